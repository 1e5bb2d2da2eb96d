#!/usr/bin/env python3
"""Stem forward (8 -> 32, bias + ReLU + per-patch sums; csrc/conv.hip stem_fwd_kernel): float64 check, a bit dump for A/B runs of two
libraries (ADYOLO_LIB), and the time at the bench shape.  usage: python tools/stem_fwd_check.py [--dump file.pt]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402

DEV = "cuda:0"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump", default=None)
    a = ap.parse_args()
    g = torch.Generator(device=DEV).manual_seed(11)
    outs = []
    for (n, h, w) in [(2, 9, 40), (3, 17, 64), (1, 64, 70)]:
        x = torch.randn(n, h, w, 8, device=DEV, generator=g)
        x[..., 7] = 0
        wt = torch.randn(32, 7, 3, 3, device=DEV, generator=g) * 0.2
        b = torch.randn(32, device=DEV, generator=g)
        wpk, _ = ops.pack_w3x3(wt, 8, want_dgrad=False, algo="direct")
        y, st = ops.conv3x3(x, wpk, 32, bias=b, relu=True, want_stats=True)
        ref = F.relu(F.conv2d(x[..., :7].permute(0, 3, 1, 2).double(), wt.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
        err = float((y.double() - ref).abs().max() / ref.abs().max())
        es = float((st[0].sum(0).double() - ref.sum((0, 1, 2))).abs().max() / ref.sum((0, 1, 2)).abs().max())
        print("stem fwd %s: y %.2e  sums %.2e %s" % ((n, h, w), err, es, "OK" if max(err, es) < 2e-5 else "FAIL"))
        outs += [y.cpu(), st.cpu()]
    x = torch.randn(64, 2400, 64, 8, device=DEV, generator=g)
    wt = torch.randn(32, 7, 3, 3, device=DEV, generator=g) * 0.2
    b = torch.randn(32, device=DEV, generator=g)
    wpk, _ = ops.pack_w3x3(wt, 8, want_dgrad=False, algo="direct")
    for _ in range(3):
        y, st = ops.conv3x3(x, wpk, 32, bias=b, relu=True, want_stats=True)
    torch.cuda.synchronize()
    outs += [y[:2].cpu(), st.cpu()]
    ts = []
    for _ in range(10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.conv3x3(x, wpk, 32, bias=b, relu=True, want_stats=True)
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    print("bench shape 64 x 2400 x 64: %.3f ms (median of 10)" % ts[len(ts) // 2])
    if a.dump:
        torch.save(outs, a.dump)


if __name__ == "__main__":
    main()
