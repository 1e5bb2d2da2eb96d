#!/usr/bin/env python3
"""Stem weight gradient (8 -> 32, csrc/conv.hip stem_wgrad_kernel): float64 check at odd shapes, then timing at the bench shape.
With ADYOLO_REF_LIB=<other libadyolo_hip.so> the reference library's result is also compared BIT FOR BIT (second process).
usage: python tools/stem_wgrad_check.py [--dump file.pt]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402

DEV = "cuda:0"


def ref64(x, dy, cin_real):
    xx = x.double().permute(0, 3, 1, 2)[:, :cin_real].contiguous().requires_grad_(False)
    w = torch.zeros(dy.shape[3], cin_real, 3, 3, dtype=torch.float64, device=DEV, requires_grad=True)
    y = F.conv2d(xx, w, padding=1)
    (g,) = torch.autograd.grad(y, w, dy.double().permute(0, 3, 1, 2))
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dump", default=None)
    a = ap.parse_args()
    g = torch.Generator(device=DEV).manual_seed(3)
    outs = []
    worst = 0.0
    for (n, h, w) in [(2, 7, 5), (3, 12, 64), (2, 9, 70), (1, 33, 130), (5, 4, 63), (2, 64, 64), (3, 1, 8), (2, 2, 200)]:
        x = torch.randn(n, h, w, 8, device=DEV, generator=g)
        x[..., 7] = 0
        dy = torch.randn(n, h, w, 32, device=DEV, generator=g)
        dw = ops.conv3x3_wgrad(x, dy, 7)
        torch.cuda.synchronize()
        r = ref64(x, dy, 7)
        err = float((dw.double() - r).abs().max() / r.abs().max())
        worst = max(worst, err)
        print("stem wgrad %s: rel err vs float64 %.2e %s" % ((n, h, w), err, "OK" if err < 2e-5 else "FAIL"))
        outs.append(dw.cpu())
    x = torch.randn(64, 2400, 64, 8, device=DEV, generator=g)
    dy = torch.randn(64, 2400, 64, 32, device=DEV, generator=g)
    for _ in range(3):
        dw = ops.conv3x3_wgrad(x, dy, 7)
    torch.cuda.synchronize()
    outs.append(dw.cpu())
    ts = []
    for _ in range(10):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        ops.conv3x3_wgrad(x, dy, 7)
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    print("bench shape 64 x 2400 x 64: %.3f ms (median of 10, kernel + slab reduction), worst error %.2e" % (ts[len(ts) // 2], worst))
    if a.dump:
        torch.save(outs, a.dump)


if __name__ == "__main__":
    main()
