#!/usr/bin/env python3
"""GPU check of the PERSISTENT F(4x4,3x3) kernel (csrc/wino4p.hpp) against the one-patch-per-workgroup kernel (csrc/wino4.hip,
ADYOLO_W4_PERSIST=0) and a float64 convolution, for the five operand combinations it is instantiated for, then a same-process
A/B timing of the two at the bench workload's shapes.
usage: python tools/wino4/persist_check.py [--skip-check] [--skip-bench] [--batch 64] [--iters 5] [--stages 2,3,4,23,34]"""
import argparse
import os
import sys

os.environ.setdefault("ADYOLO_W4_MIN_K", "32")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402

DEV = "cuda:0"


def relerr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / max(1e-30, ref.abs().max()))


def combos(n, h, w, cin, cout, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    r = lambda *s: torch.randn(*s, device=DEV, generator=g)                                            # noqa: E731
    dy, aux = r(n, h, w, cout), r(n, h, w, cout)
    aff = (torch.rand(cin, device=DEV, generator=g) + 0.5, r(cin))
    mean, invstd = r(cout) * 0.1, torch.rand(cout, device=DEV, generator=g) + 0.5
    nb = n * h * w * cout // 64
    bits1 = torch.randint(-2 ** 62, 2 ** 62, (nb,), dtype=torch.int64, device=DEV, generator=g)
    bits2 = torch.randint(-2 ** 62, 2 ** 62, (nb,), dtype=torch.int64, device=DEV, generator=g)
    return {
        "plain (0)": dict(),
        "plain + affine + relu (0)": dict(relu=True, in_affine=aff),
        "fwd conv1: affine, relu, stats (1)": dict(relu=True, in_affine=aff, want_stats=True),
        "fwd conv2: affine, stats (1)": dict(in_affine=aff, want_stats=True),
        "dgrad conv2: stats vs bn aux (9)": dict(want_stats=True, stat_bn=(aux, mean, invstd)),
        "dgrad conv1, projection after a pooled boundary: addend (2)": dict(addend=dy),
        "dgrad conv1 of the first block: addend + bits, stats vs the stem's aux (15)": dict(addend=dy, addend_mask=bits1, want_stats=True,
                                                                                           stat_bn=(aux, mean, invstd)),
        "dgrad conv1, projection: addend, stats vs aux, stat bits (27)": dict(addend=dy, want_stats=True, stat_bn=(aux, mean, invstd),
                                                                              stat_mask=bits2),
        "dgrad conv1, identity: addend + bits, stats vs aux, stat bits (31)": dict(addend=dy, addend_mask=bits1, want_stats=True,
                                                                                   stat_bn=(aux, mean, invstd), stat_mask=bits2),
    }


def run(x, wpk, cout, kw, persist):
    os.environ["ADYOLO_W4_PERSIST"] = "1" if persist else "0"
    ops.reload_thresholds()
    out = ops.conv3x3(x, wpk, cout, **kw)
    torch.cuda.synchronize()
    return out if isinstance(out, tuple) else (out, None)


def check(n, h, w, cin, cout):
    g = torch.Generator(device=DEV).manual_seed(n * 1000 + h * 10 + cin)
    x = torch.randn(n, h, w, cin, device=DEV, generator=g)
    wt = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / np.sqrt(9 * cin)
    wpk, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo="winograd4", allow32=True)
    assert wpk.shape[0] == 36
    wp2, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo="winograd")      # 32-channel output blocks: no one-patch F(4x4) kernel --
    worst = 0.0                                                              # compared with the F(2x2) kernel instead
    ref = F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, padding=1).permute(0, 2, 3, 1)
    for name, kw in combos(n, h, w, cin, cout).items():
        if (n * h * w * cout // 4) % 64 and ("stat_mask" in kw or "addend_mask" in kw):
            continue
        yp, sp = run(x, wpk, cout, kw, True)
        yo, so = run(x, wpk if cout % 64 == 0 else wp2, cout, kw, False)
        e = relerr(yp, yo)
        # (per-patch sums: the F(2x2) kernel's patches are 8 x 16 pixels, the F(4x4) kernels' 32 x 16 / 16 x 32 -- compare the totals)
        es = relerr(sp.sum(1), so.sum(1)) if sp is not None else 0.0
        e64 = relerr(yp, ref) if not kw else 0.0
        worst = max(worst, e, es, e64)
        flag = "" if max(e, es, e64) < 2e-5 else "   <-- FAIL"
        print("  %-72s y %.2e  stats %.2e%s%s" % (name, e, es, ("  vs float64 %.2e" % e64) if not kw else "", flag), flush=True)
    return worst


SHAPES = {1: (2400, 64, 32, 32), 2: (1200, 32, 64, 64), 3: (600, 16, 128, 128), 4: (600, 16, 256, 256),
          12: (1200, 32, 32, 64), 23: (600, 16, 64, 128), 34: (600, 16, 128, 256)}


def bench(batch, iters, stages, only=None):
    for st in stages:
        h, w, cin, cout = SHAPES[st]
        x = torch.randn(batch, h, w, cin, device=DEV)
        wt = torch.randn(cout, cin, 3, 3, device=DEV) * 0.05
        flops = 2.0 * batch * h * w * cout * 9 * cin
        wpk, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo="winograd4", allow32=True)
        wp2, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo="winograd")
        if wpk.shape[0] != 36:
            continue
        for name, kw in combos(batch, h, w, cin, cout).items():
            if only and only not in name:
                continue
            times = {}
            for rep in range(2):
                for persist in (False, True):
                    os.environ["ADYOLO_W4_PERSIST"] = "1" if persist else "0"
                    ops.reload_thresholds()
                    wp = wpk if (persist or cout % 64 == 0) else wp2        # (32 output channels: F(2x2) is the alternative)
                    fn = lambda: ops.conv3x3(x, wp, cout, **kw)                                        # noqa: E731
                    fn()
                    torch.cuda.synchronize()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(iters):
                        fn()
                    e.record()
                    torch.cuda.synchronize()
                    times.setdefault(persist, []).append(s.elapsed_time(e) / iters)
            one, per = min(times[False]), min(times[True])
            issued = flops * 9 / 36.0
            print("stage %2d %-70s %s %.3f ms  persistent %.3f ms (%.2f of the fp32 MFMA peak)  %.2fx" % (
                st, name, "one-patch" if cout % 64 == 0 else "F(2x2)   ", one, per, issued / per / 1e9 / 157.3, one / per), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--skip-bench", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--stages", default="2,3,4,23,34")
    ap.add_argument("--only", default=None, help="time only the operand combinations whose name contains this string")
    a = ap.parse_args()
    if not a.skip_check:
        worst = 0.0
        # (N, H, W, Cin, Cout): one patch per workgroup, several patches per workgroup (more than 32 patch slots per XCD), ragged
        # heights / widths, both tile shapes (W = 16: 4 tile columns), 1 / 2 / 4 channel blocks, 2 ... 32 pairs
        for shp in [(1, 16, 16, 32, 64), (2, 32, 16, 64, 64), (64, 160, 16, 64, 64), (64, 90, 16, 64, 128), (40, 70, 32, 64, 64),
                    (24, 67, 16, 128, 256), (3, 37, 40, 32, 64), (70, 33, 50, 32, 128), (9, 100, 64, 64, 64), (2, 600, 16, 256, 256),
                    (5, 8, 16, 512, 64), (48, 64, 32, 32, 64), (3, 40, 64, 32, 32), (40, 64, 64, 32, 32), (20, 72, 32, 64, 32),
                    # round 6 (resident U, whole-line staging, shared mask words): 32 -> 32 with a width that is not a multiple of 4 (no
                    # mask sharing) and with ragged patches in both directions; 64 -> 64 the same
                    (40, 52, 70, 32, 32), (70, 36, 52, 32, 32), (48, 44, 36, 64, 64)]:
            print("shape N=%d H=%d W=%d %d->%d" % shp, flush=True)
            worst = max(worst, check(*shp))
        print("WORST relative error %.3e %s" % (worst, "OK" if worst < 2e-5 else "FAIL"), flush=True)
    if not a.skip_bench:
        bench(a.batch, a.iters, [int(s) for s in a.stages.split(",")], a.only)


if __name__ == "__main__":
    main()
