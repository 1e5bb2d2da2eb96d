#!/usr/bin/env python3
"""Micro-benchmark of the conv3x3 kernels at the bench workload's layer shapes (HIP events, one process).
usage: python tools/conv_bench.py [--batch 64] [--iters 5] [--which fwd,wgrad] [--stages 1,2,3,4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402

SHAPES = {0: (2400, 64, 8, 32), 1: (2400, 64, 32, 32), 2: (1200, 32, 64, 64), 3: (600, 16, 128, 128),
          4: (600, 16, 256, 256), 12: (1200, 32, 32, 64), 23: (600, 16, 64, 128), 34: (600, 16, 128, 256)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--which", default="fwd,wgrad")
    ap.add_argument("--stages", default="1,2,3,4")
    ap.add_argument("--algo", default=None, help="direct | winograd (default: ADYOLO_CONV_ALGO)")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (DVFS check: same cycles, less power)")
    ap.add_argument("--fused", action="store_true", help="fwd: in_affine + epilogue stats + masked addend; wgrad: in_affine")
    a = ap.parse_args()
    for st in [int(s) for s in a.stages.split(",")]:
        h, w, cin, cout = SHAPES[st]
        x = torch.randn(a.batch, h, w, cin, device="cuda:0")
        wt = torch.randn(cout, max(cin, 1) if cin != 8 else 7, 3, 3, device="cuda:0") * 0.05
        wpk, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo=a.algo)
        dy = torch.randn(a.batch, h, w, cout, device="cuda:0")
        if a.zeros:
            x.zero_(); wt.zero_(); dy.zero_(); wpk.zero_()
        flops = 2.0 * a.batch * h * w * cout * 9 * cin
        aff = (torch.rand(cin, device="cuda:0") + 0.5, torch.randn(cin, device="cuda:0")) if (a.fused and cin != 8) else None
        for which in a.which.split(","):
            if which == "fwd":
                if a.fused:
                    fn = lambda: ops.conv3x3(x, wpk, cout, relu=True, in_affine=aff, want_stats=True, addend=dy, addend_mask=dy)  # noqa: E731
                else:
                    fn = lambda: ops.conv3x3(x, wpk, cout)  # noqa: E731
            else:
                fn = lambda: ops.conv3x3_wgrad(x, dy, wt.shape[1], in_affine=aff, algo=a.algo)  # noqa: E731
            fn()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / a.iters
            print("stage %2d %-5s B=%d %dx%d %d->%d : %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)"
                  % (st, which, a.batch, h, w, cin, cout, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573), flush=True)


if __name__ == "__main__":
    main()
