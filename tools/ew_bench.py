#!/usr/bin/env python3
"""The big elementwise passes of the step alone, at the bench shapes (B = 64 x 60 s): HIP-event time per launch and the fraction of
8 TB/s by algorithmic bytes (two tensors read, one written).  usage: python tools/ew_bench.py [--iters 10]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402
from adyolo_amd.ops import _c, _p, _stream, NULL  # noqa: E402

SHAPES = {1: (2400, 64, 32), 2: (1200, 32, 64), 3: (600, 16, 128), 4: (600, 16, 256)}
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
for st, (h, w, c) in SHAPES.items():
    n = a.batch
    x = torch.randn(n, h, w, c, device="cuda:0")
    dy = torch.randn(n, h, w, c, device="cuda:0")
    g, m, iv = torch.rand(c, device="cuda:0") + 0.5, torch.randn(c, device="cuda:0"), torch.rand(c, device="cuda:0") + 0.5
    sdy, sdyx = torch.randn(c, device="cuda:0"), torch.randn(c, device="cuda:0")
    dx = torch.empty_like(x)
    rows = n * h * w

    def bn_apply():
        _c("adyolo_bn_bwd_apply", _p(dy), _p(x), _p(g), _p(m), _p(iv), _p(sdy), _p(sdyx), _p(dx), NULL, NULL, NULL, NULL, rows, c, 1, 1.0, _stream())
    cc, rr = torch.randn(n, h, w, c, device="cuda:0"), torch.randn(n, h, w, c, device="cuda:0")
    sc, sh = torch.rand(c, device="cuda:0") + 0.5, torch.randn(c, device="cuda:0")
    sv = torch.rand(n, c, device="cuda:0")
    eo = torch.empty_like(cc)
    from adyolo_amd import _lib
    words = _lib.load().adyolo_relu_mask_words(n, h * w, c)
    bits = torch.empty(words, dtype=torch.int64, device="cuda:0")

    def se_fwd():
        _c("adyolo_se_tail_fwd", _p(cc), _p(rr), _p(sc), _p(sh), _p(sv), NULL, NULL, _p(eo), _p(bits), n, h * w, c, _stream())
    def se_fwd_nobits():
        _c("adyolo_se_tail_fwd", _p(cc), _p(rr), _p(sc), _p(sh), _p(sv), NULL, NULL, _p(eo), NULL, n, h * w, c, _stream())
    for name, fn in (("bn_bwd_apply", bn_apply), ("se_tail_fwd", se_fwd), ("se_tail_fwd/nobits", se_fwd_nobits)):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(a.iters):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort()
        ms = ts[len(ts) // 2]
        gb = 3.0 * x.numel() * 4 / 1e9
        print("stage %d %-14s %d x %d x %d x %d (%.2f GB per tensor): %.3f ms  %.0f GB/s  %.1f %% of 8 TB/s" % (st, name, n, h, w, c, gb / 3, ms, gb / ms * 1e3, gb / ms * 1e3 / 80), flush=True)
