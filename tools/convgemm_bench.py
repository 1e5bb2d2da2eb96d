"""Implicit-GEMM convolution (adyolo_conv_gemm) at the ResNet-Conformer's shapes (config 4: 32 clips x 800 frames): forward,
data-gradient and weight-gradient launches.  usage (GPU box): python3 tools/convgemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adyolo_amd  # noqa: E402,F401
from adyolo_amd import ops  # noqa: E402

# (name, N, H, W, Cin, Cout, KH, KW, SH, SW, PH, PW)
SHAPES = [("stage4 3x1 512->512 (F=1)", 32, 800, 1, 512, 512, 3, 1, 1, 1, 1, 0),
          ("stage3 folded 3x1 512->512 (F=2)", 32, 800, 1, 512, 512, 3, 1, 1, 1, 1, 0),
          ("stage2 3x3 128->128 (F=4)", 32, 800, 4, 128, 128, 3, 3, 1, 1, 1, 1),
          ("stage3.0 3x3 s(1,2) 128->256", 32, 800, 4, 128, 256, 3, 3, 1, 2, 1, 1),
          ("stage4.0 3x3 s(1,2) 256->512", 32, 800, 2, 256, 512, 3, 3, 1, 2, 1, 1),
          ("stem 7x7 s(1,2) 8->64", 32, 800, 64, 8, 64, 7, 7, 1, 2, 3, 3)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name, n, h, w, cin, cout, kh, kw, sh, sw, ph, pw in SHAPES:
    g = (n, h, w, cin, cout, kh, kw, sh, sw, ph, pw)
    ho, wo = ops.conv_out_hw(h, w, kh, kw, sh, sw, ph, pw)
    x = torch.randn(n, h, w, cin, device="cuda:0")
    wt = torch.randn(cout, cin, kh, kw, device="cuda:0") * 0.05
    dy = torch.randn(n, ho, wo, cout, device="cuda:0")
    wk, wkt = ops.pack_wk(wt), ops.pack_wk(wt.transpose(0, 1).contiguous())
    flops = 2.0 * n * ho * wo * cout * kh * kw * cin
    tf = timeit(lambda: ops.conv_gemm(0, x, wk, *g))
    td = timeit(lambda: ops.conv_gemm(1, dy, wkt, *g))
    tw = timeit(lambda: ops.conv_gemm(2, x, dy, *g))
    print("%-34s fwd %.3f ms (%.0f TFLOP/s)  dgrad %.3f ms (%.0f)  wgrad %.3f ms (%.0f)"
          % (name, tf, flops / tf / 1e9, td, flops / td / 1e9, tw, flops / tw / 1e9))
