"""fp32-MFMA GEMM micro-benchmark on the shapes the models use.  usage: python tools/gemm_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adyolo_amd
from adyolo_amd import ops
SHAPES = [("head fwd", 38400, 2400, 256, False, False, 1), ("gru proj", 38400, 768, 256, False, False, 1),
          ("head dX", 38400, 256, 2400, False, True, 1), ("head dW", 2400, 256, 38400, True, True, 9),
          ("down1x1 s2", 2457600, 64, 32, False, False, 1), ("down1x1 s4", 614400, 256, 128, False, False, 1),
          ("down dW s4", 256, 128, 614400, True, True, 64), ("ffn conformer", 6400, 2048, 512, False, False, 1),
          ("ffn dW", 2048, 512, 6400, True, True, 1), ("attn proj", 6400, 512, 512, False, False, 1),
          # ResNet-Conformer (config 4, B = 32 x 20 s: 25600 rows, d_model 256, FFN 1024)
          ("c4 ffn1", 25600, 1024, 256, False, False, 1), ("c4 ffn2", 25600, 256, 1024, False, False, 1),
          ("c4 ffn1 dX", 25600, 256, 1024, False, True, 1), ("c4 ffn2 dX", 25600, 1024, 256, False, True, 1),
          ("c4 ffn1 dW", 1024, 256, 25600, True, True, 16), ("c4 ffn2 dW", 256, 1024, 25600, True, True, 16),
          ("c4 proj", 25600, 256, 256, False, False, 1), ("c4 proj dW", 256, 256, 25600, True, True, 50)]
for name, m, n, k, ta, tb, splits in SHAPES:
    a = torch.randn((k, m) if ta else (m, k), device="cuda:0")
    b = torch.randn((k, n) if tb else (n, k), device="cuda:0")
    lda, ldb = a.shape[1], b.shape[1]
    f = lambda: ops.gemm(a, b, m, n, k, lda, ldb, trans_a=ta, trans_b=tb, splits=splits)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): f()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print("%-16s M=%-8d N=%-5d K=%-7d %s%s splits=%-3d %.3f ms  %.1f TFLOP/s" % (name, m, n, k, "T" if ta else "N", "T" if tb else "N", splits, ms, 2.0 * m * n * k / ms / 1e9))
