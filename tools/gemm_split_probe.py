"""Does split-K recover the tail of fractional workgroup rounds on the forward / data-gradient GEMMs of config 4?
(M = 25 600 rows, N = 256: 800 workgroups of 128 x 64 on 512 resident slots = 1.56 rounds.)  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adyolo_amd  # noqa: F401
from adyolo_amd import ops

SHAPES = [("proj 256->256", 25600, 256, 256, False, False), ("ffn2 1024->256", 25600, 256, 1024, False, False),
          ("ffn1 dX", 25600, 256, 1024, False, True), ("ffn1 256->1024", 25600, 1024, 256, False, False),
          ("pw 256->512", 25600, 512, 256, False, False), ("head 256->2400 (B=64x60s)", 38400, 2400, 256, False, False),
          ("gru proj 256->768", 38400, 768, 256, False, False)]
for name, m, n, k, ta, tb in SHAPES:
    a = torch.randn((k, m) if ta else (m, k), device="cuda:0")
    b = torch.randn((k, n) if tb else (n, k), device="cuda:0")
    lda, ldb = a.shape[1], b.shape[1]
    res = []
    for splits in (1, 2, 4):
        f = lambda: ops.gemm(a, b, m, n, k, lda, ldb, trans_a=ta, trans_b=tb, splits=splits)  # noqa: E731
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            f()
        e.record()
        torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 10)
    wgs = ((m + 127) // 128) * ((n + 63) // 64)
    print("%-28s %5d workgroups (%.2f rounds)   splits 1 / 2 / 4: %.3f / %.3f / %.3f ms   %.0f TFLOP/s unsplit"
          % (name, wgs, wgs / 512.0, res[0], res[1], res[2], 2.0 * m * n * k / res[0] / 1e9))
