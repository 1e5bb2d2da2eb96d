#!/usr/bin/env python3
"""Narrow maps (W = 4 / 8, the ResNet-Conformer's middle stages) on the persistent F(4x4) kernel with patches one / two tiles wide
(adyolo_wino4_fwd, ADYOLO_W4_NARROW): check against a float64 convolution, then time against the 16-pixel-wide patch
(ADYOLO_W4_NARROW=0) and the implicit-GEMM convolution.  usage: python tools/wino4/narrow_check.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("ADYOLO_W4_MIN_K", "32")
os.environ["ADYOLO_W4_MIN_WGS"] = "1"
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402

DEV = "cuda:0"
ops.reload_thresholds()


def run(x, wt, cin, cout, dgrad=False):
    wpk, wpkd = ops.pack_w3x3(wt, cin, want_dgrad=True, algo="winograd4")
    return ops.conv3x3(x, wpkd if dgrad else wpk, cin if dgrad else cout)


def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


ok = True
for (n, h, w, cin, cout) in [(2, 70, 4, 64, 64), (3, 133, 8, 64, 128), (1, 128, 4, 128, 64), (2, 9, 8, 64, 64), (5, 800, 4, 128, 128), (4, 800, 8, 64, 64),
                             (2, 260, 3, 64, 64), (2, 100, 7, 64, 64)]:
    g = torch.Generator(device=DEV).manual_seed(h + w + cin)
    x = torch.randn(n, h, w, cin, device=DEV, generator=g)
    wt = torch.randn(cout, cin, 3, 3, device=DEV, generator=g) / (3 * cin ** 0.5)
    y = run(x, wt, cin, cout)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), wt.double(), None, padding=1).permute(0, 2, 3, 1)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    form = ops._lib.load().adyolo_wino4_last_form()
    good = err < 2e-5 and form == 2
    ok = ok and good
    print("N=%d %dx%d %d->%d: %.2e of absmax, persistent=%s %s" % (n, h, w, cin, cout, err, form == 2, "" if good else "<-- FAIL"), flush=True)
print("CHECK %s" % ("OK" if ok else "FAIL"))
for (n, h, w, cin, cout) in [(32, 800, 4, 128, 128), (32, 800, 8, 64, 64)]:
    x = torch.randn(n, h, w, cin, device=DEV)
    wt = torch.randn(cout, cin, 3, 3, device=DEV) / (3 * cin ** 0.5)
    wpk, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo="winograd4")
    tn = t(lambda: ops.conv3x3(x, wpk, cout))
    os.environ["ADYOLO_W4_NARROW"] = "0"
    ops.reload_thresholds()
    tw = t(lambda: ops.conv3x3(x, wpk, cout))
    del os.environ["ADYOLO_W4_NARROW"]
    ops.reload_thresholds()
    wk = ops.pack_wk(wt)
    tg = t(lambda: ops.conv_gemm(0, x, wk, n, h, w, cin, cout, 3, 3, 1, 1, 1, 1))
    fl = 2.0 * n * h * w * cin * cout * 9
    print("N=%d %dx%d %d->%d forward: narrow patches %.3f ms (%.0f TFLOP/s algorithmic) | 16-wide patches %.3f ms | implicit GEMM %.3f ms (%.0f TFLOP/s)"
          % (n, h, w, cin, cout, tn, fl / tn / 1e9, tw, tg, fl / tg / 1e9), flush=True)
