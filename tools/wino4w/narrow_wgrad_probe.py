import os, sys
sys.path.insert(0, "/root/repo")
os.environ["ADYOLO_W4W_MIN_WORK"] = "1"
import torch
import adyolo_amd
from adyolo_amd import ops
ops.reload_thresholds()
def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for (n, h, w, cin, cout) in [(32, 800, 4, 128, 128), (32, 800, 8, 64, 64)]:
    x = torch.randn(n, h, w, cin, device="cuda:0"); dy = torch.randn(n, h, w, cout, device="cuda:0")
    print(ops.wgrad_form(cin, cout, "winograd4", (n, h, w)))
    t4 = t(lambda: ops.conv3x3_wgrad(x, dy, cin, algo="winograd4"))
    t2 = t(lambda: ops.conv3x3_wgrad(x, dy, cin, algo="winograd"))
    tg = t(lambda: ops.conv_gemm(2, x, dy, n, h, w, cin, cout, 3, 3, 1, 1, 1, 1))
    print("N=%d %dx%d %d->%d wgrad: F(4x4) domain %.3f ms | F(2x2) domain %.3f ms | implicit GEMM %.3f ms" % (n, h, w, cin, cout, t4, t2, tg))
