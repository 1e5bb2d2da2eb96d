import os, sys
sys.path.insert(0, "/root/repo")
import torch
import adyolo_amd
from adyolo_amd import ops
dev = "cuda:0"
rows, cin, cout = 6400, 512, 512
V = torch.randn(6, rows, cin, device=dev); U = torch.randn(6, cout, cin, device=dev); M = torch.empty(6, rows, cout, device=dev)
E = torch.randn(6, rows, cout, device=dev); dU = torch.empty(6, cout, cin, device=dev)
def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
fwd = lambda: ops.gemm_batched(V, U, M, rows, cout, cin, cin, cin, cout, False, False, 6, 1, rows * cin, 0, cout * cin, 0, rows * cout, 0)
wg = lambda: ops.gemm_batched(E, V, dU, cout, cin, rows, cout, cin, cin, True, True, 6, 1, rows * cout, 0, rows * cin, 0, cout * cin, 0)
print("fwd batched 6 x (6400 x 512 x 512): %.3f ms  (%.1f TFLOP/s)" % (t(fwd), 6 * 2 * rows * cin * cout / t(fwd) / 1e9))
print("wgrad batched 6 x (512 x 512, K = 6400): %.3f ms (%.1f TFLOP/s)" % (t(wg), 6 * 2 * rows * cin * cout / t(wg) / 1e9))
ref = torch.einsum("prk,pnk->prn", V.double(), U.double())
fwd(); torch.cuda.synchronize()
print("fwd err", float((M.double() - ref).abs().max() / ref.abs().max()))
# elementwise pass estimate: copy 52 MB -> 79 MB
x = torch.randn(32 * 800 * 512, device=dev); y = torch.empty(int(1.5 * x.numel()), device=dev)
print("copy-ish 52 MB read + 79 MB write: %.3f ms" % t(lambda: y[: x.numel()].copy_(x)))
