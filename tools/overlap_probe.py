"""Do an HBM-bound elementwise kernel and the MFMA-bound Winograd weight-gradient overlap when issued on two HIP streams?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adyolo_amd
from adyolo_amd import ops
n, h, w, c = 64, 600, 16, 256
x = torch.randn(n, h, w, c, device="cuda:0")
dy = torch.randn(n, h, w, c, device="cuda:0")
r = torch.randn(n, h, w, c, device="cuda:0")
sc, sh = torch.rand(c, device="cuda:0") + 0.5, torch.randn(c, device="cuda:0")
s = torch.rand(n, c, device="cuda:0")
wt = torch.randn(c, c, 3, 3, device="cuda:0") * 0.05
wpk, _ = ops.pack_w3x3(wt, c, want_dgrad=False)
s2 = torch.cuda.Stream()
def ew(k):
    for _ in range(k): ops.se_tail_fwd(x, r, sc, sh, s)
def wg(k):
    for _ in range(k): ops.conv3x3_wgrad(x, dy, c)
def fw(k):
    for _ in range(k): ops.conv3x3(x, wpk, c)
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
ew(2); wg(2); fw(2)
K = 10
t_ew, t_wg, t_fw = timed(lambda: ew(K * 4)), timed(lambda: wg(K)), timed(lambda: fw(K))
def both(a, ka, b, kb):
    def run():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            b(kb)
        a(ka)
        torch.cuda.current_stream().wait_stream(s2)
    return run
print("alone: %d x elementwise %.2f ms, %d x wgrad %.2f ms, %d x fwd %.2f ms" % (K * 4, t_ew, K, t_wg, K, t_fw))
print("elementwise || wgrad : %.2f ms (sum %.2f, max %.2f)" % (timed(both(ew, K * 4, wg, K)), t_ew + t_wg, max(t_ew, t_wg)))
print("fwd || wgrad         : %.2f ms (sum %.2f, max %.2f)" % (timed(both(fw, K, wg, K)), t_fw + t_wg, max(t_fw, t_wg)))
