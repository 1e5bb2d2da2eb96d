"""GRU micro-benchmark at the bench shape (B = 64, T' = 600): forward and backward of one bidirectional layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adyolo_amd
from adyolo_amd import ops
b, t = 64, 600
gx = torch.randn(b, t, 2, 384, device="cuda:0") * 0.5
whh = torch.randn(2, 384, 128, device="cuda:0") * 0.08
bhh = torch.randn(2, 384, device="cuda:0") * 0.1
dout = torch.randn(b, t, 256, device="cuda:0")
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): r = f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, r
ms_f, (out, gates, hprev) = timeit(lambda: ops.gru_fwd(gx, whh, bhh, True))
ms_b, _ = timeit(lambda: ops.gru_bwd(dout, gates, hprev, whh))
print("gru fwd %.3f ms (%.0f ns/step)  bwd %.3f ms (%.0f ns/step)  checksum %.6f" % (ms_f, ms_f * 1e6 / t, ms_b, ms_b * 1e6 / t, float(out.sum())))
ms_e, _ = timeit(lambda: ops.gru_fwd(gx, whh, bhh, False))
print("gru fwd without saving gates %.3f ms (%.0f ns/step)" % (ms_e, ms_e * 1e6 / t))
for bb in (1, 8, 32):
    g1 = gx[:bb].contiguous()
    ms1, _ = timeit(lambda: ops.gru_fwd(g1, whh, bhh, True))
    print("gru fwd B=%d: %.3f ms (%.0f ns/step)" % (bb, ms1, ms1 * 1e6 / t))
