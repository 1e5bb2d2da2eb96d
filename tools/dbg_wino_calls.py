"""Run the golden encoder step with Winograd convolutions and, at every conv3x3 call, also run the direct kernel on the
same inputs; print the calls where the two differ most (relative to the output's absmax and to its rms)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import adyolo_amd
from adyolo_amd import ops
from oracle.filler import fill_module_
from adyolo_amd.wrapper import WrapperModel
sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_kernels import _params
os.environ["ADYOLO_CONV_ALGO"] = "winograd"
g = np.load(os.path.join(R, "tests", "golden", "encoder.npz"))
x = torch.from_numpy(g["x"])
orig_pack, orig_conv = ops.pack_w3x3, ops.conv3x3
twin = {}
def pack(w, cin_pad, want_dgrad=True, algo=None):
    uf, ud = orig_pack(w, cin_pad, want_dgrad, algo="winograd")
    df, dd = orig_pack(w, cin_pad, want_dgrad, algo="direct")
    twin[uf.data_ptr()] = df
    if ud is not None:
        twin[ud.data_ptr()] = dd
    return uf, ud
rows = []
flips = []
def conv(x_, wpk, cout, **kw):
    out = orig_conv(x_, wpk, cout, **kw)
    if wpk.dim() == 4:
        ref = orig_conv(x_, twin[wpk.data_ptr()], cout, **kw)
        a, b = (out[0], ref[0]) if isinstance(out, tuple) else (out, ref)
        d = (a - b).abs()
        flips.append(int(((a > 0) != (b > 0)).sum()) if kw.get("relu") else -1)
        rows.append((float(d.max() / b.abs().max()), float(d.max() / b.pow(2).mean().sqrt()), float(x_.abs().max() / x_.pow(2).mean().sqrt()),
                     tuple(x_.shape), cout, sorted(k for k, v in kw.items() if v is not None and v is not False)))
    return out
ops.pack_w3x3, ops.conv3x3 = pack, conv
model = WrapperModel((1, 7, 64, 64), (), _params())
fill_module_(model)
model = model.to("cuda:0")
model.train(); model.encoder.lstm.dropout = 0.0
y = model.encoder(x.cuda())
nf = len(rows)
(y * torch.from_numpy(g["probe"]).cuda()).sum().backward()
torch.cuda.synchronize()
print("ReLU-mask flips per relu conv call (winograd vs direct):", [f for f in flips if f >= 0])
for i, r in enumerate(rows[:0]):
    print("%s %2d  max/absmax %.2e  max/rms %.2e  in absmax/rms %.1f  x%s -> %d  %s" % (("fwd" if i < nf else "bwd"), i, *r))
