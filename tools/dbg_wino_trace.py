"""Golden encoder step twice (direct, winograd); compare every conv3x3 / conv3x3_wgrad input and output between the runs."""
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
g = np.load(os.path.join(R, "tests", "golden", "encoder.npz"))
x = torch.from_numpy(g["x"])
orig_conv, orig_wg = ops.conv3x3, ops.conv3x3_wgrad
def run(algo):
    os.environ["ADYOLO_CONV_ALGO"] = algo
    rec = []
    def conv(x_, wpk, cout, **kw):
        out = orig_conv(x_, wpk, cout, **kw)
        o = out[0] if isinstance(out, tuple) else out
        rec.append(("conv %s->%d %s" % (tuple(x_.shape), cout, sorted(k for k, v in kw.items() if v is not None and v is not False)),
                    x_.cpu(), o.cpu(), kw.get("addend").cpu() if kw.get("addend") is not None else None))
        return out
    def wg(x_, dy, cin_real, **kw):
        out = orig_wg(x_, dy, cin_real, **kw)
        rec.append(("wgrad %s" % (tuple(x_.shape),), dy.cpu(), out.cpu(), x_.cpu()))
        return out
    ops.conv3x3, ops.conv3x3_wgrad = conv, wg
    model = WrapperModel((1, 7, 64, 64), (), _params())
    fill_module_(model)
    model = model.to("cuda:0")
    model.train(); model.encoder.lstm.dropout = 0.0
    y = model.encoder(x.cuda())
    (y * torch.from_numpy(g["probe"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    return rec
a, b = run("direct"), run("winograd")
rel = lambda p, q: float((p - q).abs().max() / q.abs().max())
print("relu-mask flips between the runs:", [int(((oa > 0) != (ob > 0)).sum()) for (na, ia, oa, xa), (nb, ib, ob, xb) in zip(a, b) if "'relu'" in na])
for (na, ia, oa, xa), (nb, ib, ob, xb) in list(zip(a, b))[:0]:
    extra = "" if xa is None else "  third %.2e" % rel(xb, xa)
    print("%-70s in %.2e  out %.2e%s" % (na[:70], rel(ib, ia), rel(ob, oa), extra))
