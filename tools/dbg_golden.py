import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
import adyolo_amd
from oracle.filler import fill_module_, fill_state_dict
from oracle import seresnet as onet
from adyolo_amd.wrapper import WrapperModel
sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_kernels import _params
g = np.load(os.path.join(R, "tests", "golden", "encoder.npz"))
x = torch.from_numpy(g["x"])
enc64, _ = onet.split_state_dict(fill_state_dict(onet.state_dict_spec()))
enc64 = {k: (v.double() if v.is_floating_point() else v) for k, v in enc64.items()}
for k, v in enc64.items():
    if v.is_floating_point() and "running" not in k:
        v.requires_grad_(True)
y64 = onet.encoder_forward(enc64, x.double(), training=True)
(y64 * torch.from_numpy(g["probe"]).double()).sum().backward()
def run(fa, wa):
    os.environ["ADYOLO_CONV_ALGO"] = fa
    os.environ["ADYOLO_WGRAD_ALGO"] = wa
    model = WrapperModel((1, 7, 64, 64), (), _params())
    fill_module_(model)
    model = model.to("cuda:0")
    model.train(); model.encoder.lstm.dropout = 0.0
    y = model.encoder(x.cuda())
    (y * torch.from_numpy(g["probe"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    named = dict(model.encoder.named_parameters())
    rows = []
    for key in g.files:
        if key.startswith("grad_"):
            t64 = enc64[key[5:]].grad.reshape(-1)
            got = named[key[5:]].grad.reshape(-1)[:t64.numel()].cpu()
            ref = torch.from_numpy(g[key].reshape(-1))
            am = float(t64.abs().max())
            if am < 1e-9: continue
            rows.append((float((got.double() - t64).abs().max()) / am, float((ref.double() - t64[:ref.numel()]).abs().max()) / am, key))
    rows.sort(reverse=True)
    print("== fwd/dgrad %s, wgrad %s: y err %.2e" % (fa, wa, float((y.cpu() - torch.from_numpy(g["y_train"])).abs().max())))
    for r in rows[:8]:
        print("   %.2e (ref %.2e) %s" % r)
for fa, wa in (("direct", "direct"), ("winograd", "direct"), ("direct", "winograd"), ("winograd", "winograd"), ("winograd", "winograd")):
    run(fa, wa)
