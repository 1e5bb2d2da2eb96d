"""Which part of the evaluation forward differs when replayed from a hipGraph?  (GPU box)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import adyolo_amd  # noqa: F401
import bench
from adyolo_amd import functional as Fn
from adyolo_amd.wrapper import WrapperModel
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.datasets import synthetic_audio

torch.manual_seed(100)
prm = bench.params("cuda:0")
model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
model.eval()
fx = FeatureExtractor(None, "cuda:0")
enc = model.encoder
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def stages(audio):
    res = {}
    feat = fx(audio, channels_last8=True)
    res["feat"] = feat
    holder = Fn.BlockLink()
    y = Fn.StemFn.apply(feat, enc.conv1.weight, enc.conv1.bias, enc.bn1.weight, enc.bn1.bias, enc.bn1, False, holder)
    res["stem"] = y
    aff = holder.affine
    link = None
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(enc, "layer%d" % li)):
            nxt = Fn.BlockLink()
            y = blk(y, link_in=link, link_out=nxt, in_affine=aff, stem_holder=holder if aff is not None else None)
            aff = None
            link = nxt
            res["l%d.%d" % (li, bi)] = y
    y = Fn.SAPFn.apply(y, enc.attention.W.weight, enc.attention.W.bias)
    res["sap"] = y
    y = Fn.BiGRULayerFn.apply(y, *enc.lstm.layer_params(0), False)
    res["gru0"] = y
    y = Fn.BiGRULayerFn.apply(y, *enc.lstm.layer_params(1), False)
    res["gru1"] = y
    y = Fn.LNTanhFn.apply(y, enc.norm.weight, enc.norm.bias, enc.norm.eps)
    res["ln"] = y
    res["head"] = model.head(y)
    return res


with torch.no_grad():
    a0 = synthetic_audio(B, 48000, seed=1).to("cuda:0")
    a1 = synthetic_audio(B, 48000, seed=2).to("cuda:0")
    stages(a0)
    torch.cuda.synchronize()
    static = a0.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = stages(static)
    for trial, a in enumerate((a0, a1, a0)):
        ref = {k: v.clone() for k, v in stages(a).items()}
        static.copy_(a)
        g.replay()
        torch.cuda.synchronize()
        bad = [(k, float((outs[k] - ref[k]).abs().max())) for k in ref if not torch.equal(outs[k], ref[k])]
        print("B=%d trial %d: first differing stages:" % (B, trial), bad[:4])
