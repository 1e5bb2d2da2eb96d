"""First-step gradients of the whole model under three arithmetics (Winograd fp32, direct fp32, Winograd bf16x3): relative L2
differences, to tell ReLU-mask switches (present between ANY two arithmetics) from a systematic error.  GPU box only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adyolo_amd  # noqa: E402,F401
import bench  # noqa: E402
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion  # noqa: E402
from adyolo_amd.features import FeatureExtractor  # noqa: E402
from adyolo_amd.datasets import synthetic_audio, synthetic_targets  # noqa: E402
from adyolo_amd.train import TrainStep  # noqa: E402

b, n = 4, 24000 * 8
audio = synthetic_audio(b, n, seed=77).to("cuda:0")
target = synthetic_targets(b, n // 2400, 12, seed=77).to("cuda:0")
g = {}
for name, algo, math in (("wino_f32", "winograd", "f32"), ("direct_f32", "direct", "f32"), ("wino_bf16x3", "winograd", "bf16x3")):
    os.environ["ADYOLO_CONV_ALGO"], os.environ["ADYOLO_MATH"] = algo, math
    torch.manual_seed(100)
    prm = bench.params("cuda:0")
    model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
    tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=False)
    loss = float(tr.step(audio, target))
    g[name] = (tr.flat.flat_grad.double().clone(), loss)
    del tr, model
for a, c in (("wino_f32", "direct_f32"), ("wino_f32", "wino_bf16x3"), ("direct_f32", "wino_bf16x3")):
    ga, gc = g[a][0], g[c][0]
    print("%-12s vs %-12s  loss %.7f / %.7f   grad rel L2 %.3e   max abs %.3e (largest gradient %.3e)"
          % (a, c, g[a][1], g[c][1], float((ga - gc).norm() / ga.norm()), float((ga - gc).abs().max()), float(ga.abs().max())))
