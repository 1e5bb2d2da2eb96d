import os, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
def run(algo, steps=40):
    os.environ["ADYOLO_CONV_ALGO"] = algo
    import importlib
    import adyolo_amd
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    torch.manual_seed(100)
    prm = bench.params("cuda:0")
    B, n = 8, 24000 * 10
    T = n // 600
    model = WrapperModel((1, 7, T, 64), (), prm).to("cuda:0")
    model.encoder.lstm.dropout = 0.0
    tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)
    audio = synthetic_audio(B, n, seed=7).to("cuda:0")
    target = synthetic_targets(B, T // 4, 12, seed=7).to("cuda:0")
    out = []
    for i in range(steps):
        out.append(float(tr.step(audio, target)))
    return out
a = run("direct"); b = run("winograd")
for i in (0, 1, 2, 5, 10, 20, 30, 39):
    print("step %2d  direct %.5f  winograd %.5f  rel diff %.2e" % (i, a[i], b[i], abs(a[i] - b[i]) / abs(a[i])))
import math
assert all(math.isfinite(v) for v in a + b) and b[-1] < b[0]
