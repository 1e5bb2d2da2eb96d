import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, warnings
import adyolo_amd
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.datasets import synthetic_audio, synthetic_targets
from adyolo_amd.train import TrainStep
import bench
n = 24000 * 4
audio = synthetic_audio(4, n, seed=9).to("cuda:0")
target = synthetic_targets(4, n // 2400, 12, seed=9).to("cuda:0")
prm = bench.params("cuda:0"); prm["args"]["encoder"] = "resnet-conformer"
torch.manual_seed(100)
model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
for m in model.modules():
    if type(m).__name__ == "MultiHeadAttention":
        m.p = 0.0
tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=True)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    losses = [float(tr.step(audio, target)) for _ in range(5)]
print("losses", losses)
print("captures", tr.graphs.captures, "eager_only", len(tr.graphs.eager_only), "replays", tr.graphs.replays, [str(x.message)[:200] for x in w])
