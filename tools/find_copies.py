"""Which ATen ops of one train step end in a device-to-device copy (profiling aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import adyolo_amd
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.datasets import synthetic_audio, synthetic_targets
from adyolo_amd.train import TrainStep
torch.manual_seed(100)
prm = bench.params("cuda:0")
b, n = 8, 24000 * 10
model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)
audio = synthetic_audio(b, n, seed=3).to("cuda:0")
target = synthetic_targets(b, n // 2400, 12, seed=3).to("cuda:0")
tr.step(audio, target); torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.step(audio, target)
    torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::stack", "aten::zeros", "aten::zero_", "aten::fill_", "aten::add_", "aten::mul", "aten::empty_like"):
        st = [s for s in (ev.stack or []) if "ad-yolo_amd" in s or "adyolo" in s]
        cnt[(ev.name, st[0] if st else "?", str(ev.input_shapes)[:60])] += 1
for (name, where, shp), c in cnt.most_common(40):
    print("%4d  %-18s %s  %s" % (c, name, where[-90:], shp))
