#!/usr/bin/env python3
"""Which torch-side launches (aten copies / adds / fills / cats: `__amd_rocclr_copyBuffer`, `at::native::*elementwise*`,
`CatArrayBatchedCopy`, `multi_tensor_apply`) does one eager train step make, and from where?  One step under torch.profiler with
Python stacks, aggregated by (aten op, innermost repo frame).  usage: python tools/torch_launch_sources.py [--encoder se-resnet34]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import adyolo_amd  # noqa: F401,E402
import bench  # noqa: E402
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion  # noqa: E402
from adyolo_amd.features import FeatureExtractor  # noqa: E402
from adyolo_amd.datasets import synthetic_audio, synthetic_targets  # noqa: E402
from adyolo_amd.train import TrainStep  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--encoder", default="se-resnet34")
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--seconds", type=int, default=20)
a = ap.parse_args()
n = 24000 * a.seconds
prm = bench.params("cuda:0")
prm["args"]["encoder"] = a.encoder
torch.manual_seed(100)
model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=False)
audio = synthetic_audio(a.batch, n, seed=1).to("cuda:0")
target = synthetic_targets(a.batch, n // 2400, 12, seed=1).to("cuda:0")
for _ in range(3):
    tr.step(audio, target)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(audio, target)
    torch.cuda.synchronize()
agg = collections.Counter()
kern = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA or not ev.kernels:
        continue
    # innermost aten op that owns the kernels: skip parents whose children own them (kernels are attached to the launching op)
    name = ev.name
    if not name.startswith("aten::") and not name.startswith("autograd::") and "Backward" not in name:
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "/ad-yolo_amd/" in fr or "/adyolo_amd" in fr or "bench.py" in fr:
            site = fr.split("/root/repo/")[-1] if "/root/repo/" in fr else fr
            break
    for k in ev.kernels:
        agg[(name, site, k.name[:60])] += 1
        kern[k.name[:60]] += 1
# copies (hipMemcpyAsync D2D shows up as __amd_rocclr_copyBuffer in a kernel trace, not as a kernel here): every aten::copy_ /
# clone / contiguous / cat / stack / fill_ / zero_ / add_ call with the chain of its enclosing ops
cops = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        continue
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::cat", "aten::_foreach_add_", "aten::mul", "aten::sum"):
        chain, q = [], ev.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name)
            q = q.cpu_parent
        site = "?"
        for fr in (ev.stack or []):
            if "/root/repo/" in fr and "torch_launch_sources" not in fr:
                site = fr.split("/root/repo/")[-1]
                break
        cops[(ev.name, " < ".join(chain), site)] += 1
print("host-side aten calls that launch copies / fills / adds, with their enclosing ops:")
for (name, chain, site), c in sorted(cops.items(), key=lambda kv: -kv[1]):
    print("  %4d  %-18s %-80s %s" % (c, name, chain[:80], site))
print("torch-side kernels of ONE eager step (%s, %d x %d s): %d launches" % (a.encoder, a.batch, a.seconds, sum(kern.values())))
for k, c in kern.most_common():
    print("  %4d  %s" % (c, k))
print("by (aten op, first repo frame on the stack, kernel):")
for (name, site, k), c in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("  %4d  %-28s %-70s %s" % (c, name, site, k))
