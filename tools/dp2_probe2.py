"""Log the notification / launch sequence of the bucketed all-reduce on the real model (1 process, forced hooks, gloo)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", ADYOLO_FORCE_DP_HOOKS="1")
import torch
import adyolo_amd, bench
from adyolo_amd import dist as adist
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.datasets import synthetic_audio, synthetic_targets
from adyolo_amd.train import TrainStep
adist.init_from_env("gloo")
torch.manual_seed(100)
prm = bench.params("cuda:0")
b, n = 2, 24000 * 4
model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
model.encoder.lstm.dropout = 0.0
tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)
names = {id(p): k for k, p in model.named_parameters()}
red = tr.reducer
log = []
orig_launch = red._launch
def launch(bk):
    log.append("LAUNCH bucket %d" % bk)
    orig_launch(bk)
red._launch = launch
orig_notify = red.notify
def notify(idx):
    log.append("sink  %-44s bucket %d pending-before %d" % (names[id(tr.flat.params[idx])], red._bucket_of[idx], red._pending[red._bucket_of[idx]]))
    orig_notify(idx)
red.notify = notify
for h in red._hooks:
    h.remove()
red._hooks = []
def mk(i):
    def hook(_p):
        log.append("hook  %-44s bucket %d pending-before %d" % (names[id(tr.flat.params[i])], red._bucket_of[i], red._pending[red._bucket_of[i]]))
        bk = red._bucket_of[i]
        red._pending[bk] -= 1
        if red._pending[bk] == 0:
            red.fired_from_hooks += 1
            red._launch(bk)
    return hook
for i, p in enumerate(tr.flat.params):
    red._hooks.append(p.register_post_accumulate_grad_hook(mk(i)))
audio = synthetic_audio(b, n, seed=30).to("cuda:0")
target = synthetic_targets(b, n // 2400, 12, seed=40).to("cuda:0")
tr.step(audio, target)
torch.cuda.synchronize()
from collections import Counter
cnt = Counter(l.split()[1] for l in log if not l.startswith("LAUNCH"))
print("notifications:", len([l for l in log if not l.startswith("LAUNCH")]), "params:", len(tr.flat.params), "dups:", [k for k, v in cnt.items() if v > 1][:20])
for l in log:
    if l.startswith("LAUNCH") or "pending-before 1" in l.split("bucket")[1][2:] or "layer3.5" in l or "layer3.4.se.fc.2.bias" in l or "layer4.0" in l:
        print(l)
