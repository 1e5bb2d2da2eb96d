"""Localise a hipGraph capture failure: capture growing prefixes of the step, each in its own child process.
usage (GPU box): python tools/graph_probe.py            (prints one line per stage: ok / rc)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = ["features", "forward", "loss", "backward", "full", "trainer"]


def child(stage):
    sys.path.insert(0, ROOT)
    import faulthandler
    faulthandler.enable()
    import torch
    import adyolo_amd  # noqa: F401
    import bench
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    torch.manual_seed(100)
    prm = bench.params("cuda:0")
    model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
    crit = WrapperCriterion(prm)
    fx = FeatureExtractor(None, "cuda:0")
    audio = synthetic_audio(2, 48000, seed=1).to("cuda:0")
    target = synthetic_targets(2, 20, 12, seed=1).to("cuda:0")
    tr = TrainStep(model, crit, fx, prm, graph=(stage == "trainer"))
    if stage == "trainer":
        for i in range(4):
            print("step", i, float(tr.step(audio, target)), flush=True)
        return
    tr.step_eager(audio, target)
    torch.cuda.synchronize()
    model.train()

    def body():
        feat = fx(audio, channels_last8=True)
        if stage == "features":
            return feat
        out = model(feat, channels_last8=True)
        if stage == "forward":
            return out
        tr.optimizer.zero_grad()
        loss = crit(out, target)
        if stage == "loss":
            return loss
        loss.backward()
        if stage == "backward":
            return loss
        tr.optimizer.step()
        return loss
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        res = body()
    print("captured", stage, flush=True)
    g.replay()
    torch.cuda.synchronize()
    print("replayed", stage, float(res.reshape(-1)[0]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for st in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), st], capture_output=True, text=True)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            print("%-9s rc=%d  %s" % (st, r.returncode, tail), flush=True)
            if r.returncode != 0:
                print("\n".join(r.stderr.strip().splitlines()[-25:]))
