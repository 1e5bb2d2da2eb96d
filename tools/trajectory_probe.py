"""30 train steps of SE-ResNet34 + AD-YOLO on 8 x 20 s synthetic clips with each convolution algorithm: the loss trajectories must
track each other (sanity run beyond the 3-5 steps of the trajectory tests).  usage (GPU box): python tools/trajectory_probe.py"""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    import adyolo_amd  # noqa: F401
    import bench
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    n = 24000 * 20
    prm = bench.params("cuda:0")
    torch.manual_seed(100)
    model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
    tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=True)
    losses = []
    for i in range(30):
        audio = synthetic_audio(8, n, seed=100 + i % 4).to("cuda:0")
        target = synthetic_targets(8, n // 2400, 12, seed=100 + i % 4).to("cuda:0")
        losses.append(float(tr.step(audio, target)))
    print(" ".join("%.5f" % v for v in losses))
else:
    out = {}
    for algo in ("winograd4", "winograd", "direct"):
        env = dict(os.environ, ADYOLO_CONV_ALGO=algo, ADYOLO_W4_MIN_WGS="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "run"], env=env, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln and ln[0].isdigit()]
        assert line, r.stderr[-2000:]
        out[algo] = [float(v) for v in line[-1].split()]
        print("%-10s" % algo, " ".join("%.4f" % v for v in out[algo][::3]))
    for algo in ("winograd4", "winograd"):
        d = max(abs(a - b) / abs(b) for a, b in zip(out[algo], out["direct"]))
        print("max relative deviation of %s from direct over 30 steps: %.2e" % (algo, d))
