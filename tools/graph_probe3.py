"""Stress the evaluation-forward graph: which side (eager or replay) is wrong when they differ?  (GPU box)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import adyolo_amd  # noqa: F401
import bench
from adyolo_amd.wrapper import WrapperModel
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.postprocess import LabelPostProcessor
from adyolo_amd.graph import ForwardGraphs
from adyolo_amd.datasets import synthetic_audio

prm = bench.params("cuda:0")
prm["train_config"].update(conf_thresh=0.5, clss_thresh=0.5, unify_thresh=15.0, nms="conn-merge")
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    torch.manual_seed(100)
    model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
    model.eval()
    fx = FeatureExtractor(None, "cuda:0")
    post = LabelPostProcessor(prm)
    fg = ForwardGraphs(model, fx, post)
    for seconds in (2, 3):
        for i in range(4):
            audio = synthetic_audio(1, 24000 * seconds, seed=100 + 10 * seconds + i).to("cuda:0")
            with torch.no_grad():
                ref = model(fx(audio, channels_last8=True), channels_last8=True)
                ref2 = model(fx(audio, channels_last8=True), channels_last8=True)
            out, dec = fg(audio)
            out = out.clone()
            with torch.no_grad():
                ref3 = model(fx(audio, channels_last8=True), channels_last8=True)
            e12, e13, eo1, eo3 = (bool(torch.equal(a, b)) for a, b in ((ref, ref2), (ref, ref3), (out, ref), (out, ref3)))
            if not (e12 and e13 and eo1):
                bad += 1
                print("rep %d clip %ds call %d: ref==ref2 %s ref==ref3 %s out==ref %s out==ref3 %s  |out-ref| %.3e |ref-ref3| %.3e"
                      % (rep, seconds, i, e12, e13, eo1, eo3, float((out - ref).abs().max()), float((ref - ref3).abs().max())))
print("mismatches:", bad)
