#!/usr/bin/env python3
"""K1 micro-benchmark: feature extraction at the bench workload (B=64 x 60 s), HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd.features import FeatureExtractor  # noqa: E402

b, n = int(os.environ.get("B", 64)), 24000 * int(os.environ.get("S", 60))
audio = (torch.randn(b, n, 4, device="cuda:0") * 0.1).contiguous()
fx = FeatureExtractor(None, "cuda:0")
for layout in (True, False):
    fx(audio, channels_last8=layout)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fx(audio, channels_last8=layout)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    gb = FeatureExtractor.algorithmic_bytes(b, n) / 1e9
    print("K1 layout=%s: %.3f ms  %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % ("nhwc8" if layout else "nchw7", ms, gb / ms * 1e3, gb / ms * 1e3 / 80))
