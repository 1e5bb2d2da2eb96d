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

# MIC feature set (BASELINE config 5): log-mel of the microphones (K1) + six GCC-PHAT channels (K1m)
from adyolo_amd.features import MicFeatureExtractor  # noqa: E402
from adyolo_amd import _lib  # noqa: E402
from adyolo_amd.ops import _p, _stream  # noqa: E402
mfx = MicFeatureExtractor(None, "cuda:0")
out = torch.zeros((b, n // 600, 64, 32), dtype=torch.float32, device="cuda:0")
gcc = lambda: _lib.call("adyolo_feat_gcc_phat", _p(audio), _p(None), _p(mfx.k1.twiddle), _p(mfx.gcc_mean), _p(mfx.gcc_rstd), _p(out), b, n, 32, 4, _stream())  # noqa: E731
for name, fn in (("K1m GCC-PHAT kernel alone", gcc), ("MIC feature set (K1 + K1m + assembly)", lambda: mfx(audio))):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fn()
    e.record()
    torch.cuda.synchronize()
    print("%s: %.3f ms" % (name, s.elapsed_time(e) / 5))
