#!/usr/bin/env python3
"""Mapping probes for the F(4x4) kernel on tiny cases (one-hot inputs / identity filters)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
import adyolo_amd  # noqa
from adyolo_amd import ops

torch.set_printoptions(linewidth=200, precision=3, sci_mode=False)
H, W, cin, cout = int(os.environ.get("H", 16)), int(os.environ.get("W", 16)), int(os.environ.get("CIN", 32)), 64

def run(x, wt):
    wpk, _ = ops.pack_w3x3(wt.cuda(), cin, want_dgrad=False, algo="winograd4")
    y = ops.conv3x3(x.cuda().contiguous(), wpk, cout)
    torch.cuda.synchronize()
    return y.cpu()

def ref(x, wt):
    return F.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, padding=1).permute(0, 2, 3, 1).float()

# 1) identity centre tap, random x: y[..., co] = x[..., co] for co < cin
wt = torch.zeros(cout, cin, 3, 3)
for c in range(cin):
    wt[c, c, 1, 1] = 1.0
x = torch.randn(1, H, W, cin)
y = run(x, wt)
r = ref(x, wt)
print("centre-tap identity: max err %.3e  (absmax ref %.3e)" % ((y - r).abs().max(), r.abs().max()))
err = (y - r).abs().amax(dim=3)[0]
print("per-pixel max err (rows = y):")
print((err > 1e-3).int())
errc = (y - r).abs().amax(dim=(0, 1, 2))
print("per-channel max err:", errc)
# 2) one-hot pixel, all taps = 1 for one (co, ci) pair
for (py, px, c) in [(5, 6, 3), (0, 0, 0), (15, 15, 31), (9, 2, 17)]:
    wt = torch.zeros(cout, cin, 3, 3)
    wt[40, c] = torch.arange(1, 10).float().view(3, 3)
    x = torch.zeros(1, H, W, cin)
    x[0, py, px, c] = 1.0
    y = run(x, wt)
    r = ref(x, wt)
    print("one-hot (%d,%d,c%d): err %.3e" % (py, px, c, (y - r).abs().max()))
    if (y - r).abs().max() > 1e-3:
        nz = (y.abs() > 1e-4).nonzero()
        print("  nonzero outputs (first 20):", nz[:20].tolist())
        print("  expected nonzero:", (r.abs() > 1e-4).nonzero()[:12].tolist())
        print("  got   ch40 patch:\n", y[0, max(0, py - 2):py + 3, max(0, px - 2):px + 3, 40])
        print("  ref   ch40 patch:\n", r[0, max(0, py - 2):py + 3, max(0, px - 2):px + 3, 40])
