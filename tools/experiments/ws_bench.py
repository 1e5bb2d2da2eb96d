#!/usr/bin/env python3
"""Stage-1 Winograd forward: the wave-specialised persistent kernel (adyolo_wino_fwd_ws) against wino_fwd_kernel<1, true>.
usage: python tools/ws_bench.py [--blocks 256]"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import adyolo_amd  # noqa: F401
from adyolo_amd import ops, _lib

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", default="256")
ap.add_argument("--batch", type=int, default=64)
a = ap.parse_args()
lib = _lib.load()
P, I = ctypes.c_void_p, ctypes.c_int
lib.adyolo_wino_fwd_ws.argtypes = [P, P, P, P, I, I, I, I, I, P]
lib.adyolo_wino_fwd_ws.restype = I
n, h, w, c = a.batch, 2400, 64, 32
x = torch.randn(n, h, w, c, device="cuda:0")
wt = torch.randn(c, c, 3, 3, device="cuda:0") * 0.05
b = torch.randn(c, device="cuda:0")
wpk, _ = ops.pack_w3x3(wt, c, want_dgrad=False, algo="winograd")
ref = ops.conv3x3(x, wpk, c, bias=b, relu=True)
y = torch.empty_like(ref)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(blocks):
    rc = lib.adyolo_wino_fwd_ws(x.data_ptr(), wpk.data_ptr(), b.data_ptr(), y.data_ptr(), n, h, w, 1, blocks, st)
    assert rc == 0, lib.adyolo_last_error()


def timeit(fn, it=200):
    for _ in range(100):                 # long enough for the clocks to settle
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


print("reference kernel: %.3f ms" % timeit(lambda: ops.conv3x3(x, wpk, c, bias=b, relu=True)))
for blocks in [int(v) for v in a.blocks.split(",")]:
    y.zero_()
    run(blocks)
    torch.cuda.synchronize()
    err = float((y - ref).abs().max())
    print("ws kernel, %4d workgroups: %.3f ms, max |diff| vs reference kernel %.3g (%s)" %
          (blocks, timeit(lambda: run(blocks)), err, "bit-equal" if torch.equal(y, ref) else "differs"))
