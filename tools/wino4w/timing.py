#!/usr/bin/env python3
"""Where a workgroup of the F(4x4)-domain weight-gradient kernel spends a step: s_memtime stamps written by wave 0 of ONE
workgroup of a launch built with -DW4W_TIMING=1 (bash tools/build_variant.sh w4w_timing wino4w.hip -DW4W_TIMING=1; run with
ADYOLO_LIB=ad-yolo_amd/variants/lib_w4w_timing.so).  Stamps per step: 0 top, 1 operands of block 0 ready (window + dy reads,
H transforms), 2 block 0 issued (36 MFMAs with the staging tasks between them), 3 block 1, 4 end-of-step barrier passed."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import _lib  # noqa: E402

SHAPES = {1: (2400, 64, 32, 32), 2: (1200, 32, 64, 64), 3: (600, 16, 128, 128), 4: (600, 16, 256, 256)}
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.adyolo_w4w_timing_buffer.argtypes = [ctypes.c_void_p]
tb = torch.zeros(12 * 16, dtype=torch.int64, device="cuda:0")
assert lib.adyolo_w4w_timing_buffer(tb.data_ptr()) == 0
B = int(os.environ.get("B", "64"))
for st in (4, 2, 1):
    h, w, cin, cout = SHAPES[st]
    for aff in (0, 1):
        x = torch.randn(B, h, w, cin, device="cuda:0")
        dy = torch.randn(B, h, w, cout, device="cuda:0")
        sc, sh = torch.rand(cin, device="cuda:0") + 0.5, torch.randn(cin, device="cuda:0")
        ns = _lib.load().adyolo_wino4_wgrad_slabs(B, h, w, cin, cout)
        slabs = torch.empty(ns * 36 * cin * cout, device="cuda:0")
        du = torch.empty(36 * cin * cout, device="cuda:0")
        dw = torch.empty(cout, cin, 3, 3, device="cuda:0")
        s = torch.cuda.current_stream().cuda_stream
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(3):
            tb.zero_()
            ev0.record()
            _lib.call("adyolo_wino4_wgrad", x.data_ptr(), dy.data_ptr(), sc.data_ptr() if aff else None, sh.data_ptr() if aff else None,
                      slabs.data_ptr(), du.data_ptr(), dw.data_ptr(), B, h, w, cin, cin, cout, s)
            ev1.record()
        torch.cuda.synchronize()
        t = tb.cpu().view(12, 16).double()
        labels = ["operands (LDS reads + transforms)", "block 0 (36 MFMAs + staging)", "block 1", "end barrier"]
        print("stage %d affine=%d: launch + finish %.3f ms" % (st, aff, ev0.elapsed_time(ev1)), flush=True)
        for k in range(3, 9):
            if t[k, 0] == 0 or t[k + 1, 0] == 0:
                continue
            seq = [0, 1, 2, 3, 4]
            parts = ["%s %.0f" % (lab, float(t[k, b] - t[k, a])) for a, b, lab in zip(seq[:-1], seq[1:], labels)]
            print("  step %d: %.0f ticks (to next top %.0f) | %s" % (k, float(t[k, 4] - t[k, 0]), float(t[k + 1, 0] - t[k, 0]),
                                                                   " | ".join(parts)), flush=True)
