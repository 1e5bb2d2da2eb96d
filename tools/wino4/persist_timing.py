#!/usr/bin/env python3
"""Where a persistent F(4x4) workgroup spends a patch: s_memtime stamps written by ONE workgroup of a plain launch built with
-DW4P_TIMING=1 (tools/build_variant.sh w4p_timing wino4p_e0.hip -DW4P_TIMING=1; run with ADYOLO_LIB=ad-yolo_amd/variants/lib_w4p_timing.so).
Stamps per patch: 0 pair loop starts, 1 pair loop done, 2-5 writer half of epilogue round 0-3 done, 6-9 reader half done,
10 epilogue done (statistics included), 11 first A fragments of the next patch ready."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("ADYOLO_W4_MIN_K", "32")
import torch  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops, _lib  # noqa: E402

SHAPES = {1: (2400, 64, 32, 32), 2: (1200, 32, 64, 64), 3: (600, 16, 128, 128), 4: (600, 16, 256, 256)}
for st in (1, 2, 3, 4):
    h, w, cin, cout = SHAPES[st]
    x = torch.randn(64, h, w, cin, device="cuda:0")
    wt = torch.randn(cout, cin, 3, 3, device="cuda:0") * 0.05
    wpk, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo="winograd4")
    y = torch.empty(64, h, w, cout, device="cuda:0")
    tb = torch.zeros(12 * 16, dtype=torch.int64, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        _lib.call("adyolo_wino4_fwd", x.data_ptr(), wpk.data_ptr(), None, None, None, None, None, y.data_ptr(), None, None,
                  tb.data_ptr(), None, None, 64, h, w, cin, cout, 0, 0, s)
    torch.cuda.synchronize()
    t = tb.cpu().view(12, 16).double()
    # s_memtime ticks: everything relative to the stamp before it
    for p in range(1, 5):
        if t[p, 0] == 0 or t[p + 1, 0] == 0:
            continue
        period = float(t[p + 1, 0] - t[p, 0])
        seq = [0, 1, 2, 6, 3, 7, 4, 8, 5, 9, 10, 11]
        labels = ["pair loop", "->writer0", "reader0", "writer1", "reader1", "writer2", "reader2", "writer3", "reader3", "stats/end", "A prep"]
        parts = []
        for a, b, lab in zip(seq[:-1], seq[1:], labels):
            parts.append("%s %.0f" % (lab, float(t[p, b] - t[p, a])))
        parts.append("to next loop %.0f" % float(t[p + 1, 0] - t[p, 11]))
        if t[p, 12] > 0:
            last = 9 if t[p, 9] > 0 else 7                   # reader stamp of the last round (NB = 2: 9, NB = 1: 7)
            w = 5 if t[p, 5] > 0 else 3
            parts.append("LAST ROUND: xi pass %.0f, next-patch loads issued %.0f, pixels %.0f, B reload issued %.0f" % (
                float(t[p, 12] - t[p, w]), float(t[p, 13] - t[p, 12]), float(t[p, 14] - t[p, 13]), float(t[p, last] - t[p, 14])))
        print("stage %d patch %d: period %.0f ticks | %s" % (st, p, period, " | ".join(parts)), flush=True)
        if p < 4 and t[6 + p, 0] > 0:
            q = t[6 + p]
            print("    pair loop, steps 0-6-12-18-24-30 of pair 0 | pair 1: %s | %s | pair 0 -> pair 1: %.0f" % (
                " ".join("%.0f" % float(q[i + 1] - q[i]) for i in range(5)), " ".join("%.0f" % float(q[6 + i + 1] - q[6 + i]) for i in range(5)),
                float(q[6] - q[0])), flush=True)
