#!/usr/bin/env python3
"""Scan a hipcc -S listing for vector-memory loads that are waited for almost immediately: per kernel, the loads whose destination
register is covered by an s_waitcnt vmcnt(N) fewer than `near` instructions later with at most N younger loads in between (i.e.
the wait really is for them).  Such a pair in a hot loop is an exposed memory latency -- the pattern behind round 6's sunk
leftover-row requests.  usage: isa_load_wait_scan.py file.s [near=12] [kernel substring]"""
import re
import sys

path = sys.argv[1]
near = int(sys.argv[2]) if len(sys.argv) > 2 else 12
sub = sys.argv[3] if len(sys.argv) > 3 else ""
kern, out = None, {}
window = []                                   # (index, is_load)
idx = 0
for line in open(path):
    s = line.strip()
    m = re.match(r"^(_Z\w+):", s)
    if m:
        kern, window, idx = m.group(1), [], 0
        continue
    if kern is None or not s or s[0] in ";." or s.startswith(".L"):
        if s.startswith(".Lfunc_end"):
            kern = None
        continue
    idx += 1
    op = s.split()[0]
    if op.startswith(("buffer_load", "global_load")):
        window.append(idx)
    elif op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", s)
        if m and window:
            n = int(m.group(1))
            waited = window[:len(window) - n] if n < len(window) else []
            hits = [i for i in waited if idx - i <= near]
            if hits and sub in kern:
                out.setdefault(kern, []).append((idx, len(hits), idx - hits[-1]))
            window = window[len(window) - n:] if n < len(window) else window
for k, v in out.items():
    short = re.sub(r"EEvPKf.*", "", k).replace("_ZN6adyolo2w4", "").replace("_ZN6adyolo", "")
    print("%-60s %3d places: %s" % (short[:60], len(v), " ".join("@%d(%d loads, %d instr)" % t for t in v[:12])))
