#!/usr/bin/env python3
"""Per-kernel time of ONE training step from a rocprofv3 rocpd database (steps are delimited by adam_kernel launches).
usage: python tools/step_profile.py <results.db> [top]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = list(db.execute("select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.start, d.end from "
                       "rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
step = rows[marks[-2] + 1:marks[-1] + 1]
print("kernels %d, sum %.3f ms, span %.3f ms" % (len(step), sum(r[5] - r[4] for r in step) / 1e6, (step[-1][5] - step[0][4]) / 1e6))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    name = r[0].replace("_ZN6adyolo", "").replace(".kd", "")[:48]
    k = (name, r[1], r[2], r[3]) if "--by-grid" in sys.argv else (name,)
    agg[k][0] += 1
    agg[k][1] += (r[5] - r[4]) / 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%9.1f us x%4d  %s" % (v[1], v[0], " ".join(str(x) for x in k)))
