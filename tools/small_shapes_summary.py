"""gpurun_out/<tag>_<config>_kernel_stats.csv + the bench entry printed under rocprof -> gpurun_out/<tag>_small_shapes.json:
kernel_ms_per_step = total kernel time of the traced process / steps it executed (warm-ups included: every step launches
the same kernels; the model construction's few fill kernels are in the total too)."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = {}
for cfg in ("train_bs16x20s", "train_bs8x20s", "eval_bs1x60s", "eval_bs8x60s", "conformer_bs32x20s"):
    stats = os.path.join(ROOT, "gpurun_out", "%s_%s_kernel_stats.csv" % (tag, cfg))
    log = os.path.join(ROOT, "gpurun_out", "%s_%s_under_rocprof.log" % (tag, cfg))
    if not (os.path.exists(stats) and os.path.exists(log)):
        continue
    ent = None
    for line in open(log):
        if line.startswith("{"):
            ent = json.loads(line)
    if ent is None:
        continue
    steps = sum(ent.get("steps_executed", {"all": ent["steps"] + 3}).values())
    rows = list(csv.DictReader(open(stats)))
    total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    launches = sum(int(r["Calls"]) for r in rows)
    top = sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:6]
    out[cfg] = {"kernel_ms_per_step": round(total_ns / 1e6 / steps, 3), "launches_per_step": round(launches / steps, 1),
                "steps_traced": steps, "wall_ms_per_step_under_the_profiler": ent["ms_per_step"],
                "top_kernels_ms_per_step": {r["Name"][:70]: round(float(r["TotalDurationNs"]) / 1e6 / steps, 3) for r in top}}
path = os.path.join(ROOT, "gpurun_out", "%s_small_shapes.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
