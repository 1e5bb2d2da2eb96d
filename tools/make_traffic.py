"""profiles/r01_traffic.json from two rocprofv3 --pmc passes over bench.py (FETCH_SIZE, WRITE_SIZE; separate passes).
HBM bytes per launch of the dominant kernel family = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: FETCH_SIZE/WRITE_SIZE are in
KiB and on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section).
usage: python tools/make_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <algo> <kernel-substring> <out.json>"""
import csv
import json
import os
import sys


def mean_counter(path, sub, name):
    vals = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if sub in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / max(1, len(vals)), len(vals)


def main():
    fetch_csv, write_csv, algo, sub, out = sys.argv[1:6]
    fetch, nf = mean_counter(fetch_csv, sub, "FETCH_SIZE")
    write, nw = mean_counter(write_csv, sub, "WRITE_SIZE")
    doc = {}
    if os.path.exists(out):
        with open(out) as f:
            doc = json.load(f)
    doc[algo] = {"kernel": sub, "launches_fetch_pass": nf, "launches_write_pass": nw,
                 "FETCH_SIZE_KiB_mean": round(fetch, 1), "WRITE_SIZE_KiB_mean": round(write, 1),
                 "hbm_bytes_per_launch": int((2.0 * fetch + write) * 1024.0),
                 "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1 "
                           "--no-cpu-baseline --no-extra --no-pipeline --no-parity`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE halving correction)"}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print(json.dumps(doc[algo]))


if __name__ == "__main__":
    main()
