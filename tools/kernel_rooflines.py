"""Per-kernel HBM rooflines keyed on the FULL demangled kernel name (template arguments kept) AND the launch grid, from
rocprofv3 passes over one command:
    --pmc FETCH_SIZE --kernel-trace     ->  <fetch counter_collection.csv>
    --pmc WRITE_SIZE --kernel-trace     ->  <write counter_collection.csv>
    [--kernel-trace --stats             ->  <kernel_trace.csv> for the durations; default: the dispatch timestamps the
                                            counter passes carry themselves]
bytes per dispatch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports
half the bytes of wide coalesced reads -- MI355X_MICROARCH.md, HBM section), GB/s against 8 TB/s.

Round 2's tools/stage_rooflines.py aggregated by the STRIPPED base name: all template instantiations of a kernel shared
one byte count (a fraction above 1 came out of that).  Here a group is (name with template arguments, grid size): the
stage-1 and stage-2..4 launches of `wino_fwd_kernel<...>` are separate rows, and no row can mix shapes.

usage: python tools/kernel_rooflines.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>
                                        [--trace kernel_trace.csv] [--only substring]
"""
import collections
import csv
import json
import re
import sys

PEAK_GBS = 8000.0


def clean(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)               # argument list of the demangled signature
    return name.replace("adyolo::", "").strip()


def read_counter(path, counter, only):
    """-> {(name, grid): [values]}, {(name, grid): [durations ns]} in dispatch order"""
    vals, durs = collections.defaultdict(list), collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter or (only and only not in r["Kernel_Name"]):
                continue
            k = (clean(r["Kernel_Name"]), int(r["Grid_Size"]))
            vals[k].append(float(r["Counter_Value"]))
            durs[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return vals, durs


def read_trace(path, only):
    durs = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if only and only not in r["Kernel_Name"]:
                continue
            grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            durs[(clean(r["Kernel_Name"]), grid)].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return durs


def main():
    args = sys.argv[1:]
    trace = only = None
    if "--trace" in args:
        i = args.index("--trace")
        trace = args[i + 1]
        del args[i:i + 2]
    if "--only" in args:
        i = args.index("--only")
        only = args[i + 1]
        del args[i:i + 2]
    fetch_p, write_p, out = args[:3]
    fetch, dur_f = read_counter(fetch_p, "FETCH_SIZE", only)
    write, dur_w = read_counter(write_p, "WRITE_SIZE", only)
    durs = read_trace(trace, only) if trace else None
    rows = []
    total_ns = 0.0
    for k in sorted(set(fetch) & set(write)):
        if durs is not None and k in durs:
            d = durs[k]
        else:
            d = dur_f[k] + dur_w[k]
        mean = lambda v: sum(v) / len(v)        # noqa: E731
        # the first dispatch of a group often runs cold (and the warm-up step of the bench differs from no other): use the median duration
        ds = sorted(d)
        avg_ns = ds[len(ds) // 2]
        by = (2.0 * mean(fetch[k]) + mean(write[k])) * 1024.0
        gbs = by / avg_ns
        total_ns += sum(d) / (2 if durs is None else 1)
        rows.append({"kernel": k[0], "grid": k[1], "dispatches": len(fetch[k]), "median_us": round(avg_ns / 1e3, 1),
                     "fetch_MB": round(2.0 * mean(fetch[k]) * 1024 / 1e6, 2), "write_MB": round(mean(write[k]) * 1024 / 1e6, 2),
                     "hbm_MB_per_dispatch": round(by / 1e6, 2), "GBps": round(gbs, 1),
                     "frac_of_hbm_peak": round(gbs / PEAK_GBS, 4),
                     "total_ms": round(sum(d) / (2 if durs is None else 1) / 1e6, 3)})
    rows.sort(key=lambda x: -x["total_ms"])
    over = [x for x in rows if x["frac_of_hbm_peak"] > 1.0]
    doc = {"method": "per (full template name, grid): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes / median dispatch duration "
                     "(%s) vs 8 TB/s" % ("kernel trace of the --stats pass" if trace else "timestamps of the two counter passes"),
           "rows_above_peak": len(over), "kernels": rows}
    json.dump(doc, open(out, "w"), indent=1)
    for x in rows[:45]:
        print("%-52s grid %9d x%4d %8.1f us  rd %8.1f wr %8.1f MB %7.1f GB/s %5.1f%%  (%.1f ms)" %
              (x["kernel"][:52], x["grid"], x["dispatches"], x["median_us"], x["fetch_MB"], x["write_MB"], x["GBps"],
               100 * x["frac_of_hbm_peak"], x["total_ms"]))
    if over:
        print("WARNING: %d rows above the HBM peak (counter passes saw different work?)" % len(over))


if __name__ == "__main__":
    main()
