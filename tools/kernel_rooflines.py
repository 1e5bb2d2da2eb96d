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
stage-1 and stage-2..4 launches of `wino_fwd_kernel<...>` are separate rows.  Launches of one instantiation with one grid can
still be different layers (stage 2 and stage 4 of the ResNet launch the same 19 200 workgroups): a group is therefore split
further into duration clusters (a gap of more than 1.3x between consecutive sorted durations starts a new cluster); the
clusters of the three passes are matched in ascending order of duration.

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
    """-> {(name, grid): [(duration ns, value)]} in dispatch order"""
    rows = collections.defaultdict(list)
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter or (only and only not in r["Kernel_Name"]):
                continue
            k = (clean(r["Kernel_Name"]), int(r["Grid_Size"]))
            rows[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), float(r["Counter_Value"])))
    return rows


def clusters(pairs, gap=1.3):
    """[(duration, value)] -> list of clusters (each a list of pairs), ascending duration; new cluster at a > gap x jump"""
    ps = sorted(pairs)
    out = [[ps[0]]]
    for a, b in zip(ps, ps[1:]):
        if b[0] > gap * a[0] and len(ps) >= 8:
            out.append([])
        out[-1].append(b)
    # tiny clusters (cold first launches, profiler hiccups) are merged into their lower neighbour
    merged = [out[0]]
    for c in out[1:]:
        if len(c) < max(2, len(ps) // 20):
            merged[-1].extend(c)
        else:
            merged.append(c)
    return merged


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
    fetch = read_counter(fetch_p, "FETCH_SIZE", only)
    write = read_counter(write_p, "WRITE_SIZE", only)
    tr = read_trace(trace, only) if trace else None
    rows = []
    mean = lambda v: sum(v) / len(v)        # noqa: E731
    for k in sorted(set(fetch) & set(write)):
        cf, cw = clusters(fetch[k]), clusters(write[k])
        ct = clusters([(d, 0.0) for d in tr[k]]) if tr is not None and k in tr else None
        if len(cf) != len(cw) or (ct is not None and len(ct) != len(cf)):
            cf, cw = [sorted(fetch[k])], [sorted(write[k])]          # the passes disagree on the clustering: keep the group whole
            ct = [sorted((d, 0.0) for d in tr[k])] if ct is not None else None
        for ci in range(len(cf)):
            durs = [d for d, _ in (ct[ci] if ct is not None else cf[ci] + cw[ci])]
            ds = sorted(durs)
            med_ns = ds[len(ds) // 2]              # median: the first dispatch of a group often runs cold
            f_kib, w_kib = mean([v for _, v in cf[ci]]), mean([v for _, v in cw[ci]])
            by = (2.0 * f_kib + w_kib) * 1024.0
            gbs = by / med_ns
            tot = sum(durs) / (1 if ct is not None else 2)
            rows.append({"kernel": k[0], "grid": k[1], "cluster": "%d/%d" % (ci + 1, len(cf)), "dispatches": len(cf[ci]),
                         "median_us": round(med_ns / 1e3, 1), "fetch_MB": round(2.0 * f_kib * 1024 / 1e6, 2),
                         "write_MB": round(w_kib * 1024 / 1e6, 2), "hbm_MB_per_dispatch": round(by / 1e6, 2),
                         "GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_GBS, 4), "total_ms": round(tot / 1e6, 3)})
    rows.sort(key=lambda x: -x["total_ms"])
    over = [x for x in rows if x["frac_of_hbm_peak"] > 1.0]
    doc = {"method": "per (full template name, grid): (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes / median dispatch duration "
                     "(%s) vs 8 TB/s" % ("kernel trace of the --stats pass" if trace else "timestamps of the two counter passes"),
           "rows_above_peak": len(over), "kernels": rows}
    json.dump(doc, open(out, "w"), indent=1)
    for x in rows[:45]:
        print("%-48s grid %9d %3s x%4d %8.1f us  rd %8.1f wr %8.1f MB %7.1f GB/s %5.1f%%  (%.1f ms)" %
              (x["kernel"][:48], x["grid"], x["cluster"], x["dispatches"], x["median_us"], x["fetch_MB"], x["write_MB"], x["GBps"],
               100 * x["frac_of_hbm_peak"], x["total_ms"]))
    if over:
        print("WARNING: %d rows above the HBM peak (counter passes saw different work?)" % len(over))


if __name__ == "__main__":
    main()
