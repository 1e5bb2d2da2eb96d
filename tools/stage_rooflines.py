"""HBM roofline of every kernel of a profiled bench run: PMC bytes per dispatch ((2 x FETCH_SIZE + WRITE_SIZE) x 1024, the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md) / average dispatch time (rocprofv3 --kernel-trace --stats) vs 8 TB/s.
usage: python tools/stage_rooflines.py <kernel_stats.csv> <pmc_fetch_summary.txt> <pmc_write_summary.txt> <out.json>"""
import csv
import json
import re
import sys

PEAK = 8000.0


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^>]*>)?)", name)
    return (m.group(1) if m else name)[:60]


def read_pmc(path, counter):
    """summary text of tools/pmc_summary.py -> {short kernel name: KiB per dispatch (weighted over grids)}"""
    acc, cur, n = {}, None, 0
    for line in open(path):
        m = re.match(r"(\S.*?)\s+grid=\d+\s+\((\d+) dispatches\)", line)
        if m:
            cur, n = m.group(1), int(m.group(2))
            continue
        m = re.match(r"\s+%s\s+mean ([\d.e+]+)" % counter, line)
        if m and cur:
            s, c = acc.get(cur, (0.0, 0))
            acc[cur] = (s + float(m.group(1)) * n, c + n)
    return {k: s / c for k, (s, c) in acc.items() if c}


def main():
    stats, fetch_p, write_p, out = sys.argv[1:5]
    fetch, write = read_pmc(fetch_p, "FETCH_SIZE"), read_pmc(write_p, "WRITE_SIZE")
    rows, total = [], 0.0
    for r in csv.DictReader(open(stats)):
        total += float(r["TotalDurationNs"])
    for r in csv.DictReader(open(stats)):
        k = short(r["Name"])
        if k not in fetch or k not in write:
            k0 = k.split("<")[0]                       # the PMC summary may carry the name without template arguments
            if k0 in fetch and k0 in write:
                fetch[k], write[k] = fetch[k0], write[k0]
            else:
                continue
        by = (2.0 * fetch[k] + write[k]) * 1024.0
        avg = float(r["AverageNs"])
        gbs = by / avg
        rows.append({"kernel": k, "calls": int(r["Calls"]), "avg_us": round(avg / 1e3, 1),
                     "share_of_gpu_time": round(float(r["TotalDurationNs"]) / total, 4),
                     "hbm_MB_per_dispatch": round(by / 1e6, 2), "GBps": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK, 4)})
    rows.sort(key=lambda x: -x["share_of_gpu_time"])
    json.dump({"method": __doc__.strip().split("\n")[0], "kernels": rows}, open(out, "w"), indent=1)
    for x in rows[:30]:
        print("%-60s %6.1f us %8.2f MB %7.1f GB/s %5.1f%%  (%.1f%% of GPU time)" % (x["kernel"], x["avg_us"], x["hbm_MB_per_dispatch"], x["GBps"], 100 * x["frac_of_hbm_peak"], 100 * x["share_of_gpu_time"]))


if __name__ == "__main__":
    main()
