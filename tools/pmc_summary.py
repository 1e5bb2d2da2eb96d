"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel (short name), per counter, mean value per dispatch.
usage: python tools/pmc_summary.py <counter_collection.csv> [name-substring]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[\d, ]+>)?)", name)
    return (m.group(1) if m else name)[:60]


def main():
    path, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if sub and sub not in r["Kernel_Name"]:
                continue
            acc[short(r["Kernel_Name"]) + "  grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        n = max(len(v) for v in cs.values())
        print("%s  (%d dispatches)" % (k, n))
        for c, v in sorted(cs.items()):
            print("    %-28s mean %.6g" % (c, sum(v) / len(v)))


if __name__ == "__main__":
    main()
