#!/usr/bin/env python3
"""Per-kernel table (FULL template name) from the counter passes of tools/pmc_r04.sh: matrix-pipe busy share, wave-cycle shares
(waiting / issuing), instructions per MFMA, LDS bank-conflict share, HBM bytes and GB/s.  usage: python tools/pmc_table.py
gpurun_out/pmc_r04s_ [substring ...]   (prefix of the five pass directories; substrings select kernels, default: wino)"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name[:60]


def main():
    prefix = sys.argv[1]
    subs = sys.argv[2:] or ["wino"]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for d in sorted(glob.glob(prefix + "*")):
        for path in glob.glob(d + "/*counter_collection.csv"):
            with open(path, newline="") as f:
                seen = set()
                for r in csv.DictReader(f):
                    if not any(s in r["Kernel_Name"] for s in subs):
                        continue
                    key = short(r["Kernel_Name"]) + " grid=" + r["Grid_Size"]
                    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r["Dispatch_Id"] not in seen:
                        seen.add(r["Dispatch_Id"])
                        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for key in sorted(acc):
        c = {k: sum(v) / len(v) for k, v in acc[key].items()}
        if c.get("SQ_INSTS_MFMA", 0) <= 0:
            continue
        ms = sorted(dur[key])[len(dur[key]) // 2]
        print(key)
        print("    median duration under the counters %.3f ms" % ms)
        if "GRBM_GUI_ACTIVE" in c:
            print("    matrix pipe busy                   %.1f %%   (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))"
                  % (100.0 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)))
        if "SQ_WAVE_CYCLES" in c:
            w = c["SQ_WAVE_CYCLES"]
            print("    wave cycles: waiting (s_waitcnt, barriers) %.1f %%, waiting to issue %.1f %%, issuing %.1f %%"
                  % (100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w))
        if c.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            print("    LDS bank-conflict cycles           %.1f %% of all LDS-array cycles (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)"
                  % (100 * c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
        if "SQ_INSTS_VALU" in c:
            m = c["SQ_INSTS_MFMA"]
            print("    per MFMA: %.2f VALU, %.2f LDS, %.2f VMEM reads, %.2f SALU instructions"
                  % ((c["SQ_INSTS_VALU"] - m) / m if c["SQ_INSTS_VALU"] > m else c["SQ_INSTS_VALU"] / m, c["SQ_INSTS_LDS"] / m,
                     c["SQ_INSTS_VMEM_RD"] / m, c["SQ_INSTS_SALU"] / m))
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            b = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            print("    HBM bytes per launch               %.2f GB = %.2f TB/s over the launch (%.2f of 8 TB/s)" % (b / 1e9, b / (ms * 1e-3) / 1e12,
                                                                                                             b / (ms * 1e-3) / 8e12))


if __name__ == "__main__":
    main()
