# FETCH_SIZE / WRITE_SIZE of every operand set of the persistent kernel at stage 1 (the fused data gradients are bandwidth-shaped: where do their bytes go?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/fx1_$C
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/fx1_$C -o runc --output-format csv -- python3 $R/tools/wino4/persist_check.py --skip-check --iters 3 --stages 1,2 > $R/gpurun_out/fx1_$C.log 2>&1
done
python3 - $R/gpurun_out/fx1_ > $R/gpurun_out/r06/w4p_fetch_stage1_sets.txt <<'P'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "*/*counter_collection.csv"):
    for r in csv.DictReader(open(path, newline="")):
        if "wino4p" not in r["Kernel_Name"]: continue
        m = re.search(r"wino4p_fwd_kernel<[^>]*>", r["Kernel_Name"])
        key = m.group(0)
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key in sorted(acc):
    c = {k: sorted(v)[len(v)//2] for k, v in acc[key].items()}
    print("%-50s n=%3d  %8.1f us  FETCH x2 %8.1f MB  WRITE %8.1f MB" % (key, len(dur[key]), sorted(dur[key])[len(dur[key])//2], 2*c.get("FETCH_SIZE",0)*1024/1e6, c.get("WRITE_SIZE",0)*1024/1e6))
P
cat $R/gpurun_out/r06/w4p_fetch_stage1_sets.txt
cd $R && timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
