cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stemprof -o run --output-format csv -- python3 $R/tools/stem_wgrad_check.py > $R/gpurun_out/r06/stemprof.log 2>&1
grep -h "stem_wgrad\|wgrad_reduce" $R/gpurun_out/stemprof/*kernel_stats.csv | cut -c1-200
python3 - <<'P'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
for p in glob.glob(R + "/gpurun_out/stemprof/*kernel_trace.csv"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(p)) if "stem_wgrad_kernel" in r["Kernel_Name"]]
    print("stem_wgrad_kernel durations (us), last 10:", [round(x, 1) for x in d[-10:]])
P
