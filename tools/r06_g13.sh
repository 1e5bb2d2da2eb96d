# W4P_KILLDUP 2 (in-tree) vs 1 vs 0: correctness of the in-tree build, then per-launch times of every operand combination at stages 1-4
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 tools/wino4/persist_check.py --skip-bench > gpurun_out/r06/persist_check4.txt 2>&1; echo "FAIL lines: $(grep -c FAIL gpurun_out/r06/persist_check4.txt)"; tail -1 gpurun_out/r06/persist_check4.txt
for rep in 1 2; do
for lib in "" ad-yolo_amd/variants/lib_kd1.so ad-yolo_amd/variants/lib_kd0.so; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 400 python3 tools/wino4/persist_check.py --skip-check --stages 1,2,3,4 --iters 8 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140
  else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 400 python3 tools/wino4/persist_check.py --skip-check --stages 1,2,3,4 --iters 8 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140; fi
done; done > gpurun_out/r06/w4p_killdup_ab.txt 2>&1
python3 - <<'P'
import re, collections
t = collections.defaultdict(lambda: collections.defaultdict(list)); lib = None
for l in open("gpurun_out/r06/w4p_killdup_ab.txt"):
    m = re.match(r"== lib=\[(.*)\] rep", l)
    if m: lib = m.group(1).split("lib_")[-1].replace(".so", "") or "kd2"; continue
    m = re.match(r"stage\s+(\d+) (.*?)\s+(?:F\(2x2\)|one-patch)\s+[\d.]+ ms persistent ([\d.]+) ms", l)
    if m: t[(m.group(1), m.group(2))][lib].append(float(m.group(3)))
for k in sorted(t):
    r = {a: min(v) for a, v in t[k].items()}
    print("stage %s %-72s kd2 %.3f  kd1 %.3f  kd0 %.3f   kd2/kd0 %.3f" % (k[0], k[1][:72], r.get("kd2", 0), r.get("kd1", 0), r.get("kd0", 0), r.get("kd2", 0) / max(1e-9, r.get("kd0", 1))))
P
