# resident-U form of the 32 -> 32 persistent kernel (W4P_BRES 1, in-tree) against the B ring (variant bres0): correctness, then stage-1 A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 tools/wino4/persist_check.py --skip-bench > gpurun_out/r06/persist_check7.txt 2>&1; echo "FAIL lines: $(grep -c FAIL gpurun_out/r06/persist_check7.txt)"; grep -c "OK\|ok" gpurun_out/r06/persist_check7.txt; tail -1 gpurun_out/r06/persist_check7.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_scale.py -x -q -m gpu -k "conv or block or seed100 or narrow or bench_shape" 2>&1 | tail -2
for rep in 1 2 3; do
for lib in ad-yolo_amd/variants/lib_bres0.so ""; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 400 python3 tools/wino4/persist_check.py --skip-check --stages 1 --iters 8 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140
  else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 400 python3 tools/wino4/persist_check.py --skip-check --stages 1 --iters 8 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140; fi
done; done > gpurun_out/r06/w4p_bres_ab.txt 2>&1
python3 - <<'P'
import re, collections
t = collections.defaultdict(lambda: collections.defaultdict(list)); lib = None
for l in open("gpurun_out/r06/w4p_bres_ab.txt"):
    m = re.match(r"== lib=\[(.*)\] rep", l)
    if m: lib = "ring" if m.group(1) else "resident"; continue
    m = re.match(r"stage\s+(\d+) (.*?)\s+(?:F\(2x2\)|one-patch)\s+[\d.]+ ms persistent ([\d.]+) ms", l)
    if m: t[(m.group(1), m.group(2))][lib].append(float(m.group(3)))
for k in sorted(t):
    r = {a: min(v) for a, v in t[k].items()}
    print("stage %s %-72s resident %.3f  ring %.3f  ratio %.3f" % (k[0], k[1][:72], r["resident"], r["ring"], r["resident"] / r["ring"]))
P
