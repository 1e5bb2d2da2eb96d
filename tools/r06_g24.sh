# shared ReLU-mask words in the epilogue (MKM, in-tree) against one dword per pixel and component (variant mk0)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
OUT=$R/gpurun_out/r06/w4p_mkdedup_ab.txt
: > $OUT
timeout 900 python3 tools/wino4/persist_check.py --skip-bench > gpurun_out/r06/persist_check9.txt 2>&1; echo "FAIL lines: $(grep -c FAIL gpurun_out/r06/persist_check9.txt)"; tail -1 gpurun_out/r06/persist_check9.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_scale.py -x -q -m gpu -k "conv or block or seed100 or narrow or bench_shape" 2>&1 | tail -2
for rep in 1 2 3; do
for v in mk0 base; do
  if [ $v = base ]; then unset ADYOLO_LIB; else export ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_$v.so; fi
  echo "== $v rep $rep" >> $OUT
  timeout 400 python3 $R/tools/wino4/persist_check.py --skip-check --stages 1,2,3,4 --iters 8 --only "bits" 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140 >> $OUT
done; done
python3 - <<'P' >> $OUT
import re, collections, os
t = collections.defaultdict(lambda: collections.defaultdict(list)); lib = None
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/w4p_mkdedup_ab.txt"):
    m = re.match(r"== (\w+) rep", l)
    if m: lib = m.group(1); continue
    m = re.match(r"stage\s+(\d+) (.*?)\s+(?:F\(2x2\)|one-patch)\s+[\d.]+ ms persistent ([\d.]+) ms", l)
    if m and lib: t[(m.group(1), m.group(2))][lib].append(float(m.group(3)))
print("== summary (min of 3 x 8 launches)")
for k in sorted(t):
    r = {a: min(v) for a, v in t[k].items()}
    print("stage %s %-72s shared %.3f  per pixel %.3f  ratio %.3f" % (k[0], k[1][:72], r["base"], r["mk0"], r["base"] / r["mk0"]))
P
tail -14 $OUT
