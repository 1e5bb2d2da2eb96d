# step-16 requests of the per-pair bodies kept in place by a compiler barrier (W4P_REQ_FENCE 1, in-tree) against the free form (variant fence0)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
OUT=$R/gpurun_out/r06/w4p_reqfence_ab.txt
: > $OUT
timeout 1200 python3 tools/wino4/persist_check.py --skip-bench > gpurun_out/r06/persist_check13.txt 2>&1; echo "FAIL lines: $(grep -c FAIL gpurun_out/r06/persist_check13.txt)"; tail -1 gpurun_out/r06/persist_check13.txt
for rep in 1 2 3; do
for v in fence0 base; do
  if [ $v = base ]; then unset ADYOLO_LIB; else export ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_$v.so; fi
  echo "== $v rep $rep" >> $OUT
  timeout 600 python3 $R/tools/wino4/persist_check.py --skip-check --stages 1 --iters 8 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140 >> $OUT
done; done
python3 - <<'P' >> $OUT
import re, collections, os
t = collections.defaultdict(lambda: collections.defaultdict(list)); lib = None
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/w4p_reqfence_ab.txt"):
    m = re.match(r"== (\w+) rep", l)
    if m: lib = m.group(1); continue
    m = re.match(r"stage\s+(\d+) (.*?)\s+(?:F\(2x2\)|one-patch)\s+[\d.]+ ms persistent ([\d.]+) ms", l)
    if m and lib: t[(m.group(1), m.group(2))][lib].append(float(m.group(3)))
print("== summary (min of 2 x 8 launches)")
for k in sorted(t):
    r = {a: min(v) for a, v in t[k].items()}
    print("stage %s %-72s fenced %.3f  free %.3f  ratio %.3f" % (k[0], k[1][:72], r["base"], r["fence0"], r["base"] / r["fence0"]))
P
tail -38 $OUT
