# epilogue of the 64-channel-block kernels without mask sharing (stages 3 / 4, operand sets 27 / 31): mask words requested after the writer half (variant opo2) against before it (in-tree)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
OUT=$R/gpurun_out/r06/w4p_oporder_ab.txt
: > $OUT
ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_opo2.so timeout 1200 python3 tools/wino4/persist_check.py --skip-bench > gpurun_out/r06/persist_check14.txt 2>&1; echo "FAIL lines: $(grep -c FAIL gpurun_out/r06/persist_check14.txt)"; tail -1 gpurun_out/r06/persist_check14.txt
for rep in 1 2 3; do
for v in base opo2; do
  if [ $v = base ]; then unset ADYOLO_LIB; else export ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_$v.so; fi
  echo "== $v rep $rep" >> $OUT
  timeout 600 python3 $R/tools/wino4/persist_check.py --skip-check --stages 3,4 --iters 8 --only "stat bits" 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140 >> $OUT
done; done
python3 - <<'P' >> $OUT
import re, collections, os
t = collections.defaultdict(lambda: collections.defaultdict(list)); lib = None
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/w4p_oporder_ab.txt"):
    m = re.match(r"== (\w+) rep", l)
    if m: lib = m.group(1); continue
    m = re.match(r"stage\s+(\d+) (.*?)\s+(?:F\(2x2\)|one-patch)\s+[\d.]+ ms persistent ([\d.]+) ms", l)
    if m and lib: t[(m.group(1), m.group(2))][lib].append(float(m.group(3)))
print("== summary (min of 3 x 8 launches)")
for k in sorted(t):
    r = {a: min(v) for a, v in t[k].items()}
    print("stage %s %-72s masks late %.3f  masks early %.3f  ratio %.3f" % (k[0], k[1][:72], r["opo2"], r["base"], r["opo2"] / r["base"]))
P
tail -6 $OUT
