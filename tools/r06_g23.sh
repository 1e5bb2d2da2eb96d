# FULL-line staging for 32 -> 32 layers with operand sets 15 / 27 / 31 (in-tree) against the half-line form (variant full0)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
OUT=$R/gpurun_out/r06/w4p_full_ab.txt
: > $OUT
cd $R
timeout 900 python3 tools/wino4/persist_check.py --skip-bench > gpurun_out/r06/persist_check8.txt 2>&1; echo "FAIL lines: $(grep -c FAIL gpurun_out/r06/persist_check8.txt)"; tail -1 gpurun_out/r06/persist_check8.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_scale.py -x -q -m gpu -k "conv or block or seed100 or narrow or bench_shape" 2>&1 | tail -2
cd /tmp
for rep in 1 2 3; do
for v in full0 base; do
  if [ $v = base ]; then unset ADYOLO_LIB; else export ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_$v.so; fi
  echo "== $v rep $rep" >> $OUT
  timeout 400 python3 $R/tools/wino4/persist_check.py --skip-check --stages 1 --iters 8 --only "dgrad conv1" 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140 >> $OUT
done; done
for v in base full0; do
  if [ $v = base ]; then unset ADYOLO_LIB; else export ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_$v.so; fi
  rm -rf $R/gpurun_out/fx3_${v}
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/fx3_${v} -o runc --output-format csv -- python3 $R/tools/wino4/persist_check.py --skip-check --iters 3 --stages 1 --only "dgrad conv1" > $R/gpurun_out/fx3_$v.log 2>&1
  echo "== FETCH $v" >> $OUT
  python3 - $R/gpurun_out/fx3_${v} >> $OUT <<'P'
import csv, glob, sys, collections, re
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(path, newline="")):
        if "wino4p" not in r["Kernel_Name"]: continue
        acc[re.search(r"wino4p_fwd_kernel<[^>]*>", r["Kernel_Name"]).group(0)].append(float(r["Counter_Value"]))
for key in sorted(acc):
    v = sorted(acc[key]); print("%-56s FETCH x2 %8.1f MB" % (key, 2 * v[len(v)//2] * 1024 / 1e6))
P
done
python3 - <<'P' >> $OUT
import re, collections, os
t = collections.defaultdict(lambda: collections.defaultdict(list)); lib = None
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/w4p_full_ab.txt"):
    m = re.match(r"== (\w+) rep", l)
    if m: lib = m.group(1); continue
    m = re.match(r"stage\s+(\d+) (.*?)\s+(?:F\(2x2\)|one-patch)\s+[\d.]+ ms persistent ([\d.]+) ms", l)
    if m and lib: t[(m.group(1), m.group(2))][lib].append(float(m.group(3)))
print("== summary (min of 3 x 8 launches)")
for k in sorted(t):
    r = {a: min(v) for a, v in t[k].items()}
    print("stage %s %-72s full lines %.3f  half lines %.3f  ratio %.3f" % (k[0], k[1][:72], r["base"], r["full0"], r["base"] / r["full0"]))
P
tail -16 $OUT
