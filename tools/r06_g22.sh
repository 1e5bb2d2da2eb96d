# cache policy of the fused operands' loads (addend, statistics input: each read once) in the persistent kernel: time and traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
OUT=$R/gpurun_out/r06/w4p_opaux_ab.txt
: > $OUT
for rep in 1 2; do
for v in base opaux2 opaux1; do
  if [ $v = base ]; then unset ADYOLO_LIB; else export ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_$v.so; fi
  echo "== $v rep $rep" >> $OUT
  timeout 400 python3 $R/tools/wino4/persist_check.py --skip-check --stages 1,2,3,4 --iters 8 --only "dgrad" 2>/dev/null | grep "stage" | sed 's/  */ /g' | cut -c1-140 >> $OUT
done; done
for v in base opaux2; do
  if [ $v = base ]; then unset ADYOLO_LIB; else export ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_$v.so; fi
  rm -rf $R/gpurun_out/fx2_${v}
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/fx2_${v} -o runc --output-format csv -- python3 $R/tools/wino4/persist_check.py --skip-check --iters 3 --stages 1,2 --only "dgrad" > $R/gpurun_out/fx2_$v.log 2>&1
  echo "== FETCH $v" >> $OUT
  python3 - $R/gpurun_out/fx2_${v} >> $OUT <<'P'
import csv, glob, sys, collections, re
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(path, newline="")):
        if "wino4p" not in r["Kernel_Name"]: continue
        acc[re.search(r"wino4p_fwd_kernel<[^>]*>", r["Kernel_Name"]).group(0)].append(float(r["Counter_Value"]))
for key in sorted(acc):
    v = sorted(acc[key]); print("%-50s FETCH x2 %8.1f MB" % (key, 2 * v[len(v)//2] * 1024 / 1e6))
P
done
python3 - <<'P' >> $OUT
import re, collections, os
t = collections.defaultdict(lambda: collections.defaultdict(list)); lib = None
for l in open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06/w4p_opaux_ab.txt"):
    m = re.match(r"== (\w+) rep", l)
    if m: lib = m.group(1); continue
    m = re.match(r"stage\s+(\d+) (.*?)\s+(?:F\(2x2\)|one-patch)\s+[\d.]+ ms persistent ([\d.]+) ms", l)
    if m and lib: t[(m.group(1), m.group(2))][lib].append(float(m.group(3)))
print("== summary (min of 2 x 8 launches)")
for k in sorted(t):
    r = {a: min(v) for a, v in t[k].items()}
    print("stage %s %-66s base %.3f  nt %.3f (%.3f)  sc0 %.3f (%.3f)" % (k[0], k[1][:66], r["base"], r["opaux2"], r["opaux2"] / r["base"], r["opaux1"], r["opaux1"] / r["base"]))
P
tail -45 $OUT
