# stage-1 / stage-2 read excess of the persistent forward kernel: FETCH_SIZE and duration per variant (store policy, load policy, no stores)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
OUT=$R/gpurun_out/r06/w4p_fetch_excess${TAG}.txt
: > $OUT
for v in ${VARIANTS:-base e1_st0 e1_x1 e1_x2 e1_x3 e1_nost}; do
  if [ $v = base ]; then unset ADYOLO_LIB; else export ADYOLO_LIB=$R/ad-yolo_amd/variants/lib_$v.so; fi
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/fx_${v}_$C
    timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/fx_${v}_$C -o runc --output-format csv -- python3 $R/tools/wino4/persist_check.py --skip-check --iters 3 --stages 1,2,3,4 --only "fwd conv2" > $R/gpurun_out/fx_${v}_$C.log 2>&1
  done
  echo "== $v" >> $OUT
  python3 - $R/gpurun_out/fx_${v}_ >> $OUT <<'P'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "*/*counter_collection.csv"):
    for r in csv.DictReader(open(path, newline="")):
        if "wino4p" not in r["Kernel_Name"]: continue
        m = re.search(r"wino4p_fwd_kernel<[^>]*>", r["Kernel_Name"])
        key = m.group(0) + " lds=" + r.get("LDS_Block_Size", "?") + " wg=" + r["Grid_Size"]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key in sorted(acc):
    c = {k: sorted(v)[len(v)//2] for k, v in acc[key].items()}
    print("%-60s n=%3d  %8.1f us  FETCH x2 %8.1f MB  WRITE %8.1f MB" % (key, len(dur[key]), sorted(dur[key])[len(dur[key])//2], 2*c.get("FETCH_SIZE",0)*1024/1e6, c.get("WRITE_SIZE",0)*1024/1e6))
P
done
cat $OUT
