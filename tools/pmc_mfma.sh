cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for A in winograd direct; do
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/pmcm_$A -o runc --output-format csv -- python3 $R/tools/conv_bench.py --which fwd,wgrad --algo $A --stages 4 --iters 3 > $R/gpurun_out/pmcm_$A.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcm_$A/runc_counter_collection.csv _kernel
python3 - <<PY
import csv,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open("$R/gpurun_out/pmcm_$A/runc_kernel_trace.csv")):
    d[r["Kernel_Name"][:40]].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in d.items():
    if "kernel" in k and ("wino" in k or "conv3x3" in k): print(k, "avg us", sum(v)/len(v)/1e3, "n", len(v))
PY
done
