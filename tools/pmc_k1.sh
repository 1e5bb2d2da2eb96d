# usage (GPU box): bash tools/pmc_k1.sh <tag>   -> gpurun_out/<tag>_k1_pmc_*.txt (K1 counters, separate passes per counter group)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
run() { # $1 = pass name, rest = counters
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_k1_${tag}_$name -o run --output-format csv -- python3 $R/tools/feat_bench.py > $R/gpurun_out/${tag}_k1_pmc_$name.log 2>&1
  python3 $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_k1_${tag}_$name -name "*counter_collection.csv" | head -1) feat_ > $R/gpurun_out/${tag}_k1_pmc_$name.txt
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE
cat $R/gpurun_out/${tag}_k1_pmc_sq1.txt $R/gpurun_out/${tag}_k1_pmc_sq2.txt $R/gpurun_out/${tag}_k1_pmc_fetch.txt $R/gpurun_out/${tag}_k1_pmc_write.txt $R/gpurun_out/${tag}_k1_pmc_grbm.txt
