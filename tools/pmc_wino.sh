# usage (GPU box): bash tools/pmc_wino.sh <stages> <which>   -> SQ stall counters + L2 hit + HBM bytes of the Winograd kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ST=${1:-4}; WH=${2:-fwd}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmcw$i -o runc --output-format csv -- python3 $R/tools/conv_bench.py --which $WH --algo winograd --stages $ST --iters 2 > $R/gpurun_out/pmcw$i.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcw$i/runc_counter_collection.csv wino_ | grep -v "^$"
done
