# usage (GPU box): bash tools/pmc_r05.sh  -> SQ / memory counters of the 3x3 convolution kernels at stage 1 (persistent F(4x4) forward, NB = 1; F(4x4)-domain weight gradient <NB = 1>)
# and stage 4 (persistent F(4x4), F(4x4)-domain weight gradient <NB = 2>), fused operands, B = 64 (round 5: the kernels that replaced round 4's).
# Counters in their own passes, kernel trace only (gpurun refuses --pmc together with the runtime trace domains).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_pmc_stage1_vs_stage4.txt
: > $OUT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_r05s_$i -o runc --output-format csv -- python3 $R/tools/conv_bench.py --which fwd,wgrad --stages 1,4 --fused --iters 2 > $R/gpurun_out/pmc_r05s_$i.log 2>&1
  echo "== pass $i: $C" >> $OUT
  python3 $R/tools/pmc_summary.py $(find $R/gpurun_out/pmc_r05s_$i -name "*counter_collection.csv" | head -1) wino | grep -v "^$" >> $OUT
done
cat $OUT
# LDS-array view (the guide's pair: SQ_LDS_BANK_CONFLICT = extra cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles)
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_r05s_6 -o runc --output-format csv -- python3 $R/tools/conv_bench.py --which fwd,wgrad --stages 1,4 --fused --iters 2 > $R/gpurun_out/pmc_r05s_6.log 2>&1
