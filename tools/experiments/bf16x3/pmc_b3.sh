# usage (GPU box): bash tools/pmc_b3.sh  -> SQ counters of the fp32 and the bf16x3 Winograd forward kernels at the stage-2..4 bench shapes
# (tools/b3_bench.py launches both): matrix-pipe busy cycles, vector / LDS activity, waits, LDS bank conflicts.  Separate passes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmcb3_$i -o runc --output-format csv -- python3 $R/tools/b3_bench.py > $R/gpurun_out/pmcb3_$i.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmcb3_$i/runc_counter_collection.csv wino_fwd | grep -v "^$"
done
