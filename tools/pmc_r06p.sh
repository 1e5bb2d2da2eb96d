# usage (GPU box): bash tools/pmc_r06p.sh -> SQ / memory counters of the PERSISTENT F(4x4) kernel (forward conv2 operand set: affine + statistics)
# at stages 1 (NB = 1) and 4, B = 64; counters in their own passes, kernel trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_r06p_$i -o runc --output-format csv -- python3 $R/tools/wino4/persist_check.py --skip-check --iters 2 --stages 1,2,3,4 --only "fwd conv2" > $R/gpurun_out/pmc_r06p_$i.log 2>&1
done
python3 $R/tools/pmc_table.py $R/gpurun_out/pmc_r06p_ wino4p > $R/gpurun_out/r06_pmc_persistent.txt 2>&1
cat $R/gpurun_out/r06_pmc_persistent.txt
