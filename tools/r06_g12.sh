# dead pair-1 requests against an empty buffer (W4P_KILLDUP): correctness, then A/B of time and FETCH_SIZE (EPI 1 kernels; variant = the old form)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 tools/wino4/persist_check.py --skip-bench > gpurun_out/r06/persist_check3.txt 2>&1; echo "FAIL lines: $(grep -c FAIL gpurun_out/r06/persist_check3.txt)"; tail -2 gpurun_out/r06/persist_check3.txt
for rep in 1 2; do
for lib in "" ad-yolo_amd/variants/lib_e1_kd0.so; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 300 python3 tools/wino4/persist_check.py --skip-check --stages 1,2,3,4 --iters 8 --only "fwd conv" 2>/dev/null | grep "stage" | cut -c1-150
  else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python3 tools/wino4/persist_check.py --skip-check --stages 1,2,3,4 --iters 8 --only "fwd conv" 2>/dev/null | grep "stage" | cut -c1-150; fi
done; done > gpurun_out/r06/w4p_killdup_ab.txt 2>&1
cat gpurun_out/r06/w4p_killdup_ab.txt
VARIANTS="base e1_kd0" TAG=3 bash tools/r06_g11.sh
