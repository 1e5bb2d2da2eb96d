cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 900 python3 tools/wino4/persist_check.py --skip-bench > gpurun_out/r06/persist_check2.txt 2>&1; echo "FAIL lines: $(grep -c FAIL gpurun_out/r06/persist_check2.txt)"; tail -2 gpurun_out/r06/persist_check2.txt
timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_scale.py -x -q -m gpu -k "conv or block or seed100 or narrow or bench_shape" 2>&1 | tail -3
for rep in 1 2; do
for lib in "" ad-yolo_amd/variants/lib_w4p_before.so; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 300 python3 tools/wino4/persist_check.py --skip-check --stages 1,2,3,4 --iters 8 2>/dev/null | grep "stage" | awk '{print $0}' | cut -c1-150
  else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python3 tools/wino4/persist_check.py --skip-check --stages 1,2,3,4 --iters 8 2>/dev/null | grep "stage" | cut -c1-150; fi
done; done > gpurun_out/r06/w4p_unroll2_ab.txt 2>&1
cat gpurun_out/r06/w4p_unroll2_ab.txt
