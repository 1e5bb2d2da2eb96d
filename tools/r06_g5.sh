cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for rep in 1 2 3; do
for lib in "" ad-yolo_amd/variants/lib_ew_r5.so; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 200 python3 tools/ew_bench.py 2>/dev/null | grep se_tail; else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python3 tools/ew_bench.py 2>/dev/null | grep se_tail; fi
done; done > gpurun_out/r06/se_tail_fwd_ab.txt 2>&1
cat gpurun_out/r06/se_tail_fwd_ab.txt
timeout 600 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity_scale.py -x -q -m gpu -k "block or seed100 or se_" 2>&1 | tail -3
