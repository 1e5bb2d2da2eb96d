cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wgrad" 2>&1 | tail -5
timeout 300 python3 tools/wino4w/gpu_check.py 2>&1 | tail -25
for rep in 1 2; do
for lib in "" ad-yolo_amd/variants/lib_w4w_r5.so; do
  echo "== lib=[$lib] rep $rep"
  for fused in "" "--fused"; do
  if [ -z "$lib" ]; then timeout 200 python3 tools/conv_bench.py --which wgrad --stages 1,2,3,4,12,23,34 --iters 10 $fused 2>/dev/null
  else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python3 tools/conv_bench.py --which wgrad --stages 1,2,3,4,12,23,34 --iters 10 $fused 2>/dev/null; fi
  done
done; done > gpurun_out/r06/w4w_unroll_ab.txt 2>&1
cat gpurun_out/r06/w4w_unroll_ab.txt
