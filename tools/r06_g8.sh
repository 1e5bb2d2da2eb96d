cd $GRAFT_REPO_ROOT
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv3x3 or stem or block" 2>&1 | tail -3
for rep in 1 2 3; do
for lib in "" ad-yolo_amd/variants/lib_conv_r5.so; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 200 python3 tools/conv_bench.py --which fwd --stages 0 --iters 20 2>/dev/null
  else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python3 tools/conv_bench.py --which fwd --stages 0 --iters 20 2>/dev/null; fi
done; done
