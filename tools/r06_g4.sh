cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for rep in 1 2; do
for lib in "" ad-yolo_amd/variants/lib_ew_noseg.so; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 200 python3 tools/ew_bench.py 2>/dev/null; else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python3 tools/ew_bench.py 2>/dev/null; fi
done; done > gpurun_out/r06/ew_seg_ab.txt 2>&1
cat gpurun_out/r06/ew_seg_ab.txt
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bn or block or stem" 2>&1 | tail -3
