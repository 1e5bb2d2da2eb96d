cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 120 tools/micro/store_burst > gpurun_out/r06/store_burst.txt 2>&1
echo "--- conv ab" 
for rep in 1 2; do
for lib in "" ad-yolo_amd/variants/lib_stg12k.so ad-yolo_amd/variants/lib_stg34k.so; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 200 python3 tools/conv_bench.py --which fwd --stages 1,2,3 --iters 10 2>/dev/null
  else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python3 tools/conv_bench.py --which fwd --stages 1,2,3 --iters 10 2>/dev/null; fi
done; done > gpurun_out/r06/stagger_ab.txt 2>&1
timeout 1500 python3 -m pytest tests/test_gpu_graph.py "tests/test_gpu_parity_scale.py::test_seed100_training_step_matches_reference" "tests/test_gpu_parity_scale.py::test_dispatch_table_at_the_bench_shape" -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r06/pytest1.txt
cat gpurun_out/r06/store_burst.txt gpurun_out/r06/stagger_ab.txt; tail -30 gpurun_out/r06/pytest1.txt
