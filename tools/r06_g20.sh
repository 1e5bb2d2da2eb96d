cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "se_tail or avgpool or se_basic" 2>&1 | tail -3
timeout 1500 python3 -m pytest tests/test_gpu_parity_scale.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for fp in 0 1; do
  echo "== ADYOLO_FUSE_POOL_BWD=$fp rep $rep"
  ADYOLO_FUSE_POOL_BWD=$fp timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-pipeline --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], 'fwd', d['stages']['encoder_fwd']['ms'], 'loss', d.get('final_loss'))"
done; done > gpurun_out/r06/fuse_pool_bwd_ab.txt 2>&1
cat gpurun_out/r06/fuse_pool_bwd_ab.txt
