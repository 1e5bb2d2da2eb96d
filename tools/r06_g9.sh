cd $GRAFT_REPO_ROOT
for rep in 1 2; do for ev in torch timing; do
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pipeline --no-parity --event-kind $ev 2>gpurun_out/r06_ev.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$ev', d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['stages']['conv3x3_wgrad']['avg_launch_ms'], d['stages']['conv3x3_wgrad']['share_of_step'], d['stages']['k1_features']['ms'], d['stages']['_elementwise_total'], d['stages']['encoder_fwd']['ms'])
" || tail -5 gpurun_out/r06_ev.err
done; done
