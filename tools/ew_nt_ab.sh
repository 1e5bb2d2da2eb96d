# usage (GPU box): bash tools/ew_nt_ab.sh <libA|-> <libB> -- the bench's stage figures with two libraries, A B A B ("-" = in-tree)
a=${1:--}; b=$2
for lib in $a $b $a $b; do
  [ "$lib" = "-" ] && lib=""
  echo "== lib: ${lib:-in-tree}"
  ADYOLO_LIB=$lib python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-pipeline --no-parity 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
st=d['stages']
k=st['conv3x3_fwd_dgrad']['kernels']['wino4p_fwd_kernel']
print('ms_per_step', d['ms_per_step'], ' wino4p avg', k['avg_launch_ms'], ' se_tail_fwd', st['se_tail_fwd']['ms_per_step'], st['se_tail_fwd']['frac_of_hbm_peak'], ' se_tail_bwd', st['se_tail_bwd (reduce + fc + apply)']['ms_per_step'], ' bn_bwd', st['bn_bwd (reduce + apply)']['ms_per_step'], ' ew_total', st['_elementwise_total']['ms_per_step'])
"
done
