# usage (GPU box): bash tools/small_shapes_profile.sh <tag>
# rocprofv3 kernel trace of every bench.py extra config (eager variant: a known number of identical steps) -> sum of kernel
# durations per step -> gpurun_out/<tag>_small_shapes.json (copy to profiles/r03_small_shapes.json) + per-config kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=${1:-r04}
for cfg in train_bs16x20s train_bs8x20s eval_bs1x60s eval_bs8x60s conformer_bs32x20s; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_$cfg -o run --output-format csv -- python3 $R/bench.py --only-extra $cfg --extra-mode eager > $R/gpurun_out/${tag}_${cfg}_under_rocprof.log 2>&1
  cp $(find $R/gpurun_out/prof_${tag}_$cfg -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_${cfg}_kernel_stats.csv
done
cd $R
python3 tools/small_shapes_summary.py $tag
