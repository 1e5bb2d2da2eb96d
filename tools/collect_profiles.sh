# usage (here, after `gpurun -- bash tools/round_end.sh <tag>` has merged gpurun_out/): bash tools/collect_profiles.sh <tag>
# copies the round's judged summaries from gpurun_out/ (scratch) into profiles/ (tracked) under the names the earlier rounds use
set -e
R=$(cd $(dirname $0)/.. && pwd)
tag=${1:-r06}
g=$R/gpurun_out; p=$R/profiles
cp $g/${tag}_kernel_stats.csv $p/${tag}_bench_b64x60s_kernel_stats.csv
cp $g/${tag}_bench_under_rocprof.log $p/${tag}_bench_b64x60s_under_rocprof.log
cp $g/${tag}_bench_line.json $p/${tag}_bench_line.json
cp $g/${tag}_traffic.json $p/${tag}_traffic.json
cp $g/${tag}_pmc_fetch_summary.txt $p/${tag}_pmc_fetch_size_summary.txt
cp $g/${tag}_pmc_write_summary.txt $p/${tag}_pmc_write_size_summary.txt
cp $g/${tag}_stage_rooflines.json $g/${tag}_stage_rooflines.txt $p/
cp $g/${tag}_small_shapes.json $p/
for c in train_bs16x20s train_bs8x20s eval_bs1x60s eval_bs8x60s conformer_bs32x20s; do cp $g/${tag}_${c}_kernel_stats.csv $p/; done
tail -3 $g/${tag}_pytest_gpu.log > $p/${tag}_pytest_gpu_tail.txt
ls $p | grep "^${tag}_" | wc -l
