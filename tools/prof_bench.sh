# usage (on the GPU box): bash tools/prof_bench.sh <tag> [bench args...]   -> gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o run --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/${tag}_bench_under_rocprof.log 2>&1
f=$(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/${tag}_kernel_stats.csv
tail -1 $R/gpurun_out/${tag}_bench_under_rocprof.log | cut -c1-300
