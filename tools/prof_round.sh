# usage (GPU box): bash tools/prof_round.sh <tag>     -> kernel stats, bench log and HBM-traffic PMC passes under gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o run --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline --no-parity > $R/gpurun_out/${tag}_bench_under_rocprof.log 2>&1
cp $(find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_$tag -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline --no-parity > $R/gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write_$tag -o run --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline --no-parity > $R/gpurun_out/${tag}_pmc_write.log 2>&1
cd $R
python3 tools/make_traffic.py $(find gpurun_out/pmc_fetch_$tag -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_write_$tag -name "*counter_collection.csv" | head -1) winograd wino_fwd_kernel gpurun_out/${tag}_traffic.json
# round 4: the F(4x4,3x3) kernel of the default algorithm (bench.py looks its algorithm's key up: roofline.traffic)
python3 tools/make_traffic.py $(find gpurun_out/pmc_fetch_$tag -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_write_$tag -name "*counter_collection.csv" | head -1) winograd4 wino4p_fwd_kernel gpurun_out/${tag}_traffic.json
python3 tools/pmc_summary.py $(find gpurun_out/pmc_fetch_$tag -name "*counter_collection.csv" | head -1) adyolo > gpurun_out/${tag}_pmc_fetch_summary.txt
python3 tools/pmc_summary.py $(find gpurun_out/pmc_write_$tag -name "*counter_collection.csv" | head -1) adyolo > gpurun_out/${tag}_pmc_write_summary.txt
# per (full template name, grid) HBM rooflines; durations from the kernel trace of the --stats pass (no counters running)
python3 tools/kernel_rooflines.py $(find gpurun_out/pmc_fetch_$tag -name "*counter_collection.csv" | head -1) $(find gpurun_out/pmc_write_$tag -name "*counter_collection.csv" | head -1) gpurun_out/${tag}_stage_rooflines.json --trace $(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1) > gpurun_out/${tag}_stage_rooflines.txt
tail -1 gpurun_out/${tag}_bench_under_rocprof.log | cut -c1-200
