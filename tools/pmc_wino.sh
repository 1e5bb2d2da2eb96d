cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES --kernel-trace -d $R/gpurun_out/pmcw1 -o runc --output-format csv -- python3 $R/tools/conv_bench.py --which fwd --algo winograd --stages 2,4 --iters 2 > $R/gpurun_out/pmcw1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $R/gpurun_out/pmcw2 -o runc --output-format csv -- python3 $R/tools/conv_bench.py --which fwd --algo winograd --stages 2,4 --iters 2 > $R/gpurun_out/pmcw2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcw3 -o runc --output-format csv -- python3 $R/tools/conv_bench.py --which fwd --algo winograd --stages 2,4 --iters 2 > $R/gpurun_out/pmcw3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmcw4 -o runc --output-format csv -- python3 $R/tools/conv_bench.py --which fwd --algo winograd --stages 2,4 --iters 2 > $R/gpurun_out/pmcw4.log 2>&1
cd $R; for i in 1 2 3 4; do f=$(ls gpurun_out/pmcw$i/*/*counter_collection.csv gpurun_out/pmcw$i/*counter_collection.csv 2>/dev/null | head -1); python3 tools/pmc_summary.py $f wino_fwd; done
