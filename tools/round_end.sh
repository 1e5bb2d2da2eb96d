# usage (GPU box): bash tools/round_end.sh <tag>   -> the GPU suite, the round's profiles and the full bench line under gpurun_out/
R=$GRAFT_REPO_ROOT
tag=${1:-r06}
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1
tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 900 bash tools/prof_round.sh $tag
cd $R
timeout 900 bash tools/small_shapes_profile.sh $tag > gpurun_out/${tag}_small_shapes.log 2>&1
cd $R
timeout 900 python3 bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
cut -c1-400 gpurun_out/${tag}_bench_line.json
