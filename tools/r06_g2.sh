cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
timeout 120 tools/micro/lds_write_rate > gpurun_out/r06/lds_write_rate.txt 2>&1
cat gpurun_out/r06/lds_write_rate.txt
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r06/pytest_full1.txt
cat gpurun_out/r06/pytest_full1.txt
