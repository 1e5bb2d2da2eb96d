# usage (on the GPU box): bash tools/clock_watch.sh <tag> -- samples sclk / power with rocm-smi while bench.py runs
R=$GRAFT_REPO_ROOT
tag=$1
rm -f $R/gpurun_out/${tag}_clocks.txt
python3 $R/bench.py --steps 60 --warmup 3 --no-stages --no-cpu-baseline > $R/gpurun_out/${tag}_bench.log 2>&1 &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power" | sed -e 's/.*(\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*(W): /W /' | tr '\n' ' ' >> $R/gpurun_out/${tag}_clocks.txt
  echo >> $R/gpurun_out/${tag}_clocks.txt
  sleep 0.3
done
wait $BP
tail -1 $R/gpurun_out/${tag}_bench.log | cut -c1-200
