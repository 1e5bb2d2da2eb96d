# usage (on the GPU box): bash tools/ab_bench.sh [bench args]  -- bench.py with the in-tree library and with every library
# under ad-yolo_amd/whatif/ (ADYOLO_LIB), same box, back to back, twice
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "in-tree: $(python3 $R/bench.py --steps 10 --warmup 3 --no-stages --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  for f in $R/ad-yolo_amd/whatif/*.so; do
    echo "$(basename $f): $(ADYOLO_LIB=$f python3 $R/bench.py --steps 10 --warmup 3 --no-stages --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
  done
done
