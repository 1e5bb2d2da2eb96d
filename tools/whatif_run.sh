# usage (on the GPU box): bash tools/whatif_run.sh <which> <stages>  -- times conv_bench with each timing-only variant library
R=$GRAFT_REPO_ROOT
which=${1:-fwd}; stages=${2:-1,2,3,4}
echo "== base"; python3 $R/tools/conv_bench.py --which $which --stages $stages --iters 10
for f in $R/ad-yolo_amd/whatif/*.so; do
  echo "== $(basename $f)"
  ADYOLO_LIB=$f timeout 300 python3 $R/tools/conv_bench.py --which $which --stages $stages --iters 10
done
