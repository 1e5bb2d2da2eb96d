# usage (GPU box): bash tools/ab_conv.sh [conv_bench args]  -- conv micro-benchmark with the in-tree library and every variant library
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== in-tree (rep $rep)"; python3 tools/conv_bench.py "$@" 2>/dev/null
for v in ad-yolo_amd/variants/lib_*.so; do echo "== $v (rep $rep)"; ADYOLO_LIB=$GRAFT_REPO_ROOT/$v python3 tools/conv_bench.py "$@" 2>/dev/null; done
done
