# usage (GPU box): bash tools/feat_ab.sh   -> K1 timings of the in-tree library and of every ad-yolo_amd/variants/lib_*.so
cd $GRAFT_REPO_ROOT
echo "== in-tree"; python3 tools/feat_bench.py
for v in ad-yolo_amd/variants/lib_*.so; do echo "== $v"; ADYOLO_LIB=$GRAFT_REPO_ROOT/$v python3 tools/feat_bench.py; done
