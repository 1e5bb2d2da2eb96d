# usage (GPU box): bash tools/b3_ab.sh  -> tools/b3_bench.py on the in-tree library and on every ad-yolo_amd/variants/lib_*.so
cd $GRAFT_REPO_ROOT
echo "== in-tree"; python3 tools/b3_bench.py 2>&1 | grep -v amdgpu.ids | grep -v "32->"
for v in ad-yolo_amd/variants/lib_*.so; do echo "== $v"; ADYOLO_LIB=$GRAFT_REPO_ROOT/$v python3 tools/b3_bench.py 2>&1 | grep -v amdgpu.ids | grep -v "32->"; done
