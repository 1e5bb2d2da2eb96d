# usage (GPU box): bash tools/whatif_gemm.sh  -- gemm_bench with the in-tree library and each timing-only variant under ad-yolo_amd/whatif/
R=$GRAFT_REPO_ROOT
echo "== in-tree"; python3 $R/tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
for f in $R/ad-yolo_amd/whatif/*.so; do
  echo "== $(basename $f)"; ADYOLO_LIB=$f timeout 300 python3 $R/tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
done
