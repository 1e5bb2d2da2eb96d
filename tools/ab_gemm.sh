# usage (GPU box): bash tools/ab_gemm.sh  -- gemm_bench + the ResNet-Conformer bench with the in-tree library and each library under ad-yolo_amd/whatif/
R=$GRAFT_REPO_ROOT
echo "== in-tree"; python3 $R/tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
python3 $R/bench.py --encoder resnet-conformer --batch 32 --seconds 20 --steps 5 --warmup 2 --no-stages --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
for f in $R/ad-yolo_amd/whatif/*.so; do
  echo "== $(basename $f)"; ADYOLO_LIB=$f python3 $R/tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids
  ADYOLO_LIB=$f python3 $R/bench.py --encoder resnet-conformer --batch 32 --seconds 20 --steps 5 --warmup 2 --no-stages --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
done
