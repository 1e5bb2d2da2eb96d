# usage (on the GPU box): bash tools/wino4/whatif_run.sh [stages]  -- times the F(4x4) kernel with each timing-only variant library
R=${GRAFT_REPO_ROOT:-/root/repo}
stages=${1:-3,4}
echo "== base"; python3 $R/tools/wino4/gpu_check.py --skip-check --stages $stages 2>&1 | grep stage | grep plain
for f in $R/ad-yolo_amd/whatif/lib_w4_*.so; do
  echo "== $(basename $f)"
  ADYOLO_LIB=$f timeout 300 python3 $R/tools/wino4/gpu_check.py --skip-check --stages $stages 2>&1 | grep stage | grep plain
done
