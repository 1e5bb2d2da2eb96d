# usage (on the GPU box): bash tools/wino4/whatif_p_run.sh [stages]  -- times the persistent F(4x4) kernel (plain launches) with each
# timing-only variant library ad-yolo_amd/variants/lib_w4p_<mask>.so (tools/build_variant.sh w4p_<mask> wino4p_e0.hip -DW4P_WHATIF=<mask>)
R=${GRAFT_REPO_ROOT:-/root/repo}
stages=${1:-2,3,4}
echo "== base"; python3 $R/tools/wino4/persist_check.py --skip-check --stages $stages --only "plain (0)" 2>&1 | grep "^stage"
for f in $R/ad-yolo_amd/variants/lib_w4p_*.so; do
  echo "== $(basename $f)"
  ADYOLO_LIB=$f timeout 300 python3 $R/tools/wino4/persist_check.py --skip-check --stages $stages --only "plain (0)" 2>&1 | grep "^stage"
done
