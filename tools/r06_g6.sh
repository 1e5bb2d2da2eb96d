cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "" ad-yolo_amd/variants/lib_br12.so ad-yolo_amd/variants/lib_br18.so; do
  echo "== lib=[$lib] rep $rep"
  if [ -z "$lib" ]; then timeout 200 python3 tools/conv_bench.py --which fwd --stages 2,3,4,12,23,34 --iters 10 2>/dev/null
  else ADYOLO_LIB=$GRAFT_REPO_ROOT/$lib timeout 200 python3 tools/conv_bench.py --which fwd --stages 2,3,4,12,23,34 --iters 10 2>/dev/null; fi
done; done
