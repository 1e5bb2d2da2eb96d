cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
for rep in 1 2; do
echo "== new"; timeout 300 python3 tools/stem_fwd_check.py --dump /tmp/new.pt 2>&1 | tail -4
echo "== old"; ADYOLO_LIB=$GRAFT_REPO_ROOT/ad-yolo_amd/variants/lib_stemfold.so timeout 300 python3 tools/stem_fwd_check.py --dump /tmp/old.pt 2>&1 | tail -1
done > gpurun_out/r06/stem_fwd_ab.txt 2>&1
python3 -c "
import torch
a=torch.load('/tmp/new.pt'); b=torch.load('/tmp/old.pt')
print('bit-identical to the old kernel:', [bool(torch.equal(x,y)) for x,y in zip(a,b)])" >> gpurun_out/r06/stem_fwd_ab.txt 2>&1
cat gpurun_out/r06/stem_fwd_ab.txt
