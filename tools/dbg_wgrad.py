import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import adyolo_amd
from adyolo_amd import ops
torch.manual_seed(0)
def run(n,h,w,cin,cout,mean,aff,zm):
    x=(torch.relu(torch.randn(n,cin,h,w))+mean)
    dy=torch.randn(n,cout,h,w)
    if zm: dy=dy-dy.mean(dim=(0,2,3),keepdim=True)
    sc=torch.rand(cin)+0.5; sh=torch.randn(cin)
    xa=x*sc[None,:,None,None]+sh[None,:,None,None] if aff else x
    wr=torch.zeros(cout,cin,3,3,dtype=torch.float64,requires_grad=True)
    F.conv2d(xa.double(),wr,None,padding=1).backward(dy.double())
    t=wr.grad
    xg=x.permute(0,2,3,1).contiguous().cuda(); dyg=dy.permute(0,2,3,1).contiguous().cuda()
    a=(sc.cuda(),sh.cuda()) if aff else None
    out=[]
    for algo in ("direct","winograd"):
        dw=ops.conv3x3_wgrad(xg,dyg,cin,in_affine=a,algo=algo).cpu().double()
        out.append(float((dw-t).abs().max()/t.abs().max()))
    print("n%d %dx%d %d->%d mean %g aff %d zm %d: direct %.2e  winograd %.2e"%(n,h,w,cin,cout,mean,aff,zm,out[0],out[1]))
run(2,16,16,128,128,0,0,0)
run(2,16,16,128,128,0,1,1)
run(2,16,16,128,128,3,1,1)
run(2,16,16,128,256,0,0,1)
run(2,32,32,64,64,0,1,1)
run(2,64,64,32,32,0,1,1)
run(8,300,16,128,128,0,1,1)   # many items per workgroup
run(4,200,64,32,32,0,0,0)
run(16,130,32,64,64,0,1,0)
