import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, numpy as np, torch.nn.functional as F
from hiddenpose_amd import hip_ops as ops
def cl(x): return x.permute(0,2,3,4,1).contiguous()
def ncdhw(x): return x.permute(0,4,1,2,3).contiguous()
g=torch.Generator().manual_seed(1)
cin,cout,dims=32,96,(1,4,4,4)
B,D,H,W=dims
x=torch.randn(B,cin,D,H,W,generator=g); w=torch.randn(cout,cin,3,3,3,generator=g)*0.05
for j in range(3):
  for tapsel in [None]:
    gy=torch.zeros(B,cout,2,2,2); gy[:,32*j:32*j+32]=torch.randn(B,32,2,2,2,generator=g)
    xd,wd=x.double().requires_grad_(True),w.double().requires_grad_(True)
    ref=F.conv3d(xd,wd,stride=2,padding=1); (ref*gy.double()).sum().backward()
    xc=cl(x).cuda(); wc=w.cuda(); desc=ops._desc(xc,cout,3,2,1,False)
    dx,dw=ops._conv_grads(desc,xc,wc,cl(gy).cuda(),True)
    e=(ncdhw(dx).cpu().double()-xd.grad)
    print('K slice',j,'total rel err',float(e.norm()/xd.grad.norm()), 'got norm',float(dx.norm()),'exp norm',float(xd.grad.norm()))
    for cls in range(8):
        pd,ph,pw=(cls>>2)&1,(cls>>1)&1,cls&1
        ee=e[:,:,pd::2,ph::2,pw::2]; rr=xd.grad[:,:,pd::2,ph::2,pw::2]
        print('   cls',cls,float(ee.norm()/rr.norm()))
