import sys; sys.path.insert(0,'.')
import torch, numpy as np
from hiddenpose_amd import testing as hpt
from hiddenpose_amd.config import make_cfg
from hiddenpose_amd.NlosPose import NlosPose
from hiddenpose_amd.train_epoch import build_training, compute_loss
def rel(a,b): return float((a.double()-b.double()).norm()/b.double().norm())
out={}
for mode in ("fp32","bf16","bf16s"):
    T=N=32; B=2
    cfg=make_cfg(T,N,conv_precision=mode); m=NlosPose(cfg); hpt.fill_module(m); m=m.cuda().train()
    c,vc,opt,_=build_training(cfg,m)
    loss,jl,vl,heat,ref=compute_loss(m,c,vc,hpt.synthetic_meas(B,T,N).cuda(),hpt.synthetic_vol(B,T,N).cuda(),hpt.synthetic_joints(B,T//2).cuda())
    loss.backward(); torch.cuda.synchronize()
    out[mode]=(loss.item(),heat.detach().float(),{k:p.grad.detach().clone() for k,p in m.named_parameters()})
    print(mode,"loss",loss.item(), "heat dtype", heat.dtype, flush=True)
for mode in ("bf16","bf16s"):
    l,h,g=out[mode]; l0,h0,g0=out["fp32"]
    cos={k:float((g[k].double().flatten()@g0[k].double().flatten())/(g[k].double().norm()*g0[k].double().norm()+1e-300)) for k in ["pose_net.conv1.weight","pose_net.layer1.0.conv2.weight","pose_net.layer2.0.conv2.weight","pose_net.layer4.1.conv3.weight","pose_net.head.features.0.weight","pose_net.bn1.weight","feature_extraction.weights"]}
    print(mode,"loss ratio",l/l0-1,"heat rel",rel(h,h0),"cos",{k.split('.',1)[1]:round(v,4) for k,v in cos.items()})
