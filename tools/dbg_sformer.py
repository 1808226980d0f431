import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, ctypes as C, math
import torch.nn.functional as F
from hiddenpose_amd import _lib
from oracle import nlospose_oracle as O
L=_lib.lib(); st=torch.cuda.current_stream().cuda_stream
def rel(a,b): return float((a.double()-b.double()).norm()/b.double().norm())
g=torch.Generator().manual_seed(1)
# patchify
B,Fr,Cc,H,W,ps=2,3,1,16,16,4
v=torch.rand(B,Fr,Cc,H,W,generator=g).cuda()
hp=H//ps
ref=v.reshape(B,Fr,Cc,hp,ps,hp,ps).permute(0,1,3,5,4,6,2).reshape(B*Fr*hp*hp,ps*ps*Cc)
out=torch.empty_like(ref); L.hp_sformer_patchify(v.data_ptr(),out.data_ptr(),B,Fr,Cc,H,W,ps,st); print('patchify',rel(out,ref))
# layernorm
x=torch.randn(50,64,generator=g).cuda(); gm=torch.randn(64,generator=g).cuda(); bt=torch.randn(64,generator=g).cuda()
y=torch.empty_like(x); L.hp_layernorm_forward(x.data_ptr(),y.data_ptr(),50,64,gm.data_ptr(),bt.data_ptr(),1e-5,0,0,st); print('ln',rel(y,F.layer_norm(x,(64,),gm,bt)))
# geglu
u=torch.randn(33,2*40,generator=g).cuda(); gg=torch.empty(33,40,device='cuda'); L.hp_geglu_forward(u.data_ptr(),gg.data_ptr(),33,40,st)
a,gt=u.chunk(2,-1); print('geglu',rel(gg,a*F.gelu(gt)))
# linear
from hiddenpose_amd.NlosPoseSformer import _linear
xx=torch.randn(100,64,generator=g).cuda(); w=torch.randn(48,64,generator=g).cuda(); b=torch.randn(48,generator=g).cuda()
print('linear',rel(_linear(xx,w,b),F.linear(xx,w,b)))
# qkv prepare + attention
for dh in (16,32):
    Bq,heads,nj,n,fr=2,2,24,16,3; ntok=nj+fr*n; inner=heads*dh
    qkv=torch.randn(Bq,ntok,3*inner,generator=g).cuda()
    sc=torch.logspace(0.,math.log(10/2)/math.log(2),dh//4,base=2)
    sin,cos=O.axial_rotary_tables(4,4,sc); sin,cos=sin.cuda().contiguous(),cos.cuda().contiguous()
    Q=torch.empty(Bq,heads,ntok,dh,device='cuda'); K=torch.empty_like(Q); V=torch.empty_like(Q)
    L.hp_sformer_qkv_prepare(qkv.data_ptr(),Q.data_ptr(),K.data_ptr(),V.data_ptr(),Bq,ntok,heads,dh,nj,n,dh**-0.5,sin.data_ptr(),cos.data_ptr(),sin.shape[-1],st)
    q,k,vv=(t.reshape(Bq,ntok,heads,dh).permute(0,2,1,3) for t in qkv.chunk(3,-1)); q=q*dh**-0.5
    rot=lambda t: t*cos+O._rotate_every_two(t)*sin
    qr=q.clone(); kr=k.clone()
    qr[:,:,nj:]=rot(q[:,:,nj:].reshape(Bq,heads,fr,n,dh)).reshape(Bq,heads,fr*n,dh)
    kr[:,:,nj:]=rot(k[:,:,nj:].reshape(Bq,heads,fr,n,dh)).reshape(Bq,heads,fr*n,dh)
    print('dh',dh,'prep q',rel(Q,qr),'k',rel(K,kr),'v',rel(V,vv))
    att=torch.zeros(Bq,ntok,inner,device='cuda')
    L.hp_sformer_attention(Q.data_ptr(),K.data_ptr(),V.data_ptr(),att.data_ptr(),Bq,heads,dh,ntok,nj,n,fr,st)
    jout=torch.softmax(qr[:,:,:nj]@kr.transpose(-1,-2),-1)@vv
    pq=qr[:,:,nj:].reshape(Bq,heads,fr,n,dh); pk=kr[:,:,nj:].reshape(Bq,heads,fr,n,dh); pv=vv[:,:,nj:].reshape(Bq,heads,fr,n,dh)
    kk=torch.cat((kr[:,:,None,:nj].expand(-1,-1,fr,-1,-1),pk),3); v2=torch.cat((vv[:,:,None,:nj].expand(-1,-1,fr,-1,-1),pv),3)
    pout=torch.softmax(pq@kk.transpose(-1,-2),-1)@v2
    refo=torch.cat((jout,pout.reshape(Bq,heads,fr*n,dh)),2).permute(0,2,1,3).reshape(Bq,ntok,inner)
    print('   attn joint',rel(att[:,:nj],refo[:,:nj]),'patch',rel(att[:,nj:],refo[:,nj:]))
