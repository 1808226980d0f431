import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, ctypes as C
import torch.nn.functional as F
from hiddenpose_amd import _lib, testing as hpt
from hiddenpose_amd.NlosPoseSformer import NlosPoseSformer, _linear
from oracle import nlospose_oracle as O
def rel(a,b): return float((a.double().cpu()-b.double().cpu()).norm()/b.double().cpu().norm())
kw=dict(dim=64, num_frames=4, num_joints=24, image_size=32, patch_size=8, channels=1, depth=2, heads=4, dim_head=16, out_dim=128)
m=NlosPoseSformer(**kw); hpt.fill_module(m,'sformer.')
video=torch.rand(2,4,1,32,32,generator=torch.Generator().manual_seed(77))
sd={'sformer.'+k:v for k,v in m.state_dict().items()}
# oracle intermediates
p='sformer.'; b,f,c,H,W=video.shape; ps=8; hp=wp=4; nj=24
t=video.reshape(b,f,c,hp,ps,wp,ps).permute(0,1,3,5,4,6,2).reshape(b,f*hp*wp,ps*ps*c)
tok=F.linear(t,sd[p+'to_patch_embedding.weight'],sd[p+'to_patch_embedding.bias'])
x0=torch.cat((sd[p+'joints_token'].expand(b,-1,-1),tok),1)
sin,cos=O.axial_rotary_tables(hp,wp,sd[p+'image_rot_emb.scales'])
lp=p+'layers.0.'
h0=F.layer_norm(x0,x0.shape[-1:],sd[lp+'1.norm.weight'],sd[lp+'1.norm.bias'])
a0=O._sformer_attention(h0,sd,lp+'1.fn.',4,f,sin,cos,nj)
x1=x0+a0
# module pieces
L=_lib.lib(); mc=m.cuda(); dev=torch.device('cuda'); st=torch.cuda.current_stream().cuda_stream
vid=video.cuda().contiguous(); n=16; ntok=nj+f*n; dim=64; heads=4; dh=16
tokens=torch.empty(b*f*n,ps*ps*c,device=dev); L.hp_sformer_patchify(vid.data_ptr(),tokens.data_ptr(),b,f,c,H,W,ps,st)
emb=_linear(tokens,mc.to_patch_embedding.weight,mc.to_patch_embedding.bias)
x=torch.empty(b,ntok,dim,device=dev); x[:,:nj]=mc.joints_token; x[:,nj:]=emb.view(b,f*n,dim)
print('x0',rel(x,x0))
sp=mc.layers[0][1]; a=sp.fn; rows=b*ntok
h=torch.empty_like(x); L.hp_layernorm_forward(x.data_ptr(),h.data_ptr(),rows,dim,sp.norm.weight.data_ptr(),sp.norm.bias.data_ptr(),sp.norm.eps,0,0,st)
print('h0',rel(h,h0))
qkv=_linear(h.view(rows,dim),a.to_qkv.weight,None); print('qkv',rel(qkv,F.linear(h0,sd[lp+'1.fn.to_qkv.weight']).reshape(rows,-1)))
sin_t,cos_t=mc.image_rot_emb.tables(hp,wp,dev); print('tables',rel(sin_t,sin),rel(cos_t,cos), sin_t.shape)
q=torch.empty(b,heads,ntok,dh,device=dev); k=torch.empty_like(q); v=torch.empty_like(q)
L.hp_sformer_qkv_prepare(qkv.data_ptr(),q.data_ptr(),k.data_ptr(),v.data_ptr(),b,ntok,heads,dh,nj,n,C.c_float(a.scale),sin_t.data_ptr(),cos_t.data_ptr(),sin_t.shape[-1],st)
att=torch.empty(b,ntok,heads*dh,device=dev)
L.hp_sformer_attention(q.data_ptr(),k.data_ptr(),v.data_ptr(),att.data_ptr(),b,heads,dh,ntok,nj,n,f,st)
proj=_linear(att.view(rows,heads*dh),a.to_out[0].weight,a.to_out[0].bias); print('attn out',rel(proj.view(b,ntok,dim),a0))
L.hp_leaky_add_forward(x.data_ptr(),proj.data_ptr(),x.data_ptr(),x.numel(),C.c_float(1.0),st); print('x1',rel(x,x1))
print('scale',a.scale,type(a.scale))
# reference attention from the SAME Q,K,V the kernel saw
jout=torch.softmax(q[:,:,:nj]@k.transpose(-1,-2),-1)@v
pq=q[:,:,nj:].reshape(b,heads,f,n,dh); pk=k[:,:,nj:].reshape(b,heads,f,n,dh); pv=v[:,:,nj:].reshape(b,heads,f,n,dh)
kk=torch.cat((k[:,:,None,:nj].expand(-1,-1,f,-1,-1),pk),3); v2=torch.cat((v[:,:,None,:nj].expand(-1,-1,f,-1,-1),pv),3)
logits=pq@kk.transpose(-1,-2)
pout=torch.softmax(logits,-1)@v2
refo=torch.cat((jout,pout.reshape(b,heads,f*n,dh)),2)  # (b,h,N,d)
atth=att.view(b,ntok,heads,dh).permute(0,2,1,3)
print('joint',rel(atth[:,:,:nj],refo[:,:,:nj]),'patch',rel(atth[:,:,nj:],refo[:,:,nj:]))
print('logit range',float(logits.min()),float(logits.max()), 'joint logits',float((q[:,:,:nj]@k.transpose(-1,-2)).abs().max()))
for hh in range(heads):
    for ff in range(f):
        sl=slice(nj+ff*n,nj+(ff+1)*n)
        print(' head',hh,'frame',ff,rel(atth[:,hh,sl],refo[:,hh,sl]))
# oracle q,k,v
qkvo=F.linear(h0,sd[lp+'1.fn.to_qkv.weight'])
qo,ko,vo=(t.reshape(b,ntok,heads,dh).permute(0,2,1,3) for t in qkvo.chunk(3,-1)); qo=qo*dh**-0.5
rot=lambda t: t*cos+O._rotate_every_two(t)*sin
qo=qo.clone(); ko=ko.clone()
qo[:,:,nj:]=rot(qo[:,:,nj:].reshape(b,heads,f,n,dh)).reshape(b,heads,f*n,dh)
ko[:,:,nj:]=rot(ko[:,:,nj:].reshape(b,heads,f,n,dh)).reshape(b,heads,f*n,dh)
print('Q',rel(q,qo),'K',rel(k,ko),'V',rel(v,vo))
print('proj via torch from att',rel(F.linear(att,a.to_out[0].weight,a.to_out[0].bias),a0), 'proj hip vs torch',rel(proj.view(b,ntok,dim),F.linear(att,a.to_out[0].weight,a.to_out[0].bias)))
