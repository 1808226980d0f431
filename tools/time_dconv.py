import sys, time; sys.path.insert(0, '.')
import torch
from hiddenpose_amd import _lib
L = _lib.lib()
st = torch.cuda.current_stream().cuda_stream
for (cin, cout, rep, dims) in [(1,1,1,(4,512,128,128)), (1,4,0,(4,512,128,128)), (4,4,0,(4,512,128,128)), (8,4,0,(4,512,128,128)), (4,8,0,(4,256,64,64)), (16,32,0,(4,64,16,16)), (64,16,0,(4,64,16,16))]:
    B,D,H,W = dims
    x = torch.randn(B,cin,D,H,W,device='cuda'); g = torch.randn(B,cout,D,H,W,device='cuda')
    w = torch.randn(cout,cin,3,3,3,device='cuda'); y=torch.empty_like(g); gx=torch.empty_like(x)
    dw = torch.empty(cout,cin,3,3,3,device='cuda'); db = torch.empty(cout,device='cuda')
    ws = torch.empty(int(L.hp_dconv3_backward_data_workspace_bytes(B,cin,D,H,W,rep))//4+1, device='cuda')
    def t(fn):
        fn(); torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); return (time.perf_counter()-t0)/5*1e3
    a=t(lambda: L.hp_dconv3_forward(x.data_ptr(),w.data_ptr(),None,y.data_ptr(),B,cin,cout,D,H,W,rep,st))
    b=t(lambda: L.hp_dconv3_backward_data(g.data_ptr(),w.data_ptr(),gx.data_ptr(),B,cin,cout,D,H,W,rep,ws.data_ptr(),st))
    wsw = torch.empty(int(L.hp_dconv3_backward_weight_workspace_bytes(B,cin,cout,D,H,W))//4, device='cuda')
    c=t(lambda: L.hp_dconv3_backward_weight(x.data_ptr(),g.data_ptr(),dw.data_ptr(),db.data_ptr(),B,cin,cout,D,H,W,rep,wsw.data_ptr(),st))
    gf=2*B*D*H*W*27*cin*cout/1e9; gb=(cin+cout)*B*D*H*W*4/1e9
    print(f"{cin}->{cout} rep{rep} {dims}: fwd {a:.3f} ms ({gf/a:.0f} GF/s, {gb/a*1e3:.0f} GB/s) dgrad {b:.3f} wgrad {c:.3f} ms ({gf/c:.0f} GF/s)")
