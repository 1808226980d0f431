"""Stem BatchNorm + ReLU + MaxPool3d(3,2,1) forward / backward standalone (dev tool): python tools/dbg/time_stem_pool.py [B D H W]"""
import sys; sys.path.insert(0, '.')
import torch
from hiddenpose_amd import _lib
L = _lib.lib()
B, D, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (4, 512, 128, 128)
C = 64
z = torch.randn(B, D, H, W, C, device='cuda')
p = torch.empty(B, D // 2, H // 2, W // 2, C, device='cuda')
mean = torch.zeros(C, device='cuda'); rstd = torch.ones(C, device='cuda'); gamma = torch.randn(C, device='cuda'); beta = torch.randn(C, device='cuda')
ws = torch.empty(int(L.hp_stem_bn_pool_workspace_bytes(C)) // 4 + 4, device='cuda')
st = torch.cuda.current_stream().cuda_stream
def fwd():
    _lib.check(L.hp_stem_bn_relu_pool_forward(z.data_ptr(), p.data_ptr(), B, D, H, W, C, mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(), st), "fwd")
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
t = timeit(fwd)
gb = (z.numel() + p.numel()) * 4 / 1e9
print(f"stem bn+relu+pool fwd {B}x{D}x{H}x{W}: {t:.3f} ms  {gb / t:.2f} TB/s (z once + pooled)   checksum {float(p.double().sum()):.6e}")
dp = torch.randn_like(p); dz = torch.empty_like(z)
dgamma = torch.empty(C, device='cuda'); dbeta = torch.empty(C, device='cuda')
def bwd():
    _lib.check(L.hp_stem_bn_relu_pool_backward(z.data_ptr(), p.data_ptr(), dp.data_ptr(), dz.data_ptr(), B, D, H, W, C, mean.data_ptr(), rstd.data_ptr(),
                                               gamma.data_ptr(), beta.data_ptr(), 1, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), st), "bwd")
fwd()
t = timeit(bwd)
print(f"stem bn+relu+pool bwd (reduce + apply): {t:.3f} ms   checksum {float(dz.double().abs().sum()):.6e} {float(dgamma.double().sum()):.6e}")
