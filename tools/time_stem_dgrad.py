"""Stem data gradient (7^3, 64 -> 1 channel) standalone: python tools/time_stem_dgrad.py [B D H W]  (default: the headline's 4 512 128 128)"""
import sys, time; sys.path.insert(0, '.')
import torch, ctypes as C
from hiddenpose_amd import _lib, hip_ops as ops
L = _lib.lib()
B, D, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) > 4 else (4, 512, 128, 128)
dz = torch.randn(B, D, H, W, 64, device='cuda')
w = torch.randn(64, 1, 7, 7, 7, device='cuda')
x = torch.zeros(B, D, H, W, 1, device='cuda')
desc = ops._desc(x, 64, 7, 1, 3, False)
_, wd = ops._pack(desc, w, False, True)
dx = torch.empty_like(x)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        L.hp_conv3d_backward_data(C.byref(desc), dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, ops._stream(x))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"stem dgrad B={B} {D}x{H}x{W}: {dt*1e3:.2f} ms  -> {2*B*D*H*W*343*64/dt/1e12:.1f} TFLOP/s")
