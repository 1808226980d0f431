"""Per-layer timing of the thin-channel 3^3 convolutions at the U-Net / FeatureExtraction shapes of a T x N x N volume:
    python tools/time_dconv_layers.py [T N]      (default 1024 256; HP_TIME_DCONV_PRECISION=bf16: forward / data gradient
    on the bf16 matrix cores)"""
import sys; sys.path.insert(0, '.')
import os
import torch
from hiddenpose_amd import _lib
T, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 256)
PREC = 1 if os.environ.get('HP_TIME_DCONV_PRECISION', 'fp32') == 'bf16' else 0
L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
layers = [("FE 1->1 replicate", 1, 1, 1, 1), ("FE 1->1 zero (box)", 1, 1, 1, 0), ("conv 1->4", 1, 4, 1, 0), ("conv/dec4 4->4", 4, 4, 1, 0),
          ("dec4 8->4", 8, 4, 1, 0), ("enc1 4->8", 4, 8, 2, 0), ("enc1 8->8", 8, 8, 2, 0), ("dec3 16->4", 16, 4, 2, 0),
          ("enc2 8->16", 8, 16, 4, 0), ("enc2 16->16", 16, 16, 4, 0), ("dec2 32->8", 32, 8, 4, 0), ("enc3 32->32", 32, 32, 8, 0),
          ("dec1 64->16", 64, 16, 8, 0)]
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
a = torch.randn(1024 * 256 * 256, device='cuda'); b_ = torch.empty_like(a)
print('calibration: torch copy of 268 MB: %.3f ms' % timeit(lambda: b_.copy_(a)), flush=True)
del a, b_
tot = [0, 0, 0]
for name, cin, cout, ds, rep in layers:
    D, H, W = T // ds, N // ds, N // ds
    x = torch.randn(1, cin, D, H, W, device='cuda'); g = torch.randn(1, cout, D, H, W, device='cuda'); w = torch.randn(cout, cin, 3, 3, 3, device='cuda')
    y = torch.empty_like(g); gx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty(cout, device='cuda')
    ws = torch.empty(int(L.hp_dconv3_backward_data_workspace_bytes(1, cin, D, H, W, rep)) // 4 + 1, device='cuda')
    wsw = torch.empty(int(L.hp_dconv3_backward_weight_workspace_bytes(1, cin, cout, D, H, W)) // 4, device='cuda')
    f = timeit(lambda: L.hp_dconv3_forward_fused_p(x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), None, 1, cin, cout, D, H, W, rep, 1.0, PREC, st))
    d = timeit(lambda: L.hp_dconv3_backward_data_p(g.data_ptr(), w.data_ptr(), gx.data_ptr(), 1, cin, cout, D, H, W, rep, PREC, ws.data_ptr(), st))
    wg = timeit(lambda: L.hp_dconv3_backward_weight_p(x.data_ptr(), g.data_ptr(), dw.data_ptr(), db.data_ptr(), 1, cin, cout, D, H, W, rep, PREC, wsw.data_ptr(), st))
    V = D * H * W
    gf = 2 * 27 * cin * cout * V / 1e9
    gb = 4 * V * (cin + cout) / 1e9
    print(f"{name:22s} {D}x{H}x{W}: fwd {f:6.3f} ms ({gf/f:6.1f} GF/ms... {gf/f:5.1f} TF/s, {gb/f*1e3:5.0f} GB/s)  dgrad {d:6.3f}  wgrad {wg:6.3f}", flush=True)
    tot[0] += f; tot[1] += d; tot[2] += wg
    del x, g, y, gx, ws, wsw
print("sum (one of each): fwd %.2f dgrad %.2f wgrad %.2f" % tuple(tot))
