"""Per-layer two-sided bound of the posenet3d_50 convolutions in the bf16-storage mode (BASELINE configs[2]'s per-GPU share):
for each of the 33 distinct shapes x {forward, data gradient, weight gradient} the measured time of the kernel through the C ABI
against max(FLOP / MFMA peak, algorithmic bytes / HBM rate) -- which of the two bounds the layer, and how far the kernel is from it.
    python tools/bound_table.py [T N B] [precision] > profiles/roundN_bound_table_<precision>.txt
Peaks: dense bf16 MFMA 2.5 PFLOP/s, fp32 MFMA 157.3 TFLOP/s, HBM 6.3 TB/s achievable (8 TB/s peak; MI355X_MICROARCH.md)."""
import ctypes as C
import sys
import time

sys.path.insert(0, ".")
import torch

from hiddenpose_amd import _lib
from hiddenpose_amd import hip_ops as ops

L = _lib.lib()
T, N, B = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (512, 128, 4)))
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16s"
ops.set_conv_precision(prec)
HALF = prec == "bf16s"
MFMA = 157.3e12 if prec == "fp32" else 2.5e15
HBM = 6.3e12
COUNT = {"stem": 1, "deconv0": 1, "deconv1": 1, "deconv2": 1, "head": 1}   # launches per step of each distinct shape


def layers():
    out = [("stem", 1, 64, 7, 1, 3, False, (T, N, N), 1)]
    d = (T // 2, N // 2, N // 2)
    inpl = 64
    for li, (nb, pl) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
        for bi in range(2):
            s = 2 if (bi == 0 and li > 0) else 1
            do = tuple(v // s for v in d)
            rep = 1 if bi == 0 else nb - 1          # blocks 1.. share block 1's shapes
            out.append((f"l{li+1}.{bi}.conv1", inpl, pl, 1, 1, 0, False, d, rep))
            out.append((f"l{li+1}.{bi}.conv2", pl, pl, 3, s, 1, False, d, rep))
            out.append((f"l{li+1}.{bi}.conv3", pl, pl * 4, 1, 1, 0, False, do, rep))
            if bi == 0:
                out.append((f"l{li+1}.{bi}.down", inpl, pl * 4, 1, s, 0, False, d, 1))
            inpl, d = pl * 4, do
    cin = 2048
    for i in range(3):
        out.append((f"deconv{i}", cin, 256, 4, 2, 1, True, d, 1))
        cin, d = 256, tuple(2 * v for v in d)
    out.append(("head", 256, 24, 1, 1, 0, False, d, 1))
    return out


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print(f"# T={T} N={N} B={B} precision={prec}: time vs max(FLOP / {MFMA/1e12:.0f} TF, bytes / {HBM/1e12:.1f} TB/s); 'x' = launches of this shape per step")
print(f"# {'layer':13s} {'shape':22s} {'x':>2s} | " + " | ".join(f"{d:>5s}  ms    bound  ms   of-bound  by" for d in ("fwd", "dgrad", "wgrad")))
tot = {"fwd": [0.0, 0.0], "dgrad": [0.0, 0.0], "wgrad": [0.0, 0.0]}
for name, cin, cout, k, s, p, tr, d, rep in layers():
    half = HALF and cin > 1
    eb = 2 if half else 4
    dt = torch.bfloat16 if half else torch.float32
    x = torch.randn(B, *d, cin, device="cuda").to(dt)
    w = torch.randn((cin, cout, k, k, k) if tr else (cout, cin, k, k, k), device="cuda") * 0.05
    desc = ops._desc(x, cout, k, s, p, tr)
    do = ops._out_dims(desc)
    y = torch.empty(B, *do, cout, device="cuda", dtype=dt)
    gy = torch.randn(B, *do, cout, device="cuda").to(dt)
    st = ops._stream(x)
    whf, whd = ops._w_half(desc, half, cin), ops._w_half(desc, half, cout)
    wf, _ = ops._pack(desc, w, True, False, half=whf)
    _, wd = ops._pack(desc, w, False, True, half=whd)
    io_f = (ops.HP_IO_X | ops.HP_IO_Y if half else 0) | (ops.HP_IO_W if whf else 0)
    io_d = (ops.HP_IO_X | ops.HP_IO_DX | ops.HP_IO_DY if half else 0) | (ops.HP_IO_W if whd else 0)
    io_w = ops.HP_IO_X | ops.HP_IO_DX | ops.HP_IO_DY if half else 0
    mout = B * do[0] * do[1] * do[2]
    taps = 8 if tr else k ** 3
    flops = 2.0 * mout * taps * cin * cout
    xb, yb, wb = x.numel() * eb, y.numel() * (2 if HALF else 4), w.numel() * 4
    dx = torch.empty_like(x)
    dwp = torch.empty(int(L.hp_conv3d_packed_weight_elems(C.byref(desc))), device="cuda")
    stats = torch.empty(_lib.STATS_SLOTS * 2 * cout, dtype=torch.float64, device="cuda")
    row = []
    for what in ("fwd", "dgrad", "wgrad"):
        if what == "fwd":
            desc.io = io_f
            t = timeit(lambda: L.hp_conv3d_forward(C.byref(desc), x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st))
            nbytes = xb + yb + wb
        elif what == "dgrad":
            desc.io = io_d
            t = timeit(lambda: L.hp_conv3d_backward_data(C.byref(desc), gy.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, st))
            nbytes = xb + yb + wb
        else:
            desc.io = io_w
            t = timeit(lambda: L.hp_conv3d_backward_weight(C.byref(desc), x.data_ptr(), gy.data_ptr(), dwp.data_ptr(), st))
            nbytes = xb + yb + wb
        bm, bh = flops / MFMA, nbytes / HBM
        bound = max(bm, bh)
        row.append(f"{t*1e3:9.3f}  {bound*1e3:8.3f}  {bound/t:7.2f}  {'mfma' if bm >= bh else 'hbm':>4s}")
        tot[what][0] += t * rep
        tot[what][1] += bound * rep
    print(f"  {name:13s} {cin:4d}->{cout:4d} k{k}s{s}{'T' if tr else ' '} {str(d):>0s} {rep:2d} | " + " | ".join(row), flush=True)
    del x, y, gy, dx, dwp
print("# per step (x launches): " + "  ".join(f"{k}: {v[0]*1e3:.1f} ms measured, {v[1]*1e3:.1f} ms bound ({v[1]/v[0]:.2f})" for k, v in tot.items()))
