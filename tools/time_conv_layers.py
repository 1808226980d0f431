"""Per-layer timing of the posenet3d_50 convolution kernels through the C ABI (dev tool)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, ".")
import torch

from hiddenpose_amd import _lib
from hiddenpose_amd import hip_ops as ops

L = _lib.lib()
T, N, B = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (128, 128, 4)))
which = sys.argv[4] if len(sys.argv) > 4 else "fwd,dgrad,wgrad"
ops.set_conv_precision(sys.argv[5] if len(sys.argv) > 5 else "fp32")
ONLY = sys.argv[6].split(",") if len(sys.argv) > 6 else None   # layer-name prefixes
HALF = ops.get_conv_precision() == "bf16s"   # bf16 tensors in HBM (stem excepted), bf16 packed weights where the kernels take them


def layers():
    out = [("stem", 1, 64, 7, 1, 3, False, (T, N, N))]
    d = (T // 2, N // 2, N // 2)
    inpl = 64
    for li, (nb, pl) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
        for bi in range(nb):
            s = 2 if (bi == 0 and li > 0) else 1
            do = tuple(v // s for v in d)
            if bi < 2:  # blocks 2.. repeat block 1's shapes
                out.append((f"l{li+1}.{bi}.conv1", inpl, pl, 1, 1, 0, False, d))
                out.append((f"l{li+1}.{bi}.conv2", pl, pl, 3, s, 1, False, d))
                out.append((f"l{li+1}.{bi}.conv3", pl, pl * 4, 1, 1, 0, False, do))
                if bi == 0:
                    out.append((f"l{li+1}.{bi}.down", inpl, pl * 4, 1, s, 0, False, d))
            inpl, d = pl * 4, do
    cin = 2048
    for i in range(3):
        out.append((f"deconv{i}", cin, 256, 4, 2, 1, True, d))
        cin, d = 256, tuple(2 * v for v in d)
    out.append(("head", 256, 24, 1, 1, 0, False, d))
    return out


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print(f"T={T} N={N} B={B} precision={ops.get_conv_precision()}")
for name, cin, cout, k, s, p, tr, d in layers():
    if ONLY and not any(name.startswith(o) for o in ONLY):
        continue
    half = HALF and cin > 1
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
    flops = 2.0 * mout * (8 if tr else k ** 3) * cin * cout
    dx = torch.empty_like(x)
    dwp = torch.empty(int(L.hp_conv3d_packed_weight_elems(C.byref(desc))), device="cuda")
    res = []
    if "fwd" in which:
        desc.io = io_f
        t = timeit(lambda: L.hp_conv3d_forward(C.byref(desc), x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), None, st))
        res.append(f"fwd {t*1e3:8.3f} ms {flops/t/1e12:6.1f} TF")
        if os.environ.get("HP_TIME_STATS"):   # the same with the BatchNorm statistics epilogue (fp64 atomics per block and column)
            stats = torch.empty(_lib.STATS_SLOTS * 2 * cout, dtype=torch.float64, device="cuda")
            t = timeit(lambda: L.hp_conv3d_forward(C.byref(desc), x.data_ptr(), wf.data_ptr(), None, y.data_ptr(), stats.data_ptr(), st))
            res.append(f"fwd+stats {t*1e3:8.3f} ms {flops/t/1e12:6.1f} TF")
    if "dgrad" in which:
        desc.io = io_d
        t = timeit(lambda: L.hp_conv3d_backward_data(C.byref(desc), gy.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, st))
        res.append(f"dgrad {t*1e3:8.3f} ms {flops/t/1e12:6.1f} TF")
    if "wgrad" in which:
        desc.io = io_w
        t = timeit(lambda: L.hp_conv3d_backward_weight(C.byref(desc), x.data_ptr(), gy.data_ptr(), dwp.data_ptr(), st))
        res.append(f"wgrad {t*1e3:8.3f} ms {flops/t/1e12:6.1f} TF")
    print(f"{name:14s} {cin:5d}->{cout:5d} k{k} s{s} in{d} GF {flops/1e9:8.1f} | " + " | ".join(res), flush=True)
    del x, y, gy, dx, dwp
