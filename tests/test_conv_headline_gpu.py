"""Parity evidence under the headline number's own shape (BASELINE configs[1]: batch 4, 128 x 128 x 512, fp32).

Every convolution of posenet3d_50 (models/posenet3d_50.py:59-153,176-184: the 33 distinct shapes of
tools/time_conv_layers.py) runs ONCE at the bench geometry through the C ABI -- forward (with the BatchNorm statistics
epilogue), data gradient (plain, with an addend, with the masked addend of the identity shortcut, in place for the
strided 1^3 shortcuts) and weight gradient -- and is spot-checked against float64 dot products:

* forward / data gradient: 256 output rows x every output channel (>= 6144 outputs per call).  The input patches of the
  sampled rows are gathered on the device with plain torch indexing, copied to the host and reduced there in NumPy
  float64 (each output is a K <= 64 * 2048 reduction).
* weight gradient, sparse: the tensor indexed by the reduction's row space (dy; x for a transposed convolution) is
  zero except for ~40 rows, so dW is a sum of ~40 outer products that the host evaluates in float64 for the FULL
  dW tensor -- any row that is read from the wrong address, dropped or counted twice shows in every element it touches.
* weight gradient, dense: random operands; 16 x 16 channel blocks x up to 8 taps are reduced over all M rows in
  float64 on the device (torch.matmul -- an M-long reduction per output is not host work at M = 4.2e6 .. 3.4e7).

Bars (relative L2 over the sampled outputs / largest error over the rms of the expected values): 1e-5 / 1e-4 forward and
data gradient (measured <= 2.2e-6 / 2.5e-5: fp32 rounding of a K <= 131072 sum; ONE dropped product of a K = 16384 sum is
8e-3 of the rms), 2e-5 / 1e-4 dense weight gradient (measured <= 3.2e-6 / 1.2e-5; a dropped 32-row step of an M = 4.2e6
reduction is 2.7e-3), 5e-7 / 1e-5 sparse-row weight gradient (measured 8e-8 / 1.4e-6).

Rows are sampled deliberately: the first and last rows of the tensor (last M tile), the rows either side of every
2^31-byte mark of every tensor the call touches (layer 1's tensors are 4.29 GB: the kernels address rows with 32-bit
offsets relative to a per-block base and decide on host-side span tests whether they may), and -- for the weight
gradient -- the first and last row of several `msplit` chunks as hp_conv3d_backward_weight_split reports them.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import pytest
import torch

from hiddenpose_amd import _lib
from hiddenpose_amd import hip_ops as ops

pytestmark = pytest.mark.gpu

B, T, N = 4, 512, 128
ROWS = 256


def _layers():
    out = [("stem", 1, 64, 7, 1, 3, False, (T, N, N))]
    d = (T // 2, N // 2, N // 2)
    inpl = 64
    for li, (nb, pl) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
        for bi in range(2):  # blocks 2.. repeat block 1's shapes
            s = 2 if (bi == 0 and li > 0) else 1
            do = tuple(v // s for v in d)
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


LAYERS = _layers()
assert len(LAYERS) == 33


# ---------------------------------------------------------------- gathers (plain torch indexing on the device)
def _unravel(rows, dims):
    d0, d1, d2 = dims
    return rows // (d0 * d1 * d2), (rows // (d1 * d2)) % d0, (rows // d2) % d1, rows % d2


def _gather(src, src_dims, rows, dst_dims, k, s, p, divisible):
    """(R, k^3, C): for every destination row (flat index over (B, *dst_dims)) and tap (kd, kh, kw) the source row at
    dst * s - p + tap  (divisible = False: what a convolution reads), or at (dst + p - tap) / s where that is an integer
    (divisible = True: what its transpose reads); zeros where the position falls outside the source volume."""
    dev = src.device
    b, *dc = _unravel(rows, dst_dims)
    taps = torch.arange(k, device=dev)
    pos, ok = [], []
    for a in range(3):
        if divisible:
            num = dc[a][:, None] + p - taps[None, :]
            q = torch.div(num, s, rounding_mode="floor")
            v = (num - q * s == 0) & (q >= 0) & (q < src_dims[a])
        else:
            q = dc[a][:, None] * s - p + taps[None, :]
            v = (q >= 0) & (q < src_dims[a])
        pos.append(q.clamp(0, src_dims[a] - 1))
        ok.append(v)
    flat = ((b[:, None, None, None] * src_dims[0] + pos[0][:, :, None, None]) * src_dims[1] + pos[1][:, None, :, None]) * src_dims[2] \
        + pos[2][:, None, None, :]
    valid = ok[0][:, :, None, None] & ok[1][:, None, :, None] & ok[2][:, None, None, :]
    c = src.shape[-1]
    got = src.reshape(-1, c)[flat.reshape(-1)].reshape(rows.numel(), k ** 3, c)
    return got * valid.reshape(rows.numel(), k ** 3, 1).to(got.dtype)


def _marks(nbytes_per_row, M):
    """Rows either side of every 2^31-byte mark of a tensor with M rows of `nbytes_per_row` bytes."""
    out = []
    mark = 1 << 31
    while mark < M * nbytes_per_row:
        r = mark // nbytes_per_row
        out += [r - 1, r, r + 1]
        mark += 1 << 31
    return [r for r in out if 0 <= r < M]


def _sample_rows(M, special, seed, n=ROWS):
    g = torch.Generator().manual_seed(seed)
    special = sorted({int(r) for r in special if 0 <= r < M})[:n - 32]
    edge = [0, 1, 2, 31, 32, 127, 128, M - 129, M - 128, M - 33, M - 32, M - 2, M - 1]
    rows = sorted(set(special) | {r for r in edge if 0 <= r < M})
    extra = torch.randint(0, M, (n - len(rows),), generator=g).tolist()
    return torch.tensor(rows + extra, dtype=torch.int64)


def _to_rowspace(rows, from_dims, to_dims):
    """Flat rows of a (B, *from_dims) grid mapped to the rows of the (B, *to_dims) grid at the same place (stride 2: the
    halved / doubled coordinate)."""
    if not rows:
        return []
    b, d, h, w = _unravel(torch.tensor(rows, dtype=torch.int64), from_dims)
    c = [(v * t) // f for v, t, f in zip((d, h, w), to_dims, from_dims)]
    return (((b * to_dims[0] + c[0]) * to_dims[1] + c[1]) * to_dims[2] + c[2]).tolist()


def _check(got, want, what, rel_bar, max_bar):
    got = np.asarray(got, np.float64)
    rms = float(np.sqrt(np.mean(want ** 2)))
    rel = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300))
    mx = float(np.abs(got - want).max() / max(rms, 1e-300))
    assert rel < rel_bar and mx < max_bar, f"{what}: rel-L2 {rel:.2e} (bar {rel_bar:.0e}), max |err| / rms {mx:.2e} (bar {max_bar:.0e})"
    return rel, mx


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_headline_geometry_convolution_vs_float64(layer, capsys):
    name, cin, cout, k, s, p, tr, din = layer
    L = _lib.lib()
    assert ops.get_conv_precision() == "fp32"
    dev = torch.device("cuda", 0)
    seed = 1000 + [l[0] for l in LAYERS].index(name)
    torch.manual_seed(seed)
    x = torch.randn(B, *din, cin, device=dev)
    w = torch.randn((cin, cout, k, k, k) if tr else (cout, cin, k, k, k), device=dev) * 0.05
    desc = ops._desc(x, cout, k, s, p, tr)
    dout = ops._out_dims(desc)
    Mi, Mo = B * din[0] * din[1] * din[2], B * dout[0] * dout[1] * dout[2]
    st = ops._stream(x)
    wf, wd = ops._pack(desc, w, True, True)
    desc.io = 0
    w64 = w.double().cpu().numpy().reshape(w.shape[0], w.shape[1], k ** 3)
    # packed-as-matrix forms: [tap][gathered channel][produced channel]
    if tr:
        w_fwd = np.ascontiguousarray(w64.transpose(2, 0, 1)).reshape(k ** 3 * cin, cout)     # Wt[ci, co, t]
        w_dgr = np.ascontiguousarray(w64.transpose(2, 1, 0)).reshape(k ** 3 * cout, cin)
    else:
        w_fwd = np.ascontiguousarray(w64.transpose(2, 1, 0)).reshape(k ** 3 * cin, cout)     # W[co, ci, t]
        w_dgr = np.ascontiguousarray(w64.transpose(2, 0, 1)).reshape(k ** 3 * cout, cin)
    report = []

    # ---------------------------------------------------------------- forward (+ statistics epilogue / bias)
    y = torch.empty(B, *dout, cout, device=dev)
    head = name == "head"
    bias = torch.randn(cout, device=dev) if head else None
    stats = None if head else torch.empty(_lib.STATS_SLOTS * 2 * cout, dtype=torch.float64, device=dev)
    _lib.check(L.hp_conv3d_forward(C.byref(desc), x.data_ptr(), wf.data_ptr(), _lib.ptr(bias), y.data_ptr(), _lib.ptr(stats), st),
               "hp_conv3d_forward")
    special = _marks(cout * 4, Mo) + _to_rowspace(_marks(cin * 4, Mi), din, dout)
    rows = _sample_rows(Mo, special, seed).to(dev)
    patches = _gather(x, din, rows, dout, k, 2 if tr else s, p, divisible=tr)
    want = patches.double().cpu().numpy().reshape(rows.numel(), -1) @ w_fwd
    if head:
        want = want + bias.double().cpu().numpy()[None, :]
    got = y.reshape(Mo, cout)[rows].cpu().numpy()
    report.append(("fwd",) + _check(got, want, f"{name} forward", 1e-5, 1e-4))
    if stats is not None:
        ym = y.reshape(B, -1, cout)
        s1 = sum(ym[b].sum(0, dtype=torch.float64) for b in range(B))
        s2 = sum((ym[b].double() ** 2).sum(0) for b in range(B))
        got_s = stats.view(_lib.STATS_SLOTS, 2 * cout).sum(0).cpu().numpy()     # the epilogue's partial vectors (HP_STATS_SLOTS)
        # tile sums are fp32 before they meet in fp64 atomics: |error of the sum| ~ 2^-24 sqrt(rows per tile) rms(y) per tile
        np.testing.assert_allclose(got_s[cout:], s2.cpu().numpy(), rtol=1e-6, err_msg=f"{name} statistics epilogue (sum of squares)")
        np.testing.assert_allclose(got_s[:cout], s1.cpu().numpy(), rtol=0, atol=1e-6 * float(s2.max().sqrt()),
                                   err_msg=f"{name} statistics epilogue (sum)")
        del ym, s1, s2
    del y, patches

    # ---------------------------------------------------------------- data gradient
    gy = torch.randn(B, *dout, cout, device=dev)
    dx = torch.empty_like(x)
    _lib.check(L.hp_conv3d_backward_data(C.byref(desc), gy.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, st), "hp_conv3d_backward_data")
    special = _marks(cin * 4, Mi) + _to_rowspace(_marks(cout * 4, Mo), dout, din)
    rows_i = _sample_rows(Mi, special, seed + 1).to(dev)
    patches = _gather(gy, dout, rows_i, din, k, 2 if tr else s, p, divisible=not tr)
    want = patches.double().cpu().numpy().reshape(rows_i.numel(), -1) @ w_dgr
    got = dx.reshape(Mi, cin)[rows_i].cpu().numpy()
    report.append(("dgrad",) + _check(got, want, f"{name} data gradient", 1e-5, 1e-4))
    del patches
    # the fused variants the model uses: the second contribution to the same tensor summed in the epilogue
    stem = name == "stem"
    strided_1 = (not tr) and k == 1 and s == 2
    masked = (not tr) and (not stem) and s == 1 and cin > 32 and name.endswith(".1.conv1")
    addend = torch.randn_like(x)
    dx2 = addend.clone() if strided_1 else torch.empty_like(x)
    if masked:
        mask = torch.randint(0, 16, (Mi * cin // 4,), dtype=torch.uint8, device=dev)
        _lib.check(L.hp_conv3d_backward_data_masked(C.byref(desc), gy.data_ptr(), wd.data_ptr(), dx2.data_ptr(), addend.data_ptr(),
                                                    mask.data_ptr(), st), "hp_conv3d_backward_data_masked")
        bits = (mask.int()[:, None] >> torch.arange(4, device=dev, dtype=torch.int32)[None, :]) & 1
        addend = addend * bits.reshape(addend.shape).to(addend.dtype)
        del bits, mask
    else:
        _lib.check(L.hp_conv3d_backward_data(C.byref(desc), gy.data_ptr(), wd.data_ptr(), dx2.data_ptr(),
                                             dx2.data_ptr() if strided_1 else addend.data_ptr(), st), "hp_conv3d_backward_data")
    # (a + b) in fp32 against a, b given: one rounding of the sum
    err = float((dx2 - (dx + addend)).abs().max())
    scale = float(dx.abs().max() + addend.abs().max())
    # (the stem's data gradient meets in fp32 atomics: two runs differ by their summation order)
    assert err <= (1e-5 if stem else 2.5e-7) * scale, f"{name} data gradient + {'masked ' if masked else 'in-place ' if strided_1 else ''}addend: {err:.3e} of {scale:.3e}"
    del dx, dx2, addend

    # ---------------------------------------------------------------- weight gradient
    n_packed = int(L.hp_conv3d_packed_weight_elems(C.byref(desc)))
    dwp = torch.empty(n_packed, device=dev)

    def wgrad(xx, gg):
        _lib.check(L.hp_conv3d_backward_weight(C.byref(desc), xx.data_ptr(), gg.data_ptr(), dwp.data_ptr(), st), "hp_conv3d_backward_weight")
        if ops._same_as_packed(desc):
            return dwp.view_as(w).clone()
        dw = torch.empty_like(w)
        _lib.check(L.hp_conv3d_unpack_wgrad(C.byref(desc), dwp.data_ptr(), dw.data_ptr(), st), "hp_conv3d_unpack_wgrad")
        return dw

    # dense: 16 x 16 channel blocks, <= 8 taps, reduced over all M rows in float64 on the device
    dw = wgrad(x, gy)
    g = torch.Generator().manual_seed(seed + 2)
    co_s = torch.randperm(cout, generator=g)[:16].sort().values.to(dev)
    ci_s = torch.randperm(cin, generator=g)[:16].sort().values.to(dev)
    tap_s = torch.randperm(k ** 3, generator=g)[:8].sort().values.tolist()
    dense_got, dense_want = [], []
    gyc = gy.reshape(B, *dout, cout)[..., co_s].double()            # (B, Do, Ho, Wo, 16)
    xc = x.reshape(B, *din, cin)[..., ci_s].double()
    for t in tap_s:
        kd, kh, kw = t // (k * k), (t // k) % k, t % k
        if tr:   # dWt[ci, co, t] = sum_i x[i, ci] dy[2 i - 1 + tap, co]
            pad = torch.nn.functional.pad(gyc, (0, 0, p, p, p, p, p, p))
            sh = pad[:, kd:kd + 2 * din[0]:2, kh:kh + 2 * din[1]:2, kw:kw + 2 * din[2]:2, :]
            blk = xc.reshape(Mi, -1).T @ sh.reshape(Mi, -1)                                      # (ci, co)
            dense_got.append(dw[ci_s][:, co_s][:, :, kd, kh, kw].double())
        else:    # dW[co, ci, t] = sum_o dy[o, co] x[s o - p + tap, ci]
            pad = torch.nn.functional.pad(xc, (0, 0, p, p, p, p, p, p))
            sh = pad[:, kd:kd + s * dout[0]:s, kh:kh + s * dout[1]:s, kw:kw + s * dout[2]:s, :]
            blk = gyc.reshape(Mo, -1).T @ sh.reshape(Mo, -1)                                     # (co, ci)
            dense_got.append(dw[co_s][:, ci_s][:, :, kd, kh, kw].double())
        dense_want.append(blk)
        del pad, sh
    report.append(("wgrad",) + _check(torch.stack(dense_got).cpu().numpy(), torch.stack(dense_want).cpu().numpy(),
                                      f"{name} weight gradient (dense)", 2e-5, 1e-4))
    del gyc, xc, dw, dense_got, dense_want

    # sparse: ~40 live rows at the seams -> the FULL dW in float64 on the host
    msplit, chunk = C.c_long(0), C.c_long(0)
    _lib.check(L.hp_conv3d_backward_weight_split(C.byref(desc), C.byref(msplit), C.byref(chunk)), "hp_conv3d_backward_weight_split")
    msplit, chunk = msplit.value, chunk.value
    Mr = Mi if tr else Mo                                   # the reduction's row space
    seams = []
    for c in sorted({1, 2, msplit // 2, msplit - 1, msplit}):
        if 0 < c and c * chunk - 1 < Mr:
            seams += [c * chunk - 1, c * chunk]
    live = sorted({r for r in ([0, Mr - 1, Mr - 2, Mr - 33] + seams + _marks((cin if tr else cout) * 4, Mr)
                               + (_to_rowspace(_marks(cout * 4, Mo), dout, din) if tr else _to_rowspace(_marks(cin * 4, Mi), din, dout))
                               + torch.randint(0, Mr, (8,), generator=g).tolist()) if 0 <= r < Mr})[:48]
    live_t = torch.tensor(live, dtype=torch.int64, device=dev)
    if tr:
        xs = torch.zeros_like(x)
        xs.reshape(Mi, cin)[live_t] = x.reshape(Mi, cin)[live_t]
        dw = wgrad(xs, gy)
        rows_v = xs.reshape(Mi, cin)[live_t].double().cpu().numpy()                               # (S, ci)
        pt = _gather(gy, dout, live_t, din, k, 2, p, divisible=False).double().cpu().numpy()       # (S, taps, co)
        want = (rows_v.T @ pt.reshape(len(live), -1)).reshape(cin, k ** 3, cout).transpose(0, 2, 1)   # Wt[ci, co, t]
        del xs
    else:
        gs = torch.zeros_like(gy)
        gs.reshape(Mo, cout)[live_t] = gy.reshape(Mo, cout)[live_t]
        dw = wgrad(x, gs)
        rows_v = gs.reshape(Mo, cout)[live_t].double().cpu().numpy()                              # (S, co)
        pt = _gather(x, din, live_t, dout, k, s, p, divisible=False).double().cpu().numpy()        # (S, taps, ci)
        want = (rows_v.T @ pt.reshape(len(live), -1)).reshape(cout, k ** 3, cin).transpose(0, 2, 1)   # W[co, ci, t]
        del gs
    got = dw.cpu().numpy().reshape(want.shape)
    report.append((f"wgrad-sparse[{len(live)} rows, msplit {msplit} x {chunk}]",) + _check(got, want, f"{name} weight gradient (sparse rows)", 5e-7, 1e-5))
    with capsys.disabled():
        print(f"\n[{name:12s} {cin:4d}->{cout:4d} k{k} s{s}{' T' if tr else ''} in{din}] " +
              "  ".join(f"{w_} rel {a:.1e} max {b:.1e}" for w_, a, b in report), end="")


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_headline_geometry_convolution_bf16_storage_vs_float64(layer, capsys):
    """The same 33 shapes at the same geometry in BASELINE configs[2]'s per-GPU arithmetic (`bf16s`: bf16 tensors in HBM, bf16
    matrix cores, fp32 accumulation; the stem's tensors stay fp32): forward with the statistics epilogue, data gradient and weight
    gradient (dense block and sparse rows) through the C ABI exactly as hip_ops calls them, against float64 evaluations on the
    bf16-ROUNDED operands.  Products of bf16 values are exact in fp32, so fp32 results (weight gradients, the stem, the head's
    output) keep the fp32 bars; bf16 outputs add their own rounding (2^-9 per element: measured rel-L2 1.7e-3 on every layer;
    largest error over rms 1e-2 .. 3.5e-2 -- the strided shortcut gradients are 7/8 zeros, so their largest elements are many rms:
    bars 5e-3 / 6e-2).  This is what puts the bf16 tile loaders (`k_igemm<..., XH, GL>`), the 2-byte row offsets of `k_wgrad_blh` and its
    host-side span tests under the 2.1 GB tensors of layer 1."""
    name, cin, cout, k, s, p, tr, din = layer
    L = _lib.lib()
    dev = torch.device("cuda", 0)
    seed = 3000 + [l[0] for l in LAYERS].index(name)
    torch.manual_seed(seed)
    prev = ops.set_conv_precision("bf16s")
    try:
        half = cin > 1                                     # the stem reads its single-channel input as fp32
        head = name == "head"
        dt = torch.bfloat16 if half else torch.float32
        x = torch.randn(B, *din, cin, device=dev).to(dt)
        w = (torch.randn((cin, cout, k, k, k) if tr else (cout, cin, k, k, k), device=dev) * 0.05)
        desc = ops._desc(x, cout, k, s, p, tr)
        dout = ops._out_dims(desc)
        Mi, Mo = B * din[0] * din[1] * din[2], B * dout[0] * dout[1] * dout[2]
        st = ops._stream(x)
        whf, whd = ops._w_half(desc, half, cin), ops._w_half(desc, half, cout)
        wf, _ = ops._pack(desc, w, True, False, half=whf)
        _, wd = ops._pack(desc, w, False, True, half=whd)
        wr = w.bfloat16().double().cpu().numpy().reshape(w.shape[0], w.shape[1], k ** 3)     # what the matrix cores multiply
        if tr:
            w_fwd = np.ascontiguousarray(wr.transpose(2, 0, 1)).reshape(k ** 3 * cin, cout)
            w_dgr = np.ascontiguousarray(wr.transpose(2, 1, 0)).reshape(k ** 3 * cout, cin)
        else:
            w_fwd = np.ascontiguousarray(wr.transpose(2, 1, 0)).reshape(k ** 3 * cin, cout)
            w_dgr = np.ascontiguousarray(wr.transpose(2, 0, 1)).reshape(k ** 3 * cout, cin)
        xr = x.bfloat16()                                   # (identity for the bf16 tensors; the stem's fp32 input as rounded on load)
        eb = 2 if half else 4
        report = []

        # ---------------------------------------------------------------- forward
        y_half = half and not head                          # the head's output stays fp32 (hip_ops._ConvBiasToNCDHW)
        y = torch.empty(B, *dout, cout, device=dev, dtype=torch.bfloat16 if y_half else torch.float32)
        bias = torch.randn(cout, device=dev) if head else None
        stats = None if head else torch.empty(_lib.STATS_SLOTS * 2 * cout, dtype=torch.float64, device=dev)
        desc.io = (ops.HP_IO_X if half else 0) | (ops.HP_IO_Y if y_half else 0) | (ops.HP_IO_W if whf else 0)
        _lib.check(L.hp_conv3d_forward(C.byref(desc), x.data_ptr(), wf.data_ptr(), _lib.ptr(bias), y.data_ptr(), _lib.ptr(stats), st),
                   "hp_conv3d_forward")
        special = _marks(cout * (2 if y_half else 4), Mo) + _to_rowspace(_marks(cin * eb, Mi), din, dout)
        rows = _sample_rows(Mo, special, seed).to(dev)
        want = _gather(xr, din, rows, dout, k, 2 if tr else s, p, divisible=tr).double().cpu().numpy().reshape(rows.numel(), -1) @ w_fwd
        if head:
            want = want + bias.double().cpu().numpy()[None, :]
        got = y.reshape(Mo, cout)[rows].float().cpu().numpy()
        report.append(("fwd",) + (_check(got, want, f"{name} forward", 5e-3, 6e-2) if y_half else _check(got, want, f"{name} forward", 1e-5, 1e-4)))
        if stats is not None:                               # from the fp32 accumulators, i.e. before y was rounded
            s2 = sum((y.reshape(B, -1, cout)[b].double() ** 2).sum(0) for b in range(B)).cpu().numpy()
            np.testing.assert_allclose(stats.view(_lib.STATS_SLOTS, 2 * cout).sum(0).cpu().numpy()[cout:], s2, rtol=3e-3 if y_half else 1e-6)
        del y

        # ---------------------------------------------------------------- data gradient
        gy = torch.randn(B, *dout, cout, device=dev).to(dt)
        dx = torch.empty_like(x)
        desc.io = (ops.HP_IO_X | ops.HP_IO_DX | ops.HP_IO_DY if half else 0) | (ops.HP_IO_W if whd else 0)
        _lib.check(L.hp_conv3d_backward_data(C.byref(desc), gy.data_ptr(), wd.data_ptr(), dx.data_ptr(), None, st), "hp_conv3d_backward_data")
        gr = gy.bfloat16()
        special = _marks(cin * eb, Mi) + _to_rowspace(_marks(cout * eb, Mo), dout, din)
        rows_i = _sample_rows(Mi, special, seed + 1).to(dev)
        want = _gather(gr, dout, rows_i, din, k, 2 if tr else s, p, divisible=not tr).double().cpu().numpy().reshape(rows_i.numel(), -1) @ w_dgr
        got = dx.reshape(Mi, cin)[rows_i].float().cpu().numpy()
        report.append(("dgrad",) + (_check(got, want, f"{name} data gradient", 5e-3, 6e-2) if half else _check(got, want, f"{name} data gradient", 1e-5, 1e-4)))
        del dx

        # ---------------------------------------------------------------- weight gradient (fp32 out: exact products of bf16 operands)
        desc.io = ops.HP_IO_X | ops.HP_IO_DX | ops.HP_IO_DY if half else 0
        dwp = torch.empty(int(L.hp_conv3d_packed_weight_elems(C.byref(desc))), device=dev)

        def wgrad(xx, gg):
            _lib.check(L.hp_conv3d_backward_weight(C.byref(desc), xx.data_ptr(), gg.data_ptr(), dwp.data_ptr(), st), "hp_conv3d_backward_weight")
            if ops._same_as_packed(desc):
                return dwp.view_as(w).clone()
            dw = torch.empty_like(w)
            _lib.check(L.hp_conv3d_unpack_wgrad(C.byref(desc), dwp.data_ptr(), dw.data_ptr(), st), "hp_conv3d_unpack_wgrad")
            return dw

        dw = wgrad(x, gy)
        g = torch.Generator().manual_seed(seed + 2)
        co_s = torch.randperm(cout, generator=g)[:16].sort().values.to(dev)
        ci_s = torch.randperm(cin, generator=g)[:16].sort().values.to(dev)
        tap_s = torch.randperm(k ** 3, generator=g)[:4].sort().values.tolist()
        dense_got, dense_want = [], []
        gyc = gr.reshape(B, *dout, cout)[..., co_s].double()
        xc = xr.reshape(B, *din, cin)[..., ci_s].double()
        for t in tap_s:
            kd, kh, kw = t // (k * k), (t // k) % k, t % k
            if tr:
                pad = torch.nn.functional.pad(gyc, (0, 0, p, p, p, p, p, p))
                sh = pad[:, kd:kd + 2 * din[0]:2, kh:kh + 2 * din[1]:2, kw:kw + 2 * din[2]:2, :]
                dense_want.append(xc.reshape(Mi, -1).T @ sh.reshape(Mi, -1))
                dense_got.append(dw[ci_s][:, co_s][:, :, kd, kh, kw].double())
            else:
                pad = torch.nn.functional.pad(xc, (0, 0, p, p, p, p, p, p))
                sh = pad[:, kd:kd + s * dout[0]:s, kh:kh + s * dout[1]:s, kw:kw + s * dout[2]:s, :]
                dense_want.append(gyc.reshape(Mo, -1).T @ sh.reshape(Mo, -1))
                dense_got.append(dw[co_s][:, ci_s][:, :, kd, kh, kw].double())
            del pad, sh
        report.append(("wgrad",) + _check(torch.stack(dense_got).cpu().numpy(), torch.stack(dense_want).cpu().numpy(),
                                          f"{name} weight gradient (dense)", 2e-5, 1e-4))
        del gyc, xc, dw
        msplit, chunk = C.c_long(0), C.c_long(0)
        _lib.check(L.hp_conv3d_backward_weight_split(C.byref(desc), C.byref(msplit), C.byref(chunk)), "hp_conv3d_backward_weight_split")
        msplit, chunk = msplit.value, chunk.value
        Mr = Mi if tr else Mo
        seams = []
        for c in sorted({1, 2, msplit // 2, msplit - 1, msplit}):
            if 0 < c and c * chunk - 1 < Mr:
                seams += [c * chunk - 1, c * chunk]
        live = sorted({r for r in ([0, Mr - 1, Mr - 2, Mr - 33] + seams + _marks((cin if tr else cout) * eb, Mr)
                                   + torch.randint(0, Mr, (8,), generator=g).tolist()) if 0 <= r < Mr})[:48]
        live_t = torch.tensor(live, dtype=torch.int64, device=dev)
        if tr:
            xs = torch.zeros_like(x)
            xs.reshape(Mi, cin)[live_t] = x.reshape(Mi, cin)[live_t]
            dw = wgrad(xs, gy)
            rows_v = xs.reshape(Mi, cin)[live_t].bfloat16().double().cpu().numpy()
            pt = _gather(gr, dout, live_t, din, k, 2, p, divisible=False).double().cpu().numpy()
            want = (rows_v.T @ pt.reshape(len(live), -1)).reshape(cin, k ** 3, cout).transpose(0, 2, 1)
            del xs
        else:
            gs = torch.zeros_like(gy)
            gs.reshape(Mo, cout)[live_t] = gy.reshape(Mo, cout)[live_t]
            dw = wgrad(x, gs)
            rows_v = gs.reshape(Mo, cout)[live_t].bfloat16().double().cpu().numpy()     # (the stem's fp32 dz is rounded on load)
            pt = _gather(xr, din, live_t, dout, k, s, p, divisible=False).double().cpu().numpy()
            want = (rows_v.T @ pt.reshape(len(live), -1)).reshape(cout, k ** 3, cin).transpose(0, 2, 1)
            del gs
        report.append((f"wgrad-sparse[{len(live)} rows]",) + _check(dw.cpu().numpy().reshape(want.shape), want,
                                                                     f"{name} weight gradient (sparse rows)", 1e-6, 1e-5))
    finally:
        ops.set_conv_precision(prev)
    with capsys.disabled():
        print(f"\n[bf16s {name:12s} {cin:4d}->{cout:4d} k{k} s{s}{' T' if tr else ''}] " +
              "  ".join(f"{w_} rel {a:.1e} max {b:.1e}" for w_, a, b in report), end="")
