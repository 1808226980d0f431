#!/usr/bin/env python3
"""Headline benchmark: NlosPose training step (forward + loss + backward + Adam) on
synthetic 128x128x512 transients, batch 4 per GPU, fp32 (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; for N > 1 gradients are averaged with bucketed RCCL all-reduce
overlapped with backward (hiddenpose_amd/data_parallel.py).  Per-GPU batch is fixed,
so scaling is weak.  Rank 0 prints ONE JSON line.  Started WITHOUT a launcher (no WORLD_SIZE
in the environment) and --gpus N > 1, this process starts the N ranks itself -- fresh child
processes under torch.distributed.run, before anything here has touched the GPU -- relays
their output and exits with their status.

Extra objects on the line:
  roofline     -- the dominant hand-written HIP kernel of the timed region, timed with HIP
                  events on its launch stream inside the library (hp_profile_*); achieved =
                  algorithmic bytes (or flops) per launch / mean launch duration.
  cpu_baseline -- the oracle (CPU restatement of the reference, oracle/) timed on this
                  host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (T, N, per-GPU batch)
    "t512": (512, 128, 4),     # BASELINE.json metric: "128x128x512 meas", configs[1] batch 4
    "native": (128, 128, 4),   # the reference's own training shape (train.py:77-86)
    "tiny": (32, 32, 2),
}

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense f32-input MFMA peak
MFMA_BF16_PEAK_TFLOPS = 2516.8  # same guide: dense bf16 MFMA = 16 x the f32-input rate (~2.5 PF)


def posenet_conv_flops(T: int, N: int, B: int) -> dict:
    """Algorithmic FLOPs (2 x MAC) per training step of every implicit-GEMM kernel family of
    posenet3d_50 (models/posenet3d_50.py:156-270), keyed by the profiling name used in
    csrc/conv_kernels.hip.  Forward, data gradient and weight gradient of a convolution all
    cost 2 * out_voxels * k^3 * Cin * Cout."""
    f = {"conv_igemm_stem": 0, "conv_igemm_k1": 0, "conv_igemm_k3": 0, "conv_igemm_deconv": 0, "conv_igemm_dgrad": 0,
         "conv_stem_dgrad": 0, "conv_wgrad": 0}
    vox = B * T * N * N
    stem = 2 * vox * 343 * 64
    f["conv_igemm_stem"] += stem
    f["conv_stem_dgrad"] += stem
    f["conv_wgrad"] += stem
    v = vox // 8  # after MaxPool3d(3,2,1)
    inpl = 64

    def add(name, flops):
        f[name] += flops
        f["conv_igemm_dgrad"] += flops
        f["conv_wgrad"] += flops

    for li, (nb, pl) in enumerate(zip((3, 4, 6, 3), (64, 128, 256, 512))):
        for bi in range(nb):
            stride = 2 if (bi == 0 and li > 0) else 1
            vin, vout = v, v // (stride ** 3)
            add("conv_igemm_k1", 2 * vin * inpl * pl)            # conv1 at the input resolution
            add("conv_igemm_k3", 2 * vout * 27 * pl * pl)        # conv2 carries the stride
            add("conv_igemm_k1", 2 * vout * pl * pl * 4)         # conv3
            if bi == 0:
                add("conv_igemm_k1", 2 * vout * inpl * pl * 4)   # shortcut type 'B'
            inpl, v = pl * 4, vout
    cin = 2048
    for _ in range(3):
        v *= 8
        add("conv_igemm_deconv", 2 * v * 8 * cin * 256)          # k4 s2: 8 taps reach each output voxel
        cin = 256
    add("conv_igemm_k1", 2 * v * 256 * 24)
    return f


def algorithmic_work(kernel: str, T: int, N: int, B: int):
    """(bound, unit amount per launch) for each hand-written kernel; derivations in DESIGN.md.
    V = T*N*N voxels, P = ceil(B/2) packed pairs, complex64 = 8 B."""
    V = T * N * N
    P = (B + 1) // 2
    table = {
        # real volumes in (4 B/voxel/sample) -> packed spectrum halves out (2 x 8 B/voxel/pair)
        "lct_axis_fwd_t": ("hbm", 4 * V * B + 16 * V * P),
        "lct_axis_fwd_h": ("hbm", 16 * V * P + 32 * V * P),
        # rows in + rows out + one read of the inverse PSF spectrum (8V complex)
        "lct_axis_mid_w": ("hbm", 32 * V * P * 2 + 64 * V),
        "lct_axis_inv_h": ("hbm", 32 * V * P + 16 * V * P),
        "lct_axis_inv_t": ("hbm", 16 * V * P + 4 * V * B),
    }
    return table.get(kernel)


def cpu_model_string() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def csrc_sha() -> str:
    """sha1 over the kernel sources: a PMC profile is only quoted on a bench line produced by the SAME kernels."""
    import glob
    import hashlib

    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "hiddenpose_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "hiddenpose_amd", "csrc", "*.cpp")) +
                    glob.glob(os.path.join(ROOT, "hiddenpose_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def cpu_baseline(threads: int, full: bool = False):
    """BASELINE.md section 3 protocol: the oracle (CPU restatement of the reference) on ONE 128x128x128 cube
    (= 1/4 of a 128x128x512 sample, `scaled_from`), 1 warm-up + 3 timed iterations, median; legs: forward only
    (eval, no_grad), the full train step (forward + losses + backward + Adam) and the LCT alone on one 512x128x128 volume.
    `value` is the train-step leg scaled to 128x128x512 samples/s, i.e. the same unit as the GPU line.  The default run
    is a bounded sample (about a minute of CPU work); `full` (--cpu-baseline-full) adds the remaining legs of the
    protocol: batch 4 at 128^3 (forward, train step) and the LCT at 1024x256x256."""
    import statistics

    from hiddenpose_amd import testing as hpt
    from hiddenpose_amd.config import make_cfg
    from hiddenpose_amd.NlosPose import NlosPose
    from oracle import nlospose_oracle as O

    torch.set_num_threads(threads)
    T, N, B = 128, 128, 1
    model = NlosPose(make_cfg(T, N))
    sd = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point and "running" not in k else v.clone())
          for k, v in model.state_dict().items()}
    k = O.LCTConstants(N, T, 0.04)
    meas = hpt.synthetic_meas(B, T, N)
    vol = hpt.synthetic_vol(B, T, N)
    joints = hpt.synthetic_joints(B, T // 2).reshape(B, -1)
    opt = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-3)

    def train():
        t0 = time.perf_counter()
        loss, *_ = O.train_loss(meas, vol, joints, sd, k)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return time.perf_counter() - t0

    def fwd():
        t0 = time.perf_counter()
        with torch.no_grad():
            O.train_loss(meas, vol, joints, sd, k)
        return time.perf_counter() - t0

    fwd()
    t_fwd = statistics.median(fwd() for _ in range(3))
    train()
    t_train = statistics.median(train() for _ in range(3))

    def lct_leg(T_, N_):
        kk = O.LCTConstants(N_, T_, 5.12 / T_)
        x = hpt.synthetic_meas(1, T_, N_)

        def run():
            t0 = time.perf_counter()
            with torch.no_grad():
                O.lct_forward(x, kk)
            return time.perf_counter() - t0

        run()
        return statistics.median(run() for _ in range(3))

    extra = {"lct_forward_s_512x128x128": round(lct_leg(512, 128), 3)}
    if full:
        B4 = 4
        meas4, vol4 = hpt.synthetic_meas(B4, T, N), hpt.synthetic_vol(B4, T, N)
        joints4 = hpt.synthetic_joints(B4, T // 2).reshape(B4, -1)

        def run4(grad):
            t0 = time.perf_counter()
            if grad:
                loss, *_ = O.train_loss(meas4, vol4, joints4, sd, k)
                opt.zero_grad()
                loss.backward()
                opt.step()
            else:
                with torch.no_grad():
                    O.train_loss(meas4, vol4, joints4, sd, k)
            return time.perf_counter() - t0

        run4(False)
        extra["forward_s_batch4_128"] = round(statistics.median(run4(False) for _ in range(3)), 3)
        run4(True)
        extra["train_step_s_batch4_128"] = round(statistics.median(run4(True) for _ in range(3)), 3)
        extra["lct_forward_s_1024x256x256"] = round(lct_leg(1024, 256), 3)
    return {"value": round(0.25 / t_train, 6), "unit": "samples/s", "cores": threads, "kind": "port", **extra,
            "cpu_model": cpu_model_string(), "protocol": "1 warm-up + 3 timed, median",
            "scaled_from": "one 128x128x128 cube = 1/4 of a 128x128x512 sample (times x4)",
            "train_step_s_per_cube": round(t_train, 3), "forward_s_per_cube": round(t_fwd, 3),
            "forward_value": round(0.25 / t_fwd, 6),
            "sample": f"batch 1 (B=1): 1 cube of 128x128x128, oracle (torch CPU fp32, {threads} threads): forward + losses {t_fwd:.2f} s, "
                      f"forward + losses + backward + Adam {t_train:.2f} s (medians of 3 after 1 warm-up); value = train leg / 4"}


def bench_sformer(args, emit=True):
    """BASELINE config 5: NlosPoseSformer (dim 256, depth 8, 8 heads x 32, patch 4, 16 frames of 128x128), batch 8,
    forward only (the reference has no training loop for this head), fp32 MFMA attention."""
    from hiddenpose_amd import _lib
    from hiddenpose_amd.NlosPoseSformer import NlosPoseSformer

    torch.cuda.set_device(0)
    torch.manual_seed(410)
    B = args.batch or 8
    kw = dict(dim=256, num_frames=16, num_joints=24, image_size=128, patch_size=4, channels=1, depth=8, heads=8,
              dim_head=32, out_dim=512)
    model = NlosPoseSformer(**kw).cuda().eval()
    model.linear_precision = args.conv_precision  # fp32 (default) or a bf16 matrix-core mode for the Linear GEMMs
    model.attention_precision = getattr(args, "attention", None) or ("bf16" if args.conv_precision == "bf16" else "fp32")
    video = torch.rand(B, 16, 1, 128, 128, device="cuda")
    for _ in range(args.warmup):
        model(video)
    torch.cuda.synchronize()
    _lib.profile_reset()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model(video)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    ntok = 24 + 16 * 1024
    attn_flops = 8 * (B * 8 * 16 * 1024 * (24 + 1024) * 32 * 4 + B * 8 * 24 * ntok * 32 * 4)
    ms = prof.get("sformer_attention_patch", (0, 0.0))[1] + prof.get("sformer_attention_joint", (0, 0.0))[1]
    ach = attn_flops * args.steps / (ms / 1e3) / 1e12 if ms else None
    apeak = MFMA_F32_PEAK_TFLOPS if model.attention_precision == "fp32" else MFMA_BF16_PEAK_TFLOPS
    line = {
        "metric": "samples/sec NlosPoseSformer forward (config 5)", "value": round(B * args.steps / dt, 3), "unit": "samples/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if (args.conv_precision == "fp32" and model.attention_precision == "fp32") else
                 f"{args.conv_precision} linears + {model.attention_precision} patch attention (MFMA, f32 soft-max and accumulation)",
        "data": "synthetic",
        "config": {"workload": f"NlosPoseSformer forward, batch {B}, 16 frames x 128x128, patch 4, dim 256, depth 8, "
                               "8 heads x 32, random-init weights"},
        "hip_kernel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in sorted(prof.items())},
        "roofline": {"kernel": "sformer_attention", "bound": "mfma", "achieved": round(ach, 2) if ach else None,
                     "peak": apeak, "unit": "TFLOP/s", "frac": round(ach / apeak, 4) if ach else None,
                     "traffic": None,
                     "note": "with 32-wide heads the 16-bit kernel is limited by the soft-max's vector work (4 MFMAs = 128 matrix-pipe "
                             "cycles per 32-key tile against ~900 cycles of exp / max / scale): the MFMA fraction is not its yardstick"
                             if model.attention_precision != "fp32" else None}}
    if emit:
        print(json.dumps(line), flush=True)
    return line


def bench_ingest(args, emit=True):
    """one dataset sample (600 x 256 x 256 RGBE image + 256^3 volume) -> network inputs
    (128^3 transient, 128^3 target) on the device, inputs resident in HBM; the oracle (NumPy restatement of
    utils/nlos_pose_dataloader.py:71-144) timed on the host as the CPU baseline."""
    from hiddenpose_amd import _lib
    from hiddenpose_amd.nlos_pose_dataloader import box_pyramid, rgbe_to_meas

    torch.cuda.set_device(0)
    g = torch.Generator("cuda").manual_seed(410)
    rgbe = torch.randint(0, 256, (600 * 256, 256, 4), dtype=torch.uint8, device="cuda", generator=g)
    rgbe[..., 3] = rgbe[..., 3] % 12 + 120
    vol = (torch.rand(256, 256, 256, device="cuda", generator=g) < 0.02).float()

    def step():
        return rgbe_to_meas(rgbe, 1), box_pyramid(vol, 1)

    for _ in range(max(1, args.warmup)):
        step()
    torch.cuda.synchronize()
    _lib.profile_reset()
    _lib.profile_enable(True)
    steps = max(args.steps, 20)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    px_all, px_kept = 600 * 256 * 256, 512 * 256 * 256
    name = "ingest_rgbe_to_meas"
    alg = 4.0 * px_kept + 4.0 * 128 ** 3  # fused pass: every kept RGBE pixel read once, the volume written once
    n, ms = prof[name]
    ach = alg / (ms / n / 1e3) / 1e9
    line = {
        "metric": "samples/sec ingest (600x256x256 .hdr pixels + 256^3 vol -> 128^3 meas, vol)", "value": round(steps / dt, 2),
        "unit": "samples/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8->f32", "data": "synthetic",
        "config": {"workload": "NlosPoseDataset.__getitem__ arithmetic, DAWNSAMPLE_CNT=1, RGBE bytes resident in HBM "
                               "(the 157 MB host->device copy of the expanded file is not in the timed region)"},
        "hip_kernel_ms_per_step": {k: round(v[1] / steps, 4) for k, v in sorted(prof.items())},
        "roofline": {"kernel": name, "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_us": round(1e3 * ms / n, 2),
                     "algorithmic_bytes": alg,
                     "note": f"the two global-max passes read 4 B x {px_all} pixels each in addition"},
    }
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import numpy as np

        import ingest_oracle as io

        rg = rgbe.cpu().numpy()
        v = vol.cpu().numpy()
        t0 = time.perf_counter()
        io.meas_from_bgr(io.rgbe_to_bgr_float(rg), 600, 512, 1)
        io.vol_pyramid(v, 1)
        cpu = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(1.0 / cpu, 3), "unit": "samples/s", "cores": 1, "kind": "port",
                                "sample": "the same single sample, NumPy float32 (decode from expanded RGBE bytes onward)"}
    if emit:
        print(json.dumps(line), flush=True)
    return line


def bench_highres(args, emit=True, also_bf16=False):
    """BASELINE config 4: 256x256x1024 transient, FeatureExtraction -> LCT -> normalize -> UNet3d only
    (forward + backward w.r.t. the FE / UNet parameters), batch 1, HBM-bandwidth roofline of the LCT."""
    from hiddenpose_amd import _lib
    from hiddenpose_amd import hip_ops as ops
    from hiddenpose_amd import testing as hpt
    from hiddenpose_amd.feature_extraction import FeatureExtraction
    from hiddenpose_amd.feature_propagation import FeaturePropagation
    from hiddenpose_amd.unet3d import UNet3d

    torch.cuda.set_device(0)
    torch.manual_seed(410)
    T, N, B = 1024, 256, args.batch or 1
    t0 = time.perf_counter()
    fe = FeatureExtraction(1, 1, stride=1).cuda()
    fp = FeaturePropagation(image_size=N, time_size=T, bin_len=5.12 / T, wall_size=2.0).cuda()
    un = UNet3d(1, 4).cuda()
    meas = hpt.synthetic_meas(B, T, N).cuda()
    fp.method.plan_for(meas.device)
    print(f"[bench] constants + plan for T={T} N={N}: {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)

    # --conv-precision bf16 / bf16s: the multi-channel 3^3 convolutions on the bf16 matrix cores -- forward, data gradient AND
    # weight gradient (hip_ops._DCONV_WGRAD_BF16, on unless HP_DCONV_WGRAD_BF16=0); fp32 tensors, fp32 accumulation, fp32 LCT /
    # norms.  (In the NlosPose model this arithmetic is opt-in: MODEL.DCONV_PRECISION = 'bf16'.)
    dconv_bf16 = getattr(args, "conv_precision", "fp32") in ("bf16", "bf16s")
    ops.set_dconv_precision("bf16" if dconv_bf16 else "fp32")

    params = [p for m in (fe, fp, un) for p in m.parameters()]

    def step():
        # gradients start from None, as after the training loop's optimizer.zero_grad(): otherwise every step also pays one
        # tiny accumulation add per parameter (68 launches here) that no training step has
        for p in params:
            p.grad = None
        f = ops.normalize_feature(fp(fe(meas), [0] * B, [T] * B))
        r = un(f)
        (r.square().mean() + f.mean()).backward()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    _lib.profile_reset()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _lib.profile_enable(False)
    ops.set_dconv_precision("fp32")
    prof = _lib.profile_read()
    V = T * N * N
    lct_ms = sum(v[1] for k, v in prof.items() if k.startswith("lct_"))
    lct_calls = 2 * args.steps  # forward + backward per step
    ach = 40.0 * V * B * lct_calls / (lct_ms / 1e3) / 1e9 if lct_ms else None
    # bytes the five passes actually move per direction: a lone volume (Hermitian route) 136 V, a packed pair 272 V
    moved = (136.0 * (B % 2) + 272.0 * (B // 2)) * V
    ach_moved = moved * lct_calls / (lct_ms / 1e3) / 1e9 if lct_ms else None
    conv_ms = sum(v[1] for k, v in prof.items() if k.startswith("dconv3_")) / args.steps
    conv_gflop = 3 * 410.0 * B   # FE + U-Net convolutions forward 410.0 GFLOP at 1024 x 256 x 256; backward = 2x
    line = {
        "metric": "samples/sec (256x256x1024 meas) FE+LCT+normalize+UNet fwd+bwd", "value": round(B * args.steps / dt, 3),
        "unit": "samples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 tensors and LCT; multi-channel 3^3 convolutions fwd / data gradient / weight gradient with bf16 operands (MFMA 4x4x4), f32 accumulation"
                 if dconv_bf16 else "f32", "data": "synthetic",
        "config": {"workload": f"FeatureExtraction+LCT+normalize_feature+UNet3d fwd+bwd, {N}x{N}x{T}, batch {B}"},
        "hip_kernel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in sorted(prof.items()) if v[1] / args.steps > 0.05},
        "roofline": {"kernel": "lct (5 passes, forward or adjoint)", "bound": "hbm", "achieved": round(ach, 1) if ach else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4) if ach else None, "traffic": None,
                     "algorithmic_bytes": "40 * T*N*N per volume and direction (SURVEY 8d)",
                     "moved_bytes_per_direction": moved, "achieved_on_moved_bytes": round(ach_moved, 1) if ach_moved else None,
                     "frac_on_moved_bytes": round(ach_moved / HBM_PEAK_GBS, 4) if ach_moved else None,
                     "ms_per_direction": round(lct_ms / lct_calls, 3) if lct_ms else None},
        "thin_channel_convolutions": {"ms_per_step": round(conv_ms, 3), "algorithmic_gflop_per_step": conv_gflop,
                                      "achieved_tflops": round(conv_gflop / conv_ms, 1) if conv_ms else None,
                                      "peak_tflops": MFMA_F32_PEAK_TFLOPS if not dconv_bf16 else None}}
    if also_bf16 and not dconv_bf16:
        # the same step with the multi-channel convolutions (forward, data gradient, weight gradient) on the bf16 matrix cores
        ops.set_dconv_precision("bf16")
        for _ in range(max(1, args.warmup)):
            step()
        torch.cuda.synchronize()
        _lib.profile_reset()
        _lib.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dtb = time.perf_counter() - t0
        _lib.profile_enable(False)
        ops.set_dconv_precision("fp32")
        pb = _lib.profile_read()
        line["bf16_thin_channel"] = {
            "ms_per_step": round(1e3 * dtb / args.steps, 3), "value": round(B * args.steps / dtb, 3), "unit": "samples/s",
            "dtype": "f32 tensors and LCT; multi-channel 3^3 convolutions fwd / data gradient / weight gradient with bf16 operands (MFMA 4x4x4)",
            "thin_channel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in sorted(pb.items()) if k.startswith("dconv3_")}}
    if emit:
        print(json.dumps(line), flush=True)
    return line


def quick_native(args, local, note, workload="native", precision=None, label="reference-native shape", dist_ctx=None,
                 batch=None):
    """The same train step at another shape / arithmetic (default: the reference's native 128x128x128, batch 4),
    1 warm-up + 3 timed steps.  `dist_ctx` = (rank, world, algo, bucket_mb): every rank runs it with its own samples and a
    bf16-wire gradient exchange, the timed region is bracketed by barriers and the slowest rank's time counts."""
    from hiddenpose_amd import testing as hpt
    from hiddenpose_amd.config import make_cfg
    from hiddenpose_amd.NlosPose import NlosPose
    from hiddenpose_amd.train_epoch import build_training, train_step

    T, N, B = WORKLOADS[workload]
    B = batch or B
    precision = precision or args.conv_precision
    dev = torch.device("cuda", local)
    cfg = make_cfg(T, N, device=local, conv_precision=precision)
    model = NlosPose(cfg).to(dev).train()
    criterion, voxel_criterion, optimizer, _ = build_training(cfg, model)
    rank, world, reducer = 0, 1, None
    if dist_ctx is not None:
        import torch.distributed as dist

        from hiddenpose_amd.data_parallel import GradBucketReducer

        rank, world, algo, bucket_mb = dist_ctx
        wire = torch.bfloat16 if precision != "fp32" else None
        reducer = GradBucketReducer(model, bucket_mb=bucket_mb, algo=algo, wire_dtype=wire)
    meas = hpt.synthetic_meas(B, T, N, "transient", seed=410 + rank * B).to(dev)
    vol = hpt.synthetic_vol(B, T, N, seed=1 + rank).to(dev)
    joints = hpt.synthetic_joints(B, T // 2, seed=2 + rank).to(dev)
    train_step(model, criterion, voxel_criterion, optimizer, meas, vol, joints, reducer)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    steps = 3
    if reducer is not None:
        reducer.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        train_step(model, criterion, voxel_criterion, optimizer, meas, vol, joints, reducer)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    dp_timing = None
    if reducer is not None:
        dp_timing = reducer.read_timing()
        reducer.remove_hooks()
    note(f"extra: {N}x{N}x{T} batch {B}/GPU x {world} {precision}: {1e3 * dt / steps:.1f} ms/step")
    out = {"workload": f"NlosPose train step, {N}x{N}x{T} ({label}), batch {B}" + ("/GPU" if world > 1 else "") + f", {precision}",
           "ms_per_step": round(1e3 * dt / steps, 3), "value": round(B * world * steps / dt, 3), "unit": "samples/s", "steps": steps,
           "warmup": 1}
    if world > 1:
        out.update({"n_gpus": world, "global_batch": B * world,
                    "parallelism": f"dp{world} ({dist_ctx[2]}, {'bf16' if precision != 'fp32' else 'fp32'} wire)",
                    "dp_timing": dp_timing})
    return out


def collective_library_info() -> dict:
    """Versions a reader needs next to a multi-GPU number (torch.distributed 'nccl' IS RCCL on ROCm)."""
    import torch.distributed as dist
    out = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None),
           "backend": dist.get_backend() if dist.is_initialized() else None}
    try:
        out["rccl"] = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception as e:  # noqa: BLE001  (gloo rehearsal, CPU build)
        out["rccl"] = f"unavailable ({type(e).__name__})"
    return out


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (torch.distributed.run,
    rendezvous on 127.0.0.1 and a free port), before THIS process has made any GPU call; their stdout / stderr are
    inherited, so rank 0's JSON line is this command's JSON line.  Returns the launcher's exit status."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    print(f"[bench] no launcher in the environment: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def device_identity(local: int) -> dict:
    pr = torch.cuda.get_device_properties(local)
    uuid = getattr(pr, "uuid", None)
    return {"device_index": local, "name": pr.name, "uuid": str(uuid) if uuid is not None else None,
            "pci_bus_id": getattr(pr, "pci_bus_id", None), "pid": os.getpid()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="t512", choices=sorted(WORKLOADS) + ["sformer", "highres", "ingest"])
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--conv-precision", default="fp32", choices=["fp32", "bf16", "bf16s", "bf16x3", "bf16x6"],
                    help="regressor convolution GEMM arithmetic; bf16s = BASELINE.json configs[2] in full (bf16 matrix cores and "
                         "bf16 activation storage, fp32 LCT / statistics / weights); bf16 = bf16 matrix cores over fp32 tensors. "
                         "The headline metric (configs[1]) is fp32, the default.")
    ap.add_argument("--attention", default=None, choices=["fp32", "bf16", "fp16"],
                    help="--workload sformer: arithmetic of the per-frame patch attention (BASELINE configs[4] words it "
                         "'MFMA fp16 attention'); default follows --conv-precision")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--no-wgrad-stream", action="store_true",
                    help="weight gradients on the main stream (default: on a second HIP stream, hip_ops.set_wgrad_async, so that "
                         "they overlap the memory-bound BatchNorm passes of backward's critical path).  Per-kernel times and the "
                         "roofline object always come from appended steps with everything on ONE stream (un-overlapped kernels).")
    ap.add_argument("--profile-steps", type=int, default=2, help="appended un-overlapped, fully profiled steps (per-kernel table, roofline)")
    ap.add_argument("--dp-algo", default=os.environ.get("HP_DP_ALGO", "all_reduce"), choices=["all_reduce", "rs_ag", "a2a"],
                    help="gradient exchange per bucket (data_parallel.GradBucketReducer): RCCL all-reduce, "
                         "reduce-scatter + all-gather, or direct all-to-all reduce-scatter + all-gather")
    ap.add_argument("--dp-wire", default=os.environ.get("HP_DP_WIRE", "auto"), choices=["auto", "fp32", "bf16"],
                    help="dtype on the wire for the gradient exchange; auto = bf16 in the bf16 convolution modes "
                         "(BASELINE configs[2]), fp32 otherwise")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="all legs of BASELINE.md section 3 in cpu_baseline (adds batch 4 at 128^3 and the LCT at 1024x256x256: minutes)")
    ap.add_argument("--no-extra", action="store_true", help="skip the short 128^3 (reference-native shape) run reported under `extra`")
    args = ap.parse_args()

    if args.workload == "sformer":
        return bench_sformer(args)
    if args.workload == "highres":
        return bench_highres(args)
    if args.workload == "ingest":
        return bench_ingest(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))  # nothing above touches the GPU (importing torch does not initialise HIP)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # rehearsal hook (single-GPU boxes): HP_DIST_BACKEND=gloo HP_SHARE_GPU=1 runs all ranks on cuda:0 over gloo
    backend = os.environ.get("HP_DIST_BACKEND", "nccl")
    if os.environ.get("HP_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from hiddenpose_amd import _lib
    from hiddenpose_amd import testing as hpt
    from hiddenpose_amd.config import make_cfg
    from hiddenpose_amd.data_parallel import GradBucketReducer
    from hiddenpose_amd.NlosPose import NlosPose
    from hiddenpose_amd.train_epoch import build_training, seed_everything, train_step

    T, N, B = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    cfg = make_cfg(T, N, device=local, conv_precision=args.conv_precision)
    from hiddenpose_amd import hip_ops as _ops
    wgrad_stream = _ops._WGRAD_ASYNC and not args.no_wgrad_stream     # HP_WGRAD_STREAM=0 in the environment switches it off as well
    _ops.set_wgrad_async(wgrad_stream)
    bf16 = args.conv_precision != "fp32"
    # split modes issue 3 / 6 bf16 MFMAs per algorithmic product: the useful-FLOP ceiling shrinks accordingly
    mfma_terms = {"fp32": 1, "bf16": 1, "bf16s": 1, "bf16x3": 3, "bf16x6": 6}[args.conv_precision]
    mfma_peak = round(MFMA_BF16_PEAK_TFLOPS / mfma_terms, 1) if bf16 else MFMA_F32_PEAK_TFLOPS
    seed_everything(410)
    model = NlosPose(cfg).to(dev)
    model.train()
    criterion, voxel_criterion, optimizer, _ = build_training(cfg, model)
    force = bool(os.environ.get("HP_FORCE_REDUCER"))  # rehearse the DP code path on a single rank
    if force and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    wire = {"auto": torch.bfloat16 if bf16 else None, "fp32": None, "bf16": torch.bfloat16}[args.dp_wire]
    reducer = GradBucketReducer(model, bucket_mb=args.bucket_mb, force_collectives=force, algo=args.dp_algo,
                                wire_dtype=wire) if (world > 1 or force) else None

    # synthetic batch of this rank (different samples per rank), resident in HBM before timing
    meas = hpt.synthetic_meas(B, T, N, "transient", seed=410 + rank * B).to(dev)
    vol = hpt.synthetic_vol(B, T, N, seed=1 + rank).to(dev)
    joints = hpt.synthetic_joints(B, T // 2, seed=2 + rank).to(dev)

    reducer_on = reducer is not None

    # A/B hook: HP_MAIN_PRIO=-1 runs the whole step on a HIGH-priority stream (the side stream of the weight gradients keeps
    # the default, lower priority), so that backward's critical path wins the workgroup slots it can use
    main_stream = None
    if os.environ.get("HP_MAIN_PRIO"):
        main_stream = torch.cuda.Stream(dev, priority=int(os.environ["HP_MAIN_PRIO"]))
        main_stream.wait_stream(torch.cuda.current_stream(dev))

    def step():
        if main_stream is not None:
            with torch.cuda.stream(main_stream):
                return train_step(model, criterion, voxel_criterion, optimizer, meas, vol, joints, reducer)
        return train_step(model, criterion, voxel_criterion, optimizer, meas, vol, joints, reducer)

    def note(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    note(f"workload {args.workload}: T={T} N={N} batch/GPU={B} world={world}")
    for i in range(args.warmup):
        t_w = time.perf_counter()
        step()
        torch.cuda.synchronize()
        note(f"warmup step {i} done in {time.perf_counter() - t_w:.2f} s "
             f"(peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)")
    torch.cuda.synchronize()
    _lib.profile_reset()
    # timed region: NO per-kernel events (an event pair per launch costs ~4 ms/step over the ~800 launches of a step, and
    # with the weight gradients on their own stream two kernels are in flight, so a kernel's duration would describe a
    # contended launch): the per-kernel table and `roofline` come from the appended steps below
    _lib.profile_enable(False)
    if reducer is not None:
        reducer.enable_timing(True)  # a pair of events per bucket on the communication stream + one pair around finish()'s wait
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = []
    for _ in range(args.steps):
        loss, _, _ = step()
        losses.append(loss)          # device scalars: read back after the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dp_timing = None
    if reducer is not None:
        dp_timing = reducer.read_timing()
        reducer.enable_timing(False)
    # appended, untimed: `--profile-steps` more steps of the same training run with every kernel family bracketed by HIP
    # events on its launch stream and the weight gradients back on the MAIN stream -- one kernel in flight at a time, so the
    # durations measure the kernels themselves.  Also timed as a whole: the un-overlapped step, for the A/B on the line.
    _ops.set_wgrad_async(False)
    step()                           # the first step after the switch re-times allocator blocks: not profiled
    torch.cuda.synchronize()
    _lib.profile_reset()
    _lib.profile_enable(1)
    psteps = max(1, args.profile_steps)
    tp0 = time.perf_counter()
    for _ in range(psteps):
        step()
    torch.cuda.synchronize()
    profiled_ms = 1e3 * (time.perf_counter() - tp0) / psteps
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    _ops.set_wgrad_async(wgrad_stream)
    dp_ab = None
    if reducer is not None and not args.no_extra:
        # the same step with each exchange algorithm, 3 timed steps each (1 warm-up), slowest rank: the first multi-GPU run
        # A/Bs them by itself (the xGMI mesh favours the direct forms, nothing here has measured it)
        dp_ab = {}
        for algo in ("all_reduce", "rs_ag", "a2a"):
            reducer.remove_hooks()
            reducer = GradBucketReducer(model, bucket_mb=args.bucket_mb, force_collectives=force, algo=algo, wire_dtype=wire)
            step()
            torch.cuda.synchronize()
            reducer.enable_timing(True)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tb = torch.tensor([time.perf_counter() - ta], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
            tm = reducer.read_timing()
            dp_ab[algo] = {"ms_per_step": round(1e3 * float(tb.item()) / 3, 3), "exchange_ms_per_step": tm["exchange_ms_per_step"],
                           "exposed_ms_per_step": tm["exposed_ms_per_step"]}
            note(f"dp-algo A/B {algo}: {dp_ab[algo]}")
        reducer.remove_hooks()
        reducer = GradBucketReducer(model, bucket_mb=args.bucket_mb, force_collectives=force, algo=args.dp_algo, wire_dtype=wire)
    ranks_seen = [device_identity(local)]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # evidence that the collectives really spanned `world` processes on distinct devices
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, dict(device_identity(local), rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", "0"))))
    dist_extra = None
    if world > 1 and args.workload == "t512" and not args.no_extra and args.conv_precision == "fp32":
        # BASELINE configs[2]: batch 32 = 8 x 4 in bf16 (matrix cores + activation storage) with fp32 LCT, bf16 on the wire;
        # at other world sizes the same per-GPU share
        if reducer is not None:
            reducer.remove_hooks()
        del step
        model = optimizer = criterion = voxel_criterion = meas = vol = joints = reducer = None
        torch.cuda.empty_cache()
        dist_extra = quick_native(args, local, note, "t512", "bf16s", "headline cube, configs[2] arithmetic, storage and wire",
                                  dist_ctx=(rank, world, args.dp_algo, args.bucket_mb))

    if rank == 0:
        samples = B * world * args.steps
        line = {
            "metric": "samples/sec (128x128x512 meas) fwd+bwd" if args.workload == "t512"
                      else f"samples/sec ({N}x{N}x{T} meas) fwd+bwd",
            "value": round(samples / dt, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.conv_precision if bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"NlosPose train step (fwd+L2Joint+BCEDice loss+bwd+Adam), {N}x{N}x{T} transients, "
                                   f"batch {B}/GPU, " + (f"{args.conv_precision} convolutions (bf16 matrix cores, {mfma_terms} plane product(s), fp32 "
                                   "accumulation) with fp32 LCT, U-Net, norms, losses and " + ("bf16 regressor activations / activation gradients in HBM (fp32 raw conv outputs, statistics, weights)" if args.conv_precision == "bf16s" else "fp32 tensors in HBM") if bf16 else "fp32") + ", random-init weights"
                                   + ((", weight gradients on a second stream" if _ops._WGRAD_SIDE_MODE == 1 else
                                       ", weight gradients of the 3^3 / 4^3 convolutions (and of small 1^3 ones) on a second stream")
                                      if wgrad_stream else ", weight gradients on the main stream"),
                       "global_batch": B * world, "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                       "backend": (dist.get_backend() if dist.is_initialized() else None),
                       "exchange": args.dp_algo if reducer_on else None, "ranks": ranks_seen,
                       # what a reader needs to explain the FIRST multi-GPU run from this one line: library versions, the
                       # communication-related environment in effect, per-bucket exchange times and the exposed communication
                       "collective_library": collective_library_info() if reducer_on else None,
                       "comm_env": {k: v for k, v in sorted(os.environ.items())
                                    if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_", "HP_DP_", "HP_WGRAD", "HP_DIST", "HP_SHARE", "TORCH_NCCL"))},
                       "dp_timing": dp_timing,
                       "parallelism": f"dp{world}" + (f" ({args.dp_algo}, {'bf16' if wire is not None else 'fp32'} wire, "
                                                      f"{args.bucket_mb:g} MB buckets)" if reducer_on else ""), "hip_stages": sorted(__import__("hiddenpose_amd.hip_ops", fromlist=["x"]).HIP_STAGES),
                       "aten_stages": sorted(__import__("hiddenpose_amd.hip_ops", fromlist=["x"]).ATEN_STAGES)},
            "loss": round(float(loss.item()), 6),
            # the same synthetic batch every step, Adam lr 1e-3 from the reference's initialisation: the loss must move
            "loss_per_timed_step": [round(float(v.item()), 4) for v in losses],
        }
        roof = None
        if prof:
            name, (n, ms) = max(prof.items(), key=lambda kv: kv[1][1])
            work = algorithmic_work(name, T, N, B)
            conv = posenet_conv_flops(T, N, B)
            if name in conv and n:
                # one profiling name covers every layer's launch of that kernel family:
                # achieved = total algorithmic FLOPs of the family / total time of its launches
                ach = conv[name] * psteps / (ms / 1e3) / 1e12
                roof = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": mfma_peak,
                        "unit": "TFLOP/s", "frac": round(ach / mfma_peak, 4), "traffic": None, "launches": n,
                        "avg_launch_us": round(1e3 * ms / n, 2),
                        "gflop_per_step": round(conv[name] / 1e9, 1)}
                # HBM bytes of this kernel family per step from the committed PMC passes (rocprofv3 --pmc
                # FETCH_SIZE / WRITE_SIZE in separate runs, FETCH x2 on gfx950); only valid for the same workload
                pmc = os.path.join(ROOT, "profiles", "t512_pmc_hbm_traffic.json")
                if args.workload == "t512" and B == 4 and not bf16 and os.path.exists(pmc):
                    fam = json.load(open(pmc))
                    key = name if name in fam else None
                    if fam.get("_csrc_sha") != csrc_sha():
                        # the counters were collected on other kernels than the ones just timed: quote nothing
                        roof["stale_profile"] = (f"profiles/t512_pmc_hbm_traffic.json was collected at csrc sha "
                                                 f"{fam.get('_csrc_sha')}, this run is {csrc_sha()}: traffic not quoted")
                    elif key:
                        per_step = n / psteps
                        roof["traffic"] = round(fam[key] / per_step, 3)
                        roof["traffic_unit"] = ("GB of HBM per launch, mean over the family's launches (PMC FETCH_SIZE x2 + WRITE_SIZE, "
                                                "profiles/t512_pmc_hbm_traffic.json, same csrc sha)")
                        roof["traffic_gb_per_step"] = fam[key]
                        roof["gflop_per_launch"] = round(conv[name] / 1e9 / per_step, 1)
            elif work and n:
                bound, amount = work
                avg_s = ms / n / 1e3
                if bound == "hbm":
                    ach, peak, unit = amount / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
                else:
                    ach, peak, unit = amount / avg_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
                roof = {"kernel": name, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                        "frac": round(ach / peak, 4), "traffic": None, "launches": n,
                        "avg_launch_us": round(1e3 * ms / n, 2)}
            line["hip_kernel_ms_per_step"] = {k: round(v[1] / psteps, 3) for k, v in sorted(prof.items())}
            if roof is not None:
                roof["note"] = (f"per-kernel times: HIP events over {psteps} appended step(s) of the same run with every kernel on ONE stream "
                                f"(un-overlapped, {profiled_ms:.1f} ms/step incl. ~4 ms of event overhead); the timed region runs the weight "
                                "gradients on a second stream" if wgrad_stream else
                                f"per-kernel times: HIP events over {psteps} appended step(s) of the same run")
            line["unoverlapped_profiled_ms_per_step"] = round(profiled_ms, 3)
            cf = posenet_conv_flops(T, N, B)
            line["mfma_tflops_by_kernel"] = {k: round(cf[k] * psteps / (prof[k][1] / 1e3) / 1e12, 1)
                                             for k in sorted(cf) if k in prof and prof[k][1] > 0}
        line["roofline"] = roof
        if dist_extra is not None:
            line["extra"] = {"configs2_bf16s_dp": dist_extra}
        if dp_ab is not None:
            line.setdefault("extra", {})["dp_algo_ab"] = dp_ab
        if world == 1 and args.workload == "t512" and not args.no_extra:
            # SURVEY 8(d) names two shapes for configs[1]: the BASELINE-worded 128x128x512 cube (the headline above) and
            # the reference's own training shape 128^3 (train.py:77-86); the second is reported here, same step, same batch
            del step
            model = optimizer = criterion = voxel_criterion = meas = vol = joints = None
            torch.cuda.empty_cache()
            line["extra"] = {"native_128": quick_native(args, local, note)}
            if args.conv_precision == "fp32":
                # BASELINE configs[2]'s per-GPU share (bf16 arithmetic + bf16 activation storage, fp32 LCT) on the headline cube
                torch.cuda.empty_cache()
                line["extra"]["configs2_bf16s"] = quick_native(args, local, note, "t512", "bf16s",
                                                              "headline cube, configs[2] arithmetic and storage")
                # BASELINE configs[3] and configs[4] on the same driver line (short runs; their own workloads give the long form)
                torch.cuda.empty_cache()
                sub = argparse.Namespace(**vars(args))
                sub.steps, sub.warmup, sub.batch = 3, 1, 0
                hr = bench_highres(sub, emit=False, also_bf16=True)
                line["extra"]["configs3_highres"] = {k: hr[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup",
                                                                        "config", "roofline", "thin_channel_convolutions",
                                                                        "bf16_thin_channel")}
                note(f"extra: highres {hr['ms_per_step']:.1f} ms/step")
                torch.cuda.empty_cache()
                sub.conv_precision, sub.attention = "bf16", "fp16"
                sf = bench_sformer(sub, emit=False)
                line["extra"]["configs4_sformer_fp16"] = {k: sf[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup",
                                                                             "dtype", "config", "roofline")}
                note(f"extra: sformer {sf['ms_per_step']:.1f} ms/step")
                torch.cuda.empty_cache()
                sub.no_cpu_baseline = True
                ig = bench_ingest(sub, emit=False)   # one dataset sample -> network inputs on the device
                line["extra"]["ingest"] = {k: ig[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "config", "roofline")}
                note(f"extra: ingest {ig['ms_per_step']:.3f} ms/sample")
        if world == 1 and not args.no_cpu_baseline:
            # the box's CPU share for one GPU is 16 cores: more threads than that only oversubscribe
            try:
                avail = len(os.sched_getaffinity(0))
            except AttributeError:
                avail = os.cpu_count() or 1
            threads = max(1, min(16, avail))
            note(f"cpu_baseline: oracle train step on one 128^3 cube with {threads} threads ...")
            line["cpu_baseline"] = cpu_baseline(threads, full=args.cpu_baseline_full)
            note("cpu_baseline done")
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
