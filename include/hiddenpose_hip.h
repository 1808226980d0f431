/* libhiddenpose_hip.so -- C ABI of the MI355X (gfx950) NlosPose hot path.
 *
 * The reference (Hagtaril/HiddenPose) has no native/FFI seam: its hot path is
 * a chain of stock torch operators behind nn.Module.forward.  Each entry point
 * below replaces the operator sequence of one reference function; the Python
 * host side (the hiddenpose_amd Python package) mirrors the reference's module API on top of
 * these calls through ctypes.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain C, no framework types: device pointers are `void*`/`float*` into
 *    memory owned by the CALLER (PyTorch caching allocator on the Python side);
 *    the library allocates device memory only inside plan objects.
 *  - every call returns 0 on success or a negative hp_status; the message is
 *    available from hp_last_error_string() (thread local).  Nothing throws or
 *    exits across the ABI.
 *  - every compute call is asynchronous on the hipStream_t passed as `stream`
 *    (a `void*` here so that the header needs no HIP include).
 *  - tensors are fp32, contiguous, (B, C, T, H, W) as in the reference.
 *  - plans are immutable after creation and may be shared by streams/threads.
 */
#ifndef HIDDENPOSE_HIP_H
#define HIDDENPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hp_status {
  HP_OK = 0,
  HP_ERR_BAD_ARG = -1,     /* shape / size / null pointer */
  HP_ERR_UNSUPPORTED = -2, /* valid request the library has no kernel for */
  HP_ERR_HIP = -3,         /* a HIP runtime call failed; see hp_last_error_string() */
  HP_ERR_WORKSPACE = -4,   /* caller workspace too small */
  HP_ERR_NO_DEVICE = -5
} hp_status;

int hp_version(void);
const char* hp_last_error_string(void);

/* Per-kernel timing with HIP events recorded on the launch stream (off by default).
 * hp_profile_get(i, ...) synchronises the recorded events and returns the launch count
 * and total milliseconds of the i-th kernel name seen since hp_profile_reset(). */
/* on = 1: every kernel family; on = 2: only the matrix-core convolution families (names "conv_*" except the weight
 * packing), which keeps the event overhead out of a timed region; 0: off. */
int hp_profile_enable(int on);
int hp_profile_reset(void);
int hp_profile_count(void);
int hp_profile_get(int i, char* name, int name_cap, int64_t* launches, double* total_ms);

/* Stage ranges for rocprofv3's marker trace (`rocprofv3 --kernel-trace --marker-trace --stats`): SURVEY section 5's
 * "roctx ranges per stage" -- the reference has only wall-clock prints around epochs (train_epoch.py:27-31,95-104).
 * hp_range_enable(1) looks the marker library up at run time (librocprofiler-sdk-roctx, then libroctx64; neither is a link
 * dependency) and returns 1 if ranges are live, 0 if the box has no such library (not an error); hp_range_enable(0)
 * switches them off.  hp_range_push returns the calling thread's nesting depth (0 when ranges are off), hp_range_pop
 * the depth left; push / pop pair up per thread.  hp_range_start / hp_range_stop are the thread-free form (the backward
 * stages: autograd runs them on its own thread and the pass ends on the caller's): start returns an id > 0, or 0 when
 * ranges are off; stop(0) is a no-op.  Host-side only: nothing is enqueued on a stream. */
int hp_range_enable(int on);
int hp_range_push(const char* name);
int hp_range_pop(void);
int64_t hp_range_start(const char* name);
int hp_range_stop(int64_t id);

/* ------------------------------------------------------------------------
 * LCT physics layer.
 * Replaces models/feature_propagation.py: LCT._parpareparam :71-109,
 * _resamplingOperator :111-139, _definePsf :141-171, todev :173-184 (plan) and
 * LCT.forward :186-257 (hp_lct_forward), mode='lct'.
 * ---------------------------------------------------------------------- */
typedef struct hp_lct_plan hp_lct_plan;

enum { HP_MATERIAL_DIFFUSE = 0, HP_MATERIAL_SPECULAR = 1 };

/* Host-only constants (no GPU needed); used by the plan and by CPU tests.
 * gridz[T]; mtx: dense row-major T*T; psf_zidx: (2N*2N) int32, z index of the
 * PSF's 1 in every (x,y) column after the roll (first one if several);
 * psf_count: number of ones; invpsf_re/im: (2T,2N,2N) natural order, may be NULL. */
int hp_lct_host_constants(int T, int N, double bin_len, double wall_size,
                          float* gridz, float* mtx, int32_t* psf_zidx, int64_t* psf_count,
                          float* invpsf_re, float* invpsf_im);

/* Builds all constants on the host, uploads them to `device` (HIP ordinal). */
int hp_lct_plan_create(hp_lct_plan** plan, int T, int N, double bin_len, double wall_size,
                       int material, int device);
/* The same with the inverse filter of the reference's 'bp' mode (models/feature_propagation.py:93-94, models/tflct.py:59-60):
 * invpsf = conj(fftn(psf)) instead of the Wiener filter conj(F) / (1/snr + |F|^2).  The Laplacian-of-Gaussian epilogue of
 * that mode is hp_laplacian5_*. */
#define HP_LCT_MODE_LCT 0
#define HP_LCT_MODE_BP 1
int hp_lct_plan_create_mode(hp_lct_plan** plan, int T, int N, double bin_len, double wall_size,
                            int material, int mode, int device);
int hp_lct_plan_destroy(hp_lct_plan* plan);
/* Epilogue of the 'bp' mode (models/feature_propagation.py:246-253; models/tflct.py:164-174): per (planes) volume
 * (T, H, W): ReplicationPad3d(2) -> conv3d with the 5x5x5 Laplacian-of-Gaussian filter w125 (device pointer, [kt][kh][kw];
 * utils/helper.py:13-32 builds it) -> first time slice set to 0.  _backward is the adjoint (gradient w.r.t. x). */
int hp_laplacian5_forward(const float* x, const float* w125, float* y, long planes, int T, int H, int W, void* stream);
int hp_laplacian5_backward(const float* dy, const float* w125, float* dx, long planes, int T, int H, int W, void* stream);
/* Bytes of caller-provided scratch needed for a batch of B volumes (B*D in
 * the reference's naming). */
size_t hp_lct_workspace_bytes(const hp_lct_plan* plan, int batch);
/* y = LCT(x): x,y (batch, T, N, N) fp32 device pointers; tbe=0, ten=T. */
int hp_lct_forward(const hp_lct_plan* plan, const float* x, float* y, int batch,
                   void* workspace, size_t workspace_bytes, void* stream);
/* gx = LCT^T(gy) (vector-Jacobian product of hp_lct_forward). */
int hp_lct_backward(const hp_lct_plan* plan, const float* gy, float* gx, int batch,
                    void* workspace, size_t workspace_bytes, void* stream);
/* Time windows of LCT.forward (:193-200): sample b of x (B, D, tnum, H, W) is placed at time offset
 * tbes[b] of the zero-initialised y (B, D, T, H, W) (to_window = 0), or cut back out of it (to_window = 1:
 * x is written, the adjoint used by the backward pass).  tbes is a HOST array of B offsets with
 * 0 <= tbes[b] and tbes[b] + tnum <= T.  Copy engine work only (memset + strided copies on `stream`). */
int hp_lct_time_window(float* y_full, float* x_window, int B, int D, int tnum, int T, long plane, const int* tbes,
                       int to_window, void* stream);
/* Test hook: copy the device-resident inverse PSF back in natural (2T,2N,2N) order. */
int hp_lct_plan_get_invpsf(const hp_lct_plan* plan, float* invpsf_re, float* invpsf_im);

/* ------------------------------------------------------------------------
 * Pose regressor (models/posenet3d_50.py) building blocks.  Activations are
 * channels-last fp32, X[b][d][h][w][C]; all GEMM arithmetic is exact-fp32 MFMA.
 *
 * hp_conv3d_* replace nn.Conv3d / nn.ConvTranspose3d forward and both gradients
 * (posenet3d_50.py:9-24 conv3x3x3/conv1x1x1, :176-181 stem, :129-132 deconv).
 * Supported: Conv3d k in {1,3,7}, stride 1|2 (k7: the 1-channel stride-1 stem),
 * Cin % 4 == 0 otherwise (stride-2 data gradients and ConvTranspose3d k4 s2 p1: channels % 32 == 0).
 * Weights are used in a packed K-contiguous layout ([tap][Cout][Cin]; for the
 * data gradient [tap][Cin][Cout]) produced by hp_conv3d_pack_weight from the
 * torch layout and converted back for gradients by hp_conv3d_unpack_wgrad.
 *
 * precision selects the arithmetic of the three GEMMs (tensors stay fp32 in HBM):
 *   HP_PRECISION_FP32  v_mfma_f32_32x32x2_f32, bit-equal to an fp32 fmaf chain (default);
 *   HP_PRECISION_BF16  both operands rounded to bf16 (RNE) on the way into LDS,
 *                      v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the
 *                      "bf16 with fp32 LCT" training configuration (BASELINE.json configs[2]);
 *   HP_PRECISION_BF16X3 / _BF16X6  each fp32 operand split into 2 / 3 bf16 planes
 *                      (value, residual, residual of the residual) and the 3 / 6 plane
 *                      products with pa + pb < planes accumulated in fp32 on the bf16 matrix
 *                      cores: ~2^-16 / ~2^-24 relative error per product (X6 = the fp32
 *                      level, not bit-identical to FP32).  Opt-in; never the default.
 * The value is the number of bf16 operand planes.
 * ---------------------------------------------------------------------- */
#define HP_PRECISION_FP32 0
#define HP_PRECISION_BF16 1
#define HP_PRECISION_BF16X3 2
#define HP_PRECISION_BF16X6 3
#define HP_PRECISION_FP16 4 /* hp_sformer_attention only: fp16 operands (v_mfma_f32_32x32x16_f16), fp32 soft-max / accumulation */
typedef struct hp_conv_desc {
  int B, Di, Hi, Wi; /* input volume */
  int Cin, Cout;
  int k, stride, pad;
  int transposed; /* 0: Conv3d, 1: ConvTranspose3d */
  int precision;  /* HP_PRECISION_* */
  int io;         /* HP_IO_*: which activation tensors of the call are bf16 instead of fp32 (0: all fp32) */
} hp_conv_desc;
/* Element type of the activation tensors (BASELINE configs[2]: bf16 storage, fp32 accumulators and statistics).
 * Pointers of flagged tensors address 2-byte bf16 elements; weights, bias, statistics and dw stay fp32. */
#define HP_IO_X_BF16 1  /* forward input x (also the X operand of the weight gradient) */
#define HP_IO_Y_BF16 2  /* forward output y */
#define HP_IO_DY_BF16 4 /* incoming gradient dy of the data / weight gradient */
#define HP_IO_DX_BF16 8 /* data gradient dx and its addend */
#define HP_IO_W_BF16 16 /* packed weight images (w_fwd / w_dgrad of hp_conv3d_pack_weight, as read by forward / backward_data)
                         * are bf16: only with a bf16 gathered tensor, HP_PRECISION_BF16 and its channel count % 64 == 0 */

size_t hp_conv3d_packed_weight_elems(const hp_conv_desc* d);
int hp_conv3d_pack_weight(const hp_conv_desc* d, const float* w_torch, void* w_fwd, void* w_dgrad, void* stream);
int hp_conv3d_unpack_wgrad(const hp_conv_desc* d, const float* dw_packed, float* dw_torch, void* stream);
/* y = conv(x) [+ bias]; if stats != NULL it receives per-channel sums and sums of squares of y for train-mode BatchNorm
 * as HP_STATS_SLOTS partial vectors: stats[s][0 .. Cout) sums, stats[s][Cout .. 2 Cout) sums of squares over the workgroups
 * whose launch index % HP_STATS_SLOTS == s (HP_STATS_SLOTS * 2 * Cout doubles, zeroed by the call; hp_bn_train_finalize* adds
 * the slots).  Why slots: every workgroup ends with one fp64 atomic per column and statistic, and atomics on ONE address
 * serialise at ~16 ns each -- 32768 M tiles per address made the 64-channel 1^3 layers of layer 1 take 0.73 instead of
 * 0.21 ms (bf16 storage) / 0.88 instead of 0.49 ms (fp32) at the headline shape. */
#define HP_STATS_SLOTS 32
int hp_conv3d_forward(const hp_conv_desc* d, const void* x, const float* w_fwd, const float* bias, void* y,
                      double* stats, void* stream);
/* dx = conv^T(dy) [+ addend]: `addend` (same shape as dx, may be NULL) lets a second gradient contribution to the
 * same tensor (residual / shortcut branch) be summed in the epilogue instead of by a separate pass.  For a strided
 * 1^3 convolution (gradient reaches every second voxel per axis) addend may BE dx: the sum is then formed in place
 * and no zero-filled or copied tensor is produced. */
int hp_conv3d_backward_data(const hp_conv_desc* d, const void* dy, const float* w_dgrad, void* dx,
                            const void* addend, void* stream);
/* Same with the addend gated by a byte mask (hp_bn_apply's relu_mask layout: one byte per channel quad of dx):
 * dx = conv^T(dy) + addend (.) mask.  Bottleneck.forward's identity shortcut (posenet3d_50.py:90-93): the shortcut
 * gradient is the block's output gradient times the sign mask of the block output, which therefore is never
 * written as a tensor.  Dense (stride-1) data gradients with > 32 input channels, a multiple of 4, only. */
int hp_conv3d_backward_data_masked(const hp_conv_desc* d, const void* dy, const float* w_dgrad, void* dx,
                                   const void* addend, const unsigned char* addend_mask, void* stream);
/* hp_conv3d_backward_data[_masked] that ALSO takes the backward reduction of the BatchNorm unit `a` in front of the
 * convolution (y_a = [relu](BN(z_a)) is this convolution's input and has no other consumer, so dx IS dy_a): while a tile of
 * dx is in hand the kernel reads the same tile of z_a once and accumulates  sum g  and  sum g * (z_a - mean) * rstd  per
 * channel, g = dx (.) [gamma * zhat + beta > 0] (relu != 0) or dx (Bottleneck.forward / DeconvHead.forward,
 * posenet3d_50.py:75-95,129-153); for a unit WITH a residual (the block's output unit, whose output also feeds the next block's
 * identity shortcut: dx is then the complete block-output gradient, addend included) relu_mask = the byte mask hp_bn_apply wrote.  sums: HP_STATS_SLOTS x 2 x Cin doubles (partial vectors, zeroed by the call), to be
 * handed to hp_bn_backward_presummed, whose separate pass over dy_a and z_a it replaces.  Exact-fp32 tensors, a dense
 * stride-1 (or ConvTranspose3d) data gradient with B*D*H*W % 128 == 0 and Cin % 64 == 0 (% 128 beyond 64) only: *fused is
 * set to 1 when the sums were taken, to 0 when the call fell back to the plain data gradient (sums untouched). */
int hp_conv3d_backward_data_bnsums(const hp_conv_desc* d, const void* dy, const float* w_dgrad, void* dx, const void* addend,
                                   const unsigned char* addend_mask, const float* z, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, int relu, const unsigned char* relu_mask, double* sums,
                                   int* fused, void* stream);
/* dw_packed (same layout as w_fwd) is zeroed and accumulated by the call. */
int hp_conv3d_backward_weight(const hp_conv_desc* d, const void* x, const void* dy, float* dw_packed, void* stream);
/* How hp_conv3d_backward_weight splits its reduction over M = B * output voxels for this descriptor: `msplit` chunks of
 * `chunk_rows` consecutive rows (the last one shorter) whose partial sums meet in fp32 atomics.  A query for tests that
 * aim samples at the chunk seams (tests/test_conv_headline_gpu.py); the dedicated stem / dense-1^3 kernels split
 * differently and ignore it. */
int hp_conv3d_backward_weight_split(const hp_conv_desc* d, long* msplit, long* chunk_rows);

/* BatchNorm3d (posenet3d_50.py:70-95,133,182) on [M][C] channels-last matrices.  The raw convolution output z, the
 * statistics and every parameter are fp32; `io` says which ACTIVATION tensors of a call are bf16 (0: all fp32): */
#define HP_BN_ACT_BF16 1 /* forward: y, and a plain residual `res` */
#define HP_BN_DY_BF16 2  /* backward: the incoming gradient dy */
#define HP_BN_DZ_BF16 4  /* backward: the outgoing gradients dz (and g_out) */
#define HP_BN_Z_BF16 8   /* the raw convolution output z (and, in hp_bn_apply_res_bn, the raw shortcut output) */
/* stats: the HP_STATS_SLOTS x 2C partial sums of hp_conv3d_forward (see there). */
int hp_bn_train_finalize(const double* stats, long M, int C, float eps, float momentum, float* mean, float* rstd,
                         float* running_mean, float* running_var, void* stream);
/* Same, and BatchNorm3d's `num_batches_tracked` (int64, device) is incremented by the same launch (may be NULL). */
int hp_bn_train_finalize_counted(const double* stats, long M, int C, float eps, float momentum, float* mean, float* rstd,
                                 float* running_mean, float* running_var, long long* num_batches_tracked, void* stream);
int hp_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* rstd,
                     void* stream);
/* y = act((z - mean) * rstd * gamma + beta [+ res]);  res may be NULL; relu = 0|1.
 * relu_mask (may be NULL): M*C/4 bytes, bit k of byte q = [y[4q + k] > 0] -- what the backward of a unit WITH a
 * residual needs of y (1 byte instead of 16 per channel quad). */
int hp_bn_apply(const void* z, const void* res, void* y, long M, int C, const float* mean, const float* rstd,
                const float* gamma, const float* beta, int relu, unsigned char* relu_mask, int io, void* stream);
/* Same, with the residual given as the RAW output of the shortcut convolution and that shortcut's BatchNorm
 * (res_mean .. res_beta, all C) applied on the fly: y = act(BN(z) + BN_res(res)) -- Bottleneck.forward with a
 * `downsample` branch (posenet3d_50.py:86-93) without materialising the normalised shortcut tensor.  The result is
 * bit-identical to hp_bn_apply on a stored BN_res(res).  (res is that raw fp32 tensor whatever `io` says.) */
int hp_bn_apply_res_bn(const void* z, const void* res, void* y, long M, int C, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, int relu, unsigned char* relu_mask, const float* res_mean,
                       const float* res_rstd, const float* res_gamma, const float* res_beta, int io, void* stream);
size_t hp_bn_backward_workspace_bytes(int C);
/* g = dy * [y > 0] (stored to g_out if not NULL: gradient of the residual branch);
 * dz = gradient w.r.t. the BatchNorm input; dgamma/dbeta may be NULL.
 * With relu_mask and neither g_out nor y, g is never stored: both passes apply the byte mask to dy.  The same call
 * (relu = 1, relu_mask = the byte mask of the unit that consumed this unit's output as its residual) serves a
 * shortcut unit whose incoming gradient is that unit's masked output gradient. */
/* The ReLU mask comes from relu_mask (hp_bn_apply's byte mask) if given, else from y, else -- unit without a
 * residual -- it is rebuilt from z and beta_for_mask. */
int hp_bn_backward(const void* dy, const float* y, const void* z, void* g_out, void* dz, long M, int C,
                   const float* mean, const float* rstd, const float* gamma, const float* beta_for_mask, int relu,
                   int train, float* dgamma, float* dbeta, const unsigned char* relu_mask, void* workspace, int io,
                   void* stream);
/* hp_bn_backward of a unit WITHOUT residual whose two sums were already taken by the data gradient that produced dy
 * (hp_conv3d_backward_data_bnsums): only the coefficient kernel and the apply pass dz = a g + b z + c run; the ReLU mask is
 * relu_mask (a residual unit's byte mask) if given, else rebuilt from z and beta_for_mask as in hp_bn_backward.  workspace: hp_bn_backward_workspace_bytes(C). */
int hp_bn_backward_presummed(const void* dy, const void* z, void* dz, long M, int C, const float* mean, const float* rstd,
                             const float* gamma, const float* beta_for_mask, int relu, int train, float* dgamma, float* dbeta,
                             const unsigned char* relu_mask, const double* sums, void* workspace, int io, void* stream);
/* Two BatchNorm units fed by the same gradient g = dy (.) relu_mask -- bn3 of the main branch (a) and the BatchNorm
 * of the shortcut convolution (b) of a Bottleneck with `downsample` (posenet3d_50.py:86-93): one reduction and one
 * apply pass serve both (dy and the mask are read once per pass instead of twice).  Same arithmetic per unit as
 * hp_bn_backward; C <= 1024.  workspace: 2 * (hp_bn_backward_workspace_bytes(C) rounded up to 16) bytes. */
int hp_bn_backward_dual(const void* dy, const unsigned char* relu_mask, long M, int C, const void* z_a, void* dz_a,
                        const float* mean_a, const float* rstd_a, const float* gamma_a, int train_a, float* dgamma_a,
                        float* dbeta_a, const void* z_b, void* dz_b, const float* mean_b, const float* rstd_b,
                        const float* gamma_b, int train_b, float* dgamma_b, float* dbeta_b, void* workspace, int io,
                        void* stream);
/* fp32 <-> bf16 copies of n elements (n % 4 == 0): the stem's fused BN+ReLU+pool kernels stay fp32; its pooled output
 * and the gradient that comes back to it cross the bf16 boundary through these. */
int hp_cast_f32_to_bf16(const float* x, void* y, long n, void* stream);
int hp_cast_bf16_to_f32(const void* x, float* y, long n, void* stream);
/* MaxPool3d(kernel 3, stride 2, padding 1) (posenet3d_50.py:184), channels-last. */
int hp_maxpool3d_k3s2_forward(const float* x, float* y, int B, int D, int H, int W, int C, void* stream);
int hp_maxpool3d_k3s2_backward(const float* x, const float* y, const float* dy, float* dx, int B, int D, int H, int W,
                               int C, void* stream);
/* Stem: BatchNorm3d + ReLU + MaxPool3d(3,2,1) fused (posenet3d_50.py:253-257); the normalised 64-channel
 * full-resolution volume is never written.  z (B,D,H,W,C) is the raw stem convolution output. */
size_t hp_stem_bn_pool_workspace_bytes(int C);
int hp_stem_bn_relu_pool_forward(const float* z, float* pooled, int B, int D, int H, int W, int C, const float* mean,
                                 const float* rstd, const float* gamma, const float* beta, void* workspace, void* stream);
int hp_stem_bn_relu_pool_backward(const float* z, const float* pooled, const float* dpooled, float* dz, int B, int D, int H,
                                  int W, int C, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                  int train, float* dgamma, float* dbeta, void* workspace, void* stream);
/* [B][V][C] -> [B][C][V] (to_channels_first = 1) or back (0). */
int hp_layout_transpose(const float* in, float* out, int B, long V, int C, int to_channels_first, void* stream);

/* ------------------------------------------------------------------------
 * Thin-channel 3x3x3 convolutions, planar (B, C, D, H, W) fp32, stride 1, "same" size:
 * FeatureExtraction / ResConv3D (models/feature_extraction.py:147-158,167,228-256;
 * replicate_pad = 1 for the ReplicationPad3d(1)+Conv3d pairs, 0 for the zero-padded box
 * filter) and UNet3d's DoubleConv convolutions (unet/unet3d.py:15-23).
 * Weights in the torch layout (Cout, Cin, 3, 3, 3); any channel counts (4-wide matrix-core blocks, padded).
 * ---------------------------------------------------------------------- */
int hp_dconv3_forward(const float* x, const float* w, const float* bias, float* y, int B, int cin, int cout, int D,
                      int H, int W, int replicate_pad, void* stream);
/* the same convolution with a fused epilogue: y = leaky(conv(x) + bias [+ residual], slope) (slope 1 = none;
 * ResConv3D, feature_extraction.py:228-256) and, when stats != NULL, per (b, co) {sum y, sum y^2} as doubles
 * (2*B*cout, zeroed here) for the GroupNorm that follows (unet3d.py:17-18) */
int hp_dconv3_forward_fused(const float* x, const float* w, const float* bias, const float* residual, float* y,
                            double* stats, int B, int cin, int cout, int D, int H, int W, int replicate_pad, float slope,
                            void* stream);
size_t hp_dconv3_backward_data_workspace_bytes(int B, int cin, int D, int H, int W, int replicate_pad);
int hp_dconv3_backward_data(const float* gy, const float* w, float* gx, int B, int cin, int cout, int D, int H, int W,
                            int replicate_pad, void* workspace, void* stream);
/* The same two calls with an arithmetic choice (BASELINE configs[2], "bf16 with fp32 LCT": the U-Net's convolutions on
 * the bf16 matrix cores).  precision = HP_PRECISION_FP32: identical to the calls above.  HP_PRECISION_BF16: x (or gy) and
 * w are rounded to bf16 (nearest even) on their way into LDS, products are exact, accumulation is fp32
 * (v_mfma_f32_4x4x4_16b_bf16: four input channels per instruction); tensors stay fp32 in memory.  Single-channel
 * (1 -> 1) layers always run exact. */
int hp_dconv3_forward_fused_p(const float* x, const float* w, const float* bias, const float* residual, float* y,
                              double* stats, int B, int cin, int cout, int D, int H, int W, int replicate_pad, float slope,
                              int precision, void* stream);
int hp_dconv3_backward_data_p(const float* gy, const float* w, float* gx, int B, int cin, int cout, int D, int H, int W,
                              int replicate_pad, int precision, void* workspace, void* stream);
/* weight gradient with the same arithmetic choice: HP_PRECISION_BF16 rounds x and gy to bf16 (K = four consecutive voxels
 * per v_mfma_f32_4x4x4_16b_bf16), accumulates in fp32 and sums the bias gradient exactly; layers with cin == 1, W % 4 != 0
 * or a gy not aligned to 16 bytes run exact.  Workspace as for hp_dconv3_backward_weight. */
int hp_dconv3_backward_weight_p(const float* x, const float* gy, float* dw, float* dbias, int B, int cin, int cout, int D,
                                int H, int W, int replicate_pad, int precision, void* workspace, void* stream);
/* dw (Cout,Cin,3,3,3) and dbias (Cout, may be NULL) are overwritten.  workspace (device, sized by the query)
 * holds per-workgroup partial sums that a second kernel adds in a fixed order: no atomics, run-to-run
 * bit-identical. */
size_t hp_dconv3_backward_weight_workspace_bytes(int B, int cin, int cout, int D, int H, int W);
int hp_dconv3_backward_weight(const float* x, const float* gy, float* dw, float* dbias, int B, int cin, int cout, int D,
                              int H, int W, int replicate_pad, void* workspace, void* stream);

/* ------------------------------------------------------------------------
 * UNet3d memory-bound stages (unet/unet3d.py), planar (B, C, D, H, W) fp32; V = D*H*W.
 * ---------------------------------------------------------------------- */
size_t hp_groupnorm_workspace_bytes(int B, int C);
/* y = relu(GroupNorm_G(z) * gamma + beta)  (unet3d.py:17-18,22-23); mean/rstd: B*G floats kept for backward */
int hp_groupnorm_relu_forward(const float* z, float* y, int B, int C, int G, long V, const float* gamma,
                              const float* beta, float eps, float* mean, float* rstd, void* workspace, void* stream);
/* the same with (a) scale/shift, B*C floats each: the per-(b,c) affine map y = relu(z*scale + shift), kept for backward;
 * (b) optional chan_stats: per (b,c) {sum z, sum z^2} as doubles from the producing convolution's epilogue
 * (hp_dconv3_forward_fused) -- the statistics pass over z is then skipped */
int hp_groupnorm_relu_forward_v2(const float* z, float* y, int B, int C, int G, long V, const float* gamma,
                                 const float* beta, float eps, const double* chan_stats, float* mean, float* rstd,
                                 float* scale, float* shift, void* workspace, void* stream);
/* backward of the above; the ReLU mask is rebuilt from z with the forward's scale/shift (y is not read) */
int hp_groupnorm_relu_backward_v2(const float* dy, const float* z, float* dz, int B, int C, int G, long V,
                                  const float* gamma, const float* mean, const float* rstd, const float* scale,
                                  const float* shift, float* dgamma, float* dbeta, void* workspace, void* stream);
/* MaxPool3d(2,2) (unet3d.py:35); planes = B*C */
int hp_maxpool3d_k2_forward(const float* x, float* y, long planes, int D, int H, int W, void* stream);
int hp_maxpool3d_k2_backward(const float* x, const float* dy, float* dx, long planes, int D, int H, int W, void* stream);
/* the same with a second gradient of x summed in the same pass: the skip tensor of a U-Net level feeds the pool and the
 * decoder's concatenation (unet3d.py:31-39, 42-62).  add: (B, >= C, D, H, W) read in place -- sample b, channel c at
 * add + b * add_batch_stride + c * D*H*W (the first C channels of the concatenation's gradient) */
int hp_maxpool3d_k2_backward_add(const float* x, const float* dy, const float* add, long add_batch_stride, float* dx, int B, int C,
                                 int D, int H, int W, void* stream);
/* nn.Upsample(scale_factor=2, mode='trilinear', align_corners=True) (unet3d.py:47), written into /
 * read from channel slice [c_off, c_off+C) of a (B, Ctot, 2D, 2H, 2W) tensor: the torch.cat of :61 */
int hp_upsample_trilinear2x_forward(const float* x, float* y, int B, int C, int D, int H, int W, int Ctot, int c_off,
                                    void* stream);
/* The same interpolation as three 1-D passes in the reference's order (w, h, d: identical arithmetic) through a
 * caller-provided workspace: two loads per element and pass, 16-byte stores into the concat slice. */
size_t hp_upsample_trilinear2x_forward_workspace_bytes(int B, int C, int D, int H, int W);
int hp_upsample_trilinear2x_forward_ws(const float* x, float* y, int B, int C, int D, int H, int W, int Ctot, int c_off,
                                       void* workspace, void* stream);
int hp_upsample_trilinear2x_backward(const float* dy, float* dx, int B, int C, int D, int H, int W, int Ctot, int c_off,
                                     void* stream);
/* The same adjoint as three 1-D passes (w, h, d) through a caller-provided workspace: 5 loads per element and pass instead
 * of up to 125 in the fused gather (falls back to it for extents the passes do not cover). */
size_t hp_upsample_trilinear2x_backward_workspace_bytes(int B, int C, int D, int H, int W);
int hp_upsample_trilinear2x_backward_ws(const float* dy, float* dx, int B, int C, int D, int H, int W, int Ctot, int c_off,
                                        void* workspace, void* stream);
/* (B,C,V) -> channel slice of (B,Ctot,V) (gather = 0) or the reverse (gather = 1) */
int hp_channel_slice_copy(const float* src, float* dst, int B, int C, long V, int Ctot, int c_off, int gather,
                          void* stream);
/* 1x1x1 convolution of UNet3d's `Out` (unet3d.py:65-71) */
int hp_conv1x1_forward(const float* x, const float* w, const float* bias, float* y, int B, int cin, int cout, long V,
                       void* stream);
int hp_conv1x1_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B, int cin,
                        int cout, long V, void* stream);
/* The same convolution with the sum `y + addend` written in the same pass (models/NlosPose.py:57: `feature + refine`, the
 * regressor's input, leaves with the refined volume), and its backward with a second incoming gradient dy2 of y (the one
 * that arrives through that sum) added on load.  addend / sum_out and dy2 may be NULL. */
int hp_conv1x1_forward_sum(const float* x, const float* w, const float* bias, const float* addend, float* y, float* sum_out,
                           int B, int cin, int cout, long V, void* stream);
int hp_conv1x1_backward_sum(const float* x, const float* w, const float* dy, const float* dy2, float* dx, float* dw, float* db,
                            int B, int cin, int cout, long V, void* stream);

/* ------------------------------------------------------------------------
 * Elementwise / reduction stages around the convolutions.
 * ---------------------------------------------------------------------- */
/* y = leaky_relu(a [+ b], slope); slope = 1 is a plain add (feature_extraction.py:170,252-255; NlosPose.py:57) */
int hp_leaky_add_forward(const float* a, const float* b, float* y, long n, float slope, void* stream);
int hp_leaky_backward(const float* dy, const float* y, float* g, long n, float slope, void* stream);
/* normalize_feature (feature_propagation.py:273-286): per volume (x - min)/(max(x - min) + 1e-15) * gain, NO ReLU.
 * keys: 2*nvol uint64 (min/max value+index) kept for backward; backward workspace: 2*nvol doubles */
int hp_normalize_feature_forward(const float* x, float* y, int nvol, long V, float gain, void* keys, void* stream);
int hp_normalize_feature_backward(const float* dy, const float* x, float* dx, int nvol, long V, float gain,
                                  const void* keys, void* workspace, void* stream);
/* softmax_integral_tensor (utils/criterion.py:96-153): heat (BJ, D, H, W) -> joints (BJ, 3) = E[w], E[h], E[d]
 * in voxel units; stat: 2*BJ floats (max, sum) kept for backward */
int hp_softargmax_forward(const float* heat, float* joints, float* stat, int BJ, int D, int H, int W, void* stream);
int hp_softargmax_backward(const float* heat, const float* joints, const float* stat, const float* gjoints, float* dheat,
                           int BJ, int D, int H, int W, void* stream);
/* weighted_mse_loss (utils/criterion.py:156-162): loss = sum((pred - gt)^2 * weights) * scale, scale = 1/B */
int hp_weighted_mse_forward(const float* pred, const float* gt, const float* weights, long n, float scale, float* loss,
                            void* stream);
int hp_weighted_mse_backward(const float* pred, const float* gt, const float* weights, const float* gloss, long n, float scale,
                             float* dpred, void* stream);
/* VisibleNet's projection (models/feature_propagation.py:303-310): per (plane, pixel) the 4 largest values along depth
 * (descending) and their depth coordinate (D-1-idx)/(D-1); x (planes, D, HW) -> vals, dep (planes, 4, HW) */
int hp_depth_top4(const float* x, float* vals, float* dep, long planes, int D, long HW, void* stream);
/* addnoise_dataset (utils/nlos_pose_dataloader_noise.py:167-172): replicate-border Gaussian blur of the flattened
 * measurement with 2*radius+1 normalised taps, then (do_poisson) one Poisson draw per sample, mean = blurred value;
 * sample i is a function of (seed, i) only */
int hp_noise_blur_poisson(const float* x, float* y, long n, const float* taps, int radius, int do_poisson,
                          unsigned long long seed, void* stream);
/* BCEDiceLoss (utils/criterion.py:348-385): BCEWithLogits(mean) + 1 - (2 sum(sig t) + eps)/(sum sig + sum t) over all
 * n elements.  acc: 4 doubles kept for backward. */
int hp_bce_dice_forward(const float* logit, const float* target, long n, float eps, double* acc, float* loss,
                        void* stream);
int hp_bce_dice_backward(const float* logit, const float* target, const double* acc, const float* gloss, float* dlogit,
                         long n, float eps, void* stream);
/* The same loss with the Dice sums taken over a batch that spans several devices (data parallelism; the reference's
 * Dice is batch-global, utils/criterion.py:358-368): `partial` leaves {sum bce, sum sig t, sum sig, sum t} of the
 * LOCAL n elements in acc[0..3]; the caller sums acc[1..3] over all ranks (one 3-scalar all-reduce); `finalize`
 * gives loss = acc[0]/n + 1 - (2 acc[1] + eps)/(acc[2] + acc[3]); `backward_scaled` multiplies the Dice part of the
 * gradient by dice_scale (= world size when the gradients are afterwards AVERAGED over ranks). */
int hp_bce_dice_partial(const float* logit, const float* target, long n, double* acc, void* stream);
int hp_bce_dice_finalize(const double* acc, long n, float eps, float* loss, void* stream);
int hp_bce_dice_backward_scaled(const float* logit, const float* target, const double* acc, const float* gloss,
                                float* dlogit, long n, float eps, float dice_scale, void* stream);

/* ------------------------------------------------------------------------
 * NlosPoseSformer inference (models/NlosPoseSformer.py; BASELINE config 5).  Linear layers use
 * hp_conv3d_forward with k = 1 (a 1x1x1 convolution over channels-last rows is a Linear; its packed
 * weight layout equals the torch (out, in) layout).
 * ---------------------------------------------------------------------- */
/* nn.Linear with a fused residual: y (M, N) = x (M, K) @ w (N, K)^T + bias + addend.  bias and addend may be
 * NULL; y may alias addend (x = x + f(x) in place, :117-118).  The implicit-GEMM kernel of hp_conv3d_forward with
 * k = 1; precision as in hp_conv_desc. */
int hp_linear_forward(const float* x, const float* w, const float* bias, const float* addend, float* y, long M, int K, int N,
                      int precision, void* stream);
/* The feed-forward's first Linear with its GEGLU (models/NlosPoseSformer.py:252-262: `x, gates = u.chunk(2, dim=-1);
 * x * gelu(gates)`) in the GEMM's epilogue: u (M, 2*hidden) is never written.  w (N2 = 2*hidden, K) and bias (N2, may be
 * NULL) are the Linear's OWN parameters, as they lie in its state_dict: the kernel pairs value row 64 t + c with gate row
 * hidden + 64 t + c inside its weight gather (one 128-column tile = 64 values + their 64 gates), so no reordered copy of the
 * weights exists (round 3 took a pre-paired copy, which a `param.data` write could leave stale).  y: (M, hidden).
 * hidden % 64 == 0, i.e. N2 % 128 == 0.  Same arithmetic choices as hp_linear_forward. */
int hp_linear_geglu_forward(const float* x, const float* w, const float* bias, float* y, long M, int K, int N2,
                            int precision, void* stream);
/* rearrange 'b f c (h p1) (w p2) -> (b f h w) (p1 p2 c)'  (:104) */
int hp_sformer_patchify(const float* video, float* tokens, int B, int frames, int C, int H, int W, int patch,
                        void* stream);
/* nn.LayerNorm over the last dim (:185-194, :76-79).  rows_per_batch > 0 selects rows
 * (r / rows_per_batch) * batch_stride_rows + r % rows_per_batch of x (the joint tokens of every batch). */
int hp_layernorm_forward(const float* x, float* y, long rows, int dim, const float* gamma, const float* beta, float eps,
                         int rows_per_batch, long batch_stride_rows, void* stream);
/* nn.GELU() (erf form; models/tokenpose.py:271-277 FeedForward), elementwise over n values */
int hp_gelu_forward(const float* x, float* y, long n, void* stream);
/* GEGLU (:197-201): g = u[:, :hidden] * gelu(u[:, hidden:]) with the exact (erf) GELU */
int hp_geglu_forward(const float* u, float* g, long rows, int hidden, void* stream);
/* chunk(3) + 'b n (h d) -> (b h) n d' + q * scale + axial RoPE on the patch tokens (:160-172, :298-313).
 * K0 receives the keys WITHOUT the rotary embedding: the joint queries attend before it is applied (:305). */
int hp_sformer_qkv_prepare(const float* qkv, float* Q, float* K, float* K0, float* V, int B, int Ntok, int heads, int dh,
                           int num_joints, int patches_per_frame, float scale, const float* sin_t, const float* cos_t,
                           int rot_dim, void* stream);
/* spatial attention with joint tokens (:284-319): joint queries attend to all tokens, patch queries to
 * [joint tokens | patches of their frame]; out (B, Ntok, heads*dh) with heads merged. */
size_t hp_sformer_attention_workspace_bytes(int B, int heads, int dh);
/* precision: HP_PRECISION_FP32 (exact-fp32 MFMA), HP_PRECISION_BF16 or HP_PRECISION_FP16 (patch-token attention with
 * bf16 / fp16 operands on the 16-bit matrix cores, fp32 soft-max and accumulation, dim_head 32; the 24 joint queries stay
 * fp32).  HP_PRECISION_FP16 is BASELINE configs[4]'s "MFMA fp16 attention" (models/NlosPoseSformer.py:284-319). */
int hp_sformer_attention(const float* Q, const float* K, const float* K0, const float* V, float* out, int B, int heads,
                         int dh, int Ntok, int num_joints, int patches_per_frame, int frames, int precision,
                         void* workspace, void* stream);

/* ------------------------------------------------------------------------
 * Measurement ingest (utils/nlos_pose_dataloader.py:71-144, utils/loadrealdata.py:6-15): the per-sample
 * CPU work of the reference's Dataset.__getitem__, moved to the device.
 * ---------------------------------------------------------------------- */
/* HOST function: Radiance .hdr container (what cv2.imread(file, -1) parses at :74) -> flat R,G,B,E bytes,
 * row-major (height, width, 4).  Header "#?...", "FORMAT=32-bit_rle_rgbe", blank line, "-Y H +X W";
 * scanlines either new-style run-length encoded (2, 2, W_hi, W_lo prefix) or flat.  Call with rgbe = NULL
 * to query width/height.  No device is touched. */
int hp_rgbe_decode(const unsigned char* file, size_t nbytes, int* width, int* height, unsigned char* rgbe,
                   size_t rgbe_capacity);
/* DEVICE: flat RGBE image of `frames` stacked (H, W) frames -> normalised gray transient volume:
 *   float BGR = mantissa * 2^(e-136); / max; gray = 0.114 B + 0.587 G + 0.299 R; / max    (:74-83)
 *   'reshape (frames H) W -> frames H W', keep the first keep_frames                        (:107)
 *   (m[::2] + m[1::2]) / 2 along time, then downsample_cnt rounds of t, h, w pair averages (:113-117)
 * meas: (keep_frames / 2^(cnt+1), H / 2^cnt, W / 2^cnt) fp32.  maxima: 2 floats of device scratch that
 * receive the two global maxima (the caller may read maxima[0] to apply the reference's
 * 'abs(meas.max()) < 1e-10 -> wrong Meas File' rule at :75). */
int hp_ingest_rgbe_to_meas(const unsigned char* rgbe, int frames, int H, int W, int keep_frames, int downsample_cnt,
                           float* meas, float* maxima, void* stream);
/* Noise variant of the ingest (utils/nlos_pose_dataloader_noise.py:86-118: gray of the RAW image -- its first "/ max" is
 * commented out at :92 --, addnoise_dataset, "/ max" of the noisy image, then the same crop and pyramid):
 *   hp_ingest_rgbe_to_gray   RGBE bytes -> gray[npx] = (0.114 B + 0.587 G) + 0.299 R of the decoded floats; maxima[0] = the
 *                            largest decoded channel value (the :88 'wrong Meas File' test)
 *   hp_noise_blur_poisson    addnoise_dataset on the flattened gray image (declared above)
 *   hp_ingest_image_to_meas  image / max(image) -> '(t h) w -> t h w'[:keep_frames] -> time pairs -> box rounds, as
 *                            hp_ingest_rgbe_to_meas does from the bytes.  float64 != 0: divide and average in double and
 *                            round once at the end (NumPy on the int64 Poisson counts); 0: float32 throughout (a float32
 *                            image, e.g. blur only).  maxima[0] receives the image maximum. */
int hp_ingest_rgbe_to_gray(const unsigned char* rgbe, long npx, float* gray, float* maxima, void* stream);
int hp_ingest_image_to_meas(const float* image, int frames, int H, int W, int keep_frames, int downsample_cnt, int float64,
                            float* meas, float* maxima, void* stream);
/* One round of the reference's box pyramid on a float volume addressed by element strides:
 * out[d,h,w] = pair averages along d, then h, then w (each (a+b)/2), contiguous (D/2, H/2, W/2)  (:114-121) */
int hp_box_downsample_round(const float* in, float* out, int D, int H, int W, long stride_d, long stride_h,
                            long stride_w, void* stream);
/* (m[::2] + m[1::2]) / 2 along the leading axis of a strided volume -> contiguous (D/2, H, W).  With the
 * strides of 'h w t -> t w h' this is loadrealdata.py:9-10 without materialising the rearranged array. */
int hp_pair_average_axis0(const float* in, float* out, int D, int H, int W, long stride_d, long stride_h, long stride_w,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIDDENPOSE_HIP_H */
