// HP_BUILD: -ffp-contract=off
// Measurement ingest on the device: what utils/nlos_pose_dataloader.py:71-144
// does per sample on a CPU worker with OpenCV + NumPy -- Radiance .hdr -> float BGR -> /max -> gray ->
// /max -> (600,256,256)[:512] -> pairwise time average -> DAWNSAMPLE_CNT rounds of 2x2x2 box averaging --
// and the same box pyramid for the ground-truth volume and for utils/loadrealdata.py:6-15.
//
// Host side (file parsing, no arithmetic): header + run-length expansion of the .hdr into flat RGBE bytes (rgbe_host.cpp).
// Device side: two max reductions over the whole image (both normalisations are global) and ONE fused
// pass that decodes, normalises, converts to gray, crops and averages: every output voxel reads its
// 2^(cnt+1) x 2^cnt x 2^cnt pixels straight from the RGBE bytes (4 B/pixel) and nothing intermediate is
// written.  HBM-bound: 4 B read per kept input pixel per pass.
//
// Arithmetic order is the reference's, operation by operation in float32 with no FMA contraction, so the
// result is bit-identical to the NumPy evaluation (oracle/ingest_oracle.py):
//   v = mantissa * 2^(e-136)            (OpenCV rgbe2float: no +0.5, e == 0 -> 0)
//   v1 = v / M1,  M1 = max over all channels and pixels of v
//   gray = (0.114f*B1 + 0.587f*G1) + 0.299f*R1      (cv2.COLOR_BGR2GRAY on float32)
//   p = gray / M2, M2 = max gray
//   time pairs (a+b)/2, then per round: t pairs, h pairs, w pairs, each (a+b)/2.
#include <algorithm>
#include <cstdint>
#include <cstring>

#include "hp_internal.h"

// hipcc contracts a*b+c into an FMA by default (HIP's __fmul_rn / __fadd_rn are plain operators, and the
// backend fuses across the fp-contract pragma): this file reproduces NumPy float32 arithmetic bit for bit,
// so hiddenpose_amd/build.py compiles it with the flag named on the first line.

namespace hp {

__device__ __forceinline__ float rgbe_scale(unsigned e) {
  // ldexp(1, e - 136) as float32; e in 1..255 -> exponent -135..119: subnormal below -126
  return e ? ldexpf(1.0f, (int)e - 136) : 0.f;
}

// gray of pixel `px` (packed R,G,B,E little-endian word) after the first normalisation
__device__ __forceinline__ float gray_of(unsigned px, float m1) {
  const float f = rgbe_scale(px >> 24);
  const float r = __fdiv_rn(__fmul_rn((float)(px & 255u), f), m1);
  const float g = __fdiv_rn(__fmul_rn((float)((px >> 8) & 255u), f), m1);
  const float b = __fdiv_rn(__fmul_rn((float)((px >> 16) & 255u), f), m1);
  return __fadd_rn(__fadd_rn(__fmul_rn(0.114f, b), __fmul_rn(0.587f, g)), __fmul_rn(0.299f, r));
}

// values are >= 0, so the IEEE bit pattern orders like the value
__device__ __forceinline__ void block_max_to(float v, float* out) {
  __shared__ float red[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    float m = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
    atomicMax((unsigned*)out, __float_as_uint(m));
  }
}

// pass 1: M1 = max decoded channel value; pass 2 (GRAY): M2 = max gray(v / M1)
template <bool GRAY>
__global__ __launch_bounds__(256) void k_rgbe_max(const unsigned* __restrict__ px, long n, const float* __restrict__ m1p,
                                                  float* __restrict__ out) {
  float m = 0.f;
  const float m1 = GRAY ? *m1p : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned p = px[i];
    if constexpr (GRAY) {
      m = fmaxf(m, gray_of(p, m1));
    } else {
      const unsigned mant = max(max(p & 255u, (p >> 8) & 255u), (p >> 16) & 255u);
      m = fmaxf(m, __fmul_rn((float)mant, rgbe_scale(p >> 24)));
    }
  }
  block_max_to(m, out);
}

struct IngestGeom {
  int H, W;       // image rows per frame, columns
  int To, Ho, Wo; // output volume
};

// level-L value at (t, h, w) of the reference's pyramid; level 0 = time-pair average of normalised gray
template <int L>
__device__ __forceinline__ float pyramid(const unsigned* __restrict__ px, const IngestGeom& g, float m1, float m2, int t,
                                         int h, int w) {
  if constexpr (L == 0) {
    const long row = (long)g.H * g.W;
    const long o = (long)(2 * t) * row + (long)h * g.W + w;
    const float a = __fdiv_rn(gray_of(px[o], m1), m2);
    const float b = __fdiv_rn(gray_of(px[o + row], m1), m2);
    return __fmul_rn(__fadd_rn(a, b), 0.5f);
  } else {
    float vw[2];
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      float vh[2];
#pragma unroll
      for (int dh = 0; dh < 2; ++dh) {
        const float a = pyramid<L - 1>(px, g, m1, m2, 2 * t, 2 * h + dh, 2 * w + dw);
        const float b = pyramid<L - 1>(px, g, m1, m2, 2 * t + 1, 2 * h + dh, 2 * w + dw);
        vh[dh] = __fmul_rn(__fadd_rn(a, b), 0.5f);
      }
      vw[dw] = __fmul_rn(__fadd_rn(vh[0], vh[1]), 0.5f);
    }
    return __fmul_rn(__fadd_rn(vw[0], vw[1]), 0.5f);
  }
}

template <int CNT>
__global__ __launch_bounds__(256) void k_rgbe_to_meas(const unsigned* __restrict__ px, float* __restrict__ out, IngestGeom g,
                                                      const float* __restrict__ mx) {
  const long n = (long)g.To * g.Ho * g.Wo;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w = (int)(i % g.Wo), h = (int)((i / g.Wo) % g.Ho), t = (int)(i / ((long)g.Wo * g.Ho));
  out[i] = pyramid<CNT>(px, g, mx[0], mx[1], t, h, w);
}

// one reference round on a float volume: t pairs, then h pairs, then w pairs, each (a+b)/2
__global__ __launch_bounds__(256) void k_box_round(const float* __restrict__ in, float* __restrict__ out, int Do, int Ho,
                                                   int Wo, long sd, long sh, long sw) {
  const long n = (long)Do * Ho * Wo;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w = (int)(i % Wo), h = (int)((i / Wo) % Ho), d = (int)(i / ((long)Wo * Ho));
  float vw[2];
#pragma unroll
  for (int dw = 0; dw < 2; ++dw) {
    float vh[2];
#pragma unroll
    for (int dh = 0; dh < 2; ++dh) {
      const float* p = in + (long)(2 * d) * sd + (long)(2 * h + dh) * sh + (long)(2 * w + dw) * sw;
      vh[dh] = __fmul_rn(__fadd_rn(p[0], p[sd]), 0.5f);
    }
    vw[dw] = __fmul_rn(__fadd_rn(vh[0], vh[1]), 0.5f);
  }
  out[i] = __fmul_rn(__fadd_rn(vw[0], vw[1]), 0.5f);
}

// leading-axis pair average with arbitrary input strides (the first step of both loaders)
__global__ __launch_bounds__(256) void k_pair_avg(const float* __restrict__ in, float* __restrict__ out, int Do, int Ho, int Wo,
                                                  long sd, long sh, long sw) {
  const long n = (long)Do * Ho * Wo;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w = (int)(i % Wo), h = (int)((i / Wo) % Ho), d = (int)(i / ((long)Wo * Ho));
  const float* p = in + (long)(2 * d) * sd + (long)h * sh + (long)w * sw;
  out[i] = __fmul_rn(__fadd_rn(p[0], p[sd]), 0.5f);
}

// ---------------------------------------------------------------- noise variant (utils/nlos_pose_dataloader_noise.py:86-118)
// The noise dataset converts the RAW float image to gray (its first "/ max" is commented out, :92), perturbs it
// (addnoise_dataset: hp_noise_blur_poisson) and only then normalises by the global maximum of the noisy image, so the gray
// image exists in memory between two kernels:
//   k_rgbe_gray_raw      RGBE bytes -> gray (0.114 B + 0.587 G) + 0.299 R of the decoded floats, and the :88 maximum
//   k_image_max          maximum of the perturbed image (values >= 0)
//   k_image_to_meas<R>   / max, '(t h) w -> t h w'[:keep], time pairs and the box rounds.  R = double when the image
//                        holds Poisson counts (numpy: int64 / int64 max -> float64, averaged in float64, cast at the
//                        end), float for a blur-only image (float32 throughout)
__global__ __launch_bounds__(256) void k_rgbe_gray_raw(const unsigned* __restrict__ px, long n, float* __restrict__ gray,
                                                       float* __restrict__ mx) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned p = px[i];
    const float f = rgbe_scale(p >> 24);
    const float r = __fmul_rn((float)(p & 255u), f), g = __fmul_rn((float)((p >> 8) & 255u), f),
                b = __fmul_rn((float)((p >> 16) & 255u), f);
    gray[i] = __fadd_rn(__fadd_rn(__fmul_rn(0.114f, b), __fmul_rn(0.587f, g)), __fmul_rn(0.299f, r));
    m = fmaxf(m, fmaxf(fmaxf(r, g), b));
  }
  block_max_to(m, mx);
}

__global__ __launch_bounds__(256) void k_image_max(const float* __restrict__ x, long n, float* __restrict__ mx) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, x[i]);
  block_max_to(m, mx);
}

template <typename R, int L>
__device__ __forceinline__ R image_pyramid(const float* __restrict__ img, const IngestGeom& g, R mx, int t, int h, int w) {
  if constexpr (L == 0) {
    const long row = (long)g.H * g.W;
    const long o = (long)(2 * t) * row + (long)h * g.W + w;
    return ((R)img[o] / mx + (R)img[o + row] / mx) / (R)2;
  } else {
    R vw[2];
#pragma unroll
    for (int dw = 0; dw < 2; ++dw) {
      R vh[2];
#pragma unroll
      for (int dh = 0; dh < 2; ++dh)
        vh[dh] = (image_pyramid<R, L - 1>(img, g, mx, 2 * t, 2 * h + dh, 2 * w + dw) +
                  image_pyramid<R, L - 1>(img, g, mx, 2 * t + 1, 2 * h + dh, 2 * w + dw)) / (R)2;
      vw[dw] = (vh[0] + vh[1]) / (R)2;
    }
    return (vw[0] + vw[1]) / (R)2;
  }
}

template <typename R, int CNT>
__global__ __launch_bounds__(256) void k_image_to_meas(const float* __restrict__ img, float* __restrict__ out, IngestGeom g,
                                                       const float* __restrict__ mx) {
  const long n = (long)g.To * g.Ho * g.Wo;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w = (int)(i % g.Wo), h = (int)((i / g.Wo) % g.Ho), t = (int)(i / ((long)g.Wo * g.Ho));
  out[i] = (float)image_pyramid<R, CNT>(img, g, (R)mx[0], t, h, w);
}

}  // namespace hp

using namespace hp;

extern "C" int hp_ingest_rgbe_to_meas(const unsigned char* rgbe, int frames, int H, int W, int keep_frames,
                                      int downsample_cnt, float* meas, float* maxima, void* stream) {
  HP_REQUIRE(rgbe && meas && maxima, "hp_ingest_rgbe_to_meas: null argument");
  HP_REQUIRE(frames >= 1 && H >= 1 && W >= 1 && keep_frames >= 2 && keep_frames <= frames, "ingest: bad frame counts");
  HP_REQUIRE(downsample_cnt >= 0 && downsample_cnt <= 2, "ingest: downsample_cnt must be 0, 1 or 2 (got %d)", downsample_cnt);
  const int div = 1 << downsample_cnt;
  HP_REQUIRE(keep_frames % (2 * div) == 0 && H % div == 0 && W % div == 0,
             "ingest: %d frames x %d x %d is not divisible by the averaging pyramid", keep_frames, H, W);
  HP_REQUIRE(((uintptr_t)rgbe & 3) == 0, "ingest: RGBE buffer must be 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const long npx = (long)frames * H * W;
  HP_CHECK_HIP(hipMemsetAsync(maxima, 0, 2 * sizeof(float), st));
  const unsigned nb = (unsigned)std::min<long>((npx + 255) / 256, 256 * 16);
  {
    HP_PROF("ingest_rgbe_max", st);
    hipLaunchKernelGGL((k_rgbe_max<false>), dim3(nb), dim3(256), 0, st, (const unsigned*)rgbe, npx, nullptr, maxima);
    hipLaunchKernelGGL((k_rgbe_max<true>), dim3(nb), dim3(256), 0, st, (const unsigned*)rgbe, npx, maxima, maxima + 1);
  }
  IngestGeom g{H, W, keep_frames / (2 * div), H / div, W / div};
  const long nout = (long)g.To * g.Ho * g.Wo;
  const unsigned ob = (unsigned)((nout + 255) / 256);
  {
    HP_PROF("ingest_rgbe_to_meas", st);
    switch (downsample_cnt) {
      case 0: hipLaunchKernelGGL((k_rgbe_to_meas<0>), dim3(ob), dim3(256), 0, st, (const unsigned*)rgbe, meas, g, maxima); break;
      case 1: hipLaunchKernelGGL((k_rgbe_to_meas<1>), dim3(ob), dim3(256), 0, st, (const unsigned*)rgbe, meas, g, maxima); break;
      default: hipLaunchKernelGGL((k_rgbe_to_meas<2>), dim3(ob), dim3(256), 0, st, (const unsigned*)rgbe, meas, g, maxima); break;
    }
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_box_downsample_round(const float* in, float* out, int D, int H, int W, long stride_d, long stride_h,
                                       long stride_w, void* stream) {
  HP_REQUIRE(in && out, "hp_box_downsample_round: null argument");
  HP_REQUIRE(D >= 2 && H >= 2 && W >= 2 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "box round: even sizes needed (%d,%d,%d)", D, H, W);
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)(D / 2) * (H / 2) * (W / 2);
  HP_PROF("ingest_box_round", st);
  hipLaunchKernelGGL(k_box_round, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, D / 2, H / 2, W / 2, stride_d,
                     stride_h, stride_w);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_pair_average_axis0(const float* in, float* out, int D, int H, int W, long stride_d, long stride_h,
                                     long stride_w, void* stream) {
  HP_REQUIRE(in && out, "hp_pair_average_axis0: null argument");
  HP_REQUIRE(D >= 2 && D % 2 == 0 && H >= 1 && W >= 1, "pair average: even leading size needed (%d)", D);
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)(D / 2) * H * W;
  HP_PROF("ingest_pair_avg", st);
  hipLaunchKernelGGL(k_pair_avg, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, out, D / 2, H, W, stride_d, stride_h,
                     stride_w);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_ingest_rgbe_to_gray(const unsigned char* rgbe, long npx, float* gray, float* maxima, void* stream) {
  HP_REQUIRE(rgbe && gray && maxima && npx > 0, "hp_ingest_rgbe_to_gray: bad argument");
  HP_REQUIRE(((uintptr_t)rgbe & 3) == 0, "ingest: RGBE buffer must be 4-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  HP_CHECK_HIP(hipMemsetAsync(maxima, 0, sizeof(float), st));
  const unsigned nb = (unsigned)std::min<long>((npx + 255) / 256, 256 * 16);
  HP_PROF("ingest_rgbe_gray", st);
  hipLaunchKernelGGL(k_rgbe_gray_raw, dim3(nb), dim3(256), 0, st, (const unsigned*)rgbe, npx, gray, maxima);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_ingest_image_to_meas(const float* image, int frames, int H, int W, int keep_frames, int downsample_cnt,
                                       int float64, float* meas, float* maxima, void* stream) {
  HP_REQUIRE(image && meas && maxima, "hp_ingest_image_to_meas: null argument");
  HP_REQUIRE(frames >= 1 && H >= 1 && W >= 1 && keep_frames >= 2 && keep_frames <= frames, "ingest: bad frame counts");
  HP_REQUIRE(downsample_cnt >= 0 && downsample_cnt <= 2, "ingest: downsample_cnt must be 0, 1 or 2 (got %d)", downsample_cnt);
  const int div = 1 << downsample_cnt;
  HP_REQUIRE(keep_frames % (2 * div) == 0 && H % div == 0 && W % div == 0,
             "ingest: %d frames x %d x %d is not divisible by the averaging pyramid", keep_frames, H, W);
  hipStream_t st = (hipStream_t)stream;
  const long npx = (long)frames * H * W;
  HP_CHECK_HIP(hipMemsetAsync(maxima, 0, sizeof(float), st));
  const unsigned nb = (unsigned)std::min<long>((npx + 255) / 256, 256 * 16);
  {
    HP_PROF("ingest_image_max", st);
    hipLaunchKernelGGL(k_image_max, dim3(nb), dim3(256), 0, st, image, npx, maxima);
  }
  IngestGeom g{H, W, keep_frames / (2 * div), H / div, W / div};
  const long nout = (long)g.To * g.Ho * g.Wo;
  const dim3 ob((unsigned)((nout + 255) / 256)), tb(256);
  {
    HP_PROF("ingest_image_to_meas", st);
#define HP_I2M(R_)                                                                                                  \
  switch (downsample_cnt) {                                                                                         \
    case 0: hipLaunchKernelGGL((k_image_to_meas<R_, 0>), ob, tb, 0, st, image, meas, g, maxima); break;             \
    case 1: hipLaunchKernelGGL((k_image_to_meas<R_, 1>), ob, tb, 0, st, image, meas, g, maxima); break;             \
    default: hipLaunchKernelGGL((k_image_to_meas<R_, 2>), ob, tb, 0, st, image, meas, g, maxima); break;            \
  }
    if (float64) { HP_I2M(double) } else { HP_I2M(float) }
#undef HP_I2M
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
