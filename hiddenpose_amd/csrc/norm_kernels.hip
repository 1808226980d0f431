// Memory-bound companions of the convolution engine (channels-last fp32, [M][C] matrices):
// BatchNorm3d train/eval apply with fused residual add + ReLU, its two-pass backward,
// MaxPool3d(3,2,1) forward/backward, and layout changes at the regressor boundary.
// All of them stream 16 bytes per lane; per-channel reductions go through LDS and end in
// one fp64 atomic per channel per block.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>

#include "hp_internal.h"

namespace hp {

constexpr int ET = 256;
#ifndef HP_BN_UNR_H
#define HP_BN_UNR_H 8
#endif

// Streaming hints (round 4): the BatchNorm passes read and write every activation tensor exactly once -- tens of GB per step
// that displace the operands the weight gradients on the other stream re-read through L2 / MALL.  Their 16-byte accesses carry
// the non-temporal bit (`global_load_dwordx4 ... nt`): same-box A/B/A/B at the headline shape 468.7 / 469.9 -> 466.9 / 466.8
// ms/step, and the passes themselves are not slower alone (apply 16.9 -> 16.3, reduce 8.5 -> 8.2 ms/step).  -DHP_BN_NT=0
// (HP_EXTRA_DEFS, hiddenpose_amd/build.py) builds without it.  Not for the stem's pool kernels: their windows overlap and
// live on L2 reuse.
#ifndef HP_BN_NT
#define HP_BN_NT 1
#endif
typedef float hp_f4v __attribute__((ext_vector_type(4)));
typedef unsigned hp_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
#if HP_BN_NT
  const hp_f4v t = __builtin_nontemporal_load(reinterpret_cast<const hp_f4v*>(p));
  return make_float4(t.x, t.y, t.z, t.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ uint4 ld_stream(const uint4* p) {
#if HP_BN_NT
  const hp_u4v t = __builtin_nontemporal_load(reinterpret_cast<const hp_u4v*>(p));
  return make_uint4(t.x, t.y, t.z, t.w);
#else
  return *p;
#endif
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
#if HP_BN_NT
  __builtin_nontemporal_store((hp_f4v){v.x, v.y, v.z, v.w}, reinterpret_cast<hp_f4v*>(p));
#else
  *p = v;
#endif
}
__device__ __forceinline__ void st_stream(uint4* p, uint4 v) {
#if HP_BN_NT
  __builtin_nontemporal_store((hp_u4v){v.x, v.y, v.z, v.w}, reinterpret_cast<hp_u4v*>(p));
#else
  *p = v;
#endif
}

// Element-type plumbing of the BatchNorm passes.  IOM = 0: every tensor fp32, IOM = 1: every tensor bf16 -- both known
// at compile time, so a pass issues all its loads back to back (a run-time type test per access put a conversion, and
// with it a wait, between consecutive loads: the bf16 reduce pass ran SLOWER than the fp32 one) -- and a lane moves
// 16 bytes either way: one channel quad of fp32 or two of bf16 (Q).  IOM = 2: mixed types, run-time flags, one quad.
template <int IOM>
struct QIO {
  static constexpr int Q = IOM == 1 ? 2 : 1;
  static __device__ __forceinline__ void ld(const void* p, long g, int half, float4 (&o)[Q]) {  // group g = quads g*Q ..
    if constexpr (IOM == 0) {
      o[0] = ld_stream(reinterpret_cast<const float4*>(p) + g);
    } else if constexpr (IOM == 1) {
      const uint4 u = ld_stream(reinterpret_cast<const uint4*>(p) + g);
      o[0] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                         __uint_as_float(u.y & 0xffff0000u));
      o[1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                         __uint_as_float(u.w & 0xffff0000u));
    } else {
      o[0] = hp_ld4(p, 4 * g, half);
    }
  }
  static __device__ __forceinline__ void st(void* p, long g, const float4 (&v)[Q], int half) {
    if constexpr (IOM == 0) {
      st_stream(reinterpret_cast<float4*>(p) + g, v[0]);
    } else if constexpr (IOM == 1) {
      typedef __attribute__((ext_vector_type(2))) float f2;
      typedef __attribute__((ext_vector_type(2))) __bf16 h2;
      const h2 a = __builtin_convertvector((f2){v[0].x, v[0].y}, h2), b = __builtin_convertvector((f2){v[0].z, v[0].w}, h2);
      const h2 c = __builtin_convertvector((f2){v[1].x, v[1].y}, h2), d = __builtin_convertvector((f2){v[1].z, v[1].w}, h2);
      st_stream(reinterpret_cast<uint4*>(p) + g, make_uint4(__builtin_bit_cast(unsigned int, a), __builtin_bit_cast(unsigned int, b),
                                                           __builtin_bit_cast(unsigned int, c), __builtin_bit_cast(unsigned int, d)));
    } else {
      hp_st4(p, 4 * g, v[0], half);
    }
  }
  // byte-per-quad masks of group g
  static __device__ __forceinline__ unsigned ldm(const unsigned char* m, long g) {
    if constexpr (Q == 2) return *(reinterpret_cast<const unsigned short*>(m) + g);
    else return m[g];
  }
  static __device__ __forceinline__ void stm(unsigned char* m, long g, unsigned v) {
    if constexpr (Q == 2) *(reinterpret_cast<unsigned short*>(m) + g) = (unsigned short)v;
    else m[g] = (unsigned char)v;
  }
};
__device__ __forceinline__ float4 f4_fma(float4 v, float4 a, float4 b) {
  return make_float4(fmaf(v.x, a.x, b.x), fmaf(v.y, a.y, b.y), fmaf(v.z, a.z, b.z), fmaf(v.w, a.w, b.w));
}
__device__ __forceinline__ float4 f4_mask(float4 g, unsigned mk) {
  return make_float4((mk & 1u) ? g.x : 0.f, (mk & 2u) ? g.y : 0.f, (mk & 4u) ? g.z : 0.f, (mk & 8u) ? g.w : 0.f);
}
__device__ __forceinline__ float4 f4_gate(float4 g, float4 y) {
  return make_float4(y.x > 0.f ? g.x : 0.f, y.y > 0.f ? g.y : 0.f, y.z > 0.f ? g.z : 0.f, y.w > 0.f ? g.w : 0.f);
}
// dz = a g + b z + c
__device__ __forceinline__ float4 f4_dz(float4 g, float4 v, float4 a, float4 b, float4 k) {
  return make_float4(a.x * g.x + b.x * v.x + k.x, a.y * g.y + b.y * v.y + k.y, a.z * g.z + b.z * v.z + k.z, a.w * g.w + b.w * v.w + k.w);
}

// stats[s][0:C] = sum, stats[s][C:2C] = sum of squares (fp64) over the rows of slot s
__global__ void k_bn_finalize(const double* __restrict__ stats, long M, int C, float eps, float momentum,
                              float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ running_mean,
                              float* __restrict__ running_var, long long* __restrict__ num_batches_tracked) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;  // BatchNorm3d's counter, in the same launch
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;   // the convolution epilogue's HP_STATS_SLOTS partial vectors (include/hiddenpose_hip.h)
  for (int sl = 0; sl < HP_STATS_SLOTS; ++sl) {
    s1 += stats[(size_t)sl * 2 * C + c];
    s2 += stats[(size_t)sl * 2 * C + C + c];
  }
  const double m = s1 / (double)M;
  double var = s2 / (double)M - m * m;
  if (var < 0) var = 0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
    running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
  }
}

__global__ void k_bn_eval_stats(const float* __restrict__ running_mean, const float* __restrict__ running_var, int C,
                                float eps, float* __restrict__ mean, float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  rstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// y = act((z - mean) * rstd * gamma + beta [+ res])
template <int IOM>
__global__ __launch_bounds__(ET) void k_bn_apply(const void* __restrict__ z, int z_half, const void* __restrict__ res, int res_half,
                                                 void* __restrict__ y, int y_half, long n4, int C4, const float4* __restrict__ mean,
                                                 const float4* __restrict__ rstd, const float4* __restrict__ gamma,
                                                 const float4* __restrict__ beta, int relu,
                                                 unsigned char* __restrict__ mask_out, const float4* __restrict__ rmean,
                                                 const float4* __restrict__ rrstd, const float4* __restrict__ rgamma,
                                                 const float4* __restrict__ rbeta) {
  using IO = QIO<IOM>;
  constexpr int Q = IO::Q;
  const long ng = n4 / Q;
  const int CG = C4 / Q;
  // The grid stride is usually a multiple of the channel groups per row: a thread then meets the same channels in every
  // trip and keeps their affine maps in registers (the per-trip parameter loads -- 4 to 8 per tensor load -- had the pass
  // bound by vector-memory instructions rather than by bytes, most of all with two channel quads per lane).
  const long stride = (long)gridDim.x * ET;
  const bool fixed = stride % CG == 0;
  float4 scv[Q], shv[Q], sc2v[Q], sh2v[Q];
  auto params = [&](int c0) {
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      const int c = c0 + e;
      const float4 m = mean[c], r = rstd[c], g = gamma[c], b = beta[c];
      // y = z * sc + sh with sc = rstd * gamma, sh = beta - mean * sc: the backward rebuilds the ReLU mask with
      // exactly this expression
      scv[e] = make_float4(r.x * g.x, r.y * g.y, r.z * g.z, r.w * g.w);
      shv[e] = make_float4(b.x - m.x * scv[e].x, b.y - m.y * scv[e].y, b.z - m.z * scv[e].z, b.w - m.w * scv[e].w);
      if (rmean) {  // the residual is a raw convolution output with a BatchNorm of its own still to be applied
        const float4 m2 = rmean[c], r2 = rrstd[c], g2 = rgamma[c], b2 = rbeta[c];
        sc2v[e] = make_float4(r2.x * g2.x, r2.y * g2.y, r2.z * g2.z, r2.w * g2.w);
        sh2v[e] = make_float4(b2.x - m2.x * sc2v[e].x, b2.y - m2.y * sc2v[e].y, b2.z - m2.z * sc2v[e].z, b2.w - m2.w * sc2v[e].w);
      }
    }
  };
  if (fixed) params((int)(((long)blockIdx.x * ET + threadIdx.x) % CG) * Q);
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < ng; i += stride) {
    if (!fixed) params((int)(i % CG) * Q);
    float4 v[Q], q[Q], o[Q];
    IO::ld(z, i, z_half, v);
    if (res) IO::ld(res, i, res_half, q);
    unsigned mk = 0u;
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      o[e] = f4_fma(v[e], scv[e], shv[e]);
      if (res) {
        float4 qq = q[e];
        if (rmean) qq = f4_fma(qq, sc2v[e], sh2v[e]);
        o[e].x += qq.x;
        o[e].y += qq.y;
        o[e].z += qq.z;
        o[e].w += qq.w;
      }
      if (relu) o[e] = make_float4(fmaxf(o[e].x, 0.f), fmaxf(o[e].y, 0.f), fmaxf(o[e].z, 0.f), fmaxf(o[e].w, 0.f));
      // one byte per channel quad: which outputs are positive (the backward reads this instead of y: 1 B, not 16)
      mk |= (unsigned)((o[e].x > 0.f ? 1 : 0) | (o[e].y > 0.f ? 2 : 0) | (o[e].z > 0.f ? 4 : 0) | (o[e].w > 0.f ? 8 : 0)) << (8 * e);
    }
    IO::st(y, i, o, y_half);
    if (mask_out) IO::stm(mask_out, i, mk);
  }
}

// Pass 1 of the backward: g = dy * [y > 0] (written to g_out), per channel sum(g) and
// sum(g * zhat) into red[0:C], red[C:2C] (fp64).  Thread t owns channel quad (t % C4) --
// the block strides over rows so that a thread always sees the same channels.
template <int UNR, int IOM>
__global__ __launch_bounds__(ET) void k_bn_bwd_reduce(const void* __restrict__ dy, int dy_half, const float4* __restrict__ y,
                                                      const void* __restrict__ z, int z_half, void* __restrict__ g_out, int g_half, long M,
                                                      int C4, const float4* __restrict__ mean,
                                                      const float4* __restrict__ rstd, int relu,
                                                      double* __restrict__ red, const float4* __restrict__ gamma,
                                                      const float4* __restrict__ beta, const unsigned char* __restrict__ mask) {
  using IO = QIO<IOM>;
  constexpr int Q = IO::Q;
  __shared__ float4 ssum[Q][ET], sdot[Q][ET];
  const int tid = threadIdx.x;
  const int CG = C4 / Q;                             // channel groups (Q quads each) per row
  const int lanes_per_row = CG < ET ? CG : ET;       // threads covering one row pass
  const int rows_per_pass = ET / lanes_per_row;      // >= 1 when CG <= ET
  const int cg_passes = (CG + ET - 1) / ET;          // > 1 when CG > ET
  const int my_row = tid / lanes_per_row, my_cg0 = tid % lanes_per_row;
  for (int cp = 0; cp < cg_passes; ++cp) {
    const int cg = my_cg0 + cp * ET;
    float4 s[Q], d[Q];
#pragma unroll
    for (int e = 0; e < Q; ++e) s[e] = d[e] = make_float4(0, 0, 0, 0);
    if (cg < CG && my_row < rows_per_pass) {
      float4 m[Q], r[Q], sc[Q], sh[Q];
#pragma unroll
      for (int e = 0; e < Q; ++e) {
        m[e] = mean[cg * Q + e];
        r[e] = rstd[cg * Q + e];
        // without a residual the ReLU mask follows from z alone: the forward output need not be read
        sc[e] = sh[e] = make_float4(0, 0, 0, 0);
        if (relu && !y) {
          const float4 ga = gamma[cg * Q + e], be = beta[cg * Q + e];
          sc[e] = make_float4(r[e].x * ga.x, r[e].y * ga.y, r[e].z * ga.z, r[e].w * ga.w);
          sh[e] = make_float4(be.x - m[e].x * sc[e].x, be.y - m[e].y * sc[e].y, be.z - m[e].z * sc[e].z, be.w - m[e].w * sc[e].w);
        }
      }
      // UNR rows per trip: all their loads are issued before the first use (the pass is a pure read stream and
      // needs ~15 MB in flight chip-wide to run at HBM speed; one row per trip keeps ~6 MB in flight)
      const long stride = (long)gridDim.x * rows_per_pass;
      for (long row0 = (long)blockIdx.x * rows_per_pass + my_row; row0 < M; row0 += UNR * stride) {
        float4 gq[UNR][Q], vq[UNR][Q], yq[UNR];
        unsigned mq[UNR];
        bool okq[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const long row = row0 + u * stride;
          okq[u] = row < M;
          const long i = (okq[u] ? row : row0) * CG + cg;
          IO::ld(dy, i, dy_half, gq[u]);
          IO::ld(z, i, z_half, vq[u]);
          mq[u] = (relu && mask) ? IO::ldm(mask, i) : 0u;
          if constexpr (Q == 1) yq[u] = (relu && !mask && y) ? y[i] : make_float4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (!okq[u]) continue;
          const long i = (row0 + u * stride) * CG + cg;
          float4 g[Q];
#pragma unroll
          for (int e = 0; e < Q; ++e) {
            g[e] = gq[u][e];
            const float4 v = vq[u][e];
            if (relu && mask) {
              g[e] = f4_mask(g[e], mq[u] >> (8 * e));
            } else if (relu) {
              float4 yy;
              if constexpr (Q == 1) {
                if (y) yy = yq[u];
                else yy = f4_fma(v, sc[e], sh[e]);
              } else {
                yy = f4_fma(v, sc[e], sh[e]);
              }
              g[e] = f4_gate(g[e], yy);
            }
            s[e].x += g[e].x;
            s[e].y += g[e].y;
            s[e].z += g[e].z;
            s[e].w += g[e].w;
            d[e].x += g[e].x * (v.x - m[e].x) * r[e].x;
            d[e].y += g[e].y * (v.y - m[e].y) * r[e].y;
            d[e].z += g[e].z * (v.z - m[e].z) * r[e].z;
            d[e].w += g[e].w * (v.w - m[e].w) * r[e].w;
          }
          if (g_out) IO::st(g_out, i, g, g_half);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      ssum[e][tid] = s[e];
      sdot[e][tid] = d[e];
    }
    __syncthreads();
    if (my_row == 0 && cg < CG) {
      const int C = C4 * 4;
#pragma unroll
      for (int e = 0; e < Q; ++e) {
        float4 ss = s[e], dd = d[e];
        for (int rr = 1; rr < rows_per_pass; ++rr) {
          const float4 a = ssum[e][rr * lanes_per_row + my_cg0], b = sdot[e][rr * lanes_per_row + my_cg0];
          ss.x += a.x; ss.y += a.y; ss.z += a.z; ss.w += a.w;
          dd.x += b.x; dd.y += b.y; dd.z += b.z; dd.w += b.w;
        }
        const int cq = cg * Q + e;
        atomicAdd(red + cq * 4 + 0, (double)ss.x);
        atomicAdd(red + cq * 4 + 1, (double)ss.y);
        atomicAdd(red + cq * 4 + 2, (double)ss.z);
        atomicAdd(red + cq * 4 + 3, (double)ss.w);
        atomicAdd(red + C + cq * 4 + 0, (double)dd.x);
        atomicAdd(red + C + cq * 4 + 1, (double)dd.y);
        atomicAdd(red + C + cq * 4 + 2, (double)dd.z);
        atomicAdd(red + C + cq * 4 + 3, (double)dd.w);
      }
    }
    __syncthreads();
  }
}

// dgamma = sum(g zhat), dbeta = sum(g); coefficients of pass 2:
//   dz = a * g + b * z + c   with  a = gamma rstd, b = -a rstd mean(g zhat), c = -a mean(g) - b mean
__global__ void k_bn_bwd_coef(const double* __restrict__ red, long M, int C, const float* __restrict__ mean,
                              const float* __restrict__ rstd, const float* __restrict__ gamma, int train,
                              float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ ca,
                              float* __restrict__ cb, float* __restrict__ cc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double sg = red[c], sd = red[C + c];
  if (dgamma) dgamma[c] = (float)sd;
  if (dbeta) dbeta[c] = (float)sg;
  const double a = (double)gamma[c] * rstd[c];
  if (train) {
    const double b = -a * rstd[c] * (sd / (double)M);
    ca[c] = (float)a;
    cb[c] = (float)b;
    cc[c] = (float)(-a * (sg / (double)M) - b * mean[c]);
  } else {  // eval mode: statistics are constants
    ca[c] = (float)a;
    cb[c] = 0.f;
    cc[c] = 0.f;
  }
}

template <int IOM>
__global__ __launch_bounds__(ET) void k_bn_bwd_apply(const void* __restrict__ g, int g_half, const void* __restrict__ z, int z_half,
                                                     void* __restrict__ dz, int dz_half, long n4, int C4,
                                                     const float4* __restrict__ ca, const float4* __restrict__ cb,
                                                     const float4* __restrict__ cc) {
  using IO = QIO<IOM>;
  constexpr int Q = IO::Q;
  const long ng = n4 / Q;
  const int CG = C4 / Q;
  const long stride = (long)gridDim.x * ET;
  const bool fixed = stride % CG == 0;  // see k_bn_apply
  float4 av[Q], bv[Q], kv[Q];
  auto params = [&](int c0) {
#pragma unroll
    for (int e = 0; e < Q; ++e) av[e] = ca[c0 + e], bv[e] = cb[c0 + e], kv[e] = cc[c0 + e];
  };
  if (fixed) params((int)(((long)blockIdx.x * ET + threadIdx.x) % CG) * Q);
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < ng; i += stride) {
    if (!fixed) params((int)(i % CG) * Q);
    float4 gg[Q], v[Q], o[Q];
    IO::ld(g, i, g_half, gg);
    IO::ld(z, i, z_half, v);
#pragma unroll
    for (int e = 0; e < Q; ++e) o[e] = f4_dz(gg[e], v[e], av[e], bv[e], kv[e]);
    IO::st(dz, i, o, dz_half);
  }
}

// Pass 2 for units without a residual: the masked gradient is not parked in memory by pass 1; the ReLU mask is
// rebuilt here from z (the same affine map, bit for bit), so the unit's backward moves 5 tensors instead of 6.
template <int IOM>
__global__ __launch_bounds__(ET) void k_bn_bwd_apply_mask(const void* __restrict__ dy, int dy_half, const void* __restrict__ z, int z_half,
                                                          void* __restrict__ dz, int dz_half, long n4, int C4,
                                                          const float4* __restrict__ ca, const float4* __restrict__ cb,
                                                          const float4* __restrict__ cc, const float4* __restrict__ mean,
                                                          const float4* __restrict__ rstd, const float4* __restrict__ gamma,
                                                          const float4* __restrict__ beta, int relu) {
  using IO = QIO<IOM>;
  constexpr int Q = IO::Q;
  const long ng = n4 / Q;
  const int CG = C4 / Q;
  const long stride = (long)gridDim.x * ET;
  const bool fixed = stride % CG == 0;  // see k_bn_apply
  float4 av[Q], bv[Q], kv[Q], scv[Q], shv[Q];
  auto params = [&](int c0) {
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      const int c = c0 + e;
      av[e] = ca[c], bv[e] = cb[c], kv[e] = cc[c];
      if (relu) {
        const float4 m = mean[c], r = rstd[c], ga = gamma[c], be = beta[c];
        scv[e] = make_float4(r.x * ga.x, r.y * ga.y, r.z * ga.z, r.w * ga.w);
        shv[e] = make_float4(be.x - m.x * scv[e].x, be.y - m.y * scv[e].y, be.z - m.z * scv[e].z, be.w - m.w * scv[e].w);
      }
    }
  };
  if (fixed) params((int)(((long)blockIdx.x * ET + threadIdx.x) % CG) * Q);
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < ng; i += stride) {
    if (!fixed) params((int)(i % CG) * Q);
    float4 gg[Q], v[Q], o[Q];
    IO::ld(dy, i, dy_half, gg);
    IO::ld(z, i, z_half, v);
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      if (relu) gg[e] = f4_gate(gg[e], f4_fma(v[e], scv[e], shv[e]));
      o[e] = f4_dz(gg[e], v[e], av[e], bv[e], kv[e]);
    }
    IO::st(dz, i, o, dz_half);
  }
}

// Pass 2 for units whose ReLU mask is the byte mask of the forward apply (units with a residual, or a shortcut unit
// fed by such a unit's output gradient): g = dy (.) mask is rebuilt from the byte instead of being parked by pass 1,
// so neither this unit nor the consumer of the shortcut gradient moves a masked copy of dy through memory.
template <int IOM>
__global__ __launch_bounds__(ET) void k_bn_bwd_apply_bytemask(const void* __restrict__ dy, int dy_half, const void* __restrict__ z, int z_half,
                                                              void* __restrict__ dz, int dz_half, long n4, int C4,
                                                              const float4* __restrict__ ca, const float4* __restrict__ cb,
                                                              const float4* __restrict__ cc,
                                                              const unsigned char* __restrict__ mask) {
  using IO = QIO<IOM>;
  constexpr int Q = IO::Q;
  const long ng = n4 / Q;
  const int CG = C4 / Q;
  const long stride = (long)gridDim.x * ET;
  const bool fixed = stride % CG == 0;  // see k_bn_apply
  float4 av[Q], bv[Q], kv[Q];
  auto params = [&](int c0) {
#pragma unroll
    for (int e = 0; e < Q; ++e) av[e] = ca[c0 + e], bv[e] = cb[c0 + e], kv[e] = cc[c0 + e];
  };
  if (fixed) params((int)(((long)blockIdx.x * ET + threadIdx.x) % CG) * Q);
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < ng; i += stride) {
    if (!fixed) params((int)(i % CG) * Q);
    float4 gg[Q], v[Q], o[Q];
    IO::ld(dy, i, dy_half, gg);
    const unsigned mk = IO::ldm(mask, i);
    IO::ld(z, i, z_half, v);
#pragma unroll
    for (int e = 0; e < Q; ++e) o[e] = f4_dz(f4_mask(gg[e], mk >> (8 * e)), v[e], av[e], bv[e], kv[e]);
    IO::st(dz, i, o, dz_half);
  }
}

// ---- Two BatchNorm units that receive the same gradient g = dy (.) mask (Bottleneck with a shortcut convolution:
// bn3 of the main branch and the shortcut's BatchNorm both feed the residual sum): one reduction and one apply pass
// read dy and the byte mask once for both.  C4 <= ET is required (channel quads per row fit a workgroup pass).
template <int IOM>
__global__ __launch_bounds__(ET) void k_bn_bwd_reduce_dual(const void* __restrict__ dy, int dy_half, const unsigned char* __restrict__ mask,
                                                           const void* __restrict__ za, const void* __restrict__ zb, int z_half, long M,
                                                           int C4, const float4* __restrict__ mean_a,
                                                           const float4* __restrict__ rstd_a, const float4* __restrict__ mean_b,
                                                           const float4* __restrict__ rstd_b, double* __restrict__ red_a,
                                                           double* __restrict__ red_b) {
  using IO = QIO<IOM>;
  constexpr int Q = IO::Q;
  constexpr int UNR = 2;
  __shared__ float4 ssum[Q][ET], sda[Q][ET], sdb[Q][ET];
  const int tid = threadIdx.x;
  const int CG = C4 / Q;
  const int rows_per_pass = ET / CG;
  const int my_row = tid / CG, cg = tid % CG;
  float4 s[Q], da[Q], db[Q];
#pragma unroll
  for (int e = 0; e < Q; ++e) s[e] = da[e] = db[e] = make_float4(0, 0, 0, 0);
  if (my_row < rows_per_pass) {
    float4 ma[Q], ra[Q], mb[Q], rb[Q];
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      ma[e] = mean_a[cg * Q + e];
      ra[e] = rstd_a[cg * Q + e];
      mb[e] = mean_b[cg * Q + e];
      rb[e] = rstd_b[cg * Q + e];
    }
    const long stride = (long)gridDim.x * rows_per_pass;
    for (long row0 = (long)blockIdx.x * rows_per_pass + my_row; row0 < M; row0 += UNR * stride) {
      float4 gq[UNR][Q], vaq[UNR][Q], vbq[UNR][Q];
      unsigned mq[UNR];
      bool okq[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const long row = row0 + u * stride;
        okq[u] = row < M;
        const long i = (okq[u] ? row : row0) * CG + cg;
        IO::ld(dy, i, dy_half, gq[u]);
        mq[u] = IO::ldm(mask, i);
        IO::ld(za, i, z_half, vaq[u]);
        IO::ld(zb, i, z_half, vbq[u]);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (!okq[u]) continue;
#pragma unroll
        for (int e = 0; e < Q; ++e) {
          const float4 g = f4_mask(gq[u][e], mq[u] >> (8 * e)), va = vaq[u][e], vb = vbq[u][e];
          s[e].x += g.x; s[e].y += g.y; s[e].z += g.z; s[e].w += g.w;
          da[e].x += g.x * (va.x - ma[e].x) * ra[e].x;
          da[e].y += g.y * (va.y - ma[e].y) * ra[e].y;
          da[e].z += g.z * (va.z - ma[e].z) * ra[e].z;
          da[e].w += g.w * (va.w - ma[e].w) * ra[e].w;
          db[e].x += g.x * (vb.x - mb[e].x) * rb[e].x;
          db[e].y += g.y * (vb.y - mb[e].y) * rb[e].y;
          db[e].z += g.z * (vb.z - mb[e].z) * rb[e].z;
          db[e].w += g.w * (vb.w - mb[e].w) * rb[e].w;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < Q; ++e) {
    ssum[e][tid] = s[e];
    sda[e][tid] = da[e];
    sdb[e][tid] = db[e];
  }
  __syncthreads();
  if (my_row == 0) {
    const int C = C4 * 4;
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      float4 ss = s[e], aa = da[e], bb = db[e];
      for (int rr = 1; rr < rows_per_pass; ++rr) {
        const float4 a = ssum[e][rr * CG + cg], b = sda[e][rr * CG + cg], c = sdb[e][rr * CG + cg];
        ss.x += a.x; ss.y += a.y; ss.z += a.z; ss.w += a.w;
        aa.x += b.x; aa.y += b.y; aa.z += b.z; aa.w += b.w;
        bb.x += c.x; bb.y += c.y; bb.z += c.z; bb.w += c.w;
      }
      const int cq = cg * Q + e;
      const float sv[4] = {ss.x, ss.y, ss.z, ss.w}, av[4] = {aa.x, aa.y, aa.z, aa.w}, bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(red_a + cq * 4 + k, (double)sv[k]);
        atomicAdd(red_b + cq * 4 + k, (double)sv[k]);
        atomicAdd(red_a + C + cq * 4 + k, (double)av[k]);
        atomicAdd(red_b + C + cq * 4 + k, (double)bv[k]);
      }
    }
  }
}

template <int IOM>
__global__ __launch_bounds__(ET) void k_bn_bwd_apply_dual(const void* __restrict__ dy, int dy_half, const unsigned char* __restrict__ mask,
                                                          const void* __restrict__ za, const void* __restrict__ zb, int z_half,
                                                          void* __restrict__ dza, void* __restrict__ dzb, int dz_half, long n4, int C4,
                                                          const float4* __restrict__ caa, const float4* __restrict__ cba,
                                                          const float4* __restrict__ cca, const float4* __restrict__ cab,
                                                          const float4* __restrict__ cbb, const float4* __restrict__ ccb) {
  using IO = QIO<IOM>;
  constexpr int Q = IO::Q;
  const long ng = n4 / Q;
  const int CG = C4 / Q;
  const long stride = (long)gridDim.x * ET;
  const bool fixed = stride % CG == 0;  // see k_bn_apply
  float4 aa[Q], ba[Q], ka[Q], ab[Q], bb[Q], kb[Q];
  auto params = [&](int c0) {
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      aa[e] = caa[c0 + e], ba[e] = cba[c0 + e], ka[e] = cca[c0 + e];
      ab[e] = cab[c0 + e], bb[e] = cbb[c0 + e], kb[e] = ccb[c0 + e];
    }
  };
  if (fixed) params((int)(((long)blockIdx.x * ET + threadIdx.x) % CG) * Q);
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < ng; i += stride) {
    if (!fixed) params((int)(i % CG) * Q);
    float4 gg[Q], va[Q], vb[Q], oa[Q], ob[Q];
    IO::ld(dy, i, dy_half, gg);
    const unsigned mk = IO::ldm(mask, i);
    IO::ld(za, i, z_half, va);
    IO::ld(zb, i, z_half, vb);
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      const float4 g = f4_mask(gg[e], mk >> (8 * e));
      oa[e] = f4_dz(g, va[e], aa[e], ba[e], ka[e]);
      ob[e] = f4_dz(g, vb[e], ab[e], bb[e], kb[e]);
    }
    IO::st(dza, i, oa, dz_half);
    IO::st(dzb, i, ob, dz_half);
  }
}

// MaxPool3d(kernel 3, stride 2, pad 1), channels-last
__global__ __launch_bounds__(ET) void k_maxpool3_fwd(const float4* __restrict__ x, float4* __restrict__ y, int B, int D,
                                                     int H, int W, int C4) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long total = (long)B * Do * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < total; i += (long)gridDim.x * ET) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % Wo);
    t /= Wo;
    const int oh = (int)(t % Ho);
    t /= Ho;
    const int od = (int)(t % Do);
    const int b = (int)(t / Do);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int a = -1; a <= 1; ++a) {
      const int z = 2 * od + a;
      if ((unsigned)z >= (unsigned)D) continue;
      for (int bb = -1; bb <= 1; ++bb) {
        const int yy = 2 * oh + bb;
        if ((unsigned)yy >= (unsigned)H) continue;
        for (int cc = -1; cc <= 1; ++cc) {
          const int xx = 2 * ow + cc;
          if ((unsigned)xx >= (unsigned)W) continue;
          const float4 v = x[((((long)b * D + z) * H + yy) * W + xx) * C4 + c];
          m.x = fmaxf(m.x, v.x);
          m.y = fmaxf(m.y, v.y);
          m.z = fmaxf(m.z, v.z);
          m.w = fmaxf(m.w, v.w);
        }
      }
    }
    y[i] = m;
  }
}

// gather form of the backward: dx[i] = sum over the <= 8 windows containing i of dy[o] [x[i] == y[o]]
__global__ __launch_bounds__(ET) void k_maxpool3_bwd(const float4* __restrict__ x, const float4* __restrict__ y,
                                                     const float4* __restrict__ dy, float4* __restrict__ dx, int B, int D,
                                                     int H, int W, int C4) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long total = (long)B * D * H * W * C4;
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < total; i += (long)gridDim.x * ET) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int w = (int)(t % W);
    t /= W;
    const int h = (int)(t % H);
    t /= H;
    const int d = (int)(t % D);
    const int b = (int)(t / D);
    const float4 v = x[i];
    float4 acc = make_float4(0, 0, 0, 0);
    // windows o with 2o-1 <= i <= 2o+1  ->  o in {floor(i/2), floor((i+1)/2)} (deduplicated)
    const int d0 = d / 2, d1 = (d + 1) / 2, h0 = h / 2, h1 = (h + 1) / 2, w0 = w / 2, w1 = (w + 1) / 2;
    for (int a = 0; a < 2; ++a) {
      const int od = a ? d1 : d0;
      if ((a && d1 == d0) || od >= Do) continue;
      for (int bb = 0; bb < 2; ++bb) {
        const int oh = bb ? h1 : h0;
        if ((bb && h1 == h0) || oh >= Ho) continue;
        for (int cc = 0; cc < 2; ++cc) {
          const int ow = cc ? w1 : w0;
          if ((cc && w1 == w0) || ow >= Wo) continue;
          const long o = ((((long)b * Do + od) * Ho + oh) * Wo + ow) * C4 + c;
          const float4 m = y[o], g = dy[o];
          acc.x += v.x == m.x ? g.x : 0.f;
          acc.y += v.y == m.y ? g.y : 0.f;
          acc.z += v.z == m.z ? g.z : 0.f;
          acc.w += v.w == m.w ? g.w : 0.f;
        }
      }
    }
    dx[i] = acc;
  }
}

// voxel index -> (b, d, h, w); shifts when all three extents are powers of two (runtime integer division costs ~40
// instructions per quotient and these kernels do three per element)
struct VoxDecode {
  int D, H, W, sd, sh, sw;  // s* = log2 or -1
};
__host__ inline VoxDecode make_decode(int D, int H, int W) {
  auto lg = [](int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
  };
  VoxDecode q{D, H, W, lg(D), lg(H), lg(W)};
  if (q.sd < 0 || q.sh < 0 || q.sw < 0) q.sd = q.sh = q.sw = -1;
  return q;
}
__device__ __forceinline__ void decode_vox(const VoxDecode& q, long v, int& b, int& d, int& h, int& w) {
  if (q.sw >= 0) {
    w = (int)(v & (q.W - 1));
    h = (int)((v >> q.sw) & (q.H - 1));
    d = (int)((v >> (q.sw + q.sh)) & (q.D - 1));
    b = (int)(v >> (q.sw + q.sh + q.sd));
  } else {
    long t = v;
    w = (int)(t % q.W);
    t /= q.W;
    h = (int)(t % q.H);
    t /= q.H;
    d = (int)(t % q.D);
    b = (int)(t / q.D);
  }
}

// ---- stem: BatchNorm + ReLU + MaxPool3d(3,2,1) without materialising the normalised 64-channel volume.
// y = relu(z * sc + sh) with sc = rstd * gamma, sh = beta - mean * sc  (recomputed wherever it is needed)
__device__ __forceinline__ float4 bn_relu4(float4 v, float4 sc, float4 sh) {
  return make_float4(fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f), fmaxf(fmaf(v.z, sc.z, sh.z), 0.f),
                     fmaxf(fmaf(v.w, sc.w, sh.w), 0.f));
}

__global__ __launch_bounds__(ET) void k_bn_relu_pool3_fwd(const float4* __restrict__ z, float4* __restrict__ p, int B, int D,
                                                          int H, int W, int C4, const float4* __restrict__ sc,
                                                          const float4* __restrict__ sh, VoxDecode vq, int c4_shift,
                                                          unsigned xcd_group) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long total = (long)B * Do * Ho * Wo * C4;
  // Workgroup b runs on XCD b % 8, and neighbouring output rows / planes share input rows / planes: with the plain order the
  // four 256-quad chunks of an output row land on four XCDs and every L2 fetches its own copy of the shared rows (2.25x the
  // tensor through the fabric).  xcd_group = 0xffffffff: each XCD walks a contiguous eighth of the pooled tensor (3.06 ->
  // 2.41 ms at 4 x 512 x 128 x 128 x 64); 0 < xcd_group: runs of xcd_group chunks per XCD in every round of gridDim.x chunks.
  auto pool_one = [&](long i, int c, int ow, int oh, int od, int b) {
    const float4 a = sc[c], s0 = sh[c];
    float4 m = make_float4(0.f, 0.f, 0.f, 0.f);  // ReLU output is >= 0 and every window holds a valid voxel
    // A window position outside the volume is CLAMPED onto the border voxel, which the window holds anyway (the maximum does
    // not change): 27 unconditional loads, requested together.  With `continue` on the bounds every load sat behind a
    // branch and was waited for on the spot (27 serial L2 round trips per output quad).
    const float4* const zb = z + (long)b * D * H * W * C4 + c;
#pragma unroll
    for (int dz = -1; dz <= 1; ++dz) {
      const long zo = (long)max(2 * od + dz, 0) * H;   // (2 od + 1 <= D - 1: D is even)
      float4 v[9];
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx)
          v[(dy + 1) * 3 + dx + 1] = zb[((zo + max(2 * oh + dy, 0)) * W + max(2 * ow + dx, 0)) * C4];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const float4 r = bn_relu4(v[t], a, s0);
        m.x = fmaxf(m.x, r.x);
        m.y = fmaxf(m.y, r.y);
        m.z = fmaxf(m.z, r.z);
        m.w = fmaxf(m.w, r.w);
      }
    }
    p[i] = m;
  };
  unsigned lb = blockIdx.x;
  long stride = (long)gridDim.x * ET;
  if (xcd_group == 0xffffffffu) {   // slab mode: XCD k walks the k-th eighth of the tensor, gridDim.x / 8 chunks at a time
    const unsigned slot = blockIdx.x >> 3, xcd = blockIdx.x & 7u, per = gridDim.x >> 3;
    const long rounds = total / ((long)gridDim.x * ET);   // (the launcher checked: whole rounds)
    lb = 0;
    stride = (long)per * ET;
    const long first = ((long)xcd * rounds * per + slot) * ET + threadIdx.x, last = (long)(xcd + 1) * rounds * per * ET;
    for (long i = first; i < last; i += stride) {
      int c, ow, oh, od, b;
      c = (int)(i & (C4 - 1));
      decode_vox(vq, i >> c4_shift, b, od, oh, ow);
      pool_one(i, c, ow, oh, od, b);
    }
    return;
  }
  if (xcd_group > 0) {
    const unsigned slot = blockIdx.x >> 3, xcd = blockIdx.x & 7u;
    lb = (slot / xcd_group) * (8u * xcd_group) + xcd * xcd_group + slot % xcd_group;
  }
  for (long i = (long)lb * ET + threadIdx.x; i < total; i += stride) {
    int c, ow, oh, od, b;
    if (vq.sw >= 0 && c4_shift >= 0) {  // pooled extents and C4 are powers of two: shifts instead of five divisions
      c = (int)(i & (C4 - 1));
      decode_vox(vq, i >> c4_shift, b, od, oh, ow);
    } else {
      c = (int)(i % C4);
      long t = i / C4;
      ow = (int)(t % Wo);
      t /= Wo;
      oh = (int)(t % Ho);
      t /= Ho;
      od = (int)(t % Do);
      b = (int)(t / Do);
    }
    pool_one(i, c, ow, oh, od, b);
  }
}

// g[i] = [y_i > 0] * sum over the <= 8 windows o containing i of dp[o] * [y_i == p[o]]   (y recomputed from z)
__device__ __forceinline__ float4 stem_pool_grad(const float4* __restrict__ p, const float4* __restrict__ dp, float4 y,
                                                 int b, int d, int h, int w, int c, int D, int H, int W, int C4) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  float4 acc = make_float4(0, 0, 0, 0);
  const int d0 = d / 2, d1 = (d + 1) / 2, h0 = h / 2, h1 = (h + 1) / 2, w0 = w / 2, w1 = (w + 1) / 2;
  for (int a = 0; a < 2; ++a) {
    const int od = a ? d1 : d0;
    if ((a && d1 == d0) || od >= Do) continue;
    for (int bb = 0; bb < 2; ++bb) {
      const int oh = bb ? h1 : h0;
      if ((bb && h1 == h0) || oh >= Ho) continue;
      for (int cc = 0; cc < 2; ++cc) {
        const int ow = cc ? w1 : w0;
        if ((cc && w1 == w0) || ow >= Wo) continue;
        const long o = ((((long)b * Do + od) * Ho + oh) * Wo + ow) * C4 + c;
        const float4 m = p[o], gg = dp[o];
        acc.x += (y.x == m.x) ? gg.x : 0.f;
        acc.y += (y.y == m.y) ? gg.y : 0.f;
        acc.z += (y.z == m.z) ? gg.z : 0.f;
        acc.w += (y.w == m.w) ? gg.w : 0.f;
      }
    }
  }
  acc.x = y.x > 0.f ? acc.x : 0.f;
  acc.y = y.y > 0.f ? acc.y : 0.f;
  acc.z = y.z > 0.f ? acc.z : 0.f;
  acc.w = y.w > 0.f ? acc.w : 0.f;
  return acc;
}

// pass 1: per channel sum(g), sum(g * zhat) -> red[0:C], red[C:2C].  Requires 4*C4 <= ET... thread t owns channel
// quad t % C4 and strides over voxels, so per-thread sums stay per channel (C4 = 16 for the 64-channel stem).
__global__ __launch_bounds__(ET) void k_stem_bwd_reduce(const float4* __restrict__ z, const float4* __restrict__ p,
                                                        const float4* __restrict__ dp, int B, int D, int H, int W, int C4,
                                                        const float4* __restrict__ sc, const float4* __restrict__ sh,
                                                        const float4* __restrict__ mean, const float4* __restrict__ rstd,
                                                        double* __restrict__ red, VoxDecode vq) {
  __shared__ float4 ssum[ET], sdot[ET];
  const int tid = threadIdx.x;
  const int cq = tid % C4, vrow = tid / C4, vpb = ET / C4;
  const long nvox = (long)B * D * H * W;
  const float4 a = sc[cq], s0 = sh[cq], mu = mean[cq], rs = rstd[cq];
  float4 s = make_float4(0, 0, 0, 0), dd = make_float4(0, 0, 0, 0);
  for (long v = (long)blockIdx.x * vpb + vrow; v < nvox; v += (long)gridDim.x * vpb) {
    int b, d, h, w;
    decode_vox(vq, v, b, d, h, w);
    const float4 zv = z[v * C4 + cq];
    const float4 g = stem_pool_grad(p, dp, bn_relu4(zv, a, s0), b, d, h, w, cq, D, H, W, C4);
    s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
    dd.x += g.x * (zv.x - mu.x) * rs.x;
    dd.y += g.y * (zv.y - mu.y) * rs.y;
    dd.z += g.z * (zv.z - mu.z) * rs.z;
    dd.w += g.w * (zv.w - mu.w) * rs.w;
  }
  ssum[tid] = s;
  sdot[tid] = dd;
  __syncthreads();
  if (vrow == 0) {
    for (int rr = 1; rr < vpb; ++rr) {
      const float4 x1 = ssum[rr * C4 + cq], x2 = sdot[rr * C4 + cq];
      s.x += x1.x; s.y += x1.y; s.z += x1.z; s.w += x1.w;
      dd.x += x2.x; dd.y += x2.y; dd.z += x2.z; dd.w += x2.w;
    }
    const int C = C4 * 4;
    atomicAdd(red + cq * 4 + 0, (double)s.x);
    atomicAdd(red + cq * 4 + 1, (double)s.y);
    atomicAdd(red + cq * 4 + 2, (double)s.z);
    atomicAdd(red + cq * 4 + 3, (double)s.w);
    atomicAdd(red + C + cq * 4 + 0, (double)dd.x);
    atomicAdd(red + C + cq * 4 + 1, (double)dd.y);
    atomicAdd(red + C + cq * 4 + 2, (double)dd.z);
    atomicAdd(red + C + cq * 4 + 3, (double)dd.w);
  }
}

// pass 2: dz = ca * g + cb * z + cc with g recomputed
__global__ __launch_bounds__(ET) void k_stem_bwd_apply(const float4* __restrict__ z, const float4* __restrict__ p,
                                                       const float4* __restrict__ dp, float4* __restrict__ dz, int B, int D,
                                                       int H, int W, int C4, const float4* __restrict__ sc,
                                                       const float4* __restrict__ sh, const float4* __restrict__ ca,
                                                       const float4* __restrict__ cb, const float4* __restrict__ cc,
                                                       VoxDecode vq, int c4_shift) {
  const long total = (long)B * D * H * W * C4;
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < total; i += (long)gridDim.x * ET) {
    const int c = c4_shift >= 0 ? (int)(i & (C4 - 1)) : (int)(i % C4);
    int b, d, h, w;
    decode_vox(vq, c4_shift >= 0 ? i >> c4_shift : i / C4, b, d, h, w);
    const float4 zv = z[i];
    const float4 g = stem_pool_grad(p, dp, bn_relu4(zv, sc[c], sh[c]), b, d, h, w, c, D, H, W, C4);
    const float4 a = ca[c], bq = cb[c], k = cc[c];
    dz[i] = make_float4(fmaf(a.x, g.x, fmaf(bq.x, zv.x, k.x)), fmaf(a.y, g.y, fmaf(bq.y, zv.y, k.y)),
                        fmaf(a.z, g.z, fmaf(bq.z, zv.z, k.z)), fmaf(a.w, g.w, fmaf(bq.w, zv.w, k.w)));
  }
}

// Tiled form of both passes for the 64-channel stem.  One workgroup owns 2 (d) x 2 (h) x 32 (w) voxels: the
// <= 2 x 2 x 17 pooled windows they can belong to are staged ONCE in LDS (values and gradients, 34 KB) instead of
// being fetched through L2 by every voxel (6.8 window reads per voxel on average there: that cache traffic,
// not HBM, bounded the untiled kernels).  Iteration `it` of a thread handles a fixed (d parity, h parity, w
// parity), so the number of windows per voxel (1 or 2 per axis) is a compile-time constant and a wave never
// diverges; windows beyond the pooled extent are staged with a zero gradient.
constexpr int ST_W = 32, ST_HALO = ST_W / 2 + 1;

template <bool APPLY>
__global__ __launch_bounds__(ET) void k_stem_bwd_tiled(const float4* __restrict__ z, const float4* __restrict__ p,
                                                       const float4* __restrict__ dp, float4* __restrict__ dz, int B, int D,
                                                       int H, int W, const float4* __restrict__ sc, const float4* __restrict__ sh,
                                                       const float4* __restrict__ k0, const float4* __restrict__ k1,
                                                       const float4* __restrict__ k2, double* __restrict__ red, long ntiles,
                                                       int xcd_slab) {
  constexpr int C4 = 16;
  __shared__ float4 sp[2 * 2 * ST_HALO * C4], sg[2 * 2 * ST_HALO * C4];
  const int tid = threadIdx.x, cq = tid & (C4 - 1), vrow = tid >> 4;  // 16 voxel slots x 16 channel quads
  const int Do = D / 2, Ho = H / 2, Wo = W / 2, tw = (W + ST_W - 1) / ST_W;
  const float4 a = sc[cq], s0 = sh[cq];
  // reduce: k0 = mean, k1 = rstd;  apply: k0 = ca, k1 = cb, k2 = cc
  const float4 q0 = k0[cq], q1 = k1[cq], q2 = APPLY ? k2[cq] : make_float4(0, 0, 0, 0);
  float4 s = make_float4(0, 0, 0, 0), dd = make_float4(0, 0, 0, 0);
  // XCD-aware walk (workgroup b runs on XCD b % 8): neighbouring tiles share their pooled windows, so each XCD takes a
  // contiguous eighth of the tile list (whole eighths only) and its L2 serves the shared windows
  const bool slab = xcd_slab && (ntiles & 7) == 0 && (gridDim.x & 7u) == 0;
  const long t_first = slab ? (long)(blockIdx.x & 7u) * (ntiles >> 3) + (blockIdx.x >> 3) : (long)blockIdx.x;
  const long t_last = slab ? (long)((blockIdx.x & 7u) + 1) * (ntiles >> 3) : ntiles;
  const long t_step = slab ? (long)(gridDim.x >> 3) : (long)gridDim.x;
  for (long tile = t_first; tile < t_last; tile += t_step) {
    long t = tile;
    const int wt = (int)(t % tw);
    t /= tw;
    const int hb = (int)(t % Ho);
    t /= Ho;
    const int db = (int)(t % Do);
    const int b = (int)(t / Do);
    const int w0 = wt * ST_W, ow0 = w0 / 2;
    // this thread's 8 voxels: rows (d, h) = (2 db + (it >> 2), 2 hb + ((it >> 1) & 1)), w = w0 + 2 vrow + (it & 1)
    float4 zv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int d = 2 * db + (it >> 2), h = 2 * hb + ((it >> 1) & 1), w = w0 + 2 * vrow + (it & 1);
      zv[it] = w < W ? z[((((long)b * D + d) * H + h) * W + w) * C4 + cq] : make_float4(0, 0, 0, 0);
    }
    __syncthreads();  // the previous tile's window reads are finished
    for (int idx = tid; idx < 2 * 2 * ST_HALO * C4; idx += ET) {
      const int c = idx & (C4 - 1), e = idx >> 4;
      const int owl = e % ST_HALO, r = e / ST_HALO;
      const int od = db + (r >> 1), oh = hb + (r & 1), ow = ow0 + owl;
      const bool ok = od < Do && oh < Ho && ow < Wo;
      const long o = ((((long)b * Do + od) * Ho + oh) * Wo + ow) * C4 + c;
      sp[idx] = ok ? p[o] : make_float4(0, 0, 0, 0);
      sg[idx] = ok ? dp[o] : make_float4(0, 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int pd = it >> 2, ph = (it >> 1) & 1, pw = it & 1;  // parities = extra windows per axis
      const int d = 2 * db + pd, h = 2 * hb + ph, w = w0 + 2 * vrow + pw;
      if (w >= W) continue;
      const float4 y = bn_relu4(zv[it], a, s0);
      float4 g = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i <= pd; ++i)
#pragma unroll
        for (int j = 0; j <= ph; ++j)
#pragma unroll
          for (int k = 0; k <= pw; ++k) {
            const int e = ((i * 2 + j) * ST_HALO + vrow + k) * C4 + cq;
            const float4 m = sp[e], gg = sg[e];
            g.x += (y.x == m.x) ? gg.x : 0.f;
            g.y += (y.y == m.y) ? gg.y : 0.f;
            g.z += (y.z == m.z) ? gg.z : 0.f;
            g.w += (y.w == m.w) ? gg.w : 0.f;
          }
      g.x = y.x > 0.f ? g.x : 0.f;
      g.y = y.y > 0.f ? g.y : 0.f;
      g.z = y.z > 0.f ? g.z : 0.f;
      g.w = y.w > 0.f ? g.w : 0.f;
      const float4 v = zv[it];
      if constexpr (APPLY) {
        st_stream(dz + ((((long)b * D + d) * H + h) * W + w) * C4 + cq,
                  make_float4(fmaf(q0.x, g.x, fmaf(q1.x, v.x, q2.x)), fmaf(q0.y, g.y, fmaf(q1.y, v.y, q2.y)),
                              fmaf(q0.z, g.z, fmaf(q1.z, v.z, q2.z)), fmaf(q0.w, g.w, fmaf(q1.w, v.w, q2.w))));
      } else {
        s.x += g.x; s.y += g.y; s.z += g.z; s.w += g.w;
        dd.x += g.x * (v.x - q0.x) * q1.x;
        dd.y += g.y * (v.y - q0.y) * q1.y;
        dd.z += g.z * (v.z - q0.z) * q1.z;
        dd.w += g.w * (v.w - q0.w) * q1.w;
      }
    }
  }
  if constexpr (!APPLY) {
    __syncthreads();
    float4* ssum = sp;
    float4* sdot = sg;
    ssum[tid] = s;
    sdot[tid] = dd;
    __syncthreads();
    if (vrow == 0) {
      for (int rr = 1; rr < ET / C4; ++rr) {
        const float4 x1 = ssum[rr * C4 + cq], x2 = sdot[rr * C4 + cq];
        s.x += x1.x; s.y += x1.y; s.z += x1.z; s.w += x1.w;
        dd.x += x2.x; dd.y += x2.y; dd.z += x2.z; dd.w += x2.w;
      }
      const int C = C4 * 4;
      atomicAdd(red + cq * 4 + 0, (double)s.x);
      atomicAdd(red + cq * 4 + 1, (double)s.y);
      atomicAdd(red + cq * 4 + 2, (double)s.z);
      atomicAdd(red + cq * 4 + 3, (double)s.w);
      atomicAdd(red + C + cq * 4 + 0, (double)dd.x);
      atomicAdd(red + C + cq * 4 + 1, (double)dd.y);
      atomicAdd(red + C + cq * 4 + 2, (double)dd.z);
      atomicAdd(red + C + cq * 4 + 3, (double)dd.w);
    }
  }
}

__global__ void k_bn_scale_shift(const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, int C, float* __restrict__ sc, float* __restrict__ sh) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float a = rstd[c] * gamma[c];
  sc[c] = a;
  sh[c] = beta[c] - mean[c] * a;
}

// [B][V][C] <-> [B][C][V] through a 32x32 LDS tile
__global__ void k_transpose_vc(const float* __restrict__ in, float* __restrict__ out, long V, int C, int to_ncv) {
  __shared__ float t[32][33];
  const int b = blockIdx.z;
  const long v0 = (long)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in 0..7
  const float* ib = in + (long)b * V * C;
  float* ob = out + (long)b * V * C;
  if (to_ncv) {  // in [V][C] -> out [C][V]
    for (int r = ty; r < 32; r += 8) {
      const long v = v0 + r;
      const int c = c0 + tx;
      t[r][tx] = (v < V && c < C) ? ib[v * C + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + r;
      const long v = v0 + tx;
      if (v < V && c < C) ob[(long)c * V + v] = t[tx][r];
    }
  } else {  // in [C][V] -> out [V][C]
    for (int r = ty; r < 32; r += 8) {
      const int c = c0 + r;
      const long v = v0 + tx;
      t[r][tx] = (v < V && c < C) ? ib[(long)c * V + v] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const long v = v0 + r;
      const int c = c0 + tx;
      if (v < V && c < C) ob[v * C + c] = t[tx][r];
    }
  }
}

__global__ __launch_bounds__(ET) void k_cast(const void* __restrict__ x, int x_half, void* __restrict__ y, int y_half, long n4) {
  for (long i = (long)blockIdx.x * ET + threadIdx.x; i < n4; i += (long)gridDim.x * ET) hp_st4(y, 4 * i, hp_ld4(x, 4 * i, x_half), y_half);
}

// Grid of the streaming BatchNorm passes (grid-stride loops over channel quads): THREE workgroups per CU.  Round 4 sweep at the
// headline shape (per step, un-overlapped): forward apply 18.7 ms with 2048 workgroups (8 per CU: rounds 1-3), 17.5 with 1024,
// 16.9 with 768, 17.6 with 512, 17.8 / 17.6 with 640 / 896; backward apply 23.7 / 22.6 / 21.7 / 21.4 with 2048 / 1024 / 768 / 512
// -- fewer concurrent streams, and every CU with the same number of them.  (The planar U-Net / GroupNorm / pooling passes do
// not care: 2048 stays there.)  HP_BN_GRID / HP_BN_BWD_GRID: A/B hooks.
static unsigned grid_for(long n, int per_block = ET) {
  static const long cap = getenv("HP_BN_GRID") ? atol(getenv("HP_BN_GRID")) : 256 * 3;
  return (unsigned)std::min<long>((n + per_block - 1) / per_block, cap);
}
static unsigned grid_for_bwd(long n, int per_block = ET) {   // the backward apply passes
  static const long cap = getenv("HP_BN_BWD_GRID") ? atol(getenv("HP_BN_BWD_GRID")) : 256 * 3;
  return (unsigned)std::min<long>((n + per_block - 1) / per_block, cap);
}

// IOM of QIO for a call: 0 = every tensor fp32, 1 = every tensor bf16 (and an even number of channel quads), 2 = mixed
static int io_mode(int C4, std::initializer_list<int> halves) {
  int n = 0, h = 0;
  for (int v : halves) {
    ++n;
    h += v ? 1 : 0;
  }
  if (h == 0) return 0;
  return (h == n && C4 % 2 == 0) ? 1 : 2;
}
// launch KERN<IOM> for the run-time io mode `iom`
#define HP_LAUNCH_IOM(KERN, iom, grid, st, ...)                                                      \
  do {                                                                                              \
    if ((iom) == 0) hipLaunchKernelGGL(KERN<0>, dim3(grid), dim3(ET), 0, st, __VA_ARGS__);          \
    else if ((iom) == 1) hipLaunchKernelGGL(KERN<1>, dim3(grid), dim3(ET), 0, st, __VA_ARGS__);     \
    else hipLaunchKernelGGL(KERN<2>, dim3(grid), dim3(ET), 0, st, __VA_ARGS__);                     \
  } while (0)

}  // namespace hp

using namespace hp;

extern "C" int hp_bn_train_finalize(const double* stats, long M, int C, float eps, float momentum, float* mean,
                                    float* rstd, float* running_mean, float* running_var, void* stream) {
  return hp_bn_train_finalize_counted(stats, M, C, eps, momentum, mean, rstd, running_mean, running_var, nullptr, stream);
}

extern "C" int hp_bn_train_finalize_counted(const double* stats, long M, int C, float eps, float momentum, float* mean,
                                            float* rstd, float* running_mean, float* running_var,
                                            long long* num_batches_tracked, void* stream) {
  HP_REQUIRE(stats && mean && rstd && M > 0 && C > 0, "hp_bn_train_finalize: bad argument");
  hipLaunchKernelGGL(k_bn_finalize, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, stats, M, C, eps, momentum,
                     mean, rstd, running_mean, running_var, num_batches_tracked);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_bn_eval_stats(const float* running_mean, const float* running_var, int C, float eps, float* mean,
                                float* rstd, void* stream) {
  HP_REQUIRE(running_mean && running_var && mean && rstd && C > 0, "hp_bn_eval_stats: bad argument");
  hipLaunchKernelGGL(k_bn_eval_stats, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, running_mean, running_var,
                     C, eps, mean, rstd);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_bn_apply(const void* z, const void* res, void* y, long M, int C, const float* mean, const float* rstd,
                           const float* gamma, const float* beta, int relu, unsigned char* relu_mask, int io, void* stream) {
  return hp_bn_apply_res_bn(z, res, y, M, C, mean, rstd, gamma, beta, relu, relu_mask, nullptr, nullptr, nullptr, nullptr, io, stream);
}

extern "C" int hp_bn_apply_res_bn(const void* z, const void* res, void* y, long M, int C, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, int relu, unsigned char* relu_mask,
                                  const float* res_mean, const float* res_rstd, const float* res_gamma, const float* res_beta,
                                  int io, void* stream) {
  HP_REQUIRE(z && y && mean && rstd && gamma && beta && M > 0 && C > 0 && C % 4 == 0, "hp_bn_apply: bad argument");
  HP_REQUIRE(!res_mean || (res && res_rstd && res_gamma && res_beta), "hp_bn_apply_res_bn: incomplete residual BatchNorm");
  const long n4 = M * (C / 4);
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("bn_apply", st);
  // io & HP_BN_ACT_BF16: y (and a plain residual, which is an activation too) are bf16; a residual that comes with its own
  // BatchNorm is the shortcut convolution's raw fp32 output
  const int act_half = (io & HP_BN_ACT_BF16) ? 1 : 0, z_half = (io & HP_BN_Z_BF16) ? 1 : 0;
  const int res_half = res_mean ? z_half : act_half;
  const int iom = res ? io_mode(C / 4, {z_half, act_half, res_half}) : io_mode(C / 4, {z_half, act_half});
  HP_LAUNCH_IOM(k_bn_apply, iom, grid_for(n4 / (iom == 1 ? 2 : 1)), st, z, z_half, res, res_half, y, act_half, n4,
                C / 4, (const float4*)mean, (const float4*)rstd, (const float4*)gamma, (const float4*)beta, relu, relu_mask,
                (const float4*)res_mean, (const float4*)res_rstd, (const float4*)res_gamma, (const float4*)res_beta);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// workspace: 2*C doubles (reduction) + 3*C floats (coefficients)
extern "C" size_t hp_bn_backward_workspace_bytes(int C) { return sizeof(double) * 2 * C + sizeof(float) * 3 * C; }

extern "C" int hp_bn_backward(const void* dy, const float* y, const void* z, void* g_out, void* dz, long M, int C,
                              const float* mean, const float* rstd, const float* gamma, const float* beta_for_mask,
                              int relu, int train, float* dgamma, float* dbeta, const unsigned char* relu_mask,
                              void* workspace, int io, void* stream) {
  const int dy_half = (io & HP_BN_DY_BF16) ? 1 : 0, dz_half = (io & HP_BN_DZ_BF16) ? 1 : 0, z_half = (io & HP_BN_Z_BF16) ? 1 : 0;
  HP_REQUIRE(dy && z && dz && mean && rstd && gamma && workspace && M > 0 && C > 0 && C % 4 == 0,
             "hp_bn_backward: bad argument");
  HP_REQUIRE(!relu || y || relu_mask || beta_for_mask,
             "hp_bn_backward: relu needs the forward output, its mask, or beta to rebuild the mask");
  // y may be NULL when there was no residual: the ReLU mask is then recomputed from z (bit-identical affine map)
  hipStream_t st = (hipStream_t)stream;
  double* red = (double*)workspace;
  float* ca = (float*)(red + 2 * C);
  float* cb = ca + C;
  float* cc = cb + C;
  HP_CHECK_HIP(hipMemsetAsync(red, 0, sizeof(double) * 2 * C, st));
  const int C4 = C / 4;
  // g buffer: caller-provided g_out (residual units: g is also the gradient of the shortcut), or none at all when
  // pass 2 can rebuild the mask itself (no forward output given: mask from z, or no ReLU); dy must not alias dz then
  const bool remask = !g_out && !y && !relu_mask && dy != dz;
  // byte mask given and nobody wants g: pass 2 re-applies the byte mask to dy (nothing parked)
  const bool bytemask = relu && relu_mask && !g_out && !y;
  void* gbuf = (remask || bytemask) ? nullptr : g_out ? g_out : dz;
  {
    HP_PROF("bn_bwd_reduce", st);
    // the forward output y, where a caller still passes it, is fp32: such calls take the mixed path
    const int iom = y ? (dy_half || z_half || (gbuf && dz_half) ? 2 : 0)
                      : gbuf ? io_mode(C4, {dy_half, z_half, dz_half}) : io_mode(C4, {dy_half, z_half});
    const int CG = C4 / (iom == 1 ? 2 : 1);
    const int rows_per_pass = CG < ET ? ET / CG : 1;
    // at most one workgroup per CU, and at least HP_BN_RED_TRIPS (default 16) trips of 4 rows per thread: a workgroup's fixed
    // cost (parameter loads, the cross-row reduction, 2C fp64 atomics) is paid per workgroup, and on the small tensors of the
    // deep layers a grid sized by rows alone spends most of its time there
    static const int trips_min = getenv("HP_BN_RED_TRIPS") ? atoi(getenv("HP_BN_RED_TRIPS")) : 16;
    // (one workgroup per CU since round 4: 8.66 ms/step at the headline shape against 9.2 with two, 9.1 with three, 9.7 with four)
    static const long red_wgs = getenv("HP_BN_RED_WGS") ? atol(getenv("HP_BN_RED_WGS")) : 256;
    const unsigned nb = (unsigned)std::max<long>(1, std::min<long>(M / ((long)rows_per_pass * 4 * trips_min), red_wgs));
    if (iom == 0)
      hipLaunchKernelGGL((k_bn_bwd_reduce<4, 0>), dim3(nb), dim3(ET), 0, st, dy, dy_half, (const float4*)y, z, z_half, gbuf, dz_half, M,
                         C4, (const float4*)mean, (const float4*)rstd, relu, red, (const float4*)gamma, (const float4*)beta_for_mask,
                         relu_mask);
    else if (iom == 1)
      hipLaunchKernelGGL((k_bn_bwd_reduce<HP_BN_UNR_H, 1>), dim3(nb), dim3(ET), 0, st, dy, dy_half, (const float4*)y, z, z_half, gbuf, dz_half, M,
                         C4, (const float4*)mean, (const float4*)rstd, relu, red, (const float4*)gamma, (const float4*)beta_for_mask,
                         relu_mask);
    else
      hipLaunchKernelGGL((k_bn_bwd_reduce<4, 2>), dim3(nb), dim3(ET), 0, st, dy, dy_half, (const float4*)y, z, z_half, gbuf, dz_half, M,
                         C4, (const float4*)mean, (const float4*)rstd, relu, red, (const float4*)gamma, (const float4*)beta_for_mask,
                         relu_mask);
  }
  hipLaunchKernelGGL(k_bn_bwd_coef, dim3((C + 127) / 128), dim3(128), 0, st, red, M, C, mean, rstd, gamma, train, dgamma,
                     dbeta, ca, cb, cc);
  {
    HP_PROF("bn_bwd_apply", st);
    const long n4 = M * C4;
    if (bytemask) {
      const int iom = io_mode(C4, {dy_half, z_half, dz_half});
      HP_LAUNCH_IOM(k_bn_bwd_apply_bytemask, iom, grid_for_bwd(n4 / (iom == 1 ? 2 : 1)), st, dy, dy_half, z, z_half, dz, dz_half, n4, C4,
                    (const float4*)ca, (const float4*)cb, (const float4*)cc, relu_mask);
    } else if (remask) {
      const int iom = io_mode(C4, {dy_half, z_half, dz_half});
      HP_LAUNCH_IOM(k_bn_bwd_apply_mask, iom, grid_for_bwd(n4 / (iom == 1 ? 2 : 1)), st, dy, dy_half, z, z_half, dz, dz_half, n4, C4,
                    (const float4*)ca, (const float4*)cb, (const float4*)cc, (const float4*)mean, (const float4*)rstd,
                    (const float4*)gamma, (const float4*)beta_for_mask, relu);
    } else {
      const int iom = io_mode(C4, {z_half, dz_half});
      HP_LAUNCH_IOM(k_bn_bwd_apply, iom, grid_for_bwd(n4 / (iom == 1 ? 2 : 1)), st, (const void*)gbuf, dz_half, z, z_half, dz, dz_half, n4,
                    C4, (const float4*)ca, (const float4*)cb, (const float4*)cc);
    }
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

__global__ void k_bn_sum_slots(const double* __restrict__ slots, double* __restrict__ red, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int sl = 0; sl < HP_STATS_SLOTS; ++sl) s += slots[(size_t)sl * n + i];
  red[i] = s;
}

extern "C" int hp_bn_backward_presummed(const void* dy, const void* z, void* dz, long M, int C, const float* mean,
                                        const float* rstd, const float* gamma, const float* beta_for_mask, int relu, int train,
                                        float* dgamma, float* dbeta, const unsigned char* relu_mask, const double* sums,
                                        void* workspace, int io, void* stream) {
  const int dy_half = (io & HP_BN_DY_BF16) ? 1 : 0, dz_half = (io & HP_BN_DZ_BF16) ? 1 : 0, z_half = (io & HP_BN_Z_BF16) ? 1 : 0;
  HP_REQUIRE(dy && z && dz && mean && rstd && gamma && sums && workspace && M > 0 && C > 0 && C % 4 == 0 && dy != dz,
             "hp_bn_backward_presummed: bad argument");
  HP_REQUIRE(!relu || relu_mask || beta_for_mask, "hp_bn_backward_presummed: relu needs the byte mask, or beta to rebuild the mask from z");
  hipStream_t st = (hipStream_t)stream;
  double* red = (double*)workspace;
  float* ca = (float*)(red + 2 * C);
  float* cb = ca + C;
  float* cc = cb + C;
  const int C4 = C / 4;
  hipLaunchKernelGGL(k_bn_sum_slots, dim3((2 * C + 127) / 128), dim3(128), 0, st, sums, red, 2 * C);
  hipLaunchKernelGGL(k_bn_bwd_coef, dim3((C + 127) / 128), dim3(128), 0, st, red, M, C, mean, rstd, gamma, train, dgamma,
                     dbeta, ca, cb, cc);
  {
    HP_PROF("bn_bwd_apply", st);
    const long n4 = M * C4;
    const int iom = io_mode(C4, {dy_half, z_half, dz_half});
    if (relu && relu_mask)
      HP_LAUNCH_IOM(k_bn_bwd_apply_bytemask, iom, grid_for_bwd(n4 / (iom == 1 ? 2 : 1)), st, dy, dy_half, z, z_half, dz, dz_half, n4, C4,
                    (const float4*)ca, (const float4*)cb, (const float4*)cc, relu_mask);
    else
      HP_LAUNCH_IOM(k_bn_bwd_apply_mask, iom, grid_for_bwd(n4 / (iom == 1 ? 2 : 1)), st, dy, dy_half, z, z_half, dz, dz_half, n4, C4,
                    (const float4*)ca, (const float4*)cb, (const float4*)cc, (const float4*)mean, (const float4*)rstd,
                    (const float4*)gamma, (const float4*)beta_for_mask, relu);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_bn_backward_dual(const void* dy, const unsigned char* relu_mask, long M, int C, const void* z_a,
                                   void* dz_a, const float* mean_a, const float* rstd_a, const float* gamma_a, int train_a,
                                   float* dgamma_a, float* dbeta_a, const void* z_b, void* dz_b, const float* mean_b,
                                   const float* rstd_b, const float* gamma_b, int train_b, float* dgamma_b, float* dbeta_b,
                                   void* workspace, int io, void* stream) {
  const int dy_half = (io & HP_BN_DY_BF16) ? 1 : 0, dz_half = (io & HP_BN_DZ_BF16) ? 1 : 0, z_half = (io & HP_BN_Z_BF16) ? 1 : 0;
  HP_REQUIRE(dy && relu_mask && z_a && dz_a && mean_a && rstd_a && gamma_a && z_b && dz_b && mean_b && rstd_b && gamma_b &&
                 workspace && M > 0 && C > 0 && C % 4 == 0 && C / 4 <= ET,
             "hp_bn_backward_dual: bad argument (C must be a multiple of 4, at most %d)", 4 * ET);
  hipStream_t st = (hipStream_t)stream;
  // workspace = two single-unit workspaces back to back
  const size_t half = (hp_bn_backward_workspace_bytes(C) + 15) / 16 * 16;
  double* red_a = (double*)workspace;
  double* red_b = (double*)((char*)workspace + half);
  float *ca_a = (float*)(red_a + 2 * C), *cb_a = ca_a + C, *cc_a = cb_a + C;
  float *ca_b = (float*)(red_b + 2 * C), *cb_b = ca_b + C, *cc_b = cb_b + C;
  HP_CHECK_HIP(hipMemsetAsync(workspace, 0, 2 * half, st));
  const int C4 = C / 4;
  {
    HP_PROF("bn_bwd_reduce", st);
    const int iom = io_mode(C4, {dy_half, z_half});
    const int rows_per_pass = ET / (C4 / (iom == 1 ? 2 : 1));
    const unsigned nb = (unsigned)std::min<long>((M + rows_per_pass - 1) / rows_per_pass, 256 * 2);
    HP_LAUNCH_IOM(k_bn_bwd_reduce_dual, iom, nb, st, dy, dy_half, relu_mask, z_a, z_b, z_half, M, C4, (const float4*)mean_a,
                  (const float4*)rstd_a, (const float4*)mean_b, (const float4*)rstd_b, red_a, red_b);
  }
  hipLaunchKernelGGL(k_bn_bwd_coef, dim3((C + 127) / 128), dim3(128), 0, st, red_a, M, C, mean_a, rstd_a, gamma_a, train_a,
                     dgamma_a, dbeta_a, ca_a, cb_a, cc_a);
  hipLaunchKernelGGL(k_bn_bwd_coef, dim3((C + 127) / 128), dim3(128), 0, st, red_b, M, C, mean_b, rstd_b, gamma_b, train_b,
                     dgamma_b, dbeta_b, ca_b, cb_b, cc_b);
  {
    HP_PROF("bn_bwd_apply", st);
    const long n4 = M * C4;
    const int iom = io_mode(C4, {dy_half, z_half, dz_half});
    HP_LAUNCH_IOM(k_bn_bwd_apply_dual, iom, grid_for_bwd(n4 / (iom == 1 ? 2 : 1)), st, dy, dy_half, relu_mask, z_a, z_b, z_half, dz_a, dz_b,
                  dz_half, n4, C4, (const float4*)ca_a, (const float4*)cb_a, (const float4*)cc_a, (const float4*)ca_b,
                  (const float4*)cb_b, (const float4*)cc_b);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_maxpool3d_k3s2_forward(const float* x, float* y, int B, int D, int H, int W, int C, void* stream) {
  HP_REQUIRE(x && y && C % 4 == 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "hp_maxpool3d_k3s2_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("maxpool3_fwd", st);
  const long n = (long)B * (D / 2) * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(k_maxpool3_fwd, dim3(grid_for(n)), dim3(ET), 0, st, (const float4*)x, (float4*)y, B, D, H, W, C / 4);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_maxpool3d_k3s2_backward(const float* x, const float* y, const float* dy, float* dx, int B, int D, int H,
                                          int W, int C, void* stream) {
  HP_REQUIRE(x && y && dy && dx && C % 4 == 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0,
             "hp_maxpool3d_k3s2_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("maxpool3_bwd", st);
  const long n = (long)B * D * H * W * (C / 4);
  hipLaunchKernelGGL(k_maxpool3_bwd, dim3(grid_for(n)), dim3(ET), 0, st, (const float4*)x, (const float4*)y,
                     (const float4*)dy, (float4*)dx, B, D, H, W, C / 4);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_layout_transpose(const float* in, float* out, int B, long V, int C, int to_channels_first,
                                   void* stream) {
  HP_REQUIRE(in && out && B > 0 && V > 0 && C > 0, "hp_layout_transpose: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("layout_transpose", st);
  hipLaunchKernelGGL(k_transpose_vc, dim3((unsigned)((V + 31) / 32), (C + 31) / 32, B), dim3(256), 0, st, in, out, V, C,
                     to_channels_first);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// ---- stem: fused BatchNorm + ReLU + MaxPool3d(3,2,1) (posenet3d_50.py:253-257); z channels-last (B,D,H,W,C)
// workspace (floats): 2*C scale/shift | 3*C coefficients | then 2*C doubles for the reduction
extern "C" size_t hp_stem_bn_pool_workspace_bytes(int C) { return sizeof(float) * 5 * C + sizeof(double) * 2 * C + 16; }

extern "C" int hp_stem_bn_relu_pool_forward(const float* z, float* pooled, int B, int D, int H, int W, int C, const float* mean,
                                            const float* rstd, const float* gamma, const float* beta, void* workspace,
                                            void* stream) {
  HP_REQUIRE(z && pooled && mean && rstd && gamma && beta && workspace && C % 4 == 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0,
             "hp_stem_bn_relu_pool_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  float* sc = (float*)workspace;
  float* sh = sc + C;
  hipLaunchKernelGGL(k_bn_scale_shift, dim3((C + 127) / 128), dim3(128), 0, st, mean, rstd, gamma, beta, C, sc, sh);
  HP_PROF("stem_bn_relu_pool_fwd", st);
  const long n = (long)B * (D / 2) * (H / 2) * (W / 2) * (C / 4);
  // two workgroups per CU: 2.28 ms at the headline shape against 2.50 with eight and 2.88 with three (768 x 256 threads does not
  // divide the pooled tensor, which costs the slab order below)
  const unsigned grid = (unsigned)std::min<long>((n + ET - 1) / ET, 256 * 2);
  // slab order (each XCD walks a contiguous eighth of the pooled tensor) when the launch is whole rounds and the extents
  // decode with shifts; otherwise runs of 32 chunks per XCD and round, or the plain order.  HP_POOL_XCD_SLAB=0: plain (A/B runs)
  static const bool xslab = !(getenv("HP_POOL_XCD_SLAB") && atoi(getenv("HP_POOL_XCD_SLAB")) == 0);
  unsigned xg = 0u;
  if (xslab) {
    if (grid % 8 == 0 && n % ((long)grid * ET) == 0 && is_pow2(C / 4) && make_decode(D / 2, H / 2, W / 2).sw >= 0) xg = 0xffffffffu;
    else if (grid % 256 == 0) xg = 32u;
  }
  hipLaunchKernelGGL(k_bn_relu_pool3_fwd, dim3(grid), dim3(ET), 0, st, (const float4*)z, (float4*)pooled, B, D, H, W, C / 4,
                     (const float4*)sc, (const float4*)sh, make_decode(D / 2, H / 2, W / 2), is_pow2(C / 4) ? ilog2(C / 4) : -1, xg);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_stem_bn_relu_pool_backward(const float* z, const float* pooled, const float* dpooled, float* dz, int B, int D,
                                             int H, int W, int C, const float* mean, const float* rstd, const float* gamma,
                                             const float* beta, int train, float* dgamma, float* dbeta, void* workspace,
                                             void* stream) {
  HP_REQUIRE(z && pooled && dpooled && dz && mean && rstd && gamma && beta && workspace && C % 4 == 0 && (ET % (C / 4)) == 0,
             "hp_stem_bn_relu_pool_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  float* sc = (float*)workspace;
  float* sh = sc + C;
  float* ca = sh + C;
  float* cb = ca + C;
  float* cc = cb + C;
  double* red = (double*)(((uintptr_t)(cc + C) + 15) & ~(uintptr_t)15);
  hipLaunchKernelGGL(k_bn_scale_shift, dim3((C + 127) / 128), dim3(128), 0, st, mean, rstd, gamma, beta, C, sc, sh);
  HP_CHECK_HIP(hipMemsetAsync(red, 0, sizeof(double) * 2 * C, st));
  const long nvox = (long)B * D * H * W;
  const int C4 = C / 4;
  const bool tiled = C == 64 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0;
  const long ntiles = (long)B * (D / 2) * (H / 2) * ((W + ST_W - 1) / ST_W);
  static const int xslab = !(getenv("HP_POOL_XCD_SLAB") && atoi(getenv("HP_POOL_XCD_SLAB")) == 0);   // 0: plain order (A/B runs)
  if (tiled) {
    HP_PROF("stem_bn_pool_bwd_reduce", st);
    hipLaunchKernelGGL(k_stem_bwd_tiled<false>, dim3((unsigned)std::min<long>(ntiles, 256 * 8)), dim3(ET), 0, st, (const float4*)z,
                       (const float4*)pooled, (const float4*)dpooled, (float4*)nullptr, B, D, H, W, (const float4*)sc,
                       (const float4*)sh, (const float4*)mean, (const float4*)rstd, (const float4*)nullptr, red, ntiles, xslab);
  } else {
    HP_PROF("stem_bn_pool_bwd_reduce", st);
    const int vpb = ET / C4;
    hipLaunchKernelGGL(k_stem_bwd_reduce, dim3((unsigned)std::min<long>((nvox + vpb - 1) / vpb, 256 * 8)), dim3(ET), 0, st,
                       (const float4*)z, (const float4*)pooled, (const float4*)dpooled, B, D, H, W, C4, (const float4*)sc,
                       (const float4*)sh, (const float4*)mean, (const float4*)rstd, red, make_decode(D, H, W));
  }
  hipLaunchKernelGGL(k_bn_bwd_coef, dim3((C + 127) / 128), dim3(128), 0, st, red, nvox, C, mean, rstd, gamma, train, dgamma, dbeta,
                     ca, cb, cc);
  if (tiled) {
    HP_PROF("stem_bn_pool_bwd_apply", st);
    // (eight workgroups per CU: 3.66 ms at the headline shape against 3.9-4.0 with 64 per CU, 3.8 with four, 4.6 with two)
    static const long stem_apply_cap = getenv("HP_STEM_APPLY_GRID") ? atol(getenv("HP_STEM_APPLY_GRID")) : 256 * 8;
    hipLaunchKernelGGL(k_stem_bwd_tiled<true>, dim3((unsigned)std::min<long>(ntiles, stem_apply_cap)), dim3(ET), 0, st, (const float4*)z,
                       (const float4*)pooled, (const float4*)dpooled, (float4*)dz, B, D, H, W, (const float4*)sc, (const float4*)sh,
                       (const float4*)ca, (const float4*)cb, (const float4*)cc, (double*)nullptr, ntiles, xslab);
  } else {
    HP_PROF("stem_bn_pool_bwd_apply", st);
    hipLaunchKernelGGL(k_stem_bwd_apply, dim3(grid_for(nvox * C4)), dim3(ET), 0, st, (const float4*)z, (const float4*)pooled,
                       (const float4*)dpooled, (float4*)dz, B, D, H, W, C4, (const float4*)sc, (const float4*)sh, (const float4*)ca,
                       (const float4*)cb, (const float4*)cc, make_decode(D, H, W), is_pow2(C4) ? ilog2(C4) : -1);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_cast_f32_to_bf16(const float* x, void* y, long n, void* stream) {
  HP_REQUIRE(x && y && n > 0 && n % 4 == 0, "hp_cast_f32_to_bf16: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("cast_bf16", st);
  hipLaunchKernelGGL(k_cast, dim3(grid_for(n / 4)), dim3(ET), 0, st, (const void*)x, 0, y, 1, n / 4);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_cast_bf16_to_f32(const void* x, float* y, long n, void* stream) {
  HP_REQUIRE(x && y && n > 0 && n % 4 == 0, "hp_cast_bf16_to_f32: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("cast_bf16", st);
  hipLaunchKernelGGL(k_cast, dim3(grid_for(n / 4)), dim3(ET), 0, st, x, 1, (void*)y, 0, n / 4);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
