// Small memory-bound stages around the convolutions:
//  * FeatureExtraction's leaky-ReLU / residual elementwise steps (feature_extraction.py:228-256)
//  * normalize_feature (feature_propagation.py:273-286): per-volume min/max rescale to [0,10]
//  * soft-argmax decode (utils/criterion.py:96-153) with its backward
//  * BCE-with-logits + batch-global Dice loss (utils/criterion.py:348-385)
// Reductions: wavefront shuffles -> LDS -> one atomic (or one store) per block.
#include <algorithm>
#include <cfloat>

#include "hp_internal.h"

namespace hp {

constexpr int MT = 256;

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wmin(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}

// y = leaky(a [+ b], slope)
__global__ __launch_bounds__(MT) void k_leaky_add(const float4* __restrict__ a, const float4* __restrict__ b,
                                                  float4* __restrict__ y, long n4, float slope) {
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < n4; i += (long)gridDim.x * MT) {
    float4 v = a[i];
    if (b) {
      const float4 w = b[i];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    v.x = v.x > 0.f ? v.x : v.x * slope;
    v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope;
    v.w = v.w > 0.f ? v.w : v.w * slope;
    y[i] = v;
  }
}
// g = dy * (y > 0 ? 1 : slope)   (leaky ReLU keeps the sign, so the output tells the branch)
__global__ __launch_bounds__(MT) void k_leaky_bwd(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                  float4* __restrict__ g, long n4, float slope) {
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < n4; i += (long)gridDim.x * MT) {
    const float4 d = dy[i], v = y[i];
    g[i] = make_float4(v.x > 0.f ? d.x : d.x * slope, v.y > 0.f ? d.y : d.y * slope, v.z > 0.f ? d.z : d.z * slope,
                       v.w > 0.f ? d.w : d.w * slope);
  }
}

// ---- normalize_feature.  Pass 1: per-volume min / max with first-occurrence indices.
// Orderable encoding: key = (float bits made monotone) << 32 | index  -> one 64-bit atomic.
__device__ __forceinline__ unsigned int mono(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unmono(unsigned int m) {
  return __uint_as_float((m & 0x80000000u) ? (m & 0x7fffffffu) : ~m);
}

__global__ __launch_bounds__(MT) void k_minmax(const float* __restrict__ x, unsigned long long* __restrict__ keys, long V) {
  // keys[2*vol] = min key (value, index) ; keys[2*vol+1] = max key (value, ~index so the FIRST index wins)
  const long vol = blockIdx.y;
  const float* p = x + vol * V;
  unsigned long long kmin = ~0ull, kmax = 0ull;
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < V; i += (long)gridDim.x * MT) {
    const unsigned int m = mono(p[i]);
    const unsigned long long a = ((unsigned long long)m << 32) | (unsigned int)i;
    const unsigned long long b = ((unsigned long long)m << 32) | (unsigned int)(~(unsigned int)i);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(keys + 2 * vol, kmin);
    atomicMax(keys + 2 * vol + 1, kmax);
  }
}

__global__ __launch_bounds__(MT) void k_normalize_apply(const float* __restrict__ x, float* __restrict__ y,
                                                        const unsigned long long* __restrict__ keys, long V, float gain) {
  const long vol = blockIdx.y;
  const float mn = unmono((unsigned int)(keys[2 * vol] >> 32)), mx = unmono((unsigned int)(keys[2 * vol + 1] >> 32));
  const float inv = 1.0f / ((mx - mn) + 1e-15f);
  const float4* p = (const float4*)(x + vol * V);
  float4* o = (float4*)(y + vol * V);
  const long n4 = V / 4;
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < n4; i += (long)gridDim.x * MT) {
    float4 v = p[i];
    v.x = (v.x - mn) * inv * gain;
    v.y = (v.y - mn) * inv * gain;
    v.z = (v.z - mn) * inv * gain;
    v.w = (v.w - mn) * inv * gain;
    o[i] = v;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < V; i += MT) y[vol * V + i] = (x[vol * V + i] - mn) * inv * gain;
}

// backward: y = gain * (x - mn) / R, R = (mx - mn) + eps.
// dx_i = gain dy_i / R ; at argmax: -= gain S2 / R^2 ; at argmin: += gain S2 / R^2 - gain S1 / R
// with S1 = sum dy, S2 = sum dy (x - mn)
__global__ __launch_bounds__(MT) void k_normalize_bwd_reduce(const float* __restrict__ dy, const float* __restrict__ x,
                                                             const unsigned long long* __restrict__ keys,
                                                             double* __restrict__ acc, long V) {
  __shared__ float sh[2 * MT / 64];
  const long vol = blockIdx.y;
  const float mn = unmono((unsigned int)(keys[2 * vol] >> 32));
  float s1 = 0.f, s2 = 0.f;
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < V; i += (long)gridDim.x * MT) {
    const float d = dy[vol * V + i];
    s1 += d;
    s2 += d * (x[vol * V + i] - mn);
  }
  s1 = wsum(s1);
  s2 = wsum(s2);
  if ((threadIdx.x & 63) == 0) {
    sh[(threadIdx.x >> 6) * 2] = s1;
    sh[(threadIdx.x >> 6) * 2 + 1] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
    for (int w = 0; w < MT / 64; ++w) {
      a += sh[2 * w];
      b += sh[2 * w + 1];
    }
    atomicAdd(acc + 2 * vol, (double)a);
    atomicAdd(acc + 2 * vol + 1, (double)b);
  }
}

__global__ __launch_bounds__(MT) void k_normalize_bwd_apply(const float* __restrict__ dy, float* __restrict__ dx,
                                                            const unsigned long long* __restrict__ keys,
                                                            const double* __restrict__ acc, long V, float gain) {
  const long vol = blockIdx.y;
  const unsigned long long kmin = keys[2 * vol], kmax = keys[2 * vol + 1];
  const float mn = unmono((unsigned int)(kmin >> 32)), mx = unmono((unsigned int)(kmax >> 32));
  const unsigned int imin = (unsigned int)kmin, imax = ~(unsigned int)kmax;
  const double R = (double)(mx - mn) + 1e-15;
  const float inv = (float)((double)gain / R);
  const float t2 = (float)((double)gain * acc[2 * vol + 1] / (R * R)), t1 = (float)((double)gain * acc[2 * vol] / R);
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < V; i += (long)gridDim.x * MT) {
    float v = dy[vol * V + i] * inv;
    if ((unsigned int)i == imax) v -= t2;
    if ((unsigned int)i == imin) v += t2 - t1;
    dx[vol * V + i] = v;
  }
}

// ---- soft-argmax over (b, joint) heat-maps of D*H*W logits, each split over SA_CHUNKS workgroups so that the
// 4 x 24 maps fill the chip:  (1) max logit per map (ordered-integer atomicMax into stat[2 bj]),
// (2) sum exp(l - max) and the three coordinate moments (fp32 atomics into stat[2 bj + 1], out[3 bj ..]),
// (3) out /= sum and the max back as a float.
// out[bj*3 + {0,1,2}] = E[w], E[h], E[d];  stat[bj*2 + {0,1}] = max logit, sum exp(l - max)
constexpr int SA_CHUNKS = 32;

__global__ __launch_bounds__(MT) void k_softargmax_max(const float* __restrict__ heat, float* __restrict__ stat, long V) {
  __shared__ float sh[MT / 64];
  const long bj = blockIdx.y;
  const long per = (V + SA_CHUNKS - 1) / SA_CHUNKS, beg = blockIdx.x * per, end = min(V, beg + per);
  const float* p = heat + bj * V;
  float m = -FLT_MAX;
  for (long i = beg + threadIdx.x; i < end; i += MT) m = fmaxf(m, p[i]);
  m = wmax(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < MT / 64; ++w) m = fmaxf(m, sh[w]);
    atomicMax((unsigned int*)(stat + bj * 2), mono(m));
  }
}

__global__ __launch_bounds__(MT) void k_softargmax_moments(const float* __restrict__ heat, float* __restrict__ out,
                                                           float* __restrict__ stat, int D, int H, int W) {
  __shared__ float sh[4 * MT / 64];
  const long bj = blockIdx.y;
  const long V = (long)D * H * W;
  const long per = (V + SA_CHUNKS - 1) / SA_CHUNKS, beg = blockIdx.x * per, end = min(V, beg + per);
  const float* p = heat + bj * V;
  const float m = unmono(*(const unsigned int*)(stat + bj * 2));
  float s = 0.f, ex = 0.f, ey = 0.f, ez = 0.f;
  for (long i = beg + threadIdx.x; i < end; i += MT) {
    const float e = __expf(p[i] - m);
    const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((long)W * H));
    s += e;
    ex += e * (float)x;
    ey += e * (float)y;
    ez += e * (float)z;
  }
  s = wsum(s);
  ex = wsum(ex);
  ey = wsum(ey);
  ez = wsum(ez);
  if ((threadIdx.x & 63) == 0) {
    float* q = sh + (threadIdx.x >> 6) * 4;
    q[0] = s; q[1] = ex; q[2] = ey; q[3] = ez;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0, b = 0, c = 0, d = 0;
    for (int w = 0; w < MT / 64; ++w) {
      a += sh[4 * w]; b += sh[4 * w + 1]; c += sh[4 * w + 2]; d += sh[4 * w + 3];
    }
    atomicAdd(stat + bj * 2 + 1, a);
    atomicAdd(out + bj * 3 + 0, b);
    atomicAdd(out + bj * 3 + 1, c);
    atomicAdd(out + bj * 3 + 2, d);
  }
}

__global__ void k_softargmax_finish(float* __restrict__ out, float* __restrict__ stat, int BJ) {
  const int bj = blockIdx.x * blockDim.x + threadIdx.x;
  if (bj >= BJ) return;
  const float a = stat[bj * 2 + 1];
  out[bj * 3 + 0] /= a;
  out[bj * 3 + 1] /= a;
  out[bj * 3 + 2] /= a;
  stat[bj * 2] = unmono(*(const unsigned int*)(stat + bj * 2));
}

// d logit_i = p_i * sum_a g_a (coord_a(i) - E_a)
__global__ __launch_bounds__(MT) void k_softargmax_bwd(const float* __restrict__ heat, const float* __restrict__ out,
                                                       const float* __restrict__ stat, const float* __restrict__ gout,
                                                       float* __restrict__ dheat, int D, int H, int W, int chunks) {
  const long bj = blockIdx.x / chunks;
  const int chunk = blockIdx.x % chunks;
  const long V = (long)D * H * W;
  const float m = stat[bj * 2], inv = 1.0f / stat[bj * 2 + 1];
  const float gx = gout[bj * 3], gy = gout[bj * 3 + 1], gz = gout[bj * 3 + 2];
  const float ex = out[bj * 3], ey = out[bj * 3 + 1], ez = out[bj * 3 + 2];
  for (long i = (long)chunk * MT + threadIdx.x; i < V; i += (long)chunks * MT) {
    const float pr = __expf(heat[bj * V + i] - m) * inv;
    const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((long)W * H));
    dheat[bj * V + i] = pr * (gx * ((float)x - ex) + gy * ((float)y - ey) + gz * ((float)z - ez));
  }
}

// ---- BCE-with-logits + Dice.  acc[0..3] = sum bce_i, sum sig*t, sum sig, sum t  (fp64)
__global__ __launch_bounds__(MT) void k_bce_dice_reduce(const float* __restrict__ logit, const float* __restrict__ target,
                                                        double* __restrict__ acc, long n) {
  __shared__ float sh[4 * MT / 64];
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < n; i += (long)gridDim.x * MT) {
    const float x = logit[i], t = target[i];
    // max(x,0) - x t + log(1 + exp(-|x|))
    a += fmaxf(x, 0.f) - x * t + log1pf(__expf(-fabsf(x)));
    const float s = 1.0f / (1.0f + __expf(-x));
    b += s * t;
    c += s;
    d += t;
  }
  a = wsum(a); b = wsum(b); c = wsum(c); d = wsum(d);
  if ((threadIdx.x & 63) == 0) {
    float* q = sh + (threadIdx.x >> 6) * 4;
    q[0] = a; q[1] = b; q[2] = c; q[3] = d;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    float s = 0.f;
    for (int w = 0; w < MT / 64; ++w) s += sh[4 * w + threadIdx.x];
    atomicAdd(acc + threadIdx.x, (double)s);
  }
}

__global__ void k_bce_dice_finish(const double* __restrict__ acc, long n, float eps, float* __restrict__ loss) {
  const double bce = acc[0] / (double)n;
  const double dice = (2.0 * acc[1] + (double)eps) / (acc[2] + acc[3]);
  loss[0] = (float)(bce + 1.0 - dice);
}

// d loss / d x_i = gl * [ (sig - t)/n  -  dDice/dsig_i * sig (1 - sig) ],
// dDice/dsig_i = (2 t_i U - (2 I + eps)) / U^2,  U = sum sig + sum t
__global__ __launch_bounds__(MT) void k_bce_dice_bwd(const float* __restrict__ logit, const float* __restrict__ target,
                                                     const double* __restrict__ acc, const float* __restrict__ gl,
                                                     float* __restrict__ dlogit, long n, float eps, float dice_scale) {
  const double U = acc[2] + acc[3];
  const float invn = (float)(1.0 / (double)n);
  const float c1 = (float)((double)dice_scale * 2.0 / U);
  const float c0 = (float)((double)dice_scale * (2.0 * acc[1] + (double)eps) / (U * U));
  const float g = gl[0];
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < n; i += (long)gridDim.x * MT) {
    const float x = logit[i], t = target[i];
    const float s = 1.0f / (1.0f + __expf(-x));
    dlogit[i] = g * ((s - t) * invn - (t * c1 - c0) * s * (1.0f - s));
  }
}


// ---------------------------------------------------------------- weighted MSE on decoded joints (utils/criterion.py:156-162)
// loss = sum((pred - gt)^2 * w) * scale   (scale = 1/B when size_average); one workgroup: n = 72 B elements
__global__ __launch_bounds__(MT) void k_weighted_mse(const float* __restrict__ pred, const float* __restrict__ gt,
                                                     const float* __restrict__ wt, long n, float scale, float* __restrict__ loss) {
  __shared__ float sh[MT / 64];
  float a = 0.f;
  for (long i = threadIdx.x; i < n; i += MT) {
    const float d = pred[i] - gt[i];
    a += d * d * wt[i];
  }
  a = wsum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < MT / 64; ++w) s += sh[w];
    loss[0] = s * scale;
  }
}
__global__ __launch_bounds__(MT) void k_weighted_mse_bwd(const float* __restrict__ pred, const float* __restrict__ gt,
                                                         const float* __restrict__ wt, const float* __restrict__ gl, long n,
                                                         float scale, float* __restrict__ dpred) {
  const float g = gl[0] * 2.0f * scale;
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < n; i += (long)gridDim.x * MT) dpred[i] = g * (pred[i] - gt[i]) * wt[i];
}

// ---------------------------------------------------------------- VisibleNet (models/feature_propagation.py:289-312)
// x (planes = B*C, D, HW), already ReLU'd and normalised: the four largest values along depth per (plane, pixel), in
// descending order (ties: the smaller depth index first), and their depth coordinate (D-1-idx)/(D-1).
// out: vals (planes, 4, HW) and dep (planes, 4, HW).
__global__ __launch_bounds__(MT) void k_depth_top4(const float* __restrict__ x, float* __restrict__ vals, float* __restrict__ dep,
                                                   long planes, int D, long HW) {
  const long total = planes * HW;
  for (long i = (long)blockIdx.x * MT + threadIdx.x; i < total; i += (long)gridDim.x * MT) {
    const long pl = i / HW, px = i - pl * HW;
    const float* p = x + pl * D * HW + px;
    float v0 = -INFINITY, v1 = -INFINITY, v2 = -INFINITY, v3 = -INFINITY;
    int i0 = 0, i1 = 0, i2 = 0, i3 = 0;
    for (int d = 0; d < D; ++d) {
      const float v = p[(long)d * HW];
      if (v > v3) {  // strict: an equal value met later does not displace an earlier one
        if (v > v0) { v3 = v2; i3 = i2; v2 = v1; i2 = i1; v1 = v0; i1 = i0; v0 = v; i0 = d; }
        else if (v > v1) { v3 = v2; i3 = i2; v2 = v1; i2 = i1; v1 = v; i1 = d; }
        else if (v > v2) { v3 = v2; i3 = i2; v2 = v; i2 = d; }
        else { v3 = v; i3 = d; }
      }
    }
    const float dm = (float)(D - 1);  // IEEE division, as the reference's (depdim - 1 - idx) / (depdim - 1)
    float* vo = vals + pl * 4 * HW + px;
    float* dq = dep + pl * 4 * HW + px;
    vo[0] = v0; vo[HW] = v1; vo[2 * HW] = v2; vo[3 * HW] = v3;
    dq[0] = __fdiv_rn((float)(D - 1 - i0), dm); dq[HW] = __fdiv_rn((float)(D - 1 - i1), dm);
    dq[2 * HW] = __fdiv_rn((float)(D - 1 - i2), dm); dq[3 * HW] = __fdiv_rn((float)(D - 1 - i3), dm);
  }
}

// ---------------------------------------------------------------- noise augmentation (utils/nlos_pose_dataloader_noise.py:167-172)
// Gaussian blur of the FLATTENED measurement (one long 1-D signal, replicate border) with K = 2R+1 normalised taps, then
// one Poisson draw per sample with the blurred value as its mean (counter-based generator: sample i depends only on
// (seed, i), so the result does not depend on the launch geometry).
__device__ __forceinline__ unsigned int hp_hash32(unsigned int a, unsigned int b) {
  unsigned int h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u + (a << 6) + (a >> 2));
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float hp_uniform(unsigned long long seed, unsigned long long idx, unsigned int k) {
  const unsigned int h = hp_hash32(hp_hash32((unsigned int)seed ^ (unsigned int)(idx >> 32), (unsigned int)idx),
                                   (unsigned int)(seed >> 32) + 0x632BE5ABu * (k + 1u));
  return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1)
}
__device__ float hp_poisson(float lam, unsigned long long seed, unsigned long long idx) {
  if (!(lam > 0.f)) return 0.f;
  if (lam < 12.f) {  // multiplication method (Knuth): exact
    const float L = __expf(-lam);
    float prod = 1.f;
    unsigned int k = 0;
    do {
      prod *= hp_uniform(seed, idx, k);
      ++k;
    } while (prod > L && k < 200u);
    return (float)(k - 1);
  }
  // transformed rejection (Hoermann's PTRS): exact for lam >= 10
  const float slam = sqrtf(lam), loglam = __logf(lam);
  const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
  const float invalpha = 1.1239f + 1.1328f / (b - 3.4f), vr = 0.9277f - 3.6224f / (b - 2.f);
  for (unsigned int k = 0; k < 64u; ++k) {
    const float U = hp_uniform(seed, idx, 2 * k) - 0.5f, V = hp_uniform(seed, idx, 2 * k + 1);
    const float us = 0.5f - fabsf(U);
    const float kf = floorf((2.f * a / us + b) * U + lam + 0.43f);
    if (us >= 0.07f && V <= vr) return kf;
    if (kf < 0.f || (us < 0.013f && V > us)) continue;
    if (__logf(V) + __logf(invalpha) - __logf(a / (us * us) + b) <= -lam + kf * loglam - lgammaf(kf + 1.f)) return kf;
  }
  return floorf(lam + 0.5f);
}
__global__ __launch_bounds__(MT) void k_blur1d_poisson(const float* __restrict__ x, float* __restrict__ y, long n,
                                                       const float* __restrict__ taps, int R, int do_poisson,
                                                       unsigned long long seed) {
  extern __shared__ float sm[];  // taps (2R+1) then the tile with its halo (MT + 2R)
  float* tp = sm;
  float* tile = sm + (2 * R + 1);
  for (int i = threadIdx.x; i < 2 * R + 1; i += MT) tp[i] = taps[i];
  const long base = (long)blockIdx.x * MT;
  for (int i = threadIdx.x; i < MT + 2 * R; i += MT) {
    long j = base + i - R;
    j = j < 0 ? 0 : (j >= n ? n - 1 : j);  // BORDER_REPLICATE
    tile[i] = x[j];
  }
  __syncthreads();
  const long o = base + threadIdx.x;
  if (o >= n) return;
  float s = 0.f;
  for (int k = 0; k <= 2 * R; ++k) s = fmaf(tp[k], tile[threadIdx.x + k], s);
  y[o] = do_poisson ? hp_poisson(s, seed, (unsigned long long)o) : s;
}

// ---------------------------------------------------------------- 'bp' mode epilogue (models/feature_propagation.py:246-253)
// volume -> ReplicationPad3d(2) -> conv3d with the 5^3 Laplacian-of-Gaussian filter (utils/helper.py:13-32) -> first time
// slice zeroed.  One input channel, one output channel: a 125-tap stencil.  A workgroup owns an 4 x 8 x 32 output tile and
// stages its 8 x 12 x 36 halo (indices clamped = replication padding) in LDS once; the 125 weights sit in LDS as well.
constexpr int LP_Z = 4, LP_Y = 8, LP_X = 32, LP_R = 2;
constexpr int LP_HZ = LP_Z + 2 * LP_R, LP_HY = LP_Y + 2 * LP_R, LP_HX = LP_X + 2 * LP_R, LP_PX = LP_HX + 1;
__global__ __launch_bounds__(256) void k_laplacian5_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, int T, int H, int W, int tz, int ty, int tx) {
  __shared__ float tile[LP_HZ * LP_HY * LP_PX];
  __shared__ float wt[125];
  const int tid = threadIdx.x;
  int b = blockIdx.x;
  const int bx = b % tx;
  b /= tx;
  const int by = b % ty;
  b /= ty;
  const int bz = b % tz;
  const long plane = b / tz;
  const float* xp = x + plane * (long)T * H * W;
  float* yp = y + plane * (long)T * H * W;
  const int z0 = bz * LP_Z, y0 = by * LP_Y, x0 = bx * LP_X;
  if (tid < 125) wt[tid] = w[tid];
  for (int i = tid; i < LP_HZ * LP_HY * LP_HX; i += 256) {
    const int hx = i % LP_HX, hy = (i / LP_HX) % LP_HY, hz = i / (LP_HX * LP_HY);
    const int gz = min(max(z0 + hz - LP_R, 0), T - 1), gy = min(max(y0 + hy - LP_R, 0), H - 1), gx = min(max(x0 + hx - LP_R, 0), W - 1);
    tile[(hz * LP_HY + hy) * LP_PX + hx] = xp[((long)gz * H + gy) * W + gx];
  }
  __syncthreads();
  const int lx = tid & 31, ly = tid >> 5;
  const int ox = x0 + lx, oy = y0 + ly;
  if (ox >= W || oy >= H) return;
#pragma unroll
  for (int lz = 0; lz < LP_Z; ++lz) {
    const int oz = z0 + lz;
    if (oz >= T) break;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int bb = 0; bb < 5; ++bb)
#pragma unroll
        for (int c = 0; c < 5; ++c) acc = fmaf(wt[(a * 5 + bb) * 5 + c], tile[((lz + a) * LP_HY + ly + bb) * LP_PX + lx + c], acc);
    yp[((long)oz * H + oy) * W + ox] = oz == 0 ? 0.f : acc;   // volumn[:, :1] = 0 (:252)
  }
}

// adjoint of the same map: gx[i] = sum over the padded positions p that replicate voxel i (p = i inside; the two positions
// beyond the border as well for a border voxel, per axis) of sum_tap w[tap] * g[p - tap + 2], g taken as 0 outside the volume
// and in its first time slice.  A gather: no atomics, run-to-run identical.
__global__ __launch_bounds__(256) void k_laplacian5_bwd(const float* __restrict__ g, const float* __restrict__ w,
                                                        float* __restrict__ gx, int T, int H, int W, long total) {
  __shared__ float wt[125];
  if (threadIdx.x < 125) wt[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ix = (int)(i % W), iy = (int)((i / W) % H), iz = (int)((i / ((long)W * H)) % T);
  const float* gp = g + (i / ((long)T * H * W)) * (long)T * H * W;
  const int pz0 = iz == 0 ? -LP_R : iz, pz1 = iz == T - 1 ? T - 1 + LP_R : iz;
  const int py0 = iy == 0 ? -LP_R : iy, py1 = iy == H - 1 ? H - 1 + LP_R : iy;
  const int px0 = ix == 0 ? -LP_R : ix, px1 = ix == W - 1 ? W - 1 + LP_R : ix;
  float acc = 0.f;
  for (int pz = pz0; pz <= pz1; ++pz)
    for (int py = py0; py <= py1; ++py)
      for (int px = px0; px <= px1; ++px)
        for (int a = 0; a < 5; ++a) {
          const int oz = pz - a + LP_R;   // output voxel whose tap a reads padded position pz
          if (oz < 1 || oz >= T) continue;  // slice 0 of the output is zeroed: its gradient does not flow
          for (int b = 0; b < 5; ++b) {
            const int oy = py - b + LP_R;
            if (oy < 0 || oy >= H) continue;
#pragma unroll
            for (int c = 0; c < 5; ++c) {
              const int ox = px - c + LP_R;
              if (ox < 0 || ox >= W) continue;
              acc = fmaf(wt[(a * 5 + b) * 5 + c], gp[((long)oz * H + oy) * W + ox], acc);
            }
          }
        }
  gx[i] = acc;
}

static unsigned mgrid(long n) { return (unsigned)std::max<long>(1, std::min<long>((n + MT - 1) / MT, 256 * 8)); }

}  // namespace hp

using namespace hp;

extern "C" int hp_leaky_add_forward(const float* a, const float* b, float* y, long n, float slope, void* stream) {
  HP_REQUIRE(a && y && n % 4 == 0, "hp_leaky_add_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("leaky_add", st);
  hipLaunchKernelGGL(k_leaky_add, dim3(mgrid(n / 4)), dim3(MT), 0, st, (const float4*)a, (const float4*)b, (float4*)y, n / 4, slope);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_leaky_backward(const float* dy, const float* y, float* g, long n, float slope, void* stream) {
  HP_REQUIRE(dy && y && g && n % 4 == 0, "hp_leaky_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("leaky_bwd", st);
  hipLaunchKernelGGL(k_leaky_bwd, dim3(mgrid(n / 4)), dim3(MT), 0, st, (const float4*)dy, (const float4*)y, (float4*)g, n / 4, slope);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// keys: 2*nvol uint64 saved for backward
extern "C" int hp_normalize_feature_forward(const float* x, float* y, int nvol, long V, float gain, void* keys, void* stream) {
  HP_REQUIRE(x && y && keys && nvol > 0 && V > 0 && V < (1l << 32), "hp_normalize_feature_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  // min slots start at all-ones, max slots at zero
  HP_CHECK_HIP(hipMemsetAsync(keys, 0, sizeof(unsigned long long) * 2 * nvol, st));
  HP_CHECK_HIP(hipMemset2DAsync(keys, 16, 0xff, 8, nvol, st));
  const unsigned chunks = (unsigned)std::max<long>(1, std::min<long>(2048 / nvol, (V + MT - 1) / MT));
  {
    HP_PROF("normalize_minmax", st);
    hipLaunchKernelGGL(k_minmax, dim3(chunks, nvol), dim3(MT), 0, st, x, (unsigned long long*)keys, V);
  }
  {
    HP_PROF("normalize_apply", st);
    hipLaunchKernelGGL(k_normalize_apply, dim3(chunks, nvol), dim3(MT), 0, st, x, y, (const unsigned long long*)keys, V, gain);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// workspace: 2*nvol doubles
extern "C" int hp_normalize_feature_backward(const float* dy, const float* x, float* dx, int nvol, long V, float gain,
                                             const void* keys, void* workspace, void* stream) {
  HP_REQUIRE(dy && x && dx && keys && workspace && nvol > 0, "hp_normalize_feature_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_CHECK_HIP(hipMemsetAsync(workspace, 0, sizeof(double) * 2 * nvol, st));
  const unsigned chunks = (unsigned)std::max<long>(1, std::min<long>(2048 / nvol, (V + MT - 1) / MT));
  HP_PROF("normalize_bwd", st);
  hipLaunchKernelGGL(k_normalize_bwd_reduce, dim3(chunks, nvol), dim3(MT), 0, st, dy, x, (const unsigned long long*)keys,
                     (double*)workspace, V);
  hipLaunchKernelGGL(k_normalize_bwd_apply, dim3(chunks, nvol), dim3(MT), 0, st, dy, dx, (const unsigned long long*)keys,
                     (const double*)workspace, V, gain);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_softargmax_forward(const float* heat, float* joints, float* stat, int BJ, int D, int H, int W, void* stream) {
  HP_REQUIRE(heat && joints && stat && BJ > 0, "hp_softargmax_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  // zero bits: the smallest ordered key for the max, 0.0f for the sums
  HP_CHECK_HIP(hipMemsetAsync(stat, 0, sizeof(float) * 2 * (size_t)BJ, st));
  HP_CHECK_HIP(hipMemsetAsync(joints, 0, sizeof(float) * 3 * (size_t)BJ, st));
  HP_PROF("softargmax_fwd", st);
  hipLaunchKernelGGL(k_softargmax_max, dim3(SA_CHUNKS, BJ), dim3(MT), 0, st, heat, stat, (long)D * H * W);
  hipLaunchKernelGGL(k_softargmax_moments, dim3(SA_CHUNKS, BJ), dim3(MT), 0, st, heat, joints, stat, D, H, W);
  hipLaunchKernelGGL(k_softargmax_finish, dim3((BJ + 63) / 64), dim3(64), 0, st, joints, stat, BJ);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_softargmax_backward(const float* heat, const float* joints, const float* stat, const float* gjoints,
                                      float* dheat, int BJ, int D, int H, int W, void* stream) {
  HP_REQUIRE(heat && joints && stat && gjoints && dheat && BJ > 0, "hp_softargmax_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long V = (long)D * H * W;
  const int chunks = (int)std::max<long>(1, std::min<long>(32, (V + MT * 8 - 1) / (MT * 8)));
  HP_PROF("softargmax_bwd", st);
  hipLaunchKernelGGL(k_softargmax_bwd, dim3((unsigned)(BJ * chunks)), dim3(MT), 0, st, heat, joints, stat, gjoints, dheat, D, H,
                     W, chunks);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// acc: 4 doubles kept for backward; loss: 1 float
extern "C" int hp_bce_dice_forward(const float* logit, const float* target, long n, float eps, double* acc, float* loss,
                                   void* stream) {
  HP_REQUIRE(logit && target && acc && loss && n > 0, "hp_bce_dice_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 4, st));
  HP_PROF("bce_dice_fwd", st);
  hipLaunchKernelGGL(k_bce_dice_reduce, dim3(mgrid(n)), dim3(MT), 0, st, logit, target, acc, n);
  hipLaunchKernelGGL(k_bce_dice_finish, dim3(1), dim3(1), 0, st, acc, n, eps, loss);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_bce_dice_backward_scaled(const float* logit, const float* target, const double* acc, const float* gloss,
                                           float* dlogit, long n, float eps, float dice_scale, void* stream) {
  HP_REQUIRE(logit && target && acc && gloss && dlogit && n > 0, "hp_bce_dice_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("bce_dice_bwd", st);
  hipLaunchKernelGGL(k_bce_dice_bwd, dim3(mgrid(n)), dim3(MT), 0, st, logit, target, acc, gloss, dlogit, n, eps, dice_scale);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_bce_dice_backward(const float* logit, const float* target, const double* acc, const float* gloss,
                                    float* dlogit, long n, float eps, void* stream) {
  return hp_bce_dice_backward_scaled(logit, target, acc, gloss, dlogit, n, eps, 1.0f, stream);
}

extern "C" int hp_bce_dice_partial(const float* logit, const float* target, long n, double* acc, void* stream) {
  HP_REQUIRE(logit && target && acc && n > 0, "hp_bce_dice_partial: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 4, st));
  HP_PROF("bce_dice_fwd", st);
  hipLaunchKernelGGL(k_bce_dice_reduce, dim3(mgrid(n)), dim3(MT), 0, st, logit, target, acc, n);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_bce_dice_finalize(const double* acc, long n, float eps, float* loss, void* stream) {
  HP_REQUIRE(acc && loss && n > 0, "hp_bce_dice_finalize: bad argument");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_bce_dice_finish, dim3(1), dim3(1), 0, st, acc, n, eps, loss);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_weighted_mse_forward(const float* pred, const float* gt, const float* weights, long n, float scale, float* loss,
                                       void* stream) {
  HP_REQUIRE(pred && gt && weights && loss && n > 0, "hp_weighted_mse_forward: bad argument");
  hipLaunchKernelGGL(k_weighted_mse, dim3(1), dim3(MT), 0, (hipStream_t)stream, pred, gt, weights, n, scale, loss);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_weighted_mse_backward(const float* pred, const float* gt, const float* weights, const float* gloss, long n,
                                        float scale, float* dpred, void* stream) {
  HP_REQUIRE(pred && gt && weights && gloss && dpred && n > 0, "hp_weighted_mse_backward: bad argument");
  hipLaunchKernelGGL(k_weighted_mse_bwd, dim3(mgrid(n)), dim3(MT), 0, (hipStream_t)stream, pred, gt, weights, gloss, n, scale, dpred);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_depth_top4(const float* x, float* vals, float* dep, long planes, int D, long HW, void* stream) {
  HP_REQUIRE(x && vals && dep && planes > 0 && D >= 4 && HW > 0, "hp_depth_top4: bad argument (depth must be >= 4)");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("depth_top4", st);
  hipLaunchKernelGGL(k_depth_top4, dim3(mgrid(planes * HW)), dim3(MT), 0, st, x, vals, dep, planes, D, HW);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_noise_blur_poisson(const float* x, float* y, long n, const float* taps, int radius, int do_poisson,
                                     unsigned long long seed, void* stream) {
  HP_REQUIRE(x && y && taps && n > 0 && radius >= 0 && radius <= 2048, "hp_noise_blur_poisson: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("noise_blur_poisson", st);
  const size_t shm = sizeof(float) * (size_t)(2 * radius + 1 + MT + 2 * radius);
  hipLaunchKernelGGL(k_blur1d_poisson, dim3((unsigned)((n + MT - 1) / MT)), dim3(MT), shm, st, x, y, n, taps, radius, do_poisson, seed);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_laplacian5_forward(const float* x, const float* w125, float* y, long planes, int T, int H, int W, void* stream) {
  HP_REQUIRE(x && w125 && y && planes > 0 && T >= 1 && H >= 1 && W >= 1, "hp_laplacian5_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int tz = (T + LP_Z - 1) / LP_Z, ty = (H + LP_Y - 1) / LP_Y, tx = (W + LP_X - 1) / LP_X;
  const long blocks = planes * tz * ty * tx;
  HP_REQUIRE(blocks < (1l << 31), "hp_laplacian5_forward: too many tiles");
  HP_PROF("laplacian5_fwd", st);
  hipLaunchKernelGGL(k_laplacian5_fwd, dim3((unsigned)blocks), dim3(256), 0, st, x, w125, y, T, H, W, tz, ty, tx);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_laplacian5_backward(const float* dy, const float* w125, float* dx, long planes, int T, int H, int W, void* stream) {
  HP_REQUIRE(dy && w125 && dx && planes > 0 && T >= 1 && H >= 1 && W >= 1, "hp_laplacian5_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const long total = planes * T * H * W;
  HP_REQUIRE((total + 255) / 256 < (1l << 31), "hp_laplacian5_backward: too many voxels");
  HP_PROF("laplacian5_bwd", st);
  hipLaunchKernelGGL(k_laplacian5_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, w125, dx, T, H, W, total);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
