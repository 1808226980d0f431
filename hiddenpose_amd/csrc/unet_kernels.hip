// Memory-bound stages of UNet3d (unet/unet3d.py) on planar fp32 (B, C, D, H, W):
// GroupNorm(4)+ReLU forward/backward, MaxPool3d(2), trilinear x2 upsampling with
// align_corners=True written straight into the concatenation buffer, and the 1x1x1 output
// convolution.  One pass per tensor wherever the statistics allow it, 16-byte accesses.
#include <algorithm>

#include "hp_internal.h"

namespace hp {

constexpr int UT = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// block-wide sum of two values; result valid in thread 0
__device__ __forceinline__ void block_sum2(float& a, float& b, float* sh) {
  a = wave_sum(a);
  b = wave_sum(b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sh[wave * 2] = a;
    sh[wave * 2 + 1] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = 0.f;
    b = 0.f;
    for (int w = 0; w < UT / 64; ++w) {
      a += sh[w * 2];
      b += sh[w * 2 + 1];
    }
  }
  __syncthreads();
}

// ---- GroupNorm forward: per (b, channel) sum / sum of squares (fp64 atomics), then apply.
// grid (chunks, B*C); acc[(b*C + c)*2 + {0,1}]
__global__ __launch_bounds__(UT) void k_plane_stats(const float* __restrict__ z, double* __restrict__ acc, long V) {
  __shared__ float sh[2 * UT / 64];
  const long plane = blockIdx.y;
  const float4* p = (const float4*)(z + plane * V);
  const long n4 = V / 4;
  float s = 0.f, q = 0.f;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < n4; i += (long)gridDim.x * UT) {
    const float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
    q += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < V; i += UT) {
      const float v = z[plane * V + i];
      s += v;
      q += v * v;
    }
  block_sum2(s, q, sh);
  if (threadIdx.x == 0) {
    atomicAdd(acc + plane * 2, (double)s);
    atomicAdd(acc + plane * 2 + 1, (double)q);
  }
}

// per (b, group): mean / rstd from the per-channel sums; per (b, c): scale/shift of the affine map
__global__ void k_gn_finalize(const double* __restrict__ acc, int B, int C, int G, long V, float eps,
                              const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ mean,
                              float* __restrict__ rstd, float* __restrict__ scale, float* __restrict__ shift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C, cpg = C / G, g = c / cpg;
  double s = 0, q = 0;
  for (int k = 0; k < cpg; ++k) {
    s += acc[((long)b * C + g * cpg + k) * 2];
    q += acc[((long)b * C + g * cpg + k) * 2 + 1];
  }
  const double n = (double)cpg * (double)V;
  const double m = s / n;
  double var = q / n - m * m;
  if (var < 0) var = 0;
  const double r = 1.0 / sqrt(var + (double)eps);
  if (c % cpg == 0) {
    mean[b * G + g] = (float)m;
    rstd[b * G + g] = (float)r;
  }
  scale[i] = (float)(r * gamma[c]);
  shift[i] = (float)(beta[c] - m * r * gamma[c]);
}

// y = relu(z * scale[plane] + shift[plane]);  grid (chunks, B*C)
__global__ __launch_bounds__(UT) void k_affine_relu(const float* __restrict__ z, float* __restrict__ y,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    long V) {
  const long plane = blockIdx.y;
  const float a = scale[plane], c = shift[plane];
  const float4* p = (const float4*)(z + plane * V);
  float4* o = (float4*)(y + plane * V);
  const long n4 = V / 4;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < n4; i += (long)gridDim.x * UT) {
    float4 v = p[i];
    v.x = fmaxf(fmaf(v.x, a, c), 0.f);
    v.y = fmaxf(fmaf(v.y, a, c), 0.f);
    v.z = fmaxf(fmaf(v.z, a, c), 0.f);
    v.w = fmaxf(fmaf(v.w, a, c), 0.f);
    o[i] = v;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < V; i += UT) y[plane * V + i] = fmaxf(fmaf(z[plane * V + i], a, c), 0.f);
}

// ---- GroupNorm backward.  g = dy * [y > 0];  per plane: S1 = sum g, S2 = sum g * z
// The ReLU mask is rebuilt from z with the forward's own expression (fmaf(z, scale, shift) > 0), so the normalised
// tensor y is not read back: two streams per pass instead of three.
__global__ __launch_bounds__(UT) void k_gn_bwd_reduce(const float* __restrict__ dy, const float* __restrict__ z,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      double* __restrict__ acc, long V) {
  __shared__ float sh[2 * UT / 64];
  const long plane = blockIdx.y;
  const float fa = scale[plane], fc = shift[plane];
  const float4* pd = (const float4*)(dy + plane * V);
  const float4* pz = (const float4*)(z + plane * V);
  const long n4 = V / 4;
  float s = 0.f, q = 0.f;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < n4; i += (long)gridDim.x * UT) {
    const float4 d = pd[i], zz = pz[i];
    const float g0 = fmaf(zz.x, fa, fc) > 0.f ? d.x : 0.f, g1 = fmaf(zz.y, fa, fc) > 0.f ? d.y : 0.f,
                g2 = fmaf(zz.z, fa, fc) > 0.f ? d.z : 0.f, g3 = fmaf(zz.w, fa, fc) > 0.f ? d.w : 0.f;
    s += g0 + g1 + g2 + g3;
    q += g0 * zz.x + g1 * zz.y + g2 * zz.z + g3 * zz.w;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < V; i += UT) {
      const float zv = z[plane * V + i];
      const float g = fmaf(zv, fa, fc) > 0.f ? dy[plane * V + i] : 0.f;
      s += g;
      q += g * zv;
    }
  block_sum2(s, q, sh);
  if (threadIdx.x == 0) {
    atomicAdd(acc + plane * 2, (double)s);
    atomicAdd(acc + plane * 2 + 1, (double)q);
  }
}

// zhat = (z - m) r.  dgamma_c = sum_b (S2 - m S1) r ; dbeta_c = sum_b S1
// dz = r gamma_c g - r (A + zhat Bq)/n,  A = sum_{c in grp} gamma_c S1_c,  Bq = sum gamma_c (S2_c - m S1_c) r
//    = ca * g + cb * z + cc   per plane
__global__ void k_gn_bwd_coef(const double* __restrict__ acc, int B, int C, int G, long V, const float* __restrict__ mean,
                              const float* __restrict__ rstd, const float* __restrict__ gamma, float* __restrict__ dgamma,
                              float* __restrict__ dbeta, float* __restrict__ ca, float* __restrict__ cb,
                              float* __restrict__ cc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int cpg = C / G;
  if (i < C) {  // parameter gradients: reduce over the batch
    double dg = 0, dbt = 0;
    const int g = i / cpg;
    for (int b = 0; b < B; ++b) {
      const double m = mean[b * G + g], r = rstd[b * G + g];
      const double s1 = acc[((long)b * C + i) * 2], s2 = acc[((long)b * C + i) * 2 + 1];
      dg += (s2 - m * s1) * r;
      dbt += s1;
    }
    dgamma[i] = (float)dg;
    dbeta[i] = (float)dbt;
  }
  if (i < B * C) {
    const int b = i / C, c = i % C, g = c / cpg;
    const double m = mean[b * G + g], r = rstd[b * G + g];
    double A = 0, Bq = 0;
    for (int k = 0; k < cpg; ++k) {
      const int cj = g * cpg + k;
      const double s1 = acc[((long)b * C + cj) * 2], s2 = acc[((long)b * C + cj) * 2 + 1];
      A += gamma[cj] * s1;
      Bq += gamma[cj] * (s2 - m * s1) * r;
    }
    const double n = (double)cpg * (double)V;
    // dz = r*gamma*g - r/n * (A + (z - m) r Bq)
    ca[i] = (float)(r * gamma[c]);
    cb[i] = (float)(-r * r * Bq / n);
    cc[i] = (float)(-r * A / n + r * r * Bq * m / n);
  }
}

__global__ __launch_bounds__(UT) void k_gn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ z,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     float* __restrict__ dz, const float* __restrict__ ca,
                                                     const float* __restrict__ cb, const float* __restrict__ cc, long V) {
  const long plane = blockIdx.y;
  const float a = ca[plane], b = cb[plane], c = cc[plane];
  const float fa = scale[plane], fc = shift[plane];
  const float4* pd = (const float4*)(dy + plane * V);
  const float4* pz = (const float4*)(z + plane * V);
  float4* po = (float4*)(dz + plane * V);
  const long n4 = V / 4;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < n4; i += (long)gridDim.x * UT) {
    const float4 d = pd[i], zz = pz[i];
    float4 o;
    o.x = fmaf(a, fmaf(zz.x, fa, fc) > 0.f ? d.x : 0.f, fmaf(b, zz.x, c));
    o.y = fmaf(a, fmaf(zz.y, fa, fc) > 0.f ? d.y : 0.f, fmaf(b, zz.y, c));
    o.z = fmaf(a, fmaf(zz.z, fa, fc) > 0.f ? d.z : 0.f, fmaf(b, zz.z, c));
    o.w = fmaf(a, fmaf(zz.w, fa, fc) > 0.f ? d.w : 0.f, fmaf(b, zz.w, c));
    po[i] = o;
  }
  if (blockIdx.x == 0)
    for (long i = n4 * 4 + threadIdx.x; i < V; i += UT) {
      const long j = plane * V + i;
      dz[j] = fmaf(a, fmaf(z[j], fa, fc) > 0.f ? dy[j] : 0.f, fmaf(b, z[j], c));
    }
}

// ---- MaxPool3d(2, 2): one thread per pooled voxel
__global__ __launch_bounds__(UT) void k_maxpool2_fwd(const float* __restrict__ x, float* __restrict__ y, long planes, int D,
                                                     int H, int W) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long total = planes * Do * Ho * Wo;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < total; i += (long)gridDim.x * UT) {
    const int ow = (int)(i % Wo);
    long t = i / Wo;
    const int oh = (int)(t % Ho);
    t /= Ho;
    const int od = (int)(t % Do);
    const long pl = t / Do;
    const float* p = x + ((pl * D + 2 * od) * H + 2 * oh) * W + 2 * ow;
    float m = -INFINITY;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float2 v = *(const float2*)(p + ((long)a * H + b) * W);
        m = fmaxf(m, fmaxf(v.x, v.y));
      }
    y[i] = m;
  }
}

// gradient goes to the FIRST maximum of each window (torch semantics)
// `add` (optional): a second gradient of the pooled tensor's INPUT, summed in the same pass -- the U-Net's skip tensor feeds the
// pool and the decoder's concatenation, and its two gradients used to meet in a separate accumulation pass.  `add` is read in
// place from the concatenation's gradient: channel c of sample b lies at add + (b * add_bstride + c * D*H*W).
__global__ __launch_bounds__(UT) void k_maxpool2_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                     float* __restrict__ dx, long planes, int D, int H, int W,
                                                     const float* __restrict__ add, long add_bstride, int C) {
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const long total = planes * Do * Ho * Wo;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < total; i += (long)gridDim.x * UT) {
    const int ow = (int)(i % Wo);
    long t = i / Wo;
    const int oh = (int)(t % Ho);
    t /= Ho;
    const int od = (int)(t % Do);
    const long pl = t / Do;
    const long base = ((pl * D + 2 * od) * H + 2 * oh) * W + 2 * ow;
    float v[8];
    float m = -INFINITY;
    int arg = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      v[k] = x[base + ((long)(k >> 2) * H + ((k >> 1) & 1)) * W + (k & 1)];
      if (v[k] > m) {
        m = v[k];
        arg = k;
      }
    }
    const float g = dy[i];
    const float* ap = nullptr;
    if (add) {
      const long bb = pl / C;
      ap = add + bb * add_bstride + (pl - bb * C) * ((long)D * H * W) + (base - pl * ((long)D * H * W));
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int k0 = a * 4 + b * 2;
        float2 o = make_float2(arg == k0 ? g : 0.f, arg == k0 + 1 ? g : 0.f);
        if (add) {
          const float2 e = *(const float2*)(ap + ((long)a * H + b) * W);
          o.x += e.x;
          o.y += e.y;
        }
        *(float2*)(dx + base + ((long)a * H + b) * W) = o;
      }
  }
}

constexpr int UPS_TAB = 2048;  // D + H + W of the INPUT must fit the per-axis tables of the two kernels below
// ---- trilinear x2, align_corners=True.  src = o * (n-1)/(2n-1).
// Forward writes into channel slice [c_off, c_off + C) of a (B, Ctot, 2D, 2H, 2W) buffer.
__device__ __forceinline__ void lerp_src(int o, int n, int& i0, int& i1, float& w1) {
  const float s = n > 1 ? (float)o * (float)(n - 1) / (float)(2 * n - 1) : 0.f;
  i0 = (int)s;
  if (i0 > n - 1) i0 = n - 1;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  w1 = s - (float)i0;
}

__global__ __launch_bounds__(UT) void k_upsample2_fwd(const float* __restrict__ x, float* __restrict__ y, int B, int C, int D,
                                                      int H, int W, int Ctot, int c_off, int sw, int sh, int sd) {
  // interpolation sources per output position of each axis (lerp_src, unchanged), built once per workgroup
  __shared__ int f_i0[2 * UPS_TAB], f_i1[2 * UPS_TAB];
  __shared__ float f_w[2 * UPS_TAB];
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  for (int e = threadIdx.x; e < Do + Ho + Wo; e += UT) {
    const int n = e < Do ? D : e < Do + Ho ? H : W;
    const int o = e < Do ? e : e < Do + Ho ? e - Do : e - Do - Ho;
    lerp_src(o, n, f_i0[e], f_i1[e], f_w[e]);
  }
  __syncthreads();
  const long total = (long)B * C * Do * Ho * Wo;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < total; i += (long)gridDim.x * UT) {
    int ow, oh, od, c, b;
    if (sw >= 0) {  // power-of-two extents: shifts (five 64-bit divisions cost ~200 instructions per element)
      ow = (int)(i & (Wo - 1));
      oh = (int)((i >> sw) & (Ho - 1));
      od = (int)((i >> (sw + sh)) & (Do - 1));
      const unsigned bc = (unsigned)(i >> (sw + sh + sd));
      c = (int)(bc % (unsigned)C);
      b = (int)(bc / (unsigned)C);
    } else {
      ow = (int)(i % Wo);
      long t = i / Wo;
      oh = (int)(t % Ho);
      t /= Ho;
      od = (int)(t % Do);
      t /= Do;
      c = (int)(t % C);
      b = (int)(t / C);
    }
    const int d0 = f_i0[od], d1 = f_i1[od], h0 = f_i0[Do + oh], h1 = f_i1[Do + oh], w0 = f_i0[Do + Ho + ow], w1 = f_i1[Do + Ho + ow];
    const float fd = f_w[od], fh = f_w[Do + oh], fw = f_w[Do + Ho + ow];
    const float* p = x + ((long)b * C + c) * D * H * W;
    auto at = [&](int a, int bb, int cc) { return p[((long)a * H + bb) * W + cc]; };
    const float v00 = at(d0, h0, w0) * (1.f - fw) + at(d0, h0, w1) * fw;
    const float v01 = at(d0, h1, w0) * (1.f - fw) + at(d0, h1, w1) * fw;
    const float v10 = at(d1, h0, w0) * (1.f - fw) + at(d1, h0, w1) * fw;
    const float v11 = at(d1, h1, w0) * (1.f - fw) + at(d1, h1, w1) * fw;
    const float v0 = v00 * (1.f - fh) + v01 * fh, v1 = v10 * (1.f - fh) + v11 * fh;
    y[(((long)b * Ctot + c_off + c) * Do + od) * Ho * Wo + (long)oh * Wo + ow] = v0 * (1.f - fd) + v1 * fd;
  }
}

// adjoint in gather form: every input voxel sums the (<= 4 per axis) outputs that read it
__device__ __forceinline__ int adj_range(int i, int n, int& lo) {
  // outputs o with i0(o) == i or i1(o) == i lie in [ceil((i-1)/r), floor((i+1)/r)], r = (n-1)/(2n-1)
  if (n == 1) {
    lo = 0;
    return 2;
  }
  const float inv = (float)(2 * n - 1) / (float)(n - 1);
  int a = (int)floorf((float)(i - 1) * inv) - 1, b = (int)ceilf((float)(i + 1) * inv) + 1;
  if (a < 0) a = 0;
  if (b > 2 * n - 1) b = 2 * n - 1;
  lo = a;
  return b - a + 1;
}

// Per-axis adjoint tables in LDS: for input position p the outputs lo .. lo + cnt - 1 read it with weights wt[0 .. cnt)
// (cnt <= 5).  They depend on the axis length only, so a workgroup builds them once (D + H + W entries) instead of
// re-deriving up to 15 interpolation sources per voxel; the arithmetic of every weight is lerp_src's, unchanged.

__global__ __launch_bounds__(UT) void k_upsample2_bwd(const float* __restrict__ dy, float* __restrict__ dx, int B, int C,
                                                      int D, int H, int W, int Ctot, int c_off, int sw, int sh, int sd) {
  __shared__ int t_lo[UPS_TAB], t_cnt[UPS_TAB];
  __shared__ float t_wt[UPS_TAB * 5];
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  for (int e = threadIdx.x; e < D + H + W; e += UT) {
    const int n = e < D ? D : e < D + H ? H : W;
    const int pos = e < D ? e : e < D + H ? e - D : e - D - H;
    int lo;
    const int m = adj_range(pos, n, lo);
    int first = -1, cnt = 0;
    float wts[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < m; ++a) {
      int i0, i1;
      float f;
      lerp_src(lo + a, n, i0, i1, f);
      const float wgt = (i0 == pos ? 1.f - f : 0.f) + (i1 == pos ? f : 0.f);
      if (wgt != 0.f) {
        if (first < 0) first = lo + a;
        const int k = lo + a - first;  // outputs with a non-zero weight are consecutive; zero weights in between stay 0
        if (k < 5) {
          wts[k] = wgt;
          cnt = k + 1;
        }
      }
    }
    t_lo[e] = first < 0 ? 0 : first;
    t_cnt[e] = cnt;
#pragma unroll
    for (int k = 0; k < 5; ++k) t_wt[e * 5 + k] = wts[k];
  }
  __syncthreads();
  const long total = (long)B * C * D * H * W;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < total; i += (long)gridDim.x * UT) {
    int w, h, d, c, b;
    if (sw >= 0) {
      w = (int)(i & (W - 1));
      h = (int)((i >> sw) & (H - 1));
      d = (int)((i >> (sw + sh)) & (D - 1));
      const unsigned bc = (unsigned)(i >> (sw + sh + sd));
      c = (int)(bc % (unsigned)C);
      b = (int)(bc / (unsigned)C);
    } else {
      w = (int)(i % W);
      long t = i / W;
      h = (int)(t % H);
      t /= H;
      d = (int)(t % D);
      t /= D;
      c = (int)(t % C);
      b = (int)(t / C);
    }
    const int ed = d, eh = D + h, ew = D + H + w;
    const int od0 = t_lo[ed], nd = t_cnt[ed], oh0 = t_lo[eh], nh = t_cnt[eh], ow0 = t_lo[ew], nw = t_cnt[ew];
    float ww[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) ww[k] = t_wt[ew * 5 + k];
    const float* p = dy + ((long)b * Ctot + c_off + c) * Do * Ho * Wo;
    float acc = 0.f;
    for (int a = 0; a < nd; ++a) {
      const float wa = t_wt[ed * 5 + a];
      for (int bb = 0; bb < nh; ++bb) {
        const float wab = wa * t_wt[eh * 5 + bb];
        const float* row = p + ((long)(od0 + a) * Ho + (oh0 + bb)) * Wo;
        // the <= 5 samples of a row are read unconditionally (clamped) and weighted 0 beyond nw
        float r[5];
#pragma unroll
        for (int cc = 0; cc < 5; ++cc) r[cc] = row[min(ow0 + cc, Wo - 1)];
#pragma unroll
        for (int cc = 0; cc < 5; ++cc)
          if (cc < nw) acc += wab * ww[cc] * r[cc];
      }
    }
    dx[i] = acc;
  }
}

// ---- separable adjoint: trilinear interpolation is a product of three 1-D maps, so its adjoint is three 1-D passes
// (w, h, d), each summing the <= 5 outputs of ONE axis that read an input position:
//     dst[plane][r][p][in] = sum_k wt[p][k] * src[plane'][r][lo[p] + k][in],   p < n (the axis shrinks 2n -> n)
// 5 loads per element and pass instead of up to 125 in the fused gather; the traffic of the three passes together is
// 2.6x the gradient tensor (1 + 1/2, 1/2 + 1/4, 1/4 + 1/8).  VEC = 4: four consecutive `in` per thread (h and d passes).
// The first pass reads the channel slice [c_off, c_off + C) of the (B, Ctot, ...) gradient: plane' = b * Ctot + c_off + c.
template <int VEC>
__global__ __launch_bounds__(UT) void k_ups_adj_axis(const float* __restrict__ src, float* __restrict__ dst, unsigned total,
                                                     int n, unsigned R, unsigned inner, int C, int Ctot, int c_off, int s_in,
                                                     int s_n, int s_r) {
  __shared__ int t_lo[1024], t_cnt[1024];
  __shared__ float t_wt[1024 * 5];
  for (int pos = threadIdx.x; pos < n; pos += UT) {
    int lo;
    const int m = adj_range(pos, n, lo);
    int first = -1, cnt = 0;
    float wts[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < m; ++a) {
      int i0, i1;
      float f;
      lerp_src(lo + a, n, i0, i1, f);
      const float wgt = (i0 == pos ? 1.f - f : 0.f) + (i1 == pos ? f : 0.f);
      if (wgt != 0.f) {
        if (first < 0) first = lo + a;
        const int k = lo + a - first;
        if (k < 5) {
          wts[k] = wgt;
          cnt = k + 1;
        }
      }
    }
    t_lo[pos] = first < 0 ? 0 : first;
    t_cnt[pos] = cnt;
#pragma unroll
    for (int k = 0; k < 5; ++k) t_wt[pos * 5 + k] = wts[k];
  }
  __syncthreads();
  const unsigned innerv = inner / VEC;
  for (unsigned i = blockIdx.x * UT + threadIdx.x; i < total; i += gridDim.x * UT) {
    unsigned in, p, r, pl;
    if (s_in >= 0) {  // power-of-two extents
      in = i & (innerv - 1);
      p = (i >> s_in) & (unsigned)(n - 1);
      r = (i >> (s_in + s_n)) & (R - 1);
      pl = i >> (s_in + s_n + s_r);
    } else {
      in = i % innerv;
      unsigned t = i / innerv;
      p = t % (unsigned)n;
      t /= (unsigned)n;
      r = t % R;
      pl = t / R;
    }
    const unsigned spl = Ctot > 0 ? (pl / (unsigned)C) * (unsigned)Ctot + (unsigned)c_off + pl % (unsigned)C : pl;
    const int lo = t_lo[p], cnt = t_cnt[p];
    const float* sp = src + (((long)spl * R + r) * (2 * n) + lo) * (long)inner + (long)in * VEC;
    const int last = 2 * n - 1 - lo;  // clamp: rows beyond the axis are read (weight 0) at its last position
    if constexpr (VEC == 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      float4 v[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] = *(const float4*)(sp + (long)min(k, last) * inner);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float wk = k < cnt ? t_wt[p * 5 + k] : 0.f;
        acc.x += wk * v[k].x;
        acc.y += wk * v[k].y;
        acc.z += wk * v[k].z;
        acc.w += wk * v[k].w;
      }
      *(float4*)(dst + (((long)pl * R + r) * n + p) * (long)inner + (long)in * 4) = acc;
    } else {
      float acc = 0.f, v[5];
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] = sp[(long)min(k, last) * inner];
#pragma unroll
      for (int k = 0; k < 5; ++k) acc += (k < cnt ? t_wt[p * 5 + k] : 0.f) * v[k];
      dst[(((long)pl * R + r) * n + p) * (long)inner + in] = acc;
    }
  }
}

// ---- separable forward, in the reference's own order of interpolation (w, then h, then d: the same expression tree as
// the fused kernel): dst[plane'][r][o][in] = src[..][i0(o)][in] * (1 - f(o)) + src[..][i1(o)][in] * f(o), o < 2n.
// Two loads per element and pass instead of eight; the last pass (d) writes the channel slice of the concat buffer with
// 16-byte stores.
template <int VEC>
__global__ __launch_bounds__(UT) void k_ups_interp_axis(const float* __restrict__ src, float* __restrict__ dst, unsigned total,
                                                        int n, unsigned R, unsigned inner, int C, int Ctot, int c_off, int s_in,
                                                        int s_n, int s_r) {
  __shared__ int f_i0[2048], f_i1[2048];
  __shared__ float f_w[2048];
  for (int o = threadIdx.x; o < 2 * n; o += UT) lerp_src(o, n, f_i0[o], f_i1[o], f_w[o]);
  __syncthreads();
  const unsigned innerv = inner / VEC;
  for (unsigned i = blockIdx.x * UT + threadIdx.x; i < total; i += gridDim.x * UT) {
    unsigned in, o, r, pl;
    if (s_in >= 0) {
      in = i & (innerv - 1);
      o = (i >> s_in) & (unsigned)(2 * n - 1);
      r = (i >> (s_in + s_n)) & (R - 1);
      pl = i >> (s_in + s_n + s_r);
    } else {
      in = i % innerv;
      unsigned t = i / innerv;
      o = t % (unsigned)(2 * n);
      t /= (unsigned)(2 * n);
      r = t % R;
      pl = t / R;
    }
    const unsigned dpl = Ctot > 0 ? (pl / (unsigned)C) * (unsigned)Ctot + (unsigned)c_off + pl % (unsigned)C : pl;
    const int i0 = f_i0[o], i1 = f_i1[o];
    const float f = f_w[o];
    const float* sp = src + ((long)pl * R + r) * n * (long)inner + (long)in * VEC;
    float* dp = dst + (((long)dpl * R + r) * (2 * n) + o) * (long)inner + (long)in * VEC;
    if constexpr (VEC == 4) {
      const float4 a = *(const float4*)(sp + (long)i0 * inner), b = *(const float4*)(sp + (long)i1 * inner);
      *(float4*)dp = make_float4(a.x * (1.f - f) + b.x * f, a.y * (1.f - f) + b.y * f, a.z * (1.f - f) + b.z * f, a.w * (1.f - f) + b.w * f);
    } else {
      *dp = sp[(long)i0 * inner] * (1.f - f) + sp[(long)i1 * inner] * f;
    }
  }
}

// copy (B, C, V) into channel slice [c_off, c_off+C) of (B, Ctot, V), or back (gather = 1)
__global__ __launch_bounds__(UT) void k_channel_slice_copy(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                           int C, long V, int Ctot, int c_off, int gather) {
  const long n4 = (long)B * C * V / 4, v4 = V / 4;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < n4; i += (long)gridDim.x * UT) {
    const long pl = i / v4, r = i - pl * v4;
    const int b = (int)(pl / C), c = (int)(pl % C);
    const long big = (((long)b * Ctot + c_off + c) * v4 + r);
    if (gather)
      ((float4*)dst)[i] = ((const float4*)src)[big];
    else
      ((float4*)dst)[big] = ((const float4*)src)[i];
  }
}

// ---- 1x1x1 convolution with few channels (UNet `Out`: 4 -> 1)
// addend / sum_out (optional, shaped like y): sum_out = y + addend in the same pass (NlosPose.py:57: the regressor's input
// `feature + refine` leaves with the refined volume instead of through a launch of its own)
__global__ __launch_bounds__(UT) void k_conv1_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float* __restrict__ y, int B, int cin,
                                                  int cout, long V, const float* __restrict__ addend, float* __restrict__ sum_out) {
  const long total = (long)B * V;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < total; i += (long)gridDim.x * UT) {
    const long b = i / V, v = i - b * V;
    for (int co = 0; co < cout; ++co) {
      float s = bias ? bias[co] : 0.f;
      for (int ci = 0; ci < cin; ++ci) s = fmaf(x[(b * cin + ci) * V + v], w[co * cin + ci], s);
      y[(b * cout + co) * V + v] = s;
      if (sum_out) sum_out[(b * cout + co) * V + v] = addend[(b * cout + co) * V + v] + s;
    }
  }
}

template <int CIN, int COUT>
__global__ __launch_bounds__(UT) void k_conv1_bwd(const float* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ dy, float* __restrict__ dx,
                                                  float* __restrict__ dw, float* __restrict__ db, int B, long V,
                                                  const float* __restrict__ dy2) {
  // per-thread partial sums, block reduction, one atomic per block and entry (dy2, optional: a second gradient of the
  // same output, added on load -- the gradient that arrives through `feature + refine`)
  __shared__ float sh[2 * UT / 64];
  const long total = (long)B * V;
  float pw[CIN * COUT], pb[COUT], wr[CIN * COUT];
#pragma unroll
  for (int k = 0; k < CIN * COUT; ++k) {
    pw[k] = 0.f;
    wr[k] = w[k];
  }
#pragma unroll
  for (int k = 0; k < COUT; ++k) pb[k] = 0.f;
  for (long i = (long)blockIdx.x * UT + threadIdx.x; i < total; i += (long)gridDim.x * UT) {
    const long b = i / V, v = i - b * V;
    float g[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      g[co] = dy[(b * COUT + co) * V + v] + (dy2 ? dy2[(b * COUT + co) * V + v] : 0.f);
      pb[co] += g[co];
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      const float xv = x[(b * CIN + ci) * V + v];
      float s = 0.f;
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
        s = fmaf(g[co], wr[co * CIN + ci], s);
        pw[co * CIN + ci] = fmaf(g[co], xv, pw[co * CIN + ci]);
      }
      dx[(b * CIN + ci) * V + v] = s;
    }
  }
#pragma unroll
  for (int k = 0; k < CIN * COUT; ++k) {
    float a = pw[k], z = 0.f;
    block_sum2(a, z, sh);
    if (threadIdx.x == 0) atomicAdd(dw + k, a);
  }
#pragma unroll
  for (int k = 0; k < COUT; ++k) {
    float a = pb[k], z = 0.f;
    block_sum2(a, z, sh);
    if (threadIdx.x == 0 && db) atomicAdd(db + k, a);
  }
}

// log2 of the three extents when all are powers of two, else -1 each
static void pow2_shifts(int d, int h, int w, int& sd, int& sh, int& sw) {
  auto lg = [](int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
  };
  sd = lg(d), sh = lg(h), sw = lg(w);
  if (sd < 0 || sh < 0 || sw < 0) sd = sh = sw = -1;
}
static unsigned ugrid(long n) { return (unsigned)std::max<long>(1, std::min<long>((n + UT - 1) / UT, 256 * 8)); }
static unsigned chunks_for(long V, long planes) {
  // enough blocks to fill the chip without shredding small planes
  const long want = std::max<long>(1, 2048 / std::max<long>(planes, 1));
  return (unsigned)std::max<long>(1, std::min<long>(want, (V / 4 + UT - 1) / UT));
}

}  // namespace hp

using namespace hp;

extern "C" size_t hp_groupnorm_workspace_bytes(int B, int C) { return sizeof(double) * 2 * B * C + sizeof(float) * 3 * B * C; }

// y = relu(GroupNorm(z)); mean/rstd: (B*G) floats and scale/shift: (B*C) floats of the affine map, all saved for
// backward; workspace: hp_groupnorm_workspace_bytes.  chan_stats (optional): per (b, c) {sum, sum of squares} of z as
// doubles, e.g. from the producing convolution's epilogue (hp_dconv3_forward_fused) -- the statistics pass is skipped.
extern "C" int hp_groupnorm_relu_forward_v2(const float* z, float* y, int B, int C, int G, long V, const float* gamma,
                                            const float* beta, float eps, const double* chan_stats, float* mean, float* rstd,
                                            float* scale, float* shift, void* workspace, void* stream) {
  HP_REQUIRE(z && y && gamma && beta && mean && rstd && scale && shift && workspace && C % G == 0,
             "hp_groupnorm_relu_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  double* acc = (double*)workspace;
  const dim3 grid(chunks_for(V, (long)B * C), (unsigned)(B * C));
  if (!chan_stats) {
    HP_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 2 * B * C, st));
    HP_PROF("gn_stats", st);
    hipLaunchKernelGGL(k_plane_stats, grid, dim3(UT), 0, st, z, acc, V);
  }
  hipLaunchKernelGGL(k_gn_finalize, dim3((B * C + 127) / 128), dim3(128), 0, st, chan_stats ? chan_stats : acc, B, C, G, V, eps,
                     gamma, beta, mean, rstd, scale, shift);
  {
    HP_PROF("gn_apply_relu", st);
    hipLaunchKernelGGL(k_affine_relu, grid, dim3(UT), 0, st, z, y, scale, shift, V);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_groupnorm_relu_forward(const float* z, float* y, int B, int C, int G, long V, const float* gamma,
                                         const float* beta, float eps, float* mean, float* rstd, void* workspace,
                                         void* stream) {
  HP_REQUIRE(workspace, "hp_groupnorm_relu_forward: bad argument");
  float* scale = (float*)((double*)workspace + 2 * (long)B * C);
  return hp_groupnorm_relu_forward_v2(z, y, B, C, G, V, gamma, beta, eps, nullptr, mean, rstd, scale, scale + (long)B * C,
                                      workspace, stream);
}

// dz, dgamma, dbeta of y = relu(GroupNorm(z)) given dy; scale/shift: the forward's affine map (the ReLU mask is
// rebuilt from z with it, y is not read)
extern "C" int hp_groupnorm_relu_backward_v2(const float* dy, const float* z, float* dz, int B, int C, int G, long V,
                                             const float* gamma, const float* mean, const float* rstd, const float* scale,
                                             const float* shift, float* dgamma, float* dbeta, void* workspace, void* stream) {
  HP_REQUIRE(dy && z && dz && gamma && mean && rstd && scale && shift && dgamma && dbeta && workspace && C % G == 0,
             "hp_groupnorm_relu_backward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  double* acc = (double*)workspace;
  float* ca = (float*)(acc + 2 * (long)B * C);
  float* cb = ca + (long)B * C;
  float* cc = cb + (long)B * C;
  HP_CHECK_HIP(hipMemsetAsync(acc, 0, sizeof(double) * 2 * B * C, st));
  const dim3 grid(chunks_for(V, (long)B * C), (unsigned)(B * C));
  {
    HP_PROF("gn_bwd_reduce", st);
    hipLaunchKernelGGL(k_gn_bwd_reduce, grid, dim3(UT), 0, st, dy, z, scale, shift, acc, V);
  }
  hipLaunchKernelGGL(k_gn_bwd_coef, dim3((B * C + 127) / 128), dim3(128), 0, st, acc, B, C, G, V, mean, rstd, gamma, dgamma,
                     dbeta, ca, cb, cc);
  {
    HP_PROF("gn_bwd_apply", st);
    hipLaunchKernelGGL(k_gn_bwd_apply, grid, dim3(UT), 0, st, dy, z, scale, shift, dz, ca, cb, cc, V);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_maxpool3d_k2_forward(const float* x, float* y, long planes, int D, int H, int W, void* stream) {
  HP_REQUIRE(x && y && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "hp_maxpool3d_k2_forward: even sizes required");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("maxpool2_fwd", st);
  hipLaunchKernelGGL(k_maxpool2_fwd, dim3(ugrid(planes * (D / 2) * (H / 2) * (W / 2))), dim3(UT), 0, st, x, y, planes, D, H, W);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_maxpool3d_k2_backward(const float* x, const float* dy, float* dx, long planes, int D, int H, int W,
                                        void* stream) {
  HP_REQUIRE(x && dy && dx && D % 2 == 0 && H % 2 == 0 && W % 2 == 0, "hp_maxpool3d_k2_backward: even sizes required");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("maxpool2_bwd", st);
  hipLaunchKernelGGL(k_maxpool2_bwd, dim3(ugrid(planes * (D / 2) * (H / 2) * (W / 2))), dim3(UT), 0, st, x, dy, dx, planes, D, H,
                     W, (const float*)nullptr, 0l, 1);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_maxpool3d_k2_backward_add(const float* x, const float* dy, const float* add, long add_batch_stride, float* dx, int B,
                                            int C, int D, int H, int W, void* stream) {
  HP_REQUIRE(x && dy && add && dx && B > 0 && C > 0 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0,
             "hp_maxpool3d_k2_backward_add: bad argument (even sizes required)");
  HP_REQUIRE(add_batch_stride >= (long)C * D * H * W && (add_batch_stride & 1) == 0 && ((uintptr_t)add & 7) == 0,
             "hp_maxpool3d_k2_backward_add: the addend's samples must not overlap and its rows must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("maxpool2_bwd", st);
  const long planes = (long)B * C;
  hipLaunchKernelGGL(k_maxpool2_bwd, dim3(ugrid(planes * (D / 2) * (H / 2) * (W / 2))), dim3(UT), 0, st, x, dy, dx, planes, D, H,
                     W, add, add_batch_stride, C);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_upsample_trilinear2x_forward(const float* x, float* y, int B, int C, int D, int H, int W, int Ctot,
                                               int c_off, void* stream) {
  HP_REQUIRE(x && y && c_off >= 0 && c_off + C <= Ctot, "hp_upsample_trilinear2x_forward: bad argument");
  HP_REQUIRE(D + H + W <= UPS_TAB, "hp_upsample_trilinear2x_forward: D + H + W must not exceed %d", UPS_TAB);
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("upsample2_fwd", st);
  int sw, sh, sd;
  pow2_shifts(2 * D, 2 * H, 2 * W, sd, sh, sw);
  hipLaunchKernelGGL(k_upsample2_fwd, dim3(ugrid((long)B * C * D * H * W * 8)), dim3(UT), 0, st, x, y, B, C, D, H, W, Ctot, c_off, sw,
                     sh, sd);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_upsample_trilinear2x_backward(const float* dy, float* dx, int B, int C, int D, int H, int W, int Ctot,
                                                int c_off, void* stream) {
  HP_REQUIRE(dy && dx && c_off >= 0 && c_off + C <= Ctot, "hp_upsample_trilinear2x_backward: bad argument");
  HP_REQUIRE(D + H + W <= UPS_TAB, "hp_upsample_trilinear2x_backward: D + H + W must not exceed %d", UPS_TAB);
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("upsample2_bwd", st);
  int sw, sh, sd;
  pow2_shifts(D, H, W, sd, sh, sw);
  hipLaunchKernelGGL(k_upsample2_bwd, dim3(ugrid((long)B * C * D * H * W)), dim3(UT), 0, st, dy, dx, B, C, D, H, W, Ctot, c_off, sw, sh,
                     sd);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// workspace of the separable forward: x after the w pass (B,C,D,H,2W) and after the h pass (B,C,D,2H,2W)
extern "C" size_t hp_upsample_trilinear2x_forward_workspace_bytes(int B, int C, int D, int H, int W) {
  return sizeof(float) * ((size_t)B * C * D * H * 2 * W + (size_t)B * C * D * 2 * H * 2 * W);
}

extern "C" int hp_upsample_trilinear2x_forward_ws(const float* x, float* y, int B, int C, int D, int H, int W, int Ctot, int c_off,
                                                  void* workspace, void* stream) {
  HP_REQUIRE(x && y && workspace && c_off >= 0 && c_off + C <= Ctot, "hp_upsample_trilinear2x_forward_ws: bad argument");
  const size_t n1 = (size_t)B * C * D * H * 2 * W, n2 = (size_t)B * C * D * 2 * H * 2 * W, n3 = (size_t)B * C * 2 * D * 2 * H * 2 * W;
  if (n3 >= (1ull << 32) || D > 1024 || H > 1024 || W > 1024 || W % 2 != 0)
    return hp_upsample_trilinear2x_forward(x, y, B, C, D, H, W, Ctot, c_off, stream);
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("upsample2_fwd", st);
  float* t1 = (float*)workspace;
  float* t2 = t1 + n1;
  auto lg = [](unsigned v) {
    int l = 0;
    while ((1u << l) < v) ++l;
    return (1u << l) == v ? l : -1;
  };
  auto launch = [&](int vec, const float* src, float* dst, size_t total, int n, unsigned R, unsigned inner, int ctot) {
    const unsigned innerv = inner / (unsigned)vec;
    int s_in = lg(innerv), s_n = lg((unsigned)(2 * n)), s_r = lg(R);
    if (s_in < 0 || s_n < 0 || s_r < 0) s_in = s_n = s_r = -1;
    const unsigned tot = (unsigned)(total / (size_t)vec);
    if (vec == 4)
      hipLaunchKernelGGL(k_ups_interp_axis<4>, dim3(ugrid(tot)), dim3(UT), 0, st, src, dst, tot, n, R, inner, C, ctot, c_off, s_in, s_n, s_r);
    else
      hipLaunchKernelGGL(k_ups_interp_axis<1>, dim3(ugrid(tot)), dim3(UT), 0, st, src, dst, tot, n, R, inner, C, ctot, c_off, s_in, s_n, s_r);
  };
  launch(1, x, t1, n1, W, (unsigned)(D * H), 1u, 0);                         // w: (.., D*H rows, W)   -> (.., 2W)
  launch(4, t1, t2, n2, H, (unsigned)D, (unsigned)(2 * W), 0);               // h: (.., D, H, 2W)      -> (.., 2H, 2W)
  launch(4, t2, y, n3, D, 1u, (unsigned)(4 * H * W), Ctot);                  // d: (.., D, 2H*2W)      -> slice of (B, Ctot, 2D, 2H*2W)
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// workspace of the separable backward: the gradient after the w pass (B,C,2D,2H,W) and after the h pass (B,C,2D,H,W)
extern "C" size_t hp_upsample_trilinear2x_backward_workspace_bytes(int B, int C, int D, int H, int W) {
  return sizeof(float) * ((size_t)B * C * 2 * D * 2 * H * W + (size_t)B * C * 2 * D * H * W);
}

extern "C" int hp_upsample_trilinear2x_backward_ws(const float* dy, float* dx, int B, int C, int D, int H, int W, int Ctot,
                                                   int c_off, void* workspace, void* stream) {
  HP_REQUIRE(dy && dx && workspace && c_off >= 0 && c_off + C <= Ctot, "hp_upsample_trilinear2x_backward_ws: bad argument");
  const size_t n1 = (size_t)B * C * 2 * D * 2 * H * W, n2 = (size_t)B * C * 2 * D * H * W, n3 = (size_t)B * C * D * H * W;
  // 32-bit element indices inside a pass, tables of 1024 positions per axis, float4 along w in the h and d passes
  if (n1 >= (1ull << 32) || D > 1024 || H > 1024 || W > 1024 || W % 4 != 0)
    return hp_upsample_trilinear2x_backward(dy, dx, B, C, D, H, W, Ctot, c_off, stream);
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("upsample2_bwd", st);
  float* t1 = (float*)workspace;
  float* t2 = t1 + n1;
  auto lg = [](unsigned v) {
    int l = 0;
    while ((1u << l) < v) ++l;
    return (1u << l) == v ? l : -1;
  };
  auto launch = [&](int vec, const float* src, float* dst, size_t total, int n, unsigned R, unsigned inner, int ctot) {
    const unsigned innerv = inner / (unsigned)vec;
    int s_in = lg(innerv), s_n = lg((unsigned)n), s_r = lg(R);
    if (s_in < 0 || s_n < 0 || s_r < 0) s_in = s_n = s_r = -1;
    const unsigned tot = (unsigned)(total / (size_t)vec);
    if (vec == 4)
      hipLaunchKernelGGL(k_ups_adj_axis<4>, dim3(ugrid(tot)), dim3(UT), 0, st, src, dst, tot, n, R, inner, C, ctot, c_off, s_in, s_n, s_r);
    else
      hipLaunchKernelGGL(k_ups_adj_axis<1>, dim3(ugrid(tot)), dim3(UT), 0, st, src, dst, tot, n, R, inner, C, ctot, c_off, s_in, s_n, s_r);
  };
  launch(1, dy, t1, n1, W, (unsigned)(2 * D * 2 * H), 1u, Ctot);              // w: (.., 2D*2H rows, 2W) -> (.., W)
  launch(4, t1, t2, n2, H, (unsigned)(2 * D), (unsigned)W, 0);                 // h: (.., 2D, 2H, W)      -> (.., H, W)
  launch(4, t2, dx, n3, D, 1u, (unsigned)(H * W), 0);                          // d: (.., 2D, H*W)        -> (.., D, H*W)
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_channel_slice_copy(const float* src, float* dst, int B, int C, long V, int Ctot, int c_off, int gather,
                                     void* stream) {
  HP_REQUIRE(src && dst && V % 4 == 0 && c_off >= 0 && c_off + C <= Ctot, "hp_channel_slice_copy: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("channel_slice_copy", st);
  hipLaunchKernelGGL(k_channel_slice_copy, dim3(ugrid((long)B * C * V / 4)), dim3(UT), 0, st, src, dst, B, C, V, Ctot, c_off,
                     gather);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_conv1x1_forward(const float* x, const float* w, const float* bias, float* y, int B, int cin, int cout,
                                  long V, void* stream) {
  return hp_conv1x1_forward_sum(x, w, bias, nullptr, y, nullptr, B, cin, cout, V, stream);
}

extern "C" int hp_conv1x1_forward_sum(const float* x, const float* w, const float* bias, const float* addend, float* y,
                                      float* sum_out, int B, int cin, int cout, long V, void* stream) {
  HP_REQUIRE(x && w && y && cin <= 8 && cout <= 8, "hp_conv1x1_forward: at most 8 channels");
  HP_REQUIRE((addend == nullptr) == (sum_out == nullptr), "hp_conv1x1_forward_sum: addend and sum_out go together");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("conv1x1_fwd", st);
  hipLaunchKernelGGL(k_conv1_fwd, dim3(ugrid((long)B * V)), dim3(UT), 0, st, x, w, bias, y, B, cin, cout, V, addend, sum_out);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_conv1x1_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int B,
                                   int cin, int cout, long V, void* stream) {
  return hp_conv1x1_backward_sum(x, w, dy, nullptr, dx, dw, db, B, cin, cout, V, stream);
}

extern "C" int hp_conv1x1_backward_sum(const float* x, const float* w, const float* dy, const float* dy2, float* dx, float* dw,
                                       float* db, int B, int cin, int cout, long V, void* stream) {
  HP_REQUIRE(x && w && dy && dx && dw, "hp_conv1x1_backward: null argument");
  if (!(cin == 4 && cout == 1)) {
    set_error("hp_conv1x1_backward: only the 4 -> 1 output convolution of UNet3d(1,4) is built (got %d -> %d)", cin, cout);
    return HP_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  HP_CHECK_HIP(hipMemsetAsync(dw, 0, sizeof(float) * cin * cout, st));
  if (db) HP_CHECK_HIP(hipMemsetAsync(db, 0, sizeof(float) * cout, st));
  HP_PROF("conv1x1_bwd", st);
  hipLaunchKernelGGL((k_conv1_bwd<4, 1>), dim3(std::min(ugrid((long)B * V), 1024u)), dim3(UT), 0, st, x, w, dy, dx, dw, db, B, V, dy2);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
