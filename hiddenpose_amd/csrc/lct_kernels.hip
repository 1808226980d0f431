// LCT physics layer on gfx950: plan + forward/backward.
//
// Reference operator (models/feature_propagation.py:186-257):
//   y = mtx^T . crop( Re F3^-1( invpsf . F3( pad( mtx . (g^4 . x) ) ) ) )
// with F3 a (2T,2N,2N) FFT of the (T,N,N) volume zero-padded at the high end.
//
// MI355X formulation (HBM-bound, everything fp32):
//  * two real volumes of a batch are packed as ONE complex volume (re = sample
//    2p, im = sample 2p+1).  invpsf is Hermitian, so the padded Wiener filter is a
//    real-linear operator and the two results come back as Re / Im.  No
//    real-to-complex symmetry code, no wasted half spectrum.
//  * a zero-padded 2L-point DFT splits into two L-point DFTs (even bins: the
//    data itself; odd bins: the data times e^{-i pi n/L}); symmetrically the first
//    L outputs of the inverse are the sum of two L-point inverse DFTs.  Every pass
//    therefore works on L-point transforms and the padding is never stored.
//  * forward passes are decimation-in-frequency (natural in, digit-reversed
//    out), inverse passes decimation-in-time (digit-reversed in, natural out), so
//    no reordering pass exists: invpsf is stored once in the matching permuted
//    order, pre-scaled by 1/(8 T N N).
//  * pass order T -> H -> [W . filter . W^-1] -> H^-1 -> T^-1.  The t-resampling
//    band operators (and g^4) are fused into the first and last pass, the filter
//    into the middle one; the heaviest pass (filter) streams contiguous rows.
//  * backward = the same pipeline with conj(invpsf) and the two band operators
//    swapped (adjoint; mtxi = mtx^T).
//  * a LONE volume (batch 1, or the last volume of an odd batch) has no partner to share a complex
//    volume with.  It takes the Hermitian route instead: the T-axis spectrum of real data satisfies
//    X(2T - f) = conj X(f), so only the T + 1 planes f = 0..T are kept ("compact rows") and every later
//    pass -- H, W . filter . W^-1, H^-1 -- runs on half the rows and reads half of invpsf: 136 V bytes
//    instead of the 272 V a half-empty pair would move.  Inside the T passes two adjacent real columns
//    (w, w+1) ride through one complex FFT and are untangled / re-entangled in LDS
//    (E_a = (Z(k) + conj Z(-k))/2, E_b = (Z(k) - conj Z(-k))/2i), so the T pass costs the FFT work of a
//    pair pass and its global accesses are 64-byte (real) / 128-byte (complex) row segments.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <thread>
#include <vector>

#include "hp_internal.h"
#include "lct_host.h"

namespace hp {

constexpr int NT = 256;  // threads per workgroup (4 waves of 64)

struct BandDev {
  int32_t* off = nullptr;
  int32_t* idx = nullptr;
  float* val = nullptr;
};

}  // namespace hp

struct hp_lct_plan {
  int T = 0, N = 0, material = 0, device = 0;
  hp::LctHost host;
  std::vector<int> permT, permN;  // array position along a doubled axis -> natural frequency
  float2* Hdev = nullptr;         // [2T][2N][2N] permuted, scaled
  float2 *twT = nullptr, *hsT = nullptr, *twN = nullptr, *hsN = nullptr;
  hp::BandDev fwd_in, fwd_out, bwd_in, bwd_out;
  // lone-volume (Hermitian) path: natural T-frequency k -> position after fft_dif<T>, and compact row -> row of Hdev
  uint16_t* posT = nullptr;
  int32_t* crow = nullptr;
  bool force_lone = false;  // every volume on the lone route (long T: its T passes move 64/128-byte row segments)
};

namespace hp {

// ---------------------------------------------------------------- device FFT
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {  // a * conj(b)
  return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// L-point transforms along dim 0 of an LDS tile s[L][LD] for columns 0..WT-1.
// tw[p] = exp(-2 pi i p / L).  Forward: DIF, natural in -> digit-reversed out.
template <int L, int WT, int LD>
__device__ __forceinline__ void fft_dif(float2* s, const float2* tw, int tid) {
#pragma unroll
  for (int n = L; n >= 4; n >>= 2) {
    const int m = n >> 2;
    const int tstep = L / n;
    for (int b = tid; b < (L / 4) * WT; b += NT) {
      const int col = b % WT, bf = b / WT;
      const int j = bf % m, blk = bf / m;
      float2* p = s + (blk * n + j) * LD + col;
      float2 a0 = p[0], a1 = p[m * LD], a2 = p[2 * m * LD], a3 = p[3 * m * LD];
      float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
      float2 t3 = make_float2(d.y, -d.x);  // -i * (a1 - a3)
      float2 y0 = cadd(t0, t2), y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
      if (m > 1) {
        y1 = cmul(y1, tw[tstep * j]);
        y2 = cmul(y2, tw[tstep * 2 * j]);
        y3 = cmul(y3, tw[tstep * 3 * j]);
      }
      p[0] = y0;
      p[m * LD] = y1;
      p[2 * m * LD] = y2;
      p[3 * m * LD] = y3;
    }
    __syncthreads();
  }
  if constexpr ((L & 0x55555555) == 0) {  // odd log2: final radix-2 stage on adjacent pairs
    for (int b = tid; b < (L / 2) * WT; b += NT) {
      const int col = b % WT, blk = b / WT;
      float2* p = s + (blk * 2) * LD + col;
      float2 a0 = p[0], a1 = p[LD];
      p[0] = cadd(a0, a1);
      p[LD] = csub(a0, a1);
    }
    __syncthreads();
  }
}

// Inverse (unnormalised): DIT, digit-reversed in -> natural out; exact mirror of fft_dif.
template <int L, int WT, int LD>
__device__ __forceinline__ void fft_dit(float2* s, const float2* tw, int tid) {
  if constexpr ((L & 0x55555555) == 0) {
    for (int b = tid; b < (L / 2) * WT; b += NT) {
      const int col = b % WT, blk = b / WT;
      float2* p = s + (blk * 2) * LD + col;
      float2 a0 = p[0], a1 = p[LD];
      p[0] = cadd(a0, a1);
      p[LD] = csub(a0, a1);
    }
    __syncthreads();
  }
  constexpr int n0 = ((L & 0x55555555) == 0) ? 8 : 4;
#pragma unroll
  for (int n = n0; n <= L; n <<= 2) {
    const int m = n >> 2;
    const int tstep = L / n;
    for (int b = tid; b < (L / 4) * WT; b += NT) {
      const int col = b % WT, bf = b / WT;
      const int j = bf % m, blk = bf / m;
      float2* p = s + (blk * n + j) * LD + col;
      float2 a0 = p[0], a1 = p[m * LD], a2 = p[2 * m * LD], a3 = p[3 * m * LD];
      if (m > 1) {
        a1 = cmulc(a1, tw[tstep * j]);
        a2 = cmulc(a2, tw[tstep * 2 * j]);
        a3 = cmulc(a3, tw[tstep * 3 * j]);
      }
      float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
      float2 t3 = make_float2(-d.y, d.x);  // +i * (a1 - a3)
      p[0] = cadd(t0, t2);
      p[m * LD] = cadd(t1, t3);
      p[2 * m * LD] = csub(t0, t2);
      p[3 * m * LD] = csub(t1, t3);
    }
    __syncthreads();
  }
}

struct PassGeom {
  long in_pair_stride, out_pair_stride;    // elements per packed pair
  int chunks_per_outer;                    // tiles per outer index
  long in_outer_stride, out_outer_stride;  // elements
  long in_axis_stride, out_axis_stride;    // elements between consecutive axis samples
};

// Forward pass along one axis: L samples in, 2L (parity q, position) out.
// REAL_IN: input is the real batch (B,T,N,N); samples 2p / 2p+1 become re / im and the
// band operator (t-resampling, fused g^k scaling) is applied on the way in.
// Band operator (the sqrt-time resampling, a sparse row-compressed L x L matrix with ~2 entries per row, <= 32 in a few rows
// near t = 0) applied to the L x WT tile in LDS, for L >= NT: a thread owns whole rows n = tid + NT j (all WT complex columns),
// so a row's table entries are fetched ONCE (not once per column), the offsets of all its rows first, then the first two
// entries of every row speculatively (clamped index, zero weight past the row's end): two dependent global round trips per
// thread instead of one per entry and element; rows with more than two entries finish in a short loop.  The LDS reads are
// whole rows (WT x 8 bytes contiguous).
template <int L, int WT>
__device__ __forceinline__ void band_rows(const float2* __restrict__ s, const int32_t* __restrict__ boff,
                                          const int32_t* __restrict__ bidx, const float* __restrict__ bval, int tid,
                                          float2 (&u)[(L / NT) * WT]) {
  constexpr int RPT = L / NT;
  const int nnz = boff[L];
  int t0[RPT], t1[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    t0[j] = boff[tid + NT * j];
    t1[j] = boff[tid + NT * j + 1];
  }
  int bi[RPT][2];
  float bv[RPT][2];
#pragma unroll
  for (int j = 0; j < RPT; ++j)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int t = min(t0[j] + k, nnz - 1);
      bi[j][k] = bidx[t];
      bv[j][k] = t0[j] + k < t1[j] ? bval[t] : 0.f;
    }
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const float2* r0 = s + bi[j][0] * WT;
    const float2* r1 = s + bi[j][1] * WT;
#pragma unroll
    for (int c = 0; c < WT; ++c) {
      const float2 a = r0[c], b = r1[c];
      u[j * WT + c] = make_float2(bv[j][0] * a.x + bv[j][1] * b.x, bv[j][0] * a.y + bv[j][1] * b.y);
    }
    for (int t = t0[j] + 2; t < t1[j]; ++t) {   // the few long rows
      const float cf = bval[t];
      const float2* rr = s + bidx[t] * WT;
#pragma unroll
      for (int c = 0; c < WT; ++c) {
        const float2 v = rr[c];
        u[j * WT + c].x += cf * v.x;
        u[j * WT + c].y += cf * v.y;
      }
    }
  }
}

template <int L, int WT, bool REAL_IN>
__global__ __launch_bounds__(NT) void k_axis_fwd(const float* __restrict__ xr, const float2* __restrict__ in,
                                                 float2* __restrict__ out, int batch, long vol, PassGeom g,
                                                 const float2* __restrict__ tw, const float2* __restrict__ hs,
                                                 const int32_t* __restrict__ boff, const int32_t* __restrict__ bidx,
                                                 const float* __restrict__ bval) {
  constexpr int EPT = (L * WT) / NT;
  static_assert(EPT >= 1 && (L * WT) % NT == 0, "tile must cover the workgroup");
  __shared__ float2 s[L * WT];
  __shared__ float2 stw[L];
  __shared__ float2 shs[L];
  const int tid = threadIdx.x;
  const int pair = blockIdx.z;
  const int outer = blockIdx.x / g.chunks_per_outer, chunk = blockIdx.x % g.chunks_per_outer;
  const long ibase = (long)outer * g.in_outer_stride + (long)chunk * WT;
  const long obase = (long)pair * g.out_pair_stride + (long)outer * g.out_outer_stride + (long)chunk * WT;
  for (int i = tid; i < L; i += NT) {
    stw[i] = tw[i];
    shs[i] = hs[i];
  }
  float2 u[EPT];
  if constexpr (REAL_IN) {
    const float* x0 = xr + (long)(2 * pair) * vol;
    const bool has1 = (2 * pair + 1) < batch;
    const float* x1 = xr + (long)(2 * pair + 1) * vol;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT, col = e % WT;
      const long gi = ibase + (long)n * g.in_axis_stride + col;
      s[e] = make_float2(x0[gi], has1 ? x1[gi] : 0.0f);
    }
    __syncthreads();
    if constexpr (L >= NT) {
      band_rows<L, WT>(s, boff, bidx, bval, tid, u);   // u[j * WT + c] = element (row tid + NT j, column c)
    } else {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * NT, n = e / WT, col = e % WT;
        float2 acc = make_float2(0.f, 0.f);
        for (int t = boff[n]; t < boff[n + 1]; ++t) {
          const float c = bval[t];
          const float2 v = s[bidx[t] * WT + col];
          acc.x += c * v.x;
          acc.y += c * v.y;
        }
        u[k] = acc;
      }
    }
  } else {
    const float2* ip = in + (long)pair * g.in_pair_stride;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT, col = e % WT;
      u[k] = ip[ibase + (long)n * g.in_axis_stride + col];
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    __syncthreads();
    if constexpr (REAL_IN && L >= NT) {   // row-owning element order of band_rows
#pragma unroll
      for (int j = 0; j < L / NT; ++j) {
        const int n = tid + NT * j;
        const float2 h = shs[n];
#pragma unroll
        for (int c = 0; c < WT; ++c) s[n * WT + c] = q ? cmul(u[j * WT + c], h) : u[j * WT + c];
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * NT, n = e / WT;
        s[e] = q ? cmul(u[k], shs[n]) : u[k];
      }
    }
    __syncthreads();
    fft_dif<L, WT, WT>(s, stw, tid);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT, col = e % WT;
      out[obase + (long)(q * L + n) * g.out_axis_stride + col] = s[e];
    }
  }
}

// Inverse pass along one axis: 2L (parity, position) in, first L samples out.
// REAL_OUT: band operator on the way out, re -> sample 2p, im -> sample 2p+1.
template <int L, int WT, bool REAL_OUT>
__global__ __launch_bounds__(NT) void k_axis_inv(const float2* __restrict__ in, float2* __restrict__ out,
                                                 float* __restrict__ yr, int batch, long vol, PassGeom g,
                                                 const float2* __restrict__ tw, const float2* __restrict__ hs,
                                                 const int32_t* __restrict__ boff, const int32_t* __restrict__ bidx,
                                                 const float* __restrict__ bval) {
  constexpr int EPT = (L * WT) / NT;
  __shared__ float2 s[L * WT];
  __shared__ float2 stw[L];
  __shared__ float2 shs[L];
  const int tid = threadIdx.x;
  const int pair = blockIdx.z;
  const int outer = blockIdx.x / g.chunks_per_outer, chunk = blockIdx.x % g.chunks_per_outer;
  const long ibase = (long)pair * g.in_pair_stride + (long)outer * g.in_outer_stride + (long)chunk * WT;
  const long obase = (long)outer * g.out_outer_stride + (long)chunk * WT;
  for (int i = tid; i < L; i += NT) {
    stw[i] = tw[i];
    shs[i] = hs[i];
  }
  float2 acc[EPT];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT, col = e % WT;
      s[e] = in[ibase + (long)(q * L + n) * g.in_axis_stride + col];
    }
    __syncthreads();
    fft_dit<L, WT, WT>(s, stw, tid);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT;
      acc[k] = q ? cadd(acc[k], cmulc(s[e], shs[n])) : s[e];
    }
  }
  if constexpr (REAL_OUT) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) s[tid + k * NT] = acc[k];
    __syncthreads();
    float* y0 = yr + (long)(2 * pair) * vol;
    const bool has1 = (2 * pair + 1) < batch;
    float* y1 = yr + (long)(2 * pair + 1) * vol;
    if constexpr (L >= NT) {   // whole rows per thread (band_rows); WT consecutive floats per volume and row
      float2 o[(L / NT) * WT];
      band_rows<L, WT>(s, boff, bidx, bval, tid, o);
#pragma unroll
      for (int j = 0; j < L / NT; ++j) {
        const long gi = obase + (long)(tid + NT * j) * g.out_axis_stride;
#pragma unroll
        for (int c = 0; c < WT; c += 4) {
          *reinterpret_cast<float4*>(y0 + gi + c) = make_float4(o[j * WT + c].x, o[j * WT + c + 1].x, o[j * WT + c + 2].x, o[j * WT + c + 3].x);
          if (has1)
            *reinterpret_cast<float4*>(y1 + gi + c) = make_float4(o[j * WT + c].y, o[j * WT + c + 1].y, o[j * WT + c + 2].y, o[j * WT + c + 3].y);
        }
      }
      return;
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT, col = e % WT;
      float2 a = make_float2(0.f, 0.f);
      for (int t = boff[n]; t < boff[n + 1]; ++t) {
        const float c = bval[t];
        const float2 v = s[bidx[t] * WT + col];
        a.x += c * v.x;
        a.y += c * v.y;
      }
      const long gi = obase + (long)n * g.out_axis_stride + col;
      y0[gi] = a.x;
      if (has1) y1[gi] = a.y;
    }
  } else {
    float2* op = out + (long)pair * g.out_pair_stride;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT, col = e % WT;
      op[obase + (long)n * g.out_axis_stride + col] = acc[k];
    }
  }
}


// twiddles of the L-point transform read from the half-step table hs[p] = exp(-i pi p / L) (one LDS table instead
// of two: the 1024-point lone-volume tile leaves room for exactly one)
struct TwFromHalf {
  const float2* hs;
  int L;
  __device__ __forceinline__ float2 operator()(int p) const {
    const int q = 2 * p;
    if (q < L) return hs[q];
    const float2 v = hs[q - L];
    return make_float2(-v.x, -v.y);
  }
};

template <int L, int WT, int LD, typename TW>
__device__ __forceinline__ void fft_dif_f(float2* s, TW tw, int tid) {
#pragma unroll
  for (int n = L; n >= 4; n >>= 2) {
    const int m = n >> 2;
    const int tstep = L / n;
    for (int b = tid; b < (L / 4) * WT; b += NT) {
      const int col = b % WT, bf = b / WT;
      const int j = bf % m, blk = bf / m;
      float2* p = s + (blk * n + j) * LD + col;
      float2 a0 = p[0], a1 = p[m * LD], a2 = p[2 * m * LD], a3 = p[3 * m * LD];
      float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
      float2 t3 = make_float2(d.y, -d.x);
      float2 y0 = cadd(t0, t2), y1 = cadd(t1, t3), y2 = csub(t0, t2), y3 = csub(t1, t3);
      if (m > 1) {
        y1 = cmul(y1, tw(tstep * j));
        y2 = cmul(y2, tw(tstep * 2 * j));
        y3 = cmul(y3, tw(tstep * 3 * j));
      }
      p[0] = y0;
      p[m * LD] = y1;
      p[2 * m * LD] = y2;
      p[3 * m * LD] = y3;
    }
    __syncthreads();
  }
  if constexpr ((L & 0x55555555) == 0) {
    for (int b = tid; b < (L / 2) * WT; b += NT) {
      const int col = b % WT, blk = b / WT;
      float2* p = s + (blk * 2) * LD + col;
      float2 a0 = p[0], a1 = p[LD];
      p[0] = cadd(a0, a1);
      p[LD] = csub(a0, a1);
    }
    __syncthreads();
  }
}

template <int L, int WT, int LD, typename TW>
__device__ __forceinline__ void fft_dit_f(float2* s, TW tw, int tid) {
  if constexpr ((L & 0x55555555) == 0) {
    for (int b = tid; b < (L / 2) * WT; b += NT) {
      const int col = b % WT, blk = b / WT;
      float2* p = s + (blk * 2) * LD + col;
      float2 a0 = p[0], a1 = p[LD];
      p[0] = cadd(a0, a1);
      p[LD] = csub(a0, a1);
    }
    __syncthreads();
  }
  constexpr int n0 = ((L & 0x55555555) == 0) ? 8 : 4;
#pragma unroll
  for (int n = n0; n <= L; n <<= 2) {
    const int m = n >> 2;
    const int tstep = L / n;
    for (int b = tid; b < (L / 4) * WT; b += NT) {
      const int col = b % WT, bf = b / WT;
      const int j = bf % m, blk = bf / m;
      float2* p = s + (blk * n + j) * LD + col;
      float2 a0 = p[0], a1 = p[m * LD], a2 = p[2 * m * LD], a3 = p[3 * m * LD];
      if (m > 1) {
        a1 = cmulc(a1, tw(tstep * j));
        a2 = cmulc(a2, tw(tstep * 2 * j));
        a3 = cmulc(a3, tw(tstep * 3 * j));
      }
      float2 t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), d = csub(a1, a3);
      float2 t3 = make_float2(-d.y, d.x);
      p[0] = cadd(t0, t2);
      p[m * LD] = cadd(t1, t3);
      p[2 * m * LD] = csub(t0, t2);
      p[3 * m * LD] = csub(t1, t3);
    }
    __syncthreads();
  }
}

// Lone-volume T passes.  Tile: L time samples x WT complex columns = 2 WT adjacent real columns (w even in re, w+1
// in im).  Compact rows: c = k (even bins f = 2k, k = 0..L/2), then c = L/2 + 1 + k (odd bins f = 2k+1, k < L/2).
template <int L, int WT>
__global__ __launch_bounds__(NT) void k_axis_fwd_t_lone(const float* __restrict__ x, float2* __restrict__ out, long plane,
                                                       const float2* __restrict__ hs, const uint16_t* __restrict__ pos,
                                                       const int32_t* __restrict__ boff, const int32_t* __restrict__ bidx,
                                                       const float* __restrict__ bval) {
  constexpr int EPT = (L * WT) / NT;
  static_assert(EPT >= 1 && (L * WT) % NT == 0, "tile must cover the workgroup");
  __shared__ float2 s[L * WT];
  __shared__ float2 shs[L];
  __shared__ uint16_t spos[L];
  const int tid = threadIdx.x;
  const long col0 = (long)blockIdx.x * (2 * WT);  // first real column (flattened h*N + w) of this tile
  for (int i = tid; i < L; i += NT) {
    shs[i] = hs[i];
    spos[i] = pos[i];
  }
  float2 u[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid + k * NT, n = e / WT, col = e % WT;
    s[e] = *reinterpret_cast<const float2*>(x + (long)n * plane + col0 + 2 * col);
  }
  __syncthreads();
  constexpr bool ROWS = L >= NT;   // a thread owns whole rows (band_rows); shorter transforms keep one element per (row, column)
  if constexpr (ROWS) {
    band_rows<L, WT>(s, boff, bidx, bval, tid, u);
  } else {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT, col = e % WT;
      float2 acc = make_float2(0.f, 0.f);
      for (int t = boff[n]; t < boff[n + 1]; ++t) {
        const float c = bval[t];
        const float2 v = s[bidx[t] * WT + col];
        acc.x += c * v.x;
        acc.y += c * v.y;
      }
      u[k] = acc;
    }
  }
  const TwFromHalf tw{shs, L};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    __syncthreads();
    if constexpr (ROWS) {
#pragma unroll
      for (int j = 0; j < L / NT; ++j) {
        const int n = tid + NT * j;
        const float2 h = shs[n];
#pragma unroll
        for (int c = 0; c < WT; ++c) s[n * WT + c] = q ? cmul(u[j * WT + c], h) : u[j * WT + c];
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int e = tid + k * NT, n = e / WT;
        s[e] = q ? cmul(u[k], shs[n]) : u[k];
      }
    }
    __syncthreads();
    fft_dif_f<L, WT, WT>(s, tw, tid);
    const int nk = q ? L / 2 : L / 2 + 1;
    const int c0 = q ? L / 2 + 1 : 0;
    for (int i = tid; i < nk * WT; i += NT) {
      const int k = i / WT, col = i % WT;
      const int kp = q ? (L - 1 - k) : ((L - k) & (L - 1));
      const float2 za = s[spos[k] * WT + col], zb = s[spos[kp] * WT + col];
      // column a = (Z(k) + conj Z(k'))/2, column b = (Z(k) - conj Z(k'))/(2i)
      const float4 o = make_float4(0.5f * (za.x + zb.x), 0.5f * (za.y - zb.y), 0.5f * (za.y + zb.y), 0.5f * (zb.x - za.x));
      *reinterpret_cast<float4*>(out + (long)(c0 + k) * plane + col0 + 2 * col) = o;
    }
  }
}

template <int L, int WT>
__global__ __launch_bounds__(NT) void k_axis_inv_t_lone(const float2* __restrict__ in, float* __restrict__ y, long plane,
                                                       const float2* __restrict__ hs, const uint16_t* __restrict__ pos,
                                                       const int32_t* __restrict__ boff, const int32_t* __restrict__ bidx,
                                                       const float* __restrict__ bval) {
  constexpr int EPT = (L * WT) / NT;
  __shared__ float2 s[L * WT];
  __shared__ float2 shs[L];
  __shared__ uint16_t spos[L];
  const int tid = threadIdx.x;
  const long col0 = (long)blockIdx.x * (2 * WT);
  for (int i = tid; i < L; i += NT) {
    shs[i] = hs[i];
    spos[i] = pos[i];
  }
  const TwFromHalf tw{shs, L};
  float2 acc[EPT];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    __syncthreads();
    const int nk = q ? L / 2 : L / 2 + 1;
    const int c0 = q ? L / 2 + 1 : 0;
    for (int i = tid; i < nk * WT; i += NT) {
      const int k = i / WT, col = i % WT;
      const int kp = q ? (L - 1 - k) : ((L - k) & (L - 1));
      const float4 v = *reinterpret_cast<const float4*>(in + (long)(c0 + k) * plane + col0 + 2 * col);
      // Z(k) = A + i B ;  Z(k') = conj A + i conj B  (A, B: spectra of the two real output columns)
      if (kp == k) {
        s[spos[k] * WT + col] = make_float2(v.x, v.z);  // self-conjugate bins are real: keep the real parts
      } else {
        s[spos[k] * WT + col] = make_float2(v.x - v.w, v.y + v.z);
        s[spos[kp] * WT + col] = make_float2(v.x + v.w, v.z - v.y);
      }
    }
    __syncthreads();
    fft_dit_f<L, WT, WT>(s, tw, tid);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, n = e / WT;
      acc[k] = q ? cadd(acc[k], cmulc(s[e], shs[n])) : s[e];
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < EPT; ++k) s[tid + k * NT] = acc[k];
  __syncthreads();
  if constexpr (L >= NT) {
    float2 o[(L / NT) * WT];
    band_rows<L, WT>(s, boff, bidx, bval, tid, o);
#pragma unroll
    for (int j = 0; j < L / NT; ++j) {
      float* dst = y + (long)(tid + NT * j) * plane + col0;   // 2 WT adjacent real columns of row n
#pragma unroll
      for (int c = 0; c < WT; c += 2)
        *reinterpret_cast<float4*>(dst + 2 * c) = make_float4(o[j * WT + c].x, o[j * WT + c].y, o[j * WT + c + 1].x, o[j * WT + c + 1].y);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid + k * NT, n = e / WT, col = e % WT;
    float2 a = make_float2(0.f, 0.f);
    for (int t = boff[n]; t < boff[n + 1]; ++t) {
      const float c = bval[t];
      const float2 v = s[bidx[t] * WT + col];
      a.x += c * v.x;
      a.y += c * v.y;
    }
    *reinterpret_cast<float2*>(y + (long)n * plane + col0 + 2 * col) = a;
  }
}

// Middle pass on contiguous rows of length L (the W axis), in place:
//   row <- crop( F^-1( H . F( pad(row) ) ) ),  H read once, coalesced.
template <int L, int RT>
__global__ __launch_bounds__(NT) void k_axis_mid(float2* __restrict__ data, const float2* __restrict__ H,
                                                 long pair_stride, long rows_per_pair, int conj_h,
                                                 const float2* __restrict__ tw, const float2* __restrict__ hs,
                                                 const int32_t* __restrict__ crow, int rows_per_plane) {
  constexpr int EPT = (L * RT) / NT;
  constexpr int LD = RT + 1;
  __shared__ float2 s[L * LD];
  __shared__ float2 stw[L];
  __shared__ float2 shs[L];
  const int tid = threadIdx.x;
  const int pair = blockIdx.z;
  const long row0 = (long)blockIdx.x * RT;
  float2* base = data + (long)pair * pair_stride + row0 * L;
  // lone-volume path: data plane c (compact row) uses plane crow[c] of H; a tile never straddles two planes
  const long hrow0 = crow ? (long)crow[row0 / rows_per_plane] * rows_per_plane + row0 % rows_per_plane : row0;
  const float2* hbase = H + hrow0 * (2 * L);
  for (int i = tid; i < L; i += NT) {
    stw[i] = tw[i];
    shs[i] = hs[i];
  }
  float2 u[EPT], acc[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) u[k] = base[tid + k * NT];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, r = e / L, n = e % L;
      s[n * LD + r] = q ? cmul(u[k], shs[n]) : u[k];
    }
    __syncthreads();
    fft_dif<L, RT, LD>(s, stw, tid);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, r = e / L, n = e % L;
      const float2 h = hbase[(long)r * (2 * L) + q * L + n];
      const float2 v = s[n * LD + r];
      s[n * LD + r] = conj_h ? cmulc(v, h) : cmul(v, h);
    }
    __syncthreads();
    fft_dit<L, RT, LD>(s, stw, tid);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int e = tid + k * NT, r = e / L, n = e % L;
      const float2 v = s[n * LD + r];
      acc[k] = q ? cadd(acc[k], cmulc(v, shs[n])) : v;
    }
  }
  (void)rows_per_pair;
#pragma unroll
  for (int k = 0; k < EPT; ++k) base[tid + k * NT] = acc[k];
}

// ------------------------------------------------------------------ host side
static constexpr int tile_width(int L) { return L <= 256 ? 16 : (L == 512 ? 8 : 4); }

// position (after the DIF stages of fft_dif<L>) -> natural frequency of the L-point DFT
static std::vector<int> dif_position_to_freq(int L) {
  std::vector<int> radices;
  int n = L;
  while (n >= 4) {
    radices.push_back(4);
    n >>= 2;
  }
  if (n == 2) radices.push_back(2);
  std::vector<int> k(L);
  for (int pos = 0; pos < L; ++pos) {
    int rem = pos, span = L, mult = 1, freq = 0;
    for (int R : radices) {
      span /= R;
      int digit = rem / span;
      rem %= span;
      freq += digit * mult;
      mult *= R;
    }
    k[pos] = freq;
  }
  return k;
}

// array position along a doubled axis (parity q, position) -> natural bin of the 2L-point DFT
static std::vector<int> doubled_axis_perm(int L) {
  std::vector<int> k = dif_position_to_freq(L), p(2 * L);
  for (int q = 0; q < 2; ++q)
    for (int pos = 0; pos < L; ++pos) p[q * L + pos] = 2 * k[pos] + q;
  return p;
}

template <typename T>
static int upload(T** dst, const std::vector<T>& v) {
  HP_CHECK_HIP(hipMalloc((void**)dst, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) HP_CHECK_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return HP_OK;
}

static int upload_band(BandDev& b, const SparseRows& m) {
  int rc;
  if ((rc = upload(&b.off, m.off))) return rc;
  if ((rc = upload(&b.idx, m.idx))) return rc;
  return upload(&b.val, m.val);
}

static void free_band(BandDev& b) {
  if (b.off) (void)hipFree(b.off);
  if (b.idx) (void)hipFree(b.idx);
  if (b.val) (void)hipFree(b.val);
  b = BandDev();
}

static int upload_twiddles(int L, float2** tw, float2** hs) {
  std::vector<float2> a(L), b(L);
  const double PI = 3.14159265358979323846;
  for (int p = 0; p < L; ++p) {
    a[p] = make_float2((float)cos(-2.0 * PI * p / L), (float)sin(-2.0 * PI * p / L));
    b[p] = make_float2((float)cos(-PI * p / L), (float)sin(-PI * p / L));
  }
  int rc;
  if ((rc = upload(tw, a))) return rc;
  return upload(hs, b);
}

static bool supported_len(int L) { return is_pow2(L) && L >= 16 && L <= 1024; }

// Plan management runs on the plan's device but leaves the caller's current device as it found it (a finalizer may
// destroy a plan at any point of a multi-device process).
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != dev) (void)hipSetDevice(dev);
    else prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

template <int L, bool REAL_IN>
static void launch_fwd(dim3 grid, hipStream_t st, const float* xr, const float2* in, float2* out, int batch, long vol,
                       PassGeom g, const float2* tw, const float2* hs, const BandDev& b) {
  constexpr int WT = tile_width(L);
  HP_PROF(REAL_IN ? "lct_axis_fwd_t" : "lct_axis_fwd_h", st);
  hipLaunchKernelGGL((k_axis_fwd<L, WT, REAL_IN>), grid, dim3(NT), 0, st, xr, in, out, batch, vol, g, tw, hs, b.off,
                     b.idx, b.val);
}
template <int L, bool REAL_OUT>
static void launch_inv(dim3 grid, hipStream_t st, const float2* in, float2* out, float* yr, int batch, long vol,
                       PassGeom g, const float2* tw, const float2* hs, const BandDev& b) {
  constexpr int WT = tile_width(L);
  HP_PROF(REAL_OUT ? "lct_axis_inv_t" : "lct_axis_inv_h", st);
  hipLaunchKernelGGL((k_axis_inv<L, WT, REAL_OUT>), grid, dim3(NT), 0, st, in, out, yr, batch, vol, g, tw, hs, b.off,
                     b.idx, b.val);
}
template <int L>
static void launch_mid(dim3 grid, hipStream_t st, float2* data, const float2* H, long pair_stride, long rows,
                       int conj_h, const float2* tw, const float2* hs, const int32_t* crow) {
  constexpr int RT = tile_width(L);
  static_assert((2 * L) % RT == 0, "a row tile must stay inside one (2N)-row plane");
  HP_PROF("lct_axis_mid_w", st);
  hipLaunchKernelGGL((k_axis_mid<L, RT>), grid, dim3(NT), 0, st, data, H, pair_stride, rows, conj_h, tw, hs, crow, 2 * L);
}

static constexpr int lone_tile_width(int L) { return L <= 512 ? 16 : 8; }  // complex columns: 64 KB of LDS at most

template <int L>
static void launch_fwd_t_lone(hipStream_t st, const float* x, float2* out, long plane, const hp_lct_plan* p, const BandDev& b) {
  constexpr int WT = lone_tile_width(L);
  HP_PROF("lct_lone_fwd_t", st);
  hipLaunchKernelGGL((k_axis_fwd_t_lone<L, WT>), dim3((unsigned)(plane / (2 * WT))), dim3(NT), 0, st, x, out, plane, p->hsT,
                     p->posT, b.off, b.idx, b.val);
}
template <int L>
static void launch_inv_t_lone(hipStream_t st, const float2* in, float* y, long plane, const hp_lct_plan* p, const BandDev& b) {
  constexpr int WT = lone_tile_width(L);
  HP_PROF("lct_lone_inv_t", st);
  hipLaunchKernelGGL((k_axis_inv_t_lone<L, WT>), dim3((unsigned)(plane / (2 * WT))), dim3(NT), 0, st, in, y, plane, p->hsT,
                     p->posT, b.off, b.idx, b.val);
}

#define HP_DISPATCH_LEN(L, CALL)                 \
  switch (L) {                                   \
    case 16: { constexpr int LL = 16; CALL; } break;     \
    case 32: { constexpr int LL = 32; CALL; } break;     \
    case 64: { constexpr int LL = 64; CALL; } break;     \
    case 128: { constexpr int LL = 128; CALL; } break;   \
    case 256: { constexpr int LL = 256; CALL; } break;   \
    case 512: { constexpr int LL = 512; CALL; } break;   \
    case 1024: { constexpr int LL = 1024; CALL; } break; \
    default: break;                              \
  }

static size_t lone_elems(const hp_lct_plan* p) { return (size_t)3 * (p->T + 1) * p->N * p->N; }  // float2 elements

static int run_lct(const hp_lct_plan* p, const float* x, float* y, int batch, void* ws, size_t ws_bytes,
                   hipStream_t st, bool backward) {
  HP_REQUIRE(p && x && y && ws, "hp_lct: null argument");
  HP_REQUIRE(batch >= 1, "hp_lct: batch must be >= 1 (got %d)", batch);
  const size_t need = hp_lct_workspace_bytes(p, batch);
  if (ws_bytes < need) {
    set_error("hp_lct: workspace too small (%zu < %zu)", ws_bytes, need);
    return HP_ERR_WORKSPACE;
  }
  const int T = p->T, N = p->N;
  // complete pairs ride as complex volumes; a last odd volume takes the Hermitian (lone) route
  const int P = p->force_lone ? 0 : batch / 2;
  const int nlone = batch - 2 * P;
  const long vol = (long)T * N * N;
  float2* C1 = (float2*)ws;          // [P][2T][N][N]
  float2* C2 = C1 + (long)P * 2 * vol;  // [P][2T][2N][N]
  const BandDev& bin = backward ? p->bwd_in : p->fwd_in;
  const BandDev& bout = backward ? p->bwd_out : p->fwd_out;
  const int wtT = tile_width(T), wtN = tile_width(N);

  PassGeom g;
  if (P > 0) {
    const int pb = 2 * P;  // volumes on the pair route
    // 1. T forward (+ band in)
    g = PassGeom{0, 2 * vol, (int)((long)N * N / wtT), 0, 0, (long)N * N, (long)N * N};
    HP_DISPATCH_LEN(T, (launch_fwd<LL, true>(dim3(g.chunks_per_outer, 1, P), st, x, nullptr, C1, pb, vol, g, p->twT,
                                             p->hsT, bin)));
    // 2. H forward
    g = PassGeom{2 * vol, 4 * vol, N / wtN, (long)N * N, 2L * N * N, N, N};
    HP_DISPATCH_LEN(N, (launch_fwd<LL, false>(dim3(2 * T * g.chunks_per_outer, 1, P), st, nullptr, C1, C2, pb, vol, g,
                                              p->twN, p->hsN, bin)));
    // 3. W forward . filter . W inverse (in place)
    {
      const long rows = 2L * T * 2 * N;
      HP_DISPATCH_LEN(N, (launch_mid<LL>(dim3((unsigned)(rows / wtN), 1, P), st, C2, p->Hdev, 4 * vol, rows,
                                         backward ? 1 : 0, p->twN, p->hsN, nullptr)));
    }
    // 4. H inverse
    g = PassGeom{4 * vol, 2 * vol, N / wtN, 2L * N * N, (long)N * N, N, N};
    HP_DISPATCH_LEN(N, (launch_inv<LL, false>(dim3(2 * T * g.chunks_per_outer, 1, P), st, C2, C1, nullptr, pb, vol, g,
                                              p->twN, p->hsN, bout)));
    // 5. T inverse (+ band out)
    g = PassGeom{2 * vol, 0, (int)((long)N * N / wtT), 0, 0, (long)N * N, (long)N * N};
    HP_DISPATCH_LEN(T, (launch_inv<LL, true>(dim3(g.chunks_per_outer, 1, P), st, C1, nullptr, y, pb, vol, g, p->twT,
                                             p->hsT, bout)));
  }
  // lone volumes: T + 1 compact rows (Hermitian half of the T spectrum), same H / W passes on half the rows
  float2* L1 = C1 + (long)P * 6 * vol;                       // [T+1][N][N]
  float2* L2 = L1 + (long)(T + 1) * N * N;                   // [T+1][2N][N]
  const long plane = (long)N * N;
  for (int i = 0; i < nlone; ++i) {
    const float* xi = x + (long)(2 * P + i) * vol;
    float* yi = y + (long)(2 * P + i) * vol;
    HP_DISPATCH_LEN(T, (launch_fwd_t_lone<LL>(st, xi, L1, plane, p, bin)));
    g = PassGeom{0, 0, N / wtN, plane, 2 * plane, N, N};
    HP_DISPATCH_LEN(N, (launch_fwd<LL, false>(dim3((T + 1) * g.chunks_per_outer, 1, 1), st, nullptr, L1, L2, 1, vol, g,
                                              p->twN, p->hsN, bin)));
    {
      const long rows = (long)(T + 1) * 2 * N;
      HP_DISPATCH_LEN(N, (launch_mid<LL>(dim3((unsigned)(rows / wtN), 1, 1), st, L2, p->Hdev, 0, rows, backward ? 1 : 0,
                                         p->twN, p->hsN, p->crow)));
    }
    g = PassGeom{0, 0, N / wtN, 2 * plane, plane, N, N};
    HP_DISPATCH_LEN(N, (launch_inv<LL, false>(dim3((T + 1) * g.chunks_per_outer, 1, 1), st, L2, L1, nullptr, 1, vol, g,
                                              p->twN, p->hsN, bout)));
    HP_DISPATCH_LEN(T, (launch_inv_t_lone<LL>(st, L1, yi, plane, p, bout)));
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

}  // namespace hp

using namespace hp;

extern "C" int hp_lct_plan_create(hp_lct_plan** out, int T, int N, double bin_len, double wall_size, int material,
                                  int device) {
  return hp_lct_plan_create_mode(out, T, N, bin_len, wall_size, material, HP_LCT_MODE_LCT, device);
}

extern "C" int hp_lct_plan_create_mode(hp_lct_plan** out, int T, int N, double bin_len, double wall_size, int material,
                                       int mode, int device) {
  HP_REQUIRE(out, "hp_lct_plan_create: null out");
  HP_REQUIRE(mode == HP_LCT_MODE_LCT || mode == HP_LCT_MODE_BP, "hp_lct_plan_create: bad mode %d", mode);
  *out = nullptr;
  if (!supported_len(T) || !supported_len(N)) {
    set_error("hp_lct_plan_create: T and N must be powers of two in [16,1024] (got T=%d N=%d)", T, N);
    return HP_ERR_UNSUPPORTED;
  }
  HP_REQUIRE(material == HP_MATERIAL_DIFFUSE || material == HP_MATERIAL_SPECULAR, "bad material %d", material);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("hp_lct_plan_create: no HIP device available");
    return HP_ERR_NO_DEVICE;
  }
  HP_REQUIRE(device >= 0 && device < ndev, "hp_lct_plan_create: device %d out of range (%d devices)", device, ndev);
  DeviceGuard guard(device);

  hp_lct_plan* p = new hp_lct_plan();
  p->T = T;
  p->N = N;
  p->material = material;
  p->device = device;
  lct_host_build(T, N, bin_len, wall_size, p->host);
  p->permT = doubled_axis_perm(T);
  p->permN = doubled_axis_perm(N);

  // band operators. forward: in = mtx.diag(g^k), out = mtx^T ; backward: in = mtx, out = diag(g^k).mtx^T
  const int gp = material == HP_MATERIAL_DIFFUSE ? 4 : 2;
  std::vector<float> gk(T);
  for (int t = 0; t < T; ++t) {
    float g = p->host.gridz[t], v = g * g;
    gk[t] = gp == 4 ? v * v : v;
  }
  SparseRows fin = p->host.mtx;
  for (size_t e = 0; e < fin.idx.size(); ++e) fin.val[e] *= gk[fin.idx[e]];
  SparseRows mT = p->host.mtx.transposed(T);
  SparseRows bout = mT;
  for (int r = 0; r < T; ++r)
    for (int e = bout.off[r]; e < bout.off[r + 1]; ++e) bout.val[e] *= gk[r];
  int rc = HP_OK;
  auto fail = [&](int code) {
    hp_lct_plan_destroy(p);
    return code;
  };
  if ((rc = upload_band(p->fwd_in, fin))) return fail(rc);
  if ((rc = upload_band(p->fwd_out, mT))) return fail(rc);
  if ((rc = upload_band(p->bwd_in, p->host.mtx))) return fail(rc);
  if ((rc = upload_band(p->bwd_out, bout))) return fail(rc);
  if ((rc = upload_twiddles(T, &p->twT, &p->hsT))) return fail(rc);
  if ((rc = upload_twiddles(N, &p->twN, &p->hsN))) return fail(rc);
  {
    const std::vector<int> freq = dif_position_to_freq(T);  // position -> natural frequency of the T-point DFT
    std::vector<uint16_t> pos(T);
    for (int q = 0; q < T; ++q) pos[freq[q]] = (uint16_t)q;
    std::vector<int32_t> crow(T + 1);
    for (int k = 0; k <= T / 2; ++k) crow[k] = pos[k];                        // even bins f = 2k: rows [0, T) of Hdev
    for (int k = 0; k < T / 2; ++k) crow[T / 2 + 1 + k] = T + pos[k];          // odd bins  f = 2k+1: rows [T, 2T)
    if ((rc = upload(&p->posT, pos))) return fail(rc);
    if ((rc = upload(&p->crow, crow))) return fail(rc);
    const char* env = getenv("HP_LCT_FORCE_LONE");
    p->force_lone = env ? atoi(env) != 0 : false;
  }

  // inverse PSF spectrum, permuted to the DIF/DIT bin order of the device passes and
  // pre-scaled by 1/(2T.2N.2N); built slice by slice (threads over kz) in pinned-size chunks.
  const int N2 = 2 * N, M2 = 2 * T;
  const size_t sl = (size_t)N2 * N2;
  const double scale = 1.0 / ((double)M2 * N2 * N2);
  if (hipMalloc((void**)&p->Hdev, sizeof(float2) * sl * M2) != hipSuccess) {
    set_error("hp_lct_plan_create: hipMalloc of %zu bytes for invpsf failed", sizeof(float2) * sl * M2);
    return fail(HP_ERR_HIP);
  }
  {
    std::vector<float2> hostH(sl * M2);
    unsigned nth = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    const std::vector<int>& pT = p->permT;
    const std::vector<int>& pN = p->permN;
    const LctHost& hh = p->host;
    for (unsigned t = 0; t < nth; ++t)
      th.emplace_back([&, t]() {
        std::vector<std::complex<double>> buf(sl);
        for (int at = (int)t; at < M2; at += (int)nth) {
          lct_invpsf_slice(hh, pT[at], buf.data(), nullptr, mode == HP_LCT_MODE_LCT);
          float2* dst = hostH.data() + (size_t)at * sl;
          for (int ah = 0; ah < N2; ++ah) {
            const std::complex<double>* src = buf.data() + (size_t)pN[ah] * N2;
            for (int aw = 0; aw < N2; ++aw) {
              std::complex<double> v = src[pN[aw]] * scale;
              dst[(size_t)ah * N2 + aw] = make_float2((float)v.real(), (float)v.imag());
            }
          }
        }
      });
    for (auto& t : th) t.join();
    if (hipMemcpy(p->Hdev, hostH.data(), sizeof(float2) * sl * M2, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("hp_lct_plan_create: upload of invpsf failed");
      return fail(HP_ERR_HIP);
    }
  }
  *out = p;
  return HP_OK;
}

extern "C" int hp_lct_plan_destroy(hp_lct_plan* p) {
  if (!p) return HP_OK;
  DeviceGuard guard(p->device);
  if (p->Hdev) (void)hipFree(p->Hdev);
  for (float2* q : {p->twT, p->hsT, p->twN, p->hsN})
    if (q) (void)hipFree(q);
  if (p->posT) (void)hipFree(p->posT);
  if (p->crow) (void)hipFree(p->crow);
  free_band(p->fwd_in);
  free_band(p->fwd_out);
  free_band(p->bwd_in);
  free_band(p->bwd_out);
  delete p;
  return HP_OK;
}

extern "C" size_t hp_lct_workspace_bytes(const hp_lct_plan* p, int batch) {
  if (!p || batch < 1) return 0;
  const size_t P = p->force_lone ? 0 : (size_t)batch / 2;
  const size_t nlone = (size_t)batch - 2 * P;
  return (P * 6 * (size_t)p->T * p->N * p->N + (nlone ? hp::lone_elems(p) : 0)) * sizeof(float2);
}

extern "C" int hp_lct_forward(const hp_lct_plan* p, const float* x, float* y, int batch, void* ws, size_t ws_bytes,
                              void* stream) {
  return run_lct(p, x, y, batch, ws, ws_bytes, (hipStream_t)stream, false);
}

extern "C" int hp_lct_backward(const hp_lct_plan* p, const float* gy, float* gx, int batch, void* ws, size_t ws_bytes,
                               void* stream) {
  return run_lct(p, gy, gx, batch, ws, ws_bytes, (hipStream_t)stream, true);
}

extern "C" int hp_lct_plan_get_invpsf(const hp_lct_plan* p, float* re, float* im) {
  HP_REQUIRE(p && re && im, "hp_lct_plan_get_invpsf: null argument");
  const int N2 = 2 * p->N, M2 = 2 * p->T;
  const size_t sl = (size_t)N2 * N2;
  std::vector<float2> h(sl * M2);
  DeviceGuard guard(p->device);
  HP_CHECK_HIP(hipMemcpy(h.data(), p->Hdev, sizeof(float2) * sl * M2, hipMemcpyDeviceToHost));
  const double unscale = (double)M2 * N2 * N2;
  for (int at = 0; at < M2; ++at)
    for (int ah = 0; ah < N2; ++ah)
      for (int aw = 0; aw < N2; ++aw) {
        const float2 v = h[(size_t)at * sl + (size_t)ah * N2 + aw];
        const size_t o = (size_t)p->permT[at] * sl + (size_t)p->permN[ah] * N2 + p->permN[aw];
        re[o] = (float)(v.x * unscale);
        im[o] = (float)(v.y * unscale);
      }
  return HP_OK;
}

extern "C" int hp_lct_time_window(float* y_full, float* x_window, int B, int D, int tnum, int T, long plane, const int* tbes,
                                  int to_window, void* stream) {
  HP_REQUIRE(y_full && x_window && tbes && B >= 1 && D >= 1 && tnum >= 1 && tnum <= T && plane >= 1,
             "hp_lct_time_window: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const size_t row = sizeof(float) * (size_t)tnum * plane, full = sizeof(float) * (size_t)T * plane;
  if (!to_window) HP_CHECK_HIP(hipMemsetAsync(y_full, 0, full * (size_t)B * D, st));
  for (int b = 0; b < B; ++b) {
    HP_REQUIRE(tbes[b] >= 0 && tbes[b] + tnum <= T, "hp_lct_time_window: window [%d, %d) of sample %d leaves [0, %d)", tbes[b],
               tbes[b] + tnum, b, T);
    float* yf = y_full + ((size_t)b * D * T + tbes[b]) * plane;
    float* xw = x_window + (size_t)b * D * tnum * plane;
    // D rows (channels) of tnum*plane floats: pitch T*plane in the full tensor, tnum*plane in the window
    if (to_window) HP_CHECK_HIP(hipMemcpy2DAsync(xw, row, yf, full, row, (size_t)D, hipMemcpyDeviceToDevice, st));
    else HP_CHECK_HIP(hipMemcpy2DAsync(yf, full, xw, row, row, (size_t)D, hipMemcpyDeviceToDevice, st));
  }
  return HP_OK;
}
