// NlosPoseSformer inference path (models/NlosPoseSformer.py; BASELINE config 5):
// patch embedding, LayerNorm, axial-RoPE + head split, spatial attention with joint tokens,
// GEGLU.  The Linear layers run on the exact-fp32 MFMA GEMM of conv_kernels.hip (a 1x1x1
// "convolution" over channels-last rows IS a Linear).
//
// Attention is a flash-style single pass in exact-fp32 MFMA, organised so that NO cross-lane
// data movement is needed for the soft-max:
//   S^T = K Q^T  -> accumulator column (= lane) is one QUERY, registers are 32 keys: the running
//                   max / sum of a query is a reduction over the lane's own registers plus one
//                   xor-32 shuffle;
//   O^T += V^T P -> P is consumed as the B operand exactly as it lies in the accumulator
//                   (the MFMA's k order is permuted to the accumulator's row order, and V^T rows
//                   are fetched in that same order), so the lane keeps owning its query.
// Joint-token queries (24 rows against every token) split the key range over the 4 waves of
// one workgroup and merge the partial (max, sum, O) triples through LDS.
#include <algorithm>
#include <cfloat>

#include <type_traits>

#include "hp_internal.h"

namespace hp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int ST = 256;

// 'b f c (h p1) (w p2) -> (b f h w) (p1 p2 c)'
__global__ __launch_bounds__(ST) void k_patchify(const float* __restrict__ v, float* __restrict__ out, int B, int Fr,
                                                 int C, int H, int W, int ps) {
  const int hp = H / ps, wp = W / ps, pd = ps * ps * C;
  const long total = (long)B * Fr * hp * wp * pd;
  for (long i = (long)blockIdx.x * ST + threadIdx.x; i < total; i += (long)gridDim.x * ST) {
    const int e = (int)(i % pd);
    long t = i / pd;
    const int pw = (int)(t % wp);
    t /= wp;
    const int ph = (int)(t % hp);
    t /= hp;  // t = b*Fr + f
    const int c = e % C, p2 = (e / C) % ps, p1 = e / (C * ps);
    out[i] = v[((t * C + c) * H + ph * ps + p1) * W + pw * ps + p2];
  }
}

// LayerNorm over the last dim (one wave per row), eps inside the sqrt, affine.  in_row(r) lets the final
// norm pick the joint-token rows out of the token matrix: row r -> (r / rpb) * batch_stride + (r % rpb).
__global__ __launch_bounds__(ST) void k_layernorm(const float* __restrict__ x, float* __restrict__ y, long rows, int dim,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  float eps, int rpb, long batch_stride_rows) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (ST / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long src = rpb > 0 ? (row / rpb) * batch_stride_rows + (row % rpb) : row;
  const float* p = x + src * dim;
  float s = 0.f;
  for (int i = lane; i < dim; i += 64) s += p[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)dim;
  float q = 0.f;
  for (int i = lane; i < dim; i += 64) {
    const float d = p[i] - mean;
    q += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = rsqrtf(q / (float)dim + eps);
  float* o = y + row * dim;
  for (int i = lane; i < dim; i += 64) o[i] = (p[i] - mean) * rstd * gamma[i] + beta[i];
}

// u (rows, 2H) -> g (rows, H): g = u[:, :H] * gelu_erf(u[:, H:])
__global__ __launch_bounds__(ST) void k_geglu(const float* __restrict__ u, float* __restrict__ g, long rows, int Hd) {
  const long total = rows * Hd;
  for (long i = (long)blockIdx.x * ST + threadIdx.x; i < total; i += (long)gridDim.x * ST) {
    const long r = i / Hd;
    const int c = (int)(i - r * Hd);
    const float a = u[r * 2 * Hd + c], gt = u[r * 2 * Hd + Hd + c];
    g[i] = a * 0.5f * gt * (1.0f + erff(gt * 0.70710678118654752f));
  }
}

// plain GELU (erf form: nn.GELU() of models/tokenpose.py:274), rows x dim elementwise
__global__ __launch_bounds__(ST) void k_gelu(const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * ST + threadIdx.x; i < n; i += (long)gridDim.x * ST) {
    const float v = x[i];
    y[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  }
}

// qkv (B, Ntok, 3*inner) -> Q, K, V (B, heads, Ntok, dh); q scaled; axial RoPE on the patch tokens'
// q and k: t' = t * cos + rotate_every_two(t) * sin on the first rot_dim dims (tables (n, rot_dim)).
__global__ __launch_bounds__(ST) void k_qkv_prepare(const float* __restrict__ qkv, float* __restrict__ Q,
                                                    float* __restrict__ K, float* __restrict__ K0,
                                                    float* __restrict__ V, int B, int Ntok,
                                                    int heads, int dh, int nj, int n, float scale,
                                                    const float* __restrict__ sin_t, const float* __restrict__ cos_t,
                                                    int rot_dim) {
  const int inner = heads * dh;
  const long total = (long)B * Ntok * inner;
  for (long i = (long)blockIdx.x * ST + threadIdx.x; i < total; i += (long)gridDim.x * ST) {
    const int d = (int)(i % dh);
    long t = i / dh;
    const int h = (int)(t % heads);
    t /= heads;
    const int tok = (int)(t % Ntok);
    const int b = (int)(t / Ntok);
    const float* src = qkv + ((long)b * Ntok + tok) * 3 * inner + h * dh;
    float q = src[d] * scale, k = src[inner + d];
    const float v = src[2 * inner + d], k_plain = k;
    if (tok >= nj && d < rot_dim) {
      const int pos = (tok - nj) % n;
      const float sn = sin_t[pos * rot_dim + d], cs = cos_t[pos * rot_dim + d];
      const int dp = d ^ 1;  // partner of the pair; rotate_every_two: even -> -x[d+1], odd -> x[d-1]
      const float qp = src[dp] * scale, kp = src[inner + dp];
      const float sgn = (d & 1) ? 1.f : -1.f;
      q = q * cs + sgn * qp * sn;
      k = k * cs + sgn * kp * sn;
    }
    const long o = (((long)b * heads + h) * Ntok + tok) * dh + d;
    Q[o] = q;
    K[o] = k;
    K0[o] = k_plain;  // the joint queries attend BEFORE the rotary embedding is applied (:305 precedes :311)
    V[o] = v;
  }
}

// mode 0 (patch): grid (ceil(n/128), B*heads*frames); queries = the frame's n patch tokens,
//                 keys = [nj joint tokens | the frame's n patch tokens].
// mode 1 (joint): grid (1, B*heads); queries = the nj joint tokens, keys = all Ntok tokens, split over waves.
template <int DH>
__global__ __launch_bounds__(ST) void k_attention(const float* __restrict__ Q, const float* __restrict__ K,
                                                  const float* __restrict__ V, float* __restrict__ out, int heads,
                                                  int Ntok, int nj, int n, int frames, int mode,
                                                  float* __restrict__ part) {
  constexpr int LD = DH + 1;
  __shared__ float Ks[4][32 * LD];
  __shared__ float Vs[4][32 * LD];
  __shared__ float mrg_m[4][32], mrg_l[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  int bh, f = 0;
  if (mode == 0) {
    bh = blockIdx.y / frames;
    f = blockIdx.y % frames;
  } else {
    bh = blockIdx.y;
  }
  const int b = bh / heads, head = bh % heads;
  const float* Qb = Q + (long)bh * Ntok * DH;
  const float* Kb = K + (long)bh * Ntok * DH;
  const float* Vb = V + (long)bh * Ntok * DH;
  const int nkeys = mode == 0 ? nj + n : Ntok;
  const int nq = mode == 0 ? n : nj;
  // this lane's query
  const int qi = mode == 0 ? blockIdx.x * 128 + wave * 32 + col : col;
  const bool qvalid = qi < nq;
  const int qtok = mode == 0 ? nj + f * n + qi : qi;
  float qreg[DH / 2];
#pragma unroll
  for (int s = 0; s < DH / 2; ++s) qreg[s] = qvalid ? Qb[(long)qtok * DH + 2 * s + half] : 0.f;

  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m = -FLT_MAX, l = 0.f;

  const int ntiles = (nkeys + 31) / 32;
  // joint mode: blockIdx.x is a key split; inside the split the tiles are dealt to the 4 waves
  const int nsplit = mode == 0 ? 1 : (int)gridDim.x;
  const int tiles_per_split = (ntiles + nsplit - 1) / nsplit;
  const int tile0 = mode == 0 ? 0 : (int)blockIdx.x * tiles_per_split;
  const int tile_end = mode == 0 ? ntiles : min(ntiles, tile0 + tiles_per_split);
  const int steps = mode == 0 ? ntiles : (tiles_per_split + 3) / 4;
  for (int it = 0; it < steps; ++it) {
    const int tile = mode == 0 ? it : tile0 + it * 4 + wave;
    __syncthreads();
    if (mode == 0) {  // one tile for the whole block, staged by all 256 threads into region 0
      for (int i = tid; i < 32 * DH; i += ST) {
        const int kr = i / DH, d = i - kr * DH;
        const int kj = tile * 32 + kr;
        float kv = 0.f, vv = 0.f;
        if (kj < nkeys) {
          const int tok = kj < nj ? kj : nj + f * n + (kj - nj);
          kv = Kb[(long)tok * DH + d];
          vv = Vb[(long)tok * DH + d];
        }
        Ks[0][kr * LD + d] = kv;
        Vs[0][kr * LD + d] = vv;
      }
    } else {  // every wave stages its own tile
      for (int i = lane; i < 32 * DH; i += 64) {
        const int kr = i / DH, d = i - kr * DH;
        const int kj = tile * 32 + kr;
        float kv = 0.f, vv = 0.f;
        if (kj < nkeys && tile < tile_end) {
          kv = Kb[(long)kj * DH + d];
          vv = Vb[(long)kj * DH + d];
        }
        Ks[wave][kr * LD + d] = kv;
        Vs[wave][kr * LD + d] = vv;
      }
    }
    __syncthreads();
    const float* ks = mode == 0 ? Ks[0] : Ks[wave];
    const float* vs = mode == 0 ? Vs[0] : Vs[wave];
    if (tile >= tile_end) continue;  // (joint mode tail; barriers above stay uniform)
    // S^T[key][query]
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) sacc = __builtin_amdgcn_mfma_f32_32x32x2f32(ks[col * LD + 2 * s + half], qreg[s], sacc, 0, 0, 0);
    float tm = -FLT_MAX;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (key >= nkeys) sacc[r] = -FLT_MAX;
      tm = fmaxf(tm, sacc[r]);
    }
    tm = fmaxf(tm, __shfl_xor(tm, 32));
    const float mn = fmaxf(m, tm);
    const float alpha = __expf(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = sacc[r] > -FLT_MAX ? __expf(sacc[r] - mn) : 0.f;
      sacc[r] = pr;
      ps += pr;
    }
    ps += __shfl_xor(ps, 32);
    l = l * alpha + ps;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    // O^T[d][query] += V^T[d][key] P[key][query], k order = accumulator row order of P
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int key = (s & 3) + 8 * (s >> 2) + 4 * half;
      const float a = col < DH ? vs[key * LD + col] : 0.f;
      oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, sacc[s], oacc, 0, 0, 0);
    }
  }

  const int inner = heads * DH;
  if (mode == 0) {
    // normalise, transpose through LDS (region `wave` of Ks is free now) and store 128-byte rows
    __syncthreads();
    float* os = Ks[wave];  // [32 queries][LD]
    const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (d < DH) os[col * LD + d] = oacc[r] * inv;
    }
    __syncthreads();
    for (int i = lane; i < 32 * DH; i += 64) {
      const int qr = i / DH, d = i - qr * DH;
      const int q2 = blockIdx.x * 128 + wave * 32 + qr;
      if (q2 < nq) out[((long)b * Ntok + nj + f * n + q2) * inner + head * DH + d] = os[qr * LD + d];
    }
  } else {
    // merge the 4 waves' partial results for the same 32 queries
    __syncthreads();
    float* os = Ks[wave];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
      if (d < DH) os[col * LD + d] = oacc[r];
    }
    if (half == 0) {
      mrg_m[wave][col] = m;
      mrg_l[wave][col] = l;
    }
    __syncthreads();
    // partial record of this split for every (query, d): unnormalised O, plus (max, sum) per query
    float* rec = part + ((long)blockIdx.y * nsplit + blockIdx.x) * (32 * (DH + 2));
    for (int i = tid; i < 32 * DH; i += ST) {
      const int qr = i / DH, d = i - qr * DH;
      float M = -FLT_MAX;
      for (int w = 0; w < 4; ++w) M = fmaxf(M, mrg_m[w][qr]);
      float Lsum = 0.f, o = 0.f;
      for (int w = 0; w < 4; ++w) {
        const float sc = mrg_l[w][qr] > 0.f ? __expf(mrg_m[w][qr] - M) : 0.f;
        Lsum += mrg_l[w][qr] * sc;
        o += Ks[w][qr * LD + d] * sc;
      }
      rec[qr * (DH + 2) + d] = o;
      if (d == 0) {
        rec[qr * (DH + 2) + DH] = M;
        rec[qr * (DH + 2) + DH + 1] = Lsum;
      }
    }
  }
}


// Patch-token attention on the bf16 matrix cores (dim_head 32; BASELINE config 5 "MFMA fp16 attention").
// Same dataflow as k_attention mode 0 -- S^T = K Q^T puts one query per lane, the soft-max stays in fp32
// registers, O^T += V^T P consumes P exactly as it lies in the accumulator -- but each 32-key tile is
// 2 + 2 v_mfma_f32_32x32x16_bf16 instead of 16 + 16 fp32 MFMAs.  The MFMA k-slots of step t are mapped to the
// accumulator rows the lane already holds (slot (half, j) <-> register 8t + j <-> key (j&3) + 8(2t + (j>>2)) + 4 half),
// and the V tile is staged transposed with its keys in that slot order, so both P and V^T are one 16-byte read.
// H = __bf16 (v_mfma_f32_32x32x16_bf16) or _Float16 (v_mfma_f32_32x32x16_f16: the "MFMA fp16 attention" of BASELINE
// configs[4] to the letter; 11 significand bits instead of 8 on Q, K, V and the probabilities, same rate, same dataflow).
template <typename H>
__device__ __forceinline__ f32x16 mfma_h16(const __attribute__((ext_vector_type(8))) H a, const __attribute__((ext_vector_type(8))) H b,
                                           f32x16 c) {
  if constexpr (std::is_same<H, _Float16>::value) return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <typename H>
__global__ __launch_bounds__(ST) void k_attention_patch_h16(const float* __restrict__ Q, const float* __restrict__ K,
                                                             const float* __restrict__ V, float* __restrict__ out, int heads,
                                                             int Ntok, int nj, int n, int frames) {
  constexpr int DH = 32, LDB = DH + 8;  // 80-byte rows: the 16-byte fragment reads of 32 rows spread over all banks
  using bf16x8s = __attribute__((ext_vector_type(8))) H;
  __shared__ __attribute__((aligned(16))) H Kh[32 * LDB];   // [key][d]
  __shared__ __attribute__((aligned(16))) H Vt[32 * LDB];   // [d][key slot]
  __shared__ float os[4][32 * (DH + 1)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  const int bh = blockIdx.y / frames, f = blockIdx.y % frames;
  const int b = bh / heads, head = bh % heads;
  const float* Qb = Q + (long)bh * Ntok * DH;
  const float* Kb = K + (long)bh * Ntok * DH;
  const float* Vb = V + (long)bh * Ntok * DH;
  const int nkeys = nj + n;
  const int qi = blockIdx.x * 128 + wave * 32 + col;
  const bool qvalid = qi < n;
  const int qtok = nj + f * n + qi;
  // B operand of S^T: this lane's query, d = 16 s + 8 half .. + 7.  Scores are kept in the log2 domain (the query carries
  // log2 e): exp(s - m) = exp2(s' - m') is then ONE v_exp_f32 per score instead of a multiply + v_exp_f32.
  constexpr float LOG2E = 1.4426950408889634f;
  bf16x8s qf[2];
  {
    // four 16-byte loads in flight (a query row past n re-reads the last one: its result is not stored); the element-wise
    // conditional form compiled to 16 loads, each waited for on the spot
    const float* qrow = Qb + (long)(nj + f * n + min(qi, n - 1)) * DH + 8 * half;
    float4 q4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q4[u] = *(const float4*)(qrow + 16 * (u >> 1) + 4 * (u & 1));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qf[u >> 1][4 * (u & 1) + 0] = (H)(q4[u].x * LOG2E);
      qf[u >> 1][4 * (u & 1) + 1] = (H)(q4[u].y * LOG2E);
      qf[u >> 1][4 * (u & 1) + 2] = (H)(q4[u].z * LOG2E);
      qf[u >> 1][4 * (u & 1) + 3] = (H)(q4[u].w * LOG2E);
    }
  }

  f32x16 oacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) oacc[r] = 0.f;
  float m = -FLT_MAX, l = 0.f;
  const int ntiles = (nkeys + 31) / 32;
  // With 32-wide heads this kernel is bound by the soft-max's vector work, not by the matrix cores (4 MFMAs = 128 cycles per
  // 32-key tile against ~900 cycles of compare / exp / scale per wave), so that work is what is trimmed: key masking only in
  // the ragged last tile, the running-maximum rescale of the 16 accumulators only when some query's maximum actually grew
  // (wave-uniform test; after the first tiles it rarely does), and the next tile's K / V rows requested BEFORE the current
  // tile's arithmetic (they used to be loaded and waited for, one by one, between the two barriers).
  typedef __attribute__((ext_vector_type(4))) H h4;
  const int kr = tid >> 3, dq = (tid & 7) * 4;   // 32 keys x 8 channel quads = one float4 of K and of V per thread
  const int slot = (kr >> 4) * 16 + ((kr >> 2) & 1) * 8 + (kr & 3) + 4 * ((kr >> 3) & 1);
  auto request = [&](int tile, float4& kv, float4& vv) {
    const int kj = min(tile * 32 + kr, nkeys - 1);   // rows past the last key: any finite row (their scores are masked)
    const int tok = kj < nj ? kj : nj + f * n + (kj - nj);
    kv = *(const float4*)(Kb + (long)tok * DH + dq);
    vv = *(const float4*)(Vb + (long)tok * DH + dq);
  };
  float4 kv, vv;
  request(0, kv, vv);
  for (int tile = 0; tile < ntiles; ++tile) {
    __syncthreads();
    *(h4*)(Kh + kr * LDB + dq) = (h4){(H)kv.x, (H)kv.y, (H)kv.z, (H)kv.w};
    Vt[(dq + 0) * LDB + slot] = (H)vv.x;
    Vt[(dq + 1) * LDB + slot] = (H)vv.y;
    Vt[(dq + 2) * LDB + slot] = (H)vv.z;
    Vt[(dq + 3) * LDB + slot] = (H)vv.w;
    if (tile + 1 < ntiles) request(tile + 1, kv, vv);
    __syncthreads();
    f32x16 sacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
      sacc = mfma_h16<H>(*(const bf16x8s*)(Kh + col * LDB + 16 * s2 + 8 * half), qf[s2], sacc);
    const bool ragged = tile == ntiles - 1 && (nkeys & 31) != 0;   // workgroup-uniform
    if (ragged) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (key >= nkeys) sacc[r] = -FLT_MAX;
      }
    }
    float tm = fmaxf(fmaxf(fmaxf(sacc[0], sacc[1]), fmaxf(sacc[2], sacc[3])), fmaxf(fmaxf(sacc[4], sacc[5]), fmaxf(sacc[6], sacc[7])));
    tm = fmaxf(tm, fmaxf(fmaxf(fmaxf(sacc[8], sacc[9]), fmaxf(sacc[10], sacc[11])), fmaxf(fmaxf(sacc[12], sacc[13]), fmaxf(sacc[14], sacc[15]))));
    tm = fmaxf(tm, __shfl_xor(tm, 32));
    if (__builtin_amdgcn_ballot_w64(tm > m) != 0ull) {   // some query of this wave has a new maximum: rescale
      const float mn = fmaxf(m, tm);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      l *= alpha;
      m = mn;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    }
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pr = __builtin_amdgcn_exp2f(sacc[r] - m);   // masked keys: exp2(-huge) = 0
      sacc[r] = pr;
      ps += pr;
    }
    ps += __shfl_xor(ps, 32);
    l += ps;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8s pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (H)sacc[8 * t + j];
      oacc = mfma_h16<H>(*(const bf16x8s*)(Vt + col * LDB + 16 * t + 8 * half), pf, oacc);
    }
  }
  // normalise, transpose through LDS and store 128-byte rows
  const int inner = heads * DH;
  float* o = os[wave];
  const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
    o[col * (DH + 1) + d] = oacc[r] * inv;
  }
  __syncthreads();
  for (int i = lane; i < 32 * DH; i += 64) {
    const int qr = i / DH, d = i - qr * DH;
    const int q2 = blockIdx.x * 128 + wave * 32 + qr;
    if (q2 < n) out[((long)b * Ntok + nj + f * n + q2) * inner + head * DH + d] = o[qr * (DH + 1) + d];
  }
}

// merge the key splits of the joint-token attention: one thread per (bh, query, d)
__global__ void k_attention_joint_merge(const float* __restrict__ part, float* __restrict__ out, int BH, int heads, int dh,
                                        int Ntok, int nq, int nsplit) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BH * nq * dh) return;
  const int d = i % dh, qr = (i / dh) % nq, bh = i / (dh * nq);
  const int b = bh / heads, head = bh % heads;
  const int rs = dh + 2;
  float M = -FLT_MAX;
  for (int sp = 0; sp < nsplit; ++sp) M = fmaxf(M, part[((long)bh * nsplit + sp) * 32 * rs + qr * rs + dh]);
  float L = 0.f, o = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) {
    const float* rec = part + ((long)bh * nsplit + sp) * 32 * rs + qr * rs;
    const float sc = rec[dh + 1] > 0.f ? __expf(rec[dh] - M) : 0.f;
    L += rec[dh + 1] * sc;
    o += rec[d] * sc;
  }
  out[((long)b * Ntok + qr) * heads * dh + head * dh + d] = o / L;
}

static unsigned sgrid(long n) { return (unsigned)std::max<long>(1, std::min<long>((n + ST - 1) / ST, 256 * 8)); }

}  // namespace hp

using namespace hp;

extern "C" int hp_sformer_patchify(const float* video, float* tokens, int B, int frames, int C, int H, int W, int patch,
                                   void* stream) {
  HP_REQUIRE(video && tokens && patch > 0 && H % patch == 0 && W % patch == 0, "hp_sformer_patchify: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("sformer_patchify", st);
  hipLaunchKernelGGL(k_patchify, dim3(sgrid((long)B * frames * C * H * W)), dim3(ST), 0, st, video, tokens, B, frames, C, H, W,
                     patch);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_layernorm_forward(const float* x, float* y, long rows, int dim, const float* gamma, const float* beta,
                                    float eps, int rows_per_batch, long batch_stride_rows, void* stream) {
  HP_REQUIRE(x && y && gamma && beta && rows > 0 && dim > 0, "hp_layernorm_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("layernorm", st);
  hipLaunchKernelGGL(k_layernorm, dim3((unsigned)((rows + 3) / 4)), dim3(ST), 0, st, x, y, rows, dim, gamma, beta, eps,
                     rows_per_batch, batch_stride_rows);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_geglu_forward(const float* u, float* g, long rows, int hidden, void* stream) {
  HP_REQUIRE(u && g && rows > 0 && hidden > 0, "hp_geglu_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("geglu", st);
  hipLaunchKernelGGL(k_geglu, dim3(sgrid(rows * hidden)), dim3(ST), 0, st, u, g, rows, hidden);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_gelu_forward(const float* x, float* y, long n, void* stream) {
  HP_REQUIRE(x && y && n > 0, "hp_gelu_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("gelu", st);
  hipLaunchKernelGGL(k_gelu, dim3(sgrid(n)), dim3(ST), 0, st, x, y, n);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_sformer_qkv_prepare(const float* qkv, float* Q, float* K, float* K0, float* V, int B, int Ntok, int heads,
                                      int dh,
                                      int num_joints, int patches_per_frame, float scale, const float* sin_t,
                                      const float* cos_t, int rot_dim, void* stream) {
  HP_REQUIRE(qkv && Q && K && K0 && V && (rot_dim == 0 || (sin_t && cos_t)) && rot_dim <= dh && rot_dim % 2 == 0,
             "hp_sformer_qkv_prepare: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("sformer_qkv_prepare", st);
  hipLaunchKernelGGL(k_qkv_prepare, dim3(sgrid((long)B * Ntok * heads * dh)), dim3(ST), 0, st, qkv, Q, K, K0, V, B, Ntok, heads, dh,
                     num_joints, patches_per_frame, scale, sin_t, cos_t, rot_dim);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

constexpr int JOINT_SPLITS = 32;
extern "C" size_t hp_sformer_attention_workspace_bytes(int B, int heads, int dh) {
  return sizeof(float) * (size_t)B * heads * JOINT_SPLITS * 32 * (dh + 2);
}

extern "C" int hp_sformer_attention(const float* Q, const float* K, const float* K0, const float* V, float* out, int B,
                                    int heads, int dh,
                                    int Ntok, int num_joints, int patches_per_frame, int frames, int precision,
                                    void* workspace, void* stream) {
  HP_REQUIRE(Q && K && K0 && V && out && workspace, "hp_sformer_attention: null argument");
  HP_REQUIRE(num_joints <= 32 && Ntok == num_joints + frames * patches_per_frame, "hp_sformer_attention: bad token layout");
  HP_REQUIRE(precision == HP_PRECISION_FP32 || precision == HP_PRECISION_BF16 || precision == HP_PRECISION_FP16,
             "hp_sformer_attention: precision %d not built", precision);
  HP_REQUIRE(precision == HP_PRECISION_FP32 || dh == 32, "hp_sformer_attention: the 16-bit patch attention is built for dim_head 32");
  if (dh != 16 && dh != 24 && dh != 32) {
    set_error("hp_sformer_attention: dim_head %d not built (16, 24, 32)", dh);
    return HP_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = (Ntok + 31) / 32;
  const int nsplit = std::max(1, std::min(JOINT_SPLITS, ntiles / 4));
  float* part = (float*)workspace;
  const dim3 gp((patches_per_frame + 127) / 128, B * heads * frames), gj(nsplit, B * heads);
  {
    HP_PROF("sformer_attention_patch", st);
    if (dh == 32 && precision == HP_PRECISION_BF16)
      hipLaunchKernelGGL(k_attention_patch_h16<__bf16>, gp, dim3(ST), 0, st, Q, K, V, out, heads, Ntok, num_joints, patches_per_frame,
                         frames);
    else if (dh == 32 && precision == HP_PRECISION_FP16)
      hipLaunchKernelGGL(k_attention_patch_h16<_Float16>, gp, dim3(ST), 0, st, Q, K, V, out, heads, Ntok, num_joints, patches_per_frame,
                         frames);
    else if (dh == 32) hipLaunchKernelGGL((k_attention<32>), gp, dim3(ST), 0, st, Q, K, V, out, heads, Ntok, num_joints, patches_per_frame, frames, 0, part);
    else if (dh == 24) hipLaunchKernelGGL((k_attention<24>), gp, dim3(ST), 0, st, Q, K, V, out, heads, Ntok, num_joints, patches_per_frame, frames, 0, part);
    else hipLaunchKernelGGL((k_attention<16>), gp, dim3(ST), 0, st, Q, K, V, out, heads, Ntok, num_joints, patches_per_frame, frames, 0, part);
  }
  if (num_joints > 0) {  // (TokenPose's all-to-all attention has no separate joint / class queries)
    HP_PROF("sformer_attention_joint", st);
    if (dh == 32) hipLaunchKernelGGL((k_attention<32>), gj, dim3(ST), 0, st, Q, K0, V, out, heads, Ntok, num_joints, patches_per_frame, frames, 1, part);
    else if (dh == 24) hipLaunchKernelGGL((k_attention<24>), gj, dim3(ST), 0, st, Q, K0, V, out, heads, Ntok, num_joints, patches_per_frame, frames, 1, part);
    else hipLaunchKernelGGL((k_attention<16>), gj, dim3(ST), 0, st, Q, K0, V, out, heads, Ntok, num_joints, patches_per_frame, frames, 1, part);
    const int total = B * heads * num_joints * dh;
    hipLaunchKernelGGL(k_attention_joint_merge, dim3((total + 255) / 256), dim3(256), 0, st, part, out, B * heads, heads, dh, Ntok,
                       num_joints, nsplit);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
