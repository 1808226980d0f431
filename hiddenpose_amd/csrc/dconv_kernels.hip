// Direct 3x3x3 convolutions for the thin-channel stages (FeatureExtraction: 1 channel;
// UNet3d: 4..64 channels), planar fp32 (B, C, D, H, W) as in the reference.
//
// With 1..64 channels a 32x32 MFMA tile would be mostly padding, but gfx950 also has
// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products per instruction, exact fp32, 64 FLOP/clk/SIMD
// (twice the VALU FMA rate) -- a 4-channel x 4-channel (or 4-channel x 4-voxel) block is exactly this
// layer family's shape.  All three directions use it, with the input planes of a 4-channel chunk held in
// a 4-slot LDS ring while the workgroup slides along z:
//
//   forward     y = conv(x, w) + bias                 zero or replicate padding
//   data grad   the same kernel with swapped channel strides and flipped taps
//               (replicate padding: "full" correlation on the +1 halo domain, then a fold)
//   weight grad 16 voxels x (4 co x 4 ci) per instruction, atomic-free two-stage reduction
#include <algorithm>

#include "hp_internal.h"

namespace hp {


// adjoint of replicate padding by 1: dx[p] = sum of the halo-domain cells that clamp to p
__global__ void k_fold_replicate(const float* __restrict__ dpad, float* __restrict__ dx, long nvol, int D, int H, int W) {
  const int De = D + 2, He = H + 2, We = W + 2;
  const long per = (long)D * H * W, total = nvol * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long vol = i / per;
    long r = i - vol * per;
    const int z = (int)(r / ((long)H * W));
    r -= (long)z * H * W;
    const int yy = (int)(r / W), xx = (int)(r - (long)yy * W);
    const int z0 = z == 0 ? 0 : z + 1, z1 = z == D - 1 ? D + 1 : z + 1;
    const int y0 = yy == 0 ? 0 : yy + 1, y1 = yy == H - 1 ? H + 1 : yy + 1;
    const int x0 = xx == 0 ? 0 : xx + 1, x1 = xx == W - 1 ? W + 1 : xx + 1;
    float s = 0.f;
    const float* p = dpad + vol * (long)De * He * We;
    for (int a = z0; a <= z1; ++a)
      for (int bb = y0; bb <= y1; ++bb)
        for (int c = x0; c <= x1; ++c) s += p[((long)a * He + bb) * We + c];
    dx[i] = s;
  }
}

// Weight gradient on the matrix cores, for any channel count: v_mfma_f32_4x4x1_16b_f32 computes 16 independent
// 4x4 outer products per instruction (64 FLOP/clk/SIMD, twice the VALU FMA rate, exact fp32).  Block b of the
// 16 = voxel x + b of a 16-voxel x-run; row i = output channel co0 + i, column j = input channel ci0 + j:
//     D_b[i][j] += g[co0+i][v_b] * x[ci0+j][v_b + tap]
// so lane 4b+i supplies g, lane 4b+j supplies the shifted x, and each lane keeps D_b[.][j] (4 registers) per
// tap: 27 taps x 4 = 108 accumulators (+4 for the bias gradient: the same product with x = 1).
// A workgroup owns an 8 (y) x 64 (x) column of one (batch, 4 co, 4 ci) task and slides along z; the padded x
// planes live in a 4-slot LDS ring (plane z+2 is fetched while plane z is multiplied: one barrier per plane).
// Every shifted operand is ONE ds_read_b32 with a compile-time offset (row pitch 68: the 64 lanes of a read
// fall on every bank exactly twice, the minimum for 256 bytes); g comes straight from global memory, one load
// per 27 MFMAs.  At the end the 16 blocks are summed with shuffles, the 4 waves combine in LDS and the
// workgroup issues one atomic per weight (all workgroups add into the same few hundred floats, so the number
// of atomic packets has to stay small).
constexpr int WG_TY = 8, WG_TX = 64, WG_PX = 68, WG_PY = WG_TY + 2, WG_PLANE = 4 * WG_PY * WG_PX;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int PADMODE>
__global__ __launch_bounds__(256, 2) void k_dconv3_wgrad_mfma(const float* __restrict__ x, const float* __restrict__ g,
                                                           int cin, int cout,
                                                           int D, int H, int W, int tiles_x, int tiles_y, int zchunk,
                                                           int cig_n, float* __restrict__ partial) {
  __shared__ float ring[4 * WG_PLANE];
  __shared__ float part[4][28 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = lane >> 2, sub = lane & 3;
  int t_ = blockIdx.x;
  const int bx = t_ % tiles_x;
  t_ /= tiles_x;
  const int by = t_ % tiles_y;
  const int bz = t_ / tiles_y;
  const int b = blockIdx.y;
  const int cig = blockIdx.z % cig_n, cog = blockIdx.z / cig_n;
  const int x0 = bx * WG_TX, y0 = by * WG_TY;
  const int zb = bz * zchunk, ze = min(D, zb + zchunk);
  const int co = cog * 4 + sub;
  const bool co_ok = co < cout;
  const long cs = (long)D * H * W;
  const float* gch = g + ((long)b * cout + (co_ok ? co : 0)) * cs;
  const float* xb = x + ((long)b * cin + cig * 4) * cs;
  const bool want_db = cig == 0;

  // Staging map, fixed for the whole z walk: element e = tid + 256 k of the [4 ch][10 rows][68 cols] plane image
  // (the LDS offset IS e) comes from offset soff[k] inside the channel-0 z-plane, or is zero when bit k of smask
  // is clear (zero padding, columns 66/67, channels past cin).
  constexpr int SK = (WG_PLANE + 255) / 256;
  int soff[SK];
  unsigned smask = 0;
#pragma unroll
  for (int k = 0; k < SK; ++k) {
    const int e = tid + 256 * k;
    const int c = e / (WG_PY * WG_PX);
    const int r = e - c * (WG_PY * WG_PX);
    const int ly = r / WG_PX, lx = r - ly * WG_PX;
    int yy = y0 + ly - 1, xx = x0 + lx - 1;
    bool ok = e < WG_PLANE && lx < 66 && cig * 4 + c < cin;
    if (PADMODE == 1) {
      yy = min(max(yy, 0), H - 1);
      xx = min(max(xx, 0), W - 1);
    } else {
      ok = ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    }
    soff[k] = ok ? (int)((long)c * cs + (long)yy * W + xx) : 0;
    smask |= ok ? (1u << k) : 0u;
  }
  // a plane is fetched into registers (stage_load) before the multiply phase and parked in its ring slot
  // (stage_store) after it, so the global-memory latency hides behind the MFMAs
  auto stage_load = [&](int zp, float (&v)[SK]) {
    int zz = zp;
    bool zok = (unsigned)zz < (unsigned)D;
    if (PADMODE == 1) zz = min(max(zz, 0), D - 1), zok = true;
    const float* src = xb + (long)zz * H * W;
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = (zok && ((smask >> k) & 1u)) ? src[soff[k]] : 0.f;
  };
  auto stage_store = [&](int zp, const float (&v)[SK]) {
    float* dst = ring + (zp & 3) * WG_PLANE + tid;
#pragma unroll
    for (int k = 0; k < SK; ++k)
      if (tid + 256 * k < WG_PLANE) dst[256 * k] = v[k];
  };
  auto stage = [&](int zp) {
    float v[SK];
    stage_load(zp, v);
    stage_store(zp, v);
  };

  f32x4 acc[28];
#pragma unroll
  for (int t = 0; t < 28; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (zb < ze) {
    stage(zb - 1);
    stage(zb);
    stage(zb + 1);
  }
  __syncthreads();
  // this lane's offset inside a plane for row r, run q: (sub * PY + (2*wave + r) + dy) * PX + q*16 + blk + dx
  const int lbase = (sub * WG_PY + 2 * wave) * WG_PX + blk;
  for (int z = zb; z < ze; ++z) {
    float nxt[SK];
    if (z + 1 < ze) stage_load(z + 2, nxt);
    __builtin_amdgcn_sched_barrier(0);  // keep the fetch ahead of the multiply phase (the scheduler would sink it)
    const float* p0 = ring + ((z - 1) & 3) * WG_PLANE + lbase;
    const float* p1 = ring + (z & 3) * WG_PLANE + lbase;
    const float* p2 = ring + ((z + 1) & 3) * WG_PLANE + lbase;
    // 8 (row, run) pairs per plane, one g value each; the next pair's g is in flight while this one multiplies
    auto g_at = [&](int rq) -> float {
      const int y = y0 + 2 * wave + (rq >> 2), xv = x0 + (rq & 3) * 16 + blk;
      return (co_ok && y < H && xv < W) ? gch[((long)z * H + y) * W + xv] : 0.f;
    };
    float gnext = g_at(0);
#pragma unroll 1
    for (int rq = 0; rq < 8; ++rq) {
      const float gv = gnext;
      if (rq + 1 < 8) gnext = g_at(rq + 1);
      const int off = (rq >> 2) * WG_PX + (rq & 3) * 16;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const float* pl = (dz == 0 ? p0 : dz == 1 ? p1 : p2) + off;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            acc[(dz * 3 + dy) * 3 + dx] =
                __builtin_amdgcn_mfma_f32_4x4x1f32(gv, pl[dy * WG_PX + dx], acc[(dz * 3 + dy) * 3 + dx], 0, 0, 0);
      }
      if (want_db) acc[27] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, 1.0f, acc[27], 0, 0, 0);
    }
    if (z + 1 < ze) stage_store(z + 2, nxt);  // slot (z+2)&3 was last read as plane z-2: its readers passed the previous barrier
    __syncthreads();
  }
  // sum the 16 blocks: lanes with equal (lane & 3) hold the same (., j) column of different voxels
#pragma unroll
  for (int t = 0; t < 28; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[t][i];
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      acc[t][i] = v;
    }
  if (blk == 0) {
#pragma unroll
    for (int t = 0; t < 28; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) part[wave][(i * 4 + sub) * 28 + t] = acc[t][i];  // [co i][ci j][tap]
  }
  __syncthreads();
  // partial[pair][workgroup of the pair][448]: plain stores, summed by k_dconv3_wgrad_reduce in a fixed order
  const long wg = (long)blockIdx.y * gridDim.x + blockIdx.x, nwg = (long)gridDim.x * gridDim.y;
  float* out = partial + ((long)blockIdx.z * nwg + wg) * (16 * 28);
  for (int u = tid; u < 16 * 28; u += 256) out[u] = part[0][u] + part[1][u] + part[2][u] + part[3][u];
}


// Single input channel (FeatureExtraction, first U-Net conv): the 4 columns of the outer product would be 3/4
// padding, so they carry four TAPS instead -- column j of step m is tap 4m + j, 7 instructions cover the 27 taps:
//     D_b[i][j] += g[co0+i][v_b] * x[0][v_b + tap(4m+j)]
// Same ring (one channel per plane image), same partial layout ([co][ci = 0][tap]) and reduction kernel.
constexpr int W1_PLANE = WG_PY * WG_PX;
template <int PADMODE>
__global__ __launch_bounds__(256, 2) void k_dconv3_wgrad_mfma_c1(const float* __restrict__ x, const float* __restrict__ g,
                                                                 int cout, int D, int H, int W, int tiles_x, int tiles_y,
                                                                 int zchunk, float* __restrict__ partial) {
  __shared__ float ring[4 * W1_PLANE];
  __shared__ float part[4][28 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = lane >> 2, sub = lane & 3;
  int t_ = blockIdx.x;
  const int bx = t_ % tiles_x;
  t_ /= tiles_x;
  const int by = t_ % tiles_y;
  const int bz = t_ / tiles_y;
  const int b = blockIdx.y, cog = blockIdx.z;
  const int x0 = bx * WG_TX, y0 = by * WG_TY;
  const int zb = bz * zchunk, ze = min(D, zb + zchunk);
  const int co = cog * 4 + sub;
  const bool co_ok = co < cout;
  const long cs = (long)D * H * W;
  const float* gch = g + ((long)b * cout + (co_ok ? co : 0)) * cs;
  const float* xb = x + (long)b * cs;
  for (int u = tid; u < 4 * 28 * 16; u += 256) (&part[0][0])[u] = 0.f;

  constexpr int SK = (W1_PLANE + 255) / 256;
  int soff[SK];
  unsigned smask = 0;
#pragma unroll
  for (int k = 0; k < SK; ++k) {
    const int e = tid + 256 * k;
    const int ly = e / WG_PX, lx = e - ly * WG_PX;
    int yy = y0 + ly - 1, xx = x0 + lx - 1;
    bool ok = e < W1_PLANE && lx < 66;
    if (PADMODE == 1) {
      yy = min(max(yy, 0), H - 1);
      xx = min(max(xx, 0), W - 1);
    } else {
      ok = ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    }
    soff[k] = ok ? (int)((long)yy * W + xx) : 0;
    smask |= ok ? (1u << k) : 0u;
  }
  // a plane is fetched into registers (stage_load) before the multiply phase and parked in its ring slot
  // (stage_store) after it, so the global-memory latency hides behind the MFMAs
  auto stage_load = [&](int zp, float (&v)[SK]) {
    int zz = zp;
    bool zok = (unsigned)zz < (unsigned)D;
    if (PADMODE == 1) zz = min(max(zz, 0), D - 1), zok = true;
    const float* src = xb + (long)zz * H * W;
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = (zok && ((smask >> k) & 1u)) ? src[soff[k]] : 0.f;
  };
  auto stage_store = [&](int zp, const float (&v)[SK]) {
    float* dst = ring + (zp & 3) * W1_PLANE + tid;
#pragma unroll
    for (int k = 0; k < SK; ++k)
      if (tid + 256 * k < W1_PLANE) dst[256 * k] = v[k];
  };
  auto stage = [&](int zp) {
    float v[SK];
    stage_load(zp, v);
    stage_store(zp, v);
  };
  // this lane's tap of step m: t = 4m + sub -> (dz, in-plane offset); t = 27 does not exist (operand forced to 0)
  int tdz[7], toff[7];
#pragma unroll
  for (int m = 0; m < 7; ++m) {
    const int t = min(4 * m + sub, 26);
    tdz[m] = t / 9;
    toff[m] = ((t / 3) % 3) * WG_PX + t % 3;
  }
  const bool last_ok = sub != 3;  // tap 27

  f32x4 acc[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (zb < ze) {
    stage(zb - 1);
    stage(zb);
    stage(zb + 1);
  }
  __syncthreads();
  const int lbase = (2 * wave) * WG_PX + blk;
  for (int z = zb; z < ze; ++z) {
    float nxt[SK];
    if (z + 1 < ze) stage_load(z + 2, nxt);
    __builtin_amdgcn_sched_barrier(0);  // keep the fetch ahead of the multiply phase (the scheduler would sink it)
    const float* pm[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) pm[m] = ring + ((z - 1 + tdz[m]) & 3) * W1_PLANE + lbase + toff[m];
    auto g_at = [&](int rq) -> float {
      const int y = y0 + 2 * wave + (rq >> 2), xv = x0 + (rq & 3) * 16 + blk;
      return (co_ok && y < H && xv < W) ? gch[((long)z * H + y) * W + xv] : 0.f;
    };
    float gnext = g_at(0);
#pragma unroll 1
    for (int rq = 0; rq < 8; ++rq) {
      const float gv = gnext;
      if (rq + 1 < 8) gnext = g_at(rq + 1);
      const int off = (rq >> 2) * WG_PX + (rq & 3) * 16;
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        float xv = pm[m][off];
        if (m == 6) xv = last_ok ? xv : 0.f;
        acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, xv, acc[m], 0, 0, 0);
      }
      acc[7] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv, 1.0f, acc[7], 0, 0, 0);
    }
    if (z + 1 < ze) stage_store(z + 2, nxt);
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[m][i];
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      acc[m][i] = v;
    }
  if (blk == 0) {
#pragma unroll
    for (int m = 0; m < 7; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (4 * m + sub < 27) part[wave][(i * 4) * 28 + 4 * m + sub] = acc[m][i];  // [co i][ci 0][tap]
    if (sub == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) part[wave][(i * 4) * 28 + 27] = acc[7][i];
    }
  }
  __syncthreads();
  const long wg = (long)blockIdx.y * gridDim.x + blockIdx.x, nwg = (long)gridDim.x * gridDim.y;
  float* out = partial + ((long)blockIdx.z * nwg + wg) * (16 * 28);
  for (int u = tid; u < 16 * 28; u += 256) out[u] = part[0][u] + part[1][u] + part[2][u] + part[3][u];
}

// dw[co][ci][tap] / db[co] = sum over the workgroups of the (co group, ci group) pair, in a fixed order:
// grid (pair, 7 column chunks of 64); 1024 threads = 16 slices of the workgroup range x 64 columns.
__global__ __launch_bounds__(1024) void k_dconv3_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw,
                                                              float* __restrict__ db, int cin, int cout, int cig_n, long nwg) {
  __shared__ float red[16][64];
  const int pair = blockIdx.x, cig = pair % cig_n, cog = pair / cig_n;
  const int slice = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int u = blockIdx.y * 64 + lane;
  const float* src = partial + (long)pair * nwg * (16 * 28) + u;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long w = slice;
  for (; w + 48 < nwg; w += 64) {
    s0 += src[w * (16 * 28)];
    s1 += src[(w + 16) * (16 * 28)];
    s2 += src[(w + 32) * (16 * 28)];
    s3 += src[(w + 48) * (16 * 28)];
  }
  for (; w < nwg; w += 16) s0 += src[w * (16 * 28)];
  red[slice][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (slice == 0) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][lane];
    const int t = u % 28, j = (u / 28) & 3, i = u / (28 * 4);
    const int oc = cog * 4 + i, ic = cig * 4 + j;
    if (oc < cout) {
      if (t < 27) {
        if (ic < cin) dw[((long)oc * cin + ic) * 27 + t] = v;
      } else if (db && cig == 0 && j == 0) {
        db[oc] = v;
      }
    }
  }
}


// Forward / data-gradient 3x3x3 convolution on the same 4x4x1 matrix-core instruction.  Roles: row i = output
// channel co0 + i (lane 4b+i supplies the weight, identical in all 16 blocks), column j = voxel 4b + j of a 64-voxel
// x-run (lane l supplies x[l + tap]), K = one (input channel, tap) per instruction:
//     D_b[i][j] += w[co0+i][ci][tap] * x[ci][v_{4b+j} + tap]        -> lane l keeps out[co0 .. co0+3][v_l]
// A workgroup owns an 8 (y) x 64 (x) output column of one (batch, 4 co) task and slides along z with the input
// planes of a 4-channel chunk in the same 4-slot LDS ring as the weight gradient; the 27 weights of the current
// input channel sit in registers (7 broadcast ds_read_b128 per channel and plane).  More than 4 input channels:
// the chunks are separate z sweeps and every sweep after the first adds into y.
template <int PADMODE>
__global__ __launch_bounds__(256, 2) void k_dconv3_mfma(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int cin,
                                                         int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int pad,
                                                         long wsco, long wsci, int flip, int tiles_x, int tiles_y, int zchunk) {
  __shared__ float ring[4 * WG_PLANE];
  __shared__ __attribute__((aligned(16))) float wl[4 * 4 * 28];  // [ci][co][28]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 3;
  int t_ = blockIdx.x;
  const int bx = t_ % tiles_x;
  t_ /= tiles_x;
  const int by = t_ % tiles_y;
  const int bz = t_ / tiles_y;
  const int b = blockIdx.y, cog = blockIdx.z;
  const int x0 = bx * WG_TX, y0 = by * WG_TY;
  const int zb = bz * zchunk, ze = min(Do, zb + zchunk);
  const long ics = (long)Di * Hi * Wi, ocs = (long)Do * Ho * Wo;
  constexpr int SK = (WG_PLANE + 255) / 256;

  for (int c0 = 0; c0 < cin; c0 += 4) {
    const int nci = min(4, cin - c0);
    const float* xb = x + ((long)b * cin + c0) * ics;
    __syncthreads();  // previous sweep: ring and weight image are free
    // weights of this (4 co, 4 ci) pair: wl[ci][co][tap]
    for (int e = tid; e < 4 * 4 * 28; e += 256) {
      const int tap = e % 28, co = (e / 28) & 3, ci = e / (28 * 4);
      float v = 0.f;
      if (tap < 27 && cog * 4 + co < cout && ci < nci)
        v = w[(long)(cog * 4 + co) * wsco + (long)(c0 + ci) * wsci + (flip ? 26 - tap : tap)];
      wl[e] = v;
    }
    // staging map (see the weight-gradient kernel): element e = tid + 256 k of the [4][10][68] plane image
    int soff[SK];
    unsigned smask = 0;
#pragma unroll
    for (int k = 0; k < SK; ++k) {
      const int e = tid + 256 * k;
      const int c = e / (WG_PY * WG_PX);
      const int r = e - c * (WG_PY * WG_PX);
      const int ly = r / WG_PX, lx = r - ly * WG_PX;
      int yy = y0 + ly - pad, xx = x0 + lx - pad;
      bool ok = e < WG_PLANE && lx < 66 && c < nci;
      if (PADMODE == 1) {
        yy = min(max(yy, 0), Hi - 1);
        xx = min(max(xx, 0), Wi - 1);
      } else {
        ok = ok && (unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi;
      }
      soff[k] = ok ? (int)((long)c * ics + (long)yy * Wi + xx) : 0;
      smask |= ok ? (1u << k) : 0u;
    }
    // a plane is fetched into registers (stage_load) before the multiply phase and parked in its ring slot
    // (stage_store) after it, so the global-memory latency hides behind the MFMAs
    auto stage_load = [&](int zin, float (&v)[SK]) {
      int zz = zin;
      bool zok = (unsigned)zz < (unsigned)Di;
      if (PADMODE == 1) zz = min(max(zz, 0), Di - 1), zok = true;
      const float* src = xb + (long)zz * Hi * Wi;
#pragma unroll
      for (int k = 0; k < SK; ++k) v[k] = (zok && ((smask >> k) & 1u)) ? src[soff[k]] : 0.f;
    };
    auto stage_store = [&](int zin, const float (&v)[SK]) {
      float* dst = ring + (zin & 3) * WG_PLANE + tid;
#pragma unroll
      for (int k = 0; k < SK; ++k)
        if (tid + 256 * k < WG_PLANE) dst[256 * k] = v[k];
    };
    auto stage = [&](int zin) {
      float v[SK];
      stage_load(zin, v);
      stage_store(zin, v);
    };
    if (zb < ze) {
      stage(zb - pad);
      stage(zb - pad + 1);
      stage(zb - pad + 2);
    }
    __syncthreads();
    const int lbase = (2 * wave) * WG_PX + lane;
    for (int z = zb; z < ze; ++z) {
      float nxt[SK];
      if (z + 1 < ze) stage_load(z - pad + 3, nxt);
      __builtin_amdgcn_sched_barrier(0);  // keep the fetch ahead of the multiply phase (the scheduler would sink it)
      const float* p0 = ring + ((z - pad) & 3) * WG_PLANE + lbase;
      const float* p1 = ring + ((z - pad + 1) & 3) * WG_PLANE + lbase;
      const float* p2 = ring + ((z - pad + 2) & 3) * WG_PLANE + lbase;
      f32x4 acc[2];
      acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
      for (int ci = 0; ci < nci; ++ci) {
        float wr[28];
        const f32x4* wp = (const f32x4*)(wl + (ci * 4 + sub) * 28);
#pragma unroll
        for (int q = 0; q < 7; ++q) {
          const f32x4 t4 = wp[q];
          wr[4 * q] = t4[0];
          wr[4 * q + 1] = t4[1];
          wr[4 * q + 2] = t4[2];
          wr[4 * q + 3] = t4[3];
        }
        const int co_ = ci * WG_PY * WG_PX;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int dz = 0; dz < 3; ++dz) {
            const float* pl = (dz == 0 ? p0 : dz == 1 ? p1 : p2) + co_ + r * WG_PX;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
              for (int dx = 0; dx < 3; ++dx)
                acc[r] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[(dz * 3 + dy) * 3 + dx], pl[dy * WG_PX + dx], acc[r], 0, 0, 0);
          }
      }
      // lane l holds out[co0 + i][row][x0 + l] in acc[row][i]
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int oy = y0 + 2 * wave + r, ox = x0 + lane;
        if (oy < Ho && ox < Wo) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int oc = cog * 4 + i;
            if (oc < cout) {
              float* yp = y + ((long)b * cout + oc) * ocs + ((long)z * Ho + oy) * Wo + ox;
              *yp = acc[r][i] + (c0 == 0 ? (bias ? bias[oc] : 0.f) : *yp);
            }
          }
        }
      }
      if (z + 1 < ze) stage_store(z - pad + 3, nxt);  // the slot held plane z - pad - 1, last read one barrier ago
      __syncthreads();
    }
  }
}

// generic entry: y (B,cout,Do,Ho,Wo) = conv3(x (B,cin,Di,Hi,Wi)) with weight(co,ci,tap) = w[co*wsco + ci*wsci + tap']
static int run_dconv(const float* x, const float* w, const float* bias, float* y, int B, int cin, int cout, int Di, int Hi,
                     int Wi, int Do, int Ho, int Wo, int pad, long wsco, long wsci, int flip, int padmode,
                     hipStream_t st) {
  const int tiles_x = (Wo + WG_TX - 1) / WG_TX, tiles_y = (Ho + WG_TY - 1) / WG_TY, cog_n = (cout + 3) / 4;
  const long cols = (long)tiles_x * tiles_y * B * cog_n;
  int zsplit = (int)std::max<long>(1, std::min<long>((Do + 7) / 8, (1536 + cols - 1) / cols));
  const int zchunk = (Do + zsplit - 1) / zsplit;
  zsplit = (Do + zchunk - 1) / zchunk;
  dim3 grid((unsigned)(tiles_x * tiles_y * zsplit), (unsigned)B, (unsigned)cog_n);
  if (padmode)
    hipLaunchKernelGGL((k_dconv3_mfma<1>), grid, dim3(256), 0, st, x, w, bias, y, cin, cout, Di, Hi, Wi, Do, Ho, Wo, pad, wsco,
                       wsci, flip, tiles_x, tiles_y, zchunk);
  else
    hipLaunchKernelGGL((k_dconv3_mfma<0>), grid, dim3(256), 0, st, x, w, bias, y, cin, cout, Di, Hi, Wi, Do, Ho, Wo, pad, wsco,
                       wsci, flip, tiles_x, tiles_y, zchunk);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

}  // namespace hp

using namespace hp;

extern "C" int hp_dconv3_forward(const float* x, const float* w, const float* bias, float* y, int B, int cin, int cout,
                                 int D, int H, int W, int replicate_pad, void* stream) {
  HP_REQUIRE(x && w && y && B > 0 && cin > 0 && cout > 0, "hp_dconv3_forward: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("dconv3_fwd", st);
  return run_dconv(x, w, bias, y, B, cin, cout, D, H, W, D, H, W, 1, (long)cin * 27, 27, 0, replicate_pad, st);
}

extern "C" size_t hp_dconv3_backward_data_workspace_bytes(int B, int cin, int D, int H, int W, int replicate_pad) {
  return replicate_pad ? sizeof(float) * (size_t)B * cin * (D + 2) * (H + 2) * (W + 2) : 0;
}

extern "C" int hp_dconv3_backward_data(const float* gy, const float* w, float* gx, int B, int cin, int cout, int D, int H,
                                       int W, int replicate_pad, void* workspace, void* stream) {
  HP_REQUIRE(gy && w && gx && B > 0, "hp_dconv3_backward_data: bad argument");
  hipStream_t st = (hipStream_t)stream;
  // gx[ci] = sum_co corr(gy[co], flipped w[co][ci]): roles of the channel strides swap
  if (!replicate_pad) {
    HP_PROF("dconv3_dgrad", st);
    return run_dconv(gy, w, nullptr, gx, B, cout, cin, D, H, W, D, H, W, 1, 27, (long)cin * 27, 1, 0, st);
  }
  HP_REQUIRE(workspace, "hp_dconv3_backward_data: replicate padding needs the workspace");
  float* dpad = (float*)workspace;
  {
    HP_PROF("dconv3_dgrad", st);
    int rc = run_dconv(gy, w, nullptr, dpad, B, cout, cin, D, H, W, D + 2, H + 2, W + 2, 2, 27, (long)cin * 27, 1, 0, st);
    if (rc) return rc;
  }
  const long total = (long)B * cin * D * H * W;
  hipLaunchKernelGGL(k_fold_replicate, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, dpad, gx,
                     (long)B * cin, D, H, W);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

namespace {
struct WgradGeom {
  int cog_n, cig_n, tiles_x, tiles_y, zsplit, zchunk;
  long nwg;  // workgroups per channel pair
};
WgradGeom wgrad_geom(int B, int cin, int cout, int D, int H, int W) {
  WgradGeom q;
  q.cog_n = (cout + 3) / 4;
  q.cig_n = (cin + 3) / 4;
  q.tiles_x = (W + hp::WG_TX - 1) / hp::WG_TX;
  q.tiles_y = (H + hp::WG_TY - 1) / hp::WG_TY;
  // z range per workgroup: ~1024 workgroups (2 resident per CU), at least 8 planes each: 2 halo planes are
  // re-read and one 448-value reduction is paid per chunk
  const long cols = (long)q.tiles_x * q.tiles_y * B * q.cog_n * q.cig_n;
  q.zsplit = (int)std::max<long>(1, std::min<long>((D + 7) / 8, (1024 + cols - 1) / cols));
  q.zchunk = (D + q.zsplit - 1) / q.zsplit;
  q.zsplit = (D + q.zchunk - 1) / q.zchunk;
  q.nwg = (long)q.tiles_x * q.tiles_y * q.zsplit * B;
  return q;
}
}  // namespace

extern "C" size_t hp_dconv3_backward_weight_workspace_bytes(int B, int cin, int cout, int D, int H, int W) {
  if (B < 1 || cin < 1 || cout < 1 || D < 1 || H < 1 || W < 1) return 0;
  const WgradGeom q = wgrad_geom(B, cin, cout, D, H, W);
  return sizeof(float) * (size_t)q.cog_n * q.cig_n * q.nwg * 16 * 28;
}

extern "C" int hp_dconv3_backward_weight(const float* x, const float* gy, float* dw, float* dbias, int B, int cin,
                                         int cout, int D, int H, int W, int replicate_pad, void* workspace, void* stream) {
  HP_REQUIRE(x && gy && dw && workspace && B > 0, "hp_dconv3_backward_weight: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const WgradGeom q = wgrad_geom(B, cin, cout, D, H, W);
  dim3 grid((unsigned)(q.tiles_x * q.tiles_y * q.zsplit), (unsigned)B, (unsigned)(q.cog_n * q.cig_n));
  float* partial = (float*)workspace;
  HP_PROF("dconv3_wgrad", st);
  if (cin == 1) {
    if (replicate_pad)
      hipLaunchKernelGGL((k_dconv3_wgrad_mfma_c1<1>), grid, dim3(256), 0, st, x, gy, cout, D, H, W, q.tiles_x, q.tiles_y, q.zchunk,
                         partial);
    else
      hipLaunchKernelGGL((k_dconv3_wgrad_mfma_c1<0>), grid, dim3(256), 0, st, x, gy, cout, D, H, W, q.tiles_x, q.tiles_y, q.zchunk,
                         partial);
  } else if (replicate_pad) {
    hipLaunchKernelGGL((k_dconv3_wgrad_mfma<1>), grid, dim3(256), 0, st, x, gy, cin, cout, D, H, W, q.tiles_x, q.tiles_y,
                       q.zchunk, q.cig_n, partial);
  } else {
    hipLaunchKernelGGL((k_dconv3_wgrad_mfma<0>), grid, dim3(256), 0, st, x, gy, cin, cout, D, H, W, q.tiles_x, q.tiles_y,
                       q.zchunk, q.cig_n, partial);
  }
  hipLaunchKernelGGL(k_dconv3_wgrad_reduce, dim3((unsigned)(q.cog_n * q.cig_n), 7), dim3(1024), 0, st, partial, dw, dbias, cin, cout,
                     q.cig_n, q.nwg);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
