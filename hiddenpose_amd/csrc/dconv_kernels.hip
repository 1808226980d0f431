// Direct 3x3x3 convolutions for the thin-channel stages (FeatureExtraction: 1 channel;
// UNet3d: 4..64 channels), planar fp32 (B, C, D, H, W) as in the reference.
//
// These layers are bandwidth / VALU work, not matrix work: with 4..32 output channels an
// MFMA tile would be mostly padding, and the fp32 MFMA rate equals the fp32 VALU rate on
// gfx950 anyway.  So: one thread owns VX consecutive x voxels and ALL output channels; the
// input tile with its halo is staged in LDS once per 4-channel chunk; weights are wave
// uniform and come through the scalar cache; every LDS value feeds 3*COUT FMAs.
//
//   forward     y = conv(x, w) + bias                 zero or replicate padding
//   data grad   the same kernel with swapped channel strides and flipped taps
//               (replicate padding: "full" correlation on the +1 halo domain, then a fold)
//   weight grad one thread per (co, ci, dz, dy) row of 3 taps, sliding along x in LDS
#include <algorithm>

#include "hp_internal.h"

namespace hp {

constexpr int DT = 256;
constexpr int TZ = 4, TY = 8, CC = 4;  // tile depth/height, input-channel chunk

template <int COUT, int VX, int PADMODE>
__global__ __launch_bounds__(DT) void k_dconv3(const float* __restrict__ x, const float* __restrict__ w,
                                               const float* __restrict__ bias, float* __restrict__ y, int cin, int Di,
                                               int Hi, int Wi, int Do, int Ho, int Wo, int pad, long wsco, long wsci,
                                               int flip, int tiles_x, int tiles_y) {
  constexpr int TX = 8 * VX;
  constexpr int LZ = TZ + 2, LY = TY + 2, LX = TX + 2, LVOL = LZ * LY * LX;
  __shared__ float xs[CC * LVOL];
  const int tid = threadIdx.x;
  const int txg = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
  int t = blockIdx.x;
  const int bx = t % tiles_x;
  t /= tiles_x;
  const int by = t % tiles_y;
  const int bz = t / tiles_y;
  const int b = blockIdx.y;
  const int oz0 = bz * TZ, oy0 = by * TY, ox0 = bx * TX;
  const long in_cs = (long)Di * Hi * Wi;

  float acc[COUT][VX];
#pragma unroll
  for (int co = 0; co < COUT; ++co)
#pragma unroll
    for (int v = 0; v < VX; ++v) acc[co][v] = 0.f;

  for (int c0 = 0; c0 < cin; c0 += CC) {
    __syncthreads();
    for (int i = tid; i < CC * LVOL; i += DT) {
      const int c = i / LVOL;
      int r = i - c * LVOL;
      const int lz = r / (LY * LX);
      r -= lz * (LY * LX);
      const int ly = r / LX, lx = r - ly * LX;
      int gz = oz0 + lz - pad, gy = oy0 + ly - pad, gx = ox0 + lx - pad;
      float v = 0.f;
      if (c0 + c < cin) {
        if (PADMODE == 1) {
          gz = min(max(gz, 0), Di - 1);
          gy = min(max(gy, 0), Hi - 1);
          gx = min(max(gx, 0), Wi - 1);
          v = x[((long)b * cin + c0 + c) * in_cs + ((long)gz * Hi + gy) * Wi + gx];
        } else if ((unsigned)gz < (unsigned)Di && (unsigned)gy < (unsigned)Hi && (unsigned)gx < (unsigned)Wi) {
          v = x[((long)b * cin + c0 + c) * in_cs + ((long)gz * Hi + gy) * Wi + gx];
        }
      }
      xs[i] = v;
    }
    __syncthreads();
    const int cn = min(CC, cin - c0);
    for (int c = 0; c < cn; ++c) {
      const float* wc = w + (long)(c0 + c) * wsci;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const float* row = xs + c * LVOL + ((tz + dz) * LY + (ty + dy)) * LX + txg * VX;
          float r[VX + 2];
#pragma unroll
          for (int j = 0; j < VX + 2; ++j) r[j] = row[j];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int tap = (dz * 3 + dy) * 3 + dx;
            const int ti = flip ? 26 - tap : tap;
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
              const float wv = wc[(long)co * wsco + ti];
#pragma unroll
              for (int v = 0; v < VX; ++v) acc[co][v] = fmaf(r[v + dx], wv, acc[co][v]);
            }
          }
        }
    }
  }
  const int oz = oz0 + tz, oy = oy0 + ty, ox = ox0 + txg * VX;
  if (oz < Do && oy < Ho) {
    const long out_cs = (long)Do * Ho * Wo;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
      const float bv = bias ? bias[co] : 0.f;
      float* yp = y + ((long)b * COUT + co) * out_cs + ((long)oz * Ho + oy) * Wo + ox;
#pragma unroll
      for (int v = 0; v < VX; ++v)
        if (ox + v < Wo) yp[v] = acc[co][v] + bv;
    }
  }
}

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

// Weight gradient: dW[co][ci][tap] += sum_v g[co][v] * x[ci][v + tap - 1],  db[co] += sum_v g[co][v].
// Block = one voxel tile (4 x 8 x 32) x one (8 co, 4 ci) channel chunk; thread = one
// (co, ci, dz, dy) row, sliding a 3-wide window along x so each FMA costs 2/3 LDS read.
constexpr int WCO = 8, WTX = 32;
template <int PADMODE>
__global__ __launch_bounds__(DT) void k_dconv3_wgrad(const float* __restrict__ x, const float* __restrict__ g,
                                                     float* __restrict__ dw, float* __restrict__ db, int cin, int cout,
                                                     int D, int H, int W, int tiles_x, int tiles_y, int co_chunks,
                                                     int ci_chunks) {
  constexpr int LZ = TZ + 2, LY = TY + 2, LX = WTX + 2, LVOL = LZ * LY * LX;
  constexpr int GV = TZ * TY * WTX, GLD = GV + 1;
  __shared__ float xs[CC * LVOL];
  __shared__ float gs[WCO * GLD];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int bx = t % tiles_x;
  t /= tiles_x;
  const int by = t % tiles_y;
  const int bz = t / tiles_y;
  const int b = blockIdx.y;
  const int cic = blockIdx.z % ci_chunks, coc = blockIdx.z / ci_chunks;
  (void)co_chunks;
  const int co0 = coc * WCO, ci0 = cic * CC;
  const int z0 = bz * TZ, y0 = by * TY, x0 = bx * WTX;
  const long cs = (long)D * H * W;
  for (int i = tid; i < CC * LVOL; i += DT) {
    const int c = i / LVOL;
    int r = i - c * LVOL;
    const int lz = r / (LY * LX);
    r -= lz * (LY * LX);
    const int ly = r / LX, lx = r - ly * LX;
    int gz = z0 + lz - 1, gy = y0 + ly - 1, gx = x0 + lx - 1;
    float v = 0.f;
    if (ci0 + c < cin) {
      if (PADMODE == 1) {
        gz = min(max(gz, 0), D - 1);
        gy = min(max(gy, 0), H - 1);
        gx = min(max(gx, 0), W - 1);
        v = x[((long)b * cin + ci0 + c) * cs + ((long)gz * H + gy) * W + gx];
      } else if ((unsigned)gz < (unsigned)D && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
        v = x[((long)b * cin + ci0 + c) * cs + ((long)gz * H + gy) * W + gx];
      }
    }
    xs[i] = v;
  }
  for (int i = tid; i < WCO * GV; i += DT) {
    const int c = i / GV;
    int r = i - c * GV;
    const int lz = r / (TY * WTX);
    r -= lz * (TY * WTX);
    const int ly = r / WTX, lx = r - ly * WTX;
    const int gz = z0 + lz, gy = y0 + ly, gx = x0 + lx;
    float v = 0.f;
    if (co0 + c < cout && gz < D && gy < H && gx < W) v = g[((long)b * cout + co0 + c) * cs + ((long)gz * H + gy) * W + gx];
    gs[c * GLD + (i - c * GV)] = v;
  }
  __syncthreads();
  // rows: (co, ci, dz, dy) -> 8 * 4 * 9 = 288 rows over 256 threads
  for (int row = tid; row < WCO * CC * 9; row += DT) {
    const int dy = row % 3, dz = (row / 3) % 3, ci = (row / 9) % CC, co = row / (9 * CC);
    if (co0 + co >= cout || ci0 + ci >= cin) continue;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float* gp = gs + co * GLD;
    const float* xp = xs + ci * LVOL;
    for (int z = 0; z < TZ; ++z)
      for (int yy = 0; yy < TY; ++yy) {
        const float* xr = xp + ((z + dz) * LY + (yy + dy)) * LX;
        const float* gr = gp + (z * TY + yy) * WTX;
        float w0 = xr[0], w1 = xr[1];
#pragma unroll 8
        for (int xx = 0; xx < WTX; ++xx) {
          const float w2 = xr[xx + 2];
          const float gv = gr[xx];
          a0 = fmaf(gv, w0, a0);
          a1 = fmaf(gv, w1, a1);
          a2 = fmaf(gv, w2, a2);
          w0 = w1;
          w1 = w2;
        }
      }
    float* o = dw + ((long)(co0 + co) * cin + ci0 + ci) * 27 + (dz * 3 + dy) * 3;
    atomicAdd(o + 0, a0);
    atomicAdd(o + 1, a1);
    atomicAdd(o + 2, a2);
  }
  if (db && cic == 0 && tid < WCO && co0 + tid < cout) {
    float s = 0.f;
    const float* gp = gs + tid * GLD;
    for (int i = 0; i < GV; ++i) s += gp[i];
    atomicAdd(db + co0 + tid, s);
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
  auto stage = [&](int zp) {
    float* dst = ring + (zp & 3) * WG_PLANE + tid;
    int zz = zp;
    bool zok = (unsigned)zz < (unsigned)D;
    if (PADMODE == 1) zz = min(max(zz, 0), D - 1), zok = true;
    const float* src = xb + (long)zz * H * W;
    float v[SK];
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = (zok && ((smask >> k) & 1u)) ? src[soff[k]] : 0.f;
#pragma unroll
    for (int k = 0; k < SK; ++k)
      if (tid + 256 * k < WG_PLANE) dst[256 * k] = v[k];
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
    if (z + 1 < ze) stage(z + 2);  // slot (z+2)&3 was last read as plane z-2: its readers passed the previous barrier
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
    auto stage = [&](int zin) {  // input plane zin (any integer) -> ring slot zin & 3
      float* dst = ring + (zin & 3) * WG_PLANE + tid;
      int zz = zin;
      bool zok = (unsigned)zz < (unsigned)Di;
      if (PADMODE == 1) zz = min(max(zz, 0), Di - 1), zok = true;
      const float* src = xb + (long)zz * Hi * Wi;
      float v[SK];
#pragma unroll
      for (int k = 0; k < SK; ++k) v[k] = (zok && ((smask >> k) & 1u)) ? src[soff[k]] : 0.f;
#pragma unroll
      for (int k = 0; k < SK; ++k)
        if (tid + 256 * k < WG_PLANE) dst[256 * k] = v[k];
    };
    if (zb < ze) {
      stage(zb - pad);
      stage(zb - pad + 1);
      stage(zb - pad + 2);
    }
    __syncthreads();
    const int lbase = (2 * wave) * WG_PX + lane;
    for (int z = zb; z < ze; ++z) {
      if (z + 1 < ze) stage(z - pad + 3);  // the slot it replaces held plane z - pad - 1, last read one barrier ago
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
      __syncthreads();
    }
  }
}

template <int COUT, int VX>
static void launch_dconv(int padmode, dim3 grid, hipStream_t st, const float* x, const float* w, const float* bias,
                         float* y, int cin, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int pad, long wsco, long wsci,
                         int flip, int tiles_x, int tiles_y) {
  if (padmode)
    hipLaunchKernelGGL((k_dconv3<COUT, VX, 1>), grid, dim3(DT), 0, st, x, w, bias, y, cin, Di, Hi, Wi, Do, Ho, Wo, pad,
                       wsco, wsci, flip, tiles_x, tiles_y);
  else
    hipLaunchKernelGGL((k_dconv3<COUT, VX, 0>), grid, dim3(DT), 0, st, x, w, bias, y, cin, Di, Hi, Wi, Do, Ho, Wo, pad,
                       wsco, wsci, flip, tiles_x, tiles_y);
}

static int vx_for(int cout) { return cout <= 8 ? 4 : 2; }

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
  if (replicate_pad)
    hipLaunchKernelGGL((k_dconv3_wgrad_mfma<1>), grid, dim3(256), 0, st, x, gy, cin, cout, D, H, W, q.tiles_x, q.tiles_y,
                       q.zchunk, q.cig_n, partial);
  else
    hipLaunchKernelGGL((k_dconv3_wgrad_mfma<0>), grid, dim3(256), 0, st, x, gy, cin, cout, D, H, W, q.tiles_x, q.tiles_y,
                       q.zchunk, q.cig_n, partial);
  hipLaunchKernelGGL(k_dconv3_wgrad_reduce, dim3((unsigned)(q.cog_n * q.cig_n), 7), dim3(1024), 0, st, partial, dw, dbias, cin, cout,
                     q.cig_n, q.nwg);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
