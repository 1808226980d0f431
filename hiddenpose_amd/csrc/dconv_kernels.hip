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

// Workgroup b of a launch runs on XCD b % 8 (each XCD has its own L2), and neighbouring (y, x) tiles share their halo rows
// and columns: with the plain order no two neighbours ever meet in one L2.  Each XCD takes a contiguous eighth of the tile
// list instead (whole launches of a multiple of 8 workgroups per grid row only, so that b % 8 is the XCD in every row).
__device__ int g_xcd_slab = 1;
__device__ __forceinline__ int xcd_slab_tile(unsigned b, unsigned n) {
  return ((n & 7u) || !g_xcd_slab) ? (int)b : (int)((b & 7u) * (n >> 3) + (b >> 3));
}


// adjoint of replicate padding by 1: dx[p] = sum of the halo-domain cells that clamp to p.  grid (x tiles, H, nvol * D):
// no index divisions, interior cells are one coalesced read; only the six faces sum 2 / 4 / 8 cells.
__global__ __launch_bounds__(256) void k_fold_replicate(const float* __restrict__ dpad, float* __restrict__ dx, int D, int H, int W) {
  const int xx = blockIdx.x * 256 + threadIdx.x;
  if (xx >= W) return;
  const int yy = blockIdx.y, z = blockIdx.z % D;
  const long vol = blockIdx.z / D;
  const int He = H + 2, We = W + 2;
  const int z0 = z == 0 ? 0 : z + 1, z1 = z == D - 1 ? D + 1 : z + 1;
  const int y0 = yy == 0 ? 0 : yy + 1, y1 = yy == H - 1 ? H + 1 : yy + 1;
  const int x0 = xx == 0 ? 0 : xx + 1, x1 = xx == W - 1 ? W + 1 : xx + 1;
  const float* p = dpad + vol * (long)(D + 2) * He * We;
  float s = 0.f;
  for (int a = z0; a <= z1; ++a)
    for (int bb = y0; bb <= y1; ++bb)
      for (int c = x0; c <= x1; ++c) s += p[((long)a * He + bb) * We + c];
  dx[((vol * D + z) * H + yy) * (long)W + xx] = s;
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
  int t_ = xcd_slab_tile(blockIdx.x, gridDim.x);
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
  const float* xb = x + ((long)b * cin + cig * 4) * cs;
  const bool want_db = cig == 0;

  // Staging map, fixed for the whole z walk: element e = tid + 256 k of the [4 ch][10 rows][68 cols] plane image
  // (the LDS offset IS e) comes from offset soff[k] inside the channel-0 z-plane, or is zero when bit k of smask
  // is clear (zero padding, columns 66/67, channels past cin).
  constexpr int SK = (WG_PLANE + 255) / 256;
  // buffer loads (see k_dconv3_mfma): fixed per-thread byte offsets, out-of-range = the zero padding, plane in the scalar offset
  constexpr unsigned OOB = 0x80000000u;
  // num_records: the rest of sample b's channels from this group on (hp_extent) -- a slip stays inside the sample's tensor
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, hp_extent((long)(cin - cig * 4) * cs, 0, 4), 0x00020000);
  unsigned soff[SK];
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
    soff[k] = ok ? (unsigned)(((long)c * cs + (long)yy * W + xx) * 4) : OOB;
  }
  // a plane is fetched into registers (stage_load) before the multiply phase and parked in its ring slot
  // (stage_store) after it, so the global-memory latency hides behind the MFMAs
  // steady-state form (plane inside the volume) and generic form: see k_dconv3_mfma -- a zero-fill on the other side of a
  // branch makes the compiler wait for the prefetch in flight (vmcnt(0)) at the join
  auto stage_load_fast = [&](int zp, float (&v)[SK]) {
    const int zz = PADMODE == 1 ? min(max(zp, 0), D - 1) : zp;
    const unsigned zs = (unsigned)((long)zz * H * W * 4);
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, soff[k], zs, 0));
  };
  auto stage_load = [&](int zp, float (&v)[SK]) {
    if (PADMODE == 1 || (unsigned)zp < (unsigned)D) {
      stage_load_fast(zp, v);
    } else {
#pragma unroll
      for (int k = 0; k < SK; ++k) v[k] = 0.f;
    }
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
  // g of one plane: 8 (row, run) values per lane
  // g through a buffer descriptor at this workgroup's 4 output channels: lane part (channel sub, x run) fixed, row and plane scalar
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)(g + ((long)b * cout + cog * 4) * cs), 0, hp_extent((long)(cout - cog * 4) * cs, 0, 4), 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  unsigned goff[4];
#pragma unroll
  for (int qx = 0; qx < 4; ++qx) {
    const int xv = x0 + qx * 16 + blk;
    goff[qx] = (co_ok && xv < W) ? (unsigned)(((long)sub * cs + xv) * 4) : OOB;
  }
  unsigned goffr[8];   // the same with the row's validity folded in: the steady-state loads carry no condition
#pragma unroll
  for (int rq = 0; rq < 8; ++rq) goffr[rq] = y0 + 2 * wv + (rq >> 2) < H ? goff[rq & 3] : OOB;
  auto g_load_fast = [&](int z, float (&gv)[8]) {
#pragma unroll
    for (int rq = 0; rq < 8; ++rq)
      gv[rq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, goffr[rq], (unsigned)(((long)z * H + y0 + 2 * wv + (rq >> 2)) * W * 4), 0));
  };
  auto g_load = [&](int z, float (&gv)[8]) {
#pragma unroll
    for (int rq = 0; rq < 8; ++rq) {
      const int y = y0 + 2 * wv + (rq >> 2);   // scalar
      if (z < ze && y < H)
        gv[rq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, goff[rq & 3], (unsigned)(((long)z * H + y) * W * 4), 0));
      else
        gv[rq] = 0.f;
    }
  };
  auto multiply = [&](int z, const float (&gv)[8]) {
    const float* p0 = ring + ((z - 1) & 3) * WG_PLANE + lbase;
    const float* p1 = ring + (z & 3) * WG_PLANE + lbase;
    const float* p2 = ring + ((z + 1) & 3) * WG_PLANE + lbase;
    // 24 groups of 9 taps ((row, run) x dz), software-pipelined through two 9-register operand sets: the LDS reads of group
    // g + 1 are in flight during the 9 MFMAs of group g.  (One (row, run) at a time -- 27 reads, then 27 MFMAs that wait for
    // them -- exposed an LDS round trip eight times per plane: the matrix pipe was 42 % busy at two waves per SIMD.)
    float ops[2][9];
    auto ld9 = [&](int g_, float (&v)[9]) {
      const int rq = g_ / 3, dz = g_ % 3;
      const float* pl = (dz == 0 ? p0 : dz == 1 ? p1 : p2) + (rq >> 2) * WG_PX + (rq & 3) * 16;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) v[dy * 3 + dx] = pl[dy * WG_PX + dx];
    };
    ld9(0, ops[0]);
#pragma unroll
    for (int g_ = 0; g_ < 24; ++g_) {
      if (g_ + 1 < 24) ld9(g_ + 1, ops[(g_ + 1) & 1]);
      const int rq = g_ / 3, dz = g_ % 3;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9)
        acc[dz * 9 + t9] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv[rq], ops[g_ & 1][t9], acc[dz * 9 + t9], 0, 0, 0);
      if (dz == 2 && want_db) acc[27] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv[rq], 1.0f, acc[27], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep the groups in order (the scheduler would hoist all 216 LDS reads)
    }
  };
  // Everything a plane needs from global memory is requested one full step (g) or two steps (x planes) before it is
  // used: with single-step prefetch and g fetched inside its own step the kernel was bound by the few KB it kept in
  // flight per CU, not by the matrix pipe.
  float nA[SK], nB[SK], gA[8], gB[8];
  if (zb < ze) g_load(zb, gA);
  if (zb + 1 < ze) stage_load(zb + 2, nA);
  // steady state: both steps of the pair, their g rows and their prefetched planes lie inside the chunk / the volume
  int z = zb;
  for (; z + 3 < ze && (PADMODE == 1 || z + 4 < D); z += 2) {
    stage_load_fast(z + 3, nB);
    g_load_fast(z + 1, gB);
    __builtin_amdgcn_sched_barrier(0);  // keep the fetches ahead of the multiply phase (the scheduler would sink them)
    multiply(z, gA);
    stage_store(z + 2, nA);  // slot (z+2)&3 was last read as plane z-2: its readers passed the previous barrier
    __syncthreads();
    stage_load_fast(z + 4, nA);
    g_load_fast(z + 2, gA);
    __builtin_amdgcn_sched_barrier(0);
    multiply(z + 1, gB);
    stage_store(z + 3, nB);
    __syncthreads();
  }
  for (; z < ze; z += 2) {   // chunk end, generic form
    if (z + 2 < ze) stage_load(z + 3, nB);
    g_load(z + 1, gB);
    __builtin_amdgcn_sched_barrier(0);  // keep the fetches ahead of the multiply phase (the scheduler would sink them)
    multiply(z, gA);
    if (z + 1 < ze) stage_store(z + 2, nA);  // slot (z+2)&3 was last read as plane z-2: its readers passed the previous barrier
    __syncthreads();
    if (z + 1 < ze) {
      if (z + 3 < ze) stage_load(z + 4, nA);
      g_load(z + 2, gA);
      __builtin_amdgcn_sched_barrier(0);
      multiply(z + 1, gB);
      if (z + 2 < ze) stage_store(z + 3, nB);
      __syncthreads();
    }
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


// bf16-operand weight gradient (HP_PRECISION_BF16): v_mfma_f32_4x4x4_16b_bf16 with K = FOUR CONSECUTIVE VOXELS of an x-run,
//     D_b[i][j] += sum_{k<4} g[co0+i][v_{4b+k}] * x[ci0+j][v_{4b+k} + tap]
// so one instruction covers a whole 64-voxel row (16 blocks x 4) where the exact kernel needs four.  Lane 4b+i supplies four
// consecutive g values (one 16-byte global load, rounded to bf16), lane 4b+j the four x values starting at voxel 4b + dx of
// the shifted row: the ring keeps the planes PLANAR in bf16 (one row = 68 halves), a lane reads the eight halves 4b .. 4b+7
// of its row once per (dz, dy) -- two 8-byte-aligned ds_read_b64 -- and cuts the three dx windows out of them in registers
// (dx = 1: two v_alignbit; dx = 2: the middle four halves, no instruction): 3 MFMAs per 2 LDS reads instead of 12 per 12.
// The bias gradient is summed exactly (fp32, from the unrounded g).  Needs W % 4 == 0 and a 16-byte aligned g (the caller
// falls back to the exact kernel otherwise).  Same tiling, z walk, partial layout and reduction kernel as above.
using wbf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using wbf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using ws16x4 = __attribute__((ext_vector_type(4))) short;
using wf32x2 = __attribute__((ext_vector_type(2))) float;
using wu32x2 = __attribute__((ext_vector_type(2))) unsigned;
__device__ __forceinline__ wu32x2 wpack_bf16x4(float a, float b, float c, float d) {
  const wbf16x2 lo = __builtin_convertvector((wf32x2){a, b}, wbf16x2);
  const wbf16x2 hi = __builtin_convertvector((wf32x2){c, d}, wbf16x2);
  return (wu32x2){__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}
constexpr int WH_ROW = WG_PX;                       // halves per row (68: 136 bytes, rows stay 8-byte aligned)
constexpr int WH_PLANE = 4 * WG_PY * WH_ROW;        // halves per ring slot (4 channels x 10 rows)
constexpr int WH_GROUPS = WH_ROW / 4;               // 17 four-cell groups per row
constexpr int WH_ITEMS = 4 * WG_PY * WH_GROUPS;     // 680 staging items (channel, row, group) per plane

template <int PADMODE>
__global__ __launch_bounds__(256, 2) void k_dconv3_wgrad_bf16(const float* __restrict__ x, const float* __restrict__ g,
                                                              int cin, int cout, int D, int H, int W, int tiles_x, int tiles_y,
                                                              int zchunk, int cig_n, float* __restrict__ partial) {
  __shared__ __attribute__((aligned(16))) unsigned short ring[4 * WH_PLANE];
  __shared__ float part[4][28 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int blk = lane >> 2, sub = lane & 3;
  int t_ = xcd_slab_tile(blockIdx.x, gridDim.x);
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
  const float* xb = x + ((long)b * cin + cig * 4) * cs;
  const bool want_db = cig == 0;
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, hp_extent((long)(cin - cig * 4) * cs, 0, 4), 0x00020000);
  // staging item it = tid + 256 k: (channel c, row ly, group gq) -> cells lx = 4 gq .. 4 gq + 3 of that row, packed into one
  // 8-byte LDS write; every cell has its own fixed offset (out of range = zero padding / past the volume / past cin)
  constexpr int SKI = (WH_ITEMS + 255) / 256;
  unsigned soff[SKI][4];
#pragma unroll
  for (int k = 0; k < SKI; ++k) {
    const int it = tid + 256 * k;
    const int c = it / (WG_PY * WH_GROUPS);
    const int r = it - c * (WG_PY * WH_GROUPS);
    const int ly = r / WH_GROUPS, gq = r - ly * WH_GROUPS;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int lx = 4 * gq + m;
      int yy = y0 + ly - 1, xx = x0 + lx - 1;
      bool ok = it < WH_ITEMS && lx < 66 && cig * 4 + c < cin;
      if (PADMODE == 1) {
        yy = min(max(yy, 0), H - 1);
        xx = min(max(xx, 0), W - 1);
      } else {
        ok = ok && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      }
      soff[k][m] = ok ? (unsigned)(((long)c * cs + (long)yy * W + xx) * 4) : OOB;
    }
  }
  auto stage_load_fast = [&](int zp, float (&v)[SKI][4]) {   // steady-state form: plane inside the volume
    const int zz = PADMODE == 1 ? min(max(zp, 0), D - 1) : zp;
    const unsigned zs = (unsigned)((long)zz * H * W * 4);
#pragma unroll
    for (int k = 0; k < SKI; ++k)
#pragma unroll
      for (int m = 0; m < 4; ++m) v[k][m] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, soff[k][m], zs, 0));
  };
  auto stage_load = [&](int zp, float (&v)[SKI][4]) {
    if (PADMODE == 1 || (unsigned)zp < (unsigned)D) {
      stage_load_fast(zp, v);
    } else {
#pragma unroll
      for (int k = 0; k < SKI; ++k)
#pragma unroll
        for (int m = 0; m < 4; ++m) v[k][m] = 0.f;
    }
  };
  auto stage_store = [&](int zp, const float (&v)[SKI][4]) {
    wu32x2* dst = (wu32x2*)(ring + (zp & 3) * WH_PLANE) + tid;   // item it occupies halves 4 it .. 4 it + 3 of the slot
#pragma unroll
    for (int k = 0; k < SKI; ++k)
      if (tid + 256 * k < WH_ITEMS) dst[256 * k] = wpack_bf16x4(v[k][0], v[k][1], v[k][2], v[k][3]);
  };
  auto stage = [&](int zp) {
    float v[SKI][4];
    stage_load(zp, v);
    stage_store(zp, v);
  };

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float sdb = 0.f;   // exact bias-gradient partial of output channel `sub` (this lane's four voxels per row)

  if (zb < ze) {
    stage(zb - 1);
    stage(zb);
    stage(zb + 1);
  }
  __syncthreads();
  // this lane's window inside a slot for row r: halves (sub * PY + 2 * wave + r + dy) * ROW + 4 * blk .. + 7
  const int lbase = (sub * WG_PY + 2 * wave) * WH_ROW + 4 * blk;
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)(g + ((long)b * cout + cog * 4) * cs), 0, hp_extent((long)(cout - cog * 4) * cs, 0, 4), 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int xv = x0 + 4 * blk;
  const unsigned goff = (co_ok && xv < W) ? (unsigned)(((long)sub * cs + xv) * 4) : OOB;   // W % 4 == 0: the four voxels are in or out together
  unsigned goffr[2];
#pragma unroll
  for (int r = 0; r < 2; ++r) goffr[r] = y0 + 2 * wv + r < H ? goff : OOB;
  auto g_load_fast = [&](int z, float4 (&gv)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
      gv[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, goffr[r], (unsigned)(((long)z * H + y0 + 2 * wv + r) * W * 4), 0));
  };
  auto g_load = [&](int z, float4 (&gv)[2]) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int y = y0 + 2 * wv + r;   // scalar
      if (z < ze && y < H) {
        gv[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(grs, goff, (unsigned)(((long)z * H + y) * W * 4), 0));
      } else {
        gv[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto multiply = [&](int z, const float4 (&gv)[2]) {
    const unsigned short* p0 = ring + ((z - 1) & 3) * WH_PLANE + lbase;
    const unsigned short* p1 = ring + (z & 3) * WH_PLANE + lbase;
    const unsigned short* p2 = ring + ((z + 1) & 3) * WH_PLANE + lbase;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const ws16x4 ga = __builtin_bit_cast(ws16x4, wpack_bf16x4(gv[r].x, gv[r].y, gv[r].z, gv[r].w));
      if (want_db) sdb += (gv[r].x + gv[r].y) + (gv[r].z + gv[r].w);
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const unsigned short* pl = (dz == 0 ? p0 : dz == 1 ? p1 : p2) + r * WH_ROW;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const wu32x2 lo = *(const wu32x2*)(pl + dy * WH_ROW), hi = *(const wu32x2*)(pl + dy * WH_ROW + 4);
          const wu32x2 w1 = {__builtin_amdgcn_alignbit(lo[1], lo[0], 16), __builtin_amdgcn_alignbit(hi[0], lo[1], 16)};
          const wu32x2 w2 = {lo[1], hi[0]};
          const int t = (dz * 3 + dy) * 3;
          acc[t] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ga, __builtin_bit_cast(ws16x4, lo), acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ga, __builtin_bit_cast(ws16x4, w1), acc[t + 1], 0, 0, 0);
          acc[t + 2] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ga, __builtin_bit_cast(ws16x4, w2), acc[t + 2], 0, 0, 0);
        }
      }
    }
  };
  float nA[SKI][4], nB[SKI][4];
  float4 gA[2], gB[2];
  if (zb < ze) g_load(zb, gA);
  if (zb + 1 < ze) stage_load(zb + 2, nA);
  // steady state: both steps of the pair, their g rows and their prefetched planes lie inside the chunk / the volume
  int z = zb;
  for (; z + 3 < ze && (PADMODE == 1 || z + 4 < D); z += 2) {
    stage_load_fast(z + 3, nB);
    g_load_fast(z + 1, gB);
    __builtin_amdgcn_sched_barrier(0);  // keep the fetches ahead of the multiply phase (the scheduler would sink them)
    multiply(z, gA);
    stage_store(z + 2, nA);  // slot (z+2)&3 was last read as plane z-2: its readers passed the previous barrier
    __syncthreads();
    stage_load_fast(z + 4, nA);
    g_load_fast(z + 2, gA);
    __builtin_amdgcn_sched_barrier(0);
    multiply(z + 1, gB);
    stage_store(z + 3, nB);
    __syncthreads();
  }
  for (; z < ze; z += 2) {   // chunk end, generic form
    if (z + 2 < ze) stage_load(z + 3, nB);
    g_load(z + 1, gB);
    __builtin_amdgcn_sched_barrier(0);
    multiply(z, gA);
    if (z + 1 < ze) stage_store(z + 2, nA);
    __syncthreads();
    if (z + 1 < ze) {
      if (z + 3 < ze) stage_load(z + 4, nA);
      g_load(z + 2, gA);
      __builtin_amdgcn_sched_barrier(0);
      multiply(z + 1, gB);
      if (z + 2 < ze) stage_store(z + 3, nB);
      __syncthreads();
    }
  }
  // sum the 16 blocks: lanes with equal (lane & 3) hold the same (., j) column of different voxels
#pragma unroll
  for (int t = 0; t < 27; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = acc[t][i];
      v += __shfl_xor(v, 4);
      v += __shfl_xor(v, 8);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      acc[t][i] = v;
    }
  sdb += __shfl_xor(sdb, 4);
  sdb += __shfl_xor(sdb, 8);
  sdb += __shfl_xor(sdb, 16);
  sdb += __shfl_xor(sdb, 32);
  if (blk == 0) {
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) part[wave][(i * 4 + sub) * 28 + t] = acc[t][i];  // [co i][ci j][tap]
    // slot 27 = bias gradient of output channel i, read by the reduction from column j = 0: this lane owns channel `sub`
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) part[wave][(sub * 4 + jj) * 28 + 27] = jj == 0 ? sdb : 0.f;
  }
  __syncthreads();
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
  int t_ = xcd_slab_tile(blockIdx.x, gridDim.x);
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
  const float* xb = x + (long)b * cs;
  for (int u = tid; u < 4 * 28 * 16; u += 256) (&part[0][0])[u] = 0.f;

  constexpr int SK = (W1_PLANE + 255) / 256;
  constexpr unsigned OOB = 0x80000000u;   // buffer loads as in k_dconv3_wgrad_mfma
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, hp_extent(cs, 0, 4), 0x00020000);
  unsigned soff[SK];
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
    soff[k] = ok ? (unsigned)(((long)yy * W + xx) * 4) : OOB;
  }
  // a plane is fetched into registers (stage_load) before the multiply phase and parked in its ring slot
  // (stage_store) after it, so the global-memory latency hides behind the MFMAs
  // steady-state form (plane inside the volume) and generic form: see k_dconv3_mfma -- a zero-fill on the other side of a
  // branch makes the compiler wait for the prefetch in flight (vmcnt(0)) at the join
  auto stage_load_fast = [&](int zp, float (&v)[SK]) {
    const int zz = PADMODE == 1 ? min(max(zp, 0), D - 1) : zp;
    const unsigned zs = (unsigned)((long)zz * H * W * 4);
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, soff[k], zs, 0));
  };
  auto stage_load = [&](int zp, float (&v)[SK]) {
    if (PADMODE == 1 || (unsigned)zp < (unsigned)D) {
      stage_load_fast(zp, v);
    } else {
#pragma unroll
      for (int k = 0; k < SK; ++k) v[k] = 0.f;
    }
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
  // g through a buffer descriptor at this workgroup's 4 output channels: lane part (channel sub, x run) fixed, row and plane scalar
  const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc((void*)(g + ((long)b * cout + cog * 4) * cs), 0, hp_extent((long)(cout - cog * 4) * cs, 0, 4), 0x00020000);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  unsigned goff[4];
#pragma unroll
  for (int qx = 0; qx < 4; ++qx) {
    const int xv = x0 + qx * 16 + blk;
    goff[qx] = (co_ok && xv < W) ? (unsigned)(((long)sub * cs + xv) * 4) : OOB;
  }
  unsigned goffr[8];   // the same with the row's validity folded in: the steady-state loads carry no condition
#pragma unroll
  for (int rq = 0; rq < 8; ++rq) goffr[rq] = y0 + 2 * wv + (rq >> 2) < H ? goff[rq & 3] : OOB;
  auto g_load_fast = [&](int z, float (&gv)[8]) {
#pragma unroll
    for (int rq = 0; rq < 8; ++rq)
      gv[rq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, goffr[rq], (unsigned)(((long)z * H + y0 + 2 * wv + (rq >> 2)) * W * 4), 0));
  };
  auto g_load = [&](int z, float (&gv)[8]) {
#pragma unroll
    for (int rq = 0; rq < 8; ++rq) {
      const int y = y0 + 2 * wv + (rq >> 2);   // scalar
      if (z < ze && y < H)
        gv[rq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, goff[rq & 3], (unsigned)(((long)z * H + y) * W * 4), 0));
      else
        gv[rq] = 0.f;
    }
  };
  auto multiply = [&](int z, const float (&gv)[8]) {
    const float* pm[7];
#pragma unroll
    for (int m = 0; m < 7; ++m) pm[m] = ring + ((z - 1 + tdz[m]) & 3) * W1_PLANE + lbase + toff[m];
#pragma unroll
    for (int rq = 0; rq < 8; ++rq) {
      const int off = (rq >> 2) * WG_PX + (rq & 3) * 16;
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        float xv = pm[m][off];
        if (m == 6) xv = last_ok ? xv : 0.f;
        acc[m] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv[rq], xv, acc[m], 0, 0, 0);
      }
      acc[7] = __builtin_amdgcn_mfma_f32_4x4x1f32(gv[rq], 1.0f, acc[7], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  float nA[SK], nB[SK], gA[8], gB[8];  // see k_dconv3_wgrad_mfma: g one step ahead, x planes two steps ahead
  if (zb < ze) g_load(zb, gA);
  if (zb + 1 < ze) stage_load(zb + 2, nA);
  // steady state: both steps of the pair, their g rows and their prefetched planes lie inside the chunk / the volume
  int z = zb;
  for (; z + 3 < ze && (PADMODE == 1 || z + 4 < D); z += 2) {
    stage_load_fast(z + 3, nB);
    g_load_fast(z + 1, gB);
    __builtin_amdgcn_sched_barrier(0);  // keep the fetches ahead of the multiply phase (the scheduler would sink them)
    multiply(z, gA);
    stage_store(z + 2, nA);  // slot (z+2)&3 was last read as plane z-2: its readers passed the previous barrier
    __syncthreads();
    stage_load_fast(z + 4, nA);
    g_load_fast(z + 2, gA);
    __builtin_amdgcn_sched_barrier(0);
    multiply(z + 1, gB);
    stage_store(z + 3, nB);
    __syncthreads();
  }
  for (; z < ze; z += 2) {   // chunk end, generic form
    if (z + 2 < ze) stage_load(z + 3, nB);
    g_load(z + 1, gB);
    __builtin_amdgcn_sched_barrier(0);
    multiply(z, gA);
    if (z + 1 < ze) stage_store(z + 2, nA);
    __syncthreads();
    if (z + 1 < ze) {
      if (z + 3 < ze) stage_load(z + 4, nA);
      g_load(z + 2, gA);
      __builtin_amdgcn_sched_barrier(0);
      multiply(z + 1, gB);
      if (z + 2 < ze) stage_store(z + 3, nB);
      __syncthreads();
    }
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


struct EpiVals {
  float a[2][4], r[2][4];  // [row of the wave][output channel]: addend (bias or partial sum), residual
};

// Forward / data-gradient 3x3x3 convolution on the same 4x4x1 matrix-core instruction.  Roles: row i = output
// channel co0 + i (lane 4b+i supplies the weight, identical in all 16 blocks), column j = voxel 4b + j of a 64-voxel
// x-run (lane l supplies x[l + tap]), K = one (input channel, tap) per instruction:
//     D_b[i][j] += w[co0+i][ci][tap] * x[ci][v_{4b+j} + tap]        -> lane l keeps out[co0 .. co0+3][v_l]
// A workgroup owns an 8 (y) x 64 (x) output column of one (batch, 4 co) task and slides along z with the input
// planes of a 4-channel chunk in the same 4-slot LDS ring as the weight gradient; the 27 weights of the current
// input channel sit in registers (7 broadcast ds_read_b128 per channel and plane).  More than 4 input channels:
// the chunks are separate z sweeps and every sweep after the first adds into y.
template <int PADMODE, int S, bool PLAIN>
__global__ __launch_bounds__(256, 2) void k_dconv3_mfma(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int cin,
                                                         int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int pad,
                                                         long wsco, long wsci, int flip, int tiles_x, int tiles_y, int zchunk,
                                                         const float* __restrict__ res, double* __restrict__ stats, float slope) {
  __shared__ float ring[4 * WG_PLANE];
  __shared__ float sred[4][8];
  f32x4 st1v = {0.f, 0.f, 0.f, 0.f}, st2v = {0.f, 0.f, 0.f, 0.f};   // per-channel sum / sum of squares of this lane's outputs
  __shared__ __attribute__((aligned(16))) float wl[4 * 4 * 28];  // [ci][co][28]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 3;
  int t_ = xcd_slab_tile(blockIdx.x, gridDim.x);
  const int bx = t_ % tiles_x;
  t_ /= tiles_x;
  const int by = t_ % tiles_y;
  const int bz = t_ / tiles_y;
  const int b = blockIdx.y, cog = blockIdx.z;
  const int x0 = bx * WG_TX, y0 = by * WG_TY;
  const int zb = bz * zchunk, ze = min(Do, zb + zchunk);
  const long ics = (long)Di * Hi * Wi, ocs = (long)Do * Ho * Wo;
  constexpr int SK = (WG_PLANE + 255) / 256;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const bool xin = x0 + lane < Wo;
  const unsigned xoff = xin ? (unsigned)((x0 + lane) * 4) : 0x80000000u;
  const long ybase = ((long)b * cout + cog * 4) * ocs;   // this workgroup's 4 output channels: offsets within them fit 31 bits
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias && cog * 4 + i < cout ? bias[cog * 4 + i] : 0.f;

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
    // staging map (see the weight-gradient kernel): element e = tid + 256 k of the [4][10][68] plane image.  The plane is
    // fetched with BUFFER loads: descriptor base = this sweep's 4-channel chunk, per-thread byte offset fixed for the sweep
    // (cells outside the volume carry an out-of-range offset: the hardware returns the zero padding), the plane in the scalar
    // offset -- no vector address arithmetic per plane (while a wave streams fp32 MFMAs the other waves of its SIMD issue no
    // vector-ALU instruction: tools/micro/mfma_coexec.hip; each one here is paid in matrix-pipe time).
    constexpr unsigned OOB = 0x80000000u;
    unsigned soff[SK];
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
      soff[k] = ok ? (unsigned)(((long)c * ics + (long)yy * Wi + xx) * 4) : OOB;
    }
    // a plane is fetched into registers (stage_load) before the multiply phase and parked in its ring slot
    // (stage_store) after it, so the global-memory latency hides behind the MFMAs
    // Two forms of every per-step piece.  The STEADY-STATE form (suffix _fast) carries no per-step condition: its plane is
    // known to lie inside the volume and its step inside the chunk, so the loop body is straight-line code around the
    // loop-invariant branches.  The generic form (chunk ends, tiny volumes) branches and zero-fills.  Why two: with the
    // conditions inside the main loop the compiler merged the rotating register sets with v_mov copies at the joins, and a
    // copy (or a zero-fill) of a register whose load is still in flight needs s_waitcnt vmcnt(0) -- the prefetch was waited
    // for at the top of every step, whatever the number of sets (seen in the ISA; round 3).
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, hp_extent((long)(cin - c0) * ics, 0, 4), 0x00020000);
    auto stage_load_fast = [&](int zin, float (&v)[SK]) {   // plane known to lie inside the volume (or clamped into it)
      const int zz = PADMODE == 1 ? min(max(zin, 0), Di - 1) : zin;
      const unsigned zs = (unsigned)((long)zz * Hi * Wi * 4);
#pragma unroll
      for (int k = 0; k < SK; ++k) v[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, soff[k], zs, 0));
    };
    auto stage_load = [&](int zin, float (&v)[SK]) {
      int zz = zin;
      bool zok = (unsigned)zz < (unsigned)Di;
      if (PADMODE == 1) zz = min(max(zz, 0), Di - 1), zok = true;
      if (zok) {  // workgroup-uniform
        stage_load_fast(zz, v);
      } else {
#pragma unroll
        for (int k = 0; k < SK; ++k) v[k] = 0.f;
      }
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
    // What the epilogue of a step adds to its accumulators -- the bias (first sweep) or the partial sum already in y (later
    // sweeps), and the residual (last sweep) -- is REQUESTED before the step's plane prefetch and used after its multiply
    // phase: one memory round trip hidden behind the MFMAs instead of eight exposed ones (a load per output row and channel,
    // each waited for on the spot), and the wait for it leaves the younger plane prefetches in flight.
    const bool last = c0 + 4 >= cin;
    const bool need_a = c0 != 0, need_r = !PLAIN && last && res != nullptr;   // sweep constants
    // `ok` of an output row / channel of a step and its scalar offset
    auto out_ok = [&](int z, bool valid, int r, int i) { return valid && y0 + 2 * wv + r < Ho && cog * 4 + i < cout; };
    auto out_off = [&](int z, int r, int i) { return (unsigned)(((long)i * ocs + ((long)z * Ho + (y0 + 2 * wv + r)) * Wo) * 4); };
    // steady-state form: no step / row conditions beyond the loop-invariant ones, and NO write to ev on the paths that do not
    // load (a v_mov into a register some other path loads into costs an s_waitcnt vmcnt(0) at the join)
    const unsigned y_rec = hp_extent((long)(cout - cog * 4) * ocs, 0, 4);   // the rest of sample b's output channels
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(y + ybase), 0, y_rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)((res ? res : y) + ybase), 0, y_rec, 0x00020000);
    auto epi_load_fast = [&](int z, EpiVals& ev) {
      if (need_a) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i)   // rows past Ho / channels past cout: a harmless in-range read of memory this workgroup owns or zero
            ev.a[r][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, out_ok(z, true, r, i) ? xoff : 0x80000000u, out_ok(z, true, r, i) ? out_off(z, r, i) : 0u, 0));
      }
      if (need_r) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            ev.r[r][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, out_ok(z, true, r, i) ? xoff : 0x80000000u, out_ok(z, true, r, i) ? out_off(z, r, i) : 0u, 0));
      }
    };
    // The epilogue is written on 4-channel VECTORS (packed fp32 adds / fmas: two instructions per four values) and carries
    // only what the launch needs (PLAIN: no residual, no activation; statistics unmasked when the tile lies inside the row):
    // next to fp32 MFMAs every vector-ALU instruction costs ~6 cycles of matrix-pipe time, and the scalar form -- add,
    // residual select, leaky compare / multiply / two selects, row mask, square, accumulate, per element -- was 150 of them per
    // step against 216 MFMAs (SQ_INSTS_VALU - SQ_INSTS_MFMA, round 3).
    const f32x4 bv4 = {bv[0], bv[1], bv[2], bv[3]};
    const bool xfull = x0 + WG_TX <= Wo;   // scalar
    auto epilogue_fast = [&](int z, const f32x4 (&acc)[2], const EpiVals& ev) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (y0 + 2 * wv + r < Ho) {   // loop-invariant
          f32x4 v = acc[r];
          if (need_a) v += (f32x4){ev.a[r][0], ev.a[r][1], ev.a[r][2], ev.a[r][3]};
          else v += bv4;
          if (last) {
            if constexpr (!PLAIN) {
              if (need_r) v += (f32x4){ev.r[r][0], ev.r[r][1], ev.r[r][2], ev.r[r][3]};
              if (slope != 1.0f) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
              }
            }
            if (stats) {   // (channels past cout hold exact zeros: zero weights, zero bias)
              if (xfull) {
                st1v += v;
                st2v += v * v;
              } else {
                const f32x4 vs = {xin ? v[0] : 0.f, xin ? v[1] : 0.f, xin ? v[2] : 0.f, xin ? v[3] : 0.f};
                st1v += vs;
                st2v += vs * vs;
              }
            }
          }
          // (`__builtin_bit_cast` applied directly to a vector ELEMENT yields element 0 on this compiler: go through a scalar)
          const float vo[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (cog * 4 + i < cout)   // loop-invariant
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vo[i]), yrs, xoff, out_off(z, r, i), 0);
        }
      }
    };
    auto epi_load = [&](int z, EpiVals& ev) {   // generic form: called for steps inside the chunk only
      epi_load_fast(z, ev);
    };
    // one output plane: multiply against the three ring planes, store
    auto mma = [&](int z, f32x4 (&acc)[2]) {
      const float* p0 = ring + ((z - pad) & 3) * WG_PLANE + lbase;
      const float* p1 = ring + ((z - pad + 1) & 3) * WG_PLANE + lbase;
      const float* p2 = ring + ((z - pad + 2) & 3) * WG_PLANE + lbase;
      // (six accumulation chains -- row x dz -- instead of two were measured: 20 % slower)
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
    };
    // lane l holds out[co0 + i][row][x0 + l] in acc[row][i].  Buffer stores / loads: the lane part of the address is fixed
    // for the whole kernel (x position; beyond the row: out of range = dropped / zero), row, plane and channel are scalar.
    // Two planes are in flight at any time (register sets A / B): with a single plane the kernel was bound by the bytes it
    // kept outstanding (~22 KB per CU against a ~2 us loaded memory latency), not by the matrix pipe or LDS.
    // S register sets rotate: the plane parked after step z (ring slot of plane z - pad - 1, last read one barrier ago) was
    // requested S - 1 steps earlier, so S - 1 planes are in flight per workgroup at any time
    float nr[S][SK];
#pragma unroll
    for (int i = 0; i < S - 1; ++i) stage_load(zb + i - pad + 3, nr[i]);   // (generic form: a tiny volume ends here already)
    // steady state: whole groups of S steps whose prefetched planes all lie inside the volume -- no per-step condition at all
    int z = zb;
    for (; z + S <= ze && (PADMODE == 1 || z + 2 * S + 1 - pad < Di); z += S) {
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) {
        const int zz = z + s_;
        EpiVals ev;
        f32x4 acc[2];
        epi_load_fast(zz, ev);
        stage_load_fast(zz + S - 1 - pad + 3, nr[(s_ + S - 1) % S]);
        __builtin_amdgcn_sched_barrier(0);  // keep the fetches ahead of the multiply phase (the scheduler would sink them)
        mma(zz, acc);
        epilogue_fast(zz, acc, ev);
        stage_store(zz - pad + 3, nr[s_]);
        __syncthreads();
      }
    }
    // the last steps of a chunk, generic form (planes past the volume are zero-filled, steps past the end skipped)
    for (; z < ze; z += S) {
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) {
        const int zz = z + s_;
        if (zz < ze) {  // workgroup-uniform
          EpiVals ev;
          f32x4 acc[2];
          epi_load_fast(zz, ev);
          if (zz + S < ze) stage_load(zz + S - 1 - pad + 3, nr[(s_ + S - 1) % S]);
          mma(zz, acc);
          epilogue_fast(zz, acc, ev);
          if (zz + 1 < ze) stage_store(zz - pad + 3, nr[s_]);
          __syncthreads();
        }
      }
    }
  }
  if (stats) {
    // per output channel of this group: all 64 lanes of the four waves hold partial sums
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = st1v[i], q = st2v[i];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        a += __shfl_xor(a, o);
        q += __shfl_xor(q, o);
      }
      if (lane == 0) {
        sred[wave][i] = a;
        sred[wave][4 + i] = q;
      }
    }
    __syncthreads();
    if (tid < 8) {
      const int co = cog * 4 + (tid & 3);
      if (co < cout) {
        const float v = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
        atomicAdd(stats + ((long)b * cout + co) * 2 + (tid >> 2), (double)v);
      }
    }
  }
}


// bf16-operand variant of the kernel above (HP_PRECISION_BF16: BASELINE configs[2], "bf16 with fp32 LCT"): tensors stay
// planar fp32 in HBM, operands are rounded to bf16 (nearest even) on their way into LDS, accumulation is fp32.
// v_mfma_f32_4x4x4_16b_bf16 takes FOUR input channels per instruction (K = 4) where the fp32 4x4x1 takes one:
//     D_b[i][j] += sum_{k<4} w[co0+i][c0+k][tap] * x[c0+k][v_{4b+j} + tap]
// so the ring keeps the planes channel-interleaved -- one 8-byte element (4 bf16 channels) per voxel cell -- and a shifted
// operand is ONE ds_read_b64 (64 consecutive lanes = 512 contiguous bytes: conflict-free) per 1024 MACs: a quarter of the
// matrix-core and LDS instructions of the exact kernel.  NQ = channel quads per sweep (4 or 8 input channels; the 8-channel
// ring has the footprint of the fp32 kernel's 4-channel one, so an 8 -> 4 layer is ONE sweep and y is written once); the
// 27 * NQ weight operands of a sweep live in registers.  Roles, tiling, z walk and epilogue are those of k_dconv3_mfma.
using dbf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using dbf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using ds16x4 = __attribute__((ext_vector_type(4))) short;
using df32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ ds16x4 pack_bf16x4(float a, float b, float c, float d) {
  const dbf16x2 lo = __builtin_convertvector((df32x2){a, b}, dbf16x2);
  const dbf16x2 hi = __builtin_convertvector((df32x2){c, d}, dbf16x2);
  return __builtin_bit_cast(ds16x4, (dbf16x4){lo[0], lo[1], hi[0], hi[1]});
}

template <int PADMODE, int NQ, int S, bool PLAIN>
__global__ __launch_bounds__(256, 2) void k_dconv3_bf16(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int cin,
                                                         int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int pad,
                                                         long wsco, long wsci, int flip, int tiles_x, int tiles_y, int zchunk,
                                                         const float* __restrict__ res, double* __restrict__ stats, float slope) {
  constexpr int CELLS = WG_PY * WG_PX;   // voxel cells of a staged plane (10 rows x 68)
  constexpr int SLOT = NQ * CELLS;       // 8-byte elements per ring slot
  constexpr int SKC = (CELLS + 255) / 256;
  __shared__ __attribute__((aligned(16))) ds16x4 ring[4 * SLOT];
  __shared__ __attribute__((aligned(16))) ds16x4 wl[NQ * 4 * 28];  // [quad][co][tap]
  __shared__ float sred[4][8];
  f32x4 st1v = {0.f, 0.f, 0.f, 0.f}, st2v = {0.f, 0.f, 0.f, 0.f};   // per-channel sum / sum of squares of this lane's outputs
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 3;
  int t_ = xcd_slab_tile(blockIdx.x, gridDim.x);
  const int bx = t_ % tiles_x;
  t_ /= tiles_x;
  const int by = t_ % tiles_y;
  const int bz = t_ / tiles_y;
  const int b = blockIdx.y, cog = blockIdx.z;
  const int x0 = bx * WG_TX, y0 = by * WG_TY;
  const int zb = bz * zchunk, ze = min(Do, zb + zchunk);
  const long ics = (long)Di * Hi * Wi, ocs = (long)Do * Ho * Wo;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const bool xin = x0 + lane < Wo;
  constexpr unsigned OOB = 0x80000000u;
  const unsigned xoff = xin ? (unsigned)((x0 + lane) * 4) : OOB;
  const long ybase = ((long)b * cout + cog * 4) * ocs;
  float bv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bv[i] = bias && cog * 4 + i < cout ? bias[cog * 4 + i] : 0.f;

  for (int c0 = 0; c0 < cin; c0 += 4 * NQ) {
    const int nci = min(4 * NQ, cin - c0);
    __syncthreads();  // previous sweep: ring and weight image are free
    for (int e = tid; e < NQ * 4 * 28; e += 256) {
      const int tap = e % 28, co = (e / 28) & 3, q = e / (28 * 4);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (tap < 27 && cog * 4 + co < cout) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (4 * q + k < nci) v[k] = w[(long)(cog * 4 + co) * wsco + (long)(c0 + 4 * q + k) * wsci + (flip ? 26 - tap : tap)];
      }
      wl[e] = pack_bf16x4(v[0], v[1], v[2], v[3]);
    }
    // one descriptor per channel quad (a quad's four planes stay inside the 2 GB a descriptor spans); the per-thread cell
    // offsets are fixed for the sweep, channel and plane go into the scalar offset
    const float* xq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) xq[q] = x + ((long)b * cin + min(c0 + 4 * q, cin - 1)) * ics;
    unsigned soff[SKC];
#pragma unroll
    for (int k = 0; k < SKC; ++k) {
      const int e = tid + 256 * k;
      const int ly = e / WG_PX, lx = e - ly * WG_PX;
      int yy = y0 + ly - pad, xx = x0 + lx - pad;
      bool ok = e < CELLS && lx < 66;
      if (PADMODE == 1) {
        yy = min(max(yy, 0), Hi - 1);
        xx = min(max(xx, 0), Wi - 1);
      } else {
        ok = ok && (unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi;
      }
      soff[k] = ok ? (unsigned)(((long)yy * Wi + xx) * 4) : OOB;
    }
    // steady-state form: plane inside the volume; a channel past cin re-reads the sweep's last channel (its weights are zero)
    __amdgpu_buffer_rsrc_t xrs[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      xrs[q] = __builtin_amdgcn_make_buffer_rsrc((void*)xq[q], 0, hp_extent((long)(cin - min(c0 + 4 * q, cin - 1)) * ics, 0, 4), 0x00020000);
    auto stage_load_fast = [&](int zin, float (&v)[SKC][4 * NQ]) {
      const int zz = PADMODE == 1 ? min(max(zin, 0), Di - 1) : zin;
      const unsigned zp = (unsigned)((long)zz * Hi * Wi * 4);
#pragma unroll
      for (int c = 0; c < 4 * NQ; ++c) {
        const int cc = min(c, nci - 1);   // scalar
        const unsigned zs = zp + (unsigned)((long)(cc & 3) * ics * 4);
#pragma unroll
        for (int k = 0; k < SKC; ++k)
          v[k][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32((cc >> 2) == 0 ? xrs[0] : xrs[NQ - 1], soff[k], zs, 0));
      }
    };
    auto stage_load = [&](int zin, float (&v)[SKC][4 * NQ]) {   // generic form (see k_dconv3_mfma)
      int zz = zin;
      bool zok = (unsigned)zz < (unsigned)Di;
      if (PADMODE == 1) zz = min(max(zz, 0), Di - 1), zok = true;
      if (zok) {  // workgroup-uniform
        stage_load_fast(zz, v);
      } else {
#pragma unroll
        for (int k = 0; k < SKC; ++k)
#pragma unroll
          for (int c = 0; c < 4 * NQ; ++c) v[k][c] = 0.f;
      }
    };
    auto stage_store = [&](int zin, const float (&v)[SKC][4 * NQ]) {
      ds16x4* dst = ring + (zin & 3) * SLOT + tid;
#pragma unroll
      for (int k = 0; k < SKC; ++k)
        if (tid + 256 * k < CELLS) {
#pragma unroll
          for (int q = 0; q < NQ; ++q) dst[q * CELLS + 256 * k] = pack_bf16x4(v[k][4 * q], v[k][4 * q + 1], v[k][4 * q + 2], v[k][4 * q + 3]);
        }
    };
    auto stage = [&](int zin) {
      float v[SKC][4 * NQ];
      stage_load(zin, v);
      stage_store(zin, v);
    };
    if (zb < ze) {
      stage(zb - pad);
      stage(zb - pad + 1);
      stage(zb - pad + 2);
    }
    __syncthreads();
    // the sweep's weight operands: lane 4b+i supplies output channel i
    ds16x4 wr[NQ][27];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int t = 0; t < 27; ++t) wr[q][t] = wl[(q * 4 + sub) * 28 + t];
    const int lbase = (2 * wave) * WG_PX + lane;
    // What the epilogue of a step adds to its accumulators -- the bias (first sweep) or the partial sum already in y (later
    // sweeps), and the residual (last sweep) -- is REQUESTED before the step's plane prefetch and used after its multiply
    // phase: one memory round trip hidden behind the MFMAs instead of eight exposed ones (a load per output row and channel,
    // each waited for on the spot), and the wait for it leaves the younger plane prefetches in flight.
    const bool last = c0 + 4 * NQ >= cin;
    const bool need_a = c0 != 0, need_r = !PLAIN && last && res != nullptr;   // sweep constants
    // `ok` of an output row / channel of a step and its scalar offset
    auto out_ok = [&](int z, bool valid, int r, int i) { return valid && y0 + 2 * wv + r < Ho && cog * 4 + i < cout; };
    auto out_off = [&](int z, int r, int i) { return (unsigned)(((long)i * ocs + ((long)z * Ho + (y0 + 2 * wv + r)) * Wo) * 4); };
    // steady-state form: no step / row conditions beyond the loop-invariant ones, and NO write to ev on the paths that do not
    // load (a v_mov into a register some other path loads into costs an s_waitcnt vmcnt(0) at the join)
    const unsigned y_rec = hp_extent((long)(cout - cog * 4) * ocs, 0, 4);   // the rest of sample b's output channels
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)(y + ybase), 0, y_rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void*)((res ? res : y) + ybase), 0, y_rec, 0x00020000);
    auto epi_load_fast = [&](int z, EpiVals& ev) {
      if (need_a) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i)   // rows past Ho / channels past cout: a harmless in-range read of memory this workgroup owns or zero
            ev.a[r][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, out_ok(z, true, r, i) ? xoff : 0x80000000u, out_ok(z, true, r, i) ? out_off(z, r, i) : 0u, 0));
      }
      if (need_r) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            ev.r[r][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrs, out_ok(z, true, r, i) ? xoff : 0x80000000u, out_ok(z, true, r, i) ? out_off(z, r, i) : 0u, 0));
      }
    };
    // The epilogue is written on 4-channel VECTORS (packed fp32 adds / fmas: two instructions per four values) and carries
    // only what the launch needs (PLAIN: no residual, no activation; statistics unmasked when the tile lies inside the row):
    // next to fp32 MFMAs every vector-ALU instruction costs ~6 cycles of matrix-pipe time, and the scalar form -- add,
    // residual select, leaky compare / multiply / two selects, row mask, square, accumulate, per element -- was 150 of them per
    // step against 216 MFMAs (SQ_INSTS_VALU - SQ_INSTS_MFMA, round 3).
    const f32x4 bv4 = {bv[0], bv[1], bv[2], bv[3]};
    const bool xfull = x0 + WG_TX <= Wo;   // scalar
    auto epilogue_fast = [&](int z, const f32x4 (&acc)[2], const EpiVals& ev) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        if (y0 + 2 * wv + r < Ho) {   // loop-invariant
          f32x4 v = acc[r];
          if (need_a) v += (f32x4){ev.a[r][0], ev.a[r][1], ev.a[r][2], ev.a[r][3]};
          else v += bv4;
          if (last) {
            if constexpr (!PLAIN) {
              if (need_r) v += (f32x4){ev.r[r][0], ev.r[r][1], ev.r[r][2], ev.r[r][3]};
              if (slope != 1.0f) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
              }
            }
            if (stats) {   // (channels past cout hold exact zeros: zero weights, zero bias)
              if (xfull) {
                st1v += v;
                st2v += v * v;
              } else {
                const f32x4 vs = {xin ? v[0] : 0.f, xin ? v[1] : 0.f, xin ? v[2] : 0.f, xin ? v[3] : 0.f};
                st1v += vs;
                st2v += vs * vs;
              }
            }
          }
          // (`__builtin_bit_cast` applied directly to a vector ELEMENT yields element 0 on this compiler: go through a scalar)
          const float vo[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (cog * 4 + i < cout)   // loop-invariant
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vo[i]), yrs, xoff, out_off(z, r, i), 0);
        }
      }
    };
    auto epi_load = [&](int z, EpiVals& ev) {   // generic form: called for steps inside the chunk only
      epi_load_fast(z, ev);
    };
    auto mma = [&](int z, f32x4 (&acc)[2]) {
      const ds16x4* p0 = ring + ((z - pad) & 3) * SLOT + lbase;
      const ds16x4* p1 = ring + ((z - pad + 1) & 3) * SLOT + lbase;
      const ds16x4* p2 = ring + ((z - pad + 2) & 3) * SLOT + lbase;
      acc[0] = acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int dz = 0; dz < 3; ++dz) {
          const ds16x4* pl = (dz == 0 ? p0 : dz == 1 ? p1 : p2) + q * CELLS;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int r = 0; r < 2; ++r)
                acc[r] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wr[q][(dz * 3 + dy) * 3 + dx], pl[(r + dy) * WG_PX + dx], acc[r], 0, 0, 0);
        }
    };
    float nr[S][SKC][4 * NQ];
#pragma unroll
    for (int i = 0; i < S - 1; ++i) stage_load(zb + i - pad + 3, nr[i]);   // (generic form: a tiny volume ends here already)
    // steady state: whole groups of S steps whose prefetched planes all lie inside the volume -- no per-step condition at all
    int z = zb;
    for (; z + S <= ze && (PADMODE == 1 || z + 2 * S + 1 - pad < Di); z += S) {
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) {
        const int zz = z + s_;
        EpiVals ev;
        f32x4 acc[2];
        epi_load_fast(zz, ev);
        stage_load_fast(zz + S - 1 - pad + 3, nr[(s_ + S - 1) % S]);
        __builtin_amdgcn_sched_barrier(0);  // keep the fetches ahead of the multiply phase (the scheduler would sink them)
        mma(zz, acc);
        epilogue_fast(zz, acc, ev);
        stage_store(zz - pad + 3, nr[s_]);
        __syncthreads();
      }
    }
    // the last steps of a chunk, generic form (planes past the volume are zero-filled, steps past the end skipped)
    for (; z < ze; z += S) {
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) {
        const int zz = z + s_;
        if (zz < ze) {  // workgroup-uniform
          EpiVals ev;
          f32x4 acc[2];
          epi_load_fast(zz, ev);
          if (zz + S < ze) stage_load(zz + S - 1 - pad + 3, nr[(s_ + S - 1) % S]);
          mma(zz, acc);
          epilogue_fast(zz, acc, ev);
          if (zz + 1 < ze) stage_store(zz - pad + 3, nr[s_]);
          __syncthreads();
        }
      }
    }
  }
  if (stats) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float a = st1v[i], q = st2v[i];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        a += __shfl_xor(a, o);
        q += __shfl_xor(q, o);
      }
      if (lane == 0) {
        sred[wave][i] = a;
        sred[wave][4 + i] = q;
      }
    }
    __syncthreads();
    if (tid < 8) {
      const int co = cog * 4 + (tid & 3);
      if (co < cout) {
        const float v = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
        atomicAdd(stats + ((long)b * cout + co) * 2 + (tid >> 2), (double)v);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// Forward / data-gradient 3x3x3 convolution on v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD).
//
// The 4x4x1 kernel above needs one shifted ds_read_b32 per MFMA for 256 MACs: with the weights in registers it is
// bound by the LDS read rate (128 B/clk/CU) at about half the matrix rate.  Here one 16x16x4 MFMA takes
//     rows    m = 16 consecutive voxels of an x-run,
//     K       k = 4 input channels of a chunk (single-channel layers: the 3 dy taps),
//     columns n = dz * 4 + co   (3 z-taps x 4 output channels = 12 of 16 columns),
// i.e. 768 useful MACs per 256-byte LDS read (3x the 4x4x1 kernel) at 3/4 of the matrix rate.  The product of
// INPUT plane z' with the dz-th z-tap belongs to OUTPUT plane z' - dz + pad, so a workgroup that slides along z keeps,
// per tile, the partial sums of the two output planes still waiting for input (q0, q1: lanes n < 4) and folds the
// three column groups of each new product into them with two DPP row shifts:
//     out[z'-2+pad] = q1 + P[dz=2]   (complete: bias / residual / leaky / GroupNorm statistics, 16-byte stores)
//     q1 = q0 + P[dz=1],  q0 = P[dz=0]
// Only ONE input plane is live in LDS at a time (two slots: the next plane is fetched into registers during the
// multiply phase and parked afterwards, one barrier per plane); channel stride and row pitch are = 16 (mod 32) floats
// so that the four 16-lane segments of an operand read fall on disjoint banks.
constexpr int F_TY = 8, F_TX = 64, F_PX = 80, F_ROWS = F_TY + 2;
constexpr int F_CS = F_ROWS * F_PX + 16;  // 816 = 16 (mod 32)
constexpr int F_MAXC = 8;                 // input channels staged per launch

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// CIN1: single input channel, K slots carry the dy taps.  Otherwise cin_here (<= 8) channels in chunks of 4.
template <int PADMODE, bool CIN1>
__global__ __launch_bounds__(256, 2) void k_dconv3_f16(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ res,
                                                        float* __restrict__ y, double* __restrict__ stats, int cin, int c_base,
                                                        int cin_here, int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                        int pad, long wsco, long wsci, int flip, int tiles_x, int tiles_y,
                                                        int zchunk, int accumulate, float slope) {
  constexpr int NCH = CIN1 ? 1 : F_MAXC;
  constexpr int SLOT = NCH * F_CS;
  __shared__ float img[2 * SLOT];
  __shared__ float sred[4][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lk = lane >> 4, ln = lane & 15;  // K slot / column (A: row = ln) of this lane's operands
  int t_ = xcd_slab_tile(blockIdx.x, gridDim.x);
  const int bx = t_ % tiles_x;
  t_ /= tiles_x;
  const int by = t_ % tiles_y;
  const int bz = t_ / tiles_y;
  const int b = blockIdx.y, cog = blockIdx.z;
  const int x0 = bx * F_TX, y0 = by * F_TY;
  const int zb = bz * zchunk, ze = min(Do, zb + zchunk);
  const long ics = (long)Di * Hi * Wi, ocs = (long)Do * Ho * Wo;
  const int nchunk = CIN1 ? 1 : (cin_here + 3) / 4;

  // B operand (weights), in registers for the whole walk: lane (k = lk, n = ln) -> w[co][ci][dz][dy][dx]
  const int wdz = ln >> 2, wco = cog * 4 + (ln & 3);
  const bool wn_ok = ln < 12 && wco < cout;
  float wreg[CIN1 ? 3 : 18];
  if constexpr (CIN1) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int tap = (wdz * 3 + min(lk, 2)) * 3 + dx;
      wreg[dx] = (wn_ok && lk < 3) ? w[(long)wco * wsco + (long)c_base * wsci + (flip ? 26 - tap : tap)] : 0.f;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ci = 4 * j + lk;
        const int tap = wdz * 9 + t;
        wreg[j * 9 + t] = (wn_ok && ci < cin_here) ? w[(long)wco * wsco + (long)(c_base + ci) * wsci + (flip ? 26 - tap : tap)] : 0.f;
      }
  }

  // staging map: element e = tid + 256 k (k < 3) of ONE channel's [10 rows][66 cols] halo image; the same three
  // (row, column) slots serve every channel, so the tables cost 6 registers whatever the channel count
  constexpr int PER_CH = F_ROWS * 66;
  constexpr int SKC = (PER_CH + 255) / 256;  // 3
  constexpr int SK = NCH * SKC;
  int soff[SKC], loff[SKC];
  unsigned smask = 0;
  const float* xb = x + ((long)b * cin + c_base) * ics;
  const int nstage = CIN1 ? 1 : cin_here;
#pragma unroll
  for (int k = 0; k < SKC; ++k) {
    const int e = tid + 256 * k;
    const int ly = e / 66, lx = e - ly * 66;
    int yy = y0 + ly - pad, xx = x0 + lx - pad;
    bool ok = e < PER_CH;
    if (PADMODE == 1) {
      yy = min(max(yy, 0), Hi - 1);
      xx = min(max(xx, 0), Wi - 1);
    } else {
      ok = ok && (unsigned)yy < (unsigned)Hi && (unsigned)xx < (unsigned)Wi;
    }
    soff[k] = ok ? (int)((long)yy * Wi + xx) : 0;
    loff[k] = e < PER_CH ? ly * F_PX + lx : -1;
    smask |= ok ? (1u << k) : 0u;
  }
  auto stage_load = [&](int zin, float (&v)[SK]) {
    int zz = zin;
    bool zok = (unsigned)zz < (unsigned)Di;
    if (PADMODE == 1) zz = min(max(zz, 0), Di - 1), zok = true;
    const float* src = xb + (long)zz * Hi * Wi;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int k = 0; k < SKC; ++k)
        v[c * SKC + k] = (zok && c < nstage && ((smask >> k) & 1u)) ? src[(long)c * ics + soff[k]] : 0.f;
  };
  auto stage_store = [&](int slot, const float (&v)[SK]) {
    float* dst = img + slot * SLOT;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int k = 0; k < SKC; ++k)
        if (loff[k] >= 0) dst[c * F_CS + loff[k]] = v[c * SKC + k];
  };
  // columns 66..79 and the 16-float tail of a channel are never read with a non-zero weight except through
  // shifted reads of tiles at the right edge (x offsets <= 65): nothing to clear.

  f32x4 q0[8], q1[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) q0[t] = q1[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float s1 = 0.f, s2 = 0.f;

  const int zfirst = zb - pad, zlast = ze - 1 - pad + 2;
  if (zb < ze) {
    float v0[SK];
    stage_load(zfirst, v0);
    stage_store(0, v0);
  }
  __syncthreads();
  // this lane's operand base inside a slot
  const int abase = CIN1 ? (2 * wave + min(lk, 2)) * F_PX + ln : lk * F_CS + (2 * wave) * F_PX + ln;
  const bool vec_ok = (Wo & 3) == 0;
  const int oco = cog * 4 + (ln & 3);
  const bool st_lane = ln < 4 && oco < cout;
  const float bv = (st_lane && bias && !accumulate) ? bias[oco] : 0.f;
  for (int zp = zfirst, it = 0; zb < ze && zp <= zlast; ++zp, ++it) {
    float nxt[SK];
    if (zp < zlast) stage_load(zp + 1, nxt);
    __builtin_amdgcn_sched_barrier(0);
    const float* a0 = img + (it & 1) * SLOT + abase;
    f32x4 P[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) P[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (CIN1) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int t = 0; t < 8; ++t)
          P[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[(t >> 2) * F_PX + (t & 3) * 16 + dx], wreg[dx], P[t], 0, 0, 0);
    } else {
      for (int j = 0; j < nchunk; ++j) {
        const float* aj = a0 + j * 4 * F_CS;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float wv = j ? wreg[9 + dy * 3 + dx] : wreg[dy * 3 + dx];
#pragma unroll
            for (int t = 0; t < 8; ++t)
              P[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[((t >> 2) + dy) * F_PX + (t & 3) * 16 + dx], wv, P[t], 0, 0, 0);
          }
      }
    }
    const int zo = zp - 2 + pad;  // the output plane completed by this input plane
    const bool zst = zo >= zb && zo < ze;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x4 d;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float p = P[t][i];
        d[i] = q1[t][i] + dpp_mov<0x108>(p);        // row_shl:8 -> columns dz = 2
        q1[t][i] = q0[t][i] + dpp_mov<0x104>(p);    // row_shl:4 -> columns dz = 1
        q0[t][i] = p;
      }
      const int oy = y0 + 2 * wave + (t >> 2), ox = x0 + (t & 3) * 16 + 4 * lk;
      if (zst && st_lane && oy < Ho && ox < Wo) {
        const long o = ((long)b * cout + oco) * ocs + ((long)zo * Ho + oy) * Wo + ox;
        if (vec_ok) {  // Wo % 4 == 0: the four voxels are inside the row and 16-byte aligned
          float4 r4 = make_float4(d[0] + bv, d[1] + bv, d[2] + bv, d[3] + bv);
          if (accumulate) {
            const float4 pv = *reinterpret_cast<const float4*>(y + o);
            r4.x += pv.x; r4.y += pv.y; r4.z += pv.z; r4.w += pv.w;
          }
          if (res) {
            const float4 rv = *reinterpret_cast<const float4*>(res + o);
            r4.x += rv.x; r4.y += rv.y; r4.z += rv.z; r4.w += rv.w;
          }
          if (slope != 1.0f) {
            r4.x = r4.x > 0.f ? r4.x : r4.x * slope; r4.y = r4.y > 0.f ? r4.y : r4.y * slope;
            r4.z = r4.z > 0.f ? r4.z : r4.z * slope; r4.w = r4.w > 0.f ? r4.w : r4.w * slope;
          }
          s1 += (r4.x + r4.y) + (r4.z + r4.w);
          s2 += (r4.x * r4.x + r4.y * r4.y) + (r4.z * r4.z + r4.w * r4.w);
          *reinterpret_cast<float4*>(y + o) = r4;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (ox + i < Wo) {
              float r = d[i] + bv;
              if (accumulate) r += y[o + i];
              if (res) r += res[o + i];
              if (slope != 1.0f) r = r > 0.f ? r : r * slope;
              s1 += r;
              s2 += r * r;
              y[o + i] = r;
            }
        }
      }
    }
    if (zp < zlast) stage_store((it + 1) & 1, nxt);
    __syncthreads();
  }
  if (stats) {
    // per output channel: lanes (ln = co, any lk) of all four waves
    s1 += __shfl_xor(s1, 16);
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 16);
    s2 += __shfl_xor(s2, 32);
    if (lane < 4) {
      sred[wave][lane] = s1;
      sred[wave][4 + lane] = s2;
    }
    __syncthreads();
    if (tid < 8) {
      const int co = cog * 4 + (tid & 3);
      if (co < cout) {
        const float v = (sred[0][tid] + sred[1][tid]) + (sred[2][tid] + sred[3][tid]);
        atomicAdd(stats + ((long)b * cout + co) * 2 + (tid >> 2), (double)v);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------
// Single-channel 3x3x3 stencil (FeatureExtraction: six 1 -> 1 convolutions per pass, forward and data gradient).
// 54 FLOP per 8 bytes is far below the machine balance, and on the matrix-core kernels above a 1 -> 1 layer pays the
// whole per-plane staging / epilogue overhead for a quarter of the MFMAs (0.8 ms per layer at 256 x 256 x 1024 against
// 0.11 ms of HBM time).  Here a thread owns four consecutive x outputs of one row and slides along z with the
// 3 planes x 3 rows x 6 columns it needs in registers: one new plane = 18 loads (served by L1/L2: neighbouring
// threads share all but their own four columns) for 108 FMAs; weights are wave-uniform scalars.  Same generic
// interface as run_dconv (pad 1 forward, pad 2 for the replicate-padding gradient on the halo domain, flipped taps).
// FOLD (with PADMODE 0, pad 1, flip 1): the data gradient of a REPLICATE-padded convolution in one pass.  The adjoint of
// "clamp the read position" folds the weight of every tap that pointed outside the volume onto the border voxel's own
// position, independently per axis: at coordinate 0 the centre weight of that axis gains the weight of index 2 (flipped
// order), at the last coordinate that of index 0 -- the zero-padded correlation with position-dependent weights on the six
// faces.  y rows are fixed per thread (weights folded once), z planes per step (centre-plane weights summed in a uniform
// branch), x ends are two corrections per plane for the lanes that own x = 0 / x = W - 1.  Replaces the correlation on the
// (D+2)(H+2)(W+2) halo domain + k_fold_replicate (a second pass over a larger volume).
template <int PADMODE, bool FOLD = false, bool ASYNC = false>
__global__ __launch_bounds__(256) void k_stencil_c1(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, const float* __restrict__ res,
                                                    float* __restrict__ y, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                    int pad, int flip, int zchunk, float slope) {
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int ox = (blockIdx.x * 64 + tx) * 4, oy = blockIdx.y * 4 + ty;
  const int nzc = (Do + zchunk - 1) / zchunk;
  const int b = blockIdx.z / nzc, zb = (blockIdx.z % nzc) * zchunk, ze = min(Do, zb + zchunk);
  if (oy >= Ho) return;  // a wave is one row: it leaves as a whole (lanes past Wo stay: they feed their neighbours)
  const bool live = ox < Wo;
  float wt[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) wt[t] = w[flip ? 26 - t : t];
  float lo_f = 0.f, hi_f[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (FOLD) {
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float lo = oy == 0 ? wt[dz * 9 + 6 + dx] : 0.f, hi = oy == Ho - 1 ? wt[dz * 9 + dx] : 0.f;
        wt[dz * 9 + 3 + dx] += lo + hi;
      }
    lo_f = ox == 0 ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) hi_f[i] = ox + i == Wo - 1 ? 1.f : 0.f;
  }
  const float bv = bias ? bias[0] : 0.f;
  const float* xb = x + (long)b * Di * Hi * Wi;
  // row / column addressing of the 3 x 6 window, fixed for the whole walk
  int roff[3], coff[6];
  bool rok[3], cok[6];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    int yy = oy - pad + r;
    rok[r] = PADMODE == 1 || (unsigned)yy < (unsigned)Hi;
    yy = min(max(yy, 0), Hi - 1);
    roff[r] = yy * Wi;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    int xx = ox - pad + c;
    cok[c] = PADMODE == 1 || (unsigned)xx < (unsigned)Wi;
    coff[c] = min(max(xx, 0), Wi - 1);
  }
  // Rows whose length is a multiple of 4: one aligned 16-byte load per row and lane (columns ox .. ox+3); the two
  // window columns outside it come from the neighbouring lanes (whole-wave DPP shifts), only lanes 0 / 63 load them.
  const bool vload = (Wi & 3) == 0 && (pad == 1 || pad == 2);
  const bool own_ok = ox + 3 < Wi;
  auto load_plane = [&](int zin, float (&v)[3][6]) {
    bool zok = PADMODE == 1 || (unsigned)zin < (unsigned)Di;
    const float* src = xb + (long)min(max(zin, 0), Di - 1) * Hi * Wi;
    if (vload) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const bool ok = zok && rok[r];
        const float4 own = (ok && own_ok) ? *reinterpret_cast<const float4*>(src + roff[r] + ox) : make_float4(0.f, 0.f, 0.f, 0.f);
        float lz = dpp_mov<0x138>(own.z), lw = dpp_mov<0x138>(own.w);  // wave_shr:1 -> lane i reads lane i-1
        float rx = dpp_mov<0x130>(own.x);                               // wave_shl:1 -> lane i reads lane i+1
        if (tx == 0) {
          lw = (ok && (PADMODE == 1 || ox >= 1)) ? src[roff[r] + max(ox - 1, 0)] : 0.f;
          lz = (ok && (PADMODE == 1 || ox >= 2)) ? src[roff[r] + max(ox - 2, 0)] : 0.f;
        }
        if (tx == 63 || ox + 4 >= Wi) rx = (ok && (PADMODE == 1 || ox + 4 < Wi)) ? src[roff[r] + min(ox + 4, Wi - 1)] : 0.f;
        if (pad == 1) {
          v[r][0] = lw; v[r][1] = own.x; v[r][2] = own.y; v[r][3] = own.z; v[r][4] = own.w; v[r][5] = rx;
        } else {
          v[r][0] = lz; v[r][1] = lw; v[r][2] = own.x; v[r][3] = own.y; v[r][4] = own.z; v[r][5] = own.w;
        }
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c) v[r][c] = (zok && rok[r] && cok[c]) ? src[roff[r] + coff[c]] : 0.f;
  };
  const bool vec = (Wo & 3) == 0;
  auto emit = [&](int z, const float (&p0)[3][6], const float (&p1)[3][6], const float (&p2)[3][6]) {
    float o[4] = {bv, bv, bv, bv};
    if constexpr (FOLD) {
      float wc[9];   // centre-plane weights: the plane that would lie outside folds onto it (z is uniform)
#pragma unroll
      for (int t = 0; t < 9; ++t) wc[t] = wt[9 + t] + (z == 0 ? wt[18 + t] : 0.f) + (z == Do - 1 ? wt[t] : 0.f);
      auto plane = [&](const float (&pp)[3][6], const float* w9) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = fmaf(w9[dy * 3 + dx], pp[dy][i + dx], o[i]);
        // x ends: centre position gains the weight of the tap that pointed outside (index 2 at x = 0, index 0 at x = W - 1)
        float elo = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) elo = fmaf(w9[dy * 3 + 2], pp[dy][1], elo);
        o[0] = fmaf(lo_f, elo, o[0]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (vec && i < 3) continue;   // rows that are whole quads end in a lane's last output
          float ehi = 0.f;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) ehi = fmaf(w9[dy * 3], pp[dy][i + 1], ehi);
          o[i] = fmaf(hi_f[i], ehi, o[i]);
        }
      };
      plane(p0, wt);
      plane(p1, wc);
      plane(p2, wt + 18);
    } else {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] = fmaf(wt[dy * 3 + dx], p0[dy][i + dx], o[i]);
            o[i] = fmaf(wt[9 + dy * 3 + dx], p1[dy][i + dx], o[i]);
            o[i] = fmaf(wt[18 + dy * 3 + dx], p2[dy][i + dx], o[i]);
          }
    }
    const long ob = (((long)b * Do + z) * Ho + oy) * Wo + ox;
    if (!live) return;
    if (vec) {
      float4 r4 = make_float4(o[0], o[1], o[2], o[3]);
      if (res) {
        const float4 rv = *reinterpret_cast<const float4*>(res + ob);
        r4.x += rv.x; r4.y += rv.y; r4.z += rv.z; r4.w += rv.w;
      }
      if (slope != 1.0f) {
        r4.x = r4.x > 0.f ? r4.x : r4.x * slope; r4.y = r4.y > 0.f ? r4.y : r4.y * slope;
        r4.z = r4.z > 0.f ? r4.z : r4.z * slope; r4.w = r4.w > 0.f ? r4.w : r4.w * slope;
      }
      *reinterpret_cast<float4*>(y + ob) = r4;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (ox + i < Wo) {
          float r = o[i] + (res ? res[ob + i] : 0.f);
          if (slope != 1.0f) r = r > 0.f ? r : r * slope;
          y[ob + i] = r;
        }
    }
  };
  // Five plane buffers in rotation: three under the stencil, two in flight.  With the next plane requested only when the
  // current one was done, a wave kept 3 KB outstanding and the kernel ran at the memory LATENCY (1.7 TB/s).
  float p0[3][6], p1[3][6], p2[3][6], p3[3][6], p4[3][6];
  if constexpr (ASYNC) {   // the launcher's promise: rows of whole quads (Wi % 4 == 0) and pad == 1
    // Asynchronous form (rows of whole quads, pad 1: every FeatureExtraction layer).  load_plane above post-processes a plane
    // (lane exchange, border selects) right after requesting it, i.e. it WAITS for it, and its conditional loads / zero
    // fills make the compiler copy plane registers at the joins (s_waitcnt vmcnt(0) again): the "two planes in flight" never
    // were.  Here a request is three unconditional buffer loads per row -- the lane's own quad, and one word each for the
    // first / last lane of the row, every other lane carrying an out-of-range offset (returns 0, moves nothing) -- into the
    // RAW image {own.x .. own.w, left, right}; the image is turned into the 6-column window (finalize) two steps later,
    // right before the plane's first use.  Rows outside the volume are read clamped and their WEIGHTS zeroed (same
    // products); planes outside it go through a descriptor of zero records (chunk ends only, generic loop below).
    constexpr unsigned OOB = 0x80000000u;
    const long plane_elems = (long)Hi * Wi;
    const bool first = tx == 0, lastl = tx == 63 || ox + 4 >= Wi;
    const unsigned vo_own = own_ok ? (unsigned)(ox * 4) : OOB;
    const unsigned vo_l = (first && (PADMODE == 1 || ox >= 1)) ? (unsigned)(max(ox - 1, 0) * 4) : OOB;
    const unsigned vo_r = (lastl && (PADMODE == 1 || ox + 4 < Wi)) ? (unsigned)(min(ox + 4, Wi - 1) * 4) : OOB;
    if (PADMODE == 0) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
        if (!rok[r]) {   // wave-uniform: the row lies outside the volume -> its taps contribute nothing
#pragma unroll
          for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) wt[dz * 9 + r * 3 + dx] = 0.f;
        }
    }
    const int ro0 = __builtin_amdgcn_readfirstlane(roff[0]), ro1 = __builtin_amdgcn_readfirstlane(roff[1]),
              ro2 = __builtin_amdgcn_readfirstlane(roff[2]);   // a wave is one row: scalar
    auto request = [&](int zin, bool zok, float (&v)[3][6]) {   // zok: scalar
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, zok ? hp_extent((long)Di * plane_elems, 0, 4) : 0u, 0x00020000);
      const long zp = (long)min(max(zin, 0), Di - 1) * plane_elems;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const unsigned so = (unsigned)((zp + (r == 0 ? ro0 : r == 1 ? ro1 : ro2)) * 4);
        const float4 own = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo_own, so, 0));
        v[r][0] = own.x; v[r][1] = own.y; v[r][2] = own.z; v[r][3] = own.w;
        v[r][4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo_l, so, 0));
        v[r][5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo_r, so, 0));
      }
    };
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, hp_extent((long)Di * plane_elems, 0, 4), 0x00020000);
    auto request_fast = [&](int zin, float (&v)[3][6]) {   // plane inside the volume (PADMODE 1: clamped into it)
      const long zp = (long)(PADMODE == 1 ? min(max(zin, 0), Di - 1) : zin) * plane_elems;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const unsigned so = (unsigned)((zp + (r == 0 ? ro0 : r == 1 ? ro1 : ro2)) * 4);
        const float4 own = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo_own, so, 0));
        v[r][0] = own.x; v[r][1] = own.y; v[r][2] = own.z; v[r][3] = own.w;
        v[r][4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo_l, so, 0));
        v[r][5] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo_r, so, 0));
      }
    };
    auto finalize = [&](float (&v)[3][6]) {   // raw image -> window columns ox-1 .. ox+4
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const float ox_ = v[r][0], oy_ = v[r][1], oz_ = v[r][2], ow_ = v[r][3];
        const float lw = dpp_mov<0x138>(ow_), rx = dpp_mov<0x130>(ox_);   // lane i reads lane i-1 / i+1
        const float l = first ? v[r][4] : lw, rr = lastl ? v[r][5] : rx;
        v[r][0] = l; v[r][1] = ox_; v[r][2] = oy_; v[r][3] = oz_; v[r][4] = ow_; v[r][5] = rr;
      }
    };
    auto zin_ok = [&](int zin) { return PADMODE == 1 || (unsigned)zin < (unsigned)Di; };
    request(zb - 1, zin_ok(zb - 1), p0);
    request(zb, zin_ok(zb), p1);
    request(zb + 1, zin_ok(zb + 1), p2);
    request(zb + 2, zin_ok(zb + 2), p3);
    finalize(p0);
    finalize(p1);
    int z = zb;
    // steady state: groups of five steps inside the chunk whose requested planes (up to z + 7) lie inside the volume
    for (; z + 5 <= ze && (PADMODE == 1 || z + 7 < Di); z += 5) {
      request_fast(z + 3, p4); finalize(p2); emit(z, p0, p1, p2);
      request_fast(z + 4, p0); finalize(p3); emit(z + 1, p1, p2, p3);
      request_fast(z + 5, p1); finalize(p4); emit(z + 2, p2, p3, p4);
      request_fast(z + 6, p2); finalize(p0); emit(z + 3, p3, p4, p0);
      request_fast(z + 7, p3); finalize(p1); emit(z + 4, p4, p0, p1);
    }
    for (; z < ze; z += 5) {   // chunk end, generic form
      request(z + 3, zin_ok(z + 3), p4); finalize(p2); emit(z, p0, p1, p2);
      if (z + 1 < ze) { request(z + 4, zin_ok(z + 4), p0); finalize(p3); emit(z + 1, p1, p2, p3); }
      if (z + 2 < ze) { request(z + 5, zin_ok(z + 5), p1); finalize(p4); emit(z + 2, p2, p3, p4); }
      if (z + 3 < ze) { request(z + 6, zin_ok(z + 6), p2); finalize(p0); emit(z + 3, p3, p4, p0); }
      if (z + 4 < ze) { request(z + 7, zin_ok(z + 7), p3); finalize(p1); emit(z + 4, p4, p0, p1); }
    }
    return;
  }
  load_plane(zb - pad, p0);
  load_plane(zb - pad + 1, p1);
  load_plane(zb - pad + 2, p2);
  load_plane(zb - pad + 3, p3);
  for (int z = zb; z < ze; z += 5) {
    load_plane(z - pad + 4, p4);
    emit(z, p0, p1, p2);
    if (z + 1 < ze) {
      load_plane(z - pad + 5, p0);
      emit(z + 1, p1, p2, p3);
    }
    if (z + 2 < ze) {
      load_plane(z - pad + 6, p1);
      emit(z + 2, p2, p3, p4);
    }
    if (z + 3 < ze) {
      load_plane(z - pad + 7, p2);
      emit(z + 3, p3, p4, p0);
    }
    if (z + 4 < ze) {
      load_plane(z - pad + 8, p3);
      emit(z + 4, p4, p0, p1);
    }
  }
}

// generic entry: y (B,cout,Do,Ho,Wo) = conv3(x (B,cin,Di,Hi,Wi)) with weight(co,ci,tap) = w[co*wsco + ci*wsci + tap']
// The 16x16x4 kernel is numerically identical but measured 1.5x SLOWER than the 4x4x1 kernel at 256x256x1024 (its
// 12-of-16-column MFMA peaks at 3/4 of the matrix rate and the matrix pipe stays ~40 % busy in both), so it is
// opt-in (HP_DCONV_16X16=1) until its issue pattern is understood.
static bool use_mfma_c1() {  // A/B switch: single-channel layers on the matrix-core kernel instead of the stencil kernel
  static const bool v = [] {
    const char* e = getenv("HP_DCONV_C1_MFMA");
    return e && atoi(e) != 0;
  }();
  return v;
}
static bool stencil_async() {  // A/B switch: HP_STENCIL_ASYNC=0 keeps the single-channel stencil on its synchronous plane loads
  static const bool v = [] {
    const char* e = getenv("HP_STENCIL_ASYNC");
    return !e || atoi(e) != 0;
  }();
  return v;
}
static bool use_f16_dconv() {
  static const bool v = [] {
    const char* e = getenv("HP_DCONV_16X16");
    return e && atoi(e) != 0;
  }();
  return v;
}

// HP_DCONV_XCD_SLAB=0 keeps the plain tile order (A/B runs)
static void xcd_slab_init() {
  static const bool done = [] {
    if (const char* e = getenv("HP_DCONV_XCD_SLAB")) {
      const int v = atoi(e) != 0;
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_xcd_slab), &v, sizeof(int));
    }
    return true;
  }();
  (void)done;
}

static int run_dconv(const float* x, const float* w, const float* bias, const float* res, float* y, double* stats, float slope,
                     int B, int cin, int cout, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int pad, long wsco, long wsci,
                     int flip, int padmode, hipStream_t st, int precision = HP_PRECISION_FP32) {
  xcd_slab_init();
  // the kernels address a 4-channel group of planes through one buffer descriptor (31-bit byte offsets: beyond it loads read
  // zero and stores are dropped)
  // (the single-channel stencil addresses ONE plane set per sample: 4 bytes per voxel inside its descriptor, 2^29 voxels)
  const bool stencil = cin == 1 && cout == 1 && !stats && !use_mfma_c1();
  const long vox_bytes = stencil ? 4 : 16;
  HP_REQUIRE((long)Di * Hi * Wi * vox_bytes < (1l << 31) && (long)Do * Ho * Wo * vox_bytes < (1l << 31),
             "thin-channel convolution: a volume of more than 2^%d voxels per channel is not supported (%d x %d x %d)", stencil ? 29 : 27, Di,
             Hi, Wi);
  const int tiles_x = (Wo + WG_TX - 1) / WG_TX, tiles_y = (Ho + WG_TY - 1) / WG_TY, cog_n = (cout + 3) / 4;
  const long cols = (long)tiles_x * tiles_y * B * cog_n;
  static const long blocks_target = getenv("HP_DCONV_BLOCKS") ? atol(getenv("HP_DCONV_BLOCKS")) : 1536;
  int zsplit = (int)std::max<long>(1, std::min<long>((Do + 7) / 8, (blocks_target + cols - 1) / cols));
  const int zchunk = (Do + zsplit - 1) / zsplit;
  zsplit = (Do + zchunk - 1) / zchunk;
  dim3 grid((unsigned)(tiles_x * tiles_y * zsplit), (unsigned)B, (unsigned)cog_n);
  if (stencil) {
    static const int zc_env = getenv("HP_STENCIL_ZC") ? atoi(getenv("HP_STENCIL_ZC")) : 0;
    const int zc = zc_env > 0 ? zc_env : (Do >= 256 ? 64 : std::max(8, (Do + 3) / 4));
    dim3 g1((unsigned)((Wo + 255) / 256), (unsigned)((Ho + 3) / 4), (unsigned)(B * ((Do + zc - 1) / zc)));
    const bool async = (Wi & 3) == 0 && pad == 1 && stencil_async();
#define HP_ST_LAUNCH(PM, AS) \
  hipLaunchKernelGGL((k_stencil_c1<PM, false, AS>), g1, dim3(256), 0, st, x, w, bias, res, y, Di, Hi, Wi, Do, Ho, Wo, pad, flip, zc, slope)
    if (padmode) {
      if (async) HP_ST_LAUNCH(1, true); else HP_ST_LAUNCH(1, false);
    } else {
      if (async) HP_ST_LAUNCH(0, true); else HP_ST_LAUNCH(0, false);
    }
#undef HP_ST_LAUNCH
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
  }
  if (stats) HP_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * (size_t)B * cout, st));
  const bool plain = !res && slope == 1.0f;   // no residual, no activation: the epilogue instantiation without either
  if (precision == HP_PRECISION_BF16) {
    // 8 input channels per sweep from 8 channels on (same LDS footprint as the exact kernel), 4 below
#define HP_DBF_LAUNCH(PM, NQ, PL)                                                                                           \
  hipLaunchKernelGGL((k_dconv3_bf16<PM, NQ, 2, PL>), grid, dim3(256), 0, st, x, w, bias, y, cin, cout, Di, Hi, Wi, Do, Ho, Wo, pad, wsco, \
                     wsci, flip, tiles_x, tiles_y, zchunk, res, stats, slope)
#define HP_DBF_SETS(PM, NQ)                                          \
  do {                                                               \
    if (plain) HP_DBF_LAUNCH(PM, NQ, true);                          \
    else HP_DBF_LAUNCH(PM, NQ, false);                               \
  } while (0)
    if (cin > 4) {
      if (padmode) HP_DBF_SETS(1, 2); else HP_DBF_SETS(0, 2);
    } else {
      if (padmode) HP_DBF_SETS(1, 1); else HP_DBF_SETS(0, 1);
    }
#undef HP_DBF_SETS
#undef HP_DBF_LAUNCH
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
  }
  if (!use_f16_dconv()) {
#define HP_DMF_LAUNCH(PM, PL)                                                                                               \
  hipLaunchKernelGGL((k_dconv3_mfma<PM, 2, PL>), grid, dim3(256), 0, st, x, w, bias, y, cin, cout, Di, Hi, Wi, Do, Ho, Wo, pad, wsco, \
                     wsci, flip, tiles_x, tiles_y, zchunk, res, stats, slope)
    if (padmode) {
      if (plain) HP_DMF_LAUNCH(1, true); else HP_DMF_LAUNCH(1, false);
    } else {
      if (plain) HP_DMF_LAUNCH(0, true); else HP_DMF_LAUNCH(0, false);
    }
#undef HP_DMF_LAUNCH
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
  }
  static_assert(F_TY == WG_TY && F_TX == WG_TX, "both kernels share the workgroup tiling");
  for (int c_base = 0; c_base < cin; c_base += F_MAXC) {
    const int here = std::min(F_MAXC, cin - c_base);
    const bool last = c_base + here >= cin;
    const int acc = c_base > 0 ? 1 : 0;
    // residual / activation / statistics belong to the complete sum: the last channel block applies them
    const float* r = last ? res : nullptr;
    double* sp = last ? stats : nullptr;
    const float sl = last ? slope : 1.0f;
#define HP_F16_LAUNCH(PM, C1)                                                                                              \
  hipLaunchKernelGGL((k_dconv3_f16<PM, C1>), grid, dim3(256), 0, st, x, w, bias, r, y, sp, cin, c_base, here, cout, Di, Hi, Wi, \
                     Do, Ho, Wo, pad, wsco, wsci, flip, tiles_x, tiles_y, zchunk, acc, sl)
    if (cin == 1) {
      if (padmode) HP_F16_LAUNCH(1, true); else HP_F16_LAUNCH(0, true);
    } else {
      if (padmode) HP_F16_LAUNCH(1, false); else HP_F16_LAUNCH(0, false);
    }
#undef HP_F16_LAUNCH
  }
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
  return run_dconv(x, w, bias, nullptr, y, nullptr, 1.0f, B, cin, cout, D, H, W, D, H, W, 1, (long)cin * 27, 27, 0, replicate_pad, st);
}

extern "C" int hp_dconv3_forward_fused(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                       double* stats, int B, int cin, int cout, int D, int H, int W, int replicate_pad,
                                       float slope, void* stream) {
  HP_REQUIRE(x && w && y && B > 0 && cin > 0 && cout > 0, "hp_dconv3_forward_fused: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("dconv3_fwd", st);
  return run_dconv(x, w, bias, residual, y, stats, slope, B, cin, cout, D, H, W, D, H, W, 1, (long)cin * 27, 27, 0, replicate_pad, st);
}

extern "C" int hp_dconv3_forward_fused_p(const float* x, const float* w, const float* bias, const float* residual, float* y,
                                         double* stats, int B, int cin, int cout, int D, int H, int W, int replicate_pad,
                                         float slope, int precision, void* stream) {
  HP_REQUIRE(x && w && y && B > 0 && cin > 0 && cout > 0, "hp_dconv3_forward_fused_p: bad argument");
  HP_REQUIRE(precision == HP_PRECISION_FP32 || precision == HP_PRECISION_BF16, "hp_dconv3_forward_fused_p: precision must be fp32 or bf16");
  hipStream_t st = (hipStream_t)stream;
  HP_PROF("dconv3_fwd", st);
  return run_dconv(x, w, bias, residual, y, stats, slope, B, cin, cout, D, H, W, D, H, W, 1, (long)cin * 27, 27, 0, replicate_pad, st,
                   cin == 1 && cout == 1 ? HP_PRECISION_FP32 : precision);
}

extern "C" size_t hp_dconv3_backward_data_workspace_bytes(int B, int cin, int D, int H, int W, int replicate_pad) {
  // (single-input-channel layers whose output is single-channel too fold inside the stencil kernel and need none; the query
  // has no cout, so a 1 -> n layer is told the halo volume of its ONE input channel: B * (D+2)(H+2)(W+2) floats)
  return replicate_pad ? sizeof(float) * (size_t)B * cin * (D + 2) * (H + 2) * (W + 2) : 0;
}

extern "C" int hp_dconv3_backward_data_p(const float* gy, const float* w, float* gx, int B, int cin, int cout, int D, int H,
                                         int W, int replicate_pad, int precision, void* workspace, void* stream);

extern "C" int hp_dconv3_backward_data(const float* gy, const float* w, float* gx, int B, int cin, int cout, int D, int H,
                                       int W, int replicate_pad, void* workspace, void* stream) {
  return hp_dconv3_backward_data_p(gy, w, gx, B, cin, cout, D, H, W, replicate_pad, HP_PRECISION_FP32, workspace, stream);
}

extern "C" int hp_dconv3_backward_data_p(const float* gy, const float* w, float* gx, int B, int cin, int cout, int D, int H,
                                         int W, int replicate_pad, int precision, void* workspace, void* stream) {
  HP_REQUIRE(gy && w && gx && B > 0, "hp_dconv3_backward_data: bad argument");
  HP_REQUIRE(precision == HP_PRECISION_FP32 || precision == HP_PRECISION_BF16, "hp_dconv3_backward_data_p: precision must be fp32 or bf16");
  if (cin == 1 && cout == 1) precision = HP_PRECISION_FP32;  // single-channel layers stay on the exact stencil kernel
  hipStream_t st = (hipStream_t)stream;
  // gx[ci] = sum_co corr(gy[co], flipped w[co][ci]): roles of the channel strides swap
  if (!replicate_pad) {
    HP_PROF("dconv3_dgrad", st);
    return run_dconv(gy, w, nullptr, nullptr, gx, nullptr, 1.0f, B, cout, cin, D, H, W, D, H, W, 1, 27, (long)cin * 27, 1, 0, st, precision);
  }
  const bool stencil = cin == 1 && cout == 1 && !use_mfma_c1();
  HP_REQUIRE((long)D * H * W * (stencil ? 4 : 16) < (1l << 31),
             "thin-channel data gradient: a volume of more than 2^%d voxels per channel is not supported (%d x %d x %d)", stencil ? 29 : 27, D,
             H, W);
  if (stencil) {
    // single-channel layers (FeatureExtraction): the replicate fold inside the stencil kernel, one pass over (D, H, W)
    HP_PROF("dconv3_dgrad", st);
    static const int zc_env = getenv("HP_STENCIL_ZC") ? atoi(getenv("HP_STENCIL_ZC")) : 0;
    const int zc = zc_env > 0 ? zc_env : (D >= 256 ? 64 : std::max(8, (D + 3) / 4));
    dim3 g1((unsigned)((W + 255) / 256), (unsigned)((H + 3) / 4), (unsigned)(B * ((D + zc - 1) / zc)));
    if ((W & 3) == 0 && stencil_async())
      hipLaunchKernelGGL((k_stencil_c1<0, true, true>), g1, dim3(256), 0, st, gy, w, (const float*)nullptr, (const float*)nullptr, gx, D, H,
                         W, D, H, W, 1, 1, zc, 1.0f);
    else
      hipLaunchKernelGGL((k_stencil_c1<0, true, false>), g1, dim3(256), 0, st, gy, w, (const float*)nullptr, (const float*)nullptr, gx, D, H,
                         W, D, H, W, 1, 1, zc, 1.0f);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
  }
  HP_REQUIRE(workspace, "hp_dconv3_backward_data: replicate padding needs the workspace");
  float* dpad = (float*)workspace;
  {
    HP_PROF("dconv3_dgrad", st);
    int rc = run_dconv(gy, w, nullptr, nullptr, dpad, nullptr, 1.0f, B, cout, cin, D, H, W, D + 2, H + 2, W + 2, 2, 27, (long)cin * 27, 1, 0, st,
                       precision);
    if (rc) return rc;
  }
  HP_REQUIRE((long)B * cin * D < 65536 && H < 65536, "hp_dconv3_backward_data: volume too large for the fold grid");
  hipLaunchKernelGGL(k_fold_replicate, dim3((unsigned)((W + 255) / 256), (unsigned)H, (unsigned)((long)B * cin * D)), dim3(256), 0,
                     st, dpad, gx, D, H, W);
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
  // z range per workgroup: ~512 workgroups = ONE resident wave of 2 per CU (1024 for the single-channel kernel), at least 8
  // planes each: 2 halo planes are re-read, and a partial tile is written and reduced per workgroup.  Sweep at the U-Net shapes
  // of 1024 x 256 x 256 and 512 x 128 x 128 (gpurun_out/r4/dblocks*.log): 256 / 384 / 512 / 768 / 1024 / 1536 workgroups ->
  // layer sums 6.74 / 6.01 / 5.19 / 5.41 / 5.05 / 5.25 ms and 1.16 / 1.17 / 0.98 / 1.10 / 1.08 / 1.15 ms: whole waves win, and among
  // them the fewest (round 4; 1024 before)
  const long cols = (long)q.tiles_x * q.tiles_y * B * q.cog_n * q.cig_n;
  static const long blocks_env = getenv("HP_DCONV_WBLOCKS") ? atol(getenv("HP_DCONV_WBLOCKS")) : 0;
  const long blocks_target = blocks_env > 0 ? blocks_env : (cin == 1 && cout == 1) ? 1024 : 512;
  q.zsplit = (int)std::max<long>(1, std::min<long>((D + 7) / 8, (blocks_target + cols - 1) / cols));
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

extern "C" int hp_dconv3_backward_weight_p(const float* x, const float* gy, float* dw, float* dbias, int B, int cin,
                                           int cout, int D, int H, int W, int replicate_pad, int precision, void* workspace,
                                           void* stream);

extern "C" int hp_dconv3_backward_weight(const float* x, const float* gy, float* dw, float* dbias, int B, int cin,
                                         int cout, int D, int H, int W, int replicate_pad, void* workspace, void* stream) {
  return hp_dconv3_backward_weight_p(x, gy, dw, dbias, B, cin, cout, D, H, W, replicate_pad, HP_PRECISION_FP32, workspace, stream);
}

extern "C" int hp_dconv3_backward_weight_p(const float* x, const float* gy, float* dw, float* dbias, int B, int cin,
                                           int cout, int D, int H, int W, int replicate_pad, int precision, void* workspace,
                                           void* stream) {
  xcd_slab_init();
  HP_REQUIRE(x && gy && dw && workspace && B > 0, "hp_dconv3_backward_weight: bad argument");
  HP_REQUIRE(precision == HP_PRECISION_FP32 || precision == HP_PRECISION_BF16, "hp_dconv3_backward_weight_p: precision must be fp32 or bf16");
  hipStream_t st = (hipStream_t)stream;
  HP_REQUIRE((long)D * H * W * 16 < (1l << 31),
             "thin-channel weight gradient: a volume of more than 2^27 voxels per channel is not supported (%d x %d x %d)", D, H, W);
  const WgradGeom q = wgrad_geom(B, cin, cout, D, H, W);
  dim3 grid((unsigned)(q.tiles_x * q.tiles_y * q.zsplit), (unsigned)B, (unsigned)(q.cog_n * q.cig_n));
  float* partial = (float*)workspace;
  HP_PROF("dconv3_wgrad", st);
  // bf16 operands: multi-channel layers whose rows allow the 16-byte g loads; everything else stays on the exact kernels
  if (precision == HP_PRECISION_BF16 && cin > 1 && W % 4 == 0 && ((uintptr_t)gy & 15) == 0) {
    if (replicate_pad)
      hipLaunchKernelGGL((k_dconv3_wgrad_bf16<1>), grid, dim3(256), 0, st, x, gy, cin, cout, D, H, W, q.tiles_x, q.tiles_y,
                         q.zchunk, q.cig_n, partial);
    else
      hipLaunchKernelGGL((k_dconv3_wgrad_bf16<0>), grid, dim3(256), 0, st, x, gy, cin, cout, D, H, W, q.tiles_x, q.tiles_y,
                         q.zchunk, q.cig_n, partial);
  } else if (cin == 1) {
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
