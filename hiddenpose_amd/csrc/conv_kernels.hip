// 3-D convolution engine of the pose regressor (posenet3d_50) on gfx950.
//
// All activations inside the regressor are channels-last fp32: X[b][d][h][w][C].
// Every convolution flavour of models/posenet3d_50.py -- 1^3 / 3^3 / 7^3 Conv3d, stride 1
// or 2, ConvTranspose3d(k4,s2,p1), and all their data / weight gradients -- is ONE
// implicit GEMM  out[m][n] = sum_k A[m][k] * W[k][n]  with
//     m = an output voxel of one "class" (a parity class for the stride-2 scatter cases),
//     k = (tap, input channel),  n = output channel,
// computed on the matrix cores with exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: bit-equal to
// an fmaf chain, no reduced precision).  A is never materialised: each 128x32 A tile is
// gathered straight from the channels-last input (128-byte contiguous channel runs per
// voxel) into LDS; W is pre-packed [tap][n][c] so both operands are K-contiguous.
//
// Workgroup = 256 threads = 4 waves; block tile 128(M) x BN(N) x 32(K), each wave owns
// 32x32 MFMA tiles (2x2 for BN=128).  LDS rows are padded to 33 floats so that the
// per-lane fragment reads (32 consecutive rows, fixed k) and the staging writes are
// bank-conflict free.  Global loads of tile t+1 are issued before the MFMAs of tile t.
// Epilogue: optional bias, per-channel sum / sum-of-squares for train-mode BatchNorm
// (fp64 atomics, one pair per column per block), 128-byte coalesced channels-last stores.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "hp_internal.h"

namespace hp {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using s16x4 = __attribute__((ext_vector_type(4))) short;

// four fp32 -> four bf16 (round to nearest even; two v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16x4 to_bf16x4(const float4 v) {
  const bf16x2 lo = __builtin_convertvector((f32x2){v.x, v.y}, bf16x2);
  const bf16x2 hi = __builtin_convertvector((f32x2){v.z, v.w}, bf16x2);
  return (bf16x4){lo[0], lo[1], hi[0], hi[1]};
}

enum IgemmMode { MODE_CONV = 0, MODE_DECONV = 1, MODE_DGRAD_S2K3 = 2, MODE_STEM = 3 };

struct IgemmGeom {
  int mode;
  int B, Di, Hi, Wi, Cin;  // gathered tensor (channels-last); Cin = K extent per tap
  int Do, Ho, Wo, Nout;    // written tensor (channels-last)
  int gd, gh, gw;          // per-class grid of M positions
  int os;                  // written position = g*os + class parity
  int k, s, p, flip;       // MODE_CONV: gathered pos = g*s + (flip ? p - kk : kk - p)
  long M;                  // B*gd*gh*gw
  int kpt;                 // K tiles per tap (Cin/32, or padded taps/32 for the stem)
  int sw, sh, sd;          // log2 of gw, gh, gd when all three are powers of two, else -1
  int tn;                  // > 0: 1-D grid with the N tiles of one M tile adjacent on one XCD (set by the launcher)
  int xh, yh, ah;          // element type of the gathered tensor / written tensor / addend: 0 = fp32, 1 = bf16 (hp_ld4 / hp_st4)
  int wh;                  // packed weights are bf16 (bf16-storage tiles only)
  int geglu;               // 1: GEGLU epilogue (hp_linear_geglu_forward): columns [0, 64) of every 128-column tile are values, [64, 128)
                           //    their gates; Y has Nout / 2 columns and receives value * gelu(gate)
  int slab;                // > 0: M tiles per XCD -- XCD k (workgroup b runs on XCD b % 8) walks M tiles k * slab .. (k + 1) * slab - 1
  long welems;             // elements of the packed weight image this geometry reads (descriptor extents)
  // BNS instantiations (hp_conv3d_backward_data_bnsums): the written tensor is the output gradient dy_a of the BatchNorm unit
  // that produced this convolution's input; its raw output z_a and parameters, and where the two backward sums go
  const float* bn_z;
  const float *bn_mean, *bn_rstd, *bn_gamma, *bn_beta;
  double* bn_sums;         // HP_STATS_SLOTS x 2 x Nout doubles: sum g, sum g * zhat with g = dy_a (.) [y_a > 0]
  int bn_relu;
  const unsigned char* bn_mask;   // unit WITH residual: [y_a > 0] as hp_bn_apply's byte mask (one byte per channel quad) instead of z_a's sign
};

// m -> (b, z, y, x) on the per-class grid; shifts when the grid is a power of two (the usual case)
__device__ __forceinline__ void grid_coords(const IgemmGeom& g, long m, int& b, int& z, int& y, int& x) {
  if (g.sw >= 0) {
    const unsigned int u = (unsigned int)m;  // M < 2^32 is checked on the host for this path
    x = (int)(u & (unsigned)(g.gw - 1));
    y = (int)((u >> g.sw) & (unsigned)(g.gh - 1));
    z = (int)((u >> (g.sw + g.sh)) & (unsigned)(g.gd - 1));
    b = (int)(u >> (g.sw + g.sh + g.sd));
  } else {
    long t = m;
    x = (int)(t % g.gw);
    t /= g.gw;
    y = (int)(t % g.gh);
    t /= g.gh;
    z = (int)(t % g.gd);
    b = (int)(t / g.gd);
  }
}

// Split of four fp32 values into NP bf16 planes: plane 0 = bf16(v), plane p = bf16(v - sum of earlier planes).
// NP = 1 is plain bf16 rounding; NP = 2 keeps 16 significant bits, NP = 3 all 24 (each residual is exact in fp32).
template <int NP>
__device__ __forceinline__ void split_bf16(const float4 v, bf16x4 (&out)[NP]) {
  float4 r = v;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    out[p] = to_bf16x4(r);
    if (p + 1 < NP) {
      r.x -= (float)out[p][0];
      r.y -= (float)out[p][1];
      r.z -= (float)out[p][2];
      r.w -= (float)out[p][3];
    }
  }
}
// Products kept for an NP-plane split of both operands, smallest terms first: every pair (pa, pb) with
// pa + pb < NP.  NP = 1: 1 MFMA, 2: 3 MFMAs (error ~2^-16 per product), 3: 6 MFMAs (error ~2^-24, the fp32 level).
template <int NP>
struct SplitTerms;
template <>
struct SplitTerms<1> {
  static constexpr int N = 1;
  static constexpr int A[1] = {0}, B[1] = {0};
};
template <>
struct SplitTerms<2> {
  static constexpr int N = 3;
  static constexpr int A[3] = {1, 0, 0}, B[3] = {0, 1, 0};
};
template <>
struct SplitTerms<3> {
  static constexpr int N = 6;
  static constexpr int A[6] = {2, 0, 1, 1, 0, 0}, B[6] = {0, 2, 1, 0, 1, 0};
};

constexpr int BM = 128, BK = 32, LDK = BK + 1, CT = 256;
constexpr int LDH = BK + 8;  // bf16 tile row: 80 bytes, so the 16-byte fragment reads of 32 rows spread over all banks

__host__ __device__ inline int class_ntaps(const IgemmGeom& g, int cls) {
  switch (g.mode) {
    case MODE_CONV: return g.k * g.k * g.k;
    case MODE_DECONV: return 8;
    case MODE_DGRAD_S2K3: return (1 + ((cls >> 2) & 1)) * (1 + ((cls >> 1) & 1)) * (1 + (cls & 1));
    default: return 1;
  }
}

// input offset (relative to g*s) and weight slab of tap `t` of class `cls`
__device__ __forceinline__ void tap_info(const IgemmGeom& g, int cls, int t, int& dz, int& dy, int& dx, int& widx) {
  if (g.mode == MODE_CONV) {
    const int kk = g.k;
    // literal divisors for the kernel sizes in use: a runtime integer division is ~40 scalar instructions, and this
    // runs once per tap inside the K loop (every second K tile of a 64-channel 3^3 layer)
    int a, b, c;
    if (kk == 3) {
      a = t / 9, b = (t / 3) % 3, c = t % 3;
    } else if (kk == 4) {
      a = t >> 4, b = (t >> 2) & 3, c = t & 3;
    } else if (kk == 1) {
      a = b = c = 0;
    } else {
      a = t / (kk * kk), b = (t / kk) % kk, c = t % kk;
    }
    dz = g.flip ? g.p - a : a - g.p;
    dy = g.flip ? g.p - b : b - g.p;
    dx = g.flip ? g.p - c : c - g.p;
    widx = t;
  } else if (g.mode == MODE_DECONV) {
    // out[2g+p] = sum_j x[g + p - j] * w[(1-p) + 2j],  j in {0,1}  (k4, s2, p1)
    const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
    const int jd = (t >> 2) & 1, jh = (t >> 1) & 1, jw = t & 1;
    dz = pd - jd;
    dy = ph - jh;
    dx = pw - jw;
    widx = (((1 - pd) + 2 * jd) * 4 + ((1 - ph) + 2 * jh)) * 4 + ((1 - pw) + 2 * jw);
  } else {  // MODE_DGRAD_S2K3: dX[2g+p] = p ? dY[g+1] w[0] + dY[g] w[2] : dY[g] w[1]
    const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
    const int nw = 1 + pw, nh = 1 + ph;
    const int jw = t % nw, jh = (t / nw) % nh, jd = t / (nw * nh);
    const int kd = pd ? 2 * jd : 1, kh = ph ? 2 * jh : 1, kw = pw ? 2 * jw : 1;
    dz = pd ? 1 - jd : 0;
    dy = ph ? 1 - jh : 0;
    dx = pw ? 1 - jw : 0;
    widx = (kd * 3 + kh) * 3 + kw;
  }
}

template <int BN>
struct TileCfg {
  static constexpr int WN = BN >= 64 ? 2 : 1;       // waves along N
  static constexpr int WM = 4 / WN;                 // waves along M
  static constexpr int TM = BM / (WM * 32);         // 32x32 tiles per wave along M
  static constexpr int TN = BN / (WN * 32);
};

// XH: the gathered tensor is bf16 in memory (NP = 1 only): K tiles of 64 channels, so that a thread still moves 16 bytes per
// row and load (8 bf16) -- half the loads, K tiles and barriers per MAC of the fp32-storage path -- and the A tile goes to
// LDS as loaded, without a conversion.
// 128 zero bytes: where a direct-to-LDS load (GL tiles) has to deliver zeros (padding taps, rows past M / Nout)
__device__ __attribute__((aligned(256))) unsigned int g_zero_row[64];

// GL (with XH, bf16 packed weights): both tiles go from memory straight into LDS (global_load_lds_dwordx4: no staging
// registers, no ds_write pass), into two buffers of unpadded 128-byte rows, so that the loads of tile t+1 are in flight
// while tile t is multiplied and one barrier per K tile suffices.  A wave-instruction of such a load writes 1 KB
// linearly (lane l -> base + 16 l = 8 rows x 8 chunks), so the bank spread comes from the SOURCE side: LDS chunk c of row r
// holds memory chunk c ^ ((r >> 1) & 7), and the fragment reads apply the same XOR (16 rows of a ds_read_b128 lane group
// then fall on 16 distinct 16-byte slots).
// BL (exact-fp32 tiles, Cin a multiple of 32): both tiles are fetched with BUFFER loads -- descriptor base = this
// workgroup's lowest gathered address, per-thread byte offset fixed for the whole kernel, the (tap, channel tile)
// displacement in the scalar offset -- so a K tile's eight loads need no vector address arithmetic, and rows that must
// read zeros (padding taps, rows past M / N_out) simply carry an offset beyond the descriptor's range: the hardware
// returns 0.  Why it matters: tools/micro/mfma_coexec.hip -- while one wave streams fp32 MFMAs, vector-ALU, vector-memory
// and LDS-read instructions of the OTHER waves of that SIMD do not issue at all, i.e. every such instruction of the K loop
// is paid in matrix-pipe time whichever wave executes it.
template <int BN, bool STEM, bool STATS, int NP, bool XH = false, bool GL = false, bool BL = false, int BNS = 0>
__global__ __launch_bounds__(CT, (NP == 0 && BN == 128) ? 3 : 1) void k_igemm(const void* __restrict__ Xv, const float* __restrict__ Wp,
                                              const float* __restrict__ bias, void* __restrict__ Y,
                                              double* __restrict__ stats, const void* __restrict__ addend,
                                              const unsigned char* __restrict__ amask, IgemmGeom g) {
  const float* const X = (const float*)Xv;  // fp32 view (stem: always fp32; otherwise only used when g.xh == 0)
  // Runtime element types only in the single-plane bf16 kernels (NP = 1): the exact-fp32 and split kernels keep plain
  // fp32 accesses (the uniform branch per access cost the fp32 headline step 3 %).
  constexpr bool IOH = NP == 1;
  auto ld_x4 = [&](long off) -> float4 {
    if constexpr (IOH) return hp_ld4(Xv, off, g.xh);
    else return *(const float4*)(X + off);
  };
  auto ld_a4 = [&](long off) -> float4 {
    if constexpr (IOH) return hp_ld4(addend, off, g.ah);
    else return *(const float4*)((const float*)addend + off);
  };
  auto ld_a1 = [&](long off) -> float {
    if constexpr (IOH) return hp_ld1(addend, off, g.ah);
    else return ((const float*)addend)[off];
  };
  auto st_y4 = [&](long off, float4 v) {
    if constexpr (IOH) hp_st4(Y, off, v, g.yh);
    else *(float4*)((float*)Y + off) = v;
  };
  auto st_y1 = [&](long off, float v) {
    if constexpr (IOH) hp_st1(Y, off, v, g.yh);
    else ((float*)Y)[off] = v;
  };
  using C = TileCfg<BN>;
  // one LDS arena: A and B tiles during the K loop, then the output staging tile of the epilogue
  constexpr bool BF = NP > 0;  // NP = 0: exact-fp32 MFMA; NP >= 1: bf16 MFMA on NP operand planes
  constexpr int NPL = NP > 0 ? NP : 1;
  static_assert(!XH || (NP == 1 && !STEM), "bf16-storage tiles: single bf16 plane, not the stem");
  static_assert(!GL || (XH && BN >= 64), "direct-to-LDS tiles: bf16 storage, 64 or 128 columns");
  static_assert(!BL || (!XH && !STEM), "buffer-load tiles: fp32 tensors in memory (any arithmetic), not the stem");
  constexpr int BKT = XH ? 64 : BK;      // K extent of a tile
  constexpr int LDX = GL ? BKT : BKT + 8;  // XH: bf16 tile row (144 bytes: 16-byte fragment reads of 16 rows hit 16 distinct slots)
  constexpr int EPL = XH ? 8 : 4;        // elements per thread, row and load
  constexpr int ARENA0 = (BM + BN) * LDK > NP * (BM + BN) * LDH / 2 ? (BM + BN) * LDK : NP * (BM + BN) * LDH / 2;
  constexpr int ARENA1 = XH && (BM + BN) * LDX / 2 > ARENA0 ? (BM + BN) * LDX / 2 : ARENA0;
  constexpr int GLBUF = (BM + BN) * 64;  // bf16 elements per buffer (GL)
  constexpr int ARENA = GL ? GLBUF : ARENA1;  // GL: two buffers of (BM + BN) x 64 bf16 = (BM + BN) * 64 floats
  __shared__ __attribute__((aligned(16))) float smem[ARENA];
  float* const As = smem;
  float* const Bs = smem + BM * LDK;
  // BF: the same arena holds the tiles as NP bf16 planes each (operands split once, on the way into LDS)
  __bf16* const Ah = (__bf16*)smem;
  __bf16* const Bh = Ah + NPL * BM * (XH ? LDX : LDH);
  __shared__ float red[STATS ? 2 * BN * C::WM : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int cls = blockIdx.z;
  // Several N tiles: block b runs on XCD b % 8, so the i-th block of an XCD takes N tile i % tn of M tile
  // (i / tn) * 8 + xcd -- the tn blocks that gather the same A rows run back to back on one XCD and all but the
  // first are served by its L2 (with the N tiles on grid.y they are a whole grid.x apart and re-read HBM).
  // Measured and not adopted: classes adjacent as well (the unequal tap counts of the stride-2 data-gradient
  // classes then unbalance the tail: 3.9 -> 6.6 ms), runs of 2-8 consecutive M tiles per XCD (no change).
  unsigned mt_idx = blockIdx.x, nt_idx = blockIdx.y;
  if (g.tn > 0) {
    const unsigned i = blockIdx.x >> 3;
    nt_idx = i % (unsigned)g.tn;
    mt_idx = g.slab > 0 ? (blockIdx.x & 7u) * (unsigned)g.slab + i / (unsigned)g.tn : (i / (unsigned)g.tn) * 8u + (blockIdx.x & 7u);
    if ((long)mt_idx * BM >= g.M) return;
  } else if (g.slab > 0) {
    mt_idx = (blockIdx.x & 7u) * (unsigned)g.slab + (blockIdx.x >> 3);
    if ((long)mt_idx * BM >= g.M) return;
  }
  const long m0 = (long)mt_idx * BM;
  const int n0 = nt_idx * BN;
  const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;

  // ---- per-thread gather rows: r0 + 32 i
  const int kq0 = tid & 7, r0 = tid >> 3;
  // GL: this thread's LDS chunk kq0 of rows r0 + 32 i is filled from memory chunk kq0 ^ swizzle(row) (same for every i)
  const int kq = GL ? (kq0 ^ ((r0 >> 1) & 7)) : kq0;
  int rb[4], rz[4], ry[4], rx[4];
  bool rv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long m = m0 + r0 + 32 * i;
    rv[i] = m < g.M;
    grid_coords(g, rv[i] ? m : 0, rb[i], rz[i], ry[i], rx[i]);
    rx[i] *= g.s;
    ry[i] *= g.s;
    rz[i] *= g.s;
  }
  const int ntaps = STEM ? 1 : class_ntaps(g, cls);
  const int kpt = XH ? (g.Cin + BKT - 1) / BKT : g.kpt;
  const int KT = ntaps * kpt;

  // Gather addressing without per-tile index arithmetic: element offset = rowoff[i] (this thread's voxel and
  // channel quad, fixed) + a wave-uniform tap offset; whether tap t falls inside the volume for row i is decided
  // once, as bit t of vmask[i] (<= 64 taps per class), and shifted out one bit per tap.
  long rowoff[4], wrow[BN / 32];
  unsigned long long vmask[4];
  bool wvalid[BN / 32];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rowoff[i] = ((((long)rb[i] * g.Di + rz[i]) * g.Hi + ry[i]) * g.Wi + rx[i]) * g.Cin + kq * EPL;
    vmask[i] = 0ull;
  }
  if constexpr (!STEM) {
    for (int t = ntaps - 1; t >= 0; --t) {
      int dz, dy, dx, widx;
      tap_info(g, cls, t, dz, dy, dx, widx);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = rv[i] && (unsigned)(rz[i] + dz) < (unsigned)g.Di && (unsigned)(ry[i] + dy) < (unsigned)g.Hi &&
                        (unsigned)(rx[i] + dx) < (unsigned)g.Wi;
        vmask[i] = (vmask[i] << 1) | (ok ? 1ull : 0ull);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    const int n = n0 + r0 + 32 * i;
    wvalid[i] = n < g.Nout;
    // GEGLU epilogue: column c of the 128-column tile t is the Linear's row 64 t + c (c < 64: value) or hidden + 64 t + c - 64
    // (gate) -- the pairing is done HERE, in the gather's row index, so no reordered copy of the weights exists anywhere
    const int nsrc = g.geglu ? ((n & 64) ? (g.Nout >> 1) + ((n >> 7) << 6) + (n & 63) : ((n >> 7) << 6) + (n & 63)) : n;
    wrow[i] = (long)nsrc * g.Cin + kq * EPL;
  }

  // BL: descriptors (wave-uniform) and per-thread byte offsets
  constexpr unsigned OOB = 0x80000000u;  // = num_records: any offset from here on reads as zero
  long rowbase = 0;   // element offset of the tile's first row: the X descriptor's base (with min_xoff)
  unsigned xrow32[4] = {0u, 0u, 0u, 0u}, xvoff[4] = {OOB, OOB, OOB, OOB}, wvoff[BN / 32];
  long min_xoff = 0;  // smallest tap displacement of this class (elements): the scalar offsets are taken relative to it
  if constexpr (BL) {
    for (int t = 0; t < ntaps; ++t) {
      int dz, dy, dx, widx;
      tap_info(g, cls, t, dz, dy, dx, widx);
      const long xo = (((long)dz * g.Hi + dy) * g.Wi + dx) * g.Cin;
      min_xoff = (t == 0 || xo < min_xoff) ? xo : min_xoff;
    }
    // the tile's rows ascend with m, so its first row has the lowest base address
    int b0, z0, y0, x0;
    grid_coords(g, m0 < g.M ? m0 : 0, b0, z0, y0, x0);
    rowbase = ((((long)b0 * g.Di + z0 * g.s) * g.Hi + y0 * g.s) * g.Wi + x0 * g.s) * g.Cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) xrow32[i] = rv[i] ? (unsigned)((rowoff[i] - rowbase) * 4) : OOB;
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) wvoff[i] = wvalid[i] ? (unsigned)(wrow[i] * 4) : OOB;
  }
  // num_records = what is left of the tensor from the descriptor's base (hp_extent): offsets made of OOB stay out of range
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(X + (rowbase + min_xoff)), 0, hp_extent((long)g.B * g.Di * g.Hi * g.Wi * g.Cin, rowbase + min_xoff, 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)Wp, 0, hp_extent(g.welems, 0, 4), 0x00020000);
  float4 ra[4], rbw[BN / 32];
  uint4 ra8[XH ? 4 : 1];        // XH: 8 bf16 per row as loaded
  float4 rbw2[XH ? BN / 32 : 1];  // XH: second half of the 8 weights per row (fp32 weights)
  uint4 rb8[XH ? BN / 32 : 1];    // XH: 8 bf16 weights per row (bf16 weights, g.wh)
  // running (tap, channel tile) of the NEXT tile to gather: no division in the K loop
  int ld_tap = 0, ld_ci = 0;
  long ld_xoff = 0, ld_woff = 0;  // wave-uniform: tap displacement in X, tap slab in the packed weights
  unsigned ninv_lo[4] = {0u, 0u, 0u, 0u}, ninv_hi[4] = {0u, 0u, 0u, 0u};   // BL: ~vmask, indexed by the tap itself
  if constexpr (BL) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ninv_lo[i] = ~(unsigned)vmask[i];
      ninv_hi[i] = ~(unsigned)(vmask[i] >> 32);
    }
  }
  auto tap_offsets = [&](int t) {
    int dz, dy, dx, widx;
    tap_info(g, cls, t, dz, dy, dx, widx);
    ld_xoff = (((long)dz * g.Hi + dy) * g.Wi + dx) * g.Cin;
    ld_woff = (long)widx * g.Nout * g.Cin;
    if constexpr (BL) {  // rows for which this tap falls outside the volume read zeros: an out-of-range offset
      // bit t of the inverted mask lifted into bit 31 of the (< 2^31) row offset: v_bfe_u32 + v_lshl_or_b32 per row and tap
      // (the 64-bit shift + test + select form was ~20 vector instructions per tap: with two K tiles per tap (64 channels)
      // a tenth of the kernel's matrix-pipe time)
      const unsigned tb = (unsigned)t & 31u;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        xvoff[i] = xrow32[i] | (__builtin_amdgcn_ubfe(t < 32 ? ninv_lo[i] : ninv_hi[i], tb, 1u) << 31);
    }
  };
  if (!STEM) tap_offsets(0);
  auto load_tile = [&](int kt) {
    if constexpr (STEM) {
      // single input channel, 7^3 taps spread along K: element (row, kk) = x[voxel + off(kk)]
      const int kbase = kt * BK + kq * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = kbase + j;
          const int a = kk / 49, b = (kk / 7) % 7, c = kk % 7;
          const int z = rz[i] + a - 3, y = ry[i] + b - 3, x = rx[i] + c - 3;
          const bool ok = rv[i] && kk < 343 && (unsigned)z < (unsigned)g.Di && (unsigned)y < (unsigned)g.Hi &&
                          (unsigned)x < (unsigned)g.Wi;
          v[j] = ok ? X[(unsigned)(((rb[i] * g.Di + z) * g.Hi + y) * g.Wi + x)] : 0.f;
        }
        ra[i] = make_float4(v[0], v[1], v[2], v[3]);
      }
#pragma unroll
      for (int i = 0; i < BN / 32; ++i) {
        const int n = n0 + r0 + 32 * i;
        rbw[i] = n < g.Nout ? *(const float4*)(Wp + (long)n * (g.kpt * BK) + kbase) : make_float4(0, 0, 0, 0);
      }
    } else {
      const int cofs = ld_ci * BKT;
      const bool cok = cofs + kq * EPL < g.Cin;
      const long xb = ld_xoff + cofs;
      const float* const wb = Wp + (ld_woff + cofs);
      if constexpr (XH) {
        const unsigned short* const xh = (const unsigned short*)Xv + xb;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          ra8[i] = (cok && (vmask[i] & 1ull)) ? *(const uint4*)(xh + rowoff[i]) : make_uint4(0u, 0u, 0u, 0u);
        if (g.wh) {
          const unsigned short* const wb16 = (const unsigned short*)Wp + (ld_woff + cofs);
#pragma unroll
          for (int i = 0; i < BN / 32; ++i)
            rb8[i] = (cok && wvalid[i]) ? *(const uint4*)(wb16 + wrow[i]) : make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
          for (int i = 0; i < BN / 32; ++i) {
            rbw[i] = (cok && wvalid[i]) ? *(const float4*)(wb + wrow[i]) : make_float4(0, 0, 0, 0);
            rbw2[i] = (cok && wvalid[i]) ? *(const float4*)(wb + wrow[i] + 4) : make_float4(0, 0, 0, 0);
          }
        }
      } else if constexpr (BL) {
        const unsigned xs = (unsigned)((ld_xoff - min_xoff + cofs) * 4), ws = (unsigned)((ld_woff + cofs) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvoff[i], xs, 0));
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) rbw[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff[i], ws, 0));
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          ra[i] = (cok && (vmask[i] & 1ull)) ? ld_x4(xb + rowoff[i]) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
          rbw[i] = (cok && wvalid[i]) ? *(const float4*)(wb + wrow[i]) : make_float4(0, 0, 0, 0);
      }
      if (++ld_ci == kpt) {
        ld_ci = 0;
        if constexpr (!BL) {
#pragma unroll
          for (int i = 0; i < 4; ++i) vmask[i] >>= 1;
        }
        if (++ld_tap < ntaps) tap_offsets(ld_tap);
      }
    }
  };
  // byte offsets of this thread's staging rows, made OPAQUE so that they stay in registers: the rows are 32 x 33 floats
  // apart (4224 bytes, beyond the 8-bit dword offsets of ds_write2_b32), and left to itself the compiler re-derives every
  // address with a v_add per K tile -- 14 vector instructions per tile, each ~6 cycles of matrix-pipe time
  unsigned lds_a[4], lds_b[BN / 32];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lds_a[i] = (unsigned)(((r0 + 32 * i) * LDK + kq * 4) * 4);
    asm volatile("" : "+v"(lds_a[i]));
  }
#pragma unroll
  for (int i = 0; i < BN / 32; ++i) {
    lds_b[i] = (unsigned)((BM * LDK + (r0 + 32 * i) * LDK + kq * 4) * 4);   // from the arena's start: Bs = smem + BM * LDK
    asm volatile("" : "+v"(lds_b[i]));
  }
  auto store_tile = [&]() {
    if constexpr (XH) {
#pragma unroll
      for (int i = 0; i < 4; ++i) *(uint4*)(Ah + (r0 + 32 * i) * LDX + kq * 8) = ra8[i];
      if (g.wh) {
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) *(uint4*)(Bh + (r0 + 32 * i) * LDX + kq * 8) = rb8[i];
      } else {
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
          const bf16x4 lo = to_bf16x4(rbw[i]), hi = to_bf16x4(rbw2[i]);
          *(bf16x4*)(Bh + (r0 + 32 * i) * LDX + kq * 8) = lo;
          *(bf16x4*)(Bh + (r0 + 32 * i) * LDX + kq * 8 + 4) = hi;
        }
      }
      return;
    }
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x4 pl[NPL];
        split_bf16<NPL>(ra[i], pl);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(bf16x4*)(Ah + (p * BM + r0 + 32 * i) * LDH + kq * 4) = pl[p];
      }
#pragma unroll
      for (int i = 0; i < BN / 32; ++i) {
        bf16x4 pl[NPL];
        split_bf16<NPL>(rbw[i], pl);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(bf16x4*)(Bh + (p * BN + r0 + 32 * i) * LDH + kq * 4) = pl[p];
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* d = (float*)((char*)As + lds_a[i]);
      d[0] = ra[i].x;
      d[1] = ra[i].y;
      d[2] = ra[i].z;
      d[3] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
      float* d = (float*)((char*)smem + lds_b[i]);
      d[0] = rbw[i].x;
      d[1] = rbw[i].y;
      d[2] = rbw[i].z;
      d[3] = rbw[i].w;
    }
  };

  f32x16 acc[C::TM][C::TN];
#pragma unroll
  for (int a = 0; a < C::TM; ++a)
#pragma unroll
    for (int b = 0; b < C::TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  if constexpr (GL) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const unsigned short* const xh16 = (const unsigned short*)Xv;
    const unsigned short* const wh16 = (const unsigned short*)Wp;
    const bool all_w = n0 + BN <= g.Nout;  // every weight row of this block's N tile exists (workgroup-uniform)
    // issue the loads of K tile (ld_tap, ld_ci) into buffer `buf`, then advance the running tile
    auto stage = [&](int buf) {
      __bf16* const ab = Ah + buf * GLBUF;
      __bf16* const bb = ab + BM * 64;
      const long xb = ld_xoff + ld_ci * BKT;
      const long wbo = ld_woff + ld_ci * BKT;
      // Rows that need zeros (padding taps, rows past M / N_out) are rare: one wave-wide test decides whether the loads of
      // this tile need the per-lane choice between the row and the zero row at all
      const unsigned m4 = (unsigned)(vmask[0] & 1ull) & (unsigned)(vmask[1] & 1ull) & (unsigned)(vmask[2] & 1ull) &
                          (unsigned)(vmask[3] & 1ull);
      const unsigned short* const xt = xh16 + xb;
      if (__builtin_amdgcn_ballot_w64(m4 == 0u) == 0ull) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(xt + rowoff[i]), (lptr_t)(ab + (i * 32 + wave * 8) * 64), 16, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const void* src = (vmask[i] & 1ull) ? (const void*)(xt + rowoff[i]) : (const void*)g_zero_row;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ab + (i * 32 + wave * 8) * 64), 16, 0, 0);
        }
      }
      const unsigned short* const wt = wh16 + wbo;
      if (all_w) {
#pragma unroll
        for (int i = 0; i < BN / 32; ++i)
          __builtin_amdgcn_global_load_lds((gptr_t)(wt + wrow[i]), (lptr_t)(bb + (i * 32 + wave * 8) * 64), 16, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
          const void* src = wvalid[i] ? (const void*)(wt + wrow[i]) : (const void*)g_zero_row;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(bb + (i * 32 + wave * 8) * 64), 16, 0, 0);
        }
      }
      if (++ld_ci == kpt) {
        ld_ci = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) vmask[i] >>= 1;
        if (++ld_tap < ntaps) tap_offsets(ld_tap);
      }
    };
    const int swz = ((lane & 31) >> 1) & 7, hf = lane >> 5;
    int ck[4];  // element offset of this lane's 16 bytes of K step ks inside its (swizzled) row
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) ck[ks] = (((2 * ks + hf) ^ swz) * 8);
    const int arow = (wm * C::TM * 32 + (lane & 31)) * 64, brow = BM * 64 + (wn * C::TN * 32 + (lane & 31)) * 64;
    stage(0);
    for (int kt = 0; kt < KT; ++kt) {
      // tile kt has landed (this wave's loads; the barrier covers the other waves') and every wave is through the reads of
      // the buffer that tile kt + 1 is about to overwrite
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (kt + 1 < KT) stage((kt + 1) & 1);
      const __bf16* const tb = Ah + (kt & 1) * GLBUF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 ha[C::TM], hb[C::TN];
#pragma unroll
        for (int i = 0; i < C::TM; ++i) ha[i] = *(const bf16x8*)(tb + arow + i * 32 * 64 + ck[ks]);
#pragma unroll
        for (int j = 0; j < C::TN; ++j) hb[j] = *(const bf16x8*)(tb + brow + j * 32 * 64 + ck[ks]);
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  const float* ap = As + (wm * C::TM * 32 + (lane & 31)) * LDK + (lane >> 5);
  const float* bp = Bs + (wn * C::TN * 32 + (lane & 31)) * LDK + (lane >> 5);
  constexpr int LDF = XH ? LDX : LDH;  // bf16 fragment row stride
  const __bf16* ahp = Ah + (wm * C::TM * 32 + (lane & 31)) * LDF + 8 * (lane >> 5);
  const __bf16* bhp = Bh + (wn * C::TN * 32 + (lane & 31)) * LDF + 8 * (lane >> 5);
  // kt = -1 is the prologue: one call site for the gather keeps the pipeline uniform
  for (int kt = -1; kt < (GL ? -1 : KT); ++kt) {
    if (kt + 1 < KT) load_tile(kt + 1);
    if (BF && kt >= 0) {
      // v_mfma_f32_32x32x16_bf16: lane (row = lane&31, half = lane>>5) feeds k = 8*half .. 8*half+7
#pragma unroll
      for (int ks = 0; ks < BKT / 16; ++ks) {
        bf16x8 ha[NPL][C::TM], hb[NPL][C::TN];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
#pragma unroll
          for (int i = 0; i < C::TM; ++i) ha[p][i] = *(const bf16x8*)(ahp + (p * BM + i * 32) * LDF + ks * 16);
#pragma unroll
          for (int j = 0; j < C::TN; ++j) hb[p][j] = *(const bf16x8*)(bhp + (p * BN + j * 32) * LDF + ks * 16);
        }
        using ST = SplitTerms<NPL>;
#pragma unroll
        for (int t = 0; t < ST::N; ++t)
#pragma unroll
          for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[ST::A[t]][i], hb[ST::B[t]][j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
    if (!BF && kt >= 0) {
      // fragments of step kk+1 are fetched from LDS while the MFMAs of step kk run (two register sets)
      float fa[2][C::TM], fb[2][C::TN];
#pragma unroll
      for (int i = 0; i < C::TM; ++i) fa[0][i] = ap[i * 32 * LDK];
#pragma unroll
      for (int j = 0; j < C::TN; ++j) fb[0][j] = bp[j * 32 * LDK];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk + 1 < BK / 2) {
#pragma unroll
          for (int i = 0; i < C::TM; ++i) fa[nxt][i] = ap[i * 32 * LDK + 2 * (kk + 1)];
#pragma unroll
          for (int j = 0; j < C::TN; ++j) fb[nxt][j] = bp[j * 32 * LDK + 2 * (kk + 1)];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this step's MFMAs
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
          for (int j = 0; j < C::TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    if (kt + 1 < KT) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue
  const bool dense_out = (g.os == 1 && g.gd == g.Do && g.gh == g.Ho && g.gw == g.Wo);
  if constexpr (BN >= 64) {
    // The accumulators leave through LDS so that every lane moves 16 contiguous bytes of one output row
    // (bias / shortcut-gradient addend are read the same way); 64 tile rows per round.
    constexpr int LDO = BN + 4, Q = BN / 4;
    static_assert(64 * LDO <= ARENA, "staging tile must fit the arena");
    static_assert(C::WM == 2 && C::TM == 2, "epilogue assumes 2 waves x 2 MFMA tiles along M");
    const bool vec_ok = (g.Nout & 3) == 0;
    // Whole tiles of a dense output (every 1^3 / 3^3 stride-1 forward and data gradient away from the last M tile): rows are
    // m0 + trow, so the stores (and the addend / mask loads) are buffer accesses with a fixed per-thread offset and the row
    // group in the scalar offset -- no per-row index or 64-bit address arithmetic (exact-fp32 kernels pay every vector
    // instruction in matrix-pipe time; the K = 64 layers spent ~15 % of their time in this epilogue).
    const bool fast = !IOH && dense_out && vec_ok && m0 + BM <= g.M && n0 + BN <= g.Nout && !g.geglu;
    const long tile_el = m0 * g.Nout + n0;
    // (used by the `fast` whole-tile path only: fp32 tensors of M x Nout elements; extents: hp_extent)
    const long y_el = g.M * g.Nout;
    const unsigned y_rec = hp_extent(y_el, tile_el, 4);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((float*)Y + tile_el), 0, y_rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars =
        __builtin_amdgcn_make_buffer_rsrc((void*)((addend ? (const float*)addend : (const float*)Y) + tile_el), 0, y_rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((amask ? amask : (const unsigned char*)Y) + (tile_el >> 2)), 0, hp_extent(y_el >> 2, tile_el >> 2, 1), 0x00020000);
    constexpr int RPK = CT / Q;  // staging rows per round of 256 threads
    const int r_l = tid / Q, q_l = tid % Q;
    const unsigned vo = (unsigned)((r_l * g.Nout + 4 * q_l) * 4);
    float4 bv = make_float4(0, 0, 0, 0);
    if (fast && bias) bv = *(const float4*)(bias + n0 + 4 * q_l);
    // BNS: the tile just computed IS dy_a of the BatchNorm unit a in front of this convolution (this launch is the data gradient
    // of the only consumer of y_a): its backward sums  s = sum g,  d = sum g * zhat  (g = dy_a (.) [y_a > 0], zhat = (z_a - mean)
    // * rstd; the expressions are k_bn_bwd_reduce's) are taken here, from the tile in hand and ONE read of the z_a tile, instead
    // of by a separate pass over dy_a and z_a.  Host-checked: every tile of a BNS launch is whole (the `fast` path).
    // BNS = 1: units without residual (the ReLU mask follows from z_a's sign after the affine map); BNS = 2: a block's output
    // unit, gated by the byte mask of its output.  Two instantiations, because the byte-mask loads cost the lean one 6 % of its
    // time when they were a runtime branch inside it (data gradients 117.7 -> 124.6 ms/step at the headline shape).
    [[maybe_unused]] __amdgpu_buffer_rsrc_t zrs = yrs, bmrs = mrs;
    [[maybe_unused]] float4 b_m = make_float4(0, 0, 0, 0), b_r = b_m, b_sc = b_m, b_sh = b_m, b_s = b_m, b_d = b_m;
    if constexpr (BNS) {
      zrs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.bn_z + tile_el), 0, y_rec, 0x00020000);
      if constexpr (BNS == 2)
        bmrs = __builtin_amdgcn_make_buffer_rsrc((void*)(g.bn_mask + (tile_el >> 2)), 0, hp_extent(y_el >> 2, tile_el >> 2, 1), 0x00020000);
      b_m = *(const float4*)(g.bn_mean + n0 + 4 * q_l);
      b_r = *(const float4*)(g.bn_rstd + n0 + 4 * q_l);
      if (BNS == 1 && g.bn_relu) {
        const float4 ga = *(const float4*)(g.bn_gamma + n0 + 4 * q_l), be = *(const float4*)(g.bn_beta + n0 + 4 * q_l);
        b_sc = make_float4(b_r.x * ga.x, b_r.y * ga.y, b_r.z * ga.z, b_r.w * ga.w);
        b_sh = make_float4(be.x - b_m.x * b_sc.x, be.y - b_m.y * b_sc.y, be.z - b_m.z * b_sc.z, be.w - b_m.w * b_sc.w);
      }
    }
#pragma unroll
    for (int h = 0; h < C::TM; ++h) {
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < C::TN; ++j) smem[rl * LDO + wn * C::TN * 32 + j * 32 + (lane & 31)] = acc[h][j][r];
      }
      // The slab's addend rows (and their sign masks) are requested TOGETHER, here -- acc[h] is dead, its registers hold them --
      // and consumed after the barrier: one memory round trip per 32-row slab.  Requested inside the store loop below, each
      // load was waited for on the spot (16 exposed round trips per tile: the 1^3 data gradients with K = 64 .. 256 spent
      // more time there than in their MFMAs).
      constexpr int K2 = (64 * Q) / CT;
      float4 avb[K2];
      unsigned mkb[K2];
      [[maybe_unused]] float4 zvb[BNS ? K2 : 1];
      [[maybe_unused]] unsigned bmb[BNS == 2 ? K2 : 1];
      if constexpr (BNS) {
#pragma unroll
        for (int k2 = 0; k2 < K2; ++k2) {
          const int rowc = k2 * RPK;
          const unsigned so = (unsigned)(((rowc >> 5) * (C::TM * 32) + h * 32 + (rowc & 31)) * g.Nout * 4);
          zvb[k2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(zrs, vo, so, 0));
          if constexpr (BNS == 2) bmb[k2] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(bmrs, vo >> 4, so >> 4, 0);
        }
      }
      if (fast && addend) {
#pragma unroll
        for (int k2 = 0; k2 < K2; ++k2) {
          const int rowc = k2 * RPK;
          const unsigned so = (unsigned)(((rowc >> 5) * (C::TM * 32) + h * 32 + (rowc & 31)) * g.Nout * 4);
          avb[k2] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ars, vo, so, 0));
        }
        if (amask) {
#pragma unroll
          for (int k2 = 0; k2 < K2; ++k2) {
            const int rowc = k2 * RPK;
            const unsigned so = (unsigned)(((rowc >> 5) * (C::TM * 32) + h * 32 + (rowc & 31)) * g.Nout * 4);
            mkb[k2] = __builtin_amdgcn_raw_buffer_load_b8(mrs, vo >> 4, so >> 4, 0);
          }
        }
      }
      __syncthreads();
      if constexpr (BN == 128) {
        if (g.geglu && dense_out && m0 + BM <= g.M && n0 + BN <= g.Nout) {   // whole tile: value quads only, no bounds
          constexpr int QH = Q / 2;   // 16 value quads per row, each paired with the quad 64 columns further
          float* const yg = (float*)Y + m0 * (long)(g.Nout / 2) + n0 / 2;
          float4 bv4 = make_float4(0, 0, 0, 0), bg4 = bv4;
          const int qg = tid % QH, rg = tid / QH;
          if (bias) {   // the Linear's own bias: value entries n0 / 2 .., gate entries hidden + n0 / 2 ..
            bv4 = *(const float4*)(bias + n0 / 2 + 4 * qg);
            bg4 = *(const float4*)(bias + g.Nout / 2 + n0 / 2 + 4 * qg);
          }
          auto gl = [](float a, float t) { return a * 0.5f * t * (1.0f + erff(t * 0.70710678118654752f)); };
#pragma unroll
          for (int k2 = 0; k2 < (64 * QH) / CT; ++k2) {
            const int row64 = rg + k2 * (CT / QH);
            const int trow = (row64 >> 5) * (C::TM * 32) + h * 32 + (row64 & 31);
            float4 v = *(const float4*)(smem + row64 * LDO + 4 * qg), gt = *(const float4*)(smem + row64 * LDO + BN / 2 + 4 * qg);
            v.x += bv4.x; v.y += bv4.y; v.z += bv4.z; v.w += bv4.w;
            gt.x += bg4.x; gt.y += bg4.y; gt.z += bg4.z; gt.w += bg4.w;
            *(float4*)(yg + (long)trow * (g.Nout / 2) + 4 * qg) = make_float4(gl(v.x, gt.x), gl(v.y, gt.y), gl(v.z, gt.z), gl(v.w, gt.w));
          }
          continue;
        }
      }
      if (fast) {
#pragma unroll
        for (int k2 = 0; k2 < K2; ++k2) {
          const int rowc = k2 * RPK;                                             // staging row = r_l + rowc
          const int trow_c = (rowc >> 5) * (C::TM * 32) + h * 32 + (rowc & 31);  // tile row = r_l + trow_c (RPK divides 32)
          const unsigned so = (unsigned)(trow_c * g.Nout * 4);
          float4 v = *(const float4*)(smem + (r_l + rowc) * LDO + 4 * q_l);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          if (addend) {
            float4 av = avb[k2];
            if (amask) {
              const unsigned mk = mkb[k2];
              av.x = (mk & 1u) ? av.x : 0.f;
              av.y = (mk & 2u) ? av.y : 0.f;
              av.z = (mk & 4u) ? av.z : 0.f;
              av.w = (mk & 8u) ? av.w : 0.f;
            }
            v.x += av.x; v.y += av.y; v.z += av.z; v.w += av.w;
          }
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, vo, so, 0);
          if constexpr (BNS) {
            const float4 zz = zvb[k2];
            float4 gg = v;
            if constexpr (BNS == 2) {
              const unsigned bm = bmb[k2];
              gg.x = (bm & 1u) ? v.x : 0.f;
              gg.y = (bm & 2u) ? v.y : 0.f;
              gg.z = (bm & 4u) ? v.z : 0.f;
              gg.w = (bm & 8u) ? v.w : 0.f;
            } else if (g.bn_relu) {
              gg.x = fmaf(zz.x, b_sc.x, b_sh.x) > 0.f ? v.x : 0.f;
              gg.y = fmaf(zz.y, b_sc.y, b_sh.y) > 0.f ? v.y : 0.f;
              gg.z = fmaf(zz.z, b_sc.z, b_sh.z) > 0.f ? v.z : 0.f;
              gg.w = fmaf(zz.w, b_sc.w, b_sh.w) > 0.f ? v.w : 0.f;
            }
            b_s.x += gg.x; b_s.y += gg.y; b_s.z += gg.z; b_s.w += gg.w;
            b_d.x += gg.x * (zz.x - b_m.x) * b_r.x;
            b_d.y += gg.y * (zz.y - b_m.y) * b_r.y;
            b_d.z += gg.z * (zz.z - b_m.z) * b_r.z;
            b_d.w += gg.w * (zz.w - b_m.w) * b_r.w;
          }
        }
        continue;
      }
      if constexpr (IOH) {
        // bf16 output (and addend): 8 channels = 16 bytes per lane (the memory-bound 1^3 layers spend their time here)
        if (g.yh && (g.Nout & 7) == 0 && (!addend || g.ah)) {
          constexpr int Q8 = BN / 8;
#pragma unroll
          for (int k2 = 0; k2 < (64 * Q8) / CT; ++k2) {
            const int idx = tid + k2 * CT;
            const int row64 = idx / Q8, q8 = idx - row64 * Q8;
            const int trow = (row64 >> 5) * (C::TM * 32) + h * 32 + (row64 & 31);
            const long m = m0 + trow;
            const int n = n0 + 8 * q8;
            if (m >= g.M || n >= g.Nout) continue;
            long orow = m;
            if (!dense_out) {
              int b, z, y, x;
              grid_coords(g, m, b, z, y, x);
              orow = (long)(unsigned)(((b * g.Do + z * g.os + pd) * g.Ho + y * g.os + ph) * g.Wo + x * g.os + pw);
            }
            float4 v0 = *(const float4*)(smem + row64 * LDO + 8 * q8), v1 = *(const float4*)(smem + row64 * LDO + 8 * q8 + 4);
            const long yo = orow * g.Nout + n;
            if (bias) {
              const float4 b0 = *(const float4*)(bias + n), b1 = *(const float4*)(bias + n + 4);
              v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w;
              v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
            }
            if (addend) {
              const uint4 u = *(const uint4*)((const unsigned short*)addend + yo);
              float4 a0 = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                                      __uint_as_float(u.y & 0xffff0000u));
              float4 a1 = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                                      __uint_as_float(u.w & 0xffff0000u));
              if (amask) {
                const unsigned mk = *(const unsigned short*)(amask + (yo >> 2));
                a0.x = (mk & 1u) ? a0.x : 0.f; a0.y = (mk & 2u) ? a0.y : 0.f; a0.z = (mk & 4u) ? a0.z : 0.f; a0.w = (mk & 8u) ? a0.w : 0.f;
                a1.x = (mk & 0x100u) ? a1.x : 0.f; a1.y = (mk & 0x200u) ? a1.y : 0.f; a1.z = (mk & 0x400u) ? a1.z : 0.f;
                a1.w = (mk & 0x800u) ? a1.w : 0.f;
              }
              v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w;
              v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
            }
            const bf16x4 l = to_bf16x4(v0), hgh = to_bf16x4(v1);
            union {
              bf16x4 h4[2];
              uint4 u;
            } o;
            o.h4[0] = l;
            o.h4[1] = hgh;
            *(uint4*)((unsigned short*)Y + yo) = o.u;
          }
          continue;
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < (64 * Q) / CT; ++k2) {
        const int idx = tid + k2 * CT;
        const int row64 = idx / Q, q = idx - row64 * Q;
        const int trow = (row64 >> 5) * (C::TM * 32) + h * 32 + (row64 & 31);
        const long m = m0 + trow;
        const int n = n0 + 4 * q;
        if (m >= g.M || n >= g.Nout) continue;
        long orow = m;
        if (!dense_out) {
          int b, z, y, x;
          grid_coords(g, m, b, z, y, x);
          orow = (long)(unsigned)(((b * g.Do + z * g.os + pd) * g.Ho + y * g.os + ph) * g.Wo + x * g.os + pw);
        }
        float4 v = *(const float4*)(smem + row64 * LDO + 4 * q);
        const long yo = orow * g.Nout + n;
        if constexpr (BN == 128) {   // every 128-column instantiation (hp_linear_geglu_forward launches no other; tensors are fp32)
          if (g.geglu) {  // workgroup-uniform.  u = x W^T + b is never written: Y[m][n0 / 2 + c] = u[c] * gelu(u[64 + c])
            if (q >= Q / 2) continue;
            float4 gt = *(const float4*)(smem + row64 * LDO + 4 * q + BN / 2);
            if (bias) {
              const float4 bv = *(const float4*)(bias + n0 / 2 + 4 * q), bg = *(const float4*)(bias + g.Nout / 2 + n0 / 2 + 4 * q);
              v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
              gt.x += bg.x; gt.y += bg.y; gt.z += bg.z; gt.w += bg.w;
            }
            auto gl = [](float a, float t) { return a * 0.5f * t * (1.0f + erff(t * 0.70710678118654752f)); };  // k_geglu's expression
            *(float4*)((float*)Y + orow * (g.Nout / 2) + (n0 / 2 + 4 * q)) = make_float4(gl(v.x, gt.x), gl(v.y, gt.y), gl(v.z, gt.z), gl(v.w, gt.w));
            continue;
          }
        }
        if (vec_ok) {
          if (bias) {
            const float4 bv = *(const float4*)(bias + n);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
          }
          if (addend) {
            float4 av = ld_a4(yo);
            if (amask) {  // byte per channel quad (hp_bn_apply's relu_mask): the addend is dy (.) mask, never stored
              const unsigned mk = amask[(orow * g.Nout + n) >> 2];
              av.x = (mk & 1u) ? av.x : 0.f;
              av.y = (mk & 2u) ? av.y : 0.f;
              av.z = (mk & 4u) ? av.z : 0.f;
              av.w = (mk & 8u) ? av.w : 0.f;
            }
            v.x += av.x; v.y += av.y; v.z += av.z; v.w += av.w;
          }
          st_y4(yo, v);
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 4; ++e)
            if (n + e < g.Nout) st_y1(yo + e, vv[e] + (bias ? bias[n + e] : 0.f) + (addend ? ld_a1(yo + e) : 0.f));
        }
      }
    }
    if constexpr (BNS) {
      // column sums over the tile's rows: the RPK threads that share a column quad meet in the (now free) staging arena, then
      // one fp64 atomic per column and statistic into this workgroup's slot (HP_STATS_SLOTS, as the forward statistics)
      __syncthreads();
      float4* const part = (float4*)smem;        // [RPK][Q][2]
      part[(r_l * Q + q_l) * 2 + 0] = b_s;
      part[(r_l * Q + q_l) * 2 + 1] = b_d;
      __syncthreads();
      if (tid < Q) {
        float4 ss = part[tid * 2], dd = part[tid * 2 + 1];
#pragma unroll
        for (int rr = 1; rr < RPK; ++rr) {
          const float4 a = part[(rr * Q + tid) * 2], b = part[(rr * Q + tid) * 2 + 1];
          ss.x += a.x; ss.y += a.y; ss.z += a.z; ss.w += a.w;
          dd.x += b.x; dd.y += b.y; dd.z += b.z; dd.w += b.w;
        }
        double* const slot = g.bn_sums + (size_t)((blockIdx.x + blockIdx.z) & (HP_STATS_SLOTS - 1)) * 2 * g.Nout + n0 + 4 * tid;
        atomicAdd(slot + 0, (double)ss.x); atomicAdd(slot + 1, (double)ss.y); atomicAdd(slot + 2, (double)ss.z); atomicAdd(slot + 3, (double)ss.w);
        atomicAdd(slot + g.Nout + 0, (double)dd.x); atomicAdd(slot + g.Nout + 1, (double)dd.y);
        atomicAdd(slot + g.Nout + 2, (double)dd.z); atomicAdd(slot + g.Nout + 3, (double)dd.w);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < C::TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * C::TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const long m = m0 + row;
        if (m >= g.M) continue;
        long orow = m;
        if (!dense_out) {
          int b, z, y, x;
          grid_coords(g, m, b, z, y, x);
          orow = (long)(unsigned)(((b * g.Do + z * g.os + pd) * g.Ho + y * g.os + ph) * g.Wo + x * g.os + pw);
        }
#pragma unroll
        for (int j = 0; j < C::TN; ++j) {
          const int n = n0 + wn * C::TN * 32 + j * 32 + (lane & 31);
          if (n < g.Nout)
            st_y1(orow * g.Nout + n, acc[i][j][r] + (bias ? bias[n] : 0.f) + (addend ? ld_a1(orow * g.Nout + n) : 0.f));
        }
      }
    }
  }
  if constexpr (STATS) {
    // column sums over this block's rows (rows past M hold exact zeros)
#pragma unroll
    for (int j = 0; j < C::TN; ++j) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          s += v;
          q += v * v;
        }
      s += __shfl_xor(s, 32);
      q += __shfl_xor(q, 32);
      if (lane < 32) {
        const int col = wn * C::TN * 32 + j * 32 + lane;
        red[(wm * BN + col) * 2 + 0] = s;
        red[(wm * BN + col) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < BN) {
      const int n = n0 + tid;
      if (n < g.Nout) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int w = 0; w < C::WM; ++w) {
          s += (double)red[(w * BN + tid) * 2 + 0];
          q += (double)red[(w * BN + tid) * 2 + 1];
        }
        // (the launch index, not the M tile: with the XCD slab order the tiles in flight differ by multiples of the slab length)
        double* const slot = stats + (size_t)((blockIdx.x + blockIdx.z) & (HP_STATS_SLOTS - 1)) * 2 * g.Nout;   // see hp_conv3d_forward
        atomicAdd(slot + n, s);
        atomicAdd(slot + g.Nout + n, q);
      }
    }
  }
}

// ---------------------------------------------------------------- weight gradient
// dW[slab][n][c] += sum_m dY[orow(m)][n] * X[gather(m, tap)][c]   (reduction over voxels)
// A = dY^T tile, B = gathered X tile; both are staged exactly as they lie in memory
// ([m][channel]), which is already the conflict-free MFMA fragment order.  The voxel range
// is split over blockIdx.z; partial tiles are combined with fp32 atomics.
constexpr int WG_KM = 32;  // voxels per step

// TT x TT weight tile per workgroup (TT = 128: 2x2 MFMA tiles per wave; TT = 64: one), 32 voxels per
// step, global loads of step s+1 in flight while the MFMAs of step s run.  With TT = 64 a block may own
// NTAP consecutive taps of a single-class convolution: the dY tile is staged once and reused by the
// NTAP gathered X tiles (1 + NTAP LDS fragment reads feed NTAP MFMAs).
// XH: X and dY are both bf16 in memory (NP = 1): rows are staged with 16-byte loads of 8 bf16 -- half the load
// instructions per step -- and go to LDS as loaded.
template <bool STEM, int TT, int NTAP, int NP, bool XH = false>
__global__ __launch_bounds__(CT) void k_wgrad(const void* __restrict__ Xv, const void* __restrict__ dY,
                                              float* __restrict__ dW, IgemmGeom g, int tiles_c, int msplit,
                                              int tiles_total, int tap_groups) {
  const float* const X = (const float*)Xv;  // stem: always fp32.  g.xh / g.yh: element type of X / dY (NP = 1 only)
  constexpr bool IOH = NP == 1;
  auto ld_y4 = [&](long off) -> float4 {
    if constexpr (IOH) return hp_ld4(dY, off, g.yh);
    else return *(const float4*)((const float*)dY + off);
  };
  auto ld_y1 = [&](long off) -> float {
    if constexpr (IOH) return hp_ld1(dY, off, g.yh);
    else return ((const float*)dY)[off];
  };
  auto ld_x4 = [&](long off) -> float4 {
    if constexpr (IOH) return hp_ld4(Xv, off, g.xh);
    else return *(const float4*)(X + off);
  };
  static_assert(!XH || (NP == 1 && !STEM), "bf16-storage staging: single bf16 plane, not the stem");
  // voxels per step (128 / 64 per step for the bf16-storage variant were measured: 6 % slower than 32)
  constexpr int KM = WG_KM;
  constexpr int EPL = XH ? 8 : 4;       // elements per thread, row and load
  constexpr int QN = TT / EPL;          // 16-byte loads per staged row
  constexpr int RPP = CT / QN;          // rows per staging pass
  constexpr int RPT = KM / RPP;      // rows per thread
  static_assert(RPT >= 1, "staging pass covers at most one step");
  constexpr int WT = TT / 64;           // 32x32 tiles per wave per dim
  static_assert(NTAP == 1 || TT == 64, "multi-tap blocks use the 64x64 tile");
  constexpr bool BF = NP > 0;
  constexpr int NPL = NP > 0 ? NP : 1;
  constexpr int LDT = TT + 32;
  constexpr int YSZ = XH ? KM * LDT / 2 : (KM * TT > NP * KM * LDT / 2 ? KM * TT : NP * KM * LDT / 2);  // floats
  __shared__ __attribute__((aligned(16))) float Ys[YSZ];
  __shared__ __attribute__((aligned(16))) float Xs[NTAP * YSZ];
  // BF: the tiles are kept as bf16 [voxel][TT + 32]; the matrix cores want 8 consecutive voxels (the GEMM K
  // axis) per lane, which the transposing LDS read (ds_read_b64_tr_b16) delivers from this row-major
  // image.  Row stride = 16 dwords mod 32, so the 4 rows x 64 bytes one half-wave touches cover all banks.
  // planes: Yh[p][voxel][LDT], Xh[tap][p][voxel][LDT]
  __bf16* const Yh = (__bf16*)Ys;
  __bf16* const Xh = (__bf16*)Xs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, tap group fastest: blocks that run together share the same voxel range, so the dY rows and
  // the (shifted) X rows they gather are served by L2 / Infinity Cache instead of being re-fetched per tap
  // Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8).  The blocks that share a voxel
  // range (same split: every tap and tile re-reads the same dY / X rows) are therefore renumbered so that XCD k
  // owns a contiguous range of logical ids = whole splits: each per-XCD L2 then streams only its own voxel ranges
  // instead of all of them.  Pure renumbering: any placement gives the same result.
  unsigned bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int bid_tap = bid % tap_groups;
  const int bid_tile = (bid / tap_groups) % tiles_total;
  const int bid_split = bid / (tap_groups * tiles_total);
  const int tile_n = bid_tile / tiles_c, tile_c = bid_tile % tiles_c;
  const int n0 = tile_n * TT, c0 = tile_c * TT;
  int cls = 0, tap = bid_tap * NTAP;  // (class, tap group)
  if (!STEM && NTAP == 1) {
    while (tap >= class_ntaps(g, cls)) {
      tap -= class_ntaps(g, cls);
      ++cls;
    }
  }
  const int ntaps_cls = STEM ? 1 : class_ntaps(g, cls);
  const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
  int dz[NTAP], dy[NTAP], dx[NTAP], widx[NTAP];
  bool tv[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    dz[t] = dy[t] = dx[t] = widx[t] = 0;
    tv[t] = tap + t < ntaps_cls;
    if (!STEM && tv[t]) tap_info(g, cls, tap + t, dz[t], dy[t], dx[t], widx[t]);
  }
  const long chunk = ((g.M + msplit - 1) / msplit + KM - 1) / KM * KM;
  const long mbeg = (long)bid_split * chunk, mend = mbeg + chunk < g.M ? mbeg + chunk : g.M;
  const int Kc = STEM ? g.kpt * BK : g.Cin;  // extent of the c axis
  const int sq = tid % QN, sr = tid / QN;
  const int wn = wave >> 1, wc = wave & 1;

  f32x16 acc[NTAP][WT][WT];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;

  float4 vy[RPT], vx[NTAP][RPT];
  uint4 hy[XH ? RPT : 1], hx[XH ? NTAP : 1][XH ? RPT : 1];
  auto load_step = [&](long mb) {
    if constexpr (XH) {
      const unsigned short* const Xh16 = (const unsigned short*)Xv;
      const unsigned short* const Yh16 = (const unsigned short*)dY;
#pragma unroll
      for (int h = 0; h < RPT; ++h) {
        const long m = mb + sr + RPP * h;
        hy[h] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) hx[t][h] = make_uint4(0u, 0u, 0u, 0u);
        if (m < mend) {
          int b, z, y, x;
          grid_coords(g, m, b, z, y, x);
          const long orow = (long)(unsigned)(((b * g.Do + z * g.os + pd) * g.Ho + y * g.os + ph) * g.Wo + x * g.os + pw);
          const int n = n0 + sq * 8;
          if (n + 7 < g.Nout) hy[h] = *(const uint4*)(Yh16 + orow * g.Nout + n);
          const int c = c0 + sq * 8;
#pragma unroll
          for (int t = 0; t < NTAP; ++t) {
            const int zz = z * g.s + dz[t], yy = y * g.s + dy[t], xx = x * g.s + dx[t];
            if (tv[t] && (unsigned)zz < (unsigned)g.Di && (unsigned)yy < (unsigned)g.Hi && (unsigned)xx < (unsigned)g.Wi &&
                c + 7 < g.Cin)
              hx[t][h] = *(const uint4*)(Xh16 + (long)(unsigned)(((b * g.Di + zz) * g.Hi + yy) * g.Wi + xx) * g.Cin + c);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
      const long m = mb + sr + RPP * h;
      vy[h] = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < NTAP; ++t) vx[t][h] = make_float4(0, 0, 0, 0);
      if (m < mend) {
        int b, z, y, x;
        grid_coords(g, m, b, z, y, x);
        const long orow = (long)(unsigned)(((b * g.Do + z * g.os + pd) * g.Ho + y * g.os + ph) * g.Wo + x * g.os + pw);
        const int n = n0 + sq * 4;
        if (n + 3 < g.Nout) {
          vy[h] = ld_y4(orow * g.Nout + n);
        } else if (n < g.Nout) {
          float t4[4] = {0, 0, 0, 0};
          for (int e = 0; e < 4; ++e)
            if (n + e < g.Nout) t4[e] = ld_y1(orow * g.Nout + n + e);
          vy[h] = make_float4(t4[0], t4[1], t4[2], t4[3]);
        }
        if constexpr (STEM) {
          float t4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int kk = c0 + sq * 4 + e;
            const int a = kk / 49, bb = (kk / 7) % 7, cc = kk % 7;
            const int zz = z + a - 3, yy = y + bb - 3, xx = x + cc - 3;
            const bool ok = kk < 343 && (unsigned)zz < (unsigned)g.Di && (unsigned)yy < (unsigned)g.Hi &&
                            (unsigned)xx < (unsigned)g.Wi;
            t4[e] = ok ? X[(unsigned)(((b * g.Di + zz) * g.Hi + yy) * g.Wi + xx)] : 0.f;
          }
          vx[0][h] = make_float4(t4[0], t4[1], t4[2], t4[3]);
        } else {
          const int c = c0 + sq * 4;
#pragma unroll
          for (int t = 0; t < NTAP; ++t) {
            const int zz = z * g.s + dz[t], yy = y * g.s + dy[t], xx = x * g.s + dx[t];
            if (tv[t] && (unsigned)zz < (unsigned)g.Di && (unsigned)yy < (unsigned)g.Hi && (unsigned)xx < (unsigned)g.Wi &&
                c < g.Cin)
              vx[t][h] = ld_x4((long)(unsigned)(((b * g.Di + zz) * g.Hi + yy) * g.Wi + xx) * g.Cin + c);
          }
        }
      }
    }
  };

  if (mbeg < mend) load_step(mbeg);
  for (long mb = mbeg; mb < mend; mb += KM) {
    __syncthreads();  // fragment reads of the previous step are done
#pragma unroll
    for (int h = 0; h < RPT; ++h) {
      if constexpr (XH) {
        *(uint4*)(Yh + (sr + RPP * h) * LDT + sq * 8) = hy[h];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) *(uint4*)(Xh + (t * KM + sr + RPP * h) * LDT + sq * 8) = hx[t][h];
      } else if constexpr (BF) {
        bf16x4 pl[NPL];
        split_bf16<NPL>(vy[h], pl);
#pragma unroll
        for (int p = 0; p < NPL; ++p) *(bf16x4*)(Yh + (p * KM + sr + RPP * h) * LDT + sq * 4) = pl[p];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          split_bf16<NPL>(vx[t][h], pl);
#pragma unroll
          for (int p = 0; p < NPL; ++p)
            *(bf16x4*)(Xh + ((t * NPL + p) * KM + sr + RPP * h) * LDT + sq * 4) = pl[p];
        }
      } else {
        *(float4*)(Ys + (sr + RPP * h) * TT + sq * 4) = vy[h];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) *(float4*)(Xs + (t * KM + sr + RPP * h) * TT + sq * 4) = vx[t][h];
      }
    }
    __syncthreads();
    if (mb + KM < mend) load_step(mb + KM);
    if constexpr (BF) {
      // 16-lane group gq of the wave: voxels 8*(gq>>1) .. +7 of the K step, channels 16*(gq&1) .. +15 of the
      // 32-wide tile; lane 4q+p of the group addresses row q, columns 4p .. 4p+3 (two reads: rows +0, +4)
      const int gq = lane >> 4, li = lane & 15;
      const int trow = 8 * (gq >> 1) + (li >> 2), tcol = 16 * (gq & 1) + 4 * (li & 3);
      typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
      auto tr8 = [&](const __bf16* base) -> bf16x8 {
        union {
          s16x4 h[2];
          bf16x8 f;
        } u;
        u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
        u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * LDT));
        return u.f;
      };
      using ST = SplitTerms<NPL>;
#pragma unroll
      for (int ks = 0; ks < KM / 16; ++ks) {
        bf16x8 ha[NPL][WT];
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
          for (int i = 0; i < WT; ++i)
            ha[p][i] = tr8(Yh + (p * KM + ks * 16 + trow) * LDT + wn * (TT / 2) + i * 32 + tcol);
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
          bf16x8 hb[NPL][WT];
#pragma unroll
          for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int j = 0; j < WT; ++j)
              hb[p][j] = tr8(Xh + ((t * NPL + p) * KM + ks * 16 + trow) * LDT + wc * (TT / 2) + j * 32 + tcol);
#pragma unroll
          for (int u = 0; u < ST::N; ++u)
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
              for (int j = 0; j < WT; ++j)
                acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[ST::A[u]][i], hb[ST::B[u]][j], acc[t][i][j], 0, 0, 0);
        }
      }
      continue;
    }
    // two fragment register sets: step kk+1 is read from LDS while the MFMAs of step kk run
    float fa[2][WT], fb[2][NTAP][WT];
    auto frag = [&](int set, int kk) {
      const int mrow = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < WT; ++i) fa[set][i] = Ys[mrow * TT + wn * (TT / 2) + i * 32 + (lane & 31)];
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int j = 0; j < WT; ++j) fb[set][t][j] = Xs[(t * KM + mrow) * TT + wc * (TT / 2) + j * 32 + (lane & 31)];
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < KM / 2; ++kk) {
      const int cur = kk & 1;
      if (kk + 1 < KM / 2) frag(cur ^ 1, kk + 1);
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this step's MFMAs
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < WT; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][t][j], acc[t][i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // dW layout: [slab][n][Kc]
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    if (!tv[t]) continue;
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn * (TT / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          const int c = c0 + wc * (TT / 2) + j * 32 + (lane & 31);
          if (n < g.Nout && c < Kc) atomicAdd(dW + ((long)widx[t] * g.Nout + n) * Kc + c, acc[t][i][j][r]);
        }
  }
}

// ---------------------------------------------------------------- weight gradient, exact fp32, lean address stream
// Same decomposition as k_wgrad (TT x TT weight tile, 32 voxels per step, NTAP taps share the staged dY tile) for
// power-of-two class grids, rebuilt around one measurement (tools/micro/mfma_coexec.hip): while a wave streams fp32 MFMAs,
// no other wave of its SIMD issues vector-ALU or vector-memory instructions, so every such instruction of the step is paid
// in matrix-pipe time -- and k_wgrad spends ~250 of them per step and thread on coordinates, bounds and 64-bit addresses.
// Here a thread owns ONE voxel row per step (its TT/32 float4 of dY and of each tap's X row), the coordinates come from
// three shifts, row offsets are 32-bit multiply-adds relative to the block's first row, and the loads are buffer loads whose
// out-of-range offset IS the zero fill of padding taps and of rows past the range: ~20 vector instructions per step.
// UNI (class grids at least 8 wide): the 8 rows a wave loads per step are 8 consecutive x positions of one (b, z, y) line,
// so that line's coordinates, its bounds tests and both row offsets are SCALAR work (the wave's scalar offset of the buffer
// loads); per lane there remain a fixed offset and, per tap, the x bound of the line's two ends: 3 vector instructions.
template <int TT, int NTAP, bool UNI>
__global__ __launch_bounds__(CT) void k_wgrad_bl(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ dW,
                                                 IgemmGeom g, int tiles_c, int msplit, int tiles_total, int tap_groups) {
  constexpr int KM = WG_KM;        // voxels per step
  constexpr int QT = TT / 32;      // float4 per thread, row and tensor (8 threads per row)
  constexpr int WT = TT / 64;
  constexpr unsigned OOB = 0x80000000u;
  static_assert(NTAP == 1 || TT == 64, "multi-tap blocks use the 64x64 tile");
  __shared__ __attribute__((aligned(16))) float Ys[KM * TT];
  __shared__ __attribute__((aligned(16))) float Xs[NTAP * KM * TT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // see k_wgrad
  const int bid_tap = bid % tap_groups;
  const int bid_tile = (bid / tap_groups) % tiles_total;
  const int bid_split = bid / (tap_groups * tiles_total);
  const int n0 = (bid_tile / tiles_c) * TT, c0 = (bid_tile % tiles_c) * TT;
  int cls = 0, tap = bid_tap * NTAP;
  if (NTAP == 1) {
    while (tap >= class_ntaps(g, cls)) {
      tap -= class_ntaps(g, cls);
      ++cls;
    }
  }
  const int ntaps_cls = class_ntaps(g, cls);
  const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
  int dz[NTAP], dy[NTAP], dx[NTAP], widx[NTAP];
  bool tv[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    dz[t] = dy[t] = dx[t] = widx[t] = 0;
    tv[t] = tap + t < ntaps_cls;
    if (tv[t]) tap_info(g, cls, tap + t, dz[t], dy[t], dx[t], widx[t]);
  }
  const long chunk = ((g.M + msplit - 1) / msplit + KM - 1) / KM * KM;
  const long mbeg = (long)bid_split * chunk, mend = mbeg + chunk < g.M ? mbeg + chunk : g.M;
  const int wn = wave >> 1, wc = wave & 1;
  const int r = tid >> 3, q = tid & 7;  // this thread's row of the step and its first channel quad (then q + 8, ...)

  // scalar: coordinates of the block's first row; every row offset below is taken relative to it (32-bit)
  const unsigned m0u = (unsigned)(mbeg < g.M ? mbeg : 0);
  const int x0 = (int)(m0u & (unsigned)(g.gw - 1)), y0 = (int)((m0u >> g.sw) & (unsigned)(g.gh - 1));
  const int zb0 = (int)(m0u >> (g.sw + g.sh));  // b * gd + z
  const long orow0 = ((long)(zb0 * g.os + pd) * g.Ho + y0 * g.os + ph) * g.Wo + x0 * g.os + pw;
  // lowest X row any tap of this block can touch: the first row displaced by the smallest tap offset
  long tmin = 0;
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const long o = ((long)dz[t] * g.Hi + dy[t]) * g.Wi + dx[t];
    tmin = (t == 0 || o < tmin) ? o : tmin;
  }
  const long xrow0 = ((long)(zb0 * g.s) * g.Hi + y0 * g.s) * g.Wi + x0 * g.s + tmin;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(dY + orow0 * g.Nout + n0), 0, hp_extent((long)g.B * g.Do * g.Ho * g.Wo * g.Nout, orow0 * g.Nout + n0, (int)sizeof(*dY)), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(X + xrow0 * g.Cin + c0), 0, hp_extent((long)g.B * g.Di * g.Hi * g.Wi * g.Cin, xrow0 * g.Cin + c0, (int)sizeof(*X)), 0x00020000);
  // per-axis strides of the two row indices, in rows (scalars)
  const int oy_s = g.os * g.Wo, oz_s = g.os * g.Ho * g.Wo;            // dY row  = zb * oz_s + y * oy_s + x * os + const
  const int xy_s = g.s * g.Wi, xz_s = g.s * g.Hi * g.Wi;              // X row   = zb * xz_s + y * xy_s + x * s  + tap + const
  const int ybytes = g.Nout * 4, xbytes = g.Cin * 4;
  int toff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) toff[t] = (int)((((long)dz[t] * g.Hi + dy[t]) * g.Wi + dx[t]) - tmin);
  const bool nfull = n0 + TT <= g.Nout, cfull = c0 + TT <= g.Cin;  // this block's channel ranges are whole (workgroup-uniform)

  f32x16 acc[NTAP][WT][WT];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;

  float4 vy[QT], vx[NTAP][QT];
  // UNI: this wave's 8 rows of a step start at row 8 * wv; lane part of every offset (x position lx of the line, channel quad q)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lx = (tid >> 3) & 7;
  const unsigned ylane = (unsigned)(lx * g.os * ybytes + q * 16), xlane = (unsigned)(lx * g.s * xbytes + q * 16);
  const int lxs = lx * g.s;
  auto load_step = [&](long mb) {
    if constexpr (UNI) {
      const unsigned mw = (unsigned)mb + 8u * (unsigned)wv;  // first of this wave's 8 rows: a multiple of 8 (scalar)
      const bool live = (long)mw < mend;                    // the range ends on a multiple of 8: all 8 rows or none
      const int xw = (int)(mw & (unsigned)(g.gw - 1)), y = (int)((mw >> g.sw) & (unsigned)(g.gh - 1));
      const int zb = (int)(mw >> (g.sw + g.sh)), z = zb & (g.gd - 1);
      const int dzb = zb - zb0, dyy = y - y0, dxx = xw - x0;
      const unsigned ys = (unsigned)((dzb * oz_s + dyy * oy_s + dxx * g.os) * ybytes);
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        const bool ok = live && (nfull || n0 + (q + 8 * j) * 4 + 3 < g.Nout);
        vy[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(yrs, ok ? ylane + 128u * j : OOB, ys, 0));
      }
      const int xr = dzb * xz_s + dyy * xy_s + dxx * g.s;
      const int zi = z * g.s, yi = y * g.s, xi = xw * g.s;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const bool line = live && tv[t] && (unsigned)(zi + dz[t]) < (unsigned)g.Di && (unsigned)(yi + dy[t]) < (unsigned)g.Hi;  // scalar
        const bool in = line && (unsigned)(lxs + xi + dx[t]) < (unsigned)g.Wi;
        const unsigned xs = (unsigned)((xr + toff[t]) * xbytes);
#pragma unroll
        for (int j = 0; j < QT; ++j) {
          const bool ok = in && (cfull || c0 + (q + 8 * j) * 4 + 3 < g.Cin);
          vx[t][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? xlane + 128u * j : OOB, xs, 0));
        }
      }
      return;
    }
    const unsigned m = (unsigned)mb + (unsigned)r;
    const bool live = (long)m < mend;
    const int x = (int)(m & (unsigned)(g.gw - 1)), y = (int)((m >> g.sw) & (unsigned)(g.gh - 1));
    const int zb = (int)(m >> (g.sw + g.sh)), z = zb & (g.gd - 1);
    const int dzb = zb - zb0, dyy = y - y0, dxx = x - x0;
    unsigned yo = (unsigned)((dzb * oz_s + dyy * oy_s + dxx * g.os) * ybytes) + (unsigned)(q * 16);
    yo = live ? yo : OOB;
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      const bool ok = nfull || n0 + (q + 8 * j) * 4 + 3 < g.Nout;
      vy[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(yrs, ok ? yo + 128u * j : OOB, 0, 0));
    }
    const int xr = dzb * xz_s + dyy * xy_s + dxx * g.s;
    const int zi = z * g.s, yi = y * g.s, xi = x * g.s;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      const bool in = live && tv[t] && (unsigned)(zi + dz[t]) < (unsigned)g.Di && (unsigned)(yi + dy[t]) < (unsigned)g.Hi &&
                      (unsigned)(xi + dx[t]) < (unsigned)g.Wi;
      const unsigned xo = in ? (unsigned)((xr + toff[t]) * xbytes) + (unsigned)(q * 16) : OOB;
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        const bool ok = cfull || c0 + (q + 8 * j) * 4 + 3 < g.Cin;
        vx[t][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? xo + 128u * j : OOB, 0, 0));
      }
    }
  };

  if (mbeg < mend) load_step(mbeg);
  for (long mb = mbeg; mb < mend; mb += KM) {
    __syncthreads();  // fragment reads of the previous step are done
#pragma unroll
    for (int j = 0; j < QT; ++j) {
      *(float4*)(Ys + r * TT + (q + 8 * j) * 4) = vy[j];
#pragma unroll
      for (int t = 0; t < NTAP; ++t) *(float4*)(Xs + (t * KM + r) * TT + (q + 8 * j) * 4) = vx[t][j];
    }
    __syncthreads();
    if (mb + KM < mend) load_step(mb + KM);
    float fa[2][WT], fb[2][NTAP][WT];
    auto frag = [&](int set, int kk) {
      const int mrow = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < WT; ++i) fa[set][i] = Ys[mrow * TT + wn * (TT / 2) + i * 32 + (lane & 31)];
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int j = 0; j < WT; ++j) fb[set][t][j] = Xs[(t * KM + mrow) * TT + wc * (TT / 2) + j * 32 + (lane & 31)];
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < KM / 2; ++kk) {
      const int cur = kk & 1;
      if (kk + 1 < KM / 2) frag(cur ^ 1, kk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < WT; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][t][j], acc[t][i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    if (!tv[t]) continue;
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = n0 + wn * (TT / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int c = c0 + wc * (TT / 2) + j * 32 + (lane & 31);
          if (n < g.Nout && c < g.Cin) atomicAdd(dW + ((long)widx[t] * g.Nout + n) * g.Cin + c, acc[t][i][j][e]);
        }
  }
}

// The same lean address stream for the bf16-storage weight gradient (X and dY bf16 in memory, v_mfma_f32_32x32x16_bf16,
// fragments through ds_read_b64_tr_b16 as in k_wgrad's XH path).  There the step is only 8 MFMAs per wave (256 cycles of
// matrix pipe) against ~120 vector instructions of coordinate and address arithmetic per thread: the kernel was bound by its
// own address stream.  Wave-uniform rows (see k_wgrad_bl, UNI): a wave's 8 rows per pass are 8 consecutive x positions of one
// line, everything but the x bound of a tap is scalar, and the 16-byte loads are buffer loads with the zero fill in the range
// check.  KMH voxels per step (8 threads per row, KMH / 32 passes).
template <int TT, int NTAP, int KMH>
__global__ __launch_bounds__(CT) void k_wgrad_blh(const unsigned short* __restrict__ X, const unsigned short* __restrict__ dY,
                                                  float* __restrict__ dW, IgemmGeom g, int tiles_c, int msplit, int tiles_total,
                                                  int tap_groups) {
  constexpr int QT = TT / 64;      // 16-byte loads (8 bf16) per thread, row and tensor: 8 threads cover a row of TT channels
  constexpr int NPASS = KMH / 32;  // 32 rows per pass
  constexpr int WT = TT / 64;
  constexpr int LDT = TT + 32;     // bf16 row stride: 16 dwords mod 32 (see k_wgrad)
  constexpr unsigned OOB = 0x80000000u;
  static_assert(NTAP == 1 || TT == 64, "multi-tap blocks use the 64x64 tile");
  __shared__ __attribute__((aligned(16))) __bf16 Yh[KMH * LDT];
  __shared__ __attribute__((aligned(16))) __bf16 Xh[NTAP * KMH * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int bid_tap = bid % tap_groups;
  const int bid_tile = (bid / tap_groups) % tiles_total;
  const int bid_split = bid / (tap_groups * tiles_total);
  const int n0 = (bid_tile / tiles_c) * TT, c0 = (bid_tile % tiles_c) * TT;
  int cls = 0, tap = bid_tap * NTAP;
  if (NTAP == 1) {
    while (tap >= class_ntaps(g, cls)) {
      tap -= class_ntaps(g, cls);
      ++cls;
    }
  }
  const int ntaps_cls = class_ntaps(g, cls);
  const int pd = (cls >> 2) & 1, ph = (cls >> 1) & 1, pw = cls & 1;
  int dz[NTAP], dy[NTAP], dx[NTAP], widx[NTAP];
  bool tv[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    dz[t] = dy[t] = dx[t] = widx[t] = 0;
    tv[t] = tap + t < ntaps_cls;
    if (tv[t]) tap_info(g, cls, tap + t, dz[t], dy[t], dx[t], widx[t]);
  }
  const long chunk = ((g.M + msplit - 1) / msplit + KMH - 1) / KMH * KMH;
  const long mbeg = (long)bid_split * chunk, mend = mbeg + chunk < g.M ? mbeg + chunk : g.M;
  const int wn = wave >> 1, wc = wave & 1;
  const int r = tid >> 3, q = tid & 7;
  const unsigned m0u = (unsigned)(mbeg < g.M ? mbeg : 0);
  const int x0 = (int)(m0u & (unsigned)(g.gw - 1)), y0 = (int)((m0u >> g.sw) & (unsigned)(g.gh - 1));
  const int zb0 = (int)(m0u >> (g.sw + g.sh));
  const long orow0 = ((long)(zb0 * g.os + pd) * g.Ho + y0 * g.os + ph) * g.Wo + x0 * g.os + pw;
  long tmin = 0;
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const long o = ((long)dz[t] * g.Hi + dy[t]) * g.Wi + dx[t];
    tmin = (t == 0 || o < tmin) ? o : tmin;
  }
  const long xrow0 = ((long)(zb0 * g.s) * g.Hi + y0 * g.s) * g.Wi + x0 * g.s + tmin;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(dY + orow0 * g.Nout + n0), 0, hp_extent((long)g.B * g.Do * g.Ho * g.Wo * g.Nout, orow0 * g.Nout + n0, (int)sizeof(*dY)), 0x00020000);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(X + xrow0 * g.Cin + c0), 0, hp_extent((long)g.B * g.Di * g.Hi * g.Wi * g.Cin, xrow0 * g.Cin + c0, (int)sizeof(*X)), 0x00020000);
  const int oy_s = g.os * g.Wo, oz_s = g.os * g.Ho * g.Wo, xy_s = g.s * g.Wi, xz_s = g.s * g.Hi * g.Wi;
  const int ybytes = g.Nout * 2, xbytes = g.Cin * 2;
  int toff[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) toff[t] = (int)((((long)dz[t] * g.Hi + dy[t]) * g.Wi + dx[t]) - tmin);
  const bool nfull = n0 + TT <= g.Nout, cfull = c0 + TT <= g.Cin;

  f32x16 acc[NTAP][WT][WT];
#pragma unroll
  for (int t = 0; t < NTAP; ++t)
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][i][j][e] = 0.f;

  uint4 hy[NPASS][QT], hx[NPASS][NTAP][QT];
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lx = (tid >> 3) & 7;
  const unsigned ylane = (unsigned)(lx * g.os * ybytes + q * 16), xlane = (unsigned)(lx * g.s * xbytes + q * 16);
  const int lxs = lx * g.s;
  auto load_step = [&](long mb) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const unsigned mw = (unsigned)mb + 32u * ps + 8u * (unsigned)wv;  // this wave's 8 rows of the pass (scalar, multiple of 8)
      const bool live = (long)mw < mend;
      const int xw = (int)(mw & (unsigned)(g.gw - 1)), y = (int)((mw >> g.sw) & (unsigned)(g.gh - 1));
      const int zb = (int)(mw >> (g.sw + g.sh)), z = zb & (g.gd - 1);
      const int dzb = zb - zb0, dyy = y - y0, dxx = xw - x0;
      const unsigned ys = (unsigned)((dzb * oz_s + dyy * oy_s + dxx * g.os) * ybytes);
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        const bool ok = live && (nfull || n0 + (q + 8 * j) * 8 + 7 < g.Nout);
        hy[ps][j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(yrs, ok ? ylane + 128u * j : OOB, ys, 0));
      }
      const int xr = dzb * xz_s + dyy * xy_s + dxx * g.s;
      const int zi = z * g.s, yi = y * g.s, xi = xw * g.s;
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        const bool line = live && tv[t] && (unsigned)(zi + dz[t]) < (unsigned)g.Di && (unsigned)(yi + dy[t]) < (unsigned)g.Hi;
        const bool in = line && (unsigned)(lxs + xi + dx[t]) < (unsigned)g.Wi;
        const unsigned xs = (unsigned)((xr + toff[t]) * xbytes);
#pragma unroll
        for (int j = 0; j < QT; ++j) {
          const bool ok = in && (cfull || c0 + (q + 8 * j) * 8 + 7 < g.Cin);
          hx[ps][t][j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? xlane + 128u * j : OOB, xs, 0));
        }
      }
    }
  };

  const int gq = lane >> 4, li = lane & 15;
  const int trow = 8 * (gq >> 1) + (li >> 2), tcol = 16 * (gq & 1) + 4 * (li & 3);
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  auto tr8 = [&](const __bf16* base) -> bf16x8 {
    union {
      s16x4 h[2];
      bf16x8 f;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * LDT));
    return u.f;
  };
  if (mbeg < mend) load_step(mbeg);
  for (long mb = mbeg; mb < mend; mb += KMH) {
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
      for (int j = 0; j < QT; ++j) {
        *(uint4*)(Yh + (32 * ps + r) * LDT + (q + 8 * j) * 8) = hy[ps][j];
#pragma unroll
        for (int t = 0; t < NTAP; ++t) *(uint4*)(Xh + (t * KMH + 32 * ps + r) * LDT + (q + 8 * j) * 8) = hx[ps][t][j];
      }
    __syncthreads();
    if (mb + KMH < mend) load_step(mb + KMH);
#pragma unroll
    for (int ks = 0; ks < KMH / 16; ++ks) {
      bf16x8 ha[WT];
#pragma unroll
      for (int i = 0; i < WT; ++i) ha[i] = tr8(Yh + (ks * 16 + trow) * LDT + wn * (TT / 2) + i * 32 + tcol);
#pragma unroll
      for (int t = 0; t < NTAP; ++t) {
        bf16x8 hb[WT];
#pragma unroll
        for (int j = 0; j < WT; ++j) hb[j] = tr8(Xh + (t * KMH + ks * 16 + trow) * LDT + wc * (TT / 2) + j * 32 + tcol);
#pragma unroll
        for (int i = 0; i < WT; ++i)
#pragma unroll
          for (int j = 0; j < WT; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[i], hb[j], acc[t][i][j], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    if (!tv[t]) continue;
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
      for (int j = 0; j < WT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int n = n0 + wn * (TT / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          const int c = c0 + wc * (TT / 2) + j * 32 + (lane & 31);
          if (n < g.Nout && c < g.Cin) atomicAdd(dW + ((long)widx[t] * g.Nout + n) * g.Cin + c, acc[t][i][j][e]);
        }
  }
}

// ---------------------------------------------------------------- weight gradient of a dense 1^3 convolution, thin side
// dW[n][c] = sum_m dY[m][n] * X[m][c] when one side has 64 channels and the other 256 (layer-1 Bottlenecks): the
// square 64 x 64 tile of k_wgrad leaves every wave a single MFMA tile (2 fragment reads per MFMA, ~80 TFLOP/s, and
// the 64-channel operand re-read by four blocks).  Here ONE block holds the whole TN x TC = 256 x 64 (or 64 x 256)
// gradient, every wave a 64 x 64 part of it (2 x 2 MFMA tiles, 1 fragment read per MFMA), and both operands stream
// through exactly once; the voxel range is split over blocks (XCD-aware numbering as in k_wgrad), partial sums meet
// in fp32 atomics.  No taps, no coordinate decode: row m of X and row m of dY belong together.
template <int TN, int TC>
__global__ __launch_bounds__(CT) void k_wgrad_dense(const float* __restrict__ X, const float* __restrict__ dY,
                                                    float* __restrict__ dW, long M, int Nout, int Cin, int msplit,
                                                    int tiles_c) {
  constexpr int QY = TN / 4, QX = TC / 4;            // float4 per staged row
  constexpr int PY = CT / QY, PX = CT / QX;          // rows per staging pass
  constexpr int RY = WG_KM / PY, RX = WG_KM / PX;    // rows per thread and step
  constexpr int WN_ = TN >= TC ? 4 : 1, WC_ = 4 / WN_;  // wave grid: every wave owns 64 x 64
  static_assert(TN / WN_ == 64 && TC / WC_ == 64, "64 x 64 per wave");
  __shared__ __attribute__((aligned(16))) float Ys[WG_KM * TN];
  __shared__ __attribute__((aligned(16))) float Xs[WG_KM * TC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned bid = blockIdx.x;
  if ((gridDim.x & 7u) == 0u) bid = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int tiles_total = gridDim.x / msplit;
  const int bid_tile = bid % tiles_total, bid_split = bid / tiles_total;
  const int n0 = (bid_tile / tiles_c) * TN, c0 = (bid_tile % tiles_c) * TC;
  const long chunk = ((M + msplit - 1) / msplit + WG_KM - 1) / WG_KM * WG_KM;
  const long mbeg = (long)bid_split * chunk, mend = mbeg + chunk < M ? mbeg + chunk : M;
  const int yq = tid % QY, yr = tid / QY, xq = tid % QX, xr = tid / QX;
  const int wn = wave / WC_, wc = wave % WC_;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 vy[RY], vx[RX];
  auto load_step = [&](long mb) {
#pragma unroll
    for (int h = 0; h < RY; ++h) {
      const long m = mb + yr + PY * h;
      vy[h] = m < mend ? *(const float4*)(dY + m * Nout + n0 + yq * 4) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int h = 0; h < RX; ++h) {
      const long m = mb + xr + PX * h;
      vx[h] = m < mend ? *(const float4*)(X + m * Cin + c0 + xq * 4) : make_float4(0, 0, 0, 0);
    }
  };
  if (mbeg < mend) load_step(mbeg);
  for (long mb = mbeg; mb < mend; mb += WG_KM) {
    __syncthreads();  // fragment reads of the previous step are done
#pragma unroll
    for (int h = 0; h < RY; ++h) *(float4*)(Ys + (yr + PY * h) * TN + yq * 4) = vy[h];
#pragma unroll
    for (int h = 0; h < RX; ++h) *(float4*)(Xs + (xr + PX * h) * TC + xq * 4) = vx[h];
    __syncthreads();
    if (mb + WG_KM < mend) load_step(mb + WG_KM);
    float fa[2][2], fb[2][2];
    auto frag = [&](int set, int kk) {
      const int mrow = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[set][i] = Ys[mrow * TN + wn * 64 + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[set][j] = Xs[mrow * TC + wc * 64 + j * 32 + (lane & 31)];
    };
    frag(0, 0);
#pragma unroll
    for (int kk = 0; kk < WG_KM / 2; ++kk) {
      const int cur = kk & 1;
      if (kk + 1 < WG_KM / 2) frag(cur ^ 1, kk + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int c = c0 + wc * 64 + j * 32 + (lane & 31);
        atomicAdd(dW + (long)n * Cin + c, acc[i][j][r]);
      }
}

// ---------------------------------------------------------------- stem data gradient
// dX[j + k - 3] += sum_n dZ[j][n] * w[n][k]   (7^3 taps, 64 channels -> 1 channel)
// As an implicit GEMM this has N = 1, so it is computed the other way round: per 4x4x8 voxel
// patch, P[j][k] = dZ[j][:] . w[:][k] is a dense 128 x 64 GEMM per tap chunk on the matrix
// cores, followed by a col2im fold of the tap columns into a 10x10x14 LDS patch that is
// flushed to dX with one fp32 atomic per touched voxel.
// LDS float atomics are ~100x slower than plain LDS traffic on gfx950, so the fold is a
// gather: a chunk is four (kd, kh) tap rows x (7 kw, padded to 8) = 32 columns; each wave (one z slice
// of the patch) parks its 32x32 P tile in a staging array addressed by OUTPUT cell
// [row][jy][cx = jx + kw][kw], then every lane owns output cells and sums their <= 4 x 7 terms.
constexpr int SP_Z = 4, SP_Y = 4, SP_X = 8, SP_M = SP_Z * SP_Y * SP_X;  // 128 voxels
constexpr int SR_Z = SP_Z + 6, SR_Y = SP_Y + 6, SR_X = SP_X + 6, SR_N = SR_Z * SR_Y * SR_X;
constexpr int SLD = 65;
constexpr int SG_KW = 9;                                  // padded kw slots per cell
constexpr int SG_N = 4 * SP_Y * SR_X * SG_KW;             // staging floats per wave
constexpr int SD_CHUNKS = 12;                             // 48 (kd, kh) tap rows, four per chunk; the 49th rides in the spare columns

template <bool HB>
__global__ __launch_bounds__(CT, 3) void k_stem_dgrad(const float* __restrict__ dZ, const float* __restrict__ Wt,
                                                   float* __restrict__ dX, int D, int H, int W, int pz, int py, int px,
                                                   int zchunk) {
  // HB: the patch GEMM runs on the bf16 matrix cores (bf16 modes): the weight tile is kept as bf16 rows of SHD = 72 elements
  // (144 bytes: the 16-byte fragment reads of 16 rows fall on 16 distinct 16-byte slots), operands rounded on their way in,
  // 4 v_mfma_f32_32x32x16_bf16 per chunk instead of 32 fp32 ones -- the fold, unchanged, then sets the pace.
  constexpr int SHD = 72;
  __shared__ __attribute__((aligned(16))) float Bs[HB ? 32 * SHD / 2 : 32 * SLD];
  __bf16* const Bh = (__bf16*)Bs;
  __shared__ float stage[4 * SG_N];
  __shared__ float patch[SR_N];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // A workgroup walks a run of patches along z in its (y, x) column.  The output patch is a ring of 10 planes: moving
  // on by one patch (4 planes) completes the 4 oldest planes -- only those are flushed to dX (fp32 atomics: the halo in
  // y and x is shared with the neighbouring columns) and become the 4 newest, zeroed; the other 6 carry over.  The dZ
  // tile of the next patch is fetched into registers while this one is multiplied and folded.
  int t = blockIdx.x;
  const int bx = t % px;
  t /= px;
  const int by = t % py;
  const int zs = t / py;
  const int bz_beg = zs * zchunk, bz_end = min(pz, bz_beg + zchunk);
  const int b = blockIdx.y;
  const int y0 = by * SP_Y, x0 = bx * SP_X;
  for (int i = tid; i < SR_N; i += CT) patch[i] = 0.f;
  // The dZ tile (128 voxels x 64 channels) never passes through LDS: the K order of the patch GEMM is free as long as both
  // operands agree, so lane (voxel row = lane & 31 of this wave's z slice, half = lane >> 5) takes channels 32 * half ..
  // 32 * half + 31 -- 128 contiguous bytes, eight 16-byte buffer loads with a lane-fixed offset -- and these ARE its 32
  // (fp32) / 4 (bf16) A operands for all 13 chunks; the weight fragment of step kk is row 32 * half + kk of the weight tile.
  // A voxel row outside the volume in y / x has its offset out of range, a z slice outside an empty descriptor: zeros.
  const int a_ry = (lane & 31) >> 3, a_rx = lane & 7;
  const unsigned avoff = (y0 + a_ry < H && x0 + a_rx < W) ? (unsigned)((((wave * H + a_ry) * W + a_rx) * 64 + 32 * (lane >> 5)) * 4) : 0x80000000u;
  float4 areg[8];
  auto fetch_a = [&](int bz) {
    const float* base = dZ + ((((long)b * D + bz * SP_Z) * H + y0) * W + x0) * 64;
    const long base_el = ((((long)b * D + bz * SP_Z) * H + y0) * W + x0) * 64;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(
        (void*)base, 0, bz * SP_Z + wave < D ? hp_extent((long)gridDim.y * D * H * W * 64, base_el, 4) : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < 8; ++j) areg[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(ars, avoff + 16u * j, 0, 0));
  };
  if (bz_beg < bz_end) fetch_a(bz_beg);
  int ring = 0;  // ring slot of the patch's first plane (z0 - 3)
  const __bf16* const bhp = Bh + (lane & 31) * SHD + 32 * (lane >> 5);
  const float* bp = Bs + (lane & 31) * SLD + 32 * (lane >> 5);
  float* sw = stage + wave * SG_N;
  const int col = lane & 31, gi_l = col >> 3, kw_l = col & 7, half = lane >> 5;
  // staging write address of accumulator register r: lane part + compile-time register part
  const int sw_lane = ((gi_l * SP_Y) * SR_X + 4 * half + kw_l) * SG_KW + kw_l;
  // Tap columns: the 49 (kd, kh) rows of 7 kw taps are numbered flat = kd * 7 + kh and taken four at a time, each padded to 8
  // columns -- 12 chunks of 32 columns hold rows 0 .. 47 (round 2: 14 chunks, one kd per chunk pair with a 3-row second half).
  // The 49th row (kd = kh = 6) rides in the padding: column (row 0 of the chunk, slot 7) of chunk c < 7 carries its tap kw = c.
  // Wt is [flat][kw][64], so a chunk's weight tile is one contiguous run: weight chunk c+1 is fetched into registers while
  // chunk c is multiplied and folded, as two 16-byte buffer loads per thread with a lane-fixed offset and the chunk as the
  // scalar offset (the 16 threads of the spare column step through row 48 instead).  The other slot-7 columns re-read kw = 6
  // and the spare column of chunks 7 .. 11 reads past the end (zeros): neither is ever parked or folded.
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, 343 * 64 * 4, 0x00020000);
  unsigned wvoff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int i = tid + h * CT;  // 512 float4 = 32 columns x 16 quads
    const int tr = i >> 4, q = i & 15;
    wvoff[h] = tr == 7 ? (unsigned)((48 * 7 * 64 + q * 4) * 4) : (unsigned)((((tr >> 3) * 7 + min(tr & 7, 6)) * 64 + q * 4) * 4);
  }
  const int wvstep = (tid >> 4) == 7 ? (64 - 4 * 7 * 64) * 4 : 0;   // spare column: tap kw = chunk of row 48, against the scalar offset
  float4 wv[2];
  auto fetch_w = [&](int chunk) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      wv[h] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrs, h == 0 ? (unsigned)((int)wvoff[0] + chunk * wvstep) : wvoff[1],
                                                                                chunk * (4 * 7 * 64 * 4), 0));
  };
  // Fold bookkeeping, fixed per lane: lane < 56 owns the staging position (jy, cx) of this wave's slice and sums, for each
  // of the chunk's four rows gi, the 7 kw slots parked there: sum_kw P[(jy, cx - kw)][(gi, kw)] -- the part of output cell
  // (row jy + kh, cx) that comes from voxel row jy.  Slots (cx, kw) with cx - kw outside the 8-voxel run are never written
  // by the park and stay zero from the start, so no term needs a validity test.  The four parts of a lane then go to four
  // different patch rows, one row at a time (lanes of one step hit distinct cells; LDS operations of a wave are ordered);
  // a part's cell is (plane, row) of its flat row -- scalar -- plus the lane number.
  const bool f_on = lane < SP_Y * SR_X;
  const int fbase = f_on ? lane * SG_KW : 0;
  for (int i = tid; i < 4 * SG_N; i += CT) stage[i] = 0.f;
  // The spare column's P values (row 48, tap kw = c in chunk c) are parked in the staging slots 7 and 8 that the regular
  // columns leave free: chunk c writes plane (row c & 3, slot 7 + (c >> 2)) at cx = jx + c -- every (plane, cell) is written
  // by one chunk only, so the cells it does not reach stay zero as well, and the last step of a patch sums the seven planes.
  const bool p48 = col == 7;
  // this wave's rows of the dZ tile stay in registers for all 12 chunks of a patch
  float afr[HB ? 1 : 32];
  bf16x8 ha[HB ? 4 : 1];
  // Software pipeline over the 12 tap chunks: while the matrix cores run chunk c (a dependent chain of 32 MFMAs
  // on one accumulator tile), the same wave folds the P tile of chunk c-1 that it parked in its private staging
  // array.  Next to fp32 MFMAs every other vector instruction of the SIMD costs matrix-pipe time, so the fold
  // is kept short: 16 LDS reads (pairs), 16 adds (packed), 4 read-add-write steps with scalar cell addresses.  Workgroup
  // barriers are needed only around the shared weight tile; the patch planes touched by the four waves within one step
  // are distinct.
  auto step = [&](auto mm_c, auto fold_c, int chunk) {
    constexpr bool MM = decltype(mm_c)::value, FOLD = decltype(fold_c)::value;
    const int pc = chunk - 1;  // the chunk being folded
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x2 fp2[4][3];
    float fp1[4], fpart[4];
    float fb[2] = {0.f, 0.f}, nb[2] = {0.f, 0.f};   // weight operands, read a pair of K steps ahead (one ds_read2 per pair)
    bf16x8 hb[HB ? 4 : 1];
    if constexpr (MM && HB) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) hb[ks] = *(const bf16x8*)(bhp + ks * 8);
    }
    if constexpr (MM && !HB) {
      fb[0] = bp[0];
      fb[1] = bp[1];
    }
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) {
      if constexpr (MM && !HB) {
        if ((kk & 1) == 0 && kk + 2 < 32) {
          nb[0] = bp[kk + 2];
          nb[1] = bp[kk + 3];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[kk], fb[kk & 1], acc, 0, 0, 0);
      }
      if constexpr (MM && HB) {  // the 4 K steps of the chunk, spread over the fold slots
        if ((kk & 7) == 0) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[kk >> 3], hb[kk >> 3], acc, 0, 0, 0);
      }
      if constexpr (FOLD) {
        // slots 0 .. 15: the 7 kw terms of row gi = kk / 4 as three pairs and a single (unconditional staging reads);
        // slots 16 .. 19: their sum; slots 20 .. 23: the part of row gi goes to its patch row of this wave's plane
        if (kk < 16) {
          const int gi = kk >> 2, j = kk & 3;  // compile-time after unrolling
          const float* src = sw + fbase + gi * (SP_Y * SR_X * SG_KW) + 2 * j;
          if (j < 3) fp2[gi][j] = (f32x2){src[0], src[1]};
          else fp1[gi] = src[0];
        } else if (kk < 20) {
          const int gi = kk - 16;
          const f32x2 s2 = fp2[gi][0] + fp2[gi][1] + fp2[gi][2];
          fpart[gi] = (s2[0] + s2[1]) + fp1[gi];
        } else if (kk < 24) {
          const int gi = kk - 20;
          const int flat = 4 * pc + gi;  // scalar
          const int kdp = flat / 7, khp = flat - 7 * kdp;
          int slot = ring + wave + kdp;  // plane (wave + kd) of the patch
          slot = slot >= SR_Z ? slot - SR_Z : slot;
          if (f_on) patch[(slot * SR_Y + khp) * SR_X + lane] += fpart[gi];
          // the next row's update of ANOTHER lane reads the cell this lane just wrote: keep the four read-add-write
          // steps in program order (per lane their addresses differ, so the compiler would otherwise be free to hoist
          // the later reads above this write)
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr (MM && !HB) {
        if (kk & 1) {
          fb[0] = nb[0];
          fb[1] = nb[1];
        }
      }
      // fp32: pin the fold slot to its MFMA; bf16: only 4 MFMAs per chunk -- the compiler is free to batch the fold's reads
      if constexpr (!HB) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MM) {
      // park P: register r is voxel (jy = r>>2, jx = (r&3) + 4*half) of this wave's z slice
      const int s48 = (((chunk & 3) * SP_Y) * SR_X + chunk) * SG_KW + 7 + (chunk >> 2);  // scalar
      const int swl = p48 ? s48 + 4 * half * SG_KW : sw_lane;
      if (p48 ? chunk < 7 : kw_l < 7) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sw[swl + ((r >> 2) * SR_X + (r & 3)) * SG_KW] = acc[r];
      }
    }
    if constexpr (FOLD && !MM) {
      // last step of the patch: the 49th tap row -- output row 6 + jy of plane wave + 6
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < 7; ++c) s += sw[fbase + ((c & 3) * SP_Y * SR_X) * SG_KW + 7 + (c >> 2)];
      int slot = ring + wave + 6;
      slot = slot >= SR_Z ? slot - SR_Z : slot;
      if (f_on) patch[(slot * SR_Y + 6) * SR_X + lane] += s;
    }
  };
  for (int bz = bz_beg; bz < bz_end; ++bz) {
  const int z0 = bz * SP_Z;
  if constexpr (HB) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const bf16x4 lo = to_bf16x4(areg[2 * ks]), hi = to_bf16x4(areg[2 * ks + 1]);
      ha[ks] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      afr[4 * j] = areg[j].x;
      afr[4 * j + 1] = areg[j].y;
      afr[4 * j + 2] = areg[j].z;
      afr[4 * j + 3] = areg[j].w;
    }
  }
  if (bz + 1 < bz_end) fetch_a(bz + 1);
  fetch_w(0);
  for (int chunk = 0; chunk <= SD_CHUNKS; ++chunk) {
    __syncthreads();  // every wave is through the fragment reads of the previous weight tile and its patch update
    if (chunk < SD_CHUNKS) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int i = tid + h * CT;
        const int tr = i >> 4, q = i & 15;
        if constexpr (HB) {
          *(bf16x4*)(Bh + tr * SHD + q * 4) = to_bf16x4(wv[h]);
        } else {
          float* d = Bs + tr * SLD + q * 4;
          d[0] = wv[h].x;
          d[1] = wv[h].y;
          d[2] = wv[h].z;
          d[3] = wv[h].w;
        }
      }
    }
    __syncthreads();
    if (chunk + 1 < SD_CHUNKS) fetch_w(chunk + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (chunk == 0) step(std::true_type{}, std::false_type{}, chunk);
    else if (chunk < SD_CHUNKS) step(std::true_type{}, std::true_type{}, chunk);
    else step(std::false_type{}, std::true_type{}, chunk);
  }
  __syncthreads();
  // planes z0 - 3 .. z0 are complete (the next patch starts at z0 + 1); the last patch of the run flushes all 10
  const int nfl = (bz + 1 < bz_end ? SP_Z : SR_Z) * SR_Y * SR_X;
  for (int i = tid; i < nfl; i += CT) {
    const int cz = i / (SR_Y * SR_X), rem = i - cz * (SR_Y * SR_X), cy = rem / SR_X, cx = rem - cy * SR_X;
    const int z = z0 + cz - 3, y = y0 + cy - 3, x = x0 + cx - 3;
    int slot = ring + cz;
    slot = slot >= SR_Z ? slot - SR_Z : slot;
    float* const cell = patch + slot * (SR_Y * SR_X) + rem;
    if ((unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
      atomicAdd(dX + (((long)b * D + z) * H + y) * W + x, *cell);
    *cell = 0.f;
  }
  ring = ring + SP_Z >= SR_Z ? ring + SP_Z - SR_Z : ring + SP_Z;
  }
}



// ---------------------------------------------------------------- stem weight gradient, whole dW per workgroup
// dW[co][tap] = sum_v dZ[v][co] * x[v + tap]  (64 x 343).  As a GEMM its output is only 64 x 352, so ONE
// workgroup can hold all of it in accumulators (2 x 11 MFMA tiles over 4 waves): dZ -- the 8.6 GB operand -- is
// then read exactly once instead of once per 64-wide tap tile, and the tap operand needs no gathered tile at
// all: per 4x4x8-voxel tile the input patch (10 x 10 x 14) sits in LDS and lane (tap, half) reads
// patch[voxel(k) + offset(tap)] directly.  Patch pitches 39 (row) and 401 (plane) are = 7 and 49 mod 32, so the
// 32 consecutive taps of a fragment read fall on 32 consecutive banks.  A workgroup walks a contiguous range of
// tiles, keeps its partial dW in registers and adds it to memory once at the end (fp32 atomics).
constexpr int SWG_PR = 39, SWG_PP = 401, SWG_PATCH = 10 * SWG_PP, SWG_LDY = 65;

__global__ __launch_bounds__(256, 2) void k_stem_wgrad_full(const float* __restrict__ X, const float* __restrict__ dZ,
                                                            float* __restrict__ dW, int D, int H, int W, int pz, int py,
                                                            int px, long tiles_total, int tiles_per_wg, int kpad) {
  __shared__ float Ys[SP_M * SWG_LDY];
  __shared__ float patch[SWG_PATCH + 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  // this wave's tap tiles: wave, wave + 4, wave + 8 (11 tiles of 32 taps cover 352 >= 343)
  int toff[3];
  bool tval[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tap = 32 * (wave + 4 * j) + col;
    tval[j] = wave + 4 * j < 11 && tap < 343;
    const int a = tap / 49, bb = (tap / 7) % 7, c = tap % 7;
    toff[j] = tval[j] ? a * SWG_PP + bb * SWG_PR + c : 0;
  }
  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (tid < 8) patch[SWG_PATCH + tid] = 0.f;
  const float* const ysrc = Ys + half * SWG_LDY + col;   // this lane's dZ fragment source: + voxel pair * 2 * SWG_LDY
  const float* psrc[3];                                   // this lane's tap cell of the patch origin, per tap tile: + vbase(voxel pair)
#pragma unroll
  for (int j = 0; j < 3; ++j) psrc[j] = patch + toff[j] + half;

  const long t0 = (long)blockIdx.x * tiles_per_wg, t1 = min(tiles_total, t0 + tiles_per_wg);
  // tile t+1 is fetched into registers while tile t is multiplied
  float4 zreg[SP_M / 16];
  float preg[6];
  const int zq = tid & 15, zr0 = tid >> 4;
  auto fetch = [&](long t) {
    long r_ = t;
    const int bx = (int)(r_ % px);
    r_ /= px;
    const int by = (int)(r_ % py);
    r_ /= py;
    const int bz = (int)(r_ % pz);
    const int b = (int)(r_ / pz);
    const int z0 = bz * SP_Z, y0 = by * SP_Y, x0 = bx * SP_X;
#pragma unroll
    for (int pss = 0; pss < SP_M / 16; ++pss) {  // dZ tile: 128 voxels x 64 channels
      const int r = zr0 + 16 * pss;
      const int z = z0 + (r >> 5), y = y0 + ((r >> 3) & 3), x = x0 + (r & 7);
      zreg[pss] = (z < D && y < H && x < W) ? *(const float4*)(dZ + ((((long)b * D + z) * H + y) * W + x) * 64 + zq * 4)
                                           : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {  // input patch with its 3-voxel halo
      const int e = tid + 256 * k;
      const int pzz = e / 140, rem = e - pzz * 140, pyy = rem / 14, pxx = rem - pyy * 14;
      const int z = z0 + pzz - 3, y = y0 + pyy - 3, x = x0 + pxx - 3;
      preg[k] = (e < 1400 && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                    ? X[(((long)b * D + z) * H + y) * W + x]
                    : 0.f;
    }
  };
  if (t0 < t1) fetch(t0);
  for (long t = t0; t < t1; ++t) {
    __syncthreads();  // previous tile's fragment reads are done
#pragma unroll
    for (int pss = 0; pss < SP_M / 16; ++pss) {
      float* d = Ys + (zr0 + 16 * pss) * SWG_LDY + zq * 4;
      d[0] = zreg[pss].x;
      d[1] = zreg[pss].y;
      d[2] = zreg[pss].z;
      d[3] = zreg[pss].w;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int e = tid + 256 * k;
      const int pzz = e / 140, rem = e - pzz * 140, pyy = rem / 14, pxx = rem - pyy * 14;
      if (e < 1400) patch[pzz * SWG_PP + pyy * SWG_PR + pxx] = preg[k];
    }
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    // The K loop is fully unrolled with the lane part of every LDS address fixed for the whole kernel (ysrc, psrc[j]) and the
    // voxel part a compile-time immediate of the ds_read: voxel v = 2 kk + half lies at vbase(2 kk) + half (2 kk is even, so
    // the + 1 of the upper half-wave never carries out of the x field).  Computing the address per read -- add, select of
    // the zero cell for taps >= 343, shift -- was 42 vector instructions per 24 MFMAs (SQ_INSTS_VALU: 2.5 per MFMA), each
    // ~6 cycles of matrix-pipe time next to fp32 MFMAs.  Taps >= 343 now read tap 0's cell: their products are never stored.
#pragma unroll
    for (int kk = 0; kk < SP_M / 2; ++kk) {
      constexpr int ZZ = 0;  // (placeholder so that the lambdas below see constant expressions after unrolling)
      const int ve = 2 * kk + ZZ;  // even voxel of the pair: (z, y, x) = (ve >> 5, (ve >> 3) & 3, ve & 7)
      const int vbase = (ve >> 5) * SWG_PP + ((ve >> 3) & 3) * SWG_PR + (ve & 7);
      float fa[2], fb[3];
      fa[0] = ysrc[ve * SWG_LDY];
      fa[1] = ysrc[ve * SWG_LDY + 32];
#pragma unroll
      for (int j = 0; j < 3; ++j) fb[j] = psrc[j][vbase];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      if ((kk & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // groups of 24 MFMAs: the scheduler must not hoist all 320 reads
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (wave + 4 * j >= 11) continue;
    const int tap = 32 * (wave + 4 * j) + col;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (tap < 343) atomicAdd(dW + (long)co * kpad + tap, acc[i][j][r]);
      }
  }
}

// ---------------------------------------------------------------- stem weight gradient on the bf16 matrix cores
// The same decomposition as k_stem_wgrad_full (one workgroup holds the whole 64 x 352 gradient, dZ is read once, the
// tap operand comes straight from the input patch in LDS) for the bf16 modes: v_mfma_f32_32x32x16_bf16 takes 16 voxels
// per instruction, 8 consecutive ones per lane.  A = dZ^T: the tile is staged [voxel][channel] as it lies in memory and
// read through ds_read_b64_tr_b16 (as in k_wgrad).  B[voxel][tap] = x[voxel + tap]: the 8 consecutive voxels of a lane
// are the 8-voxel x-run of one (z, y) row of the 4 x 4 x 8 tile, i.e. 8 consecutive bf16 of the patch starting at
// an arbitrary element -- the patch is therefore kept in FOUR copies shifted by 0..3 elements, so that every run is two
// 8-byte-aligned ds_read_b64 of copy (start & 3).  Patch pitches 28 / 281 keep those reads at <= 2 lanes per bank.
// Operands are rounded to bf16 on their way into LDS, accumulation is fp32.
constexpr int SWH_PR = 28, SWH_PP = 281, SWH_COPY = 738 * 4 /* bf16 per copy: 736 + 2 quad-words, the bank shift between copies */, SWH_LDT = 96;

__global__ __launch_bounds__(256, 2) void k_stem_wgrad_bf16(const float* __restrict__ X, const float* __restrict__ dZ,
                                                            float* __restrict__ dW, int D, int H, int W, int pz, int py,
                                                            int px, long tiles_total, int tiles_per_wg, int kpad) {
  __shared__ __attribute__((aligned(16))) __bf16 Yh[SP_M * SWH_LDT];
  __shared__ __attribute__((aligned(16))) __bf16 patch[4 * SWH_COPY + 16];   // last 16: zeros for invalid taps
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  // this wave's tap tiles: wave, wave + 4, wave + 8 (11 tiles of 32 taps cover 352 >= 343)
  int toff[3];
  bool tval[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tap = 32 * (wave + 4 * j) + col;
    tval[j] = wave + 4 * j < 11 && tap < 343;
    const int a = tap / 49, bb = (tap / 7) % 7, c = tap % 7;
    toff[j] = tval[j] ? a * SWH_PP + bb * SWH_PR + c : 0;
  }
  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (tid < 16) patch[4 * SWH_COPY + tid] = (__bf16)0.f;

  const long t0 = (long)blockIdx.x * tiles_per_wg, t1 = min(tiles_total, t0 + tiles_per_wg);
  // tile t+1 is fetched into registers while tile t is multiplied
  float4 zreg[SP_M / 16];
  float preg[6];
  const int zq = tid & 15, zr0 = tid >> 4;
  auto fetch = [&](long t) {
    long r_ = t;
    const int bx = (int)(r_ % px);
    r_ /= px;
    const int by = (int)(r_ % py);
    r_ /= py;
    const int bz = (int)(r_ % pz);
    const int b = (int)(r_ / pz);
    const int z0 = bz * SP_Z, y0 = by * SP_Y, x0 = bx * SP_X;
#pragma unroll
    for (int pss = 0; pss < SP_M / 16; ++pss) {  // dZ tile: 128 voxels x 64 channels
      const int r = zr0 + 16 * pss;
      const int z = z0 + (r >> 5), y = y0 + ((r >> 3) & 3), x = x0 + (r & 7);
      zreg[pss] = (z < D && y < H && x < W) ? *(const float4*)(dZ + ((((long)b * D + z) * H + y) * W + x) * 64 + zq * 4)
                                           : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {  // input patch with its 3-voxel halo
      const int e = tid + 256 * k;
      const int pzz = e / 140, rem = e - pzz * 140, pyy = rem / 14, pxx = rem - pyy * 14;
      const int z = z0 + pzz - 3, y = y0 + pyy - 3, x = x0 + pxx - 3;
      preg[k] = (e < 1400 && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                    ? X[(((long)b * D + z) * H + y) * W + x]
                    : 0.f;
    }
  };
  // fragment addressing of the transposing read (see k_wgrad): 16-lane group gq -> voxels 8*(gq>>1).., channels 16*(gq&1)..
  const int gq = lane >> 4, li = lane & 15;
  const int trow = 8 * (gq >> 1) + (li >> 2), tcol = 16 * (gq & 1) + 4 * (li & 3);
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  auto tr8 = [&](const __bf16* base) -> bf16x8 {
    union {
      s16x4 h[2];
      bf16x8 f;
    } u;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base + 4 * SWH_LDT));
    return u.f;
  };
  if (t0 < t1) fetch(t0);
  for (long t = t0; t < t1; ++t) {
    __syncthreads();  // previous tile's fragment reads are done
#pragma unroll
    for (int pss = 0; pss < SP_M / 16; ++pss) *(bf16x4*)(Yh + (zr0 + 16 * pss) * SWH_LDT + zq * 4) = to_bf16x4(zreg[pss]);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int e = tid + 256 * k;
      const int pzz = e / 140, rem = e - pzz * 140, pyy = rem / 14, pxx = rem - pyy * 14;
      if (e < 1400) {
        const int pi = pzz * SWH_PP + pyy * SWH_PR + pxx;
        const __bf16 v = (__bf16)preg[k];
#pragma unroll
        for (int sft = 0; sft < 4; ++sft)
          if (pi - sft >= 0) patch[sft * SWH_COPY + pi - sft] = v;   // copy s holds element i + s at index i
      }
    }
    __syncthreads();
    if (t + 1 < t1) fetch(t + 1);
#pragma unroll 2
    for (int ks = 0; ks < SP_M / 16; ++ks) {
      // K = 16 voxels: (z, y) rows 2 ks and 2 ks + 1 of the tile, 8 voxels along x each; this lane's row:
      const int rr = 2 * ks + half;
      const int vbase = (rr >> 2) * SWH_PP + (rr & 3) * SWH_PR;
      bf16x8 ha[2], hb[3];
#pragma unroll
      for (int i = 0; i < 2; ++i) ha[i] = tr8(Yh + (ks * 16 + trow) * SWH_LDT + i * 32 + tcol);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int e = vbase + toff[j];
        const __bf16* src = tval[j] ? patch + (e & 3) * SWH_COPY + (e & ~3) : patch + 4 * SWH_COPY;
        union {
          uint2 q[2];
          bf16x8 f;
        } u;
        u.q[0] = *(const uint2*)(src);
        u.q[1] = *(const uint2*)(src + 4);
        hb[j] = u.f;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[i], hb[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (wave + 4 * j >= 11) continue;
    const int tap = 32 * (wave + 4 * j) + col;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (tap < 343) atomicAdd(dW + (long)co * kpad + tap, acc[i][j][r]);
      }
  }
}

// ---------------------------------------------------------------- stem forward on the bf16 matrix cores
// out[v][n] = sum_tap x[v + tap] w[n][tap] for the bf16 modes.  K = taps, ordered (kd, kh) pair by pair with the 7 kw taps of a
// pair padded to 8 (zero weight): the 8 consecutive K elements a lane feeds to v_mfma_f32_32x32x16_bf16 are then 8
// consecutive voxels of one input row -- two aligned ds_read_b64 of the four-times-shifted input patch of k_stem_wgrad_bf16 --
// and a K step of 16 is two (kd, kh) pairs (25 steps for 49 pairs + 1 empty).  Rows of an MFMA tile = a 4 (y) x 8 (x) slice of
// voxels, columns = 32 output channels; a wave owns two z slices x 64 channels.  Workgroup = 8 waves = a 16 x 4 x 8 voxel tile
// (22 x 10 x 14 patch); the weights sit in LDS as [pair][channel][8] bf16 (50 KB) for the whole run of tiles of a persistent
// workgroup, the next patch is fetched into registers during the multiply, BatchNorm statistics collect in registers and
// leave in one fp64 atomic per channel and workgroup.  fp32 in, fp32 out, operands rounded to bf16 on their way into LDS.
constexpr int SFH_TZ = 16, SFH_TY = 4, SFH_TX = 8, SFH_PZ = SFH_TZ + 6, SFH_PR = 28, SFH_PP = 10 * SFH_PR;
constexpr int SFH_COPY = SFH_PZ * SFH_PP + 8;           // bf16 per shifted copy (+ 2 quad-words: bank shift between copies)
constexpr int SFH_W = 50 * 64 * 8;                      // weight image, bf16
constexpr int SFH_NP = SFH_PZ * 10 * 14;                // patch elements
constexpr int SFH_LD = (SFH_NP + 511) / 512;            // patch elements per thread

__global__ __launch_bounds__(512) void k_stem_fwd_bf16(const float* __restrict__ X, const float* __restrict__ Wp,
                                                       float* __restrict__ Y, double* __restrict__ stats, int D, int H, int W,
                                                       int kpad, int pz, int py, int px, long tiles_total, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float sfh_smem[];
  __bf16* const wl = (__bf16*)sfh_smem;
  __bf16* const patch = wl + SFH_W;
  float* const red = (float*)(patch + 4 * SFH_COPY);  // [8 waves][64 channels][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, half = lane >> 5;
  for (int sidx = tid; sidx < 50 * 64; sidx += 512) {
    const int pr = sidx >> 6, n = sidx & 63;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (pr < 49 && c < 7) ? Wp[(long)n * kpad + pr * 7 + c] : 0.f;
    *(bf16x4*)(wl + sidx * 8) = to_bf16x4(make_float4(v[0], v[1], v[2], v[3]));
    *(bf16x4*)(wl + sidx * 8 + 4) = to_bf16x4(make_float4(v[4], v[5], v[6], v[7]));
  }
  for (int i = tid; i < 4 * SFH_COPY / 2; i += 512) ((unsigned int*)patch)[i] = 0u;  // row pads stay zero for good
  const long t0 = (long)blockIdx.x * tiles_per_wg, t1 = min(tiles_total, t0 + tiles_per_wg);
  float preg[SFH_LD];
  auto tile_origin = [&](long t, int& b, int& z0, int& y0, int& x0) {
    long r_ = t;
    x0 = (int)(r_ % px) * SFH_TX;
    r_ /= px;
    y0 = (int)(r_ % py) * SFH_TY;
    r_ /= py;
    z0 = (int)(r_ % pz) * SFH_TZ;
    b = (int)(r_ / pz);
  };
  auto fetch = [&](long t) {
    int b, z0, y0, x0;
    tile_origin(t, b, z0, y0, x0);
#pragma unroll
    for (int k = 0; k < SFH_LD; ++k) {
      const int e = tid + 512 * k;
      const int pzz = e / 140, rem = e - pzz * 140, pyy = rem / 14, pxx = rem - pyy * 14;
      const int z = z0 + pzz - 3, y = y0 + pyy - 3, x = x0 + pxx - 3;
      preg[k] = (e < SFH_NP && (unsigned)z < (unsigned)D && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
                    ? X[(((long)b * D + z) * H + y) * W + x]
                    : 0.f;
    }
  };
  // A operand: lane (row = voxel (y = col >> 3, x = col & 7) of z slice 2 wave + mt, K half) starts at patch element
  // abase[mt] + (kd * PP + kh * PR) of its half's (kd, kh) pair
  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) abase[mt] = (2 * wave + mt) * SFH_PP + (col >> 3) * SFH_PR + (col & 7);
  float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
  if (t0 < t1) fetch(t0);
  for (long t = t0; t < t1; ++t) {
    __syncthreads();  // the previous tile's fragment reads are done (first trip: weights and zeroed patch are in place)
#pragma unroll
    for (int k = 0; k < SFH_LD; ++k) {
      const int e = tid + 512 * k;
      const int pzz = e / 140, rem = e - pzz * 140, pyy = rem / 14, pxx = rem - pyy * 14;
      if (e < SFH_NP) {
        const int pi = pzz * SFH_PP + pyy * SFH_PR + pxx;
        const __bf16 v = (__bf16)preg[k];
#pragma unroll
        for (int sft = 0; sft < 4; ++sft)
          if (pi - sft >= 0) patch[sft * SFH_COPY + pi - sft] = v;  // copy s holds element i + s at index i
      }
    }
    __syncthreads();
    int b, z0, y0, x0;
    tile_origin(t, b, z0, y0, x0);
    if (t + 1 < t1) fetch(t + 1);
    f32x16 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    int kd = 0, kh = half;  // pair 2 ks + half
#pragma unroll 5
    for (int ks = 0; ks < 25; ++ks) {
      const int pr = 2 * ks + half;
      const int off = pr < 49 ? kd * SFH_PP + kh * SFH_PR : 0;
      bf16x8 ha[2], hb[2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int e = abase[mt] + off;
        const __bf16* const src = patch + (e & 3) * SFH_COPY + (e & ~3);
        union {
          uint2 q[2];
          bf16x8 f;
        } u;
        u.q[0] = *(const uint2*)(src);
        u.q[1] = *(const uint2*)(src + 4);
        ha[mt] = u.f;
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) hb[nt] = *(const bf16x8*)(wl + ((pr * 64) + nt * 32 + col) * 8);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[mt], hb[nt], acc[mt][nt], 0, 0, 0);
      kh += 2;
      if (kh >= 7) {
        kh -= 7;
        ++kd;
      }
    }
    // epilogue: register r of a tile is voxel row (r & 3) + 8 (r >> 2) + 4 half, column = channel nt * 32 + col
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int z = z0 + 2 * wave + mt;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int y = y0 + (i >> 3), x = x0 + (i & 7);
        if (z < D && y < H && x < W) {
          float* const yo = Y + ((((long)b * D + z) * H + y) * W + x) * 64 + col;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const float v = acc[mt][nt][r];
            yo[nt * 32] = v;
            ssum[nt] += v;
            ssq[nt] += v * v;
          }
        }
      }
    }
  }
  if (stats) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const float sv = ssum[nt] + __shfl_xor(ssum[nt], 32), qv = ssq[nt] + __shfl_xor(ssq[nt], 32);
      if (half == 0) {
        red[(wave * 64 + nt * 32 + col) * 2 + 0] = sv;
        red[(wave * 64 + nt * 32 + col) * 2 + 1] = qv;
      }
    }
    __syncthreads();
    if (tid < 64) {
      double sv = 0.0, qv = 0.0;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        sv += (double)red[(w * 64 + tid) * 2 + 0];
        qv += (double)red[(w * 64 + tid) * 2 + 1];
      }
      double* const slot = stats + (size_t)(blockIdx.x & (HP_STATS_SLOTS - 1)) * 128;
      atomicAdd(slot + tid, sv);
      atomicAdd(slot + 64 + tid, qv);
    }
  }
}

// ---------------------------------------------------------------- stem forward on the 4x4x1 matrix-core instruction
// The stem (1 -> 64 channels, 7^3 taps) as a GEMM has K = 343 taps of a single-channel volume: the generic kernel
// gathers its A tile element by element.  With v_mfma_f32_4x4x1_16b_f32 (16 blocks of a 4x4 outer product; row i =
// output channel 4 cg + i, column j / block b = voxel 4b + j of a 64-voxel x-run) every tap needs ONE shifted
// ds_read_b32 of the input plane, shared by the 16 channel groups, and no gather at all:
//     D_b[i][j] += w[4 cg + i][tap] * x[v_{4b+j} + tap]
// A workgroup owns an 8 (y) x 64 (x) output column and slides along z: 8-slot LDS ring of padded input planes
// (14 rows x 72 columns), the whole weight tensor resident in LDS as [tap][i][cg] (88 KB; 4 broadcast 16-byte reads
// per tap), 128 accumulators per wave (2 rows x 16 groups x 4).  Epilogue per row: transpose through LDS, per-channel
// sum / sum of squares for the BatchNorm statistics, 128-byte channels-last stores.  Exact fp32 (fmaf chain per
// output, taps in ascending order as in the generic kernel).
constexpr int SF_TY = 8, SF_TX = 64, SF_RY = SF_TY + 6, SF_RX = 72, SF_PLANE = SF_RY * SF_RX;
using f32x4c = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void k_stem_fwd_mfma(const float* __restrict__ X, const float* __restrict__ Wp,
                                                       float* __restrict__ Y, double* __restrict__ stats, int D, int H, int W,
                                                       int kpad, int tiles_x, int tiles_y, int zchunk) {
  extern __shared__ __attribute__((aligned(16))) float sf_smem[];
  float* const wl = sf_smem;                       // [343][4][16]
  float* const ring = wl + 343 * 64;               // [8][SF_RY][SF_RX]
  float* const otile = ring + 8 * SF_PLANE;        // [4 waves][64 voxels][33]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 3;
  int t_ = blockIdx.x;
  const int bx = t_ % tiles_x;
  t_ /= tiles_x;
  const int by = t_ % tiles_y;
  const int bz = t_ / tiles_y;
  const int b = blockIdx.y;
  const int x0 = bx * SF_TX, y0 = by * SF_TY;
  const int zb = bz * zchunk, ze = min(D, zb + zchunk);
  const float* xb = X + (long)b * D * H * W;

  for (int e = tid; e < 343 * 64; e += 256) {
    const int cg = e & 15, i = (e >> 4) & 3, tap = e >> 6;
    wl[e] = Wp[(long)(cg * 4 + i) * kpad + tap];
  }
  // staging map of one padded plane: element e = tid + 256 k of [14][72]
  constexpr int SK = (SF_PLANE + 255) / 256;
  int soff[SK];
  unsigned smask = 0;
#pragma unroll
  for (int k = 0; k < SK; ++k) {
    const int e = tid + 256 * k;
    const int ly = e / SF_RX, lx = e - ly * SF_RX;
    const int yy = y0 + ly - 3, xx = x0 + lx - 3;
    const bool ok = e < SF_PLANE && lx < SF_TX + 6 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    soff[k] = ok ? yy * W + xx : 0;
    smask |= ok ? (1u << k) : 0u;
  }
  // a plane is REQUESTED before a step's multiply phase and parked in its ring slot after it (the slot of plane z - 4 is free
  // during step z): with one workgroup per CU (153 KB of LDS) nothing else hides the latency of a load that is waited for
  // on the spot -- the synchronous form cost a full memory round trip per 5 us step
  auto stage_load = [&](int zin, float (&v)[SK]) {
    const bool zok = (unsigned)zin < (unsigned)D;
    const float* src = xb + (long)(zok ? zin : 0) * H * W;
#pragma unroll
    for (int k = 0; k < SK; ++k) v[k] = (zok && ((smask >> k) & 1u)) ? src[soff[k]] : 0.f;
  };
  auto stage_store = [&](int zin, const float (&v)[SK]) {
    float* dst = ring + (zin & 7) * SF_PLANE + tid;
#pragma unroll
    for (int k = 0; k < SK; ++k)
      if (tid + 256 * k < SF_PLANE) dst[256 * k] = v[k];
  };
  auto stage = [&](int zin) {
    float v[SK];
    stage_load(zin, v);
    stage_store(zin, v);
  };
  if (zb < ze)
    for (int zin = zb - 3; zin <= zb + 3; ++zin) stage(zin);
  __syncthreads();

  float* const ot = otile + wave * (64 * 33);
  float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};  // this lane's channels 32 hh + (lane & 31), lanes < 32 only
  for (int z = zb; z < ze; ++z) {
    float nv[SK];
    if (z + 1 < ze) stage_load(z + 4, nv);  // replaces plane z - 4, last read one barrier ago
    __builtin_amdgcn_sched_barrier(0);      // keep the request ahead of the multiply phase
    f32x4c acc[2][16];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int cg = 0; cg < 16; ++cg) acc[r][cg] = (f32x4c){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int a = 0; a < 7; ++a) {
      const float* pl = ring + ((z + a - 3) & 7) * SF_PLANE + (2 * wave) * SF_RX + lane;
#pragma unroll
      for (int bb = 0; bb < 7; ++bb) {
        const float* wt = wl + ((a * 7 + bb) * 7) * 64 + sub * 16;
#pragma unroll
        for (int c = 0; c < 7; ++c) {
          float wr[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4c t4 = *(const f32x4c*)(wt + c * 64 + 4 * q);
            wr[4 * q] = t4[0];
            wr[4 * q + 1] = t4[1];
            wr[4 * q + 2] = t4[2];
            wr[4 * q + 3] = t4[3];
          }
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const float xv = pl[(r + bb) * SF_RX + c];
#pragma unroll
            for (int cg = 0; cg < 16; ++cg) acc[r][cg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wr[cg], xv, acc[r][cg], 0, 0, 0);
          }
        }
      }
    }
    // lane l holds out[4 cg + i][row][x0 + l] in acc[row][cg][i]
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int oy = y0 + 2 * wave + r;
      const bool row_ok = oy < H;
      const long orow = (((long)b * D + z) * H + oy) * W + x0;  // first voxel of the run
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int cg = 0; cg < 8; ++cg)
#pragma unroll
          for (int i = 0; i < 4; ++i) ot[lane * 33 + cg * 4 + i] = acc[r][8 * hh + cg][i];
        // (LDS accesses of one wave are ordered; no barrier needed for a wave-private tile)
        if (row_ok) {
          if (stats && lane < 32) {
            float s = 0.f, q = 0.f;
            const int nv = min(64, W - x0);
            for (int v = 0; v < nv; ++v) {
              const float val = ot[v * 33 + lane];
              s += val;
              q += val * val;
            }
            ssum[hh] += s;
            ssq[hh] += q;
          }
          const int q4 = lane & 7;
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int v = it * 8 + (lane >> 3);
            if (x0 + v < W) {
              const float* sp = ot + v * 33 + 4 * q4;
              *(float4*)(Y + (orow + v) * 64 + 32 * hh + 4 * q4) = make_float4(sp[0], sp[1], sp[2], sp[3]);
            }
          }
        }
      }
    }
    if (z + 1 < ze) stage_store(z + 4, nv);
    __syncthreads();
  }
  if (stats && lane < 32) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      double* const slot = stats + (size_t)(blockIdx.x & (HP_STATS_SLOTS - 1)) * 128;
      atomicAdd(slot + 32 * hh + lane, (double)ssum[hh]);
      atomicAdd(slot + 64 + 32 * hh + lane, (double)ssq[hh]);
    }
  }
}

// ---------------------------------------------------------------- weight (un)packing
// torch Conv3d weight (Cout,Cin,k,k,k)  <->  packed [tap][Cout][Cin]   (transposed=0)
// torch ConvTranspose3d weight (Cin,Cout,k,k,k) <-> packed [tap][Cout][Cin]   (transposed=1)
// `swap` writes [tap][Cin][Cout] instead (operand of the data-gradient GEMM).
__global__ void k_pack_weight(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int taps,
                              int transposed, int swap, int to_torch) {
  const long total = (long)Cout * Cin * taps;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // i enumerates the packed layout
    const int inner = (int)(i % (swap ? Cout : Cin));
    long r = i / (swap ? Cout : Cin);
    const int outer = (int)(r % (swap ? Cin : Cout));
    const int t = (int)(r / (swap ? Cin : Cout));
    const int co = swap ? inner : outer, ci = swap ? outer : inner;
    const long ti = transposed ? ((long)ci * Cout + co) * taps + t : ((long)co * Cin + ci) * taps + t;
    if (to_torch)
      ((float*)w)[ti] = out[i];
    else
      out[i] = w[ti];
  }
}

// The same mapping through a 64 x taps LDS tile so that BOTH sides move contiguous runs: the torch side in runs
// of `taps` floats (t fastest), the packed side in runs of 64 inner channels.  grid (inner chunks of 64, outer).
__global__ __launch_bounds__(256) void k_pack_weight_tiled(const float* __restrict__ w, float* __restrict__ out, int Cout,
                                                           int Cin, int taps, int transposed, int swap, int to_torch,
                                                           int out_half = 0) {
  __shared__ float tile[64][65];
  const int A = swap ? Cin : Cout, Bn = swap ? Cout : Cin;  // packed [t][A][Bn]
  const int a = blockIdx.y, bn0 = blockIdx.x * 64;
  const int nb = min(64, Bn - bn0), n = nb * taps;
  auto torch_index = [&](int bn, int t) -> long {
    const int co = swap ? bn : a, ci = swap ? a : bn;
    return transposed ? ((long)ci * Cout + co) * taps + t : ((long)co * Cin + ci) * taps + t;
  };
  if (!to_torch) {
    for (int e = threadIdx.x; e < n; e += 256) {
      const int bl = e / taps, t = e - bl * taps;
      tile[bl][t] = w[torch_index(bn0 + bl, t)];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < taps * 64; e += 256) {
      const int t = e >> 6, bl = e & 63;
      if (bl < nb) hp_st1(out, ((long)t * A + a) * Bn + bn0 + bl, tile[bl][t], out_half);
    }
  } else {
    for (int e = threadIdx.x; e < taps * 64; e += 256) {
      const int t = e >> 6, bl = e & 63;
      if (bl < nb) tile[bl][t] = out[((long)t * A + a) * Bn + bn0 + bl];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n; e += 256) {
      const int bl = e / taps, t = e - bl * taps;
      ((float*)w)[torch_index(bn0 + bl, t)] = tile[bl][t];
    }
  }
}

// stem: torch (64,1,7,7,7) <-> [64][kpad] (taps along K, zero padded)
__global__ void k_pack_stem(const float* __restrict__ w, float* __restrict__ out, int Cout, int kpad, int to_torch) {
  const long total = (long)Cout * kpad;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % kpad), n = (int)(i / kpad);
    if (to_torch) {
      if (k < 343) ((float*)w)[(long)n * 343 + k] = out[i];
    } else {
      out[i] = k < 343 ? w[(long)n * 343 + k] : 0.f;
    }
  }
}

// ---------------------------------------------------------------- host dispatch
struct ConvPlan {
  IgemmGeom fwd, dgrad, wgrad;
  int fwd_classes = 1, dgrad_classes = 1, wgrad_tapsum = 1;
  bool stem = false, dgrad_zero_fill = false;
  int planes = 0;
};

static void set_shifts(IgemmGeom& g) {
  g.sw = g.sh = g.sd = -1;
  if (g.mode >= 0 && is_pow2(g.gw) && is_pow2(g.gh) && is_pow2(g.gd) && g.M < (1l << 32)) {
    g.sw = ilog2(g.gw);
    g.sh = ilog2(g.gh);
    g.sd = ilog2(g.gd);
  }
}

static int make_plan(const hp_conv_desc& d, ConvPlan& p) {
  HP_REQUIRE(d.B >= 1 && d.Cin >= 1 && d.Cout >= 1, "conv: bad channel/batch sizes");
  HP_REQUIRE(d.precision >= HP_PRECISION_FP32 && d.precision <= HP_PRECISION_BF16X6, "conv: unknown precision %d", d.precision);
  HP_REQUIRE(d.io == 0 || d.precision == HP_PRECISION_BF16, "conv: bf16 activation tensors (io = %d) go with HP_PRECISION_BF16", d.io);
  p.planes = d.precision;  // HP_PRECISION_* = number of bf16 operand planes (0: exact-fp32 MFMA)
  const int k = d.k, s = d.stride, pad = d.pad;
  IgemmGeom f{};
  if (!d.transposed) {
    const int Do = (d.Di + 2 * pad - k) / s + 1, Ho = (d.Hi + 2 * pad - k) / s + 1, Wo = (d.Wi + 2 * pad - k) / s + 1;
    p.stem = (d.Cin == 1 && k == 7 && s == 1 && pad == 3);
    if (!p.stem) HP_REQUIRE(d.Cin % 4 == 0, "conv: Cin must be a multiple of 4 (got %d)", d.Cin);
    HP_REQUIRE(s == 1 || s == 2, "conv: stride must be 1 or 2");
    const int kpt = p.stem ? (343 + 31) / 32 : (d.Cin + 31) / 32;
    f = IgemmGeom{p.stem ? MODE_STEM : MODE_CONV, d.B, d.Di, d.Hi, d.Wi, d.Cin, Do, Ho, Wo, d.Cout, Do, Ho, Wo, 1,
                  k, s, pad, 0, (long)d.B * Do * Ho * Wo, kpt};
    p.fwd = f;
    p.fwd_classes = 1;
    // weight gradient: same gather as forward
    p.wgrad = f;
    p.wgrad_tapsum = p.stem ? 1 : k * k * k;
    // data gradient
    IgemmGeom g{};
    if (p.stem) {
      // gradient w.r.t. the single-channel volume: N = 1 (correct, but 1/32 of the MFMA tile is used)
      g = IgemmGeom{MODE_CONV, d.B, Do, Ho, Wo, d.Cout, d.Di, d.Hi, d.Wi, 1, d.Di, d.Hi, d.Wi, 1,
                    7, 1, 3, 1, (long)d.B * d.Di * d.Hi * d.Wi, d.Cout / 32};
      p.dgrad_classes = 1;
    } else if (s == 1) {
      HP_REQUIRE(d.Cout % 4 == 0, "conv dgrad: Cout must be a multiple of 4 (got %d)", d.Cout);
      g = IgemmGeom{MODE_CONV, d.B, Do, Ho, Wo, d.Cout, d.Di, d.Hi, d.Wi, d.Cin, d.Di, d.Hi, d.Wi, 1,
                    k, 1, pad, 1, (long)d.B * d.Di * d.Hi * d.Wi, (d.Cout + 31) / 32};
      p.dgrad_classes = 1;
    } else if (k == 3 && pad == 1) {
      HP_REQUIRE(d.Di % 2 == 0 && d.Hi % 2 == 0 && d.Wi % 2 == 0 && d.Cout % 32 == 0, "conv dgrad s2: even dims needed");
      g = IgemmGeom{MODE_DGRAD_S2K3, d.B, Do, Ho, Wo, d.Cout, d.Di, d.Hi, d.Wi, d.Cin, d.Di / 2, d.Hi / 2, d.Wi / 2, 2,
                    3, 1, 1, 0, (long)d.B * (d.Di / 2) * (d.Hi / 2) * (d.Wi / 2), d.Cout / 32};
      p.dgrad_classes = 8;
    } else if (k == 1 && pad == 0) {
      HP_REQUIRE(d.Di % 2 == 0 && d.Hi % 2 == 0 && d.Wi % 2 == 0 && d.Cout % 32 == 0, "conv dgrad s2: even dims needed");
      g = IgemmGeom{MODE_CONV, d.B, Do, Ho, Wo, d.Cout, d.Di, d.Hi, d.Wi, d.Cin, Do, Ho, Wo, 2,
                    1, 1, 0, 0, (long)d.B * Do * Ho * Wo, d.Cout / 32};
      p.dgrad_classes = 1;
      p.dgrad_zero_fill = true;
    } else {
      set_error("conv: unsupported stride-2 kernel %d pad %d", k, pad);
      return HP_ERR_UNSUPPORTED;
    }
    p.dgrad = g;
  } else {
    HP_REQUIRE(k == 4 && s == 2 && pad == 1, "deconv: only k4 s2 p1 is supported");
    HP_REQUIRE(d.Cin % 32 == 0 && d.Cout % 32 == 0, "deconv: channels must be multiples of 32");
    const int Do = 2 * d.Di, Ho = 2 * d.Hi, Wo = 2 * d.Wi;
    f = IgemmGeom{MODE_DECONV, d.B, d.Di, d.Hi, d.Wi, d.Cin, Do, Ho, Wo, d.Cout, d.Di, d.Hi, d.Wi, 2,
                  4, 1, 1, 0, (long)d.B * d.Di * d.Hi * d.Wi, d.Cin / 32};
    p.fwd = f;
    p.fwd_classes = 8;
    p.wgrad = f;
    p.wgrad_tapsum = 64;
    // dX[i] = sum_k dY[2i - 1 + k] w[k] : a stride-2 k4 p1 gather of dY
    p.dgrad = IgemmGeom{MODE_CONV, d.B, Do, Ho, Wo, d.Cout, d.Di, d.Hi, d.Wi, d.Cin, d.Di, d.Hi, d.Wi, 1,
                        4, 2, 1, 0, (long)d.B * d.Di * d.Hi * d.Wi, d.Cout / 32};
    p.dgrad_classes = 1;
  }
  HP_REQUIRE((long)d.B * d.Di * d.Hi * d.Wi * (d.transposed ? 8 : 1) < (1l << 31), "conv: more than 2^31 voxels per tensor");
  set_shifts(p.fwd);
  set_shifts(p.dgrad);
  set_shifts(p.wgrad);
  p.fwd.welems = p.wgrad.welems = p.stem ? (long)d.Cout * ((343 + 31) / 32 * 32) : (long)d.Cout * d.Cin * k * k * k;
  p.dgrad.welems = (long)d.Cout * d.Cin * k * k * k;
  return HP_OK;
}

template <bool STEM, bool STATS, int NP, bool XH = false>
static void launch_igemm_bn(const IgemmGeom& g, int classes, const void* X, const float* W, const float* bias, void* Y,
                            double* stats, const void* addend, const unsigned char* amask, hipStream_t st) {
  if constexpr (NP == 1 && !STEM && !XH) {
    // bf16 tensor in memory, channels a multiple of 64: the 64-deep K tile variant
    if (g.xh && g.Cin % 64 == 0) return launch_igemm_bn<STEM, STATS, NP, true>(g, classes, X, W, bias, Y, stats, addend, amask, st);
  }
  const unsigned mt = (unsigned)((g.M + BM - 1) / BM);
  // bf16 packed weights as well: tiles loaded straight into LDS (HP_IGEMM_GL=0 keeps the register-staged tiles: A/B runs)
  static const bool gl_on = !(getenv("HP_IGEMM_GL") && atoi(getenv("HP_IGEMM_GL")) == 0);
  const bool gl = XH && g.wh && gl_on;
  // exact-fp32 tiles with whole 32-channel K tiles: buffer loads (HP_IGEMM_BL=0 keeps the flat loads: A/B runs)
  static const bool bl_on = !(getenv("HP_IGEMM_BL") && atoi(getenv("HP_IGEMM_BL")) == 0);
  // (also the bf16 / split-bf16 modes on fp32 tensors: same loads, operands split on their way into LDS -- their flat
  // loads sat behind per-row branches and were waited for one by one)
  const bool bl = !XH && !g.xh && !g.wh && !STEM && bl_on && g.Cin % 32 == 0 && (long)g.Nout * g.Cin * class_ntaps(g, 0) * 4 * (g.mode == MODE_DECONV ? 8 : 1) < (1l << 31);
  // each XCD walks a contiguous eighth of the M tiles: the halo rows / planes that neighbouring M tiles share are then served
  // by ONE L2 instead of being fetched by several (same step time, L2 fills of the family 422 -> 383 GB per headline step on
  // one device).  HP_IGEMM_SLAB=0: M tile m on XCD m % 8 as before (A/B runs)
  static const bool slab_on = !(getenv("HP_IGEMM_SLAB") && atoi(getenv("HP_IGEMM_SLAB")) == 0);
  if (slab_on && g.Nout <= 64) {
    IgemmGeom gs = g;
    gs.slab = (int)((mt + 7) / 8);
    const dim3 grid((mt + 7) / 8 * 8, 1, classes);
    if constexpr (!XH && !STEM) {
      if (bl && g.Nout > 32) {
        hipLaunchKernelGGL((k_igemm<64, false, STATS, NP, false, false, true>), grid, dim3(CT), 0, st, X, W, bias, Y, stats, addend, amask, gs);
        return;
      }
    }
  }
  if (g.Nout > 64) {
    const unsigned tn = (unsigned)((g.Nout + 127) / 128);
    IgemmGeom gg = g;
    gg.tn = tn > 1 ? (int)tn : 0;  // XCD-aware 1-D grid (see k_igemm)
    gg.slab = slab_on ? (int)((mt + 7) / 8) : 0;
    const dim3 grid = (tn > 1 || gg.slab) ? dim3((mt + 7) / 8 * 8 * (tn > 1 ? tn : 1), 1, classes) : dim3(mt, 1, classes);
    if constexpr (XH) {
      if (gl) {
        hipLaunchKernelGGL((k_igemm<128, STEM, STATS, NP, true, true>), grid, dim3(CT), 0, st, X, W, bias, Y, stats, addend, amask, gg);
        return;
      }
    }
    if constexpr (!XH && !STEM) {
      if (bl) {
        hipLaunchKernelGGL((k_igemm<128, false, STATS, NP, false, false, true>), grid, dim3(CT), 0, st, X, W, bias, Y, stats, addend, amask, gg);
        return;
      }
    }
    hipLaunchKernelGGL((k_igemm<128, STEM, STATS, NP, XH>), grid, dim3(CT), 0, st, X, W, bias, Y, stats, addend, amask, gg);
  } else if (g.Nout > 32) {
    if constexpr (XH) {
      if (gl) {
        hipLaunchKernelGGL((k_igemm<64, STEM, STATS, NP, true, true>), dim3(mt, 1, classes), dim3(CT), 0, st, X, W, bias, Y, stats, addend, amask, g);
        return;
      }
    }
    if constexpr (!XH && !STEM) {
      if (bl) {
        hipLaunchKernelGGL((k_igemm<64, false, STATS, NP, false, false, true>), dim3(mt, 1, classes), dim3(CT), 0, st, X, W, bias, Y, stats, addend, amask, g);
        return;
      }
    }
    hipLaunchKernelGGL((k_igemm<64, STEM, STATS, NP, XH>), dim3(mt, 1, classes), dim3(CT), 0, st, X, W, bias, Y, stats, addend, amask, g);
  } else {
    hipLaunchKernelGGL((k_igemm<32, STEM, STATS, NP, XH>), dim3(mt, 1, classes), dim3(CT), 0, st, X, W, bias, Y, stats, addend, amask, g);
  }
}

template <int NP>
static void launch_igemm_p(const IgemmGeom& g, int classes, bool stem, const void* X, const float* W, const float* bias,
                           void* Y, double* stats, const void* addend, const unsigned char* amask, hipStream_t st) {
  if (stem) {
    if (stats) launch_igemm_bn<true, true, NP>(g, classes, X, W, bias, Y, stats, addend, amask, st);
    else launch_igemm_bn<true, false, NP>(g, classes, X, W, bias, Y, stats, addend, amask, st);
  } else {
    if (stats) launch_igemm_bn<false, true, NP>(g, classes, X, W, bias, Y, stats, addend, amask, st);
    else launch_igemm_bn<false, false, NP>(g, classes, X, W, bias, Y, stats, addend, amask, st);
  }
}

static void launch_igemm(const IgemmGeom& g, int classes, bool stem, int planes, const void* X, const float* W,
                         const float* bias, void* Y, double* stats, const void* addend, hipStream_t st,
                         const unsigned char* amask = nullptr) {
  switch (planes) {
    case 1: launch_igemm_p<1>(g, classes, stem, X, W, bias, Y, stats, addend, amask, st); break;
    case 2: launch_igemm_p<2>(g, classes, stem, X, W, bias, Y, stats, addend, amask, st); break;
    case 3: launch_igemm_p<3>(g, classes, stem, X, W, bias, Y, stats, addend, amask, st); break;
    default: launch_igemm_p<0>(g, classes, stem, X, W, bias, Y, stats, addend, amask, st); break;
  }
}

}  // namespace hp

using namespace hp;

extern "C" size_t hp_conv3d_packed_weight_elems(const hp_conv_desc* d) {
  if (!d) return 0;
  if (!d->transposed && d->Cin == 1 && d->k == 7) return (size_t)d->Cout * ((343 + 31) / 32 * 32);
  return (size_t)d->Cout * d->Cin * d->k * d->k * d->k;
}

extern "C" int hp_conv3d_pack_weight(const hp_conv_desc* d, const float* w_torch, void* w_fwd_v, void* w_dgrad_v,
                                     void* stream) {
  float* const w_fwd = (float*)w_fwd_v;      // bf16 images (HP_IO_W_BF16) are written through hp_st1
  float* const w_dgrad = (float*)w_dgrad_v;
  HP_REQUIRE(d && w_torch, "hp_conv3d_pack_weight: null argument");
  ConvPlan p;
  int rc = make_plan(*d, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int taps = d->k * d->k * d->k;
  if (p.stem) {
    const int kpad = p.fwd.kpt * BK;
    HP_PROF("conv_pack_weight", st);
    if (w_fwd) hipLaunchKernelGGL(k_pack_stem, dim3(64), dim3(256), 0, st, w_torch, w_fwd, d->Cout, kpad, 0);
    if (w_dgrad) hipLaunchKernelGGL(k_pack_weight, dim3(64), dim3(256), 0, st, w_torch, w_dgrad, d->Cout, 1, 343, 0, 1, 0);
  } else {
    HP_REQUIRE(taps <= 64, "conv: at most 64 taps (got %d)", taps);
    HP_PROF("conv_pack_weight", st);
    const int wh = (d->io & HP_IO_W_BF16) ? 1 : 0;  // packed images as bf16 (the bf16-storage tiles read them as they lie)
    if (w_fwd)
      hipLaunchKernelGGL(k_pack_weight_tiled, dim3((d->Cin + 63) / 64, d->Cout), dim3(256), 0, st, w_torch, w_fwd, d->Cout,
                         d->Cin, taps, d->transposed, 0, 0, wh);
    if (w_dgrad)
      hipLaunchKernelGGL(k_pack_weight_tiled, dim3((d->Cout + 63) / 64, d->Cin), dim3(256), 0, st, w_torch, w_dgrad, d->Cout,
                         d->Cin, taps, d->transposed, 1, 0, wh);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_conv3d_unpack_wgrad(const hp_conv_desc* d, const float* dw_packed, float* dw_torch, void* stream) {
  HP_REQUIRE(d && dw_packed && dw_torch, "hp_conv3d_unpack_wgrad: null argument");
  ConvPlan p;
  int rc = make_plan(*d, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int taps = d->k * d->k * d->k;
  HP_PROF("conv_pack_weight", st);
  if (p.stem) {
    hipLaunchKernelGGL(k_pack_stem, dim3(64), dim3(256), 0, st, dw_torch, (float*)dw_packed, d->Cout, p.fwd.kpt * BK, 1);
  } else {
    hipLaunchKernelGGL(k_pack_weight_tiled, dim3((d->Cin + 63) / 64, d->Cout), dim3(256), 0, st, dw_torch, (float*)dw_packed,
                       d->Cout, d->Cin, taps, d->transposed, 0, 1);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_conv3d_forward(const hp_conv_desc* d, const void* x, const float* w_fwd, const float* bias, void* y,
                                 double* stats, void* stream) {
  HP_REQUIRE(d && x && w_fwd && y, "hp_conv3d_forward: null argument");
  ConvPlan p;
  int rc = make_plan(*d, p);
  if (rc) return rc;
  HP_REQUIRE(!(p.stem && (d->io & HP_IO_X_BF16)), "hp_conv3d_forward: the stem reads its single-channel input as fp32");
  p.fwd.xh = (d->io & HP_IO_X_BF16) ? 1 : 0;
  p.fwd.yh = (d->io & HP_IO_Y_BF16) ? 1 : 0;
  p.fwd.wh = (d->io & HP_IO_W_BF16) ? 1 : 0;
  HP_REQUIRE(!p.fwd.wh || (p.fwd.xh && p.planes == 1 && !p.stem && d->Cin % 64 == 0),
             "hp_conv3d_forward: bf16 packed weights go with a bf16 input, HP_PRECISION_BF16 and Cin %% 64 == 0");
  hipStream_t st = (hipStream_t)stream;
  if (stats) HP_CHECK_HIP(hipMemsetAsync(stats, 0, sizeof(double) * 2 * d->Cout * HP_STATS_SLOTS, st));
  if (p.stem && p.planes == 1 && d->Cout == 64 && !bias && !p.fwd.yh) {
    // single-plane bf16 arithmetic: persistent workgroups (one per CU: 100 KB of LDS), 16 x 4 x 8 voxel tiles
    const int pz = (d->Di + SFH_TZ - 1) / SFH_TZ, py = (d->Hi + SFH_TY - 1) / SFH_TY, px = (d->Wi + SFH_TX - 1) / SFH_TX;
    const long tiles = (long)d->B * pz * py * px;
    const int per_wg = (int)std::max<long>(1, (tiles + 511) / 512);
    const size_t lds = sizeof(__bf16) * (SFH_W + 4 * SFH_COPY) + sizeof(float) * 8 * 64 * 2;
    static std::once_flag lds_once_h;
    static hipError_t lds_rc_h = hipSuccess;
    std::call_once(lds_once_h, [&] {
      lds_rc_h = hipFuncSetAttribute((const void*)k_stem_fwd_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    HP_CHECK_HIP(lds_rc_h);
    HP_PROF("conv_igemm_stem", st);
    hipLaunchKernelGGL(k_stem_fwd_bf16, dim3((unsigned)((tiles + per_wg - 1) / per_wg)), dim3(512), lds, st, (const float*)x, w_fwd,
                       (float*)y, stats, d->Di, d->Hi, d->Wi, p.fwd.kpt * BK, pz, py, px, tiles, per_wg);
  } else if (p.stem && p.planes == 0 && d->Cout == 64 && !bias && !p.fwd.yh) {
    // dedicated 4x4x1-MFMA stem kernel: ~1024 workgroups, one resident per CU (138 KB of LDS)
    const int tiles_x = (d->Wi + SF_TX - 1) / SF_TX, tiles_y = (d->Hi + SF_TY - 1) / SF_TY;
    const long cols = (long)tiles_x * tiles_y * d->B;
    int zsplit = (int)std::max<long>(1, std::min<long>((d->Di + 15) / 16, (1024 + cols - 1) / cols));
    const int zchunk = (d->Di + zsplit - 1) / zsplit;
    zsplit = (d->Di + zchunk - 1) / zchunk;
    const size_t lds = sizeof(float) * (343 * 64 + 8 * SF_PLANE + 4 * 64 * 33);
    // opt in to > 64 KB of dynamic LDS once per process (idempotent; std::call_once keeps concurrent callers safe)
    static std::once_flag lds_once;
    static hipError_t lds_rc = hipSuccess;
    std::call_once(lds_once, [&] {
      lds_rc = hipFuncSetAttribute((const void*)k_stem_fwd_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    HP_CHECK_HIP(lds_rc);
    HP_PROF("conv_igemm_stem", st);
    hipLaunchKernelGGL(k_stem_fwd_mfma, dim3((unsigned)(tiles_x * tiles_y * zsplit), (unsigned)d->B), dim3(256), lds, st,
                       (const float*)x, w_fwd, (float*)y, stats, d->Di, d->Hi, d->Wi, p.fwd.kpt * BK, tiles_x, tiles_y, zchunk);
  } else {
    HP_PROF(p.stem ? "conv_igemm_stem" : d->transposed ? "conv_igemm_deconv" : d->k == 1 ? "conv_igemm_k1" : "conv_igemm_k3", st);
    launch_igemm(p.fwd, p.fwd_classes, p.stem, p.planes, x, w_fwd, bias, y, stats, nullptr, st);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_linear_forward(const float* x, const float* w, const float* bias, const float* addend, float* y, long M, int K,
                                 int N, int precision, void* stream) {
  HP_REQUIRE(x && w && y && M >= 1 && M < (1l << 31) && K >= 4 && N >= 1, "hp_linear_forward: bad argument");
  hp_conv_desc d{1, 1, 1, (int)M, K, N, 1, 1, 0, 0, precision, 0};
  ConvPlan p;
  int rc = make_plan(d, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  {
    HP_PROF("linear_fwd", st);
    launch_igemm(p.fwd, 1, false, p.planes, x, w, bias, y, nullptr, addend, st);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

extern "C" int hp_linear_geglu_forward(const float* x, const float* w, const float* bias, float* y, long M, int K,
                                       int N2, int precision, void* stream) {
  HP_REQUIRE(x && w && y && M >= 1 && M < (1l << 31) && K >= 4, "hp_linear_geglu_forward: bad argument");
  HP_REQUIRE(N2 >= 128 && N2 % 128 == 0, "hp_linear_geglu_forward: 2 * hidden must be a multiple of 128 (got %d)", N2);
  hp_conv_desc d{1, 1, 1, (int)M, K, N2, 1, 1, 0, 0, precision, 0};
  ConvPlan p;
  int rc = make_plan(d, p);
  if (rc) return rc;
  p.fwd.geglu = 1;
  hipStream_t st = (hipStream_t)stream;
  {
    HP_PROF("linear_fwd", st);
    launch_igemm(p.fwd, 1, false, p.planes, x, w, bias, y, nullptr, nullptr, st);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// Data gradient whose output is the incoming gradient of the BatchNorm unit in front of the convolution: the same exact-fp32
// whole-tile kernels with the BNS epilogue (see k_igemm).  Returns false when the geometry is not one they cover.
static bool launch_igemm_bns(const IgemmGeom& g, const void* X, const float* W, void* Y, const void* addend,
                             const unsigned char* amask, hipStream_t st) {
  const bool dense_out = g.os == 1 && g.gd == g.Do && g.gh == g.Ho && g.gw == g.Wo;
  if (g.mode != MODE_CONV || !dense_out || g.M % BM != 0 || g.Nout <= 32 || g.Nout % (g.Nout > 64 ? 128 : 64) != 0 || g.xh || g.yh ||
      g.wh || g.ah)
    return false;
  const unsigned mt = (unsigned)(g.M / BM);
  static const bool bl_on = !(getenv("HP_IGEMM_BL") && atoi(getenv("HP_IGEMM_BL")) == 0);
  static const bool slab_on = !(getenv("HP_IGEMM_SLAB") && atoi(getenv("HP_IGEMM_SLAB")) == 0);
  const bool bl = bl_on && g.Cin % 32 == 0 && (long)g.Nout * g.Cin * class_ntaps(g, 0) * 4 < (1l << 31);
  IgemmGeom gg = g;
  gg.slab = slab_on ? (int)((mt + 7) / 8) : 0;
#define HP_BNS_LAUNCH(BN_, BL_)                                                                                                     \
  do {                                                                                                                             \
    if (g.bn_mask)                                                                                                                 \
      hipLaunchKernelGGL((k_igemm<BN_, false, false, 0, false, false, BL_, 2>), grid, dim3(CT), 0, st, X, W, nullptr, Y, nullptr,   \
                         addend, amask, gg);                                                                                       \
    else                                                                                                                           \
      hipLaunchKernelGGL((k_igemm<BN_, false, false, 0, false, false, BL_, 1>), grid, dim3(CT), 0, st, X, W, nullptr, Y, nullptr,   \
                         addend, amask, gg);                                                                                       \
  } while (0)
  if (g.Nout <= 64) {   // one N tile of 64 columns (launch_igemm_bn's first two branches)
    const dim3 grid = gg.slab ? dim3((mt + 7) / 8 * 8, 1, 1) : dim3(mt, 1, 1);
    if (bl) HP_BNS_LAUNCH(64, true);
    else HP_BNS_LAUNCH(64, false);
    return true;
  }
  const unsigned tn = (unsigned)(g.Nout / 128);
  gg.tn = tn > 1 ? (int)tn : 0;
  const dim3 grid = (tn > 1 || gg.slab) ? dim3((mt + 7) / 8 * 8 * (tn > 1 ? tn : 1), 1, 1) : dim3(mt, 1, 1);
  if (bl) HP_BNS_LAUNCH(128, true);
  else HP_BNS_LAUNCH(128, false);
#undef HP_BNS_LAUNCH
  return true;
}

extern "C" int hp_conv3d_backward_data_bnsums(const hp_conv_desc* d, const void* dy, const float* w_dgrad, void* dx, const void* addend,
                                              const unsigned char* addend_mask, const float* z, const float* mean,
                                              const float* rstd, const float* gamma, const float* beta, int relu,
                                              const unsigned char* relu_mask, double* sums, int* fused, void* stream) {
  HP_REQUIRE(d && dy && w_dgrad && dx && z && mean && rstd && sums && fused && (!relu || relu_mask || (gamma && beta)),
             "hp_conv3d_backward_data_bnsums: null argument");
  *fused = 0;
  ConvPlan p;
  int rc = make_plan(*d, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (p.planes == 0 && !p.stem && !p.dgrad_zero_fill && d->io == 0 && p.dgrad_classes == 1 &&
      (!addend_mask || (addend && d->Cin % 4 == 0 && d->Cin > 32))) {
    IgemmGeom g = p.dgrad;
    g.bn_z = z;
    g.bn_mean = mean;
    g.bn_rstd = rstd;
    g.bn_gamma = gamma;
    g.bn_beta = beta;
    g.bn_sums = sums;
    g.bn_relu = relu ? 1 : 0;
    g.bn_mask = relu ? relu_mask : nullptr;
    HP_CHECK_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 2 * g.Nout * HP_STATS_SLOTS, st));
    HP_PROF("conv_igemm_dgrad", st);
    if (launch_igemm_bns(g, dy, w_dgrad, dx, addend, addend_mask, st)) {
      HP_CHECK_HIP(hipGetLastError());
      *fused = 1;
      return HP_OK;
    }
  }
  return hp_conv3d_backward_data_masked(d, dy, w_dgrad, dx, addend, addend_mask, stream);
}

extern "C" int hp_conv3d_backward_data(const hp_conv_desc* d, const void* dy, const float* w_dgrad, void* dx,
                                       const void* addend, void* stream) {
  return hp_conv3d_backward_data_masked(d, dy, w_dgrad, dx, addend, nullptr, stream);
}

extern "C" int hp_conv3d_backward_data_masked(const hp_conv_desc* d, const void* dy, const float* w_dgrad, void* dx,
                                              const void* addend, const unsigned char* addend_mask, void* stream) {
  HP_REQUIRE(d && dy && w_dgrad && dx, "hp_conv3d_backward_data: null argument");
  ConvPlan p;
  int rc = make_plan(*d, p);
  if (rc) return rc;
  const int dyh = (d->io & HP_IO_DY_BF16) ? 1 : 0, dxh = (d->io & HP_IO_DX_BF16) ? 1 : 0;
  HP_REQUIRE(!(p.stem && (dyh || dxh)), "hp_conv3d_backward_data: the stem's data gradient is fp32 in, fp32 out");
  p.dgrad.xh = dyh;
  p.dgrad.yh = p.dgrad.ah = dxh;
  p.dgrad.wh = (d->io & HP_IO_W_BF16) ? 1 : 0;
  HP_REQUIRE(!p.dgrad.wh || (dyh && p.planes == 1 && !p.stem && p.dgrad.Cin % 64 == 0),
             "hp_conv3d_backward_data: bf16 packed weights go with a bf16 dy, HP_PRECISION_BF16 and Cout %% 64 == 0");
  const size_t esz = dxh ? 2 : 4;
  HP_REQUIRE(!addend_mask || (addend && !p.stem && !p.dgrad_zero_fill && d->Cin % 4 == 0 && d->Cin > 32),
             "hp_conv3d_backward_data_masked: a masked addend needs a dense data gradient with > 32 input channels, a multiple of 4");
  hipStream_t st = (hipStream_t)stream;
  if (p.stem) {
    HP_REQUIRE(d->Cout == 64, "stem data gradient: 64 output channels expected (got %d)", d->Cout);
    HP_REQUIRE((long)d->Hi * d->Wi <= (1l << 21), "stem data gradient: H * W up to 2^21 (31-bit offsets inside a 4-plane dZ tile), got %d x %d", d->Hi, d->Wi);
    const size_t nb = sizeof(float) * (size_t)d->B * d->Di * d->Hi * d->Wi;
    if (addend) HP_CHECK_HIP(hipMemcpyAsync(dx, addend, nb, hipMemcpyDeviceToDevice, st));
    else HP_CHECK_HIP(hipMemsetAsync(dx, 0, nb, st));
    const int pz = (d->Di + SP_Z - 1) / SP_Z, py = (d->Hi + SP_Y - 1) / SP_Y, px = (d->Wi + SP_X - 1) / SP_X;
    HP_PROF("conv_stem_dgrad", st);
    // single-plane bf16 arithmetic: the patch GEMM on the bf16 matrix cores (fp32 in, fp32 out either way)
    // 3 workgroups resident per CU (768 slots), each walking zchunk patches along z: the z split that minimises
    // (rounds of 768 workgroups) x (patches per workgroup + ~2 for the extra ring start / full flush of a split)
    const long cols = (long)py * px * d->B;
    int zsplit = 1;
    long best = -1;
    for (int zs = 1; zs <= std::min(pz, 32); ++zs) {
      const long cost = ((cols * zs + 767) / 768) * ((pz + zs - 1) / zs + 2);
      if (best < 0 || cost < best) best = cost, zsplit = zs;
    }
    if (const char* e = getenv("HP_STEM_DGRAD_ZSPLIT")) zsplit = std::max(1, std::min(pz, atoi(e)));  // tests: long runs on small volumes
    const int zchunk = (pz + zsplit - 1) / zsplit;
    zsplit = (pz + zchunk - 1) / zchunk;
    hipLaunchKernelGGL(p.planes == 1 ? k_stem_dgrad<true> : k_stem_dgrad<false>, dim3((unsigned)(py * px * zsplit), (unsigned)d->B), dim3(CT), 0,
                       st, (const float*)dy, w_dgrad, (float*)dx, d->Di, d->Hi, d->Wi, pz, py, px, zchunk);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
  }
  if (p.dgrad_zero_fill) {  // strided 1^3 convolution: only every second voxel per axis receives a contribution
    const size_t nb = esz * (size_t)d->B * d->Di * d->Hi * d->Wi * d->Cin;
    if (addend == dx) {}  // in place: the untouched voxels already hold the other contribution
    else if (addend) HP_CHECK_HIP(hipMemcpyAsync(dx, addend, nb, hipMemcpyDeviceToDevice, st));
    else HP_CHECK_HIP(hipMemsetAsync(dx, 0, nb, st));
  }
  {
    HP_PROF("conv_igemm_dgrad", st);
    launch_igemm(p.dgrad, p.dgrad_classes, false, p.planes, dy, w_dgrad, nullptr, dx, nullptr, addend, st, addend_mask);
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}

// How the generic weight-gradient kernels split one call: TT x TT tiles of dW[Cout][K], tap groups, and `msplit` chunks of
// the M = B * voxels reduction whose partial sums meet in fp32 atomics.
struct WgradSplit {
  int TT, tap_groups, tiles_c, tiles_total;
  bool multitap;
  long msplit;
};
static WgradSplit wgrad_split(const ConvPlan& p) {
  const IgemmGeom& g = p.wgrad;
  const int Kc = p.stem ? g.kpt * BK : g.Cin;
  WgradSplit w{};
  w.TT = (g.Nout >= 128 && Kc >= 128) ? 128 : 64;
  // 64-wide tiles of a plain 3^3 convolution: 4 taps per block share the staged dY tile
  w.multitap = !p.stem && w.TT == 64 && g.mode == MODE_CONV && g.k == 3;
  w.tap_groups = w.multitap ? (27 + 3) / 4 : p.wgrad_tapsum;
  const int tiles_n = (g.Nout + w.TT - 1) / w.TT;
  w.tiles_c = (Kc + w.TT - 1) / w.TT;
  const long base_blocks = (long)tiles_n * w.tiles_c * w.tap_groups;
  // Upper end: ~4096 blocks (the large layers: plenty of rows per block, and many short blocks balance best), at least 4
  // steps per block.  Below it, the count that minimises  waves x (rows per block + R0):  every block pays a fixed price --
  // prologue, and TT x TT fp32 atomics in the epilogue -- worth R0 rows of its main loop (fitted on layers 3 / 4 of the
  // bench shape: ~96 rows with fp32 operands, ~768 with bf16 ones, whose rows are five times cheaper; 768 resident blocks).
  // With M = 8192 ... 65536 rows the old rule alone cut 4096 blocks of 128 ... 256 rows: layer 4's 1^3 weight gradients
  // ran 0.235 ms (fp32) / 0.211 ms (bf16s) against 0.157 / 0.055 with a sixth of the blocks (round 4).
  long hi = std::max<long>(1, (4096 + base_blocks - 1) / base_blocks);
  hi = std::min<long>(std::min<long>(hi, std::max<long>(1, g.M / (4 * WG_KM))), 4096);
  static const int fixed_env = getenv("HP_WGRAD_R0") ? atoi(getenv("HP_WGRAD_R0")) : -1;   // 0: the old rule
  const long r0 = fixed_env >= 0 ? fixed_env : ((g.xh && g.yh) ? 768 : 96);
  long best = hi;
  // (the search below is a few thousand cost evaluations: remembered per (M, blocks, R0) -- a training loop asks for the same
  // ~60 geometries every step)
  struct Memo { long M, base, r0, hi, best; };
  static std::mutex memo_mu;
  static std::vector<Memo> memo;
  bool known = false;
  {
    std::lock_guard<std::mutex> lk(memo_mu);
    for (const Memo& m : memo)
      if (m.M == g.M && m.base == base_blocks && m.r0 == r0 && m.hi == hi) {
        best = m.best;
        known = true;
        break;
      }
  }
  if (r0 > 0 && !known) {
    const long slots = 768;
    auto cost = [&](long ms) {
      const long rows = ((g.M + ms - 1) / ms + WG_KM - 1) / WG_KM * WG_KM;
      const long blocks = base_blocks * ms;
      double c = (double)((blocks + slots - 1) / slots) * (double)(rows + r0);
      if (blocks & 7) c *= 1.02;   // the XCD renumbering of the kernels needs a grid that is a multiple of 8
      return c;
    };
    double cmin = cost(hi);
    for (long ms = 1; ms < hi; ++ms) cmin = std::min(cmin, cost(ms));
    for (long ms = hi; ms >= 1; --ms)   // the largest count within 3 % of the minimum: more, shorter blocks balance better
      if (cost(ms) <= 1.03 * cmin) {
        best = ms;
        break;
      }
  }
  // the kernels round a chunk up to whole steps of WG_KM rows: drop the chunks that this rounding leaves empty
  for (; !known;) {
    const long chunk = ((g.M + best - 1) / best + WG_KM - 1) / WG_KM * WG_KM;
    const long need = (g.M + chunk - 1) / chunk;
    if (need >= best) break;
    best = need;
  }
  if (!known) {
    std::lock_guard<std::mutex> lk(memo_mu);
    if (memo.size() < 4096) memo.push_back({g.M, base_blocks, r0, hi, best});
  }
  w.msplit = best;
  w.tiles_total = tiles_n * w.tiles_c;
  return w;
}

extern "C" int hp_conv3d_backward_weight_split(const hp_conv_desc* d, long* msplit, long* chunk_rows) {
  HP_REQUIRE(d && msplit && chunk_rows, "hp_conv3d_backward_weight_split: null argument");
  ConvPlan p;
  int rc = make_plan(*d, p);
  if (rc) return rc;
  p.wgrad.xh = (d->io & HP_IO_X_BF16) ? 1 : 0;   // as hp_conv3d_backward_weight sets them: the split depends on the operand type
  p.wgrad.yh = (d->io & HP_IO_DY_BF16) ? 1 : 0;
  const WgradSplit ws = wgrad_split(p);
  *msplit = ws.msplit;
  *chunk_rows = ((p.wgrad.M + ws.msplit - 1) / ws.msplit + WG_KM - 1) / WG_KM * WG_KM;  // as the kernels round it
  return HP_OK;
}

extern "C" int hp_conv3d_backward_weight(const hp_conv_desc* d, const void* x, const void* dy, float* dw_packed,
                                         void* stream) {
  HP_REQUIRE(d && x && dy && dw_packed, "hp_conv3d_backward_weight: null argument");
  ConvPlan p;
  int rc = make_plan(*d, p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int xh = (d->io & HP_IO_X_BF16) ? 1 : 0, dyh = (d->io & HP_IO_DY_BF16) ? 1 : 0;
  HP_REQUIRE(!(p.stem && xh), "hp_conv3d_backward_weight: the stem reads its single-channel input as fp32");
  p.wgrad.xh = xh;
  p.wgrad.yh = dyh;  // element type of dY
  const IgemmGeom& g = p.wgrad;
  const int Kc = p.stem ? g.kpt * BK : g.Cin;
  HP_CHECK_HIP(hipMemsetAsync(dw_packed, 0, sizeof(float) * hp_conv3d_packed_weight_elems(d), st));
  if (p.stem && p.planes <= 1 && d->Cout == 64 && !dyh) {
    const int pz = (d->Di + SP_Z - 1) / SP_Z, py = (d->Hi + SP_Y - 1) / SP_Y, px = (d->Wi + SP_X - 1) / SP_X;
    const long tiles = (long)d->B * pz * py * px;
    const int per_wg = (int)std::max<long>(1, (tiles + 1023) / 1024);  // ~1024 workgroups, 2 resident per CU
    HP_PROF("conv_wgrad", st);
    hipLaunchKernelGGL(p.planes == 1 ? k_stem_wgrad_bf16 : k_stem_wgrad_full, dim3((unsigned)((tiles + per_wg - 1) / per_wg)), dim3(256),
                       0, st, (const float*)x, (const float*)dy, dw_packed, d->Di,
                       d->Hi, d->Wi, pz, py, px, tiles, per_wg, Kc);
    HP_CHECK_HIP(hipGetLastError());
    return HP_OK;
  }
  // dense 1^3 convolution with a 64-channel side and a 256-multiple side: one block per voxel split holds TN x TC
  if (!p.stem && p.planes == 0 && !xh && !dyh && g.mode == MODE_CONV && g.k == 1 && g.s == 1 && g.os == 1 && g.M >= (1l << 16)) {
    const bool wide_n = g.Nout % 256 == 0 && g.Cin == 64, wide_c = g.Nout == 64 && g.Cin % 256 == 0;
    if (wide_n || wide_c) {
      const int tiles = wide_n ? g.Nout / 256 : g.Cin / 256;
      // ~1024 blocks = one resident wave of 4 per CU (768 / 2048 / 4096 measured 16 % / 2 % / 5 % slower)
      long ms = std::max<long>(1, std::min<long>(1024 / tiles, g.M / (4 * WG_KM)));
      ms = std::max<long>(8, ms / 8 * 8);  // a multiple of 8 keeps the XCD renumbering on
      HP_PROF("conv_wgrad", st);
      if (wide_n)
        hipLaunchKernelGGL((k_wgrad_dense<256, 64>), dim3((unsigned)(tiles * ms)), dim3(CT), 0, st, (const float*)x, (const float*)dy, dw_packed, g.M, g.Nout,
                           g.Cin, (int)ms, 1);
      else
        hipLaunchKernelGGL((k_wgrad_dense<64, 256>), dim3((unsigned)(tiles * ms)), dim3(CT), 0, st, (const float*)x, (const float*)dy, dw_packed, g.M, g.Nout,
                           g.Cin, (int)ms, tiles);
      HP_CHECK_HIP(hipGetLastError());
      return HP_OK;
    }
  }
  const WgradSplit ws = wgrad_split(p);
  const int TT = ws.TT, tap_groups = ws.tap_groups, tiles_c = ws.tiles_c, tiles_total = ws.tiles_total;
  const bool multitap = ws.multitap;
  const long msplit = ws.msplit;
  dim3 grid((unsigned)((long)tiles_total * tap_groups * msplit));
  {
    HP_PROF("conv_wgrad", st);
    static const bool wbl_on = !(getenv("HP_WGRAD_BL") && atoi(getenv("HP_WGRAD_BL")) == 0);
    // lean fp32 kernel: power-of-two class grid, whole channel quads, and every row offset of a block within 2^31 bytes
    const long chunk_rows = (g.M + msplit - 1) / msplit + WG_KM;
    const long span_y = chunk_rows * g.os * g.os * g.os * (long)g.Nout * 4, span_x = (chunk_rows * g.s * g.s * g.s + 4l * g.Hi * g.Wi) * (long)g.Cin * 4;
    const bool exact_grid = g.Di == g.gd * g.s && g.Hi == g.gh * g.s && g.Wi == g.gw * g.s && g.Do == g.gd * g.os &&
                            g.Ho == g.gh * g.os && g.Wo == g.gw * g.os;
    if (wbl_on && !p.stem && p.planes == 0 && !xh && !dyh && g.sw >= 0 && exact_grid && g.Nout % 4 == 0 && g.Cin % 4 == 0 &&
        span_y < (1l << 30) && span_x < (1l << 30)) {
#define HP_WBL(TT_, NTAP_)                                                                                                       \
  do {                                                                                                                           \
    if (uni)                                                                                                                     \
      hipLaunchKernelGGL((k_wgrad_bl<TT_, NTAP_, true>), grid, dim3(CT), 0, st, (const float*)x, (const float*)dy, dw_packed, g, \
                         tiles_c, (int)msplit, tiles_total, tap_groups);                                                         \
    else                                                                                                                         \
      hipLaunchKernelGGL((k_wgrad_bl<TT_, NTAP_, false>), grid, dim3(CT), 0, st, (const float*)x, (const float*)dy, dw_packed, g, \
                         tiles_c, (int)msplit, tiles_total, tap_groups);                                                         \
  } while (0)
      const bool uni = g.gw >= 8 && !(getenv("HP_WGRAD_UNI") && atoi(getenv("HP_WGRAD_UNI")) == 0);
      if (TT == 128) HP_WBL(128, 1);
      else if (multitap) HP_WBL(64, 4);
      else HP_WBL(64, 1);
#undef HP_WBL
      HP_CHECK_HIP(hipGetLastError());
      return HP_OK;
    }
    static const int wblh_km = getenv("HP_WGRAD_BLH") ? atoi(getenv("HP_WGRAD_BLH")) : 32;   // 0: off; 32 / 64: voxels per step
    if (wblh_km > 0 && !p.stem && p.planes == 1 && xh && dyh && g.sw >= 0 && g.gw >= 8 && exact_grid && g.Cin % 8 == 0 &&
        g.Nout % 8 == 0 && span_y < (1l << 30) && span_x < (1l << 30)) {
      const unsigned short* xs_ = (const unsigned short*)x;
      const unsigned short* ys_ = (const unsigned short*)dy;
#define HP_WBLH(TT_, NTAP_)                                                                                                          \
  do {                                                                                                                                \
    if (wblh_km == 64)                                                                                                                \
      hipLaunchKernelGGL((k_wgrad_blh<TT_, NTAP_, 64>), grid, dim3(CT), 0, st, xs_, ys_, dw_packed, g, tiles_c, (int)msplit,           \
                         tiles_total, tap_groups);                                                                                    \
    else                                                                                                                              \
      hipLaunchKernelGGL((k_wgrad_blh<TT_, NTAP_, 32>), grid, dim3(CT), 0, st, xs_, ys_, dw_packed, g, tiles_c, (int)msplit,           \
                         tiles_total, tap_groups);                                                                                    \
  } while (0)
      if (TT == 128) HP_WBLH(128, 1);
      else if (multitap) HP_WBLH(64, 4);
      else HP_WBLH(64, 1);
#undef HP_WBLH
      HP_CHECK_HIP(hipGetLastError());
      return HP_OK;
    }
#define HP_WGRAD_P(STEM_, TT_, NTAP_, NP_)                                                                   \
  hipLaunchKernelGGL((k_wgrad<STEM_, TT_, NTAP_, NP_>), grid, dim3(CT), 0, st, x, dy, dw_packed, g, tiles_c, \
                     (int)msplit, tiles_total, tap_groups)
#define HP_WGRAD_XH(TT_, NTAP_)                                                                                   \
  hipLaunchKernelGGL((k_wgrad<false, TT_, NTAP_, 1, true>), grid, dim3(CT), 0, st, x, dy, dw_packed, g, tiles_c, \
                     (int)msplit, tiles_total, tap_groups)
#define HP_WGRAD(STEM_, TT_, NTAP_)                                                                                   \
  do {                                                                                                               \
    if (!STEM_ && p.planes == 1 && xh && dyh && g.Cin % 8 == 0 && g.Nout % 8 == 0) {                                 \
      HP_WGRAD_XH(TT_, NTAP_);                                                                                       \
      break;                                                                                                         \
    }                                                                                                                \
    switch (p.planes) {                                                                                              \
      case 1: HP_WGRAD_P(STEM_, TT_, NTAP_, 1); break;                                                               \
      case 2: HP_WGRAD_P(STEM_, TT_, NTAP_, 2); break;                                                               \
      case 3: HP_WGRAD_P(STEM_, TT_, NTAP_, 3); break;                                                               \
      default: HP_WGRAD_P(STEM_, TT_, NTAP_, 0); break;                                                              \
    }                                                                                                                \
  } while (0)
    if (p.stem) HP_WGRAD(true, 64, 1);
    else if (TT == 128) HP_WGRAD(false, 128, 1);
    else if (multitap) HP_WGRAD(false, 64, 4);
    else HP_WGRAD(false, 64, 1);
#undef HP_WGRAD
#undef HP_WGRAD_XH
#undef HP_WGRAD_P
  }
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
