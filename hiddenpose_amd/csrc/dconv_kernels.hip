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
  const int VX = vx_for(cout), TX = 8 * VX;
  const int tiles_x = (Wo + TX - 1) / TX, tiles_y = (Ho + TY - 1) / TY, tiles_z = (Do + TZ - 1) / TZ;
  dim3 grid((unsigned)(tiles_x * tiles_y * tiles_z), (unsigned)B);
#define HP_DCONV_CASE(CO, V)                                                                                      \
  case CO:                                                                                                        \
    launch_dconv<CO, V>(padmode, grid, st, x, w, bias, y, cin, Di, Hi, Wi, Do, Ho, Wo, pad, wsco, wsci, flip, tiles_x, \
                        tiles_y);                                                                                 \
    break;
  switch (cout) {
    HP_DCONV_CASE(1, 4)
    HP_DCONV_CASE(4, 4)
    HP_DCONV_CASE(8, 4)
    HP_DCONV_CASE(16, 2)
    HP_DCONV_CASE(32, 2)
    HP_DCONV_CASE(64, 2)
    default:
      set_error("direct conv: unsupported output channel count %d (1,4,8,16,32,64)", cout);
      return HP_ERR_UNSUPPORTED;
  }
#undef HP_DCONV_CASE
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

extern "C" int hp_dconv3_backward_weight(const float* x, const float* gy, float* dw, float* dbias, int B, int cin,
                                         int cout, int D, int H, int W, int replicate_pad, void* stream) {
  HP_REQUIRE(x && gy && dw && B > 0, "hp_dconv3_backward_weight: bad argument");
  hipStream_t st = (hipStream_t)stream;
  HP_CHECK_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)cout * cin * 27, st));
  if (dbias) HP_CHECK_HIP(hipMemsetAsync(dbias, 0, sizeof(float) * cout, st));
  const int tiles_x = (W + WTX - 1) / WTX, tiles_y = (H + TY - 1) / TY, tiles_z = (D + TZ - 1) / TZ;
  const int co_chunks = (cout + WCO - 1) / WCO, ci_chunks = (cin + CC - 1) / CC;
  dim3 grid((unsigned)(tiles_x * tiles_y * tiles_z), (unsigned)B, (unsigned)(co_chunks * ci_chunks));
  HP_PROF("dconv3_wgrad", st);
  if (replicate_pad)
    hipLaunchKernelGGL((k_dconv3_wgrad<1>), grid, dim3(DT), 0, st, x, gy, dw, dbias, cin, cout, D, H, W, tiles_x, tiles_y,
                       co_chunks, ci_chunks);
  else
    hipLaunchKernelGGL((k_dconv3_wgrad<0>), grid, dim3(DT), 0, st, x, gy, dw, dbias, cin, cout, D, H, W, tiles_x, tiles_y,
                       co_chunks, ci_chunks);
  HP_CHECK_HIP(hipGetLastError());
  return HP_OK;
}
