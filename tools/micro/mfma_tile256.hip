// Microbenchmark: would a 256x256 block tile (8 waves, 128x64 per wave) lift the fp32 implicit-GEMM ceiling?
// Same per-thread traffic per K tile as the 128x128 kernel (8 float4 global loads, 32 LDS writes, 2 barriers),
// twice the MFMAs per wave.  Dev tool, not part of the library.
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int TM, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters, const float* __restrict__ src, long mask) {
  constexpr int CT = WAVES * 64;
  constexpr int ROWS = 64 * TM * (WAVES / 2) / 2 + 128;  // A rows + B rows (just sized generously)
  __shared__ float smem[(WAVES == 8 ? 512 : 256) * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (WAVES == 8 ? 512 : 256) * 33; i += CT) smem[i] = (float)(i % 7) * 0.25f;
  __syncthreads();
  f32x16 acc[TM][2];
  for (int a = 0; a < TM; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const float* ap = smem + ((wave & 1) * TM * 32 + (lane & 31)) * 33 + (lane >> 5);
  const float* bp = smem + 256 * 33 + ((wave >> 1) * 64 + (lane & 31)) * 33 + (lane >> 5);
  const int kq = tid & 7, r0 = tid >> 3;
  float4 st[8];
  long goff = ((long)blockIdx.x * 977 + r0) * 64 + kq * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = *(const float4*)(src + ((goff + (long)i * 32 * 64) & mask));
    goff += 8 * 32 * 64 + 64 * 13;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float fa[TM], fb[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = ap[i * 32 * 33 + 2 * kk];
      fb[0] = bp[2 * kk];
      fb[1] = bp[32 * 33 + 2 * kk];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float* d = smem + ((r0 + (CT / 8) * i) % (WAVES == 8 ? 512 : 256)) * 33 + kq * 4;
      d[0] = st[i].x;
      d[1] = st[i].y;
      d[2] = st[i].z;
      d[3] = st[i].w;
    }
    __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < TM; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * CT + tid] = s;
}

template <int TM, int WAVES>
void run(const char* name, int blocks, int iters, float* d, const float* src, long mask) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<TM, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, iters, src, mask);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<TM, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, iters, src, mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flops = (double)blocks * WAVES * iters * 16 * TM * 2 * (2.0 * 32 * 32 * 2);
  printf("%-52s blocks %5d  %8.3f ms  %7.1f TF\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  float *d, *src;
  hipMalloc(&d, sizeof(float) * 512 * 4096);
  const long big = 1l << 28;
  hipMalloc(&src, sizeof(float) * big);
  hipMemset(src, 0, sizeof(float) * big);
  const int iters = 1000;
  run<2, 4>("128x128 tile, 4 waves (64x64 each), 2 blocks/CU", 512, iters, d, src, (1l << 22) - 1);
  run<2, 4>("128x128 tile, 4 waves (64x64 each), 3 blocks/CU", 768, iters, d, src, (1l << 22) - 1);
  run<4, 8>("256x256 tile, 8 waves (128x64 each), 1 block/CU", 256, iters, d, src, (1l << 22) - 1);
  run<4, 8>("256x256 tile, 8 waves (128x64 each), 2 blocks/CU", 512, iters, d, src, (1l << 22) - 1);
  run<4, 4>("256x128 tile, 4 waves (128x64 each), 2 blocks/CU", 512, iters, d, src, (1l << 22) - 1);
  return 0;
}
