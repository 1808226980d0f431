// Microbenchmark: which exact-f32 MFMA shape holds the higher clock under load on gfx950 (dev tool).
// Same output tile per wave (64 x 64), same LDS bytes per MAC, random operands, 2 or 3 workgroups per CU:
//   A: v_mfma_f32_32x32x2_f32, 2 x 2 tiles of 32 x 32 (the shape k_igemm / k_wgrad use)
//   B: v_mfma_f32_16x16x4_f32, 4 x 4 tiles of 16 x 16
// REG variants keep the fragments in registers (no LDS reads) to separate the matrix pipe from the LDS path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int LDM = 144;  // K-major tile rows: 128 + 16 floats, so the 4 k rows of a 16x16x4 fragment cover all banks

template <int SHAPE, bool REG>
__global__ __launch_bounds__(256) void k(float* out, int iters, const float* __restrict__ init) {
  __shared__ __attribute__((aligned(16))) float As[32 * LDM], Bs[32 * LDM];  // [k][m], [k][n]: K tile of 32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * LDM; i += 256) {
    As[i] = init[(blockIdx.x * 7919 + i) & 0xFFFFF];
    Bs[i] = init[(blockIdx.x * 104729 + i + 4096) & 0xFFFFF];
  }
  __syncthreads();
  const int m0 = (wave >> 1) * 64, n0 = (wave & 1) * 64;
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float fa[2] = {As[lane], As[lane + 64]}, fb[2] = {Bs[lane], Bs[lane + 64]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        if (!REG) {
          const int krow = 2 * kk + (lane >> 5);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fa[i] = As[krow * LDM + m0 + i * 32 + (lane & 31)];
            fb[i] = Bs[krow * LDM + n0 + i * 32 + (lane & 31)];
          }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  } else {
    f32x4 acc[4][4];
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b)
        for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.f;
    float fa[4], fb[4];
    for (int i = 0; i < 4; ++i) {
      fa[i] = As[lane + 64 * i];
      fb[i] = Bs[lane + 64 * i];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if (!REG) {
          const int krow = 4 * kk + (lane >> 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            fa[i] = As[krow * LDM + m0 + i * 16 + (lane & 15)];
            fb[i] = Bs[krow * LDM + n0 + i * 16 + (lane & 15)];
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b)
        for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  }
  out[blockIdx.x * 256 + tid] = s;
}

template <int SHAPE, bool REG>
void run(const char* name, int blocks, int iters, float* d, const float* init) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<SHAPE, REG>), dim3(blocks), dim3(256), 0, 0, d, iters, init);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<SHAPE, REG>), dim3(blocks), dim3(256), 0, 0, d, iters, init);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 10;
  const double flops = (double)blocks * 4 * iters * (2.0 * 64 * 64 * 32);
  printf("%-44s blocks %5d  %8.3f ms  %7.1f TF\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  float *d, *init;
  hipMalloc(&d, sizeof(float) * 256 * 4096);
  const int n = 1 << 20;
  float* h = (float*)malloc(sizeof(float) * n);
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMalloc(&init, sizeof(float) * n);
  hipMemcpy(init, h, sizeof(float) * n, hipMemcpyHostToDevice);
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep)
    for (int bpc : {2, 3}) {
      printf("-- %d workgroup(s) per CU, random operands\n", bpc);
      run<32, false>("32x32x2  fragments from LDS", 256 * bpc, iters, d, init);
      run<16, false>("16x16x4  fragments from LDS", 256 * bpc, iters, d, init);
      run<32, true>("32x32x2  fragments in registers", 256 * bpc, iters, d, init);
      run<16, true>("16x16x4  fragments in registers", 256 * bpc, iters, d, init);
    }
  return 0;
}
