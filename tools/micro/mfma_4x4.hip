// Microbenchmark (dev tool): issue rate of v_mfma_f32_4x4x1_16b_f32 against v_pk_fma_f32 on gfx950.
//   v0: 8 independent 4x4x1 accumulator chains per wave     v1: 2 chains (as k_dconv3_mfma)
//   v2: v_pk_fma_f32, 8 independent chains                  v3: 32x32x2 (reference: 64 FLOP/clk/SIMD)
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_4x4.hip -o /tmp/mfma_4x4 ; run: /tmp/mfma_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  const int lane = threadIdx.x & 63;
  float a = a0 + lane * 1e-3f, b = b0 - lane * 1e-3f;
  if (V == 0 || V == 1) {
    constexpr int NC = V == 0 ? 8 : 2;
    f32x4 acc[NC];
    for (int c = 0; c < NC; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 64 / NC; ++u)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < NC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else if (V == 2) {
    f32x2 acc[8];
    for (int c = 0; c < 8; ++c) acc[c] = (f32x2){0.f, 0.f};
    const f32x2 av = {a, a * 0.5f}, bv = {b, b * 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = __builtin_elementwise_fma(av, bv, acc[c]);
    }
    float s = 0.f;
    for (int c = 0; c < 8; ++c) s += acc[c][0] + acc[c][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int V>
static void run(const char* name, double flop_per_inst, int inst_per_iter) {
  float* out;
  hipMalloc(&out, 2048 * 256 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(2048), dim3(256), 0, 0, out, 10, 1.0f, 2.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(2048), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 2048.0 * 4 * iters * inst_per_iter * flop_per_inst;
  printf("%-40s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run<0>("mfma 4x4x1_16b f32, 8 chains", 512, 64);
  run<1>("mfma 4x4x1_16b f32, 2 chains", 512, 64);
  run<2>("v_pk_fma_f32, 8 chains", 256, 64);
  run<3>("mfma 32x32x2 f32, 4 chains", 4096, 64);
  return 0;
}
