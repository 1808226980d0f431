// Microbenchmark (dev tool), companion of mfma_slots.hip: what do WIDE LDS instructions cost a wave between its fp32 MFMAs?
// Loop body = 4 independent v_mfma_f32_32x32x2_f32 (256 cycles of matrix pipe) + K LDS instructions of one kind; reported:
// cycles per loop iteration.  Question behind it: is an LDS instruction's cost per instruction (then b128 fragments with a
// permuted K order would cut the GEMM kernels' LDS overhead 4x) or per byte?
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ unsigned long long g_cyc[256 * 4];

template <int KIND, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16384; i += 64 * WAVES) lds[i] = i;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  const float fa = 1.f + lane, fb = 0.5f * lane;
  f32x4 x4[8];
  f32x2 x2[8];
  float x1[8];
  for (int i = 0; i < 8; ++i) x4[i] = (f32x4){0, 0, 0, 0}, x2[i] = (f32x2){0, 0}, x1[i] = 0;
  const f32x4 w4 = {fa, fb, fa, fb};
  const f32x2 w2 = {fa, fb};
  // rows of 36 floats (144 B): the conflict-free pitch for 16-byte fragment reads; lane -> row (lane & 31), half (lane >> 5)
  const unsigned rd = (unsigned)(((lane & 31) * 36 + 16 * (lane >> 5)) * 4 + (wave & 3) * 8192);
  const unsigned wr = (unsigned)(lane * 16 + (wave & 3) * 8192 + 32768);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < (K + 3 - a) / 4; ++j) {
        const int u = (a * ((K + 3) / 4) + j) & 7;
        if (KIND == 0) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x1[u]) : "v"(rd), "n"(0) : "memory");
        if (KIND == 1) asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:1" : "=v"(x2[u]) : "v"(rd) : "memory");
        if (KIND == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(x2[u]) : "v"(rd) : "memory");
        if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(x4[u]) : "v"(rd) : "memory");
        if (KIND == 4) asm volatile("ds_write_b32 %0, %1" ::"v"(wr), "v"(fa) : "memory");
        if (KIND == 5) asm volatile("ds_write2_b32 %0, %1, %2 offset0:0 offset1:1" ::"v"(wr), "v"(fa), "v"(fb) : "memory");
        if (KIND == 6) asm volatile("ds_write_b64 %0, %1" ::"v"(wr), "v"(w2) : "memory");
        if (KIND == 7) asm volatile("ds_write_b128 %0, %1" ::"v"(wr), "v"(w4) : "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 16; ++i) r += acc[a][i];
  for (int i = 0; i < 8; ++i) r += x1[i] + x2[i][0] + x2[i][1] + x4[i][0] + x4[i][3];
  out[blockIdx.x * 64 * WAVES + tid] = r;
  if (lane == 0 && tid < 256) g_cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND, int K, int WAVES>
double run(float* d) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, iters);
  hipDeviceSynchronize();
  static unsigned long long h[256 * 4];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cyc), sizeof(h));
  double s = 0;
  for (int i = 0; i < 256 * 4; ++i) s += (double)h[i];
  return s / (256 * 4) / iters;
}

template <int KIND>
void row(const char* name, float* d) {
  printf("%-16s 1 wave/SIMD: K=0 %6.1f  K=4 %6.1f  K=8 %6.1f  K=16 %6.1f  K=32 %6.1f | 2 waves/SIMD: K=0 %6.1f K=8 %6.1f K=16 %6.1f K=32 %6.1f  (cycles per 4 MFMAs per wave)\n",
         name, run<KIND, 0, 4>(d), run<KIND, 4, 4>(d), run<KIND, 8, 4>(d), run<KIND, 16, 4>(d), run<KIND, 32, 4>(d),
         run<KIND, 0, 8>(d), run<KIND, 8, 8>(d), run<KIND, 16, 8>(d), run<KIND, 32, 8>(d));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* d;
  hipMalloc(&d, sizeof(float) * 256 * 512);
  row<0>("ds_read_b32", d);
  row<1>("ds_read2_b32", d);
  row<2>("ds_read_b64", d);
  row<3>("ds_read_b128", d);
  row<4>("ds_write_b32", d);
  row<5>("ds_write2_b32", d);
  row<6>("ds_write_b64", d);
  row<7>("ds_write_b128", d);
  return 0;
}
