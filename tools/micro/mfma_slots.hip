// Microbenchmark (dev tool): what does ONE wave per SIMD pay for K other instructions placed between its fp32 MFMAs?
// Loop body = 4 independent v_mfma_f32_32x32x2_f32 (256 cycles of matrix pipe) + K instructions of one kind; reported:
// cycles per loop iteration (s_memtime), i.e. 256 = free, 256 + K * c = each instruction costs c cycles of pipe time.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__device__ unsigned long long g_cyc[256 * 4];

template <int KIND, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, const float* src, int iters) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  lds[tid] = tid;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  const float fa = 1.f + lane, fb = 0.5f * lane;
  float x[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  float4 ld[8];
  for (int i = 0; i < 8; ++i) ld[i] = make_float4(0, 0, 0, 0);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1u << 30, 0x00020000);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < (K + 3 - a) / 4; ++j) {
        const int u = (a * ((K + 3) / 4) + j) & 7;
        if (KIND == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(x[u]) : "v"(fa));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[u]) : "v"(fa), "v"(fb));
        if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(x[u]) : "v"((unsigned)(lane * 4 + 256 * u)) : "memory");
        if (KIND == 3) asm volatile("ds_write_b32 %0, %1" ::"v"((unsigned)(lane * 4 + 256 * u + 8192)), "v"(fa) : "memory");
        if (KIND == 4) ld[u] = *(const float4*)(src + ((it * 64 + lane) & 4095) * 4 + u * 16384);
        if (KIND == 5) ld[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(lane * 16 + u * 65536), (it & 255) * 1024, 0));
        if (KIND == 6) asm volatile("s_nop 0");
        if (KIND == 7) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[u]) : "v"(fa), "v"(fb));
      }
    }
    if (KIND == 2 || KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (KIND == 4 || KIND == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 16; ++i) r += acc[a][i];
  for (int i = 0; i < 8; ++i) r += x[i] + ld[i].x + ld[i].w;
  out[blockIdx.x * 64 * WAVES + tid] = r;
  if (lane == 0 && tid < 256) g_cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND, int K, int WAVES>
double run(float* d, const float* src) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<KIND, K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, src, iters);
  hipDeviceSynchronize();
  static unsigned long long h[256 * 4];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cyc), sizeof(h));
  double s = 0;
  for (int i = 0; i < 256 * 4; ++i) s += (double)h[i];
  return s / (256 * 4) / iters;
}

template <int KIND>
void row(const char* name, float* d, const float* src) {
  printf("%-26s 1 wave/SIMD: K=0 %6.1f  K=4 %6.1f  K=8 %6.1f  K=16 %6.1f  K=32 %6.1f | 2 waves/SIMD: K=0 %6.1f K=8 %6.1f K=16 %6.1f K=32 %6.1f  (cycles per 4 MFMAs per wave)\n",
         name, run<KIND, 0, 4>(d, src), run<KIND, 4, 4>(d, src), run<KIND, 8, 4>(d, src), run<KIND, 16, 4>(d, src), run<KIND, 32, 4>(d, src),
         run<KIND, 0, 8>(d, src), run<KIND, 8, 8>(d, src), run<KIND, 16, 8>(d, src), run<KIND, 32, 8>(d, src));
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float *d, *src;
  hipMalloc(&d, sizeof(float) * 256 * 512);
  hipMalloc(&src, 1u << 26);
  hipMemset(src, 0, 1u << 26);
  row<0>("v_mov_b32", d, src);
  row<1>("v_fma_f32", d, src);
  row<7>("v_cndmask_b32", d, src);
  row<6>("s_nop", d, src);
  row<2>("ds_read_b32", d, src);
  row<3>("ds_write_b32", d, src);
  row<4>("global_load_dwordx4 (L2)", d, src);
  row<5>("buffer_load_dwordx4 (L2)", d, src);
  return 0;
}
