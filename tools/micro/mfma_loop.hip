// Microbenchmark: ceiling of the k_igemm inner loop on gfx950 (dev tool, not part of the library).
//   v0: bare v_mfma_f32_32x32x2_f32, 4 independent accumulators per wave
//   v1: + the 4 ds_read_b32 fragment reads per k-step of the real kernel (LDK = 33 rows)
//   v2: v1 + one barrier per 16 k-steps
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int V>
__global__ __launch_bounds__(256) void k(float* out, int iters, int lds_floats_dummy, const float* __restrict__ src, long src_mask) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 256 * 33];
  __shared__ __attribute__((aligned(16))) float dma[2 * 256 * 32];  // LDS-DMA landing zone (two dense tiles)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2 * 256 * 33; i += 256) smem[i] = (float)(i % 7) * 0.25f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const float* ap = smem + ((wave >> 1) * 64 + (lane & 31)) * 33 + (lane >> 5);
  const float* bp = smem + 128 * 33 + ((wave & 1) * 64 + (lane & 31)) * 33 + (lane >> 5);
  float fa[2] = {1.f + lane, 2.f}, fb[2] = {0.5f, 0.25f * lane};
  const int kq = tid & 7, r0 = tid >> 3;
  float4 st[8];
  for (int i = 0; i < 8; ++i) st[i] = make_float4(0.1f * i, 0.2f, 0.3f, 0.4f);
  long goff = ((long)blockIdx.x * 977 + r0) * 64 + kq * 4;
  for (int it = 0; it < iters; ++it) {
    if (V == 7) {
      // 8 LDS-DMA pieces per wave per tile: lane i of piece p lands at dma + (wave*8 + p) * 256 floats + 4 i
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((goff + (long)i * 32 * 64) & src_mask)),
                                         (__attribute__((address_space(3))) void*)(dma + ((it & 1) * 8192 + (wave * 8 + i) * 256)), 16, 0, 0);
      goff += 8 * 32 * 64 + 64 * 13;
    }
    if (V == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) st[i] = *(const float4*)(src + ((goff + (long)i * 32 * 64) & src_mask));
      goff += 8 * 32 * 64 + 64 * 13;
    }
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      if (V == 5 && (kk & 1) == 0) {
        st[kk >> 1] = *(const float4*)(src + ((goff + (long)(kk >> 1) * 32 * 64) & src_mask));
        if (kk == 14) goff += 8 * 32 * 64 + 64 * 13;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (V == 6 && kk < 8) {
        // 4-byte loads: four per k-step for the first 8 k-steps (32 loads, same bytes)
        float* sp = (float*)&st[kk];
        const float* gp = src + ((goff + (long)kk * 32 * 64) & src_mask);
        sp[0] = gp[0]; sp[1] = gp[1]; sp[2] = gp[2]; sp[3] = gp[3];
        if (kk == 7) goff += 8 * 32 * 64 + 64 * 13;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (V >= 1) {
        fa[0] = ap[2 * kk];
        fa[1] = ap[32 * 33 + 2 * kk];
        fb[0] = bp[2 * kk];
        fb[1] = bp[32 * 33 + 2 * kk];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (V >= 2) __syncthreads();
    if (V >= 3 && V != 7) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float* d = smem + (r0 + 32 * i) * 33 + kq * 4;
        d[0] = st[i].x;
        d[1] = st[i].y;
        d[2] = st[i].z;
        d[3] = st[i].w;
      }
      __syncthreads();
    }
  }
  float s = dma[tid];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int V>
void run(const char* name, int blocks, int iters, float* d, const float* src, long mask) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters, 0, src, mask);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(256), 0, 0, d, iters, 0, src, mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flops = (double)blocks * 4 * iters * 16 * 4 * (2.0 * 32 * 32 * 2);
  printf("%-40s blocks %5d  %8.3f ms  %7.1f TF\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  float* d;
  hipMalloc(&d, sizeof(float) * 256 * 4096);
  const int iters = 2000;
  float* src;
  const long big = 1l << 30;  // 4 GiB of floats
  hipMalloc(&src, sizeof(float) * big);
  hipMemset(src, 0, sizeof(float) * big);
  for (int bpc : {2, 3}) {
    printf("-- %d block(s) per CU\n", bpc);
    run<2>("MFMA + ds_read + barrier", 256 * bpc, iters, d, src, 0);
    run<3>("  + 32 LDS writes + 2nd barrier / tile", 256 * bpc, iters, d, src, 0);
    run<4>("  + 8 float4 loads / tile, 16 MiB window (L2)", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<4>("  + 8 float4 loads / tile, 4 GiB window (HBM)", 256 * bpc, iters, d, src, big - 1);
    run<7>("  LDS-DMA: 8 x 1 KiB pieces / wave / tile (L2)", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<5>("  loads spread: 1 float4 per 2 k-steps (L2)", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<6>("  loads as 32 dword loads over 8 k-steps (L2)", 256 * bpc, iters, d, src, (1l << 22) - 1);
  }
  return 0;
}
