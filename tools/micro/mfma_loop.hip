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
  constexpr int SMEM_ROWS = (V == 10 || V == 12) ? 512 : 256;   // two buffers only where a variant uses them: occupancy as in the real kernel
  __shared__ __attribute__((aligned(16))) float smem[SMEM_ROWS * 36];
  __shared__ __attribute__((aligned(16))) float dma[(V == 7 || V == 9) ? 2 * 256 * 32 : 64];  // LDS-DMA landing zone (two dense tiles)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < SMEM_ROWS * 36; i += 256) smem[i] = (float)(i % 7) * 0.25f;
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
  // V8 / V9 (round 3): the same eight 16-byte loads per thread and tile as BUFFER loads whose per-thread offset is fixed and
  // whose tile displacement is a scalar -- no vector instruction per load.  V8: into registers (+ the 32 LDS writes of V3);
  // V9: straight into LDS (LDS-DMA), no writes at all.  (V4 / V7 spend ~6 vector-ALU instructions per load on `goff`.)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x80000000u, 0x00020000);
  const unsigned voff = (unsigned)(((blockIdx.x * 977 + r0) * 64 + kq * 4) * 4) & 0x3fffffu;
  for (int it = 0; it < iters; ++it) {
    if (V == 8 || V == 9) {
      const unsigned so = (unsigned)(((it * 2053) & 0x3ff) << 12) & (unsigned)(src_mask * 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (V == 8)
          st[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + i * 4096, 0));
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dma + ((it & 1) * 8192 + (wave * 8 + i) * 256)), 16,
                                                   voff, so + i * 4096, 0, 0);
      }
    }
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
    if (V >= 10) {
      // V10: two LDS buffers (pitch 33, dword reads / writes), ONE barrier per tile; V11: one buffer, 16-byte fragment reads and
      // staging writes (pitch 36, K permuted: lane half h owns K quads 2s + h), two barriers; V12: both
      constexpr bool DB = V == 10 || V == 12, W4 = V == 11 || V == 12;
      constexpr int LP = W4 ? 36 : 33;
      const float* rb = smem + (DB ? (it & 1) * 256 * LP : 0);
      float* wb = smem + (DB ? ((it + 1) & 1) * 256 * LP : 0);
      const unsigned so = (unsigned)(((it * 2053) & 0x3ff) << 12) & (unsigned)(src_mask * 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) st[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, so + i * 4096, 0));
      if (W4) {
        const float* a4 = rb + ((wave >> 1) * 64 + (lane & 31)) * LP + 4 * (lane >> 5);
        const float* b4 = rb + 128 * LP + ((wave & 1) * 64 + (lane & 31)) * LP + 4 * (lane >> 5);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const float4 qa0 = *(const float4*)(a4 + 8 * s4), qa1 = *(const float4*)(a4 + 32 * LP + 8 * s4);
          const float4 qb0 = *(const float4*)(b4 + 8 * s4), qb1 = *(const float4*)(b4 + 32 * LP + 8 * s4);
          const float xa0[4] = {qa0.x, qa0.y, qa0.z, qa0.w}, xa1[4] = {qa1.x, qa1.y, qa1.z, qa1.w};
          const float xb0[4] = {qb0.x, qb0.y, qb0.z, qb0.w}, xb1[4] = {qb1.x, qb1.y, qb1.z, qb1.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa0[e], xb0[e], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa0[e], xb1[e], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa1[e], xb0[e], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa1[e], xb1[e], acc[1][1], 0, 0, 0);
          }
        }
      } else {
        const float* a1 = rb + ((wave >> 1) * 64 + (lane & 31)) * LP + (lane >> 5);
        const float* b1 = rb + 128 * LP + ((wave & 1) * 64 + (lane & 31)) * LP + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
          fa[0] = a1[2 * kk];
          fa[1] = a1[32 * LP + 2 * kk];
          fb[0] = b1[2 * kk];
          fb[1] = b1[32 * LP + 2 * kk];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
      }
      if (!DB) __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float* d = wb + (r0 + 32 * i) * LP + kq * 4;
        if (W4) {
          *(float4*)d = st[i];
        } else {
          d[0] = st[i].x;
          d[1] = st[i].y;
          d[2] = st[i].z;
          d[3] = st[i].w;
        }
      }
      __syncthreads();
      continue;
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
    if (V >= 3 && V != 7 && V != 9) {
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
  float s = dma[tid & 63];
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
    run<8>("  BUFFER loads, scalar tile offset -> regs (L2)", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<9>("  BUFFER loads -> LDS (DMA), scalar offset (L2)", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<10>("  BUFFER -> regs, TWO LDS buffers, 1 barrier", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<11>("  BUFFER -> regs, 16-byte LDS reads+writes", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<12>("  BUFFER -> regs, two buffers + 16-byte LDS", 256 * bpc, iters, d, src, (1l << 22) - 1);
  }
  return 0;
}
