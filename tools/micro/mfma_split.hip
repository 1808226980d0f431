// Microbenchmark: does moving the global loads + LDS staging of the implicit-GEMM loop into separate producer
// waves (4 MFMA waves + 4 load waves per workgroup, double-buffered LDS tiles, one barrier per K tile) lift the
// ceiling that the fused loop (every wave loads, stages and multiplies: tools/micro/mfma_loop.hip) runs into?
//   F: fused, 256 threads (as k_igemm): 8 float4 loads + 32 LDS writes + 64 MFMA per wave per tile, 2 barriers
//   S: split, 512 threads: waves 0-3 only ds_read + MFMA, waves 4-7 only load + stage, 1 barrier
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ void mma_tile(const float* ap, const float* bp, f32x16 (&acc)[2][2]) {
  float fa[2][2], fb[2][2];
  fa[0][0] = ap[0];
  fa[0][1] = ap[32 * 33];
  fb[0][0] = bp[0];
  fb[0][1] = bp[32 * 33];
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk + 1 < 16) {
      fa[nxt][0] = ap[2 * (kk + 1)];
      fa[nxt][1] = ap[32 * 33 + 2 * (kk + 1)];
      fb[nxt][0] = bp[2 * (kk + 1)];
      fb[nxt][1] = bp[32 * 33 + 2 * (kk + 1)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i], fb[cur][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : 256) void k(float* out, int iters, const float* __restrict__ src, long src_mask) {
  __shared__ __attribute__((aligned(16))) float smem[(SPLIT ? 2 : 1) * 256 * 33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < (SPLIT ? 2 : 1) * 256 * 33; i += blockDim.x) smem[i] = src[(blockIdx.x * 7919 + i) & 0xFFFFF];
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int cw = wave & 3, ltid = tid & 255;
  const int aoff = ((cw >> 1) * 64 + (lane & 31)) * 33 + (lane >> 5);
  const int boff = 128 * 33 + ((cw & 1) * 64 + (lane & 31)) * 33 + (lane >> 5);
  const int kq = ltid & 7, r0 = ltid >> 3;
  float4 st[8];
  for (int i = 0; i < 8; ++i) st[i] = make_float4(0.1f * i, 0.2f, 0.3f, 0.4f);
  long goff = ((long)blockIdx.x * 977 + r0) * 64 + kq * 4;
  if constexpr (!SPLIT) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) st[i] = *(const float4*)(src + ((goff + (long)i * 32 * 64) & src_mask));
      goff += 8 * 32 * 64 + 64 * 13;
      mma_tile(smem + aoff, smem + boff, acc);
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float* d = smem + (r0 + 32 * i) * 33 + kq * 4;
        d[0] = st[i].x; d[1] = st[i].y; d[2] = st[i].z; d[3] = st[i].w;
      }
      __syncthreads();
    }
  } else {
    const bool producer = wave >= 4;
    for (int it = 0; it < iters; ++it) {
      float* cur = smem + (it & 1) * 256 * 33;
      float* nxt = smem + ((it & 1) ^ 1) * 256 * 33;
      if (producer) {
        // stage the tile fetched during the previous iteration, then fetch the one after it
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float* d = nxt + (r0 + 32 * i) * 33 + kq * 4;
          d[0] = st[i].x; d[1] = st[i].y; d[2] = st[i].z; d[3] = st[i].w;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) st[i] = *(const float4*)(src + ((goff + (long)i * 32 * 64) & src_mask));
        goff += 8 * 32 * 64 + 64 * 13;
      } else {
        mma_tile(cur + aoff, cur + boff, acc);
      }
      __syncthreads();
    }
  }
  float s = st[0].x + st[7].w;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int r = 0; r < 16; ++r) s += acc[a][b][r];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <bool SPLIT>
void run(const char* name, int blocks, int iters, float* d, const float* src, long mask) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int nt = SPLIT ? 512 : 256;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<SPLIT>, dim3(blocks), dim3(nt), 0, 0, d, iters, src, mask);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<SPLIT>, dim3(blocks), dim3(nt), 0, 0, d, iters, src, mask);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= 5;
  const double flops = (double)blocks * 4 * iters * 16 * 4 * (2.0 * 32 * 32 * 2);
  printf("%-52s blocks %5d  %8.3f ms  %7.1f TF\n", name, blocks, ms, flops / ms / 1e9);
}

int main() {
  float* d;
  hipMalloc(&d, sizeof(float) * 512 * 4096);
  const int iters = 2000;
  float* src;
  const long big = 1l << 30;
  hipMalloc(&src, sizeof(float) * big);
  {
    const int n = 1 << 24;
    float* h = (float*)malloc(sizeof(float) * n);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (long o = 0; o < big; o += n) hipMemcpy(src + o, h, sizeof(float) * n, hipMemcpyHostToDevice);
  }
  for (int bpc : {2, 3}) {
    printf("-- %d workgroup(s) per CU\n", bpc);
    run<false>("fused  (256 thr), loads from a 16 MiB window (L2)", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<false>("fused  (256 thr), loads from a 4 GiB window (HBM)", 256 * bpc, iters, d, src, big - 1);
    run<true>("split  (512 thr), loads from a 16 MiB window (L2)", 256 * bpc, iters, d, src, (1l << 22) - 1);
    run<true>("split  (512 thr), loads from a 4 GiB window (HBM)", 256 * bpc, iters, d, src, big - 1);
  }
  return 0;
}
