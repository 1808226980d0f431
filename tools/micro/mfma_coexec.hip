// Microbenchmark (dev tool): does other work issued by a SECOND wave on the same SIMD overlap with the first wave's
// MFMAs?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a fixed count of back-to-back MFMAs, waves 4-7
// (their SIMD partners) run a fixed count of "other" instructions.  T(both) ~ max(T_mfma, T_other) = the two overlap;
// T(both) ~ T_mfma + T_other = they serialise on the SIMD.
//   MF: 0 = v_mfma_f32_32x32x2_f32 (exact fp32), 1 = v_mfma_f32_32x32x16_bf16
//   OT: 0 = v_fma_f32 chain(s), 1 = s_add chain (SALU), 2 = ds_write_b32, 3 = ds_read_b32, 4 = v_mov (independent VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ unsigned long long g_cyc[256 * 8];
template <int MF, int OT>
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_other, int who) {
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
  __shared__ float lds[4096];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  lds[tid] = tid;
  __syncthreads();
  float r = 0.f;
  if (wave < 4) {
    if (!(who & 1)) return;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
      for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    const float fa = 1.f + lane, fb = 0.5f * lane;
    bf16x8 ha, hb;
    for (int i = 0; i < 8; ++i) ha[i] = (__bf16)(float)(lane + i), hb[i] = (__bf16)(0.25f * i);
    for (int it = 0; it < n_mfma; it += 4) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (MF == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[a], 0, 0, 0);
        else acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc[a], 0, 0, 0);
      }
    }
    for (int a = 0; a < 4; ++a)
      for (int i = 0; i < 16; ++i) r += acc[a][i];
  } else {
    if (!(who & 2)) return;
    if (OT == 0) {
      float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
      for (int it = 0; it < n_other; it += 4) {
        asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(1.0001f), "v"(0.5f));
      }
      r = x0 + x1 + x2 + x3;
    } else if (OT == 1) {
      int s0 = n_other, s1 = 1, s2 = 2, s3 = 3;
      for (int it = 0; it < n_other; it += 4) {
        asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
      }
      r = s0 + s1 + s2 + s3;
    } else if (OT == 2) {
      for (int it = 0; it < n_other; it += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"((unsigned)(lane * 4 + 2048)), "v"(r), "n"(0) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    } else if (OT == 3) {
      float t0, t1, t2, t3;
      for (int it = 0; it < n_other; it += 4) {
        asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)"
                     : "=v"(t0), "=v"(t1), "=v"(t2), "=v"(t3) : "v"((unsigned)(lane * 4)) : "memory");
        r += t0 + t1 + t2 + t3;
      }
    } else if (OT == 5) {
      const float4* p = (const float4*)out;   // L2-resident window
      float4 a0, a1, a2, a3;
      for (int it = 0; it < n_other; it += 4) {
        const int o = ((it * 64 + lane) & 16383);
        asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:1024\n global_load_dwordx4 %2, %4, off offset:2048\n global_load_dwordx4 %3, %4, off offset:3072\n s_waitcnt vmcnt(0)"
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(p + o) : "memory");
        r += a0.x + a1.x + a2.x + a3.x;
      }
    } else {
      float x0 = lane, x1, x2, x3, x4;
      for (int it = 0; it < n_other; it += 4) {
        asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "=v"(x1), "=v"(x2), "=v"(x3), "=v"(x4) : "v"(x0));
      }
      r = x1 + x2 + x3 + x4;
    }
  }
  out[blockIdx.x * 512 + tid] = r;
  if ((threadIdx.x & 63) == 0) g_cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime() - t_start;
}

template <int MF, int OT>
float run(float* d, int nm, int no, int who) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MF, OT>), dim3(256), dim3(512), 0, 0, d, nm, no, who);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<MF, OT>), dim3(256), dim3(512), 0, 0, d, nm, no, who);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) printf("HIP error\n");
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

static void roles(double& mf, double& ot) {
  static unsigned long long h[256 * 8];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cyc), sizeof(h));
  mf = ot = 0;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) (w < 4 ? mf : ot) += (double)h[b * 8 + w] / (256 * 4);
}
template <int MF, int OT>
void pair(const char* name, float* d, int nm, int no) {
  double m1, o1, m2, o2, m3, o3;
  const float a = run<MF, OT>(d, nm, no, 1);
  roles(m1, o1);
  const float b = run<MF, OT>(d, nm, no, 2);
  roles(m2, o2);
  const float c = run<MF, OT>(d, nm, no, 3);
  roles(m3, o3);
  printf("%-40s alone: mfma %6.3f ms (%8.0f kcyc) other %6.3f ms (%8.0f kcyc) | together %6.3f ms: mfma waves %8.0f kcyc (x%.2f), other waves %8.0f kcyc (x%.2f)\n",
         name, a, m1 / 1e3, b, o2 / 1e3, c, m3 / 1e3, m3 / m1, o3 / 1e3, o3 / o2);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  float* d;
  hipMalloc(&d, sizeof(float) * 256 * 512);
  const int nm32 = 8000;            // 40000 x 64 cycles
  const int nm16 = 16000;           // 80000 x 32 cycles
  pair<0, 0>("fp32 32x32x2 MFMA  | v_fma_f32 (VALU)", d, nm32, 64000);
  pair<0, 0>("fp32 MFMA x4 work  | v_fma_f32 (VALU)", d, 4 * nm32, 64000);
  pair<0, 5>("fp32 32x32x2 MFMA  | global_load_dwordx4", d, nm32, 16000);
  pair<1, 5>("bf16 32x32x16 MFMA | global_load_dwordx4", d, nm16, 16000);
  pair<0, 4>("fp32 32x32x2 MFMA  | v_mov_b32 (VALU)", d, nm32, 64000);
  pair<0, 1>("fp32 32x32x2 MFMA  | s_add_u32 (SALU)", d, nm32, 64000);
  pair<0, 2>("fp32 32x32x2 MFMA  | ds_write_b32", d, nm32, 32000);
  pair<0, 3>("fp32 32x32x2 MFMA  | ds_read_b32", d, nm32, 64000);
  pair<1, 0>("bf16 32x32x16 MFMA | v_fma_f32 (VALU)", d, nm16, 64000);
  pair<1, 4>("bf16 32x32x16 MFMA | v_mov_b32 (VALU)", d, nm16, 64000);
  pair<1, 1>("bf16 32x32x16 MFMA | s_add_u32 (SALU)", d, nm16, 64000);
  pair<1, 2>("bf16 32x32x16 MFMA | ds_write_b32", d, nm16, 32000);
  return 0;
}
