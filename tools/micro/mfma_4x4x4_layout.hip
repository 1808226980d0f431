// Operand layout of v_mfma_f32_4x4x4_16b_bf16, checked against the model the thin-channel bf16 kernels assume:
//   D[lane l][reg i] = sum_k A[lane 4*(l/4) + i][k] * B[lane l][k]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
using s16x4 = __attribute__((ext_vector_type(4))) short;
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void k(const s16x4* a, const s16x4* b, f32x4* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[l], b[l], acc, 0, 0, 0);
  d[l] = acc;
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned short ha[64][4], hb[64][4];
  float A[64][4], B[64][4], D[64][4];
  srand(1);
  for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) {
    ha[l][k] = f2bf((float)(rand() % 17 - 8)); hb[l][k] = f2bf((float)(rand() % 17 - 8));
    A[l][k] = bf2f(ha[l][k]); B[l][k] = bf2f(hb[l][k]);
  }
  void *da, *db, *dd;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dd, sizeof(D));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const s16x4*)da, (const s16x4*)db, (f32x4*)dd);
  hipMemcpy(D, dd, sizeof(D), hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
    float m1 = 0, m2 = 0;
    for (int kk = 0; kk < 4; ++kk) { m1 += A[4 * (l / 4) + i][kk] * B[l][kk]; m2 += A[l][kk] * B[4 * (l / 4) + i][kk]; }
    e1 = fmax(e1, fabs(m1 - D[l][i])); e2 = fmax(e2, fabs(m2 - D[l][i]));
  }
  printf("model D[l][i] = sum_k A[4b+i][k] B[l][k]: max err %g\nmodel with A and B roles swapped: max err %g\n", e1, e2);
  for (int l = 0; l < 8; ++l) printf("lane %d: D = %g %g %g %g\n", l, D[l][0], D[l][1], D[l][2], D[l][3]);
  return 0;
}
