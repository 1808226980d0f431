// Internal helpers shared by the translation units of libhiddenpose_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include "hp_host.h"

namespace hp {

#define HP_CHECK_HIP(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      ::hp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return HP_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

// Optional per-kernel timing with HIP events on the launch stream (hp_profile_* in the C ABI).
// Disabled by default: one relaxed atomic load per launch.
struct ProfScope {
  ProfScope(const char* name, hipStream_t st);
  ~ProfScope();
  int slot = -1;
  hipStream_t st = nullptr;
  hipEvent_t stop = nullptr;
};
#define HP_PROF(name, stream) ::hp::ProfScope _hp_prof_scope(name, stream)

// ---- activation tensors stored as fp32 or bf16 (BASELINE configs[2]: bf16 with fp32 accumulators / statistics).  The
// element type of a tensor is a RUNTIME flag of the call (`half` != 0: bf16), so one kernel serves both layouts: the
// branch is uniform over the launch and the conversion is a shift (load) or one v_cvt_pk_bf16_f32 per pair (store).
#ifdef __HIPCC__
__device__ __forceinline__ float4 hp_ld4(const void* p, long i, int half) {  // elements i .. i+3 (i % 4 == 0)
  if (half) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + i);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ float hp_ld1(const void* p, long i, int half) {
  if (half) return __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(p)[i] << 16);
  return reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ unsigned short hp_f2bf(float v) {  // round to nearest even, NaN stays NaN
  typedef __attribute__((ext_vector_type(2))) float hp_f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 hp_bf16x2;
  const hp_bf16x2 h = __builtin_convertvector((hp_f32x2){v, 0.f}, hp_bf16x2);
  return __builtin_bit_cast(unsigned int, h) & 0xffffu;
}
__device__ __forceinline__ void hp_st4(void* p, long i, float4 v, int half) {
  if (half) {
    typedef __attribute__((ext_vector_type(2))) float hp_f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 hp_bf16x2;
    const hp_bf16x2 lo = __builtin_convertvector((hp_f32x2){v.x, v.y}, hp_bf16x2);
    const hp_bf16x2 hi = __builtin_convertvector((hp_f32x2){v.z, v.w}, hp_bf16x2);
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p) + i) =
        make_uint2(__builtin_bit_cast(unsigned int, lo), __builtin_bit_cast(unsigned int, hi));
    return;
  }
  *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = v;
}
// num_records of a raw buffer descriptor whose base lies `base_off` elements into a tensor of `total` elements of `esz`
// bytes: the tensor's TRUE remaining extent, so that an addressing slip reads zeros / drops the store instead of touching
// whatever is allocated next to the tensor (round 3's descriptors all said 2^31: the range check was the sign bit only).
// Capped at 2^31: offsets with bit 31 set stay out of range (the "force this lane out of range" idiom of the tile loads),
// and tensors whose remainder genuinely exceeds 2 GiB (layer 1 and the stem at the headline shape) keep that cap.
// A base below the tensor (negative base_off: tiles whose first taps reach outside the volume, masked per lane) only
// lengthens the extent; the low end is guarded by those masks, not by the descriptor.
__device__ __forceinline__ unsigned hp_extent(long total, long base_off, int esz) {
  const long bytes = (total - base_off) * esz;
  return bytes >= 0x80000000l ? 0x80000000u : bytes > 0 ? (unsigned)bytes : 0u;
}
__device__ __forceinline__ void hp_st1(void* p, long i, float v, int half) {
  if (half) reinterpret_cast<unsigned short*>(p)[i] = hp_f2bf(v);
  else reinterpret_cast<float*>(p)[i] = v;
}
#endif

}  // namespace hp
