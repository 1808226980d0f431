/* libhiddenpose_hip.so -- C ABI of the MI355X (gfx950) NlosPose hot path.
 *
 * The reference (Hagtaril/HiddenPose) has no native/FFI seam: its hot path is
 * a chain of stock torch operators behind nn.Module.forward.  Each entry point
 * below replaces the operator sequence of one reference function; the Python
 * host side (the hiddenpose_amd Python package) mirrors the reference's module API on top of
 * these calls through ctypes.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain C, no framework types: device pointers are `void*`/`float*` into
 *    memory owned by the CALLER (PyTorch caching allocator on the Python side);
 *    the library allocates device memory only inside plan objects.
 *  - every call returns 0 on success or a negative hp_status; the message is
 *    available from hp_last_error_string() (thread local).  Nothing throws or
 *    exits across the ABI.
 *  - every compute call is asynchronous on the hipStream_t passed as `stream`
 *    (a `void*` here so that the header needs no HIP include).
 *  - tensors are fp32, contiguous, (B, C, T, H, W) as in the reference.
 *  - plans are immutable after creation and may be shared by streams/threads.
 */
#ifndef HIDDENPOSE_HIP_H
#define HIDDENPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hp_status {
  HP_OK = 0,
  HP_ERR_BAD_ARG = -1,     /* shape / size / null pointer */
  HP_ERR_UNSUPPORTED = -2, /* valid request the library has no kernel for */
  HP_ERR_HIP = -3,         /* a HIP runtime call failed; see hp_last_error_string() */
  HP_ERR_WORKSPACE = -4,   /* caller workspace too small */
  HP_ERR_NO_DEVICE = -5
} hp_status;

int hp_version(void);
const char* hp_last_error_string(void);

/* Per-kernel timing with HIP events recorded on the launch stream (off by default).
 * hp_profile_get(i, ...) synchronises the recorded events and returns the launch count
 * and total milliseconds of the i-th kernel name seen since hp_profile_reset(). */
int hp_profile_enable(int on);
int hp_profile_reset(void);
int hp_profile_count(void);
int hp_profile_get(int i, char* name, int name_cap, int64_t* launches, double* total_ms);

/* ------------------------------------------------------------------------
 * LCT physics layer.
 * Replaces models/feature_propagation.py: LCT._parpareparam :71-109,
 * _resamplingOperator :111-139, _definePsf :141-171, todev :173-184 (plan) and
 * LCT.forward :186-257 (hp_lct_forward), mode='lct'.
 * ---------------------------------------------------------------------- */
typedef struct hp_lct_plan hp_lct_plan;

enum { HP_MATERIAL_DIFFUSE = 0, HP_MATERIAL_SPECULAR = 1 };

/* Host-only constants (no GPU needed); used by the plan and by CPU tests.
 * gridz[T]; mtx: dense row-major T*T; psf_zidx: (2N*2N) int32, z index of the
 * PSF's 1 in every (x,y) column after the roll (first one if several);
 * psf_count: number of ones; invpsf_re/im: (2T,2N,2N) natural order, may be NULL. */
int hp_lct_host_constants(int T, int N, double bin_len, double wall_size,
                          float* gridz, float* mtx, int32_t* psf_zidx, int64_t* psf_count,
                          float* invpsf_re, float* invpsf_im);

/* Builds all constants on the host, uploads them to `device` (HIP ordinal). */
int hp_lct_plan_create(hp_lct_plan** plan, int T, int N, double bin_len, double wall_size,
                       int material, int device);
int hp_lct_plan_destroy(hp_lct_plan* plan);
/* Bytes of caller-provided scratch needed for a batch of B volumes (B*D in
 * the reference's naming). */
size_t hp_lct_workspace_bytes(const hp_lct_plan* plan, int batch);
/* y = LCT(x): x,y (batch, T, N, N) fp32 device pointers; tbe=0, ten=T. */
int hp_lct_forward(const hp_lct_plan* plan, const float* x, float* y, int batch,
                   void* workspace, size_t workspace_bytes, void* stream);
/* gx = LCT^T(gy) (vector-Jacobian product of hp_lct_forward). */
int hp_lct_backward(const hp_lct_plan* plan, const float* gy, float* gx, int batch,
                    void* workspace, size_t workspace_bytes, void* stream);
/* Test hook: copy the device-resident inverse PSF back in natural (2T,2N,2N) order. */
int hp_lct_plan_get_invpsf(const hp_lct_plan* plan, float* invpsf_re, float* invpsf_im);

#ifdef __cplusplus
}
#endif
#endif /* HIDDENPOSE_HIP_H */
