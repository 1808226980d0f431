// Host-side constants of the light-cone transform (LCT) plan.
#pragma once
#include <complex>
#include <cstdint>
#include <vector>

namespace hp {

// Row-compressed sparse matrix (the t -> sqrt(t) resampling operator is a narrow band).
struct SparseRows {
  int rows = 0;
  std::vector<int32_t> off;  // rows+1
  std::vector<int32_t> idx;  // column of each entry
  std::vector<float> val;
  SparseRows transposed(int cols) const;
};

struct LctHost {
  int T = 0, N = 0;
  double slope = 0;
  std::vector<float> gridz;       // T          gridz = t/(T-1)
  SparseRows mtx;                 // T x T      resampling operator
  // PSF indicator: for every rolled (x,y) column the z samples that are 1
  std::vector<int32_t> mark_off;  // 4N^2+1
  std::vector<int32_t> mark_z;
  int64_t count = 0;              // number of ones
  float psf_val = 0;              // 1/sqrt(count)
};

// models/feature_propagation.py:71-171 (constants only; no FFT yet)
void lct_host_build(int T, int N, double bin_len, double wall_size, LctHost& out);

// One kz slice (2N x 2N, natural frequency order, row-major [kx][ky]) of
// invpsf = conj(F)/(1/snr + |F|^2) (mode 'lct') or conj(F) (mode 'bp', feature_propagation.py:93-94),
// F = fftn(psf) in double precision.  `work` is unused (kept for callers that size a scratch buffer).
void lct_invpsf_slice(const LctHost& h, int kz, std::complex<double>* out, std::complex<double>* work, bool wiener = true);

}  // namespace hp
