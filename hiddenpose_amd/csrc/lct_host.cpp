// Host constants of the LCT plan: resampling band matrix, PSF indicator and the
// Wiener-inverse PSF spectrum.  Behaviour follows models/feature_propagation.py
// :71-171 of the reference; float32 steps are kept in the reference's operation
// order (no FMA contraction) because the PSF is a float32 tie test.
#include "lct_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>

#include "hp_host.h"

#pragma STDC FP_CONTRACT OFF

namespace hp {

SparseRows SparseRows::transposed(int cols) const {
  SparseRows t;
  t.rows = cols;
  std::vector<int32_t> cnt(cols + 1, 0);
  for (int32_t c : idx) cnt[c + 1]++;
  for (int c = 0; c < cols; ++c) cnt[c + 1] += cnt[c];
  t.off = cnt;
  t.idx.resize(idx.size());
  t.val.resize(idx.size());
  std::vector<int32_t> cur(cnt.begin(), cnt.end() - 1);
  for (int r = 0; r < rows; ++r)
    for (int e = off[r]; e < off[r + 1]; ++e) {
      int p = cur[idx[e]]++;
      t.idx[p] = r;
      t.val[p] = val[e];
    }
  return t;
}

// _resamplingOperator (:111-139): rows r of an (M^2 x M) operator hold 1/sqrt(r+1) in
// column ceil(sqrt(r+1))-1; log2(M) pairwise row averagings (float32) leave M rows.
static void build_resampler(int M, SparseRows& out) {
  int K = 0;
  while ((1 << K) < M) ++K;
  out.rows = M;
  out.off.assign(1, 0);
  out.idx.clear();
  out.val.clear();
  std::vector<float> blk, nxt;
  std::vector<int> col(M);
  std::vector<float> v(M);
  for (int i = 0; i < M; ++i) {
    int c0 = M, c1 = -1;
    for (int r = 0; r < M; ++r) {
      float x = (float)((int64_t)i * M + r) + 1.0f;
      float s = sqrtf(x);
      col[r] = (int)(ceilf(s) - 1.0f);
      v[r] = 1.0f / s;
      c0 = std::min(c0, col[r]);
      c1 = std::max(c1, col[r]);
    }
    int w = c1 - c0 + 1;
    blk.assign((size_t)M * w, 0.0f);
    for (int r = 0; r < M; ++r) blk[(size_t)r * w + (col[r] - c0)] = v[r];
    int rows = M;
    for (int k = 0; k < K; ++k) {
      nxt.assign((size_t)(rows / 2) * w, 0.0f);
      for (int r = 0; r < rows / 2; ++r)
        for (int c = 0; c < w; ++c) {
          float s = blk[(size_t)(2 * r) * w + c] + blk[(size_t)(2 * r + 1) * w + c];
          nxt[(size_t)r * w + c] = 0.5f * s;
        }
      blk.swap(nxt);
      rows /= 2;
    }
    for (int c = 0; c < w; ++c)
      if (blk[c] != 0.0f) {
        out.idx.push_back(c0 + c);
        out.val.push_back(blk[c]);
      }
    out.off.push_back((int32_t)out.idx.size());
  }
}

void lct_host_build(int T, int N, double bin_len, double wall_size, LctHost& o) {
  o.T = T;
  o.N = N;
  const double c = 3e8;
  const double width = wall_size / 2.0;
  const double bin_resolution = bin_len / c;
  const double trange = T * c * bin_resolution;
  o.slope = width / trange;

  o.gridz.resize(T);
  for (int t = 0; t < T; ++t) o.gridz[t] = (float)t / (float)(T - 1);  // :82-83

  build_resampler(T, o.mtx);

  // _definePsf (:141-171)
  const int N2 = 2 * N, M2 = 2 * T;
  std::vector<float> x(N2), x2(N2), z(M2);
  for (int i = 0; i < N2; ++i) {
    float a = (float)i / (float)(N2 - 1);
    a = a * 2.0f;
    x[i] = a - 1.0f;
    x2[i] = x[i] * x[i];
  }
  for (int k = 0; k < M2; ++k) {
    float a = (float)k / (float)(M2 - 1);
    z[k] = a * 2.0f;
  }
  const float s = (float)((4.0 * o.slope) * (4.0 * o.slope));
  const float tol = 1e-8f;
  o.mark_off.assign((size_t)N2 * N2 + 1, 0);
  std::vector<std::vector<int32_t>> marks((size_t)N2 * N2);
  std::vector<float> b(M2);
  int64_t count = 0;
  for (int i = 0; i < N2; ++i)
    for (int j = 0; j < N2; ++j) {
      float r2 = x2[i] + x2[j];
      float sr = s * r2;
      float mn = INFINITY;
      for (int k = 0; k < M2; ++k) {
        float a = sr - z[k];
        b[k] = fabsf(a);
        mn = std::min(mn, b[k]);
      }
      int ri = (i + N) % N2, rj = (j + N) % N2;  // np.roll by N on both spatial axes
      auto& m = marks[(size_t)ri * N2 + rj];
      for (int k = 0; k < M2; ++k) {
        float d = fabsf(b[k] - mn);
        if (d < tol) {
          m.push_back(k);
          ++count;
        }
      }
    }
  o.count = count;
  o.psf_val = 1.0f / sqrtf((float)count);
  o.mark_z.clear();
  for (size_t p = 0; p < marks.size(); ++p) {
    o.mark_off[p] = (int32_t)o.mark_z.size();
    for (int32_t k : marks[p]) o.mark_z.push_back(k);
  }
  o.mark_off[marks.size()] = (int32_t)o.mark_z.size();
}

// in-place iterative radix-2 FFT (forward, e^{-i...}), n a power of two; tw[k] = e^{-2 pi i k/n}
static void fft1d(std::complex<double>* a, int n, int stride, const std::complex<double>* tw,
                  const int* rev) {
  for (int i = 0; i < n; ++i) {
    int j = rev[i];
    if (i < j) std::swap(a[(size_t)i * stride], a[(size_t)j * stride]);
  }
  for (int len = 2; len <= n; len <<= 1) {
    int half = len >> 1, step = n / len;
    for (int i = 0; i < n; i += len)
      for (int k = 0; k < half; ++k) {
        std::complex<double> u = a[(size_t)(i + k) * stride];
        std::complex<double> w = a[(size_t)(i + k + half) * stride] * tw[(size_t)k * step];
        a[(size_t)(i + k) * stride] = u + w;
        a[(size_t)(i + k + half) * stride] = u - w;
      }
  }
}

void lct_invpsf_slice(const LctHost& h, int kz, std::complex<double>* out, std::complex<double>* work, bool wiener) {
  const int N2 = 2 * h.N, M2 = 2 * h.T;
  const double PI = 3.14159265358979323846;
  // twiddles / bit reversal for the 2N-point transforms (small; rebuilt per call)
  std::vector<std::complex<double>> tw(N2);
  for (int k = 0; k < N2; ++k) tw[k] = std::complex<double>(cos(-2.0 * PI * k / N2), sin(-2.0 * PI * k / N2));
  std::vector<int> rev(N2);
  int lg = 0;
  while ((1 << lg) < N2) ++lg;
  for (int i = 0; i < N2; ++i) {
    int r = 0;
    for (int b = 0; b < lg; ++b)
      if (i & (1 << b)) r |= 1 << (lg - 1 - b);
    rev[i] = r;
  }
  (void)work;
  // z-transform of the indicator: each column is a sum of unit phasors
  const double val = (double)h.psf_val;
  std::vector<std::complex<double>> ph(M2);
  for (int m = 0; m < M2; ++m) {
    double ang = -2.0 * PI * (double)m / (double)M2;
    ph[m] = std::complex<double>(cos(ang), sin(ang));
  }
  for (int p = 0; p < N2 * N2; ++p) {
    std::complex<double> acc(0, 0);
    for (int e = h.mark_off[p]; e < h.mark_off[p + 1]; ++e) acc += ph[((int64_t)kz * h.mark_z[e]) % M2];
    out[p] = acc * val;
  }
  for (int i = 0; i < N2; ++i) fft1d(out + (size_t)i * N2, N2, 1, tw.data(), rev.data());
  for (int j = 0; j < N2; ++j) fft1d(out + j, N2, N2, tw.data(), rev.data());
  const double inv_snr = 1.0 / 1e-1;
  for (int p = 0; p < N2 * N2; ++p) {
    std::complex<double> f = out[p];
    double den = inv_snr + f.real() * f.real() + f.imag() * f.imag();
    out[p] = wiener ? std::conj(f) / den : std::conj(f);
  }
}

}  // namespace hp

using namespace hp;

// The constants as dense arrays for tests and for hosts that want them without a device (HIP-free).
extern "C" int hp_lct_host_constants(int T, int N, double bin_len, double wall_size, float* gridz, float* mtx,
                                     int32_t* psf_zidx, int64_t* psf_count, float* invpsf_re, float* invpsf_im) {
  HP_REQUIRE(is_pow2(T) && T >= 2 && N >= 1, "hp_lct_host_constants: T must be a power of two, N >= 1");
  LctHost h;
  lct_host_build(T, N, bin_len, wall_size, h);
  if (gridz) std::memcpy(gridz, h.gridz.data(), sizeof(float) * T);
  if (mtx) {
    std::memset(mtx, 0, sizeof(float) * (size_t)T * T);
    for (int r = 0; r < T; ++r)
      for (int e = h.mtx.off[r]; e < h.mtx.off[r + 1]; ++e) mtx[(size_t)r * T + h.mtx.idx[e]] = h.mtx.val[e];
  }
  const int N2 = 2 * N, M2 = 2 * T;
  if (psf_zidx)
    for (int p = 0; p < N2 * N2; ++p) psf_zidx[p] = h.mark_off[p + 1] > h.mark_off[p] ? h.mark_z[h.mark_off[p]] : -1;
  if (psf_count) *psf_count = h.count;
  if (invpsf_re && invpsf_im) {
    const size_t sl = (size_t)N2 * N2;
    unsigned nth = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nth; ++t)
      th.emplace_back([&, t]() {
        std::vector<std::complex<double>> buf(sl);
        for (int kz = (int)t; kz < M2; kz += (int)nth) {
          lct_invpsf_slice(h, kz, buf.data(), nullptr);
          for (size_t i = 0; i < sl; ++i) {
            invpsf_re[(size_t)kz * sl + i] = (float)buf[i].real();
            invpsf_im[(size_t)kz * sl + i] = (float)buf[i].imag();
          }
        }
      });
    for (auto& t : th) t.join();
  }
  return HP_OK;
}
