// ORACLE — TEST INFRASTRUCTURE ONLY. Radix-2 NTT over Goldilocks (qp-plonky2-field 1.1.1 fft/ifft/coset_fft
// semantics: natural-order in, natural-order out; values[i] = poly(shift * w^i)). SURVEY.md App. A.1, §8(a) a3/a4.
#include "plonk.h"
#include <map>
#include <mutex>
#include <omp.h>

namespace orc {

static const std::vector<u64>& twiddles(int lg, bool inverse) {
  static std::map<int, std::vector<u64>> cache; static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  int key = lg * 2 + (inverse ? 1 : 0);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  size_t n = size_t(1) << lg; u64 w = gl_root(lg); if (inverse) w = gl_inv(w);
  std::vector<u64> t(n / 2 ? n / 2 : 1); u64 x = 1;
  for (size_t i = 0; i < n / 2; i++) { t[i] = x; x = gl_mul(x, w); }
  return cache[key] = std::move(t);
}

void ntt(std::vector<u64>& a, bool inverse) {
  size_t n = a.size(); if (n <= 1) return;
  int lg = log2_exact(n);
  const std::vector<u64>& tw = twiddles(lg, inverse);
  for (size_t i = 0; i < n; i++) { size_t j = rev_bits((uint32_t)i, lg); if (i < j) std::swap(a[i], a[j]); }
  const bool par = n >= (size_t(1) << 15) && !omp_in_parallel();   // the few big single-column transforms (quotient, FRI)
  u64* A = a.data(); const u64* TW = tw.data();
  for (int s = 1; s <= lg; s++) {
    size_t m = size_t(1) << s, half = m >> 1, step = n / m;
#pragma omp parallel for schedule(static) if (par)
    for (size_t q = 0; q < n / 2; q++) {
      size_t k = (q / half) * m, j = q % half;
      u64 t = gl_mul(TW[j * step], A[k + j + half]), u = A[k + j];
      A[k + j] = gl_add(u, t); A[k + j + half] = gl_sub(u, t);
    }
  }
  if (inverse) { u64 ni = gl_inv((u64)n); for (auto& x : a) x = gl_mul(x, ni); }
}

std::vector<u64> coset_lde(const std::vector<u64>& coeffs, int rate_bits, u64 shift) {
  size_t n = coeffs.size(), N = n << rate_bits;
  std::vector<u64> v(N, 0); u64 s = 1;
  for (size_t i = 0; i < n; i++) { v[i] = gl_mul(coeffs[i], s); s = gl_mul(s, shift); }
  ntt(v, false);
  return v;
}

} // namespace orc
