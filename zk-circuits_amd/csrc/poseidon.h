// Poseidon-Goldilocks permutation / sponge (width 12, rate 8, x^7, 4+22+4 rounds) for host and gfx950 device.
// Semantics of qp-plonky2 1.1.1's PoseidonHash (hash_no_pad in overwrite mode, hash_or_noop, two_to_one) used by
// the reference's circuits, e.g. wormhole/circuit/src/nullifier.rs:53-73, unspendable_account.rs:38-63
// (SURVEY.md App. A.2). Round constants: poseidon_constants.h (generated).
#pragma once
#include "gl.h"
#include "poseidon_constants.h"
#include "poseidon_fast_constants.h"

namespace qp {

#if defined(__HIPCC__)
__device__ __constant__ const u64 POSEIDON_RC_DEV[360] = {QP_POSEIDON_RC_LIST};
#endif
static const u64 POSEIDON_RC_HOST[360] = {QP_POSEIDON_RC_LIST};

QP_HD const u64* poseidon_rc() {
#if defined(__HIP_DEVICE_COMPILE__)
  return POSEIDON_RC_DEV;
#else
  return POSEIDON_RC_HOST;
#endif
}

QP_HD u64 sbox7(u64 x) { u64 x2 = gl_sqr(x), x4 = gl_sqr(x2), x3 = gl_mul(x, x2); return gl_mul(x3, x4); }

// out[r] = sum_i st[(i+r)%12]*CIRC[i] + st[r]*DIAG[r], CIRC = {17,15,41,16,2,28,13,13,39,18,34,20}, DIAG = {8,0,...}.
// The 32-bit halves are accumulated separately (sums < 2^41) and recombined into one 128-bit reduction.
QP_HD void mds_layer(u64 st[12]) {
#if !defined(__HIP_DEVICE_COMPILE__)
  // host (transcript, witness synthesis): 32-bit halves accumulated in u64 lanes (vectorisable), one reduction per row
  static const u64 CC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u64 lo2[24], hi2[24];
  for (int i = 0; i < 12; i++) { lo2[i] = lo2[i + 12] = st[i] & GL_EPS; hi2[i] = hi2[i + 12] = st[i] >> 32; }
  u64 sl[12], sh[12];
  for (int r = 0; r < 12; r++) { sl[r] = 0; sh[r] = 0; }
  for (int i = 0; i < 12; i++)
    for (int r = 0; r < 12; r++) { sl[r] += lo2[r + i] * CC[i]; sh[r] += hi2[r + i] * CC[i]; }
  sl[0] += lo2[0] * 8; sh[0] += hi2[0] * 8;
  for (int r = 0; r < 12; r++) {
    u64 l = sl[r] + (sh[r] << 32);
    u64 h = (sh[r] >> 32) + (l < sl[r] ? 1 : 0);
    st[r] = gl_reduce128(l, h);
  }
  return;
#endif
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u64 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { lo[i] = st[i] & GL_EPS; hi[i] = st[i] >> 32; }
#pragma unroll
  for (int r = 0; r < 12; r++) {
    u64 sl = 0, sh = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) { sl += lo[(i + r) % 12] * C[i]; sh += hi[(i + r) % 12] * C[i]; }
    if (r == 0) { sl += lo[0] * 8; sh += hi[0] * 8; }
    u64 l = sl + (sh << 32);
    u64 h = (sh >> 32) + (l < sl ? 1 : 0);
    st[r] = gl_reduce128(l, h);
  }
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host fast path (transcript: ~120 sequential permutations sit on every proof's critical path).
// Lazy residues (any u64 congruent mod p), one 128->64 reduction per product / per MDS row, constants of the next round
// folded into the MDS sums; canonicalised at the end. Same function as the generic code below (tests compare).
#if defined(__clang__)
#define QP_UNROLL_ALL _Pragma("unroll")
#else
#define QP_UNROLL_ALL _Pragma("GCC unroll 16")
#endif
namespace hostp {
static inline u64 red128(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi;
  if (__builtin_expect(lo < hi_hi, 0)) t0 -= GL_EPS;          // probability ~2^-32 for random data
  u64 t1 = hi_lo * GL_EPS;
  u64 t2 = t0 + t1;
  t2 += (0 - (u64)(t2 < t1)) & GL_EPS;                        // ~50/50: keep it branch-free
  return t2;
}
static inline u64 mul(u64 a, u64 b) { unsigned __int128 p = (unsigned __int128)a * b; return red128((u64)p, (u64)(p >> 64)); }
static inline u64 sbox(u64 x) { u64 x2 = mul(x, x), x4 = mul(x2, x2), x3 = mul(x2, x); return mul(x3, x4); }
static inline u64 addc(u64 x, u64 c) { u64 s = x + c; s += (0 - (u64)(s < c)) & GL_EPS; return s; }   // lazy x + canonical c
#if defined(__x86_64__)
} }  // close hostp, qp for the intrinsics header
#include <immintrin.h>
namespace qp { namespace hostp {
// AVX2: the 12 row sums of each 32-bit half are three 4-lane vectors; vpmuludq does four 32x32->64 products at a time.
__attribute__((target("avx2"))) static inline void mds_rc_avx2(u64 st[12], const u64* rc) {
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  alignas(32) u64 lo2[24], hi2[24], sl[12], sh[12];
  for (int i = 0; i < 12; i++) { lo2[i] = lo2[i + 12] = st[i] & GL_EPS; hi2[i] = hi2[i + 12] = st[i] >> 32; }
  __m256i l0 = _mm256_setzero_si256(), l1 = l0, l2 = l0, h0 = l0, h1 = l0, h2 = l0;
  for (int i = 0; i < 12; i++) {
    const __m256i c = _mm256_set1_epi64x((long long)C[i]);
    l0 = _mm256_add_epi64(l0, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(lo2 + i)), c));
    l1 = _mm256_add_epi64(l1, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(lo2 + i + 4)), c));
    l2 = _mm256_add_epi64(l2, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(lo2 + i + 8)), c));
    h0 = _mm256_add_epi64(h0, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(hi2 + i)), c));
    h1 = _mm256_add_epi64(h1, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(hi2 + i + 4)), c));
    h2 = _mm256_add_epi64(h2, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(hi2 + i + 8)), c));
  }
  _mm256_store_si256((__m256i*)sl, l0); _mm256_store_si256((__m256i*)(sl + 4), l1); _mm256_store_si256((__m256i*)(sl + 8), l2);
  _mm256_store_si256((__m256i*)sh, h0); _mm256_store_si256((__m256i*)(sh + 4), h1); _mm256_store_si256((__m256i*)(sh + 8), h2);
  sl[0] += lo2[0] * 8; sh[0] += hi2[0] * 8;
  for (int r = 0; r < 12; r++) {
    u64 c = rc ? rc[r] : 0;
    u64 a = sl[r] + (c & GL_EPS), b = sh[r] + (c >> 32);
    u64 t = a + (b >> 32) * GL_EPS;
    u64 add = b << 32;
    u64 v = t + add;
    v += (0 - (u64)(v < add)) & GL_EPS;
    st[r] = v;
  }
}
static inline bool have_avx2() { static const bool ok = __builtin_cpu_supports("avx2"); return ok; }
#endif
static inline void mds_rc(u64 st[12], const u64* rc) {
#if defined(__x86_64__)
  if (have_avx2()) { mds_rc_avx2(st, rc); return; }
#endif
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u64 lo[12], hi[12], out[12];
  QP_UNROLL_ALL for (int i = 0; i < 12; i++) { lo[i] = st[i] & GL_EPS; hi[i] = st[i] >> 32; }
  QP_UNROLL_ALL for (int r = 0; r < 12; r++) {
    u64 c = rc ? rc[r] : 0;
    u64 sl = c & GL_EPS, sh = c >> 32;
    QP_UNROLL_ALL for (int i = 0; i < 12; i++) { sl += lo[(i + r) % 12] * C[i]; sh += hi[(i + r) % 12] * C[i]; }
    if (r == 0) { sl += lo[0] * 8; sh += hi[0] * 8; }
    // value = sl + 2^32 sh, both < 2^43
    u64 t = sl + (sh >> 32) * GL_EPS;                         // no overflow
    u64 add = sh << 32;
    u64 v = t + add;
    v += (0 - (u64)(v < add)) & GL_EPS;
    out[r] = v;
  }
  QP_UNROLL_ALL for (int r = 0; r < 12; r++) st[r] = out[r];
}
}
// ---- fast form of the 22 partial rounds (tools/gen_poseidon_fast.py derives and checks the tables): each round is one S-box,
// one 11-term dot product and 11 multiply-adds instead of a full 12x12 MDS product. Only the host uses it: on the GPU the
// small-constant MDS needs no per-term reductions and is cheaper than 22 full multiplications.
namespace hostp {
static const u64 FAST_FIRST[12] = {QP_POSEIDON_FAST_FIRST};
static const u64 FAST_Q[22] = {QP_POSEIDON_FAST_Q};
static const u64 FAST_INIT[11][11] = {QP_POSEIDON_FAST_INIT};
static const u64 FAST_WHAT[22][11] = {QP_POSEIDON_FAST_WHAT};
static const u64 FAST_V[22][11] = {QP_POSEIDON_FAST_V};
typedef unsigned __int128 u128;
// sum_j a[j]*x[j] (+ c0*x0) for 11 terms of full 64-bit factors: low and high product halves accumulate separately (no overflow for
// < 2^64 terms), then lo + 2^64*mid + 2^128*top with 2^128 == -2^32 (mod p). Result lazy.
static inline u64 dot11(const u64* a, const u64* x, u64 c0, u64 x0) {
  u128 p = (u128)c0 * x0;
  u128 alo = (u64)p, ahi = (u64)(p >> 64);
  QP_UNROLL_ALL for (int j = 0; j < 11; j++) { p = (u128)a[j] * x[j]; alo += (u64)p; ahi += (u64)(p >> 64); }
  const u128 mid = (alo >> 64) + (u64)ahi;
  const u64 top = (u64)(ahi >> 64) + (u64)(mid >> 64);            // < 32
  u64 r = red128((u64)alo, (u64)mid);
  const u64 sub = top << 32;
  u64 d = r - sub; if (__builtin_expect(r < sub, 0)) d += GL_P;
  return d;
}
}
// TRACE: also record the S-box inputs a PoseidonGate row carries (canonical): trace[0..36) = full rounds 1-3, [36..58) = the 22 partial
// rounds' state[0], [58..106) = full rounds 26-29 — wire columns 29..134 in order. The partial-round S-box input is the same element in
// the fast and the literal form (only the other eleven lanes live in another basis).
template <bool TRACE>
inline void poseidon_permute_impl(u64 st[12], u64* trace) {
  const u64* rc = POSEIDON_RC_HOST;
  auto canon = [](u64 x) { return x - ((0 - (u64)(x >= GL_P)) & GL_P); };
  QP_UNROLL_ALL for (int i = 0; i < 12; i++) st[i] = hostp::addc(st[i], rc[i]);
  int r = 0;
  for (; r < 4; r++) {
    QP_UNROLL_ALL for (int i = 0; i < 12; i++) st[i] = hostp::sbox(st[i]);
    hostp::mds_rc(st, r < 3 ? rc + 12 * (r + 1) : hostp::FAST_FIRST);
    if (TRACE && r < 3) for (int i = 0; i < 12; i++) trace[12 * r + i] = canon(st[i]);
  }
  { u64 t[11];
    for (int i = 0; i < 11; i++) t[i] = hostp::dot11(hostp::FAST_INIT[i], st + 1, 0, 0);
    for (int i = 0; i < 11; i++) st[1 + i] = t[i]; }
  for (int t = 0; t < 22; t++) {
    if (TRACE) trace[36 + t] = canon(st[0]);
    const u64 s0 = hostp::addc(hostp::sbox(st[0]), hostp::FAST_Q[t]);
    const u64 d = hostp::dot11(hostp::FAST_WHAT[t], st + 1, 25, s0);
    QP_UNROLL_ALL for (int j = 0; j < 11; j++) {
      const hostp::u128 p = (hostp::u128)s0 * hostp::FAST_V[t][j] + st[1 + j];
      st[1 + j] = hostp::red128((u64)p, (u64)(p >> 64));
    }
    st[0] = d;
  }
  QP_UNROLL_ALL for (int i = 0; i < 12; i++) st[i] = hostp::addc(st[i], rc[12 * 26 + i]);
  for (r = 26; r < 29; r++) {
    if (TRACE) for (int i = 0; i < 12; i++) trace[58 + 12 * (r - 26) + i] = canon(st[i]);
    QP_UNROLL_ALL for (int i = 0; i < 12; i++) st[i] = hostp::sbox(st[i]);
    hostp::mds_rc(st, rc + 12 * (r + 1));
  }
  if (TRACE) for (int i = 0; i < 12; i++) trace[58 + 36 + i] = canon(st[i]);
  QP_UNROLL_ALL for (int i = 0; i < 12; i++) st[i] = hostp::sbox(st[i]);
  hostp::mds_rc(st, nullptr);
  QP_UNROLL_ALL for (int i = 0; i < 12; i++) st[i] -= (0 - (u64)(st[i] >= GL_P)) & GL_P;
}
inline void poseidon_permute(u64 st[12]) { poseidon_permute_impl<false>(st, nullptr); }
inline void poseidon_permute_traced(u64 st[12], u64 trace[106]) { poseidon_permute_impl<true>(st, trace); }
// literal round-by-round form (used by the witness synthesiser's traces and by tests)
inline void poseidon_permute_generic(u64 st[12]) {
#else
QP_HD void poseidon_permute(u64 st[12]) {
#endif
  const u64* rc = poseidon_rc();
  int r = 0;
  for (int k = 0; k < 4; k++, r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = sbox7(gl_add(st[i], rc[12 * r + i]));
    mds_layer(st);
  }
  for (int k = 0; k < 22; k++, r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], rc[12 * r + i]);
    st[0] = sbox7(st[0]);
    mds_layer(st);
  }
  for (int k = 0; k < 4; k++, r++) {
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = sbox7(gl_add(st[i], rc[12 * r + i]));
    mds_layer(st);
  }
}

// Host-side sponge helpers (device kernels stream their inputs themselves).
QP_HD void hash_no_pad(const u64* x, size_t len, u64 out[4]) {
  u64 st[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t off = 0; off < len; off += 8) {
    size_t m = len - off < 8 ? len - off : 8;
    for (size_t i = 0; i < m; i++) st[i] = x[off + i];
    poseidon_permute(st);
  }
  for (int i = 0; i < 4; i++) out[i] = st[i];
}

} // namespace qp
