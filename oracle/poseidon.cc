// ORACLE — TEST INFRASTRUCTURE ONLY (see poseidon.h).
#include "poseidon.h"
#include "poseidon_fast_constants.h"
#include <mutex>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace orc {

// ---- rand_chacha 0.3.1 (Cargo.lock:644-647) ChaCha8Rng::seed_from_u64(0) + rand 0.8 gen_range(0..p),
// the recipe recorded in SURVEY.md App. A.2.
namespace {
struct ChaCha8 {
  uint32_t key[8]; uint64_t counter = 0; uint32_t buf[16]; int pos = 16;
  static inline uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
  static inline void qr(uint32_t* s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] ^= s[a]; s[d] = rotl(s[d], 16);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = rotl(s[b], 12);
    s[a] += s[b]; s[d] ^= s[a]; s[d] = rotl(s[d], 8);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = rotl(s[b], 7);
  }
  void seed_from_u64(uint64_t state) {
    // rand_core's default seed_from_u64: PCG32 stream fills the 32-byte seed.
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    for (int i = 0; i < 8; i++) {
      state = state * MUL + INC;
      uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
      uint32_t rot = (uint32_t)(state >> 59);
      key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
  }
  void block() {
    uint32_t in[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) in[4 + i] = key[i];
    in[12] = (uint32_t)counter; in[13] = (uint32_t)(counter >> 32); in[14] = 0; in[15] = 0;
    uint32_t s[16]; memcpy(s, in, sizeof s);
    for (int r = 0; r < 4; r++) {
      qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15);
      qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) buf[i] = s[i] + in[i];
    counter++; pos = 0;
  }
  uint32_t next_u32() { if (pos >= 16) block(); return buf[pos++]; }
  uint64_t next_u64() { uint64_t lo = next_u32(); uint64_t hi = next_u32(); return lo | (hi << 32); }
  uint64_t gen_range_p() {  // rand 0.8 UniformInt::sample_single for 0..p (widening multiply + zone reject)
    const uint64_t zone = P - 1;  // (p << lz(p)) - 1 with lz = 0
    for (;;) { u128 m = (u128)next_u64() * P; if ((u64)m <= zone) return (u64)(m >> 64); }
  }
};
u64 RC[360];
std::once_flag rc_once;
}

const u64* poseidon_round_constants() {
  std::call_once(rc_once, [] {
    ChaCha8 rng; rng.seed_from_u64(0);
    for (int i = 0; i < 360; i++) RC[i] = rng.gen_range_p();
  });
  return RC;
}

// Same round function as the generic template (add constants -> S-box -> MDS), written for throughput because this file
// is also bench.py's CPU baseline: state elements are lazy residues (any u64 congruent mod p), the MDS row sums are taken
// on the 32-bit halves (AVX2 vpmuludq when available, else scalar) with the next round's constants folded in, and there
// is one 128->64 reduction per product / per row. Tests pin it against the literal form below and the reference KATs.
namespace {
inline u64 red128_lazy(u64 lo, u64 hi) {
  u64 hh = hi >> 32, hl = hi & EPS;
  u64 t0 = lo - hh; if (__builtin_expect(lo < hh, 0)) t0 -= EPS;
  u64 t1 = hl * EPS, t2 = t0 + t1;
  t2 += (0 - (u64)(t2 < t1)) & EPS;
  return t2;
}
inline u64 mul_lazy(u64 a, u64 b) { u128 p = (u128)a * b; return red128_lazy((u64)p, (u64)(p >> 64)); }
inline u64 sbox_lazy(u64 x) { u64 x2 = mul_lazy(x, x), x4 = mul_lazy(x2, x2), x3 = mul_lazy(x2, x); return mul_lazy(x3, x4); }
inline void mds_epilogue(u64 st[12], const u64 sl[12], const u64 sh[12], const u64* rc) {
  for (int r = 0; r < 12; r++) {
    u64 c = rc ? rc[r] : 0;
    u64 a = sl[r] + (c & EPS), b = sh[r] + (c >> 32);
    u64 t = a + (b >> 32) * EPS, add = b << 32, v = t + add;
    v += (0 - (u64)(v < add)) & EPS;
    st[r] = v;
  }
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) void mds_rc_avx2(u64 st[12], const u64* rc) {
  alignas(32) u64 lo2[24], hi2[24], sl[12], sh[12];
  for (int i = 0; i < 12; i++) { lo2[i] = lo2[i + 12] = st[i] & EPS; hi2[i] = hi2[i + 12] = st[i] >> 32; }
  __m256i l0 = _mm256_setzero_si256(), l1 = l0, l2 = l0, h0 = l0, h1 = l0, h2 = l0;
  for (int i = 0; i < 12; i++) {
    const __m256i c = _mm256_set1_epi64x((long long)MDS_CIRC[i]);
    l0 = _mm256_add_epi64(l0, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(lo2 + i)), c));
    l1 = _mm256_add_epi64(l1, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(lo2 + i + 4)), c));
    l2 = _mm256_add_epi64(l2, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(lo2 + i + 8)), c));
    h0 = _mm256_add_epi64(h0, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(hi2 + i)), c));
    h1 = _mm256_add_epi64(h1, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(hi2 + i + 4)), c));
    h2 = _mm256_add_epi64(h2, _mm256_mul_epu32(_mm256_loadu_si256((const __m256i*)(hi2 + i + 8)), c));
  }
  _mm256_store_si256((__m256i*)sl, l0); _mm256_store_si256((__m256i*)(sl + 4), l1); _mm256_store_si256((__m256i*)(sl + 8), l2);
  _mm256_store_si256((__m256i*)sh, h0); _mm256_store_si256((__m256i*)(sh + 4), h1); _mm256_store_si256((__m256i*)(sh + 8), h2);
  sl[0] += lo2[0] * MDS_DIAG[0]; sh[0] += hi2[0] * MDS_DIAG[0];
  mds_epilogue(st, sl, sh, rc);
}
#endif
void mds_rc_scalar(u64 st[12], const u64* rc) {
  u64 lo[12], hi[12], sl[12], sh[12];
  for (int i = 0; i < 12; i++) { lo[i] = st[i] & EPS; hi[i] = st[i] >> 32; }
  for (int r = 0; r < 12; r++) {
    u64 a = lo[r] * MDS_DIAG[r], b = hi[r] * MDS_DIAG[r];
    for (int i = 0; i < 12; i++) { a += lo[(i + r) % 12] * MDS_CIRC[i]; b += hi[(i + r) % 12] * MDS_CIRC[i]; }
    sl[r] = a; sh[r] = b;
  }
  mds_epilogue(st, sl, sh, rc);
}
inline void mds_rc(u64 st[12], const u64* rc) {
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) { mds_rc_avx2(st, rc); return; }
#endif
  mds_rc_scalar(st, rc);
}
}
// Fast form of the 22 partial rounds (upstream uses an algebraically equivalent one): tables derived and checked
// by tools/gen_poseidon_fast.py; each round is one S-box, one 11-term dot product and 11 multiply-adds instead of a 12x12 MDS product.
// s += FIRST; s[1:] = INIT*s[1:]; per round s0 = sbox(s0) + Q[t]; (s0, rest) = (25*s0 + WHAT[t].rest, rest + s0*V[t]).
namespace {
const u64 FAST_FIRST[12] = {QP_POSEIDON_FAST_FIRST};
const u64 FAST_Q[22] = {QP_POSEIDON_FAST_Q};
const u64 FAST_INIT[11][11] = {QP_POSEIDON_FAST_INIT};
const u64 FAST_WHAT[22][11] = {QP_POSEIDON_FAST_WHAT};
const u64 FAST_V[22][11] = {QP_POSEIDON_FAST_V};
inline u64 addc_lazy(u64 x, u64 c) { u64 s = x + c; s += (0 - (u64)(s < c)) & EPS; return s; }
// sum_j a[j]*x[j] + c0*x0: product halves accumulate separately, then lo + 2^64 mid + 2^128 top with 2^128 == -2^32 (mod p)
inline u64 dot11(const u64* a, const u64* x, u64 c0, u64 x0) {
  u128 p = (u128)c0 * x0;
  u128 alo = (u64)p, ahi = (u64)(p >> 64);
  for (int j = 0; j < 11; j++) { p = (u128)a[j] * x[j]; alo += (u64)p; ahi += (u64)(p >> 64); }
  const u128 mid = (alo >> 64) + (u64)ahi;
  const u64 top = (u64)(ahi >> 64) + (u64)(mid >> 64);
  u64 r = red128_lazy((u64)alo, (u64)mid);
  const u64 sub = top << 32;
  u64 d = r - sub; if (__builtin_expect(r < sub, 0)) d += P;
  return d;
}
}
// The oracle's PRIMARY permutation runs the literal rounds of SURVEY.md App. A.2 — add constants, S-box (all lanes in full rounds, lane 0 in
// partial rounds), the MDS product — from the primary constants alone (round constants + MDS). The derived fast form below shares its
// tables with the product (one generator, tools/gen_poseidon_fast.py): a checker must not lean on what it checks, so the fast form is
// opt-in (poseidon_set_fast: bench.py's cpu_baseline timing leg, where the CPU side should be as fast as it can) and tests compare the two.
static bool g_poseidon_fast = false;
void poseidon_set_fast(bool on) { g_poseidon_fast = on; }
bool poseidon_is_fast() { return g_poseidon_fast; }
static void poseidon_permute_fast(u64 st[12]);
void poseidon_permute(u64 st[12]) {
  if (g_poseidon_fast) { poseidon_permute_fast(st); return; }
  const u64* rc = poseidon_round_constants();
  for (int i = 0; i < 12; i++) st[i] = addc_lazy(st[i], rc[i]);
  for (int r = 0; r < N_ROUNDS; r++) {
    const bool full = r < N_FULL_HALF || r >= N_FULL_HALF + N_PARTIAL;
    if (full) for (int i = 0; i < 12; i++) st[i] = sbox_lazy(st[i]);
    else st[0] = sbox_lazy(st[0]);
    mds_rc(st, r + 1 < N_ROUNDS ? rc + 12 * (r + 1) : nullptr);          // the MDS product, then the next round's constants
  }
  for (int i = 0; i < 12; i++) if (st[i] >= P) st[i] -= P;
}
static void poseidon_permute_fast(u64 st[12]) {
  const u64* rc = poseidon_round_constants();
  for (int i = 0; i < 12; i++) st[i] = addc_lazy(st[i], rc[i]);
  for (int r = 0; r < N_FULL_HALF; r++) {
    for (int i = 0; i < 12; i++) st[i] = sbox_lazy(st[i]);
    mds_rc(st, r + 1 < N_FULL_HALF ? rc + 12 * (r + 1) : FAST_FIRST);
  }
  { u64 t[11]; for (int i = 0; i < 11; i++) t[i] = dot11(FAST_INIT[i], st + 1, 0, 0); for (int i = 0; i < 11; i++) st[1 + i] = t[i]; }
  for (int t = 0; t < N_PARTIAL; t++) {
    const u64 s0 = addc_lazy(sbox_lazy(st[0]), FAST_Q[t]);
    const u64 d = dot11(FAST_WHAT[t], st + 1, 25, s0);
    for (int j = 0; j < 11; j++) { const u128 p = (u128)s0 * FAST_V[t][j] + st[1 + j]; st[1 + j] = red128_lazy((u64)p, (u64)(p >> 64)); }
    st[0] = d;
  }
  const int r0 = N_FULL_HALF + N_PARTIAL;
  for (int i = 0; i < 12; i++) st[i] = addc_lazy(st[i], rc[12 * r0 + i]);
  for (int r = r0; r < N_ROUNDS; r++) {
    for (int i = 0; i < 12; i++) st[i] = sbox_lazy(st[i]);
    mds_rc(st, r + 1 < N_ROUNDS ? rc + 12 * (r + 1) : nullptr);
  }
  for (int i = 0; i < 12; i++) if (st[i] >= P) st[i] -= P;
}

// Literal (slow) form through the field-generic template; tests compare it with poseidon_permute.
void poseidon_permute_generic(u64 st[12]) {
  const u64* rc = poseidon_round_constants();
  FB s[12];
  for (int i = 0; i < 12; i++) s[i] = FB(st[i]);
  for (int r = 0; r < N_ROUNDS; r++) {
    for (int i = 0; i < 12; i++) s[i] = s[i] + FB(rc[12 * r + i]);
    bool full = r < N_FULL_HALF || r >= N_FULL_HALF + N_PARTIAL;
    if (full) for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
    else s[0] = sbox7(s[0]);
    mds_layer_t(s);
  }
  for (int i = 0; i < 12; i++) st[i] = s[i].v;
}

Hash hash_no_pad(const u64* x, size_t len) {
  u64 st[12] = {0};
  for (size_t off = 0; off < len; off += SPONGE_RATE) {
    size_t m = len - off < (size_t)SPONGE_RATE ? len - off : SPONGE_RATE;
    for (size_t i = 0; i < m; i++) st[i] = x[off + i];   // overwrite mode
    poseidon_permute(st);
  }
  return Hash{st[0], st[1], st[2], st[3]};
}

Hash hash_or_noop(const u64* x, size_t len) {
  if (len <= 4) { Hash h{0, 0, 0, 0}; for (size_t i = 0; i < len; i++) h[i] = x[i]; return h; }
  return hash_no_pad(x, len);
}

Hash two_to_one(const Hash& l, const Hash& r) {
  u64 st[12] = {l[0], l[1], l[2], l[3], r[0], r[1], r[2], r[3], 0, 0, 0, 0};
  poseidon_permute(st);
  return Hash{st[0], st[1], st[2], st[3]};
}

} // namespace orc
