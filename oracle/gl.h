// ORACLE — TEST INFRASTRUCTURE ONLY. Not linked or imported by the product path.
// CPU restatement of qp-plonky2-field 1.1.1 (Cargo.lock:514-517; source NOT under
// /root/reference) Goldilocks arithmetic, following its published algorithm as
// recorded in SURVEY.md App. A.1. Types aliased by the reference at
// common/src/circuit.rs:10-12 (F = GoldilocksField, D = 2).
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>

namespace orc {

typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 P = 0xFFFFFFFF00000001ULL;
static const u64 EPS = 0xFFFFFFFFULL;             // 2^64 mod p
static const u64 GEN = 14293326489335486720ULL;   // multiplicative generator / coset shift (A.1)
static const u64 POW2_GEN = 7277203076849721926ULL; // order 2^32
static const u64 W_EXT = 7;                       // F_p[X]/(X^2-7)

static inline u64 gl_add(u64 a, u64 b) { u64 s = a + b; if (s < a || s >= P) s -= P; return s; }
static inline u64 gl_sub(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }
static inline u64 gl_neg(u64 a) { return a ? P - a : 0; }
static inline u64 gl_red128_slow(u128 x) { return (u64)(x % P); }
// 2^64 = 2^32-1, 2^96 = -1 (mod p): x = lo + hi_lo*2^64 + hi_hi*2^96.
static inline u64 gl_red128(u128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hi_hi = hi >> 32, hi_lo = hi & EPS;
  u64 t0 = lo - hi_hi; if (lo < hi_hi) t0 -= EPS;         // wrapped by 2^64 => subtract 2^32-1
  u64 t1 = hi_lo * EPS;                                   // < 2^64
  u64 t2 = t0 + t1; if (t2 < t1) t2 += EPS;               // wrapped => add 2^32-1
  if (t2 >= P) t2 -= P;
  return t2;
}
static inline u64 gl_mul(u64 a, u64 b) { return gl_red128((u128)a * b); }
static inline u64 gl_sqr(u64 a) { return gl_mul(a, a); }
static inline u64 gl_pow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = gl_mul(r, a); a = gl_sqr(a); e >>= 1; } return r; }
static inline u64 gl_inv(u64 a) { assert(a != 0); return gl_pow(a, P - 2); }
static inline u64 gl_from(u64 x) { return x % P; }
static inline u64 gl_root(int bits) { assert(bits <= 32); u64 r = POW2_GEN; for (int i = bits; i < 32; i++) r = gl_sqr(r); return r; }

struct Ext { u64 a, b; };
static inline Ext ex(u64 a, u64 b = 0) { return Ext{a, b}; }
static inline bool operator==(Ext x, Ext y) { return x.a == y.a && x.b == y.b; }
static inline bool operator!=(Ext x, Ext y) { return !(x == y); }
static inline Ext operator+(Ext x, Ext y) { return Ext{gl_add(x.a, y.a), gl_add(x.b, y.b)}; }
static inline Ext operator-(Ext x, Ext y) { return Ext{gl_sub(x.a, y.a), gl_sub(x.b, y.b)}; }
static inline Ext operator*(Ext x, Ext y) {
  return Ext{gl_add(gl_mul(x.a, y.a), gl_mul(W_EXT, gl_mul(x.b, y.b))), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))};
}
static inline Ext ex_scale(Ext x, u64 s) { return Ext{gl_mul(x.a, s), gl_mul(x.b, s)}; }
static inline Ext ex_inv(Ext x) {
  // 1/(a+bX) = (a-bX)/(a^2-7b^2)
  u64 nrm = gl_sub(gl_sqr(x.a), gl_mul(W_EXT, gl_sqr(x.b)));
  u64 ni = gl_inv(nrm);
  return Ext{gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni)};
}
static inline Ext ex_pow(Ext x, u64 e) { Ext r = ex(1); while (e) { if (e & 1) r = r * x; x = x * x; e >>= 1; } return r; }

// Field-generic helpers so gate evaluation can be written once for F and F_{p^2}.
struct FB { // base field wrapper
  u64 v;
  FB() : v(0) {}
  FB(u64 x) : v(x) {}
};
static inline FB operator+(FB x, FB y) { return FB(gl_add(x.v, y.v)); }
static inline FB operator-(FB x, FB y) { return FB(gl_sub(x.v, y.v)); }
static inline FB operator*(FB x, FB y) { return FB(gl_mul(x.v, y.v)); }
static inline FB from_base(FB*, u64 c) { return FB(c); }
static inline Ext from_base(Ext*, u64 c) { return ex(c); }

static inline uint32_t rev_bits(uint32_t x, int bits) {
  uint32_t r = 0; for (int i = 0; i < bits; i++) { r = (r << 1) | ((x >> i) & 1); } return r;
}
static inline int log2_exact(size_t n) { int k = 0; while ((size_t(1) << k) < n) k++; assert((size_t(1) << k) == n); return k; }

} // namespace orc
