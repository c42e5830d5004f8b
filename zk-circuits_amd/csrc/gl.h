// Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension F_p[X]/(X^2 - 7) for the MI355X prover.
// Compiles for the gfx950 device and for the host (witness/circuit synthesis, transcript checks).
// Arithmetic contract follows qp-plonky2-field 1.1.1's GoldilocksField (Cargo.lock:514-517), reached in the
// reference through the aliases at common/src/circuit.rs:10-12. All values are canonical (< p) at kernel boundaries.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define QP_HD __host__ __device__ __forceinline__
#else
#define QP_HD inline
#endif

namespace qp {

typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u64 GL_P = 0xFFFFFFFF00000001ULL;
static constexpr u64 GL_EPS = 0xFFFFFFFFULL;
static constexpr u64 GL_GEN = 14293326489335486720ULL;     // multiplicative generator = LDE coset shift
static constexpr u64 GL_POW2_GEN = 7277203076849721926ULL;  // generator of the 2^32 subgroup

// Branch-free (select) forms: on the host the inputs are effectively random, so branches would mispredict.
QP_HD u64 gl_add(u64 a, u64 b) {
  u64 s = a + b;
  u64 wrapped = 0 - (u64)(s < a);                 // all ones if the 64-bit add wrapped (2^64 == 2^32 - 1 mod p)
  u64 over = 0 - (u64)(s >= GL_P);                // only meaningful when not wrapped
  return s + (wrapped & GL_EPS) - (~wrapped & over & GL_P);   // canonical for canonical inputs
}
QP_HD u64 gl_sub(u64 a, u64 b) { u64 d = a - b; return d + ((0 - (u64)(a < b)) & GL_P); }
QP_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }

QP_HD void mul64wide(u64 a, u64 b, u64& lo, u64& hi) {
#if defined(__HIP_DEVICE_COMPILE__)
  lo = a * b; hi = (u64)__umul64hi((unsigned long long)a, (unsigned long long)b);
#else
  unsigned __int128 p = (unsigned __int128)a * b; lo = (u64)p; hi = (u64)(p >> 64);
#endif
}
// x = lo + 2^64 * hi  ->  canonical residue.  2^64 == 2^32 - 1, 2^96 == -1 (mod p).
QP_HD u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  u64 t0 = lo - hi_hi; t0 -= (0 - (u64)(lo < hi_hi)) & GL_EPS;
  u64 t1 = (hi_lo << 32) - hi_lo;
  u64 t2 = t0 + t1; t2 += (0 - (u64)(t2 < t1)) & GL_EPS;
  t2 -= (0 - (u64)(t2 >= GL_P)) & GL_P;
  return t2;
}
QP_HD u64 gl_mul(u64 a, u64 b) { u64 lo, hi; mul64wide(a, b, lo, hi); return gl_reduce128(lo, hi); }
QP_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }
QP_HD u64 gl_pow(u64 a, u64 e) { u64 r = 1; while (e) { if (e & 1) r = gl_mul(r, a); a = gl_sqr(a); e >>= 1; } return r; }
QP_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }
QP_HD u64 gl_root(int bits) { u64 r = GL_POW2_GEN; for (int i = bits; i < 32; i++) r = gl_sqr(r); return r; }

QP_HD u64 gl_mul7(u64 x) { u64 t2 = gl_add(x, x), t4 = gl_add(t2, t2), t8 = gl_add(t4, t4); return gl_sub(t8, x); }

struct Ext { u64 a, b; };
QP_HD Ext ex(u64 a, u64 b = 0) { Ext e; e.a = a; e.b = b; return e; }
QP_HD Ext ex_add(Ext x, Ext y) { return ex(gl_add(x.a, y.a), gl_add(x.b, y.b)); }
QP_HD Ext ex_sub(Ext x, Ext y) { return ex(gl_sub(x.a, y.a), gl_sub(x.b, y.b)); }
QP_HD Ext ex_mul(Ext x, Ext y) {
  u64 bb7 = gl_mul7(gl_mul(x.b, y.b));
  return ex(gl_add(gl_mul(x.a, y.a), bb7), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a)));
}
QP_HD Ext ex_scale(Ext x, u64 s) { return ex(gl_mul(x.a, s), gl_mul(x.b, s)); }
QP_HD Ext ex_add_base(Ext x, u64 s) { return ex(gl_add(x.a, s), x.b); }
QP_HD Ext ex_inv(Ext x) {
  u64 b27 = gl_mul7(gl_sqr(x.b));
  u64 ni = gl_inv(gl_sub(gl_sqr(x.a), b27));
  return ex(gl_mul(x.a, ni), gl_mul(gl_neg(x.b), ni));
}
QP_HD Ext ex_pow(Ext x, u64 e) { Ext r = ex(1); while (e) { if (e & 1) r = ex_mul(r, x); x = ex_mul(x, x); e >>= 1; } return r; }
QP_HD bool ex_eq(Ext x, Ext y) { return x.a == y.a && x.b == y.b; }

QP_HD u32 rev_bits(u32 x, int bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  u32 r = 0; for (int i = 0; i < bits; i++) r = (r << 1) | ((x >> i) & 1); return r;
#endif
}

} // namespace qp
