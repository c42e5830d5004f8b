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

#if defined(__HIPCC__)
// ---- gfx950 carry-flag primitives. From C the compiler rebuilds carries with 64-bit compares + v_cndmask and never uses the
// carry-out of v_mad_u64_u32; spelled with the flags a modular multiply is 14 VALU instructions instead of 21-31 and an
// add/sub 5-6 instead of 7-8. Carries live in SGPR pairs (wave masks); the asm is not volatile, so independent chains still
// interleave. (profiles/r01_valu_issue_rates.txt: v_mad_u64_u32 and 64-bit adds issue at ~4.3 cycles, 32-bit ops at ~2.4.)
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 c) { return (u64)a * b + c; }
__device__ __forceinline__ u64 mad_co(u32 a, u32 b, u64 c, u64& co) {          // a*b + c and the carry out of bit 63
  u64 d; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(a), "v"(b), "v"(c)); return d;
}
__device__ __forceinline__ u32 add_co(u32 a, u32 b, u64& co) { u32 d; asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(d), "=s"(co) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u32 addc_co(u32 a, u32 b, u64 ci, u64& co) {        // a + b + carry_in
  u32 d; asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(a), "v"(b), "s"(ci)); return d;
}
__device__ __forceinline__ u32 sub_co(u32 a, u32 b, u64& bo) { u32 d; asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(d), "=s"(bo) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ u32 subb_co(u32 a, u32 b, u64 bi, u64& bo) {        // a - b - borrow_in
  u32 d; asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(d), "=s"(bo) : "v"(a), "v"(b), "s"(bi)); return d;
}
__device__ __forceinline__ u32 sel_mask(u32 if0, u32 if1, u64 m) { u32 d; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(if0), "v"(if1), "s"(m)); return d; }
__device__ __forceinline__ u64 pack64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }

// (l1:l0) + 2^64*hi [- 1 if HAS_BQ and borrow mask bq is set: a pending 2^96 = -1 term]  ->  lazy residue (any u64 congruent mod p).
// S = lo + h0*(2^32-1) - h1 - bq lies in (-2^32, 2^65 - 2^33), so floor(S/2^64) = c - b is in {-1,0,1} and
// r_raw + (c-b)(2^32-1) cannot wrap.
template <bool HAS_BQ>
__device__ __forceinline__ u64 red_parts(u32 l0, u32 l1, u64 hi, u64 bq) {
  const u32 h0 = (u32)hi, h1 = (u32)(hi >> 32);
  u64 b, c;
  const u32 x0 = HAS_BQ ? subb_co(l0, h1, bq, b) : sub_co(l0, h1, b);
  const u32 x1 = subb_co(l1, 0u, b, b);
  u64 r = mad_co(h0, 0xFFFFFFFFu, pack64(x0, x1), c);
  r += sel_mask(0u, 0xFFFFFFFFu, c);
  r -= sel_mask(0u, 0xFFFFFFFFu, b);
  return r;
}
__device__ __forceinline__ u64 red128_lazy(u64 lo, u64 hi) { return red_parts<false>((u32)lo, (u32)(lo >> 32), hi, 0); }
// a*b -> lazy residue; a, b any u64
__device__ __forceinline__ u64 mul_lazy(u64 a, u64 b) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 cq;
  const u64 p0 = mad32(a0, b0, 0);
  const u64 p1 = mad32(a0, b1, p0 >> 32);            // < 2^64
  const u64 p2 = mad_co(a1, b0, p1, cq);             // 65 bits: cq:p2; cq has weight 2^96 = -1 (mod p)
  const u64 hi = mad32(a1, b1, p2 >> 32);
  return red_parts<true>((u32)p0, (u32)p2, hi, cq);
}
__device__ __forceinline__ u64 canon(u64 x) { return x >= GL_P ? x - GL_P : x; }
#endif

// Branch-free (select) forms: on the host the inputs are effectively random, so branches would mispredict.
QP_HD u64 gl_add(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // s = a + b < 2p. s - p == s + (2^32-1) mod 2^64, and that add carries exactly when s >= p; take it if either add carried.
  u64 c1, c2;
  const u32 s0 = add_co((u32)a, (u32)b, c1), s1 = addc_co((u32)(a >> 32), (u32)(b >> 32), c1, c1);
  const u32 t0 = add_co(s0, 0xFFFFFFFFu, c2), t1 = addc_co(s1, 0u, c2, c2);
  const u64 m = c1 | c2;
  return pack64(sel_mask(s0, t0, m), sel_mask(s1, t1, m));
#else
  u64 s = a + b;
  u64 wrapped = 0 - (u64)(s < a);                 // all ones if the 64-bit add wrapped (2^64 == 2^32 - 1 mod p)
  u64 over = 0 - (u64)(s >= GL_P);                // only meaningful when not wrapped
  return s + (wrapped & GL_EPS) - (~wrapped & over & GL_P);   // canonical for canonical inputs
#endif
}
QP_HD u64 gl_sub(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  // d = a - b; on borrow add p, i.e. subtract 2^32 - 1
  u64 bo, b2;
  const u32 d0 = sub_co((u32)a, (u32)b, bo), d1 = subb_co((u32)(a >> 32), (u32)(b >> 32), bo, bo);
  const u32 e = sel_mask(0u, 0xFFFFFFFFu, bo);
  const u32 r0 = sub_co(d0, e, b2), r1 = subb_co(d1, 0u, b2, b2);
  return pack64(r0, r1);
#else
  u64 d = a - b; return d + ((0 - (u64)(a < b)) & GL_P);
#endif
}
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
QP_HD u64 gl_mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return canon(mul_lazy(a, b));
#else
  u64 lo, hi; mul64wide(a, b, lo, hi); return gl_reduce128(lo, hi);
#endif
}
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
