// Device-only Poseidon-Goldilocks permutation tuned for gfx950 VALU issue (profiles/r01_valu_issue_rates.txt: v_mad_u64_u32
// issues at the same ~4.3 cycles as a 32-bit add-with-carry, so the design minimises instruction count, not multiplies).
//  * state elements are "lazy" residues: any u64 congruent to the value (canonicalised once, at the end);
//  * 64x64->128 products are four v_mad_u64_u32 chained through their 64-bit addends;
//  * the MDS row sums accumulate the 32-bit halves separately with v_mad_u64_u32 (constants < 2^6, sums < 2^42) and the next
//    round's constant rides in as the initial addend, so add-round-constant costs nothing after round 0;
//  * one 128->64 reduction per product / per MDS row, using 2^64 = 2^32-1 and 2^96 = -1 (mod p).
// Same function as poseidon.h (tests compare both with the oracle); reference semantics: SURVEY.md App. A.2.
#pragma once
#include "gl.h"
#include "poseidon.h"
#include "poseidon_fast_dev_constants.h"

namespace qp {

// mad32, mad_co, add_co/addc_co/sub_co/subb_co, sel_mask, pack64, red_parts, red128_lazy, mul_lazy, canon: gl.h (device section)
// 128-bit product a*b -> (lo, hi)
__device__ __forceinline__ void mul_wide(u64 a, u64 b, u64& lo, u64& hi) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 cq, unused;
  const u64 p0 = mad32(a0, b0, 0);
  const u64 p1 = mad32(a0, b1, p0 >> 32);
  const u64 p2 = mad_co(a1, b0, p1, cq);
  const u64 h = mad32(a1, b1, p2 >> 32);
  hi = pack64((u32)h, addc_co((u32)(h >> 32), 0u, cq, unused));      // the product is < 2^128: no carry out
  lo = pack64((u32)p0, (u32)p2);
}
// a0*b0 + a1*b1 (two 128-bit products, one reduction); result lazy. 2^128 == -2^32 (mod p).
__device__ __forceinline__ u64 dot2_lazy(u64 a0, u64 b0, u64 a1, u64 b1) {
  u64 l0, h0, l1, h1; mul_wide(a0, b0, l0, h0); mul_wide(a1, b1, l1, h1);
  u64 lo = l0 + l1; u64 c0 = lo < l0;
  u64 hi = h0 + h1; u64 c1 = hi < h0;
  hi += c0; c1 += (hi < c0);
  u64 r = red128_lazy(lo, hi);
  // subtract c1 * 2^32 (c1 in {0,1}); r is lazy, so guard the borrow with +p
  u64 sub = c1 << 32;
  u64 d = r - sub; if (r < sub) d += GL_P;     // r + p - sub, cannot wrap twice since sub <= 2^32
  return d;
}
// 192-bit accumulator for sums of 128-bit products (up to 2^32 terms): one carry chain of five 32-bit adds per product
struct Acc192 { u64 lo, hi; u32 top; };
__device__ __forceinline__ void acc_mul(Acc192& a, u64 x, u64 y) {
  u64 l, h, c; mul_wide(x, y, l, h);
  const u32 r0 = add_co((u32)a.lo, (u32)l, c);
  const u32 r1 = addc_co((u32)(a.lo >> 32), (u32)(l >> 32), c, c);
  const u32 r2 = addc_co((u32)a.hi, (u32)h, c, c);
  const u32 r3 = addc_co((u32)(a.hi >> 32), (u32)(h >> 32), c, c);
  a.top = addc_co(a.top, 0u, c, c);
  a.lo = pack64(r0, r1); a.hi = pack64(r2, r3);
}
// canonical value of the accumulator: lo + 2^64 hi + 2^128 top, with 2^128 == -2^32
__device__ __forceinline__ u64 acc_reduce(const Acc192& a) {
  u64 r = red128_lazy(a.lo, a.hi);
  r = r >= GL_P ? r - GL_P : r;
  u64 t = gl_mul((u64)a.top, 1ULL << 32);
  return gl_sub(r, t);
}

__device__ __forceinline__ u64 sbox7_lazy(u64 x) {
  u64 x2 = mul_lazy(x, x), x4 = mul_lazy(x2, x2), x3 = mul_lazy(x2, x);
  return mul_lazy(x3, x4);
}
// x + c for lazy x and canonical c
__device__ __forceinline__ u64 add_lazy_canon(u64 x, u64 c) {
  u64 s = x + c; if (s < c) s += GL_EPS; return s;
}

// sl + 2^32*sh for column sums sl, sh < 2^44  ->  lazy residue: 2^32*sh = (sh_lo << 32) + sh_hi*(2^32-1); t < 2^45, so after a carry
// out of the high-word add the wrapped value is < 2^45 and the +(2^32-1) correction cannot wrap again.
__device__ __forceinline__ u64 mds_row_fold(u64 sl, u64 sh) {
  const u64 t = mad32((u32)(sh >> 32), 0xFFFFFFFFu, sl);
  u64 k;
  const u32 vh = add_co((u32)(t >> 32), (u32)sh, k);
  return pack64((u32)t, vh) + sel_mask(0u, 0xFFFFFFFFu, k);
}
// MDS with the next round's constants folded in: st[r] = sum_i C[i]*st[(i+r)%12] + 8*st[0]*[r==0] + rc[r]   (lazy)
__device__ __forceinline__ void mds_rc_lazy(u64 st[12], const u64* __restrict__ rc) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { lo[i] = (u32)st[i]; hi[i] = (u32)(st[i] >> 32); }
#pragma unroll
  for (int r = 0; r < 12; r++) {
    u64 c = rc ? rc[r] : 0;
    u64 sl = (u32)c, sh = c >> 32;
#pragma unroll
    for (int i = 0; i < 12; i++) { sl = mad32(lo[(i + r) % 12], C[i], sl); sh = mad32(hi[(i + r) % 12], C[i], sh); }
    if (r == 0) { sl = mad32(lo[0], 8, sl); sh = mad32(hi[0], 8, sh); }
    // value = sl + 2^32*sh, sl, sh < 2^43:  2^32*sh = (sh_lo << 32) + sh_hi*2^64 == (sh_lo << 32) + sh_hi*(2^32-1)
    st[r] = mds_row_fold(sl, sh);
  }
}

// ---- fast form of the 22 partial rounds on the device (tables: tools/gen_poseidon_fast.py). A literal partial round is a whole 12x12 MDS
// product (288 multiply-adds + 12 row folds) around one S-box; in the fast form it is one S-box, one 11-term dot product and 11 multiply-adds
// with full 64-bit constants. What makes that cheaper here: the dot product takes each constant a as 27/27/10-bit limbs of a and of
// a' = a*2^32 (mod p), so a*x = a*x_lo + a'*x_hi is six 32x32 multiply-adds into three column sums that cannot overflow (22 terms < 2^59
// each), with ONE recombination per dot product instead of a reduction per term: ~80 instructions per dot, ~16 per multiply-add,
// 366 per round against 472 (static counts: tools/poseidon_issue_bound.py).
__device__ __constant__ const u32 PFD_INIT[11 * 11 * 6] = {QP_POSEIDON_FASTDEV_INIT};     // dense product in front of the partial rounds (limbs)
__device__ __constant__ const u32 PFD_WHAT[22 * 11 * 6] = {QP_POSEIDON_FASTDEV_WHAT};     // per round: the dot product's eleven constants (limbs)
__device__ __constant__ const u64 PFD_VQ[22 * 12] = {QP_POSEIDON_FASTDEV_VQ};             // per round: the eleven multipliers, then the next S-box's constant
__device__ __constant__ const u64 PFD_Q[22] = {QP_POSEIDON_FAST_Q};
__device__ __constant__ const u64 PFD_RC_HEAD[60] = {QP_POSEIDON_FASTDEV_RC_HEAD};        // constants of rounds 0-3, then what follows the fourth round
__device__ __constant__ const u64 PFD_RC_PAD[372] = {QP_POSEIDON_FASTDEV_RC_PAD};         // all round constants, then a row of zeros
// Forces the pending scalar loads to have landed here (an empty asm that reads one of the loaded values).
__device__ __forceinline__ void smem_touch(u32 x) { asm volatile("" :: "s"(x)); }

// a*b + c -> lazy residue; a, b, c any u64. The low half of c rides in the first multiply-add; a0*b1 + (p0 >> 32) <= 2^64 - 2^32, so the
// high half of c (< 2^32) still fits on top of it.
__device__ __forceinline__ u64 mul_add_lazy(u64 a, u64 b, u64 c) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 cq;
  const u64 p0 = mad32(a0, b0, (u64)(u32)c);
  const u64 p1 = mad32(a0, b1, p0 >> 32) + (c >> 32);
  const u64 p2 = mad_co(a1, b0, p1, cq);
  const u64 hi = mad32(a1, b1, p2 >> 32);
  return red_parts<true>((u32)p0, (u32)p2, hi, cq);
}
// the same with a wave-uniform b: its halves stay in scalar registers (a VOP3 instruction reads one of them directly)
__device__ __forceinline__ u64 mad_co_su(u32 a, u32 b_uniform, u64 c, u64& co) {
  u64 d; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(a), "s"(b_uniform), "v"(c)); return d;
}
__device__ __forceinline__ u64 mul_add_lazy_u(u64 a, u64 b_uniform, u64 c) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b_uniform, b1 = (u32)(b_uniform >> 32);
  u64 cq;
  const u64 p0 = mad32(a0, b0, (u64)(u32)c);
  const u64 p1 = mad32(a0, b1, p0 >> 32) + (c >> 32);
  const u64 p2 = mad_co_su(a1, b0, p1, cq);
  const u64 hi = mad32(a1, b1, p2 >> 32);
  return red_parts<true>((u32)p0, (u32)p2, hi, cq);
}
__device__ __forceinline__ u64 sbox7_add_lazy(u64 x, u64 c) {      // x^7 + c
  const u64 x2 = mul_lazy(x, x), x4 = mul_lazy(x2, x2), x3 = mul_lazy(x2, x);
  return mul_add_lazy(x3, x4, c);
}
// sum_{j<11} a_j * x[j] (+ 25*s0 if WITH_S0) -> lazy residue. c = the limb table of the eleven constants (wave-uniform: scalar loads).
// Column sums T_k = sum_j (a_j limb k)*x_lo + (a'_j limb k)*x_hi, value = T0 + 2^27 T1 + 2^54 T2. With T1 = t1l + 2^32 t1h, T2 = t2l + 2^32 t2h:
//   2^54 T2 = 2^32 (2^22 t2l) + 2^64 (2^22 t2h), and 2^64 == 2^32 - 1, so with X2 = 2^22 t2h (< 2^37):
//   value == X0 + 2^32 X1,  X0 = T0 + 2^27 t1l - X2 (< 2^64),  X1 = 2^27 t1h + 2^22 t2l + X2 (< 2^59).
// X0 cannot go negative: every low limb is >= 1 (asserted by the generator), so T0 >= sum_j (x_lo + x_hi) > T2 / 2^10 >= X2.
struct Col3 { u64 T0, T1, T2; };
__device__ __forceinline__ Col3 dot_begin(u64 s0) {       // 25*s0 = 25*s0_lo + 2^27 * (25 * 2^5) * s0_hi
  Col3 a; a.T0 = mad32((u32)s0, 25u, 0); a.T1 = mad32((u32)(s0 >> 32), 25u << 5, 0); a.T2 = 0; return a;
}
template <int N, typename W>
__device__ __forceinline__ void dot_terms(Col3& a, const W& c, const u64* x) {
#pragma unroll
  for (int j = 0; j < N; j++) {
    const u32 xl = (u32)x[j], xh = (u32)(x[j] >> 32);
    a.T0 = mad32(xl, c[6 * j + 0], a.T0); a.T1 = mad32(xl, c[6 * j + 1], a.T1); a.T2 = mad32(xl, c[6 * j + 2], a.T2);
    a.T0 = mad32(xh, c[6 * j + 3], a.T0); a.T1 = mad32(xh, c[6 * j + 4], a.T1); a.T2 = mad32(xh, c[6 * j + 5], a.T2);
  }
}
__device__ __forceinline__ u64 dot_finish(const Col3& a) {
  const u64 X2 = mad32((u32)(a.T2 >> 32), 1u << 22, 0);
  const u64 X1 = mad32((u32)(a.T1 >> 32), 1u << 27, mad32((u32)a.T2, 1u << 22, X2));
  const u64 X0 = mad32((u32)a.T1, 1u << 27, a.T0) - X2;
  u64 k1, k2, unused;
  const u32 l1 = add_co((u32)(X0 >> 32), (u32)X1, k1);
  const u32 h = addc_co((u32)(X1 >> 32), 0u, k1, unused);              // < 2^28
  u64 r = mad_co(h, 0xFFFFFFFFu, pack64((u32)X0, l1), k2);            // + h * 2^64; on a carry the wrapped value is < 2^60, so the fix cannot wrap
  r += sel_mask(0u, 0xFFFFFFFFu, k2);
  return r;
}
// eleven-term dot product without the s0 term (the dense 11x11 product in front of the partial rounds)
__device__ __forceinline__ u64 dot11_dev(const u32* __restrict__ c, const u64* x) {
  Col3 a; a.T0 = 0; a.T1 = 0; a.T2 = 0;
  dot_terms<11>(a, c, x);
  return dot_finish(a);
}

// The same with the twelve constants already in registers. Left alone the compiler sinks each row's scalar load down to its use (twelve
// exposed scalar-load latencies per round: 28 % of a leaf-hash wave's lifetime parked at s_waitcnt, tools/pmc_wave_stalls.py); the callers below
// request the whole row block at the top of the round, behind a scheduling barrier, so it arrives while the S-boxes run.
__device__ __forceinline__ void mds_rc_lazy_pre(u64 st[12], const u64 (&c)[12]) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  u32 lo[12], hi[12];
#pragma unroll
  for (int i = 0; i < 12; i++) { lo[i] = (u32)st[i]; hi[i] = (u32)(st[i] >> 32); }
#pragma unroll
  for (int r = 0; r < 12; r++) {
    u64 sl = (u32)c[r], sh = c[r] >> 32;
#pragma unroll
    for (int i = 0; i < 12; i++) { sl = mad32(lo[(i + r) % 12], C[i], sl); sh = mad32(hi[(i + r) % 12], C[i], sh); }
    if (r == 0) { sl = mad32(lo[0], 8, sl); sh = mad32(hi[0], 8, sh); }
    st[r] = mds_row_fold(sl, sh);
  }
}
// One round with its constants one round ahead: `c` (what this round's MDS product adds: the next round's constants) was requested a whole
// round ago; the request for the round after (`next`, 12 wave-uniform values) goes out first, behind a scheduling barrier, and replaces c.
template <bool FULL>
__device__ __forceinline__ void round_pre(u64 st[12], u64 (&c)[12], const u64* __restrict__ next) {
  u64 n[12];
#pragma unroll
  for (int i = 0; i < 12; i++) n[i] = next[i];
  __builtin_amdgcn_sched_barrier(0);
  st[0] = sbox7_lazy(st[0]);
  if (FULL) {
#pragma unroll
    for (int i = 1; i < 12; i++) st[i] = sbox7_lazy(st[i]);
  }
  mds_rc_lazy_pre(st, c);
#pragma unroll
  for (int i = 0; i < 12; i++) c[i] = n[i];
}

// Full permutation on lazy/canonical input; output canonical. FAST selects the form of the 22 partial rounds: the fast form has 8 % fewer
// instructions (18.1 k against 19.7 k) and wins wherever the chip is full (2.6 against 2.29 G permutations/s saturated); a lone wave per SIMD
// runs the literal form faster (its round is 472 mostly independent instructions, the fast round a chain S-box -> dot product -> S-box), so
// the latency-bound stages keep it.
template <bool FAST>
__device__ __forceinline__ void poseidon_permute_dev_t(u64 st[12]) {
  const u64* rc = POSEIDON_RC_DEV;
  int r = 0;
  if constexpr (!FAST) {
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = add_lazy_canon(st[i], rc[i]);
  u64 c[12];
#pragma unroll
  for (int i = 0; i < 12; i++) c[i] = PFD_RC_PAD[12 + i];
  for (; r < 4; r++) round_pre<true>(st, c, PFD_RC_PAD + 12 * (r + 2));
  for (; r < 26; r++) round_pre<false>(st, c, PFD_RC_PAD + 12 * (r + 2));
  for (; r < 30; r++) round_pre<true>(st, c, PFD_RC_PAD + 12 * (r + 2 < 30 ? r + 2 : 30));
  } else {
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = add_lazy_canon(st[i], PFD_RC_HEAD[i]);
  u64 c[12];
#pragma unroll
  for (int i = 0; i < 12; i++) c[i] = PFD_RC_HEAD[12 + i];
  for (; r < 4; r++) round_pre<true>(st, c, PFD_RC_HEAD + 12 * (r + 2 < 5 ? r + 2 : 4));
  {
    u64 t[11];
#pragma unroll
    for (int i = 0; i < 11; i++) t[i] = dot11_dev(PFD_INIT + 66 * i, st + 1);
#pragma unroll
    for (int i = 0; i < 11; i++) st[1 + i] = t[i];
  }
  // Software-pipelined by hand. Round t's eleven multiply-adds (independent of each other, off the critical path) are issued together with
  // round t+1's S-box (a chain of four dependent multiplications), so a lone wave has something to issue while the chain waits. The round's
  // 90 dwords of constants arrive by scalar loads, whose wait is all-or-nothing (lgkmcnt(0)): each chunk is requested one sub-section
  // ahead of its first use, and the scheduling barriers keep the compiler from sinking the requests down to the uses (as it does when
  // left alone: one exposed scalar-load latency per 16 dwords).
  // Chunks of a round's constants: w1 = dot terms 0-3 (24 dwords), w2 = terms 4-7 (24), w3 = terms 8-10 (18), vq = V[t] and Q[t+1] (24).
  // Requests: vq in front of the dot product's second part (used 54 instructions later), w1' in front of the S-box (used a round's
  // multiply-adds later), w2' and w3' in front of the second half of the multiply-adds; at most 78 dwords are live at once. Every request
  // follows a dummy use of the previous one, so no wait ever has a fresh request to drain.
  u32 w1[24], w2[24], w3[18];
#pragma unroll
  for (int k = 0; k < 24; k++) { w1[k] = PFD_WHAT[k]; w2[k] = PFD_WHAT[24 + k]; }
#pragma unroll
  for (int k = 0; k < 18; k++) w3[k] = PFD_WHAT[48 + k];
  u64 s0 = sbox7_add_lazy(st[0], PFD_Q[0]);
  for (int t = 0; t < 21; t++) {
    __builtin_amdgcn_sched_barrier(0);
    Col3 a = dot_begin(s0);
    dot_terms<4>(a, w1, st + 1);
    __builtin_amdgcn_sched_barrier(0);
    smem_touch(w3[0]);
    u64 vq[12];
#pragma unroll
    for (int j = 0; j < 12; j++) vq[j] = PFD_VQ[12 * t + j];
    __builtin_amdgcn_sched_barrier(0);
    dot_terms<4>(a, w2, st + 5);
    dot_terms<3>(a, w3, st + 9);
    const u64 d = dot_finish(a);
    __builtin_amdgcn_sched_barrier(0);
    smem_touch((u32)vq[0]);
#pragma unroll
    for (int k = 0; k < 24; k++) w1[k] = PFD_WHAT[66 * (t + 1) + k];
    __builtin_amdgcn_sched_barrier(0);
    const u64 s0n = sbox7_add_lazy(d, vq[11]);
#pragma unroll
    for (int j = 0; j < 5; j++) st[1 + j] = mul_add_lazy_u(s0, vq[j], st[1 + j]);
    __builtin_amdgcn_sched_barrier(0);
    smem_touch(w1[0]);
#pragma unroll
    for (int k = 0; k < 24; k++) w2[k] = PFD_WHAT[66 * (t + 1) + 24 + k];
#pragma unroll
    for (int k = 0; k < 18; k++) w3[k] = PFD_WHAT[66 * (t + 1) + 48 + k];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 5; j < 11; j++) st[1 + j] = mul_add_lazy_u(s0, vq[j], st[1 + j]);
    s0 = s0n;
  }
  __builtin_amdgcn_sched_barrier(0);
  {
    Col3 a = dot_begin(s0);
    dot_terms<4>(a, w1, st + 1);
    dot_terms<4>(a, w2, st + 5);
    dot_terms<3>(a, w3, st + 9);
    st[0] = dot_finish(a);
#pragma unroll
    for (int j = 0; j < 11; j++) st[1 + j] = mul_add_lazy_u(s0, PFD_VQ[12 * 21 + j], st[1 + j]);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = add_lazy_canon(st[i], rc[12 * 26 + i]);
#pragma unroll
  for (int i = 0; i < 12; i++) c[i] = PFD_RC_PAD[12 * 27 + i];
  for (r = 26; r < 30; r++) round_pre<true>(st, c, PFD_RC_PAD + 12 * (r + 2 < 30 ? r + 2 : 30));
  }
#pragma unroll
  for (int i = 0; i < 12; i++) st[i] = canon(st[i]);
}
__device__ __forceinline__ void poseidon_permute_dev(u64 st[12]) { poseidon_permute_dev_t<false>(st); }

// ---------------------------------------------------------------------------------------------------------------------
// Pair permutation: two adjacent lanes (2k, 2k+1) share one state; the lane with parity p holds elements 2j+p, j < 6.
// Total instruction work equals the one-permutation-per-lane form (each lane runs 6 S-boxes and 6 MDS rows; the other
// half of every row's inputs arrives through twelve quad_perm DPP moves), but a stage gets twice the waves, which is what
// the single-proof case lacks: leaf hashing of 2^17 leaves is only 2 waves/SIMD per-lane (1.43 G perm/s) versus 4 waves
// as pairs (1.58 G perm/s measured for the per-lane kernel at that occupancy), and tree levels 1-2 halve their latency.
struct PairCtx { u32 co[6]; u32 d0; int p; };
__device__ __forceinline__ void pair_init(PairCtx& cx, int p) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
#pragma unroll
  for (int d = 0; d < 6; d++) cx.co[d] = p ? C[(2 * d + 11) % 12] : C[2 * d + 1];   // multiplier of the partner's element j' for row j, d = (j'-j) mod 6
  cx.d0 = p ? 0u : 8u; cx.p = p;
}
__device__ __forceinline__ u32 dpp_swap_pair(u32 v) { return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false); }
// own[j] = element 2j+p (lazy in, lazy out); next-round constants rcn (uniform pointer to 12 values) or null
__device__ __forceinline__ void pair_mds_rc(u64 own[6], const PairCtx& cx, const u64* __restrict__ rcn) {
  const u32 CE[6] = {17, 41, 2, 13, 39, 34};      // C[2d]: multiplier of own element j' for row j, d = (j'-j) mod 6
  u32 lo[6], hi[6], olo[6], ohi[6];
#pragma unroll
  for (int j = 0; j < 6; j++) { lo[j] = (u32)own[j]; hi[j] = (u32)(own[j] >> 32); olo[j] = dpp_swap_pair(lo[j]); ohi[j] = dpp_swap_pair(hi[j]); }
#pragma unroll
  for (int j = 0; j < 6; j++) {
    u64 c = 0;
    if (rcn) { u64 c0 = rcn[2 * j], c1 = rcn[2 * j + 1]; c = cx.p ? c1 : c0; }
    u64 sl = (u32)c, sh = c >> 32;
#pragma unroll
    for (int jj = 0; jj < 6; jj++) {
      const int d = (jj - j + 6) % 6;
      sl = mad32(lo[jj], CE[d], sl); sh = mad32(hi[jj], CE[d], sh);
      sl = mad32(olo[jj], cx.co[d], sl); sh = mad32(ohi[jj], cx.co[d], sh);
    }
    if (j == 0) { sl = mad32(lo[0], cx.d0, sl); sh = mad32(hi[0], cx.d0, sh); }
    own[j] = mds_row_fold(sl, sh);
  }
}
// Both lanes of a pair must call this together (whole waves: never under divergent control flow). Output canonical.
__device__ __forceinline__ void poseidon_permute_pair(u64 own[6], const PairCtx& cx) {
  const u64* rc = POSEIDON_RC_DEV;
#pragma unroll
  for (int j = 0; j < 6; j++) { u64 c0 = rc[2 * j], c1 = rc[2 * j + 1]; own[j] = add_lazy_canon(own[j], cx.p ? c1 : c0); }
  int r = 0;
  for (; r < 4; r++) {
#pragma unroll
    for (int j = 0; j < 6; j++) own[j] = sbox7_lazy(own[j]);
    pair_mds_rc(own, cx, rc + 12 * (r + 1));
  }
  for (; r < 26; r++) {
    if (cx.p == 0) own[0] = sbox7_lazy(own[0]);
    pair_mds_rc(own, cx, rc + 12 * (r + 1));
  }
  for (; r < 29; r++) {
#pragma unroll
    for (int j = 0; j < 6; j++) own[j] = sbox7_lazy(own[j]);
    pair_mds_rc(own, cx, rc + 12 * (r + 1));
  }
#pragma unroll
  for (int j = 0; j < 6; j++) own[j] = sbox7_lazy(own[j]);
  pair_mds_rc(own, cx, nullptr);
#pragma unroll
  for (int j = 0; j < 6; j++) own[j] = canon(own[j]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Cooperative permutation: 16 consecutive lanes hold one state (lane i < 12 holds element i; lanes 12..15 idle), so a
// 64-lane wave runs 4 permutations with ~1/12 of the per-lane instruction stream each. Throughput is ~3x worse than the
// one-permutation-per-lane form, latency ~6x better: used where a stage has too few independent permutations to fill the chip
// (upper Merkle levels, FRI layer leaves). `xch` is this group's 12-u64 exchange slot in LDS; the caller's block is ONE wave
// (64 threads), so the barriers below are wave-local.
// A group's exchange slots are only touched by lanes of one wave, and a wave's LDS operations execute in order, so the
// write -> read hand-off needs no s_barrier: only the compiler must keep the order (wave_barrier is a scheduling fence, no
// instruction). QPGPU_COOP_SBARRIER restores the workgroup barrier for A/B runs.
#if defined(QPGPU_COOP_SBARRIER)
#define COOP_SYNC() __syncthreads()
#else
#define COOP_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif
struct CoopCtx {
  u32 m[12];      // this lane's MDS row: m[j] multiplies element j
  int ii;         // element index of this lane (0 for the idle lanes 12..15)
};
__device__ __forceinline__ void coop_init(CoopCtx& cx, int i) {
  const u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const int ii = i < 12 ? i : 0;
  cx.ii = ii;
#pragma unroll
  for (int j = 0; j < 12; j++) cx.m[j] = C[(j - ii + 12) % 12] + ((ii == 0 && j == 0) ? 8u : 0u);
}
// x: this lane's element (canonical or lazy in; canonical out). All 64 lanes of the wave must call this.
// `xch` points at TWO 12-slot exchange buffers (24 u64) used alternately, so one wave-local hand-off (COOP_SYNC) per round
// suffices. (volatile LDS accesses instead of the fence were measured 40 % slower: they serialise the twelve ds_reads.)
// The round loop stays rolled (30 x ~150 instructions); the next round's constant is a plain global load issued a full
// round ahead, so its latency hides behind the S-box. (Keeping the 30 per-lane constants in a register array made the
// compiler spill them to scratch once the loop was not unrolled: 296 B/lane of scratch traffic on the critical path.)
__device__ __forceinline__ u64 coop_permute(u64 x, int i, const CoopCtx& cx, u64* xch) {
  const u64* __restrict__ rcl = POSEIDON_RC_DEV + cx.ii;
  x = add_lazy_canon(x, rcl[0]);
  u64 c_next = rcl[12];
  for (int r = 0; r < 30; r++) {
    const bool full = r < 4 || r >= 26;
    const u64 c = c_next;
    c_next = r + 2 < 30 ? rcl[12 * (r + 2)] : 0;      // prefetch
    if (full || i == 0) x = sbox7_lazy(x);
    u64* buf = xch + (r & 1) * 12;
    if (i < 12) buf[i] = x;
    COOP_SYNC();
    const u64 cc = r + 1 < 30 ? c : 0;
    u64 sl = (u32)cc, sh = cc >> 32;
#pragma unroll
    for (int j = 0; j < 12; j++) { u64 v = buf[j]; sl = mad32((u32)v, cx.m[j], sl); sh = mad32((u32)(v >> 32), cx.m[j], sh); }
    x = mds_row_fold(sl, sh);
  }
  COOP_SYNC();
  return canon(x);
}

// (A DPP row_newbcast exchange instead of LDS was measured slower: profiles/r01_notes_ab_experiments.md.)

} // namespace qp
