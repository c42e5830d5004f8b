// Poseidon-Goldilocks permutation / sponge (width 12, rate 8, x^7, 4+22+4 rounds) for host and gfx950 device.
// Semantics of qp-plonky2 1.1.1's PoseidonHash (hash_no_pad in overwrite mode, hash_or_noop, two_to_one) used by
// the reference's circuits, e.g. wormhole/circuit/src/nullifier.rs:53-73, unspendable_account.rs:38-63
// (SURVEY.md App. A.2). Round constants: poseidon_constants.h (generated).
#pragma once
#include "gl.h"
#include "poseidon_constants.h"

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

QP_HD void poseidon_permute(u64 st[12]) {
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
