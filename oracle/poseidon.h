// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h header).
// CPU restatement of qp-plonky2 1.1.1's Poseidon-Goldilocks (width 12, rate 8, capacity 4,
// x^7, 4+22+4 rounds; source not on disk) per SURVEY.md App. A.2. Pinned by the reference's
// known-answer tests: wormhole/tests/src/circuit/unspendable_account_tests.rs:12-27,
// wormhole/tests/src/prover/prover_tests.rs:31-35, wormhole/tests/test-helpers/src/lib.rs:68-80.
#pragma once
#include "gl.h"
#include <array>
#include <cstring>

namespace orc {

static const int SPONGE_WIDTH = 12, SPONGE_RATE = 8, N_FULL_HALF = 4, N_PARTIAL = 22, N_ROUNDS = 30;
static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

const u64* poseidon_round_constants();  // 360 values, index 12*r+i (generated: ChaCha8Rng seed 0)

typedef std::array<u64, 12> State;
typedef std::array<u64, 4> Hash;

void poseidon_permute(u64 st[12]);            // literal rounds (primary constants only) unless poseidon_set_fast(true)
void poseidon_set_fast(bool on);               // opt into the derived fast partial rounds (shared tables with the product: timing only)
bool poseidon_is_fast();
void poseidon_permute_generic(u64 st[12]);
Hash hash_no_pad(const u64* x, size_t len);
Hash hash_or_noop(const u64* x, size_t len);
Hash two_to_one(const Hash& l, const Hash& r);

// Field-generic pieces used by the PoseidonGate constraint evaluator (base field and F_{p^2}).
template <class T> static inline T sbox7(T x) { T x2 = x * x; T x4 = x2 * x2; T x3 = x * x2; return x3 * x4; }
template <class T> static inline void mds_layer_t(T st[12]) {
  T out[12];
  for (int r = 0; r < 12; r++) {
    T acc = from_base((T*)0, 0);
    for (int i = 0; i < 12; i++) acc = acc + st[(i + r) % 12] * from_base((T*)0, MDS_CIRC[i]);
    acc = acc + st[r] * from_base((T*)0, MDS_DIAG[r]);
    out[r] = acc;
  }
  for (int r = 0; r < 12; r++) st[r] = out[r];
}

// Base-field specialisation (the CPU prover evaluates the Poseidon gate on every LDE point): 128-bit row accumulation.
template <> inline void mds_layer_t<FB>(FB st[12]) {
  u64 out[12];
  for (int r = 0; r < 12; r++) {
    u128 acc = (u128)st[r].v * MDS_DIAG[r];
    for (int i = 0; i < 12; i++) acc += (u128)st[(i + r) % 12].v * MDS_CIRC[i];
    out[r] = gl_red128(acc);
  }
  for (int r = 0; r < 12; r++) st[r] = FB(out[r]);
}

} // namespace orc
