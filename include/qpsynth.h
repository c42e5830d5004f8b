/* qpsynth — synthetic "wormhole-shaped" / "voting-shaped" circuit + witness generator (host only, no GPU).
 *
 * Seeded circuits and witnesses of the shapes the reference's artefacts pin (wormhole/bench-data/common.bin: k = 14 zk, 135 wires, 84
 * constants/sigmas columns, 16 public inputs; aggregator/data/dummy_proof.bin: k = 13) with random gate contents: the workloads of bench.py
 * and of the kernel/prover parity tests, including shapes the real circuits do not reach (2^20 rows, upstream's recursion gates). The
 * reference's actual circuits and their witness generation are include/qpcircuit.h.
 * Gate mix follows SURVEY.md §8(d) (from wormhole/circuit/src/storage_proof/mod.rs:169-243, nullifier.rs:230-238,
 * unspendable_account.rs:194-204). It is an input generator for tests and bench.py, not part of the drop-in ABI.
 */
#ifndef QPSYNTH_H
#define QPSYNTH_H
#include "qpgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct qpsynth_params {
  uint32_t degree_bits;
  uint32_t zero_knowledge;                 /* 1: standard_recursion_zk_config shape (salts + random Noop rows) */
  uint32_t n_poseidon, n_basesum, n_arith, n_constant;
  uint32_t n_copies;                       /* number of copy constraints (non-identity sigma entries) */
  uint32_t fri_num_layers; uint32_t fri_arity_bits[8];
  uint64_t circuit_seed;
  uint32_t fri_cap_height, fri_num_queries, fri_pow_bits;   /* 0 = preset default (4, 28, 16); pow_bits uses 0xFFFFFFFF for "0 bits" */
  uint32_t reserved;                       /* 1: recursion gates with the second parameter set (preset "RT2") */
  /* rows of the recursive-verifier gate set (all zero: leaf gate set only), in order ArithmeticExtension, MulExtension, Reducing,
   * ReducingExtension, RandomAccess{4}, PoseidonMds, Exponentiation, CosetInterpolation{4} — SURVEY.md App. D */
  uint32_t n_recursion[8];
} qpsynth_params;

typedef struct qpsynth_circuit qpsynth_circuit;

/* Presets: "A" leaf non-zk k=13, "B" leaf zk k=14, "V" voting-shaped k=20, "T" tiny k=6 test circuit,
 * "R" / "RT" recursion-shaped (all 14 gate types) k=13 / k=8, "RT2" = RT with other gate parameters. */
int qpsynth_preset(const char* name, qpsynth_params* out);
int qpsynth_circuit_create(const qpsynth_params* p, qpsynth_circuit** out);
void qpsynth_circuit_destroy(qpsynth_circuit* c);
const qpgpu_circuit_desc* qpsynth_circuit_desc(const qpsynth_circuit* c);
/* Full witness for proof index `seed`: wires [num_wires][n] column-major, public_inputs [16]. */
int qpsynth_witness(const qpsynth_circuit* c, uint64_t seed, uint64_t* wires, uint64_t* public_inputs);
/* Salt columns exactly as the device derives them from (opts.salt_key, opts.salt_stream): out [3][4][N] — the ChaCha8 keystream of
 * csrc/salt.h on the host (the oracle side of the parity tests gets its salts from here). */
void qpsynth_salts_keyed(const qpsynth_circuit* c, const uint8_t* key32, uint64_t stream, uint64_t* out);
/* The fixed PUBLIC key the tests and bench.py use with salt_stream = seed (a test convention: never a secret). */
#define QPSYNTH_TEST_SALT_KEY_BYTES 0x51, 0x50, 0x2d, 0x74, 0x65, 0x73, 0x74, 0x2d, 0x73, 0x61, 0x6c, 0x74, 0x2d, 0x6b, 0x65, 0x79, \
                                    0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0a, 0x0b, 0x0c, 0x0d, 0x0e, 0x0f
void qpsynth_salts(const qpsynth_circuit* c, uint64_t salt_seed, uint64_t* out);   /* = qpsynth_salts_keyed(test key, salt_seed) */
/* One ChaCha8 block of the salt stream as eight field elements (known-answer tests of csrc/salt.h). */
void qpsynth_salt_block(const uint8_t* key32, uint64_t nonce, uint64_t block, uint64_t* out8);
const char* qpsynth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
