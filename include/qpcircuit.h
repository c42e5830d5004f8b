/* qpcircuit — the reference's circuits and their witness generation, host-only C-ABI (libqpcircuit.so; no GPU code).
 *
 * SURVEY.md §8 rows a2 (generate_partial_witness) and f3 (the real wormhole witness): what stands between the reference's
 * `CircuitInputs` and the `wires[W][n]` matrix `qpgpu_prove` takes (include/qpgpu.h). Three layers, each citing what it restates:
 *   1. input encoding   = WormholeProver::commit / CircuitFragment::fill_targets (wormhole/prover/src/lib.rs:209-225,
 *                         wormhole/circuit/src/{nullifier.rs:244-253, unspendable_account.rs:210-220, storage_proof/mod.rs:246-307,
 *                         substrate_account.rs:91-97}, voting/src/lib.rs:199-260): bytes -> the values of the input targets in
 *                         allocation order (the PartialWitness);
 *   2. circuit building = WormholeCircuit::new (wormhole/circuit/src/circuit.rs:76-137) and its fragments' `circuit()` functions,
 *                         VoteCircuitData::circuit (voting/src/lib.rs:123-197), on this repo's own builder (zk-circuits_amd/circuit/builder.h);
 *   3. witness generation = qp-plonky2's generate_partial_witness: input targets -> full wire matrix + public inputs; inputs that do not
 *                         satisfy the statement fail here, as in the reference ("set twice with different values").
 * The statement, the public inputs and their order are the reference's, and so is the whole CommonCircuitData (the zk wormhole circuit's
 * common.bin is byte-identical to wormhole/bench-data/common.bin); rows, copy constraints and the circuit digest are this
 * builder's own ("parity unpinned": qp-plonky2's CircuitBuilder is not on disk), so proofs verify against the verifier data of the
 * circuit built here, not against the reference's verifier.bin.
 */
#ifndef QPCIRCUIT_H
#define QPCIRCUIT_H
#include "qpgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct qpc_circuit qpc_circuit;

/* Circuits. The fragments are what the reference's fragment tests build alone (wormhole/tests/src/circuit/{nullifier,unspendable_account,
 * storage_proof,substrate_account}_tests.rs via circuit_helpers.rs:7-31). */
enum {
  QPC_WORMHOLE = 0,             /* WormholeCircuit::new: all four fragments + connect_shared_targets; 16 public inputs */
  QPC_NULLIFIER = 1,            /* Nullifier::circuit; 4 public inputs */
  QPC_UNSPENDABLE_ACCOUNT = 2,  /* UnspendableAccount::circuit; no public inputs */
  QPC_STORAGE_PROOF = 3,        /* StorageProof::circuit; 8 public inputs (root hash, funding amount) */
  QPC_EXIT_ACCOUNT = 4,         /* SubstrateAccount::circuit (exit account as public input only); 4 public inputs */
  QPC_VOTE = 5,                 /* VoteCircuitData::circuit; 13 public inputs */
  QPC_AGGREGATION = 6           /* a tree node: built by qpc_aggregation_circuit_build, not by qpc_circuit_build */
};

/* zero_knowledge: 0 = CircuitConfig::standard_recursion_config, 1 = standard_recursion_zk_config (blinding rows + salted oracles). */
int qpc_circuit_build(uint32_t kind, uint32_t zero_knowledge, qpc_circuit** out);
void qpc_circuit_destroy(qpc_circuit* c);
/* Descriptor to hand to qpgpu_circuit_create; its arrays are owned by the circuit object. */
const qpgpu_circuit_desc* qpc_circuit_desc(const qpc_circuit* c);
size_t qpc_circuit_num_inputs(const qpc_circuit* c);
/* Rows by gate: [Noop (padding/blinding excluded), Constant, PublicInput, BaseSum, Arithmetic, Poseidon], then rows in use before blinding/padding. */
void qpc_circuit_row_counts(const qpc_circuit* c, uint32_t out[7]);
/* generate_partial_witness. inputs[num_inputs] = the input-target values in allocation order (qpw_fill_targets / qpv_fill_targets);
 * blinding_key: 32 secret bytes seeding the zk blinding rows (may be NULL for a non-zk circuit); wires [135][n] column-major, public_inputs
 * [desc->num_public_inputs]. QPGPU_ERR_WITNESS when the inputs do not satisfy the circuit (qpc_last_error names the conflict). */
int qpc_generate_witness(const qpc_circuit* c, const uint64_t* inputs, size_t num_inputs, const uint8_t* blinding_key, uint64_t* wires, uint64_t* public_inputs);
/* The same with an explicit number of host threads (0 = up to 8, what qpc_generate_witness uses; 1 = in program order on the caller's thread).
 * Circuits declare independent segments when they are built (a node's children and their FRI queries, the storage proof's nodes); a circuit
 * without them, or whose segments turn out not to be independent, is generated on one thread whatever is asked (qpc_circuit_parallel_segments). */
int qpc_generate_witness_mt(const qpc_circuit* c, const uint64_t* inputs, size_t num_inputs, const uint8_t* blinding_key, uint64_t* wires, uint64_t* public_inputs,
                            uint32_t threads);
int qpc_circuit_parallel_segments(const qpc_circuit* c);
const char* qpc_last_error(void);

/* ---- wormhole inputs (wormhole/circuit/src/inputs.rs:23-55). 32-byte digests are BytesDigest: four little-endian u64, each < p. */
#define QPW_MAX_PROOF_LEN 20          /* storage_proof/mod.rs:22 */
#define QPW_PROOF_NODE_MAX_FELTS 188  /* storage_proof/mod.rs:23: a node is at most 752 bytes (4 bytes per felt) */
typedef struct qpw_circuit_inputs {
  /* public */
  uint8_t funding_amount_le[16];       /* u128, little-endian bytes */
  uint8_t nullifier[32], root_hash[32], exit_account[32];
  /* private */
  uint8_t secret[32];
  uint64_t transfer_count;
  uint8_t funding_account[32], unspendable_account[32];
  uint32_t proof_len, reserved;        /* ProcessedStorageProof: nodes + the hex-digit index of the child hash in each node */
  const uint8_t* const* proof_nodes; const uint32_t* proof_node_lens; const uint32_t* proof_indices;
} qpw_circuit_inputs;

/* commit(): the input-target values of circuit `kind` (QPC_WORMHOLE or one of its fragments) from the inputs; *len = count written.
 * Errors as the reference's: non-canonical digest chunk, proof longer than 20 nodes, node larger than 188 felts. */
int qpw_fill_targets(uint32_t kind, const qpw_circuit_inputs* in, uint64_t* inputs, size_t cap, size_t* len);
/* Nullifier::from_preimage (nullifier.rs:53-73): H(H("~nullif~" ‖ secret ‖ transfer_count)) as digest bytes. */
int qpw_nullifier_from_preimage(const uint8_t* secret, size_t secret_len, uint64_t transfer_count, uint8_t out[32]);
/* UnspendableAccount::from_secret (unspendable_account.rs:38-63): H(H("wormhole" ‖ secret)). */
int qpw_unspendable_account_from_secret(const uint8_t secret[32], uint8_t out[32]);
/* PublicCircuitInputs::try_from_slice (inputs.rs:95-134): 16 felts -> the four public values; funding-amount limbs must be 32-bit. */
typedef struct qpw_public_inputs { uint8_t funding_amount_le[16]; uint8_t nullifier[32], root_hash[32], exit_account[32]; } qpw_public_inputs;
int qpw_public_inputs_from_felts(const uint64_t* pis, size_t count, qpw_public_inputs* out);

/* ---- vote inputs (voting/src/lib.rs:23-52): digests as four canonical felts. */
#define QPV_MAX_MERKLE_DEPTH 32
typedef struct qpv_vote_inputs {
  uint64_t proposal_id[4], merkle_root[4]; uint32_t vote, reserved; uint64_t nullifier[4];
  uint64_t private_key[4];
  const uint64_t* merkle_siblings;     /* [num_siblings][4] */
  const uint8_t* path_indices;         /* [num_path_indices] 0 = left, 1 = right */
  uint32_t num_siblings, num_path_indices, actual_merkle_depth, reserved2;
} qpv_vote_inputs;
int qpv_fill_targets(const qpv_vote_inputs* in, uint64_t* inputs, size_t cap, size_t* len);

/* ---- aggregation (wormhole/aggregator/src/circuits/tree.rs:106-143, aggregate_chunk): the circuit of one tree node — for each of
 * `num_proofs` children of the circuit `inner` (a circuit of this library — a leaf circuit or another node — under standard_recursion(_zk)_config), verify_proof
 * in-circuit against verifier data given as (private) inputs, and the children's public inputs, concatenated, as the node's public inputs.
 * The node circuit takes the inner circuit's config (tree.rs:111): zero-knowledge children give a zero-knowledge node.
 * The in-circuit verifier is this repo's own (zk-circuits_amd/circuit/recursion.cc), built on the leaf gate set plus upstream's recursion
 * gates (ArithmeticExtension, Reducing, ReducingExtension, RandomAccess, CosetInterpolation, PoseidonMds: 8.8 k rows for two zk leaf proofs);
 * its layout is unpinned like every circuit here. */
int qpc_aggregation_circuit_build(const qpgpu_circuit_desc* inner, uint32_t num_proofs, qpc_circuit** out);
/* The same with the node's own zero-knowledge setting: 2 = as the children (what the function above and the reference do), 0 / 1 = off / on.
 * A node's witness is made of public data only (its children's proofs and their verifier data), so a node without blinding rows gives nothing
 * away about the leaves' witnesses — and has half the rows (2^14 instead of 2^15 for two zk children). */
int qpc_aggregation_circuit_build_ex(const qpgpu_circuit_desc* inner, uint32_t num_proofs, uint32_t zero_knowledge, qpc_circuit** out);
/* set_verifier_data_target + set_proof_with_pis_target: the node circuit's input-target values from the inner circuit's verifier data
 * (constants_sigmas_cap [2^cap_height][4], circuit_digest) and `count` proofs (ProofWithPublicInputs::to_bytes, `stride` bytes apart).
 * *len = count of values; inputs may be NULL to query it. A proof that does not verify fails in qpc_generate_witness. */
int qpc_aggregation_fill_targets(const qpc_circuit* c, const uint64_t* constants_sigmas_cap, const uint64_t circuit_digest[4], const uint8_t* proofs, size_t stride,
                                 size_t proof_len, uint32_t count, uint64_t* inputs, size_t cap, size_t* len);

/* PoseidonHash::hash_no_pad on the host (the hash all of the above use; csrc/poseidon.h). */
void qpc_hash_no_pad(const uint64_t* felts, size_t count, uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
