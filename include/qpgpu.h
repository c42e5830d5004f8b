/* qpgpu — C ABI of the MI355X (gfx950) Plonky2 prover backend for the Quantus wormhole / voting circuits.
 *
 * This is the drop-in seam for the reference's hot path: the body of
 *   ProverCircuitData::prove(PartialWitness)  (qp-plonky2 1.1.1, reached from
 *   WormholeProver::prove  — /root/reference/wormhole/prover/src/lib.rs:233-237 — and from
 *   aggregate_chunk's circuit_data.prove(pw) — wormhole/aggregator/src/circuits/tree.rs:136)
 * after witness generation: `prove_with_partition_witness` (stages 1-10 of SURVEY.md §3.3).
 * A `gpu` feature in a qp-plonky2 fork binds these symbols with bindgen; the wrappers
 * in the reference repo need no source change.
 *
 * Conventions: all integers little-endian; field elements are canonical Goldilocks u64 (< 2^64-2^32+1);
 * matrices are column-major ([column][row], one polynomial per contiguous run of n values, natural row order).
 * Every function returns 0 on success or a negative qpgpu_status; the text is in qpgpu_last_error()
 * (thread-local). Nothing throws or aborts across this boundary.
 */
#ifndef QPGPU_H
#define QPGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum qpgpu_status {
  QPGPU_OK = 0,
  QPGPU_ERR_ARG = -1,        /* bad argument / unsupported circuit shape or gate */
  QPGPU_ERR_NO_DEVICE = -2,  /* no gfx950 device / HIP runtime failure at init */
  QPGPU_ERR_HIP = -3,        /* a HIP call failed */
  QPGPU_ERR_OOM = -4,
  QPGPU_ERR_BUFFER = -5,     /* output buffer too small */
  QPGPU_ERR_ZETA = -6,       /* zeta^n == 1 (upstream's "opening point is in the subgroup" exit) */
  QPGPU_ERR_POW = -7,        /* supplied pow_witness invalid, or grinding found nothing */
  QPGPU_ERR_WITNESS = -8,    /* non-canonical field element (>= p) in the wire matrix, the public inputs or explicit salts */
  QPGPU_ERR_VERIFY = -9      /* qpgpu_verify_batch: a proof was rejected (which and why: qpgpu_last_error) */
} qpgpu_status;

/* Gate tags = qp-plonky2's default gate-serializer ids as they appear in CommonCircuitData
 * (wormhole/bench-data/common.bin). */
enum { QPGPU_GATE_ARITHMETIC = 0, QPGPU_GATE_ARITHMETIC_EXTENSION = 1, QPGPU_GATE_BASESUM2 = 2, QPGPU_GATE_CONSTANT = 3,
       QPGPU_GATE_COSET_INTERPOLATION = 4, QPGPU_GATE_EXPONENTIATION = 5, QPGPU_GATE_MUL_EXTENSION = 8, QPGPU_GATE_NOOP = 9,
       QPGPU_GATE_POSEIDON_MDS = 10, QPGPU_GATE_POSEIDON = 11, QPGPU_GATE_PUBLIC_INPUT = 12, QPGPU_GATE_RANDOM_ACCESS = 13,
       QPGPU_GATE_REDUCING_EXTENSION = 14, QPGPU_GATE_REDUCING = 15 };

/* One entry of CommonCircuitData.gates. `param` by tag: Constant num_consts; BaseSum<2> num_limbs; Arithmetic, ArithmeticExtension,
 * MulExtension num_ops; Reducing, ReducingExtension num_coeffs; Exponentiation num_power_bits; RandomAccess bits (param2 =
 * num_copies, param3 = num_extra_constants); CosetInterpolation subgroup_bits (param2 = degree, table = the gate's
 * barycentric_weights[2^subgroup_bits], table_len <= 16). The leaf circuit (prove()) uses tags {0,2,3,9,11,12}; the others are the
 * recursive-verifier circuit's gates behind aggregate() (wormhole/aggregator/src/circuits/tree.rs:106-143). */
typedef struct qpgpu_gate { uint32_t tag; uint32_t param2; uint64_t param; uint32_t param3; uint32_t table_len; const uint64_t* table; } qpgpu_gate;

/* CommonCircuitData + ProverOnlyCircuitData content the device path needs (what the reference loads with
 * WormholeProver::new_from_bytes — wormhole/prover/src/lib.rs:105-137). */
typedef struct qpgpu_circuit_desc {
  uint32_t degree_bits, num_wires, num_routed, num_constants;      /* num_constants counts selector columns too */
  uint32_t num_challenges, quotient_degree_factor, num_partial_products, num_public_inputs;
  uint32_t num_gate_constraints, zero_knowledge;
  uint32_t fri_rate_bits, fri_cap_height, fri_num_queries, fri_pow_bits;
  uint32_t fri_num_layers; uint32_t fri_arity_bits[8];
  uint32_t num_gates, num_selector_groups;
  const qpgpu_gate* gates;                 /* [num_gates] */
  const uint32_t* selector_indices;        /* [num_gates] */
  const uint32_t* selector_group_ranges;   /* [num_selector_groups][2] = (lo, hi) */
  const uint64_t* k_is;                    /* [num_routed] */
  const uint64_t* constants_sigmas;        /* [num_constants + num_routed][n], values on the trace subgroup */
  uint64_t circuit_digest[4];
} qpgpu_circuit_desc;

typedef struct qpgpu_ctx qpgpu_ctx;
typedef struct qpgpu_circuit qpgpu_circuit;

/* Zero-knowledge circuits need prover-side randomness for the 4 salt columns of each blinded oracle (qp-plonky2 draws
 * F::rand_vec from OsRng). It is an INPUT here, in one of two forms:
 *   salts     explicit [3 oracles][4][N] leaf-order columns of canonical field elements, or
 *   salt_key  32 secret random bytes (from the OS RNG); the device expands them with the ChaCha8 keystream
 *             (key = salt_key, nonce = salt_stream, block counter = element index / 8; csrc/salt.h), one-way in the key, so the
 *             salts a FRI query opens say nothing about the unopened ones. Use a fresh key or a fresh salt_stream per proof.
 * A zero-knowledge circuit with neither is refused with QPGPU_ERR_ARG; non-zk circuits ignore both. */
typedef struct qpgpu_prove_opts {
  const uint64_t* salts;
  const uint8_t* salt_key;    /* [32] */
  uint64_t salt_stream;
  uint64_t pow_witness;       /* used when have_pow_witness != 0; otherwise the smallest valid witness is ground */
  uint32_t have_pow_witness;
  uint32_t wires_on_device;   /* wires pointer(s) are device pointers (already resident in HBM) */
  void* stream;               /* hipStream_t to enqueue on, or NULL for the context's own stream */
} qpgpu_prove_opts;

const char* qpgpu_last_error(void);
int qpgpu_device_count(void);

int qpgpu_ctx_create(int device, qpgpu_ctx** out);
void qpgpu_ctx_destroy(qpgpu_ctx* ctx);

/* Parses qp-plonky2's CommonCircuitData::to_bytes() (the `common.bin` the reference writes at
 * wormhole/circuit-builder/src/lib.rs:36-63 and loads at wormhole/prover/src/lib.rs:105-187; layout: SURVEY.md App. B.1) into
 * a circuit descriptor. The gate list / selector arrays / k_is are written into caller-provided arrays (capacities given);
 * constants_sigmas and circuit_digest are NOT in common.bin (they live in prover.bin / verifier.bin) and stay untouched.
 * Returns 0, or QPGPU_ERR_ARG for malformed / unsupported input (lookups, gates outside the leaf set). */
int qpgpu_parse_common(const uint8_t* bytes, size_t len, qpgpu_circuit_desc* desc, qpgpu_gate* gates, size_t gates_cap,
                       uint32_t* selector_indices, uint32_t* selector_group_ranges, size_t groups_cap, uint64_t* k_is, size_t k_is_cap);
/* VerifierOnlyCircuitData::to_bytes() (first 8 + 2^cap_height*32 + 32 bytes of `verifier.bin`): cap and circuit digest. */
int qpgpu_parse_verifier_only(const uint8_t* bytes, size_t len, uint64_t* cap_out, size_t cap_words, uint64_t digest_out[4]);

/* Writers, the inverse of the two parsers (what wormhole/circuit-builder/src/lib.rs:36-51 does with to_bytes()): byte-stable on the
 * reference's bench-data/common.bin and verifier.bin. The CircuitConfig fields a descriptor does not carry are written with
 * standard_recursion(_zk)_config's values (config.num_constants 2, security_bits 100, use_base_arithmetic_gate, FRI reduction strategy
 * ConstantArityBits(4,5)). out == NULL queries the size (*len). */
int qpgpu_write_common(const qpgpu_circuit_desc* desc, uint8_t* out, size_t cap, size_t* len);
int qpgpu_write_verifier_only(const uint64_t* constants_sigmas_cap, uint32_t cap_height, const uint64_t circuit_digest[4],
                              uint8_t* out, size_t cap, size_t* len);
/* Circuit package: one blob with everything qpgpu_circuit_create needs — the common data in plonky2's byte format, the circuit digest and
 * the constants/sigmas values — checksummed. It stands where prover.bin stands in the reference (wormhole/circuit-builder/src/lib.rs:53-63,
 * loaded at wormhole/prover/src/lib.rs:105-137): ProverOnlyCircuitData's own layout is pinned by nothing the reference ships, and these
 * values are all this backend needs from it. The descriptor of a read package points into the package and lives as long as it does. */
typedef struct qpgpu_package qpgpu_package;
int qpgpu_package_write(const qpgpu_circuit_desc* desc, uint8_t* out, size_t cap, size_t* len);
int qpgpu_package_read(const uint8_t* bytes, size_t len, qpgpu_package** out);
const qpgpu_circuit_desc* qpgpu_package_desc(const qpgpu_package* pkg);
void qpgpu_package_free(qpgpu_package* pkg);

/* Uploads the preprocessed columns, computes the constants/sigmas LDE + Merkle tree once and keeps it resident
 * (the work `builder.build_prover()` does at wormhole/circuit/src/circuit.rs:102-104). The handle is immutable and
 * may be shared by threads. */
int qpgpu_circuit_create(qpgpu_ctx* ctx, const qpgpu_circuit_desc* desc, qpgpu_circuit** out);
void qpgpu_circuit_destroy(qpgpu_circuit* c);
/* constants_sigmas Merkle cap (2^cap_height digests x 4 u64) — the VerifierOnlyCircuitData content. */
int qpgpu_circuit_constants_sigmas_cap(const qpgpu_circuit* c, uint64_t* out_cap);
/* Exact size of ProofWithPublicInputs::to_bytes() for this circuit. */
size_t qpgpu_proof_size(const qpgpu_circuit* c);

/* prove_with_partition_witness: full witness matrix -> ProofWithPublicInputs::to_bytes().
 * wires: [num_wires][n] column-major. Re-entrant: each call uses its own stream + workspace; when several calls are in flight in the
 * process (the reference's rayon workers, tree.rs:100) they run with qpgpu_prove_batch's kernel choices. Same bytes either way. */
int qpgpu_prove(qpgpu_circuit* c, const uint64_t* wires, const uint64_t* public_inputs, size_t num_public_inputs,
                const qpgpu_prove_opts* opts, uint8_t* out, size_t out_cap, size_t* out_len);

/* `batch` independent proofs of the same circuit, stream-pipelined on one GPU (BASELINE config #3; the unit the
 * aggregator shards per GPU, wormhole/aggregator/src/circuits/tree.rs:92-103). wires[i], public_inputs[i],
 * out + i*out_stride. opts->salt_stream / salts apply per proof as salt_stream + i / salts + i*3*4*N. */
int qpgpu_prove_batch(qpgpu_circuit* c, size_t batch, const uint64_t* const* wires, const uint64_t* const* public_inputs,
                      size_t num_public_inputs, const qpgpu_prove_opts* opts, uint8_t* out, size_t out_stride,
                      size_t* out_lens);

/* ---- Verifier: VerifierCircuitData::verify as reached from WormholeVerifier::verify (wormhole/verifier/src/lib.rs:155-159), for a
 * batch of proofs of one circuit (the aggregator takes in up to 512 leaf proofs, wormhole/aggregator/src/circuits/tree.rs:92-103).
 * desc: the CommonCircuitData shape + circuit_digest (constants_sigmas is not needed and may be NULL); constants_sigmas_cap: the
 * VerifierOnlyCircuitData cap, [2^cap_height][4] (qpgpu_parse_verifier_only reads both from verifier.bin). Same gate set as the prover.
 * Every Merkle opening is hashed on the device, the transcript and the field arithmetic run on host worker threads. */
typedef struct qpgpu_verifier qpgpu_verifier;
int qpgpu_verifier_create(qpgpu_ctx* ctx, const qpgpu_circuit_desc* desc, const uint64_t* constants_sigmas_cap, qpgpu_verifier** out);
void qpgpu_verifier_destroy(qpgpu_verifier* v);
size_t qpgpu_verifier_proof_size(const qpgpu_verifier* v);
/* proofs + i*proof_stride, each proof_len bytes of ProofWithPublicInputs::to_bytes(). results[i] = 0 (accepted) or QPGPU_ERR_VERIFY.
 * Returns 0 if every proof was accepted, QPGPU_ERR_VERIFY if any was rejected, another negative status if the batch could not run. */
int qpgpu_verify_batch(qpgpu_verifier* v, const uint8_t* proofs, size_t proof_stride, size_t proof_len, size_t count, int32_t* results);

/* Page-locked host memory for witness matrices. qpgpu_prove / qpgpu_prove_batch take the wire matrix from any host memory; from pageable
 * memory the runtime stages the 17.7 MB (leaf) / 35 MB (tree node) through its own bounce buffers on the calling thread, from memory
 * obtained here the copy is one DMA transfer that overlaps the other streams' kernels (the reference keeps its witness in a Vec: this is
 * the allocation a host that generates witnesses into a reusable buffer should use instead). Free with qpgpu_host_free. */
int qpgpu_host_alloc(size_t bytes, void** out);
void qpgpu_host_free(void* p);

/* ---- The exchange of the sharded path: independent proofs are proved one-per-GPU with no data-path collective; what
 * precedes the aggregator step is an all-gather of the fixed-size proof blobs over RCCL / xGMI (the reference hands proofs between rayon
 * workers in shared memory, wormhole/aggregator/src/circuits/tree.rs:92-103). One communicator per rank (= per GPU, per process):
 *   rank 0 calls qpgpu_comm_unique_id and hands the 128 bytes to every rank out of band (a file, a socket, the launcher's store);
 *   every rank calls qpgpu_comm_create(ctx of its GPU, rank, world, id) — collective: it returns when all ranks have joined;
 *   qpgpu_gather_proofs: every rank passes `count` blobs of `proof_size` bytes (the same on every rank) and receives all of them in
 *   rank order, out[(r * count + i) * proof_size]. Host pointers in and out; the call blocks until the data has arrived.
 * librccl is opened at run time next to the HIP runtime this library uses (QPGPU_RCCL_LIB overrides the path); without it these
 * calls return QPGPU_ERR_NO_DEVICE and everything else works. A communicator serialises its calls; use one per thread that gathers. */
#define QPGPU_COMM_ID_BYTES 128
typedef struct qpgpu_comm qpgpu_comm;
int qpgpu_comm_unique_id(uint8_t id[QPGPU_COMM_ID_BYTES]);
int qpgpu_comm_create(qpgpu_ctx* ctx, int rank, int world, const uint8_t id[QPGPU_COMM_ID_BYTES], qpgpu_comm** out);
void qpgpu_comm_destroy(qpgpu_comm* comm);
int qpgpu_comm_rank(const qpgpu_comm* comm);
int qpgpu_comm_world(const qpgpu_comm* comm);
int qpgpu_gather_proofs(qpgpu_comm* comm, const uint8_t* local, size_t count, size_t proof_size, uint8_t* out);
/* Version code of the RCCL library in use (ncclGetVersion), 0 if none could be opened. */
int qpgpu_comm_rccl_version(void);

#ifdef __cplusplus
}
#endif
#endif
