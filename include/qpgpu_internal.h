/* qpgpu_internal — entry points of libqpgpu.so that are NOT part of the drop-in ABI (include/qpgpu.h).
 *
 * Three groups, all C-ABI and all optional for an integrator:
 *   1. transcript replay: the prover's own Fiat-Shamir schedule run over finished proof bytes (host only);
 *   2. per-kernel-group profiling counters that bench.py turns into its `roofline` object;
 *   3. per-kernel test hooks (host buffers in, host buffers out) that tests/ use to compare every HIP kernel with its
 *      oracle twin and with the vectors the reference holds (SURVEY.md §4 implication (4), §8(c)).
 * tests/test_abi.py checks that the library exports exactly qpgpu.h + this header.
 */
#ifndef QPGPU_INTERNAL_H
#define QPGPU_INTERNAL_H
#include "qpgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- 1. transcript replay. Needs only the CommonCircuitData shape + circuit_digest of `desc`
 * (gates / constants_sigmas may be NULL). out: pi_hash[4] | betas[nch] | gammas[nch] | alphas[nch] | zeta[2] | fri_alpha[2] |
 * fri_betas[2*layers] | pow_response | query indices[num_queries]. Returns the number of words written or a negative status. */
int qpgpu_replay_transcript(const qpgpu_circuit_desc* desc, const uint8_t* proof, size_t len, uint64_t* out, size_t out_cap);

/* Host half of qpgpu_verify_batch on ONE proof (transcript, proof of work, vanishing identity, FRI arithmetic, final polynomial;
 * no Merkle openings, no device): 0 = these checks pass, QPGPU_ERR_VERIFY otherwise (reason in qpgpu_last_error). */
int qpgpu_test_verify_host(const qpgpu_circuit_desc* desc, const uint8_t* proof, size_t len);

/* ---- 2. profiling: HIP-event timers around kernel groups on the proving stream. mode 0 off, 1 dominant group only, 2 all groups. */
void qpgpu_profile_enable(int mode);
void qpgpu_profile_reset(void);
int qpgpu_profile_count(void);
const char* qpgpu_profile_name(int id);
int qpgpu_profile_get(int id, uint64_t* launches, double* ms, double* alg_bytes);
/* stage boundaries of the calling thread's last proof when QPGPU_TIMING=1 was set at context creation */
int qpgpu_last_stage_ms(double* out, int cap);

/* Host cores the library sizes its thread pools and hand-off wait mode for (csrc/hostcpus.h): affinity mask, capped by the cgroup CPU quota,
 * divided by LOCAL_WORLD_SIZE; re-evaluated on every call (the library itself caches its first reading). */
int qpgpu_usable_host_cpus(void);
/* Hand-offs (device -> host transcript) of this process that fell back from the publish kernel + flag to copy + stream synchronise
 * because the flag did not arrive. Expected 0; bench.py prints it. */
uint64_t qpgpu_handoff_fallbacks(void);

/* ---- 3. kernel test hooks */
int qpgpu_test_permute(uint64_t* states /*[count][12]*/, size_t count);
int qpgpu_test_permute_form(uint64_t* states /*[count][12]*/, size_t count, int form /* 0 literal, 1 fast partial rounds, -1 default */);
/* mode 0 forward DIF, 1 forward DIT, 2 inverse DIF, 3 inverse DIT (unscaled); data [count][2^log_n] in place */
int qpgpu_test_ntt(uint64_t* data, int count, int log_n, int mode);
int qpgpu_test_lde(const uint64_t* coeffs, int count, int log_n, int log_cosets, uint64_t shift, uint64_t* out);
int qpgpu_test_intt_natural(const uint64_t* vals, int count, int log_n, uint64_t* coeffs);
int qpgpu_test_bitrev(const uint64_t* in, int count, int log_n, uint64_t scale, uint64_t* out);
int qpgpu_test_merkle(const uint64_t* cols, int width, size_t num_leaves, int cap_height, uint64_t* digests);
/* one kernel form at a time (golden-vector replay): see csrc/hooks.hip for the mode lists */
int qpgpu_test_leaf_hash(const uint64_t* cols, int width, size_t num_leaves, int mode, uint64_t* digests);
int qpgpu_test_two_to_one(const uint64_t* pairs /*[count][8]*/, size_t count, int mode, uint64_t* out /*[count][4]*/);
int qpgpu_test_fri_leaf_hash(const uint64_t* a, const uint64_t* b, int arity_bits, size_t num_leaves, int mode, uint64_t* digests);
int qpgpu_test_calib_colsum(int width, size_t n, int iters);
double qpgpu_test_host_permute_ns(int iters);
double qpgpu_test_sync_roundtrip_us(int iters);
double qpgpu_test_sync_roundtrip_us2(int iters, int mode);
double qpgpu_bench_permute(size_t count, int iters);
double qpgpu_bench_permute_form(size_t count, int iters, int form /* as qpgpu_test_permute_form */);
double qpgpu_bench_lde(int count, int log_n, int iters, int force_generic);

#ifdef __cplusplus
}
#endif
#endif
