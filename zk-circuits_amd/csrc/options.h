// Tuning / diagnostic knobs of the library. Every QPGPU_* environment variable is read in ONE place (options_refresh_from_env, called
// by qpgpu_ctx_create and once at library load) into this struct; the kernels' launchers and the orchestration only read the struct.
// The knobs are process-wide: a context created later re-reads the environment for everything created after it.
#pragma once
namespace qp {
struct Options {
  int streams = 4;            // QPGPU_STREAMS: worker threads (one HIP stream + workspace each) of qpgpu_prove_batch; default 8 when
                              //   GPU_MAX_HW_QUEUES >= 8 (each worker needs hardware queues of its own to overlap), else 4
  int handoff_spin_us = -1;   // QPGPU_HANDOFF_SPIN_US: host busy-wait budget per device->host hand-off before it naps between polls
                              //   (-1: 2000 us; 50 us inside qpgpu_prove_batch workers when the process has fewer than two usable host cores per worker)
  int no_publish = 0;         // QPGPU_NO_PUBLISH=1: hipMemcpyAsync + hipStreamSynchronize instead of the publish kernel + mapped flag
  int timing = 0;             // QPGPU_TIMING=1: stage boundaries recorded with HIP events (qpgpu_last_stage_ms)
  int split_quotient = -1;    // QPGPU_SPLIT_QUOTIENT: -1 auto (split form for single proofs of leaf circuits), 0 fused kernel only, 1 split
  int split_fork_late = 1;    // QPGPU_SPLIT_FORK_LATE=0: fork the side-stream quotient work before the leaf hash instead of after it
  int split_serial = 0;       // QPGPU_SPLIT_SERIAL=1: the three split-quotient kernels back to back on the main stream (diagnostic)
  int quotient_halves = 0;    // QPGPU_QUOTIENT_HALVES=1: fused kernel's two halves on two streams (diagnostic)
  int shared_mode = 0;        // QPGPU_SHARED_MODE=1: batch workers pick the kernels tuned for a shared chip
  int l0_table = 1;           // QPGPU_L0_TABLE=0: recompute L_0(x) per point instead of the per-circuit table
  int pair_leaves = 0;        // QPGPU_PAIR_LEAVES=1: 2-lane pair form for leaf hashing of small trees
  int pair_max = -1;          // QPGPU_PAIR_MAX: largest level (parents) hashed in the pair form (-1: 2^18)
  int coop_max = -1;          // QPGPU_COOP_MAX: largest level (parents) hashed in the 16-lane form (-1: 16384 for a single proof, 4096 inside qpgpu_prove_batch)
  int fuse_levels = 1;        // QPGPU_FUSE_LEVELS=0: one launch per latency-bound tree level
  int ntt16 = 1;              // QPGPU_NTT16: 0 off, 1 sizes <= 2^12, 2 every supported size
  int ntt_strided = 1;        // QPGPU_NTT_STRIDED=0: tiled multi-pass NTT only
  int ntt_m8 = 1;             // QPGPU_NTT_M8=0: radix-16 register passes instead of radix-256 LDS passes
  int verify_coop_max = 16384; // QPGPU_VERIFY_COOP_MAX: most Merkle openings per launch checked in the 16-lane form (0: always one lane per opening)
  int poseidon_fast = -1;     // QPGPU_POSEIDON_FAST: form of the permutation's partial rounds in the leaf-hash kernels: -1 auto (fast where the chip is
                              //   full: trees of >= 2^17 leaves and batches), 0 literal everywhere, 1 fast everywhere (same digests either way)
  int quotient_shared_loads = 1; // QPGPU_QUOTIENT_SHARED_LOADS=0: the fused quotient kernel loads every gate's wires on its own also for the leaf gate set
  int pow_first_shift = 0;    // QPGPU_POW_FIRST_SHIFT: the first proof-of-work launch tests 2^(pow_bits + shift) candidates (same proof bytes: the smallest witness wins)
};
const Options& opts();
void options_refresh_from_env();
}
