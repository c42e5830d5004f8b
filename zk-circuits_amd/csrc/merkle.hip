// Batched Poseidon hashing for the Merkle-cap commitments (MerkleTree::new + PoseidonHash::
// {hash_or_noop, two_to_one} of qp-plonky2, behind wormhole/prover/src/lib.rs:233-237).
// One permutation per lane, state in VGPRs; leaves stream column-major oracle values (coalesced across lanes).
#include "kernels.h"
#include "poseidon_dev.h"
#include "salt.h"
#include "options.h"
#include <cstdlib>

namespace qp {

// one ChaCha8 block (eight salts) per thread
__global__ void k_salt_fill(u64* cols, size_t N, SaltKey key, u64 nonce, int oracle) {
  size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (b >= N / 2) return;                                   // 4N elements = N/2 blocks
  u64 o[8]; salt_block(key, nonce, (u64)oracle * (N / 2) + b, o);
#pragma unroll
  for (int k = 0; k < 8; k++) cols[b * 8 + k] = o[k];
}
void salt_fill(u64* cols, size_t N, const SaltKey& key, u64 nonce, int oracle, hipStream_t s) {
  hipLaunchKernelGGL(k_salt_fill, dim3((unsigned)((N / 2 + 255) / 256)), dim3(256), 0, s, cols, N, key, nonce, oracle);
}
__global__ void __launch_bounds__(256) k_check_canonical(const u64* data, size_t count, u64* flag) {
  bool bad = false;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) bad |= data[i] >= GL_P;
  if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr((unsigned long long*)flag, 1ULL);
}
void check_canonical(const u64* data, size_t count, u64* flag, hipStream_t s) {
  if (!count) return;
  size_t blocks = (count + 256 * 8 - 1) / (256 * 8); if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(k_check_canonical, dim3((unsigned)blocks), dim3(256), 0, s, data, count, flag);
}

__global__ void __launch_bounds__(256) k_publish(const u64* src, u64* dst, size_t words, u64* flag, u64* counter, u64 seq, u64* extra, int clear_extra) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound stage: win VALU arbitration against co-resident bulk hashing waves
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
  if (extra && blockIdx.x == 0 && threadIdx.x == 0) { dst[words] = *extra; if (clear_extra) *extra = 0; }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    bool last = true;
    if (gridDim.x > 1) {
      unsigned long long prev = atomicAdd((unsigned long long*)counter, 1ULL);
      last = prev + 1 == gridDim.x;
      if (last) *counter = 0;                     // publishes on one workspace are stream-ordered, so the reset cannot race
    }
    if (last) { __threadfence_system(); *(volatile u64*)flag = seq; }
  }
}
void publish_to_host(const u64* src, u64* dst_mapped, size_t words, u64* flag_mapped, u64* counter_dev, u64 seq, hipStream_t s, u64* extra, bool clear_extra) {
  unsigned blocks = (unsigned)((words + 2047) / 2048); if (blocks < 1) blocks = 1; if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(k_publish, dim3(blocks), dim3(256), 0, s, src, dst_mapped, words, flag_mapped, counter_dev, seq, extra, clear_extra ? 1 : 0);
}

// HALF: only the even lanes of every wave work (experiment: does a half-empty wave64 issue faster? it does not — profiles/r02 notes)
template <bool FAST, bool HALF = false>
__global__ void k_permute_batch(u64* states, size_t count) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (HALF) { if (i & 1) return; i >>= 1; }
  if (i >= count) return;
  u64 st[12];
#pragma unroll
  for (int k = 0; k < 12; k++) st[k] = states[i * 12 + k];
  poseidon_permute_dev_t<FAST>(st);
#pragma unroll
  for (int k = 0; k < 12; k++) states[i * 12 + k] = st[k];
}
// form: 0 literal partial rounds, 1 fast partial rounds, -1 the library's choice for a full chip (QPGPU_POSEIDON_FAST=0 forces the literal form)
void poseidon_permute_batch(u64* states, size_t count, hipStream_t s, int form) {
  if (!count) return;
  if (form < 0) form = opts().poseidon_fast != 0;
  if (form == 2) { hipLaunchKernelGGL((k_permute_batch<false, true>), dim3((unsigned)((2 * count + 255) / 256)), dim3(256), 0, s, states, count); return; }
  if (form == 3) { hipLaunchKernelGGL((k_permute_batch<true, true>), dim3((unsigned)((2 * count + 255) / 256)), dim3(256), 0, s, states, count); return; }
  if (form) hipLaunchKernelGGL(k_permute_batch<true>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, states, count);
  else hipLaunchKernelGGL(k_permute_batch<false>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, states, count);
}

// FUSE: the block (256 consecutive leaves) also produces its 128 level-1 and 64 level-2 nodes, saving two launches whose
// cost is pure permutation latency. It only pays when the grid is several times larger than what is resident at once
// (measured on MI355X: at N = 2^17 all 512 blocks run in lockstep, so the fused tail costs two more permutation
// latencies at half/quarter occupancy and the leaf+levels total gets 5 % WORSE; profiles/r01 notes). Enabled from 2^20 leaves.
static constexpr size_t PAIR_MAX_LEAVES = size_t(1) << 18;   // <= 4 waves/SIMD per-lane: the pair form fills the chip better
// Occupancy floor of the fast-form kernels (tools/ab_leaf_waves.sh on MI355X, same box; V = 2^23 leaves launch average / single W proof): floor
// 4 waves/SIMD 27.64 ms / 273.0 proofs/s, 5: 27.58 / 270.8, 6: 28.09 / 268.6 (the tighter budgets spill into scratch). With the floor at 4 the
// compiler settles at 81-100 VGPRs without scratch, i.e. 5-6 resident waves. The literal form is left to the compiler's own choice.
#ifndef QPGPU_LEAF_WAVES_FUSED
#define QPGPU_LEAF_WAVES_FUSED 4
#endif
#ifndef QPGPU_LEAF_WAVES
#define QPGPU_LEAF_WAVES 4
#endif
template <bool FUSE, bool FAST>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(FAST ? (FUSE ? QPGPU_LEAF_WAVES_FUSED : QPGPU_LEAF_WAVES) : 1, 8))) k_leaves_colmajor(const u64* cols, size_t col_stride, int width, size_t num_leaves, u64* digests,
                                                         u64* level1, u64* level2) {
  __shared__ u64 sm[FUSE ? 256 * 4 : 4];
  size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (!FUSE && L >= num_leaves) return;
  u64 st[12];
#pragma unroll
  for (int k = 0; k < 12; k++) st[k] = 0;
  if (width <= 4) {
    for (int k = 0; k < width; k++) st[k] = cols[(size_t)k * col_stride + L];
  } else {
    if constexpr (!FUSE) {
      // software-pipelined absorb: the next chunk's eight column loads are in flight while the current chunk is permuted (the loop
      // otherwise exposes one HBM round trip per chunk: 18 for the wires oracle). Only for trees that do not fill the chip: with
      // many blocks per SIMD queued (the FUSE variant, >= 2^20 leaves) other waves already hide the loads and the 24 extra
      // VGPRs cost an occupancy step (measured 31.3 vs 30.0 ms per launch at 2^23 leaves).
      u64 nxt[8];
#pragma unroll
      for (int k = 0; k < 8; k++) nxt[k] = k < width ? cols[(size_t)k * col_stride + L] : 0;
      for (int off = 0; off < width; off += 8) {
        const int m = width - off < 8 ? width - off : 8;
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < m) st[k] = nxt[k];
        const int off2 = off + 8;
#pragma unroll
        for (int k = 0; k < 8; k++) if (off2 + k < width) nxt[k] = cols[(size_t)(off2 + k) * col_stride + L];
        poseidon_permute_dev_t<FAST>(st);
      }
    } else {
      for (int off = 0; off < width; off += 8) {
        const int m = width - off < 8 ? width - off : 8;
        if (m == 8) {
#pragma unroll
          for (int k = 0; k < 8; k++) st[k] = cols[(size_t)(off + k) * col_stride + L];
        } else {
#pragma unroll
          for (int k = 0; k < 8; k++) if (k < m) st[k] = cols[(size_t)(off + k) * col_stride + L];
        }
        poseidon_permute_dev_t<FAST>(st);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) digests[L * 4 + k] = st[k];
  if (FUSE) {
    const unsigned t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; k++) sm[t * 4 + k] = st[k];
    __syncthreads();
    if (t < 128) {
#pragma unroll
      for (int k = 0; k < 8; k++) st[k] = sm[t * 8 + k];
#pragma unroll
      for (int k = 8; k < 12; k++) st[k] = 0;
      poseidon_permute_dev_t<FAST>(st);
      size_t node = (size_t)blockIdx.x * 128 + t;
#pragma unroll
      for (int k = 0; k < 4; k++) level1[node * 4 + k] = st[k];
    }
    __syncthreads();
    if (t < 128) {
#pragma unroll
      for (int k = 0; k < 4; k++) sm[t * 4 + k] = st[k];
    }
    __syncthreads();
    if (t < 64) {
#pragma unroll
      for (int k = 0; k < 8; k++) st[k] = sm[t * 8 + k];
#pragma unroll
      for (int k = 8; k < 12; k++) st[k] = 0;
      poseidon_permute_dev_t<FAST>(st);
      size_t node = (size_t)blockIdx.x * 64 + t;
#pragma unroll
      for (int k = 0; k < 4; k++) level2[node * 4 + k] = st[k];
    }
  }
}
// Pair form: two lanes per leaf (poseidon_dev.h). Used when the tree has too few leaves to fill the chip one-per-lane.
__global__ void __launch_bounds__(256) k_leaves_colmajor_pair(const u64* cols, size_t col_stride, int width, size_t num_leaves, u64* digests) {
  const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int p = (int)(gid & 1);
  size_t L = gid >> 1; const bool live = L < num_leaves; if (!live) L = num_leaves - 1;     // keep whole waves converged
  PairCtx cx; pair_init(cx, p);
  u64 own[6];
#pragma unroll
  for (int j = 0; j < 6; j++) own[j] = 0;
  if (width <= 4) {
#pragma unroll
    for (int j = 0; j < 2; j++) if (2 * j + p < width) own[j] = cols[(size_t)(2 * j + p) * col_stride + L];
  } else {
    for (int off = 0; off < width; off += 8) {
      const int m = width - off < 8 ? width - off : 8;
#pragma unroll
      for (int j = 0; j < 4; j++) if (2 * j + p < m) own[j] = cols[(size_t)(off + 2 * j + p) * col_stride + L];
      poseidon_permute_pair(own, cx);
    }
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < 2; j++) digests[L * 4 + 2 * j + p] = own[j];
  }
}
__global__ void __launch_bounds__(256) k_level_pair(const u64* in, u64* out, size_t parents) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound stage: win VALU arbitration against co-resident bulk hashing waves
  const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const int p = (int)(gid & 1);
  size_t node = gid >> 1; const bool live = node < parents; if (!live) node = parents - 1;
  PairCtx cx; pair_init(cx, p);
  u64 own[6];
#pragma unroll
  for (int j = 0; j < 4; j++) own[j] = in[node * 8 + 2 * j + p];
  own[4] = 0; own[5] = 0;
  poseidon_permute_pair(own, cx);
  if (live) {
#pragma unroll
    for (int j = 0; j < 2; j++) out[node * 4 + 2 * j + p] = own[j];
  }
}

// Returns the number of tree levels already produced above the leaves (0 or 2).
int merkle_leaves_colmajor(const u64* cols, size_t col_stride, int width, size_t num_leaves, int cap_height, u64* digests, hipStream_t s, bool chip_is_shared) {
  const bool fuse = (num_leaves % 256 == 0) && (num_leaves >> 2) >= (size_t(1) << cap_height) && num_leaves >= (size_t(1) << 20);
  dim3 grid((unsigned)((num_leaves + 255) / 256));
  // Form of the partial rounds (poseidon_dev.h): fast from 2 waves per SIMD up (2^17 leaves) and for every tree of a batch, whose other proofs
  // fill the chip; literal for a single proof's smaller trees. Same-box A/B (tools/ab_poseidon_form.sh, literal / fast, proofs/s): single proof of
  // 2^17-leaf trees 266.5 / 270.3, of 2^16-leaf trees 378.8 / 364.5; batches of 64: 375 / 399; 2^23 leaves (V): 7.27 / 7.80.
  const int pf = opts().poseidon_fast;
  const bool fast = pf > 0 || (pf < 0 && (chip_is_shared || num_leaves >= (size_t(1) << 17)));
  if (fuse) {
    if (fast) hipLaunchKernelGGL((k_leaves_colmajor<true, true>), grid, dim3(256), 0, s, cols, col_stride, width, num_leaves, digests, digests + num_leaves * 4,
                                 digests + (num_leaves + num_leaves / 2) * 4);
    else hipLaunchKernelGGL((k_leaves_colmajor<true, false>), grid, dim3(256), 0, s, cols, col_stride, width, num_leaves, digests, digests + num_leaves * 4,
                            digests + (num_leaves + num_leaves / 2) * 4);
    return 2;
  }
  // Measured (MI355X, 2^17 leaves): the pair form hashes leaves 9 % SLOWER than one permutation per lane (its per-lane stream is
  // not half: DPP moves, parity-selected constants and the one-sided partial rounds add ~20 % work), so it is only used for the
  // latency-bound tree levels below; QPGPU_PAIR_LEAVES=1 switches it on here for experiments.
  const bool pair_leaves = opts().pair_leaves != 0;
  if (pair_leaves && num_leaves <= PAIR_MAX_LEAVES && num_leaves >= 2) {
    hipLaunchKernelGGL(k_leaves_colmajor_pair, dim3((unsigned)((2 * num_leaves + 255) / 256)), dim3(256), 0, s, cols, col_stride, width, num_leaves, digests);
    return 0;
  }
  if (fast) hipLaunchKernelGGL((k_leaves_colmajor<false, true>), grid, dim3(256), 0, s, cols, col_stride, width, num_leaves, digests, (u64*)nullptr, (u64*)nullptr);
  else hipLaunchKernelGGL((k_leaves_colmajor<false, false>), grid, dim3(256), 0, s, cols, col_stride, width, num_leaves, digests, (u64*)nullptr, (u64*)nullptr);
  return 0;
}

__global__ void __launch_bounds__(256) k_leaves_fri(const u64* a, const u64* b, int arity_bits, size_t num_leaves, u64* digests) {
  size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (L >= num_leaves) return;
  const int arity = 1 << arity_bits, width = 2 * arity;
  const u64* pa = a + (L << arity_bits); const u64* pb = b + (L << arity_bits);
  u64 st[12];
#pragma unroll
  for (int k = 0; k < 12; k++) st[k] = 0;
  if (width <= 4) {
    for (int k = 0; k < arity; k++) { st[2 * k] = pa[k]; st[2 * k + 1] = pb[k]; }
  } else {
    for (int off = 0; off < width; off += 8) {      // width is a multiple of 8 for arity >= 4; arity 2 handled above
#pragma unroll
      for (int k = 0; k < 4; k++) { st[2 * k] = pa[off / 2 + k]; st[2 * k + 1] = pb[off / 2 + k]; }
      poseidon_permute_dev(st);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) digests[L * 4 + k] = st[k];
}
// cooperative form: 16 lanes per leaf (layers have few leaves, each a short sponge)
__global__ void __launch_bounds__(64) k_leaves_fri_coop(const u64* a, const u64* b, int arity_bits, size_t num_leaves, u64* digests) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound stage: win VALU arbitration against co-resident bulk hashing waves
  __shared__ u64 xch[4][24];
  const int g = threadIdx.x >> 4, i = threadIdx.x & 15;
  const size_t L = (size_t)blockIdx.x * 4 + g, LL = L < num_leaves ? L : num_leaves - 1;
  CoopCtx cx; coop_init(cx, i);
  const int width = 2 << arity_bits;
  const u64* pa = a + (LL << arity_bits); const u64* pb = b + (LL << arity_bits);
  u64 x = 0;
  if (width <= 4) { if (i < width) x = (i & 1) ? pb[i >> 1] : pa[i >> 1]; }
  else {
    for (int off = 0; off < width; off += 8) {
      if (i < 8) { int f = off + i; x = (f & 1) ? pb[f >> 1] : pa[f >> 1]; }
      x = coop_permute(x, i, cx, xch[g]);
    }
  }
  if (L < num_leaves && i < 4) digests[L * 4 + i] = x;
}
void merkle_leaves_fri(const u64* a, const u64* b, int arity_bits, size_t num_leaves, u64* digests, hipStream_t s) {
  if (num_leaves <= 16384) { hipLaunchKernelGGL(k_leaves_fri_coop, dim3((unsigned)((num_leaves + 3) / 4)), dim3(64), 0, s, a, b, arity_bits, num_leaves, digests); return; }
  hipLaunchKernelGGL(k_leaves_fri, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, a, b, arity_bits, num_leaves, digests);
}

// ---- upper levels. A level with many parents runs one permutation per lane; below COOP_MAX_PARENTS the stage is
// latency-bound (one per-lane permutation is ~26k dependent-issue instructions), so 16 lanes share each permutation.
__global__ void __launch_bounds__(256) k_level_lane(const u64* in, u64* out, size_t parents) {
  size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= parents) return;
  u64 st[12];
#pragma unroll
  for (int k = 0; k < 8; k++) st[k] = in[p * 8 + k];
#pragma unroll
  for (int k = 8; k < 12; k++) st[k] = 0;
  poseidon_permute_dev(st);
#pragma unroll
  for (int k = 0; k < 4; k++) out[p * 4 + k] = st[k];
}
__global__ void __launch_bounds__(64) k_level_coop(const u64* in, u64* out, size_t parents) {
  __builtin_amdgcn_s_setprio(3);   // latency-bound stage: win VALU arbitration against co-resident bulk hashing waves
  __shared__ u64 xch[4][24];
  const int g = threadIdx.x >> 4, i = threadIdx.x & 15;
  const size_t p = (size_t)blockIdx.x * 4 + g;
  CoopCtx cx; coop_init(cx, i);
  const size_t pp = p < parents ? p : parents - 1;
  u64 x = i < 8 ? in[pp * 8 + i] : 0;
  x = coop_permute(x, i, cx, xch[g]);
  if (p < parents && i < 4) out[p * 4 + i] = x;
}
// Several consecutive latency-bound levels in one launch: a block of G 16-lane groups takes 2G sibling digests and climbs
// log2(G)+1 levels, halving its active groups each time; every level is also written to the tree (query paths need all levels).
// A ~12 us level costs ~4 us of dispatch/drain as its own kernel, so the 11 small levels of a 2^17-leaf tree go from 11 launches
// to 2. The permutation's lane exchange is wave-local (COOP_SYNC), so only the level hand-off needs workgroup barriers, and waves
// whose groups have all retired skip the permutation.
template <int G>
__global__ void __launch_bounds__(16 * G) k_levels_coop_fused(const u64* __restrict__ in, u64* __restrict__ out, size_t parents0, int nlev) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ u64 xch[G][24];
  __shared__ u64 dig[G][4];
  const int g = threadIdx.x >> 4, i = threadIdx.x & 15, wave_g0 = g & ~3;
  CoopCtx cx; coop_init(cx, i);
  size_t par = parents0;
  const size_t p0 = (size_t)blockIdx.x * G + g;
  u64 x = (i < 8 && p0 < par) ? in[p0 * 8 + i] : 0;
  u64* o = out;
  for (int k = 0; k < nlev; k++) {
    const int act = G >> k;                                    // groups of this block still holding a parent
    if (wave_g0 < act) x = coop_permute(x, i, cx, xch[g]);
    const size_t q = (size_t)blockIdx.x * act + g;
    if (g < act && q < par && i < 4) { o[q * 4 + i] = x; dig[g][i] = x; }
    __syncthreads();
    u64 nx = 0;
    if (k + 1 < nlev && g < (act >> 1) && i < 8) nx = dig[2 * g + (i >> 2)][i & 3];
    __syncthreads();
    x = nx;
    o += par * 4; par >>= 1;
  }
}
static constexpr size_t COOP_MAX_PARENTS = 16384, COOP_MAX_PARENTS_BATCH = 4096;
void merkle_levels(u64* levels, size_t num_leaves, int cap_height, hipStream_t s, int levels_done, bool batch) {
  // The 2-lane form for 32 K / 64 K parents, the 16-lane form from 16 K parents down (lowest latency) — in batches too: the
  // 16-lane form costs 2.8x and the 2-lane form 1.4x the instructions of one permutation per lane (tools/valu_budget.py) and
  // batches are VALU-bound, yet per-lane levels down to 4 K parents measured 341-344 vs 356-369 proofs/s (medians of 9 batches
  // of 48, four worker streams): the longer critical path of every proof costs more overlap than the saved instructions buy.
  // With eight worker streams (round 3) the device is fuller and the switch to the 16-lane form moves down: in batches the 2-lane form
  // also takes the 8 K and 16 K-parent levels (QPGPU_COOP_MAX sweep, medians of 4 x 6 batches of 64: 16384 -> 415, 8192 -> 421, 4096 -> 424,
  // 2048 -> 425, 1024 -> 420, 256 -> 416 proofs/s; profiles/r03_batch_levels_ab.txt). A single proof keeps 16384.
  const int env_pair = opts().pair_max, env_coop = opts().coop_max;   // tuning overrides
  const size_t pair_max = env_pair >= 0 ? (size_t)env_pair : PAIR_MAX_LEAVES;
  const size_t coop_max = env_coop >= 0 ? (size_t)env_coop : (batch ? COOP_MAX_PARENTS_BATCH : COOP_MAX_PARENTS);
  size_t cnt = num_leaves, cap = size_t(1) << cap_height; u64* cur = levels;
  for (int l = 0; l < levels_done; l++) { cur += cnt * 4; cnt >>= 1; }
  while (cnt > cap) {
    size_t parents = cnt >> 1; u64* out = cur + cnt * 4;
    if (parents > (pair_max > coop_max ? pair_max : coop_max)) hipLaunchKernelGGL(k_level_lane, dim3((unsigned)((parents + 255) / 256)), dim3(256), 0, s, cur, out, parents);
    else if (parents > coop_max) hipLaunchKernelGGL(k_level_pair, dim3((unsigned)((2 * parents + 255) / 256)), dim3(256), 0, s, cur, out, parents);
    else {
      // latency-bound tail: climb up to 6 levels per launch
      const int fuse = opts().fuse_levels;
      int nlev = 0; { size_t c2 = cnt; while (c2 > cap && nlev < 6) { c2 >>= 1; nlev++; } }
      if (fuse && nlev >= 2 && parents >= 32) {
        hipLaunchKernelGGL(k_levels_coop_fused<32>, dim3((unsigned)((parents + 31) / 32)), dim3(512), 0, s, cur, out, parents, nlev);
        for (int k = 0; k < nlev; k++) { cur += cnt * 4; cnt >>= 1; }
        continue;
      }
      hipLaunchKernelGGL(k_level_coop, dim3((unsigned)((parents + 3) / 4)), dim3(64), 0, s, cur, out, parents);
    }
    cur = out; cnt = parents;
  }
}

// ---- verifier: one lane per Merkle opening (leaf sponge, then `depth` 2-to-1 steps up to the cap). All lanes of a launch share width
// and depth, so waves stay converged; the openings of a batch of proofs give thousands of independent lanes.
__global__ void __launch_bounds__(256) k_verify_paths(const u64* rows, int width, const u64* sibs, int depth, const u32* idx, const u64* caps,
                                                      const u32* cap_of, int cap_h, size_t np, u32* ok) {
  const size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (p >= np) return;
  const u64* row = rows + p * (size_t)width;
  u64 st[12];
#pragma unroll
  for (int k = 0; k < 12; k++) st[k] = 0;
  if (width <= 4) { for (int k = 0; k < width; k++) st[k] = row[k]; }
  else {
    for (int off = 0; off < width; off += 8) {
      const int m = width - off < 8 ? width - off : 8;
#pragma unroll
      for (int k = 0; k < 8; k++) if (k < m) st[k] = row[off + k];
      poseidon_permute_dev(st);
    }
  }
  u32 index = idx[p];
  const u64* sb = sibs + p * (size_t)depth * 4;
  for (int l = 0; l < depth; l++, sb += 4) {
    const bool right = index & 1;                    // this node is the right child: the sibling goes first
#pragma unroll
    for (int k = 0; k < 4; k++) { const u64 d = st[k], s2 = sb[k]; st[k] = right ? s2 : d; st[4 + k] = right ? d : s2; }
#pragma unroll
    for (int k = 8; k < 12; k++) st[k] = 0;
    poseidon_permute_dev(st);
    index >>= 1;
  }
  const u64* cap = caps + (((size_t)(cap_of ? cap_of[p] : 0)) << (cap_h + 2)) + (size_t)index * 4;
  bool same = true;
#pragma unroll
  for (int k = 0; k < 4; k++) same = same && st[k] == cap[k];
  ok[p] = same ? 1u : 0u;
}
// The same with 16 lanes per opening (one state element per lane, coop_permute): an opening is a chain of 13-31 dependent permutations, and
// with the openings of a few proofs only (28 per proof and tree) a per-lane permutation (~59 us alone on a SIMD) makes the launch pure latency.
__global__ void __launch_bounds__(64) k_verify_paths_coop(const u64* rows, int width, const u64* sibs, int depth, const u32* idx, const u64* caps,
                                                          const u32* cap_of, int cap_h, size_t np, u32* ok) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ u64 xch[4][24];
  const int g = threadIdx.x >> 4, i = threadIdx.x & 15;
  const size_t p = (size_t)blockIdx.x * 4 + g, pp = p < np ? p : np - 1;     // idle groups repeat the last opening (all 64 lanes must take part)
  CoopCtx cx; coop_init(cx, i);
  const u64* row = rows + pp * (size_t)width;
  u64 x = 0;
  if (width <= 4) { if (i < width) x = row[i]; }
  else {
    for (int off = 0; off < width; off += 8) {
      if (i < 8 && off + i < width) x = row[off + i];          // overwrite mode: the other lanes keep their state
      x = coop_permute(x, i, cx, xch[g]);
    }
  }
  u32 index = idx[pp];
  const u64* sb = sibs + pp * (size_t)depth * 4;
  for (int l = 0; l < depth; l++, sb += 4) {
    const bool right = index & 1;                              // this node is the right child: the sibling goes first
    const u64 below = __shfl(x, (int)threadIdx.x - 4);         // lanes 4..7 see the running digest of lanes 0..3
    u64 nx = 0;
    if (i < 4) nx = right ? sb[i] : x;
    else if (i < 8) nx = right ? below : sb[i - 4];
    x = coop_permute(nx, i, cx, xch[g]);
    index >>= 1;
  }
  const u64* cap = caps + (((size_t)(cap_of ? cap_of[pp] : 0)) << (cap_h + 2)) + (size_t)index * 4;
  const unsigned long long agree = __ballot(i >= 4 || x == cap[i & 3]);
  if (p < np && i == 0) ok[p] = ((agree >> (16 * g)) & 0xFFFFull) == 0xFFFFull ? 1u : 0u;
}
void verify_paths(const u64* rows, int width, const u64* sibs, int depth, const u32* idx, const u64* caps, const u32* cap_of, int cap_h,
                  size_t num_paths, u32* ok, hipStream_t s) {
  if (!num_paths) return;
  if ((long long)num_paths <= (long long)opts().verify_coop_max) { hipLaunchKernelGGL(k_verify_paths_coop, dim3((unsigned)((num_paths + 3) / 4)), dim3(64), 0, s, rows, width, sibs, depth, idx, caps, cap_of, cap_h, num_paths, ok); return; }
  hipLaunchKernelGGL(k_verify_paths, dim3((unsigned)((num_paths + 255) / 256)), dim3(256), 0, s, rows, width, sibs, depth, idx, caps, cap_of, cap_h, num_paths, ok);
}

// ---- explicit kernel-form selection (kernels.h): one launch of the named form, whatever the size
void merkle_leaves_colmajor_mode(const u64* cols, size_t col_stride, int width, size_t num_leaves, u64* digests, int mode, hipStream_t s) {
  dim3 grid((unsigned)((num_leaves + 255) / 256));
  if (mode == 1) hipLaunchKernelGGL(k_leaves_colmajor_pair, dim3((unsigned)((2 * num_leaves + 255) / 256)), dim3(256), 0, s, cols, col_stride, width, num_leaves, digests);
  else if (mode == 2) hipLaunchKernelGGL((k_leaves_colmajor<true, false>), grid, dim3(256), 0, s, cols, col_stride, width, num_leaves, digests, digests + num_leaves * 4, digests + (num_leaves + num_leaves / 2) * 4);
  else if (mode == 3) hipLaunchKernelGGL((k_leaves_colmajor<false, true>), grid, dim3(256), 0, s, cols, col_stride, width, num_leaves, digests, (u64*)nullptr, (u64*)nullptr);
  else if (mode == 4) hipLaunchKernelGGL((k_leaves_colmajor<true, true>), grid, dim3(256), 0, s, cols, col_stride, width, num_leaves, digests, digests + num_leaves * 4, digests + (num_leaves + num_leaves / 2) * 4);
  else hipLaunchKernelGGL((k_leaves_colmajor<false, false>), grid, dim3(256), 0, s, cols, col_stride, width, num_leaves, digests, (u64*)nullptr, (u64*)nullptr);
}
void merkle_level_mode(const u64* in, u64* out, size_t parents, int mode, hipStream_t s) {
  if (mode == 0) hipLaunchKernelGGL(k_level_lane, dim3((unsigned)((parents + 255) / 256)), dim3(256), 0, s, in, out, parents);
  else if (mode == 1) hipLaunchKernelGGL(k_level_pair, dim3((unsigned)((2 * parents + 255) / 256)), dim3(256), 0, s, in, out, parents);
  else if (mode == 2) hipLaunchKernelGGL(k_level_coop, dim3((unsigned)((parents + 3) / 4)), dim3(64), 0, s, in, out, parents);
  else hipLaunchKernelGGL(k_levels_coop_fused<32>, dim3((unsigned)((parents + 31) / 32)), dim3(512), 0, s, in, out, parents, 1);
}
void merkle_leaves_fri_mode(const u64* a, const u64* b, int arity_bits, size_t num_leaves, u64* digests, int mode, hipStream_t s) {
  if (mode == 1) hipLaunchKernelGGL(k_leaves_fri_coop, dim3((unsigned)((num_leaves + 3) / 4)), dim3(64), 0, s, a, b, arity_bits, num_leaves, digests);
  else hipLaunchKernelGGL(k_leaves_fri, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, a, b, arity_bits, num_leaves, digests);
}

} // namespace qp
