// Batched Poseidon hashing for the Merkle-cap commitments (MerkleTree::new + PoseidonHash::
// {hash_or_noop, two_to_one} of qp-plonky2, behind wormhole/prover/src/lib.rs:233-237).
// One permutation per lane, state in VGPRs; leaves stream column-major oracle values (coalesced across lanes).
#include "kernels.h"
#include "poseidon_dev.h"
#include "salt.h"

namespace qp {

__global__ void k_salt_fill(u64* cols, size_t N, u64 seed, int oracle) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < 4 * N) cols[i] = salt_value(seed, (u64)oracle * 4 * N + i);
}
void salt_fill(u64* cols, size_t N, u64 seed, int oracle, hipStream_t s) {
  hipLaunchKernelGGL(k_salt_fill, dim3((unsigned)((4 * N + 255) / 256)), dim3(256), 0, s, cols, N, seed, oracle);
}

__global__ void k_permute_batch(u64* states, size_t count) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= count) return;
  u64 st[12];
#pragma unroll
  for (int k = 0; k < 12; k++) st[k] = states[i * 12 + k];
  poseidon_permute_dev(st);
#pragma unroll
  for (int k = 0; k < 12; k++) states[i * 12 + k] = st[k];
}
void poseidon_permute_batch(u64* states, size_t count, hipStream_t s) {
  if (!count) return;
  hipLaunchKernelGGL(k_permute_batch, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, states, count);
}

__global__ void __launch_bounds__(256) k_leaves_colmajor(const u64* cols, size_t col_stride, int width, size_t num_leaves, u64* digests) {
  size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (L >= num_leaves) return;
  u64 st[12];
#pragma unroll
  for (int k = 0; k < 12; k++) st[k] = 0;
  if (width <= 4) {
    for (int k = 0; k < width; k++) st[k] = cols[(size_t)k * col_stride + L];
  } else {
    for (int off = 0; off < width; off += 8) {
      int m = width - off < 8 ? width - off : 8;
      if (m == 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) st[k] = cols[(size_t)(off + k) * col_stride + L];
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++) if (k < m) st[k] = cols[(size_t)(off + k) * col_stride + L];
      }
      poseidon_permute_dev(st);
    }
  }
#pragma unroll
  for (int k = 0; k < 4; k++) digests[L * 4 + k] = st[k];
}
void merkle_leaves_colmajor(const u64* cols, size_t col_stride, int width, size_t num_leaves, u64* digests, hipStream_t s) {
  hipLaunchKernelGGL(k_leaves_colmajor, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, cols, col_stride, width, num_leaves, digests);
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
void merkle_leaves_fri(const u64* a, const u64* b, int arity_bits, size_t num_leaves, u64* digests, hipStream_t s) {
  hipLaunchKernelGGL(k_leaves_fri, dim3((unsigned)((num_leaves + 255) / 256)), dim3(256), 0, s, a, b, arity_bits, num_leaves, digests);
}

// One block reduces 2^r consecutive nodes of a level by up to r levels, writing every intermediate level.
static constexpr int LV_R = 8;   // 256 nodes in, 128 threads active at the first step
__global__ void __launch_bounds__(128) k_levels(u64* in_level, size_t in_count, u64* out_levels, int steps) {
  __shared__ u64 sm[256 * 4];
  const size_t base = (size_t)blockIdx.x << LV_R;
  for (unsigned e = threadIdx.x; e < 256 * 4; e += 128) { size_t g = base * 4 + e; sm[e] = g < in_count * 4 ? in_level[g] : 0; }
  __syncthreads();
  size_t lvl_count = in_count; u64* out = out_levels; unsigned width = 256;
  for (int st_i = 0; st_i < steps; st_i++) {
    width >>= 1; lvl_count >>= 1;
    u64 st[12]; bool act = threadIdx.x < width && ((base >> (st_i + 1)) + threadIdx.x) < lvl_count;
    if (act) {
#pragma unroll
      for (int k = 0; k < 8; k++) st[k] = sm[threadIdx.x * 8 + k];
#pragma unroll
      for (int k = 8; k < 12; k++) st[k] = 0;
      poseidon_permute_dev(st);
    }
    __syncthreads();
    if (act) {
      size_t node = (base >> (st_i + 1)) + threadIdx.x;
#pragma unroll
      for (int k = 0; k < 4; k++) { sm[threadIdx.x * 4 + k] = st[k]; out[node * 4 + k] = st[k]; }
    }
    __syncthreads();
    out += lvl_count * 4;
  }
}
void merkle_levels(u64* levels, size_t num_leaves, int cap_height, hipStream_t s) {
  size_t cnt = num_leaves, cap = size_t(1) << cap_height; u64* cur = levels;
  while (cnt > cap) {
    int steps = 0; { size_t c = cnt; while (c > cap && steps < LV_R) { c >>= 1; steps++; } }
    unsigned blocks = (unsigned)((cnt + 255) >> LV_R);
    u64* out = cur + cnt * 4;
    hipLaunchKernelGGL(k_levels, dim3(blocks), dim3(128), 0, s, cur, cnt, out, steps);
    for (int i = 0; i < steps; i++) { cur += cnt * 4; cnt >>= 1; }
  }
}

} // namespace qp
