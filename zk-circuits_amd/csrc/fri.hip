// FRI prover stages on gfx950 (qp-plonky2's PolynomialBatch::prove_openings,
// fri_committed_trees, fri_proof_of_work, fri_prover_query_rounds, behind wormhole/prover/src/lib.rs:233-237).
// The batch combine is done in coefficient space like upstream (it reads n-length coefficient columns, 8x fewer bytes
// than combining LDE values); the division by (X - z) is a suffix sum against precomputed powers of z.
#include "kernels.h"
#include "poseidon_dev.h"

namespace qp {

// ---- compose: comp0[i] = sum_j alpha^j f_j[i] over the 4 oracle groups, comp1 from the first nch polynomials of g2.
// The polynomial list is cut into FRI_COMPOSE_GROUPS runs so the Horner chains are short and the grid fills the chip;
// each step acc*alpha + f is two 2-term dot products with a single reduction each.
struct ComposeArgs { PolyGroup g[4]; int total; int per_group; Ext alpha; Ext alpha_run; /* alpha^per_group */ };
__device__ __forceinline__ u64 compose_coeff(const ComposeArgs& a, int j, size_t i) {
  // explicit chain (no indexed access into the argument struct: that put the whole struct into scratch)
  int k = j;
  if (k < a.g[0].count) return a.g[0].base[(size_t)k * a.g[0].stride + i];
  k -= a.g[0].count;
  if (k < a.g[1].count) return a.g[1].base[(size_t)k * a.g[1].stride + i];
  k -= a.g[1].count;
  if (k < a.g[2].count) return a.g[2].base[(size_t)k * a.g[2].stride + i];
  k -= a.g[2].count;
  if (k < a.g[3].count) return a.g[3].base[(size_t)k * a.g[3].stride + i];
  return 0;
}
__global__ void __launch_bounds__(256) k_fri_compose_partial(ComposeArgs a, size_t n, u64* partial) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int grp = blockIdx.y, j0 = grp * a.per_group, j1 = j0 + a.per_group < a.total ? j0 + a.per_group : a.total;
  const u64 al0 = a.alpha.a, al1 = a.alpha.b, al1x7 = gl_mul7(a.alpha.b);
  u64 x0 = 0, x1 = 0;
  for (int j = j1 - 1; j >= j0; j--) {
    u64 f = compose_coeff(a, j, i);
    u64 n0 = dot2_lazy(x0, al0, x1, al1x7);      // (x0 + x1 X)(al0 + al1 X) with X^2 = 7
    u64 n1 = dot2_lazy(x0, al1, x1, al0);
    x0 = add_lazy_canon(n0, f); x1 = n1;
  }
  partial[((size_t)grp * 2) * n + i] = canon(x0); partial[((size_t)grp * 2 + 1) * n + i] = canon(x1);
}
__global__ void __launch_bounds__(256) k_fri_compose_finish(const u64* partial, size_t n, int groups, Ext alpha_run, PolyGroup g2, int nch, Ext alpha,
                                                            u64* comp0, u64* comp1) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ext acc = ex(0, 0);
  for (int g = groups - 1; g >= 0; g--) acc = ex_add(ex_mul(acc, alpha_run), ex(partial[((size_t)g * 2) * n + i], partial[((size_t)g * 2 + 1) * n + i]));
  comp0[i] = acc.a; comp0[n + i] = acc.b;
  Ext a1 = ex(0, 0);
  for (int j = nch - 1; j >= 0; j--) a1 = ex_add_base(ex_mul(a1, alpha), g2.base[(size_t)j * g2.stride + i]);
  comp1[i] = a1.a; comp1[n + i] = a1.b;
}
void fri_compose(PolyGroup g0, PolyGroup g1, PolyGroup g2, PolyGroup g3, int nch, int log_n, Ext alpha, u64* partial, u64* comp0, u64* comp1, hipStream_t s) {
  size_t n = size_t(1) << log_n;
  ComposeArgs a; a.g[0] = g0; a.g[1] = g1; a.g[2] = g2; a.g[3] = g3; a.total = g0.count + g1.count + g2.count + g3.count;
  a.per_group = (a.total + FRI_COMPOSE_GROUPS - 1) / FRI_COMPOSE_GROUPS; a.alpha = alpha; a.alpha_run = ex_pow(alpha, (u64)a.per_group);
  dim3 grid((unsigned)((n + 255) / 256), FRI_COMPOSE_GROUPS);
  hipLaunchKernelGGL(k_fri_compose_partial, grid, dim3(256), 0, s, a, n, partial);
  hipLaunchKernelGGL(k_fri_compose_finish, dim3(grid.x), dim3(256), 0, s, partial, n, FRI_COMPOSE_GROUPS, a.alpha_run, g2, nch, alpha, comp0, comp1);
}

// ---- divide: t_i = comp[i] * z^i ; S_k = sum_{i>k} t_i ; Q[k] = S_k * z^-(k+1)
__global__ void __launch_bounds__(256) k_div_prep(const u64* comp0, const u64* comp1, size_t n, const u64* zp0_a, const u64* zp0_b,
                                                  const u64* zp1_a, const u64* zp1_b, u64* scratch) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  Ext t0 = ex_mul(ex(comp0[i], comp0[n + i]), ex(zp0_a[i], zp0_b[i]));
  Ext t1 = ex_mul(ex(comp1[i], comp1[n + i]), ex(zp1_a[i], zp1_b[i]));
  scratch[i] = t0.a; scratch[n + i] = t0.b; scratch[2 * n + i] = t1.a; scratch[3 * n + i] = t1.b;
}
// Exclusive suffix sums, blocked: k_suffix_local leaves block-local suffix sums in place plus one total per block;
// k_div_finish adds the totals of all later blocks while it applies z^-(k+1) and combines the two batches.
static constexpr int SUF_BLOCK = 2048;
__global__ void __launch_bounds__(256) k_suffix_local(u64* arrs, size_t n, u64* btot) {
  __shared__ u64 sm[256];
  u64* t = arrs + (size_t)blockIdx.y * n;
  // thread r owns 8 consecutive elements; thread 0 owns the LAST 8 of the block (suffix direction)
  const size_t blk_hi = (size_t)(blockIdx.x + 1) * SUF_BLOCK < n ? (size_t)(blockIdx.x + 1) * SUF_BLOCK : n;
  const size_t blk_lo = (size_t)blockIdx.x * SUF_BLOCK;
  const size_t hi = blk_hi > (size_t)threadIdx.x * 8 + blk_lo ? blk_hi - (size_t)threadIdx.x * 8 : blk_lo;
  const size_t lo = hi > blk_lo + 8 ? hi - 8 : blk_lo;
  u64 v[8]; u64 acc = 0; int cnt = (int)(hi - lo);
  for (int r = 0; r < cnt; r++) { v[r] = t[lo + r]; acc = gl_add(acc, v[r]); }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    u64 x = threadIdx.x >= off ? sm[threadIdx.x - off] : 0;
    __syncthreads();
    sm[threadIdx.x] = gl_add(sm[threadIdx.x], x);
    __syncthreads();
  }
  u64 run = threadIdx.x ? sm[threadIdx.x - 1] : 0;
  for (int r = cnt - 1; r >= 0; r--) { t[lo + r] = run; run = gl_add(run, v[r]); }
  if (threadIdx.x == 255) btot[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = sm[255];
}
__global__ void __launch_bounds__(256) k_div_finish(const u64* scratch, size_t n, const u64* btot, int nblk, const u64* zi0_a, const u64* zi0_b,
                                                    const u64* zi1_a, const u64* zi1_b, Ext alpha_shift, u64* final_ab) {
  __shared__ u64 sm[4][256];
  const int blk = blockIdx.x / (SUF_BLOCK / 256);             // which scan block this thread block's 256 elements belong to
  for (int a = 0; a < 4; a++) {                               // sum of the totals of all later blocks, per array
    u64 acc = 0;
    for (int b = blk + 1 + threadIdx.x; b < nblk; b += 256) acc = gl_add(acc, btot[(size_t)a * nblk + b]);
    sm[a][threadIdx.x] = acc;
  }
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) for (int a = 0; a < 4; a++) sm[a][threadIdx.x] = gl_add(sm[a][threadIdx.x], sm[a][threadIdx.x + off]);
    __syncthreads();
  }
  size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (k >= n) return;
  Ext s0 = ex(gl_add(scratch[k], sm[0][0]), gl_add(scratch[n + k], sm[1][0]));
  Ext s1 = ex(gl_add(scratch[2 * n + k], sm[2][0]), gl_add(scratch[3 * n + k], sm[3][0]));
  Ext q0 = ex_mul(s0, ex(zi0_a[k + 1], zi0_b[k + 1]));
  Ext q1 = ex_mul(s1, ex(zi1_a[k + 1], zi1_b[k + 1]));
  Ext f = ex_add(ex_mul(q0, alpha_shift), q1);
  final_ab[k] = f.a; final_ab[n + k] = f.b;
}
void fri_divide_combine(const u64* comp0, const u64* comp1, int log_n, const u64* zp0_a, const u64* zp0_b, const u64* zp1_a,
                        const u64* zp1_b, const u64* zi0_a, const u64* zi0_b, const u64* zi1_a, const u64* zi1_b, Ext alpha_shift,
                        u64* scratch, u64* final_ab, hipStream_t s) {
  size_t n = size_t(1) << log_n; unsigned blocks = (unsigned)((n + 255) / 256);
  const unsigned nblk = (unsigned)((n + SUF_BLOCK - 1) / SUF_BLOCK);
  u64* btot = scratch + 4 * n;
  hipLaunchKernelGGL(k_div_prep, dim3(blocks), dim3(256), 0, s, comp0, comp1, n, zp0_a, zp0_b, zp1_a, zp1_b, scratch);
  hipLaunchKernelGGL(k_suffix_local, dim3(nblk, 4), dim3(256), 0, s, scratch, n, btot);
  hipLaunchKernelGGL(k_div_finish, dim3(blocks), dim3(256), 0, s, scratch, n, btot, (int)nblk, zi0_a, zi0_b, zi1_a, zi1_b, alpha_shift, final_ab);
}

// ---- fold
__global__ void __launch_bounds__(256) k_fri_fold(const u64* in_a, const u64* in_b, size_t n_out, int arity_bits, Ext beta, u64* out_a, u64* out_b) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n_out) return;
  const int arity = 1 << arity_bits;
  Ext acc = ex(0, 0);
  for (int j = arity - 1; j >= 0; j--) acc = ex_add(ex_mul(acc, beta), ex(in_a[(i << arity_bits) + j], in_b[(i << arity_bits) + j]));
  out_a[i] = acc.a; out_b[i] = acc.b;
}
void fri_fold(const u64* in_a, const u64* in_b, size_t n_out, int arity_bits, Ext beta, u64* out_a, u64* out_b, hipStream_t s) {
  hipLaunchKernelGGL(k_fri_fold, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, s, in_a, in_b, n_out, arity_bits, beta, out_a, out_b);
}

// ---- proof of work
__global__ void __launch_bounds__(256) k_pow_grind(Sponge12 sp, int pos, u64 base, u64 count, int bits, u64* result) {
  u64 gid = blockIdx.x * (u64)blockDim.x + threadIdx.x;
  if (gid >= count) return;
  u64 cand = base + gid;
  if (cand >= GL_P) return;
  u64 st[12];
#pragma unroll
  for (int k = 0; k < 12; k++) st[k] = sp.s[k];
#pragma unroll
  for (int k = 0; k < 8; k++) if (k == pos) st[k] = cand;
  poseidon_permute_dev(st);
  int lz = st[7] == 0 ? 64 : __clzll((long long)st[7]);
  if (lz >= bits) atomicMin((unsigned long long*)result, (unsigned long long)cand);
}
void pow_grind(Sponge12 st, int pos, u64 base, u64 count, int bits, u64* result, hipStream_t s) {
  hipLaunchKernelGGL(k_pow_grind, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, st, pos, base, count, bits, result);
}

// ---- query gather: one block per query round; bytes written unaligned (the u8 sibling counts shift everything)
__device__ __forceinline__ void put_u64(uint8_t* p, u64 v) {
#pragma unroll
  for (int b = 0; b < 8; b++) p[b] = (uint8_t)(v >> (8 * b));
}
__global__ void __launch_bounds__(256) k_query_gather(GatherArgs a, uint8_t* out) {
  const int q = blockIdx.x;
  if (q == 0 && threadIdx.x == 0 && a.pow_reset) *a.pow_reset = ~0ULL;
  size_t x = a.idx[q];
  uint8_t* p = out + (size_t)q * a.bytes_per_query;
  const size_t N = size_t(1) << a.log_N;
  const int path = a.log_N - a.cap_height;
  for (int t = 0; t < 4; t++) {
    const GatherTree& tr = a.trees[t];
    for (int k = threadIdx.x; k < tr.width; k += blockDim.x) put_u64(p + 8 * (size_t)k, tr.cols[(size_t)k * tr.stride + x]);
    p += 8 * (size_t)tr.width;
    if (threadIdx.x == 0) *p = (uint8_t)path;
    p += 1;
    // siblings: level l digest index (x >> l) ^ 1
    for (int e = threadIdx.x; e < path * 4; e += blockDim.x) {
      int l = e >> 2, w = e & 3;
      size_t off = 0, cnt = N; for (int i = 0; i < l; i++) { off += cnt; cnt >>= 1; }
      put_u64(p + 8 * (size_t)e, tr.digests[(off + ((x >> l) ^ 1)) * 4 + w]);
    }
    p += 32 * (size_t)path;
  }
  for (int li = 0; li < a.num_layers; li++) {
    const GatherFri& f = a.layers[li];
    const int arity = 1 << f.arity_bits;
    size_t ci = x >> f.arity_bits;
    for (int k = threadIdx.x; k < 2 * arity; k += blockDim.x) {
      size_t e = (ci << f.arity_bits) + (k >> 1);
      put_u64(p + 8 * (size_t)k, (k & 1) ? f.b[e] : f.a[e]);
    }
    p += 16 * (size_t)arity;
    int fpath = f.log_leaves - a.cap_height; if (fpath < 0) fpath = 0;
    if (threadIdx.x == 0) *p = (uint8_t)fpath;
    p += 1;
    const size_t leaves = size_t(1) << f.log_leaves;
    for (int e = threadIdx.x; e < fpath * 4; e += blockDim.x) {
      int l = e >> 2, w = e & 3;
      size_t off = 0, cnt = leaves; for (int i = 0; i < l; i++) { off += cnt; cnt >>= 1; }
      put_u64(p + 8 * (size_t)e, f.digests[(off + ((ci >> l) ^ 1)) * 4 + w]);
    }
    p += 32 * (size_t)fpath;
    x = ci;
  }
}
void query_gather(const GatherArgs& a, uint8_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k_query_gather, dim3((unsigned)a.nq), dim3(256), 0, s, a, out);
}

} // namespace qp
