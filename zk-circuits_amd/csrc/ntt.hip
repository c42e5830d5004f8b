// Radix-2 Goldilocks NTT / coset LDE for gfx950 (the a3/a4 rows of SURVEY.md §8: qp-plonky2-field's ifft and
// PolynomialBatch::lde_values behind wormhole/prover/src/lib.rs:233-237).
//
// Layout: one polynomial per contiguous column; LDE values are produced directly in LEAF order (bit-reversed natural
// index) so Merkle leaves, quotient reads and FRI cosets are contiguous. Each pass stages a 2^TL-element tile in LDS,
// runs up to TL butterfly levels on it and writes it back; a size-n transform with n <= 2^TL is one pass.
#include "kernels.h"

namespace qp {

static constexpr int TL_MAX = 12;       // tile = 4096 elements = 32 KiB LDS
static constexpr int NTT_THREADS = 256;

__device__ __forceinline__ u64 root_pow(const u64* rt, int log_M, u64 e) {
  u64 M = 1ULL << log_M; e &= (M - 1);
  u64 half = M >> 1;
  return e < half ? rt[e] : gl_neg(rt[e - half]);
}

__global__ void k_root_table(u64* rt, int log_M) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, half = (size_t(1) << log_M) >> 1;
  if (i < half) rt[i] = gl_pow(gl_root(log_M), i);
}
void build_root_table(u64* d_rt, int log_M, hipStream_t s) {
  size_t half = (size_t(1) << log_M) >> 1; if (!half) half = 1;
  hipLaunchKernelGGL(k_root_table, dim3((half + 255) / 256), dim3(256), 0, s, d_rt, log_M);
}

__global__ void k_lde_scale(u64* out, int log_n, int log_cosets, u64 shift, const u64* rt, int log_M) {
  size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x, n = size_t(1) << log_n;
  if (idx >= (n << log_cosets)) return;
  size_t c = idx >> log_n, i = idx & (n - 1);
  int log_N = log_n + log_cosets;
  u64 w = root_pow(rt, log_M, (u64)(c * i) << (log_M - log_N));
  out[idx] = gl_mul(gl_pow(shift, i), w);
}
void build_lde_scale(u64* d_out, int log_n, int log_cosets, u64 shift, const NttPlan& pl, hipStream_t s) {
  size_t tot = size_t(1) << (log_n + log_cosets);
  hipLaunchKernelGGL(k_lde_scale, dim3((tot + 255) / 256), dim3(256), 0, s, d_out, log_n, log_cosets, shift, pl.rt, pl.log_M);
}

__global__ void k_powers(u64* out, size_t count, u64 base, u64 scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < count) out[i] = gl_mul(scale, gl_pow(base, i));
}
void build_powers(u64* d_out, size_t count, u64 base, u64 scale, hipStream_t s) {
  hipLaunchKernelGGL(k_powers, dim3((count + 255) / 256), dim3(256), 0, s, d_out, count, base, scale);
}

struct PassArgs {
  const u64* in; size_t in_stride;      // source columns (pass may read from a different buffer than it writes)
  u64* out; size_t out_stride;
  const u64* pre;                       // optional prescale [cosets][n] (first pass of an LDE)
  const u64* post;                      // optional postscale [n] (last pass)
  const u64* rt; int log_M;
  int log_n, b_lo, b_hi, t, tile_log, dit, inverse, log_cosets;
};

// One tile: bits [b_lo, b_hi) of the element index (plus the low t bits for coalescing) live in LDS.
__global__ void __launch_bounds__(NTT_THREADS) k_ntt_pass(PassArgs a) {
  extern __shared__ u64 sm[];
  const int TL = a.tile_log, m = a.b_hi - a.b_lo, t = a.t;
  const size_t n = size_t(1) << a.log_n;
  const unsigned tid = blockIdx.y;                       // transform id = column * cosets + coset
  const unsigned col = tid >> a.log_cosets, c = tid & ((1u << a.log_cosets) - 1);
  const u64* src = a.in + (size_t)col * a.in_stride + (a.pre ? 0 : (size_t)rev_bits(c, a.log_cosets) * n);
  u64* dst = a.out + (size_t)col * a.out_stride + (size_t)rev_bits(c, a.log_cosets) * n;
  const u64* pre = a.pre ? a.pre + (size_t)c * n : nullptr;
  const size_t q = blockIdx.x;
  const int midbits = a.b_lo - t;
  const size_t mid = midbits > 0 ? (q & ((size_t(1) << midbits) - 1)) : 0, high = midbits > 0 ? (q >> midbits) : q;
  const size_t base = (high << (a.b_lo == 0 ? TL : a.b_hi)) | (mid << t);   // contiguous tiles may hold several 2^b_hi blocks
  const unsigned tmask = (1u << t) - 1;
  const unsigned E = 1u << TL;
  // load
  for (unsigned e = threadIdx.x; e < E; e += NTT_THREADS) {
    size_t i = base | ((size_t)(e >> t) << a.b_lo) | (e & tmask);
    u64 v = src[i];
    if (pre) v = gl_mul(v, pre[i]);
    sm[e] = v;
  }
  __syncthreads();
  const int shiftM = a.log_M - a.log_n;
  for (int k = 0; k < m; k++) {
    const int j = a.dit ? k : (m - 1 - k);          // tile-relative bit
    const int pb = t + j, b = a.b_lo + j;
    for (unsigned u = threadIdx.x; u < (E >> 1); u += NTT_THREADS) {
      unsigned p0 = ((u >> pb) << (pb + 1)) | (u & ((1u << pb) - 1)), p1 = p0 | (1u << pb);
      size_t i0 = base | ((size_t)(p0 >> t) << a.b_lo) | (p0 & tmask);
      u64 jj = i0 & ((size_t(1) << b) - 1);
      u64 e = (jj << (a.log_n - b - 1)) << shiftM;
      if (a.inverse) e = (1ULL << a.log_M) - e;
      u64 w = root_pow(a.rt, a.log_M, e);
      u64 x = sm[p0], y = sm[p1];
      if (a.dit) { y = gl_mul(y, w); sm[p0] = gl_add(x, y); sm[p1] = gl_sub(x, y); }
      else { sm[p0] = gl_add(x, y); sm[p1] = gl_mul(gl_sub(x, y), w); }
    }
    __syncthreads();
  }
  for (unsigned e = threadIdx.x; e < E; e += NTT_THREADS) {
    size_t i = base | ((size_t)(e >> t) << a.b_lo) | (e & tmask);
    u64 v = sm[e];
    if (a.post) v = gl_mul(v, a.post[i]);
    dst[i] = v;
  }
}

// Pass schedule: bits are split into groups of at most TL_MAX (contiguous group) / 8 (strided groups, with 2^4 contiguous
// elements per row for coalescing). DIF walks the groups from the top bit down, DIT from the bottom up.
static void run_passes(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets,
                       bool dit, bool inverse, const u64* pre, const u64* post, const NttPlan& pl, hipStream_t s) {
  struct Grp { int lo, hi; };
  Grp g[8]; int ng = 0;
  if (log_n <= TL_MAX) { g[ng++] = {0, log_n}; }
  else {
    int rest = log_n - TL_MAX, np = (rest + 7) / 8;
    int low = TL_MAX;
    // balance: total passes np+1
    int per = (log_n + np) / (np + 1); if (per > TL_MAX) per = TL_MAX; if (per < log_n - 8 * np) per = log_n - 8 * np;
    low = per; g[ng++] = {0, low};
    int b = low;
    for (int p = 0; p < np; p++) { int left = log_n - b, parts = np - p; int w = (left + parts - 1) / parts; g[ng++] = {b, b + w}; b += w; }
  }
  int cosets = 1 << log_cosets;
  for (int k = 0; k < ng; k++) {
    int gi = dit ? k : ng - 1 - k;
    PassArgs a;
    bool first = (k == 0), last = (k == ng - 1);
    a.in = first ? in : out; a.in_stride = first ? in_stride : out_stride; a.out = out; a.out_stride = out_stride;
    a.pre = first ? pre : nullptr; a.post = last ? post : nullptr; a.rt = pl.rt; a.log_M = pl.log_M;
    a.log_n = log_n; a.b_lo = g[gi].lo; a.b_hi = g[gi].hi; a.dit = dit; a.inverse = inverse; a.log_cosets = log_cosets;
    int m = a.b_hi - a.b_lo;
    if (a.b_lo == 0) { a.t = 0; a.tile_log = log_n < TL_MAX ? log_n : TL_MAX; if (a.tile_log < m) a.tile_log = m; }
    else { int t = TL_MAX - m; if (t > a.b_lo) t = a.b_lo; if (t > 4) t = 4; a.t = t; a.tile_log = m + t; }
    // few transforms (quotient iNTT, FRI layers: 2 columns): shrink the tile until the launch has enough workgroups for the chip
    while (((size_t(1) << log_n) >> a.tile_log) * (size_t)(count * cosets) < 1024) {
      if (a.b_lo == 0) { if (a.tile_log <= m || a.tile_log <= 8) break; a.tile_log--; }
      else { if (a.t <= 2) break; a.t--; a.tile_log--; }
    }
    if (!first && a.pre == nullptr && log_cosets) {
      // later passes of an LDE operate in place on the output blocks: source == destination block
      a.in = out; a.in_stride = out_stride;
    }
    size_t tiles = (size_t(1) << log_n) >> a.tile_log;
    dim3 grid((unsigned)tiles, (unsigned)(count * cosets));
    size_t lds = sizeof(u64) << a.tile_log;
    hipLaunchKernelGGL(k_ntt_pass, grid, dim3(NTT_THREADS), lds, s, a);
  }
}

void ntt_inplace(u64* data, size_t stride, int count, int log_n, bool dit, bool inverse, const u64* post, const NttPlan& pl, hipStream_t s) {
  if (count <= 0) return;
  run_passes(data, stride, data, stride, count, log_n, 0, dit, inverse, nullptr, post, pl, s);
}
void ntt_copy(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, bool dit, bool inverse, const NttPlan& pl, hipStream_t s) {
  if (count <= 0) return;
  run_passes(in, in_stride, out, out_stride, count, log_n, 0, dit, inverse, nullptr, nullptr, pl, s);
}

void lde_batch_generic(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets,
                       const u64* scale, const NttPlan& pl, hipStream_t s) {
  if (count <= 0) return;
  run_passes(in, in_stride, out, out_stride, count, log_n, log_cosets, false, false, scale, nullptr, pl, s);
}
void lde_batch(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets,
               const u64* scale, const NttPlan& pl, hipStream_t s, bool shared_gpu) {
  if (count <= 0) return;
  // one workgroup per transform in ntt16: with only a handful of transforms (FRI layers) the tiled kernel spreads better
  if (ntt16_supported(log_n) && (count << log_cosets) >= 64) { ntt16_run(in, in_stride, out, out_stride, count, log_n, log_cosets, false, false, scale, 1, pl, s); return; }
  // strided radix-16 passes: 1.4-1.6x faster than the tiled kernel when the launch owns the chip (single proof: 4.54 vs 4.68 ms);
  // with several proofs in flight its 35 KB-LDS block pass co-runs worse (277 vs 287 proofs/s), so batch mode keeps the tiled kernel.
  if (!shared_gpu && lde_strided_supported(log_n) && (count << log_cosets) >= 64) { lde_strided_run(in, in_stride, out, out_stride, count, log_n, log_cosets, scale, pl, s); return; }
  run_passes(in, in_stride, out, out_stride, count, log_n, log_cosets, false, false, scale, nullptr, pl, s);
}

__global__ void k_bitrev_permute(const u64* in, size_t in_stride, u64* out, size_t out_stride, int log_n, u64 scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t(1) << log_n)) return;
  u64 v = in[(size_t)blockIdx.y * in_stride + rev_bits((u32)i, log_n)];
  out[(size_t)blockIdx.y * out_stride + i] = scale == 1 ? v : gl_mul(v, scale);
}
void bitrev_permute(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, u64 scale, hipStream_t s) {
  if (count <= 0) return;
  size_t n = size_t(1) << log_n;
  hipLaunchKernelGGL(k_bitrev_permute, dim3((unsigned)((n + 255) / 256), (unsigned)count), dim3(256), 0, s, in, in_stride, out, out_stride, log_n, scale);
}

} // namespace qp
