// LDS-resident radix-16 Goldilocks NTT for 2^9 <= n <= 2^14 (one global read + one global write per element).
//
// One workgroup owns one transform (one column, one LDE coset): n/16 threads, 16 elements per thread. The transform is the
// in-place DIF factorisation n = 16 x 16 x ... x 2^rem: each step gathers 16 elements at stride m = N_s/16 from LDS into
// registers, runs a 16-point DIF (constant 16th-root twiddles), multiplies output f by w_{N_s}^{u f} (table lookup) and
// scatters back in place with the sub-index bit-reversed, so after the last step LDS holds the transform in bit-reversed
// order (= Merkle leaf order). The LDS image is padded one slot per 16 (pos(e) = e + e/16) so that both the stride-m
// gathers and the final contiguous groups are (nearly) bank-conflict free for 8-byte accesses.
// Used for: witness/Z iNTT (inverse, natural-order output written straight from LDS, no separate bit-reversal pass) and the
// x8 coset LDE (prescale fused into the load). Larger transforms (quotient iNTT at 2^17, the 2^20 stress config) stay on
// the tiled multi-pass kernel in ntt.hip. Rows a3/a4 of SURVEY.md §8.
#include "kernels.h"
#include "options.h"
#include "poseidon_dev.h"
#include <cstdlib>

namespace qp {

struct W16 { u64 w[8]; };   // w[j] = W^j, W = primitive 16th root (or its inverse)

__device__ __forceinline__ u64 mulc(u64 a, u64 b) { return canon(mul_lazy(a, b)); }
__device__ __forceinline__ u64 root_pow16(const u64* rt, int log_M, u64 e) {
  u64 M = 1ULL << log_M; e &= (M - 1); u64 half = M >> 1;
  return e < half ? rt[e] : gl_neg(rt[e - half]);
}
// Same value through two small windows of the table: w^e = w^(e_hi * 2^12) * w^(e_lo). A strided pass's output twiddles
// w^(low * F) walk the whole table (one distinct 64-byte line per element: at log_M = 23 a 64 MB footprint that lives in neither
// L2 nor comfortably in MALL); the two windows are 32 KB contiguous + (M/2^13) lines, so they stay in L2. One extra multiply.
__device__ __forceinline__ u64 root_pow2w(const u64* rt, int log_M, u64 e) {
  if (log_M <= 17) return root_pow16(rt, log_M, e);                 // the whole table (<= 512 KB) is L2-resident anyway
  const u64 M = 1ULL << log_M; e &= (M - 1); const u64 half = M >> 1;
  const bool neg = e >= half; e &= half - 1;
  const u64 v = gl_mul(rt[(e >> 12) << 12], rt[e & 4095]);
  return neg ? gl_neg(v) : v;
}
__device__ __forceinline__ unsigned lpos(unsigned e) { return e + (e >> 4); }

// 16-point DIF, output in bit-reversed register order: x[r] = sum_k in[k] W^(k * rev4(r))
__device__ __forceinline__ void dft16(u64 x[16], const W16& c) {
#pragma unroll
  for (int k = 0; k < 8; k++) { u64 a = x[k], b = x[k + 8]; x[k] = gl_add(a, b); u64 d = gl_sub(a, b); x[k + 8] = k ? mulc(d, c.w[k]) : d; }
#pragma unroll
  for (int h = 0; h < 16; h += 8)
#pragma unroll
    for (int k = 0; k < 4; k++) { u64 a = x[h + k], b = x[h + k + 4]; x[h + k] = gl_add(a, b); u64 d = gl_sub(a, b); x[h + k + 4] = k ? mulc(d, c.w[2 * k]) : d; }
#pragma unroll
  for (int q = 0; q < 16; q += 4)
#pragma unroll
    for (int k = 0; k < 2; k++) { u64 a = x[q + k], b = x[q + k + 2]; x[q + k] = gl_add(a, b); u64 d = gl_sub(a, b); x[q + k + 2] = k ? mulc(d, c.w[4]) : d; }
#pragma unroll
  for (int p = 0; p < 16; p += 2) { u64 a = x[p], b = x[p + 1]; x[p] = gl_add(a, b); x[p + 1] = gl_sub(a, b); }
}

struct Ntt16Args {
  const u64* in; size_t in_stride;    // column c at in + c*in_stride (n elements, natural order)
  u64* out; size_t out_stride;        // column c (and coset block) at out + c*out_stride + rev(coset)*n
  const u64* pre;                     // optional [cosets][n] prescale
  const u64* rt; int log_M;
  W16 w16; u64 scale;                 // scale applied on store (1 = none)
  int log_cosets, inverse, out_natural;
  unsigned transforms;                // grid is folded as (x, y): transform id = y * gridDim.x + x
};

template <int K>
__global__ void __launch_bounds__(1 << (K - 4)) k_ntt16(Ntt16Args a) {
  extern __shared__ u64 sm[];
  constexpr unsigned n = 1u << K, T = n >> 4;
  constexpr int S = K / 4, REM = K % 4;
  const unsigned t = threadIdx.x;
  const unsigned tid = blockIdx.y * gridDim.x + blockIdx.x;
  if (tid >= a.transforms) return;                   // whole block exits together
  const unsigned col = tid >> a.log_cosets, co = tid & ((1u << a.log_cosets) - 1);
  const u64* src = a.in + (size_t)col * a.in_stride;
  u64* dst = a.out + (size_t)col * a.out_stride + (size_t)rev_bits(co, a.log_cosets) * n;
  const u64* pre = a.pre ? a.pre + (size_t)co * n : nullptr;
  const u64 Mfull = 1ULL << a.log_M;
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { u64 v = src[k * T + t]; if (pre) v = mulc(v, pre[k * T + t]); x[k] = v; }
#pragma unroll
  for (int s = 0; s < S; s++) {
    const unsigned Ns_log = K - 4 * s, m = 1u << (Ns_log - 4);
    const unsigned B = t >> (Ns_log - 4), u = t & (m - 1);
    const unsigned base = (B << Ns_log) + u;
    if (s > 0) {
#pragma unroll
      for (int k = 0; k < 16; k++) x[k] = sm[lpos(base + k * m)];
    }
    dft16(x, a.w16);
    if (m > 1) {
      // x[r] holds frequency f = rev4(r): multiply by w_{N_s}^{u f}
      const u64 unit = (u64)u << (a.log_M - Ns_log);
#pragma unroll
      for (int r = 1; r < 16; r++) {
        const unsigned f = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
        u64 e = unit * f;
        if (a.inverse) e = Mfull - e;
        x[r] = mulc(x[r], root_pow16(a.rt, a.log_M, e));
      }
    }
    // in place: a thread writes exactly the 16 slots it read, so no barrier is needed between its reads and writes
#pragma unroll
    for (int r = 0; r < 16; r++) sm[lpos(base + r * m)] = x[r];
    __syncthreads();
  }
  if constexpr (REM > 0) {
    constexpr unsigned R = 1u << REM, G = 16 / R;
#pragma unroll
    for (unsigned j = 0; j < G; j++) {
      const unsigned e0 = (t + j * T) * R;
      u64 y[R];
#pragma unroll
      for (unsigned p = 0; p < R; p++) y[p] = sm[lpos(e0 + p)];
      if constexpr (REM == 1) { u64 p0 = y[0], p1 = y[1]; y[0] = gl_add(p0, p1); y[1] = gl_sub(p0, p1); }
      if constexpr (REM == 2) {
        u64 a0 = gl_add(y[0], y[2]), a2 = gl_sub(y[0], y[2]), a1 = gl_add(y[1], y[3]), a3 = mulc(gl_sub(y[1], y[3]), a.w16.w[4]);
        y[0] = gl_add(a0, a1); y[1] = gl_sub(a0, a1); y[2] = gl_add(a2, a3); y[3] = gl_sub(a2, a3);
      }
      if constexpr (REM == 3) {
        u64 z[8];
#pragma unroll
        for (int k = 0; k < 4; k++) { u64 p = y[k], q = y[k + 4]; z[k] = gl_add(p, q); u64 d = gl_sub(p, q); z[k + 4] = k ? mulc(d, a.w16.w[2 * k]) : d; }
#pragma unroll
        for (int h = 0; h < 8; h += 4)
#pragma unroll
          for (int k = 0; k < 2; k++) { u64 p = z[h + k], q = z[h + k + 2]; z[h + k] = gl_add(p, q); u64 d = gl_sub(p, q); z[h + k + 2] = k ? mulc(d, a.w16.w[4]) : d; }
#pragma unroll
        for (int p = 0; p < 8; p += 2) { u64 u0 = z[p], u1 = z[p + 1]; y[p] = gl_add(u0, u1); y[p + 1] = gl_sub(u0, u1); }
      }
#pragma unroll
      for (unsigned p = 0; p < R; p++) sm[lpos(e0 + p)] = y[p];
    }
    __syncthreads();
  }
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const unsigned e = k * T + t;
    const unsigned s_e = a.out_natural ? rev_bits(e, K) : e;
    u64 v = sm[lpos(s_e)];
    if (a.scale != 1) v = mulc(v, a.scale);
    dst[e] = v;
  }
}

template <int K> static void launch16(Ntt16Args a, int transforms, hipStream_t s) {
  constexpr size_t lds = ((size_t(1) << K) + (size_t(1) << (K - 4))) * 8;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)k_ntt16<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  a.transforms = (unsigned)transforms;
  const unsigned gx = transforms < 32768 ? (unsigned)transforms : 32768u, gy = ((unsigned)transforms + gx - 1) / gx;
  hipLaunchKernelGGL(k_ntt16<K>, dim3(gx, gy), dim3(1u << (K - 4)), lds, s, a);
}

// ---------------------------------------------------------------------------------------------------------------------
// Strided radix passes for transforms larger than one LDS image: the top M bits of a sub-transform of size P = 2^(b_lo+M)
// are processed as a pure 2^M-point DIF per low index (constant twiddles only), followed by ONE twiddle w_P^(low * F) per
// element — the decimation identity X[F + 2^M g] = DFT_{P/2^M}( w_P^(low F) * DFT_{2^M}(x[. , low])[F] )[g]. M <= 4 runs
// entirely in registers (each thread owns 2^M elements at stride 2^b_lo, coalesced across threads); M = 8 is two radix-16
// steps with one exchange through a 32 KB LDS tile (256 h-values x 16 consecutive low indices). After the strided passes the
// remaining 2^9..2^12 contiguous blocks are finished by k_ntt16. Replaces the tiled radix-2 kernel for forward LDEs with
// n >= 2^13 (2 passes of radix-16 steps instead of 2-3 passes of radix-2 LDS stages with per-butterfly table lookups).
struct StridedArgs {
  const u64* in; size_t in_stride; u64* out; size_t out_stride; const u64* pre; const u64* rt; int log_M; W16 w16;
  int log_n, b_lo, log_cosets, inverse, first;
};
__device__ __forceinline__ unsigned rev_m(unsigned x, int m) { return m ? (__brev(x) >> (32 - m)) : 0; }

template <int M>
__global__ void __launch_bounds__(256) k_ntt_strided_reg(StridedArgs a) {
  constexpr int R = 1 << M;
  const size_t n = size_t(1) << a.log_n;
  const unsigned tid = blockIdx.y, col = tid >> a.log_cosets, co = tid & ((1u << a.log_cosets) - 1);
  const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (g >= (n >> M)) return;
  const int b_hi = a.b_lo + M;
  const size_t low = g & ((size_t(1) << a.b_lo) - 1), high = g >> a.b_lo;
  const size_t base = (high << b_hi) | low;
  u64* dst = a.out + (size_t)col * a.out_stride + (size_t)rev_bits(co, a.log_cosets) * n;
  const u64* src = a.first ? a.in + (size_t)col * a.in_stride : dst;
  const u64* pre = (a.first && a.pre) ? a.pre + (size_t)co * n : nullptr;
  u64 x[R];
#pragma unroll
  for (int k = 0; k < R; k++) { size_t i = base + ((size_t)k << a.b_lo); u64 v = src[i]; if (pre) v = mulc(v, pre[i]); x[k] = v; }
  // radix-2^M DIF with constant twiddles W_R^j = W16^(j * 16/R); outputs land in bit-reversed register order
#pragma unroll
  for (int lvl = 0; lvl < M; lvl++) {
    const int half = R >> (lvl + 1);
#pragma unroll
    for (int blk = 0; blk < R; blk += 2 * half)
#pragma unroll
      for (int k = 0; k < half; k++) {
        u64 p = x[blk + k], q = x[blk + k + half];
        x[blk + k] = gl_add(p, q);
        u64 d = gl_sub(p, q);
        const int tw = (k << lvl) * (16 / R);            // exponent of W16
        x[blk + k + half] = tw ? mulc(d, a.w16.w[tw]) : d;
      }
  }
  const u64 unit = (u64)low << (a.log_M - b_hi);
  const u64 Mfull = 1ULL << a.log_M;
#pragma unroll
  for (int r = 0; r < R; r++) {
    u64 v = x[r];
    if (r) { u64 e = unit * rev_m(r, M); if (a.inverse) e = Mfull - e; v = mulc(v, root_pow2w(a.rt, a.log_M, e)); }
    dst[base + ((size_t)r << a.b_lo)] = v;
  }
}

__device__ __forceinline__ unsigned lpos8(unsigned p) { return p + ((p >> 8) << 4); }
__global__ void __launch_bounds__(256) k_ntt_strided8(StridedArgs a) {
  __shared__ u64 sm[4096 + 256];
  const size_t n = size_t(1) << a.log_n;
  const unsigned tid = blockIdx.y, col = tid >> a.log_cosets, co = tid & ((1u << a.log_cosets) - 1);
  const int b_hi = a.b_lo + 8;
  // tile: 16 consecutive low indices x all 256 h values of one sub-transform
  const size_t tile = blockIdx.x, lows_per_sub = size_t(1) << (a.b_lo - 4);
  const size_t high = tile / lows_per_sub, low_base = (tile % lows_per_sub) << 4;
  const unsigned l = threadIdx.x & 15, hs = threadIdx.x >> 4;
  const size_t low = low_base + l, base = (high << b_hi) | low;
  u64* dst = a.out + (size_t)col * a.out_stride + (size_t)rev_bits(co, a.log_cosets) * n;
  const u64* src = a.first ? a.in + (size_t)col * a.in_stride : dst;
  const u64* pre = (a.first && a.pre) ? a.pre + (size_t)co * n : nullptr;
  const u64 Mfull = 1ULL << a.log_M;
  u64 x[16];
#pragma unroll
  for (int k = 0; k < 16; k++) { size_t i = base + ((size_t)(k * 16 + hs) << a.b_lo); u64 v = src[i]; if (pre) v = mulc(v, pre[i]); x[k] = v; }
  dft16(x, a.w16);
  {   // internal twiddle w_256^(hs * f), f = rev4(r)
    const u64 unit = (u64)hs << (a.log_M - 8);
#pragma unroll
    for (int r = 1; r < 16; r++) {
      const unsigned f = ((r & 1) << 3) | ((r & 2) << 1) | ((r & 4) >> 1) | ((r & 8) >> 3);
      u64 e = unit * f; if (a.inverse) e = Mfull - e;
      x[r] = mulc(x[r], root_pow16(a.rt, a.log_M, e));
    }
  }
#pragma unroll
  for (int r = 0; r < 16; r++) sm[lpos8(((unsigned)(r * 16 + hs) << 4) | l)] = x[r];
  __syncthreads();
  const unsigned B = hs;       // second step: block B of 16 consecutive h' values
#pragma unroll
  for (int k = 0; k < 16; k++) x[k] = sm[lpos8(((B * 16 + k) << 4) | l)];
  dft16(x, a.w16);
  const u64 unit = (u64)low << (a.log_M - b_hi);
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const unsigned hpp = B * 16 + r, F = rev_m(hpp, 8);
    u64 v = x[r];
    if (F) { u64 e = unit * F; if (a.inverse) e = Mfull - e; v = mulc(v, root_pow2w(a.rt, a.log_M, e)); }
    dst[base + ((size_t)hpp << a.b_lo)] = v;
  }
}

bool ntt16_supported(int log_n) {
  const int mode = opts().ntt16;   // 0 off, 1 default, 2 forced up to 2^14 (A/B switch for experiments)
  // Measured on MI355X (profiles/r01 notes): at 2^13/2^14 the 64-139 KB LDS image allows one or two workgroups per CU, which
  // gains nothing for a single proof (5.18 vs 5.14 ms) and costs 12 % of stream-pipelined throughput (240 vs 273 proofs/s)
  // because it evicts co-resident kernels of other proofs. Up to 2^12 (35 KB, 256 threads) it is 1.6-1.8x faster than the
  // tiled kernel, so that is where it is used (FRI layers, small circuits); QPGPU_NTT16=2 forces it up to 2^14.
  return mode != 0 && log_n >= 9 && log_n <= (mode == 2 ? 14 : 12);
}

// Forward LDE for 2^13 <= n <= 2^24: strided passes down to a 2^9..2^12 block size, then k_ntt16 on the contiguous blocks.
bool lde_strided_supported(int log_n) {
  return opts().ntt_strided != 0 && log_n >= 13 && log_n <= 24;
}
void lde_strided_run(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets, const u64* pre,
                     const NttPlan& pl, hipStream_t s, bool inverse) {
  if (count <= 0) return;
  StridedArgs a; a.in = in; a.in_stride = in_stride; a.out = out; a.out_stride = out_stride; a.pre = pre; a.rt = pl.rt; a.log_M = pl.log_M;
  a.log_n = log_n; a.log_cosets = log_cosets; a.inverse = inverse ? 1 : 0;
  { u64 w = gl_root(4); if (inverse) w = gl_inv(w); u64 x = 1; for (int j = 0; j < 8; j++) { a.w16.w[j] = x; x = gl_mul(x, w); } }
  const size_t n = size_t(1) << log_n; const unsigned transforms = (unsigned)(count << log_cosets);
  int rem = log_n; bool first = true;
  while (rem > 12) {
    const int use_m8 = opts().ntt_m8;
    int M = rem - 12 <= 4 ? rem - 12 : (rem - 9 < 8 ? rem - 9 : 8);
    if (M > 4 && (M < 8 || !use_m8)) M = 4;          // only M <= 4 (registers) and M = 8 (one LDS exchange) exist
    a.b_lo = rem - M; a.first = first ? 1 : 0;
    if (M == 8) hipLaunchKernelGGL(k_ntt_strided8, dim3((unsigned)(n >> 12), transforms), dim3(256), 0, s, a);
    else {
      dim3 grid((unsigned)(((n >> M) + 255) / 256), transforms);
      switch (M) {
        case 1: hipLaunchKernelGGL(k_ntt_strided_reg<1>, grid, dim3(256), 0, s, a); break;
        case 2: hipLaunchKernelGGL(k_ntt_strided_reg<2>, grid, dim3(256), 0, s, a); break;
        case 3: hipLaunchKernelGGL(k_ntt_strided_reg<3>, grid, dim3(256), 0, s, a); break;
        default: hipLaunchKernelGGL(k_ntt_strided_reg<4>, grid, dim3(256), 0, s, a); break;
      }
    }
    rem -= M; first = false;
  }
  // remaining contiguous blocks of 2^rem: pure DIF of that size each, in place
  Ntt16Args b; b.in = out; b.out = out; b.pre = nullptr; b.rt = pl.rt; b.log_M = pl.log_M; b.scale = 1; b.log_cosets = 0; b.inverse = a.inverse; b.out_natural = 0;
  b.w16 = a.w16;
  // every 2^rem block of every (column, coset) output is one transform; with contiguous columns all blocks are evenly spaced
  const size_t blocks_per_col = (n << log_cosets) >> rem;
  const bool contiguous = out_stride == (n << log_cosets);
  const int launches = contiguous ? 1 : count;
  for (int cidx = 0; cidx < launches; cidx++) {
    b.out = out + (size_t)cidx * out_stride; b.in = b.out; b.in_stride = size_t(1) << rem; b.out_stride = size_t(1) << rem;
    const int tr = (int)(contiguous ? blocks_per_col * (size_t)count : blocks_per_col);
    switch (rem) {
      case 9: launch16<9>(b, tr, s); break;
      case 10: launch16<10>(b, tr, s); break;
      case 11: launch16<11>(b, tr, s); break;
      default: launch16<12>(b, tr, s); break;
    }
  }
}

void ntt16_run(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets, bool inverse,
               bool out_natural, const u64* pre, u64 scale, const NttPlan& pl, hipStream_t s) {
  if (count <= 0) return;
  Ntt16Args a; a.in = in; a.in_stride = in_stride; a.out = out; a.out_stride = out_stride; a.pre = pre; a.rt = pl.rt; a.log_M = pl.log_M;
  a.scale = scale; a.log_cosets = log_cosets; a.inverse = inverse; a.out_natural = out_natural;
  u64 w = gl_root(4); if (inverse) w = gl_inv(w);
  u64 x = 1; for (int j = 0; j < 8; j++) { a.w16.w[j] = x; x = gl_mul(x, w); }
  const int transforms = count << log_cosets;
  switch (log_n) {
    case 9: launch16<9>(a, transforms, s); break;
    case 10: launch16<10>(a, transforms, s); break;
    case 11: launch16<11>(a, transforms, s); break;
    case 12: launch16<12>(a, transforms, s); break;
    case 13: launch16<13>(a, transforms, s); break;
    case 14: launch16<14>(a, transforms, s); break;
    default: break;
  }
}

} // namespace qp
