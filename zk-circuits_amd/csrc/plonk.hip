// PLONK prover stages on gfx950: Z / partial products, quotient evaluation (a8), openings (a9) —
// the bodies of qp-plonky2's wires_permutation_partial_products_and_zs, compute_quotient_polys /
// eval_vanishing_poly_base_batch and OpeningSet::new, behind wormhole/prover/src/lib.rs:233-237.
// All oracles are column-major ([column][row]) so every per-row/per-point kernel reads coalesced.
#include "kernels.h"
#include "poseidon.h"
#include "poseidon_dev.h"
#include "../../include/qpgpu.h"

namespace qp {

__device__ __forceinline__ u64 root_pow_d(const u64* rt, int log_M, u64 e) {
  u64 M = 1ULL << log_M; e &= (M - 1); u64 half = M >> 1;
  return e < half ? rt[e] : gl_neg(rt[e - half]);
}

// ------------------------------------------------------------------ Z / partial products
// 1. k_zpp_chunks: one thread per (4 rows, chunk of `qdf` routed wires, challenge): numerator / denominator products and one
//    shared inversion (Montgomery trick over the 4 rows) -> chunk quotients q[c][row][chunk].
// 2. k_zpp_scan_local: blocks of 2048 rows: row totals, block-local exclusive prefix products, block totals.
// 3. k_zpp_finish: every block multiplies in the product of the preceding block totals (n/512 of them) and writes Z and
//    the running partial products. No single-block pass: the old one-block scan was 34 us of strided 8-byte loads.
static constexpr int ZPP_ROWS_PER_THREAD = 2;
static constexpr int SCAN_ROWS = 2;          // rows per thread in the scan kernels (small: the scan is latency-bound, not work-bound)
static constexpr int SCAN_BLOCK = 256 * SCAN_ROWS;   // rows per scan block

__global__ void __launch_bounds__(256) k_zpp_chunks(CircuitDev cd, const u64* __restrict__ wires, const u64* __restrict__ sigmas, Chal2 betas,
                                                    Chal2 gammas, u64* __restrict__ q, const u64* __restrict__ rt, int log_M) {
  const size_t n = size_t(1) << cd.log_n, quarter = n / ZPP_ROWS_PER_THREAD;
  const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t >= quarter) return;
  const int k = blockIdx.y, c = blockIdx.z, nchunks = cd.num_pp + 1, deg = cd.qdf;
  const u64 beta = betas.v[c], gamma = gammas.v[c];
  u64 nums[ZPP_ROWS_PER_THREAD], dens[ZPP_ROWS_PER_THREAD];
#pragma unroll
  for (int r = 0; r < ZPP_ROWS_PER_THREAD; r++) {
    const size_t i = t + (size_t)r * quarter;
    const u64 x = root_pow_d(rt, log_M, (u64)i << (log_M - cd.log_n));
    const u64 bx = gl_mul(beta, x);
    u64 nu = 1, de = 1;
    for (int j = k * deg; j < (k + 1) * deg && j < cd.num_routed; j++) {
      u64 w = wires[(size_t)j * n + i], sg = sigmas[(size_t)j * n + i];
      nu = gl_mul(nu, gl_add(gl_add(w, gl_mul(bx, cd.k_is[j])), gamma));
      de = gl_mul(de, gl_add(gl_add(w, gl_mul(beta, sg)), gamma));
    }
    nums[r] = nu; dens[r] = de;
  }
  u64 pref[ZPP_ROWS_PER_THREAD]; u64 acc = 1;
#pragma unroll
  for (int r = 0; r < ZPP_ROWS_PER_THREAD; r++) { pref[r] = acc; acc = gl_mul(acc, dens[r]); }
  u64 inv = gl_inv(acc);
#pragma unroll
  for (int r = ZPP_ROWS_PER_THREAD - 1; r >= 0; r--) {
    u64 dinv = gl_mul(inv, pref[r]); inv = gl_mul(inv, dens[r]);
    const size_t i = t + (size_t)r * quarter;
    q[((size_t)c * n + i) * nchunks + k] = gl_mul(nums[r], dinv);
  }
}

// small-n fallback of step 1 (n < 4): one thread per (row, chunk)
__global__ void k_zpp_chunks_small(CircuitDev cd, const u64* wires, const u64* sigmas, Chal2 betas, Chal2 gammas, u64* q, const u64* rt, int log_M) {
  const size_t n = size_t(1) << cd.log_n;
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = blockIdx.y, c = blockIdx.z, nchunks = cd.num_pp + 1, deg = cd.qdf;
  const u64 beta = betas.v[c], gamma = gammas.v[c];
  const u64 bx = gl_mul(beta, root_pow_d(rt, log_M, (u64)i << (log_M - cd.log_n)));
  u64 nu = 1, de = 1;
  for (int j = k * deg; j < (k + 1) * deg && j < cd.num_routed; j++) {
    u64 w = wires[(size_t)j * n + i], sg = sigmas[(size_t)j * n + i];
    nu = gl_mul(nu, gl_add(gl_add(w, gl_mul(bx, cd.k_is[j])), gamma));
    de = gl_mul(de, gl_add(gl_add(w, gl_mul(beta, sg)), gamma));
  }
  q[((size_t)c * n + i) * nchunks + k] = gl_mul(nu, gl_inv(de));
}

// block-local exclusive prefix products of the row totals; loc[c][i], btot[c][block]
__global__ void __launch_bounds__(256) k_zpp_scan_local(CircuitDev cd, const u64* __restrict__ q, u64* __restrict__ loc, u64* __restrict__ btot) {
  __shared__ u64 sm[256];
  const size_t n = size_t(1) << cd.log_n;
  const int c = blockIdx.y, nchunks = cd.num_pp + 1;
  const size_t base = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)threadIdx.x * SCAN_ROWS;
  u64 tot[SCAN_ROWS]; u64 acc = 1;
#pragma unroll
  for (int r = 0; r < SCAN_ROWS; r++) {
    const size_t i = base + r; u64 tr = 1;
    if (i < n) { const u64* qr = q + ((size_t)c * n + i) * nchunks; for (int k = 0; k < nchunks; k++) tr = gl_mul(tr, qr[k]); }
    tot[r] = tr; acc = gl_mul(acc, tr);
  }
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    u64 v = threadIdx.x >= off ? sm[threadIdx.x - off] : 1;
    __syncthreads();
    sm[threadIdx.x] = gl_mul(sm[threadIdx.x], v);
    __syncthreads();
  }
  u64 run = threadIdx.x ? sm[threadIdx.x - 1] : 1;
#pragma unroll
  for (int r = 0; r < SCAN_ROWS; r++) { const size_t i = base + r; if (i < n) loc[(size_t)c * n + i] = run; run = gl_mul(run, tot[r]); }
  if (threadIdx.x == 255) btot[(size_t)c * gridDim.x + blockIdx.x] = sm[255];
}

__global__ void __launch_bounds__(256) k_zpp_finish(CircuitDev cd, const u64* __restrict__ q, const u64* __restrict__ loc, const u64* __restrict__ btot,
                                                    u64* __restrict__ zpp) {
  __shared__ u64 sm[256];
  const size_t n = size_t(1) << cd.log_n;
  const int c = blockIdx.y, nchunks = cd.num_pp + 1, nblk = gridDim.x;
  // product of the totals of all blocks before this one
  u64 acc = 1;
  for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) acc = gl_mul(acc, btot[(size_t)c * nblk + b]);
  sm[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) sm[threadIdx.x] = gl_mul(sm[threadIdx.x], sm[threadIdx.x + off]); __syncthreads(); }
  const u64 offset = sm[0];
  for (int r = 0; r < SCAN_ROWS; r++) {
    const size_t i = (size_t)blockIdx.x * SCAN_BLOCK + (size_t)r * 256 + threadIdx.x;      // coalesced over the block's rows
    if (i >= n) continue;
    u64 a = gl_mul(offset, loc[(size_t)c * n + i]);
    zpp[(size_t)c * n + i] = a;
    const u64* qr = q + ((size_t)c * n + i) * nchunks;
    for (int k = 0; k < cd.num_pp; k++) { a = gl_mul(a, qr[k]); zpp[(size_t)(cd.num_challenges + c * cd.num_pp + k) * n + i] = a; }
  }
}

void zpp_compute(const CircuitDev& cd, const u64* wires_vals, const u64* sigma_vals, Chal2 betas, Chal2 gammas, u64* scratch,
                 u64* zpp_vals, const NttPlan& pl, hipStream_t s) {
  const size_t n = size_t(1) << cd.log_n; const int nch = cd.num_challenges, nchunks = cd.num_pp + 1;
  u64* q = scratch; u64* loc = scratch + (size_t)nch * n * nchunks; u64* btot = loc + (size_t)nch * n;
  if (n >= ZPP_ROWS_PER_THREAD * 64) {
    const size_t quarter = n / ZPP_ROWS_PER_THREAD;
    hipLaunchKernelGGL(k_zpp_chunks, dim3((unsigned)((quarter + 255) / 256), (unsigned)nchunks, (unsigned)nch), dim3(256), 0, s, cd, wires_vals, sigma_vals,
                       betas, gammas, q, pl.rt, pl.log_M);
  } else {
    hipLaunchKernelGGL(k_zpp_chunks_small, dim3((unsigned)((n + 63) / 64), (unsigned)nchunks, (unsigned)nch), dim3(64), 0, s, cd, wires_vals, sigma_vals,
                       betas, gammas, q, pl.rt, pl.log_M);
  }
  const unsigned nblk = (unsigned)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  hipLaunchKernelGGL(k_zpp_scan_local, dim3(nblk, (unsigned)nch), dim3(256), 0, s, cd, q, loc, btot);
  hipLaunchKernelGGL(k_zpp_finish, dim3(nblk, (unsigned)nch), dim3(256), 0, s, cd, q, loc, btot, zpp_vals);
}

// ------------------------------------------------------------------ quotient
__global__ void k_alpha_pows(u64* out, Chal2 alphas, int count) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[(size_t)blockIdx.y * count + i] = gl_pow(alphas.v[blockIdx.y], (u64)i);
}
void alpha_powers(u64* out, Chal2 alphas, int nch, int count, hipStream_t s) {
  hipLaunchKernelGGL(k_alpha_pows, dim3((count + 63) / 64, nch), dim3(64), 0, s, out, alphas, count);
}

struct Acc2 { u64 a0, a1; };
// acc += term * alpha^idx for both challenges
#define ACC_TERM(term_, idx_) do { u64 t__ = (term_); int i__ = (idx_); \
    acc_mul(g0, t__, ap0[i__]); acc_mul(g1, t__, ap1[i__]); } while (0)
// lazy a - b for lazy a (any u64) and canonical b
__device__ __forceinline__ u64 sub_lazy_canon(u64 a, u64 b) { u64 d = a - b; if (a < b) d += GL_P; return d; }

struct Pi4 { u64 v[4]; };
struct Zh8 { u64 zh[8]; u64 zh_inv[8]; };

// PART 0: everything, output = vanishing / Z_H. PART 1: L_0 and permutation terms only, PART 2: gate constraints only; both write
// their raw alpha-weighted sums to `out` ([nch][N]); k_quotient_sum adds the two and divides by Z_H. The halves are independent,
// so a single proof runs them on two streams at once (each alone is only 2 waves/SIMD and stalls on its ~200-450 loads per point).
// LS (leaf gate set, PART 0): the circuit's gates are exactly Noop / Constant{2} / PublicInput / BaseSum{63} / Arithmetic{20} / Poseidon with
// 80 routed wires in chunks of 8. The four cheap gates then take their wires from the permutation argument's chunk loads (each of wires
// 0..79 is loaded once for both) instead of loading them again gate by gate: 323 column loads per point instead of 473 (241 algorithmic).
template <int PART, bool LS = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_quotient(CircuitDev cd, const u64* __restrict__ cs, const u64* __restrict__ wl,
                                                  const u64* __restrict__ zl, Chal2 betas, Chal2 gammas,
                                                  const u64* __restrict__ apw, int ap_stride, Pi4 pih, Zh8 zh, u64 n_inv,
                                                  u64* __restrict__ out, const u64* __restrict__ rt, int log_M) {
  const int log_N = cd.log_n + cd.rate_bits;
  const size_t N = size_t(1) << log_N;
  const size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (L >= N) return;
  const u32 j = rev_bits((u32)L, log_N);
  const size_t Ln = rev_bits((u32)((j + (1u << cd.rate_bits)) & (N - 1)), log_N);
  const u64 x = gl_mul(GL_GEN, root_pow_d(rt, log_M, (u64)j << (log_M - log_N)));
  const int NC = cd.num_constants, R = cd.num_routed, nch = cd.num_challenges, npp = cd.num_pp, deg = cd.qdf;
  const u64* ap0 = apw; const u64* ap1 = apw + ap_stride;
  auto Wv = [&](int k) -> u64 { return wl[(size_t)k * N + L]; };
  u64 tot0 = 0, tot1 = 0;   // running sums for the two challenges
  int term = 0;
  // ---- L0(x) (Z_c(x) - 1)
  if (PART != 2) {
    u64 zhx = zh.zh[j & ((1u << cd.rate_bits) - 1)];
    const u64 l0 = cd.l0_lde ? cd.l0_lde[L] : gl_mul(zhx, gl_mul(n_inv, gl_inv(gl_sub(x, 1))));
    for (int c = 0; c < nch; c++) {
      u64 t = gl_mul(l0, gl_sub(zl[(size_t)c * N + L], 1));
      tot0 = gl_add(tot0, gl_mul(t, ap0[term])); tot1 = gl_add(tot1, gl_mul(t, ap1[term])); term++;
    }
  }
  // ---- LS: filters of the four cheap gates, their accumulators (alpha powers of the gate section start behind the permutation terms)
  const int nsel_ = cd.num_selectors;
  u64 f_cst = 0, f_pi = 0, f_bs = 0, f_ar = 0, ls_lc0 = 0, ls_lc1 = 0, bs_w0 = 0, bs_sum = 0;
  Acc192 m0{0, 0, 0}, m1{0, 0, 0}, b0{0, 0, 0}, b1{0, 0, 0}, a0{0, 0, 0}, a1{0, 0, 0};      // misc (Constant, PublicInput: filter folded in), BaseSum, Arithmetic
  const u64* gp0 = ap0 + (nch + nch * (npp + 1)); const u64* gp1 = ap1 + (nch + nch * (npp + 1));
  if (LS && PART == 0) {
    ls_lc0 = cs[(size_t)nsel_ * N + L]; ls_lc1 = cs[(size_t)(nsel_ + 1) * N + L];
    for (int gi = 0; gi < cd.num_gates; gi++) {
      const GateDev g = cd.gates[gi];
      if (g.tag == QPGPU_GATE_NOOP || g.tag == QPGPU_GATE_POSEIDON) continue;
      const u64 sv = cs[(size_t)g.sel_index * N + L];
      u64 filter = 1;
      for (u32 i = g.grp_lo; i < g.grp_hi; i++) if ((int)i != gi) filter = gl_mul(filter, gl_sub((u64)i, sv));
      if (nsel_ > 1) filter = gl_mul(filter, gl_sub(0xFFFFFFFFULL, sv));
      if (g.tag == QPGPU_GATE_CONSTANT) f_cst = filter; else if (g.tag == QPGPU_GATE_PUBLIC_INPUT) f_pi = filter;
      else if (g.tag == QPGPU_GATE_BASESUM2) f_bs = filter; else f_ar = filter;
    }
  }
  // ---- permutation checks: both challenges share the loads of w_j / sigma_j; factors and running products are lazy residues,
  // the alpha-weighted terms go into one 192-bit accumulator per challenge
  if (PART != 2) {
    Acc192 pa[2] = {{0, 0, 0}, {0, 0, 0}};
    u64 prev[2] = {0, 0}, bx[2] = {0, 0};
    for (int c = 0; c < nch; c++) { prev[c] = zl[(size_t)c * N + L]; bx[c] = gl_mul(betas.v[c], x); }
    for (int k = 0; k <= npp; k++) {
      u64 nu[2] = {1, 1}, de[2] = {1, 1};
      // a chunk's loads are issued together (deg <= 8): with 2 waves/SIMD the kernel depends on memory-level parallelism
      u64 wv[8], sgv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { const int jj = k * deg + u; const bool ok = u < deg && jj < R; wv[u] = ok ? Wv(jj) : 0; sgv[u] = ok ? cs[(size_t)(NC + jj) * N + L] : 0; }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int jj = k * deg + u;
        if (!(u < deg && jj < R)) continue;
        const u64 w = wv[u], sg = sgv[u], kj = cd.k_is[jj];
        for (int c = 0; c < nch; c++) {
          nu[c] = mul_lazy(nu[c], add_lazy_canon(add_lazy_canon(mul_lazy(bx[c], kj), w), gammas.v[c]));
          de[c] = mul_lazy(de[c], add_lazy_canon(add_lazy_canon(mul_lazy(betas.v[c], sg), w), gammas.v[c]));
        }
      }
      if (LS && PART == 0) {                               // the cheap gates on the chunk's wires 8k .. 8k+7
        if (k == 0) {
          for (int i = 0; i < 2; i++) { const u64 t = gl_mul(f_cst, gl_sub(i == 0 ? ls_lc0 : ls_lc1, wv[i])); acc_mul(m0, t, gp0[i]); acc_mul(m1, t, gp1[i]); }
          for (int i = 0; i < 4; i++) { const u64 t = gl_mul(f_pi, gl_sub(wv[i], pih.v[i])); acc_mul(m0, t, gp0[i]); acc_mul(m1, t, gp1[i]); }
          bs_w0 = wv[0];
        }
        if (k < 8) {                                       // BaseSum: limb l sits on wire 1 + l; constraint 1 + l = limb (limb - 1); sum of limb 2^l against wire 0
          u64 local = 0;
#pragma unroll
          for (int u = 7; u >= 0; u--) {
            if (k == 0 && u == 0) continue;
            const u64 b = wv[u];
            local = gl_add(gl_add(local, local), b);
            const u64 t = gl_mul(b, gl_sub(b, 1)); const int idx = 8 * k + u;
            acc_mul(b0, t, gp0[idx]); acc_mul(b1, t, gp1[idx]);
          }
          bs_sum = gl_add(bs_sum, gl_mul(local, k == 0 ? 1 : (u64(1) << (8 * k - 1))));      // chunk 0's Horner already starts at limb 0 (wire 1)
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {                      // Arithmetic: operations 2k, 2k + 1
          const u64 t = gl_sub(wv[4 * u + 3], gl_add(gl_mul(ls_lc0, gl_mul(wv[4 * u], wv[4 * u + 1])), gl_mul(ls_lc1, wv[4 * u + 2])));
          acc_mul(a0, t, gp0[2 * k + u]); acc_mul(a1, t, gp1[2 * k + u]);
        }
      }
      for (int c = 0; c < nch; c++) {
        const u64 next = k < npp ? zl[(size_t)(nch + c * npp + k) * N + L] : zl[(size_t)c * N + Ln];
        const u64 t = sub_lazy_canon(mul_lazy(prev[c], nu[c]), canon(mul_lazy(next, de[c])));
        const int idx = term + c * (npp + 1) + k;          // every term enters BOTH alpha combinations
        acc_mul(pa[0], t, ap0[idx]); acc_mul(pa[1], t, ap1[idx]);
        prev[c] = next;
      }
    }
    tot0 = gl_add(tot0, acc_reduce(pa[0]));
    if (nch > 1) tot1 = gl_add(tot1, acc_reduce(pa[1]));
    if (LS && PART == 0) {
      { const u64 t = gl_sub(bs_sum, bs_w0); acc_mul(b0, t, gp0[0]); acc_mul(b1, t, gp1[0]); }
      tot0 = gl_add(tot0, gl_add(acc_reduce(m0), gl_add(gl_mul(f_bs, acc_reduce(b0)), gl_mul(f_ar, acc_reduce(a0)))));
      if (nch > 1) tot1 = gl_add(tot1, gl_add(acc_reduce(m1), gl_add(gl_mul(f_bs, acc_reduce(b1)), gl_mul(f_ar, acc_reduce(a1)))));
    }
  }
  term = nch + nch * (npp + 1);
  // ---- gate constraints: sum_g filter_g * sum_j alpha^(term+j) c_{g,j}
  ap0 += term; ap1 += term;
  const int nsel = cd.num_selectors;
  const u64 lc0 = cs[(size_t)nsel * N + L], lc1 = cs[(size_t)(nsel + 1) * N + L];
  for (int gi = 0; PART != 1 && gi < cd.num_gates; gi++) {
    const GateDev g = cd.gates[gi];
    if (LS && PART == 0 && g.tag != QPGPU_GATE_POSEIDON) continue;       // done above, on the permutation argument's loads
    const u64 sv = cs[(size_t)g.sel_index * N + L];
    u64 filter = 1;
    for (u32 i = g.grp_lo; i < g.grp_hi; i++) if ((int)i != gi) filter = gl_mul(filter, gl_sub((u64)i, sv));
    if (nsel > 1) filter = gl_mul(filter, gl_sub(0xFFFFFFFFULL, sv));
    Acc192 g0{0, 0, 0}, g1{0, 0, 0};
    switch (g.tag) {
      case QPGPU_GATE_NOOP: break;
      case QPGPU_GATE_CONSTANT:
        for (u32 i = 0; i < (u32)g.param; i++) ACC_TERM(gl_sub(cs[(size_t)(nsel + i) * N + L], Wv(i)), i);
        break;
      case QPGPU_GATE_PUBLIC_INPUT:
        for (int i = 0; i < 4; i++) ACC_TERM(gl_sub(Wv(i), pih.v[i]), i);
        break;
      case QPGPU_GATE_BASESUM2: {
        const int limbs = (int)g.param;
        u64 sum = 0;
        for (int hi = limbs; hi > 0; hi -= 8) {                // limbs [hi-8, hi), highest first, loads issued together
          u64 bv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { const int l = hi - 1 - u; bv[u] = l >= 0 ? Wv(1 + l) : 0; }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int l = hi - 1 - u;
            if (l < 0) continue;
            const u64 b = bv[u];
            sum = gl_add(gl_add(sum, sum), b);
            ACC_TERM(gl_mul(b, gl_sub(b, 1)), 1 + l);
          }
        }
        ACC_TERM(gl_sub(sum, Wv(0)), 0);
        break;
      }
      case QPGPU_GATE_ARITHMETIC:
        for (u32 i0 = 0; i0 < (u32)g.param; i0 += 4) {         // four operations (16 loads) per batch
          u64 av[16];
#pragma unroll
          for (int u = 0; u < 16; u++) av[u] = (i0 + (u >> 2)) < (u32)g.param ? Wv(4 * i0 + u) : 0;
#pragma unroll
          for (int u = 0; u < 4; u++) {
            if (i0 + u >= (u32)g.param) continue;
            ACC_TERM(gl_sub(av[4 * u + 3], gl_add(gl_mul(lc0, gl_mul(av[4 * u], av[4 * u + 1])), gl_mul(lc1, av[4 * u + 2]))), i0 + u);
          }
        }
        break;
      case QPGPU_GATE_POSEIDON: {
        // State kept as lazy residues (poseidon_dev.h); the next round's constants ride in on the MDS accumulators, so
        // after each mds_rc_lazy the state already equals "MDS output + round constant", which is what the wires pin.
        const u64* rc = POSEIDON_RC_DEV;
        int k = 0;
        u64 swap = Wv(24);
        ACC_TERM(gl_mul(swap, gl_sub(swap, 1)), k); k++;
        u64 st[12];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          u64 a = Wv(i), b = Wv(i + 4), delta = Wv(25 + i);
          ACC_TERM(gl_sub(gl_mul(swap, gl_sub(b, a)), delta), k); k++;
          st[i] = gl_add(a, delta); st[i + 4] = gl_sub(b, delta);
        }
#pragma unroll
        for (int i = 8; i < 12; i++) st[i] = Wv(i);
#pragma unroll
        for (int i = 0; i < 12; i++) st[i] = add_lazy_canon(st[i], rc[i]);
        int round = 0;
        for (int r = 0; r < 4; r++, round++) {
          if (r != 0) {
#pragma unroll
            for (int i = 0; i < 12; i++) { u64 in = Wv(29 + 12 * (r - 1) + i); ACC_TERM(sub_lazy_canon(st[i], in), k); k++; st[i] = in; }
          }
#pragma unroll
          for (int i = 0; i < 12; i++) st[i] = sbox7_lazy(st[i]);
          mds_rc_lazy(st, rc + 12 * (round + 1));
        }
        for (int r0 = 0; r0 < 22; r0 += 11) {                  // the partial rounds' S-box input wires, eleven loads at a time
          u64 pin[11];
#pragma unroll
          for (int u = 0; u < 11; u++) pin[u] = Wv(65 + r0 + u);
          for (int u = 0; u < 11; u++, round++) {
            u64 in = pin[0];
#pragma unroll
            for (int q = 0; q < 10; q++) pin[q] = pin[q + 1];   // shift register: keeps the array in VGPRs under the rolled loop
            ACC_TERM(sub_lazy_canon(st[0], in), k); k++;
            st[0] = sbox7_lazy(in);
            mds_rc_lazy(st, rc + 12 * (round + 1));
          }
        }
        for (int r = 0; r < 4; r++, round++) {
#pragma unroll
          for (int i = 0; i < 12; i++) { u64 in = Wv(87 + 12 * r + i); ACC_TERM(sub_lazy_canon(st[i], in), k); k++; st[i] = in; }
#pragma unroll
          for (int i = 0; i < 12; i++) st[i] = sbox7_lazy(st[i]);
          if (r < 3) mds_rc_lazy(st, rc + 12 * (round + 1)); else mds_rc_lazy(st, nullptr);
        }
#pragma unroll
        for (int i = 0; i < 12; i++) { ACC_TERM(sub_lazy_canon(st[i], Wv(12 + i)), k); k++; }
        break;
      }
      // ---- recursive-verifier circuit gates (layouts are the recalled upstream ones, parity unpinned).
      // Extension values are pairs of base-field wires; on the coset the algebra is plain F_{p^2}.
      case QPGPU_GATE_ARITHMETIC_EXTENSION:
        for (u32 i = 0; i < (u32)g.param; i++) {
          const Ext m0 = ex(Wv(8 * i), Wv(8 * i + 1)), m1 = ex(Wv(8 * i + 2), Wv(8 * i + 3)), ad = ex(Wv(8 * i + 4), Wv(8 * i + 5));
          const Ext t = ex_sub(ex(Wv(8 * i + 6), Wv(8 * i + 7)), ex_add(ex_scale(ex_mul(m0, m1), lc0), ex_scale(ad, lc1)));
          ACC_TERM(t.a, 2 * i); ACC_TERM(t.b, 2 * i + 1);
        }
        break;
      case QPGPU_GATE_MUL_EXTENSION:
        for (u32 i = 0; i < (u32)g.param; i++) {
          const Ext m0 = ex(Wv(6 * i), Wv(6 * i + 1)), m1 = ex(Wv(6 * i + 2), Wv(6 * i + 3));
          const Ext t = ex_sub(ex(Wv(6 * i + 4), Wv(6 * i + 5)), ex_scale(ex_mul(m0, m1), lc0));
          ACC_TERM(t.a, 2 * i); ACC_TERM(t.b, 2 * i + 1);
        }
        break;
      case QPGPU_GATE_REDUCING:
      case QPGPU_GATE_REDUCING_EXTENSION: {
        const bool xc = g.tag == QPGPU_GATE_REDUCING_EXTENSION;
        const int nc = (int)g.param, start_accs = 6 + (xc ? 2 : 1) * nc;
        const Ext alpha = ex(Wv(2), Wv(3));
        Ext acc = ex(Wv(4), Wv(5));
        for (int i0 = 0; i0 < nc; i0 += 4) {                    // four steps' loads (up to 16) issued together
          u64 ca[4], cb[4], aa[4], ab[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + u; const bool ok = i < nc;
            ca[u] = ok ? (xc ? Wv(6 + 2 * i) : Wv(6 + i)) : 0; cb[u] = (ok && xc) ? Wv(7 + 2 * i) : 0;
            const int ai = i == nc - 1 ? 0 : start_accs + 2 * i;
            aa[u] = ok ? Wv(ai) : 0; ab[u] = ok ? Wv(ai + 1) : 0;
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int i = i0 + u;
            if (i >= nc) continue;
            const Ext acc_i = ex(aa[u], ab[u]);
            const Ext t = ex_sub(ex_add(ex_mul(acc, alpha), ex(ca[u], cb[u])), acc_i);
            ACC_TERM(t.a, 2 * i); ACC_TERM(t.b, 2 * i + 1);
            acc = acc_i;
          }
        }
        break;
      }
      case QPGPU_GATE_RANDOM_ACCESS: {
        const int bits = (int)g.param, copies = (int)g.p2, extra = (int)g.p3, vec = 1 << bits;   // bits <= 4 (checked at creation)
        const int routed = (2 + vec) * copies + extra;
        int k = 0;
        for (int c = 0; c < copies; c++) {
          const int base = (2 + vec) * c;
          u64 b[4], idx = 0;
#pragma unroll
          for (int i = 0; i < 4; i++) b[i] = i < bits ? Wv(routed + c * bits + i) : 0;
#pragma unroll
          for (int i = 0; i < 4; i++) if (i < bits) { ACC_TERM(gl_mul(b[i], gl_sub(b[i], 1)), k); k++; }
#pragma unroll
          for (int i = 3; i >= 0; i--) if (i < bits) idx = gl_add(gl_add(idx, idx), b[i]);
          ACC_TERM(gl_sub(idx, Wv(base)), k); k++;
          u64 list[16];
#pragma unroll
          for (int i = 0; i < 16; i++) list[i] = i < vec ? Wv(base + 2 + i) : 0;
#pragma unroll
          for (int lvl = 0; lvl < 4; lvl++) if (lvl < bits) {
#pragma unroll
            for (int jj = 0; jj < (8 >> lvl); jj++) list[jj] = gl_add(list[2 * jj], gl_mul(b[lvl], gl_sub(list[2 * jj + 1], list[2 * jj])));
          }
          ACC_TERM(gl_sub(list[0], Wv(base + 1)), k); k++;
        }
        for (int i = 0; i < extra; i++) { ACC_TERM(gl_sub(cs[(size_t)(nsel + i) * N + L], Wv((2 + vec) * copies + i)), k); k++; }
        break;
      }
      case QPGPU_GATE_POSEIDON_MDS: {
        // the small-constant MDS acts on the two coordinates independently
#pragma unroll 1
        for (int h = 0; h < 2; h++) {
          u64 st[12];
#pragma unroll
          for (int i = 0; i < 12; i++) st[i] = Wv(2 * i + h);
          mds_rc_lazy(st, nullptr);
#pragma unroll
          for (int i = 0; i < 12; i++) ACC_TERM(gl_sub(Wv(24 + 2 * i + h), canon(st[i])), 2 * i + h);
        }
        break;
      }
      case QPGPU_GATE_EXPONENTIATION: {
        const int nb = (int)g.param;
        const u64 base = Wv(0);
        u64 prev = 1;
        for (int i0 = 0; i0 < nb; i0 += 8) {                    // eight steps' loads issued together
          u64 bv[8], iv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) { const int i = i0 + u; bv[u] = i < nb ? Wv(1 + (nb - 1 - i)) : 0; iv[u] = i < nb ? Wv(2 + nb + i) : 0; }
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int i = i0 + u;
            if (i >= nb) continue;
            ACC_TERM(gl_sub(gl_mul(prev, gl_add(gl_mul(bv[u], base), gl_sub(1, bv[u]))), iv[u]), i);
            prev = gl_sqr(iv[u]);
          }
        }
        ACC_TERM(gl_sub(Wv(1 + nb), Wv(2 + nb + nb - 1)), nb);
        break;
      }
      case QPGPU_GATE_COSET_INTERPOLATION: {
        const int bits = (int)g.param, dg = (int)g.p2, np = 1 << bits, ni = (np - 2) / (dg - 1);
        const int s_point = 1 + 2 * np, s_value = s_point + 2, s_inter = s_value + 2, s_shifted = s_inter + 4 * ni;
        const Ext xs = ex(Wv(s_shifted), Wv(s_shifted + 1));
        int k = 0;
        { const Ext t = ex_sub(ex(Wv(s_point), Wv(s_point + 1)), ex_scale(xs, Wv(0))); ACC_TERM(t.a, k); k++; ACC_TERM(t.b, k); k++; }
        Ext ev = ex(0), pr = ex(1);
        int lo = 0, hi = dg;
        for (int seg = 0; seg <= ni; seg++) {
          for (int i = lo; i < hi; i++) {
            const Ext term = ex(gl_sub(xs.a, cd.coset_dom[i]), xs.b);
            const Ext val = ex_scale(ex(Wv(1 + 2 * i), Wv(2 + 2 * i)), cd.coset_w[i]);
            ev = ex_add(ex_mul(ev, term), ex_mul(val, pr)); pr = ex_mul(pr, term);
          }
          if (seg == ni) break;
          const Ext iev = ex(Wv(s_inter + 2 * seg), Wv(s_inter + 2 * seg + 1)), ipr = ex(Wv(s_inter + 2 * (ni + seg)), Wv(s_inter + 2 * (ni + seg) + 1));
          { const Ext t = ex_sub(iev, ev); ACC_TERM(t.a, k); k++; ACC_TERM(t.b, k); k++; }
          { const Ext t = ex_sub(ipr, pr); ACC_TERM(t.a, k); k++; ACC_TERM(t.b, k); k++; }
          ev = iev; pr = ipr;
          lo = 1 + (dg - 1) * (seg + 1); hi = lo + dg - 1 < np ? lo + dg - 1 : np;
        }
        { const Ext t = ex_sub(ex(Wv(s_value), Wv(s_value + 1)), ev); ACC_TERM(t.a, k); k++; ACC_TERM(t.b, k); k++; }
        break;
      }
      default: break;   // rejected at circuit creation
    }
    tot0 = gl_add(tot0, gl_mul(filter, acc_reduce(g0))); tot1 = gl_add(tot1, gl_mul(filter, acc_reduce(g1)));
  }
  const u64 zi = PART == 0 ? zh.zh_inv[j & ((1u << cd.rate_bits) - 1)] : 1;
  out[L] = PART == 0 ? gl_mul(tot0, zi) : tot0;
  if (nch > 1) out[N + L] = PART == 0 ? gl_mul(tot1, zi) : tot1;
}
__global__ void __launch_bounds__(256) k_quotient_sum(int log_N, int rate_bits, int nch, const u64* __restrict__ a, const u64* __restrict__ b, Zh8 zh,
                                                      u64* __restrict__ out) {
  const size_t N = size_t(1) << log_N;
  const size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (L >= N) return;
  const u64 zi = zh.zh_inv[rev_bits((u32)L, log_N) & ((1u << rate_bits) - 1)];
  for (int c = 0; c < nch; c++) out[(size_t)c * N + L] = gl_mul(gl_add(a[(size_t)c * N + L], b[(size_t)c * N + L]), zi);
}

__global__ void __launch_bounds__(256) k_l0_table(int log_n, int rate_bits, Zh8 zh, u64 n_inv, u64* __restrict__ out, const u64* __restrict__ rt, int log_M) {
  const int log_N = log_n + rate_bits;
  const size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (L >= (size_t(1) << log_N)) return;
  const u32 j = rev_bits((u32)L, log_N);
  const u64 x = gl_mul(GL_GEN, root_pow_d(rt, log_M, (u64)j << (log_M - log_N)));
  out[L] = gl_mul(zh.zh[j & ((1u << rate_bits) - 1)], gl_mul(n_inv, gl_inv(gl_sub(x, 1))));
}
static Zh8 quotient_zh(const CircuitDev& cd);
void l0_table(const CircuitDev& cd, u64* out, const NttPlan& pl, hipStream_t s) {
  const size_t n = size_t(1) << cd.log_n, N = n << cd.rate_bits;
  hipLaunchKernelGGL(k_l0_table, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, cd.log_n, cd.rate_bits, quotient_zh(cd), gl_inv((u64)n), out, pl.rt, pl.log_M);
}
static Zh8 quotient_zh(const CircuitDev& cd) {
  Zh8 zh; const size_t n = size_t(1) << cd.log_n;
  u64 gn = gl_pow(GL_GEN, n), w = gl_root(cd.rate_bits), t = 1;
  for (int k = 0; k < 8; k++) { zh.zh[k] = 0; zh.zh_inv[k] = 0; }
  for (int k = 0; k < (1 << cd.rate_bits) && k < 8; k++) { zh.zh[k] = gl_sub(gl_mul(gn, t), 1); zh.zh_inv[k] = gl_inv(zh.zh[k]); t = gl_mul(t, w); }
  return zh;
}
// The two halves on two streams: `part` = 1 (permutation terms -> out_part) or 2 (gate terms -> out_part); then quotient_sum.
void quotient_eval_part(int part, const CircuitDev& cd, const u64* cs_lde, const u64* wires_lde, const u64* zs_lde, Chal2 betas, Chal2 gammas,
                        const u64* alpha_pows, int ap_stride, const u64* pi_hash, u64* out_part, const NttPlan& pl, hipStream_t s) {
  const int log_N = cd.log_n + cd.rate_bits; const size_t n = size_t(1) << cd.log_n, N = size_t(1) << log_N;
  Pi4 pih; for (int i = 0; i < 4; i++) pih.v[i] = pi_hash[i];
  const Zh8 zh = quotient_zh(cd);
  dim3 grid((unsigned)((N + 255) / 256));
  if (part == 1) hipLaunchKernelGGL((k_quotient<1, false>), grid, dim3(256), 0, s, cd, cs_lde, wires_lde, zs_lde, betas, gammas, alpha_pows, ap_stride, pih, zh, gl_inv((u64)n), out_part, pl.rt, pl.log_M);
  else hipLaunchKernelGGL((k_quotient<2, false>), grid, dim3(256), 0, s, cd, cs_lde, wires_lde, zs_lde, betas, gammas, alpha_pows, ap_stride, pih, zh, gl_inv((u64)n), out_part, pl.rt, pl.log_M);
}
void quotient_sum(const CircuitDev& cd, const u64* a, const u64* b, u64* out, hipStream_t s) {
  const int log_N = cd.log_n + cd.rate_bits; const size_t N = size_t(1) << log_N;
  hipLaunchKernelGGL(k_quotient_sum, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, log_N, cd.rate_bits, cd.num_challenges, a, b, quotient_zh(cd), out);
}
// The shapes k_quotient<0, true> is written for: each of the six leaf gates at most once with the standard parameters, 80 routed wires in
// ten chunks of 8 (quotient degree factor 8, nine partial products), Poseidon present.
bool quotient_leaf_set(const CircuitDev& cd) {
  if (cd.num_routed != 80 || cd.qdf != 8 || cd.num_pp != 9 || cd.num_challenges > 2 || cd.num_gate_constraints != 123) return false;
  int seen[16] = {0}; bool poseidon = false;
  for (int i = 0; i < cd.num_gates; i++) {
    const GateDev& g = cd.gates[i];
    if (g.tag >= 16 || seen[g.tag]++) return false;
    switch (g.tag) {
      case QPGPU_GATE_NOOP: case QPGPU_GATE_PUBLIC_INPUT: break;
      case QPGPU_GATE_POSEIDON: poseidon = true; break;
      case QPGPU_GATE_CONSTANT: if (g.param != 2) return false; break;
      case QPGPU_GATE_BASESUM2: if (g.param != 63) return false; break;
      case QPGPU_GATE_ARITHMETIC: if (g.param != 20) return false; break;
      default: return false;
    }
  }
  return poseidon;
}
void quotient_eval(const CircuitDev& cd, const u64* cs_lde, const u64* wires_lde, const u64* zs_lde, Chal2 betas, Chal2 gammas,
                   const u64* alpha_pows, int ap_stride, const u64* pi_hash, u64* out, const NttPlan& pl, hipStream_t s, bool leaf_gate_set) {
  const int log_N = cd.log_n + cd.rate_bits; const size_t n = size_t(1) << cd.log_n, N = size_t(1) << log_N;
  Pi4 pih; for (int i = 0; i < 4; i++) pih.v[i] = pi_hash[i];
  Zh8 zh; u64 gn = gl_pow(GL_GEN, n), w = gl_root(cd.rate_bits), t = 1;
  for (int k = 0; k < (1 << cd.rate_bits) && k < 8; k++) { zh.zh[k] = gl_sub(gl_mul(gn, t), 1); zh.zh_inv[k] = gl_inv(zh.zh[k]); t = gl_mul(t, w); }
  if (leaf_gate_set)
    hipLaunchKernelGGL((k_quotient<0, true>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, cd, cs_lde, wires_lde, zs_lde, betas, gammas,
                       alpha_pows, ap_stride, pih, zh, gl_inv((u64)n), out, pl.rt, pl.log_M);
  else
    hipLaunchKernelGGL((k_quotient<0, false>), dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, cd, cs_lde, wires_lde, zs_lde, betas, gammas,
                       alpha_pows, ap_stride, pih, zh, gl_inv((u64)n), out, pl.rt, pl.log_M);
}

// ------------------------------------------------------------------ openings
// Power tables: out[t][i] = base_t^i via the binary expansion of i against host-supplied base_t^(2^k).
__global__ void __launch_bounds__(256) k_ext_powers4(ExtPowArgs a) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= a.count) return;
  const int t = blockIdx.y;
  Ext r = ex(1, 0);
  for (int k = 0; k < a.nbits; k++) if ((i >> k) & 1) r = ex_mul(r, a.pw[t][k]);
  a.out_a[t][i] = r.a; a.out_b[t][i] = r.b;
}
void ext_powers4(const ExtPowArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_ext_powers4, dim3((unsigned)((a.count + 255) / 256), 4), dim3(256), 0, s, a);
}
__global__ void k_ext_powers(u64* oa, u64* ob, size_t count, Ext base, Ext scale) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= count) return;
  Ext r = ex_mul(scale, ex_pow(base, i));
  oa[i] = r.a; ob[i] = r.b;
}
void ext_powers(u64* out_a, u64* out_b, size_t count, Ext base, Ext scale, hipStream_t s) {
  hipLaunchKernelGGL(k_ext_powers, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, out_a, out_b, count, base, scale);
}

// All opening groups in one launch: segment g covers `count` polynomials (columns `stride` apart) evaluated with power table
// `table` (0: zeta, 1: g*zeta) and writes its results at out[first_out ...]. partial[p][chunk] = sum over the chunk of
// poly[i] * z^i; products accumulate in 192 bits with one reduction per thread.
__global__ void __launch_bounds__(256) k_open_partial(OpenArgs a, u64* partial) {
  __shared__ u64 sa[256], sb[256];
  int seg = 0, local = blockIdx.x;
  while (seg + 1 < a.nseg && local >= a.seg[seg].count) { local -= a.seg[seg].count; seg++; }
  const OpenSeg sg = a.seg[seg];
  const u64* p = sg.polys + (size_t)local * sg.stride;
  const u64* zp_a = sg.table ? a.zp1_a : a.zp0_a; const u64* zp_b = sg.table ? a.zp1_b : a.zp0_b;
  const size_t per = a.n / a.chunks, lo = (size_t)blockIdx.y * per, hi = lo + per;
  Acc192 aa{0, 0, 0}, ab{0, 0, 0};
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) { u64 c = p[i]; acc_mul(aa, c, zp_a[i]); acc_mul(ab, c, zp_b[i]); }
  sa[threadIdx.x] = acc_reduce(aa); sb[threadIdx.x] = acc_reduce(ab);
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { sa[threadIdx.x] = gl_add(sa[threadIdx.x], sa[threadIdx.x + off]); sb[threadIdx.x] = gl_add(sb[threadIdx.x], sb[threadIdx.x + off]); }
    __syncthreads();
  }
  if (threadIdx.x == 0) { size_t o = ((size_t)(sg.first_out + local) * a.chunks + blockIdx.y) * 2; partial[o] = sa[0]; partial[o + 1] = sb[0]; }
}
__global__ void k_open_finish(const u64* partial, int count, int chunks, u64* out_ext) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= count) return;
  u64 a = 0, b = 0;
  for (int c = 0; c < chunks; c++) { a = gl_add(a, partial[((size_t)p * chunks + c) * 2]); b = gl_add(b, partial[((size_t)p * chunks + c) * 2 + 1]); }
  out_ext[2 * (size_t)p] = a; out_ext[2 * (size_t)p + 1] = b;
}
int open_chunks(int log_n) { int c = log_n - 12; if (c < 0) c = 0; if (c > 6) c = 6; return 1 << c; }
void open_polys(OpenArgs a, int log_n, u64* partial, u64* out_ext, hipStream_t s) {
  int total = 0; for (int g = 0; g < a.nseg; g++) { a.seg[g].first_out = total; total += a.seg[g].count; }
  if (total <= 0) return;
  a.n = size_t(1) << log_n; a.chunks = open_chunks(log_n);
  hipLaunchKernelGGL(k_open_partial, dim3((unsigned)total, (unsigned)a.chunks), dim3(256), 0, s, a, partial);
  hipLaunchKernelGGL(k_open_finish, dim3((unsigned)((total + 63) / 64)), dim3(64), 0, s, partial, total, a.chunks, out_ext);
}

} // namespace qp
