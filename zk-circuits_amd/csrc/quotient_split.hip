// Quotient evaluation split by what each part depends on, so that most of it leaves the critical path:
//   k_gate_terms  — the 123 gate-constraint slots  sum_g filter_g * c_{g,j}(x): needs only the wires + constants LDEs, i.e. it
//                   can run while the (latency-bound, GPU-idle) top of the WIRES Merkle tree is being built;
//   k_perm_terms  — L_0(x)(Z_c - 1) and the permutation checks: needs beta/gamma and the Z/partial-product LDE, i.e. it can run
//                   while the top of the Z tree is being built;
//   k_quotient_combine — after alpha is known: sum_i alpha^i term_i / Z_H(x) over the stored columns (a 145-column dot product).
// Same arithmetic as the fused kernel in plonk.hip, which remains the default (see quotient_split_supported for the A/B numbers).
#include "kernels.h"
#include "options.h"
#include "poseidon.h"
#include "poseidon_dev.h"
#include "../../include/qpgpu.h"

namespace qp {

__device__ __forceinline__ u64 root_pow_q(const u64* rt, int log_M, u64 e) {
  u64 M = 1ULL << log_M; e &= (M - 1); u64 half = M >> 1;
  return e < half ? rt[e] : gl_neg(rt[e - half]);
}
__device__ __forceinline__ u64 sub_lc(u64 a, u64 b) { u64 d = a - b; if (a < b) d += GL_P; return d; }   // lazy a - canonical b (lazy result)

struct SplitGates {            // per gate type: list position (for the filter) or -1, and its parameter
  int pos_noop, pos_const, pos_pi, pos_bs, pos_ar, pos_poseidon;
  int n_const, n_bs_limbs, n_ar_ops;
};

__device__ __forceinline__ u64 gate_filter(const CircuitDev& cd, int gi, const u64* __restrict__ cs, size_t N, size_t L) {
  const GateDev g = cd.gates[gi];
  const u64 sv = cs[(size_t)g.sel_index * N + L];
  u64 f = 1;
  for (u32 i = g.grp_lo; i < g.grp_hi; i++) if ((int)i != gi) f = gl_mul(f, gl_sub((u64)i, sv));
  if (cd.num_selectors > 1) f = gl_mul(f, gl_sub(0xFFFFFFFFULL, sv));
  return f;
}

struct Pi4q { u64 v[4]; };

__global__ void __launch_bounds__(256) k_gate_terms(CircuitDev cd, SplitGates sg, const u64* __restrict__ cs, const u64* __restrict__ wl, Pi4q pih,
                                                    u64* __restrict__ G) {
  const int log_N = cd.log_n + cd.rate_bits;
  const size_t N = size_t(1) << log_N;
  const size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (L >= N) return;
  auto Wv = [&](int k) -> u64 { return wl[(size_t)k * N + L]; };
  const int nsel = cd.num_selectors;
  const u64 f_pos = gate_filter(cd, sg.pos_poseidon, cs, N, L);
  const u64 f_bs = sg.pos_bs >= 0 ? gate_filter(cd, sg.pos_bs, cs, N, L) : 0;
  const u64 f_ar = sg.pos_ar >= 0 ? gate_filter(cd, sg.pos_ar, cs, N, L) : 0;
  const u64 f_c = sg.pos_const >= 0 ? gate_filter(cd, sg.pos_const, cs, N, L) : 0;
  const u64 f_pi = sg.pos_pi >= 0 ? gate_filter(cd, sg.pos_pi, cs, N, L) : 0;
  const u64 lc0 = cs[(size_t)nsel * N + L], lc1 = cs[(size_t)(nsel + 1) * N + L];
  const int bs_n = sg.pos_bs >= 0 ? sg.n_bs_limbs + 1 : 0, ar_n = sg.pos_ar >= 0 ? sg.n_ar_ops : 0;
  const int cn_n = sg.pos_const >= 0 ? sg.n_const : 0, pi_n = sg.pos_pi >= 0 ? 4 : 0;
  u64 bs0 = 0;   // BaseSum slot 0: sum_l w[1+l] 2^l - w[0]
  if (bs_n) { u64 sum = 0; for (int l = sg.n_bs_limbs - 1; l >= 0; l--) sum = gl_add(gl_add(sum, sum), Wv(1 + l)); bs0 = gl_sub(sum, Wv(0)); }
  // slot k gets the Poseidon constraint plus whatever the cheap gates put into the same slot
  auto emit = [&](int k, u64 cpos_lazy) {
    u64 v = canon(mul_lazy(f_pos, cpos_lazy));
    if (k < bs_n) { u64 t; if (k == 0) t = bs0; else { u64 b = Wv(k); t = gl_mul(b, gl_sub(b, 1)); } v = gl_add(v, gl_mul(f_bs, t)); }
    if (k < ar_n) {
      u64 m0 = Wv(4 * k), m1 = Wv(4 * k + 1), ad = Wv(4 * k + 2), o = Wv(4 * k + 3);
      v = gl_add(v, gl_mul(f_ar, gl_sub(o, gl_add(gl_mul(lc0, gl_mul(m0, m1)), gl_mul(lc1, ad)))));
    }
    if (k < cn_n) v = gl_add(v, gl_mul(f_c, gl_sub(cs[(size_t)(nsel + k) * N + L], Wv(k))));
    if (k < pi_n) v = gl_add(v, gl_mul(f_pi, gl_sub(Wv(k), pih.v[k])));
    G[(size_t)k * N + L] = v;
  };
  // ---- Poseidon gate walk (identical to plonk.hip's fused kernel)
  const u64* rc = POSEIDON_RC_DEV;
  int k = 0;
  u64 swap = Wv(24);
  emit(k++, gl_mul(swap, gl_sub(swap, 1)));
  u64 st[12];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    u64 a = Wv(i), b = Wv(i + 4), delta = Wv(25 + i);
    emit(k++, gl_sub(gl_mul(swap, gl_sub(b, a)), delta));
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
      for (int i = 0; i < 12; i++) { u64 in = Wv(29 + 12 * (r - 1) + i); emit(k++, sub_lc(st[i], in)); st[i] = in; }
    }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = sbox7_lazy(st[i]);
    mds_rc_lazy(st, rc + 12 * (round + 1));
  }
  for (int r = 0; r < 22; r++, round++) {
    u64 in = Wv(65 + r); emit(k++, sub_lc(st[0], in));
    st[0] = sbox7_lazy(in);
    mds_rc_lazy(st, rc + 12 * (round + 1));
  }
  for (int r = 0; r < 4; r++, round++) {
#pragma unroll
    for (int i = 0; i < 12; i++) { u64 in = Wv(87 + 12 * r + i); emit(k++, sub_lc(st[i], in)); st[i] = in; }
#pragma unroll
    for (int i = 0; i < 12; i++) st[i] = sbox7_lazy(st[i]);
    if (r < 3) mds_rc_lazy(st, rc + 12 * (round + 1)); else mds_rc_lazy(st, nullptr);
  }
#pragma unroll
  for (int i = 0; i < 12; i++) emit(k++, sub_lc(st[i], Wv(12 + i)));
}

struct Zh8q { u64 zh[8]; u64 zh_inv[8]; };

__global__ void __launch_bounds__(256) k_perm_terms(CircuitDev cd, const u64* __restrict__ cs, const u64* __restrict__ wl, const u64* __restrict__ zl,
                                                    Chal2 betas, Chal2 gammas, Zh8q zh, u64 n_inv, u64* __restrict__ Pm,
                                                    const u64* __restrict__ rt, int log_M) {
  const int log_N = cd.log_n + cd.rate_bits;
  const size_t N = size_t(1) << log_N;
  const size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (L >= N) return;
  const u32 j = rev_bits((u32)L, log_N);
  const size_t Ln = rev_bits((u32)((j + (1u << cd.rate_bits)) & (N - 1)), log_N);
  const u64 x = gl_mul(GL_GEN, root_pow_q(rt, log_M, (u64)j << (log_M - log_N)));
  const int NC = cd.num_constants, R = cd.num_routed, nch = cd.num_challenges, npp = cd.num_pp, deg = cd.qdf;
  int term = 0;
  {
    const u64 zhx = zh.zh[j & ((1u << cd.rate_bits) - 1)];
    const u64 l0 = cd.l0_lde ? cd.l0_lde[L] : gl_mul(zhx, gl_mul(n_inv, gl_inv(gl_sub(x, 1))));
    for (int c = 0; c < nch; c++) Pm[(size_t)(term++) * N + L] = gl_mul(l0, gl_sub(zl[(size_t)c * N + L], 1));
  }
  // both challenges share the loads of w_j and sigma_j; a chunk's loads are issued together; factors and running products are lazy
  // residues (same arithmetic as the fused kernel in plonk.hip), canonicalised when a term is stored
  u64 prev[2] = {0, 0}, bx[2] = {0, 0};
  for (int c = 0; c < nch; c++) { prev[c] = zl[(size_t)c * N + L]; bx[c] = gl_mul(betas.v[c], x); }
  for (int k = 0; k <= npp; k++) {
    u64 nu[2] = {1, 1}, de[2] = {1, 1};
    u64 wv[8], sgv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const int jj = k * deg + u; const bool ok = u < deg && jj < R; wv[u] = ok ? wl[(size_t)jj * N + L] : 0; sgv[u] = ok ? cs[(size_t)(NC + jj) * N + L] : 0; }
    u64 nxt[2];
    for (int c = 0; c < nch; c++) nxt[c] = k < npp ? zl[(size_t)(nch + c * npp + k) * N + L] : zl[(size_t)c * N + Ln];
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
    for (int c = 0; c < nch; c++) {
      Pm[(size_t)(nch + c * (npp + 1) + k) * N + L] = gl_sub(canon(mul_lazy(prev[c], nu[c])), canon(mul_lazy(nxt[c], de[c])));
      prev[c] = nxt[c];
    }
  }
}

__global__ void __launch_bounds__(256) k_quotient_combine(CircuitDev cd, const u64* __restrict__ Pm, int n_perm, const u64* __restrict__ G, int n_gate,
                                                          const u64* __restrict__ apw, int ap_stride, Zh8q zh, u64* __restrict__ out) {
  const int log_N = cd.log_n + cd.rate_bits;
  const size_t N = size_t(1) << log_N;
  const size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (L >= N) return;
  const u32 j = rev_bits((u32)L, log_N);
  const u64* ap0 = apw; const u64* ap1 = apw + ap_stride;
  Acc192 a0{0, 0, 0}, a1{0, 0, 0};
  // loads batched eight at a time: at 2 waves/SIMD the kernel lives on memory-level parallelism
  auto run = [&](const u64* __restrict__ T, int n, const u64* __restrict__ p0, const u64* __restrict__ p1) {
    int t = 0;
    for (; t + 8 <= n; t += 8) {
      u64 v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) v[k] = T[(size_t)(t + k) * N + L];
#pragma unroll
      for (int k = 0; k < 8; k++) { acc_mul(a0, v[k], p0[t + k]); acc_mul(a1, v[k], p1[t + k]); }
    }
    for (; t < n; t++) { const u64 v = T[(size_t)t * N + L]; acc_mul(a0, v, p0[t]); acc_mul(a1, v, p1[t]); }
  };
  run(Pm, n_perm, ap0, ap1);
  run(G, n_gate, ap0 + n_perm, ap1 + n_perm);
  const u64 zi = zh.zh_inv[j & ((1u << cd.rate_bits) - 1)];
  out[L] = gl_mul(acc_reduce(a0), zi);
  if (cd.num_challenges > 1) out[N + L] = gl_mul(acc_reduce(a1), zi);
}

static Zh8q make_zh(const CircuitDev& cd) {
  Zh8q zh; const size_t n = size_t(1) << cd.log_n;
  u64 gn = gl_pow(GL_GEN, n), w = gl_root(cd.rate_bits), t = 1;
  for (int k = 0; k < 8; k++) { zh.zh[k] = 0; zh.zh_inv[k] = 0; }
  for (int k = 0; k < (1 << cd.rate_bits) && k < 8; k++) { zh.zh[k] = gl_sub(gl_mul(gn, t), 1); zh.zh_inv[k] = gl_inv(zh.zh[k]); t = gl_mul(t, w); }
  return zh;
}

bool quotient_split_supported(const CircuitDev& cd) {
  // Default for single proofs of the leaf gate set up to 2^20 LDE points (QPGPU_SPLIT_QUOTIENT=0 disables; read when a circuit is
  // created). The gate-constraint terms need only wires + constants, so they run beside the wires leaf hash (which alone offers
  // 2 waves/SIMD), the permutation terms beside the Z leaf hash, and only the alpha combination stays on the critical path.
  // MI355X, config B: single proof 3.89 vs 4.00 ms. Stream-pipelined batches keep the fused kernel (336 vs 341 proofs/s: the extra
  // ~300 MB of traffic buys nothing when other proofs already fill the chip) — qpgpu_prove_batch switches it off per call.
  const int v = opts().split_quotient;
  if (v == 0) return false;
  if (cd.log_n + cd.rate_bits > 20 && v != 1) return false;     // 145 extra columns of N per workspace
  int seen[16] = {0}; bool poseidon = false;
  for (int i = 0; i < cd.num_gates; i++) {
    u32 t = cd.gates[i].tag; if (t >= 16 || seen[t]++) return false;
    if (t == QPGPU_GATE_POSEIDON) poseidon = true;
  }
  return poseidon && cd.num_gate_constraints == 123;
}
static SplitGates make_split(const CircuitDev& cd) {
  SplitGates sg{-1, -1, -1, -1, -1, -1, 0, 0, 0};
  for (int i = 0; i < cd.num_gates; i++) {
    switch (cd.gates[i].tag) {
      case QPGPU_GATE_NOOP: sg.pos_noop = i; break;
      case QPGPU_GATE_CONSTANT: sg.pos_const = i; sg.n_const = (int)cd.gates[i].param; break;
      case QPGPU_GATE_PUBLIC_INPUT: sg.pos_pi = i; break;
      case QPGPU_GATE_BASESUM2: sg.pos_bs = i; sg.n_bs_limbs = (int)cd.gates[i].param; break;
      case QPGPU_GATE_ARITHMETIC: sg.pos_ar = i; sg.n_ar_ops = (int)cd.gates[i].param; break;
      case QPGPU_GATE_POSEIDON: sg.pos_poseidon = i; break;
    }
  }
  return sg;
}
void quotient_gate_terms(const CircuitDev& cd, const u64* cs_lde, const u64* wires_lde, const u64* pi_hash, u64* G, hipStream_t s) {
  const size_t N = size_t(1) << (cd.log_n + cd.rate_bits);
  Pi4q pih; for (int i = 0; i < 4; i++) pih.v[i] = pi_hash[i];
  hipLaunchKernelGGL(k_gate_terms, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, cd, make_split(cd), cs_lde, wires_lde, pih, G);
}
void quotient_perm_terms(const CircuitDev& cd, const u64* cs_lde, const u64* wires_lde, const u64* zs_lde, Chal2 betas, Chal2 gammas, u64* Pm,
                         const NttPlan& pl, hipStream_t s) {
  const size_t n = size_t(1) << cd.log_n, N = n << cd.rate_bits;
  hipLaunchKernelGGL(k_perm_terms, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, cd, cs_lde, wires_lde, zs_lde, betas, gammas, make_zh(cd),
                     gl_inv((u64)n), Pm, pl.rt, pl.log_M);
}
void quotient_combine(const CircuitDev& cd, const u64* Pm, const u64* G, const u64* alpha_pows, int ap_stride, u64* out, hipStream_t s) {
  const size_t N = size_t(1) << (cd.log_n + cd.rate_bits);
  const int n_perm = cd.num_challenges + cd.num_challenges * (cd.num_pp + 1);
  hipLaunchKernelGGL(k_quotient_combine, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, cd, Pm, n_perm, G, cd.num_gate_constraints, alpha_pows,
                     ap_stride, make_zh(cd), out);
}

} // namespace qp
