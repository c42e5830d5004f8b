// Internal launcher interface between the host orchestration (qpgpu.hip) and the gfx950 kernels.
// Everything here enqueues on a HIP stream and never synchronises.
#pragma once
#include "gl.h"
#include <hip/hip_runtime.h>

struct qpgpu_ctx { int device; };     // include/qpgpu.h's opaque context

namespace qp {

// ---------------- NTT / LDE (ntt.hip) ----------------
// Root table rt[t] = w_M^t for t < M/2, M = 2^log_M >= every transform size used with it.
struct NttPlan {
  const u64* rt; int log_M;
};
void build_root_table(u64* d_rt, int log_M, hipStream_t s);
// d_out[i] = shift^i * w_{cosets*n}^{c*i} laid out [c][i] (c natural coset index), for the x`cosets` LDE of n coefficients.
void build_lde_scale(u64* d_out, int log_n, int log_cosets, u64 shift, const NttPlan& pl, hipStream_t s);
// d_out[i] = scale * base^i, i < count
void build_powers(u64* d_out, size_t count, u64 base, u64 scale, hipStream_t s);

// In-place size-2^log_n transforms on `count` columns (column c at data + c*stride).
//  dit = 0: DIF, natural in -> bit-reversed out.   dit = 1: DIT, bit-reversed in -> natural out.
//  post (optional, may be null): out[i] *= post[i]  (fused into the last pass).
void ntt_inplace(u64* data, size_t stride, int count, int log_n, bool dit, bool inverse, const u64* post, const NttPlan& pl, hipStream_t s);
// Same, reading from `in` and writing to `out` (the source columns are left intact).
void ntt_copy(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, bool dit, bool inverse, const NttPlan& pl, hipStream_t s);
// Low-degree extension: for each of `count` coefficient columns (n = 2^log_n coefficients at in + c*in_stride) writes
// 2^(log_n+log_cosets) values at out + c*out_stride in LEAF order (bit-reversed natural index):
// block rev(c) holds the size-n DIF transform of coset c. scale = build_lde_scale table.
// shared_gpu: other proofs' kernels run beside this launch (qpgpu_prove_batch) — selects the kernel that co-runs best.
void lde_batch(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets,
               const u64* scale, const NttPlan& pl, hipStream_t s, bool shared_gpu = false);
void lde_batch_generic(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets,
                       const u64* scale, const NttPlan& pl, hipStream_t s);   // tiled multi-pass kernel only (any size)
// out[i] = in[rev(i)] * scale  for `count` columns
void bitrev_permute(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, u64 scale, hipStream_t s);

// LDS-resident radix-16 path (ntt16.hip) for 2^9 <= n <= 2^14: in -> out, `count` columns x 2^log_cosets coset blocks
// (block rev(c) of an output column holds coset c). out_natural: write natural-order results (bit-reversal done from LDS).
bool ntt16_supported(int log_n);
void ntt16_run(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets, bool inverse,
               bool out_natural, const u64* pre, u64 scale, const NttPlan& pl, hipStream_t s);
// Device -> host hand-off without a copy engine round trip: a kernel copies `words` u64 from device memory into mapped pinned
// host memory, fences at system scope and then stores `seq` into *flag (also pinned); the host spins on the flag.
// Measured 7.5 us per hand-off versus 15.2 us for hipMemcpyAsync + hipStreamSynchronize (tools/ubench notes in DESIGN.md).
// `extra` (optional): one more device word copied to dst_mapped[words] (and zeroed afterwards if clear_extra).
void publish_to_host(const u64* src, u64* dst_mapped, size_t words, u64* flag_mapped, u64* counter_dev, u64 seq, hipStream_t s, u64* extra = nullptr, bool clear_extra = false);
bool lde_strided_supported(int log_n);
// natural-order in -> bit-reversed out (coset blocks as in lde_batch); inverse = conjugate roots, unscaled
void lde_strided_run(const u64* in, size_t in_stride, u64* out, size_t out_stride, int count, int log_n, int log_cosets, const u64* pre,
                     const NttPlan& pl, hipStream_t s, bool inverse = false);
// cols[k][L] = salt_value(key, nonce, (oracle*4 + k)*N + L) for k < 4 (salt.h: ChaCha8 keystream); N must be a multiple of 8
struct SaltKey;
void salt_fill(u64* cols, size_t N, const SaltKey& key, u64 nonce, int oracle, hipStream_t s);
// *flag |= 1 if any of data[0..count) is >= p (non-canonical input)
void check_canonical(const u64* data, size_t count, u64* flag, hipStream_t s);

// ---------------- Poseidon / Merkle (merkle.hip) ----------------
void poseidon_permute_batch(u64* states /*[count][12]*/, size_t count, hipStream_t s, int form = -1);   // form: 0 literal / 1 fast partial rounds, -1 default
// Leaf digests of `num_leaves` rows whose element j sits at cols + j*col_stride + leaf (column-major oracle), width felts per
// leaf (hash_or_noop). digests: [num_leaves][4].
// Returns how many levels above the leaves the launch already produced (the leaf kernel fuses levels 1-2 for big trees).
int merkle_leaves_colmajor(const u64* cols, size_t col_stride, int width, size_t num_leaves, int cap_height, u64* digests, hipStream_t s, bool chip_is_shared = false);
// FRI layer leaves: leaf L = 2^arity_bits extension values (a[k], b[k]) k in [L*arity, (L+1)*arity), flattened a0,b0,a1,b1...
void merkle_leaves_fri(const u64* a, const u64* b, int arity_bits, size_t num_leaves, u64* digests, hipStream_t s);
// Builds all upper levels. levels: concatenated digests, level 0 (num_leaves) first, each level half the previous, down to
// 2^cap_height digests. Returns (host-side) nothing; layout helper below.
void merkle_levels(u64* levels, size_t num_leaves, int cap_height, hipStream_t s, int levels_done = 0, bool batch = false);
// Explicit kernel-form selection (tests/golden replay; production picks the form by size inside the functions above).
void merkle_leaves_colmajor_mode(const u64* cols, size_t col_stride, int width, size_t num_leaves, u64* digests, int mode, hipStream_t s);
void merkle_level_mode(const u64* in, u64* out, size_t parents, int mode, hipStream_t s);
void merkle_leaves_fri_mode(const u64* a, const u64* b, int arity_bits, size_t num_leaves, u64* digests, int mode, hipStream_t s);
// Merkle openings (verifier): path p hashes its row (hash_or_noop of `width` felts at rows + p*width), climbs `depth` siblings
// (sibs + (p*depth + l)*4) by the bits of idx[p] and compares with digest idx[p] >> depth of its cap: caps + cap_of[p] * 4 * 2^cap_h
// (cap_of null: one shared cap). ok[p] = 1 on match.
void verify_paths(const u64* rows, int width, const u64* sibs, int depth, const u32* idx, const u64* caps, const u32* cap_of, int cap_h,
                  size_t num_paths, u32* ok, hipStream_t s);
static inline size_t merkle_level_offset(size_t num_leaves, int level) {  // in digests
  size_t off = 0; for (int l = 0; l < level; l++) { off += num_leaves; num_leaves >>= 1; } return off;
}
static inline size_t merkle_total_digests(size_t num_leaves, int cap_height) {
  size_t t = 0; for (size_t m = num_leaves;; m >>= 1) { t += m; if (m <= (size_t(1) << cap_height)) break; } return t;
}

} // namespace qp

namespace qp {

// ---------------- PLONK stages (plonk.hip) ----------------
struct GateDev { u32 tag, sel_index, grp_lo, grp_hi; u64 param; u32 p2, p3; };
static constexpr int MAX_GATES = 16;
struct CircuitDev {            // passed by value to kernels
  int log_n, rate_bits, num_wires, num_routed, num_constants, num_challenges, qdf, num_pp, num_gates, num_selectors;
  int num_gate_constraints;
  GateDev gates[MAX_GATES];
  const u64* k_is;             // device, [num_routed]
  const u64* l0_lde;           // device, [N] leaf order: L_0(x) = (x^n - 1) / (n (x - 1)) on the LDE coset (circuit constant; null: computed per point)
  u64 coset_w[16], coset_dom[16]; // CosetInterpolationGate: barycentric weights and the subgroup points (one such gate per circuit)
};
struct Chal2 { u64 v[2]; };

// Z / partial products (row a7). wires_vals [W][n], sigma_vals [R][n] (values on the subgroup), out zpp_vals [nch*(1+npp)][n]
// ordered [Z_0..Z_{nch-1}, pp_0(npp), pp_1(npp)...]. scratch: q [nch][n][npp+1] + local prefixes [nch][n] + block totals.
void zpp_compute(const CircuitDev& cd, const u64* wires_vals, const u64* sigma_vals, Chal2 betas, Chal2 gammas, u64* scratch,
                 u64* zpp_vals, const NttPlan& pl, hipStream_t s);
static inline size_t zpp_scratch_elems(int nch, size_t n, int npp) { return (size_t)nch * n * (npp + 2) + (size_t)nch * (n / 512 + 2); }

// L_0 on the LDE coset, leaf order: out[N]. One Fermat inversion per point, done once per circuit instead of once per proof and point.
void l0_table(const CircuitDev& cd, u64* out, const NttPlan& pl, hipStream_t s);
// alpha power table: out[c][i] = alpha_c^i, i < count
void alpha_powers(u64* out, Chal2 alphas, int nch, int count, hipStream_t s);
// Quotient values (row a8) on the LDE coset in leaf order: out [nch][N]. cs_lde [S][N], wires_lde [.][N] (stride N), zs_lde [.][N].
void quotient_eval(const CircuitDev& cd, const u64* cs_lde, const u64* wires_lde, const u64* zs_lde, Chal2 betas, Chal2 gammas,
                   const u64* alpha_pows, int ap_stride, const u64* pi_hash /*host values*/, u64* out, const NttPlan& pl, hipStream_t s, bool leaf_gate_set = false);
bool quotient_leaf_set(const CircuitDev& cd);       // the circuit's gates are exactly the leaf set with standard parameters (shared-load form of the fused kernel)

// The fused kernel's two independent halves (part 1: L_0 + permutation terms, part 2: gate constraints) as separate launches
// writing raw sums [nch][N]; quotient_sum adds them and divides by Z_H. Used to run both halves concurrently on two streams.
void quotient_eval_part(int part, const CircuitDev& cd, const u64* cs_lde, const u64* wires_lde, const u64* zs_lde, Chal2 betas, Chal2 gammas,
                        const u64* alpha_pows, int ap_stride, const u64* pi_hash, u64* out_part, const NttPlan& pl, hipStream_t s);
void quotient_sum(const CircuitDev& cd, const u64* a, const u64* b, u64* out, hipStream_t s);
// Split form (quotient_split.hip): gate terms G [num_gate_constraints][N] need only wires/constants; permutation terms Pm
// [nch*(2+npp)][N] need beta/gamma + the Z oracle; the combine needs alpha. Lets the first two run off the critical path.
bool quotient_split_supported(const CircuitDev& cd);
void quotient_gate_terms(const CircuitDev& cd, const u64* cs_lde, const u64* wires_lde, const u64* pi_hash, u64* G, hipStream_t s);
void quotient_perm_terms(const CircuitDev& cd, const u64* cs_lde, const u64* wires_lde, const u64* zs_lde, Chal2 betas, Chal2 gammas, u64* Pm,
                         const NttPlan& pl, hipStream_t s);
void quotient_combine(const CircuitDev& cd, const u64* Pm, const u64* G, const u64* alpha_pows, int ap_stride, u64* out, hipStream_t s);

// out_a[i], out_b[i] = scale * base^i in F_{p^2}, i < count
void ext_powers(u64* out_a, u64* out_b, size_t count, Ext base, Ext scale, hipStream_t s);
// Four power tables in one launch: out[t][i] = base_t^i (i < count), given pw[t][k] = base_t^(2^k).
struct ExtPowArgs { u64* out_a[4]; u64* out_b[4]; Ext pw[4][28]; size_t count; int nbits; };
void ext_powers4(const ExtPowArgs& a, hipStream_t s);
// Openings (row a9), all groups in one launch: segment g = `count` polynomials evaluated at zeta (table 0) or g*zeta (table 1);
// results are written consecutively in segment order as interleaved (a,b) pairs. partial: scratch of
// 2 * (total polys) * open_chunks(log_n) u64.
struct OpenSeg { const u64* polys; size_t stride; int count; int table; int first_out; };
struct OpenArgs { OpenSeg seg[8]; int nseg; const u64 *zp0_a, *zp0_b, *zp1_a, *zp1_b; size_t n; int chunks; };
int open_chunks(int log_n);
void open_polys(OpenArgs a, int log_n, u64* partial, u64* out_ext, hipStream_t s);

// ---------------- FRI (fri.hip) ----------------
struct PolyGroup { const u64* base; size_t stride; int count; };
// comp0 = sum_j alpha^j f_j over the 4 oracle coefficient groups in order; comp1 = sum_j alpha^j zs_j (j < nch). Outputs [2][n] each (a then b).
// partial: scratch of FRI_COMPOSE_GROUPS * 2 * n u64.
static constexpr int FRI_COMPOSE_GROUPS = 8;
void fri_compose(PolyGroup g0, PolyGroup g1, PolyGroup g2, PolyGroup g3, int nch, int log_n, Ext alpha, u64* partial, u64* comp0, u64* comp1, hipStream_t s);
// final[k] = alpha^nch * Q0[k] + Q1[k], Q_b = (comp_b(X) - comp_b(z_b)) / (X - z_b); tables: zp_b = z_b^i, zi_b = z_b^{-i} (i <= n).
// scratch: 4*n + 4*(n/2048 + 1) u64. final: [2][n].
void fri_divide_combine(const u64* comp0, const u64* comp1, int log_n, const u64* zp0_a, const u64* zp0_b, const u64* zp1_a,
                        const u64* zp1_b, const u64* zi0_a, const u64* zi0_b, const u64* zi1_a, const u64* zi1_b, Ext alpha_shift,
                        u64* scratch, u64* final_ab, hipStream_t s);
// out[i] = sum_{j < 2^arity_bits} beta^j in[(i << arity_bits) + j]   (in: a at in_a, b at in_b; n_out results)
void fri_fold(const u64* in_a, const u64* in_b, size_t n_out, int arity_bits, Ext beta, u64* out_a, u64* out_b, hipStream_t s);
// PoW grind (row a12): smallest candidate in [base, base+count) with >= bits leading zeros of permute(state with cand at pos)[7]
// is atomically min-ed into *result (initialise to ~0).
struct Sponge12 { u64 s[12]; };
void pow_grind(Sponge12 st, int pos, u64 base, u64 count, int bits, u64* result, hipStream_t s);

// Query gather (row a13): writes the serialized query rounds for `nq` indices into out (byte buffer).
static constexpr int MAX_FRI_LAYERS = 8, MAX_QUERIES = 64;
struct GatherTree { const u64* cols; size_t stride; int width; const u64* digests; };   // initial oracles (column-major, leaf order)
struct GatherFri { const u64* a; const u64* b; const u64* digests; int arity_bits; int log_leaves; };
struct GatherArgs {
  GatherTree trees[4]; int log_N, cap_height, num_layers; GatherFri layers[MAX_FRI_LAYERS];
  u32 idx[MAX_QUERIES]; int nq; size_t bytes_per_query;
  u64* pow_reset;              // if set: re-arm the workspace's PoW result slot (atomicMin target) for the next proof
};
void query_gather(const GatherArgs& a, uint8_t* out, hipStream_t s);

} // namespace qp
