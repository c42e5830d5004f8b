// The aggregation circuit: verify_proof in-circuit for every child of a tree node + forwarding of the children's public inputs —
// aggregate_chunk of the reference (wormhole/aggregator/src/circuits/tree.rs:106-143: add_virtual_verifier_data, add_virtual_proof_with_pis,
// verify_proof, register_public_inputs, set_verifier_data_target / set_proof_with_pis_target). Host-only C++.
//
// qp-plonky2's recursive verifier is not on disk, so this is this repo's own in-circuit verifier, a restatement of this repo's host
// verifier (csrc/verify.hip: transcript, vanishing identity at zeta, FRI openings, folds and Merkle paths; formulas SURVEY.md App. A.4-A.9)
// on the gates upstream's recursive verifier is built from: extension-field products on ArithmeticExtension rows, the
// alpha-reductions of a query's 255 row values / of the 257 openings / of the vanishing terms on Reducing / ReducingExtension rows, cap and
// coset-position look-ups on RandomAccess rows, the arity-16 folds on CosetInterpolation rows, the Poseidon gate's MDS layers (vanishing
// identity of the child) on PoseidonMds rows, Merkle steps on the Poseidon gate's swap wire. Two children give about 9.2 k rows, of which
// 6.6 k are the Poseidon rows of the 2 x 28 x 7 Merkle openings — those are the proof format's, no gate set removes them, and they alone
// keep a two-child node above 2^13 (and a zero-knowledge node, with its 8,326 blinding rows, above 2^14). QPC_NODE_LEAF_GATES=1 builds the
// round-2 form on the leaf gate set only (14 k rows; kept for comparison). The layout is unpinned like every circuit of this builder; what
// a node proves is the reference's statement: "each child proof verifies against the given verifier data, and these are their public inputs".
// The inner circuit may use the leaf gate set and the gates named above (a leaf circuit or another node circuit built here).
#include "../../include/qpcircuit.h"
#include "../csrc/proof_format.h"
#include "builder.h"
#include <cstring>
#include <cstdio>
#include <cstdlib>

namespace qpc {
using namespace qp;

typedef Ext2 ExtT;

struct G {                                   // F_{p^2} = F_p[X]/(X^2 - 7) over builder targets
  Builder& b; bool rg;                       // rg: products on ArithmeticExtension rows (ten per row) instead of four Arithmetic slots each
  ExtT cst(u64 x, u64 y = 0) { return {b.constant(x), b.constant(y)}; }
  ExtT base(Target t) { return {t, b.zero()}; }
  ExtT add(ExtT x, ExtT y) { return {b.add(x.a, y.a), b.add(x.b, y.b)}; }
  ExtT sub(ExtT x, ExtT y) { return {b.sub(x.a, y.a), b.sub(x.b, y.b)}; }
  ExtT mul_add(ExtT x, ExtT y, ExtT z) {     // x*y + z: one ArithmeticExtension operation, or four Arithmetic slots
    if (rg) return b.arith_ext(1, 1, x, y, z);
    return {b.arithmetic(1, 1, x.a, y.a, b.arithmetic(7, 1, x.b, y.b, z.a)), b.arithmetic(1, 1, x.b, y.a, b.arithmetic(1, 1, x.a, y.b, z.b))};
  }
  ExtT kmul_add(u64 k0, ExtT x, ExtT y, u64 k1, ExtT z) {   // k0*x*y + k1*z
    if (rg) return b.arith_ext(k0, k1, x, y, z);
    return scale_add(mul(x, y), k0, scale(z, k1));
  }
  ExtT mul(ExtT x, ExtT y) { return mul_add(x, y, {b.zero(), b.zero()}); }
  ExtT sq(ExtT x) { return mul(x, x); }
  ExtT mul_base(ExtT x, Target s) { return {b.mul(x.a, s), b.mul(x.b, s)}; }
  ExtT mul_base_add(ExtT x, Target s, ExtT z) { return {b.mul_add(x.a, s, z.a), b.mul_add(x.b, s, z.b)}; }
  ExtT scale(ExtT x, u64 c) { return {b.mul_const(c, x.a), b.mul_const(c, x.b)}; }
  ExtT scale_add(ExtT x, u64 c, ExtT z) { return {b.arithmetic(c % P, 1, x.a, b.one(), z.a), b.arithmetic(c % P, 1, x.b, b.one(), z.b)}; }
  ExtT add_const(ExtT x, u64 c) { return {b.add(x.a, b.constant(c)), x.b}; }
  ExtT sbox7(ExtT x) { ExtT x2 = sq(x), x4 = sq(x2), x3 = mul(x2, x); return mul(x3, x4); }
  ExtT inv(ExtT x) {                         // hint + x * inv == 1 (so x = 0 cannot be witnessed)
    ExtT i; b.ext_inverse_hint(x.a, x.b, i.a, i.b);
    ExtT pr = mul(x, i); b.connect(pr.a, b.one()); b.assert_zero(pr.b);
    return i;
  }
  void assert_eq(ExtT x, ExtT y) { b.connect(x.a, y.a); b.connect(x.b, y.b); }
};

struct Inner {                               // the circuit whose proofs are verified (CommonCircuitData)
  qpgpu_circuit_desc d{}; std::vector<qpgpu_gate> gates; std::vector<uint32_t> sel_idx, groups; std::vector<u64> k_is;
  int log_n, rate, log_N, cap_h, nch, S, W, Zp, Q, salt, nlayers, npp, npi; int arity[8];
  Layout lay{};
  std::string init(const qpgpu_circuit_desc& src) {
    d = src;
    if (!src.gates || !src.selector_indices || !src.selector_group_ranges || !src.k_is) return "null array in the inner circuit descriptor";
    gates.assign(src.gates, src.gates + src.num_gates); sel_idx.assign(src.selector_indices, src.selector_indices + src.num_gates);
    groups.assign(src.selector_group_ranges, src.selector_group_ranges + 2 * src.num_selector_groups); k_is.assign(src.k_is, src.k_is + src.num_routed);
    d.gates = gates.data(); d.selector_indices = sel_idx.data(); d.selector_group_ranges = groups.data(); d.k_is = k_is.data(); d.constants_sigmas = nullptr;
    for (const auto& g : gates) switch (g.tag) {
      case QPGPU_GATE_NOOP: case QPGPU_GATE_PUBLIC_INPUT: case QPGPU_GATE_POSEIDON: break;
      case QPGPU_GATE_CONSTANT: if (g.param != 2) return "inner Constant gate must hold 2 constants"; break;
      case QPGPU_GATE_BASESUM2: if (g.param != BASESUM_LIMBS) return "inner BaseSum gate must have 63 limbs"; break;
      case QPGPU_GATE_ARITHMETIC: if (g.param != ARITH_OPS) return "inner Arithmetic gate must hold 20 operations"; break;
      case QPGPU_GATE_POSEIDON_MDS: break;
      case QPGPU_GATE_ARITHMETIC_EXTENSION: if (g.param != ARITH_EXT_OPS) return "inner ArithmeticExtension gate must hold 10 operations"; break;
      case QPGPU_GATE_REDUCING: if (g.param != REDUCING_COEFFS) return "inner Reducing gate must hold 43 coefficients"; break;
      case QPGPU_GATE_REDUCING_EXTENSION: if (g.param != REDUCING_EXT_COEFFS) return "inner ReducingExtension gate must hold 32 coefficients"; break;
      case QPGPU_GATE_RANDOM_ACCESS: if (g.param != RA_BITS || g.param2 != RA_COPIES || g.param3 != RA_EXTRA) return "inner RandomAccess gate must be (4 bits, 4 copies, 2 constants)"; break;
      case QPGPU_GATE_COSET_INTERPOLATION:
        if (g.param != COSET_BITS || g.param2 != COSET_DEG || g.table_len != COSET_NP || !g.table) return "inner CosetInterpolation gate must be (4 bits, degree 6)";
        for (int i = 0; i < COSET_NP; i++) if (g.table[i] != Builder::coset_weights()[i]) return "inner CosetInterpolation gate: unexpected barycentric weights";
        break;
      default: return "the in-circuit verifier does not cover gate tag " + std::to_string(g.tag);
    }
    for (auto& g : gates) g.table = nullptr;                     // the weights are Builder::coset_weights() (checked above)
    if (d.num_wires != W_ || d.num_routed != R_ || d.num_challenges != 2 || d.fri_rate_bits != 3 || d.quotient_degree_factor != 8 || d.num_gate_constraints != 123)
      return "inner circuit is not a standard_recursion_config circuit";
    if (d.fri_num_layers > 8 || d.degree_bits + d.fri_rate_bits > 28 || d.fri_cap_height > d.degree_bits) return "inner FRI shape out of range";
    if (d.num_selector_groups < 1 || d.num_constants != d.num_selector_groups + 2 || d.num_gates < 1 || d.num_gates > 16) return "inner selector / constant columns out of range";
    if ((d.num_partial_products + 1) * d.quotient_degree_factor < d.num_routed || d.fri_num_queries < 1 || d.fri_num_queries > 64 || d.fri_pow_bits > 32) return "inner circuit shape out of range";
    { uint32_t lg = d.degree_bits + d.fri_rate_bits, sum = 0;
      for (uint32_t i = 0; i < d.fri_num_layers; i++) { const uint32_t a = d.fri_arity_bits[i]; if (a < 1 || a > 4 || lg < a + d.fri_cap_height) return "inner FRI arities out of range"; lg -= a; sum += a; }
      if (sum > d.degree_bits) return "inner FRI reductions exceed the degree"; }
    for (uint32_t i = 0; i < d.num_gates; i++) if (src.selector_indices[i] >= d.num_selector_groups) return "inner selector index out of range";
    log_n = d.degree_bits; rate = d.fri_rate_bits; log_N = log_n + rate; cap_h = d.fri_cap_height; nch = d.num_challenges;
    S = d.num_constants + d.num_routed; W = d.num_wires; npp = d.num_partial_products; Zp = nch * (1 + npp); Q = nch * d.quotient_degree_factor;
    salt = d.zero_knowledge ? 4 : 0; nlayers = d.fri_num_layers; npi = d.num_public_inputs;
    for (int i = 0; i < nlayers; i++) arity[i] = d.fri_arity_bits[i];
    compute_layout(d, lay);
    return "";
  }
  static const uint32_t W_ = 135, R_ = 80;
};

struct QueryT { std::vector<Target> row[4]; std::vector<Hash4> sibs[4]; std::vector<std::vector<ExtT>> evals; std::vector<std::vector<Hash4>> lsibs; };
struct ProofT { std::vector<Hash4> caps[3]; std::vector<ExtT> op; std::vector<std::vector<Hash4>> fri_caps; std::vector<QueryT> q; std::vector<ExtT> final_poly; Target pow_witness; std::vector<Target> pis; };
struct VerifierDataT { std::vector<Hash4> cap; Hash4 digest; };

static ExtT input_ext(Builder& b) { ExtT e; e.a = b.add_input(); e.b = b.add_input(); return e; }

// add_virtual_proof_with_pis: one input target per field element of the proof, in ProofWithPublicInputs::to_bytes order.
ProofT add_proof_targets(Builder& b, const Inner& in) {
  ProofT p; const size_t capn = size_t(1) << in.cap_h;
  for (auto& c : p.caps) for (size_t i = 0; i < capn; i++) c.push_back(b.add_input_hash());
  for (size_t i = 0; i < in.lay.n_openings; i++) p.op.push_back(input_ext(b));
  p.fri_caps.resize(in.nlayers);
  for (auto& c : p.fri_caps) for (size_t i = 0; i < capn; i++) c.push_back(b.add_input_hash());
  const size_t widths[4] = {(size_t)in.S, (size_t)in.W + in.salt, (size_t)in.Zp + in.salt, (size_t)in.Q + in.salt};
  const int path = in.log_N - in.cap_h;
  p.q.resize(in.d.fri_num_queries);
  for (auto& q : p.q) {
    for (int t = 0; t < 4; t++) { q.row[t] = b.add_inputs((int)widths[t]); for (int l = 0; l < path; l++) q.sibs[t].push_back(b.add_input_hash()); }
    int lg = in.log_N; q.evals.resize(in.nlayers); q.lsibs.resize(in.nlayers);
    for (int l = 0; l < in.nlayers; l++) {
      const int ab = in.arity[l];
      for (int i = 0; i < (1 << ab); i++) q.evals[l].push_back(input_ext(b));
      for (int k = 0; k < lg - ab - in.cap_h; k++) q.lsibs[l].push_back(b.add_input_hash());
      lg -= ab;
    }
  }
  for (size_t i = 0; i < in.lay.final_poly_len; i++) p.final_poly.push_back(input_ext(b));
  p.pow_witness = b.add_input();
  p.pis = b.add_inputs(in.npi);
  return p;
}
VerifierDataT add_verifier_data_targets(Builder& b, const Inner& in) {       // add_virtual_verifier_data: constants_sigmas_cap, circuit_digest
  VerifierDataT v; for (size_t i = 0; i < (size_t(1) << in.cap_h); i++) v.cap.push_back(b.add_input_hash());
  v.digest = b.add_input_hash(); return v;
}

struct ChallengerT {                          // csrc/proof_format.h Challenger over targets
  Builder& b; Target st[12]; std::vector<Target> in; Target out[8]; int n_out = 0;
  explicit ChallengerT(Builder& bb) : b(bb) { for (auto& s : st) s = b.zero(); }
  void duplex() { for (size_t i = 0; i < in.size(); i++) st[i] = in[i]; in.clear(); b.permute(st); for (int i = 0; i < 8; i++) out[i] = st[i]; n_out = 8; }
  void observe(Target x) { n_out = 0; in.push_back(x); if (in.size() == 8) duplex(); }
  void observe(const Hash4& h) { for (Target t : h.e) observe(t); }
  void observe(ExtT e) { observe(e.a); observe(e.b); }
  Target challenge() { if (!in.empty() || n_out == 0) duplex(); return out[--n_out]; }
  ExtT challenge_ext() { ExtT e; e.a = challenge(); e.b = challenge(); return e; }
};

static Hash4 hash_or_noop(Builder& b, const std::vector<Target>& v) {
  if (v.size() <= 4) { Hash4 h; for (int i = 0; i < 4; i++) h.e[i] = i < (int)v.size() ? v[i] : b.zero(); return h; }
  return b.hash_n_to_hash_no_pad(v);
}
// indicator vector of an index given by little-endian bits: ind[k] = [index == k]
static std::vector<Target> indicators(Builder& b, const std::vector<BoolT>& bits) {
  std::vector<Target> ind{b.one()};
  for (size_t j = 0; j < bits.size(); j++) {
    std::vector<Target> nx(ind.size() * 2);
    for (size_t k = 0; k < ind.size(); k++) { nx[k + ind.size()] = b.mul(ind[k], bits[j].t); nx[k] = b.sub(ind[k], nx[k + ind.size()]); }
    ind.swap(nx);
  }
  return ind;
}
// Merkle opening: leaf digest up `sibs.size()` levels with the index bits choosing the side (Poseidon gate swap), root == cap[index >> depth]
static void verify_merkle(Builder& b, const std::vector<Target>& leaf, const std::vector<Hash4>& sibs, const BoolT* bits, const std::vector<Hash4>& cap, const std::vector<Target>& cap_ind,
                          const Target* cap_index = nullptr) {
  Hash4 cur = hash_or_noop(b, leaf);
  for (size_t l = 0; l < sibs.size(); l++) {
    Target st[12]; for (int i = 0; i < 4; i++) { st[i] = cur.e[i]; st[4 + i] = sibs[l].e[i]; st[8 + i] = b.zero(); }
    b.permute_swap(st, bits[l]);
    for (int i = 0; i < 4; i++) cur.e[i] = st[i];
  }
  for (int f = 0; f < 4; f++) {
    if (cap_index && cap.size() <= (size_t)RA_VEC) {                          // one RandomAccess copy per digest element
      std::vector<Target> list; for (const Hash4& h : cap) list.push_back(h.e[f]);
      b.connect(b.random_access(*cap_index, list), cur.e[f]);
      continue;
    }
    Target sel = b.zero();
    for (size_t k = 0; k < cap.size(); k++) sel = b.mul_add(cap_ind[k], cap[k].e[f], sel);
    b.connect(sel, cur.e[f]);
  }
}
static u64 rev_bits(u64 x, int bits) { u64 r = 0; for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i); return r; }

static void mds_ext(G& g, ExtT st[12], const u64* next_rc) {                  // MDS layer, then the next round's constants (folded into the sums' start)
  static const u64 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  ExtT out[12];
  if (g.rg) {                                                                // one PoseidonMds row, then twelve base additions
    g.b.poseidon_mds(st, out);
    for (int r = 0; r < 12; r++) st[r] = next_rc ? g.add_const(out[r], next_rc[r]) : out[r];
    return;
  }
  for (int r = 0; r < 12; r++) {
    ExtT acc = next_rc ? g.cst(next_rc[r]) : g.cst(0);
    for (int i = 0; i < 12; i++) acc = g.scale_add(st[(i + r) % 12], C[i], acc);
    if (r == 0) acc = g.scale_add(st[0], 8, acc);
    out[r] = acc;
  }
  for (int r = 0; r < 12; r++) st[r] = out[r];
}

// E[X]/(X^2 - 7) over E = F_{p^2}: what a pair of extension-valued wires means at zeta (upstream's ExtensionAlgebra)
struct AlgT { ExtT a, b; };
static inline AlgT alg(const ExtT* w, int i) { return AlgT{w[i], w[i + 1]}; }
static AlgT alg_add(G& g, AlgT x, AlgT y) { return AlgT{g.add(x.a, y.a), g.add(x.b, y.b)}; }
static AlgT alg_sub(G& g, AlgT x, AlgT y) { return AlgT{g.sub(x.a, y.a), g.sub(x.b, y.b)}; }
static AlgT alg_mul(G& g, AlgT x, AlgT y) { return AlgT{g.kmul_add(7, x.b, y.b, 1, g.mul(x.a, y.a)), g.mul_add(x.a, y.b, g.mul(x.b, y.a))}; }
static AlgT alg_scale(G& g, AlgT x, ExtT s) { return AlgT{g.mul(x.a, s), g.mul(x.b, s)}; }

// Gate constraints of the leaf gate set at zeta: terms[j] += filter_g * c_{g,j} (csrc/verify.hip eval_leaf_gate_constraints).
static void eval_gate_constraints(G& g, const Inner& in, const ExtT* consts, const ExtT* wires, const Hash4& pi_hash, std::vector<ExtT>& terms) {
  Builder& b = g.b; const int nsel = (int)in.d.num_selector_groups; const u64* rc = POSEIDON_RC_HOST;
  for (uint32_t gi = 0; gi < in.d.num_gates; gi++) {
    const qpgpu_gate& gt = in.gates[gi];
    if (gt.tag == QPGPU_GATE_NOOP) continue;
    const uint32_t si = in.sel_idx[gi], lo = in.groups[2 * si], hi = in.groups[2 * si + 1];
    const ExtT sv = consts[si];
    ExtT filter = g.cst(1);
    for (uint32_t i = lo; i < hi; i++) if (i != gi) filter = g.mul(filter, g.sub(g.cst(i), sv));
    if (nsel > 1) filter = g.mul(filter, g.sub(g.cst(0xFFFFFFFFULL), sv));
    auto put = [&](size_t slot, ExtT c) { terms[slot] = g.mul_add(filter, c, terms[slot]); };
    switch (gt.tag) {
      case QPGPU_GATE_CONSTANT: for (u64 i = 0; i < gt.param; i++) put(i, g.sub(consts[nsel + i], wires[i])); break;
      case QPGPU_GATE_PUBLIC_INPUT: for (int i = 0; i < 4; i++) put(i, g.sub(wires[i], g.base(pi_hash.e[i]))); break;
      case QPGPU_GATE_BASESUM2: {
        ExtT sum = g.cst(0);
        for (int l = (int)gt.param - 1; l >= 0; l--) sum = g.scale_add(sum, 2, wires[1 + l]);
        put(0, g.sub(sum, wires[0]));
        for (u64 l = 0; l < gt.param; l++) put(1 + l, g.mul(wires[1 + l], g.sub(wires[1 + l], g.cst(1))));
        break;
      }
      case QPGPU_GATE_ARITHMETIC:
        for (u64 i = 0; i < gt.param; i++)
          put(i, g.sub(wires[4 * i + 3], g.mul_add(consts[nsel], g.mul(wires[4 * i], wires[4 * i + 1]), g.mul(consts[nsel + 1], wires[4 * i + 2]))));
        break;
      case QPGPU_GATE_POSEIDON: {
        size_t k = 0;
        const ExtT swap = wires[24];
        put(k++, g.mul(swap, g.sub(swap, g.cst(1))));
        ExtT st[12];
        for (int i = 0; i < 4; i++) {
          const ExtT a = wires[i], bb = wires[i + 4], delta = wires[25 + i];
          put(k++, g.sub(g.mul(swap, g.sub(bb, a)), delta));
          st[i] = g.add(a, delta); st[i + 4] = g.sub(bb, delta);
        }
        for (int i = 8; i < 12; i++) st[i] = wires[i];
        for (int i = 0; i < 12; i++) st[i] = g.add_const(st[i], rc[i]);            // round 0's constants; later rounds' ride on the MDS sums
        int round = 0;
        for (int r = 0; r < 4; r++, round++) {                                   // first full rounds: round 0's S-box inputs are not wired
          if (r != 0) for (int i = 0; i < 12; i++) { const ExtT w = wires[29 + 12 * (r - 1) + i]; put(k++, g.sub(st[i], w)); st[i] = w; }
          for (int i = 0; i < 12; i++) st[i] = g.sbox7(st[i]);
          mds_ext(g, st, rc + 12 * (round + 1));
        }
        for (int r = 0; r < 22; r++, round++) {                                  // partial rounds
          const ExtT w = wires[65 + r]; put(k++, g.sub(st[0], w));
          st[0] = g.sbox7(w);
          mds_ext(g, st, rc + 12 * (round + 1));
        }
        for (int r = 0; r < 4; r++, round++) {                                   // second full rounds
          for (int i = 0; i < 12; i++) { const ExtT w = wires[87 + 12 * r + i]; put(k++, g.sub(st[i], w)); st[i] = w; }
          for (int i = 0; i < 12; i++) st[i] = g.sbox7(st[i]);
          mds_ext(g, st, round + 1 < 30 ? rc + 12 * (round + 1) : nullptr);
        }
        for (int i = 0; i < 12; i++) put(k++, g.sub(st[i], wires[12 + i]));
        break;
      }
      // ---- the recursive verifier's gates (formulas of oracle/gates.cc / csrc/verify.hip). A pair of wires opened at zeta is an element
      // w0 + w1*X of the algebra E[X]/(X^2 - 7) over E = F_{p^2}.
      case QPGPU_GATE_ARITHMETIC_EXTENSION:
        for (int i = 0; i < (int)gt.param; i++) {
          const AlgT r = alg_sub(g, alg(wires, 8 * i + 6), alg_add(g, alg_scale(g, alg_mul(g, alg(wires, 8 * i), alg(wires, 8 * i + 2)), consts[nsel]), alg_scale(g, alg(wires, 8 * i + 4), consts[nsel + 1])));
          put(2 * i, r.a); put(2 * i + 1, r.b);
        }
        break;
      case QPGPU_GATE_REDUCING: case QPGPU_GATE_REDUCING_EXTENSION: {
        const bool xc = gt.tag == QPGPU_GATE_REDUCING_EXTENSION; const int nc = (int)gt.param, start_accs = 6 + (xc ? 2 : 1) * nc;
        const AlgT alpha = alg(wires, 2); AlgT acc = alg(wires, 4);
        for (int i = 0; i < nc; i++) {
          const AlgT coeff = xc ? alg(wires, 6 + 2 * i) : AlgT{wires[6 + i], g.cst(0)};
          const AlgT acc_i = i == nc - 1 ? alg(wires, 0) : alg(wires, start_accs + 2 * i);
          const AlgT r = alg_sub(g, alg_add(g, alg_mul(g, acc, alpha), coeff), acc_i);
          put(2 * i, r.a); put(2 * i + 1, r.b);
          acc = acc_i;
        }
        break;
      }
      case QPGPU_GATE_RANDOM_ACCESS: {
        const int bits = (int)gt.param, copies = (int)gt.param2, extra = (int)gt.param3, vec = 1 << bits, routed = (2 + vec) * copies + extra;
        size_t k = 0;
        for (int c = 0; c < copies; c++) {
          const int base = (2 + vec) * c; const ExtT* bw = wires + routed + c * bits;
          for (int i = 0; i < bits; i++) put(k++, g.mul(bw[i], g.sub(bw[i], g.cst(1))));
          ExtT idx = g.cst(0); for (int i = bits - 1; i >= 0; i--) idx = g.scale_add(idx, 2, bw[i]);
          put(k++, g.sub(idx, wires[base]));
          std::vector<ExtT> list(wires + base + 2, wires + base + 2 + vec);
          for (int i = 0, len = vec; i < bits; i++, len >>= 1) for (int j = 0; j < len / 2; j++) list[j] = g.mul_add(bw[i], g.sub(list[2 * j + 1], list[2 * j]), list[2 * j]);
          put(k++, g.sub(list[0], wires[base + 1]));
        }
        for (int i = 0; i < extra; i++) put(k++, g.sub(consts[nsel + i], wires[(2 + vec) * copies + i]));
        break;
      }
      case QPGPU_GATE_POSEIDON_MDS:
        for (int h = 0; h < 2; h++) {                                            // the small-constant MDS acts on the two coordinates independently
          ExtT st[12]; for (int i = 0; i < 12; i++) st[i] = wires[2 * i + h];
          mds_ext(g, st, nullptr);
          for (int i = 0; i < 12; i++) put(2 * i + h, g.sub(wires[24 + 2 * i + h], st[i]));
        }
        break;
      case QPGPU_GATE_COSET_INTERPOLATION: {
        const u64* dom = Builder::coset_domain(); const u64* wgt = Builder::coset_weights();
        const AlgT x = alg(wires, COSET_S_SHIFTED);
        size_t k = 0;
        { const AlgT r = alg_sub(g, alg(wires, COSET_S_POINT), alg_scale(g, x, wires[0])); put(k++, r.a); put(k++, r.b); }
        AlgT ev{g.cst(0), g.cst(0)}, pr{g.cst(1), g.cst(0)};
        int lo = 0, hi = COSET_DEG;
        for (int seg = 0; seg <= COSET_NI; seg++) {
          for (int i = lo; i < hi; i++) {
            const AlgT term{g.sub(x.a, g.cst(dom[i])), x.b};
            const AlgT val{g.scale(wires[1 + 2 * i], wgt[i]), g.scale(wires[2 + 2 * i], wgt[i])};
            ev = alg_add(g, alg_mul(g, ev, term), alg_mul(g, val, pr)); pr = alg_mul(g, pr, term);
          }
          if (seg == COSET_NI) break;
          const AlgT iev = alg(wires, COSET_S_INTER + 2 * seg), ipr = alg(wires, COSET_S_INTER + 2 * (COSET_NI + seg));
          { const AlgT r = alg_sub(g, iev, ev); put(k++, r.a); put(k++, r.b); }
          { const AlgT r = alg_sub(g, ipr, pr); put(k++, r.a); put(k++, r.b); }
          ev = iev; pr = ipr;
          lo = 1 + (COSET_DEG - 1) * (seg + 1); hi = lo + COSET_DEG - 1 < COSET_NP ? lo + COSET_DEG - 1 : COSET_NP;
        }
        { const AlgT r = alg_sub(g, alg(wires, COSET_S_VALUE), ev); put(k++, r.a); put(k++, r.b); }
        break;
      }
      default: break;
    }
  }
  (void)b;
}

// builder.verify_proof: every check of csrc/verify.hip's verify_host + the Merkle openings, as constraints.
void verify_proof(Builder& b, const Inner& in, const ProofT& pf, const VerifierDataT& vd, bool rg) {
  G g{b, rg};
  const int nch = in.nch; const size_t n = size_t(1) << in.log_n;
  static const bool report = getenv("QPC_TIMING") != nullptr; size_t rows_mark = b.num_rows_so_far();
  auto section = [&](const char* what) { if (report) { const size_t r = b.num_rows_so_far(); fprintf(stderr, "  verify_proof %-22s %6zu rows\n", what, r - rows_mark); rows_mark = r; } };
  // witness generation replays a proof's transcript / vanishing identity / reduced openings as one stage-1 segment and each of its queries
  // as a stage-2 segment (builder.h begin_segment): the children of a node, and then all their queries, are independent of each other
  b.begin_segment(1);
  const Hash4 pi_hash = b.hash_n_to_hash_no_pad(pf.pis);
  // ---- transcript
  ChallengerT ch(b);
  ch.observe(vd.digest); ch.observe(pi_hash); for (const auto& h : pf.caps[0]) ch.observe(h);
  Target betas[2], gammas[2], alphas[2];
  for (int i = 0; i < nch; i++) betas[i] = ch.challenge();
  for (int i = 0; i < nch; i++) gammas[i] = ch.challenge();
  for (const auto& h : pf.caps[1]) ch.observe(h);
  for (int i = 0; i < nch; i++) alphas[i] = ch.challenge();
  for (const auto& h : pf.caps[2]) ch.observe(h);
  const ExtT zeta = ch.challenge_ext();
  const size_t nop = in.lay.n_openings, a_ext = (size_t)in.S + in.W + nch;
  for (size_t i = 0; i < a_ext; i++) ch.observe(pf.op[i]);                      // constants..zs, then partial products | quotient, then zs_next
  for (size_t i = a_ext + nch; i < nop; i++) ch.observe(pf.op[i]);
  for (size_t i = a_ext; i < a_ext + nch; i++) ch.observe(pf.op[i]);
  const ExtT fri_alpha = ch.challenge_ext();
  std::vector<ExtT> fri_betas;
  for (int l = 0; l < in.nlayers; l++) { for (const auto& h : pf.fri_caps[l]) ch.observe(h); fri_betas.push_back(ch.challenge_ext()); }
  for (const ExtT& c : pf.final_poly) ch.observe(c);
  ch.observe(pf.pow_witness);
  const Target pow_resp = ch.challenge();
  if (in.d.fri_pow_bits) b.range_check(pow_resp, 64 - (int)in.d.fri_pow_bits);   // >= pow_bits leading zeros of the canonical 64-bit value
  std::vector<std::vector<BoolT>> index_bits;
  for (uint32_t q = 0; q < in.d.fri_num_queries; q++) {
    // query index = challenge mod N: the canonical value split into two 32-bit halves (hi = 2^32 - 1 forces lo = 0, so the halves are
    // the canonical representation), of which the low log_N bits are the index
    const Target x = ch.challenge();
    Target lo, hi; b.lohi_hint(x, lo, hi);
    b.connect(b.arithmetic(u64(1) << 32, 1, hi, b.one(), lo), x);
    std::vector<BoolT> lob = b.split_le(lo, 32); b.range_check(hi, 32);
    b.assert_zero(b.mul(b.is_equal(hi, b.constant(0xFFFFFFFFULL)).t, lo));
    lob.resize(in.log_N); index_bits.push_back(lob);
  }
  section("transcript + indices");
  // ---- openings by name (proof order: constants|sigmas | wires | zs | zs_next | partial products | quotient)
  const ExtT* o_cs = pf.op.data(); const ExtT* o_wires = o_cs + in.S; const ExtT* o_zs = o_wires + in.W; const ExtT* o_zs_next = o_zs + nch;
  const ExtT* o_pp = o_zs_next + nch; const ExtT* o_q = o_pp + (size_t)nch * in.npp; const ExtT* o_sigmas = o_cs + in.d.num_constants;
  // ---- vanishing identity at zeta
  {
    std::vector<ExtT> gate_terms(in.d.num_gate_constraints, g.cst(0));
    eval_gate_constraints(g, in, o_cs, o_wires, pi_hash, gate_terms);
    ExtT zeta_n = zeta; for (int i = 0; i < in.log_n; i++) zeta_n = g.sq(zeta_n);
    const ExtT zh = g.sub(zeta_n, g.cst(1));
    const ExtT l0 = g.mul(zh, g.inv(g.scale(g.sub(zeta, g.cst(1)), (u64)n)));
    std::vector<ExtT> terms;
    for (int c = 0; c < nch; c++) terms.push_back(g.mul(l0, g.sub(o_zs[c], g.cst(1))));
    const int deg = (int)in.d.quotient_degree_factor, R = (int)in.d.num_routed;
    for (int c = 0; c < nch; c++) {
      int chunk = 0;
      for (int j0 = 0; j0 < R; j0 += deg, chunk++) {
        ExtT num = g.cst(1), den = g.cst(1);
        for (int j = j0; j < j0 + deg && j < R; j++) {
          const Target bk = b.mul(betas[c], b.constant(in.k_is[j]));             // beta * k_j
          ExtT tn = g.mul_base_add(zeta, bk, o_wires[j]); tn.a = b.add(tn.a, gammas[c]);
          ExtT td = g.mul_base_add(o_sigmas[j], betas[c], o_wires[j]); td.a = b.add(td.a, gammas[c]);
          num = g.mul(num, tn); den = g.mul(den, td);
        }
        const ExtT acc_lo = chunk == 0 ? o_zs[c] : o_pp[(size_t)c * in.npp + chunk - 1];
        const ExtT acc_hi = chunk == in.npp ? o_zs_next[c] : o_pp[(size_t)c * in.npp + chunk];
        terms.push_back(g.sub(g.mul(acc_lo, num), g.mul(acc_hi, den)));
      }
    }
    terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
    for (int c = 0; c < nch; c++) {
      ExtT van = g.cst(0);
      if (rg) van = b.reduce_ext(g.base(alphas[c]), std::vector<ExtT>(terms.rbegin(), terms.rend()));      // sum_i terms[i] alpha^i on ReducingExtension rows
      else for (size_t i = terms.size(); i-- > 0;) van = g.mul_base_add(van, alphas[c], terms[i]);
      ExtT qv = g.cst(0); for (int m = deg; m-- > 0;) qv = g.mul_add(qv, zeta_n, o_q[(size_t)c * deg + m]);
      g.assert_eq(van, g.mul(zh, qv));
    }
  }
  section("vanishing identity");
  // ---- FRI: reduced openings, then per query the initial combination, the layer folds, the final polynomial and every Merkle opening
  const size_t n0 = (size_t)in.S + in.W + in.Zp + in.Q;
  std::vector<ExtT> apow((rg ? (size_t)nch : n0) + 1); apow[0] = g.cst(1); for (size_t i = 1; i < apow.size(); i++) apow[i] = g.mul(apow[i - 1], fri_alpha);
  ExtT y0 = g.cst(0), y1 = g.cst(0);
  if (rg) {                                                                      // sum_j op_j alpha^j: Horner from the last opening down, on ReducingExtension rows
    std::vector<ExtT> ops(pf.op.begin(), pf.op.begin() + a_ext); ops.insert(ops.end(), pf.op.begin() + a_ext + nch, pf.op.end());
    y0 = b.reduce_ext(fri_alpha, std::vector<ExtT>(ops.rbegin(), ops.rend()));
    for (int i = 0; i < nch; i++) y1 = g.mul_add(o_zs_next[i], apow[i], y1);
  } else { size_t j = 0;
    for (size_t i = 0; i < a_ext; i++, j++) y0 = g.mul_add(pf.op[i], apow[j], y0);
    for (size_t i = a_ext + nch; i < nop; i++, j++) y0 = g.mul_add(pf.op[i], apow[j], y0);
    for (int i = 0; i < nch; i++) y1 = g.mul_add(o_zs_next[i], apow[i], y1); }
  const ExtT z0 = zeta, z1 = g.scale(zeta, gl_root(in.log_n));
  std::vector<ExtT> beta_pow;                                                    // beta_l^(2^arity_l)
  for (int l = 0; l < in.nlayers; l++) { ExtT x = fri_betas[l]; for (int i = 0; i < in.arity[l]; i++) x = g.sq(x); beta_pow.push_back(x); }
  const size_t unsalted[4] = {(size_t)in.S, (size_t)in.W, (size_t)in.Zp, (size_t)in.Q};
  const int path = in.log_N - in.cap_h;
  const u64 wN = gl_root(in.log_N);
  section("reduced openings");
  for (uint32_t qi = 0; qi < in.d.fri_num_queries; qi++) {
    b.begin_segment(2);
    const QueryT& q = pf.q[qi]; const std::vector<BoolT>& bits = index_bits[qi];
    // cap entry = index >> path (every tree of the query): indicator sums, or RandomAccess rows fed with the index itself
    const std::vector<BoolT> cap_bits(bits.begin() + path, bits.end());
    std::vector<Target> cap_ind; Target cap_idx = b.zero();
    const bool cap_ra = rg && cap_bits.size() <= (size_t)RA_BITS;
    if (cap_ra) { for (size_t i = cap_bits.size(); i-- > 0;) cap_idx = b.arithmetic(2, 1, cap_idx, b.one(), cap_bits[i].t); }
    else cap_ind = indicators(b, cap_bits);
    const Target* cap_sel = cap_ra ? &cap_idx : nullptr;
    ExtT sum0 = g.cst(0), sum1 = g.cst(0); size_t j = 0;
    std::vector<Target> row_values;                                               // the 255 unsalted values of the four rows, oracle by oracle
    for (int t = 0; t < 4; t++) {
      const std::vector<Hash4>& cap = t == 0 ? vd.cap : pf.caps[t - 1];
      verify_merkle(b, q.row[t], q.sibs[t], bits.data(), cap, cap_ind, cap_sel);
      if (rg) row_values.insert(row_values.end(), q.row[t].begin(), q.row[t].begin() + unsalted[t]);
      else for (size_t k = 0; k < unsalted[t]; k++, j++) sum0 = g.mul_base_add(apow[j], q.row[t][k], sum0);
      if (t == 2) for (int k = 0; k < nch; k++) sum1 = g.mul_base_add(apow[k], q.row[t][k], sum1);
    }
    if (rg) sum0 = b.reduce_base(fri_alpha, std::vector<Target>(row_values.rbegin(), row_values.rend()));   // sum_j row_j alpha^j on Reducing rows
    // the point: g * w_N^rev(index)
    Target sx = b.constant(GL_GEN);
    for (int i = 0; i < in.log_N; i++) sx = b.mul(sx, b.mul_add(bits[i].t, b.constant(gl_sub(gl_pow(wN, u64(1) << (in.log_N - 1 - i)), 1)), b.one()));
    ExtT old_eval = g.mul(g.sub(sum0, y0), g.inv(g.sub(g.base(sx), z0)));
    old_eval = g.mul_add(old_eval, apow[nch], g.mul(g.sub(sum1, y1), g.inv(g.sub(g.base(sx), z1))));
    int lg = in.log_N, used = 0;
    for (int l = 0; l < in.nlayers; l++) {
      const int ab = in.arity[l], m = 1 << ab;
      const std::vector<BoolT> within(bits.begin() + used, bits.begin() + used + ab);
      { std::vector<Target> leaf; for (const ExtT& e : q.evals[l]) { leaf.push_back(e.a); leaf.push_back(e.b); }
        verify_merkle(b, leaf, q.lsibs[l], bits.data() + used + ab, pf.fri_caps[l], cap_ind, cap_sel); }
      if (rg) {                                                                  // evals[within] == the value carried from the layer below (two RandomAccess copies)
        Target widx = b.zero(); for (int i = ab; i-- > 0;) widx = b.arithmetic(2, 1, widx, b.one(), within[i].t);
        std::vector<Target> la, lb; for (int k = 0; k < m; k++) { la.push_back(q.evals[l][k].a); lb.push_back(q.evals[l][k].b); }
        g.assert_eq(ExtT{b.random_access(widx, la), b.random_access(widx, lb)}, old_eval);
      } else { const std::vector<Target> ind = indicators(b, within);
        ExtT sel = g.cst(0); for (int k = 0; k < m; k++) sel = g.mul_base_add(q.evals[l][k], ind[k], sel);
        g.assert_eq(sel, old_eval); }
      // interpolate the coset {c0 g^i} -> evals[rev(i)] at beta (barycentric): P(beta) = (beta^m - c0^m) / (m c0^m) * sum_i y_i x_i / (beta - x_i)
      const u64 gm = gl_root(ab);
      Target c0 = sx;                                                            // sx * g^(-rev(within))
      for (int i = 0; i < ab; i++) c0 = b.mul(c0, b.mul_add(within[i].t, b.constant(gl_sub(gl_pow(gm, (u64)m - (u64(1) << (ab - 1 - i))), 1)), b.one()));
      const ExtT beta = fri_betas[l];
      if (rg && ab == COSET_BITS) {                                              // one CosetInterpolation row: the polynomial through (c0 g^i, evals[rev(i)]) at beta
        ExtT vals[COSET_NP]; for (int i = 0; i < m; i++) vals[i] = q.evals[l][rev_bits((u64)i, ab)];
        old_eval = b.coset_interpolate(c0, vals, beta);
        for (int i = 0; i < ab; i++) sx = b.mul(sx, sx);
        used += ab; lg -= ab;
        continue;
      }
      ExtT acc = g.cst(0);
      for (int i = 0; i < m; i++) {
        const Target xi = b.mul(c0, b.constant(gl_pow(gm, (u64)i)));
        const ExtT di = g.inv(ExtT{b.sub(beta.a, xi), beta.b});
        acc = g.mul_add(g.mul_base(q.evals[l][rev_bits((u64)i, ab)], xi), di, acc);
      }
      Target cm = c0; for (int i = 0; i < ab; i++) cm = b.mul(cm, cm);
      const Target mcm = b.mul_const((u64)m, cm), inv_mcm = b.inverse_hint(mcm);
      b.connect(b.mul(mcm, inv_mcm), b.one());
      const ExtT zb = ExtT{b.sub(beta_pow[l].a, cm), beta_pow[l].b};
      old_eval = g.mul(g.mul_base(zb, inv_mcm), acc);
      for (int i = 0; i < ab; i++) sx = b.mul(sx, sx);
      used += ab; lg -= ab;
    }
    ExtT fx = g.cst(0); for (size_t i = pf.final_poly.size(); i-- > 0;) fx = g.mul_base_add(fx, sx, pf.final_poly[i]);
    g.assert_eq(fx, old_eval);
    (void)lg;
  }
  b.end_segment();
  section("queries");
}

} // namespace qpc

using namespace qpc;

// ---- C-ABI (include/qpcircuit.h): the aggregation circuit of one tree level
#include "circuit_object.h"
namespace { struct AggInfo { Inner inner; uint32_t count; }; }
static void qpc_internal_set_error(const char* m) { qpc_set_error(m); }
static qpc_circuit* qpc_internal_new_circuit(uint32_t kind, bool zk) { return new qpc_circuit(kind, zk); }
static Builder* qpc_internal_builder(qpc_circuit* c) { return &c->b; }
static void qpc_internal_set_aux(qpc_circuit* c, void* aux, void (*del)(void*)) { c->aux = aux; c->aux_del = del; }
static void* qpc_internal_aux(const qpc_circuit* c) { return c->kind == QPC_AGGREGATION ? c->aux : nullptr; }

extern "C" int qpc_aggregation_circuit_build_ex(const qpgpu_circuit_desc* inner_desc, uint32_t num_proofs, uint32_t zero_knowledge, qpc_circuit** out);
extern "C" int qpc_aggregation_circuit_build(const qpgpu_circuit_desc* inner_desc, uint32_t num_proofs, qpc_circuit** out) {
  return qpc_aggregation_circuit_build_ex(inner_desc, num_proofs, 2, out);
}
extern "C" int qpc_aggregation_circuit_build_ex(const qpgpu_circuit_desc* inner_desc, uint32_t num_proofs, uint32_t zero_knowledge, qpc_circuit** out) {
  try {
    if (!inner_desc || !out || num_proofs < 1 || num_proofs > 16) { qpc_internal_set_error("bad argument"); return QPGPU_ERR_ARG; }
    AggInfo* info = new AggInfo(); info->count = num_proofs;
    const std::string e = info->inner.init(*inner_desc);
    if (!e.empty()) { delete info; qpc_internal_set_error(e.c_str()); return QPGPU_ERR_ARG; }
    // CircuitBuilder::new(common_data.config.clone()) (tree.rs:111): zk children give a zk node, unless the caller decides otherwise (_ex)
    if (zero_knowledge > 2) { delete info; qpc_internal_set_error("zero_knowledge must be 0, 1 or 2 (= as the children)"); return QPGPU_ERR_ARG; }
    qpc_circuit* c = qpc_internal_new_circuit(QPC_AGGREGATION, zero_knowledge == 2 ? inner_desc->zero_knowledge != 0 : zero_knowledge != 0);
    qpc_internal_set_aux(c, info, [](void* p) { delete (AggInfo*)p; });
    Builder& b = *qpc_internal_builder(c);
    const VerifierDataT vd = add_verifier_data_targets(b, info->inner);         // tree.rs:113-114
    for (uint32_t i = 0; i < num_proofs; i++) {                                 // tree.rs:117-126
      const ProofT pt = add_proof_targets(b, info->inner);
      verify_proof(b, info->inner, pt, vd, getenv("QPC_NODE_LEAF_GATES") == nullptr);
      for (Target t : pt.pis) b.register_public_input(t);
    }
    b.build();
    *out = c; return 0;
  } catch (const std::bad_alloc&) { qpc_internal_set_error("host memory allocation failed"); return QPGPU_ERR_OOM; }
  catch (...) { qpc_internal_set_error("internal error"); return QPGPU_ERR_ARG; }
}

// set_verifier_data_target + set_proof_with_pis_target (tree.rs:131-135): verifier data, then each proof's field elements in byte order
extern "C" int qpc_aggregation_fill_targets(const qpc_circuit* c, const uint64_t* constants_sigmas_cap, const uint64_t circuit_digest[4], const uint8_t* proofs, size_t stride,
                                            size_t proof_len, uint32_t count, uint64_t* inputs, size_t cap, size_t* len) {
  try {
    const AggInfo* info = c ? (const AggInfo*)qpc_internal_aux(c) : nullptr;
    if (!info || !constants_sigmas_cap || !circuit_digest || !proofs || !len) { qpc_internal_set_error("bad argument (not an aggregation circuit?)"); return QPGPU_ERR_ARG; }
    const Inner& in = info->inner; const Layout& L = in.lay;
    if (count != info->count) { qpc_internal_set_error(("this node circuit takes " + std::to_string(info->count) + " proofs").c_str()); return QPGPU_ERR_ARG; }
    if (proof_len != L.total || stride < proof_len) { qpc_internal_set_error(("proof length " + std::to_string(proof_len) + ", the inner circuit's proofs have " + std::to_string(L.total) + " bytes").c_str()); return QPGPU_ERR_ARG; }
    std::vector<u64> v;
    auto canon = [](u64 x) { return x >= P ? x - P : x; };
    for (size_t i = 0; i < (size_t(4) << in.cap_h); i++) v.push_back(canon(constants_sigmas_cap[i]));
    for (int i = 0; i < 4; i++) v.push_back(canon(circuit_digest[i]));
    for (uint32_t pi = 0; pi < count; pi++) {
      const uint8_t* pf = proofs + (size_t)pi * stride;
      auto words = [&](size_t off, size_t cnt) { for (size_t i = 0; i < cnt; i++) { u64 x; memcpy(&x, pf + off + 8 * i, 8); v.push_back(canon(x)); } };
      words(0, L.queries_off / 8);                                               // caps, openings, FRI caps
      const size_t widths[4] = {(size_t)in.S, (size_t)in.W + in.salt, (size_t)in.Zp + in.salt, (size_t)in.Q + in.salt};
      const size_t path = in.log_N - in.cap_h;
      for (uint32_t q = 0; q < in.d.fri_num_queries; q++) {
        size_t off = L.queries_off + (size_t)q * L.bytes_per_query;
        for (int t = 0; t < 4; t++) {
          words(off, widths[t]); off += 8 * widths[t];
          if (pf[off] != path) { qpc_internal_set_error("malformed proof: initial Merkle proof length"); return QPGPU_ERR_ARG; }
          words(off + 1, 4 * path); off += 1 + 32 * path;
        }
        int lg = in.log_N;
        for (int l = 0; l < in.nlayers; l++) {
          const int ab = in.arity[l]; const size_t fp = lg - ab - in.cap_h;
          words(off, size_t(2) << ab); off += size_t(16) << ab;
          if (pf[off] != fp) { qpc_internal_set_error("malformed proof: FRI Merkle proof length"); return QPGPU_ERR_ARG; }
          words(off + 1, 4 * fp); off += 1 + 32 * fp; lg -= ab;
        }
      }
      words(L.final_poly_off, 2 * L.final_poly_len + 1);                         // final polynomial, pow witness
      { u64 npi; memcpy(&npi, pf + L.pis_off, 8); if (npi != (u64)in.npi) { qpc_internal_set_error("malformed proof: public input count"); return QPGPU_ERR_ARG; } }
      words(L.pis_off + 8, in.npi);
    }
    *len = v.size();
    if (!inputs) return 0;
    if (cap < v.size()) { qpc_internal_set_error("output buffer too small"); return QPGPU_ERR_BUFFER; }
    memcpy(inputs, v.data(), v.size() * sizeof(u64)); return 0;
  } catch (...) { qpc_internal_set_error("internal error"); return QPGPU_ERR_ARG; }
}
