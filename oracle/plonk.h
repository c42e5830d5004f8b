// ORACLE — TEST INFRASTRUCTURE ONLY (see gl.h header).
// Shared data model for the CPU restatement of qp-plonky2 1.1.1's prover/verifier
// (source not on disk; call sites wormhole/prover/src/lib.rs:233-237,
// wormhole/verifier/src/lib.rs:155-159). Byte layouts: SURVEY.md App. B, validated
// against wormhole/bench-data/{common,verifier,proof}.bin.
#pragma once
#include "gl.h"
#include "poseidon.h"
#include <vector>
#include <string>
#include <stdexcept>
#include <utility>

namespace orc {

enum GateTag { G_ARITH = 0, G_ARITH_EXT = 1, G_BASESUM2 = 2, G_CONSTANT = 3, G_COSET_INTERP = 4, G_EXP = 5,
               G_LOOKUP = 6, G_LOOKUP_TABLE = 7, G_MUL_EXT = 8, G_NOOP = 9, G_POSEIDON_MDS = 10, G_POSEIDON = 11,
               G_PUBLIC_INPUT = 12, G_RANDOM_ACCESS = 13, G_REDUCING_EXT = 14, G_REDUCING = 15 };

// param/p2/p3/table by gate: ArithmeticExtension, MulExtension: num_ops; Reducing, ReducingExtension: num_coeffs;
// Exponentiation: num_power_bits; RandomAccess: (bits, num_copies, num_extra_constants); CosetInterpolation: (subgroup_bits, degree,
// table = barycentric weights[2^subgroup_bits]).
struct Gate { int tag; u64 param; u64 p2 = 0, p3 = 0; std::vector<u64> table; };

struct Common {
  int num_wires = 0, num_routed = 0, cfg_num_constants = 0, security_bits = 0, num_challenges = 0, max_qdf = 0;
  bool use_base_arith = false, zk = false;
  int rate_bits = 0, cap_height = 0, num_queries = 0, pow_bits = 0;
  std::vector<int> arity_bits; int degree_bits = 0; bool hiding = false;
  std::vector<int> selector_indices; std::vector<std::pair<int, int>> groups;
  int qdf = 0, num_gate_constraints = 0, num_constants = 0, num_public_inputs = 0;
  std::vector<u64> k_is; int num_partial_products = 0; std::vector<Gate> gates;

  size_t n() const { return size_t(1) << degree_bits; }
  size_t lde_n() const { return size_t(1) << (degree_bits + rate_bits); }
  int num_selectors() const { return (int)groups.size(); }
  int num_preprocessed() const { return num_constants + num_routed; }
  int num_zs_pp() const { return num_challenges * (1 + num_partial_products); }
  int num_quotient() const { return num_challenges * qdf; }
  int salt() const { return hiding ? 4 : 0; }
  size_t final_poly_len() const { int b = degree_bits; for (int a : arity_bits) b -= a; return size_t(1) << b; }
};

struct VerifierOnly { int cap_height = 0; std::vector<Hash> constants_sigmas_cap; Hash circuit_digest; };

struct MerkleProofData { std::vector<Hash> siblings; };
struct InitialTreeEntry { std::vector<u64> evals; MerkleProofData proof; };
struct QueryStep { std::vector<Ext> evals; MerkleProofData proof; };
struct QueryRound { std::vector<InitialTreeEntry> initial; std::vector<QueryStep> steps; };

struct Proof {
  std::vector<Hash> wires_cap, zs_pp_cap, quotient_cap;
  std::vector<Ext> o_constants, o_sigmas, o_wires, o_zs, o_zs_next, o_pp, o_quotient;
  std::vector<std::vector<Hash>> fri_caps;
  std::vector<QueryRound> queries;
  std::vector<Ext> final_poly;
  u64 pow_witness = 0;
  std::vector<u64> public_inputs;
};

Common parse_common(const uint8_t* b, size_t len, size_t* used = nullptr);
VerifierOnly parse_verifier_only(const uint8_t* b, size_t len, size_t* used = nullptr);
Proof parse_proof(const uint8_t* b, size_t len, const Common& c);
std::vector<uint8_t> serialize_proof(const Proof& p);

// Challenger (A.5)
struct Challenger {
  u64 st[12] = {0}; std::vector<u64> in, out;
  void observe(u64 x) { out.clear(); in.push_back(x); if (in.size() == 8) duplex(); }
  void observe_hash(const Hash& h) { for (u64 x : h) observe(x); }
  void observe_cap(const std::vector<Hash>& c) { for (auto& h : c) observe_hash(h); }
  void observe_ext(Ext e) { observe(e.a); observe(e.b); }
  void duplex() { for (size_t i = 0; i < in.size(); i++) st[i] = in[i]; in.clear(); poseidon_permute(st); out.assign(st, st + 8); }
  u64 challenge() { if (!in.empty() || out.empty()) duplex(); u64 r = out.back(); out.pop_back(); return r; }
  Ext challenge_ext() { u64 a = challenge(); u64 b = challenge(); return Ext{a, b}; }
};

// Merkle (A.4)
bool merkle_verify_to_cap(const u64* leaf, size_t len, size_t index, const std::vector<Hash>& cap, const MerkleProofData& pr);
struct MerkleTree {
  size_t num_leaves = 0; int cap_height = 0;
  std::vector<std::vector<Hash>> levels;  // levels[0] = leaf digests, last = cap
  const std::vector<Hash>& cap() const { return levels.back(); }
  MerkleProofData prove(size_t index) const {
    MerkleProofData p; for (size_t l = 0; l + 1 < levels.size(); l++) { p.siblings.push_back(levels[l][index ^ 1]); index >>= 1; } return p;
  }
};
// leaves given row-major: leaf i = rows[i*width .. +width)
MerkleTree merkle_build(const u64* rows, size_t num_leaves, size_t width, int cap_height);

// Gate constraint evaluation, generic over T in {FB, Ext}. constants = full local constants (selectors first).
template <class T> void eval_gate_constraints(const Common& c, const T* consts, const T* wires, const T pi_hash[4], T* out /*num_gate_constraints*/);
extern template void eval_gate_constraints<FB>(const Common&, const FB*, const FB*, const FB*, FB*);
extern template void eval_gate_constraints<Ext>(const Common&, const Ext*, const Ext*, const Ext*, Ext*);

// Verifier. Returns empty string on accept, else reason.
std::string verify(const Proof& p, const Common& c, const VerifierOnly& v);

// FFT (natural order in/out)
void ntt(std::vector<u64>& a, bool inverse);
std::vector<u64> coset_lde(const std::vector<u64>& coeffs, int rate_bits, u64 shift);  // zero-pad, eval on shift*<w_N>, natural order

// Prover inputs (the "circuit package"): constants_sigmas values [S][n] column-major, natural row order.
struct ProverInputs {
  const Common* common; Hash circuit_digest;
  const u64* constants_sigmas;   // [num_preprocessed][n]
  const u64* wires;              // [num_wires][n]
  const u64* public_inputs;      // [num_public_inputs]
  const u64* salts;              // zk only: [3][4][N] (oracle wires, zs_pp, quotient), each column in LEAF (bit-reversed) order; may be null if !hiding
  bool have_pow; u64 pow_witness; // if !have_pow: smallest valid witness
  int threads;
};
struct Preprocessed {  // constants_sigmas commitment (done once per circuit)
  std::vector<std::vector<u64>> coeffs;   // [S][n]
  std::vector<u64> lde_rows;              // N x S row-major, leaf order
  MerkleTree tree;
};
Preprocessed preprocess(const Common& c, const u64* constants_sigmas, int threads);
Proof prove(const ProverInputs& in, const Preprocessed& pre, std::string* err);

} // namespace orc
