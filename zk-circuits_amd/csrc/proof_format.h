// Proof byte layout and Fiat-Shamir schedule shared by the prover (qpgpu.hip), the transcript replay and the verifier (verify.hip).
// Host-only. Reference semantics: qp-plonky2 1.1.1's Challenger and ProofWithPublicInputs::to_bytes,
// behind wormhole/prover/src/lib.rs:233-237 and wormhole/verifier/src/lib.rs:155-159.
#pragma once
#include "../../include/qpgpu.h"
#include "gl.h"
#include "poseidon.h"

namespace qp {

// ---- host-side Challenger (duplex sponge)
struct Challenger {
  u64 st[12] = {0}; u64 in[8]; int n_in = 0; u64 out[8]; int n_out = 0;
  void duplex() { for (int i = 0; i < n_in; i++) st[i] = in[i]; n_in = 0; poseidon_permute(st); for (int i = 0; i < 8; i++) out[i] = st[i]; n_out = 8; }
  void observe(u64 x) { n_out = 0; in[n_in++] = x; if (n_in == 8) duplex(); }
  void observe_n(const u64* p, size_t k) { for (size_t i = 0; i < k; i++) observe(p[i]); }
  u64 challenge() { if (n_in > 0 || n_out == 0) duplex(); return out[--n_out]; }
  Ext challenge_ext() { u64 a = challenge(); u64 b = challenge(); return ex(a, b); }
};

// The proof's Fiat-Shamir schedule, shared by the prover below and by qpgpu_replay_transcript (which
// re-derives every challenge from finished proof bytes, e.g. the reference's golden bench-data/proof.bin).
struct Transcript {
  Challenger ch; int nch;
  explicit Transcript(int nch_) : nch(nch_) {}
  void start(const u64 digest[4], const u64 pi_hash[4], const u64* wires_cap, size_t cap_words, u64* betas, u64* gammas) {
    ch.observe_n(digest, 4); ch.observe_n(pi_hash, 4); ch.observe_n(wires_cap, cap_words);
    for (int i = 0; i < nch; i++) betas[i] = ch.challenge();
    for (int i = 0; i < nch; i++) gammas[i] = ch.challenge();
  }
  void after_zs_cap(const u64* cap, size_t cap_words, u64* alphas) { ch.observe_n(cap, cap_words); for (int i = 0; i < nch; i++) alphas[i] = ch.challenge(); }
  Ext after_quotient_cap(const u64* cap, size_t cap_words) { ch.observe_n(cap, cap_words); return ch.challenge_ext(); }
  // openings in proof order: constants|sigmas|wires|zs (a ext values), zs_next (nch), partial products|quotient (b ext values);
  // observed as constants..zs, partial products, quotient, then zs_next
  Ext after_openings(const u64* op, size_t a_ext, size_t b_ext) {
    ch.observe_n(op, 2 * a_ext); ch.observe_n(op + 2 * a_ext + 2 * (size_t)nch, 2 * b_ext); ch.observe_n(op + 2 * a_ext, 2 * (size_t)nch);
    return ch.challenge_ext();
  }
  Ext fri_layer_cap(const u64* cap, size_t cap_words) { ch.observe_n(cap, cap_words); return ch.challenge_ext(); }
  void final_poly_coeff(u64 a, u64 b) { ch.observe(a); ch.observe(b); }
  u64 pow_response(u64 pow_witness) { ch.observe(pow_witness); return ch.challenge(); }
  u32 query_index(size_t N) { return (u32)(ch.challenge() % N); }
};

struct Layout {   // ProofWithPublicInputs::to_bytes() layout
  size_t cap_bytes, openings_off, n_openings, fri_caps_off, queries_off, bytes_per_query, final_poly_off, final_poly_len, pow_off, pis_off, total;
};

// ProofWithPublicInputs::to_bytes() offsets from the CommonCircuitData shape alone.
inline void compute_layout(const qpgpu_circuit_desc& d, Layout& L) {
  const int log_n = d.degree_bits, rate = d.fri_rate_bits, cap_h = d.fri_cap_height, nch = d.num_challenges, salt = d.zero_knowledge ? 4 : 0;
  const size_t S = (size_t)d.num_constants + d.num_routed, W = d.num_wires, Zp = (size_t)nch * (1 + d.num_partial_products), Q = (size_t)nch * d.quotient_degree_factor;
  const size_t capn = size_t(1) << cap_h;
  L.cap_bytes = capn * 32; L.openings_off = 3 * L.cap_bytes;
  L.n_openings = S + W + 2 * nch + (size_t)nch * d.num_partial_products + Q;
  L.fri_caps_off = L.openings_off + 16 * L.n_openings; L.queries_off = L.fri_caps_off + (size_t)d.fri_num_layers * L.cap_bytes;
  size_t path = log_n + rate - cap_h; const size_t widths[4] = {S, W + salt, Zp + salt, Q + salt};
  size_t b = 0; for (int t = 0; t < 4; t++) b += 8 * widths[t] + 1 + 32 * path;
  int lg = log_n;
  for (uint32_t l = 0; l < d.fri_num_layers; l++) { const int a = d.fri_arity_bits[l]; size_t fp = lg + rate - a - cap_h; b += (size_t(16) << a) + 1 + 32 * fp; lg -= a; }
  L.bytes_per_query = b;
  L.final_poly_off = L.queries_off + L.bytes_per_query * d.fri_num_queries; L.final_poly_len = size_t(1) << lg;
  L.pow_off = L.final_poly_off + 16 * L.final_poly_len; L.pis_off = L.pow_off + 8; L.total = L.pis_off + 8 + 8 * (size_t)d.num_public_inputs;
}


} // namespace qp
