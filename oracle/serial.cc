// ORACLE — TEST INFRASTRUCTURE ONLY. Byte formats of qp-plonky2 1.1.1 as recorded in SURVEY.md App. B
// (CommonCircuitData / VerifierOnlyCircuitData / ProofWithPublicInputs::to_bytes — used by the reference at
// wormhole/aggregator/src/util.rs:22, wormhole/verifier/benches/verifier.rs:16-26).
#include "plonk.h"
#include <cstring>

namespace orc {
namespace {
struct Rd {
  const uint8_t* b; size_t len, pos = 0;
  void need(size_t k) { if (pos + k > len) throw std::runtime_error("buffer underrun"); }
  u64 u64_() { need(8); u64 v; memcpy(&v, b + pos, 8); pos += 8; return v; }
  uint32_t u32_() { need(4); uint32_t v; memcpy(&v, b + pos, 4); pos += 4; return v; }
  uint8_t u8_() { need(1); return b[pos++]; }
  u64 felt() { u64 v = u64_(); if (v >= P) throw std::runtime_error("non-canonical field element"); return v; }
  Hash hash() { Hash h; for (auto& x : h) x = felt(); return h; }
  Ext ext() { Ext e; e.a = felt(); e.b = felt(); return e; }
};
struct Wr {
  std::vector<uint8_t> o;
  void u64_(u64 v) { size_t p = o.size(); o.resize(p + 8); memcpy(&o[p], &v, 8); }
  void u8_(uint8_t v) { o.push_back(v); }
  void hash(const Hash& h) { for (u64 x : h) u64_(x); }
  void ext(Ext e) { u64_(e.a); u64_(e.b); }
};
void read_fri_config(Rd& r, Common& c) {
  c.rate_bits = (int)r.u64_(); c.cap_height = (int)r.u64_(); c.num_queries = (int)r.u64_(); c.pow_bits = (int)r.u32_();
  uint8_t tag = r.u8_();
  if (tag == 0) { u64 n = r.u64_(); for (u64 i = 0; i < n; i++) r.u64_(); }   // Fixed(vec)
  else if (tag == 1) { r.u64_(); r.u64_(); }                                  // ConstantArityBits(a, b)
  else if (tag == 2) { r.u64_(); }                                            // MinSize(Option) — not seen in goldens
  else throw std::runtime_error("unknown FRI reduction strategy");
}
}

Common parse_common(const uint8_t* b, size_t len, size_t* used) {
  Rd r{b, len}; Common c;
  c.num_wires = (int)r.u64_(); c.num_routed = (int)r.u64_(); c.cfg_num_constants = (int)r.u64_();
  c.security_bits = (int)r.u64_(); c.num_challenges = (int)r.u64_(); c.max_qdf = (int)r.u64_();
  c.use_base_arith = r.u8_(); c.zk = r.u8_();
  read_fri_config(r, c);
  read_fri_config(r, c);                         // fri_params.config (same content)
  u64 na = r.u64_(); for (u64 i = 0; i < na; i++) c.arity_bits.push_back((int)r.u64_());
  c.degree_bits = (int)r.u64_(); c.hiding = r.u8_();
  u64 ns = r.u64_(); for (u64 i = 0; i < ns; i++) c.selector_indices.push_back((int)r.u64_());
  u64 ng = r.u64_(); for (u64 i = 0; i < ng; i++) { int lo = (int)r.u64_(); int hi = (int)r.u64_(); c.groups.push_back({lo, hi}); }
  c.qdf = (int)r.u64_(); c.num_gate_constraints = (int)r.u64_(); c.num_constants = (int)r.u64_(); c.num_public_inputs = (int)r.u64_();
  u64 nk = r.u64_(); for (u64 i = 0; i < nk; i++) c.k_is.push_back(r.felt());
  c.num_partial_products = (int)r.u64_();
  u64 nlp = r.u64_(), nls = r.u64_(), nluts = r.u64_();
  if (nlp || nls || nluts) throw std::runtime_error("lookups unsupported");
  u64 ngates = r.u64_();
  for (u64 i = 0; i < ngates; i++) {
    Gate g; g.tag = (int)r.u32_(); g.param = 0;
    switch (g.tag) {
      case G_NOOP: case G_PUBLIC_INPUT: case G_POSEIDON: case G_POSEIDON_MDS: break;
      case G_CONSTANT: case G_BASESUM2: case G_ARITH: case G_ARITH_EXT: case G_MUL_EXT: case G_REDUCING: case G_REDUCING_EXT:
      case G_EXP: g.param = r.u64_(); break;
      // [RECALLED field order, no golden recursion common data ships: SURVEY.md App. D]
      case G_RANDOM_ACCESS: g.param = r.u64_(); g.p2 = r.u64_(); g.p3 = r.u64_(); break;
      case G_COSET_INTERP: { g.param = r.u64_(); g.p2 = r.u64_(); u64 nw = r.u64_(); for (u64 k = 0; k < nw; k++) g.table.push_back(r.felt()); break; }
      default: throw std::runtime_error("gate tag not supported by oracle parser");
    }
    c.gates.push_back(g);
  }
  if (used) *used = r.pos;
  return c;
}

VerifierOnly parse_verifier_only(const uint8_t* b, size_t len, size_t* used) {
  Rd r{b, len}; VerifierOnly v;
  v.cap_height = (int)r.u64_();
  for (int i = 0; i < (1 << v.cap_height); i++) v.constants_sigmas_cap.push_back(r.hash());
  v.circuit_digest = r.hash();
  if (used) *used = r.pos;
  return v;
}

Proof parse_proof(const uint8_t* b, size_t len, const Common& c) {
  Rd r{b, len}; Proof p;
  int capn = 1 << c.cap_height;
  auto cap = [&](std::vector<Hash>& v) { for (int i = 0; i < capn; i++) v.push_back(r.hash()); };
  auto exts = [&](std::vector<Ext>& v, int k) { for (int i = 0; i < k; i++) v.push_back(r.ext()); };
  cap(p.wires_cap); cap(p.zs_pp_cap); cap(p.quotient_cap);
  exts(p.o_constants, c.num_constants); exts(p.o_sigmas, c.num_routed); exts(p.o_wires, c.num_wires);
  exts(p.o_zs, c.num_challenges); exts(p.o_zs_next, c.num_challenges);
  exts(p.o_pp, c.num_challenges * c.num_partial_products); exts(p.o_quotient, c.num_quotient());
  p.fri_caps.resize(c.arity_bits.size());
  for (auto& fc : p.fri_caps) cap(fc);
  int widths[4] = {c.num_preprocessed(), c.num_wires + c.salt(), c.num_zs_pp() + c.salt(), c.num_quotient() + c.salt()};
  for (int q = 0; q < c.num_queries; q++) {
    QueryRound qr;
    for (int t = 0; t < 4; t++) {
      InitialTreeEntry e; for (int i = 0; i < widths[t]; i++) e.evals.push_back(r.felt());
      int ns = r.u8_(); for (int i = 0; i < ns; i++) e.proof.siblings.push_back(r.hash());
      qr.initial.push_back(std::move(e));
    }
    for (size_t l = 0; l < c.arity_bits.size(); l++) {
      QueryStep s; exts(s.evals, 1 << c.arity_bits[l]);
      int ns = r.u8_(); for (int i = 0; i < ns; i++) s.proof.siblings.push_back(r.hash());
      qr.steps.push_back(std::move(s));
    }
    p.queries.push_back(std::move(qr));
  }
  exts(p.final_poly, (int)c.final_poly_len());
  p.pow_witness = r.felt();
  u64 npi = r.u64_();
  for (u64 i = 0; i < npi; i++) p.public_inputs.push_back(r.felt());
  if (r.pos != len) throw std::runtime_error("trailing bytes after proof");
  return p;
}

std::vector<uint8_t> serialize_proof(const Proof& p) {
  Wr w;
  for (auto& h : p.wires_cap) w.hash(h);
  for (auto& h : p.zs_pp_cap) w.hash(h);
  for (auto& h : p.quotient_cap) w.hash(h);
  for (auto* v : {&p.o_constants, &p.o_sigmas, &p.o_wires, &p.o_zs, &p.o_zs_next, &p.o_pp, &p.o_quotient}) for (Ext e : *v) w.ext(e);
  for (auto& fc : p.fri_caps) for (auto& h : fc) w.hash(h);
  for (auto& q : p.queries) {
    for (auto& e : q.initial) { for (u64 x : e.evals) w.u64_(x); w.u8_((uint8_t)e.proof.siblings.size()); for (auto& h : e.proof.siblings) w.hash(h); }
    for (auto& s : q.steps) { for (Ext e : s.evals) w.ext(e); w.u8_((uint8_t)s.proof.siblings.size()); for (auto& h : s.proof.siblings) w.hash(h); }
  }
  for (Ext e : p.final_poly) w.ext(e);
  w.u64_(p.pow_witness);
  w.u64_(p.public_inputs.size());
  for (u64 x : p.public_inputs) w.u64_(x);
  return w.o;
}

} // namespace orc
