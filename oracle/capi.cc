// ORACLE — TEST INFRASTRUCTURE ONLY. C entry points used by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg (ctypes). Never linked by the product library.
#include "plonk.h"
#include <cstring>
#include <cstdio>
#include <omp.h>

using namespace orc;

static thread_local std::string g_err;
static int fail(const std::string& e) { g_err = e; return -1; }

struct OrcCircuit { Common common; Hash digest; std::vector<u64> constants_sigmas; Preprocessed pre; };

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
void orc_round_constants(u64* out) { memcpy(out, poseidon_round_constants(), 360 * 8); }
void orc_permute(u64* st) { poseidon_permute(st); }
void orc_set_poseidon_fast(int on) { poseidon_set_fast(on != 0); }
int orc_poseidon_is_fast() { return poseidon_is_fast() ? 1 : 0; }
void orc_permute_generic(u64* st) { poseidon_permute_generic(st); }
void orc_hash_no_pad(const u64* x, size_t len, u64* out) { Hash h = hash_no_pad(x, len); memcpy(out, h.data(), 32); }
void orc_hash_or_noop(const u64* x, size_t len, u64* out) { Hash h = hash_or_noop(x, len); memcpy(out, h.data(), 32); }
void orc_two_to_one(const u64* l, const u64* r, u64* out) { Hash a, b; memcpy(a.data(), l, 32); memcpy(b.data(), r, 32); Hash h = two_to_one(a, b); memcpy(out, h.data(), 32); }
u64 orc_mul(u64 a, u64 b) { return gl_mul(a, b); }
u64 orc_mul_slow(u64 a, u64 b) { return gl_red128_slow((u128)a * b); }
u64 orc_inv(u64 a) { return gl_inv(a); }
u64 orc_root(int bits) { return gl_root(bits); }

void orc_ntt(u64* a, size_t n, int inverse) { std::vector<u64> v(a, a + n); ntt(v, inverse != 0); memcpy(a, v.data(), n * 8); }
// out[N] natural order values of the zero-padded polynomial on shift*<w_N>
void orc_coset_lde(const u64* coeffs, size_t n, int rate_bits, u64 shift, u64* out) {
  std::vector<u64> c(coeffs, coeffs + n); auto v = coset_lde(c, rate_bits, shift); memcpy(out, v.data(), v.size() * 8);
}
// digests: all levels concatenated (level 0 first), 4 u64 per digest; returns number of digests written
size_t orc_merkle(const u64* rows, size_t num_leaves, size_t width, int cap_height, u64* digests) {
  MerkleTree t = merkle_build(rows, num_leaves, width, cap_height); size_t k = 0;
  for (auto& lvl : t.levels) for (auto& h : lvl) { memcpy(digests + 4 * k, h.data(), 32); k++; }
  return k;
}

// Verify a proof given serialized common + verifier-only data (the reference's bench-data layout).
int orc_verify_bytes(const uint8_t* common_b, size_t common_len, const uint8_t* vo_b, size_t vo_len, const uint8_t* proof_b, size_t proof_len) {
  try {
    Common c = parse_common(common_b, common_len);
    VerifierOnly v = parse_verifier_only(vo_b, vo_len);
    Proof p = parse_proof(proof_b, proof_len, c);
    std::string r = verify(p, c, v);
    if (!r.empty()) return fail(r);
    return 0;
  } catch (std::exception& e) { return fail(e.what()); }
}
// Parse + re-serialize (round trip check); returns length or -1.
long orc_proof_roundtrip(const uint8_t* common_b, size_t common_len, const uint8_t* proof_b, size_t proof_len, uint8_t* out, size_t cap) {
  try {
    Common c = parse_common(common_b, common_len);
    Proof p = parse_proof(proof_b, proof_len, c);
    auto o = serialize_proof(p);
    if (o.size() > cap) return fail("cap");
    memcpy(out, o.data(), o.size()); return (long)o.size();
  } catch (std::exception& e) { return fail(e.what()); }
}
// Transcript pins (A.6): out = [pi_hash(4), betas(2), gammas(2), alphas(2), zeta(2), fri_alpha(2), fri_beta0(2), pow_response, first 4 indices]
int orc_transcript_pins(const uint8_t* common_b, size_t common_len, const uint8_t* vo_b, size_t vo_len, const uint8_t* proof_b, size_t proof_len, u64* out) {
  try {
    Common c = parse_common(common_b, common_len); VerifierOnly vo = parse_verifier_only(vo_b, vo_len); Proof p = parse_proof(proof_b, proof_len, c);
    Hash pih = hash_no_pad(p.public_inputs.data(), p.public_inputs.size());
    Challenger ch; ch.observe_hash(vo.circuit_digest); ch.observe_hash(pih); ch.observe_cap(p.wires_cap);
    int k = 0; for (u64 x : pih) out[k++] = x;
    for (int i = 0; i < 4; i++) out[k++] = ch.challenge();
    ch.observe_cap(p.zs_pp_cap); out[k++] = ch.challenge(); out[k++] = ch.challenge();
    ch.observe_cap(p.quotient_cap); Ext z = ch.challenge_ext(); out[k++] = z.a; out[k++] = z.b;
    for (auto* v : {&p.o_constants, &p.o_sigmas, &p.o_wires, &p.o_zs, &p.o_pp, &p.o_quotient}) for (Ext e : *v) ch.observe_ext(e);
    for (Ext e : p.o_zs_next) ch.observe_ext(e);
    Ext a = ch.challenge_ext(); out[k++] = a.a; out[k++] = a.b;
    bool first = true;
    for (auto& cap : p.fri_caps) { ch.observe_cap(cap); Ext b = ch.challenge_ext(); if (first) { out[k++] = b.a; out[k++] = b.b; first = false; } }
    for (Ext e : p.final_poly) ch.observe_ext(e);
    ch.observe(p.pow_witness); out[k++] = ch.challenge();
    for (int i = 0; i < 4; i++) out[k++] = ch.challenge() % c.lde_n();
    return 0;
  } catch (std::exception& e) { return fail(e.what()); }
}


// gate_extra: NULL (leaf gate set) or [n_gates][ORC_GATE_EXTRA_STRIDE] u64 = {p2, p3, table_len, table[<=16]...}
static const int ORC_GATE_EXTRA_STRIDE = 20;
static Gate make_gate(int tag, u64 param, const u64* extra, int i) {
  Gate g; g.tag = tag; g.param = param;
  if (extra) { const u64* e = extra + (size_t)i * ORC_GATE_EXTRA_STRIDE; g.p2 = e[0]; g.p3 = e[1]; for (u64 k = 0; k < e[2] && k < 16; k++) g.table.push_back(e[3 + k]); }
  return g;
}

// ---- circuit handle: CommonCircuitData fields + constants/sigmas values (the "circuit package")
void* orc_circuit_create(int degree_bits, int num_wires, int num_routed, int num_constants, int num_challenges, int qdf,
                         int num_partial_products, int num_public_inputs, int num_gate_constraints, int zk,
                         int rate_bits, int cap_height, int num_queries, int pow_bits, int n_layers, const uint32_t* arity_bits,
                         int n_gates, const uint32_t* gate_tags, const u64* gate_params, const u64* gate_extra, const uint32_t* selector_indices,
                         int n_groups, const uint32_t* group_ranges, const u64* k_is, const u64* constants_sigmas,
                         const u64* digest, int threads) {
  try {
    auto* oc = new OrcCircuit(); Common& c = oc->common;
    c.num_wires = num_wires; c.num_routed = num_routed; c.cfg_num_constants = 2; c.security_bits = 100; c.num_challenges = num_challenges;
    c.max_qdf = qdf; c.use_base_arith = true; c.zk = zk; c.rate_bits = rate_bits; c.cap_height = cap_height; c.num_queries = num_queries;
    c.pow_bits = pow_bits; for (int i = 0; i < n_layers; i++) c.arity_bits.push_back((int)arity_bits[i]);
    c.degree_bits = degree_bits; c.hiding = zk;
    for (int i = 0; i < n_gates; i++) { c.selector_indices.push_back((int)selector_indices[i]); c.gates.push_back(make_gate((int)gate_tags[i], gate_params[i], gate_extra, i)); }
    for (int i = 0; i < n_groups; i++) c.groups.push_back({(int)group_ranges[2 * i], (int)group_ranges[2 * i + 1]});
    c.qdf = qdf; c.num_gate_constraints = num_gate_constraints; c.num_constants = num_constants; c.num_public_inputs = num_public_inputs;
    c.k_is.assign(k_is, k_is + num_routed); c.num_partial_products = num_partial_products;
    memcpy(oc->digest.data(), digest, 32);
    size_t n = c.n(); oc->constants_sigmas.assign(constants_sigmas, constants_sigmas + (size_t)c.num_preprocessed() * n);
    oc->pre = preprocess(c, oc->constants_sigmas.data(), threads);
    return oc;
  } catch (std::exception& e) { fail(e.what()); return nullptr; }
}
// Verifier-only handle: no preprocessing; the constants/sigmas cap is supplied (e.g. from the prover under test).
// Used for size-independent acceptance checks at sizes where the CPU preprocessing would take minutes.
void* orc_verifier_create(int degree_bits, int num_wires, int num_routed, int num_constants, int num_challenges, int qdf,
                          int num_partial_products, int num_public_inputs, int num_gate_constraints, int zk,
                          int rate_bits, int cap_height, int num_queries, int pow_bits, int n_layers, const uint32_t* arity_bits,
                          int n_gates, const uint32_t* gate_tags, const u64* gate_params, const u64* gate_extra, const uint32_t* selector_indices,
                          int n_groups, const uint32_t* group_ranges, const u64* k_is, const u64* cap, const u64* digest) {
  auto* oc = new OrcCircuit(); Common& c = oc->common;
  c.num_wires = num_wires; c.num_routed = num_routed; c.cfg_num_constants = 2; c.security_bits = 100; c.num_challenges = num_challenges;
  c.max_qdf = qdf; c.use_base_arith = true; c.zk = zk; c.rate_bits = rate_bits; c.cap_height = cap_height; c.num_queries = num_queries;
  c.pow_bits = pow_bits; for (int i = 0; i < n_layers; i++) c.arity_bits.push_back((int)arity_bits[i]);
  c.degree_bits = degree_bits; c.hiding = zk;
  for (int i = 0; i < n_gates; i++) { c.selector_indices.push_back((int)selector_indices[i]); c.gates.push_back(make_gate((int)gate_tags[i], gate_params[i], gate_extra, i)); }
  for (int i = 0; i < n_groups; i++) c.groups.push_back({(int)group_ranges[2 * i], (int)group_ranges[2 * i + 1]});
  c.qdf = qdf; c.num_gate_constraints = num_gate_constraints; c.num_constants = num_constants; c.num_public_inputs = num_public_inputs;
  c.k_is.assign(k_is, k_is + num_routed); c.num_partial_products = num_partial_products;
  memcpy(oc->digest.data(), digest, 32);
  oc->pre.tree.levels.emplace_back();
  for (int i = 0; i < (1 << cap_height); i++) { Hash h; memcpy(h.data(), cap + 4 * i, 32); oc->pre.tree.levels.back().push_back(h); }
  return oc;
}
void orc_circuit_destroy(void* h) { delete (OrcCircuit*)h; }
void orc_circuit_cap(void* h, u64* out) { auto* oc = (OrcCircuit*)h; size_t k = 0; for (auto& hh : oc->pre.tree.cap()) { memcpy(out + 4 * k, hh.data(), 32); k++; } }

long orc_prove(void* h, const u64* wires, const u64* public_inputs, const u64* salts, int have_pow, u64 pow_witness, int threads,
               uint8_t* out, size_t cap) {
  try {
    auto* oc = (OrcCircuit*)h;
    ProverInputs in{&oc->common, oc->digest, oc->constants_sigmas.data(), wires, public_inputs, salts, have_pow != 0, pow_witness, threads};
    std::string err; Proof p = prove(in, oc->pre, &err);
    if (!err.empty()) return fail(err);
    auto b = serialize_proof(p);
    if (b.size() > cap) return fail("output buffer too small");
    memcpy(out, b.data(), b.size()); return (long)b.size();
  } catch (std::exception& e) { return fail(e.what()); }
}
int orc_verify(void* h, const uint8_t* proof_b, size_t proof_len) {
  try {
    auto* oc = (OrcCircuit*)h;
    Proof p = parse_proof(proof_b, proof_len, oc->common);
    VerifierOnly v; v.cap_height = oc->common.cap_height; v.constants_sigmas_cap = oc->pre.tree.cap(); v.circuit_digest = oc->digest;
    std::string r = verify(p, oc->common, v);
    if (!r.empty()) return fail(r);
    return 0;
  } catch (std::exception& e) { return fail(e.what()); }
}
// Witness sanity: every gate constraint vanishes on every trace row, and the permutation argument closes
// (grand product over all routed cells is 1 for random beta/gamma). Returns 0 if consistent; else -1 with a message.
int orc_check_witness(void* h, const u64* wires, const u64* public_inputs) {
  auto* oc = (OrcCircuit*)h; const Common& c = oc->common; size_t n = c.n();
  Hash pih = hash_no_pad(public_inputs, c.num_public_inputs);
  FB pi[4] = {FB(pih[0]), FB(pih[1]), FB(pih[2]), FB(pih[3])};
  std::vector<FB> consts(c.num_constants), w(c.num_wires), out(c.num_gate_constraints);
  for (size_t i = 0; i < n; i++) {
    for (int k = 0; k < c.num_constants; k++) consts[k] = FB(oc->constants_sigmas[(size_t)k * n + i]);
    for (int k = 0; k < c.num_wires; k++) w[k] = FB(wires[(size_t)k * n + i]);
    eval_gate_constraints<FB>(c, consts.data(), w.data(), pi, out.data());
    for (int k = 0; k < c.num_gate_constraints; k++) if (out[k].v != 0) return fail("gate constraint " + std::to_string(k) + " nonzero at row " + std::to_string(i));
  }
  u64 beta = 0x123456789abcdefULL, gamma = 0xfedcba987654321ULL, num = 1, den = 1, wn = gl_root(c.degree_bits), x = 1;
  for (size_t i = 0; i < n; i++) {
    for (int j = 0; j < c.num_routed; j++) {
      u64 wv = wires[(size_t)j * n + i];
      num = gl_mul(num, gl_add(gl_add(wv, gl_mul(beta, gl_mul(c.k_is[j], x))), gamma));
      den = gl_mul(den, gl_add(gl_add(wv, gl_mul(beta, oc->constants_sigmas[(size_t)(c.num_constants + j) * n + i])), gamma));
    }
    x = gl_mul(x, wn);
  }
  if (num != den) return fail("copy constraints violated (grand product != 1)");
  return 0;
}
// Gate constraints on the listed rows only (the hint-soundness sweep hands over the rows that differ from a matrix that passed
// orc_check_witness: unchanged rows evaluate as they did). 0 = every constraint of every listed row is zero.
int orc_check_witness_rows(void* h, const u64* wires, const u64* public_inputs, const uint32_t* rows, size_t num_rows) {
  auto* oc = (OrcCircuit*)h; const Common& c = oc->common; size_t n = c.n();
  Hash pih = hash_no_pad(public_inputs, c.num_public_inputs);
  FB pi[4] = {FB(pih[0]), FB(pih[1]), FB(pih[2]), FB(pih[3])};
  std::vector<FB> consts(c.num_constants), w(c.num_wires), out(c.num_gate_constraints);
  for (size_t r = 0; r < num_rows; r++) {
    const size_t i = rows[r];
    if (i >= n) return fail("row out of range");
    for (int k = 0; k < c.num_constants; k++) consts[k] = FB(oc->constants_sigmas[(size_t)k * n + i]);
    for (int k = 0; k < c.num_wires; k++) w[k] = FB(wires[(size_t)k * n + i]);
    eval_gate_constraints<FB>(c, consts.data(), w.data(), pi, out.data());
    for (int k = 0; k < c.num_gate_constraints; k++) if (out[k].v != 0) return fail("gate constraint " + std::to_string(k) + " nonzero at row " + std::to_string(i));
  }
  return 0;
}
int orc_max_threads() { return omp_get_max_threads(); }

} // extern "C"
