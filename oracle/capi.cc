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

} // extern "C"
