// Circuit artefacts on disk (reference: wormhole/circuit-builder/src/lib.rs:36-63 writes common.bin, verifier.bin and
// prover.bin, wormhole/prover/src/lib.rs:105-187 and wormhole/verifier/src/lib.rs:98-148 load them). Host code only.
//   * writers for the two formats whose layout the reference's own files pin (CommonCircuitData, VerifierOnlyCircuitData; SURVEY.md
//     App. B.1-B.2): the inverse of qpgpu_parse_common / qpgpu_parse_verifier_only, byte-stable on wormhole/bench-data/{common,verifier}.bin;
//   * the "circuit package": one blob with everything qpgpu_circuit_create needs (the common data in plonky2's own byte format, the
//     circuit digest, the constants/sigmas values). It stands where prover.bin stands in the reference: ProverOnlyCircuitData's layout is
//     pinned by nothing on disk, and what this backend needs from it is exactly these values.
#include "../../include/qpgpu.h"
#include "gl.h"
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace qp { extern thread_local std::string g_last_error; int set_err(int code, const std::string& m); }
using namespace qp;

struct qpgpu_package {
  qpgpu_circuit_desc d{};
  std::vector<qpgpu_gate> gates; std::vector<uint32_t> sel, groups; std::vector<uint64_t> kis, cs;
};

namespace {
struct Writer {
  std::vector<uint8_t> b;
  void u8(uint8_t v) { b.push_back(v); }
  void u32(uint32_t v) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(v >> (8 * i))); }
  void u64(uint64_t v) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(v >> (8 * i))); }
};
template <class F> int guard(F&& f) {
  try { return f(); }
  catch (const std::bad_alloc&) { return set_err(QPGPU_ERR_OOM, "host memory allocation failed"); }
  catch (...) { return set_err(QPGPU_ERR_HIP, "internal error"); }
}
int emit(const std::vector<uint8_t>& b, uint8_t* out, size_t cap, size_t* len) {
  if (len) *len = b.size();
  if (!out) return 0;                                            // size query
  if (cap < b.size()) return set_err(QPGPU_ERR_BUFFER, "output buffer too small: need " + std::to_string(b.size()));
  memcpy(out, b.data(), b.size()); return 0;
}
// FriConfig as both reference configs write it: rate_bits, cap_height, num_query_rounds (u64), proof_of_work_bits (u32),
// reduction strategy ConstantArityBits(4, 5) (tag 1) — standard_recursion(_zk)_config's values.
void fri_config(Writer& w, const qpgpu_circuit_desc& d) {
  w.u64(d.fri_rate_bits); w.u64(d.fri_cap_height); w.u64(d.fri_num_queries); w.u32(d.fri_pow_bits); w.u8(1); w.u64(4); w.u64(5);
}
int common_bytes(const qpgpu_circuit_desc& d, Writer& w) {
  if (!d.gates || !d.selector_indices || !d.selector_group_ranges || !d.k_is) return set_err(QPGPU_ERR_ARG, "null array in the circuit descriptor");
  if (d.num_selector_groups < 1 || d.num_constants < d.num_selector_groups) return set_err(QPGPU_ERR_ARG, "bad constants/selectors");
  // CircuitConfig: num_wires, num_routed_wires, num_constants (the config's 2, not the circuit's column count), security_bits 100,
  // num_challenges, max_quotient_degree_factor, use_base_arithmetic_gate, zero_knowledge
  w.u64(d.num_wires); w.u64(d.num_routed); w.u64(2); w.u64(100); w.u64(d.num_challenges); w.u64(d.quotient_degree_factor);
  w.u8(1); w.u8(d.zero_knowledge ? 1 : 0);
  fri_config(w, d);
  fri_config(w, d);                                               // FriParams repeats the config ...
  w.u64(d.fri_num_layers); for (uint32_t i = 0; i < d.fri_num_layers; i++) w.u64(d.fri_arity_bits[i]);   // ... then the reduction arities,
  w.u64(d.degree_bits); w.u8(d.zero_knowledge ? 1 : 0);           // degree_bits and hiding
  w.u64(d.num_gates); for (uint32_t i = 0; i < d.num_gates; i++) w.u64(d.selector_indices[i]);
  w.u64(d.num_selector_groups); for (uint32_t i = 0; i < d.num_selector_groups; i++) { w.u64(d.selector_group_ranges[2 * i]); w.u64(d.selector_group_ranges[2 * i + 1]); }
  w.u64(d.quotient_degree_factor); w.u64(d.num_gate_constraints); w.u64(d.num_constants); w.u64(d.num_public_inputs);
  w.u64(d.num_routed); for (uint32_t i = 0; i < d.num_routed; i++) w.u64(d.k_is[i]);
  w.u64(d.num_partial_products); w.u64(0); w.u64(0); w.u64(0);    // no lookup polynomials / selectors / tables
  w.u64(d.num_gates);
  for (uint32_t i = 0; i < d.num_gates; i++) {
    const qpgpu_gate& g = d.gates[i];
    w.u32(g.tag);
    switch (g.tag) {
      case QPGPU_GATE_NOOP: case QPGPU_GATE_PUBLIC_INPUT: case QPGPU_GATE_POSEIDON: case QPGPU_GATE_POSEIDON_MDS: break;
      case QPGPU_GATE_CONSTANT: case QPGPU_GATE_BASESUM2: case QPGPU_GATE_ARITHMETIC: case QPGPU_GATE_ARITHMETIC_EXTENSION: case QPGPU_GATE_MUL_EXTENSION:
      case QPGPU_GATE_REDUCING: case QPGPU_GATE_REDUCING_EXTENSION: case QPGPU_GATE_EXPONENTIATION: w.u64(g.param); break;
      case QPGPU_GATE_RANDOM_ACCESS: w.u64(g.param); w.u64(g.param2); w.u64(g.param3); break;
      case QPGPU_GATE_COSET_INTERPOLATION:
        if (!g.table && g.table_len) return set_err(QPGPU_ERR_ARG, "CosetInterpolation gate without weights");
        w.u64(g.param); w.u64(g.param2); w.u64(g.table_len); for (uint32_t k = 0; k < g.table_len; k++) w.u64(g.table[k]); break;
      default: return set_err(QPGPU_ERR_ARG, "gate tag " + std::to_string(g.tag) + " is not supported (lookup gates)");
    }
  }
  return 0;
}
uint64_t fnv1a(const uint8_t* p, size_t n) { uint64_t h = 0xcbf29ce484222325ULL; for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ULL; } return h; }
const char MAGIC[8] = {'Q', 'P', 'G', 'P', 'U', 'P', 'K', '1'};
} // namespace

extern "C" {

int qpgpu_write_common(const qpgpu_circuit_desc* desc, uint8_t* out, size_t cap, size_t* len) { return guard([&]() -> int {
  if (!desc) return set_err(QPGPU_ERR_ARG, "null argument");
  Writer w; int rc = common_bytes(*desc, w); if (rc) return rc;
  return emit(w.b, out, cap, len);
}); }

int qpgpu_write_verifier_only(const uint64_t* constants_sigmas_cap, uint32_t cap_height, const uint64_t circuit_digest[4], uint8_t* out, size_t cap, size_t* len) { return guard([&]() -> int {
  if (!constants_sigmas_cap || !circuit_digest || cap_height > 20) return set_err(QPGPU_ERR_ARG, "bad argument");
  Writer w; w.u64(cap_height);
  for (size_t i = 0; i < (size_t(4) << cap_height); i++) w.u64(constants_sigmas_cap[i]);
  for (int i = 0; i < 4; i++) w.u64(circuit_digest[i]);
  return emit(w.b, out, cap, len);
}); }

int qpgpu_package_write(const qpgpu_circuit_desc* desc, uint8_t* out, size_t cap, size_t* len) { return guard([&]() -> int {
  if (!desc || !desc->constants_sigmas) return set_err(QPGPU_ERR_ARG, "null argument / descriptor without constants_sigmas");
  const qpgpu_circuit_desc& d = *desc;
  if (d.degree_bits > 28) return set_err(QPGPU_ERR_ARG, "degree_bits out of range");
  Writer cm; int rc = common_bytes(d, cm); if (rc) return rc;
  const size_t n = size_t(1) << d.degree_bits, S = (size_t)d.num_constants + d.num_routed;
  const size_t total = 8 + 8 + cm.b.size() + 32 + 16 + S * n * 8 + 8;
  if (len) *len = total;
  if (!out) return 0;
  if (cap < total) return set_err(QPGPU_ERR_BUFFER, "output buffer too small: need " + std::to_string(total));
  uint8_t* p = out;
  memcpy(p, MAGIC, 8); p += 8;
  { uint64_t v = cm.b.size(); memcpy(p, &v, 8); p += 8; }
  memcpy(p, cm.b.data(), cm.b.size()); p += cm.b.size();
  memcpy(p, d.circuit_digest, 32); p += 32;
  { uint64_t v = S; memcpy(p, &v, 8); p += 8; v = n; memcpy(p, &v, 8); p += 8; }
  memcpy(p, d.constants_sigmas, S * n * 8); p += S * n * 8;
  { uint64_t h = fnv1a(out, (size_t)(p - out)); memcpy(p, &h, 8); }
  return 0;
}); }

int qpgpu_package_read(const uint8_t* bytes, size_t len, qpgpu_package** out) { return guard([&]() -> int {
  if (!bytes || !out) return set_err(QPGPU_ERR_ARG, "null argument");
  if (len < 8 + 8 + 32 + 16 + 8 || memcmp(bytes, MAGIC, 8) != 0) return set_err(QPGPU_ERR_ARG, "not a circuit package (bad magic or too short)");
  { uint64_t want; memcpy(&want, bytes + len - 8, 8); if (fnv1a(bytes, len - 8) != want) return set_err(QPGPU_ERR_ARG, "circuit package checksum mismatch"); }
  uint64_t clen; memcpy(&clen, bytes + 8, 8);
  if (clen > len - (8 + 8 + 32 + 16 + 8)) return set_err(QPGPU_ERR_ARG, "circuit package truncated");
  std::unique_ptr<qpgpu_package> pk(new qpgpu_package());
  pk->gates.resize(16); pk->sel.resize(16); pk->groups.resize(32); pk->kis.resize(4096 + 16);
  int rc = qpgpu_parse_common(bytes + 16, (size_t)clen, &pk->d, pk->gates.data(), pk->gates.size(), pk->sel.data(), pk->groups.data(), pk->groups.size() / 2,
                              pk->kis.data(), pk->kis.size());
  if (rc) return rc;
  const uint8_t* p = bytes + 16 + clen;
  memcpy(pk->d.circuit_digest, p, 32); p += 32;
  uint64_t S, n; memcpy(&S, p, 8); memcpy(&n, p + 8, 8); p += 16;
  if (pk->d.degree_bits > 28 || n != (uint64_t(1) << pk->d.degree_bits) || S != (uint64_t)pk->d.num_constants + pk->d.num_routed)
    return set_err(QPGPU_ERR_ARG, "circuit package: constants/sigmas shape disagrees with the common data");
  if ((size_t)(bytes + len - 8 - p) != S * n * 8) return set_err(QPGPU_ERR_ARG, "circuit package: wrong length");
  pk->cs.resize(S * n); memcpy(pk->cs.data(), p, S * n * 8);
  for (uint64_t v : pk->cs) if (v >= GL_P) return set_err(QPGPU_ERR_ARG, "circuit package: non-canonical constants/sigmas value");
  for (int i = 0; i < 4; i++) if (pk->d.circuit_digest[i] >= GL_P) return set_err(QPGPU_ERR_ARG, "circuit package: non-canonical digest");
  pk->d.constants_sigmas = pk->cs.data();
  *out = pk.release(); return 0;
}); }

const qpgpu_circuit_desc* qpgpu_package_desc(const qpgpu_package* p) { return p ? &p->d : nullptr; }
void qpgpu_package_free(qpgpu_package* p) { delete p; }

} // extern "C"
