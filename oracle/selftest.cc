// ORACLE SELF-TEST — TEST INFRASTRUCTURE ONLY. A small native driver over the oracle (and the synthetic generator it is fed from) meant to
// run under AddressSanitizer + UBSan on the host: `make -C oracle asan` builds oracle/_asan/selftest with -fsanitize=address,undefined
// from the same sources as liboracle.so / libqpsynth.so and runs it (tests/test_oracle_prover.py::test_oracle_and_synth_under_sanitizers).
// Coverage: Poseidon vectors, NTT round trip, the verifier over the reference's golden proof (tests/golden/), prove + verify + tamper
// rejection for a non-zk, a zk and a recursion-shaped synthetic circuit.
#include "../include/qpsynth.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
typedef uint64_t u64;
extern "C" {
const char* orc_last_error();
void orc_permute(u64* st);
void orc_ntt(u64* a, size_t n, int inverse);
int orc_verify_bytes(const uint8_t*, size_t, const uint8_t*, size_t, const uint8_t*, size_t);
void* orc_circuit_create(int, int, int, int, int, int, int, int, int, int, int, int, int, int, int, const uint32_t*, int, const uint32_t*, const u64*, const u64*,
                         const uint32_t*, int, const uint32_t*, const u64*, const u64*, const u64*, int);
void orc_circuit_destroy(void*);
long orc_prove(void*, const u64*, const u64*, const u64*, int, u64, int, uint8_t*, size_t);
int orc_verify(void*, const uint8_t*, size_t);
int orc_check_witness(void*, const u64*, const u64*);
}
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest FAILED at %s:%d: %s  [%s]\n", __FILE__, __LINE__, #c, orc_last_error()); exit(1); } } while (0)

static std::vector<uint8_t> slurp(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb"); if (!f) { fprintf(stderr, "cannot open %s\n", p.c_str()); exit(2); }
  std::vector<uint8_t> b; uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + k); fclose(f); return b;
}

static void prove_and_verify(const char* preset) {
  qpsynth_params p; CHECK(qpsynth_preset(preset, &p) == 0);
  qpsynth_circuit* sc = nullptr; CHECK(qpsynth_circuit_create(&p, &sc) == 0);
  const qpgpu_circuit_desc* d = qpsynth_circuit_desc(sc);
  const size_t n = size_t(1) << d->degree_bits, N = n << d->fri_rate_bits;
  std::vector<uint32_t> tags(d->num_gates); std::vector<u64> params(d->num_gates), extra((size_t)d->num_gates * 20, 0);
  for (uint32_t i = 0; i < d->num_gates; i++) {
    tags[i] = d->gates[i].tag; params[i] = d->gates[i].param;
    extra[20 * i] = d->gates[i].param2; extra[20 * i + 1] = d->gates[i].param3; extra[20 * i + 2] = d->gates[i].table_len;
    for (uint32_t k = 0; k < d->gates[i].table_len; k++) extra[20 * i + 3 + k] = d->gates[i].table[k];
  }
  void* oc = orc_circuit_create(d->degree_bits, d->num_wires, d->num_routed, d->num_constants, d->num_challenges, d->quotient_degree_factor,
                                d->num_partial_products, d->num_public_inputs, d->num_gate_constraints, d->zero_knowledge, d->fri_rate_bits,
                                d->fri_cap_height, d->fri_num_queries, d->fri_pow_bits, d->fri_num_layers, d->fri_arity_bits, d->num_gates, tags.data(),
                                params.data(), extra.data(), d->selector_indices, d->num_selector_groups, d->selector_group_ranges, d->k_is,
                                d->constants_sigmas, d->circuit_digest, 2);
  CHECK(oc != nullptr);
  std::vector<u64> wires((size_t)d->num_wires * n), pis(d->num_public_inputs), salts;
  CHECK(qpsynth_witness(sc, 42, wires.data(), pis.data()) == 0);
  CHECK(orc_check_witness(oc, wires.data(), pis.data()) == 0);
  if (d->zero_knowledge) { salts.resize(12 * N); qpsynth_salts(sc, 42, salts.data()); }
  std::vector<uint8_t> proof(1 << 20);
  long len = orc_prove(oc, wires.data(), pis.data(), salts.empty() ? nullptr : salts.data(), 0, 0, 2, proof.data(), proof.size());
  CHECK(len > 0);
  CHECK(orc_verify(oc, proof.data(), (size_t)len) == 0);
  proof[len - 3] ^= 1; CHECK(orc_verify(oc, proof.data(), (size_t)len) != 0); proof[len - 3] ^= 1;      // a public input
  proof[100] ^= 4; CHECK(orc_verify(oc, proof.data(), (size_t)len) != 0);                               // the wires cap
  CHECK(orc_verify(oc, proof.data(), (size_t)len / 2) != 0);                                            // truncated input must be an error, not a read past the end
  orc_circuit_destroy(oc); qpsynth_circuit_destroy(sc);
  printf("  %s: prove + verify + tamper rejection ok (%ld bytes)\n", preset, len);
}

int main(int argc, char** argv) {
  const std::string golden = argc > 1 ? argv[1] : "../tests/golden";
  u64 st[12] = {0}; orc_permute(st); CHECK(st[0] == 0x3c18a9786cb0b359ULL);
  for (int i = 0; i < 12; i++) st[i] = i; orc_permute(st); CHECK(st[0] == 0xd64e1e3efc5b8e9eULL);
  { std::vector<u64> a(256), b; for (size_t i = 0; i < a.size(); i++) a[i] = i * 0x9E3779B97F4A7C15ULL % 0xFFFFFFFF00000001ULL; b = a; orc_ntt(a.data(), a.size(), 0); orc_ntt(a.data(), a.size(), 1); CHECK(a == b); }
  auto cb = slurp(golden + "/common.bin"), vb = slurp(golden + "/verifier.bin"), pb = slurp(golden + "/proof.bin");
  CHECK(orc_verify_bytes(cb.data(), cb.size(), vb.data(), 552, pb.data(), pb.size()) == 0);
  pb[pb.size() - 9] ^= 1; CHECK(orc_verify_bytes(cb.data(), cb.size(), vb.data(), 552, pb.data(), pb.size()) != 0); pb[pb.size() - 9] ^= 1;
  CHECK(orc_verify_bytes(cb.data(), cb.size() - 7, vb.data(), 552, pb.data(), pb.size()) != 0);      // malformed inputs: errors, no out-of-bounds reads
  CHECK(orc_verify_bytes(cb.data(), cb.size(), vb.data(), 552, pb.data(), pb.size() - 1000) != 0);
  printf("  golden proof accepted, tampered / truncated inputs rejected\n");
  prove_and_verify("T"); prove_and_verify("TZ"); prove_and_verify("RT");
  printf("selftest ok\n");
  return 0;
}
