#include "../../include/qpcircuit.h"
#include <cstdio>
#include <cstring>
#include <vector>
static std::vector<uint64_t> load(const char* p) { FILE* f = fopen(p, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<uint64_t> v(n / 8); if (fread(v.data(), 8, v.size(), f) != v.size()) return {}; fclose(f); return v; }
int main() {
  uint8_t key[32]; for (int i = 0; i < 32; i++) key[i] = (uint8_t)i;
  { qpc_circuit* c = nullptr; if (qpc_circuit_build(QPC_WORMHOLE, 1, &c)) { puts(qpc_last_error()); return 1; }
    auto in = load("/tmp/tsan/wormhole_inputs.bin"); const qpgpu_circuit_desc* d = qpc_circuit_desc(c);
    std::vector<uint64_t> w((size_t)d->num_wires << d->degree_bits), w1(w.size()), p(d->num_public_inputs);
    if (qpc_generate_witness_mt(c, in.data(), in.size(), key, w1.data(), p.data(), 1)) { puts(qpc_last_error()); return 2; }
    for (int rep = 0; rep < 3; rep++) { if (qpc_generate_witness_mt(c, in.data(), in.size(), key, w.data(), p.data(), 8)) { puts(qpc_last_error()); return 3; } if (memcmp(w.data(), w1.data(), w.size() * 8)) { puts("wormhole witness differs"); return 4; } }
    printf("wormhole zk: parallel %d, 8-thread witness == 1-thread witness\n", qpc_circuit_parallel_segments(c)); qpc_circuit_destroy(c); }
  { qpc_circuit* inner = nullptr; qpc_circuit_build(QPC_VOTE, 0, &inner);
    qpc_circuit* c = nullptr; if (qpc_aggregation_circuit_build(qpc_circuit_desc(inner), 2, &c)) { puts(qpc_last_error()); return 5; }
    auto in = load("/tmp/tsan/node_inputs.bin"); auto ref = load("/tmp/tsan/node_wires.bin"); const qpgpu_circuit_desc* d = qpc_circuit_desc(c);
    std::vector<uint64_t> w((size_t)d->num_wires << d->degree_bits), p(d->num_public_inputs);
    for (int rep = 0; rep < 3; rep++) { if (qpc_generate_witness_mt(c, in.data(), in.size(), nullptr, w.data(), p.data(), 8)) { puts(qpc_last_error()); return 6; } if (w.size() != ref.size() || memcmp(w.data(), ref.data(), w.size() * 8)) { puts("node witness differs"); return 7; } }
    in[5000] ^= 1;                                   // a forged child: every thread may hit the conflict
    if (qpc_generate_witness_mt(c, in.data(), in.size(), nullptr, w.data(), p.data(), 8) == 0) { puts("forged child accepted"); return 8; }
    printf("node: parallel %d, 8-thread witness == python 1-thread witness; forged child: %.60s\n", qpc_circuit_parallel_segments(c), qpc_last_error());
    qpc_circuit_destroy(c); qpc_circuit_destroy(inner); }
  return 0;
}
