// End-to-end use of the C++ host mirror (zk-circuits_amd/host/wormhole.hpp) — the analogue of the reference's
// wormhole/example/src/main.rs:13-74: build the wormhole circuit's prover, commit CircuitInputs::test_inputs(), prove, verify, print the
// public inputs; then the same call sequence on the synthetic wormhole-shaped circuit (the bench shape).
//   make -C zk-circuits_amd/csrc ../../examples/prove_example && examples/prove_example [zk]
#include "../zk-circuits_amd/host/wormhole.hpp"
#include "test_inputs.hpp"
#include <chrono>
#include <cstdio>

static double ms(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

int main(int argc, char** argv) {
  using namespace wormhole;
  const bool zk = argc > 1 && std::string(argv[1]) == "zk";
  try {
    // ---- the wormhole circuit with the reference's inputs
    auto cfg = zk ? CircuitConfig::standard_recursion_zk_config() : CircuitConfig::standard_recursion_config();
    auto t0 = std::chrono::steady_clock::now();
    WormholeProver prover = WormholeProver::create(cfg);                       // WormholeProver::new(config)
    auto t1 = std::chrono::steady_clock::now();
    CircuitInputs inputs = test_helpers::circuit_test_inputs();                // CircuitInputs::test_inputs()
    ProofWithPublicInputs proof = prover.commit(inputs).prove();               // prover.commit(&inputs)?.prove()?
    auto t2 = std::chrono::steady_clock::now();
    WormholeProver again(prover.circuit_data);                                 // same resident circuit, second proof
    CircuitInputs other = inputs; other.pub.exit_account.fill(9);
    ProofWithPublicInputs proof2 = again.commit(other).prove();
    auto t3 = std::chrono::steady_clock::now();
    std::printf("wormhole circuit (2^%u rows%s): build+upload %.1f ms, first commit+prove %.1f ms, second %.2f ms, proof %zu bytes\n",
                prover.circuit_data->common().degree_bits, zk ? ", zk" : "", ms(t0, t1), ms(t1, t2), ms(t2, t3), proof.to_bytes().size());
    PublicCircuitInputs pub = PublicCircuitInputs::try_from_slice(proof.public_inputs);
    unsigned long long amount = 0; for (int i = 7; i >= 0; i--) amount = amount << 8 | pub.funding_amount_le[i];
    std::printf("public inputs: funding_amount %llu, nullifier %02x%02x%02x%02x.., root_hash %02x%02x%02x%02x.., exit_account %02x..\n", amount,
                pub.nullifier[0], pub.nullifier[1], pub.nullifier[2], pub.nullifier[3], pub.root_hash[0], pub.root_hash[1], pub.root_hash[2], pub.root_hash[3], pub.exit_account[0]);
    if (amount != 1000000000000ULL || pub.nullifier != inputs.pub.nullifier || pub.root_hash != inputs.pub.root_hash) return 4;   // prover_tests.rs:29-43
    WormholeVerifier verifier(prover.circuit_data);                            // WormholeVerifier::new(config, Some(circuit_data))
    verifier.verify(proof); verifier.verify(proof2);
    ProofWithPublicInputs forged = proof; forged.bytes.back() ^= 1;            // verifier_tests.rs:49-66: a flipped public input must fail
    auto verdicts = verifier.verify_batch({proof, forged, proof2});
    std::printf("verified: %d %d %d\n", (int)verdicts[0], (int)verdicts[1], (int)verdicts[2]);
    if (!verdicts[0] || verdicts[1] || !verdicts[2]) return 3;
    CircuitInputs bad = inputs; bad.pub.root_hash.fill(0);                     // storage_proof_tests.rs:29-34: commit accepts, prove fails
    try { WormholeProver(prover.circuit_data).commit(bad).prove(); return 5; } catch (const std::runtime_error& e) { std::printf("invalid root hash rejected: %.96s...\n", e.what()); }
    // aggregator_tests.rs:60-112 / tree.rs:189-216: three leaves + the dummy proof, aggregated to a root whose nodes verify their children in-circuit
    WormholeProofAggregator agg(TreeAggregationConfig(2, 2));
    agg.push_proof(proof); agg.push_proof(proof2); agg.push_proof(proof);
    auto t6 = std::chrono::steady_clock::now();
    AggregatedProof root = agg.aggregate(/*dummy=*/proof, prover.circuit_data);
    auto t7 = std::chrono::steady_clock::now();
    WormholeVerifier(root.circuit_data).verify(root.proof);                    // root_proof.circuit_data.verify(root_proof.proof)
    std::printf("aggregated 4 leaves -> root (2^%u rows, %zu public inputs) in %.1f ms incl. building 2 node circuits; root verified\n",
                root.circuit_data->common().degree_bits, root.proof.public_inputs.size(), ms(t6, t7));
    if (root.proof.public_inputs.size() != 64 || !std::equal(proof2.public_inputs.begin(), proof2.public_inputs.end(), root.proof.public_inputs.begin() + 16)) return 6;
    try { prover.commit(inputs); return 2; } catch (const std::runtime_error& e) { std::printf("second commit rejected: %s\n", e.what()); }
    // ---- the synthetic wormhole-shaped circuit (bench shape): seeded full witnesses
    WormholeProver synth = WormholeProver::create(CircuitConfig::synthetic(zk ? "B" : "A"));
    auto t4 = std::chrono::steady_clock::now();
    ProofWithPublicInputs sp = synth.commit(synth.circuit_data->test_inputs()).prove();
    auto t5 = std::chrono::steady_clock::now();
    WormholeVerifier(synth.circuit_data).verify(sp);
    std::printf("synthetic shape %s: commit+prove %.2f ms, verified\n", zk ? "B" : "A", ms(t4, t5));
    return 0;
  } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
}
