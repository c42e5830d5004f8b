// End-to-end use of the C++ host mirror (zk-circuits_amd/host/wormhole.hpp) — the analogue of the reference's
// wormhole/example/src/main.rs:13-74: build the prover, commit inputs, prove, print the proof size and public inputs.
//   g++ -std=c++17 examples/prove_example.cc -Lzk-circuits_amd -lqpgpu -lqpsynth -Wl,-rpath,$PWD/zk-circuits_amd -o examples/prove_example
#include "../zk-circuits_amd/host/wormhole.hpp"
#include <chrono>
#include <cstdio>

int main(int argc, char** argv) {
  using namespace wormhole;
  const bool zk = argc > 1 && std::string(argv[1]) == "zk";
  try {
    auto cfg = zk ? CircuitConfig::standard_recursion_zk_config() : CircuitConfig::standard_recursion_config();
    auto t0 = std::chrono::steady_clock::now();
    WormholeProver prover = WormholeProver::create(cfg);                       // WormholeProver::new(config)
    auto t1 = std::chrono::steady_clock::now();
    CircuitInputs inputs = prover.circuit_data->test_inputs();                 // CircuitInputs::test_inputs()
    ProofWithPublicInputs proof = prover.commit(inputs).prove();               // prover.commit(&inputs)?.prove()?
    auto t2 = std::chrono::steady_clock::now();
    WormholeProver again(prover.circuit_data);                                 // same resident circuit, second proof
    ProofWithPublicInputs proof2 = again.commit(prover.circuit_data->test_inputs(1)).prove();
    auto t3 = std::chrono::steady_clock::now();
    std::printf("circuit build+upload %.1f ms, first prove %.1f ms, second prove %.2f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(),
                std::chrono::duration<double, std::milli>(t3 - t2).count());
    std::printf("proof bytes %zu (second %zu), public inputs:", proof.to_bytes().size(), proof2.to_bytes().size());
    for (uint64_t x : proof.public_inputs) std::printf(" %016llx", (unsigned long long)x);
    std::printf("\n");
    WormholeVerifier verifier(prover.circuit_data);                            // WormholeVerifier::new(config, Some(circuit_data))
    verifier.verify(proof); verifier.verify(proof2);
    ProofWithPublicInputs forged = proof; forged.bytes.back() ^= 1;            // verifier_tests.rs:49-66: a flipped public input must fail
    auto verdicts = verifier.verify_batch({proof, forged, proof2});
    std::printf("verified: %d %d %d\n", (int)verdicts[0], (int)verdicts[1], (int)verdicts[2]);
    if (!verdicts[0] || verdicts[1] || !verdicts[2]) return 3;
    WormholeProofAggregator agg(TreeAggregationConfig(2, 1));
    agg.push_proof(proof); agg.push_proof(proof2);
    auto leaves = agg.aggregate(proof);
    std::printf("aggregator leaf set: %zu proofs\n", leaves.size());
    try { prover.commit(inputs); return 2; } catch (const std::runtime_error& e) { std::printf("second commit rejected: %s\n", e.what()); }
    return 0;
  } catch (const std::exception& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
}
