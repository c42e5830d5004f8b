// C++ host-side mirror of the reference's drop-in wrappers for the hot path, on top of the qpgpu C-ABI
// (include/qpgpu.h). Header-only; link with -lqpgpu (-lqpsynth for the synthetic circuit source).
//
//   WormholeProver            <- /root/reference/wormhole/prover/src/lib.rs:74-237   (new / commit / prove)
//   WormholeProofAggregator   <- /root/reference/wormhole/aggregator/src/aggregator.rs:13-92 (push_proof / aggregate)
//
// Same call sequence and error behaviour as the reference (errors are std::runtime_error with the reference's text).
// The circuit comes from the synthetic wormhole-shaped generator because plonky2's CircuitBuilder needs the Rust toolchain
// (SURVEY.md §8d); a real deployment passes the descriptor built from ProverCircuitData instead.
#pragma once
#include "../../include/qpgpu.h"
#include "../../include/qpsynth.h"
#include <cstdint>
#include <array>
#include <cstring>
#include <memory>
#include <random>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

namespace wormhole {

struct CircuitConfig {
  bool zero_knowledge; const char* preset;
  static CircuitConfig standard_recursion_config() { return {false, "A"}; }       // prover_tests.rs:11
  static CircuitConfig standard_recursion_zk_config() { return {true, "B"}; }     // benches/prover.rs:12
};

struct ProofWithPublicInputs {
  std::vector<uint8_t> bytes;                 // ProofWithPublicInputs::to_bytes() layout
  std::vector<uint64_t> public_inputs;
  const std::vector<uint8_t>& to_bytes() const { return bytes; }
};

struct CircuitInputs {                        // stand-in for wormhole_circuit::inputs::CircuitInputs: a full witness
  std::vector<uint64_t> wires;                // [num_wires][n], column-major
  std::vector<uint64_t> public_inputs;        // [16]
  // zk salts = ChaCha8(salt_key, salt_stream) on the device (include/qpgpu.h). test_inputs() uses the public test key with
  // stream = seed so the bytes are reproducible; a real witness gets 32 bytes from the OS RNG (what F::rand_vec does upstream).
  std::array<uint8_t, 32> salt_key{}; uint64_t salt_stream = 0;
  static std::array<uint8_t, 32> random_salt_key() {
    std::array<uint8_t, 32> k{}; std::random_device rd;
    for (size_t i = 0; i < k.size(); i += 4) { uint32_t v = rd(); memcpy(&k[i], &v, 4); }
    return k;
  }
  CircuitInputs() : salt_key(random_salt_key()) {}
};

class CircuitData {                           // ProverCircuitData: resident on the GPU, shareable between provers
 public:
  CircuitData(const CircuitConfig& cfg, int device = 0) {
    qpsynth_params p;
    if (qpsynth_preset(cfg.preset, &p) != 0) throw std::runtime_error(qpsynth_last_error());
    if (qpsynth_circuit_create(&p, &synth_) != 0) throw std::runtime_error(qpsynth_last_error());
    desc_ = qpsynth_circuit_desc(synth_);
    if (qpgpu_ctx_create(device, &ctx_) != 0) throw std::runtime_error(qpgpu_last_error());
    if (qpgpu_circuit_create(ctx_, desc_, &circuit_) != 0) throw std::runtime_error(qpgpu_last_error());
  }
  ~CircuitData() { if (circuit_) qpgpu_circuit_destroy(circuit_); if (ctx_) qpgpu_ctx_destroy(ctx_); if (synth_) qpsynth_circuit_destroy(synth_); }
  CircuitData(const CircuitData&) = delete; CircuitData& operator=(const CircuitData&) = delete;
  const qpgpu_circuit_desc& common() const { return *desc_; }
  qpgpu_circuit* handle() const { return circuit_; }
  qpgpu_ctx* context() const { return ctx_; }
  size_t proof_size() const { return qpgpu_proof_size(circuit_); }
  CircuitInputs test_inputs(uint64_t index = 0) const {     // CircuitInputs::test_inputs() analogue (seed 0x51570000 + index)
    CircuitInputs in; size_t n = size_t(1) << desc_->degree_bits;
    in.wires.resize((size_t)desc_->num_wires * n); in.public_inputs.resize(desc_->num_public_inputs);
    static const uint8_t test_key[32] = {QPSYNTH_TEST_SALT_KEY_BYTES};
    memcpy(in.salt_key.data(), test_key, 32); in.salt_stream = 0x51570000ULL + index;
    qpsynth_witness(synth_, in.salt_stream, in.wires.data(), in.public_inputs.data());
    return in;
  }
 private:
  qpsynth_circuit* synth_ = nullptr; const qpgpu_circuit_desc* desc_ = nullptr; qpgpu_ctx* ctx_ = nullptr; qpgpu_circuit* circuit_ = nullptr;
};

class WormholeProver {
 public:
  explicit WormholeProver(std::shared_ptr<CircuitData> cd) : circuit_data(std::move(cd)), targets_(true) {}
  static WormholeProver create(const CircuitConfig& cfg, int device = 0) { return WormholeProver(std::make_shared<CircuitData>(cfg, device)); }  // lib.rs:190-202

  // lib.rs:209-225 — single use: the targets are taken.
  WormholeProver& commit(const CircuitInputs& inputs) {
    if (!targets_) throw std::runtime_error("prover has already commited to inputs");
    targets_ = false;
    const auto& d = circuit_data->common();
    if (inputs.wires.size() != (size_t)d.num_wires << d.degree_bits || inputs.public_inputs.size() != d.num_public_inputs)
      throw std::runtime_error("witness shape mismatch");
    witness_ = inputs;
    return *this;
  }
  // lib.rs:233-237 — consumes the witness.
  ProofWithPublicInputs prove() {
    if (!witness_) throw std::runtime_error("Failed to prove: prover has not commited to any inputs");
    ProofWithPublicInputs out; out.bytes.resize(circuit_data->proof_size());
    qpgpu_prove_opts o{}; o.salt_key = witness_->salt_key.data(); o.salt_stream = witness_->salt_stream;
    size_t len = 0;
    int rc = qpgpu_prove(circuit_data->handle(), witness_->wires.data(), witness_->public_inputs.data(), witness_->public_inputs.size(), &o,
                         out.bytes.data(), out.bytes.size(), &len);
    if (rc != 0) throw std::runtime_error(std::string("Failed to prove: ") + qpgpu_last_error());
    out.bytes.resize(len); out.public_inputs = witness_->public_inputs; witness_.reset();
    return out;
  }
  std::shared_ptr<CircuitData> circuit_data;
 private:
  bool targets_; std::optional<CircuitInputs> witness_;
};

// wormhole/verifier/src/lib.rs:81-159 — VerifierCircuitData::verify through qpgpu_verify_batch (device Merkle openings, host algebra).
class WormholeVerifier {
 public:
  explicit WormholeVerifier(std::shared_ptr<CircuitData> cd) : circuit_data(std::move(cd)) {     // WormholeVerifier::new(config, Some(data))
    const auto& d = circuit_data->common();
    std::vector<uint64_t> cap((size_t)4 << d.fri_cap_height);
    if (qpgpu_circuit_constants_sigmas_cap(circuit_data->handle(), cap.data()) != 0) throw std::runtime_error(qpgpu_last_error());
    if (qpgpu_verifier_create(circuit_data->context(), &d, cap.data(), &v_) != 0) throw std::runtime_error(qpgpu_last_error());
  }
  ~WormholeVerifier() { if (v_) qpgpu_verifier_destroy(v_); }
  WormholeVerifier(const WormholeVerifier&) = delete; WormholeVerifier& operator=(const WormholeVerifier&) = delete;
  void verify(const ProofWithPublicInputs& proof) const {                                          // lib.rs:155-159
    int32_t res = 0;
    if (qpgpu_verify_batch(v_, proof.bytes.data(), proof.bytes.size(), proof.bytes.size(), 1, &res) != 0)
      throw std::runtime_error(std::string("proof verification failed: ") + qpgpu_last_error());
  }
  std::vector<bool> verify_batch(const std::vector<ProofWithPublicInputs>& proofs) const {         // one flag per proof, no throw for a rejected one
    if (proofs.empty()) return {};
    const size_t len = proofs[0].bytes.size(); std::vector<uint8_t> flat(len * proofs.size()); std::vector<int32_t> res(proofs.size());
    for (size_t i = 0; i < proofs.size(); i++) { if (proofs[i].bytes.size() != len) throw std::runtime_error("one batch = proofs of one size"); memcpy(&flat[i * len], proofs[i].bytes.data(), len); }
    const int rc = qpgpu_verify_batch(v_, flat.data(), len, len, proofs.size(), res.data());
    if (rc != 0 && rc != QPGPU_ERR_VERIFY) throw std::runtime_error(qpgpu_last_error());
    std::vector<bool> ok(proofs.size()); for (size_t i = 0; i < ok.size(); i++) ok[i] = res[i] == 0; return ok;
  }
  std::shared_ptr<CircuitData> circuit_data;
 private:
  qpgpu_verifier* v_ = nullptr;
};

struct TreeAggregationConfig {                 // circuits/tree.rs:32-55
  size_t tree_branching_factor = 2; uint32_t tree_depth = 3; size_t num_leaf_proofs = 8;
  TreeAggregationConfig() = default;
  TreeAggregationConfig(size_t b, uint32_t d) : tree_branching_factor(b), tree_depth(d), num_leaf_proofs(1) { for (uint32_t i = 0; i < d; i++) num_leaf_proofs *= b; }
};

class WormholeProofAggregator {                // aggregator.rs:13-92 (buffering + padding; the recursion circuit is "next")
 public:
  explicit WormholeProofAggregator(TreeAggregationConfig cfg = {}) : config(cfg), proofs_buffer(std::vector<ProofWithPublicInputs>()) {}
  void push_proof(ProofWithPublicInputs p) {
    if (proofs_buffer) {
      if (proofs_buffer->size() >= config.num_leaf_proofs) throw std::runtime_error("tried to add proof when proof buffer is full");
      proofs_buffer->push_back(std::move(p));
    } else proofs_buffer = std::vector<ProofWithPublicInputs>{std::move(p)};
  }
  // Returns the padded leaf set the aggregator step consumes (util.rs:11-29 padding rule).
  std::vector<ProofWithPublicInputs> aggregate(const ProofWithPublicInputs& dummy) {
    if (!proofs_buffer) throw std::runtime_error("there are no proofs to aggregate");
    auto proofs = std::move(*proofs_buffer); proofs_buffer.reset();
    if (proofs.size() > config.num_leaf_proofs) throw std::runtime_error("proofs to aggregate was more than the maximum allowed");
    while (proofs.size() < config.num_leaf_proofs) proofs.push_back(dummy);
    return proofs;
  }
  TreeAggregationConfig config;
  std::optional<std::vector<ProofWithPublicInputs>> proofs_buffer;
};

} // namespace wormhole
