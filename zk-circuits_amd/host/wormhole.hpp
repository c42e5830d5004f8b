// C++ host-side mirror of the reference's drop-in wrappers for the hot path, on top of the qpgpu C-ABI
// (include/qpgpu.h). Header-only; link with -lqpgpu (-lqpsynth for the synthetic circuit source).
//
//   WormholeProver            <- /root/reference/wormhole/prover/src/lib.rs:74-237   (new / commit / prove)
//   WormholeProofAggregator   <- /root/reference/wormhole/aggregator/src/aggregator.rs:13-92 (push_proof / aggregate)
//
// Same call sequence and error behaviour as the reference (errors are std::runtime_error with the reference's text).
// Two circuit sources: the wormhole circuit built by libqpcircuit from the reference's circuit definition (include/qpcircuit.h:
// CircuitConfig::standard_recursion[_zk]_config, committed with the reference's CircuitInputs), and the synthetic wormhole-shaped
// generator (CircuitConfig::synthetic(preset), committed with a full witness: the bench shapes).
// Link with -lqpgpu -lqpcircuit -lqpsynth.
#pragma once
#include "../../include/qpgpu.h"
#include "../../include/qpcircuit.h"
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
  bool zero_knowledge; const char* preset;    // preset == nullptr: the wormhole circuit itself (libqpcircuit)
  static CircuitConfig standard_recursion_config() { return {false, nullptr}; }     // prover_tests.rs:11
  static CircuitConfig standard_recursion_zk_config() { return {true, nullptr}; }   // benches/prover.rs:12
  static CircuitConfig synthetic(const char* preset) { return {std::string(preset) == "B" || std::string(preset) == "TZ", preset}; }   // "A", "B", "T", ...
};

struct ProofWithPublicInputs {
  std::vector<uint8_t> bytes;                 // ProofWithPublicInputs::to_bytes() layout
  std::vector<uint64_t> public_inputs;
  const std::vector<uint8_t>& to_bytes() const { return bytes; }
};

using Bytes32 = std::array<uint8_t, 32>;
struct PublicCircuitInputs {                  // wormhole/circuit/src/inputs.rs:30-40
  std::array<uint8_t, 16> funding_amount_le{}; Bytes32 nullifier{}, root_hash{}, exit_account{};
  static PublicCircuitInputs try_from_slice(const std::vector<uint64_t>& pis) {       // inputs.rs:95-134
    qpw_public_inputs o; PublicCircuitInputs r;
    if (qpw_public_inputs_from_felts(pis.data(), pis.size(), &o) != 0) throw std::runtime_error(qpc_last_error());
    memcpy(r.funding_amount_le.data(), o.funding_amount_le, 16); memcpy(r.nullifier.data(), o.nullifier, 32);
    memcpy(r.root_hash.data(), o.root_hash, 32); memcpy(r.exit_account.data(), o.exit_account, 32);
    return r;
  }
};
struct PrivateCircuitInputs {                 // inputs.rs:43-55; storage_proof = ProcessedStorageProof (nodes + hex-digit child-hash indices)
  Bytes32 secret{}; std::vector<std::vector<uint8_t>> storage_proof; std::vector<uint32_t> storage_proof_indices;
  uint64_t transfer_count = 0; Bytes32 funding_account{}, unspendable_account{};
};
struct CircuitInputs { PublicCircuitInputs pub; PrivateCircuitInputs prv; };          // inputs.rs:23-27
inline Bytes32 nullifier_from_preimage(const Bytes32& secret, uint64_t transfer_count) {            // Nullifier::from_preimage(..).hash
  Bytes32 o; if (qpw_nullifier_from_preimage(secret.data(), 32, transfer_count, o.data()) != 0) throw std::runtime_error(qpc_last_error()); return o;
}
inline Bytes32 unspendable_account_from_secret(const Bytes32& secret) {                             // UnspendableAccount::from_secret(..).account_id
  Bytes32 o; if (qpw_unspendable_account_from_secret(secret.data(), o.data()) != 0) throw std::runtime_error(qpc_last_error()); return o;
}

struct WitnessInputs {                        // a full witness (synthetic circuits; circuits loaded from a package)
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
  WitnessInputs() : salt_key(random_salt_key()) {}
};

class CircuitData {                           // ProverCircuitData: resident on the GPU, shareable between provers
 public:
  CircuitData(const CircuitData&) = delete;
  CircuitData(const CircuitConfig& cfg, int device = 0) {
    if (!cfg.preset) {                                          // WormholeCircuit::new(config).build_prover() (lib.rs:190-202)
      if (qpc_circuit_build(QPC_WORMHOLE, cfg.zero_knowledge, &built_) != 0) throw std::runtime_error(qpc_last_error());
      desc_ = qpc_circuit_desc(built_);
    } else {
      qpsynth_params p;
      if (qpsynth_preset(cfg.preset, &p) != 0) throw std::runtime_error(qpsynth_last_error());
      if (qpsynth_circuit_create(&p, &synth_) != 0) throw std::runtime_error(qpsynth_last_error());
      desc_ = qpsynth_circuit_desc(synth_);
    }
    if (qpgpu_ctx_create(device, &ctx_) != 0) throw std::runtime_error(qpgpu_last_error());
    if (qpgpu_circuit_create(ctx_, desc_, &circuit_) != 0) throw std::runtime_error(qpgpu_last_error());
  }
  // The circuit of one aggregation-tree node (aggregate_chunk, tree.rs:106-128): verifies `num_proofs` proofs of `inner` in-circuit.
  CircuitData(const CircuitData& inner, uint32_t num_proofs, int device = 0) {
    if (qpc_aggregation_circuit_build(&inner.common(), num_proofs, &built_) != 0) throw std::runtime_error(qpc_last_error());
    desc_ = qpc_circuit_desc(built_);
    if (qpgpu_ctx_create(device, &ctx_) != 0) throw std::runtime_error(qpgpu_last_error());
    if (qpgpu_circuit_create(ctx_, desc_, &circuit_) != 0) throw std::runtime_error(qpgpu_last_error());
  }
  ~CircuitData() { if (circuit_) qpgpu_circuit_destroy(circuit_); if (ctx_) qpgpu_ctx_destroy(ctx_); if (synth_) qpsynth_circuit_destroy(synth_); if (built_) qpc_circuit_destroy(built_); }
  CircuitData& operator=(const CircuitData&) = delete;
  std::vector<uint64_t> constants_sigmas_cap() const {
    std::vector<uint64_t> cap((size_t)4 << desc_->fri_cap_height);
    if (qpgpu_circuit_constants_sigmas_cap(circuit_, cap.data()) != 0) throw std::runtime_error(qpgpu_last_error());
    return cap;
  }
  // set_verifier_data_target + set_proof_with_pis_target + generate_partial_witness for a node circuit: the children's proofs (of `inner`) ->
  // the node's witness. A child that does not verify throws.
  WitnessInputs aggregation_witness(const CircuitData& inner, const std::vector<ProofWithPublicInputs>& children) const {
    if (!built_) throw std::runtime_error("not an aggregation node circuit");
    if (children.empty()) throw std::runtime_error("no child proofs");
    const size_t len = children[0].bytes.size(); std::vector<uint8_t> flat(len * children.size());
    for (size_t i = 0; i < children.size(); i++) { if (children[i].bytes.size() != len) throw std::runtime_error("children of one node have one size"); memcpy(&flat[i * len], children[i].bytes.data(), len); }
    const std::vector<uint64_t> cap = inner.constants_sigmas_cap();
    std::vector<uint64_t> targets(qpc_circuit_num_inputs(built_)); size_t n_t = 0;
    if (qpc_aggregation_fill_targets(built_, cap.data(), inner.common().circuit_digest, flat.data(), len, len, (uint32_t)children.size(), targets.data(), targets.size(), &n_t) != 0)
      throw std::runtime_error(qpc_last_error());
    WitnessInputs w; const size_t n = size_t(1) << desc_->degree_bits;
    w.wires.resize((size_t)desc_->num_wires * n); w.public_inputs.resize(desc_->num_public_inputs);
    const auto blinding = WitnessInputs::random_salt_key();
    if (qpc_generate_witness(built_, targets.data(), n_t, blinding.data(), w.wires.data(), w.public_inputs.data()) != 0)
      throw std::runtime_error(std::string("a child proof of this node does not verify: ") + qpc_last_error());
    return w;
  }
  const qpgpu_circuit_desc& common() const { return *desc_; }
  qpgpu_circuit* handle() const { return circuit_; }
  qpgpu_ctx* context() const { return ctx_; }
  size_t proof_size() const { return qpgpu_proof_size(circuit_); }
  // commit() + generate_partial_witness for the built wormhole circuit: the reference's inputs -> full witness. Inputs that do not
  // satisfy the statement throw with the conflict ("... set twice with different values").
  WitnessInputs generate_witness(const CircuitInputs& ci) const {
    if (!built_) throw std::runtime_error("this circuit was not built from the wormhole circuit definition");
    qpw_circuit_inputs c{};
    memcpy(c.funding_amount_le, ci.pub.funding_amount_le.data(), 16); memcpy(c.nullifier, ci.pub.nullifier.data(), 32);
    memcpy(c.root_hash, ci.pub.root_hash.data(), 32); memcpy(c.exit_account, ci.pub.exit_account.data(), 32);
    memcpy(c.secret, ci.prv.secret.data(), 32); c.transfer_count = ci.prv.transfer_count;
    memcpy(c.funding_account, ci.prv.funding_account.data(), 32); memcpy(c.unspendable_account, ci.prv.unspendable_account.data(), 32);
    if (ci.prv.storage_proof.size() != ci.prv.storage_proof_indices.size()) throw std::runtime_error("indices length must be equal to proof length");
    std::vector<const uint8_t*> nodes; std::vector<uint32_t> lens;
    for (const auto& nd : ci.prv.storage_proof) { nodes.push_back(nd.data()); lens.push_back((uint32_t)nd.size()); }
    c.proof_len = (uint32_t)nodes.size(); c.proof_nodes = nodes.data(); c.proof_node_lens = lens.data(); c.proof_indices = ci.prv.storage_proof_indices.data();
    std::vector<uint64_t> targets(qpc_circuit_num_inputs(built_)); size_t len = 0;
    if (qpw_fill_targets(QPC_WORMHOLE, &c, targets.data(), targets.size(), &len) != 0) throw std::runtime_error(qpc_last_error());
    WitnessInputs w; const size_t n = size_t(1) << desc_->degree_bits;                // fresh OS-RNG salt key; blinding rows from a second one
    w.wires.resize((size_t)desc_->num_wires * n); w.public_inputs.resize(desc_->num_public_inputs);
    const auto blinding = WitnessInputs::random_salt_key();
    if (qpc_generate_witness(built_, targets.data(), len, blinding.data(), w.wires.data(), w.public_inputs.data()) != 0) throw std::runtime_error(qpc_last_error());
    return w;
  }
  WitnessInputs test_inputs(uint64_t index = 0) const {     // synthetic circuits: the seeded witness (seed 0x51570000 + index)
    if (!synth_) throw std::runtime_error("this circuit has no synthetic witness source");
    WitnessInputs in; size_t n = size_t(1) << desc_->degree_bits;
    in.wires.resize((size_t)desc_->num_wires * n); in.public_inputs.resize(desc_->num_public_inputs);
    static const uint8_t test_key[32] = {QPSYNTH_TEST_SALT_KEY_BYTES};
    memcpy(in.salt_key.data(), test_key, 32); in.salt_stream = 0x51570000ULL + index;
    qpsynth_witness(synth_, in.salt_stream, in.wires.data(), in.public_inputs.data());
    return in;
  }
 private:
  qpsynth_circuit* synth_ = nullptr; qpc_circuit* built_ = nullptr; const qpgpu_circuit_desc* desc_ = nullptr; qpgpu_ctx* ctx_ = nullptr; qpgpu_circuit* circuit_ = nullptr;
};

class WormholeProver {
 public:
  explicit WormholeProver(std::shared_ptr<CircuitData> cd) : circuit_data(std::move(cd)), targets_(true) {}
  static WormholeProver create(const CircuitConfig& cfg, int device = 0) { return WormholeProver(std::make_shared<CircuitData>(cfg, device)); }  // lib.rs:190-202

  // lib.rs:209-225 — single use: the targets are taken. With the reference's CircuitInputs the witness is generated here; a statement
  // that does not hold is reported by prove(), where upstream's witness generation fails.
  WormholeProver& commit(const CircuitInputs& inputs) {
    if (!targets_) throw std::runtime_error("prover has already commited to inputs");
    targets_ = false;
    try { witness_ = circuit_data->generate_witness(inputs); } catch (const std::runtime_error& e) { witness_error_ = e.what(); }
    return *this;
  }
  WormholeProver& commit(const WitnessInputs& inputs) {
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
    if (!witness_ && !witness_error_.empty()) throw std::runtime_error("Failed to prove: " + witness_error_);
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
  bool targets_; std::optional<WitnessInputs> witness_; std::string witness_error_;
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

struct AggregatedProof { ProofWithPublicInputs proof; std::shared_ptr<CircuitData> circuit_data; };     // tree.rs:24-28

// aggregate_to_tree / aggregate_level / aggregate_chunk (tree.rs:56-143): level by level, consecutive chunks of `tree_branching_factor`
// children, one node proof each. The node circuit of a level is built once (the reference rebuilds it per chunk, tree.rs:112-128).

inline AggregatedProof aggregate_to_tree(std::vector<ProofWithPublicInputs> proofs, std::shared_ptr<CircuitData> inner, const TreeAggregationConfig& config) {
  if (proofs.empty()) throw std::runtime_error("there are no proofs to aggregate");
  std::shared_ptr<CircuitData> node;
  do {
    const size_t b = config.tree_branching_factor;
    std::vector<ProofWithPublicInputs> parents;
    std::shared_ptr<CircuitData> full;                                        // the level's circuit for full chunks
    for (size_t i = 0; i < proofs.size(); i += b) {
      std::vector<ProofWithPublicInputs> chunk(proofs.begin() + i, proofs.begin() + std::min(proofs.size(), i + b));
      if (chunk.size() == b) { if (!full) full = std::make_shared<CircuitData>(*inner, (uint32_t)b); node = full; }
      else node = std::make_shared<CircuitData>(*inner, (uint32_t)chunk.size());   // a ragged last chunk: its own, narrower circuit
      WormholeProver prover(node);
      parents.push_back(prover.commit(node->aggregation_witness(*inner, chunk)).prove());
    }
    proofs = std::move(parents); inner = node;
  } while (proofs.size() > 1);
  return AggregatedProof{std::move(proofs[0]), node};
}

class WormholeProofAggregator {                // aggregator.rs:13-92
 public:
  explicit WormholeProofAggregator(TreeAggregationConfig cfg = {}) : config(cfg), proofs_buffer(std::vector<ProofWithPublicInputs>()) {}
  void push_proof(ProofWithPublicInputs p) {
    if (proofs_buffer) {
      if (proofs_buffer->size() >= config.num_leaf_proofs) throw std::runtime_error("tried to add proof when proof buffer is full");
      proofs_buffer->push_back(std::move(p));
    } else proofs_buffer = std::vector<ProofWithPublicInputs>{std::move(p)};
  }
  // The padded leaf set the aggregator step consumes (util.rs:11-29 padding rule).
  std::vector<ProofWithPublicInputs> padded_leaves(const ProofWithPublicInputs& dummy) {
    if (!proofs_buffer) throw std::runtime_error("there are no proofs to aggregate");
    auto proofs = std::move(*proofs_buffer); proofs_buffer.reset();
    if (proofs.size() > config.num_leaf_proofs) throw std::runtime_error("proofs to aggregate was more than the maximum allowed");
    while (proofs.size() < config.num_leaf_proofs) proofs.push_back(dummy);
    return proofs;
  }
  // aggregator.rs:74-92: pad with the dummy proof and aggregate to the root proof of the tree.
  AggregatedProof aggregate(const ProofWithPublicInputs& dummy, std::shared_ptr<CircuitData> leaf_circuit_data) {
    return aggregate_to_tree(padded_leaves(dummy), std::move(leaf_circuit_data), config);
  }
  TreeAggregationConfig config;
  std::optional<std::vector<ProofWithPublicInputs>> proofs_buffer;
};

} // namespace wormhole
