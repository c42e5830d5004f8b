// The reference's circuits on this repo's builder, their input encoding, and the C-ABI of include/qpcircuit.h. Host-only C++.
// Each fragment follows the reference function it names line by line in meaning (same targets in the same allocation order, same
// checks, same public inputs); gate placement is the builder's.
#include "../../include/qpcircuit.h"
#include "builder.h"
#include "circuit_object.h"
#include <cstring>
#include <memory>
#include <thread>

using namespace qpc;
using qp::hash_no_pad;

namespace {
thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

// ---- zk_circuits_common::utils (common/src/utils.rs)
void injective_string_to_felt(const char* s8, u64 out[2]) {                  // utils.rs:149-164: two little-endian u32 limbs of an 8-byte string
  for (int k = 0; k < 2; k++) { u64 v = 0; for (int i = 0; i < 4; i++) v |= (u64)(uint8_t)s8[4 * k + i] << (8 * i); out[k] = v; }
}
void injective_bytes_to_felts(const uint8_t* b, size_t len, std::vector<u64>& out) {   // utils.rs:167-180: 4 bytes per felt, little-endian, last chunk zero-padded
  for (size_t off = 0; off < len; off += 4) { u64 v = 0; for (size_t i = 0; i < 4 && off + i < len; i++) v |= (u64)b[off + i] << (8 * i); out.push_back(v); }
}
bool digest_bytes_to_felts(const uint8_t b[32], u64 out[4], std::string& why) {          // utils.rs:34-52 (BytesDigest::try_from) + 196-208
  for (int k = 0; k < 4; k++) {
    u64 v = 0; for (int i = 0; i < 8; i++) v |= (u64)b[8 * k + i] << (8 * i);
    if (v >= P) { why = "ChunkOutOfFieldRange { chunk_index: " + std::to_string(k) + ", value: " + std::to_string(v) + " }"; return false; }
    out[k] = v;
  }
  return true;
}
void digest_felts_to_bytes(const u64 f[4], uint8_t out[32]) { for (int k = 0; k < 4; k++) for (int i = 0; i < 8; i++) out[8 * k + i] = (uint8_t)(f[k] >> (8 * i)); }   // utils.rs:210-222
void u64_to_felts(u64 v, u64 out[2]) { out[0] = v >> 32; out[1] = v & 0xFFFFFFFFULL; }                         // utils.rs:126-131: high limb first
void u128_to_felts(const uint8_t le[16], u64 out[4]) {                                                         // utils.rs:101-112: four 32-bit limbs, most significant first
  for (int k = 0; k < 4; k++) { u64 v = 0; for (int i = 0; i < 4; i++) v |= (u64)le[4 * (3 - k) + i] << (8 * i); out[k] = v; }
}

// ---- targets, in the reference's allocation order
struct NullifierTargets { Hash4 hash; std::vector<Target> secret; Target transfer_count[2]; };
struct UnspendableTargets { Hash4 account_id; std::vector<Target> secret; };
struct LeafTargets { Target transfer_count[2]; Hash4 funding_account, to_account; Target funding_amount[4]; };
struct StorageProofTargets { std::vector<std::vector<Target>> proof_data; std::vector<Target> indices; Hash4 root_hash; Target proof_len; LeafTargets leaf; };
struct ExitAccountTargets { Hash4 address; };

NullifierTargets nullifier_targets(Builder& b) {                 // nullifier.rs:202-210
  NullifierTargets t; t.hash = b.add_public_input_hash(); t.secret = b.add_inputs(8); for (auto& x : t.transfer_count) x = b.add_input(); return t;
}
UnspendableTargets unspendable_targets(Builder& b) {             // unspendable_account.rs:168-178
  UnspendableTargets t; t.account_id = b.add_input_hash(); t.secret = b.add_inputs(8); return t;
}
StorageProofTargets storage_proof_targets(Builder& b) {          // storage_proof/mod.rs:36-57, leaf.rs:24-37
  StorageProofTargets t;
  for (int i = 0; i < QPW_MAX_PROOF_LEN; i++) t.proof_data.push_back(b.add_inputs(QPW_PROOF_NODE_MAX_FELTS));
  t.indices = b.add_inputs(QPW_MAX_PROOF_LEN);
  t.root_hash = b.add_public_input_hash();
  t.proof_len = b.add_input();
  for (auto& x : t.leaf.transfer_count) x = b.add_input();
  t.leaf.funding_account = b.add_input_hash(); t.leaf.to_account = b.add_input_hash();
  for (auto& x : t.leaf.funding_amount) x = b.add_public_input();
  return t;
}
ExitAccountTargets exit_account_targets(Builder& b) { ExitAccountTargets t; t.address = b.add_public_input_hash(); return t; }   // substrate_account.rs:76-82

std::vector<Target> hash_vec(const Hash4& h) { return std::vector<Target>(h.e, h.e + 4); }

// ---- common/src/gadgets.rs
BoolT xor_(Builder& b, BoolT x, BoolT y) {                        // gadgets.rs:55-65: a + b - 2ab
  Target ab = b.mul(x.t, y.t), two_ab = b.mul_const(2, ab), sum = b.add(x.t, y.t);
  return BoolT{b.sub(sum, two_ab)};
}
BoolT is_const_less_than(Builder& b, size_t left, Target right, int n_log) {   // gadgets.rs:14-42: left < right over n_log bits, most significant first
  std::vector<BoolT> right_bits = b.split_le(right, n_log);
  BoolT lt = b.constant_bool(false), eq = b.constant_bool(true);
  for (int i = n_log - 1; i >= 0; i--) {
    BoolT a = b.constant_bool((left >> i) & 1), rb = right_bits[i];
    BoolT not_a = b.not_(a), not_a_and_b = b.and_(not_a, rb), this_lt = b.and_(not_a_and_b, eq);
    lt = b.or_(lt, this_lt);
    BoolT not_xor = b.not_(xor_(b, a, rb));
    eq = b.and_(eq, not_xor);
  }
  return lt;
}

// ---- fragments
void nullifier_circuit(Builder& b, const NullifierTargets& t) {  // nullifier.rs:215-242
  u64 salt[2]; injective_string_to_felt("~nullif~", salt);
  std::vector<Target> preimage{b.constant(salt[0]), b.constant(salt[1])};
  preimage.insert(preimage.end(), t.secret.begin(), t.secret.end());
  preimage.push_back(t.transfer_count[0]); preimage.push_back(t.transfer_count[1]);
  for (Target x : preimage) b.range_check(x, 32);
  Hash4 inner = b.hash_n_to_hash_no_pad(preimage), computed = b.hash_n_to_hash_no_pad(hash_vec(inner));
  b.connect_hashes(computed, t.hash);
}
void unspendable_circuit(Builder& b, const UnspendableTargets& t) {   // unspendable_account.rs:182-208
  u64 salt[2]; injective_string_to_felt("wormhole", salt);
  std::vector<Target> preimage{b.constant(salt[0]), b.constant(salt[1])};
  for (Target x : preimage) b.range_check(x, 32);                // the secret is range-checked by the nullifier fragment (same targets once connected)
  preimage.insert(preimage.end(), t.secret.begin(), t.secret.end());
  Hash4 inner = b.hash_n_to_hash_no_pad(preimage), generated = b.hash_n_to_hash_no_pad(hash_vec(inner));
  b.connect_hashes(generated, t.account_id);
}
void storage_proof_circuit(Builder& b, const StorageProofTargets& t) {   // storage_proof/mod.rs:140-244
  const LeafTargets& lf = t.leaf;
  for (Target x : lf.transfer_count) b.range_check(x, 32);       // collect_32_bit_targets (leaf.rs:49-55)
  for (Target x : lf.funding_amount) b.range_check(x, 32);
  std::vector<Target> leaf_vec{lf.transfer_count[0], lf.transfer_count[1]};       // collect_to_vec (leaf.rs:39-47): 14 felts
  for (Target x : lf.funding_account.e) leaf_vec.push_back(x);
  for (Target x : lf.to_account.e) leaf_vec.push_back(x);
  for (Target x : lf.funding_amount) leaf_vec.push_back(x);
  const Hash4 leaf_inputs_hash = b.hash_n_to_hash_no_pad(leaf_vec);
  const Target two_pow_32 = b.constant(u64(1) << 32);
  const int n_log = 5;                                            // bits of MAX_PROOF_LEN - 1
  // Per node, independent of the other nodes (one witness-generation segment each, builder.h begin_segment): is it a proof node / the
  // node after the last one, its hash, the range checks of its felts, and the child hash it commits to.
  struct NodeOut { BoolT is_proof_node, is_leaf_node; Hash4 computed, found; };
  std::vector<NodeOut> nodes(QPW_MAX_PROOF_LEN);
  for (int i = 0; i < QPW_MAX_PROOF_LEN; i++) {
    b.begin_segment(1);
    const std::vector<Target>& node = t.proof_data[i];
    NodeOut& o = nodes[i];
    o.is_proof_node = is_const_less_than(b, i, t.proof_len, n_log);
    o.is_leaf_node = b.is_equal(b.constant(i), t.proof_len);
    o.computed = b.hash_n_to_hash_no_pad(node);
    // the child hash this node commits to: the 8 32-bit limbs starting at felt index indices[i], recombined little-endian
    Target found[4] = {b.zero(), b.zero(), b.zero(), b.zero()};
    const Target expected_index = t.indices[i];
    for (int j = 0; j < QPW_PROOF_NODE_MAX_FELTS - 8; j++) {
      b.range_check(node[j], 32);
      const BoolT is_start = b.is_equal(b.constant(j), expected_index);
      for (int h = 0; h < 4; h++) {
        const Target limbs = b.add(node[j + 2 * h], b.mul(node[j + 2 * h + 1], two_pow_32));     // lo + hi * 2^32
        found[h] = b.select(is_start, limbs, found[h]);
      }
    }
    for (int j = QPW_PROOF_NODE_MAX_FELTS - 8; j < QPW_PROOF_NODE_MAX_FELTS; j++) b.range_check(node[j], 32);
    for (int h = 0; h < 4; h++) o.found.e[h] = found[h];
  }
  b.end_segment();
  // The chain through the nodes: the first node must hash to the root, every further proof node to the hash found in its parent; and for
  // the node after the last one, the hash found in the leaf node must carry the leaf-inputs hash (limbs 1..3; the first is partly
  // overwritten by the trie's nibble prefix).
  Hash4 prev_hash = t.root_hash;
  for (int i = 0; i < QPW_MAX_PROOF_LEN; i++) {
    const NodeOut& o = nodes[i];
    for (int y = 0; y < 4; y++) b.assert_zero(b.mul(b.sub(o.computed.e[y], prev_hash.e[y]), o.is_proof_node.t));
    for (int y = 1; y < 4; y++) b.assert_zero(b.mul(b.sub(leaf_inputs_hash.e[y], prev_hash.e[y]), o.is_leaf_node.t));
    prev_hash = o.found;
  }
}

struct WormholeTargets { NullifierTargets nullifier; UnspendableTargets unspendable; StorageProofTargets storage; ExitAccountTargets exit_account; };
void wormhole_circuit(Builder& b) {                              // circuit.rs:76-137
  WormholeTargets t;
  t.nullifier = nullifier_targets(b); t.unspendable = unspendable_targets(b); t.storage = storage_proof_targets(b); t.exit_account = exit_account_targets(b);
  nullifier_circuit(b, t.nullifier); unspendable_circuit(b, t.unspendable); storage_proof_circuit(b, t.storage);   // the exit account has no constraints
  for (int i = 0; i < 8; i++) b.connect(t.nullifier.secret[i], t.unspendable.secret[i]);                         // connect_shared_targets
  for (int i = 0; i < 2; i++) b.connect(t.nullifier.transfer_count[i], t.storage.leaf.transfer_count[i]);
  b.connect_hashes(t.unspendable.account_id, t.storage.leaf.to_account);
}

void vote_circuit(Builder& b) {                                  // voting/src/lib.rs:69-100 (targets), 123-197 (circuit)
  Hash4 proposal_id = b.add_public_input_hash(), expected_root = b.add_public_input_hash();
  auto bool_input = [&](bool pub) { Target t = b.add_input(); b.assert_zero(b.arithmetic(1, P - 1, t, t, t)); if (pub) b.register_public_input(t); return BoolT{t}; };   // add_virtual_bool_target_safe
  BoolT vote = bool_input(true); (void)vote;
  Hash4 expected_nullifier = b.add_public_input_hash();
  Hash4 private_key = b.add_input_hash();
  std::vector<Hash4> siblings; for (int i = 0; i < QPV_MAX_MERKLE_DEPTH; i++) siblings.push_back(b.add_input_hash());
  std::vector<BoolT> path; for (int i = 0; i < QPV_MAX_MERKLE_DEPTH; i++) path.push_back(bool_input(false));
  Target depth = b.add_input();
  const Hash4 leaf_hash = b.hash_n_to_hash_no_pad(hash_vec(private_key));
  Hash4 cur = leaf_hash;
  const int n_log = 5;
  for (int i = 0; i < QPV_MAX_MERKLE_DEPTH; i++) {
    const BoolT active = is_const_less_than(b, i, depth, n_log);
    std::vector<Target> left, right;
    for (int k = 0; k < 4; k++) { left.push_back(b.select(path[i], siblings[i].e[k], cur.e[k])); right.push_back(b.select(path[i], cur.e[k], siblings[i].e[k])); }
    left.insert(left.end(), right.begin(), right.end());
    const Hash4 parent = b.hash_n_to_hash_no_pad(left);
    for (int k = 0; k < 4; k++) cur.e[k] = b.select(active, parent.e[k], cur.e[k]);
  }
  b.connect_hashes(cur, expected_root);
  std::vector<Target> nin = hash_vec(leaf_hash); for (Target x : proposal_id.e) nin.push_back(x);
  b.connect_hashes(b.hash_n_to_hash_no_pad(nin), expected_nullifier);
}

// ---- input encoding (fill_targets), same order as the *_targets functions above
struct Filler {
  std::vector<u64> v; std::string err;
  void digest(const uint8_t b[32], const char* what) { u64 f[4]; std::string why; if (!digest_bytes_to_felts(b, f, why)) { if (err.empty()) err = std::string(what) + ": " + why; return; } v.insert(v.end(), f, f + 4); }
  void nullifier(const qpw_circuit_inputs& in) {                  // Nullifier::new + fill_targets (nullifier.rs:36-47, 244-253)
    digest(in.nullifier, "nullifier"); injective_bytes_to_felts(in.secret, 32, v); u64 tc[2]; u64_to_felts(in.transfer_count, tc); v.insert(v.end(), tc, tc + 2);
  }
  void unspendable(const qpw_circuit_inputs& in) { digest(in.unspendable_account, "unspendable account"); injective_bytes_to_felts(in.secret, 32, v); }   // unspendable_account.rs:32-36, 210-220
  void storage(const qpw_circuit_inputs& in) {                    // StorageProof::new + fill_targets (storage_proof/mod.rs:89-126, 246-307)
    if (in.proof_len > QPW_MAX_PROOF_LEN) { if (err.empty()) err = "proof length exceeds maximum allowed length: " + std::to_string(in.proof_len) + " > " + std::to_string(QPW_MAX_PROOF_LEN); return; }
    if (in.proof_len && (!in.proof_nodes || !in.proof_node_lens || !in.proof_indices)) { if (err.empty()) err = "null storage proof arrays"; return; }
    for (uint32_t i = 0; i < QPW_MAX_PROOF_LEN; i++) {
      std::vector<u64> node;
      if (i < in.proof_len) {
        injective_bytes_to_felts(in.proof_nodes[i], in.proof_node_lens[i], node);
        if (node.size() > QPW_PROOF_NODE_MAX_FELTS) { if (err.empty()) err = "proof node at index " + std::to_string(i) + " is too large: " + std::to_string(node.size()); return; }
      }
      node.resize(QPW_PROOF_NODE_MAX_FELTS, 0); v.insert(v.end(), node.begin(), node.end());
    }
    for (uint32_t i = 0; i < QPW_MAX_PROOF_LEN; i++) v.push_back(i < in.proof_len ? in.proof_indices[i] / 8 : 0);   // hex-digit index -> felt index (8 digits per felt)
    digest(in.root_hash, "root hash");
    v.push_back(in.proof_len);
    u64 tc[2]; u64_to_felts(in.transfer_count, tc); v.insert(v.end(), tc, tc + 2);                                 // LeafInputs::try_from (leaf.rs:86-97)
    digest(in.funding_account, "funding account"); digest(in.unspendable_account, "to account");
    u64 fa[4]; u128_to_felts(in.funding_amount_le, fa); v.insert(v.end(), fa, fa + 4);
  }
  void exit_account(const qpw_circuit_inputs& in) { digest(in.exit_account, "exit account"); }
};

template <class F> int guard(F&& f) {
  try { return f(); }
  catch (const std::bad_alloc&) { return fail(QPGPU_ERR_OOM, "host memory allocation failed"); }
  catch (...) { return fail(QPGPU_ERR_ARG, "internal error"); }
}
} // namespace

void qpc_set_error(const char* m) { g_err = m; }

extern "C" {

const char* qpc_last_error(void) { return g_err.c_str(); }

int qpc_circuit_build(uint32_t kind, uint32_t zero_knowledge, qpc_circuit** out) { return guard([&]() -> int {
  if (!out || kind > QPC_VOTE) return fail(QPGPU_ERR_ARG, "unknown circuit kind");
  std::unique_ptr<qpc_circuit> c(new qpc_circuit(kind, zero_knowledge != 0));
  Builder& b = c->b;
  switch (kind) {
    case QPC_WORMHOLE: wormhole_circuit(b); break;
    case QPC_NULLIFIER: nullifier_circuit(b, nullifier_targets(b)); break;
    case QPC_UNSPENDABLE_ACCOUNT: unspendable_circuit(b, unspendable_targets(b)); break;
    case QPC_STORAGE_PROOF: storage_proof_circuit(b, storage_proof_targets(b)); break;
    case QPC_EXIT_ACCOUNT: exit_account_targets(b); break;
    case QPC_VOTE: vote_circuit(b); break;
  }
  b.build();
  *out = c.release(); return 0;
}); }
void qpc_circuit_destroy(qpc_circuit* c) { delete c; }
const qpgpu_circuit_desc* qpc_circuit_desc(const qpc_circuit* c) { return c ? &c->b.desc() : nullptr; }
size_t qpc_circuit_num_inputs(const qpc_circuit* c) { return c ? c->b.num_inputs() : 0; }
void qpc_circuit_row_counts(const qpc_circuit* c, uint32_t out[7]) { if (!c || !out) return; c->b.row_counts(out); out[6] = (uint32_t)c->b.num_gate_rows(); }

int qpc_generate_witness_mt(const qpc_circuit* c, const uint64_t* inputs, size_t num_inputs, const uint8_t* blinding_key, uint64_t* wires, uint64_t* public_inputs, uint32_t threads) { return guard([&]() -> int {
  if (!c || !inputs || !wires || (!public_inputs && c->b.desc().num_public_inputs)) return fail(QPGPU_ERR_ARG, "null argument");
  if (num_inputs != c->b.num_inputs()) return fail(QPGPU_ERR_ARG, "expected " + std::to_string(c->b.num_inputs()) + " input values, got " + std::to_string(num_inputs));
  if (threads == 0) threads = (uint32_t)std::max(1, std::min(8, qp::usable_cpus()));
  const std::string e = c->b.generate(inputs, blinding_key, wires, public_inputs, (int)threads);
  if (!e.empty()) return fail(QPGPU_ERR_WITNESS, e);
  return 0;
}); }
int qpc_generate_witness(const qpc_circuit* c, const uint64_t* inputs, size_t num_inputs, const uint8_t* blinding_key, uint64_t* wires, uint64_t* public_inputs) {
  return qpc_generate_witness_mt(c, inputs, num_inputs, blinding_key, wires, public_inputs, 0);
}
int qpc_circuit_parallel_segments(const qpc_circuit* c) { return c && c->b.parallel_replay() ? 1 : 0; }

int qpw_fill_targets(uint32_t kind, const qpw_circuit_inputs* in, uint64_t* inputs, size_t cap, size_t* len) { return guard([&]() -> int {
  if (!in || !len) return fail(QPGPU_ERR_ARG, "null argument");
  Filler f;
  switch (kind) {
    case QPC_WORMHOLE: f.nullifier(*in); f.unspendable(*in); f.storage(*in); f.exit_account(*in); break;     // commit(): lib.rs:214-223
    case QPC_NULLIFIER: f.nullifier(*in); break;
    case QPC_UNSPENDABLE_ACCOUNT: f.unspendable(*in); break;
    case QPC_STORAGE_PROOF: f.storage(*in); break;
    case QPC_EXIT_ACCOUNT: f.exit_account(*in); break;
    default: return fail(QPGPU_ERR_ARG, "not a wormhole circuit kind");
  }
  if (!f.err.empty()) return fail(QPGPU_ERR_ARG, f.err);
  *len = f.v.size();
  if (!inputs) return 0;
  if (cap < f.v.size()) return fail(QPGPU_ERR_BUFFER, "output buffer too small: need " + std::to_string(f.v.size()));
  memcpy(inputs, f.v.data(), f.v.size() * sizeof(u64)); return 0;
}); }

int qpw_nullifier_from_preimage(const uint8_t* secret, size_t secret_len, uint64_t transfer_count, uint8_t out[32]) { return guard([&]() -> int {
  if (!secret || !out) return fail(QPGPU_ERR_ARG, "null argument");
  u64 salt[2], tc[2], h[4]; injective_string_to_felt("~nullif~", salt); u64_to_felts(transfer_count, tc);
  std::vector<u64> pre(salt, salt + 2); injective_bytes_to_felts(secret, secret_len, pre); pre.insert(pre.end(), tc, tc + 2);
  hash_no_pad(pre.data(), pre.size(), h); hash_no_pad(h, 4, h);
  digest_felts_to_bytes(h, out); return 0;
}); }
int qpw_unspendable_account_from_secret(const uint8_t secret[32], uint8_t out[32]) { return guard([&]() -> int {
  if (!secret || !out) return fail(QPGPU_ERR_ARG, "null argument");
  u64 salt[2], h[4]; injective_string_to_felt("wormhole", salt);
  std::vector<u64> pre(salt, salt + 2); injective_bytes_to_felts(secret, 32, pre);
  hash_no_pad(pre.data(), pre.size(), h); hash_no_pad(h, 4, h);
  digest_felts_to_bytes(h, out); return 0;
}); }
int qpw_public_inputs_from_felts(const uint64_t* pis, size_t count, qpw_public_inputs* out) { return guard([&]() -> int {
  if (!pis || !out) return fail(QPGPU_ERR_ARG, "null argument");
  if (count != 16) return fail(QPGPU_ERR_ARG, "public inputs should contain: 16 field elements, got: " + std::to_string(count));
  digest_felts_to_bytes(pis + 0, out->nullifier); digest_felts_to_bytes(pis + 4, out->root_hash); digest_felts_to_bytes(pis + 12, out->exit_account);
  memset(out->funding_amount_le, 0, 16);
  for (int k = 0; k < 4; k++) {                                  // felts_to_u128 (utils.rs:114-121): limb k is bits [96-32k, 128-32k)
    if (pis[8 + k] > 0xFFFFFFFFULL) return fail(QPGPU_ERR_ARG, "failed to deserialize funding amount: limb " + std::to_string(k) + " = " + std::to_string(pis[8 + k]) + " is wider than 32 bits");
    for (int i = 0; i < 4; i++) out->funding_amount_le[4 * (3 - k) + i] = (uint8_t)(pis[8 + k] >> (8 * i));
  }
  return 0;
}); }

int qpv_fill_targets(const qpv_vote_inputs* in, uint64_t* inputs, size_t cap, size_t* len) { return guard([&]() -> int {   // voting/src/lib.rs:199-260
  if (!in || !len) return fail(QPGPU_ERR_ARG, "null argument");
  if (in->actual_merkle_depth > QPV_MAX_MERKLE_DEPTH)
    return fail(QPGPU_ERR_ARG, "Merkle tree depth " + std::to_string(in->actual_merkle_depth) + " exceeds maximum allowed depth " + std::to_string(QPV_MAX_MERKLE_DEPTH));
  if (in->num_siblings != in->num_path_indices)
    return fail(QPGPU_ERR_ARG, "Merkle proof length mismatch: " + std::to_string(in->num_siblings) + " siblings vs " + std::to_string(in->num_path_indices) + " path indices");
  if (in->actual_merkle_depth > in->num_siblings) return fail(QPGPU_ERR_ARG, "Merkle proof shorter than the stated depth");
  std::vector<u64> v;
  v.insert(v.end(), in->proposal_id, in->proposal_id + 4); v.insert(v.end(), in->merkle_root, in->merkle_root + 4);
  v.push_back(in->vote ? 1 : 0); v.insert(v.end(), in->nullifier, in->nullifier + 4);
  v.insert(v.end(), in->private_key, in->private_key + 4);
  for (uint32_t i = 0; i < QPV_MAX_MERKLE_DEPTH; i++) for (int k = 0; k < 4; k++) v.push_back(i < in->actual_merkle_depth ? in->merkle_siblings[4 * i + k] : 0);
  for (uint32_t i = 0; i < QPV_MAX_MERKLE_DEPTH; i++) v.push_back(i < in->actual_merkle_depth && in->path_indices[i] ? 1 : 0);
  v.push_back(in->actual_merkle_depth);
  *len = v.size();
  if (!inputs) return 0;
  if (cap < v.size()) return fail(QPGPU_ERR_BUFFER, "output buffer too small: need " + std::to_string(v.size()));
  memcpy(inputs, v.data(), v.size() * sizeof(u64)); return 0;
}); }

void qpc_hash_no_pad(const uint64_t* felts, size_t count, uint64_t out[4]) { hash_no_pad(felts, count, out); }

} // extern "C"

// ---- test hook (include/qpcircuit_internal.h): random builder programs with a native shadow evaluation
#include "../../include/qpcircuit_internal.h"
extern "C" int qpc_test_random_circuit(uint64_t seed, uint32_t zero_knowledge, uint32_t num_inputs, uint32_t num_ops, qpc_circuit** out, uint64_t* inputs,
                                       uint64_t* expected, size_t cap, size_t* num_outputs) { return guard([&]() -> int {
  using namespace qp;
  if (!out || !inputs || !expected || !num_outputs || num_inputs < 2) return fail(QPGPU_ERR_ARG, "bad argument");
  u64 s = seed * 0x9E3779B97F4A7C15ULL + 0x5157;
  auto next = [&]() { u64 z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); };
  auto felt = [&]() { u64 v = next(); return v >= P ? v - P : v; };
  const bool recursion_gates = (zero_knowledge & 2) != 0;        // bit 1: the program also draws from the recursive verifier's gates
  zero_knowledge &= 1;
  std::unique_ptr<qpc_circuit> c(new qpc_circuit(QPC_AGGREGATION + 1, zero_knowledge != 0));
  Builder& b = c->b;
  struct E2 { u64 a, b; };
  auto emul = [](E2 x, E2 y) { return E2{gl_add(gl_mul(x.a, y.a), gl_mul(7, gl_mul(x.b, y.b))), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))}; };
  auto eadd = [](E2 x, E2 y) { return E2{gl_add(x.a, y.a), gl_add(x.b, y.b)}; };
  auto escale = [](E2 x, u64 k) { return E2{gl_mul(x.a, k), gl_mul(x.b, k)}; };
  std::vector<Target> t; std::vector<u64> v;                    // targets and their native values
  for (uint32_t i = 0; i < num_inputs; i++) {
    u64 x = next() % 5 == 0 ? next() % 4 : (next() % 3 == 0 ? next() & 0xFFFFFFFFULL : felt());      // bits, 32-bit values and full field elements
    inputs[i] = x; t.push_back(b.add_input()); v.push_back(x);
  }
  auto pick = [&]() { return (size_t)(next() % t.size()); };
  auto push = [&](Target tt, u64 vv) { t.push_back(tt); v.push_back(vv); };
  auto as_bool = [&](size_t i) { BoolT r = b.is_equal(t[i], b.zero()); return std::make_pair(r, (u64)(v[i] == 0)); };   // a boolean derived from any value
  for (uint32_t op = 0; op < num_ops; op++) {
    const size_t i = pick(), j = pick(), k = pick();
    auto pick_ext = [&](Ext2& t2, E2& v2) { const size_t x = pick(), y = pick(); t2 = Ext2{t[x], t[y]}; v2 = E2{v[x], v[y]}; };
    if (recursion_gates && next() % 3 == 0) {
      switch (next() % 8) {
        case 7: {                                                   // MulExtension
          Ext2 m0, m1; E2 v0, v1; pick_ext(m0, v0); pick_ext(m1, v1);
          const u64 k0 = next() % 3 == 0 ? next() % 2 : felt();
          const Ext2 o = b.mul_ext(k0, m0, m1); const E2 ov = escale(emul(v0, v1), k0);
          push(o.a, ov.a); push(o.b, ov.b); break;
        }
        case 0: {                                                   // ArithmeticExtension
          Ext2 m0, m1, ad; E2 v0, v1, va; pick_ext(m0, v0); pick_ext(m1, v1); pick_ext(ad, va);
          const u64 k0 = next() % 3 == 0 ? next() % 2 : felt(), k1 = next() % 3 == 0 ? next() % 2 : felt();
          const Ext2 o = b.arith_ext(k0, k1, m0, m1, ad); const E2 ov = eadd(escale(emul(v0, v1), k0), escale(va, k1));
          push(o.a, ov.a); push(o.b, ov.b); break;
        }
        case 1: {                                                   // Reducing: 1 .. 100 base coefficients (one to three rows)
          Ext2 al; E2 av; pick_ext(al, av);
          const int cnt = 1 + (int)(next() % 100); std::vector<Target> cs; E2 acc{0, 0};
          for (int q = 0; q < cnt; q++) { const size_t z = pick(); cs.push_back(t[z]); acc = eadd(emul(acc, av), E2{v[z], 0}); }
          const Ext2 o = b.reduce_base(al, cs); push(o.a, acc.a); push(o.b, acc.b); break;
        }
        case 2: {                                                   // ReducingExtension: 1 .. 70 extension coefficients
          Ext2 al; E2 av; pick_ext(al, av);
          const int cnt = 1 + (int)(next() % 70); std::vector<Ext2> cs; E2 acc{0, 0};
          for (int q = 0; q < cnt; q++) { Ext2 ct; E2 cv; pick_ext(ct, cv); cs.push_back(ct); acc = eadd(emul(acc, av), cv); }
          const Ext2 o = b.reduce_ext(al, cs); push(o.a, acc.a); push(o.b, acc.b); break;
        }
        case 3: {                                                   // RandomAccess: index = four bits of a value
          Target hi_, lo_; b.lohi_hint(t[i], lo_, hi_);
          b.connect(b.arithmetic(u64(1) << 32, 1, hi_, b.one(), lo_), t[i]); b.range_check(hi_, 32);
          std::vector<BoolT> bits = b.split_le(lo_, 32);
          const int from = (int)(next() % 28);
          Target idx = b.zero(); for (int q = 3; q >= 0; q--) idx = b.arithmetic(2, 1, idx, b.one(), bits[from + q].t);
          const u64 iv = ((v[i] & 0xFFFFFFFFULL) >> from) & 15;
          std::vector<Target> list; std::vector<u64> lv; const int len = next() % 4 == 0 ? 1 + (int)(next() % 16) : 16;
          for (int q = 0; q < len; q++) { const size_t z = pick(); list.push_back(t[z]); lv.push_back(v[z]); }
          push(b.random_access(idx, list), iv < lv.size() ? lv[iv] : 0); break;
        }
        case 4: {                                                   // PoseidonMds
          Ext2 in[12], out[12]; u64 a[12], bb[12];
          for (int q = 0; q < 12; q++) { E2 ev; pick_ext(in[q], ev); a[q] = ev.a; bb[q] = ev.b; }
          b.poseidon_mds(in, out); mds_layer(a); mds_layer(bb);
          for (int q = 0; q < 12; q += 5) { push(out[q].a, a[q]); push(out[q].b, bb[q]); }
          break;
        }
        case 5: if (v[i]) {                                         // CosetInterpolation against a native Lagrange evaluation
          Ext2 vals[16], pt; E2 vv[16], pv; for (int q = 0; q < 16; q++) pick_ext(vals[q], vv[q]); pick_ext(pt, pv);
          const Ext2 o = b.coset_interpolate(t[i], vals, pt);
          const u64* dom = Builder::coset_domain(); const u64* wg = Builder::coset_weights();
          const E2 xs = escale(pv, gl_inv(v[i])); E2 sum{0, 0};
          for (int q = 0; q < 16; q++) { E2 term = escale(vv[q], wg[q]); for (int r = 0; r < 16; r++) if (r != q) term = emul(term, E2{gl_sub(xs.a, dom[r]), xs.b}); sum = eadd(sum, term); }
          push(o.a, sum.a); push(o.b, sum.b);
        } break;
        default: {                                                  // Exponentiation by up to 40 bits of a value
          Target hi_, lo_; b.lohi_hint(t[j], lo_, hi_);
          b.connect(b.arithmetic(u64(1) << 32, 1, hi_, b.one(), lo_), t[j]); b.range_check(hi_, 32);
          std::vector<BoolT> bits = b.split_le(lo_, 32);
          const int nb = 1 + (int)(next() % 32); bits.resize(nb);
          push(b.exponentiate(t[i], bits), gl_pow(v[i], (v[j] & 0xFFFFFFFFULL) & ((u64(1) << nb) - 1))); break;
        }
      }
      continue;
    }
    switch (next() % 16) {
      case 0: push(b.add(t[i], t[j]), gl_add(v[i], v[j])); break;
      case 1: push(b.sub(t[i], t[j]), gl_sub(v[i], v[j])); break;
      case 2: push(b.mul(t[i], t[j]), gl_mul(v[i], v[j])); break;
      case 3: { u64 cst = next() % 3 == 0 ? next() % 3 : felt(); push(b.mul_const(cst, t[i]), gl_mul(cst, v[i])); break; }
      case 4: push(b.mul_add(t[i], t[j], t[k]), gl_add(gl_mul(v[i], v[j]), v[k])); break;
      case 5: { u64 k0 = next() % 4 == 0 ? next() % 2 : felt(), k1 = next() % 4 == 0 ? next() % 2 : felt();
                push(b.arithmetic(k0, k1, t[i], t[j], t[k]), gl_add(gl_mul(k0, gl_mul(v[i], v[j])), gl_mul(k1, v[k]))); break; }
      case 6: { u64 cst = next() % 4 == 0 ? next() % 2 : felt(); push(b.constant(cst), cst); break; }
      case 7: { auto e = b.is_equal(t[i], t[j]); push(e.t, (u64)(v[i] == v[j])); break; }
      case 8: { auto bi = as_bool(i); push(b.select(bi.first, t[j], t[k]), bi.second ? v[j] : v[k]); break; }
      case 9: { auto bi = as_bool(i), bj = as_bool(j); u64 w = next() % 3;
                BoolT r = w == 0 ? b.and_(bi.first, bj.first) : (w == 1 ? b.or_(bi.first, bj.first) : b.not_(bi.first));
                push(r.t, w == 0 ? (bi.second & bj.second) : (w == 1 ? (bi.second | bj.second) : (1 - bi.second))); break; }
      case 10: {                                                  // split a 32-bit truncation and recombine two of its bit ranges
        Target hi_, lo_; b.lohi_hint(t[i], lo_, hi_);
        b.connect(b.arithmetic(u64(1) << 32, 1, hi_, b.one(), lo_), t[i]); b.range_check(hi_, 32);
        const u64 lov = v[i] & 0xFFFFFFFFULL;
        std::vector<BoolT> bits = b.split_le(lo_, 32);
        Target acc = b.zero(); u64 accv = 0; const int from = (int)(next() % 20), len = 1 + (int)(next() % 12);
        for (int q = len - 1; q >= 0; q--) { acc = b.arithmetic(2, 1, acc, b.one(), bits[from + q].t); accv = 2 * accv + ((lov >> (from + q)) & 1); }
        push(acc, accv); push(hi_, v[i] >> 32); break;
      }
      case 11: { std::vector<Target> in; std::vector<u64> inv; const int n = 1 + (int)(next() % 19);
                 for (int q = 0; q < n; q++) { size_t z = pick(); in.push_back(t[z]); inv.push_back(v[z]); }
                 Hash4 h = b.hash_n_to_hash_no_pad(in); u64 hv[4]; hash_no_pad(inv.data(), inv.size(), hv);
                 for (int q = 0; q < 4; q++) push(h.e[q], hv[q]); break; }
      case 12: { auto bi = as_bool(i); Target st[12]; u64 sv[12];
                 for (int q = 0; q < 12; q++) { size_t z = pick(); st[q] = t[z]; sv[q] = v[z]; }
                 b.permute_swap(st, bi.first);
                 if (bi.second) for (int q = 0; q < 4; q++) std::swap(sv[q], sv[q + 4]);
                 poseidon_permute(sv);
                 for (int q = 0; q < 12; q += 5) push(st[q], sv[q]); break; }
      case 13: { Target a = b.add(t[i], t[j]), a2 = b.add(t[j], t[i]); b.connect(a, a2); push(a2, gl_add(v[i], v[j])); break; }   // equal by construction
      case 14: if (v[i]) { Target iv = b.inverse_hint(t[i]); b.connect(b.mul(iv, t[i]), b.one()); push(iv, gl_inv(v[i])); } break;
      default: { Target ia, ib; b.ext_inverse_hint(t[i], t[j], ia, ib);
                 const u64 nrm = gl_sub(gl_sqr(v[i]), gl_mul(7, gl_sqr(v[j])));
                 if (nrm) { const u64 ni = gl_inv(nrm); push(ia, gl_mul(v[i], ni)); push(ib, gl_mul(gl_sub(0, v[j]), ni)); } break; }
    }
  }
  size_t n_out = 0;
  for (size_t i = num_inputs; i < t.size() && n_out < cap; i += 1 + next() % 3) { b.register_public_input(t[i]); expected[n_out++] = v[i]; }
  *num_outputs = n_out;
  b.build();
  *out = c.release(); return 0;
}); }

extern "C" size_t qpc_circuit_num_hints(const qpc_circuit* c) { return c ? c->b.num_hints() : 0; }
extern "C" int qpc_mutation_sweep(const qpc_circuit* c, const uint64_t* inputs, size_t num_inputs, const uint8_t* blinding_key, uint32_t what, size_t first, size_t step,
                                  uint32_t modes_mask, uint64_t seed, uint32_t full_every, qpc_witness_check_fn check, void* ctx, qpc_mutation_result* out, size_t cap,
                                  size_t* count) { return guard([&]() -> int {
  if (!c || !inputs || !check || !out || !count) return fail(QPGPU_ERR_ARG, "null argument");
  if (num_inputs != c->b.num_inputs()) return fail(QPGPU_ERR_ARG, "expected " + std::to_string(c->b.num_inputs()) + " input values, got " + std::to_string(num_inputs));
  static_assert(sizeof(qpc_mutation_result) == sizeof(Builder::MutationResult), "same layout");
  std::vector<Builder::MutationResult> res;
  const std::string e = c->b.mutation_sweep(inputs, blinding_key, what, first, step, cap, modes_mask, seed, full_every, (Builder::WitnessCheck)check, ctx, res);
  *count = std::min(res.size(), cap);
  memcpy(out, res.data(), *count * sizeof(qpc_mutation_result));
  if (!e.empty()) return fail(QPGPU_ERR_WITNESS, e);
  return 0;
}); }
