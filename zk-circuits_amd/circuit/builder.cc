// Builder::build (public-input hash, zk blinding rows, padding, sigma, descriptor) and Builder::generate (witness generation).
#include "builder.h"
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cstring>
#include <thread>
#include <cstdio>
#include <cstdlib>

namespace qpc {
using namespace qp;

namespace {
// FriReductionStrategy::ConstantArityBits(4, 5) under rate_bits 3 / cap_height 4 (the strategy both reference configs write,
// csrc/artifacts.hip fri_config): fold by 16 while the degree is above 2^5 and the folded layer still has at least a cap's worth of leaves.
// Pinned by the reference's artefacts: k = 13 -> [4,4] (aggregator/data/dummy_proof.bin), k = 14 -> [4,4,4] (bench-data/common.bin).
std::vector<uint32_t> fri_arities(uint32_t degree_bits) {
  std::vector<uint32_t> a; uint32_t d = degree_bits;
  while (d > 5 && d + 3 >= 4 + 4) { a.push_back(4); d -= 4; }
  return a;
}
uint32_t ceil_log2(size_t x) { uint32_t k = 0; while ((size_t(1) << k) < x) k++; return k; }
// per row kind: constraint degree (orders the gate list and forms the selector groups), serializer tag, parameters
const int DEGREE[K_ALL_KINDS] = {0, 1, 1, 2, 3, 7, 3, 3, 2, 2, RA_BITS + 1, 1, 4, COSET_DEG};
const uint32_t TAG[K_ALL_KINDS] = {QPGPU_GATE_NOOP, QPGPU_GATE_CONSTANT, QPGPU_GATE_PUBLIC_INPUT, QPGPU_GATE_BASESUM2, QPGPU_GATE_ARITHMETIC, QPGPU_GATE_POSEIDON,
                                   QPGPU_GATE_ARITHMETIC_EXTENSION, QPGPU_GATE_MUL_EXTENSION, QPGPU_GATE_REDUCING, QPGPU_GATE_REDUCING_EXTENSION, QPGPU_GATE_RANDOM_ACCESS,
                                   QPGPU_GATE_POSEIDON_MDS, QPGPU_GATE_EXPONENTIATION, QPGPU_GATE_COSET_INTERPOLATION};
const uint64_t PARAM[K_ALL_KINDS] = {0, 2, 0, BASESUM_LIMBS, ARITH_OPS, 0, ARITH_EXT_OPS, 13, REDUCING_COEFFS, REDUCING_EXT_COEFFS, RA_BITS, 0, EXP_BITS, COSET_BITS};
const uint32_t PARAM2[K_ALL_KINDS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, RA_COPIES, 0, 0, COSET_DEG};
const uint32_t PARAM3[K_ALL_KINDS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, RA_EXTRA, 0, 0, 0};
const uint32_t NUM_QUERIES = 28, CAP_HEIGHT = 4, POW_BITS = 16, RATE_BITS = 3;
}

const u64* Builder::coset_domain() {
  static const std::vector<u64> dom = [] { std::vector<u64> d; u64 wg = gl_root(COSET_BITS), x = 1; for (int i = 0; i < COSET_NP; i++) { d.push_back(x); x = gl_mul(x, wg); } return d; }();
  return dom.data();
}
const u64* Builder::coset_weights() {
  static const std::vector<u64> w = [] {
    const u64* dom = coset_domain(); std::vector<u64> out;
    for (int i = 0; i < COSET_NP; i++) { u64 d = 1; for (int j = 0; j < COSET_NP; j++) if (j != i) d = gl_mul(d, gl_sub(dom[i], dom[j])); out.push_back(gl_inv(d)); }
    return out;
  }();
  return w.data();
}

static double now_ms();
void Builder::build() {
  if (built_) return;
  static const bool tmb = getenv("QPC_TIMING") != nullptr; double tb0 = now_ms();
  auto lapb = [&](const char* what) { if (tmb) { double t = now_ms(); fprintf(stderr, "  build %-10s %.3f ms\n", what, t - tb0); tb0 = t; } };
  // public inputs: hashed in-circuit, the digest tied to the PublicInput gate's wires (plonky2 does the same in build()).
  { Hash4 h = hash_n_to_hash_no_pad(public_inputs_); for (int i = 0; i < 4; i++) connect(h.e[i], wire(0, i)); }
  gate_rows_ = row_kind_.size();
  uint32_t k = std::max<uint32_t>(5, ceil_log2(gate_rows_));
  if (zk_) {
    // Blinding rows, sized as plonky2's blinding_counts does: every wire polynomial is opened at zeta (D = 2 base elements) and at each FRI
    // query (one leaf, plus what the folding points and the final polynomial reveal): `regular` rows of random wires hide the wire
    // polynomials; `z` pairs of rows holding the same random routed values under a copy constraint give Z and the partial products
    // random factors (2D openings: zeta and g*zeta).
    for (;; k++) {
      const auto ar = fri_arities(k);
      size_t fold = 0, red = 0; for (uint32_t a : ar) { fold += (size_t(1) << a) - 1; red += a; }
      const size_t final_coeffs = size_t(1) << (k - red), fri_open = NUM_QUERIES * (1 + 2 * fold + 2 * final_coeffs);
      blind_plain_ = 2 + fri_open; blind_pairs_ = 4 + fri_open;
      if (gate_rows_ + blind_plain_ + 2 * blind_pairs_ <= (size_t(1) << k)) break;
    }
    blind_plain_first_ = gate_rows_; blind_pair_first_ = gate_rows_ + blind_plain_;
  }
  n_ = size_t(1) << k;
  row_kind_.resize(n_, K_NOOP); row_c0_.resize(n_, 0); row_c1_.resize(n_, 0);

  lapb("rows");
  // copy classes
  cls_.resize(cells_.size());
  { std::vector<uint32_t> index(cells_.size(), UINT32_MAX); ncls_ = 0;
    for (size_t t = 0; t < cells_.size(); t++) { uint32_t r = find((uint32_t)t); if (index[r] == UINT32_MAX) index[r] = (uint32_t)ncls_++; cls_[t] = index[r]; } }
  // the cells in use, in column-major order (witness generation writes the wire matrix column by column)
  { const size_t rows = wire_of_.size() / W;
    for (int col = 0; col < W; col++) for (size_t row = 0; row < rows; row++) { const Target t = wire_of_[row * W + (size_t)col]; if (t != NO_TARGET) wire_targets_.push_back(t); } }
  fill_dst_.reserve(wire_targets_.size()); fill_cls_.reserve(wire_targets_.size());
  for (Target t : wire_targets_) { fill_dst_.push_back((uint32_t)((size_t)cells_[t].col * n_ + cells_[t].row)); fill_cls_.push_back(cls_[t]); }

  // replay phases (builder.h begin_segment): partition the operations and check that the segments are independent as declared
  { seg_ops_[1].assign(seg_count_[1], {}); seg_ops_[2].assign(seg_count_[2], {});
    // first writer of every class in program order: 0 = input / constant / leading operation, else 1 + stage * 2^28 + segment; post = ~0
    std::vector<uint32_t> writer(ncls_, UINT32_MAX - 1);                 // UINT32_MAX - 1: not written yet
    for (Target t : inputs_) writer[cls_[t]] = 0;
    bool seen_segment = false, ok = true;
    auto reads = [&](const Op& op, std::vector<Target>& in) {
      in.clear();
      switch (op.kind) {
        case OP_ARITH: in = {op.a, op.b, op.c}; break;
        case OP_INV: case OP_LOHI: in = {op.a}; break;
        case OP_EXTINV: in = {op.a, op.b}; break;
        case OP_SPLIT: in = {op.out}; break;
        case OP_POSEIDON: for (int i = 0; i < 13; i++) in.push_back(op.out + i); break;
        default: special_reads(op, [&](Target t) { in.push_back(t); }); break;
      }
    };
    auto writes = [&](const Op& op, std::vector<Target>& out) {
      out.clear();
      switch (op.kind) {
        case OP_CONST: case OP_ARITH: case OP_INV: out = {op.out}; break;
        case OP_LOHI: case OP_EXTINV: out = {op.out, op.c}; break;
        case OP_SPLIT: for (int l = 0; l < BASESUM_LIMBS; l++) out.push_back(op.out + 1 + l); break;
        case OP_POSEIDON: for (int i = 0; i < 12; i++) out.push_back(op.out + 13 + i); break;
        default: special_writes(op, [&](Target t) { out.push_back(t); }); break;
      }
    };
    std::vector<Target> tmp;
    for (const Op& op : ops_) if (op.kind == OP_CONST) writer[cls_[op.out]] = 0;       // constants are replayed first
    for (uint32_t i = 0; i < ops_.size(); i++) {
      const Op& op = ops_[i];
      if (op.kind == OP_CONST) { const_ops_.push_back(i); continue; }
      uint32_t me;
      if (op.stage == 0) { me = seen_segment ? UINT32_MAX : 0; (seen_segment ? post_ops_ : pre_ops_).push_back(i); }
      else { seen_segment = true; me = 1 + ((uint32_t)op.stage << 28) + op.seg; seg_ops_[op.stage][op.seg].push_back(i); }
      reads(op, tmp);
      for (Target t : tmp) {
        const uint32_t w = writer[cls_[t]];
        bool legal = w == 0 || w == me || me == UINT32_MAX;               // leading data, own segment, or a trailing operation (runs last)
        if (!legal && op.stage == 2 && w != UINT32_MAX && w != UINT32_MAX - 1 && (w - 1) >> 28 == 1) legal = true;   // stage 2 may read stage 1
        if (!legal) { if (ok && getenv("QPC_TIMING")) fprintf(stderr, "  segments not independent: op %u (kind %d, stage %d seg %u) reads a class first written by %x\n", i, op.kind, op.stage, op.seg, w); ok = false; }
      }
      writes(op, tmp);
      for (Target t : tmp) if (writer[cls_[t]] == UINT32_MAX - 1) writer[cls_[t]] = me;
    }
    parallel_ok_ = ok && seen_segment;
    if (getenv("QPC_TIMING")) { size_t s1 = 0, s2 = 0; for (auto& v : seg_ops_[1]) s1 += v.size(); for (auto& v : seg_ops_[2]) s2 += v.size();
      fprintf(stderr, "  replay phases: %zu constants, %zu leading, %zu in %zu stage-1 segments, %zu in %zu stage-2 segments, %zu trailing; parallel %d\n",
              const_ops_.size(), pre_ops_.size(), s1, seg_ops_[1].size(), s2, seg_ops_[2].size(), post_ops_.size(), (int)parallel_ok_); } }

  // sigma: the routed cells of a class form one cycle; a blinding pair swaps its two cells
  std::vector<uint32_t> sigma((size_t)R * n_);
  for (size_t i = 0; i < sigma.size(); i++) sigma[i] = (uint32_t)i;
  { std::vector<uint32_t> start(ncls_ + 1, 0);                         // routed cells grouped by class (counting sort), each group one cycle
    for (Target t : wire_targets_) if (cells_[t].col < R) start[cls_[t] + 1]++;
    for (size_t c = 0; c < ncls_; c++) start[c + 1] += start[c];
    std::vector<uint32_t> cellv(start[ncls_]), fillp(start.begin(), start.end() - 1);
    for (Target t : wire_targets_) if (cells_[t].col < R) cellv[fillp[cls_[t]]++] = (uint32_t)((size_t)cells_[t].col * n_ + cells_[t].row);
    for (size_t c = 0; c < ncls_; c++) {
      const uint32_t lo = start[c], hi = start[c + 1];
      if (hi - lo > 1) for (uint32_t i = lo; i < hi; i++) sigma[cellv[i]] = cellv[i + 1 < hi ? i + 1 : lo];
    } }
  for (size_t p = 0; p < blind_pairs_; p++) for (int c = 0; c < R; c++) {
    const size_t a = (size_t)c * n_ + blind_pair_first_ + 2 * p, b = a + 1; sigma[a] = (uint32_t)b; sigma[b] = (uint32_t)a;
  }

  lapb("sigma");
  // gate list ordered by degree; selector groups as upstream forms them (a group grows while size + next degree < 8): [0,5), [5,6)
  int gate_index[K_ALL_KINDS];
  { std::vector<int> kinds; for (int i = 0; i < K_KINDS; i++) kinds.push_back(i);            // the leaf set always; a recursion gate if the circuit has such a row
    { bool used[K_ALL_KINDS] = {false}; for (size_t r = 0; r < gate_rows_; r++) used[row_kind_[r]] = true;
      for (int i = K_KINDS; i < K_ALL_KINDS; i++) { gate_index[i] = -1; if (used[i]) kinds.push_back(i); } }
    std::stable_sort(kinds.begin(), kinds.end(), [](int a, int b) { return DEGREE[a] < DEGREE[b]; });
    for (size_t i = 0; i < kinds.size(); i++) gate_index[kinds[i]] = (int)i;
    sel_idx_.assign(kinds.size(), 0); size_t start = 0;
    while (start < kinds.size()) {
      size_t size = 0;
      while (start + size < kinds.size() && (int)size + DEGREE[kinds[start + size]] < 8) size++;
      for (size_t i = 0; i < size; i++) sel_idx_[start + i] = (uint32_t)(groups_.size() / 2);
      groups_.push_back((uint32_t)start); groups_.push_back((uint32_t)(start + size)); start += size;
    }
    for (int kd : kinds) {
      qpgpu_gate g; memset(&g, 0, sizeof g); g.tag = TAG[kd]; g.param = PARAM[kd]; g.param2 = PARAM2[kd]; g.param3 = PARAM3[kd];
      if (kd == K_COSET) { g.table = coset_weights(); g.table_len = COSET_NP; }
      gates_.push_back(g);
    } }
  const int nsel = (int)(groups_.size() / 2);                  // 2 for this gate set; NC = nsel + 2 local constants
  k_is_.resize(R); { u64 x = 1; for (int j = 0; j < R; j++) { k_is_[j] = x; x = gl_mul(x, GL_GEN); } }
  constants_sigmas_.assign((size_t)(nsel + 2 + R) * n_, 0);
  u64* cs = constants_sigmas_.data();
  for (size_t i = 0; i < n_; i++) {
    const int gi = gate_index[row_kind_[i]];
    for (int sc = 0; sc < nsel; sc++) cs[(size_t)sc * n_ + i] = (int)sel_idx_[gi] == sc ? (u64)gi : 0xFFFFFFFFULL;
    cs[(size_t)nsel * n_ + i] = row_c0_[i]; cs[(size_t)(nsel + 1) * n_ + i] = row_c1_[i];
  }
  { std::vector<u64> wpow(n_); u64 w = gl_root((int)k), x = 1;
    for (size_t i = 0; i < n_; i++) { wpow[i] = x; x = gl_mul(x, w); }
    for (int j = 0; j < R; j++) for (size_t i = 0; i < n_; i++) { uint32_t t = sigma[(size_t)j * n_ + i]; cs[(size_t)(nsel + 2 + j) * n_ + i] = gl_mul(k_is_[t / n_], wpow[t % n_]); } }

  lapb("columns");
  qpgpu_circuit_desc& d = desc_; memset(&d, 0, sizeof d);
  d.degree_bits = k; d.num_wires = W; d.num_routed = R; d.num_constants = (uint32_t)(nsel + 2); d.num_challenges = 2;
  d.quotient_degree_factor = 8; d.num_partial_products = 9; d.num_public_inputs = (uint32_t)public_inputs_.size(); d.num_gate_constraints = 123;
  d.zero_knowledge = zk_; d.fri_rate_bits = RATE_BITS; d.fri_cap_height = CAP_HEIGHT; d.fri_num_queries = NUM_QUERIES; d.fri_pow_bits = POW_BITS;
  { const auto ar = fri_arities(k); d.fri_num_layers = (uint32_t)ar.size(); for (size_t i = 0; i < ar.size(); i++) d.fri_arity_bits[i] = ar[i]; }
  d.num_gates = (uint32_t)gates_.size(); d.num_selector_groups = (uint32_t)nsel; d.gates = gates_.data(); d.selector_indices = sel_idx_.data();
  d.selector_group_ranges = groups_.data(); d.k_is = k_is_.data(); d.constants_sigmas = cs;
  // Circuit digest: binds every constants/sigmas value and the shape (upstream hashes the constants/sigmas cap with the common data;
  // any four felts that commit to the circuit serve the transcript — this one is this builder's own).
  { const int ncol = nsel + 2 + R;                               // one digest per column (independent sponges, spread over host threads), then one over them
    std::vector<u64> col_digest((size_t)4 * ncol);
    auto hash_cols = [&](int lo, int hi) { for (int col = lo; col < hi; col++) hash_no_pad(cs + (size_t)col * n_, n_, &col_digest[4 * (size_t)col]); };
    const int nthr = std::max(1, std::min(8, qp::usable_cpus()));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthr; t++) pool.emplace_back(hash_cols, ncol * t / nthr, ncol * (t + 1) / nthr);
    hash_cols(0, ncol / nthr);
    for (auto& th : pool) th.join();
    std::vector<u64> buf{k, (u64)zk_, (u64)public_inputs_.size(), (u64)gate_rows_};
    buf.insert(buf.end(), col_digest.begin(), col_digest.end());
    u64 acc[4]; hash_no_pad(buf.data(), buf.size(), acc);
    memcpy(d.circuit_digest, acc, sizeof acc); }
  lapb("digest");
  // build-time indexes are not needed to generate witnesses
  std::vector<Target>().swap(wire_of_); arith_cache_ = {}; out_phase_ = {}; open_arith_ = {}; consts_ = {}; const_val_ = {}; std::vector<uint32_t>().swap(parent_); std::vector<Target>().swap(wire_targets_);
  built_ = true;
  lapb("free");
}

// Eight ChaCha8 blocks of csrc/salt.h's stream at once (GCC vector extensions; the same keystream as salt_block, block by block).
typedef uint32_t v8u __attribute__((vector_size(32)));
#define rotl8(x, r) (((x) << (r)) | ((x) >> (32 - (r))))
#define QPC_QR(a, b, c, d) do { a += b; d ^= a; d = rotl8(d, 16); c += d; b ^= c; b = rotl8(b, 12); a += b; d ^= a; d = rotl8(d, 8); c += d; b ^= c; b = rotl8(b, 7); } while (0)
__attribute__((target("avx2"))) static void salt_blocks8_avx2(const SaltKey& key, u64 nonce, u64 block0, u64* out) {
  v8u s[16], x[16];
  const uint32_t c[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
  for (int i = 0; i < 4; i++) s[i] = v8u{c[i], c[i], c[i], c[i], c[i], c[i], c[i], c[i]};
  for (int i = 0; i < 8; i++) s[4 + i] = v8u{key.k[i], key.k[i], key.k[i], key.k[i], key.k[i], key.k[i], key.k[i], key.k[i]};
  for (int l = 0; l < 8; l++) { const u64 b = block0 + l; s[12][l] = (uint32_t)b; s[13][l] = (uint32_t)(b >> 32); }
  s[14] = v8u{} + (uint32_t)nonce; s[15] = v8u{} + (uint32_t)(nonce >> 32);
  for (int i = 0; i < 16; i++) x[i] = s[i];
  for (int r = 0; r < 4; r++) {
    QPC_QR(x[0], x[4], x[8], x[12]); QPC_QR(x[1], x[5], x[9], x[13]); QPC_QR(x[2], x[6], x[10], x[14]); QPC_QR(x[3], x[7], x[11], x[15]);
    QPC_QR(x[0], x[5], x[10], x[15]); QPC_QR(x[1], x[6], x[11], x[12]); QPC_QR(x[2], x[7], x[8], x[13]); QPC_QR(x[3], x[4], x[9], x[14]);
  }
  for (int i = 0; i < 16; i++) x[i] += s[i];
  for (int l = 0; l < 8; l++) for (int i = 0; i < 8; i++) {
    const u64 z = (u64)x[2 * i][l] | ((u64)x[2 * i + 1][l] << 32);
    out[8 * l + i] = z >= GL_P ? z - GL_P : z;
  }
}
static void salt_blocks8(const SaltKey& key, u64 nonce, u64 block0, u64* out) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) { salt_blocks8_avx2(key, nonce, block0, out); return; }
  for (int l = 0; l < 8; l++) salt_block(key, nonce, block0 + l, out + 8 * l);
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
std::string Builder::generate(const u64* inputs, const uint8_t* blinding_key, u64* wires, u64* public_inputs, int threads) const {
  const size_t n = n_; static const bool tm = getenv("QPC_TIMING") != nullptr; double t0 = now_ms();      // QPC_TIMING (debug): per-phase times on stderr
  auto lap = [&](const char* what) { if (tm) { double t = now_ms(); fprintf(stderr, "  %-12s %.3f ms\n", what, t - t0); t0 = t; } };
  const u64 UNSET = ~u64(0);                                          // not a field element: marks a class no generator has written yet
  std::vector<u64> val(ncls_, UNSET);
  std::string err;
  auto name = [&](Target t) { const Cell& c = cells_[t]; return c.row < 0 ? "virtual target " + std::to_string(t) : "wire (row " + std::to_string(c.row) + ", column " + std::to_string(c.col) + ")"; };
  auto set = [&](Target t, u64 v) {
    u64& slot = val[cls_[t]];
    if (__builtin_expect(slot != UNSET && slot != v, 0)) { if (err.empty()) err = "copy-constrained targets set twice with different values: " + name(t) + " was " + std::to_string(slot) + ", now " + std::to_string(v); return false; }
    slot = v; return true;
  };
  auto get = [&](Target t) -> u64 {
    const u64 v = val[cls_[t]];
    if (__builtin_expect(v == UNSET, 0)) { if (err.empty()) err = name(t) + " is read before any generator has set it"; return 0; }
    return v;
  };
  if (zk_ && !blinding_key) return "a zero-knowledge circuit needs a blinding key";
  // Zeroing the matrix and writing the zk blinding rows (ChaCha8 keystream, one stream per wire column) touch no class value: they run
  // beside the operation replay, column ranges on up to four helper threads.
  auto zero_and_blind = [&](int c_lo, int c_hi) {
    memset(wires + (size_t)c_lo * n, 0, sizeof(u64) * (size_t)(c_hi - c_lo) * n);
    if (!zk_) return;
    const SaltKey key = salt_key_from_bytes(blinding_key);
    for (int c = c_lo; c < c_hi; c++) {                               // column by column: sequential writes
      u64 blk[64]; u64 nblk = 0; int pos = 64;
      auto rnd = [&]() { if (pos == 64) { salt_blocks8(key, 0x626c696e6400ULL /* "blind" */ + (u64)c, nblk, blk); nblk += 8; pos = 0; } return blk[pos++]; };
      u64* col = wires + (size_t)c * n;
      for (size_t r = 0; r < blind_plain_; r++) col[blind_plain_first_ + r] = rnd();
      if (c < R) for (size_t p = 0; p < blind_pairs_; p++) { const u64 v = rnd(); col[blind_pair_first_ + 2 * p] = v; col[blind_pair_first_ + 2 * p + 1] = v; }
    }
  };
  struct Joiner { std::vector<std::thread> t; void join() { for (auto& x : t) if (x.joinable()) x.join(); } ~Joiner() { join(); } } bg;
  if (zk_) { const int nb = threads > 1 ? std::min(4, threads) : 1; for (int k = 0; k < nb; k++) bg.t.emplace_back(zero_and_blind, W * k / nb, W * (k + 1) / nb); }
  else zero_and_blind(0, W);
  for (size_t i = 0; i < inputs_.size(); i++) {
    if (inputs[i] >= P) return "input target " + std::to_string(i) + " is not a canonical field element";
    if (!set(inputs_[i], inputs[i])) return err;
  }
  // one replay worker: its own error, inverse cache and Poseidon traces; class values are shared (segments write disjoint classes, and
  // re-writing a class with the value it holds stores nothing)
  static const u64 SMALL_INV = 1024;
  static const std::vector<u64> small_inv = [] {                      // 1/1 .. 1/1024 by one inversion (Montgomery's trick)
    std::vector<u64> pre(SMALL_INV + 2, 1), out(SMALL_INV + 1, 0);
    for (u64 i = 1; i <= SMALL_INV; i++) pre[i + 1] = gl_mul(pre[i], i);
    u64 inv = gl_inv(pre[SMALL_INV + 1]);
    for (u64 i = SMALL_INV; i >= 1; i--) { out[i] = gl_mul(inv, pre[i]); inv = gl_mul(inv, i); }
    return out;
  }();
  struct Worker { std::string err; std::unordered_map<u64, u64> inv_cache; std::vector<u64> ptrace; };
  u64* const vals = val.data();
  auto exec = [&](const Op& op, Worker& wk) {
    auto wset = [&](Target t, u64 v) {
      u64* slot = vals + cls_[t]; const u64 cur = __atomic_load_n(slot, __ATOMIC_RELAXED);
      if (cur == v) return;
      if (__builtin_expect(cur != UNSET, 0)) { if (wk.err.empty()) wk.err = "copy-constrained targets set twice with different values: " + name(t) + " was " + std::to_string(cur) + ", now " + std::to_string(v); return; }
      __atomic_store_n(slot, v, __ATOMIC_RELAXED);
    };
    auto wget = [&](Target t) -> u64 {
      const u64 v = __atomic_load_n(vals + cls_[t], __ATOMIC_RELAXED);
      if (__builtin_expect(v == UNSET, 0)) { if (wk.err.empty()) wk.err = name(t) + " is read before any generator has set it"; return 0; }
      return v;
    };
    switch (op.kind) {
      case OP_CONST: wset(op.out, op.k0); break;
      case OP_ARITH: { const u64 a = wget(op.a), b = wget(op.b), c = wget(op.c); wset(op.out, gl_add(gl_mul(op.k0, gl_mul(a, b)), gl_mul(op.k1, c))); break; }
      case OP_INV: {                                                 // is_equal hints: the differences repeat (index - j over all nodes), so inverses are cached
        const u64 a = wget(op.a); u64 r = 0;
        if (a && a <= SMALL_INV) r = small_inv[a];                                    // differences of small indices (is_equal(j, index)): a table
        else if (a && P - a <= SMALL_INV) r = P - small_inv[P - a];
        else if (a) { auto it = wk.inv_cache.find(a); if (it != wk.inv_cache.end()) r = it->second; else { r = gl_inv(a); wk.inv_cache.emplace(a, r); } }
        wset(op.out, r); break;
      }
      case OP_SPLIT: {
        const u64 x = wget(op.out);
        if (x >> BASESUM_LIMBS) { if (wk.err.empty()) wk.err = "range check: " + std::to_string(x) + " does not fit " + std::to_string(BASESUM_LIMBS) + " binary limbs"; break; }
        for (int l = 0; l < BASESUM_LIMBS; l++) wset(op.out + 1 + l, (x >> l) & 1);
        break;
      }
      case OP_POSEIDON: {                                           // targets of the row: inputs out..out+11, swap out+12, outputs out+13..out+24
        u64 st[12]; for (int i = 0; i < 12; i++) st[i] = wget(op.out + i);
        const u64 swap = wget(op.out + 12);
        if (swap > 1) { if (wk.err.empty()) wk.err = "Poseidon gate row " + std::to_string(op.row) + ": swap wire is not a bit"; break; }
        if (swap) for (int i = 0; i < 4; i++) std::swap(st[i], st[i + 4]);
        wk.ptrace.resize(wk.ptrace.size() + 108); wk.ptrace[wk.ptrace.size() - 108] = op.row; wk.ptrace[wk.ptrace.size() - 107] = swap;
        poseidon_permute_traced(st, wk.ptrace.data() + wk.ptrace.size() - 106);      // the row's S-box input wires, kept for the fill below
        for (int i = 0; i < 12; i++) wset(op.out + 13 + i, st[i]);
        break;
      }
      case OP_LOHI: { const u64 x = wget(op.a); wset(op.out, x & 0xFFFFFFFFULL); wset(op.c, x >> 32); break; }
      case OP_EXTINV: {                                             // (a + b*sqrt7)^-1 = (a - b*sqrt7) / (a^2 - 7 b^2)
        const u64 a = wget(op.a), b = wget(op.b);
        const u64 norm = gl_sub(gl_sqr(a), gl_mul(7, gl_sqr(b)));
        const u64 ni = norm ? gl_inv(norm) : 0;
        wset(op.out, gl_mul(a, ni)); wset(op.c, gl_mul(gl_sub(0, b), ni));
        break;
      }
      default: special_eval(op, wget, wset); break;               // the recursive verifier's gate rows
    }
  };
  const int nthr = parallel_ok_ ? std::max(1, std::min(threads, 32)) : 1;
  std::vector<Worker> workers((size_t)nthr);
  static const bool opstat = getenv("QPC_OPSTAT") != nullptr; static double kind_ms[32]; static long kind_n[32];
  auto run_list = [&](const std::vector<uint32_t>& idx, Worker& wk) {
    if (opstat) { for (uint32_t i : idx) { const double t0 = now_ms(); exec(ops_[i], wk); kind_ms[ops_[i].kind] += now_ms() - t0; kind_n[ops_[i].kind]++; if (!wk.err.empty()) return; } return; }
    for (uint32_t i : idx) { exec(ops_[i], wk); if (!wk.err.empty()) return; } };
  auto run_stage = [&](const std::vector<std::vector<uint32_t>>& segs) {           // segments of one stage over the workers (dynamic hand-out)
    if (segs.empty()) return;
    std::atomic<size_t> next{0};
    auto body = [&](Worker& wk) { for (size_t k; (k = next.fetch_add(1)) < segs.size();) { run_list(segs[k], wk); if (!wk.err.empty()) return; } };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthr; t++) pool.emplace_back(body, std::ref(workers[(size_t)t]));
    body(workers[0]);
    for (auto& th : pool) th.join();
  };
  auto first_error = [&]() { for (auto& wk : workers) if (!wk.err.empty()) return wk.err; return std::string(); };
  if (nthr > 1) {
    run_list(const_ops_, workers[0]); run_list(pre_ops_, workers[0]);
    if (first_error().empty()) run_stage(seg_ops_[1]);
    if (first_error().empty()) run_stage(seg_ops_[2]);
    if (first_error().empty()) run_list(post_ops_, workers[0]);
  } else {
    for (const Op& op : ops_) { exec(op, workers[0]); if (!workers[0].err.empty()) break; }     // program order
  }
  if (opstat) { for (int k = 0; k < 32; k++) if (kind_n[k]) { fprintf(stderr, "  op kind %2d: %7ld ops %8.3f ms\n", k, kind_n[k], kind_ms[k]); kind_ms[k] = 0; kind_n[k] = 0; } }
  err = first_error();
  if (!err.empty()) return err;
  std::vector<u64> ptrace;
  for (auto& wk : workers) { if (ptrace.empty()) ptrace.swap(wk.ptrace); else ptrace.insert(ptrace.end(), wk.ptrace.begin(), wk.ptrace.end()); }
  for (size_t i = 0; i < public_inputs_.size(); i++) public_inputs[i] = get(public_inputs_[i]);
  if (!err.empty()) return err;
  lap("ops");

  bg.join();                                                          // the matrix is zeroed (and its blinding rows written)
  lap("join");
  // the fill: routed (and special-row) cells from their classes, then the Poseidon rows' other wires — column ranges, so that with
  // several threads each writes its own part of the matrix (fill_dst_ is in column-major order)
  { const size_t m = fill_dst_.size(); const uint32_t* fd = fill_dst_.data(); const uint32_t* fc = fill_cls_.data(); const u64* vp = val.data();
    const size_t np = ptrace.size() / 108; const u64* pt = ptrace.data();
    auto fill_cells = [&](size_t lo, size_t hi) {
      for (size_t i = lo; i < hi; i++) { if (i + 24 < hi) __builtin_prefetch(vp + fc[i + 24]); const u64 v = vp[fc[i]]; wires[fd[i]] = v == UNSET ? 0 : v; }      // classes never set hold 0
    };
    auto fill_sbox = [&](int c_lo, int c_hi) {                         // the S-box inputs, column by column (sequential writes)
      for (int i = c_lo; i < c_hi; i++) { u64* col = wires + (size_t)(29 + i) * n; for (size_t r = 0; r < np; r++) col[(size_t)pt[r * 108]] = pt[r * 108 + 2 + i]; }
    };
    const int ft = threads > 1 ? std::min(threads, 8) : 1;
    if (ft > 1 && m > 65536) {
      std::vector<std::thread> pool;
      for (int t = 1; t < ft; t++) pool.emplace_back([&, t] { fill_cells(m * t / ft, m * (t + 1) / ft); fill_sbox(106 * t / ft, 106 * (t + 1) / ft); });
      fill_cells(0, m / ft); fill_sbox(0, 106 / ft);
      for (auto& th : pool) th.join();
    } else { fill_cells(0, m); fill_sbox(0, 106); }
    for (size_t r = 0; r < np; r++) if (pt[r * 108 + 1]) {             // delta = swap * (in[i+4] - in[i]) (after the routed cells are in place)
      const size_t row = (size_t)pt[r * 108];
      for (int i = 0; i < 4; i++) wires[(size_t)(25 + i) * n + row] = gl_sub(wires[(size_t)(4 + i) * n + row], wires[(size_t)i * n + row]);
    } }
  lap("fill");
  return "";
}

} // namespace qpc
