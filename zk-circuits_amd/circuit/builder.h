// Circuit builder + witness generator for the leaf gate set (Noop, Constant{2}, PublicInput, BaseSum<2>{63}, Arithmetic{20}, Poseidon) and the
// recursive verifier's gates (ArithmeticExtension, Reducing, ReducingExtension, RandomAccess, PoseidonMds, CosetInterpolation, Exponentiation;
// SURVEY.md App. D) under standard_recursion(_zk)_config: 135 wires, 80 routed, 2 constants per gate. Host-only C++.
//
// This is this repo's own builder: it offers the operations the reference's circuit definitions call on plonky2's CircuitBuilder
// (wormhole/circuit/src/{nullifier,unspendable_account,storage_proof/mod,substrate_account}.rs, common/src/gadgets.rs, voting/src/lib.rs)
// with the same meaning, and it places gates by its own rules. A circuit built here enforces the same statement as the reference's, with
// the same public inputs in the same order, but its rows / copy constraints / digest are NOT upstream's (qp-plonky2's builder is not on
// disk; gate placement is "parity unpinned"): proofs are checked by this repo's verifier data, not by the reference's verifier.bin.
//
// Model: a Target is a wire cell (row, column) or a virtual slot; connect() merges targets into copy classes (union-find). Every gate
// operand is a wire cell merged with the target it carries. Witness generation replays the operation list in build order over class
// values; writing a class twice with different values is the failure the reference's tests expect from bad inputs
// ("... set twice with different values", wormhole/tests/src/circuit/storage_proof_tests.rs:31).
#pragma once
#include "../../include/qpgpu.h"
#include "../csrc/poseidon.h"
#include "../csrc/salt.h"
#include "../csrc/hostcpus.h"
#include <map>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

namespace qpc {
using qp::u64;
typedef uint32_t Target;

enum RowKind : uint8_t { K_NOOP = 0, K_CONST = 1, K_PI = 2, K_BASESUM = 3, K_ARITH = 4, K_POSEIDON = 5, K_KINDS = 6,
                         // the recursive verifier's gates (layouts as csrc/plonk.hip and oracle/gates.cc evaluate them): used by
                         // the aggregation node circuit only, listed in a circuit's gate set only if it has such a row
                         K_ARITH_EXT = 6, K_MUL_EXT = 7, K_REDUCING = 8, K_REDUCING_EXT = 9, K_RANDOM_ACCESS = 10, K_POSEIDON_MDS = 11,
                         K_EXP = 12, K_COSET = 13, K_ALL_KINDS = 14 };
static const int W = 135, R = 80, BASESUM_LIMBS = 63, ARITH_OPS = 20, NSEL = 2, NC = 4;
// parameters under standard_recursion_config (135 wires / 80 routed / D = 2)
static const int ARITH_EXT_OPS = 10, MUL_EXT_OPS = 13, REDUCING_COEFFS = 43, REDUCING_EXT_COEFFS = 32, RA_BITS = 4, RA_COPIES = 4, RA_EXTRA = 2, RA_VEC = 16,
                 COSET_BITS = 4, COSET_NP = 16, COSET_DEG = 6, COSET_NI = (COSET_NP - 2) / (COSET_DEG - 1), EXP_BITS = 66;
static const int COSET_S_POINT = 1 + 2 * COSET_NP, COSET_S_VALUE = COSET_S_POINT + 2, COSET_S_INTER = COSET_S_VALUE + 2, COSET_S_SHIFTED = COSET_S_INTER + 4 * COSET_NI;
struct Ext2 { Target a, b; };                // an F_{p^2} value as two targets
static const u64 P = qp::GL_P;

struct Hash4 { Target e[4]; };
struct BoolT { Target t; };

// One Poseidon gate row's wires from its 12 inputs (wire layout of plonky2's PoseidonGate: inputs 0..11, outputs 12..23, swap 24, delta 25..28,
// first full rounds 29..64 (rounds 1-3), partial 65..86, second full rounds 87..134).
inline void poseidon_gate_row(u64* wires, size_t n, size_t row, const u64 in[12], u64 swap) {
  using namespace qp;
  auto Wr = [&](int col) -> u64& { return wires[(size_t)col * n + row]; };
  const u64* rc = POSEIDON_RC_HOST;
  u64 st[12];
  for (int i = 0; i < 12; i++) Wr(i) = in[i];
  Wr(24) = swap;
  for (int i = 0; i < 4; i++) {
    u64 delta = swap ? gl_sub(in[i + 4], in[i]) : 0; Wr(25 + i) = delta;
    st[i] = gl_add(in[i], delta); st[i + 4] = gl_sub(in[i + 4], delta);
  }
  for (int i = 8; i < 12; i++) st[i] = in[i];
  int round = 0;
  for (int r = 0; r < 4; r++, round++) {
    for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], rc[12 * round + i]);
    if (r) for (int i = 0; i < 12; i++) Wr(29 + 12 * (r - 1) + i) = st[i];
    for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
    mds_layer(st);
  }
  for (int r = 0; r < 22; r++, round++) {
    for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], rc[12 * round + i]);
    Wr(65 + r) = st[0]; st[0] = sbox7(st[0]);
    mds_layer(st);
  }
  for (int r = 0; r < 4; r++, round++) {
    for (int i = 0; i < 12; i++) st[i] = gl_add(st[i], rc[12 * round + i]);
    for (int i = 0; i < 12; i++) Wr(87 + 12 * r + i) = st[i];
    for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
    mds_layer(st);
  }
  for (int i = 0; i < 12; i++) Wr(12 + i) = st[i];
}

class Builder {
 public:
  explicit Builder(bool zero_knowledge) : zk_(zero_knowledge) {
    add_row(K_PI, 0, 0);                                   // row 0: the PublicInput gate (wires 0..3 = hash of the public inputs)
    zero_ = constant(0); one_ = constant(1);
  }

  // ---- targets
  Target add_virtual() { return new_target(-1, 0); }
  Target add_input() { Target t = add_virtual(); inputs_.push_back(t); return t; }          // a PartialWitness slot, in allocation order
  std::vector<Target> add_inputs(int k) { std::vector<Target> v; for (int i = 0; i < k; i++) v.push_back(add_input()); return v; }
  Hash4 add_input_hash() { Hash4 h; for (auto& e : h.e) e = add_input(); return h; }
  void register_public_input(Target t) { public_inputs_.push_back(t); }
  Target add_public_input() { Target t = add_input(); register_public_input(t); return t; }
  Hash4 add_public_input_hash() { Hash4 h; for (auto& e : h.e) e = add_public_input(); return h; }

  Target constant(u64 c) {
    c %= P;
    auto it = consts_.find(c); if (it != consts_.end()) return it->second;
    if (const_slot_ == 0) const_row_ = add_row(K_CONST, 0, 0);
    Target t = wire(const_row_, const_slot_);
    (const_slot_ == 0 ? row_c0_ : row_c1_)[const_row_] = c;
    const_slot_ ^= 1;
    consts_[c] = t; const_val_[t] = c;
    rec(Op{OP_CONST, t, 0, 0, 0, c, 0, 0});
    return t;
  }
  Target zero() const { return zero_; }
  Target one() const { return one_; }
  BoolT constant_bool(bool b) { return BoolT{b ? one_ : zero_}; }

  void connect(Target a, Target b) { unite(a, b); }
  void connect_hashes(const Hash4& a, const Hash4& b) { for (int i = 0; i < 4; i++) connect(a.e[i], b.e[i]); }
  void assert_zero(Target t) { connect(t, zero_); }

  // ---- arithmetic: out = k0 * a * b + k1 * c on an Arithmetic gate slot (20 per row; a row's slots share (k0, k1))
  Target arithmetic(u64 k0, u64 k1, Target a, Target b, Target c) {
    using namespace qp;
    u64 va = 0, vb = 0, vc = 0;
    const bool ca = is_const(a, va), cb = is_const(b, vb), cc = is_const(c, vc);
    if (ca && cb && cc) return constant(gl_add(gl_mul(k0, gl_mul(va, vb)), gl_mul(k1, vc)));
    if (k0 == 0 || (ca && va == 0) || (cb && vb == 0)) {     // no product term
      if (cc) return constant(gl_mul(k1, vc));
      if (k1 == 1) return c;
    }
    if ((k1 == 0 || (cc && vc == 0)) && k0 == 1) {           // no addend, multiplication by one
      if (ca && va == 1) return b;
      if (cb && vb == 1) return a;
    }
    if (find(a) > find(b)) std::swap(a, b);                 // the product commutes: one cache entry per unordered pair
    const ArithKey key{k0, k1, find(a), find(b), find(c)};
    auto hit = arith_cache_.find(key);
    if (hit != arith_cache_.end()) {                        // reuse, unless that would tie two independent segments together (then: a second slot)
      const uint32_t ph = out_phase_[hit->second], me = phase_code();
      if (ph == 0 || ph == me || (cur_stage_ == 2 && (ph >> 28) == 1) || cur_stage_ == 0) return hit->second;
    }
    auto& open = open_arith_[std::make_pair(k0, k1)];
    if (open.second == 0) { open.first = add_row(K_ARITH, k0, k1); }
    const uint32_t row = open.first; const int s = open.second;
    open.second = (s + 1) % ARITH_OPS;
    unite(wire(row, 4 * s), a); unite(wire(row, 4 * s + 1), b); unite(wire(row, 4 * s + 2), c);
    Target out = wire(row, 4 * s + 3);
    rec(Op{OP_ARITH, out, wire(row, 4 * s), wire(row, 4 * s + 1), wire(row, 4 * s + 2), k0, k1, 0});
    arith_cache_[key] = out; out_phase_[out] = phase_code();
    return out;
  }
  Target add(Target a, Target b) { return arithmetic(1, 1, a, one_, b); }
  Target sub(Target a, Target b) { return arithmetic(1, P - 1, a, one_, b); }
  Target mul(Target a, Target b) { return arithmetic(1, 0, a, b, zero_); }
  Target mul_const(u64 k, Target a) { return arithmetic(k % P, 0, a, one_, zero_); }
  Target mul_add(Target a, Target b, Target c) { return arithmetic(1, 1, a, b, c); }
  BoolT not_(BoolT b) { return BoolT{sub(one_, b.t)}; }
  BoolT and_(BoolT a, BoolT b) { return BoolT{mul(a.t, b.t)}; }
  BoolT or_(BoolT a, BoolT b) { return BoolT{arithmetic(P - 1, 1, a.t, b.t, add(a.t, b.t))}; }       // a + b - ab
  Target select(BoolT b, Target x, Target y) { return mul_add(b.t, sub(x, y), y); }                  // b ? x : y

  // x == y as a boolean. One hint (the inverse of the difference); constraints: diff * inv = 1 - eq, diff * eq = 0.
  BoolT is_equal(Target x, Target y) {
    Target diff = sub(x, y);
    u64 vd;
    if (is_const(diff, vd)) return constant_bool(vd == 0);
    Target inv = add_virtual();
    rec(Op{OP_INV, inv, diff, 0, 0, 0, 0, 0});
    Target ne = mul(diff, inv);
    Target eq = sub(one_, ne);
    assert_zero(mul(diff, eq));
    return BoolT{eq};
  }

  // Hint: the inverse of a + b*sqrt(7) in F_{p^2} (0 for 0) as two virtual targets; the caller constrains it.
  void ext_inverse_hint(Target a, Target b, Target& ia, Target& ib) {
    ia = add_virtual(); ib = add_virtual();
    rec(Op{OP_EXTINV, ia, a, b, ib, 0, 0, 0});
  }
  // Hint: the low and high 32-bit halves of the canonical value of x; the caller constrains them.
  void lohi_hint(Target x, Target& lo, Target& hi) { lo = add_virtual(); hi = add_virtual(); rec(Op{OP_LOHI, lo, x, 0, hi, 0, 0, 0}); }
  // Hint: the inverse of a base-field target (0 for 0); the caller constrains it.
  Target inverse_hint(Target a) { Target inv = add_virtual(); rec(Op{OP_INV, inv, a, 0, 0, 0, 0, 0}); return inv; }

  // Little-endian bits of x, which must be < 2^nbits: one BaseSum<2> row (wire 0 = sum, wires 1..63 = limbs, each constrained to {0,1});
  // limbs from nbits up are tied to zero.
  std::vector<BoolT> split_le(Target x, int nbits) {
    const uint32_t row = add_row(K_BASESUM, 0, 0);
    unite(wire(row, 0), x);
    rec(Op{OP_SPLIT, wire(row, 0), 0, 0, 0, 0, 0, row});
    std::vector<BoolT> bits;
    for (int l = 0; l < BASESUM_LIMBS; l++) { Target t = wire(row, 1 + l); if (l < nbits) bits.push_back(BoolT{t}); else assert_zero(t); }
    return bits;
  }
  void range_check(Target x, int nbits) { split_le(x, nbits); }

  // Poseidon permutation on a gate row and the no-padding sponge (rate 8, overwrite mode) over it. With `swap` set the gate exchanges
  // inputs 0..3 with 4..7 before permuting (the Merkle-path form: the bit says on which side the running digest sits).
  void permute(Target st[12]) { permute_swap(st, BoolT{zero_}); }
  void permute_swap(Target st[12], BoolT swap) {
    const uint32_t row = add_row(K_POSEIDON, 0, 0);
    const Target first = wire(row, 0);                      // targets of a fresh row are numbered in creation order: inputs first .. first+11,
    for (int i = 0; i < 12; i++) unite(wire(row, i), st[i]);
    unite(wire(row, 24), swap.t);                           // swap first+12,
    rec(Op{OP_POSEIDON, first, 0, 0, 0, 0, 0, row});
    for (int i = 0; i < 12; i++) st[i] = wire(row, 12 + i); // outputs first+13 .. first+24
  }
  Hash4 hash_n_to_hash_no_pad(const std::vector<Target>& in) {
    Target st[12]; for (auto& s : st) s = zero_;
    for (size_t off = 0; off < in.size(); off += 8) {
      for (size_t i = 0; i < 8 && off + i < in.size(); i++) st[i] = in[off + i];
      permute(st);
    }
    Hash4 h; for (int i = 0; i < 4; i++) h.e[i] = st[i];
    return h;
  }

  // ---- the recursive verifier's gates. Every wire of such a row is a target (row_target(row, col) = first target of the row + col);
  // the row's operation fills outputs and auxiliary wires from its inputs at witness generation.
  // ArithmeticExtension: out = k0 * m0 * m1 + k1 * ad in F_{p^2} (10 per row; a row's operations share (k0, k1)).
  Ext2 arith_ext(u64 k0, u64 k1, Ext2 m0, Ext2 m1, Ext2 ad) {
    auto& open = open_arith_ext_[std::make_pair(k0 % P, k1 % P)];
    if (open.second == 0) open.first = add_special_row(K_ARITH_EXT, k0 % P, k1 % P);
    const uint32_t row = open.first; const int s = open.second; open.second = (s + 1) % ARITH_EXT_OPS;
    unite(row_target(row, 8 * s), m0.a); unite(row_target(row, 8 * s + 1), m0.b); unite(row_target(row, 8 * s + 2), m1.a); unite(row_target(row, 8 * s + 3), m1.b);
    unite(row_target(row, 8 * s + 4), ad.a); unite(row_target(row, 8 * s + 5), ad.b);
    rec(Op{OP_ARITH_EXT, row_target(row, 0), 0, 0, 0, k0 % P, k1 % P, row, 0, 0, (uint32_t)s});
    return Ext2{row_target(row, 8 * s + 6), row_target(row, 8 * s + 7)};
  }
  // MulExtension: out = k0 * m0 * m1 in F_{p^2} (13 per row; the verifier here uses ArithmeticExtension throughout, this one is for completeness)
  Ext2 mul_ext(u64 k0, Ext2 m0, Ext2 m1) {
    auto& open = open_mul_ext_[k0 % P];
    if (open.second == 0) open.first = add_special_row(K_MUL_EXT, k0 % P, 0);
    const uint32_t row = open.first; const int s = open.second; open.second = (s + 1) % MUL_EXT_OPS;
    unite(row_target(row, 6 * s), m0.a); unite(row_target(row, 6 * s + 1), m0.b); unite(row_target(row, 6 * s + 2), m1.a); unite(row_target(row, 6 * s + 3), m1.b);
    rec(Op{OP_MUL_EXT, row_target(row, 0), 0, 0, 0, k0 % P, 0, row, 0, 0, (uint32_t)s});
    return Ext2{row_target(row, 6 * s + 4), row_target(row, 6 * s + 5)};
  }
  // Reducing / ReducingExtension: Horner evaluation sum_i coeff_i * alpha^(count-1-i) (acc = acc * alpha + coeff, from zero), chained over
  // as many rows as the coefficients need; the first row is padded with leading zero coefficients (upstream's reduce_base / reduce).
  Ext2 reduce_base(Ext2 alpha, const std::vector<Target>& coeffs) {
    std::vector<Ext2> c; for (Target t : coeffs) c.push_back(Ext2{t, zero_});
    return reduce_rows(alpha, c, false);
  }
  Ext2 reduce_ext(Ext2 alpha, const std::vector<Ext2>& coeffs) { return reduce_rows(alpha, coeffs, true); }
  // RandomAccess: list[index] for a 16-entry list and an index < 16 (the gate splits the index into bits of its own).
  Target random_access(Target index, const std::vector<Target>& list) {
    if (ra_slot_ == 0) ra_row_ = add_special_row(K_RANDOM_ACCESS, 0, 0);
    const uint32_t row = ra_row_; const int cp = ra_slot_; ra_slot_ = (cp + 1) % RA_COPIES;
    const int base = (2 + RA_VEC) * cp;
    unite(row_target(row, base), index);
    for (int i = 0; i < RA_VEC; i++) unite(row_target(row, base + 2 + i), i < (int)list.size() ? list[i] : zero_);
    rec(Op{OP_RANDOM_ACCESS, row_target(row, 0), 0, 0, 0, 0, 0, row, 0, 0, (uint32_t)cp});
    return row_target(row, base + 1);
  }
  // PoseidonMds: the MDS layer on 12 F_{p^2} values.
  void poseidon_mds(const Ext2 in[12], Ext2 out[12]) {
    const uint32_t row = add_special_row(K_POSEIDON_MDS, 0, 0);
    for (int i = 0; i < 12; i++) { unite(row_target(row, 2 * i), in[i].a); unite(row_target(row, 2 * i + 1), in[i].b); out[i] = Ext2{row_target(row, 24 + 2 * i), row_target(row, 25 + 2 * i)}; }
    rec(Op{OP_POSEIDON_MDS, row_target(row, 0), 0, 0, 0, 0, 0, row});
  }
  // CosetInterpolation: the value at `point` of the polynomial of degree < 16 through (shift * w16^i, values[i]).
  Ext2 coset_interpolate(Target shift, const Ext2 values[COSET_NP], Ext2 point) {
    const uint32_t row = add_special_row(K_COSET, 0, 0);
    unite(row_target(row, 0), shift);
    for (int i = 0; i < COSET_NP; i++) { unite(row_target(row, 1 + 2 * i), values[i].a); unite(row_target(row, 2 + 2 * i), values[i].b); }
    unite(row_target(row, COSET_S_POINT), point.a); unite(row_target(row, COSET_S_POINT + 1), point.b);
    rec(Op{OP_COSET, row_target(row, 0), 0, 0, 0, 0, 0, row});
    return Ext2{row_target(row, COSET_S_VALUE), row_target(row, COSET_S_VALUE + 1)};
  }
  // Exponentiation: base^(sum bits_le[i] 2^i), at most 66 bits (the gate's higher power bits are tied to zero).
  Target exponentiate(Target base, const std::vector<BoolT>& bits_le) {
    const uint32_t row = add_special_row(K_EXP, 0, 0);
    unite(row_target(row, 0), base);
    for (int i = 0; i < EXP_BITS; i++) unite(row_target(row, 1 + i), i < (int)bits_le.size() ? bits_le[i].t : zero_);
    rec(Op{OP_EXP, row_target(row, 0), 0, 0, 0, 0, 0, row});
    return row_target(row, 1 + EXP_BITS);
  }
  static const u64* coset_weights();          // barycentric weights of the order-16 subgroup (the CosetInterpolation gate's table)
  static const u64* coset_domain();

  // ---- independent parts of a circuit, for witness generation on several threads. Operations recorded between begin_segment(stage) and
  // end_segment() form one segment. Replay order: constants, the operations before the first segment, all stage-1 segments (in parallel), all
  // stage-2 segments (in parallel), the remaining unsegmented operations. A stage-1 segment may read what constants, inputs, the leading
  // operations and itself produced; a stage-2 segment also what any stage-1 segment produced. build() checks this on the recorded program and
  // falls back to in-order replay if it does not hold (the circuit is the same either way).
  void begin_segment(int stage) { cur_stage_ = (uint8_t)stage; cur_seg_ = seg_count_[stage]++; seen_segment_ = true; }
  void end_segment() { cur_stage_ = 0; cur_seg_ = 0; }

  // ---- finish: public-input hash, zk blinding rows, padding, sigma, descriptor. After this the builder is a circuit.
  void build();

  // ---- the circuit
  size_t degree() const { return n_; }
  size_t num_inputs() const { return inputs_.size(); }
  size_t num_gate_rows() const { return gate_rows_; }
  const qpgpu_circuit_desc& desc() const { return desc_; }
  void row_counts(uint32_t out[K_KINDS]) const { for (int k = 0; k < K_KINDS; k++) out[k] = 0; for (size_t r = 0; r < gate_rows_; r++) if (row_kind_[r] < K_KINDS) out[row_kind_[r]]++; }
  void row_counts_all(uint32_t out[K_ALL_KINDS]) const { for (int k = 0; k < K_ALL_KINDS; k++) out[k] = 0; for (size_t r = 0; r < gate_rows_; r++) out[row_kind_[r]]++; }
  // generate_partial_witness: inputs[num_inputs()] in allocation order -> wires [135][n] column-major, public_inputs[]. `blinding_key`
  // (32 bytes) seeds the zk blinding rows. Returns "" or the failure.
  std::string generate(const u64* inputs, const uint8_t* blinding_key, u64* wires, u64* public_inputs, int threads = 1) const;
  bool parallel_replay() const { return parallel_ok_; }
  size_t num_rows_so_far() const { return row_kind_.size(); }          // while building: rows allocated up to now (QPC_TIMING section reports)

  // ---- soundness testing (mutation.cc): hint operations = the values a generator supplies instead of a gate computing them
  // (OP_INV, OP_LOHI, OP_EXTINV, the limbs of OP_SPLIT), numbered in program order.
  struct MutationResult { uint32_t index, kind, mode, verdict; u64 conflicts, changed_classes, rows; };
  typedef int (*WitnessCheck)(void* ctx, const u64* wires, const u64* public_inputs, const uint32_t* rows, size_t num_rows, int full);
  size_t num_hints() const;
  std::string mutation_sweep(const u64* inputs, const uint8_t* blinding_key, uint32_t what, size_t first, size_t step, size_t max_count,
                             uint32_t modes_mask, u64 seed, uint32_t full_every, WitnessCheck check, void* ctx, std::vector<MutationResult>& out) const;

 private:
  enum OpKind : uint8_t { OP_CONST, OP_ARITH, OP_INV, OP_SPLIT, OP_POSEIDON, OP_EXTINV, OP_LOHI,
                          OP_ARITH_EXT, OP_REDUCING, OP_REDUCING_EXT, OP_RANDOM_ACCESS, OP_POSEIDON_MDS, OP_COSET, OP_EXP, OP_MUL_EXT };   // whole-row / slot operations of the special gates
  struct Op { uint8_t kind; Target out, a, b, c; u64 k0, k1; uint32_t row; uint8_t stage = 0; uint32_t seg = 0; uint32_t slot = 0; };
  // Targets a special-gate operation reads and writes (row_target(row, col) = op.out + col), and its evaluation: `get(target)` returns a value,
  // `set(target, value)` stores one. One definition for witness generation (builder.cc) and the soundness sweep (mutation.cc).
  template <class F> static void special_reads(const Op& op, F&& f);
  template <class F> static void special_writes(const Op& op, F&& f);
  template <class G, class S> static void special_eval(const Op& op, G&& get, S&& set);
  static bool is_special(uint8_t kind) { return kind >= OP_ARITH_EXT; }
  struct Cell { int32_t row; int16_t col; };

  void rec(Op op) { op.stage = cur_stage_; op.seg = cur_seg_; ops_.push_back(op); }
  Target new_target(int32_t row, int16_t col) { cells_.push_back(Cell{row, col}); parent_.push_back((uint32_t)parent_.size()); return (Target)(cells_.size() - 1); }
  Target wire(uint32_t row, int col) {
    Target& slot = wire_of_[(size_t)row * W + (size_t)col];
    if (slot == NO_TARGET) slot = new_target((int32_t)row, (int16_t)col);
    return slot;
  }
  uint32_t add_row(uint8_t kind, u64 c0, u64 c1) {
    row_kind_.push_back(kind); row_c0_.push_back(c0); row_c1_.push_back(c1); wire_of_.resize(wire_of_.size() + W, NO_TARGET);
    return (uint32_t)row_kind_.size() - 1;
  }
  // a special-gate row: all 135 wires become targets at once, numbered in column order
  uint32_t add_special_row(uint8_t kind, u64 c0, u64 c1) {
    const uint32_t row = add_row(kind, c0, c1);
    for (int col = 0; col < W; col++) wire(row, col);
    return row;
  }
  Target row_target(uint32_t row, int col) { return wire_of_[(size_t)row * W] + (Target)col; }
  Ext2 reduce_rows(Ext2 alpha, const std::vector<Ext2>& coeffs, bool ext) {
    const int per = ext ? REDUCING_EXT_COEFFS : REDUCING_COEFFS, cw = ext ? 2 : 1;
    const size_t rows = (coeffs.size() + per - 1) / per, pad = rows * per - coeffs.size();
    Ext2 acc{zero_, zero_};
    for (size_t r = 0; r < rows; r++) {
      const uint32_t row = add_special_row(ext ? K_REDUCING_EXT : K_REDUCING, 0, 0);
      unite(row_target(row, 2), alpha.a); unite(row_target(row, 3), alpha.b); unite(row_target(row, 4), acc.a); unite(row_target(row, 5), acc.b);
      for (int i = 0; i < per; i++) {
        const size_t k = r * per + i;
        const Ext2 c = k < pad ? Ext2{zero_, zero_} : coeffs[k - pad];
        unite(row_target(row, 6 + cw * i), c.a); if (ext) unite(row_target(row, 7 + cw * i), c.b);
      }
      rec(Op{(uint8_t)(ext ? OP_REDUCING_EXT : OP_REDUCING), row_target(row, 0), 0, 0, 0, 0, 0, row});
      acc = Ext2{row_target(row, 0), row_target(row, 1)};
    }
    return acc;
  }
  uint32_t find(uint32_t x) { while (parent_[x] != x) { parent_[x] = parent_[parent_[x]]; x = parent_[x]; } return x; }
  void unite(Target a, Target b) {
    uint32_t ra = find(a), rb = find(b); if (ra == rb) return;
    // a class keeps its constant, if any (constant folding looks it up through the representative)
    auto ca = const_val_.find(ra), cb = const_val_.find(rb);
    if (ra > rb) std::swap(ra, rb);
    parent_[rb] = ra;
    if (ca != const_val_.end()) const_val_[ra] = ca->second; else if (cb != const_val_.end()) const_val_[ra] = cb->second;
  }
  bool is_const(Target t, u64& v) { auto it = const_val_.find(find(t)); if (it == const_val_.end()) return false; v = it->second; return true; }

  bool zk_; bool built_ = false;
  std::vector<Cell> cells_; std::vector<uint32_t> parent_;
  static constexpr Target NO_TARGET = 0xFFFFFFFFu;
  std::vector<Target> wire_of_;                            // [row][column] -> target of that cell, NO_TARGET until it is used
  std::vector<uint8_t> row_kind_; std::vector<u64> row_c0_, row_c1_;
  std::vector<Op> ops_;
  std::vector<Target> inputs_, public_inputs_;
  std::map<u64, Target> consts_; std::map<uint32_t, u64> const_val_;
  uint32_t const_row_ = 0; int const_slot_ = 0;
  std::map<std::pair<u64, u64>, std::pair<uint32_t, int>> open_arith_, open_arith_ext_;
  std::map<u64, std::pair<uint32_t, int>> open_mul_ext_;
  uint32_t ra_row_ = 0; int ra_slot_ = 0;
  struct ArithKey { u64 k0, k1; uint32_t a, b, c; bool operator==(const ArithKey& o) const { return k0 == o.k0 && k1 == o.k1 && a == o.a && b == o.b && c == o.c; } };
  struct ArithHash { size_t operator()(const ArithKey& k) const { u64 h = k.k0 * 0x9E3779B97F4A7C15ULL ^ k.k1 * 0xC2B2AE3D27D4EB4FULL; h ^= ((u64)k.a << 32 | k.b) * 0x165667B19E3779F9ULL; h ^= (u64)k.c * 0x27D4EB2F165667C5ULL; return (size_t)(h ^ (h >> 29)); } };
  std::unordered_map<ArithKey, Target, ArithHash> arith_cache_;
  Target zero_, one_;
  uint8_t cur_stage_ = 0; uint32_t cur_seg_ = 0, seg_count_[3] = {0, 0, 0}; bool seen_segment_ = false;
  std::unordered_map<Target, uint32_t> out_phase_;         // arithmetic outputs: where they were recorded (0 leading, stage << 28 | 1 + segment, ~0 trailing)
  uint32_t phase_code() const { return cur_stage_ ? ((uint32_t)cur_stage_ << 28) | (1 + cur_seg_) : (seen_segment_ ? UINT32_MAX : 0); }
  // after build()
  bool parallel_ok_ = false;
  std::vector<uint32_t> const_ops_, pre_ops_, post_ops_; std::vector<std::vector<uint32_t>> seg_ops_[3];      // operation indices per replay phase
  size_t n_ = 0, gate_rows_ = 0, blind_plain_first_ = 0, blind_plain_ = 0, blind_pair_first_ = 0, blind_pairs_ = 0;
  std::vector<uint32_t> cls_; size_t ncls_ = 0;
  std::vector<Target> wire_targets_; std::vector<uint32_t> fill_dst_, fill_cls_;
  std::vector<u64> constants_sigmas_, k_is_;
  std::vector<qpgpu_gate> gates_; std::vector<uint32_t> sel_idx_, groups_;
  qpgpu_circuit_desc desc_{};
};

// ---- special-gate operations: layouts of SURVEY.md App. D as oracle/gates.cc and csrc/plonk.hip evaluate them
template <class F> void Builder::special_reads(const Op& op, F&& f) {
  const Target t0 = op.out;
  switch (op.kind) {
    case OP_ARITH_EXT: for (int j = 0; j < 6; j++) f(t0 + 8 * op.slot + j); break;
    case OP_MUL_EXT: for (int j = 0; j < 4; j++) f(t0 + 6 * op.slot + j); break;
    case OP_REDUCING: for (int j = 2; j < 6 + REDUCING_COEFFS; j++) f(t0 + j); break;
    case OP_REDUCING_EXT: for (int j = 2; j < 6 + 2 * REDUCING_EXT_COEFFS; j++) f(t0 + j); break;
    case OP_RANDOM_ACCESS: { const int base = (2 + RA_VEC) * (int)op.slot; f(t0 + base); for (int i = 0; i < RA_VEC; i++) f(t0 + base + 2 + i); break; }
    case OP_POSEIDON_MDS: for (int j = 0; j < 24; j++) f(t0 + j); break;
    case OP_COSET: for (int j = 0; j < COSET_S_VALUE; j++) f(t0 + j); break;
    case OP_EXP: for (int j = 0; j <= EXP_BITS; j++) f(t0 + j); break;
    default: break;
  }
}
template <class F> void Builder::special_writes(const Op& op, F&& f) {
  const Target t0 = op.out;
  switch (op.kind) {
    case OP_ARITH_EXT: f(t0 + 8 * op.slot + 6); f(t0 + 8 * op.slot + 7); break;
    case OP_MUL_EXT: f(t0 + 6 * op.slot + 4); f(t0 + 6 * op.slot + 5); break;
    case OP_REDUCING: f(t0); f(t0 + 1); for (int j = 6 + REDUCING_COEFFS; j < 6 + REDUCING_COEFFS + 2 * (REDUCING_COEFFS - 1); j++) f(t0 + j); break;
    case OP_REDUCING_EXT: f(t0); f(t0 + 1); for (int j = 6 + 2 * REDUCING_EXT_COEFFS; j < 6 + 2 * REDUCING_EXT_COEFFS + 2 * (REDUCING_EXT_COEFFS - 1); j++) f(t0 + j); break;
    case OP_RANDOM_ACCESS: { const int routed = (2 + RA_VEC) * RA_COPIES + RA_EXTRA; f(t0 + (2 + RA_VEC) * (int)op.slot + 1); for (int i = 0; i < RA_BITS; i++) f(t0 + routed + RA_BITS * (int)op.slot + i); break; }
    case OP_POSEIDON_MDS: for (int j = 24; j < 48; j++) f(t0 + j); break;
    case OP_COSET: for (int j = COSET_S_VALUE; j < COSET_S_SHIFTED + 2; j++) f(t0 + j); break;
    case OP_EXP: for (int j = 1 + EXP_BITS; j < 2 + 2 * EXP_BITS; j++) f(t0 + j); break;
    default: break;
  }
}
template <class G, class S> void Builder::special_eval(const Op& op, G&& get, S&& set) {
  using namespace qp;
  const Target t0 = op.out;
  struct E { u64 a, b; };
  auto mul = [](E x, E y) { return E{gl_add(gl_mul(x.a, y.a), gl_mul(7, gl_mul(x.b, y.b))), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))}; };
  auto add = [](E x, E y) { return E{gl_add(x.a, y.a), gl_add(x.b, y.b)}; };
  auto scale = [](E x, u64 k) { return E{gl_mul(x.a, k), gl_mul(x.b, k)}; };
  auto gx = [&](int col) { return E{get(t0 + col), get(t0 + col + 1)}; };
  auto sx = [&](int col, E v) { set(t0 + col, v.a); set(t0 + col + 1, v.b); };
  switch (op.kind) {
    case OP_ARITH_EXT: { const int b = 8 * (int)op.slot; sx(b + 6, add(scale(mul(gx(b), gx(b + 2)), op.k0), scale(gx(b + 4), op.k1))); break; }
    case OP_MUL_EXT: { const int b = 6 * (int)op.slot; sx(b + 4, scale(mul(gx(b), gx(b + 2)), op.k0)); break; }
    case OP_REDUCING: case OP_REDUCING_EXT: {
      const bool xc = op.kind == OP_REDUCING_EXT; const int nc = xc ? REDUCING_EXT_COEFFS : REDUCING_COEFFS, start_accs = 6 + (xc ? 2 : 1) * nc;
      const E alpha = gx(2); E acc = gx(4);
      for (int i = 0; i < nc; i++) {
        const E coeff = xc ? gx(6 + 2 * i) : E{get(t0 + 6 + i), 0};
        acc = add(mul(acc, alpha), coeff);
        sx(i == nc - 1 ? 0 : start_accs + 2 * i, acc);
      }
      break;
    }
    case OP_RANDOM_ACCESS: {
      const int base = (2 + RA_VEC) * (int)op.slot, routed = (2 + RA_VEC) * RA_COPIES + RA_EXTRA;
      const u64 idx = get(t0 + base);
      for (int i = 0; i < RA_BITS; i++) set(t0 + routed + RA_BITS * (int)op.slot + i, (idx >> i) & 1);
      set(t0 + base + 1, get(t0 + base + 2 + (int)(idx & (RA_VEC - 1))));          // an index >= 16 leaves the gate's index constraint violated
      break;
    }
    case OP_POSEIDON_MDS: {
      u64 a[12], b[12];
      for (int i = 0; i < 12; i++) { a[i] = get(t0 + 2 * i); b[i] = get(t0 + 2 * i + 1); }
      mds_layer(a); mds_layer(b);
      for (int i = 0; i < 12; i++) { set(t0 + 24 + 2 * i, a[i]); set(t0 + 25 + 2 * i, b[i]); }
      break;
    }
    case OP_COSET: {
      const u64* dom = coset_domain(); const u64* wgt = coset_weights();
      const u64 shift = get(t0);
      const E point = gx(COSET_S_POINT), xs = scale(point, shift ? gl_inv(shift) : 0);
      sx(COSET_S_SHIFTED, xs);
      E ev{0, 0}, pr{1, 0};
      int lo = 0, hi = COSET_DEG;
      for (int seg = 0; seg <= COSET_NI; seg++) {
        for (int i = lo; i < hi; i++) {
          const E term{gl_sub(xs.a, dom[i]), xs.b};
          ev = add(mul(ev, term), mul(scale(gx(1 + 2 * i), wgt[i]), pr)); pr = mul(pr, term);
        }
        if (seg == COSET_NI) break;
        sx(COSET_S_INTER + 2 * seg, ev); sx(COSET_S_INTER + 2 * (COSET_NI + seg), pr);
        lo = 1 + (COSET_DEG - 1) * (seg + 1); hi = lo + COSET_DEG - 1 < COSET_NP ? lo + COSET_DEG - 1 : COSET_NP;
      }
      sx(COSET_S_VALUE, ev);
      break;
    }
    case OP_EXP: {
      const u64 base = get(t0); u64 prev = 1, v = 1;
      for (int i = 0; i < EXP_BITS; i++) {
        const u64 bit = get(t0 + 1 + (EXP_BITS - 1 - i));
        v = gl_mul(prev, gl_add(gl_mul(bit, base), gl_sub(1, bit)));
        set(t0 + 2 + EXP_BITS + i, v); prev = gl_mul(v, v);
      }
      set(t0 + 1 + EXP_BITS, v);
      break;
    }
    default: break;
  }
}

} // namespace qpc
