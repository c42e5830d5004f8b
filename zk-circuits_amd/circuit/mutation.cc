// Soundness testing of the circuits this repo authors (the wormhole / vote circuits and the in-circuit verifier of an aggregation node):
// every value a witness generator HINTS instead of computing through a gate — base and extension inverses, 32-bit halves, the limbs of a
// BaseSum row — must be pinned by constraints, or a dishonest prover may pick it freely. Builder::mutation_sweep plays that prover: for one
// hint at a time it substitutes another value, re-derives everything downstream exactly as generate() would (later hints included, honestly),
// keeps the first value wherever two generators now disagree on a copy class (a dishonest prover cannot satisfy both), writes the resulting
// wire matrix and hands it to a checker supplied by the caller (tests pass the CPU oracle's gate-constraint check). A mutation is
//   killed    the checker rejects the matrix;
//   benign    the checker accepts it and no class other than the hint's own changed (a free value nothing depends on: is_equal's
//             inverse when the difference is zero);
//   SURVIVED  the checker accepts a matrix in which other values changed: an under-constrained hint.
// Input flips (what = 1) model the other cheat: after an honest generation, one input target's class is given another value and nothing
// is re-derived.
//
// The evaluator below restates Builder::generate's per-operation semantics without its caches; every sweep starts by checking that its
// honest matrix equals generate()'s bit for bit, and re-derives sampled mutations from scratch to check the incremental propagation.
// Reference statement this protects: an aggregation node verifies its children (wormhole/aggregator/src/circuits/tree.rs:117-121).
#include "builder.h"
#include <algorithm>
#include <cstring>

namespace qpc {
using namespace qp;

namespace {
const u64 UNSET = ~u64(0);
u64 splitmix(u64& s) { u64 z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
u64 felt(u64& s) { u64 v = splitmix(s); return v >= P ? v - P : v; }
}

size_t Builder::num_hints() const {
  size_t k = 0;
  for (const Op& op : ops_) if (op.kind == OP_INV || op.kind == OP_LOHI || op.kind == OP_EXTINV || op.kind == OP_SPLIT || is_special(op.kind)) k++;
  return k;
}

std::string Builder::mutation_sweep(const u64* inputs, const uint8_t* blinding_key, uint32_t what, size_t first, size_t step, size_t max_count,
                                    uint32_t modes_mask, u64 seed, uint32_t full_every, WitnessCheck check, void* ctx,
                                    std::vector<MutationResult>& out) const {
  const size_t n = n_;
  if (!check) return "no checker";
  if (step == 0) step = 1;
  // ---- what an operation reads and writes
  auto reads = [&](const Op& op, Target* in) -> int {
    switch (op.kind) {
      case OP_ARITH: in[0] = op.a; in[1] = op.b; in[2] = op.c; return 3;
      case OP_INV: case OP_LOHI: in[0] = op.a; return 1;
      case OP_EXTINV: in[0] = op.a; in[1] = op.b; return 2;
      case OP_SPLIT: in[0] = op.out; return 1;
      case OP_POSEIDON: for (int i = 0; i < 13; i++) in[i] = op.out + i; return 13;
      case OP_CONST: return 0;
      default: { int k = 0; special_reads(op, [&](Target t) { in[k++] = t; }); return k; }
    }
  };
  // evaluates `op` on class values `v`; writes (target, value) pairs, returns their number
  auto eval = [&](const Op& op, const u64* v, Target* wt, u64* wv) -> int {
    auto g = [&](Target t) { const u64 x = v[cls_[t]]; return x == UNSET ? 0 : x; };
    switch (op.kind) {
      case OP_CONST: wt[0] = op.out; wv[0] = op.k0; return 1;
      case OP_ARITH: wt[0] = op.out; wv[0] = gl_add(gl_mul(op.k0, gl_mul(g(op.a), g(op.b))), gl_mul(op.k1, g(op.c))); return 1;
      case OP_INV: { const u64 a = g(op.a); wt[0] = op.out; wv[0] = a ? gl_inv(a) : 0; return 1; }
      case OP_SPLIT: { const u64 x = g(op.out); for (int l = 0; l < BASESUM_LIMBS; l++) { wt[l] = op.out + 1 + l; wv[l] = (x >> l) & 1; } return BASESUM_LIMBS; }
      case OP_POSEIDON: {
        u64 st[12]; for (int i = 0; i < 12; i++) st[i] = g(op.out + i);
        if (g(op.out + 12)) for (int i = 0; i < 4; i++) std::swap(st[i], st[i + 4]);
        poseidon_permute(st);
        for (int i = 0; i < 12; i++) { wt[i] = op.out + 13 + i; wv[i] = st[i]; }
        return 12;
      }
      case OP_LOHI: { const u64 x = g(op.a); wt[0] = op.out; wv[0] = x & 0xFFFFFFFFULL; wt[1] = op.c; wv[1] = x >> 32; return 2; }
      case OP_EXTINV: {
        const u64 a = g(op.a), b = g(op.b), norm = gl_sub(gl_sqr(a), gl_mul(7, gl_sqr(b))), ni = norm ? gl_inv(norm) : 0;
        wt[0] = op.out; wv[0] = gl_mul(a, ni); wt[1] = op.c; wv[1] = gl_mul(gl_sub(0, b), ni); return 2;
      }
      default: {                                           // the recursive verifier's gate rows: one definition with generate() (builder.h special_eval);
        int k = 0;                                         // a later write to the same target replaces the earlier one, reads see the row's own writes
        special_eval(op, [&](Target t) { for (int j = k; j-- > 0;) if (wt[j] == t) return wv[j]; return g(t); },
                     [&](Target t, u64 x) { wt[k] = t; wv[k] = x; k++; });
        return k;
      }
    }
    return 0;
  };
  // ---- first writer of every class in replay order (inputs and constants first)
  const int64_t NOBODY = INT64_MAX;
  std::vector<int64_t> writer(ncls_, NOBODY);
  for (Target t : inputs_) writer[cls_[t]] = -1;
  for (const Op& op : ops_) if (op.kind == OP_CONST) writer[cls_[op.out]] = -1;
  Target wt[192]; u64 wv[192]; Target rd[192];
  { std::vector<u64> dummy(ncls_, 0);
    for (size_t i = 0; i < ops_.size(); i++) {
      if (ops_[i].kind == OP_CONST) continue;
      const int k = eval(ops_[i], dummy.data(), wt, wv);
      for (int j = 0; j < k; j++) if (writer[cls_[wt[j]]] == NOBODY) writer[cls_[wt[j]]] = (int64_t)i;
    } }
  // full replay in that order; keeps the first value where writers disagree; `forced` (optional) replaces the outputs of operation `fop`
  auto replay = [&](std::vector<u64>& v, int64_t fop, int fk, const Target* ft, const u64* fv, u64& conflicts) {
    v.assign(ncls_, UNSET); conflicts = 0;
    auto put = [&](Target t, u64 x) { u64& s = v[cls_[t]]; if (s == UNSET) s = x; else if (s != x) conflicts++; };
    for (size_t i = 0; i < inputs_.size(); i++) put(inputs_[i], inputs[i]);
    for (const Op& op : ops_) if (op.kind == OP_CONST) put(op.out, op.k0);
    for (size_t i = 0; i < ops_.size(); i++) {
      if (ops_[i].kind == OP_CONST) continue;
      if ((int64_t)i == fop) { for (int j = 0; j < fk; j++) put(ft[j], fv[j]); continue; }
      const int k = eval(ops_[i], v.data(), wt, wv);
      for (int j = 0; j < k; j++) put(wt[j], wv[j]);
    }
  };
  // class values -> matrix (routed cells from classes; Poseidon rows' other wires recomputed from the row's inputs). `m` holds the blinding rows.
  auto poseidon_row = [&](u64* m, const Op& op, const u64* v) {
    u64 in[12]; for (int i = 0; i < 12; i++) { const u64 x = v[cls_[op.out + i]]; in[i] = x == UNSET ? 0 : x; }
    const u64 sw = v[cls_[op.out + 12]];
    poseidon_gate_row(m, n, op.row, in, sw == UNSET ? 0 : sw);
    for (int i = 0; i < 12; i++) { const u64 x = v[cls_[op.out + 13 + i]]; m[(size_t)(12 + i) * n + op.row] = x == UNSET ? 0 : x; }
    m[(size_t)24 * n + op.row] = sw == UNSET ? 0 : sw;
  };
  auto fill_all = [&](u64* m, const u64* v) {
    for (const Op& op : ops_) if (op.kind == OP_POSEIDON) poseidon_row(m, op, v);
    for (size_t i = 0; i < fill_dst_.size(); i++) { const u64 x = v[fill_cls_[i]]; m[fill_dst_[i]] = x == UNSET ? 0 : x; }
  };
  const size_t npi = public_inputs_.size();
  auto pis_of = [&](const u64* v, u64* pis) { for (size_t i = 0; i < npi; i++) { const u64 x = v[cls_[public_inputs_[i]]]; pis[i] = x == UNSET ? 0 : x; } };

  // ---- the honest witness: generate()'s matrix, and this evaluator's must equal it
  std::vector<u64> m0((size_t)W * n), pis0(npi ? npi : 1);
  { const std::string e = generate(inputs, blinding_key, m0.data(), pis0.data(), 1); if (!e.empty()) return "honest witness: " + e; }
  std::vector<u64> val0; u64 c0 = 0;
  replay(val0, -1, 0, nullptr, nullptr, c0);
  if (c0) return "the honest replay met " + std::to_string(c0) + " conflicts";
  std::vector<u64> mw(m0);
  fill_all(mw.data(), val0.data());
  if (memcmp(mw.data(), m0.data(), mw.size() * sizeof(u64)) != 0) return "the sweep's evaluator and generate() disagree on the honest witness";
  { std::vector<u64> p(npi ? npi : 1); pis_of(val0.data(), p.data()); if (npi && memcmp(p.data(), pis0.data(), npi * 8) != 0) return "public inputs differ between the evaluator and generate()"; }
  { std::vector<uint32_t> none; if (check(ctx, m0.data(), pis0.data(), none.data(), 0, 1) != 0) return "the checker rejects the honest witness"; }

  // ---- indexes: class -> cells, Poseidon operations
  std::vector<uint32_t> cstart(ncls_ + 1, 0), ccell(fill_dst_.size());
  for (uint32_t c : fill_cls_) cstart[c + 1]++;
  for (size_t c = 0; c < ncls_; c++) cstart[c + 1] += cstart[c];
  { std::vector<uint32_t> pos(cstart.begin(), cstart.end() - 1); for (size_t i = 0; i < fill_dst_.size(); i++) ccell[pos[fill_cls_[i]]++] = fill_dst_[i]; }
  std::vector<uint32_t> pops; for (size_t i = 0; i < ops_.size(); i++) if (ops_[i].kind == OP_POSEIDON) pops.push_back((uint32_t)i);
  std::vector<uint32_t> hint_ops;
  // hint-bearing operations: the classic hints, and every operation of a special gate (its outputs and auxiliary wires — accumulators,
  // RandomAccess bits, the shifted interpolation point, intermediates — are generator-supplied and pinned by the row's own constraints)
  for (size_t i = 0; i < ops_.size(); i++) { const uint8_t k = ops_[i].kind; if (k == OP_INV || k == OP_LOHI || k == OP_EXTINV || k == OP_SPLIT || is_special(k)) hint_ops.push_back((uint32_t)i); }

  std::vector<u64> val(val0), pis(npi ? npi : 1), scratch;
  std::vector<uint8_t> dirty(ncls_, 0), rowmark(n, 0);
  std::vector<uint32_t> dirtied, rows;
  auto touch_row = [&](uint32_t r) { if (!rowmark[r]) { rowmark[r] = 1; rows.push_back(r); } };
  auto set_class = [&](uint32_t c, u64 x) { if (val[c] != x) { val[c] = x; if (!dirty[c]) { dirty[c] = 1; dirtied.push_back(c); } } };
  size_t serial = 0;

  auto finish = [&](MutationResult& res, const std::vector<uint32_t>& own, bool stale_poseidon, int64_t fop, int fk, const Target* ft, const u64* fv) -> std::string {
    // matrix: the cells of the changed classes, and the Poseidon rows that read one
    size_t other = 0;
    for (uint32_t c : dirtied) {
      if (std::find(own.begin(), own.end(), c) == own.end()) other++;
      for (uint32_t i = cstart[c]; i < cstart[c + 1]; i++) touch_row(ccell[i] % (uint32_t)n);
    }
    if (!stale_poseidon) for (uint32_t oi : pops) {
      const Op& op = ops_[oi]; bool hit = false;
      for (int i = 0; i < 13 && !hit; i++) hit = dirty[cls_[op.out + i]];
      if (hit) { poseidon_row(mw.data(), op, val.data()); touch_row(op.row); }
    }
    for (uint32_t c : dirtied) for (uint32_t i = cstart[c]; i < cstart[c + 1]; i++) mw[ccell[i]] = val[c];
    pis_of(val.data(), pis.data());
    if (npi && memcmp(pis.data(), pis0.data(), npi * sizeof(u64)) != 0) touch_row(0);      // row 0 is the PublicInput gate: it compares its wires with the hash of these
    res.changed_classes = other; res.rows = rows.size();
    const bool full = full_every && (serial % full_every) == 0;
    serial++;
    std::string err;
    if (full && !stale_poseidon) {                       // the same mutation from scratch: full replay, full fill
      std::vector<u64> v2; u64 c2 = 0; replay(v2, fop, fk, ft, fv, c2);
      scratch = m0; fill_all(scratch.data(), v2.data());
      if (memcmp(scratch.data(), mw.data(), mw.size() * sizeof(u64)) != 0) err = "incremental propagation and full replay disagree (mutation " + std::to_string(res.index) + " mode " + std::to_string(res.mode) + ")";
      else if ((c2 == 0) != (res.conflicts == 0)) err = "conflict counts disagree between incremental propagation and full replay";
    }
    if (rows.empty()) res.verdict = res.conflicts ? 6 : 5;                   // nothing moved: tied to a class that is already fixed / no cell carries the value
    else {
      const int rc = err.empty() ? check(ctx, mw.data(), pis.data(), rows.data(), rows.size(), full ? 1 : 0) : 0;
      if (rc < 0) err = "the checker failed";
      const bool same_pis = npi == 0 || memcmp(pis.data(), pis0.data(), npi * sizeof(u64)) == 0;
      if (rc != 0) res.verdict = 0;
      else if (res.conflicts) res.verdict = 3;                               // a conflict must show as a violated gate: harness or checker is wrong
      else res.verdict = (other == 0 && same_pis) ? 1 : 2;
    }
    // restore
    for (uint32_t r : rows) { for (int col = 0; col < W; col++) mw[(size_t)col * n + r] = m0[(size_t)col * n + r]; rowmark[r] = 0; }
    rows.clear();
    for (uint32_t c : dirtied) { val[c] = val0[c]; dirty[c] = 0; }
    dirtied.clear();
    return err;
  };

  if (what == 1) {                                        // ---- input flips after generation
    for (size_t idx = first; idx < inputs_.size() && out.size() < max_count; idx += step) for (uint32_t mode = 0; mode < 2; mode++) {
      if (!(modes_mask >> mode & 1)) continue;
      u64 s = seed ^ (idx * 0x100000001B3ULL + mode);
      const uint32_t c = cls_[inputs_[idx]];
      u64 nv = mode == 0 ? gl_add(val0[c], 1) : felt(s);
      if (nv == val0[c]) nv = gl_add(nv, 1);
      MutationResult res{(uint32_t)idx, 100, mode, 0, 0, 0, 0};
      set_class(c, nv);
      const std::string e = finish(res, std::vector<uint32_t>{}, true, -1, 0, nullptr, nullptr);
      if (!e.empty()) return e;
      out.push_back(res);
    }
    return "";
  }

  // ---- hint mutations
  for (size_t h = first; h < hint_ops.size() && out.size() < max_count; h += step) {
    const uint32_t k = hint_ops[h]; const Op& hop = ops_[k];
    Target ht[192]; u64 hv[192];
    const int hk = eval(hop, val0.data(), ht, hv);                          // the honest outputs
    const uint32_t nmodes = hop.kind == OP_INV ? 3 : hop.kind == OP_LOHI ? 5 : hop.kind == OP_EXTINV ? 4 : 2;
    if (hk == 0) continue;
    for (uint32_t mode = 0; mode < nmodes; mode++) {
      if (!(modes_mask >> mode & 1)) continue;
      u64 s = seed ^ (h * 0x100000001B3ULL + mode * 0x9E37ULL);
      u64 mv[192]; memcpy(mv, hv, sizeof(u64) * (size_t)hk);
      bool skip = false;
      switch (hop.kind) {
        case OP_INV:
          if (mode == 0) mv[0] = hv[0] ? 0 : 1; else if (mode == 1) mv[0] = gl_add(hv[0], 1); else { mv[0] = felt(s); if (mv[0] == hv[0]) mv[0] = gl_add(mv[0], 2); }
          break;
        case OP_LOHI: {
          const u64 lo = hv[0], hi = hv[1], x = lo | (hi << 32);
          if (mode == 0) mv[0] = gl_add(lo, 1);
          else if (mode == 1) mv[1] = gl_add(hi, 1);
          else if (mode == 2) { mv[0] = splitmix(s) & 0xFFFFFFFFULL; mv[1] = splitmix(s) & 0xFFFFFFFFULL; if (mv[0] == lo && mv[1] == hi) mv[0] ^= 1; }
          else if (mode == 3) { if (x > ~u64(0) - P) skip = true; else { const u64 y = x + P; mv[0] = y & 0xFFFFFFFFULL; mv[1] = y >> 32; } }   // the non-canonical pair of x + p
          else { if (hi == 0) skip = true; else { mv[0] = lo + (u64(1) << 32); mv[1] = hi - 1; } }                                           // same sum, low half out of range
          break;
        }
        case OP_EXTINV:
          if (mode == 0) { mv[0] = mv[1] = 0; if (hv[0] == 0 && hv[1] == 0) mv[0] = 1; }
          else if (mode == 1) mv[0] = gl_add(hv[0], 1);
          else if (mode == 2) mv[1] = gl_add(hv[1], 1);
          else { mv[0] = felt(s); mv[1] = felt(s); if (mv[0] == hv[0] && mv[1] == hv[1]) mv[0] = gl_add(mv[0], 1); }
          break;
        case OP_SPLIT: { const int l = (int)(splitmix(s) % BASESUM_LIMBS); mv[l] = mode == 0 ? 1 - hv[l] : gl_add(hv[l], 2); break; }   // another bit / not a bit at all
        default: {                                            // a special gate's row: one of the wires its generator writes, +1 / a random value
          const int l = mode == 0 ? 0 : (int)(splitmix(s) % (u64)hk);
          if (mode == 0) mv[l] = gl_add(hv[l], 1); else { mv[l] = felt(s); if (mv[l] == hv[l]) mv[l] = gl_add(mv[l], 1); }
          break;
        }
      }
      if (skip) { out.push_back(MutationResult{(uint32_t)h, hop.kind, mode, 4, 0, 0, 0}); continue; }
      MutationResult res{(uint32_t)h, hop.kind, mode, 0, 0, 0, 0};
      std::vector<uint32_t> own;
      for (int j = 0; j < hk; j++) {
        const uint32_t c = cls_[ht[j]];
        if (mv[j] != hv[j]) own.push_back(c);
        if (writer[c] == (int64_t)k) set_class(c, mv[j]); else if (val[c] != mv[j]) res.conflicts++;
      }
      for (size_t i = (size_t)k + 1; i < ops_.size(); i++) {               // forward propagation in replay order
        const Op& op = ops_[i];
        if (op.kind == OP_CONST) continue;
        const int nr = reads(op, rd);
        bool hot = false; for (int j = 0; j < nr && !hot; j++) hot = dirty[cls_[rd[j]]];
        if (hot) {
          const int nw = eval(op, val.data(), wt, wv);
          for (int j = 0; j < nw; j++) { const uint32_t c = cls_[wt[j]]; if (writer[c] == (int64_t)i) set_class(c, wv[j]); else if (val[c] != wv[j]) res.conflicts++; }
        } else {                                                            // same outputs as in the honest run: a conflict iff its class moved
          int nw; Target first_t;
          switch (op.kind) { case OP_SPLIT: nw = BASESUM_LIMBS; first_t = op.out + 1; break; case OP_POSEIDON: nw = 12; first_t = op.out + 13; break;
                             case OP_LOHI: case OP_EXTINV: nw = -2; first_t = op.out; break;
                             case OP_ARITH: case OP_INV: nw = 1; first_t = op.out; break; default: nw = -3; first_t = op.out; }
          auto chk = [&](Target t) { const uint32_t c = cls_[t]; if (dirty[c] && writer[c] != (int64_t)i) res.conflicts++; };
          if (nw == -2) { chk(op.out); chk(op.c); }
          else if (nw == -3) special_writes(op, chk);
          else for (int j = 0; j < nw; j++) chk(first_t + j);
        }
      }
      const std::string e = finish(res, own, false, (int64_t)k, hk, ht, mv);
      if (!e.empty()) return e;
      out.push_back(res);
    }
  }
  return "";
}

} // namespace qpc
