// Synthetic circuit + witness generator (see include/qpsynth.h). Host-only C++.
#include "../../include/qpsynth.h"
#include "../csrc/poseidon.h"
#include "../csrc/salt.h"
#include <vector>
#include <string>
#include <cstring>
#include <algorithm>

using namespace qp;

namespace {
thread_local std::string g_err;
int fail(const char* m) { g_err = m; return QPGPU_ERR_ARG; }

struct Rng {  // SplitMix64
  u64 s;
  explicit Rng(u64 seed) : s(seed) {}
  u64 next() { u64 z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
  u64 felt() { u64 v = next(); return v >= GL_P ? v - GL_P : v; }
  u64 below(u64 n) { return next() % n; }
};

enum RowKind : uint8_t { K_NOOP = 0, K_CONST = 1, K_PI = 2, K_BASESUM = 3, K_ARITH = 4, K_POSEIDON = 5,
                         // recursive-verifier gate set (qpsynth_params.n_recursion[])
                         K_ARITH_EXT = 6, K_MUL_EXT = 7, K_REDUCING = 8, K_REDUCING_EXT = 9, K_RANDOM_ACCESS = 10, K_POSEIDON_MDS = 11,
                         K_EXP = 12, K_COSET = 13, K_KINDS = 14 };
const int W = 135, R = 80, NPI = 16;
// per kind: serializer tag, parameters under standard_recursion_config (135 wires / 80 routed / D = 2), constraint degree
struct KindInfo { uint32_t tag; uint64_t param; uint32_t p2, p3; int degree; };
const KindInfo KINDS[K_KINDS] = {
  {QPGPU_GATE_NOOP, 0, 0, 0, 0}, {QPGPU_GATE_CONSTANT, 2, 0, 0, 1}, {QPGPU_GATE_PUBLIC_INPUT, 0, 0, 0, 1}, {QPGPU_GATE_BASESUM2, 63, 0, 0, 2},
  {QPGPU_GATE_ARITHMETIC, 20, 0, 0, 3}, {QPGPU_GATE_POSEIDON, 0, 0, 0, 7},
  {QPGPU_GATE_ARITHMETIC_EXTENSION, 10, 0, 0, 3}, {QPGPU_GATE_MUL_EXTENSION, 13, 0, 0, 3}, {QPGPU_GATE_REDUCING, 43, 0, 0, 2},
  {QPGPU_GATE_REDUCING_EXTENSION, 32, 0, 0, 2}, {QPGPU_GATE_RANDOM_ACCESS, 4, 4, 2, 5}, {QPGPU_GATE_POSEIDON_MDS, 0, 0, 0, 1},
  {QPGPU_GATE_EXPONENTIATION, 66, 0, 0, 4}, {QPGPU_GATE_COSET_INTERPOLATION, 4, 6, 0, 6}};
// second parameter set (qpsynth_params.reserved = 1): smaller gates, so that the evaluators' parametrisation is exercised
const KindInfo KINDS_ALT[K_KINDS] = {
  {QPGPU_GATE_NOOP, 0, 0, 0, 0}, {QPGPU_GATE_CONSTANT, 2, 0, 0, 1}, {QPGPU_GATE_PUBLIC_INPUT, 0, 0, 0, 1}, {QPGPU_GATE_BASESUM2, 63, 0, 0, 2},
  {QPGPU_GATE_ARITHMETIC, 20, 0, 0, 3}, {QPGPU_GATE_POSEIDON, 0, 0, 0, 7},
  {QPGPU_GATE_ARITHMETIC_EXTENSION, 7, 0, 0, 3}, {QPGPU_GATE_MUL_EXTENSION, 9, 0, 0, 3}, {QPGPU_GATE_REDUCING, 17, 0, 0, 2},
  {QPGPU_GATE_REDUCING_EXTENSION, 11, 0, 0, 2}, {QPGPU_GATE_RANDOM_ACCESS, 2, 5, 1, 3}, {QPGPU_GATE_POSEIDON_MDS, 0, 0, 0, 1},
  {QPGPU_GATE_EXPONENTIATION, 21, 0, 0, 4}, {QPGPU_GATE_COSET_INTERPOLATION, 3, 3, 0, 3}};

struct E2 { u64 a, b; };   // F_{p^2} = F_p[X]/(X^2 - 7)
inline E2 e_add(E2 x, E2 y) { return {gl_add(x.a, y.a), gl_add(x.b, y.b)}; }
inline E2 e_sub(E2 x, E2 y) { return {gl_sub(x.a, y.a), gl_sub(x.b, y.b)}; }
inline E2 e_mul(E2 x, E2 y) { return {gl_add(gl_mul(x.a, y.a), gl_mul(7, gl_mul(x.b, y.b))), gl_add(gl_mul(x.a, y.b), gl_mul(x.b, y.a))}; }
inline E2 e_scale(E2 x, u64 k) { return {gl_mul(x.a, k), gl_mul(x.b, k)}; }

struct Cell { uint32_t row; uint16_t col; };
} // namespace

struct qpsynth_circuit {
  qpsynth_params p;
  size_t n;
  std::vector<uint8_t> kind;                 // per row
  std::vector<u64> constants_sigmas;         // [84][n]
  std::vector<u64> k_is;
  std::vector<qpgpu_gate> gates; std::vector<uint32_t> sel_idx, groups;
  int nsel = 2, NC = 4;                      // selector columns, total constant columns (selectors + 2)
  int gate_index[K_KINDS];                   // position of each row kind's gate in the gate list (-1: absent)
  std::vector<u64> coset_weights, coset_dom;
  KindInfo kinds[K_KINDS];                   // gate parameters in use (KINDS or KINDS_ALT)
  // copy constraints, in application order: value[tgt] = value[src]
  std::vector<Cell> copy_src, copy_tgt;      // phase-2 (arith inputs) first, then phase-3 (free cells)
  size_t n_phase2 = 0;
  uint32_t row_pi, row_h0, row_h1;
  qpgpu_circuit_desc desc;
};

extern "C" {

const char* qpsynth_last_error(void) { return g_err.c_str(); }
// test hook: the host transcript permutation (fast form) or the literal round-by-round one, on a 12-word state
void qpsynth_host_permute(uint64_t* st, int generic) { if (generic) poseidon_permute_generic(st); else poseidon_permute(st); }

int qpsynth_preset(const char* name, qpsynth_params* o) {
  memset(o, 0, sizeof *o);
  std::string s(name);
  if (s == "A" || s == "B") {
    o->degree_bits = s == "A" ? 13 : 14; o->zero_knowledge = s == "B";
    o->n_poseidon = 488; o->n_basesum = 3800; o->n_arith = 2600; o->n_constant = 105; o->n_copies = 24000;
    o->fri_num_layers = s == "A" ? 2 : 3; for (int i = 0; i < 3; i++) o->fri_arity_bits[i] = 4;
    o->circuit_seed = 0x5157C1C0ULL + o->degree_bits;
  } else if (s == "V") {
    // voting-shaped: 34 Poseidon : 32 BaseSum : 40 Arithmetic per instance (voting/src/lib.rs:123-197), ~2^20/128 instances
    o->degree_bits = 20; o->zero_knowledge = 0; uint32_t inst = 8000;
    o->n_poseidon = 34 * inst; o->n_basesum = 32 * inst; o->n_arith = 40 * inst; o->n_constant = 2 * inst; o->n_copies = 200 * inst;
    o->fri_num_layers = 4; for (int i = 0; i < 4; i++) o->fri_arity_bits[i] = 4;
    o->circuit_seed = 0x5157C1C0ULL + 20;
  } else if (s == "T") {
    o->degree_bits = 6; o->zero_knowledge = 0; o->n_poseidon = 6; o->n_basesum = 12; o->n_arith = 16; o->n_constant = 3; o->n_copies = 150;
    o->fri_num_layers = 1; o->fri_arity_bits[0] = 4; o->circuit_seed = 0x5157C1C0ULL + 6;
  } else if (s == "TZ") {
    o->degree_bits = 7; o->zero_knowledge = 1; o->n_poseidon = 6; o->n_basesum = 12; o->n_arith = 16; o->n_constant = 3; o->n_copies = 150;
    o->fri_num_layers = 1; o->fri_arity_bits[0] = 4; o->circuit_seed = 0x5157C1C0ULL + 7;
  } else if (s == "R" || s == "RT" || s == "RT2") {
    // recursion-shaped: every gate of the in-circuit verifier (aggregate(), tree.rs:106-143) next to the leaf set
    const bool tiny = s != "R";
    if (s == "RT2") o->reserved = 1;                     // second gate-parameter set (KINDS_ALT)
    o->degree_bits = tiny ? 8 : 13; o->zero_knowledge = 0;
    uint32_t m = tiny ? 1 : 40;
    o->n_poseidon = 12 * m; o->n_basesum = 8 * m; o->n_arith = 12 * m; o->n_constant = 3 * m; o->n_copies = 300 * m;
    const uint32_t rec[8] = {20 * m, 12 * m, 20 * m, 12 * m, 16 * m, 6 * m, 6 * m, 8 * m};
    for (int i = 0; i < 8; i++) o->n_recursion[i] = rec[i];
    o->fri_num_layers = tiny ? 1 : 2; for (int i = 0; i < 2; i++) o->fri_arity_bits[i] = 4;
    o->circuit_seed = 0x5157C1C0ULL + 0x100 + o->degree_bits;
  } else return fail("unknown preset");
  return 0;
}

int qpsynth_circuit_create(const qpsynth_params* pp, qpsynth_circuit** out) {
  if (!pp || !out) return fail("null");
  auto* c = new qpsynth_circuit(); c->p = *pp;
  const qpsynth_params& p = c->p;
  if (p.degree_bits < 5 || p.degree_bits > 24) { delete c; return fail("degree_bits out of range"); }
  size_t n = c->n = size_t(1) << p.degree_bits;
  size_t n_rec = 0; for (int i = 0; i < 8; i++) n_rec += p.n_recursion[i];
  size_t used = 3 + (size_t)p.n_poseidon + p.n_basesum + p.n_arith + p.n_constant + n_rec;
  if (used > n) { delete c; return fail("gate rows exceed 2^degree_bits"); }
  { int sum = 0; for (uint32_t i = 0; i < p.fri_num_layers; i++) sum += p.fri_arity_bits[i]; if (p.fri_num_layers > 8 || sum > (int)p.degree_bits) { delete c; return fail("bad FRI arities"); } }
  Rng rng(p.circuit_seed);
  c->kind.assign(n, K_NOOP);
  size_t r = 0;
  c->row_pi = (uint32_t)r; c->kind[r++] = K_PI;
  c->row_h0 = (uint32_t)r; c->kind[r++] = K_POSEIDON;
  c->row_h1 = (uint32_t)r; c->kind[r++] = K_POSEIDON;
  for (uint32_t i = 0; i < p.n_poseidon; i++) c->kind[r++] = K_POSEIDON;
  for (uint32_t i = 0; i < p.n_basesum; i++) c->kind[r++] = K_BASESUM;
  for (uint32_t i = 0; i < p.n_arith; i++) c->kind[r++] = K_ARITH;
  for (uint32_t i = 0; i < p.n_constant; i++) c->kind[r++] = K_CONST;
  for (int k = 0; k < 8; k++) for (uint32_t i = 0; i < p.n_recursion[k]; i++) c->kind[r++] = (uint8_t)(K_ARITH_EXT + k);
  for (int k = 0; k < K_KINDS; k++) c->kinds[k] = (p.reserved == 1 ? KINDS_ALT : KINDS)[k];
  const KindInfo* KI = c->kinds;
  const int COSET_BITS = (int)KI[K_COSET].param, COSET_NP = 1 << COSET_BITS;
  // ---- gate list: kinds in use, ordered by constraint degree; selector groups as upstream forms them
  // (a group grows while group_size + next_gate_degree < max_degree = 8). The leaf set gives [0,5),[5,6) as in common.bin.
  {
    std::vector<int> kinds;
    for (int k = 0; k < K_KINDS; k++) if (k <= K_POSEIDON || (n_rec && p.n_recursion[k - K_ARITH_EXT])) kinds.push_back(k);
    std::stable_sort(kinds.begin(), kinds.end(), [KI](int a, int b) { return KI[a].degree < KI[b].degree; });
    for (int k = 0; k < K_KINDS; k++) c->gate_index[k] = -1;
    for (size_t i = 0; i < kinds.size(); i++) c->gate_index[kinds[i]] = (int)i;
    if (kinds.size() > 16) { delete c; return fail("too many gates"); }
    { u64 wg = gl_root(COSET_BITS), x = 1; for (int i = 0; i < COSET_NP; i++) { c->coset_dom.push_back(x); x = gl_mul(x, wg); } }
    for (int i = 0; i < COSET_NP; i++) { u64 d = 1; for (int j = 0; j < COSET_NP; j++) if (j != i) d = gl_mul(d, gl_sub(c->coset_dom[i], c->coset_dom[j])); c->coset_weights.push_back(gl_inv(d)); }
    size_t start = 0; c->sel_idx.assign(kinds.size(), 0);
    while (start < kinds.size()) {
      size_t size = 0;
      while (start + size < kinds.size() && (int)size + KI[kinds[start + size]].degree < 8) size++;
      for (size_t i = 0; i < size; i++) c->sel_idx[start + i] = (uint32_t)(c->groups.size() / 2);
      c->groups.push_back((uint32_t)start); c->groups.push_back((uint32_t)(start + size)); start += size;
    }
    for (int k : kinds) {
      qpgpu_gate g; memset(&g, 0, sizeof g); g.tag = KI[k].tag; g.param = KI[k].param; g.param2 = KI[k].p2; g.param3 = KI[k].p3;
      if (k == K_COSET) { g.table = c->coset_weights.data(); g.table_len = COSET_NP; }
      c->gates.push_back(g);
    }
    c->nsel = (int)(c->groups.size() / 2); c->NC = c->nsel + 2;
  }
  const int NC = c->NC, nsel = c->nsel;

  c->k_is.resize(R); { u64 x = 1; for (int j = 0; j < R; j++) { c->k_is[j] = x; x = gl_mul(x, GL_GEN); } }
  c->constants_sigmas.assign((size_t)(NC + R) * n, 0);
  u64* cs = c->constants_sigmas.data();
  const u64 UNUSED = 0xFFFFFFFFULL;
  for (size_t i = 0; i < n; i++) {
    int k = c->kind[i];
    const int gi = c->gate_index[k];
    for (int sc = 0; sc < nsel; sc++) cs[(size_t)sc * n + i] = (int)c->sel_idx[gi] == sc ? (u64)gi : UNUSED;
    u64* lc0 = &cs[(size_t)nsel * n + i]; u64* lc1 = &cs[(size_t)(nsel + 1) * n + i];
    if (k == K_ARITH) { u64 t = rng.below(4); *lc0 = t == 0 ? 1 : (t == 1 ? rng.below(1 << 16) : rng.felt()); *lc1 = rng.below(3) == 0 ? rng.felt() : 1; }
    if (k == K_CONST) { *lc0 = rng.felt(); *lc1 = rng.felt(); }
    if (k == K_ARITH_EXT || k == K_MUL_EXT) { *lc0 = rng.below(2) ? 1 : rng.felt(); *lc1 = rng.below(2) ? 1 : rng.felt(); }
    if (k == K_RANDOM_ACCESS) { *lc0 = rng.felt(); *lc1 = rng.felt(); }
  }
  // ---- copy constraints. sigma as a permutation over routed cells (col*n + row), built by cycle splicing.
  std::vector<uint32_t> sigma((size_t)R * n);
  for (size_t i = 0; i < sigma.size(); i++) sigma[i] = (uint32_t)i;
  std::vector<uint8_t> taken((size_t)R * n, 0);
  auto cell_id = [&](Cell x) { return (size_t)x.col * n + x.row; };
  auto splice = [&](Cell src, Cell tgt) { size_t a = cell_id(src), b = cell_id(tgt); sigma[b] = sigma[a]; sigma[a] = (uint32_t)b; taken[b] = 1; };
  // fixed wiring of the public-input hash: row_h0.out[8..12) -> row_h1.in[8..12); row_h1.out[0..4) -> PI gate wires
  for (int i = 0; i < 4; i++) { c->copy_src.push_back({c->row_h0, (uint16_t)(12 + 8 + i)}); c->copy_tgt.push_back({c->row_h1, (uint16_t)(8 + i)}); }
  for (int i = 0; i < 4; i++) { c->copy_src.push_back({c->row_h1, (uint16_t)(12 + i)}); c->copy_tgt.push_back({c->row_pi, (uint16_t)i}); }
  for (size_t k = 0; k < c->copy_src.size(); k++) splice(c->copy_src[k], c->copy_tgt[k]);
  // sources for phase 2: routed cells of Poseidon / BaseSum / Constant rows (excluding the PI-hash rows' inputs)
  size_t first_pos = 3, first_arith = 3 + (size_t)p.n_poseidon + p.n_basesum, end_arith = first_arith + p.n_arith;
  size_t first_bs = 3 + (size_t)p.n_poseidon;
  auto random_src12 = [&]() -> Cell {
    size_t row = first_pos + rng.below(p.n_poseidon + p.n_basesum);
    uint16_t col = c->kind[row] == K_POSEIDON ? (uint16_t)rng.below(25) : (uint16_t)rng.below(64);
    return Cell{(uint32_t)row, col};
  };
  size_t want = p.n_copies, n2 = p.n_arith ? want / 2 : 0;
  for (size_t k = 0; k < n2; k++) {  // arithmetic inputs as targets
    size_t row = first_arith + rng.below(p.n_arith); int op = (int)rng.below(20), which = (int)rng.below(3);
    Cell t{(uint32_t)row, (uint16_t)(4 * op + which)};
    if (taken[cell_id(t)]) continue;
    Cell s = random_src12();
    c->copy_src.push_back(s); c->copy_tgt.push_back(t); splice(s, t);
  }
  c->n_phase2 = c->copy_src.size();
  // phase 3: free cells (Noop rows: all routed wires unless zk blinding rows are left alone; BaseSum rows: wires 64..79;
  // Constant rows: 2..79) copy from any gate cell including arithmetic outputs.
  size_t first_noop = used;
  for (size_t k = 0; k < want - n2; k++) {
    Cell t;
    uint64_t pick = rng.below(3);
    if (pick == 0 && first_noop < n) t = Cell{(uint32_t)(first_noop + rng.below(n - first_noop)), (uint16_t)rng.below(R)};
    else if (pick == 1 && p.n_basesum) t = Cell{(uint32_t)(first_bs + rng.below(p.n_basesum)), (uint16_t)(64 + rng.below(16))};
    else if (p.n_constant) t = Cell{(uint32_t)(end_arith + rng.below(p.n_constant)), (uint16_t)(2 + rng.below(78))};
    else continue;
    if (taken[cell_id(t)]) continue;
    Cell s;
    if (p.n_arith && rng.below(2)) s = Cell{(uint32_t)(first_arith + rng.below(p.n_arith)), (uint16_t)(4 * rng.below(20) + 3)};
    else s = random_src12();
    c->copy_src.push_back(s); c->copy_tgt.push_back(t); splice(s, t);
  }
  // sigma values: k_is[col'] * w^row'
  {
    std::vector<u64> wpow(n); u64 w = gl_root((int)p.degree_bits), x = 1;
    for (size_t i = 0; i < n; i++) { wpow[i] = x; x = gl_mul(x, w); }
    for (int j = 0; j < R; j++) for (size_t i = 0; i < n; i++) { uint32_t t = sigma[(size_t)j * n + i]; cs[(size_t)(NC + j) * n + i] = gl_mul(c->k_is[t / n], wpow[t % n]); }
  }
  qpgpu_circuit_desc& d = c->desc; memset(&d, 0, sizeof d);
  d.degree_bits = p.degree_bits; d.num_wires = W; d.num_routed = R; d.num_constants = NC; d.num_challenges = 2;
  d.quotient_degree_factor = 8; d.num_partial_products = 9; d.num_public_inputs = NPI; d.num_gate_constraints = 123;
  d.zero_knowledge = p.zero_knowledge; d.fri_rate_bits = 3;
  d.fri_cap_height = p.fri_cap_height == 0xFFFFFFFFu ? 0 : (p.fri_cap_height ? p.fri_cap_height : 4);
  d.fri_num_queries = p.fri_num_queries ? p.fri_num_queries : 28;
  d.fri_pow_bits = p.fri_pow_bits == 0xFFFFFFFFu ? 0 : (p.fri_pow_bits ? p.fri_pow_bits : 16);
  d.fri_num_layers = p.fri_num_layers; for (int i = 0; i < 8; i++) d.fri_arity_bits[i] = p.fri_arity_bits[i];
  d.num_gates = (uint32_t)c->gates.size(); d.num_selector_groups = (uint32_t)nsel; d.gates = c->gates.data(); d.selector_indices = c->sel_idx.data();
  d.selector_group_ranges = c->groups.data(); d.k_is = c->k_is.data(); d.constants_sigmas = cs;
  // Synthetic digest (the real one hashes the constants/sigmas cap + common data; any 4 felts serve the transcript).
  { u64 t[16] = {p.circuit_seed % GL_P, p.degree_bits, p.zero_knowledge, p.n_poseidon, p.n_basesum, p.n_arith, p.n_constant, p.n_copies};
    for (int i = 0; i < 8; i++) t[8 + i] = p.n_recursion[i];
    t[8] += (u64)p.reserved << 32;
    hash_no_pad(t, n_rec ? 16 : 8, d.circuit_digest); }
  *out = c; return 0;
}

void qpsynth_circuit_destroy(qpsynth_circuit* c) { delete c; }
const qpgpu_circuit_desc* qpsynth_circuit_desc(const qpsynth_circuit* c) { return &c->desc; }

static void poseidon_row(u64* wires, size_t n, size_t row, const u64 in[12], u64 swap) {
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

int qpsynth_witness(const qpsynth_circuit* c, uint64_t seed, uint64_t* wires, uint64_t* public_inputs) {
  const size_t n = c->n; const qpsynth_params& p = c->p;
  Rng rng(seed * 0xD1B54A32D192ED03ULL + 0x5157);
  memset(wires, 0, sizeof(u64) * (size_t)W * n);
  const u64* cs = c->constants_sigmas.data();
  const int nsel = c->nsel;
  const KindInfo* KI = c->kinds;
  auto Wr = [&](size_t row, int col) -> u64& { return wires[(size_t)col * n + row]; };
  auto LC = [&](size_t row, int i) -> u64 { return cs[(size_t)(nsel + i) * n + row]; };
  auto rext = [&]() -> E2 { E2 e; e.a = rng.felt(); e.b = rng.felt(); return e; };
  auto putx = [&](size_t row, int col, E2 v) { Wr(row, col) = v.a; Wr(row, col + 1) = v.b; };
  for (int i = 0; i < NPI; i++) public_inputs[i] = rng.felt();
  // PI hash rows
  { u64 in[12] = {0}; for (int i = 0; i < 8; i++) in[i] = public_inputs[i]; poseidon_row(wires, n, c->row_h0, in, 0);
    for (int i = 0; i < 8; i++) in[i] = public_inputs[8 + i]; for (int i = 0; i < 4; i++) in[8 + i] = Wr(c->row_h0, 12 + 8 + i);
    poseidon_row(wires, n, c->row_h1, in, 0);
    for (int i = 0; i < 4; i++) Wr(c->row_pi, i) = Wr(c->row_h1, 12 + i); }
  // phase 1: Poseidon / BaseSum / Constant rows
  for (size_t row = 3; row < n; row++) {
    switch (c->kind[row]) {
      case K_POSEIDON: { u64 in[12]; for (auto& x : in) x = rng.felt(); poseidon_row(wires, n, row, in, rng.next() & 1); break; }
      case K_BASESUM: { u64 v = rng.next() >> 1; Wr(row, 0) = v; for (int l = 0; l < 63; l++) Wr(row, 1 + l) = (v >> l) & 1; break; }
      case K_CONST: Wr(row, 0) = LC(row, 0); Wr(row, 1) = LC(row, 1); break;
      case K_ARITH_EXT:
        for (int op = 0; op < (int)KI[K_ARITH_EXT].param; op++) {
          E2 m0 = rext(), m1 = rext(), ad = rext();
          putx(row, 8 * op, m0); putx(row, 8 * op + 2, m1); putx(row, 8 * op + 4, ad);
          putx(row, 8 * op + 6, e_add(e_scale(e_mul(m0, m1), LC(row, 0)), e_scale(ad, LC(row, 1))));
        }
        break;
      case K_MUL_EXT:
        for (int op = 0; op < (int)KI[K_MUL_EXT].param; op++) {
          E2 m0 = rext(), m1 = rext();
          putx(row, 6 * op, m0); putx(row, 6 * op + 2, m1); putx(row, 6 * op + 4, e_scale(e_mul(m0, m1), LC(row, 0)));
        }
        break;
      case K_REDUCING: case K_REDUCING_EXT: {
        const bool xc = c->kind[row] == K_REDUCING_EXT; const int nc = (int)KI[xc ? K_REDUCING_EXT : K_REDUCING].param, start_accs = 6 + (xc ? 2 : 1) * nc;
        E2 alpha = rext(), acc = rext();
        putx(row, 2, alpha); putx(row, 4, acc);
        for (int i = 0; i < nc; i++) {
          E2 coeff; if (xc) { coeff = rext(); putx(row, 6 + 2 * i, coeff); } else { coeff.a = rng.felt(); coeff.b = 0; Wr(row, 6 + i) = coeff.a; }
          acc = e_add(e_mul(acc, alpha), coeff);
          putx(row, i == nc - 1 ? 0 : start_accs + 2 * i, acc);
        }
        break;
      }
      case K_RANDOM_ACCESS: {
        const int bits = (int)KI[K_RANDOM_ACCESS].param, copies = (int)KI[K_RANDOM_ACCESS].p2, extra = (int)KI[K_RANDOM_ACCESS].p3, vec = 1 << bits;
        const int routed = (2 + vec) * copies + extra;
        for (int cp = 0; cp < copies; cp++) {
          const int base = (2 + vec) * cp; u64 idx = rng.below(vec);
          Wr(row, base) = idx;
          for (int i = 0; i < vec; i++) Wr(row, base + 2 + i) = rng.felt();
          Wr(row, base + 1) = Wr(row, base + 2 + (int)idx);
          for (int i = 0; i < bits; i++) Wr(row, routed + bits * cp + i) = (idx >> i) & 1;
        }
        for (int i = 0; i < extra; i++) Wr(row, (2 + vec) * copies + i) = LC(row, i);
        break;
      }
      case K_POSEIDON_MDS: {
        u64 a[12], b[12];
        for (int i = 0; i < 12; i++) { a[i] = rng.felt(); b[i] = rng.felt(); Wr(row, 2 * i) = a[i]; Wr(row, 2 * i + 1) = b[i]; }
        mds_layer(a); mds_layer(b);
        for (int i = 0; i < 12; i++) { Wr(row, 24 + 2 * i) = a[i]; Wr(row, 24 + 2 * i + 1) = b[i]; }
        break;
      }
      case K_EXP: {
        const int nb = (int)KI[K_EXP].param; u64 base = rng.felt(), prev = 1;
        Wr(row, 0) = base;
        for (int i = 0; i < nb; i++) Wr(row, 1 + i) = rng.next() & 1;
        for (int i = 0; i < nb; i++) {
          u64 v = gl_mul(prev, Wr(row, 1 + (nb - 1 - i)) ? base : 1);
          Wr(row, 2 + nb + i) = v; prev = gl_mul(v, v);
        }
        Wr(row, 1 + nb) = Wr(row, 2 + nb + nb - 1);
        break;
      }
      case K_COSET: {
        const int COSET_NP = 1 << (int)KI[K_COSET].param, COSET_DEG = (int)KI[K_COSET].p2, COSET_NI = (COSET_NP - 2) / (COSET_DEG - 1);
        const int s_point = 1 + 2 * COSET_NP, s_value = s_point + 2, s_inter = s_value + 2, s_shifted = s_inter + 4 * COSET_NI;
        u64 shift = rng.felt(); if (shift == 0) shift = 1;
        E2 vals[16], point = rext(), xs = e_scale(point, gl_inv(shift));
        Wr(row, 0) = shift; putx(row, s_point, point); putx(row, s_shifted, xs);
        for (int i = 0; i < COSET_NP; i++) { vals[i] = rext(); putx(row, 1 + 2 * i, vals[i]); }
        E2 ev{0, 0}, pr{1, 0};
        int lo = 0, hi = COSET_DEG;
        for (int seg = 0; seg <= COSET_NI; seg++) {
          for (int i = lo; i < hi; i++) {
            E2 term{gl_sub(xs.a, c->coset_dom[i]), xs.b};
            ev = e_add(e_mul(ev, term), e_mul(e_scale(vals[i], c->coset_weights[i]), pr)); pr = e_mul(pr, term);
          }
          if (seg == COSET_NI) break;
          putx(row, s_inter + 2 * seg, ev); putx(row, s_inter + 2 * (COSET_NI + seg), pr);
          lo = 1 + (COSET_DEG - 1) * (seg + 1); hi = std::min(lo + COSET_DEG - 1, COSET_NP);
        }
        putx(row, s_value, ev);
        break;
      }
      default: break;
    }
  }
  // free cells start random (so that copies carry entropy and zk Noop rows are blinded)
  for (size_t row = 3; row < n; row++) {
    int k = c->kind[row];
    if (k == K_NOOP && p.zero_knowledge) for (int col = 0; col < W; col++) Wr(row, col) = rng.felt();
    else if (k == K_NOOP) for (int col = 0; col < R; col += 7) Wr(row, col) = rng.felt();
    else if (k == K_BASESUM) for (int col = 64; col < R; col++) Wr(row, col) = rng.felt();
    else if (k == K_CONST) for (int col = 2; col < R; col++) Wr(row, col) = rng.felt();
    else if (k == K_ARITH) for (int op = 0; op < 20; op++) for (int j = 0; j < 3; j++) Wr(row, 4 * op + j) = rng.felt();
  }
  // phase 2: copies into arithmetic inputs (first 8 entries are the PI-hash wiring, already consistent)
  for (size_t k = 8; k < c->n_phase2; k++) Wr(c->copy_tgt[k].row, c->copy_tgt[k].col) = Wr(c->copy_src[k].row, c->copy_src[k].col);
  for (size_t row = 3; row < n; row++) if (c->kind[row] == K_ARITH) {
    u64 c0 = LC(row, 0), c1 = LC(row, 1);
    for (int op = 0; op < 20; op++) Wr(row, 4 * op + 3) = gl_add(gl_mul(c0, gl_mul(Wr(row, 4 * op), Wr(row, 4 * op + 1))), gl_mul(c1, Wr(row, 4 * op + 2)));
  }
  // phase 3: copies into free cells
  for (size_t k = c->n_phase2; k < c->copy_src.size(); k++) Wr(c->copy_tgt[k].row, c->copy_tgt[k].col) = Wr(c->copy_src[k].row, c->copy_src[k].col);
  return 0;
}

void qpsynth_salts_keyed(const qpsynth_circuit* c, const uint8_t* key32, uint64_t stream, uint64_t* out) {
  const size_t N = c->n << 3; const SaltKey key = salt_key_from_bytes(key32);
  for (size_t b = 0; b < 12 * N / 8; b++) salt_block(key, stream, b, out + 8 * b);
}
void qpsynth_salts(const qpsynth_circuit* c, uint64_t salt_seed, uint64_t* out) {
  static const uint8_t test_key[32] = {QPSYNTH_TEST_SALT_KEY_BYTES};
  qpsynth_salts_keyed(c, test_key, salt_seed, out);
}
void qpsynth_salt_block(const uint8_t* key32, uint64_t nonce, uint64_t block, uint64_t* out8) { salt_block(salt_key_from_bytes(key32), nonce, block, out8); }

} // extern "C"
