// ORACLE — TEST INFRASTRUCTURE ONLY. Gate constraint formulas of qp-plonky2 1.1.1 (evaluate_gate_constraints,
// Gate::eval_filtered, compute_filter and the six leaf-circuit gates) per SURVEY.md App. A.7; the gate list comes
// from wormhole/bench-data/common.bin and the identities are validated at zeta on wormhole/bench-data/proof.bin.
#include "plonk.h"
#include <algorithm>

namespace orc {

template <class T> static inline T C(u64 c) { return from_base((T*)0, c); }

template <class T> static T compute_filter(int row, std::pair<int, int> grp, T s, bool many) {
  T f = C<T>(1);
  for (int i = grp.first; i < grp.second; i++) if (i != row) f = f * (C<T>((u64)i) - s);
  if (many) f = f * (C<T>(0xFFFFFFFFULL) - s);
  return f;
}

// ---- recursion-circuit gates. PARITY UNPINNED: no aggregated proof or recursion common data ships with the
// reference, so wire layouts / constraint order follow the recalled upstream definitions and are checked for self-consistency only
// (oracle verifier accepts oracle/GPU proofs of synthetic circuits that use them).
// Alg<T>: a + b*X with X^2 = 7 over T. T = FB gives F_{p^2} (prover, base-field wires on the coset); T = Ext gives upstream's
// ExtensionAlgebra over F_{p^2} (verifier, wires opened at zeta).
template <class T> struct Alg { T a, b; };
template <class T> static inline Alg<T> operator+(Alg<T> x, Alg<T> y) { return {x.a + y.a, x.b + y.b}; }
template <class T> static inline Alg<T> operator-(Alg<T> x, Alg<T> y) { return {x.a - y.a, x.b - y.b}; }
template <class T> static inline Alg<T> operator*(Alg<T> x, Alg<T> y) { return {x.a * y.a + C<T>(W_EXT) * (x.b * y.b), x.a * y.b + x.b * y.a}; }
template <class T> static inline Alg<T> scale(Alg<T> x, T s) { return {x.a * s, x.b * s}; }
template <class T> static inline Alg<T> alg(const T* w, int i) { return {w[i], w[i + 1]}; }
template <class T> static inline Alg<T> alg_base(T x) { return {x, C<T>(0)}; }
template <class T> static inline void push(T* out, int& k, Alg<T> x) { out[k++] = x.a; out[k++] = x.b; }

template <class T> static void arithmetic_ext_gate(const Gate& g, const T* lc, const T* w, T* out, int& k) {
  for (int i = 0; i < (int)g.param; i++)
    push(out, k, alg(w, 8 * i + 6) - (scale(alg(w, 8 * i) * alg(w, 8 * i + 2), lc[0]) + scale(alg(w, 8 * i + 4), lc[1])));
}
template <class T> static void mul_ext_gate(const Gate& g, const T* lc, const T* w, T* out, int& k) {
  for (int i = 0; i < (int)g.param; i++) push(out, k, alg(w, 6 * i + 4) - scale(alg(w, 6 * i) * alg(w, 6 * i + 2), lc[0]));
}
// output [0,2), alpha [2,4), old_acc [4,6), coefficients from 6, accumulators after them (the last accumulator is `output`)
template <class T> static void reducing_gate(const Gate& g, bool ext_coeffs, const T* w, T* out, int& k) {
  const int nc = (int)g.param, cw = ext_coeffs ? 2 : 1, start_accs = 6 + cw * nc;
  Alg<T> alpha = alg(w, 2), acc = alg(w, 4);
  for (int i = 0; i < nc; i++) {
    Alg<T> coeff = ext_coeffs ? alg(w, 6 + 2 * i) : alg_base(w[6 + i]);
    Alg<T> acc_i = i == nc - 1 ? alg(w, 0) : alg(w, start_accs + 2 * i);
    push(out, k, acc * alpha + coeff - acc_i);
    acc = acc_i;
  }
}
template <class T> static void random_access_gate(const Gate& g, const T* lc, const T* w, T* out, int& k) {
  const int bits = (int)g.param, copies = (int)g.p2, extra = (int)g.p3, vec = 1 << bits;
  const int routed = (2 + vec) * copies + extra;
  std::vector<T> list(vec);
  for (int c = 0; c < copies; c++) {
    const int base = (2 + vec) * c;
    const T* b = w + routed + c * bits;
    for (int i = 0; i < bits; i++) out[k++] = b[i] * (b[i] - C<T>(1));
    T idx = C<T>(0);
    for (int i = bits - 1; i >= 0; i--) idx = idx + idx + b[i];
    out[k++] = idx - w[base];
    for (int i = 0; i < vec; i++) list[i] = w[base + 2 + i];
    for (int i = 0, len = vec; i < bits; i++, len >>= 1)
      for (int j = 0; j < len / 2; j++) list[j] = list[2 * j] + b[i] * (list[2 * j + 1] - list[2 * j]);
    out[k++] = list[0] - w[base + 1];
  }
  for (int i = 0; i < extra; i++) out[k++] = lc[i] - w[(2 + vec) * copies + i];
}
template <class T> static void poseidon_mds_gate(const T* w, T* out, int& k) {
  for (int r = 0; r < 12; r++) {
    Alg<T> acc = scale(alg(w, 2 * r), C<T>(MDS_DIAG[r]));
    for (int i = 0; i < 12; i++) acc = acc + scale(alg(w, 2 * ((i + r) % 12)), C<T>(MDS_CIRC[i]));
    push(out, k, alg(w, 24 + 2 * r) - acc);
  }
}
// base 0, power bits 1+i (little-endian), output 1+n, intermediates 2+n+i
template <class T> static void exponentiation_gate(const Gate& g, const T* w, T* out, int& k) {
  const int n = (int)g.param;
  for (int i = 0; i < n; i++) {
    T prev = i == 0 ? C<T>(1) : w[2 + n + i - 1] * w[2 + n + i - 1];
    T bit = w[1 + (n - 1 - i)];
    out[k++] = prev * (bit * w[0] + (C<T>(1) - bit)) - w[2 + n + i];
  }
  out[k++] = w[1 + n] - w[2 + n + n - 1];
}
// shift 0, values 1+2i (2^bits ext values), evaluation point, evaluation value, then (eval, prod) intermediates, then the shifted point
template <class T> static void coset_interpolation_gate(const Gate& g, const T* w, T* out, int& k) {
  const int bits = (int)g.param, deg = (int)g.p2, np = 1 << bits, ni = (np - 2) / (deg - 1);
  const int s_point = 1 + 2 * np, s_value = s_point + 2, s_inter = s_value + 2, s_shifted = s_inter + 4 * ni;
  u64 dom[64]; { u64 wgen = gl_root(bits), x = 1; for (int i = 0; i < np; i++) { dom[i] = x; x = gl_mul(x, wgen); } }
  Alg<T> x = alg(w, s_shifted);
  push(out, k, alg(w, s_point) - scale(x, w[0]));
  auto partial = [&](int lo, int hi, Alg<T> ev, Alg<T> pr, Alg<T>& ev_out, Alg<T>& pr_out) {
    for (int i = lo; i < hi; i++) {
      Alg<T> term = x - alg_base(C<T>(dom[i]));
      Alg<T> val = scale(alg(w, 1 + 2 * i), C<T>(g.table[i]));
      ev = ev * term + val * pr; pr = pr * term;
    }
    ev_out = ev; pr_out = pr;
  };
  Alg<T> ev, pr;
  partial(0, deg, alg_base(C<T>(0)), alg_base(C<T>(1)), ev, pr);
  for (int i = 0; i < ni; i++) {
    Alg<T> iev = alg(w, s_inter + 2 * i), ipr = alg(w, s_inter + 2 * (ni + i));
    push(out, k, iev - ev); push(out, k, ipr - pr);
    int lo = 1 + (deg - 1) * (i + 1), hi = std::min(lo + deg - 1, np);
    partial(lo, hi, iev, ipr, ev, pr);
  }
  push(out, k, alg(w, s_value) - ev);
}

template <class T> static void poseidon_gate(const T* w, T* out, int& k) {
  const u64* rc = poseidon_round_constants();
  T swap = w[24];
  out[k++] = swap * (swap - C<T>(1));
  T st[12];
  for (int i = 0; i < 4; i++) {
    T delta = w[25 + i];
    out[k++] = swap * (w[i + 4] - w[i]) - delta;
    st[i] = w[i] + delta; st[i + 4] = w[i + 4] - delta;
  }
  for (int i = 8; i < 12; i++) st[i] = w[i];
  int round = 0;
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 12; i++) st[i] = st[i] + C<T>(rc[12 * round + i]);
    if (r != 0) for (int i = 0; i < 12; i++) { T in = w[29 + 12 * (r - 1) + i]; out[k++] = st[i] - in; st[i] = in; }
    for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
    mds_layer_t(st); round++;
  }
  for (int r = 0; r < 22; r++) {
    for (int i = 0; i < 12; i++) st[i] = st[i] + C<T>(rc[12 * round + i]);
    T in = w[65 + r]; out[k++] = st[0] - in; st[0] = sbox7(in);
    mds_layer_t(st); round++;
  }
  for (int r = 0; r < 4; r++) {
    for (int i = 0; i < 12; i++) st[i] = st[i] + C<T>(rc[12 * round + i]);
    for (int i = 0; i < 12; i++) { T in = w[87 + 12 * r + i]; out[k++] = st[i] - in; st[i] = in; }
    for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
    mds_layer_t(st); round++;
  }
  for (int i = 0; i < 12; i++) out[k++] = st[i] - w[12 + i];
}

template <class T> void eval_gate_constraints(const Common& c, const T* consts, const T* wires, const T pi_hash[4], T* out) {
  int nsel = c.num_selectors();
  for (int i = 0; i < c.num_gate_constraints; i++) out[i] = C<T>(0);
  const T* lc = consts + nsel;
  std::vector<T> tmp(c.num_gate_constraints);
  for (size_t gi = 0; gi < c.gates.size(); gi++) {
    const Gate& g = c.gates[gi];
    int si = c.selector_indices[gi];
    T filter = compute_filter<T>((int)gi, c.groups[si], consts[si], nsel > 1);
    int k = 0; T* o = tmp.data();
    switch (g.tag) {
      case G_NOOP: break;
      case G_CONSTANT: for (u64 i = 0; i < g.param; i++) o[k++] = lc[i] - wires[i]; break;
      case G_PUBLIC_INPUT: for (int i = 0; i < 4; i++) o[k++] = wires[i] - pi_hash[i]; break;
      case G_BASESUM2: {
        T sum = C<T>(0);
        for (int l = (int)g.param - 1; l >= 0; l--) sum = sum * C<T>(2) + wires[1 + l];
        o[k++] = sum - wires[0];
        for (u64 l = 0; l < g.param; l++) o[k++] = wires[1 + l] * (wires[1 + l] - C<T>(1));
        break;
      }
      case G_ARITH:
        for (u64 i = 0; i < g.param; i++) o[k++] = wires[4 * i + 3] - (lc[0] * wires[4 * i] * wires[4 * i + 1] + lc[1] * wires[4 * i + 2]);
        break;
      case G_POSEIDON: poseidon_gate<T>(wires, o, k); break;
      case G_ARITH_EXT: arithmetic_ext_gate<T>(g, lc, wires, o, k); break;
      case G_MUL_EXT: mul_ext_gate<T>(g, lc, wires, o, k); break;
      case G_REDUCING: reducing_gate<T>(g, false, wires, o, k); break;
      case G_REDUCING_EXT: reducing_gate<T>(g, true, wires, o, k); break;
      case G_RANDOM_ACCESS: random_access_gate<T>(g, lc, wires, o, k); break;
      case G_POSEIDON_MDS: poseidon_mds_gate<T>(wires, o, k); break;
      case G_EXP: exponentiation_gate<T>(g, wires, o, k); break;
      case G_COSET_INTERP: coset_interpolation_gate<T>(g, wires, o, k); break;
      default: throw std::runtime_error("gate not supported by oracle");
    }
    if (k > c.num_gate_constraints) throw std::runtime_error("gate has more constraints than num_gate_constraints");
    for (int j = 0; j < k; j++) out[j] = out[j] + o[j] * filter;
  }
}
template void eval_gate_constraints<FB>(const Common&, const FB*, const FB*, const FB*, FB*);
template void eval_gate_constraints<Ext>(const Common&, const Ext*, const Ext*, const Ext*, Ext*);

} // namespace orc
