// ORACLE — TEST INFRASTRUCTURE ONLY. Gate constraint formulas of qp-plonky2 1.1.1 (evaluate_gate_constraints,
// Gate::eval_filtered, compute_filter and the six leaf-circuit gates) per SURVEY.md App. A.7; the gate list comes
// from wormhole/bench-data/common.bin and the identities are validated at zeta on wormhole/bench-data/proof.bin.
#include "plonk.h"

namespace orc {

template <class T> static inline T C(u64 c) { return from_base((T*)0, c); }

template <class T> static T compute_filter(int row, std::pair<int, int> grp, T s, bool many) {
  T f = C<T>(1);
  for (int i = grp.first; i < grp.second; i++) if (i != row) f = f * (C<T>((u64)i) - s);
  if (many) f = f * (C<T>(0xFFFFFFFFULL) - s);
  return f;
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
      default: throw std::runtime_error("gate not supported by oracle");
    }
    for (int j = 0; j < k; j++) out[j] = out[j] + o[j] * filter;
  }
}
template void eval_gate_constraints<FB>(const Common&, const FB*, const FB*, const FB*, FB*);
template void eval_gate_constraints<Ext>(const Common&, const Ext*, const Ext*, const Ext*, Ext*);

} // namespace orc
