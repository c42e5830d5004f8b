// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of qp-plonky2 1.1.1's verify (get_challenges,
// verify_with_challenges, eval_vanishing_poly, verify_fri_proof), as called by the reference at
// wormhole/verifier/src/lib.rs:155-159. Formulas: SURVEY.md App. A.5-A.9. Pinned by accepting the golden
// instance wormhole/bench-data/{common,verifier,proof}.bin (tests/test_oracle_golden.py).
#include "plonk.h"

namespace orc {

static Ext reduce_with_powers(const std::vector<Ext>& v, Ext alpha) {
  Ext acc = ex(0); for (size_t i = v.size(); i-- > 0;) acc = acc * alpha + v[i]; return acc;
}

std::string verify(const Proof& p, const Common& c, const VerifierOnly& vo) {
  if ((int)p.public_inputs.size() != c.num_public_inputs) return "public input count";
  const size_t n = c.n(), N = c.lde_n();
  const int nch = c.num_challenges;
  Hash pi_hash = hash_no_pad(p.public_inputs.data(), p.public_inputs.size());

  // ---- challenges (A.6)
  Challenger ch;
  ch.observe_hash(vo.circuit_digest); ch.observe_hash(pi_hash); ch.observe_cap(p.wires_cap);
  std::vector<u64> betas(nch), gammas(nch), alphas(nch);
  for (auto& b : betas) b = ch.challenge();
  for (auto& g : gammas) g = ch.challenge();
  ch.observe_cap(p.zs_pp_cap);
  for (auto& a : alphas) a = ch.challenge();
  ch.observe_cap(p.quotient_cap);
  Ext zeta = ch.challenge_ext();
  for (auto* v : {&p.o_constants, &p.o_sigmas, &p.o_wires, &p.o_zs, &p.o_pp, &p.o_quotient}) for (Ext e : *v) ch.observe_ext(e);
  for (Ext e : p.o_zs_next) ch.observe_ext(e);
  Ext fri_alpha = ch.challenge_ext();
  if (p.fri_caps.size() != c.arity_bits.size()) return "fri cap count";
  std::vector<Ext> fri_betas;
  for (auto& cap : p.fri_caps) { ch.observe_cap(cap); fri_betas.push_back(ch.challenge_ext()); }
  if (p.final_poly.size() != c.final_poly_len()) return "final poly len";
  for (Ext e : p.final_poly) ch.observe_ext(e);
  ch.observe(p.pow_witness);
  u64 pow_response = ch.challenge();
  std::vector<size_t> indices(c.num_queries);
  for (auto& i : indices) i = ch.challenge() % N;

  // ---- vanishing identity at zeta (A.8)
  {
    std::vector<Ext> consts = p.o_constants, wires = p.o_wires;
    Ext pih[4] = {ex(pi_hash[0]), ex(pi_hash[1]), ex(pi_hash[2]), ex(pi_hash[3])};
    std::vector<Ext> gate_terms(c.num_gate_constraints);
    eval_gate_constraints<Ext>(c, consts.data(), wires.data(), pih, gate_terms.data());
    Ext zeta_n = ex_pow(zeta, n);
    Ext zh = zeta_n - ex(1);
    Ext l0 = zh * ex_inv(ex_scale(zeta - ex(1), (u64)n));
    std::vector<Ext> z1_terms, pp_terms;
    const int np = c.num_partial_products, deg = c.qdf;
    for (int i = 0; i < nch; i++) {
      z1_terms.push_back(l0 * (p.o_zs[i] - ex(1)));
      std::vector<Ext> accs; accs.push_back(p.o_zs[i]);
      for (int k = 0; k < np; k++) accs.push_back(p.o_pp[i * np + k]);
      accs.push_back(p.o_zs_next[i]);
      int chunk = 0;
      for (int j0 = 0; j0 < c.num_routed; j0 += deg, chunk++) {
        Ext num = ex(1), den = ex(1);
        for (int j = j0; j < j0 + deg && j < c.num_routed; j++) {
          num = num * (p.o_wires[j] + ex_scale(zeta, gl_mul(betas[i], c.k_is[j])) + ex(gammas[i]));
          den = den * (p.o_wires[j] + ex_scale(p.o_sigmas[j], betas[i]) + ex(gammas[i]));
        }
        pp_terms.push_back(accs[chunk] * num - accs[chunk + 1] * den);
      }
    }
    std::vector<Ext> terms = z1_terms; terms.insert(terms.end(), pp_terms.begin(), pp_terms.end());
    terms.insert(terms.end(), gate_terms.begin(), gate_terms.end());
    for (int i = 0; i < nch; i++) {
      Ext van = reduce_with_powers(terms, ex(alphas[i]));
      std::vector<Ext> chunk(p.o_quotient.begin() + i * c.qdf, p.o_quotient.begin() + (i + 1) * c.qdf);
      Ext q = reduce_with_powers(chunk, zeta_n);
      if (van != zh * q) return "vanishing identity failed for challenge " + std::to_string(i);
    }
  }

  // ---- FRI (A.9)
  {
    int lz = pow_response == 0 ? 64 : __builtin_clzll(pow_response);
    if (lz < c.pow_bits) return "proof of work";
    if ((int)p.queries.size() != c.num_queries) return "query count";
    // batches
    std::vector<Ext> b0;
    for (auto* v : {&p.o_constants, &p.o_sigmas, &p.o_wires, &p.o_zs, &p.o_pp, &p.o_quotient}) b0.insert(b0.end(), v->begin(), v->end());
    Ext y0 = reduce_with_powers(b0, fri_alpha), y1 = reduce_with_powers(p.o_zs_next, fri_alpha);
    u64 gn = gl_root(c.degree_bits);
    Ext z0 = zeta, z1 = ex_scale(zeta, gn);
    const std::vector<Hash>* caps[4] = {&vo.constants_sigmas_cap, &p.wires_cap, &p.zs_pp_cap, &p.quotient_cap};
    int unsalted[4] = {c.num_preprocessed(), c.num_wires, c.num_zs_pp(), c.num_quotient()};
    int lgN = c.degree_bits + c.rate_bits;
    for (int q = 0; q < c.num_queries; q++) {
      const QueryRound& qr = p.queries[q];
      size_t x_index = indices[q];
      if (qr.initial.size() != 4) return "initial tree count";
      for (int t = 0; t < 4; t++)
        if (!merkle_verify_to_cap(qr.initial[t].evals.data(), qr.initial[t].evals.size(), x_index, *caps[t], qr.initial[t].proof))
          return "initial merkle proof q" + std::to_string(q) + " t" + std::to_string(t);
      u64 sx = gl_mul(GEN, gl_pow(gl_root(lgN), rev_bits((uint32_t)x_index, lgN)));
      // combine
      std::vector<Ext> e0;
      for (int t = 0; t < 4; t++) for (int j = 0; j < unsalted[t]; j++) e0.push_back(ex(qr.initial[t].evals[j]));
      std::vector<Ext> e1; for (int j = 0; j < c.num_challenges; j++) e1.push_back(ex(qr.initial[2].evals[j]));
      Ext sum = ex(0);
      sum = sum * ex_pow(fri_alpha, e0.size()) + (reduce_with_powers(e0, fri_alpha) - y0) * ex_inv(ex(sx) - z0);
      sum = sum * ex_pow(fri_alpha, e1.size()) + (reduce_with_powers(e1, fri_alpha) - y1) * ex_inv(ex(sx) - z1);
      Ext old_eval = sum;
      if (qr.steps.size() != c.arity_bits.size()) return "step count";
      for (size_t l = 0; l < c.arity_bits.size(); l++) {
        int ab = c.arity_bits[l], arity = 1 << ab;
        const QueryStep& s = qr.steps[l];
        size_t coset_index = x_index >> ab, within = x_index & (arity - 1);
        if (s.evals[within] != old_eval) return "fri consistency q" + std::to_string(q) + " layer " + std::to_string(l);
        // interpolate {(x*g^i, evals[rev(i)])} at beta (barycentric == Lagrange)
        u64 g = gl_root(ab);
        u64 coset_start = gl_mul(sx, gl_pow(g, arity - rev_bits((uint32_t)within, ab)));
        std::vector<Ext> xs(arity), ys(arity);
        u64 gp = 1;
        for (int i = 0; i < arity; i++) { xs[i] = ex(gl_mul(coset_start, gp)); ys[i] = s.evals[rev_bits(i, ab)]; gp = gl_mul(gp, g); }
        Ext beta = fri_betas[l], acc = ex(0);
        for (int i = 0; i < arity; i++) {
          Ext num = ex(1), den = ex(1);
          for (int j = 0; j < arity; j++) if (j != i) { num = num * (beta - xs[j]); den = den * (xs[i] - xs[j]); }
          acc = acc + ys[i] * num * ex_inv(den);
        }
        old_eval = acc;
        std::vector<u64> flat; for (Ext e : s.evals) { flat.push_back(e.a); flat.push_back(e.b); }
        if (!merkle_verify_to_cap(flat.data(), flat.size(), coset_index, p.fri_caps[l], s.proof)) return "fri merkle q" + std::to_string(q) + " layer " + std::to_string(l);
        for (int i = 0; i < ab; i++) sx = gl_sqr(sx);
        x_index = coset_index;
      }
      Ext fx = ex(0);
      for (size_t i = p.final_poly.size(); i-- > 0;) fx = fx * ex(sx) + p.final_poly[i];
      if (fx != old_eval) return "final poly q" + std::to_string(q);
    }
  }
  return "";
}

} // namespace orc
