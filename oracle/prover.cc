// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of qp-plonky2 1.1.1's prove_with_partition_witness
// (the body behind WormholeProver::prove, wormhole/prover/src/lib.rs:233-237, and aggregate_chunk's
// circuit_data.prove, wormhole/aggregator/src/circuits/tree.rs:136): stages 1-10 of SURVEY.md §3.3 with the
// details of App. A/E. The upstream source is not on disk; prover-side byte-exactness against the Rust prover is
// therefore "parity unpinned" (SURVEY.md §8c). What IS pinned: every proof made here must be accepted by
// verify() (verifier.cc), which itself accepts the reference's golden wormhole/bench-data/proof.bin.
// Also serves as bench.py's cpu_baseline ("port") — OpenMP over columns / rows.
#include "plonk.h"
#include <omp.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace orc {

namespace {

struct Commitment {
  std::vector<std::vector<u64>> coeffs;  // [cols][n]
  std::vector<u64> rows;                 // N x width row-major, leaf (bit-reversed) order, salt appended
  size_t width = 0; MerkleTree tree;
};

// In-place DIF butterflies without the final bit-reversal: natural-order input, bit-reversed-order output.
void ntt_dif_noperm(u64* a, size_t n, const std::vector<u64>& tw /* w_n^j, j < n/2 */) {
  for (size_t half = n >> 1, step = 1; half >= 1; half >>= 1, step <<= 1)
    for (size_t k = 0; k < n; k += 2 * half)
      for (size_t j = 0; j < half; j++) {
        u64 u = a[k + j], v = a[k + j + half];
        a[k + j] = gl_add(u, v); a[k + j + half] = gl_mul(gl_sub(u, v), tw[j * step]);
      }
}
// PolynomialBatch::from_coeffs: LDE x2^rate on coset GEN, rows in bit-reversed order, salt appended, Merkle tree.
// Written for the CPU baseline: the leaf-order LDE of a column is 2^rate independent size-n DIF transforms (block rev(c)
// holds coset c, exactly the layout upstream reaches with transpose + reverse_index_bits), produced column-major with
// sequential writes and then transposed to rows in cache-sized tiles.
Commitment commit_coeffs(std::vector<std::vector<u64>>&& coeffs, int rate_bits, int cap_height, const u64* salt /*[4][N] or null*/) {
  const bool tmg = getenv("ORC_TIMING2") != nullptr; auto t0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* nm) { if (!tmg) return; auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "    [commit] %-12s %7.1f ms\n", nm, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; };
  Commitment cm; cm.coeffs = std::move(coeffs);
  size_t cols = cm.coeffs.size(), n = cm.coeffs[0].size(), N = n << rate_bits;
  int lgn = log2_exact(n), lgN = lgn + rate_bits; int s = salt ? 4 : 0;
  cm.width = cols + s; cm.rows.assign(N * cm.width, 0);
  std::vector<u64> tw(n / 2 ? n / 2 : 1); { u64 w = gl_root(lgn), x = 1; for (size_t j = 0; j < n / 2; j++) { tw[j] = x; x = gl_mul(x, w); } }
  const size_t cosets = size_t(1) << rate_bits;
  std::vector<u64> scale(N);   // scale[c*n + i] = (GEN * w_N^c)^i
  { u64 wN = gl_root(lgN);
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < cosets; c++) { u64 base = gl_mul(GEN, gl_pow(wN, c)), x = 1; for (size_t i = 0; i < n; i++) { scale[c * n + i] = x; x = gl_mul(x, base); } } }
  lap("alloc+scale");
  std::vector<u64> cm_lde(cols * N);
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
  for (size_t c = 0; c < cols; c++)
    for (size_t co = 0; co < cosets; co++) {
      u64* dst = &cm_lde[c * N + (size_t)rev_bits((uint32_t)co, rate_bits) * n];
      const u64* src = cm.coeffs[c].data(); const u64* sc = &scale[co * n];
      for (size_t i = 0; i < n; i++) dst[i] = gl_mul(src[i], sc[i]);
      ntt_dif_noperm(dst, n, tw);
    }
  lap("lde");
  const size_t TILE = 64;
#pragma omp parallel for schedule(static)
  for (size_t L0 = 0; L0 < N; L0 += TILE) {
    size_t L1 = L0 + TILE < N ? L0 + TILE : N;
    for (size_t c = 0; c < cols; c++) { const u64* src = &cm_lde[c * N]; for (size_t L = L0; L < L1; L++) cm.rows[L * cm.width + c] = src[L]; }
    for (int k = 0; k < s; k++) for (size_t L = L0; L < L1; L++) cm.rows[L * cm.width + cols + k] = salt[(size_t)k * N + L];
  }
  lap("transpose");
  cm.tree = merkle_build(cm.rows.data(), N, cm.width, cap_height);
  lap("merkle");
  return cm;
}

Commitment commit_values(std::vector<std::vector<u64>>&& vals, int rate_bits, int cap_height, const u64* salt) {
#pragma omp parallel for schedule(dynamic, 1)
  for (size_t c = 0; c < vals.size(); c++) ntt(vals[c], true);
  return commit_coeffs(std::move(vals), rate_bits, cap_height, salt);
}

Ext eval_poly_ext(const std::vector<u64>& c, Ext z) {
  Ext acc = ex(0);
  for (size_t i = c.size(); i-- > 0;) acc = acc * z + ex(c[i]);
  return acc;
}

void ext_coset_ntt(std::vector<Ext>& v, u64 shift) {  // in: coeffs (natural), out: values on shift*<w> natural order
  size_t n = v.size(); std::vector<u64> a(n), b(n); u64 s = 1;
  for (size_t i = 0; i < n; i++) { a[i] = gl_mul(v[i].a, s); b[i] = gl_mul(v[i].b, s); s = gl_mul(s, shift); }
  ntt(a, false); ntt(b, false);
  for (size_t i = 0; i < n; i++) v[i] = Ext{a[i], b[i]};
}

} // namespace

Preprocessed preprocess(const Common& c, const u64* cs, int threads) {
  if (threads > 0) omp_set_num_threads(threads);
  size_t n = c.n(); int S = c.num_preprocessed();
  std::vector<std::vector<u64>> vals(S);
  for (int j = 0; j < S; j++) vals[j].assign(cs + (size_t)j * n, cs + (size_t)(j + 1) * n);
  Commitment cm = commit_values(std::move(vals), c.rate_bits, c.cap_height, nullptr);
  Preprocessed p; p.coeffs = std::move(cm.coeffs); p.lde_rows = std::move(cm.rows); p.tree = std::move(cm.tree);
  return p;
}

namespace { struct StageTimer { bool on; std::chrono::steady_clock::time_point t; StageTimer() : on(getenv("ORC_TIMING") != nullptr), t(std::chrono::steady_clock::now()) {}
  void lap(const char* name) { if (!on) return; auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[oracle] %-22s %8.1f ms\n", name, std::chrono::duration<double, std::milli>(n - t).count()); t = n; } }; }

Proof prove(const ProverInputs& in, const Preprocessed& pre, std::string* err) {
  StageTimer tm;
  const Common& c = *in.common;
  if (in.threads > 0) omp_set_num_threads(in.threads);
  const size_t n = c.n(), N = c.lde_n();
  const int lgn = c.degree_bits, lgN = lgn + c.rate_bits, nch = c.num_challenges;
  const int W = c.num_wires, R = c.num_routed, S = c.num_preprocessed(), NC = c.num_constants;
  const int np = c.num_partial_products, deg = c.qdf, nchunks = np + 1;
  const u64* salt[3] = {nullptr, nullptr, nullptr};
  if (c.hiding) { if (!in.salts) { if (err) *err = "zk circuit needs salts"; return Proof(); } for (int i = 0; i < 3; i++) salt[i] = in.salts + (size_t)i * 4 * N; }
  Proof pf;
  pf.public_inputs.assign(in.public_inputs, in.public_inputs + c.num_public_inputs);
  Hash pi_hash = hash_no_pad(pf.public_inputs.data(), pf.public_inputs.size());

  // ---- 2. wires commitment
  std::vector<std::vector<u64>> wv(W);
  for (int j = 0; j < W; j++) wv[j].assign(in.wires + (size_t)j * n, in.wires + (size_t)(j + 1) * n);
  Commitment wires = commit_values(std::move(wv), c.rate_bits, c.cap_height, salt[0]);
  pf.wires_cap = wires.tree.cap();
  tm.lap("wires commit");

  Challenger ch;
  ch.observe_hash(in.circuit_digest); ch.observe_hash(pi_hash); ch.observe_cap(pf.wires_cap);
  std::vector<u64> betas(nch), gammas(nch), alphas(nch);
  for (auto& b : betas) b = ch.challenge();
  for (auto& g : gammas) g = ch.challenge();

  // ---- 4. Z and partial products
  std::vector<std::vector<u64>> zpp(c.num_zs_pp(), std::vector<u64>(n));
  {
    u64 wn = gl_root(lgn);
    std::vector<u64> xs(n); { u64 x = 1; for (size_t i = 0; i < n; i++) { xs[i] = x; x = gl_mul(x, wn); } }
    for (int cidx = 0; cidx < nch; cidx++) {
      u64 beta = betas[cidx], gamma = gammas[cidx];
      std::vector<u64> q(n * nchunks);  // chunk quotients per row
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n; i++) {
        u64 nums[16], dens[16];
        for (int k = 0; k < nchunks; k++) { nums[k] = 1; dens[k] = 1; }
        for (int j = 0; j < R; j++) {
          u64 w = in.wires[(size_t)j * n + i];
          u64 sg = in.constants_sigmas[(size_t)(NC + j) * n + i];
          u64 nu = gl_add(gl_add(w, gl_mul(beta, gl_mul(c.k_is[j], xs[i]))), gamma);
          u64 de = gl_add(gl_add(w, gl_mul(beta, sg)), gamma);
          nums[j / deg] = gl_mul(nums[j / deg], nu); dens[j / deg] = gl_mul(dens[j / deg], de);
        }
        for (int k = 0; k < nchunks; k++) q[i * nchunks + k] = gl_mul(nums[k], gl_inv(dens[k]));
      }
      u64 z = 1;
      for (size_t i = 0; i < n; i++) {
        zpp[cidx][i] = z;                                   // Z(x_i), Z(x_0) = 1
        u64 acc = z;
        for (int k = 0; k < nchunks; k++) {
          acc = gl_mul(acc, q[i * nchunks + k]);
          if (k < np) zpp[nch + cidx * np + k][i] = acc;    // partial products
        }
        z = acc;                                            // Z(g x_i)
      }
    }
  }
  tm.lap("z/partial products");
  Commitment zs = commit_values(std::move(zpp), c.rate_bits, c.cap_height, salt[1]);
  tm.lap("zs commit");
  pf.zs_pp_cap = zs.tree.cap();
  ch.observe_cap(pf.zs_pp_cap);
  for (auto& a : alphas) a = ch.challenge();

  // ---- 6. quotient polynomials
  std::vector<std::vector<u64>> qvals(nch, std::vector<u64>(N));
  {
    u64 wN = gl_root(lgN), w8 = gl_root(c.rate_bits);
    u64 gn = gl_pow(GEN, n);
    std::vector<u64> zh_inv(size_t(1) << c.rate_bits);
    std::vector<u64> w8pow(zh_inv.size());
    { u64 t = 1; for (size_t k = 0; k < zh_inv.size(); k++) { w8pow[k] = t; zh_inv[k] = gl_inv(gl_sub(gl_mul(gn, t), 1)); t = gl_mul(t, w8); } }
    u64 n_inv = gl_inv((u64)n);
    int nterms = nch + nch * nchunks + c.num_gate_constraints;
    // x_j = GEN * w_N^j and 1/(x_j - 1) for all points (running products + one inversion per chunk)
    std::vector<u64> xs(N), xinv(N);
    const size_t CH = 1024;
#pragma omp parallel for schedule(static)
    for (size_t j0 = 0; j0 < N; j0 += CH) {
      size_t j1 = j0 + CH < N ? j0 + CH : N;
      u64 x = gl_mul(GEN, gl_pow(wN, j0));
      u64 run = 1;
      for (size_t j = j0; j < j1; j++) { xs[j] = x; xinv[j] = run; run = gl_mul(run, gl_sub(x, 1)); x = gl_mul(x, wN); }
      u64 inv = gl_inv(run);
      for (size_t j = j1; j-- > j0;) { u64 d = gl_sub(xs[j], 1); xinv[j] = gl_mul(xinv[j], inv); inv = gl_mul(inv, d); }
    }
#pragma omp parallel
    {
      std::vector<FB> consts(NC), wv2(W), gate(c.num_gate_constraints), terms(nterms);
#pragma omp for schedule(static)
      for (size_t j = 0; j < N; j++) {
        size_t L = rev_bits((uint32_t)j, lgN), Ln = rev_bits((uint32_t)((j + (size_t(1) << c.rate_bits)) % N), lgN);
        u64 x = xs[j];
        const u64* cs = &pre.lde_rows[L * S]; const u64* wr = &wires.rows[L * wires.width];
        const u64* zr = &zs.rows[L * zs.width]; const u64* zn = &zs.rows[Ln * zs.width];
        for (int k = 0; k < NC; k++) consts[k] = FB(cs[k]);
        for (int k = 0; k < W; k++) wv2[k] = FB(wr[k]);
        FB pih[4] = {FB(pi_hash[0]), FB(pi_hash[1]), FB(pi_hash[2]), FB(pi_hash[3])};
        eval_gate_constraints<FB>(c, consts.data(), wv2.data(), pih, gate.data());
        u64 zh = gl_sub(gl_mul(gn, w8pow[j % zh_inv.size()]), 1);
        u64 l0 = gl_mul(zh, gl_mul(n_inv, xinv[j]));
        int t = 0;
        for (int i = 0; i < nch; i++) terms[t++] = FB(gl_mul(l0, gl_sub(zr[i], 1)));
        for (int i = 0; i < nch; i++) {
          u64 prev = zr[i];
          const u64 bx = gl_mul(betas[i], x);
          for (int k = 0; k < nchunks; k++) {
            u64 next = k < np ? zr[nch + i * np + k] : zn[i];
            u64 num = 1, den = 1;
            for (int jj = k * deg; jj < (k + 1) * deg && jj < R; jj++) {
              num = gl_mul(num, gl_add(gl_add(wr[jj], gl_mul(bx, c.k_is[jj])), gammas[i]));
              den = gl_mul(den, gl_add(gl_add(wr[jj], gl_mul(betas[i], cs[NC + jj])), gammas[i]));
            }
            terms[t++] = FB(gl_sub(gl_mul(prev, num), gl_mul(next, den)));
            prev = next;
          }
        }
        for (int k = 0; k < c.num_gate_constraints; k++) terms[t++] = gate[k];
        for (int i = 0; i < nch; i++) {
          u64 acc = 0;
          for (int k = nterms; k-- > 0;) acc = gl_add(gl_mul(acc, alphas[i]), terms[k].v);
          qvals[i][j] = gl_mul(acc, zh_inv[j % zh_inv.size()]);
        }
      }
    }
  }
  tm.lap("quotient eval");
  std::vector<std::vector<u64>> qchunks;
  {
    u64 ginv = gl_inv(GEN);
    for (int i = 0; i < nch; i++) {
      ntt(qvals[i], true);
      u64 s = 1; for (size_t k = 0; k < N; k++) { qvals[i][k] = gl_mul(qvals[i][k], s); s = gl_mul(s, ginv); }
      for (int m = 0; m < deg; m++) qchunks.emplace_back(qvals[i].begin() + m * n, qvals[i].begin() + (m + 1) * n);
    }
  }
  tm.lap("quotient intt");
  Commitment quot = commit_coeffs(std::move(qchunks), c.rate_bits, c.cap_height, salt[2]);
  tm.lap("quotient commit");
  pf.quotient_cap = quot.tree.cap();
  ch.observe_cap(pf.quotient_cap);
  Ext zeta = ch.challenge_ext();
  if (ex_pow(zeta, n) == ex(1)) { if (err) *err = "zeta in subgroup"; return Proof(); }

  // ---- 8. openings
  u64 gn1 = gl_root(lgn);
  Ext gzeta = ex_scale(zeta, gn1);
  auto open_all = [&](const std::vector<std::vector<u64>>& polys, Ext z) {
    std::vector<Ext> o(polys.size());
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t i = 0; i < polys.size(); i++) o[i] = eval_poly_ext(polys[i], z);
    return o;
  };
  {
    std::vector<Ext> ocs = open_all(pre.coeffs, zeta), ow = open_all(wires.coeffs, zeta), oz = open_all(zs.coeffs, zeta), oq = open_all(quot.coeffs, zeta);
    pf.o_constants.assign(ocs.begin(), ocs.begin() + NC); pf.o_sigmas.assign(ocs.begin() + NC, ocs.end());
    pf.o_wires = ow; pf.o_zs.assign(oz.begin(), oz.begin() + nch); pf.o_pp.assign(oz.begin() + nch, oz.end());
    pf.o_quotient = oq;
    for (int i = 0; i < nch; i++) pf.o_zs_next.push_back(eval_poly_ext(zs.coeffs[i], gzeta));
  }
  for (auto* v : {&pf.o_constants, &pf.o_sigmas, &pf.o_wires, &pf.o_zs, &pf.o_pp, &pf.o_quotient}) for (Ext e : *v) ch.observe_ext(e);
  for (Ext e : pf.o_zs_next) ch.observe_ext(e);

  tm.lap("openings");
  // ---- 9. FRI: batch combine in coefficient space (prove_openings)
  Ext fa = ch.challenge_ext();
  std::vector<Ext> final_poly(n, ex(0));
  {
    std::vector<const std::vector<u64>*> b0, b1;
    for (auto& p : pre.coeffs) b0.push_back(&p);
    for (auto& p : wires.coeffs) b0.push_back(&p);
    for (auto& p : zs.coeffs) b0.push_back(&p);
    for (auto& p : quot.coeffs) b0.push_back(&p);
    for (int i = 0; i < nch; i++) b1.push_back(&zs.coeffs[i]);
    Ext points[2] = {zeta, gzeta};
    const std::vector<const std::vector<u64>*>* batches[2] = {&b0, &b1};
    for (int b = 0; b < 2; b++) {
      const auto& polys = *batches[b];
      std::vector<Ext> comp(n);
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n; i++) { Ext acc = ex(0); for (size_t j = polys.size(); j-- > 0;) acc = acc * fa + ex((*polys[j])[i]); comp[i] = acc; }
      // divide_by_linear: (P(X) - P(z)) / (X - z), padded back to n coefficients
      std::vector<Ext> q(n, ex(0)); Ext carry = ex(0);
      for (size_t i = n; i-- > 1;) { carry = comp[i] + points[b] * carry; q[i - 1] = carry; }
      Ext sh = ex_pow(fa, polys.size());
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n; i++) final_poly[i] = final_poly[i] * sh + q[i];
    }
  }
  std::vector<Ext> coeffs(N, ex(0));
  std::copy(final_poly.begin(), final_poly.end(), coeffs.begin());
  std::vector<Ext> values = coeffs;
  ext_coset_ntt(values, GEN);

  tm.lap("fri combine + lde");
  // ---- fri_committed_trees
  std::vector<MerkleTree> fri_trees; std::vector<std::vector<u64>> fri_leaves;
  u64 shift = GEN;
  for (size_t l = 0; l < c.arity_bits.size(); l++) {
    int ab = c.arity_bits[l]; size_t arity = size_t(1) << ab, len = values.size(); int lg = log2_exact(len);
    std::vector<u64> leaves(len * 2);
    for (size_t j = 0; j < len; j++) { size_t L = rev_bits((uint32_t)j, lg); leaves[2 * L] = values[j].a; leaves[2 * L + 1] = values[j].b; }
    MerkleTree t = merkle_build(leaves.data(), len / arity, 2 * arity, c.cap_height);
    pf.fri_caps.push_back(t.cap()); ch.observe_cap(t.cap());
    fri_trees.push_back(std::move(t)); fri_leaves.push_back(std::move(leaves));
    Ext beta = ch.challenge_ext();
    std::vector<Ext> nc(len / arity);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < nc.size(); i++) { Ext acc = ex(0); for (size_t j = arity; j-- > 0;) acc = acc * beta + coeffs[i * arity + j]; nc[i] = acc; }
    coeffs.swap(nc);
    for (int i = 0; i < ab; i++) shift = gl_sqr(shift);
    values = coeffs; ext_coset_ntt(values, shift);
  }
  coeffs.resize(coeffs.size() >> c.rate_bits);
  pf.final_poly = coeffs;
  for (Ext e : pf.final_poly) ch.observe_ext(e);

  tm.lap("fri layers");
  // ---- PoW (canonical choice of this build: the smallest valid witness; upstream's parallel find_any is not deterministic)
  if (in.have_pow) pf.pow_witness = in.pow_witness;
  else {
    bool found = false; u64 best = 0;
    for (u64 base = 0; !found; base += 1 << 14) {
      u64 local_best = ~0ULL;
#pragma omp parallel for reduction(min : local_best) schedule(static)
      for (u64 k = 0; k < (1u << 14); k++) {
        Challenger c2 = ch; c2.observe(base + k); u64 r = c2.challenge();
        int lz = r == 0 ? 64 : __builtin_clzll(r);
        if (lz >= c.pow_bits && base + k < local_best) local_best = base + k;
      }
      if (local_best != ~0ULL) { found = true; best = local_best; }
    }
    pf.pow_witness = best;
  }
  ch.observe(pf.pow_witness);
  u64 pow_response = ch.challenge();
  { int lz = pow_response == 0 ? 64 : __builtin_clzll(pow_response); if (lz < c.pow_bits) { if (err) *err = "supplied pow_witness invalid"; return Proof(); } }

  tm.lap("pow");
  // ---- queries
  const std::vector<u64>* rows[4] = {&pre.lde_rows, &wires.rows, &zs.rows, &quot.rows};
  size_t widths[4] = {(size_t)S, wires.width, zs.width, quot.width};
  const MerkleTree* trees[4] = {&pre.tree, &wires.tree, &zs.tree, &quot.tree};
  for (int q = 0; q < c.num_queries; q++) {
    size_t x = ch.challenge() % N;
    QueryRound qr;
    for (int t = 0; t < 4; t++) {
      InitialTreeEntry e; e.evals.assign(rows[t]->begin() + x * widths[t], rows[t]->begin() + (x + 1) * widths[t]);
      e.proof = trees[t]->prove(x); qr.initial.push_back(std::move(e));
    }
    for (size_t l = 0; l < c.arity_bits.size(); l++) {
      int ab = c.arity_bits[l]; size_t arity = size_t(1) << ab; size_t ci = x >> ab;
      QueryStep s; for (size_t k = 0; k < arity; k++) s.evals.push_back(Ext{fri_leaves[l][2 * (ci * arity + k)], fri_leaves[l][2 * (ci * arity + k) + 1]});
      s.proof = fri_trees[l].prove(ci); qr.steps.push_back(std::move(s)); x = ci;
    }
    pf.queries.push_back(std::move(qr));
  }
  return pf;
}

} // namespace orc
