// ORACLE — TEST INFRASTRUCTURE ONLY. MerkleTree / verify_merkle_proof_to_cap of qp-plonky2 1.1.1 per SURVEY.md App. A.4.
#include "plonk.h"
#include <omp.h>

namespace orc {

bool merkle_verify_to_cap(const u64* leaf, size_t len, size_t index, const std::vector<Hash>& cap, const MerkleProofData& pr) {
  Hash d = hash_or_noop(leaf, len);
  for (auto& sib : pr.siblings) { d = (index & 1) ? two_to_one(sib, d) : two_to_one(d, sib); index >>= 1; }
  return index < cap.size() && d == cap[index];
}

MerkleTree merkle_build(const u64* rows, size_t num_leaves, size_t width, int cap_height) {
  MerkleTree t; t.num_leaves = num_leaves; t.cap_height = cap_height;
  std::vector<Hash> cur(num_leaves);
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < num_leaves; i++) cur[i] = hash_or_noop(rows + i * width, width);
  t.levels.push_back(cur);
  while (cur.size() > (size_t(1) << cap_height)) {
    std::vector<Hash> nxt(cur.size() / 2);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < nxt.size(); i++) nxt[i] = two_to_one(cur[2 * i], cur[2 * i + 1]);
    t.levels.push_back(nxt); cur.swap(nxt);
  }
  return t;
}

} // namespace orc
