// Deterministic salt stream for zk (hiding) oracles. qp-plonky2 draws F::rand_vec for the 4 salt columns of each blinded
// oracle; the source under the reference's `no_random` feature (Cargo.toml:20-22) is not knowable offline (SURVEY.md
// App. A.10), so salts are an explicit ABI input or derived from opts.salt_seed with this counter hash.
#pragma once
#include "gl.h"
namespace qp {
// index = (oracle*4 + column)*N + leaf
QP_HD u64 salt_value(u64 seed, u64 index) {
  u64 z = seed * 0xD1342543DE82EF95ULL + (index + 1) * 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; z ^= z >> 31;
  return z >= GL_P ? z - GL_P : z;
}
}
