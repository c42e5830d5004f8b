// Salt stream for zk (hiding) oracles. qp-plonky2 draws F::rand_vec (OsRng) for the 4 salt columns of each blinded oracle
// (SURVEY.md App. A.10); every FRI query opening reveals salts at known positions, so the stream must be one-way in its key:
// salts are either an explicit ABI input (opts.salts) or the ChaCha8 keystream of a caller-supplied 256-bit secret key
// (opts.salt_key), nonce = opts.salt_stream (distinguishes proofs under one key), 64-bit block counter = element index / 8.
// Each 64-byte block yields eight u64, taken mod p by one conditional subtraction (bias 2^-32 per element).
// Compiles for host (synthetic generator: the oracle side of the parity tests) and gfx950 device.
#pragma once
#include "gl.h"
namespace qp {

struct SaltKey { u32 k[8]; };      // 32 key bytes, little-endian words

QP_HD u32 rotl32(u32 x, int r) { return (x << r) | (x >> (32 - r)); }
#define QP_CHACHA_QR(a, b, c, d) do { a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7); } while (0)
// out[0..8) = the eight field elements of ChaCha8 block `block` under (key, nonce)
QP_HD void salt_block(const SaltKey& key, u64 nonce, u64 block, u64 out[8]) {
  u32 s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key.k[0], key.k[1], key.k[2], key.k[3], key.k[4], key.k[5], key.k[6], key.k[7],
               (u32)block, (u32)(block >> 32), (u32)nonce, (u32)(nonce >> 32)};
  u32 x[16];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 16; i++) x[i] = s[i];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int r = 0; r < 4; r++) {      // 4 double rounds = ChaCha8
    QP_CHACHA_QR(x[0], x[4], x[8], x[12]); QP_CHACHA_QR(x[1], x[5], x[9], x[13]); QP_CHACHA_QR(x[2], x[6], x[10], x[14]); QP_CHACHA_QR(x[3], x[7], x[11], x[15]);
    QP_CHACHA_QR(x[0], x[5], x[10], x[15]); QP_CHACHA_QR(x[1], x[6], x[11], x[12]); QP_CHACHA_QR(x[2], x[7], x[8], x[13]); QP_CHACHA_QR(x[3], x[4], x[9], x[14]);
  }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < 8; i++) {
    const u64 z = (u64)(x[2 * i] + s[2 * i]) | ((u64)(x[2 * i + 1] + s[2 * i + 1]) << 32);
    out[i] = z >= GL_P ? z - GL_P : z;
  }
}
// element index = (oracle*4 + column)*N + leaf  ->  block index / position in the block
QP_HD u64 salt_value(const SaltKey& key, u64 nonce, u64 index) { u64 o[8]; salt_block(key, nonce, index >> 3, o); return o[index & 7]; }
inline SaltKey salt_key_from_bytes(const uint8_t* b) {
  SaltKey k; for (int i = 0; i < 8; i++) k.k[i] = (u32)b[4 * i] | ((u32)b[4 * i + 1] << 8) | ((u32)b[4 * i + 2] << 16) | ((u32)b[4 * i + 3] << 24); return k;
}

}
