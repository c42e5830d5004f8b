"""Shared test harness: the product binding (zk-circuits_amd/ffi.py: C-ABI + synthetic generator) re-exported next to the oracle
binding (oracle/binding.py, the checker) plus a few test helpers.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/ (it is the checker).
"""
import ctypes, importlib, os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zk-circuits_amd")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
_ffi = importlib.import_module("zk-circuits_amd.ffi")
from oracle.binding import oracle, OracleCircuit          # noqa: E402,F401

P = _ffi.P
u64p, u32p, u8p, ptr = _ffi.u64p, _ffi.u32p, _ffi.u8p, _ffi.ptr
Gate, CircuitDesc, SynthParams, ProveOpts = _ffi.Gate, _ffi.CircuitDesc, _ffi.SynthParams, _ffi.ProveOpts
qpgpu, synth, gpu_check, TEST_SALT_KEY = _ffi.qpgpu, _ffi.synth, _ffi.gpu_check, _ffi.TEST_SALT_KEY
SynthCircuit, GpuCircuit, ReferenceCircuitShape, replay_transcript = _ffi.SynthCircuit, _ffi.GpuCircuit, _ffi.ReferenceCircuitShape, _ffi.replay_transcript
GpuVerifier = _ffi.GpuVerifier
write_common, write_verifier_only, write_package, CircuitPackage = _ffi.write_common, _ffi.write_verifier_only, _ffi.write_package, _ffi.CircuitPackage


BuiltCircuit, CircuitError, circuit_lib = _ffi.BuiltCircuit, _ffi.CircuitError, _ffi.circuit_lib
QPC_WORMHOLE, QPC_NULLIFIER, QPC_UNSPENDABLE_ACCOUNT, QPC_STORAGE_PROOF, QPC_EXIT_ACCOUNT, QPC_VOTE = range(6)


def check_witness(oc, wires, pis):
    """Oracle check of a full witness: every gate constraint on every row and every copy constraint (sigma). None or the failure."""
    L = oracle(); L.orc_check_witness.argtypes = [ctypes.c_void_p, u64p, u64p]
    rc = L.orc_check_witness(oc.h, ptr(wires), ptr(pis))
    return None if rc == 0 else L.orc_last_error().decode()


def rand_felts(rng, shape):
    """Seeded canonical field elements, with edge values sprinkled in."""
    a = rng.integers(0, P, size=shape, dtype=np.uint64)
    flat = a.reshape(-1)
    edge = np.array([0, 1, P - 1, P - 2, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 32, (1 << 63)], dtype=np.uint64)
    k = min(len(edge), flat.size)
    flat[:k] = edge[:k]
    return a


def bitrev_indices(log_n):
    n = 1 << log_n
    idx = np.arange(n, dtype=np.uint64)
    r = np.zeros(n, dtype=np.uint64)
    for b in range(log_n):
        r |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(log_n - 1 - b)
    return r.astype(np.int64)


def keyed_salts(key, stream, N):
    """The [3][4][N] salt columns of a zk proof under (32-byte key, stream): ChaCha8 keystream of csrc/salt.h in numpy, for ANY N
    (SynthCircuit.salts_keyed only serves its own circuit's N). Element index = (oracle*4 + column)*N + leaf, eight elements per block."""
    key = bytes(key)
    assert len(key) == 32 and (12 * N) % 8 == 0
    nblk = 12 * N // 8
    k = np.frombuffer(key, dtype="<u4").astype(np.uint32)
    blk = np.arange(nblk, dtype=np.uint64)
    s = np.empty((16, nblk), dtype=np.uint32)
    for i, c in enumerate((0x61707865, 0x3320646e, 0x79622d32, 0x6b206574)):
        s[i] = c
    for i in range(8):
        s[4 + i] = k[i]
    s[12] = (blk & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    s[13] = (blk >> np.uint64(32)).astype(np.uint32)
    s[14] = np.uint32(stream & 0xFFFFFFFF)
    s[15] = np.uint32((stream >> 32) & 0xFFFFFFFF)
    x = s.copy()

    def rotl(v, r):
        return (v << np.uint32(r)) | (v >> np.uint32(32 - r))

    def qr(a, b, c, d):
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 16)
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 12)
        x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl(x[d], 8)
        x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl(x[b], 7)
    with np.errstate(over="ignore"):
        for _ in range(4):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        x += s
    z = x[0::2].astype(np.uint64) | (x[1::2].astype(np.uint64) << np.uint64(32))          # [8][nblk]
    z = np.where(z >= np.uint64(P), z - np.uint64(P), z)
    return np.ascontiguousarray(z.T).reshape(3, 4, N)
