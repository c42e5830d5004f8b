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
