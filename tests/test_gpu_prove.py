"""Whole-proof parity through the C-ABI: GPU proof bytes == oracle proof bytes for the same (circuit, witness, salts,
pow_witness), and the oracle verifier (which accepts the reference's golden bench-data proof) accepts them.
Mirrors the reference's prover/verifier tests (wormhole/tests/src/prover/prover_tests.rs:14-45,
wormhole/tests/src/verifier/verifier_tests.rs:14-66)."""
import numpy as np
import pytest
from harness import *

pytestmark = pytest.mark.gpu


def first_diff(a, b):
    if len(a) != len(b):
        return "length %d vs %d" % (len(a), len(b))
    x = np.frombuffer(a, dtype=np.uint8); y = np.frombuffer(b, dtype=np.uint8)
    d = np.nonzero(x != y)[0]
    return None if d.size == 0 else "first differing byte at %d (of %d), %d bytes differ" % (d[0], len(a), d.size)


@pytest.fixture(scope="module")
def circuits():
    cache = {}

    def get(preset):
        if preset not in cache:
            sc = SynthCircuit(preset)
            cache[preset] = (sc, OracleCircuit(sc.desc), GpuCircuit(sc.desc))
        return cache[preset]
    yield get
    for sc, oc, gc in cache.values():
        gc.close()


@pytest.mark.parametrize("preset", ["T", "TZ", "A", "B"])
def test_proof_bytes_match_oracle(circuits, preset):
    sc, oc, gc = circuits(preset)
    assert np.array_equal(gc.cap(), oc.cap())                       # preprocessed commitment
    for seed in (0x51570000, 0x51570001):
        wires, pis = sc.witness(seed)
        zk = bool(sc.desc.zero_knowledge)
        salts = sc.salts(seed) if zk else None
        want = oc.prove(wires, pis, salts)
        got = gc.prove(wires, pis, salt_seed=seed)                  # device-derived salts == qpsynth_salts(seed)
        assert len(got) == gc.proof_size == len(want)
        assert first_diff(got, want) is None
        assert oc.verify(got) is None


def test_explicit_salts_and_pow_witness(circuits):
    sc, oc, gc = circuits("TZ")
    wires, pis = sc.witness(7)
    salts = sc.salts(99)
    base = oc.prove(wires, pis, salts)
    npi = sc.desc.num_public_inputs
    off = gc.proof_size - 8 * npi - 16                                   # ... | pow_witness | u64 count | public inputs
    powo = int.from_bytes(base[off: off + 8], "little")
    got = gc.prove(wires, pis, salts=salts, pow_witness=powo)
    assert first_diff(got, base) is None
    with pytest.raises(RuntimeError, match="pow_witness"):
        gc.prove(wires, pis, salts=salts, pow_witness=(powo + 1) % P)      # almost surely invalid


def test_tampered_proof_rejected(circuits):
    sc, oc, gc = circuits("T")
    wires, pis = sc.witness(3)
    pf = bytearray(gc.prove(wires, pis))
    assert oc.verify(bytes(pf)) is None
    for pos in (len(pf) - 1, len(pf) - 8 * 16, 100, 2000):              # public inputs, a cap, an opening
        bad = bytearray(pf); bad[pos] ^= 1
        assert oc.verify(bytes(bad)) is not None


def test_batch_matches_single(circuits):
    sc, oc, gc = circuits("T")
    ws, ps = zip(*[sc.witness(100 + i) for i in range(6)])
    proofs = gc.prove_batch(list(ws), list(ps), salt_seed=5)
    for i, pf in enumerate(proofs):
        assert first_diff(pf, oc.prove(ws[i], ps[i])) is None


def test_bad_arguments(circuits):
    sc, oc, gc = circuits("T")
    wires, pis = sc.witness(1)
    L = qpgpu()
    buf = ctypes.create_string_buffer(16)
    n = ctypes.c_size_t()
    rc = L.qpgpu_prove(gc.h, wires.ctypes.data, ptr(pis), len(pis), None, buf, 16, ctypes.byref(n))
    assert rc == -5 and b"too small" in L.qpgpu_last_error()
    bad = pis.copy(); bad[0] = P
    with pytest.raises(RuntimeError, match="non-canonical"):
        gc.prove(wires, bad)
