"""Host half of the product's verifier (csrc/verify.hip: transcript, vanishing identity over F_{p^2} / the extension algebra, proof of work,
FRI arithmetic, final polynomial) on CPU-made proofs of every synthetic circuit shape — no GPU needed, the Merkle openings are the device
half (tests/test_gpu_verifier.py). The proofs come from the oracle prover, the verdicts are compared with the oracle verifier's: two
independent restatements of qp-plonky2's verify (wormhole/verifier/src/lib.rs:155-159) must agree, recursion gates included."""
import ctypes, random
import numpy as np
import pytest
from harness import *
from test_oracle_prover import SHAPES


def host_verify(desc, proof):
    L = qpgpu()
    rc = L.qpgpu_test_verify_host(ctypes.byref(desc), proof, len(proof))
    assert rc in (0, -9), (rc, L.qpgpu_last_error())
    return None if rc == 0 else L.qpgpu_last_error().decode()


@pytest.mark.parametrize("preset", ["T", "TZ", "RT", "RT2"])
def test_host_half_accepts_oracle_proofs_and_rejects_tampering(preset):
    sc = SynthCircuit(preset); oc = OracleCircuit(sc.desc)
    rnd = random.Random(3)
    for seed in (1, 2):
        w, p = sc.witness(seed)
        pf = oc.prove(w, p, sc.salts(seed) if sc.desc.zero_knowledge else None)
        assert host_verify(sc.desc, pf) is None
        cap = 32 << sc.desc.fri_cap_height
        for pos in (3, cap + 7, 2 * cap + 1, 3 * cap + 5, 3 * cap + 16 * 40 + 2, len(pf) - 2, len(pf) - 8 * sc.desc.num_public_inputs - 12):
            bad = bytearray(pf); bad[pos] ^= 1 << rnd.randrange(8)
            assert host_verify(sc.desc, bytes(bad)) is not None and oc.verify(bytes(bad)) is not None, pos
    # a witness that breaks one gate row: the proof is still produced (as upstream) and the vanishing identity fails
    w, p = sc.witness(4); w = w.copy(); w[5, 2] = (int(w[5, 2]) + 1) % P
    pf = oc.prove(w, p, sc.salts(4) if sc.desc.zero_knowledge else None)
    assert "vanishing" in host_verify(sc.desc, pf) and oc.verify(pf) is not None


def test_every_recursion_gate_is_checked_by_the_host_half():
    """Perturbing a wire of each gate kind's first row must break the vanishing identity (RT carries all 14 gate types)."""
    sc = SynthCircuit("RT"); oc = OracleCircuit(sc.desc)
    pr = sc.params
    w0, p = sc.witness(9)
    assert host_verify(sc.desc, oc.prove(w0, p)) is None
    row = 3 + pr.n_poseidon + pr.n_basesum + pr.n_arith + pr.n_constant           # first recursion row (layout of synth.cc; see test_gpu_prove)
    for kind in range(8):
        if pr.n_recursion[kind] == 0:
            continue
        w = w0.copy(); w[1, row] = (int(w[1, row]) + 1) % P
        msg = host_verify(sc.desc, oc.prove(w, p))
        assert msg is not None and "vanishing" in msg, (kind, msg)
        row += pr.n_recursion[kind]


@pytest.mark.parametrize("shape", SHAPES[:4], ids=[str(i) for i in range(4)])
def test_host_half_on_varied_fri_shapes(shape):
    sc = SynthCircuit("T", **shape); oc = OracleCircuit(sc.desc)
    w, p = sc.witness(1)
    pf = oc.prove(w, p, sc.salts(1) if sc.desc.zero_knowledge else None)
    assert host_verify(sc.desc, pf) is None
    bad = bytearray(pf); bad[len(pf) // 2] ^= 2
    assert oc.verify(bytes(bad)) is not None          # (the host half alone may pass it: mid-proof bytes are mostly Merkle siblings, the device half's job)


def test_mutated_and_random_inputs_never_crash_the_host_side():
    """Byte-level fuzz of everything in the product that parses untrusted bytes on the host: the verifier's host half (proof bytes), the
    common / verifier-only parsers and the circuit-package reader. Every call must return a status (accepting is fine only for the
    untouched input)."""
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cb = open(os.path.join(G, "common.bin"), "rb").read(); vb = open(os.path.join(G, "verifier.bin"), "rb").read(); pb = open(os.path.join(G, "proof.bin"), "rb").read()
    shape = ReferenceCircuitShape(cb, vb[:552])
    L = qpgpu(); rnd = random.Random(77)
    def mutate(b):
        b = bytearray(b)
        for _ in range(rnd.choice([1, 1, 2, 8, 64])):
            pos = rnd.randrange(len(b))
            b[pos] = rnd.choice([0, 0xFF, b[pos] ^ (1 << rnd.randrange(8)), rnd.randrange(256)])
        return bytes(b)
    accepted = 0
    for _ in range(300):
        m = mutate(pb)
        rc = L.qpgpu_test_verify_host(ctypes.byref(shape.desc), m, len(m))
        assert rc in (0, -9); accepted += rc == 0
    assert accepted <= 220            # mutations that only hit Merkle siblings pass the host half (the device half catches them)
    d = CircuitDesc(); gates = (Gate * 16)(); sel = (ctypes.c_uint32 * 16)(); grp = (ctypes.c_uint32 * 16)(); kis = (ctypes.c_uint64 * 160)()
    cap = (ctypes.c_uint64 * 64)(); dig = (ctypes.c_uint64 * 4)()
    for _ in range(500):
        m = mutate(cb); cut = rnd.choice([len(m), len(m), rnd.randrange(len(m) + 1)])
        assert L.qpgpu_parse_common(m[:cut], cut, ctypes.byref(d), gates, 16, sel, grp, 8, kis, 160) in (0, -1)
        m = mutate(vb[:552]); cut = rnd.choice([len(m), rnd.randrange(len(m) + 1)])
        assert L.qpgpu_parse_verifier_only(m[:cut], cut, cap, 64, dig) in (0, 1, 2, 3, 4, -1, -5)      # returns the cap height on success
    sc = SynthCircuit("T")                            # keep the owner alive: desc points into it
    pk = write_package(sc.desc)
    ok = 0
    for _ in range(200):
        m = mutate(pk); cut = rnd.choice([len(m), len(m), rnd.randrange(len(m) + 1)])
        h = ctypes.c_void_p()
        rc = L.qpgpu_package_read(m[:cut], cut, ctypes.byref(h))
        assert rc in (0, -1)
        if rc == 0:
            ok += 1; L.qpgpu_package_free(h)
    assert ok == 0                                   # the checksum catches every mutation
    for n in (0, 1, 7, 8, 100, 5000):
        junk = bytes(rnd.randrange(256) for _ in range(n))
        h = ctypes.c_void_p()
        assert L.qpgpu_package_read(junk, n, ctypes.byref(h)) == -1
        assert L.qpgpu_parse_common(junk, n, ctypes.byref(d), gates, 16, sel, grp, 8, kis, 160) == -1
