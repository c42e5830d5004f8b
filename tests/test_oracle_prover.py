"""The oracle prover (CPU restatement of prove_with_partition_witness) against the oracle verifier, on the synthetic
circuits; plus witness-consistency and error exits. No GPU."""
import numpy as np
import pytest
from harness import *


@pytest.fixture(scope="module")
def tiny():
    sc = SynthCircuit("T"); return sc, OracleCircuit(sc.desc)


@pytest.fixture(scope="module")
def tiny_zk():
    sc = SynthCircuit("TZ"); return sc, OracleCircuit(sc.desc)


def test_synthetic_witness_satisfies_gates_and_copy_constraints(tiny, tiny_zk):
    for sc, oc in (tiny, tiny_zk):
        for seed in range(3):
            w, p = sc.witness(seed)
            assert check_witness(oc, w, p) is None


def test_invalid_witness_detected(tiny):
    sc, oc = tiny
    w, p = sc.witness(1)
    w2 = w.copy(); w2[3, 5] ^= np.uint64(1)                 # a Poseidon-row wire
    assert check_witness(oc, w2, p) is not None


def test_prove_verify_and_determinism(tiny):
    sc, oc = tiny
    w, p = sc.witness(11)
    a = oc.prove(w, p); b = oc.prove(w, p, threads=1)
    assert a == b and oc.verify(a) is None                  # deterministic incl. the smallest-witness PoW rule; thread-count invariant


def test_zk_proof_depends_on_salts_only_where_expected(tiny_zk):
    sc, oc = tiny_zk
    w, p = sc.witness(5)
    a = oc.prove(w, p, sc.salts(1)); b = oc.prove(w, p, sc.salts(2))
    assert oc.verify(a) is None and oc.verify(b) is None and a != b
    with pytest.raises(RuntimeError, match="salts"):
        oc.prove(w, p, None)


def test_invalid_witness_gives_rejected_proof(tiny):
    sc, oc = tiny
    w, p = sc.witness(2)
    w[40, 9] = (int(w[40, 9]) + 1) % P                      # breaks a gate/copy constraint; prover still emits bytes (quotient has no remainder check here)
    pf = oc.prove(w, p)
    assert oc.verify(pf) is not None


def test_supplied_pow_witness(tiny):
    sc, oc = tiny
    w, p = sc.witness(4)
    base = oc.prove(w, p)
    off = len(base) - 8 * sc.desc.num_public_inputs - 16
    pw = int.from_bytes(base[off:off + 8], "little")
    assert oc.prove(w, p, pow_witness=pw) == base
    with pytest.raises(RuntimeError, match="pow_witness"):
        oc.prove(w, p, pow_witness=(pw + 1) % P)


@pytest.mark.parametrize("preset", ["A"])
def test_leaf_shape_proof_size_and_verify(preset):
    sc = SynthCircuit(preset); oc = OracleCircuit(sc.desc)
    w, p = sc.witness(0x51570000)
    pf = oc.prove(w, p)
    assert len(pf) == 132712 and oc.verify(pf) is None


SHAPES = [
    dict(degree_bits=5, fri_arity_bits=[1], n_poseidon=2, n_basesum=4, n_arith=4, n_constant=1, n_copies=30),
    dict(degree_bits=7, fri_arity_bits=[3, 3]),                                                       # last layer tree == its cap (empty paths)
    dict(degree_bits=7, fri_arity_bits=[1, 2, 4], fri_cap_height=2),
    dict(degree_bits=7, fri_arity_bits=[2, 2, 2], fri_cap_height=0xFFFFFFFF, fri_num_queries=5),     # cap height 0: a single root
    dict(degree_bits=9, fri_arity_bits=[4, 4], fri_num_queries=40, fri_pow_bits=10),
    dict(degree_bits=6, fri_arity_bits=[], fri_pow_bits=0xFFFFFFFF),                                  # no FRI reduction, no grinding
    dict(degree_bits=9, fri_arity_bits=[4], zero_knowledge=1, fri_cap_height=3),
]


@pytest.mark.parametrize("shape", SHAPES, ids=[str(i) for i in range(len(SHAPES))])
def test_oracle_prove_verify_on_varied_shapes(shape):
    sc = SynthCircuit("T", **shape); oc = OracleCircuit(sc.desc)
    w, p = sc.witness(77)
    pf = oc.prove(w, p, sc.salts(77) if sc.desc.zero_knowledge else None)
    assert oc.verify(pf) is None
    bad = bytearray(pf); bad[len(bad) // 2] ^= 0x10
    assert oc.verify(bytes(bad)) is not None


# ---- recursive-verifier gate set. Layouts are the recalled upstream ones: parity with the Rust
# prover is UNPINNED (the reference ships no aggregated proof and no recursion-circuit common data); what is checked is that the
# base-field evaluator (prover) and the extension-algebra evaluator (verifier) agree and that every gate wire is constrained.
REC_KINDS = ["ArithmeticExtension", "MulExtension", "Reducing", "ReducingExtension", "RandomAccess", "PoseidonMds",
             "Exponentiation", "CosetInterpolation"]
REC_USED_WIRES = {"RT": [80, 78, 133, 132, 90, 48, 134, 47],      # standard_recursion_config parameters
                  "RT2": [56, 54, 55, 48, 41, 48, 44, 35]}        # second parameter set (7/9 ops, 17/11 coeffs, RA{2,5,1}, 21 bits, coset{3,3})
RA_SHAPE = {"RT": (4, 4), "RT2": (2, 5)}                          # RandomAccess (bits, copies)


@pytest.fixture(scope="module", params=["RT", "RT2"])
def rec_tiny(request):
    sc = SynthCircuit(request.param); sc.preset = request.param; return sc, OracleCircuit(sc.desc)


def rec_first_row(sc, kind):
    p = sc.params
    return 3 + p.n_poseidon + p.n_basesum + p.n_arith + p.n_constant + sum(p.n_recursion[i] for i in range(kind))


def test_recursion_gate_set_selectors_follow_upstream_grouping(rec_tiny):
    sc, _ = rec_tiny
    d = sc.desc
    assert d.num_gates == 14
    want = {"RT": [0, 6, 6, 10, 10, 12, 12, 13, 13, 14],          # degrees 0,1,1,1,2,2 | 2,3,3,3 | 4,5 | 6 | 7
            "RT2": [0, 6, 6, 11, 11, 13, 13, 14]}[sc.preset]       # degrees 0,1,1,1,2,2 | 2,3,3,3,3 | 3,4 | 7
    assert [d.selector_group_ranges[i] for i in range(2 * d.num_selector_groups)] == want
    assert d.num_constants == d.num_selector_groups + 2


def test_recursion_gates_witness_prove_verify(rec_tiny):
    sc, oc = rec_tiny
    w, p = sc.witness(3)
    assert check_witness(oc, w, p) is None
    pf = oc.prove(w, p)
    assert oc.verify(pf) is None


@pytest.mark.parametrize("kind", range(8), ids=REC_KINDS)
def test_recursion_gate_constrains_every_wire(rec_tiny, kind):
    sc, oc = rec_tiny
    w, p = sc.witness(4)
    row = rec_first_row(sc, kind)
    bits, copies = RA_SHAPE[sc.preset]; stride = 2 + (1 << bits)
    for col in range(REC_USED_WIRES[sc.preset][kind]):
        if kind == 4 and col < stride * copies and col % stride >= 2 and col % stride - 2 != int(w[col - col % stride, row]):
            continue                                         # RandomAccess: list items other than list[index] are free
        w2 = w.copy(); w2[col, row] = (int(w2[col, row]) + 1) % P
        assert check_witness(oc, w2, p) is not None, "wire %d of a %s row is unconstrained" % (col, REC_KINDS[kind])


def test_recursion_gate_violation_gives_rejected_proof(rec_tiny):
    sc, oc = rec_tiny
    w, p = sc.witness(5)
    w[9, rec_first_row(sc, 7)] = (int(w[9, rec_first_row(sc, 7)]) + 1) % P      # a CosetInterpolation value wire
    assert oc.verify(oc.prove(w, p)) is not None


def _selfcheck():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "selfcheck.json")))


@pytest.mark.parametrize("key", sorted(k for k in _selfcheck() if not k.startswith("A/")))
def test_oracle_proofs_match_the_committed_self_check_digests(key):
    """Regression fixture of this repo's own restatement (tests/golden/make_selfcheck.py) — not a reference output."""
    import hashlib
    preset, seed = key.split("/"); seed = int(seed)
    sc = SynthCircuit(preset); oc = OracleCircuit(sc.desc)
    w, p = sc.witness(seed)
    pf = oc.prove(w, p, sc.salts(seed) if sc.desc.zero_knowledge else None)
    want = _selfcheck()[key]
    assert len(pf) == want["len"] and hashlib.sha256(pf).hexdigest() == want["sha256"]
    assert hashlib.sha256(oc.cap().tobytes()).hexdigest() == want["cap_sha256"]


def test_oracle_and_synth_under_sanitizers():
    """`make -C oracle asan`: the checker and the synthetic generator rebuilt with AddressSanitizer + UBSan and driven by
    oracle/selftest.cc (golden proof verification, malformed-input rejection, prove/verify/tamper on T, TZ and RT). Host code only —
    GPU ASan is not available on this pool."""
    import subprocess
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "selftest ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
