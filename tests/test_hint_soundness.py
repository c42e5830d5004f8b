"""Hint-soundness mutation tests for the circuits this repo authored (CPU; the oracle is the checker).

A witness generator HINTS some values instead of deriving them through a gate: inverses (is_equal, barycentric folds), extension-field
inverses, 32-bit halves (canonical index decomposition), BaseSum limbs. Each must be pinned by constraints, or a dishonest prover picks it
freely — in the aggregation node that would be a forgeable aggregate (the statement to hold: a node verifies its children,
/root/reference/wormhole/aggregator/src/circuits/tree.rs:117-121; for the leaf circuit the checks of wormhole/circuit/src/*.rs).

For EVERY hint of a circuit and every mutation mode libqpcircuit's sweep (zk-circuits_amd/circuit/mutation.cc) substitutes another value,
re-derives everything downstream and hands the wire matrix to the ORACLE's gate-constraint check; sampled mutations get the oracle's
full check (every row + copy constraints). Allowed outcomes: killed, blocked by a copy constraint, or benign (nothing but the hint's own
cell moved and the public inputs are unchanged — e.g. is_equal's inverse when the difference is zero). A survivor fails the test.
Second sweep: every input target flipped AFTER generation with nothing re-derived must be rejected."""
import collections
import ctypes
import os
import numpy as np
import pytest
from harness import (BuiltCircuit, OracleCircuit, ReferenceCircuitShape, check_witness, circuit_lib, oracle, ptr, u64p,
                     QPC_WORMHOLE, QPC_NULLIFIER, QPC_UNSPENDABLE_ACCOUNT, QPC_STORAGE_PROOF, QPC_VOTE)
import importlib

pkg = importlib.import_module("zk-circuits_amd")
th = pkg.test_helpers
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

KINDS = {2: "inverse", 3: "base-sum limbs", 5: "extension inverse", 6: "lo/hi halves", 100: "input flip",
         7: "ArithmeticExtension output", 8: "Reducing accumulators", 9: "ReducingExtension accumulators", 10: "RandomAccess bits / claimed element",
         11: "PoseidonMds outputs", 12: "CosetInterpolation shifted point / intermediates / value", 13: "Exponentiation intermediates"}
VERDICTS = {0: "killed", 1: "benign", 2: "SURVIVED", 3: "INCONSISTENT", 4: "skipped", 5: "no cell", 6: "blocked"}


class MutationResult(ctypes.Structure):
    _fields_ = [("index", ctypes.c_uint32), ("kind", ctypes.c_uint32), ("mode", ctypes.c_uint32), ("verdict", ctypes.c_uint32),
                ("conflicts", ctypes.c_uint64), ("changed_classes", ctypes.c_uint64), ("rows", ctypes.c_uint64)]


CHECK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int)


def sweep(circuit, inputs, what, step=1, first=0, modes=0xFF, full_every=97, blinding_key=None, cap=200000):
    """Runs the sweep with the oracle as checker. Returns (results, stats)."""
    L, O = circuit_lib(), oracle()
    L.qpc_circuit_num_hints.restype = ctypes.c_size_t
    L.qpc_circuit_num_hints.argtypes = [ctypes.c_void_p]
    L.qpc_mutation_sweep.argtypes = [ctypes.c_void_p, u64p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t,
                                     ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32, CHECK_FN, ctypes.c_void_p, ctypes.POINTER(MutationResult),
                                     ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    O.orc_check_witness.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    O.orc_check_witness_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    oc = OracleCircuit(circuit.desc)
    stats = {"calls": 0, "full": 0, "disagree": 0}

    def cb(_ctx, wires, pis, rows, nrows, full):
        stats["calls"] += 1
        if nrows == 0:                                        # the honest matrix: the whole check
            return 0 if O.orc_check_witness(oc.h, wires, pis) == 0 else 1
        rc = O.orc_check_witness_rows(oc.h, wires, pis, rows, nrows)
        if full:                                              # sampled: the rows shortcut must agree with the check of every row + copy constraints
            stats["full"] += 1
            if (O.orc_check_witness(oc.h, wires, pis) == 0) != (rc == 0):
                stats["disagree"] += 1
                return -1
        return 0 if rc == 0 else 1
    fn = CHECK_FN(cb)
    inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
    out = (MutationResult * cap)()
    count = ctypes.c_size_t()
    rc = L.qpc_mutation_sweep(circuit.handle, ptr(inputs), len(inputs), None if blinding_key is None else bytes(blinding_key), what, first, step,
                              modes, 0x5157C0DE, full_every, fn, None, out, cap, ctypes.byref(count))
    assert rc == 0, L.qpc_last_error().decode()
    assert stats["disagree"] == 0
    return [out[i] for i in range(count.value)], stats


def summarize(results):
    table = collections.Counter((KINDS.get(r.kind, r.kind), VERDICTS[r.verdict]) for r in results)
    return dict(table)


def assert_sound(results, name, min_hints=1):
    bad = [(r.index, KINDS.get(r.kind, r.kind), r.mode, VERDICTS[r.verdict], int(r.changed_classes)) for r in results if r.verdict in (2, 3)]
    table = summarize(results)
    hints = len({r.index for r in results})
    killed = sum(1 for r in results if r.verdict in (0, 6))
    print("\n%s: %d hints / %d mutations / %d killed or blocked / %d benign / %d skipped / %d survivors  %s"
          % (name, hints, len(results), killed, sum(r.verdict == 1 for r in results), sum(r.verdict in (4, 5) for r in results), len(bad), table))
    assert not bad, "%s: under-constrained hints (index, kind, mode, verdict, other values moved): %r" % (name, bad[:20])
    assert hints >= min_hints
    return table


def _wormhole_inputs(kind):
    return th.circuit_test_inputs().fill_targets(kind)


@pytest.mark.parametrize("kind,name,min_hints", [(QPC_NULLIFIER, "nullifier", 10), (QPC_UNSPENDABLE_ACCOUNT, "unspendable account", 2),
                                                 (QPC_VOTE, "vote", 30)])
def test_small_circuits_every_hint_every_mode(kind, name, min_hints):
    c = BuiltCircuit(kind, False)
    x = th.vote_test_inputs().fill_targets() if kind == QPC_VOTE else _wormhole_inputs(kind)
    res, stats = sweep(c, x, 0, full_every=7)
    assert_sound(res, name, min_hints)
    assert stats["full"] >= 1
    flips, _ = sweep(c, x, 1, full_every=5)
    t = assert_sound(flips, name + " (input flips)")
    assert all(r.verdict in (0, 5) for r in flips), t                       # an input is either checked by a gate or carried by no cell at all


def test_wormhole_circuit_every_hint():
    """The whole leaf circuit (20 storage-proof nodes: ~3,600 is_equal inverses, ~3,800 range checks): every hint, every mode."""
    c = BuiltCircuit(QPC_WORMHOLE, False)
    x = _wormhole_inputs(QPC_WORMHOLE)
    n = circuit_lib().qpc_circuit_num_hints
    n.restype, n.argtypes = ctypes.c_size_t, [ctypes.c_void_p]
    assert n(c.handle) > 7000
    res, stats = sweep(c, x, 0, full_every=499)
    t = assert_sound(res, "wormhole")
    assert len({r.index for r in res}) == n(c.handle) and stats["full"] >= 10
    assert t.get(("inverse", "killed"), 0) > 1000 and t.get(("base-sum limbs", "killed"), 0) + t.get(("base-sum limbs", "blocked"), 0) > 7000
    flips, _ = sweep(c, x, 1, full_every=211)
    assert_sound(flips, "wormhole (input flips)")
    unused = [r.index for r in flips if r.verdict == 5]
    assert all(r.verdict in (0, 5) for r in flips) and len(unused) < len(flips) // 2


def _golden_node(with_shape=False):
    rd = lambda f: open(os.path.join(GOLD, f), "rb").read()
    common, vb, proof = rd("common.bin"), rd("verifier.bin"), rd("proof.bin")
    vo = vb[:len(vb) - len(common)] if vb.endswith(common) and len(vb) > len(common) else vb
    shape = ReferenceCircuitShape(common, vo)
    node = BuiltCircuit.aggregation(shape.desc, 1, False)
    x = node.aggregation_targets(shape.constants_sigmas_cap, shape.desc.circuit_digest, [proof])
    return (node, x, shape, proof) if with_shape else (node, x)


def test_aggregation_node_every_hint_on_the_reference_golden_proof():
    """The in-circuit verifier, on the Rust prover's own proof (bench-data/proof.bin under common.bin / verifier.bin): every hinted inverse,
    extension inverse, 32-bit half and limb of the node that verifies it is mutated; none may survive."""
    node, x = _golden_node()
    w, p = node.witness(x)
    assert check_witness(OracleCircuit(node.desc), w, p) is None
    res, stats = sweep(node, x, 0, full_every=997)
    t = assert_sound(res, "aggregation node over the golden proof", min_hints=2500)
    assert stats["full"] >= 5
    kinds = {k for k, _ in t}
    assert {"inverse", "extension inverse", "lo/hi halves", "base-sum limbs", "ArithmeticExtension output", "Reducing accumulators", "ReducingExtension accumulators",
            "RandomAccess bits / claimed element", "PoseidonMds outputs", "CosetInterpolation shifted point / intermediates / value"} <= kinds, kinds


def test_aggregation_node_rejects_every_flipped_proof_cell():
    """Every input cell of the node (the child's caps, openings, query rows, Merkle siblings, final polynomial, proof-of-work witness and public
    inputs, and the verifier data) flipped AFTER generation: the matrix must violate a gate. The one exception is by design: the verifier
    data's constants/sigmas cap is a private input of the node (tree.rs:113-114) that nothing hashes, so a cap entry that none of this
    proof's 28 queries lands under multiplies a zero indicator and is free; every entry a query does select is pinned."""
    from harness import replay_transcript
    node, x, shape, proof = _golden_node(with_shape=True)
    flips, _ = sweep(node, x, 1, full_every=1499, modes=1)
    d = shape.desc
    selected = {q >> (d.degree_bits + d.fri_rate_bits - d.fri_cap_height) for q in replay_transcript(d, proof)["query_indices"]}
    free_cap = {4 * j + e for j in range(1 << d.fri_cap_height) if j not in selected for e in range(4)}       # the cap comes first among the inputs
    assert 0 < len(free_cap) < 4 * 8
    survivors = {r.index for r in flips if r.verdict == 2}
    assert survivors == free_cap, (sorted(survivors), sorted(free_cap))
    rest = [r for r in flips if r.index not in free_cap]
    assert_sound(rest, "aggregation node (input flips)", min_hints=len(x) // 2)
    assert all(r.verdict in (0, 5) for r in rest)
    no_cell = [r.index for r in rest if r.verdict == 5]
    assert len(no_cell) <= len(x) // 50, "inputs that reach no gate: %r" % no_cell[:40]


def test_the_sweep_finds_planted_under_constrained_hints():
    """Negative control: the builder-fuzz programs (qpc_test_random_circuit) contain, on purpose, an extension inverse nothing constrains and a
    32-bit decomposition without the canonical-form check. The sweep must report survivors there — an unpinned extension inverse, and the
    non-canonical (x + p) pair of a lo/hi split — while the constrained hints of the same programs (is_equal, inverse * x = 1, limbs) die."""
    seen = collections.Counter()
    for seed in range(8):
        c, inputs, _expected = BuiltCircuit.random_program(seed, zero_knowledge=False, num_inputs=8 + seed, num_ops=150 + 40 * seed)
        res, _ = sweep(c, inputs, 0, full_every=13)
        for r in res:
            seen[(KINDS.get(r.kind), r.mode, VERDICTS[r.verdict])] += 1
        assert not [r for r in res if r.verdict == 3]
        assert not [r for r in res if r.kind == 3 and r.verdict in (1, 2)]                     # limbs: always pinned by the BaseSum row
    assert sum(v for (k, m, vd), v in seen.items() if k == "extension inverse" and vd in ("SURVIVED", "benign")) > 0, seen
    assert seen[("lo/hi halves", 3, "SURVIVED")] > 0, seen                                      # missing canonicity check found
    assert not [k for k in seen if k[0] == "lo/hi halves" and k[1] in (0, 1, 4) and k[2] == "SURVIVED"], seen   # sum and ranges are pinned


def test_node_over_node_every_hint():
    """One level up: a node that verifies a NODE proof evaluates the recursion gates' own constraints in-circuit (ArithmeticExtension, Reducing,
    ReducingExtension, RandomAccess, PoseidonMds in the extension algebra). Same sweep on that circuit: leaf = the nullifier fragment on the
    reference's inputs, level-0 node proved by the oracle, level-1 node mutated."""
    leaf = BuiltCircuit(QPC_NULLIFIER, False)
    w, p = leaf.witness(_wormhole_inputs(QPC_NULLIFIER))
    ol = OracleCircuit(leaf.desc)
    n0 = BuiltCircuit.aggregation(leaf.desc, 2, False)
    w0, p0 = n0.witness(n0.aggregation_targets(ol.cap(), leaf.desc.circuit_digest, [ol.prove(w, p)] * 2))
    o0 = OracleCircuit(n0.desc)
    pf0 = o0.prove(w0, p0)
    assert o0.verify(pf0) is None and n0.desc.num_gates > 6                       # the inner circuit carries recursion gates
    n1 = BuiltCircuit.aggregation(n0.desc, 1, False)
    x1 = n1.aggregation_targets(o0.cap(), n0.desc.circuit_digest, [pf0])
    res, stats = sweep(n1, x1, 0, full_every=1499)
    t = assert_sound(res, "node over a node proof", min_hints=2000)
    assert stats["full"] >= 3 and ("RandomAccess bits / claimed element", "killed") in t and ("Reducing accumulators", "killed") in t
    flips, _ = sweep(n1, x1, 1, full_every=2999, modes=1)
    from harness import replay_transcript
    d = n0.desc
    selected = {q >> (d.degree_bits + d.fri_rate_bits - d.fri_cap_height) for q in replay_transcript(d, pf0)["query_indices"]}
    free_cap = {4 * j + e for j in range(1 << d.fri_cap_height) if j not in selected for e in range(4)}
    assert {r.index for r in flips if r.verdict == 2} == free_cap
    assert all(r.verdict in (0, 5) for r in flips if r.index not in free_cap)
