"""The aggregation node circuit (include/qpcircuit.h qpc_aggregation_*; zk-circuits_amd/circuit/recursion.cc) on the CPU, with the oracle as
checker: aggregate_chunk of the reference (wormhole/aggregator/src/circuits/tree.rs:106-143) and its tests (tree.rs:189-243:
recursive_aggregation_tree, pair_aggregation, public_inputs_are_aggregated) on small inner circuits so that the oracle prover finishes in
seconds. The in-circuit verifier is this repo's own restatement of its host verifier (qp-plonky2's is not on disk): what is checked is that
a node circuit's witness exists exactly when every child proof verifies, that the node's proof verifies, and that levels compose."""
import importlib, os
import numpy as np
import pytest
from harness import *

pkg = importlib.import_module("zk-circuits_amd")
th = importlib.import_module("zk-circuits_amd.test_helpers")


@pytest.fixture(scope="module")
def vote_leaf():
    inner = BuiltCircuit(QPC_VOTE, False); oc = OracleCircuit(inner.desc)          # k = 8: one FRI layer of arity 16, final polynomial of 16 coefficients
    w, p = inner.witness(th.vote_test_inputs().fill_targets())
    return inner, oc, oc.prove(w, p), p


@pytest.fixture(scope="module")
def level1(vote_leaf):
    inner, oc, leaf, p = vote_leaf
    node = BuiltCircuit.aggregation(inner.desc, 2); on = OracleCircuit(node.desc)
    w, pis = node.witness(node.aggregation_targets(oc.cap(), inner.desc.circuit_digest, [leaf, leaf]))
    return node, on, w, pis


def test_pair_aggregation_and_public_inputs(vote_leaf, level1):
    """pair_aggregation + public_inputs_are_aggregated (tree.rs:218-243)."""
    inner, oc, leaf, p = vote_leaf
    node, on, w, pis = level1
    assert check_witness(on, w, pis) is None
    assert list(pis) == list(p) * 2 and node.desc.num_public_inputs == 2 * inner.desc.num_public_inputs
    assert node.desc.zero_knowledge == 0 and node.row_counts()["rows_in_use"] <= node.n
    proof = on.prove(w, pis)
    assert on.verify(proof) is None


def test_no_fri_layer_inner_circuit():
    """An inner circuit too small for any FRI reduction (nullifier fragment, k = 5: no layers, 32 final coefficients)."""
    inner = BuiltCircuit(QPC_NULLIFIER, False); oc = OracleCircuit(inner.desc)
    w, p = inner.witness(th.circuit_test_inputs().fill_targets(QPC_NULLIFIER))
    leaf = oc.prove(w, p)
    assert inner.desc.fri_num_layers == 0
    node = BuiltCircuit.aggregation(inner.desc, 1); on = OracleCircuit(node.desc)
    w1, p1 = node.witness(node.aggregation_targets(oc.cap(), inner.desc.circuit_digest, [leaf]))
    assert check_witness(on, w1, p1) is None and list(p1) == list(p)


def test_recursive_aggregation_tree_two_levels(vote_leaf, level1):
    """recursive_aggregation_tree (tree.rs:189-216): node proofs are themselves aggregated by the next level's circuit (inner circuit = a node
    circuit: 2^14 rows, three FRI layers); the root's public inputs are the four leaves'."""
    inner, oc, leaf, p = vote_leaf
    node, on, w, pis = level1
    child = on.prove(w, pis)
    top = BuiltCircuit.aggregation(node.desc, 2); ot = OracleCircuit(top.desc)
    assert top.desc.degree_bits == 14
    w2, p2 = top.witness(top.aggregation_targets(on.cap(), node.desc.circuit_digest, [child, child]))
    assert check_witness(ot, w2, p2) is None and list(p2) == list(p) * 4
    root = ot.prove(w2, p2)
    assert ot.verify(root) is None


def test_a_child_that_does_not_verify_cannot_be_witnessed(vote_leaf, level1):
    inner, oc, leaf, p = vote_leaf
    node, on, w, pis = level1
    rng = np.random.default_rng(7)
    L = len(leaf)
    positions = [0, 3 * 512 + 9, L - 1, L - 8 * len(p) - 3, L // 2] + [int(v) for v in rng.integers(0, L, 10)]     # caps, openings, public inputs, pow witness, queries
    positions += [int(v) for v in rng.integers(0, L, 25)]
    rejected = 0
    for pos in positions:                                                        # the node has a witness exactly when the host verifier accepts the child
        bad = bytearray(leaf); bad[pos] ^= 1 << int(rng.integers(0, 8))
        accepted = oc.verify(bytes(bad)) is None
        try:
            node.witness(node.aggregation_targets(oc.cap(), inner.desc.circuit_digest, [leaf, bytes(bad)]))
            witnessed = True
        except CircuitError:
            witnessed = False
        assert witnessed == accepted, (pos, accepted, witnessed)
        rejected += not accepted
    assert rejected >= len(positions) - 2
    wrong_cap = oc.cap().copy(); wrong_cap[3, 1] ^= np.uint64(1)                  # other verifier data: the constants/sigmas openings no longer reach the cap
    with pytest.raises(CircuitError):
        node.witness(node.aggregation_targets(wrong_cap, inner.desc.circuit_digest, [leaf, leaf]))
    wrong_digest = list(inner.desc.circuit_digest); wrong_digest[0] ^= 1          # other circuit digest: another transcript
    with pytest.raises(CircuitError):
        node.witness(node.aggregation_targets(oc.cap(), wrong_digest, [leaf, leaf]))


def test_zero_knowledge_children_give_a_zero_knowledge_node():
    """aggregate_chunk builds the node with the children's config (tree.rs:111); WormholeProofAggregator::default is the zk config
    (aggregator.rs:20-23). A zk node has blinding rows (2^15 rows for two children) and its proof carries salts."""
    inner = BuiltCircuit(QPC_VOTE, True); oc = OracleCircuit(inner.desc)
    w, p = inner.witness(th.vote_test_inputs().fill_targets(), bytes(range(32)))
    leaf = oc.prove(w, p, salts=SynthCircuit("B").salts_keyed(TEST_SALT_KEY, 1))
    node = BuiltCircuit.aggregation(inner.desc, 2); on = OracleCircuit(node.desc)
    assert node.desc.zero_knowledge == 1 and node.desc.degree_bits == 15
    x = node.aggregation_targets(oc.cap(), inner.desc.circuit_digest, [leaf, leaf])
    with pytest.raises(CircuitError, match="blinding key"):
        node.witness(x)
    w1, p1 = node.witness(x, bytes(range(32)))
    assert check_witness(on, w1, p1) is None and list(p1) == list(p) * 2


def test_argument_errors(vote_leaf, level1):
    inner, oc, leaf, p = vote_leaf
    node, on, w, pis = level1
    with pytest.raises(CircuitError, match="takes 2 proofs"):
        node.aggregation_targets(oc.cap(), inner.desc.circuit_digest, [leaf])
    with pytest.raises(CircuitError, match="proof length"):
        node.aggregation_targets(oc.cap(), inner.desc.circuit_digest, [leaf[:-8], leaf[:-8]])
    with pytest.raises(CircuitError, match="does not cover gate tag"):
        rt = SynthCircuit("RT")                                                   # (kept alive: the descriptor points into it)
        BuiltCircuit.aggregation(rt.desc, 2)                                      # recursion-shaped synthetic circuit: gates outside the leaf set
    leaf_shape = BuiltCircuit(QPC_WORMHOLE, False)
    with pytest.raises(CircuitError, match="not an aggregation circuit"):
        leaf_shape.aggregation_targets(oc.cap(), inner.desc.circuit_digest, [leaf])


def test_synthetic_leaf_shape_aggregates_too():
    """The synthetic wormhole-shaped circuits (bench shapes) use the leaf gate set, so their proofs aggregate through the same node circuits."""
    sc = SynthCircuit("T"); oc = OracleCircuit(sc.desc)
    w, p = sc.witness(3)
    leaf = oc.prove(w, p)
    node = BuiltCircuit.aggregation(sc.desc, 2); on = OracleCircuit(node.desc)
    w1, p1 = node.witness(node.aggregation_targets(oc.cap(), sc.desc.circuit_digest, [leaf, leaf]))
    assert check_witness(on, w1, p1) is None


def test_node_circuit_verifies_the_reference_golden_proof_in_circuit():
    """The in-circuit verifier against reference data: a node circuit for the shape of wormhole/bench-data/common.bin, given the reference's
    verifier data (verifier.bin: constants/sigmas cap + circuit digest) and the golden proof the Rust prover made (proof.bin), has a witness —
    every transcript challenge, the vanishing identity, the FRI checks and all 280 Merkle openings hold as constraints — and forwards the
    proof's 16 public inputs; any forged byte of the proof leaves it without one."""
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rd = lambda f: open(os.path.join(g, f), "rb").read()
    common, vb, proof = rd("common.bin"), rd("verifier.bin"), rd("proof.bin")
    vo = vb[:len(vb) - len(common)] if vb.endswith(common) and len(vb) > len(common) else vb
    shape = ReferenceCircuitShape(common, vo)
    node = BuiltCircuit.aggregation(shape.desc, 1); on = OracleCircuit(node.desc)
    assert node.desc.zero_knowledge == 1 and node.desc.degree_bits == 14                  # the child's config (zk); 7,150 rows + 8,326 blinding rows
    w, p = node.witness(node.aggregation_targets(shape.constants_sigmas_cap, shape.desc.circuit_digest, [proof]), bytes(range(32)))
    assert check_witness(on, w, p) is None
    assert list(p) == list(np.frombuffer(proof[-128:], dtype="<u8"))
    rng = np.random.default_rng(11)
    for pos in [len(proof) - 1, 5, 3 * 512 + 40, len(proof) - 140] + [int(v) for v in rng.integers(0, len(proof), 8)]:
        bad = bytearray(proof); bad[pos] ^= 1 << int(rng.integers(0, 8))
        with pytest.raises(CircuitError):
            node.witness(node.aggregation_targets(shape.constants_sigmas_cap, shape.desc.circuit_digest, [bytes(bad)]), bytes(range(32)))
