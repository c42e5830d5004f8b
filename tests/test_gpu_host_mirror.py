"""The host-side mirror of the reference wrappers on a real GPU: WormholeProver new/commit/prove (reference:
wormhole/prover/src/lib.rs:190-237, tests wormhole/tests/src/prover/prover_tests.rs:14-45) and the aggregator's leaf
batch (wormhole/tests/src/aggregator/aggregator_tests.rs:11-112), plus a large Merkle tree through the fused leaf kernel."""
import importlib, os
import numpy as np
import pytest
from harness import *

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("zk-circuits_amd")


def test_commit_and_prove(pkg):                                  # prover_tests.rs:14-18
    cfg = pkg.CircuitConfig.test_config()
    prover = pkg.WormholeProver.new(cfg)
    proof = prover.commit(pkg.CircuitInputs.test_inputs()).prove()
    assert len(proof) == prover.circuit_data.proof_size
    oc = OracleCircuit(prover.circuit_data.desc)
    assert oc.verify(proof.to_bytes()) is None
    w, p = prover.circuit_data.witness(0x51570000)
    assert proof.public_inputs == [int(x) for x in p]            # proof_can_be_deserialized analogue (:21-45)
    assert proof.to_bytes() == oc.prove(w, p)


def test_prover_is_single_use(pkg):                              # lib.rs:210-212
    prover = pkg.WormholeProver.new(pkg.CircuitConfig.test_config())
    prover.commit(pkg.CircuitInputs.test_inputs(1))
    with pytest.raises(RuntimeError, match="already commited"):
        prover.commit(pkg.CircuitInputs.test_inputs(2))
    prover.prove()
    with pytest.raises(RuntimeError, match="Failed to prove"):
        prover.prove()


def test_zk_config_prover(pkg):
    prover = pkg.WormholeProver.new(pkg.CircuitConfig.test_config(zero_knowledge=True))
    proof = prover.commit(pkg.CircuitInputs.test_inputs(3)).prove()
    oc = OracleCircuit(prover.circuit_data.desc)
    assert oc.verify(proof.to_bytes()) is None


def test_aggregator_leaf_batch_and_padding(pkg):                 # aggregator_tests.rs: push / aggregate / leaf public inputs
    cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.test_config())
    agg = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, 2))
    agg.prove_leaves([pkg.CircuitInputs.test_inputs(i) for i in range(3)])
    res = agg.aggregate()
    assert res.num_real == 3 and len(res.proofs) == 4
    oc = OracleCircuit(cd.desc)
    for i, pf in enumerate(res.proofs[:3]):
        assert oc.verify(pf.to_bytes()) is None
        _, p = cd.witness(0x51570000 + i)
        assert res.public_inputs[16 * i:16 * (i + 1)] == [int(x) for x in p]


def test_large_merkle_tree_fused_levels():
    rng = np.random.default_rng(77)
    width, leaves, cap = 9, 1 << 20, 4                           # >= 2^20 leaves takes the fused level-1/2 path
    cols = rand_felts(rng, (width, leaves))
    rows = np.ascontiguousarray(cols.T)
    tot = 0; m = leaves
    while True:
        tot += m
        if m <= (1 << cap):
            break
        m >>= 1
    exp = np.zeros((tot, 4), dtype=np.uint64)
    assert oracle().orc_merkle(ptr(rows), leaves, width, cap, ptr(exp)) == tot
    got = np.zeros((tot, 4), dtype=np.uint64)
    gpu_check(qpgpu().qpgpu_test_merkle(ptr(cols), width, leaves, cap, ptr(got)))
    assert np.array_equal(got, exp)


def test_cpp_host_mirror_example_runs():
    """examples/prove_example.cc drives zk-circuits_amd/host/wormhole.hpp end to end (new -> commit -> prove, aggregator buffer)."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "prove_example")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    for mode, size in (("nozk", 132712), ("zk", 148932)):
        r = subprocess.run([exe, mode], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "proof bytes %d" % size in r.stdout and "second commit rejected: prover has already commited to inputs" in r.stdout
