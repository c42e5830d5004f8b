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
    proof = prover.commit(pkg.SyntheticInputs.test_inputs()).prove()
    assert len(proof) == prover.circuit_data.proof_size
    oc = OracleCircuit(prover.circuit_data.desc)
    assert oc.verify(proof.to_bytes()) is None
    w, p = prover.circuit_data.witness(0x51570000)
    assert proof.public_inputs == [int(x) for x in p]            # proof_can_be_deserialized analogue (:21-45)
    assert proof.to_bytes() == oc.prove(w, p)


def test_prover_is_single_use(pkg):                              # lib.rs:210-212
    prover = pkg.WormholeProver.new(pkg.CircuitConfig.test_config())
    prover.commit(pkg.SyntheticInputs.test_inputs(1))
    with pytest.raises(RuntimeError, match="already commited"):
        prover.commit(pkg.SyntheticInputs.test_inputs(2))
    prover.prove()
    with pytest.raises(RuntimeError, match="Failed to prove"):
        prover.prove()


def test_zk_config_prover(pkg):
    prover = pkg.WormholeProver.new(pkg.CircuitConfig.test_config(zero_knowledge=True))
    proof = prover.commit(pkg.SyntheticInputs.test_inputs(3)).prove()
    oc = OracleCircuit(prover.circuit_data.desc)
    assert oc.verify(proof.to_bytes()) is None


def test_aggregator_leaf_batch_and_padding(pkg):                 # aggregator_tests.rs: push / aggregate / leaf public inputs
    cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.test_config())
    agg = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, 2))
    agg.prove_leaves([pkg.SyntheticInputs.test_inputs(i) for i in range(3)])
    res = agg.aggregate()
    assert res.num_real == 3 and len(res.proofs) == 4
    oc = OracleCircuit(cd.desc)
    for i, pf in enumerate(res.proofs[:3]):
        assert oc.verify(pf.to_bytes()) is None
        _, p = cd.witness(0x51570000 + i)
        assert res.public_inputs[16 * i:16 * (i + 1)] == [int(x) for x in p]


def test_aggregation_tree_with_device_chunk_prover(pkg):          # tree.rs:56-143 control flow, qpgpu_prove per chunk
    """aggregate() to a root through the device: leaves proved as a batch, every tree node proved with qpgpu_prove on a
    recursion-shaped circuit (all 14 gate types). The witness of a node comes from a STAND-IN generator seeded by its children's
    bytes (the real one is qp-plonky2's recursive-verifier CircuitBuilder on the Rust host), so the root proves nothing about the
    leaves; what is checked is the orchestration: level/chunk order, one circuit handle per level, every node proof accepted by the
    oracle verifier for its circuit and byte-equal to the oracle's proof of the same witness, and determinism of the root."""
    import hashlib
    cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.test_config())
    cfg = pkg.TreeAggregationConfig(2, 2)
    level_circuits = {}                                          # per-level cache (the reference rebuilds per chunk, tree.rs:127)
    seen = []

    def chunk_prover(chunk, level):
        if level not in level_circuits:
            sc = SynthCircuit("RT", circuit_seed=0x7EE0 + level)
            level_circuits[level] = (sc, GpuCircuit(sc.desc), OracleCircuit(sc.desc))
        sc, gc, oc = level_circuits[level]
        seed = int.from_bytes(hashlib.sha256(b"".join(bytes(c) if not hasattr(c, "to_bytes") else c.to_bytes() for c in chunk)).digest()[:7], "little")
        w, p = sc.witness(seed)
        proof = gc.prove(w, p)
        assert oc.verify(proof) is None and proof == oc.prove(w, p)
        seen.append((level, len(chunk)))
        return pkg.aggregator.AggregatedProof(proof, sc)

    def run():
        agg = pkg.WormholeProofAggregator(cd, cfg)
        agg.prove_leaves([pkg.SyntheticInputs.test_inputs(i) for i in range(3)])      # 3 real leaves + 1 dummy
        return agg.aggregate(chunk_prover=chunk_prover)

    try:
        root = run()
        assert seen == [(0, 2), (0, 2), (1, 2)] and len(level_circuits) == 2
        assert run().proof == root.proof
    finally:
        for _, gc, _ in level_circuits.values():
            gc.close()


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
        assert "proof %d bytes" % size in r.stdout and "second commit rejected: prover has already commited to inputs" in r.stdout


def test_wormhole_verifier_mirror(pkg):
    """verifier_tests.rs:14-66 against the mirror: verify OK for a proof of this prover; from the reference's verifier.bin / common.bin the
    reference's own proof verifies; a flipped public input raises "proof verification failed"; bad artefact bytes are a ValueError."""
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rd = lambda n: open(os.path.join(G, n), "rb").read()
    cfg = pkg.CircuitConfig.test_config(zero_knowledge=True)
    prover = pkg.WormholeProver.new(cfg)
    proof = prover.commit(pkg.SyntheticInputs.test_inputs(2)).prove()
    ver = pkg.WormholeVerifier.new(cfg)
    ver.verify(proof)
    bad = bytearray(proof.to_bytes()); bad[-1] ^= 1
    with pytest.raises(RuntimeError, match="proof verification failed"):
        ver.verify(bytes(bad))
    ver.close()
    ref = pkg.WormholeVerifier.new_from_bytes(rd("verifier.bin"), rd("common.bin"))      # the file as the reference writes it
    ref.verify(rd("proof.bin"))
    assert ref.verify_batch([rd("proof.bin"), rd("proof.bin")[:-8] + bytes(8)]) == [True, False]
    ref.close()
    with pytest.raises(ValueError, match="Failed to deserialize"):
        pkg.WormholeVerifier.new_from_bytes(rd("verifier.bin")[:100], rd("common.bin"))
    with pytest.raises(ValueError, match="Failed to deserialize"):
        pkg.WormholeVerifier.new_from_bytes(rd("verifier.bin")[:552], rd("common.bin")[:-5])


@pytest.mark.parametrize("arg", ["", "zk"])
def test_cpp_host_mirror_example(arg):
    """examples/prove_example.cc over zk-circuits_amd/host/wormhole.hpp (the C++ mirror of WormholeProver / WormholeVerifier /
    WormholeProofAggregator): prove twice on one resident circuit, verify both, reject a forged public input, single-use commit."""
    import subprocess
    exe = os.path.join(ROOT, "examples", "prove_example")
    if not os.path.exists(exe):
        pytest.skip("examples/prove_example not built (__graft_entry__.build() builds it)")
    r = subprocess.run([exe] + ([arg] if arg else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "verified: 1 0 1" in r.stdout and "second commit rejected" in r.stdout
    size = 148932 if arg == "zk" else 132712                              # the reference's proof sizes
    assert "proof %d bytes" % size in r.stdout and "wormhole circuit (2^%d rows" % (14 if arg == "zk" else 13) in r.stdout
    assert "funding_amount 1000000000000, nullifier a94c9623" in r.stdout           # prover_tests.rs:29-43: the real circuit on the reference's inputs
    assert "invalid root hash rejected" in r.stdout and "set twice with different values" in r.stdout
    assert "synthetic shape %s" % ("B" if arg == "zk" else "A") in r.stdout
    assert "aggregated 4 leaves -> root (2^%d rows, 64 public inputs)" % (15 if arg == "zk" else 14) in r.stdout and "root verified" in r.stdout


def test_prove_and_verify_from_files(pkg, tmp_path):
    """circuit_data_tests.rs:96-183: write the circuit artefacts, build prover and verifier from the files alone, prove and verify — and the
    proof equals the one the in-memory circuit gives for the same witness."""
    written = pkg.generate_circuit_binaries(str(tmp_path), include_prover=True, config=pkg.CircuitConfig.test_config())
    assert set(written) == {"common.bin", "verifier.bin", "prover.qpkg"}
    prover = pkg.WormholeProver.new_from_files(str(tmp_path / "prover.qpkg"), str(tmp_path / "common.bin"))
    src = pkg.WormholeCircuitData.get(pkg.CircuitConfig.test_config())
    w, p = src.witness(0x51570003)
    proof = prover.commit(pkg.SyntheticInputs(wires=w, public_inputs=p)).prove()
    ver = pkg.WormholeVerifier.new_from_files(str(tmp_path / "verifier.bin"), str(tmp_path / "common.bin"))
    ver.verify(proof); ver.close()
    again = pkg.WormholeProver(src).commit(pkg.SyntheticInputs(wires=w, public_inputs=p)).prove()
    assert proof.to_bytes() == again.to_bytes() == OracleCircuit(src.desc).prove(w, p)
    with pytest.raises(ValueError, match="common circuit data"):
        pkg.WormholeProver.new_from_bytes(open(tmp_path / "prover.qpkg", "rb").read(), open(tmp_path / "common.bin", "rb").read()[:-1])
    with pytest.raises(ValueError, match="prover data"):
        pkg.WormholeProver.new_from_bytes(b"not a package" * 10, b"")
    with pytest.raises(RuntimeError, match="no synthetic witness source"):
        pkg.WormholeProver.new_from_files(str(tmp_path / "prover.qpkg"), str(tmp_path / "common.bin")).commit(pkg.SyntheticInputs.test_inputs())


def test_prove_leaves_uses_one_blinding_key_and_one_salt_stream_per_proof(pkg, monkeypatch):
    """every leaf of a batch draws its own prover randomness (the reference draws OsRng inside every prove): prove_leaves hands a
    distinct 32-byte blinding key to each witness, two proofs of ONE statement in one batch differ, and under a caller-supplied salt key
    the ChaCha8 stream continues from call to call instead of restarting at 0."""
    import importlib
    th = importlib.import_module("zk-circuits_amd.test_helpers")
    cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.standard_recursion_zk_config())
    agg = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, 3))
    keys = []
    real = cd._built.witness

    def spy(inputs, blinding_key=None, out=None, threads=0):
        keys.append(bytes(blinding_key)); return real(inputs, blinding_key, out=out, threads=threads)
    monkeypatch.setattr(cd._built, "witness", spy)
    same = [th.circuit_test_inputs() for _ in range(4)]
    key = bytes(range(32))
    agg.prove_leaves(same, salt_key=key)
    first = [p.to_bytes() for p in agg.proofs_buffer]
    assert len(set(keys)) == 4 and all(len(k) == 32 for k in keys)                      # one blinding key per leaf
    assert len(set(first)) == 4                                                          # same statement, four different proofs
    assert agg._salt_streams[key] == 4
    agg.proofs_buffer = []
    agg.prove_leaves(same[:2], salt_key=key)
    assert agg._salt_streams[key] == 6 and len(set(keys)) == 6                           # streams 4, 5: no (key, stream) pair repeats
    v = pkg.WormholeVerifier.new(None, circuit_data=cd)
    assert v.verify_batch([pkg.ProofWithPublicInputs(b, 16) for b in first] + agg.proofs_buffer) == [True] * 6
    agg.close()
