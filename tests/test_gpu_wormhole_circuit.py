"""The built wormhole / vote circuits (libqpcircuit, include/qpcircuit.h) through the HIP prover on a real GPU: the reference's
prover tests (wormhole/tests/src/prover/prover_tests.rs:14-47), circuit-data round trip (circuit/circuit_data_tests.rs:96-183), leaf
aggregation inputs (aggregator/aggregator_tests.rs) and the vote end-to-end test (voting/src/lib.rs:345-361) with CircuitInputs::test_inputs().

Checked: GPU proof bytes == oracle prover bytes on the same witness and salts; the product verifier and the oracle verifier accept; the
proof's public inputs are the values prover_tests.rs:29-43 expects. The circuit's rows/sigma/digest are this builder's own (unpinned)."""
import importlib, json, os
import numpy as np
import pytest
from harness import *

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("zk-circuits_amd")
th = importlib.import_module("zk-circuits_amd.test_helpers")
KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))
EXPECTED = dict(funding_amount=1_000_000_000_000, nullifier=bytes(KATS["expected_nullifier_bytes"]), root_hash=bytes(KATS["expected_root_hash_bytes"]),
                exit_account=bytes([4] * 32))


def _expected(pub):
    return (pub.funding_amount, bytes(pub.nullifier), bytes(pub.root_hash), bytes(pub.exit_account)) == tuple(EXPECTED.values())


@pytest.mark.parametrize("zk", [False, True])
def test_gpu_proof_of_the_wormhole_circuit_is_byte_equal_to_the_oracle(zk):
    c = BuiltCircuit(QPC_WORMHOLE, zk)
    w, p = c.witness(th.circuit_test_inputs().fill_targets(), bytes(range(32)))
    oc, gc = OracleCircuit(c.desc), GpuCircuit(c.desc)
    assert np.array_equal(gc.cap(), oc.cap())
    salts = SynthCircuit("B").salts_keyed(TEST_SALT_KEY, 11) if zk else None
    got, want = gc.prove(w, p, salt_seed=11), oc.prove(w, p, salts=salts)
    assert got == want and len(got) == (148932 if zk else 132712)
    assert oc.verify(got) is None
    v = GpuVerifier(c.desc, gc.cap())
    assert v.verify(got) is None
    bad = bytearray(got); bad[-8 * 16] ^= 1                                   # another nullifier: rejected
    assert v.verify(bytes(bad)) is not None
    v.close(); gc.close()


def test_commit_and_prove_with_the_reference_inputs():           # prover_tests.rs:14-47
    prover = pkg.WormholeProver.new(pkg.CircuitConfig.standard_recursion_config())
    proof = prover.commit(th.circuit_test_inputs()).prove()
    assert len(proof) == 132712
    assert _expected(pkg.PublicCircuitInputs.try_from_proof(proof))
    pkg.WormholeVerifier.new(pkg.CircuitConfig.standard_recursion_config()).verify(proof)
    with pytest.raises(RuntimeError, match="already commited"):
        prover.commit(th.circuit_test_inputs())


def test_default_prover_is_the_zk_wormhole_circuit():             # WormholeProver::default (lib.rs:80-85): zk config, k = 14
    prover = pkg.WormholeProver.default()
    assert prover.circuit_data.desc.degree_bits == 14 and prover.circuit_data.desc.zero_knowledge == 1
    a = prover.commit(th.circuit_test_inputs()).prove()
    b = pkg.WormholeProver.default().commit(th.circuit_test_inputs()).prove()
    assert len(a) == 148932 and _expected(pkg.PublicCircuitInputs.try_from_proof(a))
    assert a.to_bytes() != b.to_bytes()                           # fresh blinding rows and salts per proof
    v = pkg.WormholeVerifier.new(pkg.CircuitConfig.standard_recursion_zk_config())
    assert v.verify_batch([a, b]) == [True, True]


def test_inputs_that_do_not_satisfy_the_statement_fail_in_prove():
    """storage_proof_tests.rs:29-41 at the prover level: commit() accepts, prove() fails (upstream panics inside prove with
    'set twice with different values')."""
    bad = th.circuit_test_inputs(); bad.public.root_hash = pkg.BytesDigest(bytes(32))
    prover = pkg.WormholeProver.new(pkg.CircuitConfig.standard_recursion_config()).commit(bad)
    with pytest.raises(RuntimeError, match="Failed to prove: .*set twice with different values"):
        prover.prove()


def test_circuit_binaries_round_trip_for_the_built_circuit(tmp_path):
    """circuit_data_tests.rs:96-183: write common.bin / verifier.bin / the prover artefact, load prover and verifier back from the files,
    prove the test inputs' witness and verify."""
    written = pkg.generate_circuit_binaries(str(tmp_path), include_prover=True)          # circuit-builder/src/lib.rs:11-66: non-zk wormhole circuit
    assert set(written) == {"common.bin", "verifier.bin", "prover.qpkg"}
    src = pkg.WormholeCircuitData.get(pkg.CircuitConfig.standard_recursion_config())
    w, p = src.generate_witness(th.circuit_test_inputs())
    prover = pkg.WormholeProver.new_from_files(str(tmp_path / "prover.qpkg"), str(tmp_path / "common.bin"))
    proof = prover.commit(pkg.SyntheticInputs(wires=w, public_inputs=p)).prove()
    pkg.WormholeVerifier.new_from_files(str(tmp_path / "verifier.bin"), str(tmp_path / "common.bin")).verify(proof)
    assert _expected(pkg.PublicCircuitInputs.try_from_proof(proof))
    shape = ReferenceCircuitShape(open(tmp_path / "common.bin", "rb").read(), open(tmp_path / "verifier.bin", "rb").read())
    assert shape.desc.degree_bits == 13 and shape.desc.num_public_inputs == 16


def test_aggregator_takes_real_leaves_and_pads_with_the_dummy():
    """aggregator_tests.rs:60-112: leaf proofs of the reference's inputs, padded with the circuit's dummy proof; the aggregated public
    inputs parse back into the leaves' PublicCircuitInputs."""
    cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.standard_recursion_config())
    agg = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, 2))
    other = th.circuit_test_inputs(); other.public.exit_account = pkg.BytesDigest(bytes([7] * 32))
    agg.prove_leaves([th.circuit_test_inputs(), other, th.circuit_test_inputs()], salt_seed=5)
    res = agg.aggregate()
    assert res.num_real == 3 and len(res.proofs) == 4

    class Agg:
        public_inputs = res.public_inputs
    leaves = pkg.PublicCircuitInputs.try_from_aggregated(Agg, 16, 4)
    assert _expected(leaves[0]) and leaves[1].exit_account == bytes([7] * 32) and _expected(leaves[3])     # the dummy proves the test inputs
    v = pkg.WormholeVerifier.new(pkg.CircuitConfig.standard_recursion_config())
    assert v.verify_batch(res.proofs) == [True] * 4


def test_vote_circuit_end_to_end():                               # voting/src/lib.rs:345-361
    cfg = pkg.CircuitConfig.standard_recursion_config(kind=QPC_VOTE)
    prover = pkg.WormholeProver.new(cfg)
    good = th.vote_test_inputs()
    proof = prover.commit(good).prove()
    assert proof.public_inputs == good.public_inputs.proposal_id + good.public_inputs.merkle_root + [1] + good.public_inputs.nullifier
    pkg.WormholeVerifier.new(cfg).verify(proof)
    assert OracleCircuit(prover.circuit_data.desc).verify(proof.to_bytes()) is None
    bad = th.vote_test_inputs(); bad.private_inputs.actual_merkle_depth = 1                               # test_invalid_merkle_proof
    with pytest.raises(RuntimeError, match="Failed to prove"):
        pkg.WormholeProver.new(cfg).commit(bad).prove()


def test_recursive_aggregation_tree_of_wormhole_proofs():
    """recursive_aggregation_tree / pair_aggregation / public_inputs_are_aggregated (tree.rs:189-243) and aggregator_tests.rs:60-112 with
    real recursion: four leaf proofs of the wormhole circuit (three inputs + the dummy), two levels of node circuits that verify their
    children in-circuit, root proof verified; the root's public inputs are the leaves', in order; node proofs byte-equal to the oracle's."""
    cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.standard_recursion_config())
    agg = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, 2))
    other = th.circuit_test_inputs(); other.public.exit_account = pkg.BytesDigest(bytes([7] * 32))
    agg.prove_leaves([th.circuit_test_inputs(), other, th.circuit_test_inputs()], salt_seed=5)
    prover = pkg.RecursiveChunkProver(cd)
    root = agg.aggregate(chunk_prover=prover)
    rcd = root.circuit_data
    assert rcd.desc.degree_bits == 14 and rcd.desc.num_public_inputs == 64
    pkg.WormholeVerifier.new(None, circuit_data=rcd).verify(root.proof)                       # root_proof.circuit_data.verify(root_proof.proof)
    assert OracleCircuit(rcd.desc).verify(root.proof.to_bytes()) is None

    class Agg:
        public_inputs = root.proof.public_inputs
    leaves = pkg.PublicCircuitInputs.try_from_aggregated(Agg, 16, 4)                          # aggregator_tests.rs: leaf public inputs in order
    assert _expected(leaves[0]) and leaves[1].exit_account == bytes([7] * 32) and _expected(leaves[2]) and _expected(leaves[3])
    # one level-0 node again, alone, against the oracle prover on the same witness
    node_cd, inner, cap, digest = prover.level(0, 2)
    leaf_a = pkg.WormholeProver(cd).commit(th.circuit_test_inputs()).prove()
    w, p = node_cd._built.witness(node_cd._built.aggregation_targets(cap, digest, [leaf_a.to_bytes(), leaf_a.to_bytes()]))
    got = prover([leaf_a, leaf_a], 0).proof.to_bytes()
    assert got == OracleCircuit(node_cd.desc).prove(w, p)
    # a child that does not verify cannot be aggregated
    bad = bytearray(leaf_a.to_bytes()); bad[-1] ^= 1
    with pytest.raises(RuntimeError, match="child proof of this node does not verify"):
        prover([leaf_a, bytes(bad)], 0)


def test_recursive_aggregation_of_zk_leaves():
    """Leaves under standard_recursion_zk_config (salted oracles, k = 14) aggregate through the same node circuits."""
    cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.standard_recursion_zk_config())
    a = pkg.WormholeProver(cd).commit(th.circuit_test_inputs()).prove()
    b = pkg.WormholeProver(cd).commit(th.circuit_test_inputs()).prove()
    prover = pkg.RecursiveChunkProver(cd)
    node = prover([a, b], 0)
    ncd = node.circuit_data
    assert ncd.desc.zero_knowledge == 1 and ncd.desc.degree_bits == 15          # CircuitBuilder::new(common_data.config.clone()): zk children, zk node
    pkg.WormholeVerifier.new(None, circuit_data=ncd).verify(node.proof)
    assert node.proof.public_inputs == a.public_inputs + b.public_inputs
    again = prover([a, b], 0)
    assert again.proof.to_bytes() != node.proof.to_bytes()                      # fresh blinding rows and salts
    root = prover([node.proof, again.proof], 1)                                 # a zk node verifies zk nodes
    assert root.circuit_data.desc.degree_bits == 15 and len(root.proof.public_inputs) == 64
    pkg.WormholeVerifier.new(None, circuit_data=root.circuit_data).verify(root.proof)


def test_the_reference_example(tmp_path):
    """examples/wormhole_example.py = wormhole/example/src/main.rs:13-74 (hand-built inputs with an empty storage proof)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("wormhole_example", os.path.join(ROOT, "examples", "wormhole_example.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    proof, pub = mod.main(str(tmp_path / "proof.hex"))
    assert pub.funding_amount == 10 ** 12 and bytes(pub.exit_account[:8]) == (8226349481601990196).to_bytes(8, "little")
    assert bytes.fromhex(open(tmp_path / "proof.hex").read()) == proof.to_bytes() and len(proof) == 132712


def test_aggregator_default_constructor_aggregates_to_a_root():
    """aggregator_tests.rs:60-112 with the reference's constructors: WormholeProofAggregator::default() (zk config, tree 2^3), push leaf proofs,
    aggregate() -> AggregatedProof whose public inputs parse back into the leaves' (extract_leaf_public_inputs) and whose proof verifies."""
    aggregator = pkg.WormholeProofAggregator.default().with_config(pkg.TreeAggregationConfig(2, 2))
    prover_cd = aggregator.leaf_circuit_data
    inputs = th.circuit_test_inputs()
    for _ in range(3):
        aggregator.push_proof(pkg.WormholeProver(prover_cd).commit(inputs).prove())
    aggregated = aggregator.aggregate()                                                     # pads with the circuit's dummy proof
    leaves = aggregator.extract_leaf_public_inputs(aggregated.proof)
    assert len(leaves) == 4 and all(_expected(x) for x in leaves)
    pkg.WormholeVerifier.new(None, circuit_data=aggregated.circuit_data).verify(aggregated.proof)
    with pytest.raises(RuntimeError, match="no proofs to aggregate"):
        aggregator.aggregate()


@pytest.mark.parametrize("kind", [QPC_NULLIFIER, QPC_UNSPENDABLE_ACCOUNT, QPC_STORAGE_PROOF, QPC_EXIT_ACCOUNT, QPC_VOTE])
@pytest.mark.parametrize("zk", [False, True])
def test_every_built_circuit_proves_byte_equal_to_the_oracle(kind, zk):
    """The fragment circuits the reference's fragment tests build alone (2^5 rows without zero knowledge: no FRI reduction at all; 2^14 with
    its blinding rows; no public inputs for the unspendable-account fragment) and the vote circuit (2^8), through the device prover."""
    c = BuiltCircuit(kind, zk)
    x = th.vote_test_inputs().fill_targets() if kind == QPC_VOTE else th.circuit_test_inputs().fill_targets(kind)
    w, p = c.witness(x, bytes(range(32)))
    oc, gc = OracleCircuit(c.desc), GpuCircuit(c.desc)
    salts = SynthCircuit("B").salts_keyed(TEST_SALT_KEY, 3) if zk else None
    got = gc.prove(w, p, salt_seed=3)
    assert got == oc.prove(w, p, salts=salts) and oc.verify(got) is None
    v = GpuVerifier(c.desc, gc.cap())
    assert v.verify(got) is None
    v.close(); gc.close()


def test_the_reference_golden_proof_is_aggregated_on_the_device():
    """A tree node over the reference's own proof: the node circuit built for wormhole/bench-data/common.bin verifies bench-data/proof.bin
    (made by the Rust prover) in-circuit against verifier.bin, the device proves the node, and the node's proof — whose public inputs are the
    golden proof's — verifies under the product verifier and the oracle verifier; bytes equal to the oracle prover's."""
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rd = lambda f: open(os.path.join(g, f), "rb").read()
    common, vb, proof = rd("common.bin"), rd("verifier.bin"), rd("proof.bin")
    vo = vb[:len(vb) - len(common)] if vb.endswith(common) and len(vb) > len(common) else vb
    shape = ReferenceCircuitShape(common, vo)
    node = BuiltCircuit.aggregation(shape.desc, 1)
    w, p = node.witness(node.aggregation_targets(shape.constants_sigmas_cap, shape.desc.circuit_digest, [proof]), bytes(range(32)))
    gc, oc = GpuCircuit(node.desc), OracleCircuit(node.desc)
    got = gc.prove(w, p, salt_seed=9)
    assert got == oc.prove(w, p, salts=SynthCircuit("B").salts_keyed(TEST_SALT_KEY, 9))
    assert list(np.frombuffer(got[-128:], dtype="<u8")) == list(np.frombuffer(proof[-128:], dtype="<u8"))
    v = GpuVerifier(node.desc, gc.cap())
    assert v.verify(got) is None and oc.verify(got) is None
    v.close(); gc.close()


def test_nodes_without_zero_knowledge_over_zk_leaves():
    """RecursiveChunkProver(node_zero_knowledge=False): zk leaves, nodes without blinding rows (2^14 rows instead of 2^15; a node's witness is
    its children's proofs and verifier data, all public). The default stays the reference's (nodes take their children's config)."""
    cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.standard_recursion_zk_config())
    leaves = [pkg.WormholeProver(cd).commit(th.circuit_test_inputs()).prove() for _ in range(4)]
    prover = pkg.RecursiveChunkProver(cd, node_zero_knowledge=False)
    root = pkg.aggregate_to_tree(leaves, pkg.TreeAggregationConfig(2, 2), prover)
    assert root.circuit_data.desc.zero_knowledge == 0 and root.circuit_data.desc.degree_bits == 14
    pkg.WormholeVerifier.new(None, circuit_data=root.circuit_data).verify(root.proof)
    assert root.proof.public_inputs == [x for l in leaves for x in l.public_inputs]


def test_prover_and_verifier_from_file_e2e(tmp_path):
    """circuit_data_tests.rs:96-183: write the circuit's files, create prover and verifier from the files alone, commit hand-built CircuitInputs
    (empty storage proof: the root hash is the leaf-inputs hash; funding account [2; 32], amount 1000), prove, verify. The prover artefact
    is this backend's circuit package; its digest identifies the wormhole circuit, so the loaded prover generates witnesses like a built one."""
    pkg.generate_circuit_binaries(str(tmp_path), include_prover=True)                    # circuit-builder: standard_recursion_config
    prover = pkg.WormholeProver.new_from_files(str(tmp_path / "prover.qpkg"), str(tmp_path / "common.bin"))
    verifier = pkg.WormholeVerifier.new_from_files(str(tmp_path / "verifier.bin"), str(tmp_path / "common.bin"))
    secret = bytes([1] * 32)
    funding = pkg.BytesDigest(bytes([2] * 32)); unspendable = pkg.unspendable_account_from_secret(secret)
    amount = 1000
    leaf = [0, 0] + funding.to_felts() + unspendable.to_felts() + [(amount >> (96 - 32 * i)) & 0xFFFFFFFF for i in range(4)]
    root = pkg.BytesDigest.from_felts([int(v) for v in pkg.ffi.hash_no_pad(leaf)])
    inputs = pkg.CircuitInputs(pkg.PublicCircuitInputs(amount, pkg.nullifier_from_preimage(secret, 0), root, pkg.BytesDigest(bytes([2] * 32))),
                               pkg.PrivateCircuitInputs(secret, pkg.ProcessedStorageProof([], []), 0, funding, unspendable))
    proof = prover.commit(inputs).prove()
    verifier.verify(proof)
    assert pkg.PublicCircuitInputs.try_from_proof(proof) == inputs.public


@pytest.mark.parametrize("zk", [False, True])
def test_aggregate_proofs_into_tree_like_the_reference_tests(zk):
    """aggregator_tests.rs:11-112 as written there: from_circuit_config, push one leaf proof (push_proof_to_buffer), fill the buffer
    (push_proof_to_full_buffer: the ninth push fails), aggregate the default tree of 2^3 leaves (aggregate_proofs_into_tree), every leaf's
    public inputs come back out of the aggregated proof, and the aggregated proof verifies with its own circuit data."""
    config = pkg.CircuitConfig.standard_recursion_zk_config() if zk else pkg.CircuitConfig.standard_recursion_config()
    proof = pkg.WormholeProver.new(config).commit(th.circuit_test_inputs()).prove()
    public_inputs = pkg.PublicCircuitInputs.try_from_proof(proof)
    aggregator = pkg.WormholeProofAggregator.from_circuit_config(config)
    aggregator.push_proof(proof)
    assert len(aggregator.proofs_buffer) == 1
    for _ in range(aggregator.config.num_leaf_proofs - 1):
        aggregator.push_proof(proof)
    with pytest.raises(RuntimeError, match="proof buffer is full"):
        aggregator.push_proof(proof)
    assert len(aggregator.proofs_buffer) == aggregator.config.num_leaf_proofs == 8
    aggregated = aggregator.aggregate()
    leaves = aggregator.extract_leaf_public_inputs(aggregated.proof)
    assert len(leaves) == 8 and all(l == public_inputs for l in leaves)
    pkg.WormholeVerifier.new(None, circuit_data=aggregated.circuit_data).verify(aggregated.proof)


@pytest.mark.parametrize("seed", [100, 101, 103])
def test_builder_programs_with_the_recursion_gates_prove_byte_equal_to_the_oracle(seed):
    """Random builder programs that mix in the recursive verifier's gates (ArithmeticExtension, MulExtension, Reducing, ReducingExtension, RandomAccess,
    PoseidonMds, CosetInterpolation, Exponentiation: all fourteen gate types; five selector groups, seven constant columns): the device proof of the generated witness is
    byte-equal to the oracle's and both verifiers accept it — the device's gate kernels on rows the builder placed, Exponentiation and
    MulExtension included (the node circuit does not use those two)."""
    c, inputs, expected = BuiltCircuit.random_program(seed, zero_knowledge=(seed == 101), num_inputs=10, num_ops=260, recursion_gates=True)
    w, p = c.witness(inputs, bytes(range(32)))
    assert list(p) == list(expected) and c.desc.num_gates == 14
    oc, gc = OracleCircuit(c.desc), GpuCircuit(c.desc)
    from harness import keyed_salts
    salts = keyed_salts(TEST_SALT_KEY, 4, c.N) if c.desc.zero_knowledge else None
    got = gc.prove(w, p, salt_seed=4)
    assert got == oc.prove(w, p, salts=salts) and oc.verify(got) is None
    v = GpuVerifier(c.desc, gc.cap())
    assert v.verify(got) is None
    bad = bytearray(got); bad[-3] ^= 1
    assert v.verify(bytes(bad)) is not None
    v.close(); gc.close()
