"""The reference's circuits built by libqpcircuit (include/qpcircuit.h), on the CPU: the tests of
/root/reference/wormhole/tests/src/{circuit/*_tests.rs, prover/prover_tests.rs} and voting/src/lib.rs:345-448 restated against this
repo's builder, with the oracle as the checker (constraint check of the full witness, prover, verifier).

Pinned by the reference: the Poseidon KATs (unspendable_account_tests.rs:12-27), the nullifier bytes and the public inputs of
CircuitInputs::test_inputs() (prover_tests.rs:29-43), the 7-node storage proof hash chain (test-helpers/src/lib.rs:68-80), the degree of the
built circuits (k = 13 without zero knowledge as aggregator/data/dummy_proof.bin, k = 14 with it as bench-data/common.bin).
NOT pinned: gate placement, sigma, circuit digest — this builder's own (the reference's verifier.bin cannot check these proofs)."""
import copy, importlib, json, os
import numpy as np
import pytest
from harness import *

pkg = importlib.import_module("zk-circuits_amd")
th = importlib.import_module("zk-circuits_amd.test_helpers")
KATS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))
BLIND = bytes(range(32))


@pytest.fixture(scope="module")
def circuits():
    cache = {}

    def get(kind, zk=False):
        if (kind, zk) not in cache:
            c = BuiltCircuit(kind, zk); cache[(kind, zk)] = (c, OracleCircuit(c.desc))
        return cache[(kind, zk)]
    return get


def test_native_hashes_match_the_reference_kats():
    for s, a in zip(KATS["unspendable_secrets"], KATS["unspendable_addresses"]):           # unspendable_account_tests.rs:12-27
        assert pkg.unspendable_account_from_secret(bytes.fromhex(s)).hex() == a
    assert list(pkg.nullifier_from_preimage(bytes.fromhex(th.DEFAULT_SECRET), 4)) == KATS["expected_nullifier_bytes"]   # prover_tests.rs:31-35
    assert pkg.unspendable_account_from_secret(bytes.fromhex(th.DEFAULT_SECRET)) == th.DEFAULT_TO_ACCOUNT            # circuit.rs:131-136 needs it
    assert any(pkg.nullifier_from_preimage(bytes(64), 4))                                                              # nullifier_tests.rs:50-56


def test_test_helpers_hold_the_reference_fixture_values():
    assert th.DEFAULT_SECRET == KATS["default_secret"] and list(th.DEFAULT_FUNDING_ACCOUNT) == KATS["default_funding_account"]
    assert [n for n in th.DEFAULT_STORAGE_PROOF] == KATS["storage_proof_nodes"] and th.DEFAULT_STORAGE_PROOF_INDICIES == KATS["storage_proof_indices"]
    assert th.DEFAULT_ROOT_HASH == KATS["default_root_hash"] and th.DEFAULT_FUNDING_AMOUNT == KATS["expected_funding_amount"]


@pytest.mark.parametrize("kind,zk", [(QPC_NULLIFIER, False), (QPC_UNSPENDABLE_ACCOUNT, False), (QPC_STORAGE_PROOF, False), (QPC_EXIT_ACCOUNT, False),
                                     (QPC_WORMHOLE, False), (QPC_NULLIFIER, True), (QPC_WORMHOLE, True)])     # zk: 2^14 rows each (seconds of oracle set-up): two of them
def test_fragment_witness_satisfies_every_constraint(circuits, kind, zk):
    """build_and_verify_*_proof of each *_tests.rs: the fragment alone with the shared test inputs."""
    c, oc = circuits(kind, zk)
    w, p = c.witness(th.circuit_test_inputs().fill_targets(kind), BLIND)
    assert check_witness(oc, w, p) is None
    assert len(p) == {QPC_NULLIFIER: 4, QPC_UNSPENDABLE_ACCOUNT: 0, QPC_STORAGE_PROOF: 8, QPC_EXIT_ACCOUNT: 4, QPC_WORMHOLE: 16}[kind]


def test_wormhole_circuit_shape_and_public_inputs(circuits):
    c, _ = circuits(QPC_WORMHOLE)
    czk, _ = circuits(QPC_WORMHOLE, True)
    assert c.desc.degree_bits == 13 and czk.desc.degree_bits == 14            # dummy_proof.bin / common.bin
    assert [czk.desc.fri_arity_bits[i] for i in range(czk.desc.fri_num_layers)] == [4, 4, 4]
    rows = c.row_counts()
    # one BaseSum row per range_check / split_le: 12 + 2 + 6 + 20 * (1 + 188); Poseidon: 3 + 3 + 2 + 20 * 24 + 2 for the public-input hash
    assert rows["base_sum"] == 3800 and rows["poseidon"] == 490 and rows["public_input"] == 1 and rows["rows_in_use"] <= 8192
    assert c.num_inputs == 14 + 12 + (20 * 188 + 20 + 4 + 1 + 14) + 4
    _, pis = c.witness(th.circuit_test_inputs().fill_targets())
    got = pkg.PublicCircuitInputs.try_from_slice(pis)                         # proof_can_be_deserialized, prover_tests.rs:22-47
    assert got == pkg.PublicCircuitInputs(1_000_000_000_000, pkg.BytesDigest(bytes(KATS["expected_nullifier_bytes"])),
                                          pkg.BytesDigest(bytes(KATS["expected_root_hash_bytes"])), pkg.BytesDigest(bytes([4] * 32)))
    again = BuiltCircuit(QPC_WORMHOLE)
    assert again.desc.circuit_digest[:] == c.desc.circuit_digest[:]                            # building is deterministic
    assert czk.desc.circuit_digest[:] != c.desc.circuit_digest[:]


def _fails(c, inputs, kind, needle="set twice with different values"):
    with pytest.raises(CircuitError) as e:
        c.witness(inputs.fill_targets(kind), BLIND)
    assert e.value.code == -8 and needle in str(e.value), str(e.value)


def test_storage_proof_negative_cases(circuits):
    """storage_proof_tests.rs:29-103: each tampering fails witness generation the way upstream's does."""
    c, _ = circuits(QPC_STORAGE_PROOF)
    good = th.circuit_test_inputs()
    bad = copy.deepcopy(good); bad.public.root_hash = pkg.BytesDigest(bytes(32)); _fails(c, bad, QPC_STORAGE_PROOF)                  # invalid_root_hash_fails
    bad = copy.deepcopy(good); sp = bad.private.storage_proof
    b = bytearray(sp.proof[0]); b[sp.indices[0] // 2] ^= 0xFF; sp.proof[0] = bytes(b); _fails(c, bad, QPC_STORAGE_PROOF)              # tampered_proof_fails
    bad = copy.deepcopy(good); bad.private.transfer_count = 5; _fails(c, bad, QPC_STORAGE_PROOF)                                      # invalid_nonce
    bad = copy.deepcopy(good); bad.private.unspendable_account = pkg.BytesDigest(bytes(32)); _fails(c, bad, QPC_STORAGE_PROOF)        # invalid_exit_address (to_account)
    bad = copy.deepcopy(good); bad.public.funding_amount = 1000 << 96; _fails(c, bad, QPC_STORAGE_PROOF)                              # invalid_funding_amount: limbs [1000, 0, 0, 0]


def test_storage_proof_fuzz_every_byte_flip_fails(circuits):
    """fuzz_tampered_proof (storage_proof_tests.rs:105-150, 1000 iterations there, #[ignore]d for time): 60 seeded flips here."""
    c, _ = circuits(QPC_STORAGE_PROOF)
    rng = np.random.default_rng(0x5157)
    for _ in range(60):
        bad = th.circuit_test_inputs(); sp = bad.private.storage_proof
        i = int(rng.integers(len(sp.proof))); b = bytearray(sp.proof[i]); b[int(rng.integers(len(b)))] ^= int(rng.integers(1, 256)); sp.proof[i] = bytes(b)
        with pytest.raises(CircuitError):
            c.witness(bad.fill_targets(QPC_STORAGE_PROOF))


def test_nullifier_and_shared_target_negative_cases(circuits):
    c, _ = circuits(QPC_NULLIFIER)
    bad = th.circuit_test_inputs(); s = bytearray(bad.private.secret); s[0] ^= 0xFF; bad.private.secret = bytes(s)
    _fails(c, bad, QPC_NULLIFIER)                                              # invalid_secret_fails_proof (nullifier_tests.rs:38-48)
    w, _ = circuits(QPC_WORMHOLE)
    bad = th.circuit_test_inputs(); bad.private.unspendable_account = pkg.BytesDigest(bytes(32)); _fails(w, bad, QPC_WORMHOLE)   # account_id != H(H(salt, secret))
    bad = th.circuit_test_inputs(); bad.public.nullifier = pkg.BytesDigest(bytes(32)); _fails(w, bad, QPC_WORMHOLE)
    ok = th.circuit_test_inputs(); ok.public.exit_account = pkg.BytesDigest(bytes([9] * 32))                                       # unconstrained public input
    _, pis = w.witness(ok.fill_targets())
    assert pkg.PublicCircuitInputs.try_from_slice(pis).exit_account == bytes([9] * 32)


def test_range_checks_are_enforced(circuits):
    """The 32-bit range checks of the preimage limbs (nullifier.rs:230-233): an input target wider than 32 bits cannot be witnessed."""
    c, _ = circuits(QPC_NULLIFIER)
    x = th.circuit_test_inputs().fill_targets(QPC_NULLIFIER)
    x[4] = np.uint64(1 << 32)                                                 # first secret limb
    with pytest.raises(CircuitError) as e:
        c.witness(x)
    assert "set twice with different values" in str(e.value)
    x[4] = np.uint64(P)                                                       # not a field element at all
    with pytest.raises(CircuitError) as e:
        c.witness(x)
    assert "canonical" in str(e.value)


def test_fill_targets_errors_follow_the_reference():
    good = th.circuit_test_inputs()
    bad = copy.deepcopy(good); bad.private.storage_proof = pkg.ProcessedStorageProof([b"\0" * 8] * 21, [0] * 21)
    with pytest.raises(ValueError, match="proof length exceeds maximum allowed length: 21 > 20"):          # storage_proof/mod.rs:258-264
        bad.fill_targets()
    bad = copy.deepcopy(good); bad.private.storage_proof = pkg.ProcessedStorageProof([b"\1" * (4 * 189)], [0])
    with pytest.raises(ValueError, match="proof node at index 0 is too large: 189"):                     # storage_proof/mod.rs:272-278
        bad.fill_targets()
    with pytest.raises(ValueError, match="indices length must be equal to proof length"):                # storage_proof/mod.rs:67-75
        pkg.ProcessedStorageProof([b"ab"], [])
    with pytest.raises(ValueError, match="ChunkOutOfFieldRange { chunk_index: 1"):                        # utils.rs:41-49
        pkg.BytesDigest(bytes(8) + (P).to_bytes(8, "little") + bytes(16))
    bad = copy.deepcopy(good); bad.public.root_hash = bytes([0xFF] * 32)                                  # bypassing BytesDigest: the C side checks too
    with pytest.raises(ValueError, match="root hash: ChunkOutOfFieldRange"):
        bad.fill_targets()
    with pytest.raises(ValueError, match="public inputs should contain: 16 field elements, got: 3"):      # inputs.rs:103-109
        pkg.PublicCircuitInputs.try_from_slice([1, 2, 3])
    with pytest.raises(ValueError, match="failed to deserialize funding amount"):
        pkg.PublicCircuitInputs.try_from_slice([0] * 8 + [1 << 32] + [0] * 7)


def test_public_inputs_codec_and_aggregated_parse():
    class Agg:
        public_inputs = []
    _, pis = BuiltCircuit(QPC_WORMHOLE).witness(th.circuit_test_inputs().fill_targets())
    Agg.public_inputs = list(pis) * 3
    leaves = pkg.PublicCircuitInputs.try_from_aggregated(Agg, 16, 3)          # inputs.rs:61-93
    assert len(leaves) == 3 and leaves[0] == leaves[2] and leaves[0].funding_amount == 10 ** 12
    with pytest.raises(ValueError, match=r"aggregated public inputs should contain: 64 \(= 4 leaves × 16 fields\), got: 48"):
        pkg.PublicCircuitInputs.try_from_aggregated(Agg, 16, 4)
    d = pkg.BytesDigest(bytes(range(32)))
    assert pkg.BytesDigest.from_felts(d.to_felts()) == d


@pytest.mark.parametrize("zk", [False, True])
def test_oracle_proves_and_verifies_the_wormhole_circuit(circuits, zk):
    """commit_and_prove (prover_tests.rs:14-19) with the CPU checker: witness -> oracle prover -> oracle verifier; the proof's trailing
    public inputs are the expected ones; a proof for other public inputs is rejected."""
    c, oc = circuits(QPC_WORMHOLE, zk)
    w, p = c.witness(th.circuit_test_inputs().fill_targets(), BLIND)
    salts = SynthCircuit("B").salts_keyed(TEST_SALT_KEY, 7) if zk else None   # [3][4][2^17]: same N
    proof = oc.prove(w, p, salts=salts)
    assert oc.verify(proof) is None
    assert len(proof) == (148932 if zk else 132712)                           # the reference's proof sizes for these two shapes
    assert list(np.frombuffer(proof[-128:], dtype="<u8")) == list(p)
    bad = bytearray(proof); bad[-1] ^= 1
    assert oc.verify(bytes(bad)) is not None


def test_vote_circuit(circuits):
    """voting/src/lib.rs:345-434."""
    c, oc = circuits(QPC_VOTE)
    good = th.vote_test_inputs()
    w, p = c.witness(good.fill_targets())
    assert check_witness(oc, w, p) is None                                    # test_vote_circuit_end_to_end
    assert list(p) == good.public_inputs.proposal_id + good.public_inputs.merkle_root + [1] + good.public_inputs.nullifier
    bad = copy.deepcopy(good); bad.private_inputs.actual_merkle_depth = 33
    with pytest.raises(ValueError, match="Merkle tree depth 33 exceeds maximum allowed depth 32"):       # test_invalid_merkle_depth
        bad.fill_targets()
    bad = copy.deepcopy(good); bad.private_inputs.path_indices.append(False)
    with pytest.raises(ValueError, match="Merkle proof length mismatch: 2 siblings vs 3 path indices"):  # test_merkle_proof_length_mismatch
        bad.fill_targets()
    bad = copy.deepcopy(good); bad.private_inputs.actual_merkle_depth = 1                                 # test_invalid_merkle_proof
    with pytest.raises(CircuitError, match="set twice with different values"):
        c.witness(bad.fill_targets())
    bad = copy.deepcopy(good)                                                                             # test_completely_invalid_proof
    bad.private_inputs.private_key = [12345] * 4; bad.private_inputs.merkle_siblings = [[67890] * 4, [11111] * 4]; bad.private_inputs.path_indices = [True, True]
    with pytest.raises(CircuitError, match="set twice with different values"):
        c.witness(bad.fill_targets())
    x = good.fill_targets(); x[8] = np.uint64(2)                                                          # vote must be a bit (add_virtual_bool_target_safe)
    with pytest.raises(CircuitError):
        c.witness(x)
    proof = oc.prove(w, p)
    assert oc.verify(proof) is None
    # a deeper proof: the voter at depth 5 in a tree of 32 leaves, right/left mix
    keys = [[i + 1, 0, 0, 7] for i in range(32)]
    level = [[int(v) for v in pkg.ffi.hash_no_pad(k)] for k in keys]; idx = 19; sib, path = [], []
    leaf = level[idx]
    while len(level) > 1:
        sib.append(level[idx ^ 1]); path.append(bool(idx & 1)); idx >>= 1
        level = [[int(v) for v in pkg.ffi.hash_no_pad(level[i] + level[i + 1])] for i in range(0, len(level), 2)]
    prop = [5, 6, 7, 8]
    deep = pkg.VoteCircuitData(pkg.VotePublicInputs(prop, level[0], False, [int(v) for v in pkg.ffi.hash_no_pad(leaf + prop)]),
                               pkg.VotePrivateInputs(keys[19], sib, path, 5))
    w, p = c.witness(deep.fill_targets())
    assert check_witness(oc, w, p) is None and int(p[8]) == 0


def _random_storage_inputs(rng, num_nodes):
    """A valid CircuitInputs with a made-up storage proof of `num_nodes` nodes: node i embeds H(node i+1) (four little-endian u64) at a
    felt-aligned offset, node 0 hashes to the root, the last node embeds a digest whose limbs 1..3 are those of the leaf-inputs hash — the
    statement storage_proof/mod.rs:140-244 checks, with other lengths, offsets and node sizes than the one fixture of the reference."""
    ffi = pkg.ffi
    secret = bytes(rng.integers(0, 256, 32, dtype=np.uint8))
    tc = int(rng.integers(0, 1 << 40))
    amount = int(rng.integers(0, 1 << 62)) << int(rng.integers(0, 60))
    funding = pkg.BytesDigest(b"".join(int(rng.integers(0, P, dtype=np.uint64)).to_bytes(8, "little") for _ in range(4)))
    unspendable = pkg.unspendable_account_from_secret(secret)
    leaf_felts = [tc >> 32, tc & 0xFFFFFFFF] + funding.to_felts() + unspendable.to_felts() + [(amount >> (96 - 32 * i)) & 0xFFFFFFFF for i in range(4)]
    leaf_hash = [int(v) for v in ffi.hash_no_pad(leaf_felts)]
    embedded = [int(rng.integers(0, P, dtype=np.uint64))] + leaf_hash[1:]                       # limb 0 is not compared (storage_proof/mod.rs:232-240)
    nodes, indices = [], []
    for i in reversed(range(num_nodes)):
        size = int(rng.integers(10, 189)) * 4                                    # 10..188 felts
        off = int(rng.integers(0, min(size // 4 - 8, 179) + 1)) * 4              # felt-aligned, 8 felts fit, felt index <= 179
        body = bytearray(rng.integers(0, 256, size, dtype=np.uint8))
        body[off:off + 32] = b"".join(v.to_bytes(8, "little") for v in embedded)
        nodes.insert(0, bytes(body)); indices.insert(0, off * 2)                 # hex-digit index
        felts = [int.from_bytes(body[k:k + 4], "little") for k in range(0, size, 4)]
        embedded = [int(v) for v in ffi.hash_no_pad(felts + [0] * (188 - len(felts)))]
    root = pkg.BytesDigest(b"".join(v.to_bytes(8, "little") for v in embedded))
    exit_account = pkg.BytesDigest(b"".join(int(rng.integers(0, P, dtype=np.uint64)).to_bytes(8, "little") for _ in range(4)))
    return pkg.CircuitInputs(pkg.PublicCircuitInputs(amount, pkg.nullifier_from_preimage(secret, tc), root, exit_account),
                             pkg.PrivateCircuitInputs(secret, pkg.ProcessedStorageProof(nodes, indices), tc, funding, unspendable))


@pytest.mark.parametrize("num_nodes", [1, 2, 7, 13, 19, 20])
def test_wormhole_circuit_accepts_other_valid_inputs(circuits, num_nodes):
    """The one fixture of the reference exercises one proof length and one set of offsets; these are made-up valid statements of every
    shape the circuit allows (1..20 nodes, nodes of 10..188 felts, child hash at any felt offset 0..179)."""
    c, oc = circuits(QPC_WORMHOLE)
    rng = np.random.default_rng(1000 + num_nodes)
    ci = _random_storage_inputs(rng, num_nodes)
    w, p = c.witness(ci.fill_targets())
    assert check_witness(oc, w, p) is None
    pub = pkg.PublicCircuitInputs.try_from_slice(p)
    assert pub == ci.public
    # and the same statement stops holding when any one of its parts moves
    bad = copy.deepcopy(ci); bad.private.transfer_count ^= 1; bad.public.nullifier = pkg.nullifier_from_preimage(bad.private.secret, bad.private.transfer_count)
    if num_nodes < 20:                                                            # with 20 nodes the leaf check has no iteration left (i == proof_len never happens): upstream's loop, mirrored
        _fails(c, bad, QPC_WORMHOLE)
    else:
        c.witness(bad.fill_targets())
    bad = copy.deepcopy(ci); sp = bad.private.storage_proof
    k = int(rng.integers(num_nodes)); b = bytearray(sp.proof[k]); b[int(rng.integers(len(b)))] ^= 0x10; sp.proof[k] = bytes(b)
    _fails(c, bad, QPC_WORMHOLE)


@pytest.mark.parametrize("depth", [0, 1, 17, 31, 32])
def test_vote_circuit_accepts_any_depth(circuits, depth):
    """Merkle membership along a made-up path of `depth` levels (0 = the root is the voter's leaf). MAX_MERKLE_DEPTH is 32 (voting/src/lib.rs:20)
    and fill_targets admits it, but the circuit compares the depth over n_log = bits(MAX - 1) = 5 bits (lib.rs:131-134), so depth 32 cannot be
    witnessed: upstream's behaviour, mirrored (31 is the deepest provable path)."""
    c, oc = circuits(QPC_VOTE)
    rng = np.random.default_rng(50 + depth)
    h = pkg.ffi.hash_no_pad
    key = [int(v) for v in rng.integers(0, P, 4, dtype=np.uint64)]
    sib = [[int(v) for v in rng.integers(0, P, 4, dtype=np.uint64)] for _ in range(depth)]
    path = [bool(v) for v in rng.integers(0, 2, depth)]
    leaf = [int(v) for v in h(key)]; cur = leaf
    for s_, right in zip(sib, path):
        cur = [int(v) for v in h((s_ + cur) if right else (cur + s_))]
    prop = [int(v) for v in rng.integers(0, P, 4, dtype=np.uint64)]
    data = pkg.VoteCircuitData(pkg.VotePublicInputs(prop, cur, bool(depth & 1), [int(v) for v in h(leaf + prop)]), pkg.VotePrivateInputs(key, sib, path, depth))
    if depth == 32:
        with pytest.raises(CircuitError, match="set twice with different values"):
            c.witness(data.fill_targets())
        return
    w, p = c.witness(data.fill_targets())
    assert check_witness(oc, w, p) is None and list(p) == prop + cur + [depth & 1] + data.public_inputs.nullifier
    if depth:
        bad = copy.deepcopy(data); bad.private_inputs.path_indices[depth // 2] ^= True
        with pytest.raises(CircuitError, match="set twice with different values"):
            c.witness(bad.fill_targets())


def test_the_reference_example_inputs_with_an_empty_storage_proof(circuits):
    """wormhole/example/src/main.rs:13-74: no storage-proof nodes, the root hash is the leaf-inputs hash itself (the leaf check fires at
    i = 0 = proof_len), secret [1; 32], transfer count 0, exit account id 8226349481601990196."""
    c, oc = circuits(QPC_WORMHOLE)
    secret = bytes([1] * 32)
    funding = pkg.BytesDigest(th.DEFAULT_FUNDING_ACCOUNT)                         # the example's funding account: the same bytes as the test helpers'
    unspendable = pkg.unspendable_account_from_secret(secret)
    amount = 1_000_000_000_000
    leaf_felts = [0, 0] + funding.to_felts() + unspendable.to_felts() + [(amount >> (96 - 32 * i)) & 0xFFFFFFFF for i in range(4)]
    root = pkg.BytesDigest.from_felts([int(v) for v in pkg.ffi.hash_no_pad(leaf_felts)])
    exit_account = pkg.BytesDigest((8226349481601990196).to_bytes(8, "little") + bytes(24))
    ci = pkg.CircuitInputs(pkg.PublicCircuitInputs(amount, pkg.nullifier_from_preimage(secret, 0), root, exit_account),
                           pkg.PrivateCircuitInputs(secret, pkg.ProcessedStorageProof([], []), 0, funding, unspendable))
    w, p = c.witness(ci.fill_targets())
    assert check_witness(oc, w, p) is None
    assert pkg.PublicCircuitInputs.try_from_slice(p) == ci.public
    bad = copy.deepcopy(ci); bad.public.funding_amount += 1                       # another amount: the leaf-inputs hash no longer matches the root
    _fails(c, bad, QPC_WORMHOLE)


@pytest.mark.parametrize("seed", range(12))
def test_builder_fuzz_random_programs(seed):
    """Random programs of builder operations (arithmetic with its constant-folding shortcuts, booleans, selects, equality, bit splits, hashes,
    swapped permutations, hinted inverses, connects), evaluated natively next to the builder calls (qpc_test_random_circuit): the generated
    witness must reproduce every natively computed value, satisfy every gate and copy constraint, and prove."""
    c, inputs, expected = BuiltCircuit.random_program(seed, zero_knowledge=(seed == 1), num_inputs=8 + seed, num_ops=150 + 40 * seed)
    assert c.num_inputs == len(inputs) and c.desc.num_public_inputs == len(expected) > 20
    w, p = c.witness(inputs, bytes(range(32)))
    assert list(p) == list(expected)
    oc = OracleCircuit(c.desc)
    assert check_witness(oc, w, p) is None
    if seed in (0, 2):
        assert oc.verify(oc.prove(w, p)) is None
    bad = inputs.copy(); bad[0] = (int(bad[0]) + 1) % P                           # other inputs: other values (or a failed connect / hint), never the same outputs
    try:
        _, p2 = c.witness(bad, bytes(range(32)))
        assert list(p2) != list(expected)
    except CircuitError:
        pass


@pytest.mark.parametrize("seed", range(8))
def test_builder_fuzz_with_the_recursion_gates(seed):
    """The same fuzz with the recursive verifier's gates mixed in (ArithmeticExtension, MulExtension, Reducing, ReducingExtension, RandomAccess,
    PoseidonMds, CosetInterpolation, Exponentiation — all eight, layouts of SURVEY.md App. D as the oracle and the device evaluate them): every natively computed
    value is reproduced (Horner reductions over one to three rows, list look-ups, a native Lagrange interpolation, native powers), the
    witness satisfies every gate and copy constraint of a gate list with five selector groups, and the oracle proves and verifies it."""
    c, inputs, expected = BuiltCircuit.random_program(100 + seed, zero_knowledge=(seed == 1), num_inputs=8 + seed, num_ops=150 + 40 * seed, recursion_gates=True)
    assert c.desc.num_gates == 14 and c.desc.num_selector_groups == 5 and c.desc.num_constants == 7
    tags = [c.desc.gates[i].tag for i in range(c.desc.num_gates)]
    assert tags == [9, 3, 12, 10, 2, 15, 14, 0, 1, 8, 5, 13, 4, 11]             # by degree: Noop | Constant PublicInput PoseidonMds | BaseSum Reducing ReducingExt | Arithmetic ArithmeticExt MulExt | Exp | RandomAccess | Coset | Poseidon
    w, p = c.witness(inputs, bytes(range(32)))
    assert list(p) == list(expected)
    oc = OracleCircuit(c.desc)
    assert check_witness(oc, w, p) is None
    if seed in (0, 1):
        assert oc.verify(oc.prove(w, p, salts=_salts_for(c) if c.desc.zero_knowledge else None)) is None
    bad = inputs.copy(); bad[0] = (int(bad[0]) + 1) % P
    try:
        _, p2 = c.witness(bad, bytes(range(32)))
        assert list(p2) != list(expected)
    except CircuitError:
        pass


def _salts_for(c):
    from harness import keyed_salts, TEST_SALT_KEY
    return keyed_salts(TEST_SALT_KEY, 3, c.N)


def test_built_wormhole_circuit_has_the_reference_common_circuit_data(circuits):
    """The CommonCircuitData of the wormhole circuit built here under standard_recursion_zk_config, written in plonky2's byte format by the
    product's writer, is byte-identical to the reference's wormhole/bench-data/common.bin (1,045 bytes): circuit config, FRI parameters and
    reduction arities, degree 2^14, the gate list with its parameters in upstream's order, selector indices and groups, quotient degree
    factor, 123 gate constraints, 4 constant columns, 16 public inputs, the k_i coset shifts, 9 partial products. What common data does not
    hold — rows, sigma, constants — stays this builder's own."""
    ffi = pkg.ffi
    ref = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "common.bin"), "rb").read()
    c, _ = circuits(QPC_WORMHOLE, True)
    assert ffi.write_common(c.desc) == ref
    nz, _ = circuits(QPC_WORMHOLE, False)
    assert ffi.write_common(nz.desc) != ref                                       # the non-zk circuit: degree 2^13, two reductions, no hiding
    shape = ReferenceCircuitShape(ffi.write_common(nz.desc), ffi.write_verifier_only(np.zeros((16, 4), dtype=np.uint64), 4, [0, 0, 0, 0]))
    assert shape.desc.degree_bits == 13 and shape.desc.zero_knowledge == 0 and shape.desc.fri_num_layers == 2


@pytest.mark.parametrize("name", ["dummy_proof.bin", "dummy_proof_zk.bin"])
def test_public_inputs_equal_those_of_the_reference_dummy_proofs(circuits, name):
    """wormhole/aggregator/data/dummy_proof{,_zk}.bin are proofs the reference's prover made for CircuitInputs::test_inputs()
    (prover_tests.rs:56-82): their 16 public-input felts are the ones this circuit's witness generation produces from the same inputs, and a
    proof of the non-zk circuit built here has their length."""
    blob = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name), "rb").read()
    c, _ = circuits(QPC_WORMHOLE)
    _, p = c.witness(th.circuit_test_inputs().fill_targets())
    assert list(np.frombuffer(blob[-128:], dtype="<u8")) == [int(x) for x in p]
    assert int.from_bytes(blob[-136:-128], "little") == 16 and len(blob) == pkg.ffi.proof_size_of(c.desc)


def test_amount_and_count_limb_encodings_round_trip():
    """utils_tests.rs:11-156: u128 <-> four 32-bit limbs (most significant first) and u64 <-> two limbs, through the input encoding of the
    storage-proof fragment (the last four / the first two of its leaf targets) and PublicCircuitInputs::try_from_slice."""
    base = th.circuit_test_inputs()
    for num in [0, 1, 0x1234567890abcdef, (1 << 128) - 1, (1 << 96) - 1, (1 << 64) | 0xabcdef]:
        ci = copy.deepcopy(base); ci.public.funding_amount = num
        x = [int(v) for v in ci.fill_targets(QPC_STORAGE_PROOF)]
        assert x[-4:] == [(num >> (96 - 32 * i)) & 0xFFFFFFFF for i in range(4)]
        pis = [0] * 8 + x[-4:] + [0] * 4
        assert pkg.PublicCircuitInputs.try_from_slice(pis).funding_amount == num
    for count in [0, 1, 0x12345678, (1 << 64) - 1, (1 << 32) - 1, (1 << 32) | 0xabcdef]:
        ci = copy.deepcopy(base); ci.private.transfer_count = count
        x = [int(v) for v in ci.fill_targets(QPC_NULLIFIER)]
        assert x[-2:] == [count >> 32, count & 0xFFFFFFFF]                       # Nullifier targets: hash (4), secret (8), transfer_count (2)
    limbs = [[0, 0, 0, 0], [1, 1, 1, 1], [0x1234567, 0xabcdef, 0x1234567, 0xabcdef], [0xFFFFFFFF] * 4, [0, 0xFFFFFFFF, 0, 0xFFFFFFFF], [0xFFFFFFFF, 0, 0xFFFFFFFF, 0]]
    for l in limbs:                                                                # felts -> u128 -> felts
        num = pkg.PublicCircuitInputs.try_from_slice([0] * 8 + l + [0] * 4).funding_amount
        assert [(num >> (96 - 32 * i)) & 0xFFFFFFFF for i in range(4)] == l
