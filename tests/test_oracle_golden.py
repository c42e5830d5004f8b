"""Pins the oracle (CPU restatement) on every golden vector the reference holds for the hot path:
the 14 Poseidon known-answer tests, the whole-protocol golden instance wormhole/bench-data/*, the dummy-proof shapes
and public inputs. Runs without a GPU and without /root/reference (fixtures: tests/golden/, made by make_golden.py)."""
import ctypes, hashlib, json, os
import numpy as np
import pytest
from harness import *

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(G, "kats.json")))
BLOBS = json.load(open(os.path.join(G, "blobs.json")))


def blob(name):
    return open(os.path.join(G, name), "rb").read()


# ---- codecs restated from common/src/utils.rs (how the KAT inputs are encoded)
def injective_bytes_to_felts(b):            # utils.rs:162-174: 4 bytes LE -> felt
    b = bytes(b)
    return [int.from_bytes(b[i:i + 4], "little") for i in range(0, len(b), 4)]


def injective_string_to_felt(s):            # utils.rs:145-159
    b = s.encode(); assert len(b) == 8
    return [int.from_bytes(b[0:4], "little"), int.from_bytes(b[4:8], "little")]


def digest_bytes_to_felts(b):               # utils.rs:189-201: 8 bytes LE -> felt, x4
    b = bytes(b)
    return [int.from_bytes(b[i:i + 8], "little") for i in range(0, 32, 8)]


def u64_to_felts(x):                        # utils.rs:126-131
    return [(x >> 32) & 0xFFFFFFFF, x & 0xFFFFFFFF]


def u128_to_felts(x):                       # utils.rs:104-115: 4 limbs, most significant first
    return [(x >> (96 - 32 * i)) & 0xFFFFFFFF for i in range(4)]


def hash_no_pad(felts):
    a = np.array(felts, dtype=np.uint64); out = np.zeros(4, dtype=np.uint64)
    oracle().orc_hash_no_pad(ptr(a), len(a), ptr(out))
    return [int(x) for x in out]


def test_blob_checksums():
    for name, meta in BLOBS.items():
        assert hashlib.sha256(blob(name)).hexdigest() == meta["sha256"] and len(blob(name)) == meta["bytes"]


def test_round_constants_and_permutation_vectors():
    rc = np.zeros(360, dtype=np.uint64); oracle().orc_round_constants(ptr(rc))
    assert [int(x) for x in rc[:4]] == [0xb585f766f2144405, 0x7746a55f43921ad7, 0xb2fb0d31cee799b4, 0x0f6760a4803427d7]
    assert int(rc[359]) == 0xbc8dfb627fe558fc and int(rc.max()) < P
    st = np.zeros(12, dtype=np.uint64); oracle().orc_permute(ptr(st))
    assert [int(x) for x in st[:4]] == [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca]
    st = np.arange(12, dtype=np.uint64); oracle().orc_permute(ptr(st))
    assert [int(x) for x in st[:4]] == [0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2]


def test_fast_and_generic_permutation_agree():
    L = oracle(); rng = np.random.default_rng(0)
    for _ in range(50):
        a = rand_felts(rng, 12); b = a.copy()
        L.orc_permute(ptr(a)); L.orc_permute_generic(ptr(b))
        assert np.array_equal(a, b)


def test_fast_reduction_matches_slow():
    L = oracle(); rng = np.random.default_rng(1)
    vals = [0, 1, P - 1, P - 2, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 63] + [int(x) for x in rng.integers(0, P, 200, dtype=np.uint64)]
    for a in vals[:40]:
        for b in vals:
            assert L.orc_mul(a, b) == L.orc_mul_slow(a, b) == (a * b) % P


@pytest.mark.parametrize("i", range(5))
def test_kat_unspendable_account(i):        # KATs 1-5: H(H("wormhole" || secret)) == address (unspendable_account.rs:38-63)
    secret = bytes.fromhex(KATS["unspendable_secrets"][i])
    pre = injective_string_to_felt("wormhole") + injective_bytes_to_felts(secret)
    assert len(pre) == 10
    assert hash_no_pad(hash_no_pad(pre)) == digest_bytes_to_felts(bytes.fromhex(KATS["unspendable_addresses"][i]))


def test_kat_nullifier():                   # KAT 6: nullifier.rs:53-73, expected bytes prover_tests.rs:31-35
    secret = bytes.fromhex(KATS["default_secret"])
    pre = injective_string_to_felt("~nullif~") + injective_bytes_to_felts(secret) + u64_to_felts(KATS["default_transfer_count"])
    assert hash_no_pad(hash_no_pad(pre)) == digest_bytes_to_felts(bytes(KATS["expected_nullifier_bytes"]))


def _node_felts(hexnode):
    f = injective_bytes_to_felts(bytes.fromhex(hexnode))
    assert len(f) <= 188
    return f + [0] * (188 - len(f))         # storage_proof/mod.rs:272-280 pads to PROOF_NODE_MAX_SIZE_F


def test_kat_storage_proof_hash_chain():    # KATs 7-13: storage_proof/mod.rs:180-186,197-225
    prev = digest_bytes_to_felts(bytes.fromhex(KATS["default_root_hash"]))
    assert prev == digest_bytes_to_felts(bytes(KATS["expected_root_hash_bytes"]))
    for hexnode, idx in zip(KATS["storage_proof_nodes"], KATS["storage_proof_indices"]):
        node = _node_felts(hexnode)
        assert hash_no_pad(node) == prev
        j = idx // 8
        prev = [node[j + 2 * k] + (node[j + 2 * k + 1] << 32) for k in range(4)]
    # KAT 14: leaf-inputs hash limbs 1..3 sit in the last node (storage_proof/mod.rs:159-161,235-240)
    amount = int.from_bytes(bytes(KATS["default_funding_amount_le_bytes"]), "little")
    assert amount == KATS["expected_funding_amount"]
    leaf = (u64_to_felts(KATS["default_transfer_count"]) + digest_bytes_to_felts(bytes(KATS["default_funding_account"]))
            + digest_bytes_to_felts(bytes(KATS["default_to_account"])) + u128_to_felts(amount))
    assert hash_no_pad(leaf)[1:] == prev[1:]


# ---- whole-protocol golden instance
def _bench():
    return blob("common.bin"), blob("verifier.bin")[:552], blob("proof.bin")


def test_verifier_accepts_golden_proof():
    cb, vo, pb = _bench()
    L = oracle()
    assert blob("verifier.bin")[552:] == cb                      # verifier.bin = verifier-only || common (App. B.0)
    assert L.orc_verify_bytes(cb, len(cb), vo, len(vo), pb, len(pb)) == 0, L.orc_last_error()


def test_transcript_pins():
    cb, vo, pb = _bench()
    out = (ctypes.c_uint64 * 32)()
    assert oracle().orc_transcript_pins(cb, len(cb), vo, len(vo), pb, len(pb), out) == 0
    v = [int(x) for x in out[:21]]
    assert v[0:4] == [0xa4078dae7ee85366, 0xc9ee3673f412af9b, 0x03a7b4a125718fe7, 0x0036d97887fef483]
    assert v[4:8] == [0x7ddbb1f887404dc5, 0x9e44d5322dbb81d7, 0x057855bc392710ec, 0xd05ac43c6b5580d0]
    assert v[8:10] == [0x1820e325483ad180, 0x8a3b669f9f459949]
    assert v[10:12] == [0xf0ad89ec16659844, 0x6228d75cf70da20c]
    assert v[12:14] == [0x39411f867d3ca4ab, 0x1e316e58733fb522]
    assert v[14:16] == [0x687e5086f8af3c42, 0x5e90a83117cab036]
    assert v[16] == 0x0000170b958529af and v[17:21] == [34707, 64718, 9922, 116687]


def test_every_public_input_byte_flip_is_rejected():             # mirrors wormhole/tests/src/verifier/verifier_tests.rs:49-66
    cb, vo, pb = _bench()
    L = oracle()
    for k in range(1, 16 * 8 + 1, 7):
        bad = bytearray(pb); bad[-k] ^= 0x01
        assert L.orc_verify_bytes(cb, len(cb), vo, len(vo), bytes(bad), len(bad)) != 0


def test_proof_body_tampering_is_rejected():
    cb, vo, pb = _bench()
    L = oracle()
    for pos in (0, 600, 1536 + 5, 1536 + 4112 + 3, 20000, 148932 - 150):
        bad = bytearray(pb); bad[pos] ^= 0x04
        assert L.orc_verify_bytes(cb, len(cb), vo, len(vo), bytes(bad), len(bad)) != 0


def test_serialization_roundtrip_is_byte_stable():               # circuit_data_tests.rs:73-93 style
    cb, vo, pb = _bench()
    buf = ctypes.create_string_buffer(200000)
    n = oracle().orc_proof_roundtrip(cb, len(cb), pb, len(pb), buf, len(buf))
    assert n == len(pb) and buf.raw[:n] == pb


def test_dummy_proofs_parse_with_non_zk_k13_shape_and_expected_public_inputs():
    sc = SynthCircuit("A")                                       # same CommonCircuitData shape as the reference's non-zk leaf
    assert GpuLayoutSize(sc.desc) == 132712
    for name in ("dummy_proof.bin", "dummy_proof_zk.bin"):
        pb = blob(name)
        assert len(pb) == 132712
        pis = np.frombuffer(pb[-128:], dtype="<u8")
        assert int.from_bytes(pb[-136:-128], "little") == 16
        assert [int(x) for x in pis[0:4]] == digest_bytes_to_felts(bytes(KATS["expected_nullifier_bytes"]))
        assert [int(x) for x in pis[4:8]] == digest_bytes_to_felts(bytes(KATS["expected_root_hash_bytes"]))
        assert [int(x) for x in pis[8:12]] == u128_to_felts(KATS["expected_funding_amount"])
        assert [int(x) for x in pis[12:16]] == digest_bytes_to_felts(bytes(KATS["expected_exit_account"]))


def GpuLayoutSize(d):
    """Proof size formula of SURVEY.md App. B.3 (what qpgpu_proof_size must return)."""
    S = d.num_constants + d.num_routed; W = d.num_wires; nch = d.num_challenges
    Zp = nch * (1 + d.num_partial_products); Q = nch * d.quotient_degree_factor; s = 4 if d.zero_knowledge else 0
    cap = 32 << d.fri_cap_height
    logN = d.degree_bits + d.fri_rate_bits
    path = logN - d.fri_cap_height
    per_q = sum(8 * w + 1 + 32 * path for w in (S, W + s, Zp + s, Q + s))
    lg = d.degree_bits
    for i in range(d.fri_num_layers):
        a = d.fri_arity_bits[i]
        per_q += (16 << a) + 1 + 32 * (lg + d.fri_rate_bits - a - d.fri_cap_height)
        lg -= a
    n_open = S + W + 2 * nch + nch * d.num_partial_products + Q
    return 3 * cap + 16 * n_open + d.fri_num_layers * cap + d.fri_num_queries * per_q + 16 * (1 << lg) + 8 + 8 + 8 * d.num_public_inputs


def test_proof_size_formula_matches_goldens():
    b, a = SynthCircuit("B"), SynthCircuit("A")          # keep the owners alive: desc points into them
    assert GpuLayoutSize(b.desc) == 148932
    assert GpuLayoutSize(a.desc) == 132712


@pytest.mark.parametrize("name", ["dummy_proof.bin", "dummy_proof_zk.bin"])
def test_dummy_proof_merkle_openings_are_consistent(name):
    """CPU twin of tests/test_gpu_golden_vectors.py::test_dummy_proof_merkle_openings_on_device: the oracle's hash recovers one leaf index
    per query from the wires opening of the reference's dummy proofs, and the other openings reach their caps with it."""
    from proof_format import parse_proof
    L = oracle()
    sc = SynthCircuit("A"); d = sc.desc
    p = parse_proof(d, blob(name))

    def h2(l, r):
        o = np.zeros(4, dtype=np.uint64); L.orc_two_to_one(ptr(np.ascontiguousarray(l)), ptr(np.ascontiguousarray(r)), ptr(o)); return o

    def leaf(row):
        o = np.zeros(4, dtype=np.uint64); row = np.ascontiguousarray(row); L.orc_hash_or_noop(ptr(row), len(row), ptr(o)); return o
    depth = d.degree_bits + d.fri_rate_bits - d.fri_cap_height
    for q in p["queries"][:6]:                                           # six queries keep the pure-Python search in seconds
        cand = [leaf(q["rows"][1])]
        for lvl in range(depth):
            sib = q["paths"][1][lvl]
            cand = [h2(c, sib) for c in cand] + [h2(sib, c) for c in cand]
        hits = [(c, j) for j in range(16) for c in range(len(cand)) if np.array_equal(cand[c], p["wires_cap"][j])]
        assert len(hits) == 1
        idx = hits[0][0] | (hits[0][1] << depth)
        for t, cap in ((2, p["zs_cap"]), (3, p["quotient_cap"])):
            dg, i = leaf(q["rows"][t]), idx
            for lvl in range(depth):
                dg = h2(q["paths"][t][lvl], dg) if i & 1 else h2(dg, q["paths"][t][lvl]); i >>= 1
            assert np.array_equal(dg, cap[i])
        sh = 0
        for l in range(d.fri_num_layers):
            sh += d.fri_arity_bits[l]
            dg, i = leaf(q["layer_evals"][l].reshape(-1)), idx >> sh
            for sib in q["layer_paths"][l]:
                dg = h2(sib, dg) if i & 1 else h2(dg, sib); i >>= 1
            assert np.array_equal(dg, p["fri_caps"][l][i])


def test_the_oracle_checks_with_the_literal_rounds_and_its_fast_form_agrees():
    """The oracle's primary permutation is the literal 30 rounds on the primary constants (round constants + MDS); the derived fast partial
    rounds share their tables with the product (one generator) and are opt-in for timing only. The two forms, and the field-generic
    restatement, agree on random states and on a whole proof."""
    import ctypes
    import numpy as np
    from harness import SynthCircuit, OracleCircuit, oracle, ptr, u64p, P
    L = oracle()
    L.orc_permute.argtypes = [u64p]; L.orc_permute_generic.argtypes = [u64p]
    assert L.orc_poseidon_is_fast() == 0
    rng = np.random.default_rng(11)
    st = rng.integers(0, P, size=(300, 12), dtype=np.uint64); st[0] = 0; st[1] = P - 1
    lit, fast, gen = st.copy(), st.copy(), st.copy()
    for i in range(len(st)):
        L.orc_permute(ptr(lit[i])); L.orc_permute_generic(ptr(gen[i]))
    try:
        L.orc_set_poseidon_fast(1)
        for i in range(len(st)):
            L.orc_permute(ptr(fast[i]))
        sc = SynthCircuit("T"); oc = OracleCircuit(sc.desc); w, p = sc.witness(5)
        pf_fast = oc.prove(w, p)
    finally:
        L.orc_set_poseidon_fast(0)
    assert np.array_equal(lit, gen) and np.array_equal(lit, fast)
    assert int(lit[0][0]) == 0x3c18a9786cb0b359                                  # permute([0; 12])[0], SURVEY.md App. A.2
    assert oc.prove(w, p) == pf_fast and oc.verify(pf_fast) is None
