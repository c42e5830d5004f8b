"""Reference-held vectors through the HIP kernels themselves, with no oracle in the loop (rows a5/a6):

  (i)  the 14 Poseidon known-answer tests of the reference's test sources, hashed by every form of the device leaf sponge
       (one permutation per lane, the 2-lane pair form, the fused-levels form) at the widths they have (10 / 12 / 188 / 14 felts);
  (ii) every Merkle opening of the reference's golden proof wormhole/bench-data/proof.bin: the 28 x 4 initial-oracle rows
       (84 / 139 / 24 / 20 felts) and the 28 x 3 FRI coset leaves (32 felts) are hashed on the device and walked up to the golden
       caps with the device 2-to-1 kernels (per-lane, pair, 16-lane cooperative and the fused multi-level form), at the query
       indices the PRODUCT's transcript derives from the proof bytes (tests/test_transcript_replay.py pins those).
Fixtures: tests/golden/ (made by make_golden.py from the reference's own data files and test constants)."""
import json, os
import numpy as np
import pytest
from harness import *
from proof_format import parse_proof

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KATS = json.load(open(os.path.join(G, "kats.json")))


def blob(name):
    return open(os.path.join(G, name), "rb").read()


# ---- codecs of common/src/utils.rs (how the KAT preimages are encoded; same restatement as tests/test_oracle_golden.py)
def injective_bytes_to_felts(b):
    b = bytes(b); return [int.from_bytes(b[i:i + 4], "little") for i in range(0, len(b), 4)]


def injective_string_to_felt(s):
    b = s.encode(); return [int.from_bytes(b[0:4], "little"), int.from_bytes(b[4:8], "little")]


def digest_bytes_to_felts(b):
    b = bytes(b); return [int.from_bytes(b[i:i + 8], "little") for i in range(0, 32, 8)]


def u64_to_felts(x):
    return [(x >> 32) & 0xFFFFFFFF, x & 0xFFFFFFFF]


def u128_to_felts(x):
    return [(x >> (96 - 32 * i)) & 0xFFFFFFFF for i in range(4)]


def dev_leaf_hash(rows, mode):
    """rows: [num][width] -> [num][4] digests through the device leaf kernel of the given form (hash_or_noop per row)."""
    rows = np.asarray(rows, dtype=np.uint64)
    num, width = rows.shape
    reps = 256 // np.gcd(num, 256) if mode in (2, 4) else 1     # the fused form takes whole blocks of 256 leaves: replicate the rows
    tiled = np.tile(rows, (reps, 1))
    cols = np.ascontiguousarray(tiled.T)             # column-major [width][num]
    out = np.zeros((tiled.shape[0], 4), dtype=np.uint64)
    gpu_check(qpgpu().qpgpu_test_leaf_hash(ptr(cols), width, tiled.shape[0], mode, ptr(out)))
    for r in range(1, reps):
        assert np.array_equal(out[r * num:(r + 1) * num], out[:num])
    return out[:num]


def dev_permute_hash4(d4):
    """hash_no_pad of exactly four felts = permute(d || 0^8)[0..4] (the leaf kernel's hash_or_noop would copy a 4-felt row)."""
    st = np.zeros((len(d4), 12), dtype=np.uint64); st[:, :4] = d4
    gpu_check(qpgpu().qpgpu_test_permute(ptr(st), len(d4)))
    return st[:, :4]


def dev_two_to_one(left, right, mode):
    pairs = np.ascontiguousarray(np.concatenate([left, right], axis=1), dtype=np.uint64)
    out = np.zeros((len(pairs), 4), dtype=np.uint64)
    gpu_check(qpgpu().qpgpu_test_two_to_one(ptr(pairs), len(pairs), mode, ptr(out)))
    return out


LEAF_MODES = [0, 1, 2, 3, 4]      # per-lane, 2-lane pair, fused levels; 3 / 4 = per-lane / fused with the fast form of the partial rounds


@pytest.mark.parametrize("mode", LEAF_MODES)
def test_kats_unspendable_accounts_and_nullifier_on_device(mode):
    # KATs 1-5 (unspendable_account_tests.rs:12-27; preimage rule unspendable_account.rs:38-63): width-10 rows
    pre = [injective_string_to_felt("wormhole") + injective_bytes_to_felts(bytes.fromhex(s)) for s in KATS["unspendable_secrets"]]
    inner = dev_leaf_hash(pre, mode)
    outer = dev_permute_hash4(inner)
    for i in range(5):
        assert [int(x) for x in outer[i]] == digest_bytes_to_felts(bytes.fromhex(KATS["unspendable_addresses"][i]))
    # KAT 6 (prover_tests.rs:31-35; nullifier.rs:53-73): width-12 row
    pre = [injective_string_to_felt("~nullif~") + injective_bytes_to_felts(bytes.fromhex(KATS["default_secret"])) + u64_to_felts(KATS["default_transfer_count"])]
    assert len(pre[0]) == 12
    outer = dev_permute_hash4(dev_leaf_hash(pre, mode))
    assert [int(x) for x in outer[0]] == digest_bytes_to_felts(bytes(KATS["expected_nullifier_bytes"]))


@pytest.mark.parametrize("mode", LEAF_MODES)
def test_kats_storage_proof_chain_on_device(mode):
    # KATs 7-13 (test-helpers/src/lib.rs:68-80; storage_proof/mod.rs:180-225): seven 188-felt nodes in one launch
    nodes = []
    for hexnode in KATS["storage_proof_nodes"]:
        f = injective_bytes_to_felts(bytes.fromhex(hexnode)); nodes.append(f + [0] * (188 - len(f)))
    dig = dev_leaf_hash(nodes, mode)
    prev = digest_bytes_to_felts(bytes.fromhex(KATS["default_root_hash"]))
    for k, (node, idx) in enumerate(zip(nodes, KATS["storage_proof_indices"])):
        assert [int(x) for x in dig[k]] == prev
        j = idx // 8
        prev = [node[j + 2 * t] + (node[j + 2 * t + 1] << 32) for t in range(4)]
    # KAT 14: the 14-felt leaf-inputs hash, limbs 1..3 (storage_proof/mod.rs:159-161,235-240)
    amount = int.from_bytes(bytes(KATS["default_funding_amount_le_bytes"]), "little")
    leaf = (u64_to_felts(KATS["default_transfer_count"]) + digest_bytes_to_felts(bytes(KATS["default_funding_account"]))
            + digest_bytes_to_felts(bytes(KATS["default_to_account"])) + u128_to_felts(amount))
    assert len(leaf) == 14
    assert [int(x) for x in dev_leaf_hash([leaf], mode)[0]][1:] == prev[1:]


@pytest.fixture(scope="module")
def golden():
    shape = ReferenceCircuitShape(blob("common.bin"), blob("verifier.bin")[:552])
    pb = blob("proof.bin")
    return shape, parse_proof(shape.desc, pb), replay_transcript(shape.desc, pb)


@pytest.mark.parametrize("leaf_mode,node_mode", [(0, 0), (1, 1), (2, 2), (0, 3), (3, 0), (4, 1)])
def test_golden_proof_initial_tree_paths_on_device(golden, leaf_mode, node_mode):
    """4 oracles x 28 queries: device leaf hash of the golden rows, 13 device 2-to-1 levels, then the golden caps."""
    shape, p, tr = golden
    d = shape.desc
    caps = [shape.constants_sigmas_cap, p["wires_cap"], p["zs_cap"], p["quotient_cap"]]
    widths = [84, 139, 24, 20]
    nq = d.fri_num_queries
    for t in range(4):
        rows = np.stack([q["rows"][t] for q in p["queries"]])
        assert rows.shape == (nq, widths[t])
        dig = dev_leaf_hash(rows, leaf_mode)
        idx = np.array(tr["query_indices"], dtype=np.int64)
        depth = d.degree_bits + d.fri_rate_bits - d.fri_cap_height
        for lvl in range(depth):
            sib = np.stack([q["paths"][t][lvl] for q in p["queries"]])
            odd = (idx & 1).astype(bool)[:, None]
            dig = dev_two_to_one(np.where(odd, sib, dig), np.where(odd, dig, sib), node_mode)
            idx >>= 1
        assert np.array_equal(dig, caps[t][idx]), "oracle %d: device path does not reach the golden cap" % t


@pytest.mark.parametrize("leaf_mode,node_mode", [(0, 0), (1, 2), (1, 3), (0, 1)])
def test_golden_proof_fri_layer_paths_on_device(golden, leaf_mode, node_mode):
    """3 FRI layers x 28 queries: the 16-coset leaves (32 felts) through the device FRI-leaf kernels and up to the layer caps."""
    shape, p, tr = golden
    d = shape.desc
    idx0 = np.array(tr["query_indices"], dtype=np.int64)
    lg = d.degree_bits + d.fri_rate_bits
    for l in range(d.fri_num_layers):
        a = d.fri_arity_bits[l]
        idx = idx0 >> a                                                # leaf index in this layer's tree
        ev = np.stack([q["layer_evals"][l] for q in p["queries"]])     # [28][16][2]
        aa = np.ascontiguousarray(ev[:, :, 0].reshape(-1)); bb = np.ascontiguousarray(ev[:, :, 1].reshape(-1))
        dig = np.zeros((len(idx), 4), dtype=np.uint64)
        gpu_check(qpgpu().qpgpu_test_fri_leaf_hash(ptr(aa), ptr(bb), a, len(idx), leaf_mode, ptr(dig)))
        depth = lg - a - d.fri_cap_height
        for lvl in range(depth):
            sib = np.stack([q["layer_paths"][l][lvl] for q in p["queries"]])
            odd = (idx & 1).astype(bool)[:, None]
            dig = dev_two_to_one(np.where(odd, sib, dig), np.where(odd, dig, sib), node_mode)
            idx = idx >> 1
        assert np.array_equal(dig, p["fri_caps"][l][idx]), "FRI layer %d: device path does not reach the golden cap" % l
        idx0 = idx0 >> a
        lg -= a


def test_golden_public_inputs_hash_on_device(golden):
    """hash_no_pad(public inputs) (two sponge blocks) through the device leaf kernel == the transcript pin of App. A.6."""
    shape, p, tr = golden
    got = dev_leaf_hash([p["public_inputs"]], 0)[0]
    assert [int(x) for x in got] == [0xa4078dae7ee85366, 0xc9ee3673f412af9b, 0x03a7b4a125718fe7, 0x0036d97887fef483] == tr["pi_hash"]


def _recover_indices(row_digests, paths, cap, two_to_one):
    """Finds, for every query, the leaf index whose Merkle path (given only as leaf digest + siblings) reaches the cap: at each level both
    child orders are tried (2^depth candidates per query, hashed level by level in batches), then the digest is looked up in the cap.
    Returns the indices; fails if any query has no or more than one match."""
    nq, depth = len(paths), paths[0].shape[0]
    cand = row_digests.reshape(nq, 1, 4)                                  # [query][candidate][4]
    for lvl in range(depth):
        k = cand.shape[1]
        sib = np.stack([paths[q][lvl] for q in range(nq)]).reshape(nq, 1, 4).repeat(k, axis=1)
        left = two_to_one(cand.reshape(-1, 4), sib.reshape(-1, 4)).reshape(nq, k, 4)       # node is the left child: bit = 0
        right = two_to_one(sib.reshape(-1, 4), cand.reshape(-1, 4)).reshape(nq, k, 4)      # node is the right child: bit = 1
        cand = np.concatenate([left, right], axis=1)                     # candidate c + k*bit  ->  bits accumulate LSB-first in the candidate number
    idx = []
    for q in range(nq):
        hits = [(c, j) for j in range(len(cap)) for c in np.nonzero((cand[q] == cap[j]).all(axis=1))[0]]
        assert len(hits) == 1, "query %d: %d matches" % (q, len(hits))
        idx.append(int(hits[0][0]) | (hits[0][1] << depth))
    return idx


@pytest.mark.parametrize("name", ["dummy_proof.bin", "dummy_proof_zk.bin"])
def test_dummy_proof_merkle_openings_on_device(name):
    """The reference's other golden proofs (wormhole/aggregator/data/dummy_proof*.bin, embedded at wormhole/aggregator/src/util.rs:6-9)
    cannot be verified — their circuit's verifier data is not shipped, so the transcript (and with it the query indices) cannot be
    replayed. Their Merkle structure can still be checked on the device: the leaf index of every query is recovered from the wires
    opening alone (search over the 2^12 child orders), and with that one index the Z/partial-product and quotient openings and both FRI
    layer openings (index >> 4, >> 8) must reach the caps stored in the proof. Rows here are 135 / 20 / 16 / 32 felts (non-hiding k=13
    shape): sponge lengths the bench-data proof does not exercise."""
    sc = SynthCircuit("A"); d = sc.desc                                   # same CommonCircuitData shape as the reference's non-zk leaf circuit
    p = parse_proof(d, blob(name))
    qs = p["queries"]
    two = lambda l, r: dev_two_to_one(l, r, 0)
    wires_dig = dev_leaf_hash(np.stack([q["rows"][1] for q in qs]), 0)
    idx = _recover_indices(wires_dig, [q["paths"][1] for q in qs], p["wires_cap"], two)
    assert len(set(idx)) > 20 and max(idx) < (1 << 16)
    depth = d.degree_bits + d.fri_rate_bits - d.fri_cap_height

    def walk(dig, paths, index, node_mode):
        index = np.array(index, dtype=np.int64)
        for lvl in range(paths[0].shape[0]):
            sib = np.stack([pth[lvl] for pth in paths]); odd = (index & 1).astype(bool)[:, None]
            dig = dev_two_to_one(np.where(odd, sib, dig), np.where(odd, dig, sib), node_mode); index = index >> 1
        return dig, index
    for t, cap, mode in ((2, p["zs_cap"], 1), (3, p["quotient_cap"], 2)):
        dig, top = walk(dev_leaf_hash(np.stack([q["rows"][t] for q in qs]), t - 1), [q["paths"][t] for q in qs], idx, mode)
        assert qs[0]["paths"][t].shape[0] == depth and np.array_equal(dig, cap[top]), "oracle %d" % t
    # constants/sigmas: the cap is not in the proof, but openings that land on the same cap entry must produce the same digest
    dig, top = walk(dev_leaf_hash(np.stack([q["rows"][0] for q in qs]), 0), [q["paths"][0] for q in qs], idx, 3)
    seen = {}
    for k, j in enumerate(top):
        assert seen.setdefault(int(j), dig[k].tobytes()) == dig[k].tobytes()
    shift = 0
    for l in range(d.fri_num_layers):
        a = d.fri_arity_bits[l]; shift += a
        ev = np.stack([q["layer_evals"][l] for q in qs])
        aa = np.ascontiguousarray(ev[:, :, 0].reshape(-1)); bb = np.ascontiguousarray(ev[:, :, 1].reshape(-1))
        ld = np.zeros((len(qs), 4), dtype=np.uint64)
        gpu_check(qpgpu().qpgpu_test_fri_leaf_hash(ptr(aa), ptr(bb), a, len(qs), l % 2, ptr(ld)))
        dig, top = walk(ld, [q["layer_paths"][l] for q in qs], [i >> shift for i in idx], 2)
        assert np.array_equal(dig, p["fri_caps"][l][top]), "FRI layer %d" % l
