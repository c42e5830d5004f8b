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
    reps = 256 // np.gcd(num, 256) if mode == 2 else 1     # the fused form takes whole blocks of 256 leaves: replicate the rows
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


LEAF_MODES = [0, 1, 2]


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


@pytest.mark.parametrize("leaf_mode,node_mode", [(0, 0), (1, 1), (2, 2), (0, 3)])
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
