#!/usr/bin/env python3
"""Builds the committed golden fixtures from the reference's own data (run in the build container, where /root/reference
exists; the GPU box only sees the committed outputs).

  * copies the reference's golden binary artefacts (data files its benches/tests load:
    wormhole/verifier/benches/verifier.rs:16-26, wormhole/aggregator/src/util.rs:6-9) and checks their SHA-256
    against SURVEY.md App. B.0;
  * extracts the Poseidon known-answer constants from the reference's test sources into kats.json
    (wormhole/tests/src/circuit/unspendable_account_tests.rs:12-27, wormhole/tests/src/prover/prover_tests.rs:31-35,
    wormhole/tests/test-helpers/src/lib.rs:10-80) — values only, no code.
"""
import hashlib, json, os, re, shutil, sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
BLOBS = {
    "wormhole/bench-data/common.bin": "6431bc288d9cc69dfbdceb78cc1dc96f5c8304501a85ad06767d349077946d95",
    "wormhole/bench-data/verifier.bin": "75aface0ddbc86dbdc2ab49c096554a51cdad8869ad93d456d4b5804449da67d",
    "wormhole/bench-data/proof.bin": "531c61fc52ecc5c448f465e7920dae8d360946375671f86f5cc4809e431e1013",
    "wormhole/aggregator/data/dummy_proof.bin": "dd94fdcf8482f9df49885419fac72261871ed6bd52ab35edce35f6571b26c94d",
    "wormhole/aggregator/data/dummy_proof_zk.bin": "60c2029c422590108593574545b5818f0fa84570141de7e232d7973de97b02a1",
}


def read(p):
    return open(os.path.join(REF, p)).read()


def hex_list(src, name):
    m = re.search(name + r"[^=]*=\s*\[(.*?)\];", src, re.S)
    return re.findall(r'"([0-9a-fA-F]+)"', m.group(1))


def int_list(src, name):
    m = re.search(name + r"[^=]*=\s*\[(.*?)\];", src, re.S)
    return [int(x) for x in re.findall(r"\d+", m.group(1))]


def main():
    sums = {}
    for rel, sha in BLOBS.items():
        data = open(os.path.join(REF, rel), "rb").read()
        got = hashlib.sha256(data).hexdigest()
        assert got == sha, (rel, got)
        dst = os.path.join(HERE, os.path.basename(rel))
        shutil.copyfile(os.path.join(REF, rel), dst)
        os.chmod(dst, 0o644)
        sums[os.path.basename(rel)] = {"sha256": sha, "bytes": len(data), "source": rel}
    ua = read("wormhole/tests/src/circuit/unspendable_account_tests.rs")
    th = read("wormhole/tests/test-helpers/src/lib.rs")
    pt = read("wormhole/tests/src/prover/prover_tests.rs")
    nul = re.search(r"nullifier: BytesDigest::try_from\(\[(.*?)\]\)", pt, re.S).group(1)
    root = re.search(r"root_hash: BytesDigest::try_from\(\[(.*?)\]\)", pt, re.S).group(1)
    amount = re.search(r"DEFAULT_FUNDING_AMOUNT: u128 =\s*u128::from_le_bytes\(\[(.*?)\]\)", th, re.S).group(1)
    kats = {
        "unspendable_secrets": hex_list(ua, "SECRETS"),
        "unspendable_addresses": hex_list(ua, "ADDRESSES"),
        "default_secret": re.search(r'DEFAULT_SECRET: &str = "([0-9a-f]+)"', th).group(1),
        "default_transfer_count": int(re.search(r"DEFAULT_TRANSFER_COUNT: u64 = (\d+)", th).group(1)),
        "default_funding_account": int_list(th, "DEFAULT_FUNDING_ACCOUNT"),
        "default_to_account": int_list(th, "DEFAULT_TO_ACCOUNT"),
        "default_funding_amount_le_bytes": [int(x) for x in re.findall(r"\d+", amount)],
        "default_root_hash": re.search(r'DEFAULT_ROOT_HASH: &str =\s*"([0-9a-f]+)"', th).group(1),
        "storage_proof_nodes": hex_list(th, "DEFAULT_STORAGE_PROOF"),
        "storage_proof_indices": int_list(th, "DEFAULT_STORAGE_PROOF_INDICIES"),
        "expected_nullifier_bytes": [int(x) for x in re.findall(r"\d+", nul)],
        "expected_root_hash_bytes": [int(x) for x in re.findall(r"\d+", root)],
        "expected_funding_amount": 1_000_000_000_000,
        "expected_exit_account": [4] * 32,
        "sources": ["wormhole/tests/src/circuit/unspendable_account_tests.rs:12-27", "wormhole/tests/src/prover/prover_tests.rs:29-43",
                    "wormhole/tests/test-helpers/src/lib.rs:10-80"],
    }
    assert len(kats["unspendable_secrets"]) == 5 and len(kats["storage_proof_nodes"]) == 7
    json.dump(kats, open(os.path.join(HERE, "kats.json"), "w"), indent=1)
    json.dump(sums, open(os.path.join(HERE, "blobs.json"), "w"), indent=1)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
