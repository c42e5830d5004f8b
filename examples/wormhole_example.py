#!/usr/bin/env python3
"""The reference's example (/root/reference/wormhole/example/src/main.rs:13-74) on this backend: build the inputs by hand (empty storage proof,
root hash = leaf-inputs hash), commit, prove on the GPU, parse the public inputs back, write the proof as hex.
  python examples/wormhole_example.py [out.hex]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("zk-circuits_amd")


def main(out_path="proof_from_bins.hex"):
    # The inputs of the reference's example (a deployment reads them from a chain node): no storage-proof nodes, so the root hash is the
    # leaf-inputs hash itself.
    funding_account = pkg.BytesDigest(bytes([226, 124, 203, 9, 80, 60, 124, 205, 165, 5, 178, 216, 195, 15, 149, 38, 116, 1, 238, 133,
                                             181, 154, 106, 17, 41, 228, 118, 179, 82, 141, 225, 76]))         # the funding account the reference's example uses
    secret = bytes([1] * 32)
    unspendable_account = pkg.unspendable_account_from_secret(secret)
    funding_amount = 1_000_000_000_000
    transfer_count = 0
    leaf_inputs_felts = ([transfer_count >> 32, transfer_count & 0xFFFFFFFF] + funding_account.to_felts() + unspendable_account.to_felts()
                         + [(funding_amount >> (96 - 32 * i)) & 0xFFFFFFFF for i in range(4)])
    root_hash = pkg.BytesDigest.from_felts([int(v) for v in pkg.ffi.hash_no_pad(leaf_inputs_felts)])
    exit_account = pkg.BytesDigest((8226349481601990196).to_bytes(8, "little") + bytes(24))
    inputs = pkg.CircuitInputs(
        private=pkg.PrivateCircuitInputs(secret=secret, transfer_count=0, funding_account=funding_account,
                                         storage_proof=pkg.ProcessedStorageProof([], []), unspendable_account=unspendable_account),
        public=pkg.PublicCircuitInputs(funding_amount=funding_amount, nullifier=pkg.nullifier_from_preimage(secret, 0), root_hash=root_hash,
                                       exit_account=exit_account))
    config = pkg.CircuitConfig.standard_recursion_config()
    prover = pkg.WormholeProver.new(config)
    proof = prover.commit(inputs).prove()
    public_inputs = pkg.PublicCircuitInputs.try_from_proof(proof)
    print(public_inputs)
    pkg.WormholeVerifier.new(config).verify(proof)
    with open(out_path, "w") as f:
        f.write(proof.to_bytes().hex())
    return proof, public_inputs


if __name__ == "__main__":
    main(*sys.argv[1:2])
