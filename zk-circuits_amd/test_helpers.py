"""The reference's shared test inputs as this package's `CircuitInputs` — what `CircuitInputs::test_inputs()` returns there
(/root/reference/wormhole/tests/test-helpers/src/lib.rs:25-59; the values live in data/wormhole_test_inputs.json): used by the tests, the dummy
(padding) proof of the built wormhole circuit, the examples and bench.py's default workload. The vote inputs follow voting/src/lib.rs:300-343
(create_test_inputs)."""
import json
import os
from .inputs import (BytesDigest, CircuitInputs, PrivateCircuitInputs, ProcessedStorageProof, PublicCircuitInputs, VoteCircuitData,
                     VotePrivateInputs, VotePublicInputs, nullifier_from_preimage, unspendable_account_from_secret)
from .ffi import hash_no_pad

_D = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "wormhole_test_inputs.json")))
DEFAULT_SECRET = _D["secret"]
DEFAULT_TRANSFER_COUNT = _D["transfer_count"]
DEFAULT_FUNDING_ACCOUNT = bytes(_D["funding_account"])
DEFAULT_FUNDING_AMOUNT = int.from_bytes(bytes(_D["funding_amount_le_bytes"]), "little")
DEFAULT_TO_ACCOUNT = bytes(_D["to_account"])
DEFAULT_EXIT_ACCOUNT = bytes([4] * 32)
DEFAULT_ROOT_HASH = _D["root_hash"]
DEFAULT_STORAGE_PROOF = list(_D["storage_proof"])
DEFAULT_STORAGE_PROOF_INDICIES = list(_D["storage_proof_indices"])


def storage_proof_test_inputs():
    return ProcessedStorageProof([bytes.fromhex(n) for n in DEFAULT_STORAGE_PROOF], list(DEFAULT_STORAGE_PROOF_INDICIES))


def circuit_test_inputs():
    """CircuitInputs::test_inputs() (test-helpers/src/lib.rs:25-59)."""
    secret = bytes.fromhex(DEFAULT_SECRET)
    return CircuitInputs(
        PublicCircuitInputs(DEFAULT_FUNDING_AMOUNT, nullifier_from_preimage(secret, DEFAULT_TRANSFER_COUNT), BytesDigest(bytes.fromhex(DEFAULT_ROOT_HASH)),
                            BytesDigest(DEFAULT_EXIT_ACCOUNT)),
        PrivateCircuitInputs(secret, storage_proof_test_inputs(), DEFAULT_TRANSFER_COUNT, BytesDigest(DEFAULT_FUNDING_ACCOUNT),
                             unspendable_account_from_secret(secret)))


def vote_test_inputs():
    """create_test_inputs() (voting/src/lib.rs:300-343): four voters with keys [1;32]..[4;32], voter 0, proposal [42;32], vote yes."""
    keys = [BytesDigest(bytes([i] * 32)).to_felts() for i in (1, 2, 3, 4)]
    leaves = [[int(x) for x in hash_no_pad(k)] for k in keys]
    level1 = [[int(x) for x in hash_no_pad(leaves[i] + leaves[i + 1])] for i in (0, 2)]
    root = [int(x) for x in hash_no_pad(level1[0] + level1[1])]
    proposal = BytesDigest(bytes([42] * 32)).to_felts()
    nullifier = [int(x) for x in hash_no_pad(leaves[0] + proposal)]          # H(H(private_key) ‖ proposal_id)
    return VoteCircuitData(VotePublicInputs(proposal, root, True, nullifier),
                           VotePrivateInputs(keys[0], [leaves[1], level1[1]], [False, False], 2))
