"""The wormhole circuit's inputs (reference: /root/reference/wormhole/circuit/src/inputs.rs:23-134, common/src/utils.rs) and the vote
circuit's (voting/src/lib.rs:23-52), over libqpcircuit (include/qpcircuit.h): same fields, same encodings, same error texts.

`CircuitInputs` -> `fill_targets(kind)` is WormholeProver::commit's half of the work (the PartialWitness: the input-target values in
allocation order); `ffi.BuiltCircuit.witness` is generate_partial_witness.
"""
import ctypes
from dataclasses import dataclass, field
from typing import List
import numpy as np
from . import ffi

P = ffi.P
PUBLIC_INPUTS_FELTS_LEN = 16          # inputs.rs:96 (LEAF_PI_LEN; the constant at inputs.rs:11 says 14 and is unused)
MAX_PROOF_LEN = 20
PROOF_NODE_MAX_SIZE_F = 188
MAX_MERKLE_DEPTH = 32


class BytesDigest(bytes):
    """32 bytes whose four little-endian u64 chunks are canonical field elements (utils.rs:21-52)."""

    def __new__(cls, value):
        value = bytes(value)
        if len(value) != 32:
            raise ValueError("InvalidLength { expected: 32, got: %d }" % len(value))
        for i in range(4):
            v = int.from_bytes(value[8 * i:8 * i + 8], "little")
            if v >= P:
                raise ValueError("ChunkOutOfFieldRange { chunk_index: %d, value: %d }" % (i, v))
        return bytes.__new__(cls, value)

    @staticmethod
    def from_felts(felts):                                  # utils.rs:54-73
        if len(felts) != 4:
            raise ValueError("failed to deserialize bytes digest from field elements. Expected length 4, got %d" % len(felts))
        return BytesDigest(b"".join(int(f).to_bytes(8, "little") for f in felts))

    def to_felts(self):                                     # digest_bytes_to_felts, utils.rs:196-208
        return [int.from_bytes(self[8 * i:8 * i + 8], "little") for i in range(4)]


@dataclass
class ProcessedStorageProof:
    """storage_proof/mod.rs:59-79: the trie nodes root-first, and for each the hex-digit index at which its child's hash sits."""
    proof: List[bytes]
    indices: List[int]

    def __post_init__(self):
        if len(self.proof) != len(self.indices):
            raise ValueError("indices length must be equal to proof length, actual lengths: %d, %d" % (len(self.proof), len(self.indices)))
        self.proof = [bytes(p) for p in self.proof]


@dataclass
class PublicCircuitInputs:
    funding_amount: int
    nullifier: BytesDigest
    root_hash: BytesDigest
    exit_account: BytesDigest

    @staticmethod
    def try_from_slice(pis):                                # inputs.rs:95-134
        pis = [int(x) for x in pis]
        if len(pis) != PUBLIC_INPUTS_FELTS_LEN:
            raise ValueError("public inputs should contain: %d field elements, got: %d" % (PUBLIC_INPUTS_FELTS_LEN, len(pis)))
        out = ffi.WormholePublicInputsC()
        a = np.array(pis, dtype=np.uint64)
        ffi.circuit_check(ffi.circuit_lib().qpw_public_inputs_from_felts(ffi.ptr(a), len(a), ctypes.byref(out)))
        return PublicCircuitInputs(int.from_bytes(bytes(out.funding_amount_le), "little"), BytesDigest(bytes(out.nullifier)),
                                   BytesDigest(bytes(out.root_hash)), BytesDigest(bytes(out.exit_account)))

    @staticmethod
    def try_from_proof(proof):                              # TryFrom<&ProofWithPublicInputs>, inputs.rs:137-144
        return PublicCircuitInputs.try_from_slice(proof.public_inputs)

    @staticmethod
    def try_from_aggregated(aggr, leaf_pi_len, num_leaves):  # inputs.rs:61-93
        pis = list(aggr.public_inputs)
        expected = leaf_pi_len * num_leaves
        if len(pis) != expected:
            raise ValueError("aggregated public inputs should contain: %d (= %d leaves × %d fields), got: %d" % (expected, num_leaves, leaf_pi_len, len(pis)))
        return [PublicCircuitInputs.try_from_slice(pis[i:i + leaf_pi_len]) for i in range(0, expected, leaf_pi_len)]


@dataclass
class PrivateCircuitInputs:
    secret: bytes
    storage_proof: ProcessedStorageProof
    transfer_count: int
    funding_account: BytesDigest
    unspendable_account: BytesDigest


@dataclass
class CircuitInputs:
    public: PublicCircuitInputs
    private: PrivateCircuitInputs

    def fill_targets(self, kind=ffi.QPC_WORMHOLE):
        """The input-target values of circuit `kind` (the whole wormhole circuit or one fragment): WormholeProver::commit
        (wormhole/prover/src/lib.rs:209-225) / each fragment's fill_targets."""
        c = ffi.WormholeInputsC()
        pub, prv = self.public, self.private

        def put(dst, src, n):
            src = bytes(src)
            if len(src) != n:
                raise ValueError("expected %d bytes, got %d" % (n, len(src)))
            ctypes.memmove(dst, src, n)
        put(c.funding_amount_le, int(pub.funding_amount).to_bytes(16, "little"), 16)
        put(c.nullifier, pub.nullifier, 32); put(c.root_hash, pub.root_hash, 32); put(c.exit_account, pub.exit_account, 32)
        put(c.secret, prv.secret, 32)
        c.transfer_count = prv.transfer_count
        put(c.funding_account, prv.funding_account, 32); put(c.unspendable_account, prv.unspendable_account, 32)
        sp = prv.storage_proof
        k = len(sp.proof)
        c.proof_len = k
        nodes = (ctypes.c_char_p * max(k, 1))(*sp.proof)
        lens = (ctypes.c_uint32 * max(k, 1))(*[len(p) for p in sp.proof])
        idx = (ctypes.c_uint32 * max(k, 1))(*sp.indices)
        c.proof_nodes, c.proof_node_lens, c.proof_indices = nodes, lens, idx
        L = ffi.circuit_lib()
        n = ctypes.c_size_t()
        ffi.circuit_check(L.qpw_fill_targets(kind, ctypes.byref(c), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        ffi.circuit_check(L.qpw_fill_targets(kind, ctypes.byref(c), ffi.ptr(out), len(out), ctypes.byref(n)))
        return out


def nullifier_from_preimage(secret, transfer_count):
    """Nullifier::from_preimage(..).hash as digest bytes (nullifier.rs:53-73)."""
    out = ctypes.create_string_buffer(32)
    ffi.circuit_check(ffi.circuit_lib().qpw_nullifier_from_preimage(bytes(secret), len(secret), transfer_count, out))
    return BytesDigest(out.raw)


def unspendable_account_from_secret(secret):
    """UnspendableAccount::from_secret(..).account_id as digest bytes (unspendable_account.rs:38-63)."""
    if len(secret) != 32:
        raise ValueError("Expected 32 bytes for secret")
    out = ctypes.create_string_buffer(32)
    ffi.circuit_check(ffi.circuit_lib().qpw_unspendable_account_from_secret(bytes(secret), out))
    return BytesDigest(out.raw)


@dataclass
class VotePublicInputs:
    proposal_id: List[int]
    merkle_root: List[int]
    vote: bool
    nullifier: List[int]


@dataclass
class VotePrivateInputs:
    private_key: List[int]
    merkle_siblings: List[List[int]]
    path_indices: List[bool]
    actual_merkle_depth: int


@dataclass
class VoteCircuitData:
    public_inputs: VotePublicInputs
    private_inputs: VotePrivateInputs
    _keep: list = field(default_factory=list, repr=False)

    def fill_targets(self):                                 # voting/src/lib.rs:199-260
        c = ffi.VoteInputsC()
        pub, prv = self.public_inputs, self.private_inputs
        for i in range(4):
            c.proposal_id[i], c.merkle_root[i], c.nullifier[i], c.private_key[i] = pub.proposal_id[i], pub.merkle_root[i], pub.nullifier[i], prv.private_key[i]
        c.vote = 1 if pub.vote else 0
        sib = np.ascontiguousarray(np.array(prv.merkle_siblings, dtype=np.uint64).reshape(-1, 4)) if prv.merkle_siblings else np.zeros((1, 4), dtype=np.uint64)
        path = (ctypes.c_uint8 * max(len(prv.path_indices), 1))(*[1 if b else 0 for b in prv.path_indices])
        c.merkle_siblings, c.path_indices = ffi.ptr(sib), path
        c.num_siblings, c.num_path_indices, c.actual_merkle_depth = len(prv.merkle_siblings), len(prv.path_indices), prv.actual_merkle_depth
        L = ffi.circuit_lib()
        n = ctypes.c_size_t()
        ffi.circuit_check(L.qpv_fill_targets(ctypes.byref(c), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        ffi.circuit_check(L.qpv_fill_targets(ctypes.byref(c), ffi.ptr(out), len(out), ctypes.byref(n)))
        return out
