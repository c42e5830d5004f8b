"""zk-circuits_amd — MI355X (gfx950) native Plonky2 prover backend for the Quantus wormhole / voting circuits.

Host-side mirror of the reference's drop-in API for the hot path:
  WormholeProver.{new, commit, prove}        (/root/reference/wormhole/prover/src/lib.rs:190-237)
  WormholeVerifier.{new, new_from_bytes, verify}    (/root/reference/wormhole/verifier/src/lib.rs:87-159)
  WormholeProofAggregator.{push_proof, aggregate}  (/root/reference/wormhole/aggregator/src/aggregator.rs:28-92)
on top of the C-ABI in include/qpgpu.h (libqpgpu.so: hand-written HIP kernels). The directory name contains a hyphen;
import it with importlib.import_module("zk-circuits_amd").

The product path never touches oracle/ and has no CPU fallback: loading fails loudly if libqpgpu.so is missing.
"""
from .ffi import lib, QpgpuError, CircuitDesc, ProveOpts, Gate  # noqa: F401
from .prover import (CircuitConfig, SyntheticInputs, ProofWithPublicInputs, WormholeProver,  # noqa: F401
                     WormholeCircuitData)
from .inputs import (BytesDigest, CircuitInputs, PublicCircuitInputs, PrivateCircuitInputs, ProcessedStorageProof,  # noqa: F401
                     VoteCircuitData, VotePublicInputs, VotePrivateInputs, nullifier_from_preimage, unspendable_account_from_secret)
from . import test_helpers  # noqa: F401
from .verifier import WormholeVerifier  # noqa: F401
from .circuit_builder import generate_circuit_binaries  # noqa: F401
from .aggregator import (WormholeProofAggregator, TreeAggregationConfig, AggregatedLeaves, AggregatedProof,  # noqa: F401
                         aggregate_to_tree, aggregate_level, aggregate_to_tree_distributed, gather_proofs, RecursiveChunkProver)
