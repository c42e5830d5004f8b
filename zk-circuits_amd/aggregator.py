"""WormholeProofAggregator mirror (reference: /root/reference/wormhole/aggregator/src/aggregator.rs:13-92,
circuits/tree.rs:24-103, util.rs:11-29).

Scope: the buffer semantics and error behaviour of push_proof / aggregate, the sharding of
independent leaf proofs (one per GPU stream / per rank), the RCCL gather of the fixed-size proof blobs that precedes the
aggregator step, and the aggregation tree itself: `RecursiveChunkProver` is aggregate_chunk (tree.rs:106-143) — the node circuit
verifies its children in-circuit (libqpcircuit's own recursive verifier on the leaf gate set, include/qpcircuit.h; qp-plonky2's is not
on disk, so node circuits are this repo's layout, not upstream's), its witness is generated on the host and proved on the device.
`aggregate()` returns the padded leaf set without a chunk prover and the root proof with one.
"""
import ctypes
import os
import numpy as np
from . import ffi
from .prover import ProofWithPublicInputs, SyntheticInputs

DEFAULT_TREE_BRANCHING_FACTOR = 2   # tree.rs:17
DEFAULT_TREE_DEPTH = 3              # tree.rs:20


class TreeAggregationConfig:
    def __init__(self, tree_branching_factor=DEFAULT_TREE_BRANCHING_FACTOR, tree_depth=DEFAULT_TREE_DEPTH):
        self.tree_branching_factor = tree_branching_factor
        self.tree_depth = tree_depth
        self.num_leaf_proofs = tree_branching_factor ** tree_depth


class AggregatedLeaves:
    """What the aggregator step consumes: num_leaf_proofs proof blobs (real ones first, dummy padding after) and the
    concatenated leaf public inputs (the root proof's public inputs in the reference, tree.rs:122)."""

    def __init__(self, proofs, num_real):
        self.proofs = proofs
        self.num_real = num_real
        self.public_inputs = [x for p in proofs for x in p.public_inputs]


class WormholeProofAggregator:
    def __init__(self, leaf_circuit_data, config=None, chunk_prover=None):
        """aggregator.rs:28-37. `chunk_prover`: what aggregates a chunk of children into their parent (aggregate_chunk); with none set,
        aggregate() stops at the padded leaf set. default() / from_circuit_config() attach the device prover with real recursion."""
        self.leaf_circuit_data = leaf_circuit_data
        self.config = config or TreeAggregationConfig()
        self.proofs_buffer = []
        self.chunk_prover = chunk_prover
        self._witness_pool = []         # reusable (wires, public_inputs) buffers of prove_leaves

    @staticmethod
    def from_circuit_config(circuit_config, device=0):
        """aggregator.rs:39-44: the leaf circuit's data compiled from the config; aggregate() then returns the root proof of the tree
        (every node verifies its children in-circuit, RecursiveChunkProver)."""
        from .prover import WormholeCircuitData
        cd = WormholeCircuitData.get(circuit_config, device)
        return WormholeProofAggregator(cd, None, RecursiveChunkProver(cd, device))

    @staticmethod
    def default(device=0):
        """aggregator.rs:19-24: standard_recursion_zk_config."""
        from .prover import CircuitConfig
        return WormholeProofAggregator.from_circuit_config(CircuitConfig.standard_recursion_zk_config(), device)

    def extract_leaf_public_inputs(self, aggr):
        """aggregator.rs:64-71: the leaves' PublicCircuitInputs from an aggregated (root) proof."""
        from .inputs import PublicCircuitInputs
        return PublicCircuitInputs.try_from_aggregated(aggr, self.leaf_circuit_data.num_public_inputs, self.config.num_leaf_proofs)

    def with_config(self, config):
        self.config = config
        return self

    def push_proof(self, proof):
        if self.proofs_buffer is None:
            self.proofs_buffer = [proof]
            return
        if len(self.proofs_buffer) >= self.config.num_leaf_proofs:
            raise RuntimeError("tried to add proof when proof buffer is full")
        self.proofs_buffer.append(proof)

    def prove_leaves(self, circuit_inputs_list, salt_seed=None, salt_key=None):
        """Proves independent leaf witnesses as one stream-pipelined batch on this GPU (qpgpu_prove_batch) and pushes them.
        Salts: ChaCha8(salt_key, stream i) for leaf i with a fresh OS-RNG key per call, unless `salt_seed` is given (tests: the
        public test key with stream salt_seed + i, reproducible by the oracle)."""
        cd = self.leaf_circuit_data
        n = len(circuit_inputs_list)
        cores = len(os.sched_getaffinity(0))
        real = n > 0 and not isinstance(circuit_inputs_list[0], SyntheticInputs)
        outer = 1 if n <= 1 else min(n, max(2, cores // 2))                 # witnesses generated at once
        inner = max(1, min(8, cores // outer))                              # threads inside one witness (the circuit's independent segments)
        blinding = os.urandom(32) if salt_seed is None else ffi.TEST_SALT_KEY
        if salt_seed is not None:
            key, base = ffi.TEST_SALT_KEY, salt_seed
        else:
            key, base = (salt_key or os.urandom(32)), 0
        # witness matrices of built circuits are reused from call to call (a fresh 17.7 MB matrix costs more in page faults than the generation)
        pool = self._witness_pool
        if real:
            built = cd._built
            while len(pool) < n:
                pool.append((np.empty((cd.num_wires, cd.n), dtype=np.uint64), np.zeros(cd.num_public_inputs, dtype=np.uint64)))

        def witness(ci, slot):
            if not isinstance(ci, SyntheticInputs):                         # the reference's CircuitInputs on the built wormhole circuit
                return built.witness(ci.fill_targets(built.kind), blinding, out=pool[slot], threads=inner)
            return cd.witness(ci.seed) if ci.wires is None else (ci.wires, ci.public_inputs)

        def prove(wit, first):                                             # one device batch; leaf i takes salt stream base + i
            b = len(wit)
            wp = (ctypes.c_void_p * b)(*[w.ctypes.data for w, _ in wit])
            pp = (ffi.u64p * b)(*[p.ctypes.data_as(ffi.u64p) for _, p in wit])
            o = ffi.ProveOpts()
            o.salt_stream = base + first
            keep = ffi.set_salt_key(o, key)      # noqa: F841
            buf = ctypes.create_string_buffer(cd.proof_size * b)
            lens = (ctypes.c_size_t * b)()
            ffi.check(ffi.lib().qpgpu_prove_batch(cd._circuit, b, wp, pp, cd.num_public_inputs, ctypes.byref(o), buf, cd.proof_size, lens))
            for i in range(b):
                self.push_proof(ProofWithPublicInputs(ctypes.string_at(ctypes.addressof(buf) + i * cd.proof_size, cd.proof_size), cd.num_public_inputs))
        if not real or n == 1:
            prove([witness(ci, i) for i, ci in enumerate(circuit_inputs_list)], 0)
            return
        # host threads generate witnesses (the library releases the GIL) while the device proves the groups already complete
        from concurrent.futures import ThreadPoolExecutor
        GROUP = 16
        with ThreadPoolExecutor(max_workers=outer) as ex:
            futures = [ex.submit(witness, ci, i) for i, ci in enumerate(circuit_inputs_list)]
            for first in range(0, n, GROUP):
                prove([f.result() for f in futures[first:first + GROUP]], first)

    def aggregate(self, dummy_proof=None, chunk_prover=None):
        """aggregator.rs:74-92. With a `chunk_prover` (see aggregate_to_tree) the padded leaves are aggregated to the root proof;
        without one the padded leaf set is returned (what aggregate_level consumes)."""
        if self.proofs_buffer is None:
            raise RuntimeError("there are no proofs to aggregate")
        proofs, self.proofs_buffer = self.proofs_buffer, None
        if len(proofs) > self.config.num_leaf_proofs:
            raise RuntimeError("proofs to aggregate was more than the maximum allowed")
        num_real = len(proofs)
        if num_real < self.config.num_leaf_proofs:
            # util.rs:11-29: pad with THE dummy proof of the leaf circuit (never with a real leaf: that would repeat its nullifier and
            # amounts in the aggregated public inputs), and fail if there is none or it does not fit
            if dummy_proof is None:
                if self.leaf_circuit_data is None:
                    raise RuntimeError("failed to deserialize dummy proof")
                dummy_proof = self.leaf_circuit_data.dummy_proof()
            if proofs and len(dummy_proof) != len(proofs[0]):
                raise RuntimeError("failed to deserialize dummy proof: %d bytes, leaf proofs have %d" % (len(dummy_proof), len(proofs[0])))
            proofs = proofs + [dummy_proof] * (self.config.num_leaf_proofs - num_real)
        chunk_prover = chunk_prover or self.chunk_prover
        if chunk_prover is not None:
            return aggregate_to_tree(proofs, self.config, chunk_prover)
        return AggregatedLeaves(proofs, num_real)


def gather_proofs(local_proofs, proof_size, group=None, device=None):
    """All ranks contribute their fixed-size proof blobs; every rank gets the concatenation in rank order.
    torch.distributed all_gather over RCCL (backend "nccl") on GPU tensors, gloo on CPU tensors."""
    import torch
    import torch.distributed as dist
    k = len(local_proofs)
    local_proofs = [p.to_bytes() if hasattr(p, "to_bytes") else bytes(p) for p in local_proofs]
    flat = np.frombuffer(b"".join(local_proofs), dtype=np.uint8).copy() if k else np.zeros(0, dtype=np.uint8)
    t = torch.from_numpy(flat)
    if device is not None:
        t = t.to(device)
    world = dist.get_world_size(group)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    res = []
    for o in outs:
        raw = o.cpu().numpy().tobytes()
        res += [raw[i * proof_size:(i + 1) * proof_size] for i in range(k)]
    return res


# ------------------------------------------------------------------------------------------------ tree orchestration
# Mirror of aggregate_to_tree / aggregate_level / aggregate_chunk (tree.rs:56-143). What a chunk prover does in the reference —
# build the recursive-verifier circuit for the children's common data, set the proof targets, prove — is split at the C-ABI:
# circuit building and witness generation are qp-plonky2 host code (Rust), `circuit_data.prove(pw)` is qpgpu_prove on the
# recursion circuit. Here the chunk prover is a callable, so the control flow (levels, chunking, order, which rank proves which
# node, the exchange of fixed-size blobs) is the reference's, whatever produces the parent proof.

class AggregatedProof:
    """tree.rs:24-28: the proof plus the circuit data needed to verify it / to aggregate it further."""

    def __init__(self, proof, circuit_data=None):
        self.proof = proof
        self.circuit_data = circuit_data


class RecursiveChunkProver:
    """aggregate_chunk (tree.rs:106-143) with the device as prover: for the children's circuit build the node circuit
    (add_virtual_verifier_data, add_virtual_proof_with_pis + verify_proof + register_public_inputs per child), set the verifier-data and
    proof targets, generate the witness (host threads, one node each) and prove the whole level as one device batch. The reference rebuilds
    the node circuit for every chunk (tree.rs:112-128); every chunk of a level has the same circuit, so it is built once per level here.
    Level 0 takes proofs of `leaf_circuit_data`; level k takes proofs of level k-1's node circuit."""

    def __init__(self, leaf_circuit_data, device=0, salt_seed=None, node_zero_knowledge=None):
        """`node_zero_knowledge`: None = nodes take their children's config as in the reference (zk leaves, zk nodes: 2^15 rows per node);
        False = nodes without blinding rows and salts (2^14 rows: a node's witness is public data only — its children's proofs and verifier
        data — so this gives nothing away about the leaves). `salt_seed` (tests, benches): zero-knowledge node circuits take their blinding rows and salts from the public test key with
        reproducible streams; without it both come from the OS RNG, fresh per level."""
        self.leaf = leaf_circuit_data
        self.device = device
        self.salt_seed = salt_seed
        self.node_zero_knowledge = node_zero_knowledge
        self._levels = {}               # (level, children per node) -> (WormholeCircuitData of the node circuit, inner circuit data, its cap, its digest)
        self._level_width = {}          # level -> children per node of the (full) chunks proved at that level
        self._pool = {}                 # (level, children per node) -> reusable (wires, public_inputs) buffers

    def level(self, level, width):
        from .prover import WormholeCircuitData
        key = (level, width)
        if key not in self._levels:
            inner = self.leaf if level == 0 else self.level(level - 1, self._level_width.get(level - 1, width))[0]
            node = ffi.BuiltCircuit.aggregation(inner.desc, width, self.node_zero_knowledge)
            self._levels[key] = (WormholeCircuitData.from_built(node, self.device), inner, inner.constants_sigmas_cap(), list(inner.desc.circuit_digest))
        return self._levels[key]

    def prove_chunks(self, chunks, level):
        if not chunks:
            return []
        width = len(chunks[0])
        if any(len(c) != width for c in chunks):            # a ragged last chunk gets its own (narrower) circuit, as in the reference
            return [self.prove_chunks([c], level)[0] for c in chunks]
        self._level_width.setdefault(level, width)
        cd, inner, cap, digest = self.level(level, width)
        node = cd._built

        zk = bool(cd.desc.zero_knowledge)                 # a node takes its children's config (tree.rs:111)
        blinding_key = None if not zk else (ffi.TEST_SALT_KEY if self.salt_seed is not None else os.urandom(32))

        # witness matrices (35 MB each at 2^15 rows) are reused from call to call: fresh ones cost more in page faults than in arithmetic
        pool = self._pool.setdefault((level, width), [])
        while len(pool) < len(chunks):
            pool.append((np.empty((cd.num_wires, cd.n), dtype=np.uint64), np.zeros(cd.num_public_inputs, dtype=np.uint64)))

        cores = len(os.sched_getaffinity(0))
        outer = 1 if len(chunks) == 1 else min(len(chunks), max(2, cores // 2))           # nodes generated at once
        inner = max(1, min(8, cores // outer))                                            # threads inside one node (its children / their FRI queries)

        def witness(chunk, slot=0):
            raw = [p.to_bytes() if hasattr(p, "to_bytes") else bytes(p) for p in chunk]
            try:
                return node.witness(node.aggregation_targets(cap, digest, raw), blinding_key, out=pool[slot], threads=inner)
            except ffi.CircuitError as e:
                raise RuntimeError("Failed to prove: a child proof of this node does not verify: %s" % e)
        salt_key = None if not zk else (ffi.TEST_SALT_KEY if self.salt_seed is not None else os.urandom(32))

        def prove(wit, first):                               # one device batch (qpgpu_prove_batch: worker streams inside the library)
            b = len(wit)
            wp = (ctypes.c_void_p * b)(*[w.ctypes.data for w, _ in wit])
            pp = (ffi.u64p * b)(*[p.ctypes.data_as(ffi.u64p) for _, p in wit])
            o = ffi.ProveOpts()
            if zk:                                           # proof i of the level takes salt stream base + i
                o.salt_stream = first + ((self.salt_seed + (level << 20)) if self.salt_seed is not None else 0)
                keep = ffi.set_salt_key(o, salt_key)         # noqa: F841
            buf = ctypes.create_string_buffer(cd.proof_size * b)
            lens = (ctypes.c_size_t * b)()
            ffi.check(ffi.lib().qpgpu_prove_batch(cd._circuit, b, wp, pp, cd.num_public_inputs, ctypes.byref(o), buf, cd.proof_size, lens))
            return [AggregatedProof(ProofWithPublicInputs(ctypes.string_at(ctypes.addressof(buf) + i * cd.proof_size, cd.proof_size), cd.num_public_inputs), cd)
                    for i in range(b)]
        if len(chunks) == 1:
            return prove([witness(chunks[0])], 0)
        # witnesses on host threads (generation releases the GIL), proofs in device batches of GROUP nodes: the device proves a group while
        # the host generates the next ones
        from concurrent.futures import ThreadPoolExecutor
        out = []
        with ThreadPoolExecutor(max_workers=outer) as ex:
            futures = [ex.submit(witness, c, i) for i, c in enumerate(chunks)]
            for first in range(0, len(chunks), self.GROUP):
                out += prove([f.result() for f in futures[first:first + self.GROUP]], first)
        return out

    GROUP = 8

    def __call__(self, chunk, level):
        return self.prove_chunks([chunk], level)[0]

    def close(self):
        """Frees the node circuits' device memory and the witness buffers (proofs already returned stay valid; their circuit_data does not)."""
        for cd, _, _, _ in self._levels.values():
            cd.close()
        self._levels, self._pool = {}, {}

    def proof_size_of_level(self, level, width=DEFAULT_TREE_BRANCHING_FACTOR):
        """Byte size of the proofs that ENTER `level` (aggregate_to_tree_distributed's exchange size)."""
        return self.leaf.proof_size if level == 0 else self.level(level - 1, width)[0].proof_size


def aggregate_level(proofs, chunk_prover, config, level):
    """tree.rs:92-103: consecutive chunks of `tree_branching_factor` children, one parent each, order preserved. `chunk_prover(chunk,
    level)` returns an AggregatedProof; the reference maps it with par_chunks — a prover may batch its chunks on the device instead."""
    b = config.tree_branching_factor
    chunks = [proofs[i:i + b] for i in range(0, len(proofs), b)]
    if hasattr(chunk_prover, "prove_chunks"):                     # a device prover takes the whole level as one batch (par_chunks analogue)
        return chunk_prover.prove_chunks(chunks, level)
    return [chunk_prover(c, level) for c in chunks]


def aggregate_to_tree(leaf_proofs, config, chunk_prover):
    """tree.rs:56-79: aggregate level by level until one proof is left."""
    proofs = aggregate_level(list(leaf_proofs), chunk_prover, config, 0)
    level = 1
    while len(proofs) > 1:
        proofs = aggregate_level([p.proof for p in proofs], chunk_prover, config, level)
        level += 1
    assert len(proofs) == 1
    return proofs[0]


def aggregate_to_tree_distributed(local_leaf_proofs, config, chunk_prover, proof_size_of_level, group=None, device=None):
    """The same tree over torch.distributed ranks: rank r holds a contiguous, equally sized run of the leaves.
    While a rank's run holds whole chunks, a level is local work with no communication. Once it does not, the level's proofs
    (fixed-size blobs) are all-gathered in rank order and the chunks are dealt out to ranks 0..chunks-1; every rank returns the root.
    `proof_size_of_level(level)` is the byte size of the proofs that ENTER that level (level 0: leaf proofs). One collective per
    level at the top of the tree, none below: RCCL (backend "nccl") on GPU tensors, gloo on CPU tensors."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b = config.tree_branching_factor
    proofs, level, spread = list(local_leaf_proofs), 0, True          # spread: every rank still holds its own run
    total = len(proofs) * world
    while total > 1:
        if spread and len(proofs) % b == 0 and len(proofs) >= b:
            proofs = [p.proof for p in aggregate_level(proofs, chunk_prover, config, level)]
        else:
            if spread:
                everything = gather_proofs(proofs, proof_size_of_level(level), group=group, device=device)
                spread = False
            else:
                everything = proofs                                     # already replicated on every rank
            chunks = [everything[i:i + b] for i in range(0, len(everything), b)]
            my_chunks = [c for i, c in enumerate(chunks) if i % world == rank]
            if hasattr(chunk_prover, "prove_chunks") and my_chunks:      # a device prover batches this rank's chunks of the level
                mine = [p.proof for p in chunk_prover.prove_chunks(my_chunks, level)]
            else:
                mine = [chunk_prover(c, level).proof for c in my_chunks]
            # one parent per chunk; parents are gathered back in chunk order (rank i % world proved chunk i)
            per_rank = (len(chunks) + world - 1) // world
            size = proof_size_of_level(level + 1)
            padded = mine + [bytes(size)] * (per_rank - len(mine))
            gathered = gather_proofs(padded, size, group=group, device=device)
            proofs = [gathered[(i % world) * per_rank + i // world] for i in range(len(chunks))]
        total = (total + b - 1) // b
        level += 1
    if spread:                                                          # a single rank with a single leaf run
        proofs = gather_proofs(proofs, proof_size_of_level(level), group=group, device=device)[:1]
    return AggregatedProof(proofs[0])
