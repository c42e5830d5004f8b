"""WormholeProofAggregator mirror (reference: /root/reference/wormhole/aggregator/src/aggregator.rs:13-92,
circuits/tree.rs:24-103, util.rs:11-29).

Scope: the buffer semantics and error behaviour of push_proof / aggregate, the sharding of
independent leaf proofs (one per GPU stream / per rank), the RCCL gather of the fixed-size proof blobs that precedes the
aggregator step, and the aggregation tree itself: `RecursiveChunkProver` is aggregate_chunk (tree.rs:106-143) — the node circuit
verifies its children in-circuit (libqpcircuit's own recursive verifier on the leaf + recursion gate set, include/qpcircuit.h; qp-plonky2's is not
on disk, so node circuits are this repo's layout, not upstream's), its witness is generated on the host and proved on the device.
`aggregate()` returns the padded leaf set without a chunk prover and the root proof with one.
"""
import ctypes
import os
import numpy as np
import hashlib
from . import ffi
from .hostcpus import witness_pool_shape
from .prover import ProofWithPublicInputs, SyntheticInputs

DEFAULT_TREE_BRANCHING_FACTOR = 2   # tree.rs:17
DEFAULT_TREE_DEPTH = 3              # tree.rs:20


def proof_blinding_key(salt_seed, index, domain=b"leaf"):
    """The 32-byte key of ONE proof's zk blinding rows. Production (salt_seed None): fresh OS randomness per proof, as the reference draws
    its blinding rows from OsRng inside every prove. Tests/benches: derived from the public test key, distinct per (domain, salt_seed + index).
    Sharing one key across the proofs of a batch would repeat the blinding rows in all of them (Builder::generate derives them from
    (key, column) only), and the openings of many proofs would then outnumber the shared unknowns."""
    if salt_seed is None:
        return os.urandom(32)
    return hashlib.sha256(ffi.TEST_SALT_KEY + b"|blinding|" + domain + b"|" + int(salt_seed + index).to_bytes(16, "little")).digest()


def pipelined_batches(n, group, window, workers, witness, prove):
    """Bounded producer/consumer between host witness generation and device batches: item i is generated into buffer slot i % window
    by `witness(i, slot)` on a pool of `workers` threads, and `prove(list_of_witnesses, first_index)` takes `group` consecutive items at
    a time; a slot is handed to a later item only after the batch that used it has been proved. Returns the concatenated results."""
    from concurrent.futures import ThreadPoolExecutor
    assert window >= group and window % group == 0
    out, futures, nxt = [], {}, 0
    with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
        def top_up(limit):
            nonlocal nxt
            while nxt < min(n, limit):
                futures[nxt] = ex.submit(witness, nxt, nxt % window)
                nxt += 1
        top_up(window)
        try:
            for first in range(0, n, group):
                wit = [futures.pop(i).result() for i in range(first, min(n, first + group))]
                out += prove(wit, first) or []
                top_up(first + group + window)          # the slots of the batch just proved are free again
        finally:
            for f in futures.values():
                f.cancel()
    return out


class _fast_thread_switching:
    """The scheduler threads of stream_proofs / prove_tree spend their time inside library calls (GIL released) and need the interpreter only
    for a few microseconds between them; CPython's default 5 ms switch interval would let one of them sit on a finished proof that long."""

    def __enter__(self):
        import sys
        self.old = sys.getswitchinterval(); sys.setswitchinterval(1e-4)

    def __exit__(self, *a):
        import sys
        sys.setswitchinterval(self.old)


def stream_proofs(n, circuit_data, buffers, witness, opts_for, witness_threads, provers=8):
    """Independent proofs of one circuit as a continuous flow: `witness_threads` host threads generate witnesses (witness(i, wires, pis) fills
    a page-locked matrix taken from `buffers`), `provers` threads each keep one qpgpu_prove in flight (the library picks its batch-mode
    kernels when it sees several), a matrix goes back to the free list when its proof is done. No batch boundaries: the device never drains
    while witnesses are available. opts_for(i) returns (ProveOpts, keep-alive) for proof i. Returns the n proofs' bytes in order."""
    import queue
    import threading
    cd = circuit_data
    free, ready = queue.Queue(), queue.Queue()
    for wbuf in buffers:
        free.put(wbuf)
    out = [None] * n
    nxt = [0]
    left = [n]
    lock = threading.Lock()
    done = threading.Event(); failure = []

    def fail(e):
        with lock:
            failure.append(e)
        done.set()

    def witness_thread():
        while not done.is_set():
            with lock:
                i = nxt[0]
                if i >= n:
                    return
                nxt[0] += 1
            wires = None
            while wires is None:
                if done.is_set():
                    return
                try:
                    wires = free.get(timeout=0.05)
                except queue.Empty:
                    pass
            if done.is_set():
                return
            try:
                pis = np.zeros(cd.num_public_inputs, dtype=np.uint64)
                witness(i, wires, pis)
                ready.put((i, wires, pis))
            except Exception as e:                         # noqa: BLE001
                fail(e)

    def prover_thread():
        buf = ctypes.create_string_buffer(cd.proof_size); ln = ctypes.c_size_t()
        while not done.is_set():
            try:
                i, wires, pis = ready.get(timeout=0.05)
            except queue.Empty:
                continue
            if wires is None:
                return
            try:
                o, keep = opts_for(i)                      # noqa: F841
                ffi.check(ffi.lib().qpgpu_prove(cd._circuit, wires.ctypes.data, pis.ctypes.data_as(ffi.u64p), len(pis), ctypes.byref(o), buf, cd.proof_size, ctypes.byref(ln)))
                free.put(wires)
                out[i] = ctypes.string_at(buf, ln.value)
                with lock:
                    left[0] -= 1
                    if left[0] == 0:
                        done.set()
            except Exception as e:                         # noqa: BLE001
                fail(e)
    threads = [threading.Thread(target=prover_thread, daemon=True) for _ in range(max(1, min(provers, n)))] + \
              [threading.Thread(target=witness_thread, daemon=True) for _ in range(max(1, witness_threads))]
    with _fast_thread_switching():
        for t in threads:
            t.start()
        done.wait()
        for _ in threads:                                  # wake whoever still waits on a queue
            ready.put((-1, None, None)); free.put(None)
        for t in threads:
            t.join()
    if failure:
        raise failure[0]
    return out


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
        self._witness_pool = []         # reusable (wires, public_inputs) buffers of prove_leaves: a ring of RING_GROUPS * GROUP
        self._salt_streams = {}         # caller-supplied salt key -> next unused ChaCha8 stream

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

    GROUP = 16                          # leaf proofs per device batch
    RING_GROUPS = 3                     # witness buffers kept: RING_GROUPS * GROUP (the host may run two groups ahead of the device)

    def prove_leaves(self, circuit_inputs_list, salt_seed=None, salt_key=None):
        """Proves independent leaf witnesses as stream-pipelined batches on this GPU (qpgpu_prove_batch) and pushes them.
        Prover-side randomness is per PROOF, as in the reference (OsRng inside every prove): every leaf gets its own 32-byte blinding key
        (the zk blinding rows of its witness) and its own salt stream. Salts: ChaCha8(key, stream) with a fresh OS-RNG key per call and
        stream = leaf index; with a caller-supplied `salt_key` the stream continues from where the previous call under that aggregator
        stopped — a (key, stream) pair must never repeat, so do not hand the same key to two aggregators. `salt_seed` (tests, benches):
        the PUBLIC test key with stream salt_seed + i and blinding keys derived from it, reproducible by the oracle."""
        cd = self.leaf_circuit_data
        n = len(circuit_inputs_list)
        real = n > 0 and not isinstance(circuit_inputs_list[0], SyntheticInputs)
        outer, inner = witness_pool_shape(n)        # witnesses generated at once x threads inside one witness (independent segments)
        if salt_seed is not None:
            key, base = ffi.TEST_SALT_KEY, salt_seed
        elif salt_key is not None:
            key, base = salt_key, self._salt_streams.get(bytes(salt_key), 0)
            self._salt_streams[bytes(salt_key)] = base + n
        else:
            key, base = os.urandom(32), 0
        # witness matrices of built circuits are reused (a fresh 17.7 MB matrix costs more in page faults than the generation): a ring
        # of RING_GROUPS * GROUP buffers, released by close()
        pool = self._witness_pool
        window = self.GROUP * self.RING_GROUPS
        if real:
            built = cd._built
            while len(pool) < min(n, window):                                # page-locked: the device pulls a witness with one DMA transfer
                pool.append(ffi.witness_buffer(cd.num_wires, cd.n, cd.num_public_inputs))

        def witness(i, slot):
            ci = circuit_inputs_list[i]
            if not isinstance(ci, SyntheticInputs):                         # the reference's CircuitInputs on the built wormhole circuit
                return built.witness(ci.fill_targets(built.kind), proof_blinding_key(salt_seed, i), out=pool[slot], threads=inner)
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
            prove([witness(i, i) for i in range(n)], 0)
            return
        # host threads generate witnesses (the library releases the GIL) while the device proves: a continuous flow of single proofs, or
        # (QPGPU_TREE_STREAMING=0) device batches of GROUP leaves behind a bounded ring
        if os.environ.get("QPGPU_TREE_STREAMING", "1") == "0":
            pipelined_batches(n, self.GROUP, window, outer, witness, prove)
            return

        def fill(i, wires, pis):
            built.witness(circuit_inputs_list[i].fill_targets(built.kind), proof_blinding_key(salt_seed, i), out=(wires, pis), threads=inner)

        def opts_for(i):
            o = ffi.ProveOpts(); o.salt_stream = base + i
            return o, ffi.set_salt_key(o, key)
        for raw in stream_proofs(n, cd, [b[0] for b in pool], fill, opts_for, outer):
            self.push_proof(ProofWithPublicInputs(raw, cd.num_public_inputs))

    def close(self):
        """Releases the witness buffers (and the chunk prover's circuits and buffers)."""
        for buf in self._witness_pool:
            ffi.release_witness_buffer(buf)
        self._witness_pool = []
        if self.chunk_prover is not None and hasattr(self.chunk_prover, "close"):
            self.chunk_prover.close()

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


_native_comms = {}          # (process group id, device index) -> ffi.ProofComm, or False when the native path is not to be used


def native_comm(group=None, device=None):
    """The native RCCL communicator (include/qpgpu.h qpgpu_comm_*) matching a torch.distributed process group whose backend is RCCL:
    rank 0 draws the unique id, torch.distributed carries the 128 bytes to the other ranks, every rank joins. Created once per group and
    cached; None when the group is not on RCCL, when QPGPU_NATIVE_GATHER=0, or when a rank could not create its communicator (the
    decision is agreed between the ranks, so that either all of them take the native path or none)."""
    import torch
    import torch.distributed as dist
    if os.environ.get("QPGPU_NATIVE_GATHER", "1") == "0" or dist.get_backend(group) != "nccl" or device is None:
        return None
    dev = torch.device(device)
    key = (id(group) if group is not None else 0, dev.index or 0)
    if key not in _native_comms:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ok, uid = 1, bytes(ffi.COMM_ID_BYTES)
        if rank == 0:
            try:
                uid = ffi.ProofComm.unique_id()
            except Exception:
                ok = 0
        t = torch.tensor(list(uid) + [ok], dtype=torch.uint8, device=dev)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast(t, src=src, group=group)
        raw = bytes(t.cpu().tolist())
        comm = None
        if raw[-1] == 1:
            try:
                comm = ffi.ProofComm(rank, world, raw[:-1], dev.index or 0)
            except Exception:
                comm = None
        flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) != 1:
            if comm is not None:
                comm.close()
            comm = None
        _native_comms[key] = comm if comm is not None else False
    return _native_comms[key] or None


def gather_proofs(local_proofs, proof_size, group=None, device=None):
    """All ranks contribute their fixed-size proof blobs; every rank gets the concatenation in rank order. On an RCCL group the exchange
    is the library's own qpgpu_gather_proofs (ncclAllGather behind the C-ABI, what a Rust host calls); torch.distributed's all_gather
    serves gloo groups (CPU tensors: the tests) and is the path taken if the native communicator could not be set up."""
    import torch
    import torch.distributed as dist
    k = len(local_proofs)
    local_proofs = [p.to_bytes() if hasattr(p, "to_bytes") else bytes(p) for p in local_proofs]
    comm = native_comm(group, device)
    if comm is not None:
        return comm.gather(local_proofs, proof_size)
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
        self._pool = {}                 # (wires, rows) of a node circuit -> ring of reusable page-locked witness matrices

    def level(self, level, width):
        from .prover import WormholeCircuitData
        key = (level, width)
        if key not in self._levels:
            inner = self.leaf if level == 0 else self.level(level - 1, self._level_width.get(level - 1, width))[0]
            node = ffi.BuiltCircuit.aggregation(inner.desc, width, self.node_zero_knowledge)
            self._levels[key] = (WormholeCircuitData.from_built(node, self.device), inner, inner.constants_sigmas_cap(), list(inner.desc.circuit_digest))
        return self._levels[key]

    takes_first_index = True            # prove_chunks(chunks, level, first_index): see aggregate_level

    def prove_chunks(self, chunks, level, first_index=0):
        """One parent per chunk. `first_index` = position of chunks[0] among ALL chunks of this tree level (a rank of a distributed tree
        proves a contiguous run of them): with `salt_seed` set, node i of a level takes blinding key and salt stream number first_index + i,
        so a distributed tree reproduces the single-process tree byte for byte; without it every call draws fresh OS randomness."""
        if not chunks:
            return []
        width = len(chunks[0])
        if any(len(c) != width for c in chunks):            # a ragged last chunk gets its own (narrower) circuit, as in the reference
            return [self.prove_chunks([c], level, first_index + i)[0] for i, c in enumerate(chunks)]
        self._level_width.setdefault(level, width)
        cd, inner, cap, digest = self.level(level, width)
        node = cd._built

        zk = bool(cd.desc.zero_knowledge)                 # a node takes its children's config (tree.rs:111)
        seed = None if self.salt_seed is None else self.salt_seed + (level << 20)

        # witness matrices (35 MB each at 2^15 rows) are reused: ONE ring of RING_GROUPS * GROUP page-locked matrices per matrix shape
        # (every level of a tree has the same), with each level's own public-input vectors beside it
        window = self.GROUP * self.RING_GROUPS
        mats = self._pool.setdefault((cd.num_wires, cd.n), [])
        while len(mats) < min(len(chunks), window):
            mats.append(ffi.witness_buffer(cd.num_wires, cd.n, 0))
        pool = [(mats[i][0], np.zeros(cd.num_public_inputs, dtype=np.uint64)) for i in range(min(len(chunks), window))]
        outer, inner = witness_pool_shape(len(chunks))     # nodes generated at once x threads inside one node (its children / their FRI queries)

        def witness(i, slot=0):
            raw = [p.to_bytes() if hasattr(p, "to_bytes") else bytes(p) for p in chunks[i]]
            try:                                             # every node proof has its own blinding rows
                return node.witness(node.aggregation_targets(cap, digest, raw), proof_blinding_key(seed, first_index + i, b"node") if zk else None,
                                    out=pool[slot], threads=inner)
            except ffi.CircuitError as e:
                raise RuntimeError("Failed to prove: a child proof of this node does not verify: %s" % e)
        salt_key = None if not zk else (ffi.TEST_SALT_KEY if self.salt_seed is not None else os.urandom(32))   # fresh per call; stream = node index

        def prove(wit, first):                               # one device batch (qpgpu_prove_batch: worker streams inside the library)
            b = len(wit)
            wp = (ctypes.c_void_p * b)(*[w.ctypes.data for w, _ in wit])
            pp = (ffi.u64p * b)(*[p.ctypes.data_as(ffi.u64p) for _, p in wit])
            o = ffi.ProveOpts()
            if zk:                                           # proof i of the level takes salt stream base + i
                o.salt_stream = first + ((seed + first_index) if seed is not None else 0)
                keep = ffi.set_salt_key(o, salt_key)         # noqa: F841
            buf = ctypes.create_string_buffer(cd.proof_size * b)
            lens = (ctypes.c_size_t * b)()
            ffi.check(ffi.lib().qpgpu_prove_batch(cd._circuit, b, wp, pp, cd.num_public_inputs, ctypes.byref(o), buf, cd.proof_size, lens))
            return [AggregatedProof(ProofWithPublicInputs(ctypes.string_at(ctypes.addressof(buf) + i * cd.proof_size, cd.proof_size), cd.num_public_inputs), cd)
                    for i in range(b)]
        if len(chunks) == 1:
            return prove([witness(0)], 0)
        # witnesses on host threads (generation releases the GIL), proofs in device batches of GROUP nodes: the device proves a group while
        # the host generates the next ones
        return pipelined_batches(len(chunks), self.GROUP, window, outer, witness, prove)

    def prove_tree(self, leaf_proofs, config, provers=8, first_leaf=0):
        """aggregate_to_tree (tree.rs:56-79) scheduled by DEPENDENCY instead of level by level: a node's witness generation starts when its
        last child proof exists and its proof as soon as the witness does, so the latency-bound top of the tree (4, 2, 1 nodes: a witness and a
        proof each) overlaps the device-bound levels below instead of following them. `provers` host threads each keep one qpgpu_prove in flight
        (the library switches to its batch-mode kernel choices when it sees several). The tree must be uniform (branching^depth leaves).
        With `salt_seed` set the root is byte-identical to the level-by-level tree's (node (level, i) takes the same blinding key and salt
        stream either way). `first_leaf`: position of leaf_proofs[0] among the leaves of a larger tree whose complete subtree this is (a rank of
        aggregate_to_tree_distributed): node i of a level then counts as node first_leaf / branching^(level+1) + i of that level. Returns the
        root AggregatedProof."""
        import queue
        import threading
        b = config.tree_branching_factor
        n = len(leaf_proofs)
        depth = 0
        while b ** depth < n:
            depth += 1
        if b ** depth != n or n < b:
            raise ValueError("prove_tree needs branching^depth leaves")
        for lvl in range(depth):                           # every level's circuit (built once, in this thread)
            self._level_width.setdefault(lvl, b)
            self.level(lvl, b)
        leaves = [p.to_bytes() if hasattr(p, "to_bytes") else bytes(p) for p in leaf_proofs]
        proofs = [leaves] + [[None] * (n // b ** (lvl + 1)) for lvl in range(depth)]      # proofs[l] = the proofs that ENTER level l
        pending = [[b] * len(proofs[lvl + 1]) for lvl in range(depth)]
        zk_key = None if self.salt_seed is not None else os.urandom(32)
        shapes = {(self.level(l, b)[0].num_wires, self.level(l, b)[0].n) for l in range(depth)}
        free = {}
        for shape in shapes:                               # page-locked witness matrices per shape, handed out through a free list
            mats = self._pool.setdefault(shape, [])
            while len(mats) < self.GROUP * self.RING_GROUPS:
                mats.append(ffi.witness_buffer(shape[0], shape[1], 0))
            free[shape] = queue.Queue()
            for m in mats[:self.GROUP * self.RING_GROUPS]:
                free[shape].put(m[0])
        # both queues hand out the HIGHEST level first: a parent whose children exist goes ahead of the level-0 backlog, so subtrees
        # complete early and only the last subtree's chain to the root is left at the end
        todo = queue.PriorityQueue()                       # (-level, index): witnesses to generate
        ready = queue.PriorityQueue()                      # (-level, index, wires, pis): witnesses waiting for a prover thread
        done = threading.Event(); failure = []
        lock = threading.Lock()
        outer, inner = witness_pool_shape(max(2, len(proofs[1])))

        def fail(e):
            with lock:
                failure.append(e)
            done.set()

        def make_witness(level, i):
            try:
                cd, _inner_cd, cap, digest = self.level(level, b)
                zk = bool(cd.desc.zero_knowledge)
                seed = None if self.salt_seed is None else self.salt_seed + (level << 20)
                wires = None
                while wires is None:                       # a witness matrix of this shape (they come back from the prover threads)
                    if done.is_set():
                        return
                    try:
                        wires = free[(cd.num_wires, cd.n)].get(timeout=0.05)
                    except queue.Empty:
                        pass
                if done.is_set():
                    return
                pis = np.zeros(cd.num_public_inputs, dtype=np.uint64)
                try:
                    cd._built.witness(cd._built.aggregation_targets(cap, digest, proofs[level][b * i:b * (i + 1)]),
                                      proof_blinding_key(seed, first_leaf // b ** (level + 1) + i, b"node") if zk else None,
                                      out=(wires, pis), threads=inner)
                except ffi.CircuitError as e:
                    free[(cd.num_wires, cd.n)].put(wires)
                    raise RuntimeError("Failed to prove: a child proof of this node does not verify: %s" % e)
                ready.put((-level, i, wires, pis))
            except Exception as e:                         # noqa: BLE001
                fail(e)

        def witness_thread():
            while not done.is_set():
                try:
                    neg_level, i = todo.get(timeout=0.05)
                except queue.Empty:
                    continue
                if i < 0:
                    return
                make_witness(-neg_level, i)

        def prover_thread():
            while not done.is_set():
                try:
                    item = ready.get(timeout=0.05)
                except queue.Empty:
                    continue
                level, i, wires, pis = -item[0], item[1], item[2], item[3]
                if wires is None:
                    return
                try:
                    cd = self.level(level, b)[0]
                    o = ffi.ProveOpts()
                    if cd.desc.zero_knowledge:              # proof i of a level: salt stream as in prove_chunks
                        if self.salt_seed is not None:
                            o.salt_stream = self.salt_seed + (level << 20) + first_leaf // b ** (level + 1) + i
                            keep = ffi.set_salt_key(o, ffi.TEST_SALT_KEY)      # noqa: F841
                        else:
                            o.salt_stream = (level << 32) + i
                            keep = ffi.set_salt_key(o, zk_key)                 # noqa: F841
                    buf = ctypes.create_string_buffer(cd.proof_size); ln = ctypes.c_size_t()
                    ffi.check(ffi.lib().qpgpu_prove(cd._circuit, wires.ctypes.data, pis.ctypes.data_as(ffi.u64p), len(pis), ctypes.byref(o), buf, cd.proof_size, ctypes.byref(ln)))
                    free[(cd.num_wires, cd.n)].put(wires)
                    pf = ctypes.string_at(buf, ln.value)
                    proofs[level + 1][i] = pf
                    if level + 1 == depth:
                        done.set(); return
                    with lock:
                        pending[level + 1][i // b] -= 1
                        go = pending[level + 1][i // b] == 0
                    if go:
                        todo.put((-(level + 1), i // b))
                except Exception as e:                     # noqa: BLE001
                    fail(e)
        for i in range(len(proofs[1])):
            todo.put((0, i))
        threads = [threading.Thread(target=prover_thread, daemon=True) for _ in range(max(1, provers))] + \
                  [threading.Thread(target=witness_thread, daemon=True) for _ in range(max(1, outer))]
        with _fast_thread_switching():
            for t in threads:
                t.start()
            done.wait()
            for _ in threads:                              # wake whoever still waits on a queue
                todo.put((10 ** 9, -1)); ready.put((10 ** 9, -1, None, None))
                for q in free.values():
                    q.put(None)
            for t in threads:
                t.join()
        if failure:
            raise failure[0]
        cd_root = self.level(depth - 1, b)[0]
        return AggregatedProof(ProofWithPublicInputs(proofs[depth][0], cd_root.num_public_inputs), cd_root)

    GROUP = 8                           # node proofs per device batch
    RING_GROUPS = 3                     # witness buffers kept per level shape: RING_GROUPS * GROUP

    def __call__(self, chunk, level):
        return self.prove_chunks([chunk], level)[0]

    def close(self):
        """Frees the node circuits' device memory and the witness buffers (proofs already returned stay valid; their circuit_data does not)."""
        for cd, _, _, _ in self._levels.values():
            cd.close()
        for ring in self._pool.values():
            for buf in ring:
                ffi.release_witness_buffer(buf)
        self._levels, self._pool = {}, {}

    def proof_size_of_level(self, level, width=DEFAULT_TREE_BRANCHING_FACTOR):
        """Byte size of the proofs that ENTER `level` (aggregate_to_tree_distributed's exchange size)."""
        return self.leaf.proof_size if level == 0 else self.level(level - 1, width)[0].proof_size


def _prove_chunks(chunk_prover, chunks, level, first_index):
    if hasattr(chunk_prover, "prove_chunks"):                     # a device prover takes the whole level as one batch (par_chunks analogue)
        if getattr(chunk_prover, "takes_first_index", False):
            return chunk_prover.prove_chunks(chunks, level, first_index)
        return chunk_prover.prove_chunks(chunks, level)
    return [chunk_prover(c, level) for c in chunks]


def aggregate_level(proofs, chunk_prover, config, level, first_index=0):
    """tree.rs:92-103: consecutive chunks of `tree_branching_factor` children, one parent each, order preserved. `chunk_prover(chunk,
    level)` returns an AggregatedProof; the reference maps it with par_chunks — a prover may batch its chunks on the device instead.
    `first_index`: position of the first chunk among all chunks of the level (non-zero on ranks > 0 of a distributed tree)."""
    b = config.tree_branching_factor
    chunks = [proofs[i:i + b] for i in range(0, len(proofs), b)]
    return _prove_chunks(chunk_prover, chunks, level, first_index)


def aggregate_to_tree(leaf_proofs, config, chunk_prover, streaming=None):
    """tree.rs:56-79: aggregate level by level until one proof is left. A prover that can schedule the whole tree by dependency
    (RecursiveChunkProver.prove_tree) does so for uniform trees of three or more levels unless `streaming` is False (QPGPU_TREE_STREAMING=0
    switches the default off); same root statement, and with a salt seed the same root bytes."""
    if streaming is None:
        streaming = os.environ.get("QPGPU_TREE_STREAMING", "1") != "0"
    b, n = config.tree_branching_factor, len(leaf_proofs)
    # (trees of one or two levels have no latency-bound top to hide: one device batch per level is the cheaper schedule there — 80 vs 97 ms
    # for the reference's (5, 2) bench line, 289 vs 244 ms the other way round for (2, 5); profiles/r03_stream_ab.txt)
    if streaming and hasattr(chunk_prover, "prove_tree") and n == config.num_leaf_proofs and config.tree_depth >= 3:
        return chunk_prover.prove_tree(list(leaf_proofs), config)
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
    (fixed-size blobs) are all-gathered in rank order and the chunks are dealt out in contiguous runs to the first ranks; every rank returns the root.
    `proof_size_of_level(level)` is the byte size of the proofs that ENTER that level (level 0: leaf proofs). One collective per
    level at the top of the tree, none below: RCCL (backend "nccl") on GPU tensors, gloo on CPU tensors."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b = config.tree_branching_factor
    proofs, level, spread = list(local_leaf_proofs), 0, True          # spread: every rank still holds its own run
    total = len(proofs) * world
    while total > 1:
        if spread and level == 0 and hasattr(chunk_prover, "prove_tree") and os.environ.get("QPGPU_TREE_STREAMING", "1") != "0":
            k = 0                                                       # this rank's leaves as a complete subtree of k >= 3 levels: scheduled by dependency
            while b ** k < len(proofs):
                k += 1
            if k >= 3 and b ** k == len(proofs):
                proofs = [chunk_prover.prove_tree(proofs, TreeAggregationConfig(b, k), first_leaf=rank * len(proofs)).proof]
                for _ in range(k):
                    total = (total + b - 1) // b
                level = k
                continue
        if spread and len(proofs) % b == 0 and len(proofs) >= b:
            proofs = [p.proof for p in aggregate_level(proofs, chunk_prover, config, level, first_index=rank * (len(proofs) // b))]
        else:
            if spread:
                everything = gather_proofs(proofs, proof_size_of_level(level), group=group, device=device)
                spread = False
            else:
                everything = proofs                                     # already replicated on every rank
            chunks = [everything[i:i + b] for i in range(0, len(everything), b)]
            # rank r proves the contiguous run [r * per_rank, (r + 1) * per_rank) of the level's chunks: ranks 0 .. ceil(chunks / per_rank) - 1 work
            per_rank = (len(chunks) + world - 1) // world
            my_chunks = chunks[rank * per_rank:(rank + 1) * per_rank]
            mine = [p.proof for p in _prove_chunks(chunk_prover, my_chunks, level, rank * per_rank)] if my_chunks else []
            # one parent per chunk; the gather returns them in rank order = chunk order (padding blobs of idle ranks fall behind the last chunk)
            size = proof_size_of_level(level + 1)
            padded = mine + [bytes(size)] * (per_rank - len(mine))
            gathered = gather_proofs(padded, size, group=group, device=device)
            proofs = gathered[:len(chunks)]
        total = (total + b - 1) // b
        level += 1
    if spread:                                                          # a single rank with a single leaf run
        proofs = gather_proofs(proofs, proof_size_of_level(level), group=group, device=device)[:1]
    return AggregatedProof(proofs[0])
