#!/usr/bin/env python3
"""Where the aggregation tree's time goes, level by level (run on the GPU box): for a tree over N zk wormhole leaves (default 64), per level:
nodes, wall time, time the proving thread waited for witnesses, time inside qpgpu_prove_batch, single-thread cost of one node witness.
python tools/tree_stage_timeline.py [leaves] [zk-nodes|plain-nodes]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ["QPGPU_TREE_STREAMING"] = "0"          # the level-by-level schedule is what this tool breaks down; the dependency-driven total is measured at the end
import numpy as np
import torch
torch.cuda.init()
pkg = importlib.import_module("zk-circuits_amd"); th = pkg.test_helpers
agg = importlib.import_module("zk-circuits_amd.aggregator"); ffi = pkg.ffi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
node_zk = False if (len(sys.argv) > 2 and sys.argv[2] == "plain-nodes") else None
depth = int(round(np.log2(n)))
cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.standard_recursion_zk_config())
ins = []
for i in range(n):
    ci = th.circuit_test_inputs(); ci.public.exit_account = pkg.BytesDigest(int(i + 1).to_bytes(8, "little") + bytes(24)); ins.append(ci)
prover = pkg.RecursiveChunkProver(cd, node_zero_knowledge=node_zk)
a = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, depth), prover)
a.prove_leaves(ins); a.aggregate()                      # builds every level's circuit, warms workspaces and rings
# instrument: time in the device batch vs waiting on witnesses
stats = {}
orig_batch = ffi.lib().qpgpu_prove_batch


class Timed:
    def __init__(self): self.t = 0.0; self.calls = 0
    def __call__(self, *args):
        t0 = time.perf_counter(); r = orig_batch(*args); self.t += time.perf_counter() - t0; self.calls += 1; return r
timed = Timed()
timed.argtypes = orig_batch.argtypes
ffi.lib().qpgpu_prove_batch = timed
orig_chunks = prover.prove_chunks


def prove_chunks(chunks, level, first_index=0):
    t_dev0, c0 = timed.t, timed.calls
    t0 = time.perf_counter(); out = orig_chunks(chunks, level, first_index); dt = time.perf_counter() - t0
    stats[level] = {"nodes": len(chunks), "wall_ms": round(dt * 1e3, 2), "in_prove_batch_ms": round((timed.t - t_dev0) * 1e3, 2), "batches": timed.calls - c0,
                    "ms_per_node": round(dt * 1e3 / len(chunks), 2)}
    return out
prover.prove_chunks = prove_chunks
a2 = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, depth), prover)
t0 = time.perf_counter(); a2.prove_leaves(ins); t1 = time.perf_counter(); root = a2.aggregate(); t2 = time.perf_counter()
# one node's witness alone: single thread and the pool's inner thread count
ncd, _inner, cap, digest = prover.level(0, 2)
node = ncd._built
lv = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, 1)); lv.prove_leaves(ins[:2]); two = [p.to_bytes() for p in lv.proofs_buffer]
x = node.aggregation_targets(cap, digest, two)
buf = ffi.witness_buffer(ncd.num_wires, ncd.n, ncd.num_public_inputs)
key = bytes(range(32)) if ncd.desc.zero_knowledge else None
def tm(f, k=5):
    f(); t = time.perf_counter()
    for _ in range(k): f()
    return round((time.perf_counter() - t) / k * 1e3, 2)
wit = {"targets_ms": tm(lambda: node.aggregation_targets(cap, digest, two)),
       "witness_1_thread_ms": tm(lambda: node.witness(x, key, out=buf, threads=1)),
       "witness_2_threads_ms": tm(lambda: node.witness(x, key, out=buf, threads=2)),
       "witness_8_threads_ms": tm(lambda: node.witness(x, key, out=buf, threads=8))}
gc = ffi.GpuCircuit(node.desc)
w, p = node.witness(x, key, out=buf, threads=8)
wit["prove_single_ms"] = tm(lambda: gc.prove(w, p, salt_seed=1))
wit["prove_batch8_ms_per_node"] = round(tm(lambda: gc.prove_batch([w] * 8, [p] * 8, salt_seed=1), 3) / 8, 2)
os.environ["QPGPU_TREE_STREAMING"] = "1"
prover.prove_chunks = orig_chunks
a3 = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, depth), prover)
a3.prove_leaves(ins); a3.aggregate()              # warm the single-proof path of every level
a3 = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(2, depth), prover)
t3 = time.perf_counter(); a3.prove_leaves(ins); t4 = time.perf_counter(); a3.aggregate(); t5 = time.perf_counter()
wit["dependency_driven_leaf_stage_s"] = round(t4 - t3, 4); wit["dependency_driven_tree_stage_s"] = round(t5 - t4, 4)
print(json.dumps({"leaves": n, "node_zero_knowledge": bool(ncd.desc.zero_knowledge), "node_degree_bits": int(ncd.desc.degree_bits), "usable_cpus": importlib.import_module("zk-circuits_amd.hostcpus").usable_cpus(),
                  "leaf_stage_s": round(t1 - t0, 4), "tree_stage_s": round(t2 - t1, 4), "levels": stats, "one_node": wit}, indent=1))
