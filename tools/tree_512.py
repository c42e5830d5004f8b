#!/usr/bin/env python3
"""BASELINE configs[3] on ONE GPU (rehearsal of the 8-GPU run): 512 wormhole leaf proofs from CircuitInputs (zk config), aggregated with real
recursion to one root proof, verified. Tree shape: branching 2 / depth 9 by default (the reference's default branching; 511 nodes); `branching=8` gives
the 8-ary tree of depth 3 over the same 512 leaves (73 nodes: TreeAggregationConfig is the caller's choice in the reference too, aggregator.rs:46-49,
and its bench sweeps (b, d), benches/aggregator.rs:107-130). Prints one JSON line. Run on the GPU box: python tools/tree_512.py [leaves [plain-nodes|zk-nodes [branching]]]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
torch.cuda.init()
pkg = importlib.import_module("zk-circuits_amd"); th = pkg.test_helpers
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
node_zk = None if len(sys.argv) < 3 or sys.argv[2] != "plain-nodes" else False      # "plain-nodes": nodes without zero knowledge (2^14 rows)
branching = int(sys.argv[3]) if len(sys.argv) > 3 else 2
depth = int(round(np.log(n) / np.log(branching)))
assert branching ** depth == n, "leaves must be branching^depth"
cfg = pkg.CircuitConfig.standard_recursion_zk_config()
cd = pkg.WormholeCircuitData.get(cfg)
inputs = []
for i in range(n):
    ci = th.circuit_test_inputs(); ci.public.exit_account = pkg.BytesDigest(int(i + 1).to_bytes(8, "little") + bytes(24)); inputs.append(ci)
if os.environ.get("TREE_GROUP"):
    pkg.RecursiveChunkProver.GROUP = int(os.environ["TREE_GROUP"])       # nodes per device batch (experiment)
prover = pkg.RecursiveChunkProver(cd, node_zero_knowledge=node_zk)
agg = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(branching, depth), prover)
t0 = time.perf_counter()
warm = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(branching, depth), prover)      # builds the node circuit of every level, warms workspaces
warm.prove_leaves(inputs[:2]); w = [p for p in warm.proofs_buffer]
for lvl in range(depth):
    w = [prover(([w[0], w[-1]] * branching)[:branching], lvl).proof] * 2
if os.environ.get("TREE_WARM"):
    wa = pkg.WormholeProofAggregator(cd, pkg.TreeAggregationConfig(branching, depth), prover); wa.prove_leaves(inputs); wa.aggregate()
t_build = time.perf_counter() - t0
print("node circuits built and warmed in %.1f s" % t_build, file=sys.stderr, flush=True)
t1 = time.perf_counter(); agg.prove_leaves(inputs); t2 = time.perf_counter()
print("leaves done %.2f s" % (t2 - t1), file=sys.stderr, flush=True)
root = agg.aggregate(); t3 = time.perf_counter()
pkg.WormholeVerifier.new(None, circuit_data=root.circuit_data).verify(root.proof); t4 = time.perf_counter()
leaves = agg.extract_leaf_public_inputs(root.proof)
ok = all(int.from_bytes(bytes(l.exit_account[:8]), "little") == i + 1 for i, l in enumerate(leaves))
print(json.dumps({"workload": "BASELINE configs[3] on one GPU: %d wormhole leaf proofs (zk) from CircuitInputs -> recursive aggregation tree (branching %d, depth %d) -> root" % (n, branching, depth),
                  "leaf_stage_s": round(t2 - t1, 3), "leaf_proofs_per_s": round(n / (t2 - t1), 1), "tree_stage_s": round(t3 - t2, 3), "nodes": sum(branching ** k for k in range(depth)), "branching": branching, "depth": depth,
                  "total_s": round(t3 - t1, 3), "proofs_per_s_end_to_end": round(n / (t3 - t1), 1), "root_verify_ms": round((t4 - t3) * 1e3, 2),
                  "root_degree_bits": int(root.circuit_data.desc.degree_bits), "nodes_zero_knowledge": bool(root.circuit_data.desc.zero_knowledge), "root_public_inputs": len(root.proof.public_inputs), "root_bytes": len(root.proof),
                  "leaf_public_inputs_in_order": bool(ok), "node_circuits_build_and_warm_s": round(t_build, 1)}))
