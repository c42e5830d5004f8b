#!/usr/bin/env python3
"""One workload for the profiler: proofs of an aggregation-tree node circuit (two zk leaf proofs verified in-circuit; 2^15 rows, zk) and of the
wormhole circuit itself on the reference's test inputs, witnesses resident in HBM. Prints one JSON line with the per-proof times.
  rocprofv3 --kernel-trace --stats ... -- python3 tools/prove_node.py [--steps K]"""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
torch.cuda.init()
ffi = importlib.import_module("zk-circuits_amd.ffi"); th = importlib.import_module("zk-circuits_amd.test_helpers")
ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=20); ap.add_argument("--what", default="node", choices=["node", "wormhole"])
a = ap.parse_args()
dev = torch.device("cuda", 0)
leaf = ffi.BuiltCircuit(ffi.QPC_WORMHOLE, True)
gl = ffi.GpuCircuit(leaf.desc)
lw, lp = leaf.witness(th.circuit_test_inputs().fill_targets(), ffi.TEST_SALT_KEY)
if a.what == "wormhole":
    c, g, w, p = leaf, gl, lw, lp
else:
    proofs = [gl.prove(lw, lp, salt_seed=i) for i in range(2)]
    c = ffi.BuiltCircuit.aggregation(leaf.desc, 2); g = ffi.GpuCircuit(c.desc)
    w, p = c.witness(c.aggregation_targets(gl.cap(), leaf.desc.circuit_digest, proofs), ffi.TEST_SALT_KEY)
wd = torch.from_numpy(w.view(np.int64)).to(dev); torch.cuda.synchronize()
for _ in range(3):
    g.prove(None, p, salt_seed=1, wires_dev_ptr=wd.data_ptr())
t = time.perf_counter()
for i in range(a.steps):
    pf = g.prove(None, p, salt_seed=1, wires_dev_ptr=wd.data_ptr())
dt = (time.perf_counter() - t) / a.steps
print(json.dumps({"workload": a.what, "degree_bits": int(c.desc.degree_bits), "zero_knowledge": bool(c.desc.zero_knowledge), "rows": c.row_counts(),
                  "ms_per_proof": round(dt * 1e3, 4), "proof_bytes": len(pf), "stage_ms": g.stage_ms() if hasattr(g, "stage_ms") else None}))
