#!/usr/bin/env python3
"""Where one aggregation-tree node's time goes (run on the GPU box): input encoding, witness generation (QPC_TIMING=1 prints its phases),
host-to-device copy + proof. Leaves: synthetic preset B (zk, 2^14)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
try:
    import torch
    torch.cuda.init()
except Exception:
    torch = None
ffi = importlib.import_module("zk-circuits_amd.ffi")
agg = importlib.import_module("zk-circuits_amd.aggregator")
sc = ffi.SynthCircuit("B"); gc = ffi.GpuCircuit(sc.desc)
leaves = []
for i in range(2):
    w, p = sc.witness(0x51570000 + i); leaves.append(gc.prove(w, p, salt_seed=i))


class Leaf:
    desc = sc.desc; proof_size = gc.proof_size
    def constants_sigmas_cap(self):
        return gc.cap()
pr = agg.RecursiveChunkProver(Leaf())
node = pr(leaves, 0)                       # builds + warms level 0
cd, inner, cap, digest = pr.level(0, 2)
b = cd._built
def t(f, n=10):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n * 1e3, r
ms_t, x = t(lambda: b.aggregation_targets(cap, digest, leaves))
ms_w, wp = t(lambda: b.witness(x))
ms_wr, _ = t(lambda: b.witness(x, out=wp))
g2 = ffi.GpuCircuit(b.desc)
ms_p, _ = t(lambda: g2.prove(wp[0], wp[1]))
ms_all, _ = t(lambda: pr(leaves, 0))
print("node circuit rows", b.row_counts(), "k", b.desc.degree_bits)
print("targets %.2f ms | witness %.2f ms (reused buffer %.2f) | H2D + prove %.2f ms | whole chunk prover call %.2f ms" % (ms_t, ms_w, ms_wr, ms_p, ms_all))
