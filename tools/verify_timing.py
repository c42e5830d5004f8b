#!/usr/bin/env python3
"""Phases of qpgpu_verify_batch on the reference's golden proof (run on the GPU box; QPGPU_VERIFY_TIMING=1 prints them per call)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch; torch.cuda.init()
import importlib; H = importlib.import_module("zk-circuits_amd.ffi")          # the product binding only (no oracle in a measurement tool)
g = os.path.join(ROOT, "tests", "golden"); rd = lambda f: open(os.path.join(g, f), "rb").read()
c, vb, pf = rd("common.bin"), rd("verifier.bin"), rd("proof.bin")
vo = vb[:len(vb) - len(c)] if vb.endswith(c) and len(vb) > len(c) else vb
sh = H.ReferenceCircuitShape(c, vo); gv = H.GpuVerifier(sh.desc, sh.constants_sigmas_cap)
for _ in range(3): gv.verify(pf)
t = time.perf_counter()
for _ in range(50): gv.verify(pf)
print("queues %s: single verify %.3f ms" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), (time.perf_counter() - t) / 50 * 1e3))
for n in (64, 512):
    gv.verify_batch([pf] * n)
    t = time.perf_counter()
    for _ in range(5): gv.verify_batch([pf] * n)
    dt = (time.perf_counter() - t) / 5
    print("  batch %d: %.3f ms = %.0f proofs/s" % (n, dt * 1e3, n / dt))
