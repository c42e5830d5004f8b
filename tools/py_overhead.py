#!/usr/bin/env python3
"""How much of a single proof's wall time is the Python binding? (GPU box) Times N W-preset proofs through GpuCircuit.prove (what bench.py calls)
and through a bare ctypes call of qpgpu_prove with every argument prepared once and no copy of the proof out of the reused buffer."""
import ctypes, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
torch.cuda.init()
H = importlib.import_module("zk-circuits_amd.ffi")
sc = H.SynthCircuit("B"); gc = H.GpuCircuit(sc.desc); L = H.qpgpu()
w, p = sc.witness(1); wd = torch.from_numpy(w.view(np.int64)).cuda(); torch.cuda.synchronize()
N = 200
for _ in range(20): gc.prove(None, p, salt_seed=1, wires_dev_ptr=wd.data_ptr())
t = time.perf_counter()
for _ in range(N): gc.prove(None, p, salt_seed=1, wires_dev_ptr=wd.data_ptr())
a = (time.perf_counter() - t) / N * 1e3
o = H.ProveOpts(); o.salt_stream = 1; keep = H.set_salt_key(o, H.TEST_SALT_KEY); o.wires_on_device = 1
buf = ctypes.create_string_buffer(gc.proof_size); n = ctypes.c_size_t(); pp = H.ptr(p); wp = wd.data_ptr()
t = time.perf_counter()
for _ in range(N): L.qpgpu_prove(gc.h, wp, pp, len(p), ctypes.byref(o), buf, len(buf), ctypes.byref(n))
b = (time.perf_counter() - t) / N * 1e3
print("GpuCircuit.prove %.4f ms/proof, bare qpgpu_prove %.4f ms/proof, binding overhead %.1f us" % (a, b, (a - b) * 1e3))
