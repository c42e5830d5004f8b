#!/usr/bin/env python3
"""One-off extended run of tests/test_gpu_prove.py::test_random_gate_mixes_match_oracle over many seeds (GPU box):
  QPGPU_POSEIDON_FAST=1 python tools/fuzz_gate_mixes.py 6 80      # every leaf hash in the fast form, whatever the tree size
Each seed is a random circuit package (degree, zk, gate mix incl. recursion kinds, copies, FRI schedule); GPU bytes must equal the oracle's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
import test_gpu_prove as T
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    T.test_random_gate_mixes_match_oracle(seed)
    if seed % 10 == 0: print("seed", seed, "ok", flush=True)
print("seeds %d..%d ok (QPGPU_POSEIDON_FAST=%s)" % (lo, hi - 1, os.environ.get("QPGPU_POSEIDON_FAST", "auto")))
