#!/usr/bin/env python3
"""Saturated throughput of the one-permutation-per-lane Poseidon kernel on this GPU (qpgpu_bench_permute, grid sweep), written next to the
first-principles issue bound of tools/poseidon_issue_bound.py: profiles/r02_poseidon_issue_bound.json gains
measured_saturated_G_perm_per_s (bench.py reads both for its valu_roofline object). Run on the GPU box; copy the JSON back."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch; torch.cuda.init()
H = importlib.import_module("zk-circuits_amd.ffi")
L = H.qpgpu()
SIZES = (1 << 16, 1 << 17, 1 << 18, 1 << 20, 1 << 22, 1 << 23)
sweep = [(cnt, round(L.qpgpu_bench_permute_form(cnt, 10, 1), 3)) for cnt in SIZES]            # fast form of the partial rounds (full-chip stages)
literal = [(cnt, round(L.qpgpu_bench_permute_form(cnt, 10, 0), 3)) for cnt in SIZES]          # literal form (latency-bound stages)
print("G perm/s, fast form:", sweep)
print("G perm/s, literal form:", literal)
out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_poseidon_issue_bound.json")
src = os.path.join(ROOT, "profiles", "r02_poseidon_issue_bound.json")
d = json.load(open(src)) if os.path.exists(src) else {}
d["measured_sweep_G_perm_per_s"] = sweep
d["measured_sweep_literal_form_G_perm_per_s"] = literal
d["measured_saturated_G_perm_per_s"] = max(v for _, v in sweep)
d["measured_source"] = "tools/permute_rate.py (qpgpu_bench_permute: k_permute_batch, 10 launches per grid size) on MI355X"
json.dump(d, open(out, "w"), indent=1)
