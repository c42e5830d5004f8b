"""Experiment: does a half-empty wave64 issue faster on gfx950? (k_permute_batch with only the even lanes of each wave working: forms 2 / 3.)
It does not: 2^16 permutations run at 0.86 G/s on half-empty waves against 1.21 on full ones (profiles/r02_notes_ab_experiments.md). GPU box."""
import importlib, sys
sys.path.insert(0, ".")
import torch; torch.cuda.init()
H = importlib.import_module("zk-circuits_amd.ffi"); L = H.qpgpu()
for form in (0, 2, 1, 3):
    print(form, [(c, round(L.qpgpu_bench_permute_form(c, 10, form), 3)) for c in (1 << 15, 1 << 16, 1 << 17, 1 << 18)])
