# End-of-round measurements on the final tree (GPU box, from the repo root): the whole GPU suite, the default bench line with all extras,
# the other presets, the batch lines, the permutation-rate sweep. Per-config rocprofv3 summaries: tools/profile_round2.sh.
set -e
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -2 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python tools/permute_rate.py $O/r02_poseidon_issue_bound.json > $O/permute_rate.txt 2>/dev/null
python bench.py > $O/bench_W.json 2> $O/bench_W.err
python bench.py --preset B --no-tree > $O/bench_B.json 2>/dev/null
python bench.py --preset A --no-tree > $O/bench_A.json 2>/dev/null
python bench.py --preset WN --no-tree > $O/bench_WN.json 2>/dev/null
python bench.py --preset V --steps 5 --warmup 1 --no-pipelined --no-tree > $O/bench_V.json 2>/dev/null
python bench.py --batch 64 --steps 5 --warmup 1 --no-tree > $O/bench_W64.json 2>/dev/null
python bench.py --preset B --batch 64 --steps 5 --warmup 1 --no-tree > $O/bench_B64.json 2>/dev/null
python bench.py --preset A --batch 64 --steps 5 --warmup 1 --no-tree > $O/bench_A64.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/final/bench_*.json")):
    d = json.loads(open(f).read())
    print(f.split("/")[-1], d["value"], d["ms_per_step"], (d.get("pipelined") or {}).get("proofs_per_s"), d.get("bit_exact_vs_cpu"), d.get("bit_exact_batch"))
PY
echo done
