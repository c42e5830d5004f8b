# quick same-box sanity + timing set used while tuning kernels (GPU box): kernel parity tests, permutation rate sweep, W / A / V single, W batch 64 (twice)
# columns: preset, proofs/s, leaf-hash launch ms, tree-levels group ms, FRI-layers group ms
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden_vectors.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -2
python tools/permute_rate.py gpurun_out/rate_quick.json 2>&1 | tail -2
cat > /tmp/ab_line.py <<'PY'
import sys, json
d = json.loads(sys.stdin.read()); k = d["kernels"]
print(sys.argv[1], d["value"], d["roofline"]["avg_launch_ms"], round(k.get("merkle_levels", {}).get("avg_ms", 0), 4), round(k.get("fri_layers", {}).get("avg_ms", 0), 4))
PY
for i in 1 2; do
  for p in W A V; do python bench.py --no-extras --preset $p --steps 20 2>/dev/null | python /tmp/ab_line.py $p; done
  python bench.py --no-extras --preset W --batch 64 --steps 5 2>/dev/null | python /tmp/ab_line.py W64
done
