# quick same-box sanity + timing set used while tuning kernels (GPU box): kernel parity tests, permutation rate sweep, W / A / V single, W batch 64 (twice)
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden_vectors.py tests/test_gpu_prove.py -m gpu -x -q 2>&1 | tail -2
python tools/permute_rate.py gpurun_out/rate_quick.json 2>&1 | tail -2
for i in 1 2; do
  for p in W A V; do python bench.py --no-extras --preset $p --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p', d['value'], d['roofline']['avg_launch_ms'])"; done
  python bench.py --no-extras --preset W --batch 64 --steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('W64', d['value'])"
done
