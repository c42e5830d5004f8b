set -e
D=gpurun_out/$1; mkdir -p $D
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden_vectors.py -m gpu -x -q > $D/tests.log 2>&1; tail -1 $D/tests.log
python tools/permute_rate.py $D/rate.json > $D/rate.log 2>&1; tail -1 $D/rate.log
python bench.py --no-cpu-baseline --no-tree --no-real-circuit > $D/bench.json 2> $D/bench.err
python bench.py --no-extras --preset V --steps 5 > $D/benchV.json 2>> $D/bench.err
python - <<PY
import json
for f in ['$D/bench.json','$D/benchV.json']:
    d=json.loads(open(f).read()); print(f, d['value'], d['ms_per_step'], (d.get('pipelined') or {}).get('proofs_per_s'), d['roofline']['avg_launch_ms'])
PY
