#!/bin/bash
# Fused quotient kernel: gate wires loaded per gate (0) vs the cheap gates fed from the permutation argument's chunk loads (1, leaf gate set).
for v in 0 1 0 1; do
  r=$(QPGPU_QUOTIENT_SHARED_LOADS=$v python bench.py --no-extras --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['kernels'].get('quotient_eval',{}).get('avg_ms'))")
  echo "B64 shared_loads=$v: proofs/s, quotient avg ms: $r"
done
for v in 0 1; do
  r=$(QPGPU_QUOTIENT_SHARED_LOADS=$v python bench.py --no-extras --preset V --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['ms_per_proof'], j['kernels'].get('quotient_eval',{}).get('avg_ms'))")
  echo "V shared_loads=$v: ms/proof, quotient avg ms: $r"
done
