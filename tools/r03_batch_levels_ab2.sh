#!/bin/bash
# Batch mode, tree levels above the 16-lane switch (4096 parents): 2-lane pair form (default) vs one permutation per lane (QPGPU_PAIR_MAX=0).
for v in "A=1" "QPGPU_PAIR_MAX=0" "QPGPU_PAIR_MAX=16384" "A=1" "QPGPU_PAIR_MAX=0" "QPGPU_PAIR_MAX=16384"; do
  out=""
  for rep in 1 2 3; do
    r=$(env $v python bench.py --no-extras --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'])")
    out="$out $r"
  done
  echo "[$v] proofs/s:$out"
done
