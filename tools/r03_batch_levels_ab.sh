#!/bin/bash
# Batch mode (8 worker streams): where the tree levels switch from the 2-lane pair form to the 16-lane fused form (QPGPU_COOP_MAX = largest
# level, in parents, hashed 16 lanes per permutation). Round 2 settled on 16384 with four worker streams; with eight the device is fuller.
for v in "QPGPU_COOP_MAX=16384" "QPGPU_COOP_MAX=8192" "QPGPU_COOP_MAX=4096" "QPGPU_COOP_MAX=2048" "QPGPU_COOP_MAX=1024" "QPGPU_COOP_MAX=256"; do
  out=""
  for rep in 1 2 3 4; do
    r=$(env $v python bench.py --no-extras --batch 64 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'])")
    out="$out $r"
  done
  echo "[$v] proofs/s:$out"
done
