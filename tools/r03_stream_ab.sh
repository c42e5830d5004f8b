#!/bin/bash
run() {
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = j['aggregation_tree']; r = j['wormhole_circuit']
print('tree_s', t.get('tree_stage_s'), 'wide', t.get('wide_tree', {}).get('tree_stage_s'), 'leaves_from_inputs', r['leaves_from_inputs']['proofs_per_s'], [ (m['branching'], m['depth'], m['aggregate_ms']) for m in t.get('bench_matrix', [])])"
}
echo "streaming:"; run A=1; run A=1
echo "level by level / batches:"; run QPGPU_TREE_STREAMING=0; run QPGPU_TREE_STREAMING=0
