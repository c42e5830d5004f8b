#!/bin/bash
# One rank on the whole GPU with the host pools of a crowded node: LOCAL_WORLD_SIZE=2 / 4 / 8 (usable cores 8 / 4 / 2 of the 16-core share).
run() {
  env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = j['aggregation_tree']; r = j['wormhole_circuit']
print('value', j['value'], 'pipelined', (j.get('pipelined') or {}).get('proofs_per_s'), 'tree leaf_s', t.get('leaf_stage_s'), 'tree_s', t.get('tree_stage_s'), 'wide', t.get('wide_tree', {}).get('tree_stage_s'), 'leaves_from_inputs', r['leaves_from_inputs']['proofs_per_s'], 'verify64_s', t.get('leaf_batch_verify', {}).get('seconds'))"
}
echo "LOCAL_WORLD_SIZE unset:"; run A=1
echo "LOCAL_WORLD_SIZE=2:"; run LOCAL_WORLD_SIZE=2
echo "LOCAL_WORLD_SIZE=4:"; run LOCAL_WORLD_SIZE=4
echo "LOCAL_WORLD_SIZE=8:"; run LOCAL_WORLD_SIZE=8
