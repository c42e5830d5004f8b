#!/bin/bash
# Aggregation tree: level-by-level (QPGPU_TREE_STREAMING=0) vs scheduled by dependency (default), 512 zk leaves on one GPU, two shapes.
for shape in "zk-nodes 2" "zk-nodes 8" "plain-nodes 2"; do
  for st in 0 1 0 1; do
    r=$(QPGPU_TREE_STREAMING=$st python tools/tree_512.py 512 $shape 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('leaf', j['leaf_stage_s'], 'tree', j['tree_stage_s'], 'total', j['total_s'], 'ok', j['leaf_public_inputs_in_order'])")
    echo "[$shape] streaming=$st: $r"
  done
done
