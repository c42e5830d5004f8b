#!/bin/bash
# Two ranks sharing one GPU (gloo collectives): the hand-off wait mode and pool sizing under LOCAL_WORLD_SIZE=2.
run() {
  env "$@" QP_BENCH_BACKEND=gloo QP_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = j['aggregation_tree']
print('value', j['value'], 'pipelined', (j.get('pipelined') or {}).get('proofs_per_s'), 'tree leaf_s', t.get('leaf_stage_s'), 'tree_s', t.get('tree_stage_s'), 'verify64_s', t.get('leaf_batch_verify', {}).get('seconds'))"
}
echo "default:"; run A=1
echo "QPGPU_HANDOFF_SPIN_US=2000:"; run QPGPU_HANDOFF_SPIN_US=2000
echo "QPGPU_HOST_THREADS=16:"; run QPGPU_HOST_THREADS=16
echo "QPGPU_STREAMS=4:"; run QPGPU_STREAMS=4
