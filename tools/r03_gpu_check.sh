#!/bin/bash
# Round-3 GPU pass: the tests touched this round, the default bench line, the RCCL path at world size 1 (native gather), configs[3] on one GPU.
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_comm.py tests/test_gpu_verifier.py tests/test_gpu_host_mirror.py tests/test_gpu_wormhole_circuit.py tests/test_gpu_tree_512.py -m gpu -x -q > gpurun_out/r03_gputests_2.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r03_gputests_2.log
tail -5 gpurun_out/r03_gputests_2.log
python bench.py > gpurun_out/r03_bench_W.json 2> gpurun_out/r03_bench_W.err; echo "bench rc=$?"
QP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29700 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 20 --no-tree --no-cpu-baseline > gpurun_out/r03_bench_W_forcedist.json 2> gpurun_out/r03_bench_W_forcedist.err; echo "forced-dist bench rc=$?"
tail -3 gpurun_out/r03_bench_W_forcedist.err
python tools/tree_512.py > gpurun_out/r03_tree_512_one_gpu.json 2> gpurun_out/r03_tree_512.err; echo "tree rc=$?"; cat gpurun_out/r03_tree_512_one_gpu.json
