#!/bin/bash
# Round-3 final pass, part 1: the whole GPU suite, the default bench line, the other bench configurations, the RCCL path at world size 1.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03_gputests_final.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r03_gputests_final.log
tail -14 gpurun_out/r03_gputests_final.log
python bench.py > gpurun_out/r03_bench_W.json 2> gpurun_out/r03_bench_W.err; echo "bench W rc=$?"
python bench.py --batch 64 --steps 6 --warmup 2 > gpurun_out/r03_bench_W64.json 2> gpurun_out/r03_bench_W64.err; echo "bench W64 rc=$?"
python bench.py --preset V --steps 3 --warmup 1 > gpurun_out/r03_bench_V.json 2> gpurun_out/r03_bench_V.err; echo "bench V rc=$?"
python bench.py --preset A > gpurun_out/r03_bench_A.json 2> gpurun_out/r03_bench_A.err; echo "bench A rc=$?"
QP_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29700 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r03_bench_W_forcedist.json 2> gpurun_out/r03_bench_W_forcedist.err; echo "forced-dist rc=$?"
