set -e
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q 2>&1 | tail -2 > gpurun_out/final/pytest_gpu.txt
python bench.py > gpurun_out/final/bench_B.json 2> gpurun_out/final/bench_B.err
python bench.py --preset A > gpurun_out/final/bench_A.json 2>/dev/null
python bench.py --preset R --no-cpu-baseline > gpurun_out/final/bench_R.json 2>/dev/null
python bench.py --preset V --steps 3 --warmup 1 --no-pipelined > gpurun_out/final/bench_V.json 2>/dev/null
python bench.py --batch 64 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final/bench_B_batch64.json 2>/dev/null
python tools/permute_rate.py > gpurun_out/final/permute_rate.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipelined > $GRAFT_REPO_ROOT/gpurun_out/final/bench_under_rocprof.json 2>/dev/null
cp /tmp/pf/b_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/final/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/trace_timeline.py /tmp/pf/b_kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/final/timeline_B.txt
echo done
