#!/bin/bash
# Per-config rocprofv3 summaries (run on the GPU box from the repo root: bash tools/profile_round2.sh). One config per profiler run, extras
# off (--no-extras), so that every kernel_stats.csv describes exactly one workload:
#   W  = BASELINE configs[1] (single proof of the wormhole circuit, zk, 2^14; the default)   B  = the same shape, synthetic gate contents     A = non-zk leaf 2^13      V = configs[4] (2^20 rows)      B64 = configs[2] (batch of 64)
# Outputs under gpurun_out/r02prof/: <cfg>_kernel_stats.csv, <cfg>_bench.json, timeline_B.txt; copy what is to be judged into profiles/.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # name, bench args...
  name=$1; shift
  rm -rf /tmp/pf_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$name -o $name -- python3 $R/bench.py --no-extras "$@" > $OUT/${name}_bench.json 2> $OUT/${name}_bench.err
  cp $(find /tmp/pf_$name -name "${name}_kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  echo "profiled $name"
}
run W --preset W --steps 20 --warmup 3
python3 $R/tools/trace_timeline.py $(find /tmp/pf_W -name "W_kernel_trace.csv" | head -1) > $OUT/timeline_W.txt
run B --preset B --steps 20 --warmup 3
python3 $R/tools/trace_timeline.py $(find /tmp/pf_B -name "B_kernel_trace.csv" | head -1) > $OUT/timeline_B.txt
run A --preset A --steps 20 --warmup 3
run B64 --preset B --batch 64 --steps 3 --warmup 1
run V --preset V --steps 3 --warmup 1
echo done
