#!/bin/bash
# Round-3 rocprofv3 summaries (run on the GPU box from the repo root: bash tools/profile_round3.sh). One workload per profiler run, extras off:
#   W = BASELINE configs[1] (single proof of the wormhole circuit, zk, 2^14; the default bench line)   B64 = configs[2] (batch of 64)
#   V = configs[4] (2^20 rows)   node = single proofs of the aggregation node circuit on the recursion gate set (two zk leaves, 2^15 rows)
# Outputs under gpurun_out/r03prof/: <cfg>_kernel_stats.csv, <cfg>_bench.json, timeline_W.txt; copy what is to be judged into profiles/r03_*.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03prof
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
run B64 --preset B --batch 64 --steps 3 --warmup 1
run V --preset V --steps 3 --warmup 1
rm -rf /tmp/pf_node
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_node -o node -- python3 $R/tools/prove_node.py --steps 20 > $OUT/node_prove.json 2> $OUT/node_prove.err
cp $(find /tmp/pf_node -name "node_kernel_stats.csv" | head -1) $OUT/node_kernel_stats.csv
echo done
