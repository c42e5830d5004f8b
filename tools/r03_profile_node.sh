#!/bin/bash
# rocprofv3 kernel stats of single proofs of the aggregation node circuit (two zk leaves, recursion gate set, 2^15 rows zk).
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03prof; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_node
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_node -o node -- python3 $R/tools/prove_node.py --steps 20 > $OUT/node_prove.json 2> $OUT/node_prove.err
cp $(find /tmp/pf_node -name "node_kernel_stats.csv" | head -1) $OUT/node_kernel_stats.csv
cat $OUT/node_prove.json; head -12 $OUT/node_kernel_stats.csv | cut -c1-170
