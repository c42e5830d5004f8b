#!/bin/bash
# Round-3 final pass, part 2: rocprofv3 kernel stats (W, B64, V, node), PMC traffic (W, B64, V), configs[3] on one GPU (two shapes), tree timeline.
bash tools/profile_round3.sh 2>&1 | tail -6
python tools/pmc_traffic.py --preset W > gpurun_out/r03_pmc_W.txt 2>&1; echo "pmc W rc=$?"
python tools/pmc_traffic.py --preset B --batch 64 > gpurun_out/r03_pmc_B64.txt 2>&1; echo "pmc B64 rc=$?"
python tools/pmc_traffic.py --preset V > gpurun_out/r03_pmc_V.txt 2>&1; echo "pmc V rc=$?"
cp profiles/traffic.json gpurun_out/r03_traffic.json
python tools/tree_512.py 512 zk-nodes 2 > gpurun_out/r03_tree_512_b2.json 2>/dev/null; cat gpurun_out/r03_tree_512_b2.json
python tools/tree_512.py 512 zk-nodes 8 > gpurun_out/r03_tree_512_b8.json 2>/dev/null; cat gpurun_out/r03_tree_512_b8.json
python tools/tree_512.py 512 plain-nodes 2 > gpurun_out/r03_tree_512_b2_plain.json 2>/dev/null; cat gpurun_out/r03_tree_512_b2_plain.json
python tools/tree_stage_timeline.py 64 > gpurun_out/r03_tree_timeline_zk.json 2>/dev/null
