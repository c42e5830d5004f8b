# Re-check of the tree-level kernel thresholds for a single W proof (GPU box): the 16-lane form from QPGPU_COOP_MAX parents down, the 2-lane form up to
# QPGPU_PAIR_MAX parents, one permutation per lane above. Columns: setting, proofs/s, tree-levels group ms, FRI-layers group ms.
# Round 2, final build: coop_max 4096 / 8192 / 16384 / 32768 / 65536 = 267.0 / 270.7 / 272.7 / 263.3 / 250.6 proofs/s; pair_max 262144 / 32768 / 16384 = 273.0 / 266.4 / 261.8:
# the defaults (16384, 262144) stay.
cat > /tmp/ab_line.py <<'PY'
import sys, json
d = json.loads(sys.stdin.read()); k = d["kernels"]
print(sys.argv[1], d["value"], round(k.get("merkle_levels", {}).get("avg_ms", 0), 4), round(k.get("fri_layers", {}).get("avg_ms", 0), 4), flush=True)
PY
for rep in 1 2; do
  for cm in 4096 8192 16384 32768 65536; do QPGPU_COOP_MAX=$cm python bench.py --no-extras --preset W --steps 40 2>/dev/null | python /tmp/ab_line.py "coop_max=$cm"; done
  for pm in 262144 32768 16384; do QPGPU_PAIR_MAX=$pm python bench.py --no-extras --preset W --steps 40 2>/dev/null | python /tmp/ab_line.py "pair_max=$pm"; done
done
