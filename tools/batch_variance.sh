# Run-to-run spread of the batch-of-64 lines (GPU box): five repeats per setting, proofs/s
cat > /tmp/bv.py <<'PY'
import sys, json
d = json.loads(sys.stdin.read()); print(round(d["value"], 1), end=" ", flush=True)
PY
for p in A W; do
  for env in "" "QPGPU_STREAMS=6" "QPGPU_HANDOFF_SPIN_US=50" "QP_BENCH_NO_PIN=1"; do
    echo -n "$p64 [$env]: "
    for i in 1 2 3 4 5; do env $env python bench.py --no-extras --preset $p --batch 64 --steps 5 2>/dev/null | python /tmp/bv.py; done; echo
  done
done
