# Re-check of the kernel-choice knobs on batches of 64 W proofs (GPU box), three rounds: each line = setting, proofs/s
# Round 2, final build (run-to-run spread of a batch line is +-4 %): defaults 383.9 / 399.9 / 386.2; SHARED_MODE=1 371.4 / 367.5 / 385.6; COOP_MAX=4096 375.6 / 391.9 / 389.1;
# COOP_MAX=1024 411.8 / 383.2 / 390.7; PAIR_MAX=16384 379.3 / 379.6 / 380.3; SPLIT_QUOTIENT=1 383.9 / 396.1 / 382.4; STREAMS=12 397.7 / 362.2 / 382.0 — nothing beats the defaults outside the noise.
cat > /tmp/kl.py <<'PY'
import sys, json
d = json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], flush=True)
PY
for rep in 1 2 3; do
  for env in "QPGPU_X=0" "QPGPU_SHARED_MODE=1" "QPGPU_COOP_MAX=4096" "QPGPU_COOP_MAX=1024" "QPGPU_PAIR_MAX=16384" "QPGPU_SPLIT_QUOTIENT=1" "QPGPU_STREAMS=12"; do
    env $env python bench.py --no-extras --preset W --batch 64 --steps 5 2>/dev/null | python /tmp/kl.py "$env"
  done
done
