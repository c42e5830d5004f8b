# Re-check of the kernel-choice knobs on a single W proof (GPU box), two rounds: each line = setting, proofs/s
# Round 2, final build: defaults 274.1 / 271.9; NTT16=2 266.4 / 266.4; NTT16=0 273.5 / 270.3; NTT_M8=0 271.6 / 273.4; NTT_STRIDED=0 263.2 / 262.1; SPLIT_FORK_LATE=0 264.5 / 266.3;
# SPLIT_QUOTIENT=0 260.4 / 260.1; PAIR_LEAVES=1 259.5 / 259.6; FUSE_LEVELS=0 273.1 / 271.5; L0_TABLE=0 272.7 / 272.5; POW_FIRST_SHIFT=1 273.9 / 273.2 — every default stays.
cat > /tmp/kl.py <<'PY'
import sys, json
d = json.loads(sys.stdin.read()); print(sys.argv[1], d["value"], flush=True)
PY
for rep in 1 2; do
  for env in "QPGPU_X=0" "QPGPU_NTT16=2" "QPGPU_NTT16=0" "QPGPU_NTT_M8=0" "QPGPU_NTT_STRIDED=0" "QPGPU_SPLIT_FORK_LATE=0" "QPGPU_SPLIT_QUOTIENT=0" "QPGPU_PAIR_LEAVES=1" "QPGPU_FUSE_LEVELS=0" "QPGPU_L0_TABLE=0" "QPGPU_POW_FIRST_SHIFT=1"; do
    env $env python bench.py --no-extras --preset W --steps 40 2>/dev/null | python /tmp/kl.py "$env"
  done
done
