# Same-box A/B of two builds of libqpgpu.so (GPU box; it has hipcc): build A = the working tree, build B = the working tree with the files
# named after the output directory taken from git HEAD. Alternates A/B/A/B over W, A, V single proofs and a batch of 64 W proofs.
#   bash tools/ab_two_builds.sh <outdir-name> <file-from-HEAD> [more files]
set -e
D=gpurun_out/$1; shift; mkdir -p $D
cd zk-circuits_amd/csrc
make -j16 > /dev/null 2>&1; cp ../libqpgpu.so /tmp/libA.so
mkdir -p /tmp/keep; for f in "$@"; do cp $f /tmp/keep/$(basename $f); cp ../../tools/.ab_heads/head_$(basename $f) $f; done
make -j16 > /dev/null 2>&1; cp ../libqpgpu.so /tmp/libB.so
for f in "$@"; do cp /tmp/keep/$(basename $f) $f; done
cd ../..
cat > /tmp/ab_line.py <<'PY'
import sys, json
d = json.loads(sys.stdin.read()); k = d["kernels"]
print(sys.argv[1], sys.argv[2], d["value"], d["roofline"]["avg_launch_ms"], round(k.get("merkle_levels", {}).get("avg_ms", 0), 4), flush=True)
PY
for rep in 1 2; do for b in A B; do
  cp /tmp/lib$b.so zk-circuits_amd/libqpgpu.so
  for p in W A V; do python bench.py --no-extras --preset $p --steps 30 2>/dev/null | python /tmp/ab_line.py $b $p; done
  python bench.py --no-extras --preset W --batch 64 --steps 6 2>/dev/null | python /tmp/ab_line.py $b W64
done; done | tee $D/ab.txt
cp /tmp/libA.so zk-circuits_amd/libqpgpu.so
