# A/B of the leaf-hash kernels' occupancy target (run on the GPU box): rebuilds merkle.o per setting, then V, batch and single B lines
set -e
D=gpurun_out/$1; mkdir -p $D; shift
cd zk-circuits_amd/csrc
for w in "$@"; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DQPGPU_LEAF_WAVES=$w -DQPGPU_LEAF_WAVES_FUSED=$w -c merkle.hip -o merkle.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../libqpgpu.so ntt.o ntt16.o merkle.o hooks.o qpgpu.o verify.o artifacts.o plonk.o quotient_split.o fri.o
  cd ../..
  python bench.py --no-extras --preset V --steps 5 > $D/V_w$w.json 2>> $D/err.log
  python bench.py --no-cpu-baseline --no-tree --no-real-circuit > $D/W_w$w.json 2>> $D/err.log
  python - <<PY
import json
for f in ['$D/V_w$w.json','$D/W_w$w.json']:
    d=json.loads(open(f).read()); print('waves $w', f, d['value'], d['ms_per_step'], (d.get('pipelined') or {}).get('proofs_per_s'), d['roofline']['avg_launch_ms'], flush=True)
PY
  cd zk-circuits_amd/csrc
done
