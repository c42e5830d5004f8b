# Same-box A/B of the leaf-hash kernels' partial-round form (QPGPU_POSEIDON_FAST = 0 literal / 1 fast / -1 auto), alternating, two rounds.
# usage (GPU box): bash tools/ab_poseidon_form.sh <outdir-name>
set -e
D=gpurun_out/$1; mkdir -p $D
for rep in 1 2; do
  for form in 0 1 -1; do
    export QPGPU_POSEIDON_FAST=$form
    python bench.py --no-extras --preset W --steps 40 > $D/W_f${form}_$rep.json 2>> $D/err.log
    python bench.py --no-extras --preset A --steps 40 > $D/A_f${form}_$rep.json 2>> $D/err.log
    python bench.py --no-extras --preset W --batch 64 --steps 5 > $D/W64_f${form}_$rep.json 2>> $D/err.log
    python bench.py --no-extras --preset A --batch 64 --steps 5 > $D/A64_f${form}_$rep.json 2>> $D/err.log
    python bench.py --no-extras --preset V --steps 4 > $D/V_f${form}_$rep.json 2>> $D/err.log
    python - <<PY
import json
print("rep $rep form $form:", "  ".join("%s %.2f" % (k, json.loads(open("$D/%s_f${form}_$rep.json" % k).read())["value"]) for k in ("W", "A", "W64", "A64", "V")), flush=True)
PY
  done
done
