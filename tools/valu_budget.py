#!/usr/bin/env python3
"""VALU instruction budget per proof and per kernel (run on the GPU box): one rocprofv3 --pmc SQ_INSTS_VALU pass (with
--kernel-trace only) over `bench.py`, summed per kernel and divided by the number of proofs. Tuning aid for the batch mode,
which is bound by VALU issue: 1024 SIMDs x 2.4 GHz / ~4.1 cycles per instruction = 0.6e12 wave-instructions/s."""
import csv, glob, os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = 4
out = "gpurun_out/pmc_valu"
env = dict(os.environ, TMPDIR="/tmp", QP_BENCH_SETTLE_S="0", QPGPU_SPLIT_QUOTIENT=os.environ.get("QPGPU_SPLIT_QUOTIENT", "0"))      # fused quotient: the batch mode's kernel set
import shutil; shutil.rmtree(os.path.join(ROOT, out), ignore_errors=True)
cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "--kernel-trace", "--output-format", "csv", "-d", out, "--",
       "python3", "bench.py", "--steps", str(steps), "--warmup", "1", "--no-extras"] + sys.argv[1:]
subprocess.run(cmd, check=True, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
acc = {}
for f in glob.glob(os.path.join(ROOT, out, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "SQ_INSTS_VALU":
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("qp::", "")
        a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
leaf = sum(v[0] for k, v in acc.items() if "k_leaves_colmajor" in k)
nproofs = leaf / 3.0                              # three leaf-hash launches per proof (wires, Z, quotient)
tot = sum(v[1] for v in acc.values())
print("proofs: %d, VALU wave-instructions per proof: %.1f M  (=> %.2f ms at 0.6e12/s)" % (nproofs, tot / nproofs / 1e6, tot / nproofs / 0.6e12 * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:18]:
    print("  %-34s launches/proof %5.1f  %8.2f M instr/proof  %5.1f %%" % (k[:34], v[0] / nproofs, v[1] / nproofs / 1e6, 100 * v[1] / tot))
