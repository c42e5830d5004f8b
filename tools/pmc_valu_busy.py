#!/usr/bin/env python3
"""VALU utilisation of the main kernels from PMC counters (run on the GPU box): one rocprofv3 pass per counter set
(--pmc with --kernel-trace only). Reports per kernel: VALU instructions per wave, VALU-busy share of the kernel's GPU-active cycles."""
import argparse, csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser(); ap.add_argument("--preset", default="B"); ARGS = ap.parse_args()
def run(counters, tag):
    out = "gpurun_out/pmc_busy_%s_%s" % (ARGS.preset, tag)
    shutil.rmtree(os.path.join(ROOT, out), ignore_errors=True)          # one workload per directory
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
           "python3", "bench.py", "--preset", ARGS.preset, "--steps", "2" if ARGS.preset == "V" else "3", "--warmup", "1", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(env, QP_BENCH_SETTLE_S="0"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    if r.returncode != 0:
        print(tag, "failed:", r.stderr.decode()[-300:]); return {}
    acc = {}
    for f in glob.glob(os.path.join(ROOT, out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("qp::", "")
            a = acc.setdefault(k, {}); c = row["Counter_Name"]
            a.setdefault(c, [0, 0.0]); a[c][0] += 1; a[c][1] += float(row["Counter_Value"])
    return acc
sets = [(["SQ_INSTS_VALU", "SQ_WAVES"], "a"), (["SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"], "b"), (["GRBM_GUI_ACTIVE"], "c"), (["SQ_WAVE_CYCLES", "SQ_INST_CYCLES_VMEM"], "d")]
res = {}
for cs, tag in sets:
    for k, v in run(cs, tag).items():
        res.setdefault(k, {}).update({c: x[1] / x[0] for c, x in v.items()})
print("preset %s" % ARGS.preset)
print("%-30s %12s %10s %14s %14s %12s" % ("kernel", "VALU/wave", "waves", "ACTIVE_VALU", "GUI_ACTIVE", "VALU issue %"))
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:12]:
    iv, wv = v.get("SQ_INSTS_VALU", 0), v.get("SQ_WAVES", 1)
    av, ga = v.get("SQ_ACTIVE_INST_VALU", 0), v.get("GRBM_GUI_ACTIVE", 0)
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs; a wave64 VALU instruction is counted as 4 issue cycles of its SIMD
    busy = 100.0 * av * 4 / (1024 * ga / 8) if ga else float("nan")
    print("%-30s %12.0f %10.0f %14.3g %14.3g %12.1f" % (k[:30], iv / max(wv, 1), wv, av, ga, busy))
