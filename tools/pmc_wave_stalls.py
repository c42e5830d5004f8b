#!/usr/bin/env python3
"""Where the waves of each kernel spend their cycles (run on the GPU box): rocprofv3 --pmc passes (with --kernel-trace only) over one bench
workload. Per kernel: share of wave lifetime parked at a wait (SQ_WAIT_ANY: s_waitcnt / barrier), stalled at issue (SQ_WAIT_INST_ANY) and
issuing (SQ_ACTIVE_INST_ANY; VALU and scalar shares of it), plus scalar-cache and instruction-cache miss rates.
  python3 tools/pmc_wave_stalls.py --preset W"""
import argparse, csv, glob, os, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser(); ap.add_argument("--preset", default="W"); ap.add_argument("--batch", default="1"); ARGS = ap.parse_args()
def run(counters, tag):
    out = "gpurun_out/pmc_stalls_%s_%s" % (ARGS.preset, tag)
    shutil.rmtree(os.path.join(ROOT, out), ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", out, "--",
           "python3", "bench.py", "--preset", ARGS.preset, "--batch", ARGS.batch, "--steps", "2" if ARGS.preset == "V" else "3", "--warmup", "1", "--no-extras"]
    r = subprocess.run(cmd, cwd=ROOT, env=dict(os.environ, TMPDIR="/tmp", QP_BENCH_SETTLE_S="0"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    if r.returncode != 0:
        print(tag, "failed:", r.stderr.decode()[-400:]); return {}
    acc = {}
    for f in glob.glob(os.path.join(ROOT, out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("qp::", "")
            a = acc.setdefault(k, {}); c = row["Counter_Name"]
            a.setdefault(c, [0, 0.0]); a[c][0] += 1; a[c][1] += float(row["Counter_Value"])
    return acc
sets = [(["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_VALU", "SQ_WAVES"], "a"),
        (["SQ_INSTS_SMEM", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD"], "b"),
        (["SQC_ICACHE_REQ", "SQC_ICACHE_MISSES", "SQC_DCACHE_REQ", "SQC_DCACHE_MISSES"], "c")]
res = {}
for cs, tag in sets:
    for k, v in run(cs, tag).items():
        res.setdefault(k, {}).update({c: x[1] / x[0] for c, x in v.items()})
print("preset %s batch %s (per launch averages; SQ cycle counters are quad-cycles summed over waves)" % (ARGS.preset, ARGS.batch))
print("%-28s %9s %8s %8s %8s %8s %8s %9s %9s %9s %9s" % ("kernel", "waves", "wait%", "stall%", "issue%", "valu%", "sca%", "smem/wave", "salu/wave", "I$ miss%", "K$ miss%"))
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
    wc = v.get("SQ_WAVE_CYCLES", 0) or float("nan"); wv = max(v.get("SQ_WAVES", 1), 1)
    pct = lambda c: 100.0 * v.get(c, 0) / wc
    im = 100.0 * v.get("SQC_ICACHE_MISSES", 0) / v["SQC_ICACHE_REQ"] if v.get("SQC_ICACHE_REQ") else float("nan")
    dm = 100.0 * v.get("SQC_DCACHE_MISSES", 0) / v["SQC_DCACHE_REQ"] if v.get("SQC_DCACHE_REQ") else float("nan")
    print("%-28s %9.0f %8.1f %8.1f %8.1f %8.1f %8.1f %9.1f %9.1f %9.2f %9.2f" % (k[:28], wv, pct("SQ_WAIT_ANY"), pct("SQ_WAIT_INST_ANY"), pct("SQ_ACTIVE_INST_ANY"),
          pct("SQ_ACTIVE_INST_VALU"), pct("SQ_ACTIVE_INST_SCA"), v.get("SQ_INSTS_SMEM", 0) / wv, v.get("SQ_INSTS_SALU", 0) / wv, im, dm))
