#!/usr/bin/env python3
"""Measures HBM traffic per launch of the dominant kernel with rocprofv3 PMC counters (run on the GPU box):

    python tools/pmc_traffic.py [--preset B]

Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py`, each with --kernel-trace only, as the MI355X guide
prescribes; counters are KiB. FETCH_SIZE is only calibrated for 16 B/lane streams on gfx950 (it reads 1/2 there), so a
calibration kernel with the leaf kernel's own access pattern (8 B/lane column reads, known byte count) is profiled in the
same way and its measured/known ratio rescales the leaf kernel's counter. Writes profiles/traffic.json, which bench.py
copies into roofline.traffic.
"""
import argparse, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pmc(counter, outdir, cmd):
    import shutil
    shutil.rmtree(os.path.join(ROOT, outdir), ignore_errors=True)      # one workload per directory: rows of an earlier preset must not be averaged in
    env = dict(os.environ, TMPDIR="/tmp")
    full = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", outdir, "--"] + cmd
    subprocess.run(full, check=True, cwd=ROOT, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows


def per_kernel(rows, counter):
    acc = {}
    for r in rows:
        if r.get("Counter_Name") != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        a = acc.setdefault(name, [0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--preset", default="B"); ap.add_argument("--batch", type=int, default=1); args = ap.parse_args()
    out = {}
    calib_w, calib_n = 139, 1 << 17
    calib_cmd = ["python3", "-c", "import ctypes; L=ctypes.CDLL('zk-circuits_amd/libqpgpu.so'); L.qpgpu_test_calib_colsum.argtypes=[ctypes.c_int,ctypes.c_size_t,ctypes.c_int]; L.qpgpu_test_calib_colsum(%d,%d,5)" % (calib_w, calib_n)]
    # one workload per profiler run: no aggregation-tree / pipelined extras, so every counter row belongs to this preset's proofs
    bench_cmd = ["python3", "bench.py", "--preset", args.preset, "--steps", "3" if args.preset == "V" else "4", "--warmup", "1", "--no-extras",
                 "--no-tree", "--no-pipelined"] + (["--batch", str(args.batch)] if args.batch > 1 else [])
    res = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tag = "%s%s_%s" % (args.preset, args.batch if args.batch > 1 else "", counter)
        c = per_kernel(run_pmc(counter, "gpurun_out/pmc_calib_" + tag, calib_cmd), counter)
        b = per_kernel(run_pmc(counter, "gpurun_out/pmc_bench_" + tag, bench_cmd), counter)
        res[counter] = (c, b)
    known_fetch = calib_w * calib_n * 8 / 1024.0
    known_write = calib_n * 8 / 1024.0
    ck = [k for k in res["FETCH_SIZE"][0] if "k_calib_colsum" in k][0]
    f_scale = known_fetch / res["FETCH_SIZE"][0][ck][1]
    w_scale = known_write / res["WRITE_SIZE"][0][ck][1]
    detail = {"calibration": {"kernel": ck, "known_fetch_KiB": known_fetch, "measured_fetch_KiB": res["FETCH_SIZE"][0][ck][1],
                              "fetch_scale": f_scale, "known_write_KiB": known_write, "measured_write_KiB": res["WRITE_SIZE"][0][ck][1], "write_scale": w_scale}}
    kernels = {}
    for k, (cnt, avg) in res["FETCH_SIZE"][1].items():
        w = res["WRITE_SIZE"][1].get(k, (0, 0.0))[1]
        kernels[k] = {"launches": cnt, "fetch_bytes": avg * 1024 * f_scale, "write_bytes": w * 1024 * w_scale,
                      "raw_fetch_KiB": avg, "raw_write_KiB": w}
    leaf = [k for k in kernels if "k_leaves_colmajor" in k]
    traffic = None
    if leaf:                                   # per launch, averaged over the launches of every form of the leaf kernel (plain / fused levels)
        n = sum(kernels[k]["launches"] for k in leaf)
        traffic = sum((kernels[k]["fetch_bytes"] + kernels[k]["write_bytes"]) * kernels[k]["launches"] for k in leaf) / n
    path = os.path.join(ROOT, "profiles", "traffic.json")
    allj = json.load(open(path)) if os.path.exists(path) else {}
    key = args.preset if args.batch == 1 else "%s%d" % (args.preset, args.batch)
    allj[key] = {"merkle_leaves_colmajor": traffic, "detail": detail, "kernels": kernels}
    json.dump(allj, open(path, "w"), indent=1)
    print(json.dumps({"preset": args.preset, "merkle_leaves_colmajor_bytes_per_launch": traffic, "calibration": detail["calibration"]}, indent=1))


if __name__ == "__main__":
    main()
