"""The one JSON line bench.py prints must keep the driver's contract. Checked on the committed line of the last GPU run
(profiles/r03_bench_W.json, written by `python bench.py` on MI355X: the default workload, preset W) — no GPU needed — and on bench.py's source for the parts a saved line
cannot show (single JSON line on stdout, defaults, barrier + max over ranks)."""
import json, os, re
from harness import ROOT


def test_saved_bench_line_has_the_contract_fields():
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_W.json")).read().strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["metric"] == "wormhole proofs/s" and d["unit"] == "proofs/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"].startswith("synthetic") and d["dtype"].startswith("u64") and d["n_gpus"] == 1
    assert d["config"]["preset"] == "W" and "wormhole-circuit proof" in d["config"]["workload"]         # the default workload is the wormhole circuit itself
    assert "workload" in d["config"] and "model" not in d["config"] and "configs[1]" in d["config"]["workload"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-3                      # one proof per step at N = 1
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["traffic"] is not None and r["traffic"] > 0.9 * r["alg_bytes_per_launch"]
    assert abs(r["achieved"] - r["alg_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-2
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "proofs/s" and c["sample"]
    assert d["bit_exact_vs_cpu"] is True and all(v is True or (isinstance(v, int) and v > 0) for v in d["bit_exact_batch"].values())
    # round 3: the exchange and the per-rank numbers are named, hand-off fallbacks are counted, the kernels table says where its rows come from
    assert d["rccl_ranks"] == 0 and d["gather"] is None and len(d["ms_per_proof_by_rank"]) == 1 and d["handoff_fallbacks"] == 0 and "timed steps" in d["kernels_note"]
    t = d["aggregation_tree"]
    assert t["root_verified"] and t["root_public_inputs_are_the_leaves"] and t["node_circuit_rows"] < 10000 and t["wide_tree"]["root_verified"] and t["wide_tree"]["nodes"] == 9
    g = d["reference_golden_proof_verify"]
    assert g["accepted"] and g["verify_one"]["calls"] >= 20 and g["verify_batch64"]["calls"] >= 20 and g["verify_one"]["min_ms"] <= g["verify_ms"] <= g["verify_one"]["max_ms"]
    v = d["valu_roofline"]
    assert v["source"] == "profiles/r02_poseidon_issue_bound.json" and 0.5 < v["frac"] <= 1.05 and 0.5 < v["frac_of_first_principles"] <= 1.1


def test_rocprof_summary_agrees_with_the_live_measurement():
    """profiles/r03_W_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same workload) and the HIP-event average inside bench.py must
    tell the same launch duration for the dominant kernel."""
    import csv
    d = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_W.json")).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r03_W_kernel_stats.csv"))))
    leaf = [r for r in rows if "k_leaves_colmajor" in r["Name"]]
    assert len(leaf) == 1
    prof_ms = float(leaf[0]["AverageNs"]) / 1e6
    assert abs(prof_ms - d["roofline"]["avg_launch_ms"]) / prof_ms < 0.05, (prof_ms, d["roofline"]["avg_launch_ms"])
    assert max(rows, key=lambda r: float(r["TotalDurationNs"]))["Name"] == leaf[0]["Name"]          # it is the dominant kernel


def test_bench_source_keeps_the_timing_protocol():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'add_argument\("--gpus", type=int, default=1\)', src) and "--steps" in src and "--warmup" in src
    assert "dist.barrier()" in src and "torch.cuda.synchronize()" in src and "ReduceOp.MAX" in src
    assert src.count("json_out.write") == 1                                                  # exactly one JSON line, from rank 0
    assert "from oracle" in src and src.index("from oracle") > src.index("CPU baseline")       # the checker is imported in the cpu_baseline leg only
    assert "/root/reference" not in src
