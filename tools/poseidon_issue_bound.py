#!/usr/bin/env python3
"""First-principles VALU issue bound of the one-permutation-per-lane Poseidon kernel (the body of k_leaves_colmajor / k_permute_batch).

  instructions: static gfx950 ISA of k_permute_batch (hipcc -S of csrc/merkle.hip), per loop, weighted by the trip counts of
                poseidon_permute_dev (4 full rounds, the dense product in front of the partial rounds, 21 + 1 partial rounds in the
                fast form, 4 full rounds);
  costs:        ns per wave-instruction per SIMD at saturation, measured on MI355X by tools/ubench/valu_rates2.hip
                (profiles/r02_valu_issue_rates2.txt, column wps=8 "(ns)");
  bound:        (a) in cycles: a wave64 instruction occupies its SIMD for 2 cycles if the table shows it at the full rate (v_mov_b32, v_add_u32,
                logic ops: ~1.0 ns) and 4 cycles otherwise (every multiply, carry, 64-bit and three-operand op: ~1.8-1.9 ns), so
                1024 SIMDs x 64 lanes x 2.4 GHz / (2 N_full + 4 N_half) permutations per second at the nominal clock;
                (b) in wall-clock: 1024 x 64 / sum_i(count_i x measured ns_i) (the micro-benchmark's own clock, which DVFS holds lower).

Writes profiles/r02_poseidon_issue_bound.json (read by bench.py for its valu_roofline object). Runs in the build container (no GPU)."""
import json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RATES = os.path.join(ROOT, "profiles", "r02_valu_issue_rates2.txt")
OUT = os.path.join(ROOT, "profiles", "r02_poseidon_issue_bound.json")
SIMDS, LANES = 1024, 64


def measured_costs():
    cost = {}
    for line in open(RATES):
        m = re.match(r"^(\S+)\s+(.*)$", line)
        if not m or m.group(1) == "instruction":
            continue
        ns = re.findall(r"\(\s*([0-9.]+)\)", m.group(2))
        if len(ns) == 4:
            cost[m.group(1)] = float(ns[3])
    return cost


def classify(op, cost):
    """ns per wave-instruction for an ISA mnemonic, from the measured table (nearest measured relative where not measured itself)."""
    base = re.sub(r"_e(32|64)$", "", op)[2:]                    # strip v_ and the encoding suffix
    table = {"mov_b32": "mov_b32", "add_u32": "add_u32", "sub_u32": "sub_u32", "and_b32": "and_b32", "xor_b32": "xor_b32", "or_b32": "and_b32",
             "not_b32": "and_b32", "lshrrev_b32": "lshrrev_b32", "lshlrev_b32": "lshlrev_b32", "mad_u64_u32": "mad_u64_u32",
             "lshl_add_u64": "lshl_add_u64", "lshlrev_b64": "lshlrev_b64", "cndmask_b32": "cndmask_sgpr", "add_co_u32": "add_co_sgpr",
             "addc_co_u32": "addc_co_sgpr", "sub_co_u32": "add_co_vcc", "subb_co_u32": "addc_co_sgpr", "subbrev_co_u32": "addc_co_vcc",
             "cmp_lt_u64": "cmp_lt_u32", "cmp_gt_u64": "cmp_lt_u32", "cmp_lt_u32": "cmp_lt_u32", "mov_b64": "lshl_add_u64",
             "add3_u32": "add3_u32", "mul_lo_u32": "mul_lo_u32", "mul_hi_u32": "mul_hi_u32", "lshl_or_b32": "lshl_or_b32", "and_or_b32": "and_or_b32"}
    key = table.get(base)
    if key is None or key not in cost:
        return cost["mad_u64_u32"], "unmeasured:" + base        # priced as a half-rate instruction
    return cost[key], key


def main():
    cost = measured_costs()
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "merkle.s")
        subprocess.check_call(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "--cuda-device-only", "-S",
                               os.path.join(ROOT, "zk-circuits_amd", "csrc", "merkle.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN2qp15k_permute_batchILb1EEEvPmm:"))     # the fast form of the partial rounds
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    # the three round loops of poseidon_permute_dev, in program order: full rounds 0-3 (4 trips), the fast partial rounds (21 trips; the
    # 22nd is peeled behind the loop) and full rounds 26-29 (4 trips); everything between and around them runs once
    heads = [(i, re.match(r"^(\.LBB\d+_\d+):", l).group(1)) for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:.*Loop Header", l)]
    assert len(heads) == 3, "expected the three round loops of poseidon_permute_dev"
    spans = []
    for i, lab in heads:
        back = next(j for j in range(i, len(body)) if re.search(r"s_cbranch_scc[01]\s+" + re.escape(lab) + r"\b", body[j]))
        spans.append((i, back + 1))
    names = [("full rounds 0-3", 4), ("partial rounds (fast form)", 21), ("full rounds 26-29", 4)]
    between = ["prologue (load, first constants)", "dense 11x11 product + first S-box", "last partial round + constants of round 26", "canonical form + store"]
    segs, pos = [], 0
    for k, (a, b) in enumerate(spans):
        segs.append((between[k], pos, a, 1)); segs.append((names[k][0], a, b, names[k][1])); pos = b
    segs.append((between[3], pos, len(body), 1))
    total_ns, total_valu, total_nop, detail, unmeasured = 0.0, 0, 0, [], set()
    n_full = n_half = 0
    for name, a, b, trips in segs:
        ops = [m.group(1) for l in body[a:b] for m in [re.match(r"^\s+([vs]_[a-z0-9_]+)", l)] if m]
        valu = [o for o in ops if o.startswith("v_")]
        ns = 0.0
        for o in valu:
            c, key = classify(o, cost)
            ns += c
            if c < 1.3: n_full += trips
            else: n_half += trips
            if key.startswith("unmeasured"):
                unmeasured.add(key)
        nops = sum(1 for o in ops if o == "s_nop")
        detail.append({"segment": name, "trips": trips, "valu_instr": len(valu), "s_nop": nops, "issue_ns_per_trip": round(ns, 1)})
        total_ns += ns * trips; total_valu += len(valu) * trips; total_nop += nops * trips
    bound = SIMDS * LANES / (total_ns * 1e-9)
    cycles = 2 * n_full + 4 * n_half
    CLOCK = 2.4e9
    out = {"kernel_body": "poseidon_permute_dev (k_permute_batch / k_leaves_colmajor)", "valu_instr_per_permutation": total_valu,
           "full_rate_instr": n_full, "half_rate_instr": n_half, "issue_cycles_per_wave_permutation": cycles,
           "hazard_s_nop_per_permutation": total_nop, "issue_ns_per_wave_permutation": round(total_ns, 1),
           "first_principles_G_perm_per_s": round(SIMDS * LANES * CLOCK / cycles / 1e9, 3), "clock_GHz": CLOCK / 1e9,
           "wallclock_priced_G_perm_per_s": round(bound / 1e9, 3),
           "formula": "1024 SIMDs x 64 lanes x clock / (2 N_full + 4 N_half); wallclock_priced = 1024 x 64 / sum(count_i x measured ns_i)",
           "cost_source": "profiles/r02_valu_issue_rates2.txt (tools/ubench/valu_rates2.hip, wps=8 wall-clock column)",
           "unmeasured_priced_as_mad": sorted(unmeasured), "segments": detail}
    prev = {}
    if os.path.exists(OUT):
        prev = json.load(open(OUT))
    for k in ("measured_saturated_G_perm_per_s", "measured_source"):        # kept from tools/permute_rate.py runs on the GPU box
        if k in prev:
            out[k] = prev[k]
    json.dump(out, open(OUT, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "segments"}, indent=1))


if __name__ == "__main__":
    main()
