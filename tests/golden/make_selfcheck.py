#!/usr/bin/env python3
"""Regenerates tests/golden/selfcheck.json: SHA-256 of the ORACLE's proofs of the synthetic circuits for fixed (preset, witness
seed, salt seed). These are regression fixtures of this repo's own restatement (they pin the synthetic generator, the oracle and —
through the GPU parity tests — the device path against drifting together); they are NOT reference outputs: prover-side bytes of
the Rust prover are unpinned. Run from the repo root: python tests/golden/make_selfcheck.py"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import harness as H

out = {}
for preset in ("T", "TZ", "RT", "RT2", "A"):
    sc = H.SynthCircuit(preset); oc = H.OracleCircuit(sc.desc)
    for seed in (1, 0x51570000):
        w, p = sc.witness(seed)
        pf = oc.prove(w, p, sc.salts(seed) if sc.desc.zero_knowledge else None)
        assert oc.verify(pf) is None
        out["%s/%d" % (preset, seed)] = {"len": len(pf), "sha256": hashlib.sha256(pf).hexdigest(),
                                          "cap_sha256": hashlib.sha256(oc.cap().tobytes()).hexdigest()}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "selfcheck.json"), "w"), indent=1, sort_keys=True)
print("wrote", len(out), "entries")
