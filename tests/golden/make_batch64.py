#!/usr/bin/env python3
"""Regenerates tests/golden/batch64.json: SHA-256 of the ORACLE's proofs of the 64 preset-B witnesses of BASELINE configs[2]
(witness seed = salt seed = 0x51570000 + i, the BASELINE.md §3 / SURVEY.md §8(d) convention). Regression pins of this repo's own
restatement, not reference outputs. Run from the repo root: python tests/golden/make_batch64.py  (~2 min on 8 cores)"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import harness as H

sc = H.SynthCircuit("B"); oc = H.OracleCircuit(sc.desc)
out = {"preset": "B", "seed_base": 0x51570000, "count": 64, "sha256": []}
for i in range(64):
    seed = 0x51570000 + i
    w, p = sc.witness(seed)
    pf = oc.prove(w, p, sc.salts(seed))
    assert oc.verify(pf) is None
    out["sha256"].append(hashlib.sha256(pf).hexdigest())
    print(i, out["sha256"][-1][:16], flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "batch64.json"), "w"), indent=1)
