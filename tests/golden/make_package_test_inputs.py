#!/usr/bin/env python3
"""Writes zk-circuits_amd/data/wormhole_test_inputs.json (the values CircuitInputs::test_inputs() is made of) from tests/golden/kats.json, which
make_golden.py extracted from the reference's test helpers (wormhole/tests/test-helpers/src/lib.rs:10-80). Values only.
tests/test_wormhole_circuit.py::test_test_helpers_hold_the_reference_fixture_values keeps the two files in step."""
import json, os
HERE = os.path.dirname(os.path.abspath(__file__))
k = json.load(open(os.path.join(HERE, "kats.json")))
d = {"source": "values of the reference's shared test inputs (wormhole/tests/test-helpers/src/lib.rs:10-80); data only",
     "secret": k["default_secret"], "transfer_count": k["default_transfer_count"], "funding_account": k["default_funding_account"],
     "funding_amount_le_bytes": k["default_funding_amount_le_bytes"], "to_account": k["default_to_account"], "root_hash": k["default_root_hash"],
     "storage_proof": k["storage_proof_nodes"], "storage_proof_indices": k["storage_proof_indices"]}
out = os.path.join(HERE, "..", "..", "zk-circuits_amd", "data", "wormhole_test_inputs.json")
json.dump(d, open(out, "w"), indent=1)
print("wrote", os.path.normpath(out))
