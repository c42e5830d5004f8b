"""BASELINE configs[2]: a batch of 64 independent wormhole-shaped proofs (preset B: zk, degree 2^14) through qpgpu_prove_batch on one
GPU — the unit the aggregator shards per GPU (wormhole/aggregator/src/circuits/tree.rs:92-103). Every one of the 64 proofs must be
byte-identical to the oracle's proof of the same (witness, salts), match the committed digest, and be accepted by the oracle verifier
(the verifier that accepts the reference's golden proof). Witness seed = salt seed = 0x51570000 + i."""
import hashlib, json, os
import numpy as np
import pytest
from harness import *

pytestmark = pytest.mark.gpu
SEED0 = 0x51570000
FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch64.json")))


def test_batch_of_64_config_b_proofs_matches_oracle_bytes():
    sc = SynthCircuit("B"); oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)
    try:
        assert FIX["seed_base"] == SEED0 and FIX["count"] == 64
        wits = [sc.witness(SEED0 + i) for i in range(64)]
        proofs = gc.prove_batch([w for w, _ in wits], [p for _, p in wits], salt_seed=SEED0)      # proof i uses salt seed SEED0 + i
        assert len(proofs) == 64 and all(len(pf) == gc.proof_size == 148932 for pf in proofs)
        assert len(set(proofs)) == 64
        for i, pf in enumerate(proofs):
            assert hashlib.sha256(pf).hexdigest() == FIX["sha256"][i], "proof %d differs from the committed digest" % i
            assert oc.verify(pf) is None, "proof %d rejected" % i
        for i in range(64):                                                                   # bytes, not just digests
            w, p = wits[i]
            assert proofs[i] == oc.prove(w, p, sc.salts(SEED0 + i)), "proof %d differs from the oracle's bytes" % i
        # the same batch again from device-resident witnesses (how bench.py feeds it) gives the same bytes
        import torch
        dev = [torch.from_numpy(w.view(np.int64)).cuda() for w, _ in wits[:16]]
        torch.cuda.synchronize()
        again = gc.prove_batch([t.data_ptr() for t in dev], [p for _, p in wits[:16]], salt_seed=SEED0, wires_on_device=True)
        assert again == proofs[:16]
    finally:
        gc.close()
