"""The product's batch verifier (include/qpgpu.h qpgpu_verify_batch, csrc/verify.hip) — the mirror of
WormholeVerifier::verify (wormhole/verifier/src/lib.rs:155-159) and its tests (wormhole/tests/src/verifier/verifier_tests.rs:14-91):
  * it accepts the reference's golden proof wormhole/bench-data/proof.bin given only common.bin + verifier.bin, with NO oracle in the loop:
    the product's transcript, vanishing identity, FRI arithmetic and device Merkle kernels agree with bytes the Rust prover wrote;
  * every public-input byte flip is rejected (verifier_tests.rs:49-66), and so is tampering anywhere in the proof body (the #[ignore]d
    fuzz of :68-91, sampled);
  * it agrees with the oracle verifier proof by proof on batches of device-made proofs with some of them corrupted."""
import os, random
import numpy as np
import pytest
from harness import *

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def blob(name):
    return open(os.path.join(G, name), "rb").read()


@pytest.fixture(scope="module")
def golden_verifier():
    shape = ReferenceCircuitShape(blob("common.bin"), blob("verifier.bin")[:552])
    v = GpuVerifier(shape.desc, shape.constants_sigmas_cap)
    yield shape, v
    v.close()


def test_accepts_the_reference_golden_proof(golden_verifier):
    shape, v = golden_verifier
    pb = blob("proof.bin")
    assert v.proof_size == len(pb) == 148932
    assert v.verify(pb) is None
    ok, msg = v.verify_batch([pb] * 5)
    assert ok == [True] * 5 and msg is None


def test_every_public_input_byte_flip_is_rejected(golden_verifier):          # verifier_tests.rs:49-66
    shape, v = golden_verifier
    pb = blob("proof.bin")
    bad = []
    for k in range(1, 16 * 8 + 1):
        b = bytearray(pb); b[-k] ^= 0x01; bad.append(bytes(b))
    ok, msg = v.verify_batch(bad)
    assert not any(ok) and "rejected" in msg


def test_tampering_anywhere_is_rejected_with_a_reason(golden_verifier):
    shape, v = golden_verifier
    pb = blob("proof.bin")
    d = shape.desc
    q0 = 3 * 512 + 4112 + 3 * 512                          # first query round (SURVEY.md App. B.3 size check)
    cases = {
        0: "vanishing",                                    # wires cap: changes beta/gamma -> the identity breaks
        512 + 9: "vanishing", 1024 + 17: "vanishing",      # Z cap (alpha), quotient cap (zeta)
        1536 + 5: "vanishing", 1536 + 4112 - 3: "vanishing",                      # openings
        1536 + 4112 + 40: "proof of work|FRI|final",                              # a FRI layer cap (changes every later challenge)
        q0 + 3: "FRI consistency", q0 + 84 * 8 + 1 + 40: "Merkle",                # a leaf value of oracle 0 (enters the combination), one of its siblings
        q0 + 84 * 8: "length",                                                    # the u8 sibling count
        q0 + 83 * 8 + 2: "FRI consistency|Merkle",                                # the last sigma column of that row
        q0 + 3800 + 5: "FRI consistency|Merkle",                                  # a FRI layer evaluation
        len(pb) - 128 - 8 - 8 - 64 + 3: "proof of work|final|FRI",                # final polynomial (observed before the PoW)
        len(pb) - 128 - 8 - 8 + 2: "proof of work",                               # pow witness
    }
    import re
    proofs, pats = [], []
    for pos, pat in cases.items():
        b = bytearray(pb); b[pos] ^= 0x04; proofs.append(bytes(b)); pats.append(pat)
    for pf, pat in zip(proofs, pats):
        msg = v.verify(pf)
        assert msg is not None and re.search(pat, msg), (pat, msg)
    ok, _ = v.verify_batch(proofs + [pb])
    assert ok == [False] * len(proofs) + [True]
    with pytest.raises(AssertionError):
        v.verify_batch([pb, pb[:-8]])
    ok, msg = v.verify_batch([pb[:-8]])
    assert ok == [False] and "length" in msg


def test_random_byte_flips_agree_with_the_oracle_verifier(golden_verifier):
    shape, v = golden_verifier
    pb = blob("proof.bin")
    cb, vo = blob("common.bin"), blob("verifier.bin")[:552]
    rnd = random.Random(2024)
    proofs = []
    for _ in range(48):
        b = bytearray(pb); pos = rnd.randrange(len(pb)); b[pos] ^= 1 << rnd.randrange(8); proofs.append(bytes(b))
    ok, _ = v.verify_batch(proofs)
    L = oracle()
    want = [L.orc_verify_bytes(cb, len(cb), vo, len(vo), p, len(p)) == 0 for p in proofs]
    assert ok == want and not any(ok)


@pytest.mark.parametrize("preset", ["T", "TZ", "A"])
def test_batches_of_device_proofs_match_the_oracle_verdicts(preset):
    sc = SynthCircuit(preset); oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)
    v = GpuVerifier(sc.desc, gc.cap())
    try:
        nb = 12 if preset != "A" else 6
        wits = [sc.witness(900 + i) for i in range(nb)]
        proofs = gc.prove_batch([w for w, _ in wits], [p for _, p in wits], salt_seed=900)
        ok, msg = v.verify_batch(proofs)
        assert ok == [True] * nb and msg is None
        rnd = random.Random(7)
        mixed = []
        for i, pf in enumerate(proofs):
            if i % 3 == 1:
                b = bytearray(pf); b[rnd.randrange(len(pf))] ^= 0x20; pf = bytes(b)
            mixed.append(pf)
        ok, msg = v.verify_batch(mixed)
        want = [oc.verify(pf) is None for pf in mixed]
        assert ok == want and ok.count(False) == len([i for i in range(nb) if i % 3 == 1])
        # a witness that violates a gate still proves (as upstream) and both verifiers reject it
        w, p = sc.witness(5); w = w.copy(); w[3, 1] = (int(w[3, 1]) + 1) % P
        broken = gc.prove(w, p, salt_seed=5)
        assert v.verify(broken) is not None and oc.verify(broken) is not None
    finally:
        v.close(); gc.close()


def test_recursion_shaped_proofs_verify_too():
    """The aggregation circuits' gate set (SURVEY.md App. D layouts, unpinned against Rust): device proofs of RT / RT2 are accepted, a
    broken recursion-gate row and tampered bytes are rejected, verdicts equal the oracle's."""
    for preset in ("RT", "RT2"):
        sc = SynthCircuit(preset); oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)
        v = GpuVerifier(sc.desc, gc.cap())
        try:
            wits = [sc.witness(40 + i) for i in range(6)]
            proofs = gc.prove_batch([w for w, _ in wits], [p for _, p in wits])
            pr = sc.params
            row = 3 + pr.n_poseidon + pr.n_basesum + pr.n_arith + pr.n_constant + 2
            w, p = sc.witness(9); w = w.copy(); w[6, row] = (int(w[6, row]) + 1) % P
            proofs.append(gc.prove(w, p))
            b = bytearray(proofs[0]); b[777] ^= 8; proofs.append(bytes(b))
            ok, msg = v.verify_batch(proofs)
            assert ok == [True] * 6 + [False, False] == [oc.verify(pf) is None for pf in proofs] and "rejected" in msg
        finally:
            v.close(); gc.close()


def test_lookup_gates_are_refused():
    sc = SynthCircuit("T")
    gc = GpuCircuit(sc.desc)
    try:
        bad = CircuitDesc.from_buffer_copy(sc.desc)
        bad.gates = (Gate * 6)(*[Gate(6 if i == 3 else sc.desc.gates[i].tag, 0, sc.desc.gates[i].param) for i in range(6)])
        with pytest.raises(RuntimeError, match="error -1.*not supported"):
            GpuVerifier(bad, gc.cap())
    finally:
        gc.close()


def test_both_forms_of_the_opening_kernel_agree(monkeypatch):
    """The verifier checks Merkle openings 16 lanes per opening for small batches and one lane per opening for large ones
    (QPGPU_VERIFY_COOP_MAX): both forms accept the reference's golden proof and a batch of device-made proofs, and reject the same tampering."""
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    rd = lambda f: open(os.path.join(g, f), "rb").read()
    common, vb, proof = rd("common.bin"), rd("verifier.bin"), rd("proof.bin")
    vo = vb[:len(vb) - len(common)] if vb.endswith(common) and len(vb) > len(common) else vb
    bad_sib = bytearray(proof); bad_sib[3 * 512 + 16 * 257 + 3 * 512 + 8 * 84 + 1 + 5] ^= 1        # a sibling digest of the first query's first opening
    bad_row = bytearray(proof); bad_row[3 * 512 + 16 * 257 + 3 * 512 + 3] ^= 1                      # a row element of it
    verdicts = {}
    for mode in ("16384", "0"):
        monkeypatch.setenv("QPGPU_VERIFY_COOP_MAX", mode)
        shape = ReferenceCircuitShape(common, vo)
        v = GpuVerifier(shape.desc, shape.constants_sigmas_cap)
        ok, reasons = v.verify_batch([proof, bytes(bad_sib), proof, bytes(bad_row)] * 3)
        verdicts[mode] = ok
        assert ok == [True, False, True, False] * 3, (mode, reasons)
        v.close()
    assert verdicts["16384"] == verdicts["0"]


from test_oracle_prover import SHAPES


@pytest.mark.parametrize("shape", SHAPES, ids=[str(i) for i in range(len(SHAPES))])
@pytest.mark.parametrize("coop", ["16384", "0"])
def test_varied_shapes_through_both_opening_kernels(monkeypatch, shape, coop):
    """FRI arities 1..4 (4-felt FRI leaves are not hashed), cap heights 0..4, a last layer whose tree is its cap (empty paths), no FRI layers,
    zk: the product verifier accepts device proofs of every shape and rejects a flipped Merkle sibling and a flipped public input, with the
    16-lane and the per-lane opening kernels."""
    monkeypatch.setenv("QPGPU_VERIFY_COOP_MAX", coop)
    sc = SynthCircuit("T", **shape); gc = GpuCircuit(sc.desc); oc = OracleCircuit(sc.desc)
    try:
        w, p = sc.witness(5)
        proof = gc.prove(w, p, salt_seed=5)
        v = GpuVerifier(sc.desc, gc.cap())
        bad_pi = bytearray(proof); bad_pi[-1] ^= 1
        d = sc.desc
        S = d.num_constants + d.num_routed; nop = S + d.num_wires + 4 + 2 * d.num_partial_products + 16
        queries_off = (3 + d.fri_num_layers) * (32 << d.fri_cap_height) + 16 * nop
        bad_row = bytearray(proof); bad_row[queries_off + 2] ^= 1                    # first element of the first query's constants/sigmas row
        ok, reasons = v.verify_batch([proof, bytes(bad_pi), bytes(bad_row), proof])
        assert ok == [True, False, False, True], reasons
        assert [oc.verify(x) is None for x in (proof, bytes(bad_pi), bytes(bad_row))] == [True, False, False]
        v.close()
    finally:
        gc.close()
