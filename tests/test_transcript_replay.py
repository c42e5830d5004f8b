"""The PRODUCT's own Fiat-Shamir transcript (csrc/qpgpu.hip `Transcript`, the code path qpgpu_prove runs between stages) replayed
over the reference's golden proof, without the oracle: every pin of SURVEY.md App. A.6 must come out (row a6).
Host-only code of libqpgpu.so, so this runs without a GPU. Fixtures: tests/golden/{common,verifier,proof}.bin
(= wormhole/bench-data/*, used by the reference at wormhole/verifier/benches/verifier.rs:16-32)."""
import os
import numpy as np
import pytest
from harness import *
from proof_format import parse_proof

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def blob(name):
    return open(os.path.join(G, name), "rb").read()


@pytest.fixture(scope="module")
def shape():
    return ReferenceCircuitShape(blob("common.bin"), blob("verifier.bin")[:552])


def test_product_transcript_reproduces_every_reference_pin(shape):
    r = replay_transcript(shape.desc, blob("proof.bin"))
    assert r["pi_hash"] == [0xa4078dae7ee85366, 0xc9ee3673f412af9b, 0x03a7b4a125718fe7, 0x0036d97887fef483]
    assert r["betas"] == [0x7ddbb1f887404dc5, 0x9e44d5322dbb81d7]
    assert r["gammas"] == [0x057855bc392710ec, 0xd05ac43c6b5580d0]
    assert r["alphas"] == [0x1820e325483ad180, 0x8a3b669f9f459949]
    assert r["zeta"] == [0xf0ad89ec16659844, 0x6228d75cf70da20c]
    assert r["fri_alpha"] == [0x39411f867d3ca4ab, 0x1e316e58733fb522]
    assert r["fri_betas"][:2] == [0x687e5086f8af3c42, 0x5e90a83117cab036]
    assert r["pow_response"] == [0x0000170b958529af] and r["pow_response"][0] >> (64 - 16) == 0      # >= proof_of_work_bits leading zeros
    assert r["query_indices"][:4] == [34707, 64718, 9922, 116687]
    assert len(r["query_indices"]) == 28 and max(r["query_indices"]) < (1 << 17)


def test_replay_rejects_wrong_length_and_shape(shape):
    pb = blob("proof.bin")
    with pytest.raises(RuntimeError, match="proof length"):
        replay_transcript(shape.desc, pb[:-8])
    with pytest.raises(RuntimeError, match="proof length"):                # the non-zk k=13 dummy proof does not fit the zk k=14 shape
        replay_transcript(shape.desc, blob("dummy_proof.bin"))


def test_replay_is_sensitive_to_every_observed_section(shape):
    pb = blob("proof.bin")
    base = replay_transcript(shape.desc, pb)
    p = parse_proof(shape.desc, pb)
    assert p["pow_witness"] == 0x7fffffff8000002e
    # flipping a byte in a section changes everything drawn after it and nothing drawn before it
    for pos, unchanged, changed in ((5, [], ["betas", "alphas", "zeta", "query_indices"]),                       # wires cap
                                    (512 + 5, ["betas", "gammas"], ["alphas", "zeta"]),                        # Z cap
                                    (1024 + 5, ["betas", "alphas"], ["zeta", "fri_alpha"]),                    # quotient cap
                                    (1536 + 77, ["zeta"], ["fri_alpha", "fri_betas"]),                         # an opening
                                    (len(pb) - 128 - 16 - 3, ["fri_betas"], ["pow_response", "query_indices"])):  # pow witness
        bad = bytearray(pb); bad[pos] ^= 0x10
        r = replay_transcript(shape.desc, bytes(bad))
        for k in unchanged:
            assert r[k] == base[k], (pos, k)
        for k in changed:
            assert r[k] != base[k], (pos, k)


def test_host_half_of_the_product_verifier_accepts_the_golden_proof(shape):
    """csrc/verify.hip's host half (transcript, vanishing identity with the leaf gate set over F_{p^2}, proof of work, FRI combination,
    barycentric layer folds, final polynomial) on the reference's proof — every check except the Merkle openings, which are the
    device half (tests/test_gpu_verifier.py). No oracle involved."""
    L = qpgpu()
    pb = blob("proof.bin")
    assert L.qpgpu_test_verify_host(ctypes.byref(shape.desc), pb, len(pb)) == 0, L.qpgpu_last_error()
    want = {5: b"vanishing", 512 + 5: b"vanishing", 1024 + 5: b"vanishing", 1536 + 77: b"vanishing", 1536 + 4112 - 1: b"vanishing",
            len(pb) - 3: b"vanishing",                                        # a public input: pi_hash feeds the PublicInput gate and the transcript
            len(pb) - 128 - 16 + 1: b"proof of work",
            1536 + 4112 + 1536 + 5: b"FRI consistency",                       # first initial-tree row of query 0
            1536 + 4112 + 1536 + 3800 + 9: b"FRI consistency"}                # first FRI layer evaluations of query 0
    for pos, reason in want.items():
        bad = bytearray(pb); bad[pos] ^= 0x02
        assert L.qpgpu_test_verify_host(ctypes.byref(shape.desc), bytes(bad), len(bad)) == -9 and reason in L.qpgpu_last_error(), (pos, L.qpgpu_last_error())
    assert L.qpgpu_test_verify_host(ctypes.byref(shape.desc), pb[:-1], len(pb) - 1) == -9
