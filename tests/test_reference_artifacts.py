"""Byte-exact checks against artefacts only a machine with the Rust toolchain can emit: for each circuit NAME in
tests/golden/rust/ the files NAME.{common,verifier,constants_sigmas,witness,proof}.bin (+ NAME.salts.bin for zk circuits).
Skipped while the directory is empty — the reference ships none of them (wormhole/generated-bins/ holds only .gitkeep). When they arrive:
  * CPU: the oracle verifier must accept NAME.proof.bin with NAME.common.bin / NAME.verifier.bin — for the `agg` circuit this pins the
    recursion-gate layouts of SURVEY.md App. D, which nothing on disk pins today;
  * GPU: the device's constants/sigmas cap == the cap in NAME.verifier.bin; the device proof of NAME.witness.bin with the dumped proof's
    pow_witness == NAME.proof.bin byte for byte (non-zk, or zk with NAME.salts.bin); otherwise the oracle verifier must accept it."""
import ctypes, glob, os
import numpy as np
import pytest
from harness import *
from proof_format import parse_proof

RUST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rust")
NAMES = sorted(os.path.basename(p)[:-len(".common.bin")] for p in glob.glob(os.path.join(RUST, "*.common.bin")))
needs_artifacts = pytest.mark.skipif(not NAMES, reason="no Rust-emitted artefacts under tests/golden/rust (INTEGRATION.md §6)")


def _read(name, ext):
    p = os.path.join(RUST, "%s.%s.bin" % (name, ext))
    return open(p, "rb").read() if os.path.exists(p) else None


@needs_artifacts
@pytest.mark.parametrize("name", NAMES or ["none"])
def test_oracle_verifier_accepts_the_rust_proof(name):
    cb, vb, pb = _read(name, "common"), _read(name, "verifier"), _read(name, "proof")
    L = oracle()
    vo = vb[:len(vb) - len(cb)] if vb.endswith(cb) else vb          # verifier.bin may be verifier-only || common (App. B.0)
    assert L.orc_verify_bytes(cb, len(cb), vo, len(vo), pb, len(pb)) == 0, L.orc_last_error()


@needs_artifacts
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES or ["none"])
def test_device_proof_of_the_rust_witness_matches_the_rust_proof(name):
    cb, vb, pb = _read(name, "common"), _read(name, "verifier"), _read(name, "proof")
    vo = vb[:len(vb) - len(cb)] if vb.endswith(cb) else vb
    shape = ReferenceCircuitShape(cb, vo)
    d = shape.desc
    n = 1 << d.degree_bits; S = d.num_constants + d.num_routed
    cs = np.frombuffer(_read(name, "constants_sigmas"), dtype="<u8").copy(); assert cs.size == S * n
    d.constants_sigmas = ptr(cs)
    wires = np.frombuffer(_read(name, "witness"), dtype="<u8").copy().reshape(d.num_wires, n)
    p = parse_proof(d, pb)
    gc = GpuCircuit(d)
    try:
        assert np.array_equal(gc.cap(), shape.constants_sigmas_cap)              # row a5 on the preprocessed oracle, against Rust bytes
        sb = _read(name, "salts")
        salts = np.frombuffer(sb, dtype="<u8").copy().reshape(3, 4, n << d.fri_rate_bits) if sb else None
        got = gc.prove(wires, p["public_inputs"], salts=salts, pow_witness=p["pow_witness"])
        if not d.zero_knowledge or salts is not None:
            assert got == pb, "device proof differs from the Rust prover's bytes"
        else:
            vk = OracleCircuit(d, verifier_only_cap=shape.constants_sigmas_cap)
            assert vk.verify(got) is None
    finally:
        gc.close()
