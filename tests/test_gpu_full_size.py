"""Full-size checks through size-independent properties (the oracle prover would take minutes at these sizes):
config V (BASELINE configs[4], voting-shaped, 2^20 rows) must be accepted by the oracle VERIFIER; NTT round trip and
linearity at 2^20; Merkle cap stability under the fused path is covered in test_gpu_host_mirror.py."""
import numpy as np
import pytest
from harness import *

pytestmark = pytest.mark.gpu


def test_config_V_proof_is_accepted_by_the_oracle_verifier():
    sc = SynthCircuit("V")
    assert sc.desc.degree_bits == 20
    gc = GpuCircuit(sc.desc)
    try:
        w, p = sc.witness(0x51570000)
        pf = gc.prove(w, p)
        assert len(pf) == gc.proof_size
        vo = OracleCircuit(sc.desc, verifier_only_cap=gc.cap())
        assert vo.verify(pf) is None
        bad = bytearray(pf); bad[-5] ^= 1                      # a public input
        assert vo.verify(bytes(bad)) is not None
        pf2 = gc.prove(w, p)
        assert pf2 == pf                                       # deterministic, incl. the smallest-witness PoW rule
    finally:
        gc.close()


@pytest.mark.parametrize("log_n", [16, 20])
def test_ntt_round_trip_and_linearity_at_full_size(log_n):
    rng = np.random.default_rng(log_n)
    n = 1 << log_n
    a = rand_felts(rng, (2, n))
    L = qpgpu()
    f = a.copy(); gpu_check(L.qpgpu_test_ntt(ptr(f), 2, log_n, 0))           # forward DIF -> bit-reversed
    back = f.copy(); gpu_check(L.qpgpu_test_ntt(ptr(back), 2, log_n, 3))     # inverse DIT (unscaled) -> natural
    ninv = pow(n, P - 2, P)
    idx = rng.integers(0, n, 64)
    for c in range(2):
        for i in idx:
            assert (int(back[c, i]) * ninv) % P == int(a[c, i])
    # linearity: NTT(a0 + a1) == NTT(a0) + NTT(a1)
    s = np.zeros((1, n), dtype=np.uint64)
    t = a[0] + a[1]                                            # u64 wrap-around, then fix up mod p (a0 + a1 < 2p < 2^65)
    wrapped = t < a[0]
    t = np.where(wrapped, t + np.uint64(0xFFFFFFFF), t)        # 2^64 = 2^32 - 1 (mod p)
    s[0] = np.where(t >= np.uint64(P), t - np.uint64(P), t)
    fs = s.copy(); gpu_check(L.qpgpu_test_ntt(ptr(fs), 1, log_n, 0))
    for i in idx:
        assert int(fs[0, i]) == (int(f[0, i]) + int(f[1, i])) % P
    # natural-order iNTT path used for the witness: iNTT(NTT(short poly)) gives the coefficients back
    short = np.zeros((1, n), dtype=np.uint64); short[0, :4] = [3, 5, 7, 11]
    fwd = short.copy(); gpu_check(L.qpgpu_test_ntt(ptr(fwd), 1, log_n, 0))
    nat = np.ascontiguousarray(fwd[:, bitrev_indices(log_n)])
    rec = np.zeros_like(nat); gpu_check(L.qpgpu_test_intt_natural(ptr(nat), 1, log_n, ptr(rec)))
    assert np.array_equal(rec, short)
