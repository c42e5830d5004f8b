"""Whole-proof parity through the C-ABI: GPU proof bytes == oracle proof bytes for the same (circuit, witness, salts,
pow_witness), and the oracle verifier (which accepts the reference's golden bench-data proof) accepts them.
Mirrors the reference's prover/verifier tests (wormhole/tests/src/prover/prover_tests.rs:14-45,
wormhole/tests/src/verifier/verifier_tests.rs:14-66)."""
import numpy as np
import pytest
from harness import *

pytestmark = pytest.mark.gpu


def first_diff(a, b):
    if len(a) != len(b):
        return "length %d vs %d" % (len(a), len(b))
    x = np.frombuffer(a, dtype=np.uint8); y = np.frombuffer(b, dtype=np.uint8)
    d = np.nonzero(x != y)[0]
    return None if d.size == 0 else "first differing byte at %d (of %d), %d bytes differ" % (d[0], len(a), d.size)


@pytest.fixture(scope="module")
def circuits():
    cache = {}

    def get(preset):
        if preset not in cache:
            sc = SynthCircuit(preset)
            cache[preset] = (sc, OracleCircuit(sc.desc), GpuCircuit(sc.desc))
        return cache[preset]
    yield get
    for sc, oc, gc in cache.values():
        gc.close()


# "RT"/"R": recursion-shaped circuits carrying all 14 gate types of the in-circuit verifier behind aggregate()
# (tree.rs:106-143) — parity is GPU == oracle; parity with the Rust prover is unpinned for those gates.
@pytest.mark.parametrize("preset", ["T", "TZ", "A", "B", "RT", "RT2", "R"])
def test_proof_bytes_match_oracle(circuits, preset):
    sc, oc, gc = circuits(preset)
    assert np.array_equal(gc.cap(), oc.cap())                       # preprocessed commitment
    for seed in (0x51570000, 0x51570001):
        wires, pis = sc.witness(seed)
        zk = bool(sc.desc.zero_knowledge)
        salts = sc.salts(seed) if zk else None
        want = oc.prove(wires, pis, salts)
        got = gc.prove(wires, pis, salt_seed=seed)                  # device-derived salts == qpsynth_salts(seed)
        assert len(got) == gc.proof_size == len(want)
        assert first_diff(got, want) is None
        assert oc.verify(got) is None


def test_explicit_salts_and_pow_witness(circuits):
    sc, oc, gc = circuits("TZ")
    wires, pis = sc.witness(7)
    salts = sc.salts(99)
    base = oc.prove(wires, pis, salts)
    npi = sc.desc.num_public_inputs
    off = gc.proof_size - 8 * npi - 16                                   # ... | pow_witness | u64 count | public inputs
    powo = int.from_bytes(base[off: off + 8], "little")
    got = gc.prove(wires, pis, salts=salts, pow_witness=powo)
    assert first_diff(got, base) is None
    with pytest.raises(RuntimeError, match="pow_witness"):
        gc.prove(wires, pis, salts=salts, pow_witness=(powo + 1) % P)      # almost surely invalid


def test_tampered_proof_rejected(circuits):
    sc, oc, gc = circuits("T")
    wires, pis = sc.witness(3)
    pf = bytearray(gc.prove(wires, pis))
    assert oc.verify(bytes(pf)) is None
    for pos in (len(pf) - 1, len(pf) - 8 * 16, 100, 2000):              # public inputs, a cap, an opening
        bad = bytearray(pf); bad[pos] ^= 1
        assert oc.verify(bytes(bad)) is not None


def test_batch_matches_single(circuits):
    sc, oc, gc = circuits("T")
    ws, ps = zip(*[sc.witness(100 + i) for i in range(6)])
    proofs = gc.prove_batch(list(ws), list(ps), salt_seed=5)
    for i, pf in enumerate(proofs):
        assert first_diff(pf, oc.prove(ws[i], ps[i])) is None


def test_bad_arguments(circuits):
    sc, oc, gc = circuits("T")
    wires, pis = sc.witness(1)
    L = qpgpu()
    buf = ctypes.create_string_buffer(16)
    n = ctypes.c_size_t()
    rc = L.qpgpu_prove(gc.h, wires.ctypes.data, ptr(pis), len(pis), None, buf, 16, ctypes.byref(n))
    assert rc == -5 and b"too small" in L.qpgpu_last_error()
    bad = pis.copy(); bad[0] = P
    with pytest.raises(RuntimeError, match="non-canonical"):
        gc.prove(wires, bad)


def test_non_canonical_wire_or_salt_is_rejected_and_the_handle_stays_usable(circuits):
    """include/qpgpu.h promises QPGPU_ERR_WITNESS for any input element >= p: a wire of exactly p (or p + 5) would otherwise flow
    through the lazy arithmetic and be copied raw into the query rows, giving bytes the reference cannot produce with rc 0."""
    for preset in ("T", "TZ"):
        sc, oc, gc = circuits(preset)
        wires, pis = sc.witness(11)
        zk = bool(sc.desc.zero_knowledge)
        want = oc.prove(wires, pis, sc.salts(11) if zk else None)
        for col, row, val in ((0, 0, P), (sc.W - 1, sc.n - 1, P + 5), (37, 5, 0xFFFFFFFFFFFFFFFF)):
            bad = wires.copy(); bad[col, row] = val
            with pytest.raises(RuntimeError, match="error -8.*non-canonical"):
                gc.prove(bad, pis, salt_seed=11)
            with pytest.raises(RuntimeError, match="error -8.*non-canonical"):
                gc.prove_batch([wires, bad, wires], [pis, pis, pis], salt_seed=11)
        if zk:
            salts = sc.salts(11).copy(); salts[2, 3, sc.N - 1] = P
            with pytest.raises(RuntimeError, match="error -8.*non-canonical"):
                gc.prove(wires, pis, salts=salts)
        assert gc.prove(wires, pis, salt_seed=11) == want               # the recycled workspace is clean again
        assert gc.prove_batch([wires] * 3, [pis] * 3, salt_seed=11)[0] == want


def test_zero_knowledge_needs_salts_or_a_key(circuits):
    """Prover-side randomness is an input (qp-plonky2 draws it from OsRng): a zk circuit refuses to prove without explicit salts
    or a 32-byte key; the key-derived salts are the ChaCha8 keystream the synthetic generator restates on the host."""
    sc, oc, gc = circuits("TZ")
    wires, pis = sc.witness(5)
    with pytest.raises(RuntimeError, match="error -1.*salt_key"):
        gc.prove(wires, pis, salt_key=None)
    with pytest.raises(RuntimeError, match="error -1.*salt_key"):
        gc.prove_batch([wires], [pis], salt_key=None)
    key = bytes(range(100, 132))
    got = gc.prove(wires, pis, salt_seed=77, salt_key=key)
    assert got == oc.prove(wires, pis, sc.salts_keyed(key, 77)) and oc.verify(got) is None
    assert got != gc.prove(wires, pis, salt_seed=78, salt_key=key)           # another stream, other salts
    assert got != gc.prove(wires, pis, salt_seed=77, salt_key=bytes(32))     # another key
    batch = gc.prove_batch([wires, wires], [pis, pis], salt_seed=76, salt_key=key)   # proof i uses stream 76 + i
    assert batch[1] == got
    nz, noc, ngc = circuits("T")                                             # non-zk circuits ignore both
    w, p = nz.witness(5)
    assert ngc.prove(w, p, salt_key=None) == noc.prove(w, p)


@pytest.mark.parametrize("shape", [dict(degree_bits=13, fri_arity_bits=[]), dict(degree_bits=9, fri_cap_height=12, fri_arity_bits=[])])
def test_large_handoffs_fit_the_pinned_buffer(shape):
    """The device->host hand-off buffer is sized for the largest fetch: a 2^13-coefficient final polynomial (no FRI reduction) and a
    4096-digest cap (cap height 12) both used to exceed the old fixed-size estimate."""
    sc = SynthCircuit("T", **shape); oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)
    try:
        w, p = sc.witness(3)
        got = gc.prove(w, p)
        assert first_diff(got, oc.prove(w, p)) is None and oc.verify(got) is None
    finally:
        gc.close()


def test_handoff_wait_modes_give_the_same_bytes(monkeypatch):
    """QPGPU_HANDOFF_SPIN_US=0 naps between polls from the start (what batch workers fall back to on a busy host), QPGPU_NO_PUBLISH=1
    uses copy + stream synchronise; the knobs are read when a context is created (csrc/options.h)."""
    sc = SynthCircuit("TZ"); oc = OracleCircuit(sc.desc)
    w, p = sc.witness(8); want = oc.prove(w, p, sc.salts(8))
    for env in ({"QPGPU_HANDOFF_SPIN_US": "0"}, {"QPGPU_NO_PUBLISH": "1"}, {"QPGPU_STREAMS": "2", "QPGPU_HANDOFF_SPIN_US": "5"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        gc = GpuCircuit(sc.desc)
        try:
            assert gc.prove(w, p, salt_seed=8) == want
            assert gc.prove_batch([w] * 5, [p] * 5, salt_seed=8)[0] == want
        finally:
            gc.close()
        for k in env:
            monkeypatch.delenv(k)
    GpuCircuit(sc.desc).close()                                   # back to the defaults for the tests that follow


def test_concurrent_proves_are_reentrant(circuits):
    """qpgpu_prove from several host threads on one shared circuit handle (the aggregator's rayon workers do this in the
    reference, wormhole/aggregator/src/circuits/tree.rs:100): each call gets its own stream + workspace."""
    import threading
    sc, oc, gc = circuits("T")
    wits = [sc.witness(500 + i) for i in range(8)]
    want = [oc.prove(w, p) for w, p in wits]
    got = [None] * 8
    errs = []

    def work(i):
        try:
            for _ in range(3):
                got[i] = gc.prove(wits[i][0], wits[i][1])
        except Exception as e:      # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs
    assert got == want


def test_circuit_create_destroy_cycles_and_bad_descriptors():
    sc = SynthCircuit("T")
    for _ in range(3):
        gc = GpuCircuit(sc.desc)
        w, p = sc.witness(1)
        assert len(gc.prove(w, p)) == gc.proof_size
        gc.close()
    L = qpgpu()
    ctx = ctypes.c_void_p(); assert L.qpgpu_ctx_create(0, ctypes.byref(ctx)) == 0
    assert L.qpgpu_ctx_create(99, ctypes.byref(ctypes.c_void_p())) == -1
    import copy
    bad = CircuitDesc.from_buffer_copy(sc.desc)
    bad.gates = (Gate * 6)(*[Gate(6 if i == 3 else sc.desc.gates[i].tag, 0, sc.desc.gates[i].param) for i in range(6)])   # LookupGate: not supported
    h = ctypes.c_void_p()
    assert L.qpgpu_circuit_create(ctx, ctypes.byref(bad), ctypes.byref(h)) == -1 and b"not supported" in L.qpgpu_last_error()
    bad2 = CircuitDesc.from_buffer_copy(sc.desc); bad2.fri_arity_bits[0] = 7
    assert L.qpgpu_circuit_create(ctx, ctypes.byref(bad2), ctypes.byref(h)) == -1
    L.qpgpu_ctx_destroy(ctx)


def test_device_resident_witness_and_caller_stream(circuits):
    import torch
    sc, oc, gc = circuits("TZ")
    w, p = sc.witness(9)
    wd = torch.from_numpy(w.view(np.int64)).cuda()
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    got = gc.prove(None, p, salt_seed=4, wires_dev_ptr=wd.data_ptr(), stream=st.cuda_stream)
    assert got == oc.prove(w, p, sc.salts(4))


from test_oracle_prover import SHAPES


@pytest.mark.parametrize("shape", SHAPES, ids=[str(i) for i in range(len(SHAPES))])
def test_varied_circuit_shapes_match_oracle(shape):
    """FRI arities 1..4 in mixed schedules, cap heights 0..4, no-reduction FRI, zk, tiny and odd sizes."""
    sc = SynthCircuit("T", **shape); oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)
    try:
        assert np.array_equal(gc.cap(), oc.cap())
        for seed in (1, 2):
            w, p = sc.witness(seed)
            want = oc.prove(w, p, sc.salts(seed) if sc.desc.zero_knowledge else None)
            got = gc.prove(w, p, salt_seed=seed)
            assert first_diff(got, want) is None
    finally:
        gc.close()


@pytest.mark.parametrize("mode", ["0", "1"])
def test_split_quotient_path_matches_oracle(monkeypatch, mode):
    """The split quotient (gate terms / permutation terms on a side stream + alpha combine; default for single proofs) and the
    fused kernel (QPGPU_SPLIT_QUOTIENT=0; always used by stream-pipelined batches) give the same bytes."""
    monkeypatch.setenv("QPGPU_SPLIT_QUOTIENT", mode)
    for preset in ("T", "TZ"):
        sc = SynthCircuit(preset); oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)     # env is read at circuit creation
        try:
            w, p = sc.witness(21)
            want = oc.prove(w, p, sc.salts(21) if sc.desc.zero_knowledge else None)
            assert first_diff(gc.prove(w, p, salt_seed=21), want) is None
            assert first_diff(gc.prove_batch([w, w], [p, p], salt_seed=21)[0], want) is None
        finally:
            gc.close()


@pytest.mark.parametrize("form", ["0", "1"])
def test_both_forms_of_the_leaf_hash_partial_rounds_give_the_oracle_bytes(monkeypatch, form):
    """QPGPU_POSEIDON_FAST=0 / 1 force the literal / the fast form of the permutation's 22 partial rounds in every leaf-hash launch (the default
    picks by tree size and batch mode): same proof bytes, single and batched, zk and not — trees from 2^9 to 2^17 leaves."""
    monkeypatch.setenv("QPGPU_POSEIDON_FAST", form)
    for preset in ("T", "TZ", "A", "B"):
        sc = SynthCircuit(preset); oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)     # env is read at circuit creation
        try:
            w, p = sc.witness(33)
            want = oc.prove(w, p, sc.salts(33) if sc.desc.zero_knowledge else None)
            assert first_diff(gc.prove(w, p, salt_seed=33), want) is None
            assert first_diff(gc.prove_batch([w, w], [p, p], salt_seed=33)[0], want) is None
        finally:
            gc.close()


def test_recursion_gate_violation_is_caught_downstream(circuits):
    """A broken recursion-gate row still yields bytes (as upstream), identical to the oracle's, and the verifier rejects them."""
    sc, oc, gc = circuits("RT")
    w, p = sc.witness(9)
    pr = sc.params
    row = 3 + pr.n_poseidon + pr.n_basesum + pr.n_arith + pr.n_constant + 2     # an ArithmeticExtension row
    w[6, row] = (int(w[6, row]) + 1) % P
    got = gc.prove(w, p)
    assert first_diff(got, oc.prove(w, p)) is None and oc.verify(got) is not None


def test_split_quotient_is_not_used_for_recursion_gates(monkeypatch):
    monkeypatch.setenv("QPGPU_SPLIT_QUOTIENT", "1")
    sc = SynthCircuit("RT"); oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)
    try:
        w, p = sc.witness(2)
        assert first_diff(gc.prove(w, p), oc.prove(w, p)) is None
    finally:
        gc.close()


@pytest.mark.parametrize("seed", range(6))
def test_random_gate_mixes_match_oracle(seed):
    """Randomised circuit packages: degree, zk, gate-row mix (leaf and recursion kinds, either parameter set), copy count,
    FRI schedule — GPU bytes == oracle bytes, and the oracle verifier accepts."""
    import random
    rnd = random.Random(1000 + seed)
    k = rnd.choice([7, 8, 9, 10])
    n = 1 << k
    zk = rnd.random() < 0.4
    rec = rnd.random() < 0.6
    budget = n - 3 - (n // 8 if zk else 0)                       # leave Noop rows (zk blinding wants some)
    parts = [rnd.random() for _ in range(4 + (8 if rec else 0))]
    tot = sum(parts)
    rows = [max(1, int(budget * 0.8 * p / tot)) for p in parts]
    ov = dict(degree_bits=k, zero_knowledge=int(zk), n_poseidon=rows[0], n_basesum=rows[1], n_arith=rows[2], n_constant=rows[3],
              n_copies=rnd.randrange(0, 4 * n), circuit_seed=0x1234 + seed, reserved=(rnd.randrange(2) if rec else 0))
    arities = []
    left = k - 1                                                 # every FRI layer tree must keep at least cap-many (16) leaves
    while left > 0 and len(arities) < 3 and rnd.random() < 0.8:
        a = rnd.randint(1, min(4, left)); arities.append(a); left -= a
    ov["fri_arity_bits"] = arities
    sc = SynthCircuit("T", **ov)
    if rec:
        for i in range(8):
            sc.params.n_recursion[i] = rows[4 + i]
        sc = SynthCircuit(None, **{f: getattr(sc.params, f) for f in ("degree_bits", "zero_knowledge", "n_poseidon", "n_basesum", "n_arith",
                                                                        "n_constant", "n_copies", "circuit_seed", "reserved")},
                          fri_arity_bits=arities, n_recursion=list(sc.params.n_recursion))
    oc = OracleCircuit(sc.desc); gc = GpuCircuit(sc.desc)
    try:
        w, p = sc.witness(seed)
        salts = sc.salts(seed) if zk else None
        want = oc.prove(w, p, salts)
        got = gc.prove(w, p, salt_seed=seed)
        assert first_diff(got, want) is None and oc.verify(got) is None
    finally:
        gc.close()


def test_gpu_proofs_match_the_committed_self_check_digests():
    """The same regression digests as tests/test_oracle_prover.py, straight from the device path (no oracle call)."""
    import hashlib, json, os
    chk = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "selfcheck.json")))
    for key, want in sorted(chk.items()):
        preset, seed = key.split("/"); seed = int(seed)
        sc = SynthCircuit(preset); gc = GpuCircuit(sc.desc)
        try:
            w, p = sc.witness(seed)
            pf = gc.prove(w, p, salt_seed=seed)
            assert len(pf) == want["len"] and hashlib.sha256(pf).hexdigest() == want["sha256"], key
            assert hashlib.sha256(gc.cap().tobytes()).hexdigest() == want["cap_sha256"], key
        finally:
            gc.close()
