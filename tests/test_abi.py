"""The C-ABI library loads and exports every symbol include/qpgpu.h declares; error paths that need no GPU; the host-side
mirror of the reference wrappers (WormholeProver / WormholeProofAggregator) keeps the reference's error behaviour."""
import ctypes, importlib, os, re
import numpy as np
import pytest
from harness import *


def declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    return sorted(set(re.findall(r"\b(qp(?:gpu|synth|c|w|v)_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = ctypes.CDLL(os.path.join(PKG, "libqpgpu.so"))
    syms = declared_symbols("qpgpu.h")
    assert {"qpgpu_prove", "qpgpu_prove_batch", "qpgpu_circuit_create", "qpgpu_ctx_create", "qpgpu_last_error", "qpgpu_proof_size"} <= set(syms)
    for s in syms:
        assert hasattr(L, s), s
    S = ctypes.CDLL(os.path.join(PKG, "libqpsynth.so"))
    for s in declared_symbols("qpsynth.h"):
        assert hasattr(S, s), s


def test_circuit_library_exports_exactly_its_header():
    """libqpcircuit.so (host-only: circuits + witness generation) exports include/qpcircuit.h, nothing else, and links neither the GPU
    library nor the oracle."""
    import subprocess
    so = os.path.join(PKG, "libqpcircuit.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = {l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T" and not l.split()[2].startswith("_Z")}
    declared = set(declared_symbols("qpcircuit.h")) | set(declared_symbols("qpcircuit_internal.h"))
    assert set(declared_symbols("qpcircuit_internal.h")) - set(declared_symbols("qpcircuit.h")) == {"qpc_test_random_circuit", "qpc_circuit_num_hints", "qpc_mutation_sweep"}     # the fuzz hook and the hint-soundness sweep
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    assert {"qpc_circuit_build", "qpc_generate_witness", "qpw_fill_targets", "qpv_fill_targets"} <= declared
    ldd = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    assert "liboracle" not in ldd and "libqpgpu" not in ldd and "amdhip" not in ldd


def test_library_exports_nothing_undeclared():
    """The C symbols of libqpgpu.so are exactly include/qpgpu.h (the drop-in ABI) + include/qpgpu_internal.h (transcript replay,
    profiling counters, kernel test hooks) — no stray test or helper symbol rides along."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(PKG, "libqpgpu.so")], capture_output=True, text=True).stdout
    exported = {l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T" and not l.split()[2].startswith("_Z")}
    declared = set(declared_symbols("qpgpu.h")) | set(declared_symbols("qpgpu_internal.h"))
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    internal = set(declared_symbols("qpgpu_internal.h")) - set(declared_symbols("qpgpu.h"))
    assert all(s.startswith(("qpgpu_test_", "qpgpu_bench_", "qpgpu_profile_", "qpgpu_replay_", "qpgpu_last_stage", "qpgpu_handoff_fallbacks", "qpgpu_usable_host_cpus")) for s in internal), sorted(internal)


def test_product_library_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", os.path.join(PKG, "libqpgpu.so")], capture_output=True, text=True).stdout
    assert "liboracle" not in out
    for f in os.listdir(os.path.join(PKG, "csrc")):
        if f.endswith((".hip", ".h", ".cc")):
            assert "oracle/" not in open(os.path.join(PKG, "csrc", f)).read()
    for dirpath, _, files in os.walk(PKG):                      # the Python side of the product never imports the checker either
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M) and "liboracle" not in src, f


def test_null_arguments_do_not_crash():
    L = qpgpu()
    assert L.qpgpu_ctx_create(0, None) == -1
    assert L.qpgpu_proof_size(None) == 0
    assert L.qpgpu_prove(None, None, None, 0, None, None, 0, None) == -1 and b"null" in L.qpgpu_last_error()


def test_product_poseidon_table_matches_oracle_table():
    txt = open(os.path.join(PKG, "csrc", "poseidon_constants.h")).read()
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-f]{16})ULL", txt)]
    rc = np.zeros(360, dtype=np.uint64); oracle().orc_round_constants(ptr(rc))
    assert vals == [int(x) for x in rc]


def test_synth_presets_and_desc():
    for preset, k, zk, layers in (("A", 13, 0, 2), ("B", 14, 1, 3), ("T", 6, 0, 1)):
        sc = SynthCircuit(preset)
        d = sc.desc
        assert (d.degree_bits, d.zero_knowledge, d.fri_num_layers) == (k, zk, layers)
        assert (d.num_wires, d.num_routed, d.num_constants, d.num_public_inputs, d.num_gate_constraints) == (135, 80, 4, 16, 123)
        kis = [d.k_is[i] for i in range(3)]
        assert kis == [1, 14293326489335486720, (14293326489335486720 ** 2) % P]
    with pytest.raises(ValueError):
        SynthCircuit("A", degree_bits=10)            # gate rows no longer fit


def test_proof_size_from_the_descriptor_alone():
    """ffi.proof_size_of == the reference's proof sizes for the two leaf shapes (132,712 / 148,932 bytes)."""
    ffi = importlib.import_module("zk-circuits_amd.ffi")
    a, b, w = SynthCircuit("A"), SynthCircuit("B"), BuiltCircuit(QPC_WORMHOLE, False)      # owners kept alive: the descriptors point into them
    assert ffi.proof_size_of(a.desc) == 132712 and ffi.proof_size_of(b.desc) == 148932
    assert ffi.proof_size_of(w.desc) == 132712


def test_witness_generation_is_seeded():
    sc = SynthCircuit("T")
    w1, p1 = sc.witness(7); w2, p2 = sc.witness(7); w3, _ = sc.witness(8)
    assert np.array_equal(w1, w2) and np.array_equal(p1, p2) and not np.array_equal(w1, w3)
    assert int(w1.max()) < P


def test_aggregator_buffer_semantics():
    pkg = importlib.import_module("zk-circuits_amd")
    cfg = pkg.TreeAggregationConfig(2, 2)
    assert cfg.num_leaf_proofs == 4 and pkg.TreeAggregationConfig().num_leaf_proofs == 8     # tree.rs:17-20
    agg = pkg.WormholeProofAggregator(None, cfg)
    fake = pkg.ProofWithPublicInputs(bytes(8 * 16), 16)
    for _ in range(4):
        agg.push_proof(fake)
    with pytest.raises(RuntimeError, match="proof buffer is full"):                         # aggregator.rs:53-55
        agg.push_proof(fake)
    res = agg.aggregate()
    assert len(res.proofs) == 4 and res.num_real == 4 and len(res.public_inputs) == 64
    with pytest.raises(RuntimeError, match="no proofs to aggregate"):                       # aggregator.rs:75-77
        agg.aggregate()
    agg.push_proof(fake)                                                                    # buffer restarts (aggregator.rs:57-59)
    with pytest.raises(RuntimeError, match="failed to deserialize dummy proof"):            # no dummy available: never pad with a real leaf
        agg.aggregate()
    agg.push_proof(fake)
    with pytest.raises(RuntimeError, match="failed to deserialize dummy proof"):            # a dummy of the wrong size does not fit
        agg.aggregate(dummy_proof=pkg.ProofWithPublicInputs(bytes(8 * 17), 16))
    agg.push_proof(fake)
    dummy = pkg.ProofWithPublicInputs(bytes([1]) * (8 * 16), 16)
    res = agg.aggregate(dummy_proof=dummy)
    assert res.num_real == 1 and len(res.proofs) == 4 and res.proofs[1:] == [dummy] * 3     # padded with dummies (util.rs:11-29)


def test_prover_commit_is_single_use_without_gpu():
    pkg = importlib.import_module("zk-circuits_amd")
    p = pkg.WormholeProver.__new__(pkg.WormholeProver)
    p.circuit_data = None; p._targets = False; p._witness = None
    with pytest.raises(RuntimeError, match="already commited"):                             # prover/src/lib.rs:210-212
        p.commit(pkg.SyntheticInputs.test_inputs())
    with pytest.raises(RuntimeError, match="Failed to prove"):
        p.prove()


def test_parse_reference_common_and_verifier_bins():
    """the reference's serialized circuit artefacts (wormhole/bench-data/common.bin, verifier.bin) load into
    the C-ABI descriptor; the parsed shape equals the synthetic preset B (which is what pins that preset's shape)."""
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cb = open(os.path.join(G, "common.bin"), "rb").read()
    vb = open(os.path.join(G, "verifier.bin"), "rb").read()
    L = qpgpu()
    d = CircuitDesc(); gates = (Gate * 16)(); sel = (ctypes.c_uint32 * 16)(); grp = (ctypes.c_uint32 * 16)(); kis = (ctypes.c_uint64 * 128)()
    L.qpgpu_parse_common.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(CircuitDesc), ctypes.POINTER(Gate), ctypes.c_size_t,
                                     u32p, u32p, ctypes.c_size_t, u64p, ctypes.c_size_t]
    assert L.qpgpu_parse_common(cb, len(cb), ctypes.byref(d), gates, 16, sel, grp, 8, kis, 128) == 0, L.qpgpu_last_error()
    sc = SynthCircuit("B"); ref = sc.desc          # keep the owner alive: desc points into it
    for f in ("degree_bits", "num_wires", "num_routed", "num_constants", "num_challenges", "quotient_degree_factor", "num_partial_products",
              "num_public_inputs", "num_gate_constraints", "zero_knowledge", "fri_rate_bits", "fri_cap_height", "fri_num_queries",
              "fri_pow_bits", "fri_num_layers", "num_gates", "num_selector_groups"):
        assert getattr(d, f) == getattr(ref, f), f
    assert list(d.fri_arity_bits)[:3] == [4, 4, 4]
    assert [(gates[i].tag, gates[i].param) for i in range(6)] == [(9, 0), (3, 2), (12, 0), (2, 63), (0, 20), (11, 0)]
    assert list(sel)[:6] == [0, 0, 0, 0, 0, 1] and list(grp)[:4] == [0, 5, 5, 6]
    assert [kis[i] for i in range(80)] == [ref.k_is[i] for i in range(80)]
    # truncated / trailing input is an error, not a crash
    assert L.qpgpu_parse_common(cb[:-3], len(cb) - 3, ctypes.byref(d), gates, 16, sel, grp, 8, kis, 128) == -1
    assert L.qpgpu_parse_common(cb + b"x", len(cb) + 1, ctypes.byref(d), gates, 16, sel, grp, 8, kis, 128) == -1
    cap = (ctypes.c_uint64 * 64)(); dig = (ctypes.c_uint64 * 4)()
    L.qpgpu_parse_verifier_only.argtypes = [ctypes.c_char_p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p]
    assert L.qpgpu_parse_verifier_only(vb, 552, cap, 64, dig) == 4
    assert [hex(x) for x in dig] == ["0x8713e1b2b4082ea4", "0x188a85b2cad283d6", "0xd750e61c805b7185", "0xdd6c289a955012e3"]   # SURVEY App. A.6 digest pin


def test_host_transcript_permutation_fast_form_equals_literal_form():
    """The host Challenger's permutation uses derived 'fast partial round' tables (tools/gen_poseidon_fast.py); it must equal
    the literal round-by-round permutation on random and edge-case states, and the two known-answer vectors of SURVEY App. A.2."""
    import random
    S = synth(); S.qpsynth_host_permute.argtypes = [u64p, ctypes.c_int]
    rnd = random.Random(11)
    edge = [0, 1, P - 1, P - 2, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 63]
    for it in range(3000):
        st = np.array([rnd.choice(edge) if it % 5 == 0 else rnd.randrange(P) for _ in range(12)], dtype=np.uint64)
        a, b = st.copy(), st.copy()
        S.qpsynth_host_permute(ptr(a), 0); S.qpsynth_host_permute(ptr(b), 1)
        assert np.array_equal(a, b)
    z = np.zeros(12, dtype=np.uint64); S.qpsynth_host_permute(ptr(z), 0)
    assert [int(x) for x in z[:4]] == [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca]
    s = np.arange(12, dtype=np.uint64); S.qpsynth_host_permute(ptr(s), 0)
    assert [int(x) for x in s[:4]] == [0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2]


def test_device_fast_round_tables_are_the_host_tables_in_limb_form():
    """csrc/poseidon_fast_dev_constants.h (what the leaf-hash kernels read) against csrc/poseidon_fast_constants.h (the host transcript's tables,
    checked against the literal permutation above) and the round constants: every dot-product constant a appears as the 27/27/10-bit limbs of a and
    of a * 2^32 mod p with a nonzero low limb (poseidon_dev.h's dot_finish relies on it), VQ is V[t] followed by Q[t+1], RC_PAD the 360 round
    constants followed by twelve zeros, RC_HEAD rounds 0-3 followed by FIRST. A stale generated header fails here, not on the GPU box."""
    import re
    csrc = os.path.join(ROOT, "zk-circuits_amd", "csrc")
    def macros(path):
        txt = open(path).read().replace("\\\n", " ")
        return {m.group(1): [int(x.rstrip("uUlL"), 16) for x in re.findall(r"0x[0-9a-fA-F]+[uUlL]*", m.group(2))]
                for m in re.finditer(r"#define (\w+) (.*)", txt)}
    host, dev, rc = macros(os.path.join(csrc, "poseidon_fast_constants.h")), macros(os.path.join(csrc, "poseidon_fast_dev_constants.h")), macros(os.path.join(csrc, "poseidon_constants.h"))
    rcs = next(v for v in rc.values() if len(v) == 360)
    def limbs(a):
        out = []
        for v in (a, a * (1 << 32) % P):
            out += [v & 0x7FFFFFF, (v >> 27) & 0x7FFFFFF, v >> 54]
        return out
    init, what, V, Q, first = host["QP_POSEIDON_FAST_INIT"], host["QP_POSEIDON_FAST_WHAT"], host["QP_POSEIDON_FAST_V"], host["QP_POSEIDON_FAST_Q"], host["QP_POSEIDON_FAST_FIRST"]
    assert len(init) == 121 and len(what) == 242 and len(V) == 242 and len(Q) == 22 and len(first) == 12
    assert dev["QP_POSEIDON_FASTDEV_INIT"] == [l for a in init for l in limbs(a)]
    assert dev["QP_POSEIDON_FASTDEV_WHAT"] == [l for a in what for l in limbs(a)]
    for tab in ("QP_POSEIDON_FASTDEV_INIT", "QP_POSEIDON_FASTDEV_WHAT"):
        t = dev[tab]
        assert all(t[i] >= 1 for i in range(0, len(t), 3)) and all(t[i + 2] < 1024 for i in range(0, len(t), 3))
    assert dev["QP_POSEIDON_FASTDEV_VQ"] == [x for t in range(22) for x in V[11 * t:11 * t + 11] + [Q[t + 1] if t < 21 else 0]]
    assert dev["QP_POSEIDON_FASTDEV_RC_PAD"] == rcs + [0] * 12
    assert dev["QP_POSEIDON_FASTDEV_RC_HEAD"] == rcs[:48] + first


def test_salt_stream_is_chacha8():
    """csrc/salt.h: key-derived salts are the ChaCha8 keystream (DJB layout: 64-bit block counter, 64-bit nonce), eight u64 per block,
    each taken mod p by one conditional subtraction. Known answer: the all-zero key/nonce/counter block of the ChaCha8 test vectors
    (3e00ef2f 895f40d6 ...), and a pure-Python restatement on random inputs."""
    import random, struct
    S = synth()
    out = np.zeros(8, dtype=np.uint64)
    S.qpsynth_salt_block(bytes(32), 0, 0, ptr(out))
    ks = bytes.fromhex("3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42")
    want = [z - P if z >= P else z for z in struct.unpack("<8Q", ks)]
    assert [int(x) for x in out] == want

    def rotl(x, r): return ((x << r) | (x >> (32 - r))) & 0xffffffff
    def qr(x, a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 16); x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xffffffff; x[d] = rotl(x[d] ^ x[a], 8); x[c] = (x[c] + x[d]) & 0xffffffff; x[b] = rotl(x[b] ^ x[c], 7)
    def block(key, nonce, ctr):
        s = [0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + list(struct.unpack("<8I", key)) + [ctr & 0xffffffff, ctr >> 32, nonce & 0xffffffff, nonce >> 32]
        x = s[:]
        for _ in range(4):
            qr(x, 0, 4, 8, 12); qr(x, 1, 5, 9, 13); qr(x, 2, 6, 10, 14); qr(x, 3, 7, 11, 15)
            qr(x, 0, 5, 10, 15); qr(x, 1, 6, 11, 12); qr(x, 2, 7, 8, 13); qr(x, 3, 4, 9, 14)
        w = [(a + b) & 0xffffffff for a, b in zip(x, s)]
        return [(lambda z: z - P if z >= P else z)(w[2 * i] | (w[2 * i + 1] << 32)) for i in range(8)]
    rnd = random.Random(5)
    for _ in range(50):
        key = bytes(rnd.randrange(256) for _ in range(32)); nonce = rnd.randrange(1 << 64); ctr = rnd.randrange(1 << 40)
        S.qpsynth_salt_block(key, nonce, ctr, ptr(out))
        assert [int(x) for x in out] == block(key, nonce, ctr)
    # column layout: element (oracle*4 + column)*N + leaf is word (index % 8) of block index // 8
    sc = SynthCircuit("TZ")
    salts = sc.salts_keyed(bytes(range(32)), 9)
    flat = salts.reshape(-1)
    for idx in (0, 7, 8, 12345 % flat.size, flat.size - 1):
        S.qpsynth_salt_block(bytes(range(32)), 9, idx // 8, ptr(out))
        assert int(flat[idx]) == int(out[idx % 8])
    assert np.array_equal(sc.salts(9), sc.salts_keyed(TEST_SALT_KEY, 9)) and int(flat.max()) < P


def test_artefact_writers_are_byte_stable_on_the_reference_files():
    """circuit_data_tests.rs:73-93 style: the reference's common.bin / verifier.bin parsed by the product and written back give the same bytes
    (pins the writers — and through them the common data embedded in every circuit package — to reference data); packages round-trip, carry
    the recursion gates' parameters and tables, and refuse corruption."""
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cb = open(os.path.join(G, "common.bin"), "rb").read(); vb = open(os.path.join(G, "verifier.bin"), "rb").read()
    shape = ReferenceCircuitShape(cb, vb[:552])
    assert write_common(shape.desc) == cb
    assert write_verifier_only(shape.constants_sigmas_cap, 4, list(shape.desc.circuit_digest)) == vb[:552]
    assert write_verifier_only(shape.constants_sigmas_cap, 4, list(shape.desc.circuit_digest)) + write_common(shape.desc) == vb   # bench-data layout (App. B.0)
    for preset in ("T", "TZ", "RT", "RT2"):
        sc = SynthCircuit(preset)
        pk = write_package(sc.desc)
        p = CircuitPackage(pk); d = p.desc
        assert write_package(d) == pk and write_common(d) == write_common(sc.desc)
        n = 1 << d.degree_bits; S = d.num_constants + d.num_routed
        assert np.array_equal(np.ctypeslib.as_array(d.constants_sigmas, shape=(S, n)), sc.constants_sigmas())
        assert list(d.circuit_digest) == list(sc.desc.circuit_digest)
        for i in range(d.num_gates):
            a, b = d.gates[i], sc.desc.gates[i]
            assert (a.tag, a.param, a.param2, a.param3, a.table_len) == (b.tag, b.param, b.param2, b.param3, b.table_len)
            assert [a.table[k] for k in range(a.table_len)] == [b.table[k] for k in range(b.table_len)]
        for pos in (3, 40, len(pk) // 2, len(pk) - 1):
            bad = bytearray(pk); bad[pos] ^= 1
            with pytest.raises(RuntimeError, match="error -1"):
                CircuitPackage(bytes(bad))
        with pytest.raises(RuntimeError, match="error -1"):
            CircuitPackage(pk[:-9])
