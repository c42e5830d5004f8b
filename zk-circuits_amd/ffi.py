"""ctypes binding of include/qpgpu.h, include/qpgpu_internal.h and include/qpsynth.h — the one Python binding of the product
(tests/harness.py and bench.py import it; the stub a reference maintainer would write in Rust with bindgen is in INTEGRATION.md).
Nothing here touches oracle/ (that binding lives in oracle/binding.py, test infrastructure)."""
import ctypes
import os
import threading
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
u64p = ctypes.POINTER(ctypes.c_uint64)
u32p = ctypes.POINTER(ctypes.c_uint32)
u8p = ctypes.POINTER(ctypes.c_uint8)
P = 0xFFFFFFFF00000001


def ptr(a, t=u64p):
    return a.ctypes.data_as(t)


class QpgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("qpgpu error %d: %s" % (code, msg))
        self.code = code


class Gate(ctypes.Structure):
    _fields_ = [("tag", ctypes.c_uint32), ("param2", ctypes.c_uint32), ("param", ctypes.c_uint64), ("param3", ctypes.c_uint32),
                ("table_len", ctypes.c_uint32), ("table", ctypes.POINTER(ctypes.c_uint64))]


class CircuitDesc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in (
        "degree_bits", "num_wires", "num_routed", "num_constants", "num_challenges", "quotient_degree_factor",
        "num_partial_products", "num_public_inputs", "num_gate_constraints", "zero_knowledge", "fri_rate_bits",
        "fri_cap_height", "fri_num_queries", "fri_pow_bits", "fri_num_layers")] + [
        ("fri_arity_bits", ctypes.c_uint32 * 8), ("num_gates", ctypes.c_uint32), ("num_selector_groups", ctypes.c_uint32),
        ("gates", ctypes.POINTER(Gate)), ("selector_indices", u32p), ("selector_group_ranges", u32p),
        ("k_is", u64p), ("constants_sigmas", u64p), ("circuit_digest", ctypes.c_uint64 * 4)]


class SynthParams(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in ("degree_bits", "zero_knowledge", "n_poseidon", "n_basesum", "n_arith",
                                               "n_constant", "n_copies", "fri_num_layers")] + [
        ("fri_arity_bits", ctypes.c_uint32 * 8), ("circuit_seed", ctypes.c_uint64), ("fri_cap_height", ctypes.c_uint32),
        ("fri_num_queries", ctypes.c_uint32), ("fri_pow_bits", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
        ("n_recursion", ctypes.c_uint32 * 8)]


class ProveOpts(ctypes.Structure):
    _fields_ = [("salts", u64p), ("salt_key", ctypes.POINTER(ctypes.c_uint8)), ("salt_stream", ctypes.c_uint64), ("pow_witness", ctypes.c_uint64),
                ("have_pow_witness", ctypes.c_uint32), ("wires_on_device", ctypes.c_uint32), ("stream", ctypes.c_void_p)]


# include/qpsynth.h QPSYNTH_TEST_SALT_KEY_BYTES: the fixed PUBLIC ChaCha8 key the tests and bench.py use with salt_stream = seed, so
# that the oracle side can reproduce the salts (qpsynth_salts). Production callers pass 32 bytes from the OS RNG instead.
TEST_SALT_KEY = bytes([0x51, 0x50, 0x2d, 0x74, 0x65, 0x73, 0x74, 0x2d, 0x73, 0x61, 0x6c, 0x74, 0x2d, 0x6b, 0x65, 0x79]) + bytes(range(16))


_KEY_BUFS = {}
_tls = threading.local()


def set_salt_key(opts, key):
    """Points opts.salt_key at a 32-byte key. The ctypes buffer is cached per key value (it must outlive the call anyway, and building
    one per proof costs microseconds inside timed loops); returns it."""
    key = bytes(key)
    buf = _KEY_BUFS.get(key)
    if buf is None:
        if len(key) != 32:
            raise ValueError("salt_key must be 32 bytes")
        if len(_KEY_BUFS) > 64:
            _KEY_BUFS.clear()
        raw = (ctypes.c_uint8 * 32).from_buffer_copy(key)
        buf = _KEY_BUFS[key] = (raw, ctypes.cast(raw, ctypes.POINTER(ctypes.c_uint8)))
    opts.salt_key = buf[1]
    return buf


_lib = None
_synth = None


def lib():
    """libqpgpu.so (HIP kernels + C-ABI). There is no fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        so = os.path.join(_HERE, "libqpgpu.so")
        if not os.path.exists(so):
            raise ImportError("zk-circuits_amd: %s not built — run `python -c \"import __graft_entry__ as g; g.build()\"` "
                              "(hipcc --offload-arch=gfx950)" % so)
        # the host's choice, made before the HIP runtime initialises: eight hardware queues, so that qpgpu_prove_batch's eight worker
        # streams overlap (ROCm's default of four serialises them; the library itself never touches the environment)
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
        L = ctypes.CDLL(so)
        # ---- include/qpgpu.h
        L.qpgpu_last_error.restype = ctypes.c_char_p
        L.qpgpu_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
        L.qpgpu_ctx_destroy.argtypes = [ctypes.c_void_p]
        L.qpgpu_circuit_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(CircuitDesc), ctypes.POINTER(ctypes.c_void_p)]
        L.qpgpu_circuit_destroy.argtypes = [ctypes.c_void_p]
        L.qpgpu_circuit_constants_sigmas_cap.argtypes = [ctypes.c_void_p, u64p]
        L.qpgpu_proof_size.restype = ctypes.c_size_t
        L.qpgpu_proof_size.argtypes = [ctypes.c_void_p]
        L.qpgpu_prove.argtypes = [ctypes.c_void_p, ctypes.c_void_p, u64p, ctypes.c_size_t, ctypes.POINTER(ProveOpts),
                                  ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpgpu_prove_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(u64p), ctypes.c_size_t, ctypes.POINTER(ProveOpts), ctypes.c_char_p,
                                        ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpgpu_parse_common.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(CircuitDesc), ctypes.POINTER(Gate), ctypes.c_size_t,
                                         u32p, u32p, ctypes.c_size_t, u64p, ctypes.c_size_t]
        L.qpgpu_parse_verifier_only.argtypes = [ctypes.c_char_p, ctypes.c_size_t, u64p, ctypes.c_size_t, u64p]
        L.qpgpu_write_common.argtypes = [ctypes.POINTER(CircuitDesc), ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpgpu_write_verifier_only.argtypes = [u64p, ctypes.c_uint32, u64p, ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpgpu_package_write.argtypes = [ctypes.POINTER(CircuitDesc), ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpgpu_package_read.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        L.qpgpu_package_desc.restype = ctypes.POINTER(CircuitDesc)
        L.qpgpu_package_desc.argtypes = [ctypes.c_void_p]
        L.qpgpu_package_free.argtypes = [ctypes.c_void_p]
        L.qpgpu_verifier_create.argtypes = [ctypes.c_void_p, ctypes.POINTER(CircuitDesc), u64p, ctypes.POINTER(ctypes.c_void_p)]
        L.qpgpu_verifier_destroy.argtypes = [ctypes.c_void_p]
        L.qpgpu_verifier_proof_size.restype = ctypes.c_size_t
        L.qpgpu_verifier_proof_size.argtypes = [ctypes.c_void_p]
        L.qpgpu_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32)]
        L.qpgpu_host_alloc.argtypes = [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
        L.qpgpu_host_free.argtypes = [ctypes.c_void_p]
        L.qpgpu_host_free.restype = None
        L.qpgpu_comm_unique_id.argtypes = [ctypes.c_char_p]
        L.qpgpu_comm_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        L.qpgpu_comm_destroy.argtypes = [ctypes.c_void_p]
        L.qpgpu_comm_destroy.restype = None
        L.qpgpu_comm_rank.argtypes = L.qpgpu_comm_world.argtypes = [ctypes.c_void_p]
        L.qpgpu_gather_proofs.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_char_p]
        # ---- include/qpgpu_internal.h
        L.qpgpu_replay_transcript.argtypes = [ctypes.POINTER(CircuitDesc), ctypes.c_char_p, ctypes.c_size_t, u64p, ctypes.c_size_t]
        L.qpgpu_test_verify_host.argtypes = [ctypes.POINTER(CircuitDesc), ctypes.c_char_p, ctypes.c_size_t]
        L.qpgpu_profile_get.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        L.qpgpu_profile_name.restype = ctypes.c_char_p
        L.qpgpu_last_stage_ms.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
        L.qpgpu_handoff_fallbacks.restype = ctypes.c_uint64
        L.qpgpu_test_permute.argtypes = [u64p, ctypes.c_size_t]
        L.qpgpu_test_permute_form.argtypes = [u64p, ctypes.c_size_t, ctypes.c_int]
        L.qpgpu_test_ntt.argtypes = [u64p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.qpgpu_test_lde.argtypes = [u64p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u64p]
        L.qpgpu_test_merkle.argtypes = [u64p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, u64p]
        L.qpgpu_test_intt_natural.argtypes = [u64p, ctypes.c_int, ctypes.c_int, u64p]
        L.qpgpu_test_bitrev.argtypes = [u64p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, u64p]
        L.qpgpu_test_leaf_hash.argtypes = [u64p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, u64p]
        L.qpgpu_test_two_to_one.argtypes = [u64p, ctypes.c_size_t, ctypes.c_int, u64p]
        L.qpgpu_test_fri_leaf_hash.argtypes = [u64p, u64p, ctypes.c_int, ctypes.c_size_t, ctypes.c_int, u64p]
        L.qpgpu_test_calib_colsum.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_int]
        L.qpgpu_test_host_permute_ns.restype = ctypes.c_double
        L.qpgpu_test_host_permute_ns.argtypes = [ctypes.c_int]
        L.qpgpu_test_sync_roundtrip_us.restype = L.qpgpu_test_sync_roundtrip_us2.restype = ctypes.c_double
        L.qpgpu_test_sync_roundtrip_us.argtypes = [ctypes.c_int]
        L.qpgpu_test_sync_roundtrip_us2.argtypes = [ctypes.c_int, ctypes.c_int]
        L.qpgpu_bench_permute.restype = L.qpgpu_bench_lde.restype = ctypes.c_double
        L.qpgpu_bench_permute.argtypes = [ctypes.c_size_t, ctypes.c_int]
        L.qpgpu_bench_permute_form.restype = ctypes.c_double
        L.qpgpu_bench_permute_form.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int]
        L.qpgpu_bench_lde.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        _lib = L
    return _lib


def synth_lib():
    """libqpsynth.so — host-only synthetic circuit/witness generator (include/qpsynth.h)."""
    global _synth
    if _synth is None:
        so = os.path.join(_HERE, "libqpsynth.so")
        if not os.path.exists(so):
            raise ImportError("zk-circuits_amd: %s not built — run __graft_entry__.build()" % so)
        L = ctypes.CDLL(so)
        L.qpsynth_last_error.restype = ctypes.c_char_p
        L.qpsynth_preset.argtypes = [ctypes.c_char_p, ctypes.POINTER(SynthParams)]
        L.qpsynth_circuit_create.argtypes = [ctypes.POINTER(SynthParams), ctypes.POINTER(ctypes.c_void_p)]
        L.qpsynth_circuit_destroy.argtypes = [ctypes.c_void_p]
        L.qpsynth_circuit_desc.restype = ctypes.POINTER(CircuitDesc)
        L.qpsynth_circuit_desc.argtypes = [ctypes.c_void_p]
        L.qpsynth_witness.argtypes = [ctypes.c_void_p, ctypes.c_uint64, u64p, u64p]
        L.qpsynth_salts.argtypes = [ctypes.c_void_p, ctypes.c_uint64, u64p]
        L.qpsynth_salts_keyed.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_uint64, u64p]
        L.qpsynth_salt_block.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, u64p]
        _synth = L
    return _synth


class WormholeInputsC(ctypes.Structure):
    """qpw_circuit_inputs (include/qpcircuit.h)."""
    _fields_ = [("funding_amount_le", ctypes.c_uint8 * 16), ("nullifier", ctypes.c_uint8 * 32), ("root_hash", ctypes.c_uint8 * 32),
                ("exit_account", ctypes.c_uint8 * 32), ("secret", ctypes.c_uint8 * 32), ("transfer_count", ctypes.c_uint64),
                ("funding_account", ctypes.c_uint8 * 32), ("unspendable_account", ctypes.c_uint8 * 32),
                ("proof_len", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("proof_nodes", ctypes.POINTER(ctypes.c_char_p)), ("proof_node_lens", ctypes.POINTER(ctypes.c_uint32)),
                ("proof_indices", ctypes.POINTER(ctypes.c_uint32))]


class WormholePublicInputsC(ctypes.Structure):
    _fields_ = [("funding_amount_le", ctypes.c_uint8 * 16), ("nullifier", ctypes.c_uint8 * 32), ("root_hash", ctypes.c_uint8 * 32),
                ("exit_account", ctypes.c_uint8 * 32)]


class VoteInputsC(ctypes.Structure):
    """qpv_vote_inputs (include/qpcircuit.h)."""
    _fields_ = [("proposal_id", ctypes.c_uint64 * 4), ("merkle_root", ctypes.c_uint64 * 4), ("vote", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("nullifier", ctypes.c_uint64 * 4), ("private_key", ctypes.c_uint64 * 4), ("merkle_siblings", u64p),
                ("path_indices", ctypes.POINTER(ctypes.c_uint8)), ("num_siblings", ctypes.c_uint32), ("num_path_indices", ctypes.c_uint32),
                ("actual_merkle_depth", ctypes.c_uint32), ("reserved2", ctypes.c_uint32)]


QPGPU_ERR_WITNESS = -8
QPC_WORMHOLE, QPC_NULLIFIER, QPC_UNSPENDABLE_ACCOUNT, QPC_STORAGE_PROOF, QPC_EXIT_ACCOUNT, QPC_VOTE, QPC_AGGREGATION = range(7)
_circuit = None


def circuit_lib():
    """libqpcircuit.so — host-only: the reference's circuits + witness generation (include/qpcircuit.h)."""
    global _circuit
    if _circuit is None:
        so = os.path.join(_HERE, "libqpcircuit.so")
        if not os.path.exists(so):
            raise ImportError("zk-circuits_amd: %s not built — run __graft_entry__.build()" % so)
        L = ctypes.CDLL(so)
        L.qpc_last_error.restype = ctypes.c_char_p
        L.qpc_circuit_build.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        L.qpc_circuit_destroy.argtypes = [ctypes.c_void_p]
        L.qpc_circuit_destroy.restype = None
        L.qpc_circuit_desc.restype = ctypes.POINTER(CircuitDesc)
        L.qpc_circuit_desc.argtypes = [ctypes.c_void_p]
        L.qpc_circuit_num_inputs.restype = ctypes.c_size_t
        L.qpc_circuit_num_inputs.argtypes = [ctypes.c_void_p]
        L.qpc_circuit_row_counts.restype = None
        L.qpc_circuit_row_counts.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        L.qpc_generate_witness.argtypes = [ctypes.c_void_p, u64p, ctypes.c_size_t, ctypes.c_char_p, u64p, u64p]
        L.qpc_generate_witness_mt.argtypes = [ctypes.c_void_p, u64p, ctypes.c_size_t, ctypes.c_char_p, u64p, u64p, ctypes.c_uint32]
        L.qpc_circuit_parallel_segments.argtypes = [ctypes.c_void_p]
        L.qpw_fill_targets.argtypes = [ctypes.c_uint32, ctypes.POINTER(WormholeInputsC), u64p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpw_nullifier_from_preimage.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint64, ctypes.c_char_p]
        L.qpw_unspendable_account_from_secret.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.qpw_public_inputs_from_felts.argtypes = [u64p, ctypes.c_size_t, ctypes.POINTER(WormholePublicInputsC)]
        L.qpv_fill_targets.argtypes = [ctypes.POINTER(VoteInputsC), u64p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpc_aggregation_circuit_build.argtypes = [ctypes.POINTER(CircuitDesc), ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        L.qpc_aggregation_circuit_build_ex.argtypes = [ctypes.POINTER(CircuitDesc), ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        L.qpc_aggregation_fill_targets.argtypes = [ctypes.c_void_p, u64p, u64p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, u64p,
                                                   ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpc_test_random_circuit.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p), u64p, u64p,
                                              ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.qpc_hash_no_pad.restype = None
        L.qpc_hash_no_pad.argtypes = [u64p, ctypes.c_size_t, u64p]
        _circuit = L
    return _circuit


class CircuitError(ValueError):
    """A qpcircuit call failed; `code` is the QPGPU_ERR_* status (QPGPU_ERR_WITNESS = -8: the inputs do not satisfy the circuit)."""

    def __init__(self, code, msg):
        ValueError.__init__(self, msg)
        self.code = code


def circuit_check(rc):
    if rc != 0:
        raise CircuitError(rc, circuit_lib().qpc_last_error().decode())


def proof_size_of(d):
    """Byte length of ProofWithPublicInputs::to_bytes for circuit descriptor `d` (csrc/proof_format.h compute_layout)."""
    salt = 4 if d.zero_knowledge else 0
    nch = d.num_challenges
    S, W = d.num_constants + d.num_routed, d.num_wires
    Zp, Q = nch * (1 + d.num_partial_products), nch * d.quotient_degree_factor
    cap_bytes = (1 << d.fri_cap_height) * 32
    nop = S + W + 2 * nch + nch * d.num_partial_products + Q
    path = d.degree_bits + d.fri_rate_bits - d.fri_cap_height
    per_query = sum(8 * w + 1 + 32 * path for w in (S, W + salt, Zp + salt, Q + salt))
    lg = d.degree_bits
    for layer in range(d.fri_num_layers):
        a = d.fri_arity_bits[layer]
        per_query += (16 << a) + 1 + 32 * (lg + d.fri_rate_bits - a - d.fri_cap_height)
        lg -= a
    return 3 * cap_bytes + 16 * nop + d.fri_num_layers * cap_bytes + per_query * d.fri_num_queries + 16 * (1 << lg) + 8 + 8 + 8 * d.num_public_inputs


def hash_no_pad(felts):
    a = np.ascontiguousarray(felts, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    circuit_lib().qpc_hash_no_pad(ptr(a), len(a), ptr(out))
    return out


class BuiltCircuit:
    """One of the reference's circuits built by libqpcircuit (kind = QPC_*): descriptor for qpgpu_circuit_create + witness generation."""
    ROW_KINDS = ("noop", "constant", "public_input", "base_sum", "arithmetic", "poseidon", "rows_in_use")

    def __init__(self, kind, zero_knowledge=False):
        h = ctypes.c_void_p()
        circuit_check(circuit_lib().qpc_circuit_build(kind, 1 if zero_knowledge else 0, ctypes.byref(h)))
        self.handle, self.kind = h, kind
        self.desc = circuit_lib().qpc_circuit_desc(h).contents
        self.n = 1 << self.desc.degree_bits
        self.N = self.n << self.desc.fri_rate_bits
        self.W = self.desc.num_wires
        self.num_inputs = circuit_lib().qpc_circuit_num_inputs(h)

    @classmethod
    def random_program(cls, seed, zero_knowledge=False, num_inputs=12, num_ops=300, recursion_gates=False):
        """Test hook (include/qpcircuit_internal.h): a circuit of random builder operations; returns (circuit, input values, natively computed
        values of its public inputs)."""
        self = cls.__new__(cls)
        h = ctypes.c_void_p()
        inputs = np.zeros(num_inputs, dtype=np.uint64); expected = np.zeros(8 * num_ops + 64, dtype=np.uint64); n = ctypes.c_size_t()
        circuit_check(circuit_lib().qpc_test_random_circuit(seed, (1 if zero_knowledge else 0) | (2 if recursion_gates else 0), num_inputs, num_ops, ctypes.byref(h), ptr(inputs), ptr(expected),
                                                            len(expected), ctypes.byref(n)))
        self.handle, self.kind = h, -1
        self.desc = circuit_lib().qpc_circuit_desc(h).contents
        self.n = 1 << self.desc.degree_bits
        self.N = self.n << self.desc.fri_rate_bits
        self.W = self.desc.num_wires
        self.num_inputs = circuit_lib().qpc_circuit_num_inputs(h)
        return self, inputs, expected[:n.value]

    @classmethod
    def aggregation(cls, inner_desc, num_proofs, zero_knowledge=None):
        """The circuit of one aggregation-tree node (aggregate_chunk, tree.rs:106-143): verifies `num_proofs` proofs of the circuit
        `inner_desc` in-circuit and forwards their public inputs. `zero_knowledge`: None = as the children (the reference), or False / True."""
        self = cls.__new__(cls)
        h = ctypes.c_void_p()
        circuit_check(circuit_lib().qpc_aggregation_circuit_build_ex(ctypes.byref(inner_desc), num_proofs, 2 if zero_knowledge is None else int(bool(zero_knowledge)),
                                                                     ctypes.byref(h)))
        self.handle, self.kind = h, QPC_AGGREGATION
        self.desc = circuit_lib().qpc_circuit_desc(h).contents
        self.n = 1 << self.desc.degree_bits
        self.N = self.n << self.desc.fri_rate_bits
        self.W = self.desc.num_wires
        self.num_inputs = circuit_lib().qpc_circuit_num_inputs(h)
        self.num_proofs = num_proofs
        return self

    def aggregation_targets(self, constants_sigmas_cap, circuit_digest, proofs):
        """set_verifier_data_target + set_proof_with_pis_target: the node circuit's input values from the inner circuit's verifier data and the
        child proofs (bytes)."""
        cap = np.ascontiguousarray(constants_sigmas_cap, dtype=np.uint64)
        dig = np.ascontiguousarray(list(circuit_digest), dtype=np.uint64)
        blob = b"".join(bytes(p) for p in proofs)
        plen = len(proofs[0])
        n = ctypes.c_size_t()
        L = circuit_lib()
        circuit_check(L.qpc_aggregation_fill_targets(self.handle, ptr(cap), ptr(dig), blob, plen, plen, len(proofs), None, 0, ctypes.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        circuit_check(L.qpc_aggregation_fill_targets(self.handle, ptr(cap), ptr(dig), blob, plen, plen, len(proofs), ptr(out), len(out), ctypes.byref(n)))
        return out

    def row_counts(self):
        out = (ctypes.c_uint32 * 7)()
        circuit_lib().qpc_circuit_row_counts(self.handle, out)
        return dict(zip(self.ROW_KINDS, list(out)))

    def parallel_segments(self):
        return bool(circuit_lib().qpc_circuit_parallel_segments(self.handle))

    def witness(self, inputs, blinding_key=None, out=None, threads=0):
        """generate_partial_witness: input-target values -> (wires [W][n], public_inputs). Raises CircuitError (code QPGPU_ERR_WITNESS) when the
        inputs do not satisfy the circuit. `threads`: host threads for circuits with independent segments (0 = up to 8, 1 = in order).
        `out` = a (wires, public_inputs) pair from an earlier call to write into (saves the page faults of
        a fresh 17.7 MB matrix)."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
        if out is not None:
            wires, pis = out
            assert wires.shape == (self.W, self.n) and wires.dtype == np.uint64 and wires.flags.c_contiguous
        else:
            wires = np.empty((self.W, self.n), dtype=np.uint64)          # the library zeroes it
            pis = np.zeros(self.desc.num_public_inputs, dtype=np.uint64)
        circuit_check(circuit_lib().qpc_generate_witness_mt(self.handle, ptr(inputs), len(inputs), None if blinding_key is None else bytes(blinding_key),
                                                            ptr(wires), ptr(pis), threads))
        return wires, pis

    def constants_sigmas(self):
        d = self.desc
        return np.ctypeslib.as_array(d.constants_sigmas, shape=(d.num_constants + d.num_routed, self.n))

    def __del__(self):
        try:
            circuit_lib().qpc_circuit_destroy(self.handle)
        except Exception:
            pass


def check(rc):
    if rc != 0:
        raise QpgpuError(rc, lib().qpgpu_last_error().decode())


# short aliases used by tests/ and bench.py
qpgpu = lib
synth = synth_lib
gpu_check = check


class SynthCircuit:
    """A synthetic circuit package + witness source (include/qpsynth.h)."""

    def __init__(self, preset=None, **overrides):
        S = synth()
        self.params = SynthParams()
        if preset is not None and S.qpsynth_preset(preset.encode(), ctypes.byref(self.params)) != 0:
            raise ValueError(S.qpsynth_last_error().decode())
        for k, v in overrides.items():
            if k == "fri_arity_bits":
                for i in range(8):
                    self.params.fri_arity_bits[i] = v[i] if i < len(v) else 0
                self.params.fri_num_layers = len(v)
            elif k == "n_recursion":
                for i, a in enumerate(v):
                    self.params.n_recursion[i] = a
            else:
                setattr(self.params, k, v)
        h = ctypes.c_void_p()
        if S.qpsynth_circuit_create(ctypes.byref(self.params), ctypes.byref(h)) != 0:
            raise ValueError(S.qpsynth_last_error().decode())
        self.handle = h
        self.desc = S.qpsynth_circuit_desc(h).contents
        d = self.desc
        self.n = 1 << d.degree_bits
        self.N = self.n << d.fri_rate_bits
        self.W = d.num_wires

    def witness(self, seed):
        wires = np.zeros((self.W, self.n), dtype=np.uint64)
        pis = np.zeros(self.desc.num_public_inputs, dtype=np.uint64)
        synth().qpsynth_witness(self.handle, seed, ptr(wires), ptr(pis))
        return wires, pis

    def salts(self, salt_seed):
        out = np.zeros((3, 4, self.N), dtype=np.uint64)
        synth().qpsynth_salts(self.handle, salt_seed, ptr(out))
        return out

    def salts_keyed(self, key, stream):
        out = np.zeros((3, 4, self.N), dtype=np.uint64)
        synth().qpsynth_salts_keyed(self.handle, bytes(key), stream, ptr(out))
        return out

    def constants_sigmas(self):
        d = self.desc
        S = d.num_constants + d.num_routed
        return np.ctypeslib.as_array(d.constants_sigmas, shape=(S, self.n))

    def __del__(self):
        try:
            synth().qpsynth_circuit_destroy(self.handle)
        except Exception:
            pass


class GpuCircuit:
    """qpgpu_ctx + qpgpu_circuit: the product's resident circuit handle (include/qpgpu.h)."""

    def __init__(self, desc, device=0):
        L = qpgpu()
        self.desc = desc
        self.ctx = ctypes.c_void_p()
        gpu_check(L.qpgpu_ctx_create(device, ctypes.byref(self.ctx)))
        self.h = ctypes.c_void_p()
        gpu_check(L.qpgpu_circuit_create(self.ctx, ctypes.byref(desc), ctypes.byref(self.h)))
        self.proof_size = L.qpgpu_proof_size(self.h)

    def cap(self):
        out = np.zeros((1 << self.desc.fri_cap_height, 4), dtype=np.uint64)
        gpu_check(qpgpu().qpgpu_circuit_constants_sigmas_cap(self.h, ptr(out)))
        return out

    def prove(self, wires, pis, salts=None, salt_seed=0, pow_witness=None, wires_dev_ptr=None, stream=None, salt_key=TEST_SALT_KEY):
        """salts: explicit columns; otherwise ChaCha8(salt_key, stream = salt_seed). The default key is the public test key (so that
        qpsynth_salts(salt_seed) reproduces the salts); salt_key=None sends no key at all (a zk circuit then refuses to prove)."""
        L = qpgpu()
        o = ProveOpts()
        o.salts = ptr(salts) if salts is not None else None
        o.salt_stream = salt_seed
        keep = set_salt_key(o, salt_key) if (salt_key is not None and salts is None) else None   # noqa: F841 (kept alive until the call returns)
        o.have_pow_witness = 0 if pow_witness is None else 1
        o.pow_witness = 0 if pow_witness is None else pow_witness
        o.wires_on_device = 0 if wires_dev_ptr is None else 1
        o.stream = stream
        if wires_dev_ptr is None:
            wires = np.ascontiguousarray(wires, dtype=np.uint64)
            wp = wires.ctypes.data
        else:
            wp = wires_dev_ptr
        tl = _tls.__dict__
        buf = tl.get("buf")
        if buf is None or len(buf) < self.proof_size:               # one output buffer per calling thread, reused across proofs
            buf = tl["buf"] = ctypes.create_string_buffer(self.proof_size)
        n = ctypes.c_size_t()
        gpu_check(L.qpgpu_prove(self.h, wp, ptr(pis), len(pis), ctypes.byref(o), buf, len(buf), ctypes.byref(n)))
        return ctypes.string_at(buf, n.value)          # one copy (buf.raw[:n] makes two)

    def prove_batch(self, wires_list, pis_list, salt_seed=0, wires_on_device=False, salt_key=TEST_SALT_KEY):
        L = qpgpu()
        b = len(pis_list)
        o = ProveOpts()
        o.salt_stream = salt_seed                                   # proof i uses stream salt_seed + i
        keep = set_salt_key(o, salt_key) if salt_key is not None else None   # noqa: F841
        o.wires_on_device = 1 if wires_on_device else 0
        if wires_on_device:
            wp = (ctypes.c_void_p * b)(*wires_list)
        else:
            wires_list = [np.ascontiguousarray(w, dtype=np.uint64) for w in wires_list]
            wp = (ctypes.c_void_p * b)(*[w.ctypes.data for w in wires_list])
        pp = (u64p * b)(*[ptr(p) for p in pis_list])
        buf = ctypes.create_string_buffer(self.proof_size * b)
        lens = (ctypes.c_size_t * b)()
        gpu_check(L.qpgpu_prove_batch(self.h, b, wp, pp, len(pis_list[0]), ctypes.byref(o), buf, self.proof_size, lens))
        return [ctypes.string_at(ctypes.addressof(buf) + i * self.proof_size, self.proof_size) for i in range(b)]

    def stage_ms(self):
        out = (ctypes.c_double * 16)()
        k = qpgpu().qpgpu_last_stage_ms(out, 16)
        return list(out[:k])

    def close(self):
        L = qpgpu()
        if self.h:
            L.qpgpu_circuit_destroy(self.h); self.h = None
        if self.ctx:
            L.qpgpu_ctx_destroy(self.ctx); self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _sized_write(fn, *args):
    """Calls a qpgpu_*_write function twice: size query, then into a buffer of that size."""
    n = ctypes.c_size_t()
    gpu_check(fn(*args, None, 0, ctypes.byref(n)))
    buf = ctypes.create_string_buffer(n.value)
    gpu_check(fn(*args, buf, n.value, ctypes.byref(n)))
    return buf.raw[:n.value]


def write_common(desc):
    """CommonCircuitData::to_bytes() of a descriptor (common.bin)."""
    return _sized_write(qpgpu().qpgpu_write_common, ctypes.byref(desc))


def write_verifier_only(cap, cap_height, digest):
    """VerifierOnlyCircuitData::to_bytes(): cap height, constants/sigmas cap, circuit digest (verifier.bin as circuit-builder writes it)."""
    cap = np.ascontiguousarray(cap, dtype=np.uint64); dg = np.ascontiguousarray(list(digest), dtype=np.uint64)
    return _sized_write(qpgpu().qpgpu_write_verifier_only, ptr(cap), cap_height, ptr(dg))


def write_package(desc):
    """The circuit package of a descriptor that carries constants_sigmas (this backend's stand-in for prover.bin)."""
    return _sized_write(qpgpu().qpgpu_package_write, ctypes.byref(desc))


class CircuitPackage:
    """A parsed circuit package: `desc` is ready for GpuCircuit / qpgpu_circuit_create and points into the package."""

    def __init__(self, data):
        self.h = ctypes.c_void_p()
        gpu_check(qpgpu().qpgpu_package_read(bytes(data), len(data), ctypes.byref(self.h)))
        self.desc = qpgpu().qpgpu_package_desc(self.h).contents

    def __del__(self):
        try:
            if self.h:
                qpgpu().qpgpu_package_free(self.h); self.h = None
        except Exception:
            pass


class _PinnedBlock:
    """Owner of one qpgpu_host_alloc block, exposed to numpy through the array interface: the arrays made from it (and their views) hold a
    reference to this object as their base, so the block is freed when the last of them goes away — no registry, nothing to leak."""

    def __init__(self, ptr_, shape):
        self.ptr = ptr_
        self.__array_interface__ = {"data": (ptr_, False), "shape": tuple(shape), "typestr": "<u8", "version": 3}

    def __del__(self):
        try:
            if self.ptr:
                lib().qpgpu_host_free(self.ptr); self.ptr = None
        except Exception:
            pass


def witness_buffer(num_wires, n, num_public_inputs, pinned=True):
    """A reusable (wires [num_wires][n], public_inputs) pair for BuiltCircuit.witness(out=...). With `pinned` the matrix lives in page-locked
    host memory (qpgpu_host_alloc), so qpgpu_prove(_batch) uploads it with one DMA transfer instead of staging it through bounce buffers on
    the proving thread; if page-locked memory cannot be had (no device in this process) it is ordinary memory. The memory goes back when the
    array (and every view of it) is garbage."""
    pis = np.zeros(num_public_inputs, dtype=np.uint64)
    if pinned:
        try:
            p = ctypes.c_void_p()
            if lib().qpgpu_host_alloc(num_wires * n * 8, ctypes.byref(p)) == 0 and p.value:
                wires = np.asarray(_PinnedBlock(p.value, (num_wires, n)))
                assert wires.ctypes.data == p.value and wires.flags.c_contiguous and wires.dtype == np.uint64
                return wires, pis
        except Exception:
            pass
    return np.empty((num_wires, n), dtype=np.uint64), pis


def release_witness_buffer(buf):
    """Kept for callers that release explicitly: dropping the arrays is what frees a page-locked block."""


COMM_ID_BYTES = 128          # include/qpgpu.h QPGPU_COMM_ID_BYTES


class ProofComm:
    """qpgpu_comm (include/qpgpu.h): the native RCCL communicator of this rank for the all-gather of proof blobs. `unique_id` (rank 0
    makes it with ProofComm.unique_id()) reaches the other ranks out of band; creation is collective."""

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(COMM_ID_BYTES)
        gpu_check(qpgpu().qpgpu_comm_unique_id(buf))
        return buf.raw

    def __init__(self, rank, world, unique_id, device=0):
        L = qpgpu()
        assert len(unique_id) == COMM_ID_BYTES
        self.ctx = ctypes.c_void_p()
        gpu_check(L.qpgpu_ctx_create(device, ctypes.byref(self.ctx)))
        self.h = ctypes.c_void_p()
        rc = L.qpgpu_comm_create(self.ctx, rank, world, bytes(unique_id), ctypes.byref(self.h))
        if rc != 0:
            L.qpgpu_ctx_destroy(self.ctx); self.ctx = None
            gpu_check(rc)
        self.rank, self.world = rank, world

    def gather(self, local_proofs, proof_size):
        """Every rank contributes len(local_proofs) blobs of proof_size bytes (the same count everywhere); returns all of them in rank order."""
        k = len(local_proofs)
        if k == 0:
            return []
        flat = b"".join(bytes(p) for p in local_proofs)
        assert len(flat) == k * proof_size
        out = ctypes.create_string_buffer(k * proof_size * self.world)
        gpu_check(qpgpu().qpgpu_gather_proofs(self.h, flat, k, proof_size, out))
        raw = out.raw
        return [raw[i * proof_size:(i + 1) * proof_size] for i in range(k * self.world)]

    def close(self):
        if getattr(self, "h", None):
            qpgpu().qpgpu_comm_destroy(self.h); self.h = None
        if getattr(self, "ctx", None):
            qpgpu().qpgpu_ctx_destroy(self.ctx); self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GpuVerifier:
    """qpgpu_verifier: VerifierCircuitData::verify for batches of proofs of one circuit (include/qpgpu.h). `cap` is the
    constants/sigmas Merkle cap ([2^cap_height][4]); the descriptor needs no constants_sigmas."""

    def __init__(self, desc, cap, device=0):
        L = qpgpu()
        self.desc = desc
        self.ctx = ctypes.c_void_p()
        gpu_check(L.qpgpu_ctx_create(device, ctypes.byref(self.ctx)))
        cap = np.ascontiguousarray(cap, dtype=np.uint64)
        self.h = ctypes.c_void_p()
        rc = L.qpgpu_verifier_create(self.ctx, ctypes.byref(desc), ptr(cap), ctypes.byref(self.h))
        if rc != 0:
            L.qpgpu_ctx_destroy(self.ctx); self.ctx = None
            gpu_check(rc)
        self.proof_size = L.qpgpu_verifier_proof_size(self.h)

    def verify_batch(self, proofs):
        """Returns (results, message): results[i] is True for an accepted proof; message names the first rejection (or is None)."""
        L = qpgpu()
        k = len(proofs)
        if k == 0:
            return [], None
        ln = len(proofs[0])
        assert all(len(p) == ln for p in proofs), "one batch = proofs of one size"
        res = (ctypes.c_int32 * k)()
        rc = L.qpgpu_verify_batch(self.h, b"".join(proofs), ln, ln, k, res)
        if rc not in (0, -9):
            gpu_check(rc)
        return [r == 0 for r in res], (L.qpgpu_last_error().decode() if rc else None)

    def verify(self, proof):
        ok, msg = self.verify_batch([proof])
        return None if ok[0] else msg

    def close(self):
        L = qpgpu()
        if getattr(self, "h", None):
            L.qpgpu_verifier_destroy(self.h); self.h = None
        if getattr(self, "ctx", None):
            L.qpgpu_ctx_destroy(self.ctx); self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ReferenceCircuitShape:
    """CommonCircuitData + VerifierOnlyCircuitData of a reference artefact pair (common.bin / verifier.bin), parsed by the
    product's own loaders (qpgpu_parse_common / qpgpu_parse_verifier_only). `desc` has no constants_sigmas (those live in prover.bin)."""

    def __init__(self, common_bytes, verifier_only_bytes):
        L = qpgpu()
        self.desc = CircuitDesc()
        self._gates = (Gate * 16)(); self._sel = (ctypes.c_uint32 * 16)(); self._grp = (ctypes.c_uint32 * 16)(); self._kis = (ctypes.c_uint64 * 160)()
        gpu_check(L.qpgpu_parse_common(common_bytes, len(common_bytes), ctypes.byref(self.desc), self._gates, 16, self._sel, self._grp, 8, self._kis, 160))
        cap = (ctypes.c_uint64 * (4 << 8))(); dig = (ctypes.c_uint64 * 4)()
        ch = L.qpgpu_parse_verifier_only(verifier_only_bytes, len(verifier_only_bytes), cap, len(cap), dig)
        if ch < 0:
            raise RuntimeError(L.qpgpu_last_error().decode())
        self.constants_sigmas_cap = np.array(cap[: 4 << ch], dtype=np.uint64).reshape(1 << ch, 4)
        for i in range(4):
            self.desc.circuit_digest[i] = dig[i]


def replay_transcript(desc, proof):
    """The product's Fiat-Shamir schedule over finished proof bytes (include/qpgpu_internal.h). Returns a dict of challenges."""
    L = qpgpu()
    out = (ctypes.c_uint64 * 256)()
    k = L.qpgpu_replay_transcript(ctypes.byref(desc), proof, len(proof), out, 256)
    if k < 0:
        raise RuntimeError("qpgpu error %d: %s" % (k, L.qpgpu_last_error().decode()))
    v = [int(x) for x in out[:k]]
    nch = desc.num_challenges; o = 0
    r = {}
    for name, cnt in (("pi_hash", 4), ("betas", nch), ("gammas", nch), ("alphas", nch), ("zeta", 2), ("fri_alpha", 2),
                      ("fri_betas", 2 * desc.fri_num_layers), ("pow_response", 1), ("query_indices", desc.fri_num_queries)):
        r[name] = v[o:o + cnt]; o += cnt
    assert o == k
    return r
