"""ctypes binding of include/qpgpu.h and include/qpsynth.h (the stub a reference maintainer would write in Rust with
bindgen is shown in INTEGRATION.md)."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
u64p = ctypes.POINTER(ctypes.c_uint64)
u32p = ctypes.POINTER(ctypes.c_uint32)


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
    _fields_ = [("salts", u64p), ("salt_seed", ctypes.c_uint64), ("pow_witness", ctypes.c_uint64),
                ("have_pow_witness", ctypes.c_uint32), ("wires_on_device", ctypes.c_uint32), ("stream", ctypes.c_void_p)]


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
        L = ctypes.CDLL(so)
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
        _synth = L
    return _synth


def check(rc):
    if rc != 0:
        raise QpgpuError(rc, lib().qpgpu_last_error().decode())
