"""WormholeProver mirror (reference: /root/reference/wormhole/prover/src/lib.rs:74-237).

Same shape as the reference wrapper: `WormholeProver.new(config)` builds the circuit and uploads it to the GPU once;
`commit(inputs)` fills the witness and may be called only once; `prove()` returns a `ProofWithPublicInputs` whose bytes are plonky2's
`to_bytes()` layout. Two circuit sources:
  * the wormhole circuit itself (`CircuitConfig.standard_recursion_config()` / `_zk_config()`): built by libqpcircuit from the reference's
    circuit definition, committed with the reference's `CircuitInputs` (inputs.py) — same statement and public inputs, this repo's own gate
    placement (include/qpcircuit.h);
  * synthetic wormhole-/voting-shaped circuits (`CircuitConfig.synthetic(preset)`, libqpsynth): the shapes pinned by the reference's
    common.bin, committed with a `SyntheticInputs` seed or an explicit witness — the bench workloads.
"""
import ctypes
import os
import numpy as np
from . import ffi

P = 0xFFFFFFFF00000001


class CircuitConfig:
    """The two configs the reference uses (wormhole/tests/src/prover/prover_tests.rs:11, wormhole/prover/benches/prover.rs:12) for the
    wormhole circuit (`preset` None, `kind` = which circuit libqpcircuit builds), or a synthetic shape of libqpsynth (`preset` set)."""

    def __init__(self, zero_knowledge, preset=None, kind=ffi.QPC_WORMHOLE):
        self.zero_knowledge = zero_knowledge
        self.preset = preset
        self.kind = kind

    @staticmethod
    def standard_recursion_config(kind=ffi.QPC_WORMHOLE):
        return CircuitConfig(False, None, kind)

    @staticmethod
    def standard_recursion_zk_config(kind=ffi.QPC_WORMHOLE):
        return CircuitConfig(True, None, kind)

    @staticmethod
    def synthetic(preset):
        """A/B: wormhole-shaped leaf circuits (k = 13 non-zk / k = 14 zk, the shape of the reference's common.bin), V: voting-shaped 2^20,
        T/TZ: tiny, R/RT/RT2: recursion-shaped."""
        return CircuitConfig(preset in ("B", "TZ"), preset)

    @staticmethod
    def test_config(zero_knowledge=False):
        return CircuitConfig(zero_knowledge, "TZ" if zero_knowledge else "T")

    def key(self):
        return (self.preset, self.kind, self.zero_knowledge)


class SyntheticInputs:
    """Inputs of a synthetic circuit: a witness seed for the generator, or an explicit full witness (wires [W][n], public_inputs)
    for any circuit (e.g. one loaded from a package)."""

    def __init__(self, seed=None, wires=None, public_inputs=None, salt_key=None):
        if (seed is None) == (wires is None):
            raise ValueError("give either a seed or an explicit witness")
        self.seed, self.wires, self.public_inputs, self.salt_key = seed, wires, public_inputs, salt_key

    @staticmethod
    def test_inputs(index=0):
        return SyntheticInputs(seed=0x51570000 + index)


class ProofWithPublicInputs:
    def __init__(self, data, num_public_inputs):
        self.data = data
        k = num_public_inputs
        self.public_inputs = list(np.frombuffer(data[len(data) - 8 * k:], dtype="<u8"))

    def to_bytes(self):
        return self.data

    def __len__(self):
        return len(self.data)


class WormholeCircuitData:
    """Resident circuit: built (libqpcircuit) or synthetic (libqpsynth) circuit + qpgpu_circuit handle; shared by provers of the same config."""
    _cache = {}

    def __init__(self, config, device=0):
        L = ffi.lib()
        self.config = config
        self._synth = self._built = None
        if config.preset is None:
            self._built = ffi.BuiltCircuit(config.kind, config.zero_knowledge)      # WormholeCircuit::new(config).build_prover() (lib.rs:190-202)
            self.desc = self._built.desc
        else:
            S = ffi.synth_lib()
            params = ffi.SynthParams()
            if S.qpsynth_preset(config.preset.encode(), ctypes.byref(params)) != 0:
                raise ValueError(S.qpsynth_last_error().decode())
            self._synth = ctypes.c_void_p()
            if S.qpsynth_circuit_create(ctypes.byref(params), ctypes.byref(self._synth)) != 0:
                raise ValueError(S.qpsynth_last_error().decode())
            self.desc = S.qpsynth_circuit_desc(self._synth).contents
        self.n = 1 << self.desc.degree_bits
        self.num_wires = self.desc.num_wires
        self.num_public_inputs = self.desc.num_public_inputs
        self._ctx = ctypes.c_void_p()
        ffi.check(L.qpgpu_ctx_create(device, ctypes.byref(self._ctx)))
        self._circuit = ctypes.c_void_p()
        ffi.check(L.qpgpu_circuit_create(self._ctx, ctypes.byref(self.desc), ctypes.byref(self._circuit)))
        self.proof_size = L.qpgpu_proof_size(self._circuit)

    @classmethod
    def from_package(cls, package_bytes, device=0):
        """Resident circuit from a circuit package (ffi.write_package / qpgpu_package_write) — the artefact path of
        WormholeProver::new_from_bytes (wormhole/prover/src/lib.rs:105-137). A package of the wormhole circuit commits CircuitInputs like a built one; any other needs an explicit witness."""
        self = cls.__new__(cls)
        self.config = None
        self._package = ffi.CircuitPackage(package_bytes)       # owns the arrays desc points into
        self._synth = self._built = None
        self.desc = self._package.desc
        self.n = 1 << self.desc.degree_bits
        self.num_wires = self.desc.num_wires
        self.num_public_inputs = self.desc.num_public_inputs
        L = ffi.lib()
        self._ctx = ctypes.c_void_p()
        ffi.check(L.qpgpu_ctx_create(device, ctypes.byref(self._ctx)))
        self._circuit = ctypes.c_void_p()
        ffi.check(L.qpgpu_circuit_create(self._ctx, ctypes.byref(self.desc), ctypes.byref(self._circuit)))
        self.proof_size = L.qpgpu_proof_size(self._circuit)
        # prover.bin carries the witness generators; the package carries the circuit's digest, which binds every constants/sigmas value: if it is
        # the digest of the wormhole circuit this library builds for the package's config, that circuit's generators are the package's
        try:
            b = ffi.BuiltCircuit(ffi.QPC_WORMHOLE, bool(self.desc.zero_knowledge))
            if list(b.desc.circuit_digest) == list(self.desc.circuit_digest) and b.desc.degree_bits == self.desc.degree_bits:
                self._built = b
        except (ImportError, ffi.CircuitError):
            pass
        return self

    @classmethod
    def from_built(cls, built, device=0):
        """Resident circuit from an ffi.BuiltCircuit (e.g. an aggregation node circuit)."""
        self = cls.__new__(cls)
        self.config = None
        self._synth, self._built = None, built
        self.desc = built.desc
        self.n = 1 << self.desc.degree_bits
        self.num_wires = self.desc.num_wires
        self.num_public_inputs = self.desc.num_public_inputs
        L = ffi.lib()
        self._ctx = ctypes.c_void_p()
        ffi.check(L.qpgpu_ctx_create(device, ctypes.byref(self._ctx)))
        self._circuit = ctypes.c_void_p()
        ffi.check(L.qpgpu_circuit_create(self._ctx, ctypes.byref(self.desc), ctypes.byref(self._circuit)))
        self.proof_size = L.qpgpu_proof_size(self._circuit)
        return self

    def close(self):
        """Frees the device-resident circuit (qpgpu_circuit_destroy / qpgpu_ctx_destroy); the object is unusable afterwards."""
        L = ffi.lib()
        if getattr(self, "_circuit", None):
            L.qpgpu_circuit_destroy(self._circuit); self._circuit = None
        if getattr(self, "_ctx", None):
            L.qpgpu_ctx_destroy(self._ctx); self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def common_bytes(self):
        return ffi.write_common(self.desc)

    def verifier_only_bytes(self):
        return ffi.write_verifier_only(self.constants_sigmas_cap(), self.desc.fri_cap_height, list(self.desc.circuit_digest))

    def package_bytes(self):
        return ffi.write_package(self.desc)

    @classmethod
    def get(cls, config, device=0):
        key = config.key() + (device,)
        if key not in cls._cache:
            cls._cache[key] = cls(config, device)
        return cls._cache[key]

    def generate_witness(self, circuit_inputs, blinding_key=None, threads=0):
        """generate_partial_witness for a built circuit: the reference's CircuitInputs -> (wires, public_inputs). `threads`: host threads
        for the circuit's independent segments (0 = up to 8)."""
        if self._built is None:
            raise RuntimeError("this circuit was not built from the wormhole circuit definition: commit SyntheticInputs")
        kind = self._built.kind
        targets = circuit_inputs.fill_targets() if kind == ffi.QPC_VOTE else circuit_inputs.fill_targets(kind)
        return self._built.witness(targets, blinding_key, threads=threads)

    def witness(self, seed):
        if self._synth is None:
            raise RuntimeError("this circuit has no synthetic witness source: commit CircuitInputs (built circuit) or an explicit witness (package)")
        wires = np.zeros((self.num_wires, self.n), dtype=np.uint64)
        pis = np.zeros(self.num_public_inputs, dtype=np.uint64)
        ffi.synth_lib().qpsynth_witness(self._synth, seed, wires.ctypes.data_as(ffi.u64p), pis.ctypes.data_as(ffi.u64p))
        return wires, pis

    DUMMY_SEED = 0x5157D0D0

    def dummy_proof(self):
        """The padding proof of this leaf circuit: a genuine proof of fixed, public dummy inputs, made once per circuit — what the
        reference exports with its #[ignore] tests (wormhole/tests/src/prover/prover_tests.rs:56-82) and embeds as DUMMY_PROOF_BYTES
        (wormhole/aggregator/src/util.rs:6-9). The reference's own blobs belong to its real circuit (other digest) and cannot pad a
        synthetic one."""
        if getattr(self, "_dummy", None) is None:
            if self._built is not None:       # the reference's dummy is a proof of CircuitInputs::test_inputs() (prover_tests.rs:56-82)
                from .test_helpers import circuit_test_inputs
                wires, pis = self.generate_witness(circuit_test_inputs(), ffi.TEST_SALT_KEY)
            else:
                wires, pis = self.witness(self.DUMMY_SEED)
            o = ffi.ProveOpts(); o.salt_stream = self.DUMMY_SEED
            keep = ffi.set_salt_key(o, ffi.TEST_SALT_KEY)      # noqa: F841  (public inputs of the dummy are public anyway)
            buf = ctypes.create_string_buffer(self.proof_size); n = ctypes.c_size_t()
            ffi.check(ffi.lib().qpgpu_prove(self._circuit, wires.ctypes.data, pis.ctypes.data_as(ffi.u64p), len(pis), ctypes.byref(o),
                                            buf, self.proof_size, ctypes.byref(n)))
            self._dummy = ProofWithPublicInputs(ctypes.string_at(buf, n.value), self.num_public_inputs)
        return self._dummy

    def constants_sigmas_cap(self):
        out = np.zeros((1 << self.desc.fri_cap_height, 4), dtype=np.uint64)
        ffi.check(ffi.lib().qpgpu_circuit_constants_sigmas_cap(self._circuit, out.ctypes.data_as(ffi.u64p)))
        return out


class WormholeProver:
    def __init__(self, circuit_data):
        self.circuit_data = circuit_data
        self._targets = True          # Option<CircuitTargets>: taken by commit (lib.rs:210-212)
        self._witness = None

    @staticmethod
    def new(config, device=0):
        return WormholeProver(WormholeCircuitData.get(config, device))

    @staticmethod
    def default(device=0):
        return WormholeProver.new(CircuitConfig.standard_recursion_zk_config(), device)

    @staticmethod
    def new_from_bytes(prover_only_bytes, common_bytes, device=0):
        """wormhole/prover/src/lib.rs:105-137 with this backend's artefacts: `prover_only_bytes` is the circuit package that stands where
        prover.bin stands, `common_bytes` the common.bin written next to it (it must be the common data embedded in the package). Malformed
        input raises (the reference unwraps, lib.rs:128)."""
        try:
            cd = WormholeCircuitData.from_package(prover_only_bytes, device)
        except ffi.QpgpuError as e:
            raise ValueError("Failed to deserialize prover data from bytes: %s" % e)
        if bytes(common_bytes) != cd.common_bytes():
            raise ValueError("Failed to deserialize common circuit data from bytes: it is not the common data of this prover package")
        return WormholeProver(cd)

    @staticmethod
    def new_from_files(prover_data_path, common_data_path, device=0):           # lib.rs:141-187
        return WormholeProver.new_from_bytes(open(prover_data_path, "rb").read(), open(common_data_path, "rb").read(), device)

    def commit(self, circuit_inputs):
        if not self._targets:
            raise RuntimeError("prover has already commited to inputs")
        self._targets = False
        cd = self.circuit_data
        if not isinstance(circuit_inputs, SyntheticInputs):
            # the reference's inputs on the built circuit; witness generation runs here (upstream defers it into prove(), where a
            # statement that does not hold fails with "set twice with different values": the error is kept and raised by prove())
            self._salt_key, self._salt_stream = os.urandom(32), 0
            try:
                self._witness = cd.generate_witness(circuit_inputs, os.urandom(32))
            except ffi.CircuitError as e:
                if e.code != ffi.QPGPU_ERR_WITNESS:
                    raise
                self._witness, self._witness_error = None, str(e)
            return self
        if circuit_inputs.wires is not None:
            wires = np.ascontiguousarray(circuit_inputs.wires, dtype=np.uint64)
            pis = np.ascontiguousarray(circuit_inputs.public_inputs, dtype=np.uint64)
            if wires.shape != (cd.num_wires, cd.n) or pis.shape != (cd.num_public_inputs,):
                raise ValueError("witness shape mismatch")
            # a real witness: salts come from the OS RNG, as qp-plonky2's F::rand_vec does (32 key bytes, expanded on the device)
            self._salt_key, self._salt_stream = (circuit_inputs.salt_key or os.urandom(32)), 0
        else:
            wires, pis = cd.witness(circuit_inputs.seed)
            # seeded synthetic inputs (tests, benches): the public test key, so the bytes are reproducible by the oracle
            self._salt_key, self._salt_stream = ffi.TEST_SALT_KEY, circuit_inputs.seed
        self._witness = (wires, pis)
        return self

    def prove(self, pow_witness=None):
        if self._witness is None:
            if getattr(self, "_witness_error", None):
                raise RuntimeError("Failed to prove: %s" % self._witness_error)
            raise RuntimeError("Failed to prove: prover has not commited to any inputs")
        cd = self.circuit_data
        wires, pis = self._witness
        self._witness = None                     # prove(self) consumes the prover (lib.rs:233)
        o = ffi.ProveOpts()
        o.salt_stream = self._salt_stream
        keep = ffi.set_salt_key(o, self._salt_key)      # noqa: F841
        if pow_witness is not None:
            o.have_pow_witness, o.pow_witness = 1, pow_witness
        buf = ctypes.create_string_buffer(cd.proof_size)
        n = ctypes.c_size_t()
        rc = ffi.lib().qpgpu_prove(cd._circuit, wires.ctypes.data, pis.ctypes.data_as(ffi.u64p), len(pis), ctypes.byref(o),
                                   buf, cd.proof_size, ctypes.byref(n))
        if rc != 0:
            raise RuntimeError("Failed to prove: %s" % ffi.lib().qpgpu_last_error().decode())
        return ProofWithPublicInputs(ctypes.string_at(buf, n.value), cd.num_public_inputs)
