"""WormholeVerifier mirror (reference: /root/reference/wormhole/verifier/src/lib.rs:81-159).

`new_from_bytes(verifier_bytes, common_bytes)` / `new_from_files(...)` take the artefacts `wormhole/circuit-builder` writes
(`verifier.bin`, `common.bin`; parsed by the product's own loaders qpgpu_parse_verifier_only / qpgpu_parse_common) and `verify(proof)`
is `VerifierCircuitData::verify` — here through qpgpu_verify_batch (include/qpgpu.h): Merkle openings on the device, transcript and
field arithmetic on the host. `verify_batch` is the many-proofs form the aggregator uses when it takes in leaf proofs. `new(config)`
builds the verifier data of the synthetic circuit package that stands in for `WormholeCircuit::new(config).build_verifier()`."""
from . import ffi


class WormholeVerifier:
    def __init__(self, desc, cap, device=0, _keep=None):
        self._keep = _keep                     # owner of the arrays `desc` points into
        self._v = ffi.GpuVerifier(desc, cap, device)
        self.proof_size = self._v.proof_size

    @staticmethod
    def new(config, circuit_data=None, device=0):                 # lib.rs:87-95
        from .prover import WormholeCircuitData
        cd = circuit_data or WormholeCircuitData.get(config, device)
        return WormholeVerifier(cd.desc, cd.constants_sigmas_cap(), device, _keep=cd)

    @staticmethod
    def new_from_bytes(verifier_bytes, common_bytes, device=0):   # lib.rs:98-115: string errors, no panic
        try:
            vo = bytes(verifier_bytes)
            if vo.endswith(bytes(common_bytes)) and len(vo) > len(common_bytes):
                vo = vo[:len(vo) - len(common_bytes)]             # verifier.bin as written by the reference = verifier-only || common
            shape = ffi.ReferenceCircuitShape(bytes(common_bytes), vo)
        except RuntimeError as e:
            msg = str(e)
            raise ValueError("Failed to deserialize %s from bytes" % ("verifier data" if "cap" in msg or "verifier" in msg else "common circuit data"))
        return WormholeVerifier(shape.desc, shape.constants_sigmas_cap, device, _keep=shape)

    @staticmethod
    def new_from_files(verifier_data_path, common_data_path, device=0):     # lib.rs:118-148
        try:
            return WormholeVerifier.new_from_bytes(open(verifier_data_path, "rb").read(), open(common_data_path, "rb").read(), device)
        except ValueError as e:
            raise RuntimeError("%s (%s, %s)" % (e, verifier_data_path, common_data_path))

    def verify(self, proof):                                      # lib.rs:155-159
        data = proof.to_bytes() if hasattr(proof, "to_bytes") else bytes(proof)
        msg = self._v.verify(data)
        if msg is not None:
            raise RuntimeError("proof verification failed: %s" % msg)

    def verify_batch(self, proofs):
        """[True/False] per proof (no exception for a rejected proof)."""
        return self._v.verify_batch([p.to_bytes() if hasattr(p, "to_bytes") else bytes(p) for p in proofs])[0]

    def close(self):
        self._v.close()
