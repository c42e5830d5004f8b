"""Mirror of the reference's circuit-builder (/root/reference/wormhole/circuit-builder/src/lib.rs:11-66): build the circuit, write
`common.bin`, `verifier.bin` and the prover artefact into a directory; `WormholeProver.new_from_files` / `WormholeVerifier.new_from_files`
load them back (wormhole/tests/src/circuit/circuit_data_tests.rs:96-183 proves and verifies from such files).

`common.bin` and `verifier.bin` are qp-plonky2's own byte formats (the product's writers are byte-stable on the reference's bench-data files).
The prover artefact is `prover.qpkg`, this backend's circuit package, not `prover.bin`: ProverOnlyCircuitData's layout is pinned by nothing the
reference ships (wormhole/generated-bins/ holds only .gitkeep), and the package carries exactly what the device path needs."""
import os
from .prover import CircuitConfig, WormholeCircuitData


def generate_circuit_binaries(output_dir, include_prover=True, config=None, device=0):
    cd = WormholeCircuitData.get(config or CircuitConfig.standard_recursion_config(), device)     # circuit-builder/src/lib.rs:16: non-zk config
    os.makedirs(output_dir, exist_ok=True)
    written = {}
    for name, data in (("common.bin", cd.common_bytes()), ("verifier.bin", cd.verifier_only_bytes())):
        with open(os.path.join(output_dir, name), "wb") as f:
            f.write(data)
        written[name] = len(data)
    if include_prover:
        data = cd.package_bytes()
        with open(os.path.join(output_dir, "prover.qpkg"), "wb") as f:
            f.write(data)
        written["prover.qpkg"] = len(data)
    return written
