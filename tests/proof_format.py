"""ProofWithPublicInputs::to_bytes() reader for the tests (layout: SURVEY.md App. B.3; the reference deserialises the same bytes
at wormhole/aggregator/src/util.rs:22). Pure Python/numpy, no oracle and no product code: it only slices the byte string."""
import numpy as np


def _u64s(buf, off, count):
    return np.frombuffer(buf, dtype="<u8", count=count, offset=off).copy()


def parse_proof(d, proof):
    """d: a circuit descriptor (harness.CircuitDesc or anything with the same shape fields). Returns a dict of numpy arrays."""
    nch = d.num_challenges
    S = d.num_constants + d.num_routed
    W = d.num_wires
    Zp = nch * (1 + d.num_partial_products)
    Q = nch * d.quotient_degree_factor
    salt = 4 if d.zero_knowledge else 0
    capn = 1 << d.fri_cap_height
    log_N = d.degree_bits + d.fri_rate_bits
    out = {}
    off = 0
    for name in ("wires_cap", "zs_cap", "quotient_cap"):
        out[name] = _u64s(proof, off, capn * 4).reshape(capn, 4); off += capn * 32
    n_open = S + W + 2 * nch + nch * d.num_partial_products + Q
    out["openings"] = _u64s(proof, off, 2 * n_open).reshape(n_open, 2); off += 16 * n_open
    out["fri_caps"] = []
    for _ in range(d.fri_num_layers):
        out["fri_caps"].append(_u64s(proof, off, capn * 4).reshape(capn, 4)); off += capn * 32
    widths = [S, W + salt, Zp + salt, Q + salt]
    queries = []
    for _ in range(d.fri_num_queries):
        q = {"rows": [], "paths": [], "layer_evals": [], "layer_paths": []}
        for w in widths:
            q["rows"].append(np.frombuffer(proof, dtype="<u8", count=w, offset=off).copy()); off += 8 * w
            k = proof[off]; off += 1
            assert k == log_N - d.fri_cap_height, "unexpected sibling count"
            q["paths"].append(np.frombuffer(proof, dtype="<u8", count=4 * k, offset=off).copy().reshape(k, 4)); off += 32 * k
        lg = d.degree_bits
        for l in range(d.fri_num_layers):
            a = d.fri_arity_bits[l]
            q["layer_evals"].append(np.frombuffer(proof, dtype="<u8", count=2 << a, offset=off).copy().reshape(1 << a, 2)); off += 16 << a
            k = proof[off]; off += 1
            assert k == lg + d.fri_rate_bits - a - d.fri_cap_height
            q["layer_paths"].append(np.frombuffer(proof, dtype="<u8", count=4 * k, offset=off).copy().reshape(k, 4)); off += 32 * k
            lg -= a
        queries.append(q)
    out["queries"] = queries
    fl = 1 << (d.degree_bits - sum(d.fri_arity_bits[i] for i in range(d.fri_num_layers)))
    out["final_poly"] = np.frombuffer(proof, dtype="<u8", count=2 * fl, offset=off).copy().reshape(fl, 2); off += 16 * fl
    out["pow_witness"] = int.from_bytes(proof[off:off + 8], "little"); off += 8
    npi = int.from_bytes(proof[off:off + 8], "little"); off += 8
    assert npi == d.num_public_inputs
    out["public_inputs"] = np.frombuffer(proof, dtype="<u8", count=npi, offset=off).copy(); off += 8 * npi
    assert off == len(proof), "proof length does not match the circuit shape"
    return out
