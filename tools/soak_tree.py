import importlib, os, sys, time, resource
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch; torch.cuda.init()
pkg = importlib.import_module("zk-circuits_amd"); th = importlib.import_module("zk-circuits_amd.test_helpers")
cd = pkg.WormholeCircuitData.get(pkg.CircuitConfig.standard_recursion_zk_config())
prover = pkg.RecursiveChunkProver(cd)
cfg = pkg.TreeAggregationConfig(2, int(os.environ.get("SOAK_DEPTH", "3")))          # depth >= 3: the dependency-driven scheduler (threads) runs every iteration
t0 = time.time()
for it in range(int(os.environ.get("SOAK_ITERS", "25"))):
    agg = pkg.WormholeProofAggregator(cd, cfg)
    ins = [th.circuit_test_inputs() for _ in range(cfg.num_leaf_proofs - 1)]          # one short: the dummy proof pads the last leaf
    for i, ci in enumerate(ins): ci.public.exit_account = pkg.BytesDigest(bytes([it & 0x7f, i] + [0] * 30))
    agg.prove_leaves(ins)
    root = agg.aggregate(chunk_prover=prover)
    pkg.WormholeVerifier.new(None, circuit_data=root.circuit_data).verify(root.proof)
    if it % 5 == 0:
        free, total = torch.cuda.mem_get_info()
        print("iter %d  %.1f s  host RSS %.0f MB  GPU used %.1f GB" % (it, time.time() - t0, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, (total - free) / 2**30), flush=True)
print("soak ok")
