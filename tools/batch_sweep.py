"""Stream-pipelined throughput sweep (tuning aid): proofs/s of qpgpu_prove_batch for QPGPU_STREAMS x batch sizes."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    torch.cuda.init()
    import importlib; H = importlib.import_module("zk-circuits_amd.ffi")          # the product binding only (no oracle in a measurement tool)
    sc = H.SynthCircuit(os.environ.get("PRESET", "B")); gc = H.GpuCircuit(sc.desc)
    nb = int(os.environ.get("NB", "32"))
    wd, pl = [], []
    for i in range(4):
        w, p = sc.witness(100 + i); wd.append(torch.from_numpy(w.view(np.int64)).cuda()); pl.append(p)
    ptrs = [wd[j % 4].data_ptr() for j in range(nb)]; pis = [pl[j % 4] for j in range(nb)]
    gc.prove_batch(ptrs, pis, salt_seed=1, wires_on_device=True)
    rates = []
    for _ in range(int(os.environ.get("REPS", "3"))):
        torch.cuda.synchronize(); t = time.perf_counter()
        gc.prove_batch(ptrs, pis, salt_seed=1, wires_on_device=True)
        torch.cuda.synchronize(); rates.append(nb / (time.perf_counter() - t))
    rates.sort()
    print(json.dumps({"streams": os.environ.get("QPGPU_STREAMS"), "shared": os.environ.get("QPGPU_SHARED_MODE"), "nb": nb,
                      "proofs_per_s": round(rates[len(rates) // 2], 1), "min": round(rates[0], 1), "max": round(rates[-1], 1)}))
    gc.close()
else:
    for st in [int(x) for x in os.environ.get("SWEEP_STREAMS", "1,2,3,4").split(",")]:
        for shared in os.environ.get("SWEEP_SHARED", "0,1").split(","):
            env = dict(os.environ, QPGPU_STREAMS=str(st), QPGPU_SHARED_MODE=shared)
            subprocess.run([sys.executable, __file__, "child"], env=env)
