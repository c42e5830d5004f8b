import sys, time, importlib, os
sys.path.insert(0, os.getcwd())
ffi = importlib.import_module("zk-circuits_amd.ffi"); th = importlib.import_module("zk-circuits_amd.test_helpers")
def bench(c, x, key, label):
    o = c.witness(x, key, threads=1); ref = o[0].copy()
    for thr in (1, 2, 4, 8, 16):
        ts=[]
        for _ in range(10):
            t=time.perf_counter(); c.witness(x, key, out=o, threads=thr); ts.append(time.perf_counter()-t)
        print(label, thr, "threads: min %.2f median %.2f ms" % (min(ts)*1e3, sorted(ts)[5]*1e3), bool((o[0]==ref).all()), flush=True)
for zk in (False, True):
    c = ffi.BuiltCircuit(ffi.QPC_WORMHOLE, zk)
    bench(c, th.circuit_test_inputs().fill_targets(), bytes(32), "wormhole zk=%d" % zk)
# node over two synthetic B leaves via the device
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch; torch.cuda.init()
sc = ffi.SynthCircuit("B"); gc = ffi.GpuCircuit(sc.desc)
leaves = []
for i in range(2):
    w, p = sc.witness(i); leaves.append(gc.prove(w, p, salt_seed=i))
n = ffi.BuiltCircuit.aggregation(sc.desc, 2)
bench(n, n.aggregation_targets(gc.cap(), sc.desc.circuit_digest, leaves), bytes(32), "zk node")
