import importlib, os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
torch.cuda.init()
import importlib; H = importlib.import_module("zk-circuits_amd.ffi")          # the product binding only (no oracle in a measurement tool)
ffi = importlib.import_module("zk-circuits_amd.ffi"); th = importlib.import_module("zk-circuits_amd.test_helpers"); inp = importlib.import_module("zk-circuits_amd.inputs")
dev = torch.device("cuda", 0)
def run(name, desc, witness):
    gc = H.GpuCircuit(desc)
    seeds = [0x51570000 + i for i in range(16)]
    wd, pl = [], []
    for s in seeds:
        w, p = witness(s); wd.append(torch.from_numpy(w.view(np.int64)).to(dev)); pl.append(p)
    torch.cuda.synchronize()
    for k in range(4): gc.prove(None, pl[k], salt_seed=seeds[k], wires_dev_ptr=wd[k].data_ptr())
    per = []
    pw = []
    for rep in range(3):
        for k in range(16):
            t = time.perf_counter(); pf = gc.prove(None, pl[k], salt_seed=seeds[k], wires_dev_ptr=wd[k].data_ptr()); per.append(time.perf_counter() - t)
            if rep == 0: pw.append(int.from_bytes(pf[-(8 + 8 + 128):-(8 + 128)], "little"))
    per = np.array(per[16:]) * 1e3
    spans = [0 if x < 65536 else (1 if x < 65536 * 3 else (2 if x < 65536 * 7 else 3)) for x in pw]
    print(name, "mean %.3f ms  median %.3f  min %.3f  max %.3f | grind launches per proof %.2f" % (per.mean(), np.median(per), per.min(), per.max(), 1 + np.mean(spans)), spans)
    byk = per.reshape(2, 16).mean(0)
    print("   per witness ms:", " ".join("%.2f" % v for v in byk))
    gc.close()
sc = H.SynthCircuit("B")
run("B", sc.desc, sc.witness)
c = ffi.BuiltCircuit(ffi.QPC_WORMHOLE, True)
def wit(seed):
    ci = th.circuit_test_inputs(); ci.public.exit_account = inp.BytesDigest(int(seed).to_bytes(8, "little") + bytes(24))
    return c.witness(ci.fill_targets(), int(seed).to_bytes(32, "little"))
run("W", c.desc, wit)
run("B", sc.desc, sc.witness)
