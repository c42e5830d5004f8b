import importlib, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch; torch.cuda.init()
import harness as H
g = os.path.join(os.getcwd(), "tests", "golden"); rd = lambda f: open(os.path.join(g, f), "rb").read()
c, vb, pf = rd("common.bin"), rd("verifier.bin"), rd("proof.bin")
vo = vb[:len(vb) - len(c)] if vb.endswith(c) and len(vb) > len(c) else vb
sh = H.ReferenceCircuitShape(c, vo); gv = H.GpuVerifier(sh.desc, sh.constants_sigmas_cap)
for _ in range(3): gv.verify(pf)
print("--- single", file=sys.stderr); gv.verify(pf)
print("--- batch 64", file=sys.stderr); gv.verify_batch([pf] * 64); gv.verify_batch([pf] * 64)
