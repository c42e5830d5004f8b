#!/usr/bin/env python3
"""Full-size check (BASELINE configs[4]: voting-shaped circuit at 2^20 rows, or any preset): prove on the GPU, then have
the oracle VERIFIER accept the proof (size-independent property; the CPU prover would take minutes at this size)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import harness as H

preset = sys.argv[1] if len(sys.argv) > 1 else "V"
t = time.time(); sc = H.SynthCircuit(preset); print("circuit built %.1fs (2^%d rows)" % (time.time() - t, sc.desc.degree_bits), flush=True)
t = time.time(); gc = H.GpuCircuit(sc.desc); print("gpu circuit (preprocess) %.1fs" % (time.time() - t), flush=True)
t = time.time(); w, p = sc.witness(0x51570000); print("witness %.1fs" % (time.time() - t), flush=True)
os.environ["QPGPU_TIMING"] = "1"
for i in range(2):
    t = time.time(); pf = gc.prove(w, p, salt_seed=1); dt = time.time() - t
    print("prove %.1f ms, %d bytes, stages %s" % (dt * 1e3, len(pf), [round(x, 2) for x in gc.stage_ms()]), flush=True)
vo = H.OracleCircuit(sc.desc, verifier_only_cap=gc.cap())
err = vo.verify(pf)
print("oracle verifier:", "ACCEPT" if err is None else "REJECT: " + err, flush=True)
sys.exit(0 if err is None else 1)
