#!/bin/bash
# Sanitizer runs of the host-only circuit library (libqpcircuit), from the repo root: bash tools/sanitize/run.sh
#  1. AddressSanitizer + UBSan: the library rebuilt instrumented, the circuit / aggregation test files run against it (libasan preloaded);
#  2. ThreadSanitizer: witness generation on 8 threads (the wormhole circuit's node segments; a tree node's children and FRI queries, and the
#     error path of a forged child) in a C++ driver linked with the instrumented sources; inputs dumped by Python into /tmp/tsan.
# The in-tree library is restored afterwards.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
SRC="$R/zk-circuits_amd/circuit/builder.cc $R/zk-circuits_amd/circuit/circuits.cc $R/zk-circuits_amd/circuit/recursion.cc $R/zk-circuits_amd/circuit/mutation.cc"
cp $R/zk-circuits_amd/libqpcircuit.so /tmp/libqpcircuit.so.bak
trap 'cp /tmp/libqpcircuit.so.bak $R/zk-circuits_amd/libqpcircuit.so' EXIT
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unused-function $SRC -pthread -o $R/zk-circuits_amd/libqpcircuit.so
(cd $R && ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" python -m pytest tests/test_wormhole_circuit.py tests/test_aggregation_circuit.py tests/test_hint_soundness.py -x -q -s | tail -2)
cp /tmp/libqpcircuit.so.bak $R/zk-circuits_amd/libqpcircuit.so
mkdir -p /tmp/tsan
(cd $R && python3 - <<'PY'
import sys, importlib
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from harness import *
th = importlib.import_module("zk-circuits_amd.test_helpers")
inner = BuiltCircuit(QPC_VOTE, False); oc = OracleCircuit(inner.desc)
w, p = inner.witness(th.vote_test_inputs().fill_targets()); leaf = oc.prove(w, p)
n1 = BuiltCircuit.aggregation(inner.desc, 2)
x = n1.aggregation_targets(oc.cap(), inner.desc.circuit_digest, [leaf, leaf]); x.tofile('/tmp/tsan/node_inputs.bin')
th.circuit_test_inputs().fill_targets().tofile('/tmp/tsan/wormhole_inputs.bin')
n1.witness(x, threads=1)[0].tofile('/tmp/tsan/node_wires.bin')
PY
)
g++ -O1 -g -std=c++17 -fsanitize=thread -fPIE -pie $R/tools/sanitize/witness_threads_driver.cc $SRC -pthread -o /tmp/tsan/driver
TSAN_OPTIONS="halt_on_error=1" /tmp/tsan/driver
echo "sanitizers: clean"
