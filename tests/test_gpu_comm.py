"""The native exchange of the sharded path (include/qpgpu.h qpgpu_comm_*, csrc/comm.hip): an all-gather of fixed-size proof blobs over RCCL
behind the C-ABI — SURVEY.md §8(e); the reference hands proofs between rayon workers in shared memory
(/root/reference/wormhole/aggregator/src/circuits/tree.rs:92-103). One GPU here, so the communicator has one rank; the multi-rank
schedule around it is covered on gloo (tests/test_distributed_cpu.py) and with two ranks sharing the device (tests/test_gpu_tree_512.py)."""
import importlib
import os
import subprocess
import sys
import textwrap
import numpy as np
import pytest
from harness import ROOT

ffi = importlib.import_module("zk-circuits_amd.ffi")


def test_rccl_is_found_next_to_the_hip_runtime():
    assert ffi.lib().qpgpu_comm_rccl_version() >= 20000          # ncclGetVersion code (2.x.y -> 2xxyy)


@pytest.mark.gpu
def test_native_gather_world_of_one():
    uid = ffi.ProofComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = ffi.ProofComm(0, 1, uid)
    rng = np.random.default_rng(7)
    for count, size in ((1, 148932), (3, 132712), (64, 148932), (5, 1), (2, 255), (1, 257)):          # the golden proof sizes, a batch, odd sizes
        blobs = [rng.integers(0, 256, size=size, dtype=np.uint8).tobytes() for _ in range(count)]
        assert comm.gather(blobs, size) == blobs
    assert comm.gather([], 148932) == []
    with pytest.raises(ffi.QpgpuError):
        ffi.ProofComm(1, 1, uid)                                   # rank outside the world
    comm.close()


@pytest.mark.gpu
def test_gather_proofs_takes_the_native_path_on_an_rccl_group(tmp_path):
    """aggregator.gather_proofs on a torch.distributed group whose backend is RCCL: the unique id travels over the group, the exchange itself
    is qpgpu_gather_proofs; the result equals torch.distributed's all_gather of the same blobs (QPGPU_NATIVE_GATHER=0 forces that path)."""
    script = tmp_path / "t.py"
    script.write_text(textwrap.dedent("""
        import importlib, os, sys
        sys.path.insert(0, %r)
        import torch
        torch.cuda.init(); torch.cuda.set_device(0)
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
        agg = importlib.import_module("zk-circuits_amd.aggregator")
        dev = torch.device("cuda", 0)
        blobs = [bytes([i]) * 148932 for i in range(1, 6)]
        native = agg.native_comm(None, dev)
        assert native is not None, "no native communicator on an RCCL group"
        got = agg.gather_proofs(blobs, 148932, device=dev)
        assert got == blobs
        os.environ["QPGPU_NATIVE_GATHER"] = "0"
        assert agg.native_comm(None, dev) is None
        assert agg.gather_proofs(blobs, 148932, device=dev) == blobs
        dist.barrier(); dist.destroy_process_group()
        print("ok")
    """) % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
