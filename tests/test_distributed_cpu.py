"""N>1 path on CPU: world_size-2 gloo run of the proof-blob gather that precedes the aggregator step."""
import os, subprocess, sys, textwrap
from harness import ROOT


def test_gloo_world2_gather_of_proof_blobs(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import importlib, os, sys
        sys.path.insert(0, %r)
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        agg = importlib.import_module("zk-circuits_amd.aggregator")
        size = 1000
        local = [bytes([rank * 16 + i]) * size for i in range(3)]          # 3 fixed-size blobs per rank
        allp = agg.gather_proofs(local, size)
        assert len(allp) == 3 * world
        for r in range(world):
            for i in range(3):
                assert allp[r * 3 + i] == bytes([r * 16 + i]) * size
        dist.barrier(); dist.destroy_process_group()
        print("rank", rank, "ok")
    """ % ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29611", str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
