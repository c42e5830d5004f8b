#!/usr/bin/env python3
"""Headline benchmark: wormhole proofs/s on MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = one pass of the hot path (prove_with_partition_witness through the qpgpu C-ABI) over one witness of the configured
circuit, witness already resident in HBM. Default workload = BASELINE configs[1]: a single proof of the wormhole circuit itself
(built by libqpcircuit from the reference's circuit definition, zk config, degree 2^14, the reference's test inputs: preset "W");
the synthetic shapes of round 1 stay selectable ("B" = the same shape with random gate contents). With N ranks every rank proves its own independent witnesses (weak scaling, no
data-path collective); after the K steps the fixed-size proof blobs are all-gathered over RCCL (the exchange that
precedes the aggregator step) inside the timed region.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# ROCm maps a process's HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); qpgpu_prove_batch's workers need one each to overlap.
# Must be in the environment before the HIP runtime initialises (torch does that below), so it is set here, not in the library.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def available_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota, divided by the ranks of this node — the one
    helper every host pool is sized from (zk-circuits_amd/hostcpus.py, mirrored by csrc/hostcpus.h inside the libraries)."""
    import importlib
    return importlib.import_module("zk-circuits_amd.hostcpus").usable_cpus()


def pin_to_gpu_numa_node(torch, device_index):
    """Confines this process (and the library's worker threads, which inherit the mask) to the CPUs local to the GPU's PCIe root
    (numactl --cpunodebind equivalent). Each proof has 13 GPU->host->GPU hand-offs; from the remote socket of a two-socket host
    they cost ~10 us more each (3.70 vs 3.92 ms per proof measured on the same binaries). QP_BENCH_NO_PIN=1 disables."""
    if os.environ.get("QP_BENCH_NO_PIN"):
        return None
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        txt = open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip()
        cpus = set()
        for part in txt.split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
            return "%s: %s" % (bdf, txt)
    except Exception:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--preset", default="W", help="W: the wormhole circuit itself, zk config, 2^14 rows, on the reference's test inputs (default); WN: the same, non-zk, 2^13; "
                    "synthetic shapes — A: leaf non-zk 2^13, B: leaf zk 2^14, V: voting-shaped 2^20, R: recursion-shaped 2^13")
    ap.add_argument("--batch", type=int, default=1, help=">1: each step is one stream-pipelined batch (BASELINE configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra (untimed-headline) stream-pipelined throughput measurement")
    ap.add_argument("--cpu-proofs", type=int, default=8)
    ap.add_argument("--no-tree", action="store_true", help="skip the extra aggregation-tree measurement (BASELINE configs[2]/[3] flavour)")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no pipelined / tree / real-circuit / golden-verify extras and no CPU baseline (profiler runs: one workload per trace)")
    ap.add_argument("--no-real-circuit", action="store_true", help="skip the extra line on the wormhole circuit built by libqpcircuit (the reference's test inputs)")
    args = ap.parse_args()
    if args.no_extras:
        args.no_pipelined = args.no_tree = args.no_real_circuit = args.no_cpu_baseline = True

    import importlib
    import numpy as np
    import torch
    H = importlib.import_module("zk-circuits_amd.ffi")          # the product binding; oracle/ is imported only by the cpu_baseline leg

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE" % (world, args.gpus), file=sys.stderr)
    dist = None
    # stdout carries exactly one JSON line: libraries that print banners there (RCCL prints its version block at communicator
    # creation) are pointed at stderr, and the line goes to the saved descriptor at the end
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    # Rehearsal switches for a box with fewer GPUs than ranks (never used by the driver): QP_BENCH_BACKEND=gloo runs the collectives on CPU
    # tensors (RCCL refuses two ranks on one device) and QP_BENCH_SHARE_GPU=1 maps rank r to device r % device_count.
    backend = os.environ.get("QP_BENCH_BACKEND", "nccl")
    if os.environ.get("QP_BENCH_SHARE_GPU"):
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if world > 1 or os.environ.get("QP_BENCH_FORCE_DIST"):      # the env switch lets a 1-GPU box exercise the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # where collective payloads live
    pinned = pin_to_gpu_numa_node(torch, local_rank)

    # ---- circuit (resident) + synthetic witnesses resident in HBM (outside the timed region)
    class WormholeWorkload:
        """The wormhole circuit (libqpcircuit: the reference's circuit definition on this repo's builder, include/qpcircuit.h) with the
        reference's CircuitInputs::test_inputs(): proof `seed` pays out to exit account `seed` (a public input the circuit does not
        constrain) and takes blinding key `seed`; witnesses are generated on the host before the timed region and kept in HBM."""
        def __init__(self, zk):
            import importlib
            self.ffi = importlib.import_module("zk-circuits_amd.ffi"); self.th = importlib.import_module("zk-circuits_amd.test_helpers")
            self.inp = importlib.import_module("zk-circuits_amd.inputs")
            self.c = self.ffi.BuiltCircuit(self.ffi.QPC_WORMHOLE, zk)
            self.desc, self.n, self.N, self.zk, self._salts = self.c.desc, self.c.n, self.c.N, zk, None
        def witness(self, seed):
            ci = self.th.circuit_test_inputs()
            ci.public.exit_account = self.inp.BytesDigest(int(seed).to_bytes(8, "little") + bytes(24))
            return self.c.witness(ci.fill_targets(), int(seed).to_bytes(32, "little"))
        def salts(self, seed):                                    # the device's salt stream on the host (cpu_baseline leg: the oracle needs explicit salts)
            if self._salts is None:
                self._salts = H.SynthCircuit("B" if self.zk else "A")   # same N: the keystream depends on (key, stream, N) only
            return self._salts.salts(seed)
    real_workload = args.preset in ("W", "WN")
    sc = WormholeWorkload(args.preset == "W") if real_workload else H.SynthCircuit(args.preset)
    gc = H.GpuCircuit(sc.desc, device=local_rank)
    L = H.qpgpu()
    nwit = 16                            # distinct witnesses cycled through: enough of them that the data-dependent part (the proof-of-work search:
                                         # 1 + 0.37 + 0.05 + ... grind launches expected per proof) averages out instead of reflecting four witnesses' luck
    if args.preset == "V":
        nwit = 2                         # 1.1 GB per witness
    seeds = [0x51570000 + rank * 64 + i for i in range(nwit)]
    wit_dev, pis_list = [], []
    for sd in seeds:
        w, p = sc.witness(sd)
        wit_dev.append(torch.from_numpy(w.view(np.int64)).to(dev))
        pis_list.append(p)
    torch.cuda.synchronize()

    def step(i):
        if args.batch == 1:
            k = i % nwit
            return [gc.prove(None, pis_list[k], salt_seed=seeds[k], wires_dev_ptr=wit_dev[k].data_ptr())]
        ptrs = [wit_dev[(i * args.batch + j) % nwit].data_ptr() for j in range(args.batch)]
        pl = [pis_list[(i * args.batch + j) % nwit] for j in range(args.batch)]
        return gc.prove_batch(ptrs, pl, salt_seed=seeds[0] + i * args.batch, wires_on_device=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_w = time.perf_counter()
    for i in range(args.warmup):
        step(i)
    # settle: a fresh box runs the first ~second of proofs 3-5 % slower (clocks, first-touch of the workspaces), so warm-up
    # continues (untimed) until one second has passed; the timed region below is still exactly K steps
    settle = float(os.environ.get("QP_BENCH_SETTLE_S", "1.0"))
    i = args.warmup
    while time.perf_counter() - t_w < settle:
        step(i); i += 1
    # untimed pass with every kernel group instrumented (HIP events cost a ~10 us bubble each, so the timed region below only
    # instruments the dominant group)
    L.qpgpu_profile_reset()
    L.qpgpu_profile_enable(2)
    for i in range(min(5, max(2, args.steps))):
        step(i)
    L.qpgpu_profile_enable(0)
    all_groups = {}
    for kid in range(L.qpgpu_profile_count()):
        n_l, ms, by = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
        L.qpgpu_profile_get(kid, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(by))
        if n_l.value:
            all_groups[L.qpgpu_profile_name(kid).decode()] = {"launches": n_l.value, "avg_ms": ms.value / n_l.value, "total_ms": ms.value,
                                                            "alg_bytes_per_launch": by.value / n_l.value}
    gather_path = None
    if dist is not None:
        # the first collective on a communicator connects the ranks and allocates its buffers (tens of ms): do one gather of the
        # same shape before the timed region, as a long-running aggregator would have. The gather is the library's own
        # (qpgpu_gather_proofs: ncclAllGather behind the C-ABI) when the backend is RCCL; aggregator.gather_proofs falls back to
        # torch.distributed's all_gather if the native communicator cannot be set up on every rank, and on gloo.
        agg_mod = importlib.import_module("zk-circuits_amd.aggregator")
        gdev = dev if backend == "nccl" else None
        agg_mod.gather_proofs([bytes(gc.proof_size)] * (args.steps * args.batch), gc.proof_size, device=gdev)
        gather_path = ("qpgpu_gather_proofs (ncclAllGather behind the C-ABI, RCCL %s)" % H.qpgpu().qpgpu_comm_rccl_version()
                       if agg_mod.native_comm(None, gdev) is not None else "torch.distributed all_gather (%s)" % backend)
    L.qpgpu_profile_reset()
    L.qpgpu_profile_enable(1)
    barrier()
    t0 = time.perf_counter()
    proofs = []
    for i in range(args.steps):
        proofs += step(i)
    t_steps = time.perf_counter() - t0                            # this rank's K steps, before the exchange
    if dist is not None:
        gathered = agg_mod.gather_proofs(proofs, gc.proof_size, device=gdev)
        assert len(gathered) == world * len(proofs) and gathered[rank * len(proofs)] == proofs[0]
    barrier()
    dt = time.perf_counter() - t0
    L.qpgpu_profile_enable(0)
    per_rank_ms = [t_steps * 1e3 / (args.steps * args.batch)]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        t = torch.tensor([t_steps * 1e3 / (args.steps * args.batch)], dtype=torch.float64, device=cdev)
        allt = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [float(x.item()) for x in allt]

    # ---- per-kernel-group timing -> roofline of the dominant kernel
    kernels = {}
    for kid in range(L.qpgpu_profile_count()):
        n_l, ms, by = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
        L.qpgpu_profile_get(kid, ctypes.byref(n_l), ctypes.byref(ms), ctypes.byref(by))
        if n_l.value:
            kernels[L.qpgpu_profile_name(kid).decode()] = {"launches": n_l.value, "avg_ms": ms.value / n_l.value,
                                                          "total_ms": ms.value, "alg_bytes_per_launch": by.value / n_l.value}
    timed_groups = kernels
    kernels = dict(all_groups); kernels.update(timed_groups)       # dominant group: numbers from the timed region itself
    roofline = None
    if timed_groups:
        dom = max(timed_groups, key=lambda k: timed_groups[k]["total_ms"])
        kd = kernels[dom]
        achieved = kd["alg_bytes_per_launch"] / (kd["avg_ms"] * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")       # HBM bytes per launch from rocprofv3 --pmc passes (if collected)
        if os.path.exists(tf):
            try:
                tkey = args.preset if args.batch == 1 else "%s%d" % (args.preset, args.batch)
                tj = json.load(open(tf))
                traffic = tj.get(tkey, tj.get(args.preset, {})).get(dom)
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 5), "traffic": traffic,
                    "note": "Poseidon leaf hashing is integer-VALU-bound, not HBM-bound (DESIGN.md §4); frac is still quoted against HBM peak",
                    "avg_launch_ms": round(kd["avg_ms"], 4), "alg_bytes_per_launch": kd["alg_bytes_per_launch"]}

    # proofs to byte-check against the oracle in the cpu_baseline leg: (label, bytes, witness index, salt seed)
    to_check = []
    if args.batch > 1:
        nb_ = args.batch; last = args.steps * nb_ - 1
        to_check += [("batch_first", proofs[0], 0, seeds[0]), ("batch_last", proofs[-1], last % nwit, seeds[0] + last)]
    # ---- extra, NOT the headline: stream-pipelined throughput of independent proofs on this GPU (BASELINE configs[2] style)
    pipelined = None
    if args.batch == 1 and not args.no_pipelined:
        nb = 32
        ptrs = [wit_dev[j % nwit].data_ptr() for j in range(nb)]
        pl = [pis_list[j % nwit] for j in range(nb)]
        gc.prove_batch(ptrs, pl, salt_seed=seeds[0], wires_on_device=True)          # warm the per-stream workspaces
        barrier(); t2 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            pp = gc.prove_batch(ptrs, pl, salt_seed=seeds[0], wires_on_device=True)
        barrier(); pdt = time.perf_counter() - t2
        to_check += [("pipelined_first", pp[0], 0, seeds[0]), ("pipelined_last", pp[-1], (nb - 1) % nwit, seeds[0] + nb - 1)]
        if dist is not None:
            t = torch.tensor([pdt], dtype=torch.float64, device=cdev); dist.all_reduce(t, op=dist.ReduceOp.MAX); pdt = float(t.item())
        pipelined = {"proofs_per_s": round(reps * nb * world / pdt, 2), "ms_per_proof": round(pdt / (reps * nb) * 1e3, 4),
                     "batch": nb, "streams": int(os.environ.get("QPGPU_STREAMS", "8" if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 8 else "4")), "note": "qpgpu_prove_batch: worker threads, one HIP stream + workspace each"}

    # ---- extra, NOT the headline: the wormhole circuit itself (libqpcircuit: the reference's circuit definition on this repo's builder,
    # include/qpcircuit.h) proving the reference's CircuitInputs::test_inputs(): host witness generation + device proof. Same statement and
    # public inputs as the reference; gate placement is this builder's own, so the rows differ from the synthetic shape above.
    real = None; real_check = None
    if rank == 0 and args.batch == 1 and not args.no_real_circuit and args.preset in ("A", "B", "W", "WN"):
        try:
            import importlib
            ffi_ = importlib.import_module("zk-circuits_amd.ffi"); th_ = importlib.import_module("zk-circuits_amd.test_helpers")
            inp_ = importlib.import_module("zk-circuits_amd.inputs")
            zk_ = args.preset in ("B", "W")
            t0_ = time.perf_counter(); bc = ffi_.BuiltCircuit(ffi_.QPC_WORMHOLE, zk_); build_s = time.perf_counter() - t0_
            ci = th_.circuit_test_inputs()
            reps_w = 20; t0_ = time.perf_counter()
            for _ in range(reps_w):
                w_, p_ = bc.witness(ci.fill_targets(), H.TEST_SALT_KEY)
            wit_ms = (time.perf_counter() - t0_) / reps_w * 1e3
            t0_ = time.perf_counter()
            for _ in range(reps_w):
                bc.witness(ci.fill_targets(), H.TEST_SALT_KEY, out=(w_, p_))
            wit_reuse_ms = (time.perf_counter() - t0_) / reps_w * 1e3
            gcW = H.GpuCircuit(bc.desc, device=local_rank)
            wd_ = torch.from_numpy(w_.view(np.int64)).to(dev); torch.cuda.synchronize()
            for _ in range(3):
                pfW = gcW.prove(None, p_, salt_seed=77, wires_dev_ptr=wd_.data_ptr())
            reps_p = max(5, min(args.steps, 20)); t0_ = time.perf_counter()
            for _ in range(reps_p):
                pfW = gcW.prove(None, p_, salt_seed=77, wires_dev_ptr=wd_.data_ptr())
            ms_ = (time.perf_counter() - t0_) / reps_p * 1e3
            pub_ = inp_.PublicCircuitInputs.try_from_slice(np.frombuffer(pfW[-128:], dtype="<u8"))
            real = {"circuit": "wormhole (libqpcircuit), %s" % ("standard_recursion_zk_config" if zk_ else "standard_recursion_config"),
                    "inputs": "CircuitInputs::test_inputs() (wormhole/tests/test-helpers/src/lib.rs:25-59)", "degree_bits": int(bc.desc.degree_bits),
                    "rows": bc.row_counts(), "build_circuit_s": round(build_s, 3), "witness_generation_ms": round(wit_ms, 3), "witness_generation_reused_buffer_ms": round(wit_reuse_ms, 3),
                    "prove_ms": round(ms_, 4), "proofs_per_s": round(1e3 / ms_, 2), "proof_bytes": len(pfW),
                    "public_inputs_match_prover_tests_rs": bool(pub_.funding_amount == 10 ** 12 and pub_.nullifier == ci.public.nullifier and pub_.root_hash == ci.public.root_hash),
                    "note": "same statement / public inputs as the reference's circuit; rows, sigma and digest are this builder's own (parity unpinned)"}
            real_check = (bc, w_, p_, pfW)
            gcW.close()
            # the reference's own prover bench (wormhole/prover/benches/prover.rs:11-21, `prover_create_proof`) times WormholeProver::new(config)
            # + commit(test_inputs) + prove per iteration: circuit build included. Same sequence here (host circuit build, upload, witness, proof).
            its_ = []
            for _ in range(3):
                t0_ = time.perf_counter()
                bc2 = ffi_.BuiltCircuit(ffi_.QPC_WORMHOLE, zk_); g2_ = H.GpuCircuit(bc2.desc, device=local_rank)
                w2_, p2_ = bc2.witness(ci.fill_targets(), H.TEST_SALT_KEY)
                g2_.prove(w2_, p2_, salt_seed=78)
                its_.append(time.perf_counter() - t0_)
                g2_.close()
            # from CircuitInputs to proofs, witness generation included: 64 inputs through the aggregator's leaf path (host threads generate the
            # witnesses, pageable H2D, one qpgpu_prove_batch)
            pkg_ = importlib.import_module("zk-circuits_amd")
            cd_ = pkg_.WormholeCircuitData.get(pkg_.CircuitConfig.standard_recursion_zk_config() if zk_ else pkg_.CircuitConfig.standard_recursion_config(), local_rank)
            ins_ = []
            for i_ in range(64):
                c_ = th_.circuit_test_inputs(); c_.public.exit_account = inp_.BytesDigest(int(i_ + 1).to_bytes(8, "little") + bytes(24)); ins_.append(c_)
            agg_ = pkg_.WormholeProofAggregator(cd_, pkg_.TreeAggregationConfig(2, 6))
            agg_.prove_leaves(ins_); agg_.proofs_buffer = []                 # warm: per-stream workspaces, reusable witness buffers
            t0_ = time.perf_counter(); agg_.prove_leaves(ins_); dt_ = time.perf_counter() - t0_
            real["leaves_from_inputs"] = {"proofs": 64, "seconds": round(dt_, 4), "proofs_per_s": round(64 / dt_, 1),
                                          "note": "CircuitInputs -> witness generation on host threads -> H2D -> qpgpu_prove_batch"}
            real["new_commit_prove_ms"] = round(min(its_) * 1e3, 2)
            real["new_commit_prove_note"] = "WormholeProver::new + commit + prove per iteration as prover/benches/prover.rs does (host circuit build + upload + witness + proof)"
        except Exception as e:                                        # the extra must never take the headline down
            real = {"error": repr(e)}

    # ---- extra, NOT the headline: an aggregation tree over 64 leaves per GPU (BASELINE configs[2]/[3] flavour).
    # Leaves: one qpgpu_prove_batch of this preset. Nodes: aggregate_chunk (tree.rs:106-143) for real — every node circuit verifies its two
    # children in-circuit (libqpcircuit's recursive verifier on the leaf gate set, include/qpcircuit.h; 2^14 rows per node), its witness is
    # generated on host threads and each level is proved as one device batch (aggregator.RecursiveChunkProver); level by level as
    # tree.rs:56-103. The root proof is verified. Node circuits are this repo's own layout (qp-plonky2's recursive verifier is not on disk).
    # With N > 1 ranks the top levels exchange fixed-size proof blobs over RCCL (aggregate_to_tree_distributed).
    tree = None
    tree_leaves, tree_root, tree_desc_owner = None, None, None
    if args.batch == 1 and not args.no_tree and args.preset in ("A", "B", "W", "WN"):
        try:
            import importlib
            agg = importlib.import_module("zk-circuits_amd.aggregator")
            leaves_per_rank = 64

            class LeafData:                                       # what RecursiveChunkProver needs of the leaf circuit
                desc = sc.desc; proof_size = gc.proof_size
                def constants_sigmas_cap(self):
                    return gc.cap()
            depth = int(round(np.log2(leaves_per_rank * world)))
            cfg = agg.TreeAggregationConfig(2, depth)
            ptrs = [wit_dev[j % nwit].data_ptr() for j in range(leaves_per_rank)]
            pl = [pis_list[j % nwit] for j in range(leaves_per_rank)]
            np_ = agg.RecursiveChunkProver(LeafData(), device=local_rank)
            setup_ok = 1
            try:                                                  # untimed: build the node circuit of every level (once per circuit) and warm the workspaces
                t_build = time.perf_counter()
                warm = gc.prove_batch(ptrs[:2], pl[:2], salt_seed=seeds[0], wires_on_device=True)
                for lvl in range(depth):
                    warm = [np_([warm[0], warm[-1]], lvl).proof.to_bytes()] * 2
                t_build = time.perf_counter() - t_build
                # one untimed full-width tree over this rank's leaves: per-stream workspaces of the node circuits
                agg.aggregate_to_tree(gc.prove_batch(ptrs, pl, salt_seed=seeds[0], wires_on_device=True),
                                      agg.TreeAggregationConfig(2, int(round(np.log2(leaves_per_rank)))), np_)
            except Exception:
                setup_ok = 0
            if dist is not None:                                  # all ranks enter the collectives below, or none does
                t = torch.tensor([setup_ok], dtype=torch.int32, device=cdev); dist.all_reduce(t, op=dist.ReduceOp.MIN); setup_ok = int(t.item())
            if not setup_ok:
                raise RuntimeError("aggregation-tree setup failed on a rank")
            barrier(); t3 = time.perf_counter()
            leaf_proofs = gc.prove_batch(ptrs, pl, salt_seed=seeds[0], wires_on_device=True)
            barrier(); t4 = time.perf_counter()
            if dist is not None:
                root = agg.aggregate_to_tree_distributed(leaf_proofs, cfg, np_, np_.proof_size_of_level, device=(dev if backend == "nccl" else None))
            else:
                root = agg.aggregate_to_tree(leaf_proofs, cfg, np_)
            barrier(); t5 = time.perf_counter()
            tl, tt = t4 - t3, t5 - t4
            if dist is not None:
                t = torch.tensor([tl, tt], dtype=torch.float64, device=cdev); dist.all_reduce(t, op=dist.ReduceOp.MAX); tl, tt = float(t[0]), float(t[1])
            n_leaves = leaves_per_rank * world
            root_cd = np_.level(depth - 1, 2)[0]
            tree = {"leaves": n_leaves, "branching": 2, "levels": depth, "nodes": n_leaves - 1,
                    "leaf_stage_s": round(tl, 4), "leaf_proofs_per_s": round(n_leaves / tl, 1),
                    "tree_stage_s": round(tt, 4), "total_s": round(tl + tt, 4), "root_bytes": len(root.proof),
                    "root_public_inputs": int(root_cd.num_public_inputs), "node_circuit_rows": root_cd._built.row_counts()["rows_in_use"],
                    "node_circuit_degree_bits": int(root_cd.desc.degree_bits), "node_circuits_build_s": round(t_build, 2),
                    "note": "real recursion: every node verifies its children in-circuit (this repo's recursive verifier on the leaf gate set); "
                            "tree_stage_s = host witness generation (threads) + one device batch per level"}
            root_bytes = root.proof.to_bytes() if hasattr(root.proof, "to_bytes") else bytes(root.proof)
            if rank == 0:                                         # root_proof.circuit_data.verify(root_proof.proof) (tree.rs:214), by the product verifier
                rver = H.GpuVerifier(root_cd.desc, root_cd.constants_sigmas_cap(), device=local_rank)
                tree["root_verified"] = bool(rver.verify(root_bytes) is None)
                rver.close()
                root_pis = [int(x) for x in np.frombuffer(root_bytes[-8 * root_cd.num_public_inputs:], dtype="<u8")]
                mine = [int(x) for pf in leaf_proofs for x in np.frombuffer(pf[-128:], dtype="<u8")]
                tree["root_public_inputs_are_the_leaves"] = bool(root_pis[:len(mine)] == mine and len(root_pis) == len(mine) * world)   # rank 0's run comes first
            # the reference's aggregator bench matrix (wormhole/aggregator/benches/aggregator.rs:107-130): aggregate() over b^d ready-made leaf
            # proofs, (b, d) in {(2,1..5), (3..7,2)}, node circuits built beforehand (the reference rebuilds them per chunk inside the timing)
            if dist is None:
                matrix, t_m0 = [], time.perf_counter()
                for b_, d_ in [(2, 1), (2, 2), (2, 3), (2, 4), (2, 5), (3, 2), (4, 2), (5, 2), (6, 2), (7, 2)]:
                    if time.perf_counter() - t_m0 > 90:
                        matrix.append({"branching": b_, "depth": d_, "skipped": "matrix time box (90 s) reached"}); continue
                    cfg_ = agg.TreeAggregationConfig(b_, d_)
                    lv = leaf_proofs[:cfg_.num_leaf_proofs]
                    pr_ = np_ if b_ == 2 else agg.RecursiveChunkProver(LeafData(), device=local_rank)
                    r_ = agg.aggregate_to_tree(lv, cfg_, pr_)                          # builds (b > 2) and warms
                    torch.cuda.synchronize(); ta = time.perf_counter()
                    reps_ = 2
                    for _ in range(reps_):
                        r_ = agg.aggregate_to_tree(lv, cfg_, pr_)
                    torch.cuda.synchronize(); tb = (time.perf_counter() - ta) / reps_
                    matrix.append({"branching": b_, "depth": d_, "leaves": cfg_.num_leaf_proofs, "nodes": sum(b_ ** k for k in range(d_)),
                                   "node_degree_bits": int(r_.circuit_data.desc.degree_bits), "aggregate_ms": round(tb * 1e3, 3)})
                    if pr_ is not np_:
                        pr_.close()                                                    # wider node circuits are used once: free their device memory
                tree["bench_matrix"] = matrix
            # the aggregator's intake check: all 64 leaf proofs through the product's batch verifier (device Merkle openings + host algebra)
            try:
                ver = H.GpuVerifier(sc.desc, gc.cap(), device=local_rank)
                okv, _ = ver.verify_batch(leaf_proofs)                                  # warm (worker threads, staging buffers sized for the batch)
                torch.cuda.synchronize(); tvs = []
                for _ in range(10):
                    tv = time.perf_counter(); ver.verify_batch(leaf_proofs); tvs.append(time.perf_counter() - tv)
                tvs.sort(); tv = tvs[len(tvs) // 2]
                tree["leaf_batch_verify"] = {"proofs": len(leaf_proofs), "accepted": int(sum(okv)), "seconds": round(tv, 4), "proofs_per_s": round(len(leaf_proofs) / tv, 1),
                                             "min_s": round(tvs[0], 4), "max_s": round(tvs[-1], 4), "calls": len(tvs), "note": "median of the calls"}
                ver.close()
            except Exception as e:
                tree["leaf_batch_verify"] = {"error": repr(e)[:160]}
            to_check += [("tree_leaf_first", leaf_proofs[0], 0, seeds[0]), ("tree_leaf_last", leaf_proofs[-1], (leaves_per_rank - 1) % nwit, seeds[0] + leaves_per_rank - 1)]
            tree_leaves, tree_root, tree_desc_owner = leaf_proofs, root_bytes, root_cd
            # The same leaves under a wide tree, TreeAggregationConfig(8, d): 64 leaves per rank are 8 level-0 nodes and one level-1 node, all local;
            # with 8 ranks one more level follows, fed by the single gather of the path (8 node proofs). Fewer, larger nodes: the blinding rows and
            # the power-of-two padding are paid per node, so 73 nodes of 2^16 rows cost a third of 511 nodes of 2^15.
            n_total = leaves_per_rank * world
            wide_depth = int(round(np.log(n_total) / np.log(8)))
            if 8 ** wide_depth == n_total:
                try:
                    cfg8 = agg.TreeAggregationConfig(8, wide_depth)
                    np8 = agg.RecursiveChunkProver(LeafData(), device=local_rank)
                    ok8 = 1
                    try:
                        t_b8 = time.perf_counter()
                        warm = list(leaf_proofs[:8])
                        for lvl in range(wide_depth):
                            warm = [np8(warm, lvl).proof.to_bytes()] * 8
                        t_b8 = time.perf_counter() - t_b8
                        agg.aggregate_to_tree(leaf_proofs, agg.TreeAggregationConfig(8, 2), np8)          # untimed: workspaces of the two local levels
                    except Exception:
                        ok8 = 0
                    if dist is not None:
                        t = torch.tensor([ok8], dtype=torch.int32, device=cdev); dist.all_reduce(t, op=dist.ReduceOp.MIN); ok8 = int(t.item())
                    if ok8:
                        barrier(); t6 = time.perf_counter()
                        if dist is not None:
                            root8 = agg.aggregate_to_tree_distributed(leaf_proofs, cfg8, np8, lambda lvl: np8.proof_size_of_level(lvl, 8), device=(dev if backend == "nccl" else None))
                        else:
                            root8 = agg.aggregate_to_tree(leaf_proofs, cfg8, np8)
                        barrier(); t8 = time.perf_counter() - t6
                        if dist is not None:
                            t = torch.tensor([t8], dtype=torch.float64, device=cdev); dist.all_reduce(t, op=dist.ReduceOp.MAX); t8 = float(t.item())
                        cd8 = np8.level(wide_depth - 1, 8)[0]
                        rb8 = root8.proof.to_bytes() if hasattr(root8.proof, "to_bytes") else bytes(root8.proof)
                        wide = {"branching": 8, "levels": wide_depth, "nodes": sum(8 ** k for k in range(wide_depth)), "tree_stage_s": round(t8, 4),
                                "total_s": round(tl + t8, 4), "node_circuit_rows": cd8._built.row_counts()["rows_in_use"], "node_circuit_degree_bits": int(cd8.desc.degree_bits),
                                "node_circuits_build_s": round(t_b8, 2), "root_bytes": len(rb8)}
                        if rank == 0:
                            v8 = H.GpuVerifier(cd8.desc, cd8.constants_sigmas_cap(), device=local_rank)
                            wide["root_verified"] = bool(v8.verify(rb8) is None); v8.close()
                            pis8 = [int(x) for x in np.frombuffer(rb8[-8 * cd8.num_public_inputs:], dtype="<u8")]
                            wide["root_public_inputs_are_the_leaves"] = bool(pis8[:len(mine)] == mine and len(pis8) == len(mine) * world)
                        tree["wide_tree"] = wide
                    np8.close()
                except Exception as e:
                    tree["wide_tree"] = {"error": repr(e)[:200]}
        except Exception as e:                                  # never let the extra measurement take the headline down
            tree = {"error": repr(e)[:300]}

    # ---- extra, NOT the headline: the reference's verifier bench (wormhole/verifier/benches/verifier.rs:13-32, `verifier_verify_proof`): build
    # the verifier from common.bin / verifier.bin and verify the reference's golden proof (wormhole/bench-data; the repo keeps byte-identical
    # copies as test fixtures under tests/golden). One proof per call as the reference's loop does, and 64 copies as one batch.
    golden_verify = None
    gdir = os.path.join(ROOT, "tests", "golden")
    if rank == 0 and args.batch == 1 and not args.no_extras and all(os.path.exists(os.path.join(gdir, f)) for f in ("common.bin", "verifier.bin", "proof.bin")):
        try:
            rd = lambda f: open(os.path.join(gdir, f), "rb").read()
            common_b, verifier_b, proof_b = rd("common.bin"), rd("verifier.bin"), rd("proof.bin")
            vo = verifier_b[:len(verifier_b) - len(common_b)] if verifier_b.endswith(common_b) and len(verifier_b) > len(common_b) else verifier_b
            t0_ = time.perf_counter()
            shape = H.ReferenceCircuitShape(common_b, vo); gv = H.GpuVerifier(shape.desc, shape.constants_sigmas_cap, device=local_rank)
            setup_ms = (time.perf_counter() - t0_) * 1e3
            ok1 = gv.verify(proof_b) is None                                 # the first call creates the handle's worker threads and staging buffers

            def timed(fn, reps):                                              # per-call wall times: median / min / max over `reps` calls
                ts = []
                for _ in range(reps):
                    t_ = time.perf_counter(); fn(); ts.append((time.perf_counter() - t_) * 1e3)
                ts.sort()
                return {"median_ms": round(ts[len(ts) // 2], 3), "min_ms": round(ts[0], 3), "max_ms": round(ts[-1], 3), "calls": reps}
            one = timed(lambda: gv.verify(proof_b), 50)
            batch64 = [proof_b] * 64
            okb, _ = gv.verify_batch(batch64)                                 # warm: staging buffers sized for the batch
            b64 = timed(lambda: gv.verify_batch(batch64), 20)
            one_ms, b64_ms = one["median_ms"], b64["median_ms"]
            bad = bytearray(proof_b); bad[-1] ^= 1
            golden_verify = {"proof": "wormhole/bench-data/proof.bin with common.bin / verifier.bin (the reference's golden proof)", "accepted": bool(ok1 and all(okb)),
                             "forged_public_input_rejected": bool(gv.verify(bytes(bad)) is not None), "verifier_setup_ms": round(setup_ms, 2),
                             "verify_ms": round(one_ms, 3), "batch64_ms": round(b64_ms, 3), "batch64_proofs_per_s": round(64 / (b64_ms * 1e-3), 1),
                             "verify_one": one, "verify_batch64": b64, "host_threads": available_cores(),
                             "note": "median over repeated calls on one handle (persistent host worker pool); verify_ms / batch64_ms are the medians"}
            gv.close()
        except Exception as e:
            golden_verify = {"error": repr(e)[:200]}

    # Poseidon throughput of the dominant kernel against the VALU issue ceiling. Both ceilings come from a committed
    # file: the first-principles bound (static ISA instruction counts x 2 or 4 issue cycles, tools/poseidon_issue_bound.py) and the
    # measured saturation of the same kernel body (tools/permute_rate.py on MI355X).
    valu = None
    if "merkle_leaves_colmajor" in kernels:
        d = sc.desc; s_ = 4 if d.zero_knowledge else 0
        widths = [d.num_wires + s_, d.num_challenges * (1 + d.num_partial_products) + s_, d.num_challenges * d.quotient_degree_factor + s_]
        perms = sum(-(-w // 8) for w in widths) * sc.N
        tot_ms = kernels["merkle_leaves_colmajor"]["total_ms"] / (kernels["merkle_leaves_colmajor"]["launches"] / 3.0)
        pps = perms / (tot_ms * 1e-3)
        try:
            ib = json.load(open(os.path.join(ROOT, "profiles", "r02_poseidon_issue_bound.json")))
        except Exception:
            ib = {}
        fp, sat = ib.get("first_principles_G_perm_per_s"), ib.get("measured_saturated_G_perm_per_s")
        valu = {"permutations_per_s": round(pps / 1e9, 3), "unit": "G perm/s",
                "first_principles_G_perm_per_s": fp, "frac_of_first_principles": round(pps / 1e9 / fp, 3) if fp else None,
                "saturated_G_perm_per_s": sat, "frac": round(pps / 1e9 / sat, 3) if sat else None,
                "valu_instr_per_permutation": ib.get("valu_instr_per_permutation"), "source": "profiles/r02_poseidon_issue_bound.json",
                "note": "integer-VALU issue bound = 1024 SIMDs x 64 lanes x 2.4 GHz / (2 N_full + 4 N_half) cycles per wave-permutation; "
                        "a single 2^17-leaf proof only offers 2 waves/SIMD, the saturated figure needs >= 8. Instruction count and ceilings are those of the "
                        "fast form of the partial rounds, which the leaf hash uses from 2^17 leaves up and in batches (csrc/poseidon_dev.h)"}

    # whole-proof algorithmic bytes B(n, s) against the HBM peak
    d = sc.desc; n_, N_ = sc.n, sc.N
    s_ = 4 if d.zero_knowledge else 0
    W_, R_, S_ = d.num_wires, d.num_routed, d.num_constants + d.num_routed
    Zp_, Q_ = d.num_challenges * (1 + d.num_partial_products), d.num_challenges * d.quotient_degree_factor
    alg_bytes = (8 * W_ * n_ + 16 * (W_ + Zp_) * n_ + 16 * (W_ + Zp_ + Q_ + 3 * s_) * N_ + 192 * N_ + 8 * (2 * R_ + Zp_) * n_
                 + 8 * (S_ + W_ + Zp_ + 2) * N_ + 16 * N_ + 64 * N_ + 8 * (S_ + W_ + Zp_ + Q_) * n_ + 8 * (S_ + W_ + Zp_ + Q_) * N_ + 16 * N_ + 37 * N_)
    nproofs = args.steps * args.batch * world
    whole = {"alg_bytes_per_proof": alg_bytes, "achieved_GBps": round(alg_bytes * nproofs / world / dt / 1e9, 2),
             "hbm_frac": round(alg_bytes * nproofs / world / dt / 8e12, 5), "hbm_floor_ms": round(alg_bytes / 8e12 * 1e3, 4),
             "note": "per GPU; B(n,s) from SURVEY.md §8(d); the proof is bound by integer VALU issue (Poseidon) and dependent-stage latency, not HBM"}
    out = {
        "metric": "wormhole proofs/s", "value": round(nproofs / dt, 3), "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64 (Goldilocks field, 64-bit modular integer)",
        "data": "synthetic" + (" (the reference's test inputs, CircuitInputs::test_inputs(), exit account = proof index; witness generated on the host by libqpcircuit)" if real_workload else ""),
        "config": {"workload": {"W": "single wormhole-circuit proof (the wormhole circuit built from the reference's definition, CircuitInputs::test_inputs()), zk config, degree 2^14 (BASELINE configs[1])",
                                "WN": "single wormhole-circuit proof (the wormhole circuit built from the reference's definition, CircuitInputs::test_inputs()), non-zk config, degree 2^13",
                                "A": "single wormhole-shaped leaf proof, non-zk, degree 2^13 (BASELINE configs[1] shape of the reference tests)",
                                "B": "single wormhole-shaped leaf proof, zk config, degree 2^14 (BASELINE configs[1])",
                                "V": "voting-shaped proof at 2^20 rows (BASELINE configs[4])",
                                "R": "recursion-shaped proof (all 14 gate types of the in-circuit verifier), non-zk, degree 2^13"}.get(args.preset, args.preset),
                   "preset": args.preset, "proofs_per_step": args.batch,
                   **({"batch_workload": "each step is one qpgpu_prove_batch of %d independent proofs, stream-pipelined on this GPU (BASELINE configs[2])" % args.batch} if args.batch > 1 else {}), "degree_bits": int(sc.desc.degree_bits),
                   "zero_knowledge": bool(sc.desc.zero_knowledge), "proof_bytes": gc.proof_size,
                   "host_cpus": pinned, "parallelism": "independent proofs, one process per GPU" + ("; RCCL all_gather of proof blobs after the steps" if world > 1 else "")},
        "ms_per_proof": round(dt / (args.steps * args.batch) * 1e3, 4),
        "rccl_ranks": world if (dist is not None and backend == "nccl") else 0, "gather": gather_path,
        "ms_per_proof_by_rank": [round(x, 4) for x in per_rank_ms],      # each rank's own K steps, exchange excluded (value/ms_per_step use the max-over-ranks wall time, exchange included)
        "handoff_fallbacks": int(L.qpgpu_handoff_fallbacks()),            # device->host hand-offs that fell back from publish+flag to copy+sync in this process (expected 0)
        "kernels_note": "the dominant group (merkle_leaves_colmajor) is timed inside the %d timed steps; every other group comes from an untimed instrumented pass of %d proofs before them" % (args.steps, min(5, max(2, args.steps))),
        "roofline": roofline, "whole_proof_hbm": whole, "valu_roofline": valu, "pipelined": pipelined, "aggregation_tree": tree, "wormhole_circuit": real, "reference_golden_proof_verify": golden_verify, "kernels": {k: {kk: (round(vv, 5) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in kernels.items()},
    }

    # ---- CPU baseline (rank 0, N == 1 only): the oracle restatement ("port"), OpenMP on the host cores, bounded sample
    if args.preset == "V" and not args.no_cpu_baseline:
        args.no_cpu_baseline = True      # one CPU proof at 2^20 rows is minutes of work; outside the bounded-sample budget
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.binding import oracle as _oracle, OracleCircuit        # the checker: used in this leg only
        oc = OracleCircuit(sc.desc)
        cores = min(available_cores(), _oracle().orc_max_threads())
        w0, p0 = sc.witness(seeds[0])
        zk = bool(sc.desc.zero_knowledge)
        salts = sc.salts(seeds[0]) if zk else None
        # timing: the CPU side as fast as it can be — the derived fast partial rounds (the form upstream's CPU prover uses). The byte checks
        # below run on the oracle's primary form, the literal rounds, which shares no derived table with the product.
        _oracle().orc_set_poseidon_fast(1)
        oc.prove(w0, p0, salts, threads=cores)       # warm (twiddle tables, page faults)
        t1 = time.perf_counter()
        for _ in range(args.cpu_proofs):
            cpu_pf_fast = oc.prove(w0, p0, salts, threads=cores)
        cdt = time.perf_counter() - t1
        _oracle().orc_set_poseidon_fast(0)
        cpu_pf = oc.prove(w0, p0, salts, threads=cores)
        assert cpu_pf == cpu_pf_fast, "the oracle's two forms of the permutation disagree"
        out["cpu_baseline"] = {"value": round(args.cpu_proofs / cdt, 4), "unit": "proofs/s", "cores": cores, "kind": "port",
                               "sample": "%d proofs of the same circuit/witness by oracle/ (C++ restatement of qp-plonky2's prover, OpenMP; timed with the fast form of Poseidon's partial rounds, byte-checked on the literal rounds); "
                                         "threads = cgroup CPU share of the box; the Rust reference cannot be built here (no cargo, crates not on disk)" % args.cpu_proofs,
                               "ms_per_proof": round(cdt / args.cpu_proofs * 1e3, 2)}
        out["bit_exact_vs_cpu"] = bool(cpu_pf == gc.prove(None, p0, salt_seed=seeds[0], wires_dev_ptr=wit_dev[0].data_ptr()))
        # every extra above is byte-checked too: first and last proof of each batch against the oracle's bytes for the same
        # (witness, salt seed); all tree leaves and the tree root through the oracle verifier
        bx = {}
        for label, pf, widx, sseed in to_check:
            wk, pk = sc.witness(seeds[widx])
            bx[label] = bool(pf == oc.prove(wk, pk, sc.salts(sseed) if zk else None, threads=cores))
        if tree_leaves is not None:
            bx["tree_leaves_verified"] = int(sum(1 for pf in tree_leaves if oc.verify(pf) is None))
            ocR = OracleCircuit(tree_desc_owner.desc)
            bx["tree_root_verified"] = bool(ocR.verify(tree_root) is None)
        if real_check is not None:                                     # the real-circuit proof against the oracle's bytes for the same witness and salts
            bc, w_, p_, pfW = real_check
            ocW = OracleCircuit(bc.desc)
            bx["wormhole_circuit"] = bool(pfW == ocW.prove(w_, p_, sc.salts(77) if bc.desc.zero_knowledge else None, threads=cores) and ocW.verify(pfW) is None)
        out["bit_exact_batch"] = bx
    if rank == 0:
        json_out.write(json.dumps(out) + "\n"); json_out.flush()
    gc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
