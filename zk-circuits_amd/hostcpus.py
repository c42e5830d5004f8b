"""Host cores this process may use — the Python mirror of csrc/hostcpus.h, the ONE place host thread pools are sized from.

A GPU box hands out a share of its cores (cgroup CPU quota) behind a wide affinity mask, and the ranks of one node share the host:
    usable = min(affinity mask, cgroup quota) // ranks on this node      (at least 1)
ranks on this node = LOCAL_WORLD_SIZE (torchrun exports it). QPGPU_HOST_THREADS overrides the result; QPGPU_CGROUP_CPU_MAX names another
file in cgroup-v2 `cpu.max` format (tests). The reference sizes its pool with rayon's default (one thread per visible core,
wormhole/aggregator/src/circuits/tree.rs:92-103); on a shared box that over-subscribes, hence the caps here."""
import os


def _quota_cores(path=None):
    """Cores the cgroup CPU quota allows (rounded up), or None when there is no quota."""
    explicit = path or os.environ.get("QPGPU_CGROUP_CPU_MAX")
    try:
        with open(explicit or "/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota == "max" or int(period) <= 0:
            return None
        return max(1, -(-int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    if explicit:
        return None
    try:                                                        # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            return max(1, -(-quota // period))
    except (OSError, ValueError):
        pass
    return None


def usable_cpus(affinity=None, cpu_max_path=None, local_world=None):
    """min(affinity, cgroup quota) // ranks of this node. The keyword arguments replace what is read from the system (tests)."""
    env = os.environ.get("QPGPU_HOST_THREADS")
    if env and env.isdigit() and int(env) > 0 and affinity is None and cpu_max_path is None and local_world is None:
        return int(env)
    if affinity is None:
        try:
            affinity = len(os.sched_getaffinity(0))
        except AttributeError:
            affinity = os.cpu_count() or 1
    k = max(1, int(affinity))
    q = _quota_cores(cpu_max_path)
    if q is not None:
        k = min(k, q)
    if local_world is None:
        try:
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
        except ValueError:
            local_world = 1
    if local_world > 1:
        k //= local_world
    return max(1, k)


def witness_pool_shape(jobs, cores=None, inner_cap=8):
    """(outer, inner): `outer` witnesses generated at once x `inner` threads inside one witness (a circuit's independent segments),
    with outer * inner <= cores. One job takes all the cores it can use; several jobs share them, at least two at once."""
    cores = usable_cpus() if cores is None else max(1, int(cores))
    if jobs <= 1:
        return 1, max(1, min(inner_cap, cores))
    outer = max(1, min(jobs, max(2, cores // 2), cores))
    inner = max(1, min(inner_cap, cores // outer))
    return outer, inner
