// Host cores this process may use, in one place for libqpgpu.so and libqpcircuit.so (the Python mirror is zk-circuits_amd/hostcpus.py).
// A GPU box hands out a share of its cores (cgroup CPU quota) behind a wide affinity mask, and the ranks of one node share the host:
//   usable = min(affinity mask, cgroup quota) / ranks on this node   (at least 1)
// ranks on this node = LOCAL_WORLD_SIZE (torchrun exports it; a host without it is one rank). QPGPU_HOST_THREADS overrides the result;
// QPGPU_CGROUP_CPU_MAX names another file in cgroup-v2 `cpu.max` format ("<quota> <period>" or "max <period>"; tests).
#pragma once
#include <sched.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace qp {

inline int usable_cpus_uncached() {
  if (const char* e = getenv("QPGPU_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
  int k = (int)std::thread::hardware_concurrency();
  cpu_set_t set; CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) k = a; }
  const char* path = getenv("QPGPU_CGROUP_CPU_MAX");
  bool got = false;
  if (FILE* f = fopen(path && *path ? path : "/sys/fs/cgroup/cpu.max", "r")) {
    char q[64]; long long per = 0;
    if (fscanf(f, "%63s %lld", q, &per) == 2 && per > 0) {
      got = true;
      if (strcmp(q, "max") != 0) { const long long c = (atoll(q) + per - 1) / per; if (c >= 1 && c < k) k = (int)c; }
    }
    fclose(f);
  }
  if (!got && !(path && *path)) {                                 // cgroup v1
    long long quota = -1, per = 0;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(f, "%lld", &quota) != 1) quota = -1; fclose(f); }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f, "%lld", &per) != 1) per = 0; fclose(f); }
    if (quota > 0 && per > 0) { const long long c = (quota + per - 1) / per; if (c >= 1 && c < k) k = (int)c; }
  }
  if (const char* e = getenv("LOCAL_WORLD_SIZE")) { const int r = atoi(e); if (r > 1) k /= r; }
  return k > 0 ? k : 1;
}

inline int usable_cpus() { static const int n = usable_cpus_uncached(); return n; }

}  // namespace qp
