"""How many host CPUs this process may actually use: the minimum of its affinity mask and its cgroup CPU quota.

A GPU box hands a rank a cgroup quota (e.g. ``cpu.max = 1600000 100000`` = 16 CPUs) on a host whose affinity mask shows every
hardware thread; torch then starts one OpenMP worker per hardware thread (128) and the CFS quota throttles all of them -- the
CPU oracle ran 5x slower with 128 threads than with 16 on such a box.  Used for the oracle / cpu_baseline legs, the test session
and the launcher's per-rank OpenMP budget."""
import os


def usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                                     # cgroup v2
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:                                                 # cgroup v1
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, n)
