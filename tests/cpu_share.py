"""How many host cores this process may actually use: min(scheduler affinity, cgroup CPU quota).  os.cpu_count() reports the machine's
logical CPUs (128 on a GPU box) whatever share a container was given (16 there): thread pools sized from it oversubscribe the share 8x,
and OpenMP's spin-waits then burn the quota the working threads need.  Used by the GPU tests' fp64 oracle evaluations and by
bench.py's cpu_baseline (which reports the cores it really had)."""
import os


def cgroup_quota():
    """CPU quota of this process's cgroup in cores (float), or None when unlimited / unknown."""
    try:                                            # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:                                            # cgroup v1
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            return q / p
    except (OSError, ValueError):
        pass
    return None


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    q = cgroup_quota()
    if q is not None:
        n = min(n, max(1, int(q + 0.5)))
    env = os.environ.get("UGN_HOST_CORES")         # (override: a box whose share is set some other way)
    if env:
        n = max(1, int(env))
    return max(1, n)
