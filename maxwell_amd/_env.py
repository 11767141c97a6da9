"""Host thread budget: honour the container's CPU quota instead of the visible core count."""
import os


def host_threads(default_cap=64):
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, int(int(quota) // int(period)))
    except Exception:
        pass
    try:
        return min(len(os.sched_getaffinity(0)), default_cap)
    except Exception:
        return min(os.cpu_count() or 1, default_cap)


def set_omp_threads():
    """Must run before libgomp is loaded (i.e. before any OpenMP library of this repo is opened)."""
    if "OMP_NUM_THREADS" not in os.environ:
        ranks_here = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)   # one process per GPU on this node
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(32, host_threads() // max(ranks_here, 1))))
    return int(os.environ["OMP_NUM_THREADS"])
