"""Host-side sizing per NODE, not per process.

One process per GPU means eight ranks share one host.  The reference ran ONE process with 32 DataLoader workers for
all its GPUs (main_1v.py:124, under nn.DataParallel :158-165) and one sampler process per robot (kinect2grasp.py:160-173);
every per-process default here is therefore a share of a node-level budget:

    cpus()              CPUs this process may use: the scheduler affinity mask capped by the cgroup CPU quota
    local_world()       ranks on this node (torchrun's LOCAL_WORLD_SIZE; WORLD_SIZE on a single node; else 1)
    threads_per_rank()  cpus() // local_world(), at least 1 — host thread pools (the sampler's eig pool)
    workers_per_rank(n) a NODE-total DataLoader worker count split over the node's ranks (like --batch-size)
"""
import math
import os


def _cgroup_quota():
    """CPUs granted by the cgroup CPU controller (v2 ``cpu.max`` / v1 ``cpu.cfs_quota_us``), or None if unlimited."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, math.ceil(int(quota) / int(period)))
        return None
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0 and period > 0:
            return max(1, math.ceil(quota / period))
    except (OSError, ValueError):
        pass
    return None


def cpus():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    q = _cgroup_quota()
    return max(1, min(n, q) if q else n)


def local_world():
    for key in ("LOCAL_WORLD_SIZE", "WORLD_SIZE"):
        v = os.environ.get(key)
        if v:
            try:
                return max(1, int(v))
            except ValueError:
                pass
    return 1


def threads_per_rank(cap=8):
    return max(1, min(int(cap), cpus() // local_world()))


def workers_per_rank(node_total, ranks_on_node=None):
    """``--num-workers`` is the NODE total (the reference's single process had 32 for all GPUs, main_1v.py:124): each of
    the node's ranks gets an equal share, at least one worker when any were asked for; 0 stays 0 (load in-process)."""
    node_total = int(node_total)
    if node_total <= 0:
        return 0
    r = ranks_on_node if ranks_on_node else local_world()
    return max(1, node_total // max(1, int(r)))
