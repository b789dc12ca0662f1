"""Host BLAS thread pool vs the container's CPU quota.

numpy's OpenBLAS sizes its pool from the CPUs it can SEE (64 threads on a 256-CPU MI355X host); a container usually may USE far
fewer (cgroup CPU bandwidth: 16 CPUs per 100 ms period on the boxes this was measured on).  After any BLAS call the idle pool threads
spin for a while; 64 spinning threads burn a 16-CPU quota in a quarter of the period, and the kernel then stops EVERY thread of the
cgroup until the next period begins — the host thread that drives the GPU included.  Seen from the training loop this is one 35-80 ms
"pause" of an epoch shortly after the synthetic dataset's `x @ teacher` (profiles/r06_epoch_stall_root_cause.txt: the cgroup's
nr_throttled / throttled_usec counters move in exactly the paused epoch; with the pool limited to 8 threads they never move and no
epoch pauses).  The GPU is not involved: it finishes its graph on time, the host is not running to see it.

`fit_blas_pool_to_cpu_quota()` is what the example driver and bench.py call once at start; the library itself never touches the
pool.  Needs `threadpoolctl` (ships with scikit-learn); without it the call reports what it found and changes nothing."""
import os


def cpu_quota(root="/sys/fs/cgroup", proc_cgroup="/proc/self/cgroup"):
    """CPUs (float) this process's cgroup may use per scheduling period — the nearest ancestor with a quota, cgroup v2
    (`cpu.max`) or v1 (`cpu.cfs_quota_us` / `cpu.cfs_period_us`) — or None when there is no quota or it cannot be read.
    (`root`, `proc_cgroup`: where to look; the tests point them at a fake tree.)"""
    rel = "/"
    try:
        for line in open(proc_cgroup):
            parts = line.strip().split(":", 2)
            if len(parts) == 3 and parts[1] == "":
                rel = parts[2]
    except OSError:
        return None
    while True:
        d = root + ("" if rel == "/" else rel)
        try:
            quota = open(os.path.join(d, "cpu.max")).read().split()
            if len(quota) == 2 and quota[0] != "max":
                return int(quota[0]) / float(quota[1])
        except (OSError, ValueError):
            pass
        if rel in ("", "/"):
            break
        rel = os.path.dirname(rel)
    for d in (os.path.join(root, "cpu"), os.path.join(root, "cpu,cpuacct")):
        try:
            q = int(open(os.path.join(d, "cpu.cfs_quota_us")).read())
            per = int(open(os.path.join(d, "cpu.cfs_period_us")).read())
            if q > 0 and per > 0:
                return q / float(per)
        except (OSError, ValueError):
            pass
    return None


def usable_cpus():
    """min(CPUs of the affinity mask, the cgroup quota rounded down), at least 1"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q)))
    return max(1, n)


def blas_threads():
    """size of numpy's BLAS pool, or None when threadpoolctl is missing / no BLAS pool is loaded"""
    try:
        from threadpoolctl import threadpool_info
    except Exception:                                      # noqa: BLE001
        return None
    pools = [p for p in threadpool_info() if p.get("user_api") == "blas"]
    return pools[0]["num_threads"] if pools else None


def fit_blas_pool_to_cpu_quota(share=0.5):
    """Under a CPU quota: limit the BLAS pool to `share` of the CPUs this process may actually use (never more threads than it
    has now, at least one) — half by default, which leaves the rest of the quota to the threads that drive the GPU while idle pool
    threads spin.  Without a quota: only never more threads than CPUs in the affinity mask.
    Returns {"quota_cpus", "usable_cpus", "blas_threads_before", "blas_threads"} (None where unknown)."""
    before = blas_threads()
    out = {"quota_cpus": cpu_quota(), "usable_cpus": usable_cpus(), "blas_threads_before": before, "blas_threads": before}
    if before is None:
        return out
    want = max(1, min(before, int(out["usable_cpus"] * share) if out["quota_cpus"] is not None else out["usable_cpus"]))
    if want < before:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=want, user_api="blas")      # (not used as a context manager: the limit stays)
        out["blas_threads"] = blas_threads()
    return out
