"""utils/host_threads.py: the container's CPU quota is found (cgroup v2 nearest limited ancestor, cgroup v1), and the BLAS pool
is never left larger than what the process may use (profiles/r06_epoch_stall_root_cause.txt: a 64-thread pool under a 16-CPU
quota gets the whole process throttled)."""
import os

import numpy as np  # noqa: F401  (loads the BLAS pool threadpoolctl looks for)

from tinynn_autograd_amd.utils import host_threads as ht


def _tree(tmp_path, files, cgroup_line):
    for rel, text in files.items():
        p = tmp_path / "cg" / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(text)
    (tmp_path / "cg").mkdir(exist_ok=True)
    pc = tmp_path / "proc_cgroup"
    pc.write_text(cgroup_line)
    return str(tmp_path / "cg"), str(pc)


def test_v2_quota_of_the_nearest_limited_ancestor(tmp_path):
    # the process sits two levels down; its own level and its parent are unlimited, the namespace root carries the quota
    root, pc = _tree(tmp_path, {"cpu.max": "1600000 100000\n", "process_api/cpu.max": "max 100000\n",
                                "process_api/abc/cpu.max": "max 100000\n"}, "0::/process_api/abc\n")
    assert ht.cpu_quota(root, pc) == 16.0
    root, pc = _tree(tmp_path / "b", {"cpu.max": "1600000 100000\n", "job/cpu.max": "250000 100000\n"}, "0::/job\n")
    assert ht.cpu_quota(root, pc) == 2.5


def test_v2_without_a_quota_and_v1(tmp_path):
    root, pc = _tree(tmp_path, {"cpu.max": "max 100000\n"}, "0::/\n")
    assert ht.cpu_quota(root, pc) is None
    root, pc = _tree(tmp_path / "v1", {"cpu/cpu.cfs_quota_us": "800000\n", "cpu/cpu.cfs_period_us": "100000\n"},
                     "3:cpuset:/jobs\n1:cpu:/\n0::/\n")
    assert ht.cpu_quota(root, pc) == 8.0
    root, pc = _tree(tmp_path / "v1b", {"cpu/cpu.cfs_quota_us": "-1\n", "cpu/cpu.cfs_period_us": "100000\n"}, "1:cpu:/\n")
    assert ht.cpu_quota(root, pc) is None
    assert ht.cpu_quota(str(tmp_path / "missing"), str(tmp_path / "missing_file")) is None


def test_pool_is_fitted_and_never_grows():
    before = ht.blas_threads()
    out = ht.fit_blas_pool_to_cpu_quota()
    assert set(out) == {"quota_cpus", "usable_cpus", "blas_threads_before", "blas_threads"}
    assert out["usable_cpus"] >= 1 and out["usable_cpus"] <= (os.cpu_count() or 1)
    if before is not None:                                  # threadpoolctl present (it is in this image)
        assert out["blas_threads_before"] == before
        assert 1 <= out["blas_threads"] <= before
        assert out["blas_threads"] <= out["usable_cpus"]
        if out["quota_cpus"] is not None:
            assert out["blas_threads"] <= max(1, int(out["usable_cpus"] * 0.5))
        assert ht.fit_blas_pool_to_cpu_quota()["blas_threads"] == out["blas_threads"]     # idempotent
