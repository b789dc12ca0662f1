"""N>1 path on CPU: world_size 2 and 4 over gloo (torch.distributed) with the CPU twin behind the ABI."""

import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(world, script, args=(), env_extra=None, tag=None):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1" if world > 2 else "2", PYTHONDONTWRITEBYTECODE="1")
        env.update(env_extra or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script)] + list(args), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out[-3000:])
        assert (tag or "%s rank %d/%d ok") % ((script[:-3], rank, world) if tag is None else (rank, world)) in out


def test_data_parallel_world2_matches_single_process_reference():
    """Op-level Model path and the trainer's Python-orchestrated sharded step over GlooCommunicator, world 2."""
    _run_world(2, "dp_worker.py")


@pytest.mark.parametrize("mode", ["D", "Rbucket", "Cbucket", "A", "E"])
def test_native_sharded_step_world4_over_gloo(mode):
    """The product's C++ data-parallel step (tnn_mlp_step_sharded) at world 4: strong-scaling config D with 1024/4 rows
    per rank against the reference's bs-1024 trajectory — single-collective and bucketed — and the bucketed path on
    config-C-small (one all-reduce per layer, Adam in bucket order)."""
    _run_world(4, "dp_hook_worker.py", [mode], {"TNN_BUCKET_BYTES": "1"} if mode.endswith("bucket") else None,
               tag="dp_hook_worker " + mode + " rank %d/%d ok")
