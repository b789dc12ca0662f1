"""N > 1: launching the ranks, RCCL / topology description, the transports' self-tests and per-collective latencies."""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

from .common import ROOT, _lib, da    # noqa: F401


def rccl_version_string():
    try:
        v = ctypes.c_int(0)
        lib = ctypes.CDLL("librccl.so.1")
        lib.ncclGetVersion(ctypes.byref(v))
        n = v.value
        return "%d.%d.%d" % (n // 10000, (n // 100) % 100, n % 100)
    except Exception as exc:                              # noqa: BLE001
        return "unavailable (%s)" % type(exc).__name__


def topology_object(torch):
    """hipDeviceCanAccessPeer and hipExtGetLinkTypeAndHopCount for every pair of visible devices (rank 0; no device is
    initialised by either call).  link types: HSA_AMD_LINK_INFO_TYPE_* (0 HyperTransport, 1 QPI, 2 PCIe, 3 InfiniBand, 4 xGMI)."""
    out = {}
    try:
        n = torch.cuda.device_count()
        out["visible_devices"] = n
        out["can_access_peer"] = [[bool(i == j or torch.cuda.can_device_access_peer(i, j)) for j in range(n)] for i in range(n)]
        hip = None
        with open("/proc/self/maps") as f:
            for ln in f:
                if "libamdhip64" in ln:
                    hip = ctypes.CDLL(ln.split()[-1])
                    break
        if hip is not None and n > 1:
            lt, hops = [], []
            for i in range(n):
                lt.append([]); hops.append([])
                for j in range(n):
                    a, b = ctypes.c_uint32(0), ctypes.c_uint32(0)
                    rc = hip.hipExtGetLinkTypeAndHopCount(i, j, ctypes.byref(a), ctypes.byref(b)) if i != j else 0
                    lt[-1].append(int(a.value) if (i != j and rc == 0) else None)
                    hops[-1].append(int(b.value) if (i != j and rc == 0) else 0)
            out["link_type"], out["hops"] = lt, hops
            out["link_type_names"] = {"2": "PCIe", "4": "xGMI"}
    except Exception as exc:                              # noqa: BLE001 - diagnostics never cost the line
        out["error"] = "%s: %s" % (type(exc).__name__, exc)
    return out


def collective_selftest(comm, world, rank, all_ranks):
    """Both transports against known answers, voted over the ranks: RCCL — an all-reduce of rank-dependent integers (exact
    in f32) and the 2-float all-gather; the peer-to-peer transport — its own bit-exact self-test."""
    out = {}
    was = bool(getattr(comm, "_p2p", False))
    if getattr(comm, "_rccl", False):
        if was:
            comm.set_p2p(False)
        try:
            n = 235147
            v = da.asarray(((np.arange(n) % 97) + rank + 1).astype(np.float32))
            comm.allreduce(v)
            want = world * (np.arange(n) % 97).astype(np.float64) + world * (world + 1) / 2.0
            ok = bool(np.array_equal(np.asarray(v, dtype=np.float64), want))
            g = comm.allgather(da.asarray(np.array([rank + 0.5, 2.0 * rank], dtype=np.float32)))
            ok = ok and bool(np.array_equal(np.asarray(g), np.array([[r + 0.5, 2.0 * r] for r in range(world)], dtype=np.float32)))
        except Exception as exc:                          # noqa: BLE001
            sys.stderr.write("bench: RCCL self-test raised: %s\n" % exc)
            ok = False
        out["rccl"] = all_ranks(ok)
        if was:
            comm.set_p2p(True)
    if hasattr(comm, "p2p_status"):
        st = comm.p2p_status()
        if st and st["connected"] and not st["dead"]:
            comm.set_p2p(True)
            try:
                ok = bool(comm.p2p_selftest(sizes=(235147, 4099, 2), rounds=2))
            except Exception as exc:                      # noqa: BLE001
                sys.stderr.write("bench: peer-to-peer self-test raised: %s\n" % exc)
                ok = False
            out["xgmi_p2p"] = all_ranks(ok)
            comm.set_p2p(was)
    return out


def collective_latency_table(comm, clock, reps=100):
    """us per collective at this world size, replayed from one hipGraph of `reps` back-to-back calls (max over ranks):
    the 940,588-byte all-reduce of the gradient arena + loss slot (C1) and the 2-float statistics all-gather (C2), per
    transport."""
    lib = _lib.get()
    world = comm.world
    table = {}
    was = bool(getattr(comm, "_p2p", False))
    legs = []
    if getattr(comm, "_rccl", False):
        legs.append(("rccl", False))
    st = comm.p2p_status() if hasattr(comm, "p2p_status") else None
    if st and st["connected"] and not st["dead"]:
        legs.append(("xgmi_p2p", True))
    for name, p2p in legs:
        comm.set_p2p(p2p)
        row = {}
        try:
            buf = da.asarray(np.zeros(235147, np.float32))
            st2, out2 = da.asarray(np.array([1.0, 2.0], np.float32)), da.empty((world, 2), np.float32)
            for key, fn in (("allreduce_940588_B", lambda: comm.allreduce(buf)),
                            ("allgather_2_floats_per_rank", lambda: lib.allgather(st2._ptr, out2._ptr, 2, _lib.F32))):
                g = _lib.Graph()
                with g:
                    for _ in range(reps):
                        fn()
                g.launch()
                clock.fence()
                t0 = time.perf_counter()
                for _ in range(3):
                    g.launch()
                clock.fence()
                row[key] = round(clock.max_over_ranks(time.perf_counter() - t0) / (3 * reps) * 1e6, 2)
                del g
        except Exception as exc:                          # noqa: BLE001 - diagnostics never cost the line
            row["error"] = "%s: %s" % (type(exc).__name__, exc)
        table[name] = row
    comm.set_p2p(was)
    table["unit"] = "us per collective, %d back-to-back calls per hipGraph launch, max over ranks" % reps
    return table


def self_launch(n, argv):
    """`python3 bench.py --gpus N` without a launcher: THIS process never touches the GPU; it starts N fresh children of
    the same command, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as torch.distributed.run sets them,
    rendezvous on 127.0.0.1), relays rank 0's single JSON line and exits non-zero as soon as any child does.  Children are
    ended by their exact PIDs only."""
    import signal
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    limit = float(os.environ.get("TNN_BENCH_LAUNCH_TIMEOUT_S", "1500"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TNN_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    box = {"out": b""}

    def drain():
        box["out"] = procs[0].stdout.read()
    reader = threading.Thread(target=drain, daemon=True)
    reader.start()

    def end_all():
        for q in procs:
            if q.poll() is None:
                q.send_signal(signal.SIGTERM)
        t_end = time.time() + 10.0
        for q in procs:
            try:
                q.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                q.kill()
                q.wait()

    t0, rc = time.time(), 0
    while True:
        codes = [q.poll() for q in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            sys.stderr.write("bench: rank %d exited with code %d; ending the other ranks\n" % bad[0])
            rc = bad[0][1] if bad[0][1] > 0 else 1
            grace = time.time() + 5.0                       # a clean collective failure brings the others down by itself
            while time.time() < grace and any(q.poll() is None for q in procs):
                time.sleep(0.05)
            end_all()
            break
        if all(c == 0 for c in codes):
            break
        if time.time() - t0 > limit:
            sys.stderr.write("bench: the %d ranks did not finish within %.0f s\n" % (n, limit))
            end_all()
            rc = 124
            break
        time.sleep(0.05)
    reader.join(timeout=5.0)
    lines = [ln for ln in box["out"].decode(errors="replace").splitlines() if ln.strip()]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench: rank 0 printed no result line\n")
        rc = 5
    return rc
