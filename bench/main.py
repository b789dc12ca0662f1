"""The run itself: what `python bench.py ...` does after argument parsing (bench.py at the repository root is the entry the driver calls)."""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

from .common import GLOBAL_BATCH_D, PEAK_FP32_MFMA_TFLOPS, WIDTHS_A, WIDTHS_C, WIDTHS_E, _lib, tn    # noqa: F401
from .common import brief
from .clock import Clock, measure
from .runners import FusedRun, OpsRun, fixture_check, timed_rows_check
from .roofline import attach_gemm_traffic, box_object, dw_adam_roofline_in_step, latency_roofline, time_gemms, time_gemms_bf16
from .cpu import cpu_baseline
from .lines import config_e_object, epoch_loop_object, make_line
from .multi_gpu import collective_latency_table, collective_selftest, rccl_version_string, self_launch, topology_object


HOST_POOL = {}                                             # what fit_blas_pool_to_cpu_quota() found / did (main())


def single_gpu_secondary(line, args, widths, rows, kind, res, runner, solo, use_graph, comm):
    """Rank 0, N = 1: the roofline objects, the drop-in API paths, the epoch loop, configs[2] and configs[4], the CPU baseline —
    every one an extra measurement AFTER the line's own `value` exists."""
    if args.workload == "E":
        line["roofline"] = dw_adam_roofline_in_step(runner, widths, rows, res["ms_per_step"])
        line["gemm_roofline"] = time_gemms_bf16(widths, rows)
    elif args.workload == "C":
        line["roofline"] = time_gemms(widths, rows, reps=20)
        attach_gemm_traffic(line["roofline"], "C")
    else:
        line["roofline"] = latency_roofline(widths, rows, res, runner)
        line["roofline_gemm4096"] = None               # (key order of the line; measured below, BEHIND the epoch loop)
        if args.path == "fused" and args.rows is None and comm is None:
            paths = {}
            for name, graph in (("ops_eager", False), ("ops_graph", True)):
                r = OpsRun(widths, rows, kind, 16, graph=graph)
                paths[name] = brief(measure(solo, r, 20, 200, 3, args.min_ms, rows))
                del r
            paths["note"] = ("the same step on the drop-in Tensor/ops/Dense/ReLU/SoftmaxCrossEntropyLoss/Adam/Model API (the "
                             "reference's loop body, 4 launches per step like the trainer): issued from Python op by op (eager) / "
                             "recorded with tn.capture, 16 steps on their resident batches per hipGraph, and replayed (graph)")
            paths["host_modules"] = ("compiled ahead of time from the .py sources (tinynn-autograd_amd/_host_build.py)"
                                     if tn.host_modules_compiled() else "interpreted")
            paths["host_call_wrappers"] = ("%d of %d entry points called through generated C wrappers instead of ctypes "
                                           "(tinynn-autograd_amd/_fastcall_gen.py)" % (_lib.get().fast_calls, len(_lib._SIGNATURES)))
            line["paths"] = paths
            # (kept in front of the large configurations; the "pause" earlier rounds ordered around was CPU-quota throttling of the
            # host threads — profiles/r06_epoch_stall_root_cause.txt — and is gone with the BLAS pool sized at start)
            line["epoch_loop"] = epoch_loop_object(res["value"])
            line["roofline_gemm4096"] = time_gemms(WIDTHS_C, 512, reps=20)
            attach_gemm_traffic(line["roofline_gemm4096"], "C")
            c = FusedRun(WIDTHS_C, 512, "mse", 2, use_graph=use_graph)
            rc = measure(solo, c, 3, 20, 3, 0.0, 512)
            line["config_C"] = brief(rc, workload="configs[2]: Dense 4096-4096-4096 autoencoder, bs 512, sum-of-squares/m, Adam",
                                     algorithmic_gflop_per_step=85.8993,
                                     mfma_frac_of_whole_step=round(85.8993e9 / (rc["ms_per_step"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4))
            del c
            line["config_E"] = config_e_object(solo)
        if line.get("roofline_gemm4096") is None:
            line["roofline_gemm4096"] = time_gemms(WIDTHS_C, 512, reps=20)
            attach_gemm_traffic(line["roofline_gemm4096"], "C")
    if not args.no_cpu_baseline and args.workload != "E":
        line["cpu_baseline"] = cpu_baseline(widths, rows, kind, budget_s=8.0 if args.workload == "A" else 15.0, host_pool=HOST_POOL)


def dp_world1_objects(line, args, widths, kind, warmup, steps, solo, use_graph):
    """Rank 0, N = 1, AFTER every other measurement of the line: the data-parallel step forms with a one-rank communicator (every
    collective issued; what N > 1 runs per rank) at 128 and 1024 rows per rank — the MNIST net and the reference's own example net."""
    # the data-parallel step forms with a one-rank communicator (every collective issued; what N > 1 runs per rank), AFTER
    # every other measurement of this line
    os.environ["TNN_FORCE_COMM"] = "1"
    comm1 = None
    try:
        comm1 = tn.dist.init_from_env()
        used1 = "xgmi-p2p" if getattr(comm1, "_p2p", False) else "rccl"
        dp1 = {"transport": used1, "note": "one-rank communicator, both collectives issued; rows per rank as on the N > 1 curves"}
        for rows_b in (128, 1024):
            rb_run = FusedRun(widths, rows_b, kind, 32, 0, 1, comm1, True, use_graph=use_graph)
            dp1[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b), graph_captured=rb_run.chunk is not None)
            del rb_run
        line["dp_world1"] = dp1
        ex_widths = [784, 200, 100, 70, 30, 10]
        ex_dp = {"transport": used1}
        for rows_b in (128, 1024):
            rb_run = FusedRun(ex_widths, rows_b, kind, 32, 0, 1, comm1, True, use_graph=use_graph)
            ex_dp[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b), graph_captured=rb_run.chunk is not None)
            del rb_run
        line.setdefault("reference_example_net", {})["dp_world1"] = ex_dp
    except Exception as exc:                             # noqa: BLE001 - an extra object never costs the line
        line["dp_world1"] = "unavailable: %s: %s" % (type(exc).__name__, exc)
    finally:
        os.environ.pop("TNN_FORCE_COMM", None)
        if comm1 is not None and hasattr(comm1, "close"):
            comm1.close()


def data_parallel_primary(args, comm, clock, runner, widths, rows, kind, world, rank, warmup, steps, force_dp, emit, all_ranks,
                          p2p_alive, replicas_identical):
    """The primary measurement of a data-parallel line: the same runner timed on RCCL first (north_star's named transport), on the
    xGMI peer-to-peer path second; both are reported, and `value` follows a RULE, not a best-of-two.  Returns (result, transports)."""
    # Data-parallel run.  RCCL first (north_star's named transport), the xGMI peer-to-peer path second; both are
    # reported.  The latency path carries f32 sums up to its mapped capacity: config A's 0.94 MB arena, not C's
    # 134 MB or E's 1 GB — those go to RCCL whatever the transport's state.
    transports = {}
    arena_bytes = (int(runner.trainer.arena_size) + 1) * 4
    have_rccl = bool(getattr(comm, "_rccl", False))
    have_p2p = p2p_alive() and arena_bytes <= getattr(comm, "p2p_bytes", 0)
    res_rccl = res_p2p = None
    if have_rccl:
        if have_p2p:
            comm.set_p2p(False)
            runner.capture()
        res_rccl = measure(clock, runner, warmup, steps, args.repeats, args.min_ms, rows * world)
        transports["rccl"] = brief(res_rccl, replicas_identical=replicas_identical(runner),
                                   graph_captured=runner.chunk is not None)
    primary, used = res_rccl, "rccl"
    if have_p2p:
        # nothing below may cost the RCCL result: a watchdog emits the line as it stands (RCCL as `value`) and ends
        # the process with a NON-ZERO code if the peer-to-peer run does not come back (bounded spins make that a
        # 20 s affair per stuck barrier; a hard hang is what the timer is for)
        limit = int(os.environ.get("TNN_BENCH_P2P_TIMEOUT_S", "120"))
        partial = {"line": None}

        def give_up():
            if partial["line"] is not None:
                partial["line"]["exit_code"] = 3
                partial["line"]["config"]["collectives"]["xgmi_p2p"] = "did not finish in %d s" % limit
                emit(partial["line"])
            os._exit(3)
        dog = threading.Timer(limit, give_up)
        dog.daemon = True
        if res_rccl is not None and rank == 0:
            partial["line"] = make_line(args, widths, rows, kind, world, warmup, steps, res_rccl, runner,
                                        dict(transports, used="rccl"), force_dp)
        if res_rccl is not None:
            dog.start()
        comm.set_p2p(True)
        try:
            runner.capture()
            res_p2p = measure(clock, runner, warmup, steps, args.repeats, args.min_ms, rows * world)
        except Exception as exc:                              # noqa: BLE001 - a peer timeout raises on every rank (comm.check votes)
            sys.stderr.write("bench: the peer-to-peer leg raised: %s\n" % exc)
            res_p2p = None
        if res_p2p is None:
            dog.cancel()
            transports["xgmi_p2p"] = "failed (a peer barrier timed out); transport switched off"
            if res_rccl is None:
                raise SystemExit("bench: the peer-to-peer transport failed and no RCCL communicator exists")
            comm.set_p2p(False)
            runner.capture()
        else:
            alive = p2p_alive()
            try:
                verified = all_ranks(alive and comm.p2p_selftest(sizes=(235147, 4099, 65536), rounds=6))
            except Exception as exc:                          # noqa: BLE001 - every rank must reach the vote
                sys.stderr.write("bench: post-run peer-to-peer check raised: %s\n" % exc)
                verified = all_ranks(False)
            same = replicas_identical(runner)
            dog.cancel()
            transports["xgmi_p2p"] = brief(res_p2p, barrier_timed_out=not alive, verified_after_run=verified,
                                           replicas_identical=same, graph_captured=runner.chunk is not None)
            if verified and same and alive:
                primary, used = res_p2p, "xgmi-p2p"
            elif res_rccl is None:
                raise SystemExit("bench: the peer-to-peer transport failed its checks and no RCCL communicator exists")
    transports["used"] = used
    transports["rule"] = "value = xgmi_p2p when verified bit-exact after the run, no barrier timed out and replicas identical; else rccl"
    if primary is None:
        raise SystemExit("bench: no usable transport")
    return primary, transports


def scaling_curves(line, args, comm, clock, solo, res, widths, rows, kind, world, rank, warmup, steps, use_graph, transports,
                   replicas_identical):
    """Workload A: the three scaling definitions beside `value` — this N's weak / strong / 1024-rows-per-rank points, at N = 1 the
    single-GPU batch sizes in between and the reference's own example net, at N > 1 rank 0's single-GPU references and the
    speedups computed from them.  Adds its objects to `line` (rank 0)."""
    other_rows = None
    if world > 1:
        other_rows = 128 if args.scaling == "strong" else GLOBAL_BATCH_D // world
    point = {"global_batch": rows * world, "rows_per_rank": rows, "value": round(res["value"], 1),
             "ms_per_step": round(res["ms_per_step"], 5)}
    curves = {("strong_scaling" if (world > 1 and args.scaling == "strong") else "weak_scaling"): point}
    strong_note = ("strong scaling of configs[3] (global batch 1024 split over N ranks) is bounded by launch latency, not by "
                   "work: the per-rank step costs about the same number of dependent launches whatever its row count, so the "
                   "ceiling at N ranks is (single-GPU bs-1024 step) / (bs-1024/N sharded step incl. two collectives); see "
                   "DESIGN.md §7 for the measured per-row-count steps.  The weak curve (128 rows per rank) is reported beside it.")
    if world == 1:
        # N = 1 point of the strong curve: the whole global batch of config D on one GPU
        d1 = FusedRun(widths, GLOBAL_BATCH_D, kind, 32, 0, 1, None, False, use_graph=use_graph)
        r1 = measure(solo, d1, warmup, steps, 3, args.min_ms, GLOBAL_BATCH_D)
        curves["strong_scaling"] = brief(r1, global_batch=GLOBAL_BATCH_D, rows_per_rank=GLOBAL_BATCH_D,
                                         launches_per_step=d1.launches_per_step(), note=strong_note)
        del d1
        if not args.no_extras and args.rows is None:
            # the same net at the batch sizes in between (the per-rank batches of the strong curve at N = 4 / 2)
            between = {}
            for rows_b in (256, 512):
                rb_run = FusedRun(widths, rows_b, kind, 32, 0, 1, None, False, use_graph=use_graph)
                between[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b),
                                             launches_per_step=rb_run.launches_per_step())
                del rb_run
            curves["batch_sizes"] = between
            # the reference's OWN example net (examples/mnist/run.py:59-69), same batch size and optimizer: the trainer's
            # 2 L - 2 = 8 launch step (hidden widths padded to multiples of 16, generic merged head kernel); pinned against the reference
            # by tests/golden/traj_R_example.npz
            ex_widths = [784, 200, 100, 70, 30, 10]
            ex_run = FusedRun(ex_widths, 128, kind, 64, 0, 1, None, False, use_graph=use_graph)
            curves["reference_example_net"] = brief(measure(solo, ex_run, warmup, steps, 3, args.min_ms, 128),
                                                    widths="-".join(map(str, ex_widths)), rows=128,
                                                    launches_per_step=ex_run.launches_per_step())
            del ex_run
            ex_sizes = {}
            for rows_b in (256, 512, 1024):              # the generic merged head walking 2 / 4 / 8 blocks of 128 rows
                rb_run = FusedRun(ex_widths, rows_b, kind, 32, 0, 1, None, False, use_graph=use_graph)
                ex_sizes[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b),
                                              launches_per_step=rb_run.launches_per_step())
                del rb_run
            curves["reference_example_net"]["batch_sizes"] = ex_sizes
        curves["weak_scaling_1024"] = dict(curves["strong_scaling"], note="N = 1 point of the third curve (1024 rows per rank): "
                                           "the same measurement as strong_scaling's N = 1 point")
    elif other_rows != rows:
        other = FusedRun(widths, other_rows, kind, 64 if other_rows <= 256 else 32, rank, world, comm, False,
                         use_graph=use_graph)
        ro = measure(clock, other, warmup, steps, 3, args.min_ms, other_rows * world)
        curves["weak_scaling" if args.scaling == "strong" else "strong_scaling"] = brief(
            ro, global_batch=other_rows * world, rows_per_rank=other_rows, transport=transports["used"],
            replicas_identical=replicas_identical(other))
        del other
    if world > 1:
        # third curve: 1024 rows per rank (global batch 1024 N) — the definition under which the step is long enough for the
        # exchange to amortise (ceilings of the three curves)
        w1024 = FusedRun(widths, 1024, kind, 32, rank, world, comm, False, use_graph=use_graph)
        rw = measure(clock, w1024, warmup, steps, 3, args.min_ms, 1024 * world)
        curves["weak_scaling_1024"] = brief(rw, global_batch=1024 * world, rows_per_rank=1024, transport=transports["used"],
                                            replicas_identical=replicas_identical(w1024))
        del w1024
        # the single-GPU references of ALL curves, measured in THIS run on rank 0 while the others wait; each curve's
        # speedup is computed on its own definition (weak: 128 rows on one GPU; strong: the whole 1024 rows on one GPU)
        if rank == 0:
            for rows_1 in (128, GLOBAL_BATCH_D):
                d1 = FusedRun(widths, rows_1, kind, 64 if rows_1 <= 256 else 32, 0, 1, None, False, use_graph=use_graph)
                r1 = measure(solo, d1, warmup, steps, 3, args.min_ms, rows_1)
                curves["single_gpu_bs%d" % rows_1] = brief(r1, note="rank 0 alone, no communicator")
                del d1
            for name, rows_1 in (("weak_scaling", 128), ("strong_scaling", GLOBAL_BATCH_D), ("weak_scaling_1024", 1024)):
                if name in curves:
                    curves[name]["speedup_vs_n1"] = round(
                        curves[name]["value"] / curves["single_gpu_bs%d" % rows_1]["value"], 4)
            own = "strong_scaling" if args.scaling == "strong" else "weak_scaling"
            if line is not None:
                line["speedup_vs_n1"] = curves[own]["speedup_vs_n1"]
        comm.barrier()
    if "strong_scaling" in curves:
        curves["strong_scaling"].setdefault("note", strong_note)
    if line is not None:
        line.update(curves)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="A", choices=["A", "C", "E"])
    ap.add_argument("--path", default="fused", choices=["fused", "ops", "opsgraph"])
    ap.add_argument("--rows", type=int, default=None, help="rows per GPU for workload A (default 128; N>1: 1024/N)")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="N>1, workload A: which curve `value` is on (the other one is reported beside it)")
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--min-ms", type=float, default=50.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary objects (config_C, paths, ...)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--epoch-loop-only", action="store_true", help="N = 1: print just the epoch_loop object")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` with no launcher: become the launcher BEFORE anything touches the GPU
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])

    # stdout carries exactly ONE line, the JSON result: native libraries print there too (RCCL writes a version /
    # hostname banner to stdout when a communicator is created), so file descriptor 1 is pointed at stderr for the
    # whole run and the result goes to a saved duplicate of the original stdout — once, whoever gets there first.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    emit_lock, emitted = threading.Lock(), [False]

    def emit(obj):
        with emit_lock:
            if emitted[0] or obj is None:
                return
            emitted[0] = True
            os.write(result_fd, (json.dumps(obj) + "\n").encode())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # numpy's BLAS pool is sized from the CPUs the host SHOWS (64 threads); the container may use 16 — idle pool threads spinning
    # behind a matmul then get the whole cgroup throttled for the rest of the scheduling period (the "one-off 35-80 ms epoch" of
    # rounds 4-6: profiles/r06_epoch_stall_root_cause.txt).  Every rank takes its share of half the quota.
    from tinynn_autograd_amd.utils.host_threads import fit_blas_pool_to_cpu_quota
    HOST_POOL.update(fit_blas_pool_to_cpu_quota(share=0.5 / max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    if args.gpus != world:
        args.gpus = world                                  # the launcher's WORLD_SIZE is authoritative

    # ORDER MATTERS: torch first, libtnn_hip.so second (one HIP runtime per process).  torch itself is
    # only the control plane (gloo rendezvous / barrier) and the contract's torch.cuda.synchronize().
    import torch
    import torch.distributed                              # noqa: F401
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(int(os.environ.get("TNN_DEVICE", local_rank)))   # TNN_DEVICE: ranks sharing one GPU (tests)
    lib = _lib.get()                                    # binds LOCAL_RANK's GPU; raises without HIP
    assert tn.backend_name() == "hip-gfx950", "bench.py measures the HIP library only"
    comm = tn.dist.init_from_env() if (world > 1 or os.environ.get("TNN_FORCE_COMM") == "1") else None
    force_dp = comm is not None and world == 1
    if os.environ.get("TNN_BENCH_TEST_EXIT_RANK") == str(rank) and world > 1:
        os._exit(9)                                      # test hook: this rank dies after the rendezvous (tests/test_gpu_p2p.py)
    clock = Clock(torch, comm, world)
    solo = Clock(torch, None, 1)                         # rank-local measurements (no barrier, no max over ranks)
    use_graph = not args.no_graph
    line = None
    exit_code = 0

    if args.epoch_loop_only:
        emit({"epoch_loop": epoch_loop_object(float(os.environ.get("TNN_HEADLINE", "5.98e6")))})
        return 0
    if args.workload == "A":
        widths, kind = WIDTHS_A, "softmax_nll"
        if args.rows is not None:
            rows = args.rows
        elif world > 1 and args.scaling == "strong":
            if GLOBAL_BATCH_D % world:
                raise SystemExit("strong scaling splits the global batch of %d evenly: %d ranks do not" % (GLOBAL_BATCH_D, world))
            rows = GLOBAL_BATCH_D // world
        else:
            rows = 128
        steps = args.steps if args.steps is not None else 2000
        warmup = args.warmup if args.warmup is not None else 64
        n_batches = 64 if rows <= 256 else 32
    elif args.workload == "C":
        widths, rows, kind = WIDTHS_C, args.rows or 512, "mse"
        steps = args.steps if args.steps is not None else 50
        warmup = args.warmup if args.warmup is not None else 5
        n_batches = 2
    else:
        widths, rows, kind = WIDTHS_E, args.rows or 512, "mse"
        steps = args.steps if args.steps is not None else 20
        warmup = args.warmup if args.warmup is not None else 3
        n_batches = 2

    # ---------------------------------------------------------------- primary measurement
    transports = None
    if args.workload == "E":
        args.path = "fused"
        runner = FusedRun(widths, rows, kind, n_batches, rank, world, comm, force_dp, dtype="bfloat16")
    elif args.path == "fused":
        runner = FusedRun(widths, rows, kind, n_batches, rank, world, comm, force_dp, use_graph=use_graph)
    else:
        runner = OpsRun(widths, rows, kind, min(n_batches, 16), rank, world, comm, graph=args.path == "opsgraph")

    def replicas_identical(r):
        crc = r.params_crc()
        if world > 1:
            import torch.distributed as dist
            box = [None] * world
            dist.all_gather_object(box, crc)
            return bool(all(c == box[0] for c in box))
        return True

    def all_ranks(flag):
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([1 if flag else 0])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))
        return bool(flag)

    def p2p_alive():
        st = comm.p2p_status() if comm is not None and hasattr(comm, "p2p_status") else None
        return all_ranks(bool(st and st["enabled"] and not st["dead"]))

    selftest_before = None
    if comm is not None and args.path == "fused":
        selftest_before = collective_selftest(comm, world, rank, all_ranks)
    if comm is not None and args.path == "fused" and isinstance(runner, FusedRun):
        res, transports = data_parallel_primary(args, comm, clock, runner, widths, rows, kind, world, rank, warmup, steps, force_dp, emit,
                                                all_ranks, p2p_alive, replicas_identical)
    else:
        res = measure(clock, runner, warmup, steps, args.repeats, args.min_ms, rows * world)

    if comm is not None and transports is not None:
        comm.set_p2p(transports["used"] == "xgmi-p2p")       # everything below runs on the primary transport
    check = check_timed = None
    if args.path == "fused" and args.workload == "A":
        # the reference's fixtures exist at global batches 128 and 1024: a weak-scaling line at N = 2 / 4 (global batch 256 /
        # 512) checks the same trainer + transport at config D's split instead (1024 / N rows per rank, the step form its
        # strong_scaling point times)
        rows_chk = rows
        if world > 1 and rows * world not in (128, GLOBAL_BATCH_D) and GLOBAL_BATCH_D % world == 0:
            rows_chk = GLOBAL_BATCH_D // world
        check = fixture_check(widths, rows_chk, kind, rank, world, comm, force_dp, use_graph)
        if check is not None:
            check["ok"] = all_ranks(check["ok"])
            check["rows_per_rank"], check["global_batch"] = rows_chk, rows_chk * world
            check["step_form"] = ("single-GPU 2L - 2 launch step" if comm is None else
                                  "merged 2L - 2 launch data-parallel step, %d block(s) of <= 128 rows per rank" % ((rows_chk + 127) // 128))
        if comm is not None and not args.no_extras:
            # ... and the step form actually TIMED (its own rows per rank), against the single-GPU trainer on the whole batch
            check_timed = timed_rows_check(widths, rows, kind, rank, world, comm, force_dp, use_graph, torch)
            ok_t = all_ranks(check_timed["ok"] if check_timed is not None else True)
            if check_timed is not None:
                check_timed["ok"] = ok_t
    if rank == 0:
        line = make_line(args, widths, rows, kind, world, warmup, steps, res, runner, transports, force_dp)
        line["host_blas_pool"] = dict(HOST_POOL)          # (rank 0's; every rank takes its share of half the CPU quota)
        if check is not None:
            line["parity_vs_reference_fixture"] = check
            if not check["ok"]:
                exit_code = line["exit_code"] = 4            # a fast step with the wrong losses is not a result
        if args.path == "fused" and args.workload == "A" and check_timed is not None:
            line["parity_at_timed_rows"] = check_timed
            if not check_timed["ok"]:
                exit_code = line["exit_code"] = 4

    # ---------------------------------------------------------------- scaling curves (workload A)
    if args.workload == "A" and args.path == "fused" and not args.no_extras and args.rows is None:
        scaling_curves(line, args, comm, clock, solo, res, widths, rows, kind, world, rank, warmup, steps, use_graph, transports,
                       replicas_identical)

    # ---------------------------------------------------------------- what the communicator ran on (every line with one)
    if comm is not None and args.path == "fused" and not args.no_extras:
        after = collective_selftest(comm, world, rank, all_ranks)
        table = collective_latency_table(comm, clock)
        if line is not None:
            used = transports["used"] if transports else None
            line["multi_gpu"] = {
                "world": world, "rccl_version": rccl_version_string(), "topology": topology_object(torch),
                "selftest_before_timed_runs": selftest_before, "selftest_after_timed_runs": after,
                "collective_latency": table,
                "value_from": used,
                "why": (transports or {}).get("rule"),
                "ranks_share_a_device": os.environ.get("TNN_DEVICE") is not None,
            }
        if transports is not None:
            comm.set_p2p(transports["used"] == "xgmi-p2p")

    # ---------------------------------------------------------------- forced communicator at world 1: the step forms of N > 1
    if comm is not None and world == 1 and args.workload == "A" and args.path == "fused" and not args.no_extras and args.rows is None:
        dp_sizes = {}
        for rows_b in (256, 512, 1024):                      # the per-rank batches of the strong curve and of weak_scaling_1024
            rb_run = FusedRun(widths, rows_b, kind, 32, 0, 1, comm, True, use_graph=use_graph)
            dp_sizes[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b),
                                          graph_captured=rb_run.chunk is not None)
            del rb_run
        ex_widths = [784, 200, 100, 70, 30, 10]
        ex_dp = {}
        for rows_b in (128, 1024):
            rb_run = FusedRun(ex_widths, rows_b, kind, 32, 0, 1, comm, True, use_graph=use_graph)
            ex_dp[str(rows_b)] = brief(measure(solo, rb_run, warmup, steps, 3, args.min_ms, rows_b),
                                       graph_captured=rb_run.chunk is not None)
            del rb_run
        if line is not None:
            line["dp_world1_batch_sizes"] = dict(dp_sizes, note="the data-parallel step (both collectives issued, world 1) at the "
                                                 "per-rank batches of the strong curve (256 / 512) and of weak_scaling_1024; transport: %s"
                                                 % (transports["used"] if transports else "rccl"))
            line.setdefault("reference_example_net", {"widths": "-".join(map(str, ex_widths))})["dp_world1"] = ex_dp

    # ---------------------------------------------------------------- configs[4] (bf16, 8 GPUs) on the data-parallel line
    # (a peer-to-peer-only group carries the bandwidth-sized collectives on the transport's bulk path — but only two such ranks may
    # SHARE a GPU: the skinny bf16 GEMM's in-launch split-K hand-off needs its whole 256-workgroup grid resident, which eight
    # 8192-wide trainers time-slicing one device cannot give each other; measured: the bounded wait runs out, sticky fault 1)
    e_leg_ok = getattr(comm, "_rccl", False) or (getattr(comm, "p2p_bulk_bytes", 0) > 0 and
                                                 (os.environ.get("TNN_DEVICE") is None or world <= 2))
    if (comm is not None and e_leg_ok and args.workload == "A" and args.path == "fused"
            and not args.no_extras and args.rows is None and os.environ.get("TNN_BENCH_CONFIG_E", "1") != "0"):
        # never at the price of the line: a watchdog on EVERY rank emits the line as it stands and ends the process if the
        # extra measurement does not come back (it is the first time this step form meets real links)
        limit_e = int(os.environ.get("TNN_BENCH_CONFIG_E_TIMEOUT_S", "180"))
        state_e = {"note": "did not finish in %d s" % limit_e}

        def stop_e():
            # a hung sharded-optimizer measurement is NOT a successful run: the line is emitted as it stands, marked, and the
            # process ends with a code of its own (6) so that self_launch / the driver see the failure
            if line is not None:
                line["config_E"] = state_e["note"]
                line["exit_code"] = line.get("exit_code") or 6
            emit(line)
            os._exit(exit_code or 6)
        dog_e = threading.Timer(limit_e, stop_e)
        dog_e.daemon = True
        dog_e.start()
        if getattr(comm, "_rccl", False):
            comm.set_p2p(False)                              # bandwidth-sized messages: RCCL
        # (a peer-to-peer-only group — TNN_COMM=xgmi, ranks sharing a GPU — carries them on the transport's bulk path)
        try:
            obj_e = config_e_object(clock, rank, world, comm, force_dp)
        except Exception as exc:                             # noqa: BLE001
            # the other ranks may be inside a collective of the measurement: no vote is possible — wait for the watchdogs
            state_e["note"] = "failed on rank %d: %s" % (rank, exc)
            sys.stderr.write("bench: config_E %s\n" % state_e["note"])
            time.sleep(limit_e + 30)
            obj_e = state_e["note"]
        dog_e.cancel()
        if transports is not None:
            comm.set_p2p(transports["used"] == "xgmi-p2p")
        if line is not None:
            line["config_E"] = obj_e

    # ---------------------------------------------------------------- secondary objects (rank 0, N = 1)
    if line is not None and world == 1 and not args.no_extras:
        single_gpu_secondary(line, args, widths, rows, kind, res, runner, solo, use_graph, comm)

    if (rank == 0 and world == 1 and comm is None and line is not None and not args.no_extras and args.workload == "A"
            and args.path == "fused" and args.rows is None):
        dp_world1_objects(line, args, widths, kind, warmup, steps, solo, use_graph)
    if rank == 0 and world == 1 and line is not None and not args.no_extras:
        line["box"] = box_object(line)
    emit(line)
    if comm is not None:
        comm.barrier()
        if hasattr(comm, "close"):
            comm.close()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([exit_code])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exit_code = int(t.item())
    return exit_code
