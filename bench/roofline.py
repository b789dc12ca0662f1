"""Roofline objects of the JSON line: GEMMs against the MFMA peaks, the bf16 dW + Adam launch against HBM (timed inside the step),
PMC traffic from profiles/, the latency model of the 4-launch step, what this box itself reaches."""

import argparse
import ctypes
import json
import math
import os
import sys
import threading
import time

import numpy as np

from .common import LAUNCH_BOUNDARY_US, PEAK_BF16_MFMA_TFLOPS, PEAK_FP32_MFMA_TFLOPS, PEAK_HBM_GBS, PEAK_HBM_TBS, PROFILE_ROUND, ROOT, _lib, da    # noqa: F401
from .common import events_us, gemm_list, step_algorithmic
from .runners import FusedRun


def time_gemms(widths, rows, reps=20):
    """Each fp32 GEMM of the step on operands shaped like the step's own (activations uniform in [0, 1), weights
    Xavier-uniform: the MFMA data path's power draw, and with it the sustained clock, depends on the values)."""
    lib = _lib.get()
    rs = np.random.RandomState(7)
    results, tot_flops, tot_us = [], 0.0, 0.0
    for name, ta, tb, M, N, K in gemm_list(widths, rows):
        lim = float(np.sqrt(6.0 / (K + N)))
        a = da.asarray(rs.rand(*((K, M) if ta else (M, K))).astype(np.float32))
        b = da.asarray(rs.uniform(-lim, lim, (N, K) if tb else (K, N)).astype(np.float32))
        c = da.empty((M, N), np.float32)
        lda, ldb = (M if ta else K), (K if tb else N)
        us = events_us(lambda: lib.gemm(ta, tb, M, N, K, 1.0, a._ptr, lda, b._ptr, ldb, 0.0, c._ptr, N, _lib.F32), reps)
        flops = 2.0 * M * N * K
        results.append({"gemm": name, "layout": "NT"[ta] + "NT"[tb], "M": M, "N": N, "K": K,
                        "us": round(us, 3), "tflops": round(flops / us / 1e6, 3)})
        tot_flops += flops
        tot_us += us
    achieved = tot_flops / tot_us / 1e6
    return {"bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": None,
            "kernel": "gemm_f32_mfma_kernel (v_mfma_f32_32x32x2_f32)",
            "algorithmic_gflop_per_step": round(tot_flops / 1e9, 4),
            "gemm_us_per_step": round(tot_us, 2), "per_gemm": results}


def time_gemms_bf16(widths, rows, reps=10):
    """The bf16 step's GEMMs in the K-contiguous form the bf16 trainer uses (tnn_gemm_bf16_nt)."""
    from tinynn_autograd_amd import bf16
    rs = np.random.RandomState(7)
    results, tot_flops, tot_us = [], 0.0, 0.0
    shapes = []
    for l in range(len(widths) - 1):
        shapes.append(("fwd%d" % l, rows, widths[l + 1], widths[l], np.uint16))
    for l in reversed(range(len(widths) - 1)):
        shapes.append(("dW%d" % l, widths[l], widths[l + 1], rows, np.float32))
        if l > 0:
            shapes.append(("dX%d" % l, rows, widths[l], widths[l + 1], np.uint16))
    cache = {}
    rot = 3          # operand sets per shape, used in turn: like the layers of the step, no call finds its weights in the
                     # 256 MB memory-side cache (one 8192 x 8192 bf16 matrix is 134 MB; a single re-used one would stay there)
    for name, M, N, K, out in shapes:
        key = (M, N, K, out)
        if key not in cache:
            # one host draw per operand; the other sets are device-side rescalings of it (different bits, same cost)
            a0 = da.asarray(rs.uniform(-1, 1, (M, K)).astype(np.float32))
            b0 = da.asarray(rs.uniform(-1, 1, (N, K)).astype(np.float32))
            ops = [(bf16.to_bf16(a0 * (0.75 ** i)), bf16.to_bf16(b0 * (0.75 ** i))) for i in range(rot)]
            del a0, b0
            for A, B in ops:
                bf16.gemm_nt(A, B, out_dtype=out)
            e0, e1 = _lib.Event(), _lib.Event()
            e0.record()
            for i in range(reps):
                A, B = ops[i % rot]
                bf16.gemm_nt(A, B, out_dtype=out)
            e1.record()
            rotating = e0.elapsed_ms(e1) / reps * 1e3
            A, B = ops[0]
            for _ in range(2):
                bf16.gemm_nt(A, B, out_dtype=out)
            e0, e1 = _lib.Event(), _lib.Event()
            e0.record()
            for _ in range(reps):
                bf16.gemm_nt(A, B, out_dtype=out)
            e1.record()
            cache[key] = (rotating, e0.elapsed_ms(e1) / reps * 1e3)
            del ops
        us, us_hot = cache[key]
        flops = 2.0 * M * N * K
        results.append({"gemm": name, "layout": "NT(bf16)", "M": M, "N": N, "K": K, "us": round(us, 2),
                        "tflops": round(flops / us / 1e6, 1), "us_same_operands": round(us_hot, 2),
                        "tflops_same_operands": round(flops / us_hot / 1e6, 1)})
        tot_flops += flops
        tot_us += us
    achieved = tot_flops / tot_us / 1e6
    return {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_MFMA_TFLOPS, 4), "traffic": None,
            "kernel": "sk::gemm_bf16_sk_kernel (256 x 128 tiles, K split over two workgroups, hand-off inside the launch) for the 512-row products, gemm_bf16_dma_kernel (128 x 128) for the dW shape; v_mfma_f32_32x32x16_bf16, fp32 accumulate, LDS-DMA operand rings",
            "operands": "%d sets per shape used in turn (weights come from HBM as in the step); *_same_operands: one set re-used" % rot,
            "algorithmic_gflop_per_step": round(tot_flops / 1e9, 2), "gemm_us_per_step": round(tot_us, 1),
            "per_gemm": results}


def time_dw_adam_bf16(widths, rows, reps=10):
    """configs[4]'s dominant kernel since Adam moved into the dW epilogues: gemm_bf16_dma_kernel<8, 2, false, true>
    (tnn_gemm_bf16_nt_adam) on the weight-gradient shape.  HBM-bound: per parameter it reads p, m, v (12 B) and writes
    p, m, v, the bf16 copy and its transpose (16 B); the operands add 2 x 2 B x rows / n per element."""
    from tinynn_autograd_amd import bf16
    rs = np.random.RandomState(9)
    M, N, K = widths[0], widths[1], rows
    A = bf16.to_bf16(rs.uniform(-1, 1, (M, K)).astype(np.float32))
    B = bf16.to_bf16((rs.uniform(-1, 1, (N, K)) * 1e-2).astype(np.float32))
    P, Mo, Vo = da.zeros((M, N)), da.zeros((M, N)), da.zeros((M, N))
    W16, WT16 = da.empty((M, N), np.uint16), da.empty((N, M), np.uint16)
    pows = da.asarray(np.array([0.5, 0.5, 0, 0]), dtype=np.float64)
    lib = _lib.get()

    def call():
        lib.gemm_bf16_nt_adam(M, N, K, A._ptr, K, B._ptr, K, None, P._ptr, Mo._ptr, Vo._ptr, W16._ptr, WT16._ptr,
                              1e-3, 0.9, 0.999, 1e-8, pows._ptr)
    us = events_us(call, reps)
    alg = 28.0 * M * N + 2.0 * (M + N) * K
    gbs = alg / us / 1e3
    traffic, src = None, None
    table, path = load_traffic_table()
    if table is not None:
        for name, per in table["kernels"].items():
            if "E" in per and name.startswith("gemm_bf16_dma_kernel<8, 2, false, true>"):
                traffic, src = per["E"]["fetch_bytes"] + per["E"]["write_bytes"], path
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
            "traffic": traffic,
            **({"traffic_unit": "HBM-side bytes per launch (PMC: FETCH_SIZE x 2 + WRITE_SIZE over bench.py --workload E, %s)" % src} if src else {}),
            "kernel": "gemm_bf16_dma_kernel<8, 2, false, true> (dW = a^T dz with Adam in the epilogue, %d x %d x %d)" % (M, N, K),
            "algorithmic_bytes": int(alg), "us": round(us, 1), "launches_per_step": len(widths) - 1,
            "mfma_tflops": round(2.0 * M * N * K / us / 1e6, 1),
            "model": "28 B per parameter (p, m, v read; p, m, v, bf16 copy, bf16 transpose written) + the operands; the "
                     "gradient itself never leaves the accumulators.  Timed here: the 28-B form (3 of the step's 4 launches); "
                     "the first layer's launch writes no [in, out] bf16 copy (nothing reads it: dX stops at the input) = 26 B"}


def in_step_launch_us(run, positions, reps=4):
    """HIP-event time of single launches INSIDE the step: launch k costs T(the step's launches 0 .. k) - T(launches 0 .. k - 1),
    each prefix replayed `reps` times back to back from one hipGraph (tnn_mlp_launch_window restricts tnn_mlp_step to a window
    of its primitive calls).  Unlike a stand-alone replay of one launch on one operand set, the launch finds the caches as the
    step leaves them: its operands written by the launch in front of it, everything older evicted by the step's own traffic."""
    lib, h = run.trainer._lib, run.trainer._h
    x, y = run.batches[0]
    prefix = {}
    try:
        for k in sorted(set(positions) | set(p + 1 for p in positions)):
            if k == 0:
                prefix[0] = 0.0
                continue
            lib.mlp_launch_window(h, 0, k, None)
            prefix[k] = events_us(lambda: lib.mlp_step(h, x._ptr, y._ptr, run.rows, None), reps)
    finally:
        lib.mlp_launch_window(h, 0, -1, None)
    return [prefix[p + 1] - prefix[p] for p in positions]


def dw_adam_roofline_in_step(run, widths, rows, ms_per_step):
    """config E's roofline object: the dW + Adam launch against 8 TB/s with its time taken INSIDE the step (in_step_launch_us);
    the stand-alone replay on one operand set — which stays in the memory-side cache and reads 5-20 % faster — is kept as
    isolated_*.  `run`: the bf16 FusedRun whose step is being reported."""
    # where the dW + Adam launches sit in the step's launch sequence (csrc/tnn_mlp.cpp mlp16_step_fused): L forward, the
    # loss / dz launch, then per layer, last first: transposes, dX (not for the first layer), dW + Adam; one bias launch
    L, k, dw_pos = len(widths) - 1, len(widths), []
    for l in reversed(range(L)):
        k += 1 + (1 if l > 0 else 0)
        dw_pos.append(k)
        k += 1
    in_step = in_step_launch_us(run, dw_pos) if run.launches_per_step() == k + 1 else None
    roof = time_dw_adam_bf16(widths, rows, reps=6)
    if in_step is None:
        roof["frac_of_step_time"] = round(roof["us"] * L / (ms_per_step * 1e3), 3)
        return roof
    us28 = float(np.mean(in_step[:-1])) if L > 1 else float(in_step[0])
    roof["isolated_us"], roof["isolated_achieved"], roof["isolated_frac"] = roof["us"], roof["achieved"], roof["frac"]
    roof["us"] = round(us28, 1)
    roof["achieved"] = round(roof["algorithmic_bytes"] / us28 / 1e3, 1)
    roof["frac"] = round(roof["achieved"] / PEAK_HBM_GBS, 4)
    roof["mfma_tflops"] = round(2.0 * widths[0] * widths[1] * rows / us28 / 1e6, 1)
    roof["in_step_us_per_layer_last_first"] = [round(v, 1) for v in in_step]
    roof["timed"] = ("inside the %d-launch step: T(launches 0 .. k) - T(launches 0 .. k - 1) with HIP events, the step restricted to "
                     "a prefix of its launches (tnn_mlp_launch_window); us = mean of the 28-byte launches (every layer but the first, "
                     "whose 26-byte launch is the last entry of in_step_us_per_layer_last_first)" % (k + 1))
    roof["frac_of_step_time"] = round(float(np.sum(in_step)) / (ms_per_step * 1e3), 3)
    return roof


def load_traffic_table():
    """HBM-side bytes per launch from the PMC passes committed under profiles/ (FETCH_SIZE x2 per the gfx950 correction
    + WRITE_SIZE, separate rocprofv3 --pmc passes of this same command; tools/traffic_from_pmc.py)."""
    for rnd in (PROFILE_ROUND, "r05", "r04", "r03", "r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_traffic.json" % rnd)
        if os.path.exists(path):
            return json.load(open(path)), os.path.relpath(path, ROOT)
    return None, None


def attach_gemm_traffic(roof, tag):
    table, src = load_traffic_table()
    if table is None:
        return
    total, algorithmic = 0, 0
    for gm in roof["per_gemm"]:
        akc, bkc = gm["layout"][0] == "N", gm["layout"][1] == "T"
        flags = "%s, %s" % ("true" if akc else "false", "true" if bkc else "false")
        hit = None
        for name, per in table["kernels"].items():
            if tag in per and name.startswith("gemm_f32_mfma_kernel<") and (", " + flags + ", true>") in name:
                hit = per[tag]
        if hit is None:
            return
        total += hit["fetch_bytes"] + hit["write_bytes"]
        if gm["layout"] == "TN":
            # in the profiled step the dW launches carry Adam in their epilogue (tnn_gemm_tn_adam): operands + p, m, v read
            # and written, and no gradient store
            algorithmic += 4 * (gm["M"] * gm["K"] + gm["K"] * gm["N"]) + 24 * gm["M"] * gm["N"]
        else:
            algorithmic += 4 * (gm["M"] * gm["K"] + gm["K"] * gm["N"] + gm["M"] * gm["N"])
    roof["traffic"] = int(total)
    roof["traffic_unit"] = ("bytes per step over the step's five GEMM launches, the two dW launches with their Adam epilogue "
                            "(PMC over bench.py --workload C --no-extras, %s)" % src)
    roof["algorithmic_bytes"] = int(algorithmic)


def step_traffic(tag):
    """HBM-side bytes of ONE whole step (every kernel of the step's graph) from the same table, or None."""
    table, src = load_traffic_table()
    if table is None or "steps" not in table or tag not in table["steps"]:
        return None, None
    return int(table["steps"][tag]["bytes_per_step"]), src


def latency_roofline(widths, rows, res, runner):
    """Config A/D: neither MFMA nor HBM bounds the step — the launch chain does."""
    flops, gemm_bytes, adam_bytes = step_algorithmic(widths, rows)
    step_us = res["ms_per_step"] * 1e3
    roof = {"bound": "latency", "unit": "ksteps/s", "achieved": round(1e3 / step_us, 3)}
    gem = time_gemms(widths, rows, reps=200)
    if isinstance(runner, FusedRun) and runner.comm is None:
        launches = runner.launches_per_step()
        per = runner.per_launch_us()
        floor_us = launches * LAUNCH_BOUNDARY_US
        roof.update({"peak": round(1e3 / floor_us, 3), "frac": round(floor_us / step_us, 4),
                     "launches_per_step": launches, "launch_boundary_us": LAUNCH_BOUNDARY_US,
                     "launch_floor_us_per_step": round(floor_us, 3), "step_us": round(step_us, 3),
                     "per_launch_us": per, "sum_per_launch_us": round(sum(per), 3),
                     "model": "peak = 1 / (launches x dependent-kernel boundary); each per_launch_us is that launch replayed "
                              "back to back (HIP events), i.e. boundary + kernel"})
    else:
        roof.update({"peak": None, "frac": None, "step_us": round(step_us, 3)})
    traffic, src = step_traffic("A")
    roof["traffic"] = traffic
    if src:
        roof["traffic_unit"] = "HBM-side bytes per step, all kernels of the step (PMC, %s)" % src
    if "per_launch_us" in roof and len(widths) >= 3:
        # the step's largest launch against the roofline that would bound it if anything but latency did: the first layer's
        # backward with the whole optimizer step in it (HBM: x, dz0, every parameter's p / m / v read and written)
        n_params = sum(widths[l] * widths[l + 1] + widths[l + 1] for l in range(len(widths) - 1))
        rest = n_params - (widths[0] * widths[1] + widths[1])
        alg = 4 * (rows * widths[0] + rows * widths[1]) + 24 * n_params + 4 * rest
        us = roof["per_launch_us"][-1]
        dom = {"kernel": "dense_bwd0_adam_kernel<4> (dW0 = x^T dz0 + db0 with Adam over the whole parameter arena in the launch)",
               "bound": "hbm", "achieved": round(alg / us / 1e3, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": round(alg / us / 1e3 / PEAK_HBM_GBS, 4), "traffic": None, "algorithmic_bytes": int(alg), "us": round(us, 3),
               "us_note": "HIP events over back-to-back replays of this launch alone (boundary + kernel)",
               "model": "x [rows, n_in] and dz0 [rows, n_1] read; p, m, v of every parameter read and written (24 B/param); the other "
                        "layers' gradients read (4 B/param); dW0 itself never stored"}
        table, tsrc = load_traffic_table()
        if table is not None:
            for name, per in table["kernels"].items():
                if "A" in per and name.startswith("dense_bwd0_adam_kernel<"):
                    dom["traffic"] = per["A"]["fetch_bytes"] + per["A"]["write_bytes"]
                    dom["traffic_unit"] = "HBM-side bytes per launch (PMC: FETCH_SIZE x 2 + WRITE_SIZE, %s)" % tsrc
        roof["dominant_kernel"] = dom
    roof.update({"algorithmic_bytes": int(gemm_bytes + adam_bytes), "algorithmic_gflop_per_step": round(flops / 1e9, 4),
                 "hbm_frac": round((gemm_bytes + adam_bytes) / (step_us * 1e-6) / (PEAK_HBM_TBS * 1e12), 4),
                 "mfma_frac_of_whole_step": round(flops / (step_us * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                 "gemm_frac": gem["frac"], "gemm_tflops": gem["achieved"], "gemm_us_per_step": gem["gemm_us_per_step"],
                 "per_gemm": gem["per_gemm"]})
    return roof


def box_probe():
    """tools/probes/bin/libtnn_probe.so (its own library: `make -C tinynn-autograd_amd/csrc probe`, built by
    __graft_entry__.build()) -> what this box's MFMA pipes, clocks and HBM do right now (~100 ms on the GPU)."""
    path = os.path.join(ROOT, "tools", "probes", "bin", "libtnn_probe.so")
    lib = ctypes.CDLL(path)
    lib.tnn_probe_box.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    lib.tnn_probe_box.restype = ctypes.c_int
    out = (ctypes.c_double * 10)()
    rc = lib.tnn_probe_box(out, 10)
    if rc:
        raise RuntimeError("tnn_probe_box failed with code %d" % rc)
    return {"mfma_f32_tflops": round(out[0], 1), "mfma_f32_clock_ghz": round(out[1], 3),
            "mfma_f32_tflops_step_like_operands": round(out[8], 1), "mfma_f32_clock_ghz_step_like_operands": round(out[9], 3),
            "mfma_bf16_tflops_random_operands": round(out[2], 1), "mfma_bf16_clock_ghz_random_operands": round(out[3], 3),
            "mfma_bf16_tflops_zero_operands": round(out[4], 1), "mfma_bf16_clock_ghz_zero_operands": round(out[5], 3),
            "copy_float4_gbs": round(out[6], 1), "stream_4read_3write_gbs": round(out[7], 1)}


def box_object(line):
    """What THIS box can do (libtnn_probe.so: MFMA-only loops with random / zero operands and their sustained clocks, a float4
    copy over 2 GiB), measured after everything else so that it does not disturb the timed runs — and every MFMA- or
    HBM-bound roofline object on the line gets `frac_of_box` beside its spec-peak `frac`: achieved / the same box's probe
    (bf16 against the random-operand loop: the chip clocks to its power budget and real data is not zeros)."""
    box = box_probe()
    box["note"] = ("MFMA-only loops: 8 waves per CU, 8 independent accumulators; spec peaks 157.3 (fp32) / 2500 (bf16 dense) "
                   "TFLOP/s, 8000 GB/s; frac_of_box on the roofline objects = achieved / this box's probe")

    def annotate(obj):
        if isinstance(obj, dict):
            if obj.get("bound") in ("mfma", "hbm") and isinstance(obj.get("achieved"), (int, float)):
                if obj["bound"] == "hbm":
                    ref = max(box["copy_float4_gbs"], box["stream_4read_3write_gbs"])
                else:
                    ref = box["mfma_bf16_tflops_random_operands"] if obj.get("peak") == PEAK_BF16_MFMA_TFLOPS else box["mfma_f32_tflops"]
                if ref:
                    obj["box_peak"] = ref
                    obj["frac_of_box"] = round(obj["achieved"] / ref, 4)
            for v in obj.values():
                annotate(v)
        elif isinstance(obj, list):
            for v in obj:
                annotate(v)
    annotate(line)
    return box
