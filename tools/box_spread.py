#!/usr/bin/env python3
"""Box-to-box spread of the bench line: one row per `python3 bench.py` run collected under gpurun_out/spread/run*.json (every
gpurun call lands on a fresh box of the pool).  The fractions are quoted against the spec peak AND against the box's own probe
(`box`, measured inside the same run).

    for i in 1 2 3 4 5 6 7 8; do gpurun -- 'mkdir -p gpurun_out/spread; python3 bench.py > gpurun_out/spread/run'$i'.json'; done
    python3 tools/box_spread.py > profiles/r05_box_spread.txt"""
import glob
import json
import os
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for path in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "spread", "run*.json"))):
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        continue
    d = json.loads(lines[-1])
    e, g, b = d.get("config_E", {}), d.get("roofline_gemm4096", {}), d.get("box", {})
    eg = e.get("gemm_roofline", {})
    ed = e.get("dw_adam_roofline", {})
    rows.append(dict(
        run=os.path.basename(path)[:-5], A_us=d["ms_per_step"] * 1e3, A_Msps=d["value"] / 1e6,
        R_us=d.get("reference_example_net", {}).get("ms_per_step", float("nan")) * 1e3,
        loop_ms=d.get("epoch_loop", {}).get("trainer", {}).get("steady_epoch_ms", float("nan")),
        dp1024_us=(d.get("dp_world1", {}).get("1024", {}).get("ms_per_step", float("nan")) if isinstance(d.get("dp_world1"), dict) else float("nan")) * 1e3,
        C_ms=d.get("config_C", {}).get("ms_per_step", float("nan")), g32_TF=g.get("achieved", float("nan")),
        g32_frac=g.get("frac", float("nan")), g32_box=g.get("frac_of_box", float("nan")),
        E_ms=e.get("ms_per_step", float("nan")), E_gemm_TF=eg.get("achieved", float("nan")),
        E_dw_frac=ed.get("frac", float("nan")), E_dw_box=ed.get("frac_of_box", float("nan")),
        box_f32=b.get("mfma_f32_tflops", float("nan")), box_bf16=b.get("mfma_bf16_tflops_random_operands", float("nan")),
        box_copy=b.get("copy_float4_gbs", float("nan")), box_4r3w=b.get("stream_4read_3write_gbs", float("nan"))))
cols = ["run", "A_us", "A_Msps", "R_us", "loop_ms", "dp1024_us", "C_ms", "g32_TF", "g32_frac", "g32_box", "E_ms", "E_gemm_TF", "E_dw_frac", "E_dw_box",
        "box_f32", "box_bf16", "box_copy", "box_4r3w"]
print("# one row per bench.py run (fresh box each): A = configs[1] step (us, M samples/s); R = the reference's own net (us/step);")
print("# loop_ms = the median replayed 50,000-row epoch of the reference's loop on the trainer path (ms); dp1024_us = the data-parallel step at 1024 rows per rank, one rank;")
print("# C = configs[2] step (ms); g32 = the five fp32 GEMMs of C (TFLOP/s, fraction of 157.3, fraction of the box's MFMA-only loop);")
print("# E = configs[4] step (ms), its GEMMs in aggregate (TFLOP/s), its dW + Adam launch against 8 TB/s and against the box's")
print("# 4-read / 3-write stream; box = the probe inside the same run (fp32 / bf16 MFMA-only TFLOP/s, float4 copy and 4R/3W GB/s)")
print(" ".join("%9s" % c for c in cols))
for r in rows:
    print(" ".join(("%9s" % r[c]) if c == "run" else ("%9.3f" % r[c]) for c in cols))
if len(rows) > 1:
    print(" ".join(("%9s" % "median") if c == "run" else ("%9.3f" % statistics.median([r[c] for r in rows])) for c in cols))
    print(" ".join(("%9s" % "min") if c == "run" else ("%9.3f" % min(r[c] for r in rows)) for c in cols))
    print(" ".join(("%9s" % "max") if c == "run" else ("%9.3f" % max(r[c] for r in rows)) for c in cols))
