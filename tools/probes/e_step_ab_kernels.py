#!/usr/bin/env python3
"""Per-kernel view of tools/probes/e_step_ab.py under `rocprofv3 --kernel-trace`: every step of the trace is assigned to its form
(the prep-launch forms start their backward with mse_prep_bf16_kernel, the 25-launch one with mse_bf16_kernel; run e_step_ab.py
with AB_FORMS=ct,long or AB_FORMS=default,long so that "prep" names ONE form) and the two heavy
kernels are averaged per form and per POSITION in the step (what they were launched behind), with the gap in front of them.
    rocprofv3 --kernel-trace -d out -o ab -- python3 tools/probes/e_step_ab.py ; python3 tools/probes/e_step_ab_kernels.py out/.../ab_results.db"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    for key, s in (("mse_prep_bf16", "prep"), ("mse_bf16_kernel", "mse"), ("gemm_bf16_dma_kernel<8, 2, false, true>", "dW+Adam"),
                   ("gemm_bf16_sk_kernel", "skinny"), ("transpose2_bf16", "transpose"), ("transpose_bf16", "transpose"), ("bias_bf16_multi", "bias(all)"),
                   ("bias_bf16_kernel", "bias"), ("sum_partials", "sum")):
        if key in n:
            return s
    return "other"


ev = [(short(n), s, e) for n, s, e in rows]
# split into steps at each loss launch; a step = [loss launch ... next loss launch), its forward belongs to the previous chunk but
# the per-position statistics below only need "what came before"
stats = {"prep": collections.defaultdict(list), "long": collections.defaultdict(list)}
form = None
prev = None
for name, s, e in ev:
    if name == "prep":
        form = "prep"
    elif name == "mse":
        form = "long"
    if form is not None and name in ("dW+Adam", "skinny") and prev is not None:
        stats[form][(name, "behind " + prev[0])].append(((e - s) / 1e3, (s - prev[2]) / 1e3))
    prev = (name, s, e)
for form in ("prep", "long"):
    print("%s form" % form)
    tot = collections.defaultdict(lambda: [0.0, 0])
    for (name, pos), v in sorted(stats[form].items()):
        d = sorted(x[0] for x in v)
        g = sorted(x[1] for x in v)
        print("  %-8s %-18s n %5d  duration median %7.1f us  gap in front median %5.1f us" % (name, pos, len(v), d[len(d) // 2], g[len(g) // 2]))
        tot[name][0] += sum(d)
        tot[name][1] += len(d)
    for name, (t, n) in tot.items():
        print("  %-8s all positions: mean %.1f us over %d launches" % (name, t / n, n))
