"""configs[4] step, A/B inside ONE process.  Three forms of the single-GPU bf16 step on the same trainer and the same batches, in
alternating segments of STEPS steps, HIP events on the library stream (TNN_E_STEP is read at every step; back-to-back bench.py runs
differ by +-3 % on this pool — clock drift —, alternating segments repeat to +-3 us):
  default  17 launches: prep launch, one transpose launch for both operands of each dW product just in front of it, one bias launch
  ct       13 launches: the transposed operands written by the epilogues of the GEMMs that produce a / dz (TNN_E_STEP=ct)
  long     25 launches: the sequence of separate transposes / bias / loss / partial-sum launches (TNN_E_STEP=long)
Prints per-form median / min / max over the segments and the paired differences."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import bench
from tinynn_autograd_amd import _lib

STEPS = int(os.environ.get("AB_STEPS", "20"))
ROUNDS = int(os.environ.get("AB_ROUNDS", "12"))
FORMS = tuple(os.environ.get("AB_FORMS", "default,ct,long").split(","))
NAMES = {"default": "17 launches (shipped): prep, one transpose launch per dW, one bias launch",
         "ct": "13 launches: a^T / dz^T from the producing GEMMs' epilogues",
         "long": "25 launches: separate transposes / bias / loss launches"}

run = bench.FusedRun(bench.WIDTHS_E, 512, "mse", 2, dtype="bfloat16")
ev0, ev1 = _lib.Event(), _lib.Event()


def segment(form):
    os.environ.pop("TNN_E_STEP", None)
    if form != "default":
        os.environ["TNN_E_STEP"] = form
    for i in range(3):
        run.eager_step(i)
    ev0.record()
    for i in range(STEPS):
        run.eager_step(i)
    ev1.record()
    return ev0.elapsed_ms(ev1) / STEPS * 1e3


for f_ in FORMS:
    segment(f_)                                # warm-up of every form
res = {f_: [] for f_ in FORMS}
for r in range(ROUNDS):
    for form in (FORMS if r % 2 == 0 else FORMS[::-1]):
        res[form].append(segment(form))
os.environ.pop("TNN_E_STEP", None)
print("configs[4] step, %d rounds of %d-step segments alternating on one trainer (us per step)" % (ROUNDS, STEPS))
for f_ in FORMS:
    v = np.array(res[f_])
    print("  %-78s median %7.1f  min %7.1f  max %7.1f" % (NAMES[f_], np.median(v), v.min(), v.max()))
for f_ in FORMS:
    if f_ != "default" and "default" in res:
        d = np.array(res[f_]) - np.array(res["default"])
        print("  paired difference %-7s - default: median %6.1f us  (%.2f %% of %s), min %.1f, max %.1f"
              % (f_, np.median(d), 100.0 * np.median(d) / np.median(res[f_]), f_, d.min(), d.max()))
for f_ in FORMS:
    print("  segments %-8s" % f_, " ".join("%.0f" % v for v in res[f_]))
