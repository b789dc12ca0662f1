"""configs[4] step, A/B inside ONE process: the 3L + 1 = 13 launch form (mlp16_step_fused: transposes from the producing GEMMs'
epilogues, loss + operand preparation in one launch, every dX before the first dW, one launch for the four biases) against the
25-launch sequence it replaces (TNN_E_STEP=long, read at every step).  Segments of STEPS steps alternate between the two forms on the same trainer
and the same batches, HIP events on the library stream; back-to-back bench.py runs differ by +-3 % on this pool (clock drift),
alternating segments do not.  Prints per-form median / min over the segments and the paired difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import bench
from tinynn_autograd_amd import _lib

STEPS = int(os.environ.get("AB_STEPS", "20"))
ROUNDS = int(os.environ.get("AB_ROUNDS", "12"))

run = bench.FusedRun(bench.WIDTHS_E, 512, "mse", 2, dtype="bfloat16")
lib = _lib.get()
ev0, ev1 = _lib.Event(), _lib.Event()


def segment(form):
    if form == "long":
        os.environ["TNN_E_STEP"] = "long"
    else:
        os.environ.pop("TNN_E_STEP", None)
    for i in range(3):
        run.eager_step(i)
    ev0.record()
    for i in range(STEPS):
        run.eager_step(i)
    ev1.record()
    return ev0.elapsed_ms(ev1) / STEPS * 1e3


segment("fused"), segment("long")            # warm-up of both forms
res = {"fused": [], "long": []}
for r in range(ROUNDS):
    order = ("fused", "long") if r % 2 == 0 else ("long", "fused")
    for form in order:
        res[form].append(segment(form))
os.environ.pop("TNN_E_STEP", None)
f, l = np.array(res["fused"]), np.array(res["long"])
print("configs[4] step, %d rounds of %d-step segments alternating on one trainer (us per step)" % (ROUNDS, STEPS))
print("  fused (13 launches): median %.1f  min %.1f  max %.1f" % (np.median(f), f.min(), f.max()))
print("  long  (25 launches): median %.1f  min %.1f  max %.1f" % (np.median(l), l.min(), l.max()))
print("  paired difference long - fused: median %.1f us  (%.2f %% of the long form), min %.1f, max %.1f"
      % (np.median(l - f), 100.0 * np.median(l - f) / np.median(l), (l - f).min(), (l - f).max()))
print("  segments fused:", " ".join("%.0f" % v for v in f))
print("  segments long: ", " ".join("%.0f" % v for v in l))
