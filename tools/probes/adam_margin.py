#!/usr/bin/env python3
"""Measured margin under the Adam-parameter gate of the trajectory tests (tests/parity_suite.py: |p - reference| <= 0.1 lr after
the fixture's steps): runs the Adam trajectory cases on whichever library is loaded and prints the worst deviation of
each, in units of lr.
    python3 tools/probes/adam_margin.py                      (the HIP library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_suite as P  # noqa: E402

for case in ("traj_A_adam_fused", "traj_A_adam_generic_ops", "traj_A_adam_no_arena", "traj_D_bs1024", "traj_R_example_fused",
             "traj_R_example_generic_ops", "trainer_A_adam_eager", "trainer_A_adam_graph", "trainer_D_bs1024", "trainer_R_example_graph",
             "trainer_R_example_eager", "trainer_R_example_D_graph"):
    fn = getattr(P, case, None)
    if fn is None:
        continue
    fn()
print("worst |parameter - reference| / lr after the fixture's steps (gate: 0.1)")
for k, v in sorted(P.ADAM_MARGINS.items()):
    print("  %-40s %.4f" % (k, v))
