"""Per-launch HIP-event times (tnn_mlp_launch_window) of the reference's own example net 784-200-100-70-30-10 (8 launches) beside the
benchmark net 784-256-128-10 (4 launches) at 128 / 256 / 1024 rows."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for rows in (128, 256, 1024):
    r = bench.FusedRun([784, 200, 100, 70, 30, 10], rows, "softmax_nll", 8, use_graph=True)
    print(rows, r.launches_per_step(), r.per_launch_us(100))
    r2 = bench.FusedRun(bench.WIDTHS_A, rows, "softmax_nll", 8, use_graph=True)
    print(rows, "bench net", r2.launches_per_step(), r2.per_launch_us(100))
