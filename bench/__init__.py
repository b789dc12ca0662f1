"""bench — the measurement code behind bench.py (the entry the driver calls stays bench.py at the repository root).

  common     Shared by every module of the bench package: the library imports, the peaks and widths BASELINE.json / the MI355X guide name,
  clock      The bench contract's timed region (barrier + stream sync + torch.cuda.synchronize() on both sides, max over ranks) and the
  runners    What is timed: runners that replay training steps from hipGraphs (whole-step trainer, op-level API), and the parity checks of
  roofline   Roofline objects of the JSON line: GEMMs against the MFMA peaks, the bf16 dW + Adam launch against HBM (timed inside the step),
  cpu        The CPU baseline leg: the numpy port of the reference (oracle/ref_nn.py) timed on this host — the only place besides the
  lines      Objects of the JSON line that are whole measurements of their own (config E, the epoch loop) and the line itself.
  multi_gpu  N > 1: launching the ranks, RCCL / topology description, the transports' self-tests and per-collective latencies.
  main       The run itself: what `python bench.py ...` does after argument parsing (bench.py at the repository root is the entry the driver calls).
"""

from .common import ROOT, PEAK_FP32_MFMA_TFLOPS, PEAK_BF16_MFMA_TFLOPS, PEAK_HBM_TBS, PEAK_HBM_GBS, LAUNCH_BOUNDARY_US, WIDTHS_A, WIDTHS_C, WIDTHS_E, GLOBAL_BATCH_D, PROFILE_ROUND, tn, _lib, da, synth_batches, build_net, gemm_list, step_algorithmic, events_us, brief    # noqa: F401
from .clock import Clock, measure    # noqa: F401
from .runners import Runner, FusedRun, _OpsGraph, OpsRun, fixture_check, timed_rows_check    # noqa: F401
from .roofline import time_gemms, time_gemms_bf16, time_dw_adam_bf16, in_step_launch_us, dw_adam_roofline_in_step, load_traffic_table, attach_gemm_traffic, step_traffic, latency_roofline, box_probe, box_object    # noqa: F401
from .cpu import cpu_model_name, cpu_baseline    # noqa: F401
from .lines import config_e_object, all_epochs_object, epoch_loop_object, make_line    # noqa: F401
from .multi_gpu import rccl_version_string, topology_object, collective_selftest, collective_latency_table, self_launch    # noqa: F401
from .main import main    # noqa: F401
