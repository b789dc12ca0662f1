"""bench.py is the entry the driver calls; the measurement code lives in the bench/ package (round 6).  Host-only checks: the entry
and every module import without a GPU, the package exports what tools/probes use through `import bench`, and the pure-host
pieces of the line (the with / without-the-pause accounting of the epoch loop) do what their docstrings say."""

import ast
import os
import subprocess
import sys

from conftest import ROOT


def test_entry_is_argument_parsing_and_dispatch_only():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    defs = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))]
    assert not defs, "bench.py holds no measurement code of its own"
    assert len(src.splitlines()) < 120
    main_src = open(os.path.join(ROOT, "bench", "main.py")).read()
    longest = max(n.end_lineno - n.lineno + 1 for n in ast.parse(main_src).body if isinstance(n, ast.FunctionDef))
    assert longest < 300, "bench.main.main() keeps shrinking: %d lines" % longest


def test_package_imports_without_a_gpu_and_exports_what_the_probes_use():
    code = ("import bench, bench.clock, bench.runners, bench.roofline, bench.cpu, bench.lines, bench.multi_gpu, bench.main\n"
            "for n in ('Clock', 'measure', 'FusedRun', 'OpsRun', 'config_e_object', 'box_probe', 'events_us', 'WIDTHS_A', 'WIDTHS_E',\n"
            "          'epoch_loop_object', 'cpu_baseline', 'time_gemms', 'dw_adam_roofline_in_step', 'self_launch', 'main'):\n"
            "    assert hasattr(bench, n), n\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0 and "--workload" in r.stdout and "--gpus" in r.stdout


def test_all_epochs_is_reported_with_and_without_the_paused_epochs():
    sys.path.insert(0, ROOT)
    from bench.lines import all_epochs_object
    stats = [{"steps": 0.0082, "train": 0.054}, {"steps": 0.0790, "train": 0.0791}, {"steps": 0.0082, "train": 0.0083},
             {"steps": 0.0081, "train": 0.0082}]
    total = sum(s["train"] for s in stats)
    out = all_epochs_object(stats, None, 4, 50000, total)
    assert out["paused_epochs"] == [1] and abs(out["train_ms"] - total * 1e3) < 1e-6
    assert abs(out["without_the_pause"]["pause_ms"] - (0.0790 - 0.0082) * 1e3) < 1e-6
    assert out["without_the_pause"]["value"] > out["value"]
    clean = all_epochs_object(stats[2:] * 2, None, 4, 50000, 4 * 0.0083)
    assert clean["paused_epochs"] == [] and "without_the_pause" not in clean
