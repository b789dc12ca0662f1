#!/usr/bin/env python3
"""The one-off 50-80 ms epoch of the reference's loop (epoch_stall.py: it appears once per process, in one of the first epochs, only
when a large configuration — config E: ~10 GB of buffers allocated and released — ran before) against the GPU's CLOCK STATES:
a second process that never touches the GPU samples the amdgpu sysfs nodes (current sclk / mclk / fclk / socclk level, busy
percentages, average power) about once per millisecond while this one runs the epochs; the epochs' wall-clock windows are then
laid over the samples.
    python3 tools/probes/epoch_stall_clocks.py > profiles/r06_epoch_stall_clocks.txt
Sampler mode (internal): epoch_stall_clocks.py --sample <out file> <stop file>"""
import glob
import os
import subprocess
import sys
import time

NODES = ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent", "mem_busy_percent")


def device_dirs():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        if os.path.exists(os.path.join(d, "pp_dpm_sclk")) or os.path.exists(os.path.join(d, "gpu_busy_percent")):
            out.append(d)
    return out


def current_level(text):
    """'0: 132Mhz\\n1: 2100Mhz *' -> '2100Mhz' (the starred level); a plain number -> itself"""
    for line in text.splitlines():
        if line.rstrip().endswith("*"):
            return line.split(":", 1)[1].replace("*", "").strip()
    return text.strip().replace("\n", "|")


def sample_loop(out_path, stop_path):
    dirs = device_dirs()
    files = []
    for d in dirs:
        for n in NODES:
            p = os.path.join(d, n)
            if os.path.exists(p):
                files.append((os.path.basename(os.path.dirname(d)) + "/" + n, p))
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_average")) + glob.glob(os.path.join(d, "hwmon", "hwmon*", "power1_input")):
            files.append((os.path.basename(os.path.dirname(d)) + "/power_uW", hw))
    with open(out_path, "w") as out:
        out.write("# nodes: %s\n" % " ".join(k for k, _ in files))
        if not files:
            out.write("# no amdgpu sysfs node is readable from this process\n")
            return
        while not os.path.exists(stop_path):
            t = time.time()
            vals = []
            for _, p in files:
                try:
                    with open(p) as f:
                        vals.append(current_level(f.read()))
                except OSError:
                    vals.append("?")
            out.write("%.6f %s\n" % (t, " ".join(v.replace(" ", "") for v in vals)))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--sample":
        return sample_loop(sys.argv[2], sys.argv[3])
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, root)
    tmp = os.environ.get("TMPDIR", "/tmp")
    samples, stop = os.path.join(tmp, "tnn_clock_samples.txt"), os.path.join(tmp, "tnn_clock_stop")
    for p in (samples, stop):
        if os.path.exists(p):
            os.remove(p)
    import numpy as np
    sampler = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--sample", samples, stop])   # never touches the GPU
    import torch
    import bench
    from tinynn_autograd_amd import _lib
    from tinynn_autograd_amd.examples import mnist_run
    torch.cuda.set_device(0)
    lib = _lib.get()
    marks = [("start", time.time())]
    pre, hold = os.environ.get("PRE", "E"), None
    if pre == "E":
        bench.config_e_object(bench.Clock(torch, None, 1))
        marks.append(("config E measured, its buffers released", time.time()))
    elif pre == "E_hold":
        # the same work, but the 8192-wide trainer (its ~10 GB of buffers) stays alive until the process ends
        hold = bench.FusedRun(bench.WIDTHS_E, 512, "mse", 2, dtype="bfloat16")
        bench.measure(bench.Clock(torch, None, 1), hold, 2, 6, 3, 0.0, 512)
        marks.append(("config E measured, its buffers KEPT", time.time()))
    elif pre in ("light", "light_nogap"):
        # no heavy compute: 1.5 s of the headline step's graphs back to back (sustained, latency-bound activity at ~100 W) —
        # "light": the dataset is generated on the host AFTERWARDS (an idle gap of a few 100 ms, as in bench.py);
        # "light_nogap": the dataset exists before the activity starts, the epochs follow it at once
        if pre == "light_nogap":
            dataset = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
        hold = bench.FusedRun(bench.WIDTHS_A, 128, "softmax_nll", 64)
        t_end = time.time() + 1.5
        while time.time() < t_end:
            for _ in range(50):
                hold.chunk.launch()
            lib.stream_sync()
        marks.append(("1.5 s of headline-step graphs replayed (light, sustained)", time.time()))
    elif pre == "alloc":
        # no compute at all: 10 GB of device arrays written once and released
        from tinynn_autograd_amd import device_array as da
        bufs = [da.zeros((1 << 28,), np.float32) for _ in range(10)]
        lib.stream_sync()
        del bufs
        marks.append(("10 GB allocated, zeroed and released (no compute)", time.time()))
    if pre == "light_nogap":
        (train_x, train_y), (test_x, test_y), source = dataset
    else:
        (train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
    marks.append(("dataset on the device", time.time()))
    epochs = []
    for rep in range(2):
        np.random.seed(0)
        stats = []
        t_rep = time.time()
        mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 6, 128, 1e-3, stats=stats, trainer=True)
        lib.stream_sync()
        # the loop's own per-epoch phases (s): graph launch on the host, next permutation drawn, read-back = the GPU's remaining time
        t = t_rep
        for e, s in enumerate(stats):
            dur = sum(v for k, v in s.items() if isinstance(v, float))
            epochs.append((rep, e, s))
        marks.append(("run %d of six epochs done" % rep, time.time()))
    with open(stop, "w") as f:
        f.write("stop")
    sampler.wait(timeout=10)
    rows = [l.split() for l in open(samples) if not l.startswith("#")]
    header = open(samples).readline().strip()
    print(header)
    print("# %d samples over %.2f s (%.2f ms apart on average)" % (len(rows), float(rows[-1][0]) - float(rows[0][0]) if rows else 0.0,
                                                               (float(rows[-1][0]) - float(rows[0][0])) / max(len(rows) - 1, 1) * 1e3 if rows else 0.0))
    for name, t in marks:
        print("# mark %-45s t = %.3f s" % (name, t - marks[0][1]))
    print("# per epoch (run, epoch): launch / prefetch / read-back ms — the stalled epoch is the one whose read-back is 40-80 ms")
    for rep, e, s in epochs:
        w = s.get("wall", (marks[0][1], marks[0][1]))
        print("#   run %d epoch %d  t = %.4f .. %.4f s: %s" % (rep, e, w[0] - marks[0][1], w[1] - marks[0][1],
                                                              "  ".join("%s %.2f" % (k, v * 1e3) for k, v in s.items() if isinstance(v, float))))
    # every CHANGE of any clock level / every 25th sample, relative time
    if rows:
        t0 = marks[0][1]
        names = header.split()[2:]
        cards = sorted(set(n.split("/")[0] for n in names))

        def mhz(v):
            try:
                return float(v.lower().replace("mhz", ""))
            except ValueError:
                return 0.0
        # the card this process runs on: the one whose sclk reading is highest over the run (the others idle near 100 MHz)
        best = max(cards, key=lambda c: max(mhz(r[1 + names.index(c + "/pp_dpm_sclk")]) for r in rows) if c + "/pp_dpm_sclk" in names else 0.0)
        cols = [i for i, n in enumerate(names) if n.startswith(best + "/")]
        print("# %d cards visible in sysfs; this process runs on %s (highest sclk reading); its nodes, every change of a level and every 50th sample:" % (len(cards), best))
        print("# t_rel_s  " + " ".join(names[i].split("/")[1] for i in cols))
        last = None
        for i, r in enumerate(rows):
            vals = [r[1 + j] for j in cols]
            key = tuple(v for j, v in zip(cols, vals) if "busy" not in names[j] and "power" not in names[j] and "sclk" not in names[j])
            if key != last or i % 50 == 0:
                print("%9.4f  %s" % (float(r[0]) - t0, " ".join(vals)))
                last = key
        lv = {}
        for j in cols:
            if "busy" in names[j] or "power" in names[j]:
                continue
            lv[names[j].split("/")[1]] = sorted(set(r[1 + j] for r in rows if marks[1][1] - 0.05 <= float(r[0])), key=mhz)
        print("# distinct readings of %s from the end of the large configuration to the end of the run:" % best)
        for k, v in lv.items():
            print("#   %-14s %s" % (k, " ".join(v) if len(v) <= 12 else "%s .. %s (%d values: an averaged reading)" % (v[0], v[-1], len(v))))


if __name__ == "__main__":
    main()
