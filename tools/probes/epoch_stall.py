"""Where does the one-off ~70 ms stall of one epoch of the reference's loop come from?  Six epochs of examples/mnist_run.train
(trainer path) twice, with (PRE=E) and without (PRE=none) a config-E measurement — 10 GB of buffers allocated and released — in front:
per epoch (graph launch on the host, next permutation drawn, read-back = the GPU's remaining time) in ms.  Measured: the stall appears
once, in the read-back of one early epoch (the GPU itself takes ~70 ms longer to finish that epoch's graph), only with PRE=E.  Round 5:
an idle pause of 0.3 s (SETTLE_S=0.3) with a collection and a stream sync in front of the loop does NOT absorb it — it still lands
on a random one of the first five epochs, once per process; no Python collection longer than 1 ms happens (epoch_stall_gc.py), no
HIP call longer than a graph launch shows in an API trace.  A GPU-side pause when a light load follows a heavy one (a power /
memory-clock state switch, by its size), not something the loop or the library does."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd.examples import mnist_run
torch.cuda.set_device(0)
lib = _lib.get()
pre = os.environ.get("PRE", "E")
if pre == "E":
    solo = bench.Clock(torch, None, 1)
    bench.config_e_object(solo)
settle = float(os.environ.get("SETTLE_S", "0"))          # SETTLE_S=0.3: does an idle pause absorb the stall?
if settle:
    import gc
    gc.collect()
    lib.stream_sync()
    time.sleep(settle)
(train_x, train_y), (test_x, test_y), source = mnist_run.prepare_dataset("/nonexistent", n_train=50000, n_test=10000)
for rep in range(2):
    np.random.seed(0)
    stats = []
    mnist_run.train(train_x, train_y, test_x, test_y, [256, 128], 6, 128, 1e-3, stats=stats, trainer=True)
    print(pre, rep, [(round(s["launch"]*1e3,2), round(s["prefetch"]*1e3,2), round(s["readback"]*1e3,2)) for s in stats])
