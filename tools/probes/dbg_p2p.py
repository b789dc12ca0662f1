import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import dist, _lib
comm = dist.DeviceCommunicator(0, 1)
comm.enable_p2p(8 << 20)
for n in (1, 3, 1000, 5000, 235147, 1000):
    x = (np.arange(n) % 1000 + 1).astype(np.float32)
    d = tn.asarray(x)
    comm.allreduce(d)
    _lib.synchronize()
    y = np.asarray(d)
    bad = np.nonzero(y != x)[0]
    print("n", n, "bad", bad.size, bad[:10], y[bad[:10]], x[bad[:10]], comm.p2p_status())
