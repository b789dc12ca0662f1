import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd.dist import RcclCommunicator
mode = sys.argv[1]
lib = _lib.get()
c = RcclCommunicator(0, 1, RcclCommunicator.new_unique_id())
d = tn.asarray(np.arange(8, dtype=np.float32))
c.allreduce(d)
print(mode, np.asarray(d)[:3], flush=True)
if mode == "close":
    c.close()
elif mode == "close_shutdown":
    c.close(); del d; lib.shutdown()
elif mode == "osexit":
    c.close(); sys.stdout.flush(); os._exit(0)
