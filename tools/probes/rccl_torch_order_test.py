"""Exit-time behaviour of the process when torch (bundled HIP 7.0 libs) and libtnn_hip.so (ROCm 7.2) + RCCL
share a process, by import order.  usage: rccl_torch_order_test.py {lib_then_torch|torch_then_lib|lib_then_torch_gloo}"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
mode = sys.argv[1]
if mode == "torch_then_lib":
    import torch
    import torch.distributed as dist
import tinynn_autograd_amd as tn
from tinynn_autograd_amd import _lib
from tinynn_autograd_amd.dist import RcclCommunicator
lib = _lib.get()
uid = RcclCommunicator.new_unique_id()
if mode.startswith("lib_then_torch"):
    import torch
    import torch.distributed as dist
if mode.endswith("gloo"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
c = RcclCommunicator(0, 1, uid)
d = tn.asarray(np.arange(8, dtype=np.float32))
c.allreduce(d)
a = tn.asarray(np.ones((64, 64), np.float32))
print(mode, np.asarray(d)[:3], float((a @ a).sum()), flush=True)
maps = open("/proc/self/maps").read()
print(sorted({l.split()[-1] for l in maps.splitlines() if "amdhip64" in l or "librccl" in l}), flush=True)
c.close()
if mode.endswith("gloo"):
    dist.destroy_process_group()
