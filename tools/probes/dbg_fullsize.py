import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tinynn_autograd_amd as tn
from oracle.closed_form import ClosedFormMLP
from tinynn_autograd_amd.fused import MLPTrainer
rs = np.random.RandomState(4)
widths, m = [4096, 4096, 4096], 512
a = np.sqrt(6.0 / (4096 + 4096))
W = [rs.uniform(-a, a, (4096, 4096)).astype(np.float32) for _ in range(2)]
B = [np.zeros((1, 4096), np.float32) for _ in range(2)]
x = rs.rand(m, 4096).astype(np.float32)
trainer = MLPTrainer(widths, m, loss="mse", optimizer="adam", lr=1e-3)
trainer.set_parameters([{"w": W[i], "b": B[i]} for i in range(2)])
oracle = ClosedFormMLP(W, B, loss="mse", optimizer="adam", lr=1e-3)
xd = tn.asarray(x)
for step in range(2):
    loss = float(trainer.step(xd, xd))
    ref_loss, ref_out, gW, gb = oracle.step(x, x)
    print("step", step, "loss rel", abs(loss-ref_loss)/ref_loss)
    for l in range(2):
        g = np.asarray(trainer.grad_view(l, "w")); gbd=np.asarray(trainer.grad_view(l,"b"))
        print("  dW%d err/max %.2e   db%d err/max %.2e   |gW|max %.3e" % (l, np.abs(g-gW[l]).max()/np.abs(gW[l]).max(), l, np.abs(gbd-gb[l]).max()/np.abs(gb[l]).max(), np.abs(gW[l]).max()))
for l in range(2):
    p=np.asarray(trainer.param_view(l,"w")); print(" param maxdiff", np.abs(p-oracle.W[l]).max())
