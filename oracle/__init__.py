"""TEST INFRASTRUCTURE — CPU oracle for the tinynn-autograd dense-MLP hot path.  NOT PART OF THE PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this package, and only
as the checker / the reported CPU baseline — never as the thing measured or shipped.

Contents
  ref_autograd.py  numpy restatement of core/tensor.py + core/ops.py (same op graph, same per-edge
                   recursive backward => the reference's 4x traversal, same float64 promotion)
  ref_nn.py        numpy restatement of Dense / ReLU / whole-batch softmax NLL / SGD / Adam / Model.step
  closed_form.py   closed-form fp64 MLP training step (dz = p - y/m ...) for sizes where the op-graph
                   oracle is too slow
  gen_golden.py    container-only: imports the REAL reference from /root/reference, checks the three
                   restatements above against it, and writes tests/golden/*  (the reference never travels)
  cpu_twin/        C++ twin of the C-ABI used to test the host logic without a GPU

Pinning status: PINNED.  gen_golden.py asserts ref_autograd/ref_nn/closed_form against the imported
reference (all 18 known-answer cases of test/test_autograd.py and seeded training trajectories), and the
committed fixtures in tests/golden/ carry the reference's own outputs; tests/test_oracle_golden.py
re-checks the oracle against those fixtures wherever the suite runs.
"""
