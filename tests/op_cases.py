"""Backend-neutral op cases: each case builds a tiny graph with a Tensor class `T` and an ops namespace
`O` (needs: exp, log, max, min, maximum, minimum, reshape, pad, flatten, clip, sum) and returns a dict of
named results.  The same functions are run against
    - the real reference (oracle/gen_golden.py, container only)  -> tests/golden/op_cases.json
    - the numpy oracle (oracle/ref_autograd.py)                    -> must equal the fixture
    - the device Tensor/ops (CPU twin here, HIP library on the GPU) -> must match the fixture.
Inputs are the known-answer inputs of the reference's test/test_autograd.py (cited per case) plus a few
broadcasting / tie / accumulation cases the hot path relies on.
"""

import numpy as np


def _v(t):
    return np.asarray(t.values if hasattr(t, "values") else t)


def _g(t):
    return np.asarray(t.grad)


def case_add_same_shape(T, O):            # test_autograd.py:11-18
    a, b = T([1, 3, 5], requires_grad=True), T([5, -2, -9], requires_grad=True)
    c = a + b
    c.backward([2, 2, 2])
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_add_broadcast_leading(T, O):     # test_autograd.py:20-27  (2,3)+(3,)
    a, b = T([[1, 3, 5], [2, 3, 0]], requires_grad=True), T([5, -2, -9], requires_grad=True)
    c = a + b
    c.backward([[1, 1, 1], [2, 2, 2]])
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_add_broadcast_keepdim(T, O):     # test_autograd.py:29-37  (2,3)+(1,3) = the bias gradient
    a, b = T([[1, 3, 5], [2, 3, 0]], requires_grad=True), T([[5, -2, -9]], requires_grad=True)
    c = a + b
    c.backward([[1, 1, 1], [2, 2, 2]])
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_sub_scalar_rhs(T, O):
    a = T([[1.5, -2.0], [0.25, 4.0]], requires_grad=True)
    c = 3.0 - a
    c.backward([[1, 2], [3, 4]])
    return {"out": _v(c), "ga": _g(a)}


def case_mul(T, O):                       # test_autograd.py:40-48
    a, b = T([1, 3, 5], requires_grad=True), T([5, -2, -9], requires_grad=True)
    c = a * b
    c.backward([2, 2, 2])
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_div(T, O):                       # test_autograd.py:51-59
    a, b = T([1, 2, 5], requires_grad=True), T([8, -2, -10], requires_grad=True)
    c = a / b
    c.backward([1, 1, 1])
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_pow_int_exponent(T, O):          # test_autograd.py:62-67
    a = T([1, -3, 5], requires_grad=True)
    c = a ** 3
    c.backward([2, 2, 2])
    return {"out": _v(c), "ga": _g(a)}


def case_pow_square(T, O):                # the (err ** 2) of test_autograd.py:120
    a = T([[0.5, -1.5], [2.0, 3.0]], requires_grad=True)
    c = (a ** 2).sum()
    c.backward()
    return {"out": _v(c), "ga": _g(a)}


def case_dot(T, O):                       # test_autograd.py:70-77
    a = T([[1, 3, 5], [5, -2, 9]], requires_grad=True)
    b = T([[9, 8, 9, 7], [4, 0, 3, 0], [0, 8, 2, 7]], requires_grad=True)
    c = a @ b
    c.backward([[1, 2, 3, 4], [4, 3, 2, 1]])
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_sum_all(T, O):                   # test_autograd.py:80-87
    a, b = T([1, 3, 5], requires_grad=True), T([5, -2, -9], requires_grad=True)
    c = (a + b).sum()
    c.backward(2)
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_sum_axis1(T, O):                 # the (p * labels).sum(1) of core/losses.py:28
    a = T([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]], requires_grad=True)
    c = a.sum(1)
    c.backward([1.0, -2.0])
    return {"out": _v(c), "ga": _g(a)}


def case_exp(T, O):                       # test_autograd.py:90-96
    a = T([1, 3, 5], requires_grad=True)
    c = O.exp(a)
    c.backward([1, 2, 3])
    return {"out": _v(c), "ga": _g(a)}


def case_neg(T, O):                       # test_autograd.py:99-105
    a = T([1, 3, 5], requires_grad=True)
    c = -a
    c.backward([1, 2, 3])
    return {"out": _v(c), "ga": _g(a)}


def case_maximum_ties(T, O):              # test_autograd.py:129-137 + a tie (ties -> first operand)
    a, b = T([1, 3, 5, 4], requires_grad=True), T([5, -2, 9, 4], requires_grad=True)
    c = O.maximum(a, b)
    c.backward([1, 2, 1, 7])
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_minimum_ties(T, O):              # test_autograd.py:140-148 + a tie
    a, b = T([1, 3, 5, 4], requires_grad=True), T([5, -2, 9, 4], requires_grad=True)
    c = O.minimum(a, b)
    c.backward([1, 2, 1, 7])
    return {"out": _v(c), "ga": _g(a), "gb": _g(b)}


def case_transpose(T, O):                 # test_autograd.py:151-165 (values instead of shapes only)
    data = np.arange(48, dtype=np.float64).reshape(2, 4, 6)
    a = T(data, requires_grad=True)
    c = a.T
    c.backward(np.arange(48, dtype=np.float64).reshape(6, 4, 2))
    g1 = _g(a).copy()
    a.zero_grad()
    d = a.transpose((2, 0, 1))
    d.backward(np.ones((6, 2, 4)))
    return {"out": _v(c), "ga": g1, "out2": _v(d), "ga2": _g(a)}


def case_max_all_and_axis0(T, O):         # test_autograd.py:168-179 incl. re-backward after zero_grad
    a = T([[1, 3, 5], [3, 7, -2]], requires_grad=True)
    m_all, m_ax0 = O.max(a, axis=None), O.max(a, axis=0)
    m_all.backward()
    g_all = _g(a).copy()
    a.zero_grad()
    m_ax0.backward([1, 1, 1])
    return {"out_all": _v(m_all), "out_ax0": _v(m_ax0), "g_all": g_all, "g_ax0": _g(a)}


def case_max_ties_all_get_grad(T, O):     # core/ops.py:229 — every tied element receives the gradient
    a = T([[2.0, 7.0], [7.0, 1.0]], requires_grad=True)
    c = O.max(a)
    c.backward(3.0)
    return {"out": _v(c), "ga": _g(a)}


def case_min_axis0(T, O):                 # core/ops.py:234-240
    a = T([[1, 3, 5], [3, 7, -2]], requires_grad=True)
    c = O.min(a, axis=0)
    c.backward([1, 2, 3])
    return {"out": _v(c), "ga": _g(a)}


def case_log(T, O):                       # test_autograd.py:182-189
    a = T([1, 3, 5], requires_grad=True)
    c = O.log(a)
    c.backward(np.array([1, 2, 3]))
    return {"out": _v(c), "ga": _g(a)}


def case_reshape(T, O):                   # test_autograd.py:192-198
    a = T([[1, 2, 3], [4, 5, 6]], requires_grad=True)
    c = O.reshape(a, (6,))
    c.backward(np.arange(6.0))
    return {"out": _v(c), "ga": _g(a)}


def case_pad(T, O):                       # test_autograd.py:201-209
    a = T([[1, 2, 3], [4, 5, 6]], requires_grad=True)
    c = O.pad(a, [(1, 0), (1, 0)])
    c.backward(np.arange(12.0).reshape(3, 4))
    return {"out": _v(c), "ga": _g(a)}


def case_flatten(T, O):                   # test_autograd.py:212-219
    a = T([[1, 2, 3], [4, 5, 6]], requires_grad=True)
    c = O.flatten(a)
    c.backward(np.arange(6.0))
    return {"out": _v(c), "ga": _g(a)}


def case_clip_relu(T, O):                 # test_autograd.py:222-229 + x == 0 (gradient 1, mask is >=)
    a = T([1, -3, 5, 0], requires_grad=True)
    c = O.clip(a, 0)
    c.backward(np.array([1, 2, 3, 4]))
    return {"out": _v(c), "ga": _g(a)}


def case_clip_both_bounds(T, O):          # core/ops.py:336-340 inclusive on both sides
    a = T([-2.0, -1.0, 0.5, 2.0, 3.0], requires_grad=True)
    c = O.clip(a, -1.0, 2.0)
    c.backward(np.array([1.0, 2.0, 3.0, 4.0, 5.0]))
    return {"out": _v(c), "ga": _g(a)}


def case_getitem_slice_and_gather(T, O):  # utils/data_iterator.py:27-33 patterns + the vjp scatter
    a = T(np.arange(12.0).reshape(4, 3), requires_grad=True)
    s = a[1:3]
    s.backward(np.ones((2, 3)))
    g_slice = _g(a).copy()
    a.zero_grad()
    idx = np.array([3, 0, 2])
    gth = a[idx]
    gth.backward(np.arange(9.0).reshape(3, 3))
    return {"slice": _v(s), "g_slice": g_slice, "gather": _v(gth), "g_gather": _g(a)}


def case_grad_accumulates(T, O):          # core/tensor.py:163 — two backward calls add up
    a = T([1.0, 2.0, 3.0], requires_grad=True)
    (a * 2.0).backward([1, 1, 1])
    (a * 3.0).backward([1, 1, 1])
    return {"ga": _g(a)}


def case_diamond(T, O):                   # a node used twice (the exps of core/losses.py:27)
    a = T([[0.5, -1.0], [2.0, 0.25]], requires_grad=True)
    e = O.exp(a)
    c = (e / e.sum()).sum(1)
    c.backward([1.0, 2.0])
    return {"out": _v(c), "ga": _g(a)}


def case_softmax_nll_expression(T, O):    # core/losses.py:24-32 written out on a 3x4 batch
    z = T([[1.0, 2.0, 0.5, -1.0], [0.0, 0.1, 0.2, 0.3], [3.0, -2.0, 1.0, 0.0]], requires_grad=True)
    y = T(np.eye(4)[[1, 3, 0]])
    m = 3
    exps = O.exp(z - z.max())
    p = exps / exps.sum()
    nll = -O.log((p * y).sum(1))
    loss = nll.sum() / m
    loss.backward()
    return {"out": _v(loss), "gz": _g(z)}


def case_minimal_nn_step(T, O):           # one step of test_autograd.py:108-126 on fixed data
    rs = np.random.RandomState(7)
    x = T(rs.normal(0, 1.0, (16, 3)))
    y = x * 3.14 + 30
    w = T(rs.normal(0, 1.0, (3, 3)), requires_grad=True)
    b = T(rs.normal(0, 1.0, 3), requires_grad=True)
    losses = []
    for _ in range(3):
        w.zero_grad()
        b.zero_grad()
        err = x @ w + b - y
        loss = (err ** 2).sum()
        loss.backward()
        w -= 0.001 * w.grad
        b -= 0.001 * b.grad
        losses.append(float(np.asarray(loss.values)))
    return {"losses": np.array(losses), "w": _v(w), "b": _v(b)}


CASES = {name[5:]: fn for name, fn in sorted(globals().items()) if name.startswith("case_")}

# cases whose results are exact in float32 as well as float64 (integers / dyadic rationals): compared
# with == on every backend.  The others involve 1/3, 0.1, exp, log ... : exact for the float64 oracle,
# tolerance for float32 devices.
EXACT_IN_F32 = {
    "add_same_shape", "add_broadcast_leading", "add_broadcast_keepdim", "sub_scalar_rhs", "mul",
    "pow_int_exponent", "pow_square", "dot", "sum_all", "sum_axis1", "neg", "maximum_ties",
    "minimum_ties", "transpose", "max_all_and_axis0", "max_ties_all_get_grad", "min_axis0", "reshape",
    "pad", "flatten", "clip_relu", "clip_both_bounds", "getitem_slice_and_gather", "grad_accumulates",
}
