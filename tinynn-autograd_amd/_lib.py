"""ctypes binding of include/tnn_hip.h (libtnn_hip.so, gfx950).

This is the only place that talks to the native library.  It fails loudly: a missing shared object,
a missing symbol or a machine without a GPU raise at first use — there is no CPU fallback in the
product.  (tests/ may inject the CPU twin from oracle/ explicitly through install_test_twin(); nothing
in this package ever looks for it.)
"""

import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_int64, c_size_t, c_void_p, POINTER

import sys

# the package directory — taken from the package, not from __file__: this module may run from its compiled copy under
# _compiled/ (_host_build.py)
_HERE = os.path.dirname(os.path.abspath(sys.modules[__name__.rpartition(".")[0]].__file__))
# TNN_LIB_PATH: another build of the same library (same-box A/B timing of kernel variants, tools/probes/)
LIB_PATH = os.environ.get("TNN_LIB_PATH") or os.path.join(_HERE, "lib", "libtnn_hip.so")

# dtype / op codes (must mirror include/tnn_hip.h)
F32, F64, I64, U8, BF16 = 0, 1, 2, 3, 4
OPT_MOMENTUM, OPT_RMSPROP, OPT_ADAGRAD, OPT_ADADELTA = 0, 1, 2, 3
ADD, SUB, MUL, DIV, POW, MAX, MIN = range(7)
GT, GE, LT, LE, EQ, NE = range(6)
NEG, EXP, LOG, SQRT, SQUARE, ABS, RECIP, SIGMOID, TANH, COPY = range(10)
RSUM, RMAX, RMIN = range(3)
ACT_NONE, ACT_RELU = 0, 1

_p = c_void_p
_i64p = POINTER(c_int64)

# name -> argtypes; every entry point returns int except tnn_last_error
_SIGNATURES = {
    "tnn_init": [c_int],
    "tnn_shutdown": [],
    "tnn_backend_kind": [],
    "tnn_device_props": [POINTER(c_int), POINTER(c_int), _i64p, c_char_p, c_int],
    "tnn_malloc": [c_size_t, POINTER(c_void_p)],
    "tnn_free": [_p],
    "tnn_pool_stats": [_i64p, _i64p, _i64p],
    "tnn_pool_trim": [],
    "tnn_memcpy_h2d": [_p, _p, c_size_t],
    "tnn_memcpy_d2h": [_p, _p, c_size_t],
    "tnn_memcpy_d2d": [_p, _p, c_size_t],
    "tnn_memset": [_p, c_int, c_size_t],
    "tnn_fill": [_p, c_double, c_int64, c_int],
    "tnn_stream_sync": [],
    "tnn_event_create": [POINTER(c_void_p)],
    "tnn_event_record": [_p],
    "tnn_event_elapsed_ms": [_p, _p, POINTER(ctypes.c_float)],
    "tnn_event_destroy": [_p],
    "tnn_graph_capture_begin": [],
    "tnn_graph_capture_end": [POINTER(c_void_p)],
    "tnn_graph_launch": [_p],
    "tnn_graph_destroy": [_p],
    "tnn_gemm": [c_int, c_int, c_int64, c_int64, c_int64, c_double, _p, c_int64, _p, c_int64,
                 c_double, _p, c_int64, c_int],
    "tnn_gemm_bias_act": [c_int, c_int, c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p,
                          c_int, c_int, _p, c_int64, c_int],
    "tnn_gemm_mask": [c_int, c_int, c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p,
                      c_int64, _p, c_int64, c_int],
    "tnn_gemm_tn_colsum": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, c_int64, _p, c_int],
    "tnn_gemm_tn_adam": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, _p, _p, _p, c_double, c_double, c_double, c_double, _p, c_int],
    "tnn_gemm_tn_adam_bias": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, _p, _p, _p, _p, _p, _p, _p, c_double, c_double, c_double, c_double, _p, c_int],
    "tnn_dense_bwd": [c_int64, c_int64, c_int64, _p, _p, _p, _p, _p, _p, _p, c_int],
    "tnn_ewise_binary": [c_int, _p, _i64p, _p, _i64p, _p, c_int, _i64p, c_int],
    "tnn_ewise_scalar": [c_int, _p, c_double, c_int, _p, c_int64, c_int],
    "tnn_ewise_compare": [c_int, _p, _i64p, _p, _i64p, _p, c_int, _i64p, c_int],
    "tnn_compare_scalar": [c_int, _p, c_double, _p, c_int64, c_int],
    "tnn_ewise_unary": [c_int, _p, _p, c_int64, c_int],
    "tnn_clip": [_p, c_int, c_double, c_int, c_double, _p, c_int64, c_int],
    "tnn_clip_bwd": [_p, _p, c_int, c_double, c_int, c_double, _p, c_int64, c_int],
    "tnn_mul_mask": [_p, _p, _p, c_int64, c_int],
    "tnn_mul_signmask": [_p, _p, _p, c_int64, c_int],
    "tnn_axpy": [_p, c_double, _p, c_int64, c_int],
    "tnn_cast": [_p, c_int, _p, c_int, c_int64],
    "tnn_reduce": [c_int, _p, _p, c_int64, c_int64, c_int64, c_int],
    "tnn_argmax_rows": [_p, _p, c_int64, c_int64, c_int],
    "tnn_strided_copy": [_p, _i64p, _p, c_int, _i64p, c_int],
    "tnn_strided_scatter": [_p, _p, _i64p, c_int, _i64p, c_int],
    "tnn_gather_rows": [_p, _p, _p, c_int64, c_int64, c_int64, c_int],
    "tnn_scatter_rows": [_p, _p, _p, c_int64, c_int64, c_int64, c_int],
    "tnn_gather_scalars": [_p, _p, c_int64, c_int],
    "tnn_one_hot": [_p, _p, c_int64, c_int64, c_int],
    "tnn_bias_act": [_p, _p, c_int, _p, c_int64, c_int64, c_int],
    "tnn_softmax_nll_stats": [_p, c_int64, c_int64, _p, c_int],
    "tnn_lse_merge": [_p, c_int, _p, c_int],
    "tnn_softmax_nll_fwd_bwd": [_p, _p, c_int64, c_int64, c_int64, _p, _p, _p, c_int],
    "tnn_softmax_nll_fused": [_p, _p, c_int64, c_int64, _p, _p, _p, c_int],
    "tnn_softmax_nll_fused_sharded": [_p, _p, c_int64, c_int64, c_int64, _p, _p, _p, c_int],
    "tnn_softmax_nll_fused_tick": [_p, _p, c_int64, c_int64, c_int64, c_int, _p, _p, _p, c_int, _p, c_double, c_double],
    "tnn_mlp_head": [c_int64, c_int64, c_int64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, c_int],
    "tnn_mlp_head_fits": [c_int64, c_int64, c_int64, c_int, POINTER(c_int)],
    "tnn_mlp_head_bwd_reserve": [c_int64, c_int64, c_int64, c_int64],
    "tnn_mlp_head_bwd_fits": [c_int64, c_int64, c_int64, c_int64, c_int, POINTER(c_int)],
    "tnn_mlp_head_tick": [c_int64, c_int64, c_int64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, c_int, _p, c_double, c_double],
    "tnn_mlp_head_bwd_tick": [c_int64, c_int64, c_int64, c_int64] + [_p] * 16 + [c_int, _p, c_double, c_double],
    "tnn_mlp_head_bwd_tick_ext": [c_int64, c_int64, c_int64, c_int64, c_int64] + [_p] * 8 + [c_int] + [_p] * 9 + [c_int, _p, c_double, c_double],
    "tnn_dense_fwd_head_partials": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, c_int, c_int, _p, c_int64,
                                    _p, c_int64, _p, c_int],
    "tnn_dense_fwd_head_partials_stats": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, c_int, c_int, _p, c_int64,
                                          _p, c_int64, _p, _p, _p, _p, _p, c_int, c_int],
    "tnn_dense_fwd_rows_head_stats": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, c_int, c_int, _p, c_int64, _p,
                                      c_int64, _p, _p, _p, c_int],
    "tnn_dense_fwd_rows_head_stats_merged": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, c_int, c_int, _p, c_int64, _p,
                                             c_int64, _p, _p, _p, _p, _p, c_int, c_int],
    "tnn_mse_fwd_bwd": [_p, _p, c_int64, c_int64, _p, _p, c_int],
    "tnn_mse_fwd_bwd_tick": [_p, _p, c_int64, c_int64, _p, _p, _p, c_int, _p, c_double, c_double],
    "tnn_sgd": [_p, _p, c_int64, c_double, c_int],
    "tnn_dense_bwd_first_adam": [c_int64, c_int64, c_int64, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, c_int64,
                                 c_double, c_double, c_double, c_double, _p, c_int],
    "tnn_dense_bwd_first_allreduce_adam": [c_int64, c_int64, c_int64, _p, _p, _p, c_int64, c_int64, c_int64, _p, _p, _p, c_int64,
                                           c_double, c_double, c_double, c_double, _p, c_int64, _p, c_int],
    "tnn_optim_step": [c_int, _p, _p, _p, _p, _p, c_int64, c_double, c_double, c_double, c_double, c_int],
    "tnn_adam": [_p, _p, _p, _p, c_int64, c_double, c_double, c_double, c_double, _p, _p, c_int],
    "tnn_adam_ex": [_p, _p, _p, _p, c_int64, c_double, c_double, c_double, c_double, _p, _p, c_int, c_int, _p, _p],
    "tnn_gemm_bf16_nt": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, c_int64, c_int, _p, c_int, c_int,
                         _p, c_int64],
    "tnn_gemm_bf16_nt_t": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, c_int64, _p, c_int, c_int, _p, c_int64,
                           _p, c_int64],
    "tnn_gemm_bf16_reserve": [c_int64, c_int64, c_int64],
    "tnn_mse_bf16_prep": [_p, _p, c_int64, c_int64, c_int64, _p, _p, _p, _p, _p, c_int64, _p, _p, _p, _p, c_double, c_double],
    "tnn_bias_bf16_adam_multi": [c_int, _p, c_int64, _i64p, _p, _p, _p, _p, _p, c_double, c_double, c_double, c_double, _p],
    "tnn_transpose_bf16": [_p, _p, c_int64, c_int64],
    "tnn_transpose2_bf16": [_p, _p, c_int64, c_int64, _p, _p, c_int64, c_int64],
    "tnn_cast_bf16": [_p, _p, c_int64, c_int],
    "tnn_colsum_bf16": [_p, _p, c_int64, c_int64],
    "tnn_mse_bf16": [_p, _p, c_int64, c_int64, _p, _p],
    "tnn_mse_bf16_tick": [_p, _p, c_int64, c_int64, _p, _p, _p, _p, c_double, c_double],
    "tnn_bias_bf16_adam": [_p, c_int64, c_int64, _p, _p, _p, _p, _p, c_double, c_double, c_double, c_double, _p],
    "tnn_adam_master_bf16": [_p, _p, _p, _p, _p, c_int64, c_double, c_double, c_double, c_double, _p],
    "tnn_adam_master_bf16_2d": [_p, _p, _p, _p, _p, _p, c_int64, c_int64, c_double, c_double, c_double, c_double, _p, c_int],
    "tnn_gemm_bf16_nt_adam": [c_int64, c_int64, c_int64, _p, c_int64, _p, c_int64, _p, _p, _p, _p, _p, _p, c_double, c_double, c_double, c_double, _p],
    "tnn_adam_tick": [_p, c_double, c_double],
    "tnn_adam_master_g16": [_p, _p, _p, _p, _p, c_int64, c_double, c_double, c_double, c_double, _p],
    "tnn_mlp_create": [c_int, _i64p, c_int64, c_int, c_int, c_double, c_double, c_double, c_double,
                       c_int, POINTER(c_void_p)],
    "tnn_mlp_destroy": [_p],
    "tnn_mlp_arena": [_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                      POINTER(c_void_p), _i64p],
    "tnn_mlp_optimizer_state": [_p, POINTER(ctypes.c_void_p)],
    "tnn_mlp_param_offset": [_p, c_int, c_int, _i64p, _i64p],
    "tnn_mlp_forward": [_p, _p, c_int64, _p],
    "tnn_mlp_forward_stats": [_p, _p, c_int64, _p],
    "tnn_mlp_backward": [_p, _p, _p, c_int64, c_int64, _p, _p],
    "tnn_mlp_update": [_p],
    "tnn_mlp_step": [_p, _p, _p, c_int64, _p],
    "tnn_mlp_step_sharded": [_p, _p, _p, c_int64, _p],
    "tnn_mlp_launch_window": [_p, c_int, c_int, POINTER(c_int)],
    "tnn_mlp_keep_grads": [_p, c_int],
    "tnn_mlp_sync_params": [_p],
    "tnn_mlp_activation": [_p, c_int, POINTER(c_void_p)],
    "tnn_mlp_bf16_weights": [_p, POINTER(c_void_p)],
    "tnn_mlp_masters_sharded": [_p, POINTER(c_int)],
    "tnn_mlp_gather_masters": [_p],
    "tnn_comm_unique_id": [_p],
    "tnn_comm_init": [c_int, c_int, _p],
    "tnn_comm_destroy": [],
    "tnn_comm_world": [POINTER(c_int), POINTER(c_int)],
    "tnn_allreduce": [_p, c_int64, c_int, c_int],
    "tnn_allgather": [_p, _p, c_int64, c_int],
    "tnn_allreduce_async": [_p, c_int64, c_int, c_int],
    "tnn_comm_join": [],
    "tnn_reduce_scatter": [_p, _p, c_int64, c_int],
    "tnn_comm_chain_begin": [],
    "tnn_comm_chain_end": [],
    "tnn_comm_wait_oldest": [],
    "tnn_allreduce_adam": [_p, c_int64, _p, _p, _p, c_int64, c_double, c_double, c_double, c_double, _p, c_int, c_int,
                           c_int64, _p],
    "tnn_p2p_create": [c_int, c_int, c_int64, _p],
    "tnn_p2p_connect": [_p],
    "tnn_p2p_enable": [c_int],
    "tnn_p2p_tune": [c_int],
    "tnn_p2p_status": [POINTER(c_int), POINTER(c_int), POINTER(c_int)],
    "tnn_p2p_poll_failed": [POINTER(c_int)],
    "tnn_p2p_debug": [POINTER(c_int)],
    "tnn_p2p_guard_updates": [c_int],
    "tnn_p2p_destroy": [],
}

EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["tnn_last_error"])


class TnnError(RuntimeError):
    """A native call returned non-zero; the message is tnn_last_error()."""


class _Lib(object):
    """Holds the CDLL and one bound callable per entry point that raises on failure."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise ImportError(
                "native library %s not found — build it first: "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C tinynn-autograd_amd/csrc -j8`" % path)
        self.path = path
        self.cdll = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        self.cdll.tnn_last_error.restype = c_char_p
        self.cdll.tnn_last_error.argtypes = []
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(self.cdll, name)     # AttributeError here = header/library mismatch
            fn.argtypes = argtypes
            fn.restype = c_int
            setattr(self, name[4:], self._wrap(name, fn))
        self.kind = self.cdll.tnn_backend_kind()

    def _wrap(self, name, fn):
        last_error = self.cdll.tnn_last_error

        def call(*args):
            rc = fn(*args)
            if rc != 0:
                msg = last_error()
                raise TnnError("%s failed (rc=%d): %s" % (
                    name, rc, msg.decode("utf-8", "replace") if msg else "?"))
        call.__name__ = name
        return call


_lib = None
_is_test_twin = False


class _TorchAfterLoadGuard:
    """torch bundles its own libamdhip64 / librccl and loads them by absolute path.  Imported BEFORE libtnn_hip.so they are
    the one HIP runtime both bind to; imported AFTER it, the process holds two runtimes (torch.cuda sees no device memory of
    ours, RCCL communicators cannot be shared, abort at interpreter exit).  get() imports torch itself when the environment
    says a communicator will be needed; for everybody else this finder turns the late import into a loud warning."""

    def find_spec(self, name, path=None, target=None):
        if name == "torch" and _lib is not None and not _is_test_twin:
            import warnings
            warnings.warn("torch is being imported AFTER tinynn_autograd_amd initialised its HIP runtime: the process now "
                          "holds two HIP runtimes (torch.cuda and RCCL will not see this package's memory). "
                          "`import torch` before the first tinynn_autograd_amd device call.", RuntimeWarning, stacklevel=2)
        return None


_guard_installed = False


def get():
    """The initialised library.  Raises if libtnn_hip.so is missing or no MI355X is visible."""
    global _lib, _guard_installed
    if _lib is None:
        import sys
        if "torch" not in sys.modules:
            if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("TNN_FORCE_COMM") == "1":
                import torch  # noqa: F401  (a communicator will be built: torch's HIP runtime must be the process's one)
            elif not _guard_installed:
                sys.meta_path.insert(0, _TorchAfterLoadGuard())
                _guard_installed = True
        lib = _Lib(LIB_PATH)
        # one process per GPU: torchrun's LOCAL_RANK picks the device; TNN_DEVICE overrides (e.g. several ranks
        # sharing one GPU in the peer-to-peer transport test)
        device = int(os.environ.get("TNN_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        lib.init(device)              # raises TnnError when no HIP device is visible
        _lib = lib
    return _lib


def is_loaded():
    return _lib is not None


def backend_name():
    if _lib is None:
        return "unloaded"
    return "hip-gfx950" if _lib.kind == 1 else "cpu-twin(test only)"


def install_test_twin(path):
    """TESTS ONLY: route the ABI to the CPU twin built from oracle/cpu_twin (no GPU in the container).

    Never called by the package itself; bench.py and __graft_entry__.smoke() assert that
    backend_name() == "hip-gfx950".
    """
    global _lib, _is_test_twin
    import sys
    da = sys.modules.get(__name__.rsplit(".", 1)[0] + ".device_array")
    if da is not None:
        da.trim_cache()               # buffers of the library being replaced
    lib = _Lib(path)
    if lib.kind != 2:
        raise TnnError("install_test_twin: %s is not the CPU test twin" % path)
    lib.init(0)
    _lib = lib
    _is_test_twin = True
    return lib


def device_props():
    lib = get()
    cu, clk, hbm = c_int(0), c_int(0), c_int64(0)
    name = ctypes.create_string_buffer(256)
    lib.device_props(ctypes.byref(cu), ctypes.byref(clk), ctypes.byref(hbm), name, 256)
    return {"name": name.value.decode(), "cus": cu.value, "clock_khz": clk.value,
            "hbm_bytes": hbm.value}


def pool_stats():
    lib = get()
    a, b, c = c_int64(0), c_int64(0), c_int64(0)
    lib.pool_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
    import sys
    da = sys.modules.get(__name__.rsplit(".", 1)[0] + ".device_array")
    return {"live_bytes": a.value, "cached_bytes": b.value, "device_allocs": c.value,
            "host_cached_bytes": da._cache_bytes if da is not None else 0}     # part of live_bytes: device_array's free lists


def synchronize():
    get().stream_sync()


class Event(object):
    """HIP event on the library stream (what bench.py times kernels with)."""

    def __init__(self):
        self._h = c_void_p()
        get().event_create(ctypes.byref(self._h))

    def record(self):
        get().event_record(self._h)
        return self

    def elapsed_ms(self, later):
        ms = ctypes.c_float(0)
        get().event_elapsed_ms(self._h, later._h, ctypes.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if _lib is not None and self._h:
                _lib.event_destroy(self._h)
        except Exception:
            pass


capturing = False     # a hipGraph capture is open (Graph.__enter__ .. __exit__)


class Graph(object):
    """hipGraph captured from everything enqueued on the library stream inside the `with` block."""

    def __init__(self):
        self._h = None

    def __enter__(self):
        global capturing
        get().graph_capture_begin()
        capturing = True                 # device_array's buffer cache steps aside: the pool tags capture-time buffers
        return self

    def __exit__(self, exc_type, exc, tb):
        global capturing
        capturing = False
        h = c_void_p()
        try:
            get().graph_capture_end(ctypes.byref(h))
            self._h = h
        except TnnError:
            if exc_type is None:
                raise
        return False

    def launch(self):
        get().graph_launch(self._h)

    def __del__(self):
        try:
            if _lib is not None and self._h:
                _lib.graph_destroy(self._h)
        except Exception:
            pass
