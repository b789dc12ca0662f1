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

from ._signatures import _SIGNATURES, _i64p, _p        # noqa: E402  (name -> argtypes; every entry point returns int except tnn_last_error)

EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["tnn_last_error"])


class TnnError(RuntimeError):
    """A native call returned non-zero; the message is tnn_last_error()."""


def _fast_wrappers(path):
    """{entry point: callable} from the compiled call-wrapper module (_host_build.py builds it; TNN_HOST_COMPILED=0 or a
    signature table that has changed since it was built -> {}: every call goes through ctypes)."""
    if os.environ.get("TNN_HOST_COMPILED", "1") == "0":
        return {}
    try:
        import importlib.util
        import sysconfig
        so = os.path.join(_HERE, "_compiled", "_tnn_fastcall" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
        if not os.path.exists(so):
            return {}
        from . import _fastcall_gen
        spec = importlib.util.spec_from_file_location("_tnn_fastcall", so)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        if mod.SIGNATURE_HASH != _fastcall_gen.signature_hash(_SIGNATURES):
            return {}
        return mod.bind(path, TnnError)
    except (ImportError, OSError):
        return {}


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
        fast = _fast_wrappers(path)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(self.cdll, name)     # AttributeError here = header/library mismatch
            fn.argtypes = argtypes
            fn.restype = c_int
            # the compiled call wrapper of this entry point when there is one (same function, same error behaviour,
            # arguments converted in C: _fastcall_gen.py), else the ctypes binding
            setattr(self, name[4:], fast.get(name) or self._wrap(name, fn))
        self.fast_calls = len(fast)
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
