"""DeviceArray: the HBM-resident stand-in for the numpy ndarray behind reference Tensors.

The reference keeps a numpy array in `Tensor._values` / `Tensor.grad` (core/tensor.py:20,171) and its
callers use ndarray idioms on them (`.tolist()`, `.shape`, `np.ravel`, `np.concatenate`, slicing,
arithmetic — core/optimizer.py:14-15,27,47,70-77, examples/mnist/run.py:89).  DeviceArray implements
that surface on top of the C-ABI (include/tnn_hip.h): operators, `__array_ufunc__`,
`__array_function__`, slicing/reshape views and an explicit, synchronising `__array__` (D2H).

Design points
  * always dense row-major; slices of the leading axis, `reshape`, `ravel` are zero-copy views that keep
    their base alive; a 2-D `.T` is a lazy flag consumed by matmul (so `grad @ w.T` and `x.T @ grad`
    become NT / TN GEMMs without materialising a transpose, core/ops.py:157,160).
  * Python / numpy scalars never become device buffers: they ride along as "host scalars" and are
    passed as kernel arguments (tnn_ewise_scalar).
  * no CPU fallback: an operation that has no device implementation raises TypeError.
  * dtypes: float32 (default), float64 (exact mode / explicit), int64 (indices), bool (masks).
    Integer and bool operands of arithmetic are cast to the default float on device.
"""

import ctypes
import math
import numbers

import weakref

import numpy as np

from . import _lib
from ._lib import F32, F64, I64, U8

_CODE = {np.dtype(np.float32): F32, np.dtype(np.float64): F64, np.dtype(np.int64): I64,
         np.dtype(np.bool_): U8,
         np.dtype(np.uint16): _lib.BF16}     # raw bf16 bit patterns: storage only (tinynn_autograd_amd.bf16)
_default_float = np.dtype(np.float32)
READONLY_COPY = "readonly-copy"      # DeviceArray._tag of a snapshot that must not be mistaken for a view (fused.MLPTrainer.param_view)
MAX_NDIM = 6


def set_default_float(dtype):
    """float32 (hot path, default) or float64 (bit-for-bit mode of the reference's known-answer tests)."""
    global _default_float
    dtype = np.dtype(dtype)
    if dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise ValueError("default float must be float32 or float64")
    _default_float = dtype


def get_default_float():
    return _default_float


def _i64arr(values):
    return (ctypes.c_int64 * len(values))(*values)


_prod = math.prod


def _dense_strides(shape):
    st, acc = [], 1
    for s in reversed(shape):
        st.append(acc)
        acc *= int(s)
    return tuple(reversed(st))


def _is_scalar_like(x):
    tx = type(x)
    if tx is float or tx is int:                     # (no ABC instance check for the exact Python types)
        return True
    return isinstance(x, (numbers.Number, np.generic)) or (isinstance(x, np.ndarray) and x.ndim == 0)


# ---- host-side buffer cache in front of tnn_malloc / tnn_free.  The native pool already recycles HBM, but every
# tnn_malloc / tnn_free is a ctypes call (~1.5 us each, 18 of them per eager MNIST-size training step, more than its
# nine kernel launches cost the host).  Freed buffers of an eager step are kept here by size class and handed out again
# without leaving Python; one stream orders all work, so a recycled buffer is safe to reuse at once (the native pool's own
# argument).  Buffers allocated while a hipGraph is being captured belong to that graph in the native pool and never enter
# this cache (own == 2), and nothing is taken from it during a capture.
_cache = {}                       # size class (bytes) -> [ptr, ...]
_cache_bytes = 0
_CACHE_LIMIT = 256 << 20          # beyond this the buffers go back to the native pool
_CACHE_MAX_BLOCK = 32 << 20       # large buffers are rare and expensive to keep twice


def _size_class(nbytes):
    return (nbytes + 511) & ~511 if nbytes > 0 else 512


def trim_cache():
    """Return every cached buffer to the native pool (tests that audit tnn_pool_stats, shutdown)."""
    global _cache_bytes
    lib = _lib._lib
    for cls, ptrs in _cache.items():
        for ptr in ptrs:
            if lib is not None:
                lib.free(ptr)
    _cache.clear()
    _cache_bytes = 0


class DeviceArray(object):
    __slots__ = ("_ptr", "shape", "dtype", "_base", "_hv", "_t", "_tag", "_own", "_aux", "_sc", "__weakref__")
    __array_priority__ = 1000.0

    # ------------------------------------------------------------------ construction
    def __init__(self):
        raise TypeError("use asarray()/empty()/zeros() to create a DeviceArray")

    @classmethod
    def _raw(cls, ptr, shape, dtype, base=None, hv=None, t=False):
        self = object.__new__(cls)
        self._ptr = ptr
        # (shapes reach this point as tuples of Python ints from every internal caller: checked, not rebuilt)
        if type(shape) is tuple:
            n = len(shape)
            if n == 2:
                ok = type(shape[0]) is int and type(shape[1]) is int
            elif n == 1:
                ok = type(shape[0]) is int
            else:
                ok = n == 0 or all(type(v) is int for v in shape)
        else:
            ok = False
        if not ok:
            shape = tuple(map(int, shape))
        self.shape = shape
        self.dtype = dtype if type(dtype) is np.dtype else np.dtype(dtype)
        self._base = base
        self._hv = hv
        self._t = t
        self._own = 0                # 0: the native pool's (or not owned), 1: recyclable through _cache, 2: graph-owned
        self._sc = 0                 # size class of an owned buffer (bytes), set by _new
        self._aux = None             # by-product of the producing launch riding along (core/ops.py: the partial logits)
        self._tag = None             # free-form marker: RELU_SIGN on a fused Dense+ReLU output; the producer's output array
                                     # on a gradient whose ReLU mask has already been applied (core/ops.py dense_)
        return self

    @classmethod
    def _new(cls, shape, dtype):
        global _cache_bytes
        dtype = dtype if type(dtype) is np.dtype else np.dtype(dtype)
        nbytes = _prod(shape) * dtype.itemsize
        sc = _size_class(nbytes)
        if not _lib.capturing:
            ptrs = _cache.get(sc)
            if ptrs:
                self = cls._raw(ptrs.pop(), shape, dtype)
                _cache_bytes -= sc
                self._own = 1
                self._sc = sc
                return self
        p = ctypes.c_void_p()
        _lib.get().malloc(sc, ctypes.byref(p))
        self = cls._raw(p.value, shape, dtype)
        self._own = 2 if _lib.capturing else 1
        self._sc = sc
        return self

    @classmethod
    def _scalar(cls, value):
        """Host scalar (weakly typed, like a Python number under numpy promotion)."""
        tv = type(value)
        if tv is not float and tv is not int:        # (exact Python floats / ints: nothing to normalise, no ABC instance checks)
            if isinstance(value, np.ndarray):
                value = value[()]
            if isinstance(value, (bool, np.bool_)):
                value = float(value)
            elif isinstance(value, (numbers.Integral, np.integer)):
                value = int(value)
            else:
                value = float(value)
        return cls._raw(None, (), _default_float, hv=value)

    def __del__(self):
        global _cache_bytes
        try:
            if self._base is None and self._ptr is not None and _lib._lib is not None:
                if self._own == 1 and not _lib.capturing:
                    sc = self._sc or _size_class(_prod(self.shape) * self.dtype.itemsize)
                    if sc <= _CACHE_MAX_BLOCK and _cache_bytes + sc <= _CACHE_LIMIT:
                        _cache.setdefault(sc, []).append(self._ptr)
                        _cache_bytes += sc
                        return
                _lib._lib.free(self._ptr)
        except Exception:
            pass

    # ------------------------------------------------------------------ basic properties
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return _prod(self.shape)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def itemsize(self):
        return self.dtype.itemsize

    @property
    def is_host_scalar(self):
        return self._hv is not None

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    def __repr__(self):
        if self._hv is not None:
            return "DeviceArray(host scalar %r)" % (self._hv,)
        return "DeviceArray(shape=%s, dtype=%s)" % (self.shape, self.dtype.name)

    __hash__ = object.__hash__

    def _code(self):
        return _CODE[self.dtype]

    def _dev(self):
        """Pointer to a dense row-major device buffer holding this array's logical content."""
        if self._hv is not None:
            return self._materialise_scalar()._ptr
        if self._t:
            return self._contig()._ptr
        return self._ptr

    def _materialise_scalar(self):
        out = DeviceArray._new((), self.dtype)
        _lib.get().fill(out._ptr, float(self._hv), 1, out._code())    # a kernel, not an H2D copy: capturable, no sync
        return out

    def _contig(self):
        """Same logical array without the lazy-transpose flag (materialises the transpose)."""
        if self._hv is not None:
            return self._materialise_scalar()
        if not self._t:
            return self
        rows, cols = self.shape           # logical; stored as [cols, rows]
        out = DeviceArray._new(self.shape, self.dtype)
        _lib.get().strided_copy(self._ptr, _i64arr((1, rows)), out._ptr, 2, _i64arr((rows, cols)),
                                self._code())
        return out

    # ------------------------------------------------------------------ host transfer
    def __array__(self, dtype=None, copy=None):
        if self._hv is not None:
            out = np.asarray(self._hv, dtype=self.dtype)
        else:
            src = self._contig()
            host_dtype = np.bool_ if self.dtype == np.bool_ else self.dtype
            out = np.empty(self.shape, dtype=host_dtype)
            if out.size:
                _lib.get().memcpy_d2h(out.ctypes.data, src._ptr, out.nbytes)
        if dtype is not None and np.dtype(dtype) != out.dtype:
            out = out.astype(dtype)
        return out

    def numpy(self):
        return self.__array__()

    def tolist(self):
        return self.__array__().tolist()

    def item(self):
        if self.size != 1:
            raise ValueError("can only convert an array of size 1 to a Python scalar")
        if self._hv is not None:
            return self._hv
        return self.__array__().reshape(()).item()

    def __float__(self):
        return float(self.item())

    def __int__(self):
        return int(self.item())

    def __bool__(self):
        if self.size != 1:
            raise ValueError("The truth value of an array with more than one element is ambiguous.")
        return bool(self.item())

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    # ------------------------------------------------------------------ dtype handling
    def astype(self, dtype):
        if dtype is self.dtype:                      # (np.dtype objects are singletons per type: the common no-op case)
            return self
        dtype = np.dtype(dtype)
        if dtype == self.dtype:
            return self
        if dtype not in _CODE:
            raise TypeError("unsupported device dtype %s" % dtype)
        if self._hv is not None:
            return DeviceArray._raw(None, (), dtype, hv=self._hv)
        src = self._contig()
        out = DeviceArray._new(self.shape, dtype)
        _lib.get().cast(src._ptr, src._code(), out._ptr, _CODE[dtype], self.size)
        return out

    def _as_float(self, dtype=None):
        if dtype is None:
            dtype = self.dtype if self.dtype.kind == "f" else _default_float
        return self.astype(dtype)

    def copy(self):
        src = self._contig()
        if src._hv is not None:
            return DeviceArray._raw(None, (), self.dtype, hv=self._hv)
        out = DeviceArray._new(self.shape, self.dtype)
        _lib.get().memcpy_d2d(out._ptr, src._ptr, self.nbytes)
        return out

    def fill(self, value):
        _lib.get().fill(self._ptr, float(value), self.size, self._code())

    # ------------------------------------------------------------------ shape manipulation (views)
    def reshape(self, *newshape):
        if len(newshape) == 1 and not isinstance(newshape[0], numbers.Integral):
            newshape = tuple(newshape[0])
        newshape = [int(s) for s in newshape]
        if newshape.count(-1) > 1:
            raise ValueError("can only specify one unknown dimension")
        if -1 in newshape:
            known = _prod([s for s in newshape if s != -1])
            newshape[newshape.index(-1)] = self.size // known if known else 0
        if _prod(newshape) != self.size:
            raise ValueError("cannot reshape array of size %d into shape %s" % (self.size, tuple(newshape)))
        src = self._contig()
        return DeviceArray._raw(src._ptr, newshape, src.dtype, base=src if src._base is None else src._base)

    def ravel(self):
        return self.reshape(self.size)

    def flatten(self):
        return self.ravel().copy()

    @property
    def T(self):
        return self.transpose()

    def transpose(self, *axes):
        if len(axes) == 1 and (axes[0] is None or not isinstance(axes[0], numbers.Integral)):
            axes = axes[0]
        if axes is None or len(axes) == 0:
            axes = tuple(reversed(range(self.ndim)))
        axes = tuple(int(a) % max(self.ndim, 1) for a in axes)
        if sorted(axes) != list(range(self.ndim)):
            raise ValueError("axes don't match array")
        if axes == tuple(range(self.ndim)):
            return self
        if self._hv is not None:
            return self
        if self.ndim == 2:   # lazy flag, flipped
            return DeviceArray._raw(self._ptr, (self.shape[1], self.shape[0]), self.dtype,
                                    base=self if self._base is None else self._base, t=not self._t)
        src = self._contig()
        st = _dense_strides(src.shape)
        new_shape = tuple(src.shape[a] for a in axes)
        new_st = tuple(st[a] for a in axes)
        if len(new_shape) > MAX_NDIM:
            raise TypeError("transpose supports up to %d dimensions on device" % MAX_NDIM)
        out = DeviceArray._new(new_shape, src.dtype)
        _lib.get().strided_copy(src._ptr, _i64arr(new_st), out._ptr, len(new_shape), _i64arr(new_shape),
                                src._code())
        return out

    def _broadcast_to(self, shape):
        """Materialised broadcast (np.broadcast_to + copy)."""
        shape = tuple(int(s) for s in shape)
        if self.shape == shape and not self._t:
            return self
        src = self._contig()
        st = _broadcast_strides(src.shape, shape)
        out = DeviceArray._new(shape, src.dtype)
        if out.size:
            _lib.get().strided_copy(src._ptr, _i64arr(st), out._ptr, len(shape), _i64arr(shape),
                                    src._code())
        return out

    # ------------------------------------------------------------------ indexing
    def _basic_index(self, key):
        """Resolve ints / slices / None / Ellipsis into (offset, shape, strides) over the dense base."""
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = [j for j, k in enumerate(key) if k is Ellipsis][0]
            n_real = len([k for k in key if k is not None and k is not Ellipsis])
            key = key[:i] + (slice(None),) * (self.ndim - n_real) + key[i + 1:]
        base_st = _dense_strides(self.shape)
        offset, shape, strides, dim = 0, [], [], 0
        for k in key:
            if k is None:
                shape.append(1)
                strides.append(0)
                continue
            if dim >= self.ndim:
                raise IndexError("too many indices for array")
            n = self.shape[dim]
            if isinstance(k, (numbers.Integral, np.integer)):
                k = int(k)
                if k < 0:
                    k += n
                if not 0 <= k < n:
                    raise IndexError("index %d is out of bounds for axis %d with size %d" % (k, dim, n))
                offset += k * base_st[dim]
            elif isinstance(k, slice):
                start, stop, step = k.indices(n)
                length = len(range(start, stop, step))
                offset += start * base_st[dim] if length else 0
                shape.append(length)
                strides.append(step * base_st[dim])
            else:
                raise TypeError("unsupported index %r" % (k,))
            dim += 1
        for d in range(dim, self.ndim):
            shape.append(self.shape[d])
            strides.append(base_st[d])
        return offset, tuple(shape), tuple(strides)

    @staticmethod
    def _is_advanced(key):
        return isinstance(key, (list, np.ndarray, DeviceArray)) and not _is_scalar_like(key)

    def _index_array(self, key):
        if isinstance(key, DeviceArray):
            if key.dtype == np.bool_:
                raise TypeError("boolean mask indexing is not supported on device")
            return key.astype(np.int64)._contig()
        idx = np.asarray(key)
        if idx.dtype == np.bool_:
            raise TypeError("boolean mask indexing is not supported on device")
        if idx.ndim != 1:
            raise TypeError("only 1-D integer index arrays are supported on device")
        n = self.shape[0]
        if idx.size and (idx.min() < -n or idx.max() >= n):
            raise IndexError("index out of bounds for axis 0 with size %d" % n)
        return asarray(idx.astype(np.int64))

    def __getitem__(self, key):
        if self._hv is not None:
            if key == () or key is Ellipsis:
                return self
            raise IndexError("too many indices for array")
        src = self._contig()
        if DeviceArray._is_advanced(key):
            if src.ndim < 1:
                raise IndexError("too many indices for array")
            idx = src._index_array(key)
            row = _prod(src.shape[1:])
            out = DeviceArray._new((idx.size,) + src.shape[1:], src.dtype)
            if out.size:
                _lib.get().gather_rows(src._ptr, idx._ptr, out._ptr, idx.size, row, src.shape[0],
                                       src._code())
            return out
        offset, shape, strides = src._basic_index(key)
        ptr = src._ptr + offset * src.itemsize
        owner = src if src._base is None else src._base
        if strides == _dense_strides(shape) or _prod(shape) <= 1:
            return DeviceArray._raw(ptr, shape, src.dtype, base=owner)
        if len(shape) > MAX_NDIM:
            raise TypeError("slicing supports up to %d dimensions on device" % MAX_NDIM)
        out = DeviceArray._new(shape, src.dtype)
        if out.size:
            _lib.get().strided_copy(ptr, _i64arr(strides), out._ptr, len(shape), _i64arr(shape),
                                    src._code())
        return out

    def take(self, indices, axis=0, out=None, mode="raise"):
        """np.take(a, indices, axis=0[, out]): the row gather of `a[indices]` (tnn_gather_rows), optionally INTO an existing
        array — a per-epoch permutation of a resident dataset (utils/data_iterator.py:27-28) then lands at the same HBM
        addresses every epoch, which is what lets a hipGraph captured over the epoch's batches be replayed."""
        if not (axis == 0 or (axis is None and self.ndim == 1)) or mode != "raise":
            raise TypeError("take: only axis 0 / mode 'raise' are implemented on device")
        src = self._contig()
        if src.ndim < 1:
            raise IndexError("too many indices for array")
        idx = src._index_array(indices)
        shape = (idx.size,) + src.shape[1:]
        if out is None:
            out = DeviceArray._new(shape, src.dtype)
        elif (not isinstance(out, DeviceArray) or out.shape != shape or out.dtype != src.dtype or out._t
              or out._hv is not None):
            raise ValueError("take: `out` must be a dense device array of shape %s and dtype %s" % (shape, src.dtype))
        if out.size:
            _lib.get().gather_rows(src._ptr, idx._ptr, out._ptr, idx.size, _prod(src.shape[1:]), src.shape[0],
                                   src._code())
        return out

    def __setitem__(self, key, value):
        if self._t or self._hv is not None:
            raise TypeError("cannot assign into a transposed view or host scalar")
        if self._tag is READONLY_COPY:
            raise ValueError("assignment destination is a read-only COPY (a padded trainer's logical parameter block): "
                             "use MLPTrainer.set_param(layer, key, value)")
        lib = _lib.get()
        if DeviceArray._is_advanced(key):
            idx = self._index_array(key)
            row_shape = (idx.size,) + self.shape[1:]
            val = asarray(value).astype(self.dtype)._broadcast_to(row_shape)
            if val.size:
                lib.scatter_rows(val._ptr, idx._ptr, self._ptr, idx.size, _prod(self.shape[1:]),
                                 self.shape[0], self._code())
            return
        offset, shape, strides = self._basic_index(key)
        if _prod(shape) == 0:
            return
        val = asarray(value).astype(self.dtype)._broadcast_to(shape)
        ptr = self._ptr + offset * self.itemsize
        if strides == _dense_strides(shape):
            lib.memcpy_d2d(ptr, val._dev(), _prod(shape) * self.itemsize)
        else:
            lib.strided_scatter(val._dev(), ptr, _i64arr(strides), len(shape), _i64arr(shape), self._code())

    # ------------------------------------------------------------------ arithmetic
    def __add__(self, o): return _binary(_lib.ADD, self, o)
    def __radd__(self, o): return _binary(_lib.ADD, o, self)
    def __sub__(self, o): return _binary(_lib.SUB, self, o)
    def __rsub__(self, o): return _binary(_lib.SUB, o, self)
    def __mul__(self, o): return _binary(_lib.MUL, self, o)
    def __rmul__(self, o): return _binary(_lib.MUL, o, self)
    def __truediv__(self, o): return _binary(_lib.DIV, self, o)
    def __rtruediv__(self, o): return _binary(_lib.DIV, o, self)
    def __pow__(self, o): return _binary(_lib.POW, self, o)
    def __rpow__(self, o): return _binary(_lib.POW, o, self)
    def __neg__(self): return _unary(_lib.NEG, self)
    def __pos__(self): return self
    def __abs__(self): return _unary(_lib.ABS, self)
    def __matmul__(self, o): return matmul(self, o)
    def __rmatmul__(self, o): return matmul(o, self)

    def __iadd__(self, o): return _binary(_lib.ADD, self, o, out=self)
    def __isub__(self, o): return _binary(_lib.SUB, self, o, out=self)
    def __imul__(self, o): return _binary(_lib.MUL, self, o, out=self)
    def __itruediv__(self, o): return _binary(_lib.DIV, self, o, out=self)

    def __gt__(self, o): return _compare(_lib.GT, self, o)
    def __ge__(self, o): return _compare(_lib.GE, self, o)
    def __lt__(self, o): return _compare(_lib.LT, self, o)
    def __le__(self, o): return _compare(_lib.LE, self, o)
    def __eq__(self, o): return False if o is None else _compare(_lib.EQ, self, o)
    def __ne__(self, o): return True if o is None else _compare(_lib.NE, self, o)

    def __and__(self, o):   # bool masks: a & b  ==  a * b on {0,1}
        a, b = asarray(self), asarray(o)
        if a.dtype != np.bool_ or b.dtype != np.bool_:
            raise TypeError("& is only defined for boolean device arrays")
        return _compare(_lib.NE, _binary(_lib.MUL, a, b), 0.0)

    __rand__ = __and__

    # ------------------------------------------------------------------ reductions & friends
    def sum(self, axis=None, keepdims=False, dtype=None, out=None):
        return _reduce(_lib.RSUM, self, axis, keepdims)

    def max(self, axis=None, keepdims=False, out=None):
        return _reduce(_lib.RMAX, self, axis, keepdims)

    def min(self, axis=None, keepdims=False, out=None):
        return _reduce(_lib.RMIN, self, axis, keepdims)

    def mean(self, axis=None, keepdims=False):
        s = self.sum(axis=axis, keepdims=keepdims)
        n = self.size // max(s.size, 1)
        return s / float(n)

    def argmax(self, axis=None):
        return argmax(self, axis)

    def clip(self, min=None, max=None):
        return clip(self, min, max)

    def dot(self, o):
        return matmul(self, o)

    # ------------------------------------------------------------------ numpy protocols
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        out = kwargs.pop("out", None)
        if kwargs:
            kwargs = {k: v for k, v in kwargs.items() if v is not None}
        if out is not None:
            if len(out) != 1 or not isinstance(out[0], DeviceArray):
                return NotImplemented
            out = out[0]
        if method == "__call__" and not kwargs:
            if ufunc in _UFUNC_BINARY and len(inputs) == 2:
                return _binary(_UFUNC_BINARY[ufunc], inputs[0], inputs[1], out=out)
            if ufunc in _UFUNC_COMPARE and len(inputs) == 2 and out is None:
                return _compare(_UFUNC_COMPARE[ufunc], inputs[0], inputs[1])
            if ufunc in _UFUNC_UNARY and len(inputs) == 1:
                res = _unary(_UFUNC_UNARY[ufunc], inputs[0])
                if out is not None:
                    out[...] = res
                    return out
                return res
            if ufunc is np.matmul and len(inputs) == 2 and out is None:
                return matmul(inputs[0], inputs[1])
            if ufunc is np.positive and len(inputs) == 1:
                return asarray(inputs[0])
            if ufunc is np.logical_and and len(inputs) == 2:
                return asarray(inputs[0]).__and__(inputs[1])
        if method == "reduce" and len(inputs) == 1 and ufunc in _UFUNC_REDUCE:
            axis = kwargs.get("axis", 0)
            return _reduce(_UFUNC_REDUCE[ufunc], asarray(inputs[0]), axis, bool(kwargs.get("keepdims", False)))
        raise TypeError("ufunc %s (method %s) has no device implementation; "
                        "convert explicitly with np.asarray(x) if a host copy is intended"
                        % (ufunc.__name__, method))

    def __array_function__(self, func, types, args, kwargs):
        impl = _ARRAY_FUNCTIONS.get(func)
        if impl is None:
            raise TypeError("numpy.%s has no device implementation; convert explicitly with "
                            "np.asarray(x) if a host copy is intended" % func.__name__)
        return impl(*args, **kwargs)


# ---------------------------------------------------------------------- creation helpers
def empty(shape, dtype=None):
    if type(shape) is not tuple:                     # (the hot callers pass tuples: no ABC instance check for them)
        shape = (shape,) if isinstance(shape, numbers.Integral) else tuple(shape)
    return DeviceArray._new(shape, dtype or _default_float)


def full(shape, value, dtype=None):
    out = empty(shape, dtype)
    if out.size:
        _lib.get().fill(out._ptr, float(value), out.size, out._code())
    return out


def zeros(shape, dtype=None):
    return full(shape, 0.0, dtype)


def ones(shape, dtype=None):
    return full(shape, 1.0, dtype)


def zeros_like(a, dtype=None, **_):
    a = asarray(a)
    return zeros(a.shape, dtype or a.dtype)


def ones_like(a, dtype=None, **_):
    a = asarray(a)
    return ones(a.shape, dtype or a.dtype)


def asarray(obj, dtype=None):
    """Device counterpart of np.asarray (core/tensor.py:20).

    Floating host data is stored as the default device float unless `dtype` says otherwise; integer
    arrays stay int64 (index / label arrays) and are cast to float by the first arithmetic op; Python and
    numpy scalars stay on the host as kernel arguments.
    """
    if isinstance(obj, DeviceArray):
        if dtype is not None and dtype is not obj.dtype and np.dtype(dtype) != obj.dtype:
            return obj.astype(dtype)
        return obj
    if _is_scalar_like(obj):
        s = DeviceArray._scalar(obj)
        if dtype is not None and np.dtype(dtype).kind == "f":
            s.dtype = np.dtype(dtype)
        return s
    host = np.asarray(obj)
    if host.dtype == object:
        raise TypeError("cannot move an object array to the device")
    if dtype is not None:
        target = np.dtype(dtype)
    elif host.dtype.kind == "f":
        target = _default_float
    elif host.dtype.kind in "iu":
        target = np.dtype(np.int64)
    elif host.dtype.kind == "b":
        target = np.dtype(np.bool_)
    else:
        raise TypeError("unsupported host dtype %s" % host.dtype)
    if target not in _CODE:
        raise TypeError("unsupported device dtype %s" % target)
    host = np.ascontiguousarray(host, dtype=target)
    out = DeviceArray._new(host.shape, target)
    if host.size:
        _lib.get().memcpy_h2d(out._ptr, host.ctypes.data, host.nbytes)
    return out


class LazyArray(DeviceArray):
    """An array whose producing launch is deferred until something first asks for its buffer (`_ptr`).

    Used for two things: the gradient views of a Model's arena (`view` / `defer`, see core/model.py) and, first, the logits of
    a classifier head in TRAIN mode (core/nn.py, core/ops.py dense_(lazy=True)).  When the
    loss node gets there first it produces them together with the loss and the head's backward in one launch
    (ops.softmax_nll_); any other first use — printing, argmax, a custom loss — runs the ordinary GEMM, so the values are the
    same either way.  The deferral is visible only to code that mutates the producer's inputs IN PLACE between forward()
    and the first use of the logits (they would be computed from the mutated operands)."""
    __slots__ = ("_thunk",)

    def _get_ptr(self):
        thunk = self._thunk
        if thunk is not None:
            self._thunk = None
            thunk(self)
        return DeviceArray._ptr.__get__(self, DeviceArray)

    def _set_ptr(self, value):
        DeviceArray._ptr.__set__(self, value)

    _ptr = property(_get_ptr, _set_ptr)

    @classmethod
    def deferred(cls, shape, dtype, thunk):
        self = cls._new(shape, dtype)          # the buffer exists; only its content is pending
        self._thunk = thunk
        if _lib.capturing:                     # graph.py re-arms what is still pending when the capture ends
            _CAPTURE_LAZIES.append(weakref.ref(self))
        return self

    @classmethod
    def view(cls, base, offset, shape):
        """A view of `shape` over base's buffer from element `offset` on whose content may be declared pending (`defer`):
        the gradient views of a Model's arena (core/model.py) — the launch that fills them can then wait for the optimizer."""
        ptr = DeviceArray._ptr.__get__(base, DeviceArray) if type(base) is not DeviceArray else base._ptr
        self = cls._raw(ptr + offset * base.dtype.itemsize, shape, base.dtype, base=base)
        self._thunk = None
        return self

    def defer(self, thunk):
        """Declare the content pending: `thunk(self)` runs before the first access to the buffer (or never, see drop)."""
        self._thunk = thunk

    def drop(self):
        """Forget a pending producer (the content is about to be overwritten or is no longer wanted)."""
        self._thunk = None

    @property
    def pending(self):
        return self._thunk is not None

    def fulfilled_ptr(self):
        """The buffer for the launch that fulfils the deferral some other way (and ends it)."""
        self._thunk = None
        return DeviceArray._ptr.__get__(self, DeviceArray)

    def __del__(self):
        self._thunk = None
        DeviceArray.__del__(self)


_CAPTURE_LAZIES = []       # weak references to the LazyArray.deferred objects created inside the open hipGraph capture


def take_capture_lazies():
    """End of a capture (graph.py): the deferred arrays created inside it that NOTHING has asked for yet, with their producers.
    Their launch is not in the graph; a replay refreshes the operands they would be computed from, so the captured function
    re-arms them after every replay — a later read then computes them from the replay's data (eagerly, once) instead of
    returning what the first read after some earlier replay produced."""
    out = []
    for ref in _CAPTURE_LAZIES:
        arr = ref()
        if arr is not None and arr._thunk is not None:
            out.append((ref, arr._thunk))
    del _CAPTURE_LAZIES[:]
    return out


def gather_scalars(arrays, out=None, pointers=None):
    """One vector from n 0-d device arrays that live in n separate buffers (a loop's per-step losses): ONE launch
    (tnn_gather_scalars) instead of n 4-byte copies.  `pointers`: a device array of their addresses from an earlier call
    (returned as the second value) — the arrays of a replayed hipGraph keep their addresses, so the list is uploaded once.
    Returns (vector, pointers)."""
    if pointers is None:
        arrays = [asarray(a) for a in arrays]
        if any(a._hv is not None or a.size != 1 for a in arrays) or len({a.dtype for a in arrays}) != 1:
            raise TypeError("gather_scalars: n one-element device arrays of one dtype are required")
        pointers = asarray(np.array([a._dev() for a in arrays], dtype=np.int64))
        pointers._aux = arrays                       # the buffers stay alive as long as their address list does
    n = pointers.size
    if out is None:
        out = DeviceArray._new((n,), pointers._aux[0].dtype)
    _lib.get().gather_scalars(pointers._ptr, out._ptr, n, out._code())
    return out, pointers


def from_ptr(ptr, shape, dtype, owner):
    """View over memory owned by something else (e.g. the trainer's arenas); `owner` is kept alive."""
    return DeviceArray._raw(ptr, shape, dtype, base=owner)


# ---------------------------------------------------------------------- kernels: elementwise
def _broadcast_strides(src_shape, dst_shape):
    nd = len(dst_shape)
    if len(src_shape) > nd:
        raise ValueError("cannot broadcast %s to %s" % (src_shape, dst_shape))
    padded = (1,) * (nd - len(src_shape)) + tuple(src_shape)
    dense = (0,) * (nd - len(src_shape)) + _dense_strides(src_shape)
    st = []
    for p, d, s in zip(padded, dst_shape, dense):
        if p == d:
            st.append(s if p != 1 else 0)
        elif p == 1:
            st.append(0)
        else:
            raise ValueError("operands could not be broadcast together with shapes %s %s"
                             % (src_shape, dst_shape))
    return tuple(st)


def _float_result_dtype(a, b):
    cands = [x.dtype for x in (a, b) if x._hv is None and x.dtype.kind == "f"]
    if not cands:
        return _default_float
    return np.dtype(np.float64) if np.dtype(np.float64) in cands else np.dtype(np.float32)


_PY_BIN = {
    _lib.ADD: lambda x, y: x + y, _lib.SUB: lambda x, y: x - y, _lib.MUL: lambda x, y: x * y,
    _lib.DIV: lambda x, y: x / y, _lib.POW: lambda x, y: x ** y,
    _lib.MAX: lambda x, y: x if x >= y else y, _lib.MIN: lambda x, y: x if x <= y else y,
}
_PY_CMP = {
    _lib.GT: lambda x, y: x > y, _lib.GE: lambda x, y: x >= y, _lib.LT: lambda x, y: x < y,
    _lib.LE: lambda x, y: x <= y, _lib.EQ: lambda x, y: x == y, _lib.NE: lambda x, y: x != y,
}
_SWAP_CMP = {_lib.GT: _lib.LT, _lib.GE: _lib.LE, _lib.LT: _lib.GT, _lib.LE: _lib.GE,
             _lib.EQ: _lib.EQ, _lib.NE: _lib.NE}


def _binary(op, a, b, out=None):
    a, b = asarray(a), asarray(b)
    lib = _lib.get()
    if a._hv is not None and b._hv is not None:
        res = DeviceArray._scalar(_PY_BIN[op](a._hv, b._hv))
        if out is not None:
            out[...] = res
            return out
        return res
    dt = _float_result_dtype(a, b)
    if out is not None:
        if out.dtype.kind != "f":
            raise TypeError("in-place arithmetic needs a floating device array")
        dt = out.dtype
    if a._hv is not None or b._hv is not None:
        arr, s, lhs = (b, a._hv, 1) if a._hv is not None else (a, b._hv, 0)
        arr = arr._as_float(dt)._contig()
        res = out if out is not None else DeviceArray._new(arr.shape, dt)
        if out is not None and out.shape != arr.shape:
            raise ValueError("non-broadcastable output operand")
        if res.size:
            lib.ewise_scalar(op, arr._ptr, float(s), lhs, res._ptr, res.size, res._code())
        return res
    a, b = a._as_float(dt)._contig(), b._as_float(dt)._contig()
    shape = np.broadcast_shapes(a.shape, b.shape)
    if len(shape) > MAX_NDIM:
        raise TypeError("elementwise ops support up to %d dimensions on device" % MAX_NDIM)
    if out is not None:
        if out.shape != tuple(shape) or out._t:
            raise ValueError("non-broadcastable output operand with shape %s" % (out.shape,))
        res = out
    else:
        res = DeviceArray._new(shape, dt)
    if res.size:
        if op == _lib.ADD and out is not None and a is out and b.shape == out.shape:
            lib.axpy(out._ptr, 1.0, b._ptr, out.size, out._code())     # grad += g, param += step
        else:
            lib.ewise_binary(op, a._ptr, _i64arr(_broadcast_strides(a.shape, shape)), b._ptr,
                             _i64arr(_broadcast_strides(b.shape, shape)), res._ptr, len(shape),
                             _i64arr(shape), res._code())
    return res


def _compare(cmp, a, b):
    a, b = asarray(a), asarray(b)
    lib = _lib.get()
    if a._hv is not None and b._hv is not None:
        return asarray(np.asarray(_PY_CMP[cmp](a._hv, b._hv)))
    dt = _float_result_dtype(a, b)
    if a._hv is not None or b._hv is not None:
        if a._hv is not None:
            arr, s, cmp = b, a._hv, _SWAP_CMP[cmp]
        else:
            arr, s = a, b._hv
        arr = arr._as_float(dt)._contig()
        res = DeviceArray._new(arr.shape, np.bool_)
        if res.size:
            lib.compare_scalar(cmp, arr._ptr, float(s), res._ptr, res.size, arr._code())
        return res
    a, b = a._as_float(dt)._contig(), b._as_float(dt)._contig()
    shape = np.broadcast_shapes(a.shape, b.shape)
    if len(shape) > MAX_NDIM:
        raise TypeError("comparisons support up to %d dimensions on device" % MAX_NDIM)
    res = DeviceArray._new(shape, np.bool_)
    if res.size:
        lib.ewise_compare(cmp, a._ptr, _i64arr(_broadcast_strides(a.shape, shape)), b._ptr,
                          _i64arr(_broadcast_strides(b.shape, shape)), res._ptr, len(shape),
                          _i64arr(shape), a._code())
    return res


_PY_UNA = {
    _lib.NEG: lambda x: -x, _lib.EXP: math.exp, _lib.LOG: math.log, _lib.SQRT: math.sqrt,
    _lib.SQUARE: lambda x: x * x, _lib.ABS: abs, _lib.RECIP: lambda x: 1.0 / x,
    _lib.SIGMOID: lambda x: 1.0 / (1.0 + math.exp(-x)), _lib.TANH: math.tanh, _lib.COPY: lambda x: x,
}


def _unary(op, a):
    a = asarray(a)
    if a._hv is not None:
        return DeviceArray._scalar(_PY_UNA[op](a._hv))
    a = a._as_float()._contig()
    res = DeviceArray._new(a.shape, a.dtype)
    if res.size:
        _lib.get().ewise_unary(op, a._ptr, res._ptr, res.size, res._code())
    return res


def exp(a): return _unary(_lib.EXP, a)
def log(a): return _unary(_lib.LOG, a)
def sqrt(a): return _unary(_lib.SQRT, a)
def sigmoid(a): return _unary(_lib.SIGMOID, a)
def tanh(a): return _unary(_lib.TANH, a)
def maximum(a, b): return _binary(_lib.MAX, a, b)
def minimum(a, b): return _binary(_lib.MIN, a, b)


def clip(a, a_min=None, a_max=None, **_):
    a = asarray(a)
    if a._hv is not None:
        v = a._hv
        if a_min is not None:
            v = max(v, a_min)
        if a_max is not None:
            v = min(v, a_max)
        return DeviceArray._scalar(v)
    a = a._as_float()._contig()
    res = DeviceArray._new(a.shape, a.dtype)
    if res.size:
        _lib.get().clip(a._ptr, int(a_min is not None), float(a_min or 0.0), int(a_max is not None),
                        float(a_max or 0.0), res._ptr, res.size, res._code())
    return res


def clip_bwd(grad, x, a_min=None, a_max=None):
    """grad * [(x >= a_min) & (x <= a_max)]  — core/ops.py:336-343 with the mask recomputed from x."""
    x = asarray(x)._as_float()._contig()
    grad = asarray(grad)._as_float(x.dtype)._broadcast_to(x.shape)
    res = DeviceArray._new(x.shape, x.dtype)
    if res.size:
        _lib.get().clip_bwd(grad._dev(), x._ptr, int(a_min is not None), float(a_min or 0.0),
                            int(a_max is not None), float(a_max or 0.0), res._ptr, res.size, res._code())
    return res


RELU_SIGN = "relu-mask-in-sign-bit"      # _tag of a fused Dense+ReLU output (core/ops.py dense_(relu=True))


def mul_signmask(grad, y):
    """grad where the sign bit of y is clear, else 0 (y: sign-encoded ReLU output)."""
    y = asarray(y)._contig()
    grad = asarray(grad)._as_float(y.dtype)._broadcast_to(y.shape)._contig()
    res = DeviceArray._new(y.shape, y.dtype)
    if res.size:
        _lib.get().mul_signmask(grad._ptr, y._ptr, res._ptr, res.size, res._code())
    return res


def mul_mask(grad, mask):
    """grad * mask for a boolean device mask (vjps of maximum/minimum/max/min)."""
    mask = asarray(mask)
    if mask.dtype != np.bool_:
        return _binary(_lib.MUL, grad, mask)
    grad = asarray(grad)
    shape = tuple(np.broadcast_shapes(grad.shape, mask.shape))
    g = grad._as_float()._broadcast_to(shape)
    m = mask._broadcast_to(shape)
    res = DeviceArray._new(shape, g.dtype)
    if res.size:
        _lib.get().mul_mask(g._dev(), m._dev(), res._ptr, res.size, res._code())
    return res


# ---------------------------------------------------------------------- kernels: reductions
def _reduce(rop, a, axis=None, keepdims=False):
    a = asarray(a)
    if a._hv is not None:
        return a
    a = a._as_float()._contig()
    lib = _lib.get()
    if axis is None:
        axes = tuple(range(a.ndim))
    elif isinstance(axis, (tuple, list)):
        axes = tuple(sorted(int(x) % a.ndim for x in axis))
    else:
        if a.ndim == 0:
            raise np.exceptions.AxisError("axis %d is out of bounds for array of dimension 0" % axis)
        if not -a.ndim <= int(axis) < a.ndim:
            raise np.exceptions.AxisError("axis %d is out of bounds for array of dimension %d"
                                          % (axis, a.ndim))
        axes = (int(axis) % a.ndim,)
    if a.ndim == 0 or not axes:
        return a
    contiguous = axes == tuple(range(axes[0], axes[0] + len(axes)))
    cur = a
    groups = [axes] if contiguous else [(ax,) for ax in reversed(axes)]
    for grp in groups:
        lo, hi = grp[0], grp[-1] + 1
        outer, red, inner = _prod(cur.shape[:lo]), _prod(cur.shape[lo:hi]), _prod(cur.shape[hi:])
        if red == 0 and rop != _lib.RSUM:
            raise ValueError("zero-size array to reduction operation which has no identity")
        kept = cur.shape[:lo] + (1,) * (hi - lo) + cur.shape[hi:]
        res = DeviceArray._new(kept, cur.dtype)
        if res.size:
            lib.reduce(rop, cur._ptr, res._ptr, outer, red, inner, cur._code())
        cur = res
    if not keepdims:
        cur = cur.reshape([s for i, s in enumerate(cur.shape) if i not in axes])
    return cur


def argmax(a, axis=None, **_):
    a = asarray(a)._as_float()._contig()
    if axis is None:
        a2, out_shape = a.reshape(1, a.size), ()
    elif a.ndim >= 1 and int(axis) % a.ndim == a.ndim - 1:
        a2, out_shape = a.reshape(_prod(a.shape[:-1]), a.shape[-1]), a.shape[:-1]
    else:
        perm = [i for i in range(a.ndim) if i != int(axis) % a.ndim] + [int(axis) % a.ndim]
        moved = a.transpose(perm)._contig()
        a2, out_shape = moved.reshape(_prod(moved.shape[:-1]), moved.shape[-1]), moved.shape[:-1]
    res = DeviceArray._new((a2.shape[0],), np.int64)
    if res.size:
        _lib.get().argmax_rows(a2._ptr, res._ptr, a2.shape[0], a2.shape[1], a2._code())
    return res.reshape(out_shape)


# ---------------------------------------------------------------------- kernels: matmul
def matmul(a, b):
    """a @ b for 1-D / 2-D operands; a lazy `.T` on either side selects the NT / TN / TT kernel."""
    a, b = asarray(a), asarray(b)
    if a._hv is not None or b._hv is not None or a.ndim == 0 or b.ndim == 0:
        raise ValueError("matmul: input operand does not have enough dimensions")
    if a.ndim > 2 or b.ndim > 2:
        raise TypeError("matmul supports 1-D and 2-D operands on device")
    dt = _float_result_dtype(a, b)
    a, b = a._as_float(dt), b._as_float(dt)
    squeeze_m = a.ndim == 1
    squeeze_n = b.ndim == 1
    if squeeze_m:
        a = a.reshape(1, a.shape[0])
    if squeeze_n:
        b = b.reshape(b.shape[0], 1)
    M, K = a.shape
    K2, N = b.shape
    if K != K2:
        raise ValueError("matmul: Input operand 1 has a mismatch in its core dimension 0 "
                         "(size %d is different from %d)" % (K2, K))
    res = DeviceArray._new((M, N), dt)
    if res.size:
        ta, tb = int(a._t), int(b._t)
        lda = M if ta else K       # stored [K,M] when transposed
        ldb = K if tb else N       # stored [N,K] when transposed
        _lib.get().gemm(ta, tb, M, N, K, 1.0, a._ptr, lda, b._ptr, ldb, 0.0, res._ptr, N, res._code())
    if squeeze_m and squeeze_n:
        return res.reshape(())
    if squeeze_m:
        return res.reshape(N)
    if squeeze_n:
        return res.reshape(M)
    return res


# ---------------------------------------------------------------------- numpy function overrides
def _np_concatenate(arrays, axis=0, **_):
    arrays = [asarray(x)._contig() for x in arrays]
    if not arrays:
        raise ValueError("need at least one array to concatenate")
    if axis is None:
        arrays, axis = [x.ravel() for x in arrays], 0
    nd = arrays[0].ndim
    axis = int(axis) % nd
    dts = [x.dtype for x in arrays]
    dt = dts[0] if all(d == dts[0] for d in dts) else _float_result_dtype(
        *[x for x in arrays if x.dtype.kind == "f"][:2] or arrays[:2])
    arrays = [x.astype(dt) for x in arrays]
    for x in arrays:
        if x.ndim != nd or x.shape[:axis] + x.shape[axis + 1:] != arrays[0].shape[:axis] + arrays[0].shape[axis + 1:]:
            raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    out_shape = list(arrays[0].shape)
    out_shape[axis] = sum(x.shape[axis] for x in arrays)
    if axis == 0:
        # zero-copy when the pieces are consecutive slices of one buffer (the flat gradient arena)
        p, base, ok = arrays[0]._ptr, arrays[0]._base, arrays[0]._base is not None
        for x in arrays:
            ok = ok and x._base is base and x._ptr == p
            p = (p or 0) + x.nbytes
        if ok:
            return DeviceArray._raw(arrays[0]._ptr, out_shape, dt, base=base)
        out = DeviceArray._new(out_shape, dt)
        off = 0
        for x in arrays:
            if x.size:
                _lib.get().memcpy_d2d(out._ptr + off, x._ptr, x.nbytes)
            off += x.nbytes
        return out
    out = DeviceArray._new(out_shape, dt)
    pos = 0
    for x in arrays:
        key = [slice(None)] * nd
        key[axis] = slice(pos, pos + x.shape[axis])
        out[tuple(key)] = x
        pos += x.shape[axis]
    return out


def _np_reshape(a, *shape, **kwargs):
    if not shape:
        shape = (kwargs.get("newshape", kwargs.get("shape")),)
    return asarray(a).reshape(*shape)


def _np_expand_dims(a, axis):
    a = asarray(a)._contig()
    axis = int(axis)
    if axis < 0:
        axis += a.ndim + 1
    return a.reshape(a.shape[:axis] + (1,) + a.shape[axis:])


def _np_repeat(a, repeats, axis=None):
    a = asarray(a)._contig()
    repeats = int(repeats)
    if axis is None:
        a, axis = a.ravel(), 0
    axis = int(axis) % a.ndim
    # out[..., i*repeats + r, ...] = a[..., i, ...]  ->  view a as [..., n, 1, ...] and broadcast
    expanded = a.reshape(a.shape[:axis + 1] + (1,) + a.shape[axis + 1:])
    target = a.shape[:axis + 1] + (repeats,) + a.shape[axis + 1:]
    out = expanded._broadcast_to(target)
    return out.reshape(a.shape[:axis] + (a.shape[axis] * repeats,) + a.shape[axis + 1:])


def _np_pad(a, pad_width, mode="constant", **kwargs):
    if mode != "constant" or kwargs.get("constant_values", 0) != 0:
        raise TypeError("np.pad on device supports mode='constant' with zeros only")
    a = asarray(a)._contig()
    pw = np.broadcast_to(np.asarray(pad_width, dtype=np.int64), (a.ndim, 2))
    out_shape = tuple(int(s + b + e) for s, (b, e) in zip(a.shape, pw))
    out = zeros(out_shape, a.dtype)
    key = tuple(slice(int(b), int(b) + s) for s, (b, e) in zip(a.shape, pw))
    if a.size:
        out[key] = a
    return out


def _np_where(cond, x=None, y=None):
    if x is None or y is None:
        raise TypeError("np.where(cond) without x, y has no device implementation")
    c = asarray(cond)
    c = c if c.dtype == np.bool_ else _compare(_lib.NE, c, 0.0)
    cf = c.astype(_float_result_dtype(asarray(x), asarray(y)))
    return cf * x + (1.0 - cf) * y


def _np_allclose(a, b, rtol=1e-5, atol=1e-8, equal_nan=False):
    return bool(np.allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol, equal_nan=equal_nan))


def _np_array_equal(a, b, **_):
    return bool(np.array_equal(np.asarray(a), np.asarray(b)))


_UFUNC_BINARY = {np.add: _lib.ADD, np.subtract: _lib.SUB, np.multiply: _lib.MUL,
                 np.true_divide: _lib.DIV, np.power: _lib.POW, np.maximum: _lib.MAX,
                 np.minimum: _lib.MIN, np.float_power: _lib.POW}
_UFUNC_COMPARE = {np.greater: _lib.GT, np.greater_equal: _lib.GE, np.less: _lib.LT,
                  np.less_equal: _lib.LE, np.equal: _lib.EQ, np.not_equal: _lib.NE}
_UFUNC_UNARY = {np.negative: _lib.NEG, np.exp: _lib.EXP, np.log: _lib.LOG, np.sqrt: _lib.SQRT,
                np.square: _lib.SQUARE, np.absolute: _lib.ABS, np.reciprocal: _lib.RECIP,
                np.tanh: _lib.TANH}
_UFUNC_REDUCE = {np.add: _lib.RSUM, np.maximum: _lib.RMAX, np.minimum: _lib.RMIN}

_ARRAY_FUNCTIONS = {
    np.ravel: lambda a, order="C": asarray(a).ravel(),
    np.reshape: _np_reshape,
    np.transpose: lambda a, axes=None: asarray(a).transpose(axes),
    np.concatenate: _np_concatenate,
    np.sum: lambda a, axis=None, keepdims=False, **_: _reduce(_lib.RSUM, a, axis, keepdims),
    np.max: lambda a, axis=None, keepdims=False, **_: _reduce(_lib.RMAX, a, axis, keepdims),
    np.min: lambda a, axis=None, keepdims=False, **_: _reduce(_lib.RMIN, a, axis, keepdims),
    np.mean: lambda a, axis=None, keepdims=False, **_: asarray(a).mean(axis, keepdims),
    np.argmax: argmax,
    np.zeros_like: zeros_like,
    np.ones_like: ones_like,
    np.expand_dims: _np_expand_dims,
    np.repeat: _np_repeat,
    np.clip: clip,
    np.pad: _np_pad,
    np.where: _np_where,
    np.copy: lambda a, **_: asarray(a).copy(),
    np.take: lambda a, indices, axis=None, out=None, mode="raise": asarray(a).take(indices, axis=axis, out=out, mode=mode),
    np.shape: lambda a: asarray(a).shape,
    np.ndim: lambda a: asarray(a).ndim,
    np.size: lambda a, axis=None: asarray(a).size if axis is None else asarray(a).shape[axis],
    np.dot: matmul,
    np.matmul: matmul,
    np.allclose: _np_allclose,
    np.array_equal: _np_array_equal,
    np.squeeze: lambda a, axis=None: asarray(a).reshape(
        [s for i, s in enumerate(asarray(a).shape)
         if not (s == 1 and (axis is None or i == (axis % asarray(a).ndim)))]),
}
