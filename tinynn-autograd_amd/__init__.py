"""tinynn-autograd_amd — MI355X (gfx950) backend for tinynn-autograd's dense-MLP training hot path.

`core.tensor` / `core.ops` keep the reference's Tensor / op / backward API while the arrays behind them
live in HBM and every operation is a hand-written HIP kernel reached through a ctypes C-ABI
(include/tnn_hip.h -> lib/libtnn_hip.so).  There is no CPU fallback: importing is cheap, the first
operation loads the library and fails loudly if it or the GPU is missing.

The directory name carries a hyphen (it mirrors the reference's repository name); import it as
`tinynn_autograd_amd` — the alias module at the repository root and the registration below make both
names refer to the same module objects.
"""

import sys as _sys


def _install_compiled_host_modules():
    """Prefer the ahead-of-time compiled copies of the per-operation host modules (`_host_build.py`) — each only while
    the `.py` it was built from is byte-identical to the one on disk; anything else imports the source as usual."""
    import os
    if os.environ.get("TNN_HOST_COMPILED", "1") == "0":
        return
    from . import _host_build as hb
    have = hb.read_manifest()
    if not have:
        return
    fresh = {}
    for m in hb.MODULES:
        rel, so = hb.rel_name(m), hb.compiled_path(m)
        if os.path.exists(so) and have.get(rel) == hb.source_hash(m):
            fresh[__name__ + "." + rel] = so
    stale = [m for m in hb.MODULES if __name__ + "." + hb.rel_name(m) not in fresh]
    if stale and len(stale) < len(hb.MODULES):
        import warnings
        warnings.warn("compiled host modules out of date for %s (interpreted instead): run "
                      "`python tinynn-autograd_amd/_host_build.py`" % ", ".join(stale), RuntimeWarning, stacklevel=3)

    class _CompiledHostFinder(object):
        @staticmethod
        def find_spec(name, path=None, target=None):
            so = fresh.get(name)
            if so is None:
                return None
            from importlib.util import spec_from_file_location
            return spec_from_file_location(name, so)

    _sys.meta_path.insert(0, _CompiledHostFinder)


_install_compiled_host_modules()

from . import _lib
from . import device_array
from .device_array import DeviceArray, asarray, empty, zeros, ones, set_default_float, get_default_float
from . import core
from .core import tensor, ops, layers, losses, optimizer, model, nn, initializer, evaluator
from .core.tensor import Tensor, as_tensor
from . import utils
from .utils import data_iterator, seeder
from . import dist
from . import fused
from . import graph
from .graph import capture
from .fused import MLPTrainer, trainer_from_net

__version__ = "0.1.0"


def backend_name():
    return _lib.backend_name()


def host_modules_compiled():
    """Names of the host modules that run as compiled extensions in this process (empty: all interpreted)."""
    prefix = __name__ + "."
    return sorted(n[len(prefix):] for n, m in _sys.modules.items()
                  if n.startswith(prefix) and getattr(m, "__file__", "").endswith(".so"))


def synchronize():
    _lib.synchronize()


def _register_alias(alias="tinynn_autograd_amd"):
    """Make `import tinynn_autograd_amd[.x.y]` resolve to these very module objects."""
    prefix = __name__ + "."
    for name, mod in list(_sys.modules.items()):
        if name == __name__:
            _sys.modules.setdefault(alias, mod)
        elif name.startswith(prefix):
            _sys.modules.setdefault(alias + "." + name[len(prefix):], mod)


_register_alias()
