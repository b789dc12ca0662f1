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
