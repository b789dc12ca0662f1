"""Importable alias of the package directory `tinynn-autograd_amd/` (a hyphen cannot appear in an
`import` statement).  `import tinynn_autograd_amd` returns the package itself."""

import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("tinynn-autograd_amd")
sys.modules[__name__] = _pkg
