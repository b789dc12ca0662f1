"""Backend selection for the suite.

  * On a machine with a GPU (the `gpurun` box) everything runs on libtnn_hip.so; a failure to load it is
    an ERROR, never a skip — the product has no fallback and the tests must not invent one.
  * In the build container (no /dev/kfd) the `-m "not gpu"` tests exercise the host logic on the CPU twin
    of the C-ABI (oracle/cpu_twin, test infrastructure); `-m gpu` tests are skipped there.
"""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

TWIN_SRC = os.path.join(ROOT, "oracle", "cpu_twin", "tnn_cpu.cpp")
TWIN_SO = os.path.join(ROOT, "oracle", "_build", "libtnn_cpu.so")
HAS_GPU = os.path.exists("/dev/kfd")


def build_twin():
    deps = [TWIN_SRC, os.path.join(ROOT, "include", "tnn_hip.h"),
            os.path.join(ROOT, "tinynn-autograd_amd", "csrc", "tnn_mlp.cpp")]
    if os.path.exists(TWIN_SO) and all(os.path.getmtime(TWIN_SO) >= os.path.getmtime(d) for d in deps):
        return TWIN_SO
    os.makedirs(os.path.dirname(TWIN_SO), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                           TWIN_SRC, "-o", TWIN_SO])
    return TWIN_SO


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")
    import tinynn_autograd_amd as tn
    from tinynn_autograd_amd import _lib
    if HAS_GPU:
        _lib.get()                               # raises loudly if the .so or the device is missing
        assert tn.backend_name() == "hip-gfx950"
    else:
        _lib.install_test_twin(build_twin())
    config._tnn_backend = tn.backend_name()


def pytest_report_header(config):
    return "tinynn-autograd_amd backend: %s" % getattr(config, "_tnn_backend", "?")


def pytest_collection_modifyitems(config, items):
    if HAS_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (runs on the gpurun box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _default_float32():
    """Every test starts in the product's default precision."""
    import numpy as np
    import tinynn_autograd_amd as tn
    tn.set_default_float(np.float32)
    yield
    tn.set_default_float(np.float32)
