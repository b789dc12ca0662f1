"""The C-ABI library loads and exports every symbol include/tnn_hip.h declares (no compute, no GPU)."""

import ctypes
import os
import re
import subprocess

from conftest import ROOT, build_twin

HEADER = os.path.join(ROOT, "include", "tnn_hip.h")
LIB = os.path.join(ROOT, "tinynn-autograd_amd", "lib", "libtnn_hip.so")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"TNN_API\s+[\w\s\*]+?\b(tnn_\w+)\s*\(", text)))


def exported(path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
    return {line.split()[-1] for line in out.splitlines() if " T " in line}


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 60 and "tnn_gemm" in syms and "tnn_adam" in syms and "tnn_allreduce" in syms


def test_hip_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    missing = [s for s in declared_symbols() if s not in exported(LIB)]
    assert not missing, missing
    cdll = ctypes.CDLL(LIB)                      # loads without a GPU; nothing is called
    for s in declared_symbols():
        assert hasattr(cdll, s)


def test_ctypes_binding_covers_the_header():
    from tinynn_autograd_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()


def test_cpu_twin_exports_the_same_surface():
    missing = [s for s in declared_symbols() if s not in exported(build_twin())]
    assert not missing, missing


def test_product_loader_has_no_fallback(monkeypatch, tmp_path):
    """A missing library is an ImportError, not a silent CPU path."""
    from tinynn_autograd_amd import _lib
    import pytest
    with pytest.raises(ImportError):
        _lib._Lib(str(tmp_path / "libtnn_hip.so"))
    src = open(os.path.join(ROOT, "tinynn-autograd_amd", "_lib.py")).read()
    pkg_dir = os.path.join(ROOT, "tinynn-autograd_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith(".py"):
                body = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", body, flags=re.M), os.path.join(dirpath, f)
    assert "libtnn_cpu" not in src


def test_late_torch_import_is_a_loud_warning(monkeypatch):
    """torch imported after the HIP library initialised = two HIP runtimes in one process: the loader's meta-path guard warns
    (and get() imports torch itself when WORLD_SIZE / TNN_FORCE_COMM say a communicator is coming)."""
    import warnings
    from tinynn_autograd_amd import _lib
    guard = _lib._TorchAfterLoadGuard()
    monkeypatch.setattr(_lib, "_lib", object())
    monkeypatch.setattr(_lib, "_is_test_twin", False)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        assert guard.find_spec("torch") is None and guard.find_spec("numpy") is None
    assert len(seen) == 1 and "two HIP runtimes" in str(seen[0].message)
    monkeypatch.setattr(_lib, "_lib", None)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        guard.find_spec("torch")
    assert not seen                                   # nothing loaded yet: importing torch now is the right order
