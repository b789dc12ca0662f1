"""The ahead-of-time compiled host modules (tinynn-autograd_amd/_host_build.py): loaded only while they match their
sources, ignored on request, and the interpreter takes over — loudly — for a module whose source has changed."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tinynn-autograd_amd")

PROBE = ("import sys, warnings; sys.path.insert(0, %r)\n"
         "with warnings.catch_warnings(record=True) as w:\n"
         "    warnings.simplefilter('always')\n"
         "    import tinynn_autograd_amd as tn\n"
         "print('COMPILED', ','.join(tn.host_modules_compiled()))\n"
         "print('WARNED', ' | '.join(str(x.message) for x in w if 'compiled host modules' in str(x.message)))\n")


def _probe(root, **env):
    e = dict(os.environ, TNN_HOST_COMPILED="1")          # whatever mode the test session itself runs in
    e.update(env)
    out = subprocess.run([sys.executable, "-c", PROBE % root], env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = dict(l.split(" ", 1) if " " in l else (l, "") for l in out.stdout.strip().splitlines())
    return [n for n in lines["COMPILED"].split(",") if n], lines.get("WARNED", "")


def _host_build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("_tnn_host_build_t", os.path.join(PKG, "_host_build.py"))
    hb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hb)
    return hb


def test_compiled_host_modules_follow_their_sources(tmp_path):
    hb = _host_build()
    try:
        hb.build_host()
    except ImportError:
        pytest.skip("Cython is not installed here: the host modules run interpreted")
    names = sorted(hb.rel_name(m) for m in hb.MODULES)
    assert hb.stale_modules() == []
    got, warned = _probe(ROOT)
    assert got == names and not warned
    got, warned = _probe(ROOT, TNN_HOST_COMPILED="0")
    assert got == [] and not warned

    # a copy of the package whose core/nn.py differs from what was compiled: that one module is interpreted, with a warning
    copy = tmp_path / "repo"
    shutil.copytree(PKG, copy / "tinynn-autograd_amd",
                    ignore=shutil.ignore_patterns("lib", "csrc", "__pycache__", "_tmp"))
    shutil.copy(os.path.join(ROOT, "tinynn_autograd_amd.py"), copy / "tinynn_autograd_amd.py")
    with open(copy / "tinynn-autograd_amd" / "core" / "nn.py", "a") as f:
        f.write("\n# edited after the build\n")
    got, warned = _probe(str(copy))
    assert got == [n for n in names if n != "core.nn"]
    assert "core/nn.py" in warned


def test_call_wrappers_match_the_signature_table_and_behave_like_ctypes():
    """The generated C call wrappers (_fastcall_gen.py, built by _host_build.py): one per entry point with plain integer /
    double / address parameters, bound to the SAME functions of the SAME shared object — used by _lib when their hash matches
    the signature table, ignored with TNN_HOST_COMPILED=0; errors, arity and argument types behave like the ctypes binding."""
    import numpy as np
    import tinynn_autograd_amd as tn
    from tinynn_autograd_amd import _fastcall_gen, _lib
    hb = _host_build()
    hb.build_host()
    assert not hb.fastcall_stale() and hb.read_manifest()[hb.FASTCALL] == _fastcall_gen.signature_hash(_lib._SIGNATURES)
    elig = _fastcall_gen.eligible(_lib._SIGNATURES)
    assert len(elig) > 100 and "tnn_gemm_bias_act" in elig and "tnn_mlp_step" in elig
    assert "tnn_malloc" not in elig and "tnn_device_props" not in elig            # typed out-pointers / strings stay on ctypes
    fast = _lib._fast_wrappers(_lib.get().path)
    if os.environ.get("TNN_HOST_COMPILED", "1") == "0":
        assert fast == {}
        return
    assert sorted(fast) == sorted(elig) and _lib.get().fast_calls == len(elig)
    a = tn.asarray(np.arange(16, dtype=np.float32).reshape(4, 4))
    lib = _lib.get()
    assert lib.fill is fast["tnn_fill"] or lib.fill.__name__ == "tnn_fill"
    import ctypes
    for call in (fast["tnn_fill"], lib._wrap("tnn_fill", lib.cdll.tnn_fill)):       # the wrapper, then the ctypes binding
        call(a._ptr, 2.5, 16, a._code())
        assert (np.asarray(a) == 2.5).all()
        with pytest.raises(_lib.TnnError, match="tnn_fill failed .rc=2.: tnn_fill: unknown dtype 9"):
            call(a._ptr, 1.0, 16, 9)
        with pytest.raises(TypeError):
            call(a._ptr, 1.0, 16)
        with pytest.raises((TypeError, ctypes.ArgumentError)):
            call(a._ptr, 1.0, 16.5, a._code())
    fast["tnn_fill"](a._ptr, np.float32(1.5), np.int64(16), a._code())              # (the wrapper also takes numpy scalars)
    assert (np.asarray(a) == 1.5).all()
    # an address given as a numpy integer is an ADDRESS (never the scalar's own storage); exporters of the buffer protocol
    # other than bytes / bytearray / ctypes arrays are refused with TypeError as ctypes refuses them
    fast["tnn_fill"](np.int64(a._ptr), 3.5, 16, a._code())
    assert (np.asarray(a) == 3.5).all()
    b = tn.asarray(np.zeros(8, np.float32))
    fast["tnn_memcpy_d2d"](np.int64(b._ptr), np.uint64(a._ptr), 32)
    assert (np.asarray(b) == 3.5).all()
    for bad in (np.zeros(16, np.float32), np.float32(1.0), ctypes.pointer(ctypes.c_int(0)), "text", 1.5):
        with pytest.raises(TypeError):
            fast["tnn_fill"](bad, 1.0, 16, a._code())
    host = (ctypes.c_float * 8)()                                                    # a ctypes array passes its storage
    fast["tnn_memcpy_d2h"](host, b._ptr, 32)
    assert list(host) == [3.5] * 8
    # a c_void_p INSTANCE passes its value (event / graph handles), None passes NULL
    ev = _lib.Event().record()
    fast["tnn_event_record"](ev._h)
    lib.stream_sync()
