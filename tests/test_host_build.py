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
