"""Ahead-of-time compilation of the host layer (the Python modules a training step executes per operation).

The op-level API is the reference's seam (`core/tensor.py`, `core/ops.py`), so an eager step is bounded by the interpreter:
≈ 115 µs of bookkeeping around ≈ 25 µs of kernels.  The same `.py` sources are compiled, unmodified, with
Cython into extension modules under `tinynn-autograd_amd/_compiled/`; the package's import hook (`__init__.py`) loads a
compiled module only when the hash of the `.py` it was built from matches the `.py` on disk, otherwise the interpreter runs
the source as before.  The sources stay the single truth; nothing here is needed for correctness.

    python tinynn-autograd_amd/_host_build.py          # build (or refresh) the compiled modules
    TNN_HOST_COMPILED=0 python ...                     # ignore them for one process
"""

import hashlib
import json
import os
import sys
import sysconfig

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(PKG_DIR, "_compiled")
MANIFEST = os.path.join(OUT_DIR, "manifest.json")
EXT_SUFFIX = sysconfig.get_config_var("EXT_SUFFIX") or ".so"

# what one eager training step runs through, hottest first
MODULES = ["core/tensor.py", "core/ops.py", "device_array.py", "core/layers.py", "core/nn.py", "core/model.py",
           "core/losses.py", "core/optimizer.py", "_lib.py"]


def rel_name(path):
    return path[:-3].replace("/", ".")


def source_hash(path):
    with open(os.path.join(PKG_DIR, path), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()


def compiled_path(path):
    return os.path.join(OUT_DIR, rel_name(path) + EXT_SUFFIX)


def read_manifest():
    try:
        with open(MANIFEST) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def stale_modules():
    have = read_manifest()
    return [m for m in MODULES
            if have.get(rel_name(m)) != source_hash(m) or not os.path.exists(compiled_path(m))]


FASTCALL = "_tnn_fastcall"            # the C-ABI's call wrappers (_fastcall_gen.py): plain C generated from _lib._SIGNATURES


def fastcall_path():
    return os.path.join(OUT_DIR, FASTCALL + EXT_SUFFIX)


def _fastcall_gen():
    """(_fastcall_gen module, the signature table) loaded from their files — this script also runs outside the package."""
    import importlib.util
    mods = []
    for name in ("_fastcall_gen", "_signatures"):
        spec = importlib.util.spec_from_file_location("_tnn_" + name, os.path.join(PKG_DIR, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mods.append(mod)
    return mods[0], mods[1]._SIGNATURES


def fastcall_stale():
    gen, sigs = _fastcall_gen()
    return read_manifest().get(FASTCALL) != gen.signature_hash(sigs) or not os.path.exists(fastcall_path())


def build_fastcall(verbose=False):
    """Generate and compile the call wrappers; returns True when it was (re)built."""
    from setuptools import Extension
    from setuptools.dist import Distribution
    from setuptools.command.build_ext import build_ext
    gen, sigs = _fastcall_gen()
    tmp = os.path.join(OUT_DIR, "_tmp_fastcall")
    os.makedirs(tmp, exist_ok=True)
    src = os.path.join(tmp, FASTCALL + ".c")
    with open(src, "w") as f:
        f.write(gen.generate(sigs))
    ext = Extension(FASTCALL, [src], extra_compile_args=["-O2", "-g0", "-w"], libraries=["dl"])
    dist = Distribution(dict(ext_modules=[ext], script_name="_host_build", script_args=[]))
    cmd = build_ext(dist)
    cmd.build_lib, cmd.build_temp, cmd.inplace, cmd.force = os.path.join(tmp, "lib"), os.path.join(tmp, "obj"), False, True
    cmd.ensure_finalized()
    if not verbose:
        dist.verbose = cmd.verbose = 0
    cmd.run()
    os.replace(os.path.join(tmp, "lib", FASTCALL + EXT_SUFFIX), fastcall_path())
    have = read_manifest()
    have[FASTCALL] = gen.signature_hash(sigs)
    with open(MANIFEST, "w") as f:
        json.dump(have, f, indent=1, sort_keys=True)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return True


def build_host(force=False, verbose=False):
    """Compile the stale modules (and the call wrappers); returns the list that was (re)built.  Raises if Cython or the
    compiler fails."""
    os.makedirs(OUT_DIR, exist_ok=True)
    extra = [FASTCALL] if (force or fastcall_stale()) and build_fastcall(verbose) else []
    todo = list(MODULES) if force else stale_modules()
    if not todo:
        return extra
    from Cython.Build import cythonize
    from setuptools import Extension
    from setuptools.dist import Distribution
    from setuptools.command.build_ext import build_ext
    tmp = os.path.join(OUT_DIR, "_tmp")
    os.makedirs(tmp, exist_ok=True)
    # the C files are generated from COPIES named after the dotted module path: the extension's file name is then
    # `<dotted>.<abi>.so` while its init symbol is the last component, which is all the loader looks at
    exts = []
    for m in todo:
        stem = rel_name(m)
        src = os.path.join(tmp, stem.replace(".", "__") + ".py")
        with open(os.path.join(PKG_DIR, m), "rb") as f, open(src, "wb") as out:
            out.write(f.read())
        exts.append((m, stem, src))
    modules = cythonize(
        [Extension(stem.rsplit(".", 1)[-1], [src], extra_compile_args=["-O2", "-g0", "-w"]) for _, stem, src in exts],
        language_level=3, quiet=not verbose, build_dir=tmp, nthreads=0,
        compiler_directives=dict(binding=True, embedsignature=False))
    out_dir = os.path.join(tmp, "lib")
    dist = Distribution(dict(ext_modules=modules, script_name="_host_build", script_args=[]))
    cmd = build_ext(dist)
    cmd.build_lib, cmd.build_temp, cmd.inplace, cmd.force = out_dir, os.path.join(tmp, "obj"), False, True
    cmd.parallel = min(len(modules), os.cpu_count() or 1)      # the extensions' last name components are distinct
    cmd.ensure_finalized()
    if not verbose:
        dist.verbose = cmd.verbose = 0
    cmd.run()
    built = []
    for (m, stem, _), ext in zip(exts, modules):
        os.replace(os.path.join(out_dir, ext.name + EXT_SUFFIX), compiled_path(m))
        built.append(m)
    have = read_manifest()
    for m in built:
        have[rel_name(m)] = source_hash(m)
    with open(MANIFEST, "w") as f:
        json.dump(have, f, indent=1, sort_keys=True)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)        # generated C + objects: several MB nobody needs afterwards
    return extra + built


if __name__ == "__main__":
    done = build_host(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print("compiled host modules: %s" % (", ".join(done) if done else "up to date"))
