"""Build the host-only PyTorch extension modules of geot_amd/csrc_torch/ (pybind11, no device code):

    _pointnet2_ext_cpp       pointnet2._ext's nine functions            (pointnet2/_ext_src/src/bindings.cpp:9-22)
    _pointops_cuda_cpp       pointops_cuda, the union of both trees     (pointops/src/pointops_api.cpp:8-12,
                                                                         openpoints/cpp/pointops/src/pointops_api.cpp:13-25)
    _pointnet2_batch_cpp     pointnet2_batch_cuda's nine wrappers       (openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24)
    _geot_dispatch_cpp       one forwarder per C-ABI entry point, generated from include/geot_hip.h
                             (csrc_torch/gen_dispatch.py): what the package's own launches go through

    python -m geot_amd.build_torch_ext [--force]

g++ against the torch / pybind11 / HIP headers -- there is no device code, so neither hipcc nor hipify is involved --
linked with the torch libraries and -lgeot_hip (rpath $ORIGIN: every .so sits in geot_amd/).  In-tree and
mtime-incremental like geot_amd/build.py, so the results travel with the repository snapshot to the GPU box; the four
translation units compile in parallel (each pulls in the torch headers: ~1 min of g++ apiece).

The modules are linked against libgeot_hip.so, the `exact` distance build: under GEOT_DISTANCE=fma / fma_xy load() refuses
(the callers then stay on ctypes, which binds the library the mode names) instead of silently mixing two arithmetics.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc_torch")
GENERATED = os.path.join(CSRC, "_generated_dispatch.cpp")
MODULES = {                       # module name -> source
    "_pointnet2_ext_cpp": os.path.join(CSRC, "pointnet2_ext_bindings.cpp"),
    "_pointops_cuda_cpp": os.path.join(CSRC, "pointops_cuda_bindings.cpp"),
    "_pointnet2_batch_cpp": os.path.join(CSRC, "pointnet2_batch_bindings.cpp"),
    "_geot_dispatch_cpp": GENERATED,
}
NAME = "_pointnet2_ext_cpp"       # (kept: the first module this file built)
OUT = os.path.join(HERE, NAME + ".so")
_loaded = {}


def out_path(name):
    return os.path.join(HERE, name + ".so")


def _generate():
    header = os.path.join(ROOT, "include", "geot_hip.h")
    gen = os.path.join(CSRC, "gen_dispatch.py")
    if (not os.path.exists(GENERATED)
            or os.path.getmtime(GENERATED) < max(os.path.getmtime(header), os.path.getmtime(gen))):
        tmp = "%s.%d.tmp" % (GENERATED, os.getpid())
        subprocess.check_call([sys.executable, gen, header, tmp], stdout=subprocess.DEVNULL)
        os.replace(tmp, GENERATED)


class _BuildLock:
    """One builder at a time per checkout (several ranks of one node import the package at once): an exclusive flock on a
    file next to the outputs; the others wait, then find the outputs fresh."""

    def __enter__(self):
        import fcntl
        self.fd = open(os.path.join(HERE, ".build_torch_ext.lock"), "w")
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl
        fcntl.flock(self.fd, fcntl.LOCK_UN)
        self.fd.close()
        return False


def _command(name, src):
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths("cuda") if "device_type" in ce.include_paths.__code__.co_varnames else ce.include_paths()
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-attributes",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=" + name,
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include"]
    cmd += ["-I" + p for p in inc]
    cmd += [src, "-o", "%s.%d.tmp" % (out_path(name), os.getpid()), "-L" + tlib, "-L" + HERE, "-lgeot_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
            "-ltorch", "-ltorch_python", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    return cmd


def build(force=False, verbose=False, only=None):
    """Build every module that is older than its sources (all in parallel) -> path of _pointnet2_ext_cpp.so."""
    from . import build as hip_build
    with _BuildLock():
        lib = hip_build.build()
        _generate()
        common = [os.path.join(ROOT, "include", "geot_hip.h"), os.path.join(CSRC, "binding_common.h"), lib]
        jobs = []
        for name, src in MODULES.items():
            if only is not None and name not in only:
                continue
            out = out_path(name)
            if force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in [src] + common):
                cmd = _command(name, src)
                if verbose:
                    print(" ".join(cmd), flush=True)
                jobs.append((name, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
        failed = []
        for name, proc in jobs:
            _, err = proc.communicate()
            tmp = "%s.%d.tmp" % (out_path(name), os.getpid())
            if proc.returncode != 0:
                failed.append("%s:\n%s" % (name, err[-4000:]))
                if os.path.exists(tmp):
                    os.remove(tmp)
            else:
                os.replace(tmp, out_path(name))      # readers never see a half-written module
        if failed:
            raise RuntimeError("building the torch extension modules failed:\n" + "\n".join(failed))
    return OUT


def load(name=NAME):
    """Import a built module (building first if needed).  Raises ImportError when it cannot serve this process: another
    distance mode than the one it is linked against, or an ABI version other than the header's."""
    if name in _loaded:
        return _loaded[name]
    import importlib.util
    import torch  # noqa: F401  (the torch libraries must be loaded before the extension)
    from . import _lib
    if _lib.DISTANCE != "exact":
        raise ImportError("the compiled bindings are linked against libgeot_hip.so (GEOT_DISTANCE=exact); this process runs "
                          "GEOT_DISTANCE=%s: use the ctypes binding" % _lib.DISTANCE)
    build(only=[name])
    spec = importlib.util.spec_from_file_location(name, out_path(name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if name == "_geot_dispatch_cpp" and (mod.ABI_VERSION != _lib.ABI_VERSION or mod.geot_abi_version() != _lib.ABI_VERSION):
        raise ImportError("%s was built for ABI %d / links a library of ABI %d; the package binds %d -- rebuild"
                          % (name, mod.ABI_VERSION, mod.geot_abi_version(), _lib.ABI_VERSION))
    _loaded[name] = mod
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
