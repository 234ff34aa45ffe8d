"""Build geot_amd/_pointnet2_ext_cpp.so: the host-only PyTorch extension of csrc_torch/pointnet2_ext_bindings.cpp
(pybind11 module with pointnet2._ext's nine functions, forwarding to libgeot_hip.so).

    python -m geot_amd.build_torch_ext

g++ against the torch / pybind11 / HIP headers -- there is no device code, so neither hipcc nor hipify is involved --
linked with the torch libraries and -lgeot_hip (rpath $ORIGIN: both .so files sit in geot_amd/).  In-tree and
mtime-incremental like geot_amd/build.py, so the result travels with the repository snapshot to the GPU box."""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc_torch", "pointnet2_ext_bindings.cpp")
NAME = "_pointnet2_ext_cpp"
OUT = os.path.join(HERE, NAME + ".so")


def build(force=False, verbose=False):
    from . import build as hip_build
    lib = hip_build.build()
    deps = [SRC, os.path.join(ROOT, "include", "geot_hip.h"), lib]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths("cuda") if "device_type" in ce.include_paths.__code__.co_varnames else ce.include_paths()
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-attributes",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=" + NAME,
           "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
           "-I" + os.path.join(ROOT, "include"), "-I" + sysconfig.get_paths()["include"], "-I/opt/rocm/include"]
    cmd += ["-I" + p for p in inc]
    cmd += [SRC, "-o", OUT, "-L" + tlib, "-L" + HERE, "-lgeot_hip", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip",
            "-ltorch", "-ltorch_python", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


def load():
    """Import the built module (building it first if needed)."""
    import importlib.util
    import torch  # noqa: F401  (the torch libraries must be loaded before the extension)
    path = build()
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
