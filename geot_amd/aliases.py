"""Register the HIP-backed modules under the names the reference imports.

After ``geot_amd.aliases.install()`` these reference import lines resolve to this package
(SURVEY.md section 8b "Import-time names"):

    import pointnet2._ext                                  (pointnet2/pointnet2_utils.py:23)
    import pointops_cuda                                   (pointops/functions/pointops.py:4)
    import pointnet2_batch_cuda                            (openpoints/cpp/pointnet2_batch/__init__.py:1)
    from knn_cuda import KNN                               (openpoints/models/backbone/transformer.py:11)
    from pointnet2_ops import pointnet2_utils              (examples/segmentation/train.py:39)
    import openpoints.cpp.subsampling.grid_subsampling     (openpoints/dataset/grid_sample.py:1; its `compute`)

so the reference's own wrapper files (pointnet2_utils.py, pointops.py, subsample.py, ...) run on
MI355X unchanged.  Nothing is registered implicitly: call install() before importing the
reference packages.
"""
import sys
import types

installed_binding = None


def extension_modules(binding="auto"):
    """(pointnet2._ext, pointops_cuda, pointnet2_batch_cuda, binding used) for `binding`:
    "cpp"    the host-only PyTorch cpp_extension modules of geot_amd/csrc_torch/ (pybind11 functions taking at::Tensor, as
             the reference's bindings.cpp:9-22 / pointops_api.cpp / pointnet2_api.cpp), built in-tree by build_torch_ext;
    "ctypes" geot_amd/ext/*.py (no compiler needed; the only binding under GEOT_DISTANCE=fma / fma_xy);
    "auto"   (default) cpp when the modules build and load, ctypes otherwise."""
    from .ext import pointnet2_ext, pointops_cuda, pointnet2_batch_cuda
    if binding not in ("auto", "cpp", "ctypes"):
        raise ValueError("binding must be 'auto', 'cpp' or 'ctypes'")
    if binding != "ctypes":
        try:
            from . import build_torch_ext
            mods = tuple(build_torch_ext.load(n) for n in ("_pointnet2_ext_cpp", "_pointops_cuda_cpp", "_pointnet2_batch_cpp"))
            return mods + ("cpp",)
        except Exception:      # noqa: BLE001 -- no g++ / torch headers / matching library
            if binding == "cpp":
                raise
    return pointnet2_ext, pointops_cuda, pointnet2_batch_cuda, "ctypes"


def install(force=False, binding="auto"):
    """Register the modules; binding: see extension_modules().  The binding that was used is kept in `installed_binding`."""
    global installed_binding
    pointnet2_ext, pointops_cuda, pointnet2_batch_cuda, installed_binding = extension_modules(binding)
    from . import knn_cuda
    from .pointnet2 import pointnet2_utils

    def put(name, mod):
        if force or name not in sys.modules:
            sys.modules[name] = mod

    put("pointops_cuda", pointops_cuda)
    put("pointnet2_batch_cuda", pointnet2_batch_cuda)
    put("knn_cuda", knn_cuda)
    # `import pointnet2._ext` needs a parent package; reuse a real one if the reference's is importable
    parent = sys.modules.get("pointnet2")
    if parent is None:
        parent = types.ModuleType("pointnet2")
        parent.__path__ = []
        sys.modules["pointnet2"] = parent
    put("pointnet2._ext", pointnet2_ext)
    parent._ext = sys.modules["pointnet2._ext"]
    ops = types.ModuleType("pointnet2_ops")
    ops.pointnet2_utils = pointnet2_utils
    put("pointnet2_ops", ops)
    put("pointnet2_ops.pointnet2_utils", pointnet2_utils)
    # `import pytorch_utils as pt_utils` (pointnet2/pointnet2_utils.py:14, via its sys.path append)
    from .pointnet2 import pytorch_utils
    put("pytorch_utils", pytorch_utils)
    # `from openpoints.cpp.pointnet2_batch import pointnet2_cuda` (openpoints/models/layers/*.py:7-8)
    from .openpoints import cpp as op_cpp
    from .openpoints.cpp import pointnet2_batch as op_p2b
    if "openpoints" not in sys.modules:
        pkg = types.ModuleType("openpoints")
        pkg.__path__ = []
        sys.modules["openpoints"] = pkg
        put("openpoints.cpp", op_cpp)
        sys.modules["openpoints"].cpp = op_cpp
    put("openpoints.cpp.pointnet2_batch", op_p2b)
    # `import openpoints.cpp.subsampling.grid_subsampling as cpp_subsampling` -> .compute(points, features=, classes=,
    # sampleDl=, method=, verbose=): the CPython extension of openpoints/cpp/subsampling/wrapper.cpp:58-285
    from .openpoints.dataset import grid_sample as _gs

    def compute(points, features=None, classes=None, sampleDl=0.1, method="barycenters", verbose=0):
        if method not in ("barycenters", "voxelcenters"):       # wrapper.cpp:86-90
            raise RuntimeError('Error parsing method. Valid method names are "barycenters" and "voxelcenters" ')
        return _gs.grid_subsampling(points, features=features, labels=classes, sampleDl=sampleDl, verbose=verbose)

    sub = types.ModuleType("openpoints.cpp.subsampling")
    sub.__path__ = []
    gsm = types.ModuleType("openpoints.cpp.subsampling.grid_subsampling")
    gsm.compute = compute
    sub.grid_subsampling = gsm
    put("openpoints.cpp.subsampling", sub)
    put("openpoints.cpp.subsampling.grid_subsampling", gsm)
    cpp_pkg = sys.modules.get("openpoints.cpp")
    if cpp_pkg is not None and not hasattr(cpp_pkg, "subsampling"):
        cpp_pkg.subsampling = sys.modules["openpoints.cpp.subsampling"]     # `import a.b.c as x` walks attributes
    return ["pointnet2._ext", "pointops_cuda", "pointnet2_batch_cuda", "knn_cuda", "pointnet2_ops"]
