"""CPU suite, part 2: the drop-in boundary.  The C-ABI library must load and
export every symbol include/geot_hip.h declares (no compute without a GPU), and the
product must refuse to run without it rather than fall back."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "geot_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(geot_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from geot_amd import build
    return build.build()


def test_library_exports_every_declared_symbol(built):
    out = subprocess.run(["nm", "-D", "--defined-only", built], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (geot_[a-z0-9_]+)", out))
    declared = _declared()
    assert len(declared) >= 23
    missing = [s for s in declared if s not in exported]
    assert not missing, "declared in geot_hip.h but not exported: %s" % missing
    extra = [s for s in exported if s not in declared]
    assert not extra, "exported but undeclared: %s" % extra


def test_ctypes_binding_covers_the_header(built):
    from geot_amd import _lib
    lib = _lib.load()
    assert lib.geot_abi_version() == _lib.ABI_VERSION == int(re.search(r"GEOT_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "geot_hip.h")).read()).group(1))
    assert sorted(_lib.exported_symbols()) == _declared()
    assert b"invalid" in lib.geot_error_string(1).lower()


def test_code_object_is_gfx950_only(built):
    blob = open(built, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_cpu_tensors_are_rejected_not_emulated(built):
    import torch
    from geot_amd.ext import pointnet2_ext, pointnet2_batch_cuda
    with pytest.raises(RuntimeError, match="CPU not supported"):
        pointnet2_ext.furthest_point_sampling(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        pointnet2_batch_cuda.three_nn_wrapper(1, 4, 4, torch.zeros(1, 4, 3), torch.zeros(1, 4, 3),
                                              torch.zeros(1, 4, 3), torch.zeros(1, 4, 3, dtype=torch.int32))
    # dataloader-side ops: CPU tensors are refused as well (the reference's CPU code is not re-implemented here)
    from geot_amd.openpoints.dataset import grid_subsampling, pc_norm, prepare_sample
    with pytest.raises(RuntimeError, match="CPU not supported"):
        grid_subsampling(torch.zeros(8, 3), sampleDl=0.1, device=torch.device("cpu"))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        pc_norm(torch.zeros(8, 3))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        prepare_sample(torch.zeros(8, 3), torch.zeros(8, dtype=torch.int32), torch.zeros(4, dtype=torch.int64))


def test_missing_library_fails_loudly(tmp_path):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from geot_amd import _lib\n"
            "_lib.LIB_PATH = %r\n"
            "try:\n    _lib.load()\nexcept _lib.GeotLibraryError as e:\n    print('LOUD', e)\n"
            % (ROOT, str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "LOUD" in out.stdout and "no CPU/PyTorch fallback" in out.stdout


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "geot_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                t = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", t, flags=re.M) or "libgeot_oracle" in t:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")
def test_reference_wrapper_files_import_against_our_modules(built):
    """Drop-in check: after aliases.install() the REFERENCE's own wrapper files import cleanly
    (their `import pointnet2._ext`, `import pointops_cuda`, `import pytorch_utils` lines resolve to this
    package).  Import only -- no compute without a GPU; the reference source is read in place, never copied."""
    code = ("import sys, importlib.util; sys.path.insert(0, %r)\n"
            "import geot_amd.aliases as a; a.install()\n"
            "def load(p, n):\n"
            "    s = importlib.util.spec_from_file_location(n, p); m = importlib.util.module_from_spec(s); s.loader.exec_module(m); return m\n"
            "u = load('/root/reference/pointnet2/pointnet2_utils.py', 'ref_pointnet2_utils')\n"
            "p = load('/root/reference/pointops/functions/pointops.py', 'ref_pointops')\n"
            "s = load('/root/reference/openpoints/models/layers/subsample.py', 'ref_subsample')\n"
            "assert u._ext.__name__ in ('_pointnet2_ext_cpp', 'geot_amd.ext.pointnet2_ext') and hasattr(p, 'knnquery') and hasattr(s, 'furthest_point_sample')\n"
            "print('DROPIN_OK')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert "DROPIN_OK" in out.stdout, out.stderr[-2000:]


def test_host_only_planning_entry_points(monkeypatch):
    """Entry points that only plan (sizes, path selection) run without a GPU: sanity of their contracts."""
    from geot_amd import _lib
    lib = _lib.load()
    monkeypatch.delenv("GEOT_NN_IMPL", raising=False)
    # grid kNN: big or long-list problems qualify, small ones and k > 64 do not; env overrides both ways
    assert lib.geot_knn_grid_eligible(1, 24000, 24000, 33) == 1
    assert lib.geot_knn_grid_eligible(8, 8192, 8192, 4) == 1
    assert lib.geot_knn_grid_eligible(1, 512, 24000, 32) == 1
    assert lib.geot_knn_grid_eligible(1, 1000, 1000, 8) == 0
    assert lib.geot_knn_grid_eligible(1, 24000, 24000, 65) == 0
    monkeypatch.setenv("GEOT_NN_IMPL", "wave")
    assert lib.geot_knn_grid_eligible(1, 24000, 24000, 33) == 0
    monkeypatch.setenv("GEOT_NN_IMPL", "grid")
    assert lib.geot_knn_grid_eligible(1, 3000, 3000, 8) == 1
    monkeypatch.delenv("GEOT_NN_IMPL")
    assert lib.geot_ball_grid_eligible(1, 24000, 6000, 0.1, 32) == 1
    assert lib.geot_ball_grid_eligible(1, 24000, 6000, 0.1, 65) == 0
    assert lib.geot_ball_grid_eligible(1, 24000, 6000, -1.0, 32) == 0
    # workspace sizes: monotone, 16-byte multiples, linear in the batch
    w1, w8 = lib.geot_knn_grid_ws_bytes(1, 24000), lib.geot_knn_grid_ws_bytes(8, 24000)
    assert w1 > 24000 * 24 and w1 % 16 == 0 and w8 == 8 * w1
    assert lib.geot_knn_grid_ws_bytes(-1, 5) == -1
    assert lib.geot_ntm_threed_graph_bytes(2, 1000, 32) > 4 * (2 * 1000 * (2 * 64 + 4 * 32))
    assert lib.geot_ntm_threed_loss_ws_bytes(2, 1000, 32) > 0 and lib.geot_ntm_sig_t_mean_ws_floats(8, 24000) > 0
    assert lib.geot_ntm_correct_ws_floats(8, 24000) % 289 == 0
    # gradients: the sorted-pair-stream form (csrc/tile_scatter.hip) wherever a CU's LDS holds the targets' sums
    assert lib.geot_grad_ws_needs_zero(8, 384, 8192, 24000, 3) == 0
    assert lib.geot_grad_ws_needs_zero(8, 64, 24000, 6000 * 32, 1) == 0
    assert lib.geot_grad_ws_needs_zero(1, 3, 8, 3000, 3) == 0
    assert lib.geot_grad_ws_needs_zero(1, 16, 100000, 200000, 3) == 1        # 100 k targets, long rows: channels-last atomic accumulation
    ws = lib.geot_scatter_grad_ws_floats(8, 384, 8192, 24000, 3, 1)
    assert ws >= 8 * 384 * 8192 and lib.geot_scatter_grad_ws_floats(8, 4, 8192, 24000, 3, 1) >= 2 * 8 * 24000 * 3
    assert lib.geot_scatter_grad_ws_floats(1, 16, 100000, 200000, 3, 1) == 16 * 100000
    monkeypatch.setenv("GEOT_GATHER_IMPL", "csr")
    assert lib.geot_grad_ws_needs_zero(8, 64, 24000, 6000 * 32, 1) == 1      # the older forms: 768 KB rows do not fit LDS
    assert lib.geot_grad_ws_needs_zero(1, 16, 8192, 24000, 3) == 1           # workspace too small for their index
    monkeypatch.delenv("GEOT_GATHER_IMPL")
    assert lib.geot_sa_param_floats(3, 3, (__import__("ctypes").c_int * 3)(64, 64, 128)) == 6 * 64 + 64 + 64 * 64 + 64 + 64 * 128 + 128


def test_header_is_plain_c_and_links(tmp_path):
    """include/geot_hip.h must be consumable from C (no C++ / torch types) and every declared entry point must
    resolve at link time against the shared library -- what a maintainer's cgo / JNI / C++ forwarder would do."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "geot_hip.h")).read()
    names = sorted(set(re.findall(r"\b(geot_[a-z0-9_]+)\s*\(", hdr)))
    src = tmp_path / "link_all.c"
    body = "\n".join("    p[%d] = (fn)%s;" % (i, n) for i, n in enumerate(names))
    src.write_text('#include "geot_hip.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {\n    fn p[%d];\n%s\n'
                   '    printf("%%d %%d\\n", geot_abi_version(), p[0] != 0);\n    return 0;\n}\n' % (len(names), body))
    lib_dir = os.path.join(root, "geot_amd")
    exe = tmp_path / "link_all"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
           "-L", lib_dir, "-lgeot_hip", "-Wl,-rpath," + lib_dir, "-Wl,--unresolved-symbols=ignore-in-shared-libs"]
    subprocess.check_call(cmd)
    assert len(names) >= 50


def test_no_undefined_global_names():
    """Static check (pyflakes is not installed): every LOAD_GLOBAL in the product, bench and tool sources
    resolves to a module-level name or a builtin.  A typo'd or never-imported global in a rarely taken branch --
    e.g. the N > 1 tail of bench.py, which no CPU test can execute -- otherwise surfaces only on the GPU box."""
    import ast
    import builtins
    import dis
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    for sub in ("geot_amd", "tools", "oracle"):
        files += glob.glob(os.path.join(root, sub, "**", "*.py"), recursive=True)
    bad = []
    for path in files:
        src = open(path).read()
        tree = ast.parse(src)
        defined = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__builtins__", "__spec__", "__package__"}
        for node in ast.walk(tree):     # any binding anywhere that can create a module-level name
            if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                defined.add(node.name)
            elif isinstance(node, ast.Import):
                defined.update((a.asname or a.name).split(".")[0] for a in node.names)
            elif isinstance(node, ast.ImportFrom):
                defined.update(a.asname or a.name for a in node.names)
            elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
                defined.add(node.id)
            elif isinstance(node, ast.Global):
                defined.update(node.names)

        def walk(code):
            for ins in dis.get_instructions(code):
                if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME") and ins.argval not in defined:
                    bad.append("%s: %s (line %s)" % (os.path.relpath(path, root), ins.argval, ins.starts_line))
            for c in code.co_consts:
                if hasattr(c, "co_code"):
                    walk(c)
        walk(compile(src, path, "exec"))
    assert not bad, "undefined global names:\n" + "\n".join(sorted(set(bad)))
