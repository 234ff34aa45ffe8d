"""The binding north_star names -- a host-only torch cpp_extension over the C ABI (geot_amd/csrc_torch/
pointnet2_ext_bindings.cpp: pointnet2._ext's nine pybind11 functions, pointnet2/_ext_src/src/bindings.cpp:9-22) --
against the ctypes module the package uses by default: identical outputs bit for bit (both call the same entry points),
the reference's own wrapper file running on it, and the per-call cost of each."""
import os
import sys
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def test_cpp_binding_builds_and_exports_the_nine_functions():
    """CPU: the extension compiles against this torch (g++, no device code) and exports bindings.cpp's nine names."""
    from geot_amd import build_torch_ext
    mod = build_torch_ext.load()
    names = {"gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn", "three_interpolate",
             "three_interpolate_grad", "ball_query", "group_points", "group_points_grad"}
    assert names <= set(dir(mod))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        mod.furthest_point_sampling(torch.zeros(1, 8, 3), 4)


@pytest.mark.gpu
def test_cpp_binding_equals_the_ctypes_binding():
    from geot_amd import build_torch_ext
    from geot_amd.ext import pointnet2_ext as py
    from geot_amd.synth import make_batch
    cpp = build_torch_ext.load()
    xyz = torch.from_numpy(make_batch(2, 6000, dup_frac=0.01)[0]).to(DEV)
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(2, 24, 6000, generator=g).to(DEV)
    for mod_a, mod_b in ((py, cpp),):
        ia, ib = mod_a.furthest_point_sampling(xyz, 1500), mod_b.furthest_point_sampling(xyz, 1500)
        assert torch.equal(ia, ib) and ib.dtype == torch.int32
        ca, cb = mod_a.gather_points(xyz.transpose(1, 2).contiguous(), ia), mod_b.gather_points(xyz.transpose(1, 2).contiguous(), ib)
        assert torch.equal(ca, cb)
        centres = ca.transpose(1, 2).contiguous()
        assert torch.equal(mod_a.ball_query(centres, xyz, 0.1, 32), mod_b.ball_query(centres, xyz, 0.1, 32))
        bq = mod_a.ball_query(centres, xyz, 0.1, 32)
        ga, gb = mod_a.group_points(feats, bq), mod_b.group_points(feats, bq)
        assert torch.equal(ga, gb)
        up = torch.randn(ga.shape, generator=g).to(DEV)
        assert torch.allclose(mod_a.group_points_grad(up, bq, 6000), mod_b.group_points_grad(up, bq, 6000), rtol=1e-5, atol=1e-5)
        (da, na), (db, nb) = mod_a.three_nn(xyz, centres), mod_b.three_nn(xyz, centres)
        assert torch.equal(na, nb) and torch.equal(da, db)
        w = torch.rand(2, 6000, 3, generator=g).to(DEV)
        known = torch.randn(2, 24, 1500, generator=g).to(DEV)
        assert torch.equal(mod_a.three_interpolate(known, na, w), mod_b.three_interpolate(known, nb, w))
        up2 = torch.randn(2, 24, 6000, generator=g).to(DEV)
        # (24 channels x 1500 targets: too small a workspace for the reverse index -> the channels-last atomic scatter,
        # whose fp32 sums are not bit-reproducible from call to call)
        assert torch.allclose(mod_a.three_interpolate_grad(up2, na, w, 1500), mod_b.three_interpolate_grad(up2, nb, w, 1500),
                              rtol=1e-5, atol=1e-5)
        assert torch.equal(mod_a.gather_points_grad(ca, ia, 6000), mod_b.gather_points_grad(cb, ib, 6000))
    # launches follow torch's current stream
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        on_side = cpp.furthest_point_sampling(xyz, 1500)
    s.synchronize()
    assert torch.equal(on_side, ia)
    with pytest.raises(RuntimeError, match="contiguous"):
        cpp.gather_points(xyz.transpose(1, 2), ia)


@pytest.mark.gpu
def test_reference_wrapper_file_runs_on_the_cpp_binding():
    """aliases.install(binding="cpp"): `import pointnet2._ext` resolves to the compiled module and the package's
    autograd wrappers (same code as the reference's pointnet2_utils.py call sites) produce what they do over ctypes."""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch, geot_amd.aliases as al\n"
        "al.install(binding='cpp'); import pointnet2._ext as e\n"
        "assert type(e).__name__ == 'module' and e.__name__ == '_pointnet2_ext_cpp'\n"
        "from geot_amd.synth import make_batch\n"
        "from geot_amd.ext import pointnet2_ext as py\n"
        "x = torch.from_numpy(make_batch(1, 4096)[0]).cuda()\n"
        "assert torch.equal(e.furthest_point_sampling(x, 512), py.furthest_point_sampling(x, 512)); print('ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-3000:]


@pytest.mark.gpu
def test_per_call_cost_of_both_bindings_is_reported(capsys):
    """Host cost of one small call (1 cloud x 256 points: the kernel is ~5 us, the rest is the binding) -- the number
    behind the choice of default binding; printed (the compiled binding measured 2x cheaper)."""
    from geot_amd import build_torch_ext
    from geot_amd.ext import pointnet2_ext as py
    cpp = build_torch_ext.load()
    feats = torch.randn(1, 8, 256, device=DEV)
    idx = torch.randint(0, 256, (1, 64), device=DEV, dtype=torch.int32)
    cost = {}
    for name, mod in (("ctypes", py), ("cpp", cpp)):
        for _ in range(200):
            mod.gather_points(feats, idx)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000):
            mod.gather_points(feats, idx)
        torch.cuda.synchronize()
        cost[name] = (time.perf_counter() - t0) / 2000 * 1e6
    with capsys.disabled():
        print("\n[binding cost] gather_points on (1, 8, 256): ctypes %.1f us / call, cpp_extension %.1f us / call" % (cost["ctypes"], cost["cpp"]))
    assert cost["cpp"] <= 3.0 * cost["ctypes"]       # (a timing on a shared host: a loose bound; measured 4.6 vs 9.4 us)


POINTOPS_NAMES = {"knnquery_cuda", "furthestsampling_cuda", "furthestsampling_weights_cuda", "ballquery_cuda",
                  "grouping_forward_cuda", "grouping_backward_cuda", "interpolation_forward_cuda", "interpolation_backward_cuda",
                  "subtraction_forward_cuda", "subtraction_backward_cuda", "aggregation_forward_cuda", "aggregation_backward_cuda"}
BATCH_NAMES = {"furthest_point_sampling_wrapper", "gather_points_wrapper", "gather_points_grad_wrapper", "ball_query_wrapper",
               "group_points_wrapper", "group_points_grad_wrapper", "three_nn_wrapper", "three_interpolate_wrapper",
               "three_interpolate_grad_wrapper"}


def test_the_other_two_modules_and_the_dispatcher_build_and_export_their_names():
    """CPU: pointops_cuda (the union of both reference trees: 3 + 11 names, 12 distinct) and pointnet2_batch_cuda (9) as
    compiled modules, and the generated dispatcher with one forwarder per stream-taking entry point of the header."""
    from geot_amd import build_torch_ext, _lib
    pops, batch = build_torch_ext.load("_pointops_cuda_cpp"), build_torch_ext.load("_pointnet2_batch_cpp")
    assert POINTOPS_NAMES <= set(dir(pops)) and BATCH_NAMES <= set(dir(batch))
    disp = build_torch_ext.load("_geot_dispatch_cpp")
    skipped = {"geot_sa_group_mlp_max", "geot_sa_param_floats"}           # host arrays: ctypes (gen_dispatch.py SKIP)
    assert (set(_lib.PROTOTYPES) | set(_lib.PLAIN)) - skipped <= set(dir(disp))
    assert disp.geot_abi_version() == _lib.ABI_VERSION and disp.geot_bn_slices(8, 384, 24000) == _lib.load().geot_bn_slices(8, 384, 24000)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        batch.gather_points_wrapper(1, 1, 4, 2, torch.zeros(1, 1, 4), torch.zeros(1, 2, dtype=torch.int32), torch.zeros(1, 1, 2))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        pops.grouping_forward_cuda(2, 2, 3, torch.zeros(4, 3), torch.zeros(2, 2, dtype=torch.int32), torch.zeros(2, 2, 3))


def test_install_defaults_to_the_compiled_modules():
    """aliases.install() registers the compiled modules when they load (ctypes is the no-compiler fallback)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import geot_amd.aliases as al\n"
            "al.install(); import pointnet2._ext as e, pointops_cuda as p, pointnet2_batch_cuda as b\n"
            "print(al.installed_binding, e.__name__, p.__name__, b.__name__)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.split() == ["cpp", "_pointnet2_ext_cpp", "_pointops_cuda_cpp", "_pointnet2_batch_cpp"]
    code = code.replace("al.install()", "al.install(binding='ctypes')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.split()[0] == "ctypes" and "geot_amd.ext.pointops_cuda" in r.stdout, r.stderr[-2000:]


def _zeros_like_list(ts):
    return [torch.zeros_like(t) for t in ts]


@pytest.mark.gpu
def test_pointops_cuda_module_equals_the_ctypes_module():
    """All 12 functions of the union API, compiled vs ctypes, on offset-batched ragged input: identical outputs."""
    from geot_amd import build_torch_ext
    from geot_amd.ext import pointops_cuda as py
    from geot_amd.synth import make_batch
    cpp = build_torch_ext.load("_pointops_cuda_cpp")
    g = torch.Generator().manual_seed(1)
    sizes = [3000, 1700]
    pts = torch.cat([torch.from_numpy(make_batch(1, n, start_index=i)[0][0]) for i, n in enumerate(sizes)]).to(DEV)
    offset = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=DEV)
    new_sizes = [700, 400]
    new_offset = torch.tensor(np.cumsum(new_sizes), dtype=torch.int32, device=DEV)
    n, m, ns, c = sum(sizes), sum(new_sizes), 16, 32
    out = {}
    for name, mod in (("py", py), ("cpp", cpp)):
        r = {}
        idx = torch.zeros(m, dtype=torch.int32, device=DEV)
        tmp = torch.full((n,), 1e10, device=DEV)
        mod.furthestsampling_cuda(2, max(sizes), pts, offset, new_offset, tmp, idx)
        r["fps"], r["fps_tmp"] = idx, tmp
        w = torch.rand(n, generator=g.manual_seed(3)).to(DEV)
        idx_w = torch.zeros(m, dtype=torch.int32, device=DEV)
        tmp_w = torch.full((n,), 1e10, device=DEV)
        mod.furthestsampling_weights_cuda(2, max(sizes), pts, offset, new_offset, w, tmp_w, idx_w)
        r["fps_w"] = idx_w
        q = pts[idx.long()].contiguous()
        kidx = torch.zeros((m, ns), dtype=torch.int32, device=DEV)
        kd = torch.zeros((m, ns), device=DEV)
        mod.knnquery_cuda(m, ns, pts, q, offset, new_offset, kidx, kd)
        r["knn"], r["knn_d"] = kidx, kd
        bidx = torch.zeros((m, ns), dtype=torch.int32, device=DEV)
        assert mod.ballquery_cuda(m, 0.15, ns, pts, q, offset, new_offset, bidx) == 1
        r["ball"] = bidx
        feats = torch.randn(n, c, generator=g.manual_seed(4)).to(DEV)
        grouped = torch.empty((m, ns, c), device=DEV)
        mod.grouping_forward_cuda(m, ns, c, feats, kidx, grouped)
        r["group"] = grouped
        gin = torch.zeros((n, c), device=DEV)
        mod.grouping_backward_cuda(m, ns, c, grouped, kidx, gin)
        r["group_grad"] = gin
        k3 = kidx[:, :3].contiguous()
        w3 = torch.rand(m, 3, generator=g.manual_seed(5)).to(DEV)
        inter = torch.zeros((m, c), device=DEV)            # accumulated into, as the reference allocates it (pointops.py:277)
        mod.interpolation_forward_cuda(m, c, 3, feats, k3, w3, inter)
        r["interp"] = inter
        gi = torch.zeros((n, c), device=DEV)
        mod.interpolation_backward_cuda(m, c, 3, inter, k3, w3, gi)
        r["interp_grad"] = gi
        f1 = torch.randn(m, c, generator=g.manual_seed(6)).to(DEV)
        sub = torch.zeros((m, ns, c), device=DEV)           # (pointops.py:185)
        mod.subtraction_forward_cuda(m, ns, c, f1, feats, kidx, sub)
        r["sub"] = sub
        g1, g2 = torch.zeros((m, c), device=DEV), torch.zeros((n, c), device=DEV)
        mod.subtraction_backward_cuda(m, ns, c, kidx, sub, g1, g2)
        r["sub_g1"], r["sub_g2"] = g1, g2
        wc = 8
        pos = torch.randn(m, ns, c, generator=g.manual_seed(7)).to(DEV)
        wt = torch.randn(m, ns, wc, generator=g.manual_seed(8)).to(DEV)
        sidx = (kidx % m).contiguous()          # aggregation indexes its own n (= m here) rows
        agg = torch.zeros((m, c), device=DEV)               # (pointops.py:219)
        mod.aggregation_forward_cuda(m, ns, c, wc, f1, pos, wt, sidx, agg)
        r["agg"] = agg
        ga, gp, gw = torch.zeros_like(f1), torch.zeros_like(pos), torch.zeros_like(wt)
        mod.aggregation_backward_cuda(m, ns, c, wc, f1, pos, wt, sidx, agg, ga, gp, gw)
        r["agg_gi"], r["agg_gp"], r["agg_gw"] = ga, gp, gw
        out[name] = r
    torch.cuda.synchronize()
    atomic = {"group_grad", "interp_grad", "sub_g1", "sub_g2", "agg_gi"}        # float atomics: equal up to summation order
    for k in out["py"]:
        a, b = out["py"][k], out["cpp"][k]
        if k in atomic:
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(a.abs().max())), k
        else:
            assert torch.equal(a, b), k


@pytest.mark.gpu
def test_pointnet2_batch_module_equals_the_ctypes_module():
    """All 9 wrappers, compiled vs ctypes, caller-allocated (uninitialised) outputs."""
    from geot_amd import build_torch_ext
    from geot_amd.ext import pointnet2_batch_cuda as py
    from geot_amd.synth import make_batch
    cpp = build_torch_ext.load("_pointnet2_batch_cpp")
    b, n, m, ns, c = 2, 5000, 1200, 16, 24
    xyz = torch.from_numpy(make_batch(b, n, dup_frac=0.01)[0]).to(DEV)
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(b, c, n, generator=g).to(DEV)
    out = {}
    for name, mod in (("py", py), ("cpp", cpp)):
        r = {}
        temp = torch.full((b, n), 1e10, device=DEV)
        idx = torch.empty((b, m), dtype=torch.int32, device=DEV)
        mod.furthest_point_sampling_wrapper(b, n, m, xyz, temp, idx)
        r["fps"] = idx
        gat = torch.empty((b, 3, m), device=DEV).fill_(float("nan"))
        mod.gather_points_wrapper(b, 3, n, m, xyz.transpose(1, 2).contiguous(), idx, gat)
        r["gather"] = gat
        gg = torch.zeros((b, 3, n), device=DEV)
        mod.gather_points_grad_wrapper(b, 3, n, m, gat, idx, gg)
        r["gather_grad"] = gg
        new_xyz = gat.transpose(1, 2).contiguous()
        bq = torch.zeros((b, m, ns), dtype=torch.int32, device=DEV)
        mod.ball_query_wrapper(b, n, m, 0.1, ns, new_xyz, xyz, bq)
        r["ball"] = bq
        grp = torch.empty((b, c, m, ns), device=DEV).fill_(float("nan"))
        mod.group_points_wrapper(b, c, n, m, ns, feats, bq, grp)
        r["group"] = grp
        ggp = torch.zeros((b, c, n), device=DEV)
        mod.group_points_grad_wrapper(b, c, n, m, ns, grp, bq, ggp)
        r["group_grad"] = ggp
        d2 = torch.empty((b, n, 3), device=DEV)
        i3 = torch.empty((b, n, 3), dtype=torch.int32, device=DEV)
        mod.three_nn_wrapper(b, n, m, xyz, new_xyz, d2, i3)
        r["nn_d"], r["nn_i"] = d2, i3
        w = torch.rand(b, n, 3, generator=g.manual_seed(9)).to(DEV)
        known = torch.randn(b, c, m, generator=g.manual_seed(10)).to(DEV)
        up = torch.empty((b, c, n), device=DEV).fill_(float("nan"))
        mod.three_interpolate_wrapper(b, c, m, n, known, i3, w, up)
        r["interp"] = up
        gk = torch.zeros((b, c, m), device=DEV)
        mod.three_interpolate_grad_wrapper(b, c, n, m, up, i3, w, gk)
        r["interp_grad"] = gk
        out[name] = r
    torch.cuda.synchronize()
    for k in out["py"]:
        a, b_ = out["py"][k], out["cpp"][k]
        assert not torch.isnan(b_.float()).any(), k
        if k in ("group_grad", "interp_grad"):
            assert torch.allclose(a, b_, rtol=1e-5, atol=1e-5 * float(a.abs().max())), k
        else:
            assert torch.equal(a, b_), k


@pytest.mark.gpu
def test_the_package_launches_through_the_compiled_dispatcher(capsys):
    """ext/_common.call(): the generated dispatcher serves the package's own launches (GEOT_BINDING=auto), gives the same
    results as the ctypes path, follows torch's current stream, and costs less per launch (printed)."""
    from geot_amd.ext import _common
    from geot_amd.ext import pointnet2_ext as py
    assert _common.dispatcher() is not None and hasattr(_common.dispatcher(), "geot_gather_points")
    feats = torch.randn(1, 8, 256, device=DEV)
    idx = torch.randint(0, 256, (1, 64), device=DEV, dtype=torch.int32)
    want = torch.gather(feats, 2, idx.long().unsqueeze(1).expand(-1, 8, -1))
    cost = {}
    saved = _common._dispatch
    try:
        for name, disp in (("dispatcher", saved), ("ctypes", False)):
            _common._dispatch = disp
            assert torch.equal(py.gather_points(feats, idx), want)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                on_side = py.gather_points(feats, idx)
            s.synchronize()
            assert torch.equal(on_side, want)
            out = torch.empty(1, 8, 64, device=DEV)
            args = (1, 8, 256, 64, feats.data_ptr(), idx.data_ptr(), out.data_ptr())
            for _ in range(300):
                _common.call("geot_gather_points", feats.device, *args)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3000):
                _common.call("geot_gather_points", feats.device, *args)
            cost[name] = (time.perf_counter() - t0) / 3000 * 1e6
            torch.cuda.synchronize()
    finally:
        _common._dispatch = saved
    with capsys.disabled():
        print("\n[launch cost] _common.call('geot_gather_points', ...): ctypes %.2f us, compiled dispatcher %.2f us" %
              (cost["ctypes"], cost["dispatcher"]))
    assert cost["dispatcher"] <= 2.0 * cost["ctypes"]


@pytest.mark.gpu
def test_compiled_modules_on_empty_and_ragged_inputs():
    """Edge cases through the compiled modules: zero clouds, zero samples, zero neighbours, an empty segment of an
    offset-batched call -- the same outputs as the ctypes modules (empty tensors, zero-filled where the reference zero-fills),
    no launch on nothing, and a shape violation is a RuntimeError, not a crash."""
    from geot_amd import build_torch_ext
    from geot_amd.ext import pointnet2_ext as py_ext, pointops_cuda as py_pops
    from geot_amd.synth import make_batch
    ext, pops = build_torch_ext.load("_pointnet2_ext_cpp"), build_torch_ext.load("_pointops_cuda_cpp")
    xyz = torch.from_numpy(make_batch(2, 500)[0]).to(DEV)
    for mod in (py_ext, ext):
        assert tuple(mod.furthest_point_sampling(xyz[:0], 8).shape) == (0, 8)
        assert tuple(mod.furthest_point_sampling(xyz, 0).shape) == (2, 0)
        e = torch.empty((2, 0), dtype=torch.int32, device=DEV)
        assert tuple(mod.gather_points(xyz.transpose(1, 2).contiguous(), e).shape) == (2, 3, 0)
        g = mod.gather_points_grad(torch.empty((2, 3, 0), device=DEV), e, 500)
        assert tuple(g.shape) == (2, 3, 500) and float(g.abs().sum()) == 0.0
        bq = mod.ball_query(xyz[:, :0].contiguous(), xyz, 0.1, 4)
        assert tuple(bq.shape) == (2, 0, 4)
        d, i = mod.three_nn(xyz[:, :0].contiguous(), xyz)
        assert tuple(d.shape) == (2, 0, 3) and tuple(i.shape) == (2, 0, 3)
        with pytest.raises(RuntimeError):
            mod.gather_points(xyz.transpose(1, 2).contiguous(), torch.zeros((3, 4), dtype=torch.int32, device=DEV))   # batch mismatch
    # offset-batched FPS with an EMPTY middle segment: both modules give the same indices
    pts = torch.cat([xyz[0], xyz[1]]).contiguous()
    offset = torch.tensor([500, 500, 1000], dtype=torch.int32, device=DEV)       # segment 1 holds no points
    new_offset = torch.tensor([40, 40, 90], dtype=torch.int32, device=DEV)
    outs = []
    for mod in (py_pops, pops):
        idx = torch.zeros(90, dtype=torch.int32, device=DEV)
        tmp = torch.full((1000,), 1e10, device=DEV)
        mod.furthestsampling_cuda(3, 500, pts, offset, new_offset, tmp, idx)
        outs.append(idx)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and int(outs[1][:40].max()) < 500 and int(outs[1][40:].min()) >= 500
