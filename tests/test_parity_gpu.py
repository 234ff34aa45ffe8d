"""GPU suite: the HIP path (through the C ABI) against the CPU oracle.

Bit-exact for every integer output (FPS order, ball-query / kNN / three_nn ids) and
for the fp32 squared distances (same un-contracted arithmetic); 1e-5 relative for
interpolation / grouping values and atomically-accumulated gradients.
"""
import numpy as np
import pytest
import torch

from geot_amd.synth import make_batch, make_cloud

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
RTOL = 1e-5  # north-star tolerance for float interpolation / grouping


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ext():
    from geot_amd import _lib
    from geot_amd.ext import pointnet2_ext, pointnet2_batch_cuda, pointops_cuda
    _lib.load()  # must be the HIP library; raises if missing

    class E:
        p2 = pointnet2_ext
        p2b = pointnet2_batch_cuda
        pops = pointops_cuda
    return E


@pytest.fixture(params=["multi", "single", "basic"])
def fps_impl(request, monkeypatch):
    """All three FPS kernels must be bit-exact: bucket-pruned with multi-commit rounds (default),
    bucket-pruned one sample per round (GEOT_FPS_IMPL=single), unpruned (GEOT_FPS_IMPL=basic)."""
    monkeypatch.setenv("GEOT_FPS_IMPL", request.param)
    return request.param


def fps_k1(ext, xyz, m):
    return host(ext.p2.furthest_point_sampling(dev(xyz), m))


def fps_k1p(ext, xyz, m, return_temp=False):
    b, n, _ = xyz.shape
    out = torch.full((b, m), -7, dtype=torch.int32, device=DEV)  # garbage: must be overwritten
    temp = torch.full((b, n), 1e10, dtype=torch.float32, device=DEV)
    ext.p2b.furthest_point_sampling_wrapper(b, n, m, dev(xyz), temp, out)
    return (host(out), host(temp)) if return_temp else host(out)


def fps_k2(ext, flat, off, noff, w=None):
    off_t, noff_t = dev(off, torch.int32), dev(noff, torch.int32)
    sizes = np.diff(np.concatenate([[0], off]))
    idx = torch.full((int(noff[-1]),), -7, dtype=torch.int32, device=DEV)
    tmp = torch.full((flat.shape[0],), 1e10, dtype=torch.float32, device=DEV)
    if w is None:
        ext.pops.furthestsampling_cuda(len(off), int(sizes.max()), dev(flat), off_t, noff_t, tmp, idx)
    else:
        ext.pops.furthestsampling_weights_cuda(len(off), int(sizes.max()), dev(flat), off_t, noff_t, dev(w), tmp, idx)
    return host(idx)


# ---- FPS ------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["plain", "dup1pct"])
def test_config1_golden_on_gpu(ext, golden, tag, fps_impl):
    g = golden("config1_%s.npz" % tag)
    xyz = g["xyz"]
    k1 = fps_k1(ext, xyz, 1024)
    assert np.array_equal(k1, g["fps_k1"])
    assert np.array_equal(fps_k1p(ext, xyz, 1024), g["fps_k1p"])
    assert np.array_equal(fps_k2(ext, xyz.reshape(-1, 3), np.array([4096]), np.array([1024])), g["fps_k2"])
    centres = np.take_along_axis(xyz, k1[..., None].astype(np.int64).repeat(3, -1), 1)
    bq = host(ext.p2.ball_query(dev(centres), dev(xyz), float(g["radius"]), int(g["nsample"])))
    assert np.array_equal(bq, g["ball_query"])


@pytest.mark.parametrize("n,m", [(1, 1), (2, 2), (3, 3), (5, 4), (63, 17), (64, 64), (65, 33), (100, 100),
                                 (513, 200), (777, 300), (1025, 128), (1500, 256), (2049, 300), (4097, 200),
                                 (8193, 150), (16385, 100)])
def test_fps_small_and_ragged(ext, oracle, n, m, fps_impl):
    xyz, _ = make_batch(2, n, start_index=n, dup_frac=0.05 if n > 20 else 0.0, origin_pts=2)
    assert np.array_equal(fps_k1(ext, xyz, m), oracle.fps_dense(xyz, m, 512, True))
    got, temp = fps_k1p(ext, xyz, m, return_temp=True)
    want, wtemp = oracle.fps_dense(xyz, m, 1024, False, return_temp=True)
    assert np.array_equal(got, want)
    assert np.array_equal(temp, wtemp)  # the running min-distance buffer is part of the contract


def test_fps_edge_cases(ext, oracle, fps_impl):
    xyz, _ = make_batch(1, 40, origin_pts=0)
    assert np.array_equal(fps_k1p(ext, xyz, 60), oracle.fps_dense(xyz, 60, 1024, False))  # m > n
    tiny = (np.random.default_rng(0).standard_normal((1, 50, 3)) * 0.001).astype(np.float32)
    assert (fps_k1(ext, tiny, 10) == 0).all()  # every point origin-skipped
    tie = np.zeros((1, 16, 3), dtype=np.float32)
    tie[0, 1:, 0] = 1.0
    got = fps_k1p(ext, tie, 3)
    assert got[0, 1] == 8 and np.array_equal(got, oracle.fps_dense(tie, 3, 1024, False))
    # heavy ties at full register occupancy: 24k points drawn from 300 distinct positions
    rng = np.random.default_rng(1)
    base, _ = make_cloud(300, 99, origin_pts=0)
    heavy = base[rng.integers(0, 300, 24000)][None]
    assert np.array_equal(fps_k1p(ext, heavy, 400), oracle.fps_dense(heavy, 400, 1024, False))
    assert np.array_equal(fps_k1(ext, heavy, 400), oracle.fps_dense(heavy, 400, 512, True))


def test_fps_offset_ragged_weighted(ext, oracle, fps_impl):
    sizes = [700, 1300, 64, 2048, 5000]
    ms = [100, 333, 64, 512, 1000]
    flat = np.concatenate([make_cloud(n, 50 + i, dup_frac=0.02)[0] for i, n in enumerate(sizes)])
    off, noff = np.cumsum(sizes), np.cumsum(ms)
    assert np.array_equal(fps_k2(ext, flat, off, noff), oracle.fps_offset(flat, off, noff))
    w = np.random.default_rng(3).random(flat.shape[0]).astype(np.float32)
    w[::17] = 0.0
    assert np.array_equal(fps_k2(ext, flat, off, noff, w), oracle.fps_offset(flat, off, noff, w))


def test_fps_full_size_vs_oracle_and_properties(ext, oracle, fps_impl):
    """BASELINE shapes: 24k-point clouds, the 512 / 8192 targets of the backbone."""
    xyz, _ = make_batch(2, 24000, dup_frac=0.01)
    got512 = fps_k1(ext, xyz, 512)
    assert np.array_equal(got512, oracle.fps_dense(xyz, 512, 512, True))
    flat = xyz.reshape(-1, 3)
    off = np.array([24000, 48000])
    got = fps_k2(ext, flat, off, np.array([8192, 16384]))
    assert np.array_equal(got, oracle.fps_offset(flat, off, np.array([8192, 16384])))
    # size-independent properties: starts at the segment base, stays in segment, prefix property
    for i in range(2):
        seg = got[i * 8192:(i + 1) * 8192]
        assert seg[0] == i * 24000 and (seg >= i * 24000).all() and (seg < (i + 1) * 24000).all()
    half = fps_k2(ext, flat, off, np.array([4096, 8192]))
    assert np.array_equal(half[:4096], got[:4096]) and np.array_equal(half[4096:], got[8192:8192 + 4096])
    # greedy max-min property: each pick maximises the distance to the set picked so far
    p = xyz[0].astype(np.float64)
    sel = got[:64]
    mind = np.full(24000, np.inf)
    for j in range(63):
        mind = np.minimum(mind, ((p - p[sel[j]]) ** 2).sum(1))
        assert mind[sel[j + 1]] >= mind.max() * (1 - 1e-6)


@pytest.mark.parametrize("shape", ["lattice", "line", "clusters", "dups50", "outlier", "plane_ties"])
def test_fps_adversarial_geometry(ext, oracle, fps_impl, shape):
    """Geometries that stress the pruned / multi-commit kernels: exact distance ties between waves (lattices),
    degenerate bounding boxes (line, plane), most cells empty (outlier), half the cloud duplicated, tight
    clusters.  Indices AND the final min-distance buffer must equal the oracle's, for both tie rules."""
    rng = np.random.default_rng(17)
    n, m = 12000, 700
    if shape == "lattice":
        g = np.stack(np.meshgrid(np.arange(23), np.arange(23), np.arange(23)), -1).reshape(-1, 3)[:n]
        x = (g / 32.0).astype(np.float32)
    elif shape == "line":
        x = np.zeros((n, 3), np.float32); x[:, 0] = (np.arange(n) % 4000) / 4000.0
    elif shape == "clusters":
        x = (rng.integers(0, 6, (n, 1)) * 0.15 + rng.standard_normal((n, 3)) * 1e-3).astype(np.float32)
    elif shape == "dups50":
        x = (rng.random((n, 3)) * 0.9).astype(np.float32); x[n // 2:] = x[rng.integers(0, n // 2, n - n // 2)]
    elif shape == "outlier":
        x = (rng.random((n, 3)) * 0.01).astype(np.float32); x[77] = (0.9, -0.9, 0.5)
    else:
        side = 110
        gx, gy = np.meshgrid(np.arange(side, dtype=np.float32), np.arange(side, dtype=np.float32))
        x = np.stack([gx.ravel()[:n] / 128, gy.ravel()[:n] / 128, np.full(n, 0.25, np.float32)], 1)
    x = np.ascontiguousarray(x[None])
    got, temp = fps_k1p(ext, x, m, return_temp=True)
    want, wtemp = oracle.fps_dense(x, m, 1024, False, return_temp=True)
    assert np.array_equal(got, want)
    assert np.array_equal(temp, wtemp)
    assert np.array_equal(fps_k1(ext, x, m), oracle.fps_dense(x, m, 512, True))


def test_fps_streaming_path_large_cloud(ext, oracle, fps_impl):
    xyz, _ = make_batch(1, 30000, start_index=4)  # > 24576 points: not register-resident
    assert np.array_equal(fps_k1p(ext, xyz, 300), oracle.fps_dense(xyz, 300, 1024, False))
    assert np.array_equal(fps_k1(ext, xyz, 300), oracle.fps_dense(xyz, 300, 512, True))


# ---- ball query -------------------------------------------------------------
def test_ball_query_edges(ext, oracle):
    xyz, _ = make_batch(2, 600, start_index=7, dup_frac=0.05)
    q = xyz[:, ::7].copy()
    q[:, 0] = 5.0
    for r, ns in [(0.05, 8), (0.2, 16), (0.5, 64), (3.0, 700), (0.2, 1)]:
        got = host(ext.p2.ball_query(dev(q), dev(xyz), r, ns))
        assert np.array_equal(got, oracle.ball_query(q, xyz, r, ns))
    b, m, n = 2, q.shape[1], 600
    idx = torch.full((b, m, 16), -7, dtype=torch.int32, device=DEV)  # uninitialised output
    ext.p2b.ball_query_wrapper(b, n, m, 0.2, 16, dev(q), dev(xyz), idx)
    assert np.array_equal(host(idx), oracle.ball_query(q, xyz, 0.2, 16))
    off, noff = np.array([600, 1200]), np.array([m, 2 * m])
    idx = torch.full((2 * m, 16), -7, dtype=torch.int32, device=DEV)
    ext.pops.ballquery_cuda(2 * m, 0.2, 16, dev(xyz.reshape(-1, 3)), dev(q.reshape(-1, 3)),
                            dev(off, torch.int32), dev(noff, torch.int32), idx)
    assert np.array_equal(host(idx), oracle.ballquery_offset(0.2, 16, xyz.reshape(-1, 3), q.reshape(-1, 3), off, noff))


def test_ball_query_config2_size(ext, oracle):
    """BASELINE configs[1]: 24k points, 6000 FPS centres, r=0.1, nsample=32."""
    xyz, _ = make_batch(1, 24000)
    centres_idx = fps_k1p(ext, xyz, 6000)
    centres = np.take_along_axis(xyz, centres_idx[..., None].astype(np.int64).repeat(3, -1), 1)
    got = host(ext.p2.ball_query(dev(centres), dev(xyz), 0.1, 32))
    assert np.array_equal(got, oracle.ball_query(centres, xyz, 0.1, 32))
    # properties: ascending unique prefix, then repeats of the first hit; all inside the radius
    d2 = ((xyz[0][got[0]] - centres[0][:, None, :]) ** 2).sum(-1)
    assert (d2 < 0.1 * 0.1 + 1e-6).all()
    assert (got[0][:, 0] <= got[0].min(1)).all()


# ---- three_nn / interpolate ---------------------------------------------------
def test_three_nn(ext, oracle):
    xyz, _ = make_batch(2, 900, start_index=3, dup_frac=0.05)
    known = xyz[:, ::5].copy()
    d2, idx = ext.p2.three_nn(dev(xyz), dev(known))
    wd2, widx = oracle.three_nn(xyz, known)
    assert np.array_equal(host(idx), widx) and np.array_equal(host(d2), wd2)
    for m in (1, 2, 3):
        d2, idx = ext.p2.three_nn(dev(xyz[:, :10]), dev(known[:, :m]))
        wd2, widx = oracle.three_nn(xyz[:, :10], known[:, :m])
        assert np.array_equal(host(idx), widx) and np.array_equal(host(d2), wd2)
    b, n, m = 2, 900, known.shape[1]
    d2 = torch.full((b, n, 3), -1.0, device=DEV)
    idx = torch.full((b, n, 3), -7, dtype=torch.int32, device=DEV)
    ext.p2b.three_nn_wrapper(b, n, m, dev(xyz), dev(known), d2, idx)
    wd2, widx = oracle.three_nn(xyz, known)
    assert np.array_equal(host(idx), widx) and np.array_equal(host(d2), wd2)


def test_three_nn_propagation0_size(ext, oracle):
    """propogation_0 of the backbone: 24000 unknown x 8192 known."""
    xyz, _ = make_batch(1, 24000, start_index=2, dup_frac=0.01)
    sel = fps_k2(ext, xyz.reshape(-1, 3), np.array([24000]), np.array([8192]))
    known = xyz[:, sel]
    d2, idx = ext.p2.three_nn(dev(xyz), dev(known))
    wd2, widx = oracle.three_nn(xyz, known)
    assert np.array_equal(host(idx), widx) and np.array_equal(host(d2), wd2)
    assert (host(d2)[0, sel, 0] == 0).all()  # a known point's nearest neighbour is itself


def test_three_interpolate_fwd_bwd(ext, oracle):
    rng = np.random.default_rng(11)
    b, c, m, n = 2, 37, 256, 1000
    feats = rng.standard_normal((b, c, m)).astype(np.float32)
    idx = rng.integers(0, m, (b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    out = ext.p2.three_interpolate(dev(feats), dev(idx), dev(w))
    np.testing.assert_allclose(host(out), oracle.three_interpolate(feats, idx, w), rtol=RTOL, atol=1e-6)
    go = rng.standard_normal((b, c, n)).astype(np.float32)
    g = ext.p2.three_interpolate_grad(dev(go), dev(idx), dev(w), m)
    np.testing.assert_allclose(host(g), oracle.three_interpolate_grad(go, idx, w, m), rtol=1e-4, atol=1e-5)
    out2 = torch.full((b, c, n), float("nan"), device=DEV)
    ext.p2b.three_interpolate_wrapper(b, c, m, n, dev(feats), dev(idx), dev(w), out2)
    assert torch.equal(out, out2)
    g2 = torch.zeros((b, c, m), device=DEV)
    ext.p2b.three_interpolate_grad_wrapper(b, c, n, m, dev(go), dev(idx), dev(w), g2)
    np.testing.assert_allclose(host(g2), host(g), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("b,c,m,n", [(2, 64, 2048, 6000),      # reverse index, whole rows in LDS (one part)
                                      (2, 32, 4096, 24000),     # reverse index, source parts looped in the workgroup
                                      (2, 48, 128, 3000),       # few targets, long lists: 16 lanes per target
                                      (2, 32, 512, 3500),       # 8 lanes per target
                                      (1, 40, 50, 300),         # too small for the reverse index: channels-last scatter
                                      (1, 24, 70, 0)])          # nothing to scatter: zeros
def test_three_interpolate_grad_out_overwrites_uninitialised_buffers(ext, oracle, b, c, m, n):
    """geot_three_interpolate_grad_out: grad_points and the workspace arrive full of NaNs; every element is written
    (targets no unknown point refers to become 0) and the result equals the accumulate-into-zeros entry point."""
    from geot_amd.ext._common import call, ptr
    rng = np.random.default_rng(5)
    idx = rng.integers(0, max(m - 7, 1), (b, n, 3)).astype(np.int32)          # the last targets stay untouched
    w = rng.random((b, n, 3)).astype(np.float32)
    go = rng.standard_normal((b, c, n)).astype(np.float32)
    out = torch.full((b, c, m), float("nan"), device=DEV)
    from geot_amd import _lib
    ws_floats = max(int(_lib.load().geot_scatter_grad_ws_floats(b, c, m, n, 3, 1)), b * c * m, 1)      # ABI 7: the call's own query
    ws = torch.full((ws_floats,), float("nan"), device=DEV)
    d_go, d_idx, d_w = dev(go), dev(idx), dev(w)          # named: the raw pointers below do not keep them alive
    call("geot_three_interpolate_grad_out", out.device, b, c, n, m, ptr(d_go), ptr(d_idx), ptr(d_w), ptr(out), ptr(ws))
    want = oracle.three_interpolate_grad(go, idx, w, m) if n else np.zeros((b, c, m), np.float32)
    np.testing.assert_allclose(host(out), want, rtol=1e-4, atol=1e-5)
    assert (host(out)[:, :, m - 7:] == 0).all()


def test_reference_gradcheck_vector_on_gpu(ext):
    """pointnet2/pointnet2_test.py:15-27 through the autograd wrapper."""
    from geot_amd.pointnet2 import pointnet2_utils as pu
    torch.manual_seed(0)
    feats = torch.randn(1, 2, 4, device=DEV, requires_grad=True)
    idx = torch.tensor([[[0, 1, 2], [1, 2, 3]]], dtype=torch.int32, device=DEV)
    w = torch.tensor([[[1., 1, 1], [2, 2, 2]]], device=DEV)
    assert torch.autograd.gradcheck(lambda f: pu.three_interpolate(f, idx, w), (feats,), eps=1e-2,
                                    atol=1e-1, rtol=1e-1, nondet_tol=1e-3)
    pu.three_interpolate(feats, idx, w).sum().backward()
    assert torch.equal(feats.grad[0, 0].cpu(), torch.tensor([1., 3, 3, 2]))


# ---- gather / group -----------------------------------------------------------
def test_gather_group_fwd_bwd(ext, oracle):
    rng = np.random.default_rng(5)
    b, c, n, m = 2, 19, 3000, 700
    feats = rng.standard_normal((b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, (b, m)).astype(np.int32)
    out = ext.p2.gather_points(dev(feats), dev(idx))
    assert np.array_equal(host(out), oracle.gather_points(feats, idx))
    # reference smoke check (openpoints/models/layers/subsample.py:159-185): == torch.gather
    tg = torch.gather(dev(feats), 2, dev(idx).long().unsqueeze(1).expand(-1, c, -1))
    assert torch.equal(out, tg)
    go = rng.standard_normal((b, c, m)).astype(np.float32)
    np.testing.assert_allclose(host(ext.p2.gather_points_grad(dev(go), dev(idx), n)),
                               oracle.gather_points_grad(go, idx, n), rtol=1e-4, atol=1e-5)
    gidx = rng.integers(0, n, (b, 250, 16)).astype(np.int32)
    gout = ext.p2.group_points(dev(feats), dev(gidx))
    assert np.array_equal(host(gout), oracle.group_points(feats, gidx))
    go = rng.standard_normal((b, c, 250, 16)).astype(np.float32)
    np.testing.assert_allclose(host(ext.p2.group_points_grad(dev(go), dev(gidx), n)),
                               oracle.group_points_grad(go, gidx, n), rtol=1e-4, atol=1e-4)
    o2 = torch.full((b, c, 250, 16), float("nan"), device=DEV)
    ext.p2b.group_points_wrapper(b, c, n, 250, 16, dev(feats), dev(gidx), o2)
    assert torch.equal(o2, gout)
    o3 = torch.full((b, c, m), float("nan"), device=DEV)
    ext.p2b.gather_points_wrapper(b, c, n, m, dev(feats), dev(idx), o3)
    assert torch.equal(o3, out)


# ---- kNN ----------------------------------------------------------------------
def test_knn_heap(ext, oracle):
    xyz, _ = make_batch(3, 500, start_index=21, dup_frac=0.1)
    flat = xyz.reshape(-1, 3)
    off = np.array([500, 1000, 1500])
    q = flat[::3].copy()
    noff = np.array([167, 334, 500])
    for k in (1, 3, 5, 16, 64):
        idx = torch.full((q.shape[0], k), -7, dtype=torch.int32, device=DEV)
        d2 = torch.full((q.shape[0], k), -1.0, device=DEV)
        ext.pops.knnquery_cuda(q.shape[0], k, dev(flat), dev(q), dev(off, torch.int32), dev(noff, torch.int32), idx, d2)
        wi, wd = oracle.knnquery_heap(k, flat, q, off, noff)
        assert np.array_equal(host(idx), wi) and np.array_equal(host(d2), wd)
    # fewer candidates than nsample
    idx = torch.zeros((2, 6), dtype=torch.int32, device=DEV)
    d2 = torch.zeros((2, 6), device=DEV)
    ext.pops.knnquery_cuda(2, 6, dev(flat[:4]), dev(flat[:2]), dev(np.array([4]), torch.int32),
                           dev(np.array([2]), torch.int32), idx, d2)
    wi, wd = oracle.knnquery_heap(6, flat[:4], flat[:2], np.array([4]), np.array([2]))
    assert np.array_equal(host(idx), wi) and np.array_equal(host(d2), wd)


def test_pointops_python_api(ext, oracle):
    from geot_amd.pointops.functions import pointops
    xyz, _ = make_batch(2, 1200, start_index=8, dup_frac=0.02)
    x = dev(xyz)
    pts = pointops.fps(x, 300)
    want = oracle.fps_offset(xyz.reshape(-1, 3), np.array([1200, 2400]), np.array([300, 600]))
    assert np.array_equal(host(pts), xyz.reshape(-1, 3)[want].reshape(2, 300, 3))
    # the backbone's three calls (8192/4096/2048 in the reference) reuse one run: prefix property
    with pointops.fps_prefix_scope():
        a, b2, c2 = pointops.fps(x, 300), pointops.fps(x, 150), pointops.fps(x, 75)
        assert pointops._FPS_SCOPE.cache["idx"].shape == (2, 300)        # one run served all three
        assert torch.equal(a[:, :150], b2) and torch.equal(a[:, :75], c2)
        assert np.array_equal(host(c2), xyz.reshape(-1, 3)[want.reshape(2, 300)[:, :75].reshape(-1)].reshape(2, 75, 3))
        x.mul_(1.0)  # in-place edit bumps the version: the cache must not be used
        assert pointops._FPS_SCOPE.cache["version"] != x._version
    assert pointops._FPS_SCOPE.cache is None                             # nothing is kept outside a scope
    assert torch.equal(pointops.fps(x, 150), b2)
    idx, dist = pointops.knn(x[:, :100].contiguous(), x, 5)
    wi, wd = oracle.knnquery_heap(5, xyz.reshape(-1, 3), xyz[:, :100].reshape(-1, 3), np.array([1200, 2400]),
                                  np.array([100, 200]))
    assert idx.dtype == torch.int64
    assert np.array_equal(host(idx).reshape(-1, 5), wi - np.repeat([0, 1200], 100)[:, None])
    np.testing.assert_allclose(host(dist).reshape(-1, 5), np.sqrt(wd), rtol=1e-6)


def test_knn_sorted_and_knn_cuda_module(ext, oracle, golden):
    from geot_amd.knn_cuda import KNN, knn_sorted
    xyz, _ = make_batch(2, 900, start_index=3, dup_frac=0.05)
    q = xyz[:, ::5].copy()
    for k in (1, 4, 32, 33):
        d2, idx = knn_sorted(dev(q), dev(xyz), k)
        wi, wd = oracle.knn_sorted(q, xyz, k)
        assert np.array_equal(host(idx), wi) and np.array_equal(host(d2), wd)
    d2, idx = knn_sorted(dev(q), dev(xyz[:, :2]), 4)  # fewer refs than k -> (inf, 0)
    wi, wd = oracle.knn_sorted(q, xyz[:, :2], 4)
    assert np.array_equal(host(idx), wi) and np.array_equal(host(d2), wd)
    dist, idx = KNN(4, transpose_mode=True)(dev(xyz), dev(q))
    assert idx.dtype == torch.int64 and tuple(idx.shape) == (2, q.shape[1], 4)
    wi, wd = oracle.knn_sorted(q, xyz, 4)
    assert np.array_equal(host(idx), wi)
    dist_t, idx_t = KNN(4, transpose_mode=False)(dev(xyz).transpose(1, 2).contiguous(), dev(q).transpose(1, 2).contiguous())
    assert tuple(idx_t.shape) == (2, 4, q.shape[1]) and torch.equal(idx_t.transpose(1, 2), idx)
    # the reference's own knn_point outputs (fixture from /root/reference, see make_golden.py)
    g = golden("knn_point_ref.npz")
    d2, idx = knn_sorted(dev(g["xyz"]), dev(g["xyz"]), int(g["k"]))
    assert np.array_equal(host(idx), g["idx"])
    assert np.abs(np.sqrt(host(d2)) - g["dist"]).max() < 2e-3


# ---- channels-last pointops ------------------------------------------------------
def test_channels_last_ops(ext, oracle):
    rng = np.random.default_rng(9)
    n, ns, c, w_c = 300, 8, 32, 8
    x = rng.standard_normal((n, c)).astype(np.float32)
    y = rng.standard_normal((n, c)).astype(np.float32)
    idx = rng.integers(0, n, (n, ns)).astype(np.int32)
    w = rng.random((n, ns)).astype(np.float32)
    pos = rng.standard_normal((n, ns, c)).astype(np.float32)
    ww = rng.random((n, ns, w_c)).astype(np.float32)
    go_nc = rng.standard_normal((n, c)).astype(np.float32)
    go_nsc = rng.standard_normal((n, ns, c)).astype(np.float32)
    P = ext.pops
    out = torch.full((n, ns, c), float("nan"), device=DEV)
    P.grouping_forward_cuda(n, ns, c, dev(x), dev(idx), out)
    assert np.array_equal(host(out), oracle.grouping_cl(x, idx))
    g = torch.zeros((n, c), device=DEV)
    P.grouping_backward_cuda(n, ns, c, dev(go_nsc), dev(idx), g)
    np.testing.assert_allclose(host(g), oracle.grouping_cl_grad(go_nsc, idx, n), rtol=1e-4, atol=1e-4)
    out = torch.zeros((n, c), device=DEV)
    P.interpolation_forward_cuda(n, c, ns, dev(x), dev(idx), dev(w), out)
    np.testing.assert_allclose(host(out), oracle.interpolation_cl(x, idx, w), rtol=RTOL, atol=1e-6)
    g = torch.zeros((n, c), device=DEV)
    P.interpolation_backward_cuda(n, c, ns, dev(go_nc), dev(idx), dev(w), g)
    np.testing.assert_allclose(host(g), oracle.interpolation_cl_grad(go_nc, idx, w, n), rtol=1e-4, atol=1e-4)
    out = torch.full((n, ns, c), float("nan"), device=DEV)
    P.subtraction_forward_cuda(n, ns, c, dev(x), dev(y), dev(idx), out)
    assert np.array_equal(host(out), oracle.subtraction_cl(x, y, idx))
    g1, g2 = torch.zeros((n, c), device=DEV), torch.zeros((n, c), device=DEV)
    P.subtraction_backward_cuda(n, ns, c, dev(idx), dev(go_nsc), g1, g2)
    w1, w2 = oracle.subtraction_cl_grad(idx, go_nsc)
    np.testing.assert_allclose(host(g1), w1, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(g2), w2, rtol=1e-4, atol=1e-4)
    out = torch.zeros((n, c), device=DEV)
    P.aggregation_forward_cuda(n, ns, c, w_c, dev(x), dev(pos), dev(ww), dev(idx), out)
    np.testing.assert_allclose(host(out), oracle.aggregation_cl(x, pos, ww, idx), rtol=RTOL, atol=1e-5)
    gi, gp, gw = torch.zeros((n, c), device=DEV), torch.zeros((n, ns, c), device=DEV), torch.zeros((n, ns, w_c), device=DEV)
    P.aggregation_backward_cuda(n, ns, c, w_c, dev(x), dev(pos), dev(ww), dev(idx), dev(go_nc), gi, gp, gw)
    wi, wp, wwg = oracle.aggregation_cl_grad(x, pos, ww, idx, go_nc)
    np.testing.assert_allclose(host(gi), wi, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(gp), wp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(gw), wwg, rtol=1e-4, atol=1e-4)


# ---- wrappers: autograd + stream semantics ------------------------------------------
def test_query_and_group_matches_composition(ext, oracle):
    from geot_amd.pointnet2 import pointnet2_utils as pu
    xyz, _ = make_batch(2, 2000, start_index=12)
    x = dev(xyz)
    feats = torch.randn(2, 5, 2000, device=DEV, requires_grad=True)
    inds = pu.furthest_point_sample(x, 128)
    assert inds.dtype == torch.int32 and not inds.requires_grad
    new_xyz = pu.gather_operation(x.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    grouper = pu.QueryAndGroup(0.25, 16, use_xyz=True)
    out = grouper(x, new_xyz, feats)
    assert tuple(out.shape) == (2, 8, 128, 16)
    bq = oracle.ball_query(host(new_xyz), xyz, 0.25, 16)
    gx = oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), bq) - host(new_xyz).transpose(0, 2, 1)[..., None]
    gf = oracle.group_points(host(feats), bq)
    np.testing.assert_allclose(host(out), np.concatenate([gx, gf], 1), rtol=RTOL, atol=1e-6)
    out.sum().backward()
    cnt = np.zeros((2, 2000), dtype=np.float32)
    for b in range(2):
        np.add.at(cnt[b], bq[b].reshape(-1), 1.0)
    np.testing.assert_allclose(host(feats.grad), np.repeat(cnt[:, None, :], 5, 1), rtol=1e-5)


def test_ops_run_on_the_current_stream(ext, oracle):
    xyz, _ = make_batch(1, 3000, start_index=31)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x = dev(xyz)
        got = ext.p2.furthest_point_sampling(x, 200)
        d2, idx = ext.p2.three_nn(x, x[:, :500].contiguous())
    s.synchronize()
    assert np.array_equal(host(got), oracle.fps_dense(xyz, 200, 512, True))
    assert np.array_equal(host(idx), oracle.three_nn(xyz, xyz[:, :500])[1])


def test_bad_arguments_raise(ext):
    x = torch.zeros(1, 8, 3, device=DEV)
    with pytest.raises(RuntimeError):
        ext.p2.furthest_point_sampling(x.double(), 4)
    with pytest.raises(RuntimeError):
        ext.p2.furthest_point_sampling(x.transpose(1, 2), 4)
    with pytest.raises(RuntimeError):
        ext.p2.gather_points(torch.zeros(1, 2, 8, device=DEV), torch.zeros(1, 4, dtype=torch.int64, device=DEV))
    with pytest.raises(RuntimeError):
        ext.p2.ball_query(x, x, 0.1, 10 ** 6)  # nsample beyond the LDS budget -> hipErrorInvalidValue


# ---- grid kNN: bit-identical to the brute-force kernels -----------------------------------------
def _adversarial_clouds(rng, n):
    """Reference sets that stress the grid: volume, surface, clusters, a line, a plane with exact ties
    (lattice), heavy duplicates, one far outlier (almost every cell empty), tiny extent."""
    u = rng.random((n, 3)).astype(np.float32)
    sph = rng.standard_normal((n, 3)).astype(np.float32)
    sph /= np.linalg.norm(sph, axis=1, keepdims=True)
    clus = (rng.integers(0, 5, (n, 1)) * 0.2 + rng.standard_normal((n, 3)) * 0.003).astype(np.float32)
    line = np.zeros((n, 3), np.float32); line[:, 0] = np.linspace(-1, 1, n, dtype=np.float32)
    side = int(np.ceil(np.sqrt(n)))
    gx, gy = np.meshgrid(np.arange(side, dtype=np.float32), np.arange(side, dtype=np.float32))
    lattice = np.stack([gx.ravel()[:n] / 64, gy.ravel()[:n] / 64, np.zeros(n, np.float32)], 1)   # exact ties
    dup = u.copy(); dup[n // 2:] = dup[rng.integers(0, n // 2, n - n // 2)]
    outl = u * 0.01; outl[7] = (50.0, -30.0, 10.0)
    tiny = (1.0 + u * 1e-6).astype(np.float32)
    return {"volume": u, "sphere": sph, "clusters": clus, "line": line, "lattice": lattice, "dups": dup,
            "outlier": outl.astype(np.float32), "tiny": tiny}


@pytest.mark.parametrize("k", [1, 3, 4, 33, 64])
def test_knn_grid_matches_bruteforce_bit_for_bit(ext, k, monkeypatch):
    from geot_amd.knn_cuda import knn_sorted
    from geot_amd import _lib
    rng = np.random.default_rng(100 + k)
    n = 6000
    clouds = _adversarial_clouds(rng, n)
    names = list(clouds)
    ref = np.stack([clouds[c] for c in names])                          # (8, n, 3)
    # queries: the references themselves (self-kNN), points far outside the box, and random ones
    q_out = (rng.standard_normal((len(names), 200, 3)) * 5).astype(np.float32)
    qry = np.concatenate([ref[:, :1500], q_out, ref[:, -300:] + np.float32(1e-3)], 1)
    monkeypatch.setenv("GEOT_NN_IMPL", "grid")
    assert _lib.load().geot_knn_grid_eligible(len(names), qry.shape[1], n, k) == 1
    d_g, i_g = knn_sorted(dev(qry), dev(ref), k)
    monkeypatch.setenv("GEOT_NN_IMPL", "wave")
    assert _lib.load().geot_knn_grid_eligible(len(names), qry.shape[1], n, k) == 0
    d_b, i_b = knn_sorted(dev(qry), dev(ref), k)
    for c, name in enumerate(names):
        assert torch.equal(i_g[c], i_b[c]), name
        assert torch.equal(d_g[c], d_b[c]), name


def test_knn_grid_full_size_vs_oracle_and_three_nn(ext, oracle):
    from geot_amd.knn_cuda import knn_sorted
    from geot_amd import _lib
    xyz, _ = make_batch(2, 24000, start_index=5, dup_frac=0.01)
    d, i = knn_sorted(dev(xyz[:, :3000]), dev(xyz), 33)                 # threeD_space_loss shape, a slice of queries
    wi, wd = oracle.knn_sorted(xyz[:, :3000], xyz, 33)
    assert np.array_equal(host(i), wi) and np.array_equal(host(d), wd)
    known = xyz[:, ::3].copy()                                          # 8000 known points: grid path for k = 3
    assert _lib.load().geot_knn_grid_eligible(2, 24000, 8000, 3) == 1
    d3, i3 = ext.p2.three_nn(dev(xyz), dev(known))
    wd3, wi3 = oracle.three_nn(xyz, known)
    assert np.array_equal(host(i3), wi3) and np.array_equal(host(d3), wd3)
    # fewer references than k, NaN / Inf coordinates: same (inf, 0) padding as the brute-force kernel
    few = xyz[:, :2100].copy()
    few[0, 5] = np.nan; few[1, 9, 0] = np.inf
    q = xyz[:, :2000].copy(); q[0, 3] = np.nan; q[1, 4, 2] = -np.inf
    import os
    os.environ["GEOT_NN_IMPL"] = "grid"
    try:
        dg, ig = knn_sorted(dev(q), dev(few), 64)
        os.environ["GEOT_NN_IMPL"] = "wave"
        db, ib = knn_sorted(dev(q), dev(few), 64)
    finally:
        del os.environ["GEOT_NN_IMPL"]
    assert torch.equal(ig, ib) and torch.equal(torch.nan_to_num(dg, nan=-1.0), torch.nan_to_num(db, nan=-1.0))


@pytest.mark.parametrize("radius,ns", [(0.1, 32), (0.05, 16), (0.3, 64), (0.02, 8), (2.5, 32)])
def test_ball_query_grid_matches_bruteforce_bit_for_bit(ext, oracle, radius, ns, monkeypatch):
    """Grid ball query == linear-scan kernel on adversarial reference sets (sparse, dense blocks that overflow
    the register slots, lattices, duplicates, an outlier, queries outside the box); one case vs the oracle."""
    rng = np.random.default_rng(int(radius * 1000) + ns)
    n = 6000
    clouds = _adversarial_clouds(rng, n)
    names = list(clouds)
    ref = np.stack([clouds[c] for c in names])
    q_out = (rng.standard_normal((len(names), 150, 3)) * 3).astype(np.float32)
    qry = np.ascontiguousarray(np.concatenate([ref[:, :1200], q_out, ref[:, -200:] + np.float32(1e-3)], 1))
    monkeypatch.setenv("GEOT_NN_IMPL", "grid")
    got = ext.p2.ball_query(dev(qry), dev(ref), radius, ns)
    monkeypatch.setenv("GEOT_NN_IMPL", "wave")
    want = ext.p2.ball_query(dev(qry), dev(ref), radius, ns)
    for c, name in enumerate(names):
        assert torch.equal(got[c], want[c]), name
    assert np.array_equal(host(got[:2, :300]), oracle.ball_query(qry[:2, :300], ref[:2], radius, ns))


def test_fps_multi_commit_soak_against_unpruned_kernel(ext, monkeypatch):
    """64 different clouds (sizes across all four launch geometries, with duplicates): the multi-commit pruned
    kernel must agree with the unpruned one on every index and on the final min-distance buffer."""
    rng = np.random.default_rng(2024)
    for n, m, cnt in ((3000, 700, 16), (7000, 1500, 16), (14000, 2500, 16), (24000, 3000, 16)):
        xyz = np.stack([make_cloud(n, 500 + n + i, dup_frac=0.02 * (i % 3))[0] for i in range(cnt)])
        if n == 14000:
            xyz[:, :, 2] *= 0.01                                  # nearly flat clouds
        monkeypatch.setenv("GEOT_FPS_IMPL", "multi")
        a, ta = fps_k1p(ext, xyz, m, return_temp=True)
        monkeypatch.setenv("GEOT_FPS_IMPL", "basic")
        b_, tb = fps_k1p(ext, xyz, m, return_temp=True)
        assert np.array_equal(a, b_) and np.array_equal(ta, tb), (n, m)


@pytest.mark.parametrize("b,c,m,L", [(3, 19, 777, 4001), (1, 384, 8192, 24000), (2, 70, 36864, 1500), (2, 33, 5, 3000),
                                     (1, 16, 9000, 4100)])
def test_lds_table_paths_match_plain_kernels(ext, b, c, m, L, monkeypatch):
    """table-in-LDS gather (forward) and reverse-index gather (backward) against the register-gather / atomic
    kernels on odd shapes: forward bit for bit, backward to summation order."""
    rng = np.random.default_rng(b * 1000 + c)
    feat = dev(rng.standard_normal((b, c, m)).astype(np.float32))
    idx3 = dev(rng.integers(0, m, (b, L, 3)).astype(np.int32))
    w3 = dev(rng.random((b, L, 3)).astype(np.float32))
    g = dev(rng.standard_normal((b, c, L)).astype(np.float32))
    ns = 4
    idxg = dev(rng.integers(0, m, (b, L // ns, ns)).astype(np.int32))
    gg = dev(rng.standard_normal((b, c, L // ns, ns)).astype(np.float32))
    idx1 = dev(rng.integers(0, m, (b, L)).astype(np.int32))

    def run():
        return (ext.p2.three_interpolate(feat, idx3, w3), ext.p2.group_points(feat, idxg), ext.p2.gather_points(feat, idx1),
                ext.p2.three_interpolate_grad(g, idx3, w3, m), ext.p2.group_points_grad(gg, idxg, m))
    fast = run()
    monkeypatch.setenv("GEOT_GATHER_IMPL", "plain")
    plain = run()
    for k in range(3):
        assert torch.equal(fast[k], plain[k]), k
    for k in (3, 4):
        np.testing.assert_allclose(host(fast[k]), host(plain[k]), rtol=1e-4, atol=1e-4 * float(plain[k].abs().max()))


@pytest.mark.parametrize("k", [2, 6, 17])
def test_pointops_knn_fast_path_equals_literal_heap(ext, k):
    """pointops.knn on equal segments (grid search + tie certification + heap for the tied queries) must equal
    the literal max-heap kernel element for element, duplicates and lattice ties included."""
    from geot_amd.pointops.functions import pointops
    from geot_amd import _lib
    rng = np.random.default_rng(31 + k)
    B, N = 2, 6000
    xyz, _ = make_batch(B, N, start_index=60, dup_frac=0.05)
    xyz[1] = (xyz[1] * 48).round() / 48                                  # quantised: many exact distance ties
    x = dev(xyz)
    assert _lib.load().geot_knn_grid_eligible(B, N, N, k + 1) == 1
    idx, dist = pointops.knn(x, x, k)
    flat = x.reshape(-1, 3).contiguous()
    off = dev(np.array([N, 2 * N]), torch.int32)
    widx = torch.zeros((B * N, k), dtype=torch.int32, device=DEV)
    wd2 = torch.zeros((B * N, k), dtype=torch.float32, device=DEV)
    ext.pops.knnquery_cuda(B * N, k, flat, flat, off, off, widx, wd2)
    want_local = widx.view(B, N, k).long() - (torch.arange(B, device=DEV) * N)[:, None, None]
    assert torch.equal(idx, want_local)
    assert torch.equal(dist, torch.sqrt(wd2).view(B, N, k))


@pytest.mark.parametrize("d,k", [(17, 8), (5, 3), (32, 33), (20, 64)])
def test_knn_sorted_nd_matches_numpy_restatement(ext, d, k):
    """Feature-space kNN (feature_space_loss's neighbours): ids and squared distances bit-exact against the
    numpy restatement, duplicates (exact ties) included."""
    from oracle import np_ref
    from geot_amd.openpoints.models.layers.knn import knn_point
    rng = np.random.default_rng(d * 100 + k)
    b, nq, nr = 2, 300, 900
    ref = rng.random((b, nr, d)).astype(np.float32)
    ref[:, 500:600] = ref[:, 100:200]                                    # duplicates: ties broken by index
    qry = np.concatenate([ref[:, :200], rng.random((b, nq - 200, d)).astype(np.float32)], 1)
    dist, idx = knn_point(k, dev(qry), dev(ref))
    wi, wd = np_ref.knn_sorted_nd(qry, ref, k)
    assert idx.dtype == torch.int64 and np.array_equal(host(idx), wi)
    np.testing.assert_array_equal(host(dist), np.sqrt(wd))


@pytest.mark.parametrize("B,n,m,c,cs,skip_first", [(2, 3000, 700, 20, 7, False), (1, 24000, 8192, 64, 32, True),
                                                    (2, 513, 64, 5, 0, False), (3, 1000, 3, 33, 4, True)])
def test_fp_front_end_fused_matches_chain(B, n, m, c, cs, skip_first):
    """FPInterpolateConcat (three_nn -> weights -> interpolate -> concat in place) against the reference's
    op-by-op chain on the same HIP ops: forward values identical, both gradients within the scatter tolerance."""
    from geot_amd.pointnet2 import pointnet2_utils as pu
    g = torch.Generator().manual_seed(B * n + m)
    dev = torch.device("cuda:0")
    unknown, known = torch.rand(B, n, 3, generator=g).to(dev), torch.rand(B, m, 3, generator=g).to(dev)
    kf = torch.randn(B, c, m, generator=g).to(dev).requires_grad_(True)
    uf = torch.randn(B, cs, n, generator=g).to(dev).requires_grad_(True) if cs else None
    up = torch.randn(B, c + cs, n, generator=g).to(dev)
    wide = pu.fp_interpolate_concat(unknown, known, uf, kf, skip_first)
    wide.backward(up)
    got_k, got_u = kf.grad.clone(), (uf.grad.clone() if cs else None)
    kf.grad = None
    if cs:
        uf.grad = None
    dist, idx = pu.three_nn(unknown, known)
    r = 1.0 / (dist + 1e-8)
    w = r / torch.sum(r, dim=2, keepdim=True)
    inter = pu.three_interpolate(kf, idx, w)
    want = inter if not cs else torch.cat([uf, inter] if skip_first else [inter, uf], dim=1)
    want.backward(up)
    np.testing.assert_allclose(wide.detach().cpu().numpy(), want.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(got_k.cpu().numpy(), kf.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
    if cs:
        assert torch.equal(got_u, uf.grad)
    # and against the oracle's interpolation of the oracle's neighbours
    from oracle import capi, np_ref
    d2, oi = capi.three_nn(unknown.cpu().numpy(), known.cpu().numpy())
    ow = 1.0 / (np.sqrt(d2) + np.float32(1e-8))
    ow = (ow / ow.sum(2, keepdims=True)).astype(np.float32)
    oref = np_ref.three_interpolate(kf.detach().cpu().numpy(), oi, ow)
    lo = cs if skip_first else 0
    np.testing.assert_allclose(wide.detach().cpu().numpy()[:, lo:lo + c], oref, rtol=1e-5, atol=1e-6)


def test_fp_modules_fused_front_end_equals_chain():
    from geot_amd.openpoints.models.backbone.pointnetv2 import PointNetFPModule
    from geot_amd.pointnet2.pointnet2_modules import PointnetFPModule
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    unknown, known = torch.rand(2, 2000, 3, device=dev), torch.rand(2, 500, 3, device=dev)
    uf, kf = torch.randn(2, 6, 2000, device=dev), torch.randn(2, 10, 500, device=dev)
    for mod in (PointnetFPModule([16, 32, 8]).to(dev).eval(), PointNetFPModule([16, 32, 8]).to(dev).eval()):
        assert mod.fused_front_end
        a = mod(unknown, known, uf, kf)
        mod.fused_front_end = False
        b = mod(unknown, known, uf, kf)
        np.testing.assert_allclose(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_debug_mode_rejects_out_of_range_indices(ext, monkeypatch):
    """GEOT_DEBUG=1 turns an out-of-range gather index into an IndexError (the reference would read out of
    bounds); without the flag nothing is checked and no host sync happens."""
    feats = torch.randn(2, 4, 100, device=DEV)
    good = torch.randint(0, 100, (2, 30), device=DEV, dtype=torch.int32)
    bad = good.clone()
    bad[1, 7] = 100
    monkeypatch.setenv("GEOT_DEBUG", "1")
    assert ext.p2.gather_points(feats, good).shape == (2, 4, 30)
    with pytest.raises(IndexError, match="gather_points"):
        ext.p2.gather_points(feats, bad)
    neg = torch.full((2, 5, 3), -1, dtype=torch.int32, device=DEV)
    with pytest.raises(IndexError, match="three_interpolate"):
        ext.p2.three_interpolate(feats, neg, torch.ones(2, 5, 3, device=DEV))
    with pytest.raises(IndexError, match="group_points"):
        ext.p2.group_points(feats, bad.view(2, 10, 3))


@pytest.mark.parametrize("m", [1, 2])
def test_three_nn_with_fewer_than_three_known_points(ext, oracle, m):
    """The reference kernel's initial values survive in the unused slots (index 0, distance 1e40 -> inf in fp32);
    every output element is written (the wrappers hand the kernels uninitialised buffers)."""
    rng = np.random.default_rng(m)
    q, k = rng.random((2, 300, 3)).astype(np.float32), rng.random((2, m, 3)).astype(np.float32)
    junk = torch.full((2, 300, 3), -12345.0, device=DEV)     # poison the allocator's free list
    del junk
    d2, idx = ext.p2.three_nn(dev(q), dev(k))
    wd, wi = oracle.three_nn(q, k)
    assert np.array_equal(host(idx), wi) and np.array_equal(host(d2), wd)
    assert np.isinf(host(d2)[:, :, m:]).all() and (host(idx)[:, :, m:] == 0).all()


def test_fps_exhausted_valid_points_repeat_the_smallest_key(ext, oracle, fps_impl):
    """Found by tools/fuzz_parity.py (seed 7, case 57): a cloud whose points almost all fall inside the K1 kernel's
    origin-skip ball has fewer valid points than samples; once they are exhausted every min-distance is 0 and the
    reference picks the smallest-key point again and again.  With one valid point per wave the multi-commit
    resolve used to commit the next wave's zero-valued candidate behind it."""
    rng = np.random.default_rng(7 * 100003 + 57)
    xyz = (rng.standard_normal((3, 2500, 3)) * 0.01).astype(np.float32)
    assert ((xyz ** 2).sum(-1) > 1e-3).sum(1).max() < 100
    assert np.array_equal(fps_k1(ext, xyz, 279), oracle.fps_dense(xyz, 279, 512, True))
    # the same situation spread thinly: exactly one valid point in a few waves' share of the sorted cloud
    xyz2 = np.zeros((2, 3000, 3), dtype=np.float32)
    far = rng.choice(3000, 9, replace=False)
    xyz2[:, far] = (rng.random((2, 9, 3)).astype(np.float32) + 0.5)
    got = fps_k1(ext, xyz2, 40)
    assert np.array_equal(got, oracle.fps_dense(xyz2, 40, 512, True))


def test_sliced_index_gather_gradient_matches_the_per_target_walk(monkeypatch):
    """GEOT_GATHER_IMPL=sell (opt-in: length-sorted sliced copy of the reverse index, csrc/gather_group.hip
    table_gather_sell_kernel) against the default per-target walk and a float64 scatter-add, at a shape that cuts the
    sources into parts, with hub targets (thousands of sources: the one-wave-per-list path), targets without sources,
    a ragged channel count; bit-identical from call to call."""
    from geot_amd.ext import pointnet2_ext as p2
    from geot_amd.synth import make_batch
    b, c, n, m = 2, 70, 24000, 8000
    xyz = torch.from_numpy(make_batch(b, n, start_index=41)[0]).to(DEV)
    _, idx = p2.three_nn(xyz, xyz[:, :m].contiguous())
    idx = idx.clone()
    idx[:, ::7, 0] = 5                      # ~3400 sources on one target
    idx[0, :, 2] = idx[0, :, 2] % 64        # 375 sources on each of 64 targets
    idx[:, 1::50, 1] = m - 1
    gen = torch.Generator(device=DEV).manual_seed(3)
    w = torch.rand(b, n, 3, device=DEV, generator=gen)
    g = torch.randn(b, c, n, device=DEV, generator=gen)
    ref = torch.zeros(b, c, m, dtype=torch.float64, device=DEV)
    ref.scatter_add_(2, idx.long().reshape(b, 1, n * 3).expand(-1, c, -1),
                     (g.double().unsqueeze(-1) * w.double().unsqueeze(1)).reshape(b, c, n * 3))
    outs = {}
    for impl in ("sell", "l"):
        monkeypatch.setenv("GEOT_GATHER_IMPL", impl)
        outs[impl] = p2.three_interpolate_grad(g, idx.contiguous(), w, m)
        assert torch.equal(outs[impl], p2.three_interpolate_grad(g, idx.contiguous(), w, m)), impl
        err = float((outs[impl].double() - ref).abs().max() / ref.abs().max())
        assert err < 2e-6, (impl, err)
    assert float((outs["sell"] - outs["l"]).abs().max() / outs["l"].abs().max()) < 2e-6
