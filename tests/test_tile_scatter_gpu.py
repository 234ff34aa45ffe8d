"""The channels-first gradients behind the reference API (geot_amd/csrc/tile_scatter.hip): three_interpolate_grad
(pointnet2/_ext_src/src/interpolate_gpu.cu:119-146), group_points_grad (group_points_gpu.cu:46-67), gather_points_grad
(sampling_gpu.cu:35-50) and the kNN graph feature's neighbour gradient -- against an fp64 scatter-add, and call-to-call
bit for bit (one writer per output element, one summation order)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def rel(got, want64):
    """largest error relative to the scale of the element's row (the tolerance rule of tests/test_ref_fixtures_gpu.py)"""
    got, want64 = got.double().cpu(), want64.double().cpu()
    scale = want64.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    return float(((got - want64).abs() / scale).max())


def scatter64(g, idx, w, m):
    """fp64 restatement: out[b, c, idx[b, e, t]] += w[b, e, t] * g[b, c, e]"""
    b, c, L = g.shape
    nt = idx.shape[-1]
    src = g.double().cpu().unsqueeze(-1) * (w.double().cpu().unsqueeze(1) if w is not None else 1.0)
    out = torch.zeros(b, c, m, dtype=torch.float64)
    out.scatter_add_(2, idx.long().cpu().reshape(b, 1, L * nt).expand(-1, c, -1), src.reshape(b, c, L * nt).expand(b, c, L * nt))
    return out


def case(b, c, n, m, seed, hubs=False, dup_slots=False):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, m, (b, n, 3))
    if hubs:                                   # a few targets collect most pairs, the last target none
        idx[:, : n // 2] = rng.integers(0, min(3, m), (b, n // 2, 3))
        idx[idx == m - 1] = 0
    if dup_slots:                              # the same target twice in one source's three slots
        idx[:, ::5, 1] = idx[:, ::5, 0]
    w = rng.random((b, n, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    g = rng.standard_normal((b, c, n)).astype(np.float32)
    return dev(g), dev(idx.astype(np.int32)), dev(w)


SHAPES = [  # b, c, n (sources), m (targets), hubs
    (2, 64, 24000, 8192, False),     # propogation_0's shape, 4 channels per workgroup, 25 tiles
    (1, 7, 3000, 8, False),          # the small-m shape whose gradient used to depend on atomic arrival order
    (2, 5, 1001, 1, False),          # one target; n not a multiple of 4 (scalar staging loads)
    (3, 3, 777, 300, False),         # fewer channels than a slab
    (1, 1, 50, 20, False),
    (2, 33, 8192, 512, True),        # hub targets (runs longer than a wave), an untouched target
    (1, 18, 5000, 20000, False),     # more targets than 4 channels' sums fit: 1 channel per workgroup
    (2, 16, 4099, 12000, False),     # 2 channels per workgroup, odd n
]


@pytest.mark.parametrize("b,c,n,m,hubs", SHAPES)
def test_three_interpolate_grad_is_exact_and_reproducible(b, c, n, m, hubs):
    from geot_amd.ext import pointnet2_ext as p2
    from geot_amd import _lib
    g, idx, w = case(b, c, n, m, 7 * b + c + n, hubs, dup_slots=True)
    assert _lib.load().geot_grad_ws_needs_zero(b, c, m, n, 3) == 0          # the atomic-free form takes this shape
    want = scatter64(g, idx, w, m)
    got = p2.three_interpolate_grad(g, idx, w, m)
    assert got.shape == (b, c, m)
    assert rel(got, want) <= 1e-5 * (30 if hubs else 1)                     # a hub sums thousands of fp32 terms
    for _ in range(3):
        assert torch.equal(p2.three_interpolate_grad(g, idx, w, m), got)
    if hubs:
        assert float(got[:, :, m - 1].abs().max()) == 0.0                    # untouched targets are written


@pytest.mark.parametrize("b,c,n,m,hubs", SHAPES[:4])
def test_batch_wrapper_accumulates_into_a_prefilled_buffer(b, c, n, m, hubs):
    """pointnet2_batch_cuda convention (interpolate.cpp / group_points.cpp): *_grad outputs arrive pre-filled and are added to."""
    from geot_amd.ext import pointnet2_batch_cuda as pb
    g, idx, w = case(b, c, n, m, 11 * b + c, hubs)
    torch.manual_seed(b + c + m)
    base = torch.randn(b, c, m, device=DEV)
    out = base.clone()
    pb.three_interpolate_grad_wrapper(b, c, n, m, g, idx, w, out)
    want = scatter64(g, idx, w, m)
    # (error against the magnitude of the two addends: with one target per row the sum itself may cancel to nothing)
    scale = (want.abs() + base.double().cpu().abs()).amax(dim=-1, keepdim=True).clamp_min(1e-30)
    assert float(((out.double().cpu() - (want + base.double().cpu())).abs() / scale).max()) <= 1e-5
    again = base.clone()
    pb.three_interpolate_grad_wrapper(b, c, n, m, g, idx, w, again)
    assert torch.equal(out, again)


@pytest.mark.parametrize("b,c,n,npoint,ns", [(2, 64, 24000, 1500, 32), (1, 20, 4096, 512, 16), (2, 3, 100, 7, 5), (1, 130, 30000, 700, 8)])
def test_group_and_gather_gradients(b, c, n, npoint, ns):
    from geot_amd.ext import pointnet2_ext as p2
    from geot_amd import _lib
    rng = np.random.default_rng(b + c + n)
    idx = dev(rng.integers(0, n, (b, npoint, ns)).astype(np.int32))
    g = dev(rng.standard_normal((b, c, npoint, ns)).astype(np.float32))
    assert _lib.load().geot_grad_ws_needs_zero(b, c, n, npoint * ns, 1) == 0
    got = p2.group_points_grad(g, idx, n)
    want = scatter64(g.reshape(b, c, -1), idx.reshape(b, -1, 1), None, n)
    assert rel(got, want) <= 1e-5
    assert torch.equal(p2.group_points_grad(g, idx, n), got)
    # the same ids as a plain gather (nsample folded into the sample dimension)
    got1 = p2.gather_points_grad(g.reshape(b, c, -1).contiguous(), idx.reshape(b, -1).contiguous(), n)
    assert torch.equal(got1, got)


def test_wide_gradient_and_graph_feature_take_the_same_path():
    """three_interpolate_grad_from reads c channels of a wider gradient (batch stride); the kNN graph feature's neighbour
    gradient reads the first c of 2c channels."""
    from geot_amd.ext import pointnet2_ext as p2
    from geot_amd.openpoints.models.backbone.transformer_ops import graph_feature
    b, c, n, m, skip = 2, 24, 5000, 1200, 5
    g, idx, w = case(b, c + skip, n, m, 99)
    got = p2.three_interpolate_grad_from(g, c, idx, w, m, ch_offset=0)
    assert rel(got, scatter64(g[:, :c], idx, w, m)) <= 1e-5
    got2 = p2.three_interpolate_grad_from(g, c, idx, w, m, ch_offset=skip)
    assert rel(got2, scatter64(g[:, skip:], idx, w, m)) <= 1e-5
    nq, nk, k = 3000, 1000, 4
    rng = np.random.default_rng(5)
    xq = dev(rng.standard_normal((b, c, nq)).astype(np.float32)).requires_grad_(True)
    xk = dev(rng.standard_normal((b, c, nk)).astype(np.float32)).requires_grad_(True)
    kidx = dev(rng.integers(0, nk, (b, nq, k)).astype(np.int32))
    go = dev(rng.standard_normal((b, 2 * c, nq, k)).astype(np.float32))
    graph_feature(xq, xk, kidx).backward(go)
    want_k = scatter64(go[:, :c].reshape(b, c, -1), kidx.reshape(b, -1, 1), None, nk)
    assert rel(xk.grad, want_k) <= 1e-5
    first = xk.grad.clone()
    xk.grad = None
    xq.grad = None
    graph_feature(xq, xk, kidx).backward(go)
    assert torch.equal(xk.grad, first)


def test_older_forms_agree(monkeypatch):
    """GEOT_GATHER_IMPL selects the per-target list walk (csr) / the atomic kernels (plain) for A/B runs."""
    from geot_amd.ext import pointnet2_ext as p2
    b, c, n, m = 2, 32, 6000, 2048
    g, idx, w = case(b, c, n, m, 3)
    want = scatter64(g, idx, w, m)
    got = p2.three_interpolate_grad(g, idx, w, m)
    for impl in ("csr", "plain"):
        monkeypatch.setenv("GEOT_GATHER_IMPL", impl)
        other = p2.three_interpolate_grad(g, idx, w, m)
        assert rel(other, want) <= 1e-5
        assert rel(got, other.double()) <= 2e-6
