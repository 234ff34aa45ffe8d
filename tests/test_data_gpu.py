"""Dataloader-side ops on the GPU against the oracle and the reference-generated fixture.
grid_subsampling: every output bit-exact (fp32 sums are taken in input order, as the reference's hash map
does).  pc_norm / prepare_sample: 1e-5 on the normalised scale (fp32, different summation tree)."""
import os

import numpy as np
import pytest
import torch

from oracle import np_data

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "grid_subsampling_ref.npz")


def _cloud(seed, n, fdim, ldim, scale=(1, 0.7, 0.4), shift=(0, 0, 0)):
    rng = np.random.default_rng(seed)
    p = (rng.standard_normal((n, 3)) * np.array(scale) + np.array(shift)).astype(np.float32)
    if n > 100:
        p[rng.integers(0, n, n // 50)] = p[rng.integers(0, n, n // 50)]
    f = rng.standard_normal((n, fdim)).astype(np.float32) if fdim else None
    lab = rng.integers(-2, 6, (n, ldim)).astype(np.int32) if ldim else None
    return p, f, lab


def _run(p, f, lab, dl):
    from geot_amd.openpoints.dataset import grid_subsampling
    dev = torch.device("cuda:0")
    t = lambda a: None if a is None else torch.from_numpy(a).to(dev)
    res = grid_subsampling(t(p), t(f), t(lab), sampleDl=dl)
    res = (res,) if isinstance(res, torch.Tensor) else res
    out = [r.cpu().numpy() for r in res]
    got = {"points": out.pop(0)}
    got["features"] = out.pop(0) if f is not None else None
    got["labels"] = out.pop(0) if lab is not None else None
    return got


@pytest.mark.parametrize("seed,n,dl,fdim,ldim", [
    (0, 5000, 0.1, 0, 0), (1, 20000, 0.05, 4, 1), (2, 3000, 0.3, 2, 2), (3, 1, 0.1, 1, 1), (4, 60, 10.0, 1, 1),
    (5, 40000, 0.013, 1, 1), (6, 2, 0.1, 0, 1), (7, 257, 0.02, 3, 0), (8, 4097, 0.5, 0, 3)])
def test_grid_subsampling_bit_exact(seed, n, dl, fdim, ldim):
    p, f, lab = _cloud(seed, n, fdim, ldim)
    want = np_data.grid_subsampling(p, f, lab, dl)
    got = _run(p, f, lab, dl)
    assert got["points"].shape == want["points"].shape
    assert np.array_equal(got["points"], want["points"])
    if fdim:
        assert np.array_equal(got["features"], want["features"])
    if ldim:
        assert np.array_equal(got["labels"], want["labels"])


def test_grid_subsampling_scan_sized_cloud_far_from_origin():
    # a raw intra-oral scan: ~1.2e5 vertices in millimetres, far from the origin, some cells crowded
    p, f, lab = _cloud(11, 120000, 1, 1, scale=(30, 20, 8), shift=(500, -400, 120))
    want = np_data.grid_subsampling(p, f, lab, 8.0)
    got = _run(p, f, lab, 8.0)
    assert np.array_equal(got["points"], want["points"]) and np.array_equal(got["features"], want["features"])
    assert np.array_equal(got["labels"], want["labels"])
    assert want["count"].max() > 64


def test_grid_subsampling_one_voxel_and_numpy_io():
    from geot_amd.openpoints.dataset import grid_subsampling
    p, f, lab = _cloud(12, 3000, 2, 1, shift=(10, 10, 10))     # all coordinates positive: a single voxel
    pts, feats, labs = grid_subsampling(p, features=f, labels=lab[:, 0], sampleDl=1e4)   # numpy in -> numpy out
    assert isinstance(pts, np.ndarray) and pts.shape == (1, 3) and feats.shape == (1, 2) and labs.shape == (1, 1)
    want = np_data.grid_subsampling(p, f, lab, 1e4)
    assert np.array_equal(pts, want["points"]) and np.array_equal(feats, want["features"])
    assert np.array_equal(labs, want["labels"])
    only = grid_subsampling(p, sampleDl=0.2)
    assert isinstance(only, np.ndarray) and np.array_equal(only, np_data.grid_subsampling(p, None, None, 0.2)["points"])


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_grid_subsampling_matches_reference_fixture(case):
    """Against what the reference's own compiled code returned (rows there are in hash-map order)."""
    z = np.load(GOLD)
    g = lambda k: z[case + "_" + k] if case + "_" + k in z else None
    got = _run(g("points"), g("features"), g("labels"), float(g("dl")))
    ref_p, ref_f, ref_l = g("ref_points"), g("ref_features"), g("ref_labels")
    a, b = np_data.row_order(got["points"]), np_data.row_order(ref_p)
    assert got["points"].shape == ref_p.shape and np.array_equal(got["points"][a], ref_p[b])
    if ref_f is not None:
        assert np.array_equal(got["features"][a], ref_f[b])
    if ref_l is not None:
        tied = np_data.grid_subsampling(g("points"), g("features"), g("labels"), float(g("dl")))["tied"]
        free = ~tied[a]          # both are in ascending-key order before the row sort
        assert np.array_equal(got["labels"][a][free], ref_l[b][free])


def test_grid_subsampling_rejects_bad_input():
    from geot_amd.openpoints.dataset import grid_subsampling
    dev = torch.device("cuda:0")
    with pytest.raises(RuntimeError):
        grid_subsampling(torch.zeros((5, 2), device=dev))
    with pytest.raises(RuntimeError):
        grid_subsampling(torch.zeros((5, 3), device=dev), sampleDl=0.0)
    with pytest.raises(RuntimeError):
        grid_subsampling(torch.zeros((0, 3), device=dev))


@pytest.mark.parametrize("n,m", [(120000, 24000), (9000, 24000), (1, 5)])
def test_prepare_sample_matches_numpy_pipeline(n, m):
    from geot_amd.openpoints.dataset import pc_norm, prepare_sample
    rng = np.random.default_rng(n)
    pc = (rng.standard_normal((n, 3)) * np.array([30, 20, 8]) + np.array([5, -40, 12])).astype(np.float32)
    labels = rng.integers(0, 17, n).astype(np.int32)
    sel = rng.choice(n, m, replace=n < m)               # tooth_dataset.py:133-135
    q64, c64, m64 = np_data.pc_norm_f64(pc)
    dev = torch.device("cuda:0")
    out = prepare_sample(torch.from_numpy(pc).to(dev), torch.from_numpy(labels).to(dev), torch.from_numpy(sel).to(dev))
    scale = float(out["scale"])
    if n > 1:
        assert abs(scale - m64) <= 1e-5 * m64
        np.testing.assert_allclose(out["center"].cpu().numpy(), c64, rtol=0, atol=1e-5 * m64)
        np.testing.assert_allclose(out["pos"].cpu().numpy(), q64[sel], rtol=0, atol=1e-5)
    assert out["y"].dtype == torch.int64 and np.array_equal(out["y"].cpu().numpy(), labels[sel])
    assert np.array_equal(out["class_weights"].cpu().numpy(), np_data.class_weights(labels[sel], 17))
    if n > 1:
        full, c, s = pc_norm(torch.from_numpy(pc).to(dev))
        np.testing.assert_allclose(full.cpu().numpy(), q64, rtol=0, atol=1e-5)
        assert float(s) == scale and torch.equal(c, out["center"])
        # same fp32 expression as numpy once the centroid is fixed: feed numpy OUR centroid -> identical bits
        cen = c.cpu().numpy()
        qq = pc - cen
        mm = np.max(np.sqrt(np.sum(qq ** 2, axis=1)))
        assert mm == np.float32(scale) and np.array_equal(full.cpu().numpy(), qq / mm)


def test_prepare_sample_flags_bad_index():
    from geot_amd.openpoints.dataset import prepare_sample
    dev = torch.device("cuda:0")
    pc = torch.randn(100, 3, device=dev)
    lab = torch.zeros(100, dtype=torch.int32, device=dev)
    with pytest.raises(IndexError):
        prepare_sample(pc, lab, torch.tensor([0, 5, 100], device=dev))


def test_reference_extension_name_resolves_to_the_gpu_op():
    """`import openpoints.cpp.subsampling.grid_subsampling as cpp_subsampling; cpp_subsampling.compute(...)`
    (openpoints/dataset/grid_sample.py:1-23) after aliases.install()."""
    import geot_amd.aliases
    geot_amd.aliases.install()
    import openpoints.cpp.subsampling.grid_subsampling as cpp_subsampling
    p, f, lab = _cloud(21, 2500, 2, 1)
    pts, feats, labs = cpp_subsampling.compute(p, features=f, classes=lab[:, 0], sampleDl=0.25, verbose=0)
    want = np_data.grid_subsampling(p, f, lab, 0.25)
    assert np.array_equal(pts, want["points"]) and np.array_equal(feats, want["features"])
    assert np.array_equal(labs, want["labels"])
    with pytest.raises(RuntimeError, match="Valid method names"):
        cpp_subsampling.compute(p, method="nearest")
