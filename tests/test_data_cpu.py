"""Dataloader-side ops, CPU part: the numpy restatement of grid_subsampling against the
fixture generated from the reference's own compiled code, and against that code itself when oracle/_ref was
built in this checkout; the numpy pc_norm / class-weight restatement against fp64."""
import os

import numpy as np
import pytest

from oracle import np_data

GOLD = os.path.join(os.path.dirname(__file__), "golden", "grid_subsampling_ref.npz")


def compare_with_reference_rows(got, ref_points, ref_features, ref_labels):
    """`got` (ascending voxel key) vs the reference's rows in hash-map order: same set of rows, bit for bit;
    labels wherever the vote is not tied, and one of the tied labels elsewhere."""
    assert got["points"].shape == ref_points.shape
    a, b = np_data.row_order(got["points"]), np_data.row_order(ref_points)
    assert np.array_equal(got["points"][a], ref_points[b])
    if ref_features is not None:
        assert np.array_equal(got["features"][a], ref_features[b])
    if ref_labels is not None:
        free = ~got["tied"][a]
        assert np.array_equal(got["labels"][a][free], ref_labels[b][free])


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_grid_subsampling_restatement_matches_reference_fixture(case):
    z = np.load(GOLD)
    g = lambda k: z[case + "_" + k] if case + "_" + k in z else None
    got = np_data.grid_subsampling(g("points"), g("features"), g("labels"), float(g("dl")))
    compare_with_reference_rows(got, g("ref_points"), g("ref_features"), g("ref_labels"))
    assert np.all(np.diff(got["keys"].astype(np.int64)) > 0) and got["count"].sum() == len(g("points"))


@pytest.mark.skipif(not np_data.have_reference(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("seed,n,dl,fdim,ldim", [(0, 5000, 0.1, 0, 0), (1, 20000, 0.05, 4, 1), (2, 3000, 0.3, 2, 2),
                                                   (3, 1, 0.1, 1, 1), (4, 60, 10.0, 1, 1), (5, 40000, 0.013, 1, 1)])
def test_grid_subsampling_restatement_matches_compiled_reference(seed, n, dl, fdim, ldim):
    rng = np.random.default_rng(seed)
    p = (rng.standard_normal((n, 3)) * np.array([1, 0.7, 0.4])).astype(np.float32)
    if n > 100:
        p[rng.integers(0, n, n // 50)] = p[rng.integers(0, n, n // 50)]
    f = rng.standard_normal((n, fdim)).astype(np.float32) if fdim else None
    lab = rng.integers(-2, 6, (n, ldim)).astype(np.int32) if ldim else None
    got = np_data.grid_subsampling(p, f, lab, dl)
    ref = np_data.grid_subsampling_reference(p, f, lab, dl)
    compare_with_reference_rows(got, ref["points"], ref["features"], ref["labels"])


def test_pc_norm_numpy_within_tolerance_of_fp64():
    rng = np.random.default_rng(7)
    pc = (rng.standard_normal((120000, 3)) * np.array([30, 20, 8]) + np.array([5, -40, 12])).astype(np.float32)
    q32, c32, m32 = np_data.pc_norm_numpy(pc)
    q64, c64, m64 = np_data.pc_norm_f64(pc)
    # the published tolerance for float results on this path is 1e-5 relative (BASELINE.json north_star),
    # taken on the scale of the data (the normalised cloud has max norm 1)
    assert abs(m32 - m64) <= 1e-5 * m64 and np.abs(q32 - q64).max() <= 1e-5
    assert np.isclose(np.sqrt((q32.astype(np.float64) ** 2).sum(1)).max(), 1.0, atol=1e-6)


def test_class_weights():
    lab = np.array([0, 0, 3, 16, 16, 16, 5, 0])
    w = np_data.class_weights(lab, 17)
    assert w.dtype == np.float32 and w.shape == (17,) and np.isclose(w.sum(), 1.0)
    assert w[0] == np.float32(3) / np.float32(8) and w[16] == np.float32(3) / np.float32(8) and w[1] == 0
