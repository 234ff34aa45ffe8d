"""CPU suite, part 1: the oracle itself.

The C restatement (literal simulation of the reference kernels' loops) is checked
against the independent numpy restatement (closed-form tie rules), against the
committed golden vectors, against the reference's own torch-only ``knn_point``
outputs (fixture generated from /root/reference by tests/golden/make_golden.py) and
against the reference's only test vector (pointnet2/pointnet2_test.py:15-27).
"""
import numpy as np
import pytest

from geot_amd.synth import make_batch, make_cloud
from oracle import np_ref


# ---- BASELINE configs[0]: the CPU index gate --------------------------------
@pytest.mark.parametrize("tag", ["plain", "dup1pct"])
def test_config1_golden(oracle, golden, tag):
    g = golden("config1_%s.npz" % tag)
    xyz = g["xyz"]
    # the generator is part of the contract: the fixture must be reproducible from the seed
    regen, _ = make_batch(1, 4096, dup_frac=0.01 if tag == "dup1pct" else 0.0)
    assert np.array_equal(regen, xyz)
    k1 = oracle.fps_dense(xyz, 1024, 512, True)
    k1p = oracle.fps_dense(xyz, 1024, 1024, False)
    k2 = oracle.fps_offset(xyz.reshape(-1, 3), np.array([4096]), np.array([1024]))
    assert np.array_equal(k1, g["fps_k1"])
    assert np.array_equal(k1p, g["fps_k1p"])
    assert np.array_equal(k2, g["fps_k2"])
    assert np.array_equal(k1, np_ref.fps_dense(xyz, 1024, 512, True))
    assert np.array_equal(k1p, np_ref.fps_dense(xyz, 1024, 1024, False))
    centres = np.take_along_axis(xyz, k1[..., None].astype(np.int64).repeat(3, -1), 1)
    bq = oracle.ball_query(centres, xyz, float(g["radius"]), int(g["nsample"]))
    assert np.array_equal(bq, g["ball_query"])
    assert np.array_equal(bq, np_ref.ball_query(centres, xyz, float(g["radius"]), int(g["nsample"])))
    # the origin-skip quirk is live on this cloud: K1 and K1' must differ somewhere
    mag = (xyz[0] ** 2).sum(1)
    assert (mag <= 1e-3).sum() >= 1
    assert not np.isin(np.nonzero(mag <= 1e-3)[0], k1[0]).any()


def test_block_size_rule(oracle):
    for n in list(range(1, 70)) + [127, 128, 129, 511, 512, 513, 1023, 1024, 1025, 4096, 16000, 24000, 1 << 19]:
        for cap in (512, 1024):
            expect = min(1 << (n.bit_length() - 1), cap)
            assert oracle.block_size(n, cap) == expect == np_ref.block_size(n, cap)


@pytest.mark.parametrize("n,m", [(1, 1), (2, 2), (3, 3), (5, 4), (63, 17), (64, 64), (65, 33), (100, 100),
                                 (513, 200), (777, 300), (1025, 128), (1500, 256)])
@pytest.mark.parametrize("cap,skip", [(512, True), (1024, False)])
def test_fps_small_and_ragged(oracle, n, m, cap, skip):
    xyz, _ = make_batch(2, n, start_index=n, dup_frac=0.05 if n > 20 else 0.0, origin_pts=2)
    a = oracle.fps_dense(xyz, m, cap, skip)
    b = np_ref.fps_dense(xyz, m, cap, skip)
    assert np.array_equal(a, b)
    assert (a[:, 0] == 0).all()


def test_fps_more_samples_than_points(oracle):
    # m > n: once every point is taken all distances are 0 and the tie key decides
    xyz, _ = make_batch(1, 40, origin_pts=0)
    a = oracle.fps_dense(xyz, 60, 1024, False)
    assert np.array_equal(a, np_ref.fps_dense(xyz, 60, 1024, False))
    assert len(set(a[0, :40].tolist())) == 40


def test_fps_all_points_skipped(oracle):
    xyz = (np.random.default_rng(0).standard_normal((1, 50, 3)) * 0.001).astype(np.float32)
    a = oracle.fps_dense(xyz, 10, 512, True)
    assert (a == 0).all()
    assert np.array_equal(a, np_ref.fps_dense(xyz, 10, 512, True))


def test_fps_ties_follow_bitreversed_thread_order(oracle):
    # 8 identical far points + the start point: every candidate ties, so the order is pure tie rule
    xyz = np.zeros((1, 16, 3), dtype=np.float32)
    xyz[0, 1:, 0] = 1.0
    a = oracle.fps_dense(xyz, 3, 1024, False)[0]
    # bs = 16 -> bit-reversed order of tids 1..15 under 4 bits: 8 (0001) is the smallest key
    assert a[1] == 8
    assert np.array_equal(a, np_ref.fps_dense(xyz, 3, 1024, False)[0])


def test_fps_offset_ragged_and_prefix(oracle):
    sizes = [700, 1300, 64, 2048]
    ms = [100, 333, 64, 512]
    clouds = [make_cloud(n, 50 + i, dup_frac=0.02)[0] for i, n in enumerate(sizes)]
    flat = np.concatenate(clouds)
    off, noff = np.cumsum(sizes), np.cumsum(ms)
    a = oracle.fps_offset(flat, off, noff)
    assert np.array_equal(a, np_ref.fps_offset(flat, off, noff))
    # global indices stay inside their segment, first pick = segment start
    s = 0
    for i, (n, m) in enumerate(zip(sizes, ms)):
        seg = a[noff[i] - m:noff[i]]
        assert seg[0] == s and (seg >= s).all() and (seg < s + n).all()
        s += n
    # prefix property (same n_max => same block size): fewer samples = a prefix
    half = oracle.fps_offset(flat, off, np.cumsum([m // 2 for m in ms]))
    o = 0
    for i, m in enumerate(ms):
        assert np.array_equal(half[o:o + m // 2], a[noff[i] - m:noff[i] - m + m // 2])
        o += m // 2
    # weighted variant
    w = np.random.default_rng(3).random(flat.shape[0]).astype(np.float32)
    w[::17] = 0.0  # exercises max(w, 1e-12)
    assert np.array_equal(oracle.fps_offset(flat, off, noff, w), np_ref.fps_offset(flat, off, noff, w))


def test_ball_query_edges(oracle):
    xyz, _ = make_batch(2, 600, start_index=7, dup_frac=0.05)
    q = xyz[:, ::7].copy()
    q[:, 0] = 5.0  # a query with no neighbour at all -> zeros
    for r, ns in [(0.05, 8), (0.2, 16), (0.5, 64), (3.0, 700)]:
        a = oracle.ball_query(q, xyz, r, ns)
        assert np.array_equal(a, np_ref.ball_query(q, xyz, r, ns))
        assert (a[:, 0] == 0).all()
    # offset-batched flavour = dense flavour + segment base (and zeros stay global zeros)
    off = np.array([600, 1200])
    noff = np.array([q.shape[1], 2 * q.shape[1]])
    b = oracle.ballquery_offset(0.2, 16, xyz.reshape(-1, 3), q.reshape(-1, 3), off, noff).reshape(2, -1, 16)
    d = oracle.ball_query(q, xyz, 0.2, 16)
    assert np.array_equal(b[0], d[0])
    hit = (b[1] != 0).any(1)
    assert np.array_equal(b[1][hit], d[1][hit] + 600)
    assert (b[1][~hit] == 0).all()


def test_three_nn_and_knn_sorted(oracle):
    xyz, _ = make_batch(2, 900, start_index=3, dup_frac=0.05)
    known = xyz[:, ::5].copy()
    d2a, ia = oracle.three_nn(xyz, known)
    d2b, ib = np_ref.three_nn(xyz, known)
    assert np.array_equal(ia, ib) and np.array_equal(d2a, d2b)
    for m in (1, 2):  # fewer than 3 known points -> (+inf, 0) padding
        d2, i = oracle.three_nn(xyz[:, :10], known[:, :m])
        d2n, i_n = np_ref.three_nn(xyz[:, :10], known[:, :m])
        assert np.array_equal(i, i_n) and np.array_equal(d2, d2n) and np.isinf(d2[..., m:]).all()
    for k in (1, 4, 33):
        ia, da = oracle.knn_sorted(known, xyz, k)
        ib, db = np_ref.knn_sorted(known, xyz, k)
        assert np.array_equal(ia, ib) and np.array_equal(da, db)


def test_knn_heap_matches_literal_transcription(oracle):
    xyz, _ = make_batch(2, 150, start_index=21, dup_frac=0.1)
    flat = xyz.reshape(-1, 3)
    off = np.array([150, 300])
    q = flat[::3].copy()
    noff = np.array([50, 100])
    for k in (1, 3, 5, 16):
        ia, da = oracle.knnquery_heap(k, flat, q, off, noff)
        ib, db = np_ref.knnquery_heap_literal(k, flat, q, off, noff)
        assert np.array_equal(ia, ib) and np.array_equal(da, db)
        # without exact ties the heap order equals the sorted order
    xyz2, _ = make_batch(1, 200, start_index=5, origin_pts=0)
    ia, da = oracle.knnquery_heap(8, xyz2[0], xyz2[0, :40], np.array([200]), np.array([40]))
    ib, db = oracle.knn_sorted(xyz2[:, :40], xyz2, 8)
    assert np.array_equal(da, db[0])
    assert np.array_equal(ia, ib[0])
    # fewer candidates than nsample: trailing (segment start, 1e10)
    ia, da = oracle.knnquery_heap(6, xyz2[0, :4], xyz2[0, :2], np.array([4]), np.array([2]))
    assert (da[:, 4:] == np.float32(1e10)).all() and (ia[:, 4:] == 0).all()


def test_knn_against_reference_knn_point(oracle, golden):
    """Fixture = outputs of the reference's own knn_point (cdist + topk)."""
    g = golden("knn_point_ref.npz")
    xyz = g["xyz"]
    idx, d2 = oracle.knn_sorted(xyz, xyz, int(g["k"]))
    assert np.array_equal(idx, g["idx"])
    # cdist uses the |a|^2+|b|^2-2ab expansion: ~5e-4 absolute error near zero distance
    assert np.abs(np.sqrt(d2) - g["dist"]).max() < 2e-3
    idx33, d233 = oracle.knn_sorted(xyz[:, :256], xyz, 33)
    same = (idx33 == g["idx_sub33"])
    assert same.mean() > 0.999  # near-ties may swap under cdist's expansion
    assert np.abs(np.sqrt(d233) - g["dist_sub33"]).max() < 2e-3


def test_gather_group_interpolate_definitions(oracle):
    rng = np.random.default_rng(5)
    feats = rng.standard_normal((2, 5, 40)).astype(np.float32)
    idx = rng.integers(0, 40, (2, 13)).astype(np.int32)
    assert np.array_equal(oracle.gather_points(feats, idx), np_ref.gather_points(feats, idx))
    gidx = rng.integers(0, 40, (2, 7, 4)).astype(np.int32)
    assert np.array_equal(oracle.group_points(feats, gidx), np_ref.group_points(feats, gidx))
    tidx = rng.integers(0, 40, (2, 11, 3)).astype(np.int32)
    w = rng.random((2, 11, 3)).astype(np.float32)
    assert np.allclose(oracle.three_interpolate(feats, tidx, w), np_ref.three_interpolate(feats, tidx, w), rtol=1e-6)
    # backward passes are the transposes of the forward maps: <A x, y> == <x, A^T y>
    go = rng.standard_normal((2, 5, 13)).astype(np.float32)
    lhs = (oracle.gather_points(feats, idx).astype(np.float64) * go).sum()
    rhs = (feats.astype(np.float64) * oracle.gather_points_grad(go, idx, 40)).sum()
    assert abs(lhs - rhs) < 1e-4 * max(1, abs(lhs))
    go = rng.standard_normal((2, 5, 7, 4)).astype(np.float32)
    lhs = (oracle.group_points(feats, gidx).astype(np.float64) * go).sum()
    rhs = (feats.astype(np.float64) * oracle.group_points_grad(go, gidx, 40)).sum()
    assert abs(lhs - rhs) < 1e-4 * max(1, abs(lhs))
    go = rng.standard_normal((2, 5, 11)).astype(np.float32)
    lhs = (oracle.three_interpolate(feats, tidx, w).astype(np.float64) * go).sum()
    rhs = (feats.astype(np.float64) * oracle.three_interpolate_grad(go, tidx, w, 40)).sum()
    assert abs(lhs - rhs) < 1e-4 * max(1, abs(lhs))


def test_reference_gradcheck_vector(oracle):
    """pointnet2/pointnet2_test.py:15-27: gradcheck of three_interpolate with
    idx=[[[0,1,2],[1,2,3]]], weight=[[[1,1,1],[2,2,2]]], feats (1,2,4), atol=rtol=1e-1."""
    idx = np.array([[[0, 1, 2], [1, 2, 3]]], dtype=np.int32)
    w = np.array([[[1, 1, 1], [2, 2, 2]]], dtype=np.float32)
    feats = np.random.default_rng(0).standard_normal((1, 2, 4)).astype(np.float32)
    out = oracle.three_interpolate(feats, idx, w)
    assert np.allclose(out[0, :, 0], feats[0, :, :3].sum(1), rtol=1e-6)
    assert np.allclose(out[0, :, 1], 2 * feats[0, :, 1:].sum(1), rtol=1e-6)
    go = np.ones((1, 2, 2), dtype=np.float32)
    analytic = oracle.three_interpolate_grad(go, idx, w, 4)
    eps = 1e-2
    numeric = np.zeros_like(feats)
    for c in range(2):
        for k in range(4):
            fp, fm = feats.copy(), feats.copy()
            fp[0, c, k] += eps
            fm[0, c, k] -= eps
            numeric[0, c, k] = (oracle.three_interpolate(fp, idx, w).sum() -
                                oracle.three_interpolate(fm, idx, w).sum()) / (2 * eps)
    assert np.allclose(analytic, numeric, atol=1e-1, rtol=1e-1)
    assert np.array_equal(analytic[0, 0], np.array([1, 3, 3, 2], dtype=np.float32))


def test_channels_last_ops(oracle):
    rng = np.random.default_rng(9)
    n, ns, c, w_c = 30, 5, 8, 4
    x = rng.standard_normal((n, c)).astype(np.float32)
    idx = rng.integers(0, n, (n, ns)).astype(np.int32)
    assert np.array_equal(oracle.grouping_cl(x, idx), x[idx])
    w = rng.random((n, ns)).astype(np.float32)
    assert np.allclose(oracle.interpolation_cl(x, idx, w), (x[idx] * w[..., None]).sum(1), rtol=1e-5, atol=1e-6)
    y = rng.standard_normal((n, c)).astype(np.float32)
    assert np.array_equal(oracle.subtraction_cl(x, y, idx), x[:, None, :] - y[idx])
    pos = rng.standard_normal((n, ns, c)).astype(np.float32)
    ww = rng.random((n, ns, w_c)).astype(np.float32)
    expect = ((x[idx] + pos) * np.tile(ww, (1, 1, c // w_c))).sum(1)
    assert np.allclose(oracle.aggregation_cl(x, pos, ww, idx), expect, rtol=1e-5, atol=1e-5)
    go = rng.standard_normal((n, c)).astype(np.float32)
    g_in, g_pos, g_w = oracle.aggregation_cl_grad(x, pos, ww, idx, go)
    lhs = (oracle.aggregation_cl(x, pos, ww, idx).astype(np.float64) * go).sum()
    # linear in (input+position) for fixed weight
    rhs = (x.astype(np.float64) * g_in).sum() + (pos.astype(np.float64) * g_pos).sum()
    assert abs(lhs - rhs) < 1e-3 * max(1, abs(lhs))
    g1, g2 = oracle.subtraction_cl_grad(idx, np.ones((n, ns, c), dtype=np.float32))
    assert np.allclose(g1, ns) and np.isclose(g2.sum(), -n * ns * c)


def test_nd_knn_restatement_reduces_to_the_3d_one():
    from oracle import np_ref
    rng = np.random.default_rng(12)
    ref = rng.random((2, 400, 3)).astype(np.float32)
    ref[:, 300:330] = ref[:, 10:40]
    qry = np.concatenate([ref[:, :50], rng.random((2, 70, 3)).astype(np.float32)], 1)
    i3, d3 = np_ref.knn_sorted(qry, ref, 9)
    i_n, d_n = np_ref.knn_sorted_nd(qry, ref, 9)
    assert np.array_equal(i3, i_n) and np.array_equal(d3, d_n)
