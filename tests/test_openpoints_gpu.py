"""GPU suite: the openpoints-side wrappers (models/layers, cpp/pointops/functions, transformer hot-path
callers) against the oracle."""
import numpy as np
import pytest
import torch

from geot_amd.synth import make_batch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t.to(dtype) if dtype is not None else t).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def test_layers_subsample_group_upsampling(oracle):
    from geot_amd.openpoints.models import layers as L
    from geot_amd.openpoints.models.layers.group import QueryAndGroup, KNNGroup, create_grouper
    xyz_np, _ = make_batch(2, 5000, start_index=2, dup_frac=0.01)
    xyz = dev(xyz_np)
    feats = torch.randn(2, 7, 5000, device=DEV, requires_grad=True)
    idx = L.furthest_point_sample(xyz, 600)
    assert np.array_equal(host(idx), oracle.fps_dense(xyz_np, 600, 1024, False))   # pointnet2_batch flavour
    g = L.gather_points(feats, idx)
    assert torch.equal(g, torch.gather(feats, 2, idx.long().unsqueeze(1).expand(-1, 7, -1)))
    new_xyz = L.fps(xyz, 600)
    assert torch.equal(new_xyz, torch.gather(xyz, 1, idx.long().unsqueeze(-1).expand(-1, -1, 3)))
    gx, gf = QueryAndGroup(0.2, 16, normalize_dp=True)(new_xyz, xyz, feats)
    bq = oracle.ball_query(host(new_xyz), xyz_np, 0.2, 16)
    want_x = (oracle.group_points(np.ascontiguousarray(xyz_np.transpose(0, 2, 1)), bq) -
              host(new_xyz).transpose(0, 2, 1)[..., None]) / 0.2
    np.testing.assert_allclose(host(gx), want_x, rtol=1e-5, atol=1e-6)
    assert np.array_equal(host(gf), oracle.group_points(host(feats), bq))
    assert torch.equal(gf, L.torch_grouping_operation(feats, dev(bq)))
    gf.sum().backward()
    cnt = np.zeros((2, 5000), np.float32)
    for b in range(2):
        np.add.at(cnt[b], bq[b].reshape(-1), 1.0)
    np.testing.assert_allclose(host(feats.grad), np.repeat(cnt[:, None], 7, 1), rtol=1e-5)
    kx, kf = KNNGroup(8)(new_xyz, xyz, feats.detach())
    ki, _ = oracle.knn_sorted(host(new_xyz), xyz_np, 8)
    assert np.array_equal(host(kf), oracle.group_points(host(feats), ki))
    assert isinstance(create_grouper({"NAME": "ballquery", "radius": 0.1, "nsample": 8}), QueryAndGroup)
    # three_interpolation = three_nn + inverse-distance weights + three_interpolate
    f2 = torch.randn(2, 11, 600, device=DEV, requires_grad=True)
    up = L.three_interpolation(xyz, new_xyz, f2)
    d2, i3 = oracle.three_nn(xyz_np, host(new_xyz))
    r = 1.0 / (np.sqrt(d2) + 1e-8)
    w = (r / r.sum(2, keepdims=True)).astype(np.float32)
    np.testing.assert_allclose(host(up), oracle.three_interpolate(host(f2), i3, w), rtol=1e-4, atol=1e-5)
    up.sum().backward()
    np.testing.assert_allclose(host(f2.grad), oracle.three_interpolate_grad(np.ones((2, 11, 5000), np.float32), i3, w, 600),
                               rtol=1e-3, atol=1e-3)


def test_layers_knn_matches_reference_fixture(golden, oracle):
    from geot_amd.openpoints.models.layers.knn import knn_point, KNN, DilatedKNN
    g = golden("knn_point_ref.npz")
    x = dev(g["xyz"])
    dist, idx = knn_point(int(g["k"]), x, x)
    assert idx.dtype == torch.int64 and np.array_equal(host(idx), g["idx"])      # the reference's own outputs
    assert np.abs(host(dist) - g["dist"]).max() < 2e-3
    d33, i33 = KNN(33)(x[:, :256].contiguous(), x)
    assert (host(i33) == g["idx_sub33"]).mean() > 0.999
    dil = DilatedKNN(k=4, dilation=2)(x)
    wi, _ = oracle.knn_sorted(g["xyz"], g["xyz"], 8)
    assert np.array_equal(host(dil), wi[:, :, ::2])


def test_openpoints_pointops_functions(oracle):
    from geot_amd.openpoints.cpp.pointops.functions import pointops as P
    sizes, ms = [900, 1300], [200, 300]
    clouds = np.concatenate([make_batch(1, n, start_index=60 + i, dup_frac=0.02)[0][0] for i, n in enumerate(sizes)])
    off, noff = np.cumsum(sizes), np.cumsum(ms)
    xyz, o, no = dev(clouds), dev(off, torch.int32), dev(noff, torch.int32)
    idx = P.furthestsampling(xyz, o, no)
    assert np.array_equal(host(idx), oracle.fps_offset(clouds, off, noff))
    new_xyz = xyz[idx.long()].contiguous()
    kidx, kd = P.knnquery(6, xyz, new_xyz, o, no)
    wi, wd = oracle.knnquery_heap(6, clouds, host(new_xyz), off, noff)
    assert np.array_equal(host(kidx), wi) and np.allclose(host(kd), np.sqrt(wd), rtol=1e-6)
    bidx = P.ballquery(0.15, 12, xyz, new_xyz, o, no)
    assert np.array_equal(host(bidx), oracle.ballquery_offset(0.15, 12, clouds, host(new_xyz), off, noff))
    feat = torch.randn(clouds.shape[0], 9, device=DEV, requires_grad=True)
    grp = P.grouping(feat, kidx)
    assert np.array_equal(host(grp), oracle.grouping_cl(host(feat), wi))
    go = torch.randn_like(grp)
    grp.backward(go)
    np.testing.assert_allclose(host(feat.grad), oracle.grouping_cl_grad(host(go), wi, clouds.shape[0]), rtol=1e-4, atol=1e-5)
    gx, gfeat = P.querygroup(6, xyz, new_xyz, feat.detach(), o, no)
    assert np.array_equal(host(gfeat), oracle.grouping_cl(host(feat), wi))
    np.testing.assert_allclose(host(gx), clouds[wi] - host(new_xyz)[:, None, :], rtol=1e-6, atol=1e-7)
    cat = P.queryandgroup(6, xyz, new_xyz, feat.detach(), None, o, no)
    assert cat.shape == (500, 6, 12)
    # subtraction / aggregation on a self-graph
    n = clouds.shape[0]
    sidx, _ = P.knnquery(5, xyz, xyz, o, o)
    a = torch.randn(n, 8, device=DEV, requires_grad=True)
    b = torch.randn(n, 8, device=DEV, requires_grad=True)
    sub = P.subtraction(a, b, sidx)
    assert np.array_equal(host(sub), oracle.subtraction_cl(host(a), host(b), host(sidx)))
    go = torch.randn_like(sub)
    sub.backward(go)
    w1, w2 = oracle.subtraction_cl_grad(host(sidx), host(go))
    np.testing.assert_allclose(host(a.grad), w1, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(host(b.grad), w2, rtol=1e-4, atol=1e-5)
    pos = torch.randn(n, 5, 8, device=DEV, requires_grad=True)
    wgt = torch.rand(n, 5, 4, device=DEV, requires_grad=True)
    x = torch.randn(n, 8, device=DEV, requires_grad=True)
    agg = P.aggregation(x, pos, wgt, sidx)
    np.testing.assert_allclose(host(agg), oracle.aggregation_cl(host(x), host(pos), host(wgt), host(sidx)), rtol=1e-5, atol=1e-5)
    go = torch.randn_like(agg)
    agg.backward(go)
    gi, gp, gw = oracle.aggregation_cl_grad(host(x), host(pos), host(wgt), host(sidx), host(go))
    np.testing.assert_allclose(host(x.grad), gi, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(pos.grad), gp, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(wgt.grad), gw, rtol=1e-4, atol=1e-4)
    # interpolation: fused op == torch gather form
    f = torch.randn(500, 10, device=DEV, requires_grad=True)
    i2 = P.interpolation2(new_xyz, xyz, f, no, o)
    np.testing.assert_allclose(host(i2), host(P.interpolation(new_xyz, xyz, f.detach(), no, o)), rtol=1e-5, atol=1e-6)
    i2.sum().backward()
    assert torch.isfinite(f.grad).all()


def test_transformer_group_and_graph_feature(oracle):
    from geot_amd.openpoints.models.backbone.transformer_ops import Group, fps_downsample, get_graph_feature
    from geot_amd.knn_cuda import KNN
    xyz_np, _ = make_batch(2, 24000, start_index=1, dup_frac=0.01)
    xyz = dev(xyz_np)
    neigh, center, idx = Group(num_group=512, group_size=32)(xyz)
    fi = oracle.fps_dense(xyz_np, 512, 512, True)                       # pointnet2._ext FPS (origin-skip)
    wc = np.take_along_axis(xyz_np, fi[..., None].astype(np.int64).repeat(3, -1), 1)
    ki, _ = oracle.knn_sorted(wc, xyz_np, 32)
    flat = (ki.astype(np.int64) + np.arange(2)[:, None, None] * 24000).reshape(-1)
    assert np.array_equal(host(center), wc) and np.array_equal(host(idx), flat)
    np.testing.assert_allclose(host(neigh), xyz_np.reshape(-1, 3)[flat].reshape(2, 512, 32, 3) - wc[:, :, None, :], rtol=0, atol=0)
    # DGCNN propagation glue at the dgcnn_pro_2 shapes (512 -> 4096, k = 4)
    coor_q = dev(np.ascontiguousarray(xyz_np[:, :4096].transpose(0, 2, 1)))
    coor_k = dev(np.ascontiguousarray(wc.transpose(0, 2, 1)))
    x_k = torch.randn(2, 6, 512, device=DEV)
    x_q = torch.randn(2, 6, 4096, device=DEV)
    feat = get_graph_feature(KNN(k=4, transpose_mode=False), coor_q, x_q, coor_k, x_k)
    assert feat.shape == (2, 12, 4096, 4)
    gi, _ = oracle.knn_sorted(xyz_np[:, :4096], wc, 4)                  # (B, Nq, k)
    xk = host(x_k)
    want = np.stack([xk[b][:, gi[b]] for b in range(2)])                # (B, C, Nq, k)
    np.testing.assert_allclose(host(feat[:, :6]), want - host(x_q)[..., None], rtol=1e-6, atol=1e-6)
    # fused kernel == the reference's own op chain, forward bit for bit and backward to rounding
    from geot_amd.openpoints.models.backbone.transformer_ops import get_graph_feature_unfused
    knn4 = KNN(k=4, transpose_mode=False)
    xq1, xk1 = x_q.clone().requires_grad_(True), x_k.clone().requires_grad_(True)
    xq2, xk2 = x_q.clone().requires_grad_(True), x_k.clone().requires_grad_(True)
    f1 = get_graph_feature(knn4, coor_q, xq1, coor_k, xk1)
    f2 = get_graph_feature_unfused(knn4, coor_q, xq2, coor_k, xk2)
    assert torch.equal(f1, f2)
    g = torch.randn_like(f1)
    f1.backward(g); f2.backward(g)
    np.testing.assert_allclose(host(xq1.grad), host(xq2.grad), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(host(xk1.grad), host(xk2.grad), rtol=1e-5, atol=1e-4)   # ~32 atomics per row
    for c_, nq_, nk_, k_ in ((1, 1, 1, 1), (19, 333, 77, 5), (9, 70, 300, 20)):          # ragged shapes
        a = torch.randn(2, c_, nq_, device=DEV, requires_grad=True)
        bk = torch.randn(2, c_, nk_, device=DEV, requires_grad=True)
        ii = torch.randint(0, nk_, (2, nq_, k_), device=DEV, dtype=torch.int32)
        from geot_amd.openpoints.models.backbone.transformer_ops import graph_feature
        o = graph_feature(a, bk, ii)
        nb = torch.gather(bk.unsqueeze(2).expand(-1, -1, nq_, -1), 3, ii.long().unsqueeze(1).expand(-1, c_, -1, -1))
        ref = torch.cat((nb - a.unsqueeze(-1), a.unsqueeze(-1).expand(-1, -1, -1, k_)), 1)
        assert torch.equal(o, ref)
        go = torch.randn_like(o)
        ga, gb = torch.autograd.grad(o, (a, bk), go)
        ra, rb = torch.autograd.grad(ref, (a, bk), go)
        np.testing.assert_allclose(host(ga), host(ra), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(host(gb), host(rb), rtol=1e-5, atol=1e-4)
    nc, nx = fps_downsample(coor_q, x_q, 256)
    sel = oracle.fps_dense(xyz_np[:, :4096], 256, 512, True)
    assert np.array_equal(host(nc), np.take_along_axis(host(coor_q), sel[:, None, :].astype(np.int64).repeat(3, 1), 2))


def test_alias_installer_resolves_reference_import_names():
    import importlib
    import geot_amd.aliases as aliases
    names = aliases.install()
    for n in names + ["pointnet2_ops.pointnet2_utils", "pytorch_utils", "openpoints.cpp.pointnet2_batch"]:
        importlib.import_module(n)
    import pointnet2._ext as ext
    import pointops_cuda
    from knn_cuda import KNN
    from openpoints.cpp.pointnet2_batch import pointnet2_cuda
    x = torch.rand(1, 300, 3, device=DEV)
    assert ext.furthest_point_sampling(x, 16).shape == (1, 16)
    assert hasattr(pointops_cuda, "knnquery_cuda") and hasattr(pointnet2_cuda, "three_nn_wrapper")
    assert KNN(3, transpose_mode=True)(x, x[:, :5].contiguous())[1].shape == (1, 5, 3)


def test_composite_workloads_run_and_are_finite():
    """The bench / timing workloads (geot_amd/workloads.py) at a reduced size."""
    from geot_amd import workloads as wl
    xyz_np, _ = make_batch(2, 9000, start_index=5)
    xyz = dev(xyz_np)
    hot = wl.BackboneHotPath().to(DEV)
    loss = wl.backbone_hotpath_step(hot, xyz, torch.randn(2, 384, 512, device=DEV))
    assert torch.isfinite(loss)
    nt = wl.NtmHotPath().to(DEV)
    l2 = wl.ntm_step(nt, xyz, torch.randn(2, 17, 9000, device=DEV), torch.randn(2, 17, 9000, device=DEV))
    assert torch.isfinite(l2) and torch.isfinite(nt.ema_t).all()


def test_openpoints_sa_msg_and_fp_modules(oracle):
    """openpoints PointNetSAModuleMSG / ConvPool / PointNetFPModule mirrors: composed path == fused eval
    path, sampling + grouping indices == oracle, FP module == numpy restatement of its front end."""
    from geot_amd.openpoints.models.backbone.pointnetv2 import PointNetSAModuleMSG, PointNetFPModule
    torch.manual_seed(0)
    B, N, C = 2, 2048, 6
    xyz_np, _ = make_batch(B, N, start_index=7)
    xyz = dev(xyz_np)
    feats = torch.randn(B, C, N, device=DEV)
    sa = PointNetSAModuleMSG(stride=4, radii=[0.15, 0.3], nsamples=[16, 32], channel_list=[[C, 32, 64], [C, 32, 64]],
                             aggr_args={'feature_type': 'dp_fj', 'reduction': 'max'},
                             group_args={'NAME': 'ballquery', 'normalize_dp': True}, conv_args={},
                             norm_args={'norm': 'bn'}, act_args={'act': 'relu'}).to(DEV)
    for m in sa.modules():                                    # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2); m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    sa.eval()
    with torch.no_grad():
        q_f, out_f = sa(xyz, feats)                           # fused kernel
    for la in sa.local_aggregations:
        la.SA_CONFIG_operator.fused_eval = False
    with torch.no_grad():
        q_c, out_c = sa(xyz, feats)                           # composed ops
    want_idx = oracle.fps_dense(xyz_np, N // 4, 1024, False)  # K1' rule (no origin skip, block cap 1024)
    want_q = np.take_along_axis(xyz_np, want_idx[..., None].astype(np.int64).repeat(3, -1), 1)
    assert np.array_equal(host(q_f), want_q) and np.array_equal(host(q_c), want_q)
    assert out_f.shape == (B, 128, N // 4)
    np.testing.assert_allclose(host(out_f), host(out_c), rtol=2e-4, atol=2e-4)
    # training-mode path gives gradients to the features
    sa.train()
    f2 = feats.clone().requires_grad_(True)
    sa(xyz, f2)[1].sum().backward()
    assert torch.isfinite(f2.grad).all() and f2.grad.abs().sum() > 0
    # FP module
    fp = PointNetFPModule([64 + C, 32, 16]).to(DEV).eval()
    known, kf = q_f, torch.randn(B, 64, N // 4, device=DEV)
    with torch.no_grad():
        got = fp(xyz, known, feats, kf)
    d2, idx = oracle.three_nn(xyz_np, host(known))
    r = 1.0 / (np.sqrt(d2.astype(np.float32)) + np.float32(1e-8))
    w = (r / r.sum(2, keepdims=True)).astype(np.float32)
    interp = oracle.three_interpolate(host(kf), idx, w)
    with torch.no_grad():
        want = fp.convs(torch.cat([feats, dev(interp)], 1))
    np.testing.assert_allclose(host(got), host(want), rtol=1e-4, atol=1e-5)


def test_get_pred_whole_matches_oracle(oracle):
    from geot_amd.validation import get_pred_whole
    rng = np.random.default_rng(11)
    B, N, C = 2, 3000, 17
    pts_np, _ = make_batch(B, N, start_index=30)
    logits = rng.normal(size=(B, C, N)).astype(np.float32) * 3
    center = [rng.normal(size=(1, 3)).astype(np.float32) for _ in range(B)]
    scale = [np.float32(rng.uniform(20, 40)) for _ in range(B)]
    whole = [(rng.normal(size=(m, 3)).astype(np.float32) * 0.4 * scale[i] + center[i]) for i, m in enumerate((20011, 12345))]
    preds = get_pred_whole(dev(logits), dev(pts_np), [torch.from_numpy(w) for w in whole],
                           [torch.from_numpy(c) for c in center], [torch.tensor(s) for s in scale])
    e = np.exp(logits - logits.max(1, keepdims=True))
    sm = (e / e.sum(1, keepdims=True)).astype(np.float32)
    for i in range(B):
        p = (pts_np[i] * scale[i] + center[i]).astype(np.float32)
        d2, idx = oracle.three_nn(whole[i][None], p[None])
        r = 1.0 / (np.sqrt(d2) + np.float32(1e-8))
        w = r / r.sum(2, keepdims=True)
        lw = oracle.three_interpolate(sm[i][None], idx, w.astype(np.float32))[0]      # (C, M)
        top2 = np.sort(lw, 0)[-2:]
        clear = (top2[1] - top2[0]) > 1e-5                     # ignore numerically tied vertices
        got = preds[i][0].cpu().numpy()
        assert got.shape == (whole[i].shape[0],)
        assert np.array_equal(got[clear], lw.argmax(0)[clear]) and clear.mean() > 0.99


def test_group_knn_feature_space_and_dilated():
    """group.KNN on non-3-D inputs (the reference's cdist + topk works for any C) and DilatedKNN."""
    from geot_amd.openpoints.models.layers.group import KNN, DilatedKNN
    torch.manual_seed(5)
    sup, qry = torch.randn(2, 500, 7, device="cuda:0"), torch.randn(2, 90, 7, device="cuda:0")
    dist, idx = KNN(6)(sup, qry)
    assert dist.shape == (2, 6, 90) and idx.shape == (2, 90, 6) and idx.dtype == torch.int32
    ref = torch.cdist(qry, sup).topk(6, dim=-1, largest=False, sorted=True)
    assert torch.equal(idx.long(), ref.indices)
    assert torch.allclose(dist.transpose(1, 2), ref.values, rtol=1e-4, atol=1e-5)
    pts = torch.rand(2, 400, 3, device="cuda:0")
    dil = DilatedKNN(k=4, dilation=3)(pts)
    _, full = KNN(12)(pts, pts)
    assert torch.equal(dil, full[:, :, ::3])


@pytest.mark.gpu
def test_pseudo_mask_refinement_matches_reference_loops():
    """utils/pseudo_mask.py:5-53, 55-90, 174-196 transcribed with torch index_select loops over the same pointops.knn
    neighbours (which are checked against the oracle elsewhere) -- against the mirror's single grouping launch."""
    from geot_amd.utils import pseudo_mask as pm
    from geot_amd.pointops.functions import pointops
    xyz, _ = make_batch(2, 3000, start_index=17)
    pos = torch.from_numpy(xyz).cuda()
    pred = torch.softmax(torch.randn(2, 17, 3000, device="cuda"), 1)
    n = 4
    nbrs, dist = pm.get_neigbor_tensors(pred, n, pos)
    idx, d = pointops.knn(pos, pos, n + 1)
    flat = (idx[:, :, 1:] + torch.arange(2, device="cuda").view(2, 1, 1) * 3000)
    X = pred.transpose(0, 1).contiguous().view(17, -1)
    for ii in range(n):
        want = torch.index_select(X, 1, flat[:, :, ii].reshape(-1)).view(17, 2, 3000).transpose(0, 1)
        assert torch.equal(nbrs[ii], want)
    assert torch.equal(dist, d[:, :, 1:])
    beta = torch.exp(torch.tensor(-0.5)).cuda()
    k_nb, _ = torch.topk(torch.stack(nbrs), k=1, dim=0)
    ref = pred + beta * k_nb[0] - (pred * k_nb[0]) * beta
    mask = pm.pseudo_label_refine(pred, 0.5, pos, n, 1)
    assert torch.equal(mask, ref.max(1)[0].ge(0.5))
    m2, margin = pm.pseudo_label_refine_margin(pred, 0.1, pos, n, 1)
    top2 = torch.topk(ref, 2, dim=1)[0]
    assert torch.allclose(margin, top2[:, 0] - top2[:, 1]) and torch.equal(m2, margin.ge(0.1))
    cnt = pm.neigh_acc_count(17)
    lab = pred.argmax(1)
    cnt.update(lab, pos)
    nn1 = idx[0, :, 1]
    acc = (lab[0] == lab[0][nn1])
    want = np.stack([[int((lab[0] == kk).sum()), int((acc & (lab[0] == kk)).sum())] for kk in range(17)])
    assert np.array_equal(cnt.acc_array, want)
