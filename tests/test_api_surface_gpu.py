"""API members of the reference that the other suites never executed: GroupAll (pointnet2/pointnet2_utils.py:376-422,
openpoints/models/layers/group.py:258-275) and the SetAbstraction modules built on it (npoint=None),
QueryAndGroup(sample_uniformly=True, ret_unique_cnt=True) (pointnet2_utils.py:333-342), pointops.fps_weight /
FurthestSamplingWeight (pointops/functions/pointops.py:34-44, 81-98) -- and the re-entrancy SURVEY 8(b) asks of the binding:
operators from a worker thread on its own stream while the main thread runs a backward."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def cloud(b, n, seed):
    from geot_amd.synth import make_batch
    return make_batch(b, n, start_index=seed)[0]


def test_group_all_both_flavours():
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from geot_amd.openpoints.models.layers import group as og
    b, n, c = 2, 300, 5
    xyz = dev(cloud(b, n, 3))
    feats = torch.randn(b, c, n, device=DEV)
    # pointnet2 flavour: (B, 3 + C, 1, N), xyz channels first, optionally grouped_xyz
    out, gx = pu.GroupAll(use_xyz=True, ret_grouped_xyz=True)(xyz, None, feats)
    assert out.shape == (b, 3 + c, 1, n) and gx.shape == (b, 3, 1, n)
    assert torch.equal(out[:, :3, 0], xyz.transpose(1, 2)) and torch.equal(out[:, 3:, 0], feats) and torch.equal(gx, out[:, :3])
    assert torch.equal(pu.GroupAll(use_xyz=False)(xyz, None, feats), feats.unsqueeze(2))
    assert torch.equal(pu.GroupAll()(xyz, None, None), xyz.transpose(1, 2).unsqueeze(2))
    # openpoints flavour: (grouped_xyz, grouped_features), arguments (new_xyz, xyz, features)
    gxyz, gf = og.GroupAll()(None, xyz, feats)
    assert torch.equal(gxyz, xyz.transpose(1, 2).unsqueeze(2)) and torch.equal(gf, feats.unsqueeze(2))
    assert og.GroupAll()(None, xyz, None)[1] is None


@pytest.mark.parametrize("cls", ["PointnetSAModule", "PointnetSAModuleVotes"])
def test_sa_module_without_sampling_pools_the_whole_cloud(cls):
    """npoint=None: one group holding every point (the global SA layer of the classification nets): the module equals
    SharedMLP + max over all points, forward and backward."""
    from geot_amd.pointnet2 import pointnet2_modules as pm
    b, n, c = 2, 257, 6
    torch.manual_seed(0)
    mod = getattr(pm, cls)(mlp=[c, 16, 32], npoint=None, radius=None, nsample=None, use_xyz=True).to(DEV).train()
    xyz = dev(cloud(b, n, 11))
    feats = torch.randn(b, c, n, device=DEV, requires_grad=True)
    out = mod(xyz, feats)
    new_xyz, new_feats = out[0], out[1]
    assert new_xyz is None and new_feats.shape == (b, 32, 1)
    mlp = mod.mlp_module if cls == "PointnetSAModuleVotes" else mod.mlps[0]
    want = mlp(torch.cat([xyz.transpose(1, 2), feats], 1).unsqueeze(2)).max(-1)[0]
    assert torch.allclose(new_feats, want, rtol=1e-5, atol=1e-6)
    new_feats.sum().backward()
    assert torch.isfinite(feats.grad).all() and float(feats.grad.abs().sum()) > 0


def test_query_and_group_sample_uniformly_matches_the_reference_loop():
    """pointnet2_utils.py:333-342 restated with numpy on the oracle's ball query: per region, the unique ids first, then
    draws WITH the same seeded torch.randint stream among them; unique counts returned."""
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from oracle import capi
    b, n, m, ns, r = 2, 800, 40, 16, 0.25
    xyz_np = cloud(b, n, 21)
    new_np = xyz_np[:, :m].copy()
    feats = torch.randn(b, 3, n, device=DEV)
    grouper = pu.QueryAndGroup(r, ns, use_xyz=True, ret_grouped_xyz=True, sample_uniformly=True, ret_unique_cnt=True)
    torch.manual_seed(77)
    new_features, grouped_xyz, unique_cnt = grouper(dev(xyz_np), dev(new_np), feats)
    idx = capi.ball_query(new_np, xyz_np, r, ns).astype(np.int64)
    torch.manual_seed(77)
    cnt = np.zeros((b, m))
    for ib in range(b):
        for ir in range(m):
            uniq = np.unique(idx[ib, ir])
            cnt[ib, ir] = len(uniq)
            pick = torch.randint(0, len(uniq), (ns - len(uniq),), dtype=torch.long).numpy()
            idx[ib, ir] = np.concatenate([uniq, uniq[pick]])
    assert np.array_equal(unique_cnt.numpy(), cnt) and cnt.min() >= 1 and cnt.max() <= ns and (cnt < ns).any()
    gx = np.take_along_axis(xyz_np.transpose(0, 2, 1)[:, :, None, :], idx[:, None].repeat(3, 1), axis=3) - new_np.transpose(0, 2, 1)[..., None]
    assert np.allclose(grouped_xyz.cpu().numpy(), gx, rtol=1e-6, atol=1e-7)
    gf = np.take_along_axis(feats.cpu().numpy()[:, :, None, :], idx[:, None].repeat(3, 1), axis=3)
    assert np.array_equal(new_features[:, 3:].cpu().numpy(), gf) and torch.equal(new_features[:, :3], grouped_xyz)


@pytest.mark.parametrize("b,n,k", [(2, 3000, 64), (3, 513, 200)])
def test_fps_weight_equals_the_weighted_oracle(b, n, k):
    """pointops.fps_weight: d = float(double(d) * max(double(w), 1e-12)) before the running minimum
    (pointops/src/sampling/sampling_cuda_kernel.cu:221-225); first pick = point 0 of every cloud."""
    from geot_amd.pointops.functions import pointops
    from oracle import capi
    rng = np.random.default_rng(b + n)
    xyz = cloud(b, n, 5)
    w = rng.random((b, n)).astype(np.float32)
    w[:, ::7] = 0.0                                         # weights below the 1e-12 floor
    got = pointops.fps_weight(dev(xyz), k, dev(w))
    off = (np.arange(1, b + 1) * n).astype(np.int32)
    noff = (np.arange(1, b + 1) * k).astype(np.int32)
    want_idx = capi.fps_offset(xyz.reshape(-1, 3), off, noff, weights=w.reshape(-1))
    assert got.shape == (b, k, 3)
    assert np.array_equal(got.cpu().numpy(), xyz.reshape(-1, 3)[want_idx].reshape(b, k, 3))
    # the autograd.Function itself (global indices, int32, non-differentiable)
    idx = pointops.furthestsampling_weight(dev(xyz.reshape(-1, 3)), dev(off), dev(noff), dev(w.reshape(-1)))
    assert idx.dtype == torch.int32 and not idx.requires_grad and np.array_equal(idx.cpu().numpy(), want_idx)
    # weights of one: plain FPS
    ones = pointops.fps_weight(dev(xyz), k, torch.ones(b, n, device=DEV))
    assert torch.equal(ones, pointops.fps(dev(xyz), k))


def test_binding_is_reentrant_from_a_second_thread_and_stream():
    """Forward ops from a worker thread on a stream of its own while the main thread's autograd engine runs the gradient
    kernels of the same binding: every result equals the single-threaded one (launchers keep no per-call global state)."""
    from geot_amd.pointnet2 import pointnet2_utils as pu
    b, n, m, c = 2, 6000, 1500, 48
    xyz = dev(cloud(b, n, 31))
    known = xyz[:, :m].contiguous()
    feats_k = torch.randn(b, c, m, device=DEV)
    feats_n = torch.randn(b, c, n, device=DEV)

    def forward_ops():
        inds = pu.furthest_point_sample(xyz, 500)
        new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
        bq = pu.ball_query(0.1, 16, xyz, new_xyz)
        grouped = pu.grouping_operation(feats_n, bq)
        d, i3 = pu.three_nn(xyz, known)
        return inds, bq, grouped, d, i3

    def backward_ops():
        f = feats_k.clone().requires_grad_(True)
        g = feats_n.clone().requires_grad_(True)
        d, i3 = pu.three_nn(xyz, known)
        w = 1.0 / (d + 1e-8)
        w = w / w.sum(2, keepdim=True)
        up = pu.three_interpolate(f, i3, w)
        bq = pu.ball_query(0.1, 16, xyz, known)
        grp = pu.grouping_operation(g, bq)
        (up.square().sum() + grp.square().sum()).backward()
        return f.grad, g.grad

    ref_fwd, ref_bwd = forward_ops(), backward_ops()
    torch.cuda.synchronize()
    results, errors = {}, []

    def worker():
        try:
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                for it in range(6):
                    results[it] = forward_ops()
                side.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=worker)
    th.start()
    main = [backward_ops() for _ in range(6)]
    th.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for it in range(6):
        for a, bb in zip(results[it], ref_fwd):
            assert torch.equal(a, bb)
        for a, bb in zip(main[it], ref_bwd):
            assert torch.equal(a, bb)          # the gradients have one writer per element (csrc/tile_scatter.hip): bit-equal


def _oracle_groups(p_np, x_np, m, radius, ns):
    """FPS (pointnet2_batch semantics: reduction width 1024, no origin skip) + ball query + gathers with the CPU oracle."""
    from oracle import capi
    idx = capi.fps_dense(p_np, m, 1024, False).astype(np.int64)
    centre = np.take_along_axis(p_np, idx[..., None].repeat(3, -1), 1)
    bq = capi.ball_query(centre, p_np, radius, ns).astype(np.int64)
    dp = np.take_along_axis(p_np.transpose(0, 2, 1)[:, :, None, :], bq[:, None].repeat(3, 1), 3) - centre.transpose(0, 2, 1)[..., None]
    fj = np.take_along_axis(x_np[:, :, None, :], bq[:, None].repeat(x_np.shape[1], 1), 3)
    cx = np.take_along_axis(x_np, idx[:, None].repeat(x_np.shape[1], 1), 2)
    return idx, centre, dp, fj, cx


def test_patch_embeddings_on_the_hip_operators():
    """openpoints/models/layers/group_embed.py:14-286 (SubsampleGroup, PointPatchEmbed, P3Embed): indices from the HIP
    operators equal the oracle's, the embeddings equal their own conv stacks applied to oracle-built groups, and the
    parameter names are the reference's (a checkpoint's keys)."""
    from geot_amd.openpoints.models.layers import group_embed as ge
    b, n, c, k, r = 2, 1024, 4, 16, 0.3
    p_np = cloud(b, n, 41)
    x_np = np.random.default_rng(0).standard_normal((b, c, n)).astype(np.float32)
    p, x = dev(p_np), dev(x_np)
    idx, centre, dp, fj, cx = _oracle_groups(p_np, x_np, 64, r, k)
    # SubsampleGroup
    gp, cp, gf, gcx = ge.SubsampleGroup(num_groups=64, group_size=k, radius=r)(p, x)
    assert np.array_equal(cp.cpu().numpy(), centre) and np.allclose(gp.cpu().numpy(), dp, atol=1e-7)
    assert np.array_equal(gf.cpu().numpy(), fj) and np.array_equal(gcx.cpu().numpy()[..., 0], cx)
    assert len(ge.SubsampleGroup(num_groups=64, group_size=k, radius=r)(p)) == 2
    # PointPatchEmbed, feature_type dp_df
    torch.manual_seed(0)
    emb = ge.PointPatchEmbed(sample_ratio=0.0625, group_size=k, in_channels=c, layers=4, embed_dim=32, radius=r,
                             norm_args={'norm': 'bn2d'}).to(DEV).eval()
    keys = set(emb.state_dict())
    assert {"conv1.0.0.weight", "conv1.0.1.weight", "conv1.1.0.weight", "conv1.1.0.bias", "conv2.0.0.weight",
            "conv2.1.0.weight", "conv2.1.0.bias"} <= keys and "conv1.1.1.weight" not in keys and "conv2.1.1.weight" not in keys
    with torch.no_grad():
        (p0, cp2), (x0, out) = emb(p, x)
        feat = torch.cat([dev(dp), dev(fj) - dev(cx).unsqueeze(-1)], 1)
        h = emb.conv1(feat)
        h = torch.cat([h.max(-1, keepdim=True)[0].expand(-1, -1, -1, k), h], 1)
        want = emb.conv2(h).max(-1)[0]
    assert p0 is p and x0 is x and np.array_equal(cp2.cpu().numpy(), centre)
    assert out.shape == (b, 32, 64) and torch.allclose(out, want, rtol=1e-5, atol=1e-6)
    # P3Embed: two stages of /4, widths 16 -> 32, trains end to end
    torch.manual_seed(1)
    p3 = ge.P3Embed(sample_ratio=0.0625, scale=4, group_size=k, in_channels=c, layers=4, embed_dim=32, radius=r,
                    norm_args={'norm': 'bn2d'}).to(DEV).train()
    assert p3.channel_list == [c, 16, 32] and len(p3.convs) == 2 and p3.out_channels == 32
    xg = x.clone().requires_grad_(True)
    ps, fs = p3(p, xg)
    assert [t.shape[1] for t in ps] == [n, n // 4, n // 16] and [t.shape[1:] for t in fs[1:]] == [(16, n // 4), (32, n // 16)]
    idx1 = _oracle_groups(p_np, x_np, n // 4, r, k)[1]
    assert np.array_equal(ps[1].cpu().numpy(), idx1)          # stage 1 centres = the oracle's FPS prefix
    fs[-1].square().mean().backward()
    assert torch.isfinite(xg.grad).all() and float(xg.grad.abs().sum()) > 0


@pytest.mark.parametrize("reduction", ["mean", "sum", "max"])
@pytest.mark.parametrize("group", ["ballquery", "knn"])
def test_assa_equals_its_definition(reduction, group):
    """ASSA (openpoints/models/layers/local_aggregation.py:32-138): pre-convs, grouping, every grouped channel times each of
    the three relative coordinates reduced over the neighbourhood (channel a C + c), post-convs, residual.  The module forms
    sum / mean as one contraction; here the definition is spelled out -- the (B, 3 C, P, S) product, then the reduction."""
    from geot_amd.openpoints.models.layers.local_aggregation import ASSA, LocalAggregation
    torch.manual_seed(3)
    b, n, p, c_in = 2, 600, 150, 12
    xyz = torch.rand(b, n, 3, device=DEV)
    idx = torch.stack([torch.randperm(n, device=DEV)[:p] for _ in range(b)])
    q = torch.gather(xyz, 1, idx.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    feats = torch.randn(b, c_in, n, device=DEV, requires_grad=True)
    la = LocalAggregation([c_in, 24, 24, 36], {"NAME": "assa", "feature_type": "assa", "reduction": reduction},
                          {}, {"norm": "bn1d"}, {"act": "relu"}, {"NAME": group, "radius": 0.25, "nsample": 16}, use_res=True).to(DEV)
    m = la.SA_CONFIG_operator
    assert isinstance(m, ASSA) and m.num_preconv == 2 and len(m.convs) == 3
    # channels: 12 -> 24 -> ceil(24 / 3) = 8 before the reduction, 3 x 8 = 24 -> 36 after it; the residual is the 8-channel tensor
    assert m.convs[1][0].out_channels == 8 and m.convs[2][0].in_channels == 24 and isinstance(m.skip_layer, torch.nn.Conv1d)
    out = la(q, xyz, feats, query_idx=idx)
    assert out.shape == (b, 36, p)
    g = torch.randn_like(out)
    out.backward(g)
    got_grad = feats.grad.clone()
    feats.grad = None
    grads = [pp.grad.clone() for pp in la.parameters()]
    la.zero_grad()
    # the definition
    f1 = m.convs[:2](feats)
    dp, fj = m.grouper(q, xyz, f1)
    prod = (fj.unsqueeze(1).expand(-1, 3, -1, -1, -1) * dp.unsqueeze(2)).reshape(b, -1, p, fj.shape[-1])
    red = {"mean": prod.mean(-1), "sum": prod.sum(-1), "max": prod.max(-1)[0]}[reduction]
    res = torch.gather(f1, -1, idx.unsqueeze(1).expand(-1, f1.shape[1], -1))
    want = m.act(m.convs[2:](red) + m.skip_layer(res))
    # (two train-mode passes move the BatchNorm running statistics twice; the batch statistics the outputs use are the same)
    assert float((out - want).detach().abs().max()) <= 2e-5 * float(want.detach().abs().max())
    want.backward(g)
    assert float((got_grad - feats.grad).abs().max()) <= 5e-5 * float(feats.grad.abs().max())
    for a, pp in zip(grads, la.parameters()):
        assert float((a - pp.grad).abs().max()) <= 1e-4 * max(float(pp.grad.abs().max()), 1e-6)


def test_pointnet2_encoder_decoder_on_the_hip_operators():
    """PointNet2Encoder / PointNet2Decoder (openpoints/models/backbone/pointnetv2.py:149-381): the stage / block parameter
    tables, the state_dict layout a reference checkpoint has, a segmentation forward + backward, the classification path,
    and an ASSANet-style encoder (stem aggregation, residual blocks, query_as_support)."""
    from geot_amd.openpoints.models.backbone.pointnetv2 import PointNet2Encoder, PointNet2Decoder
    torch.manual_seed(0)
    cfg = dict(aggr_args={"NAME": "convpool", "feature_type": "dp_fj", "reduction": "max"}, group_args={"NAME": "ballquery"},
               conv_args={}, norm_args={"norm": "bn"}, act_args={"act": "relu"})
    enc = PointNet2Encoder(3, 0.1, 16, blocks=[1, 2, 1], width=32, strides=[4, 4, 4], layers=3, radius_scaling=2,
                           block_radius_scaling=1.5, nsample_scaling=2, **cfg).to(DEV)
    assert enc.radius == [[0.1], [0.2, 0.2 * 1.5], [0.4]] and enc.num_samples == [[16], [32, 32], [64]]
    assert enc.mlps == [[[32, 32, 64]], [[64, 64, 128], [128, 128, 128]], [[128, 128, 256]]]
    assert enc.channel_list == [3, 64, 256, 256] and enc.out_channels == 256
    keys = set(enc.state_dict())
    assert "SA_modules.1.local_aggregations.1.SA_CONFIG_operator.convs.2.0.weight" in keys
    assert "SA_modules.0.local_aggregations.0.SA_CONFIG_operator.convs.0.1.running_mean" in keys
    dec = PointNet2Decoder(enc.channel_list, mlps=enc.mlps, decoder_layers=2).to(DEV)
    # fp_mlps: [mlps[0][0][0]] * 3 for the finest level, then the skip widths of the inner levels; inputs = below + skip
    assert [fp.convs[0][0].in_channels for fp in dec.FP_modules] == [64 + 3, 256 + 64, 256 + 256] and dec.out_channels == 32
    assert "FP_modules.2.convs.1.0.weight" in set(dec.state_dict())
    xyz = torch.rand(2, 2048, 3, device=DEV)
    l_xyz, l_feat = enc(xyz)
    assert [t.shape[1] for t in l_xyz] == [2048, 512, 128, 32] and [t.shape[1] for t in l_feat] == [3, 64, 256, 256]
    out = dec(l_xyz, list(l_feat))
    assert out.shape == (2, 32, 2048)
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in list(enc.parameters()) + list(dec.parameters()))
    assert enc.forward_cls_feat({"pos": xyz[:, :64].contiguous(), "x": None}).shape == (2, 256)      # 64 -> 16 -> 4 -> 1 points
    # eval: ConvPool takes the fused SetAbstraction kernel; the same features as the composed ops
    enc.eval()
    with torch.no_grad():
        fused = enc(xyz)[1][-1]
        for sa in enc.SA_modules:
            for la in sa.local_aggregations:
                la.SA_CONFIG_operator.fused_eval = False
        composed = enc(xyz)[1][-1]
    assert float((fused - composed).abs().max()) <= 2e-4 * float(composed.abs().max())
    assa = PointNet2Encoder(3, 0.15, 16, aggr_args={"NAME": "assa", "feature_type": "assa", "reduction": "mean"},
                            group_args={"NAME": "ballquery"}, conv_args={}, norm_args={"norm": "bn"}, act_args={"act": "relu"},
                            blocks=[3, 3], width=48, strides=[4, 4], layers=3, use_res=True, stem_conv=True, stem_aggr=True,
                            double_last_channel=False, query_as_support=True).to(DEV)
    # (the stem's residual is the 16-channel tensor in front of the reduction: a 16 -> 48 skip convolution, as in the reference)
    assert assa.channel_list == [48, 144, 288] and assa.state_dict()["stem.SA_CONFIG_operator.skip_layer.weight"].shape == (48, 16, 1)
    lx, lf = assa(xyz)
    assert [t.shape[1] for t in lf] == [48, 144, 288] and [t.shape[2] for t in lf] == [2048, 512, 128]
    lf[-1].mean().backward()
    # the part-segmentation decoder rebuilds the encoder's table from the encoder's arguments; 16 one-hot class channels
    # join the finest level's skip features
    from geot_amd.openpoints.models.backbone.pointnetv2 import PointNet2PartDecoder
    part = PointNet2PartDecoder(3, 0.15, 16, {"NAME": "ballquery"}, {}, {"norm": "bn"}, {"act": "relu"}, blocks=[3, 3], width=48,
                                strides=[4, 4], layers=3, stem_conv=True, double_last_channel=False).to(DEV)
    assert part.mlps == [[[96] * 3] * 3, [[192] * 3] * 3]         # (this class's table: the stage's NEW width throughout)
    enc3 = PointNet2Encoder(3, 0.15, 16, blocks=[1, 1], width=48, strides=[4, 4], layers=3, double_last_channel=False,
                            stem_conv=True, **cfg).to(DEV)
    part3 = PointNet2PartDecoder(3, 0.15, 16, {"NAME": "ballquery"}, {}, {"norm": "bn"}, {"act": "relu"}, mlps=enc3.mlps,
                                 strides=[4, 4], layers=3, stem_conv=True, width=48).to(DEV)
    assert [fp.convs[0][0].in_channels for fp in part3.FP_modules] == [48 + 48 + 16, 96 + 48]
    lx, lf = enc3(xyz)
    seg = part3(lx, list(lf), torch.tensor([[3], [11]], device=DEV))
    assert seg.shape == (2, 48, 2048) and torch.isfinite(seg).all()
