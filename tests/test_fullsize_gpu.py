"""Full-size parity (BASELINE.json's sizes) in the GPU suite -- VERDICT r01 weak 10: the exact calls the bench
makes, compared with the oracle at 24 000 points, not at toy sizes; plus bounded versions of the soak scripts
(tools/fps_soak.py, tools/grid_soak.py) so that the accelerated searches meet their brute-force kernels on
full-size adversarial clouds inside `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch

from geot_amd.synth import make_batch, make_cloud, make_logits, region_labels

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("n,m", [(24000, 6000), (24000, 8192), (16000, 4000), (16000, 8192)])
def test_fps_k1_full_size_vs_oracle(n, m, oracle):
    """configs[1]'s own call: pointnet2 FPS (K1: <= 512-thread tie rule, origin skip) 24 000 -> 6000 / 8192, indices
    AND the final min-distance array; the same at the authors' 16 000-point operating point (default.yaml:6)."""
    from geot_amd.ext import pointnet2_ext as p2
    xyz = make_batch(2, n, start_index=100, dup_frac=0.01)[0]
    x = torch.from_numpy(xyz).to(DEV)
    want, wtemp = oracle.fps_dense(xyz, m, 512, True, return_temp=True)
    got = p2.furthest_point_sampling(x, m)
    assert np.array_equal(got.cpu().numpy(), want)
    from geot_amd.ext import pointnet2_batch_cuda as p2b            # K1' exposes temp: same kernel family, cap 1024
    out = torch.empty(2, m, dtype=torch.int32, device=DEV)
    temp = torch.full((2, n), 1e10, device=DEV)
    p2b.furthest_point_sampling_wrapper(2, n, m, x, temp, out)
    w2, t2 = oracle.fps_dense(xyz, m, 1024, False, return_temp=True)
    assert np.array_equal(out.cpu().numpy(), w2) and np.array_equal(temp.cpu().numpy(), t2)
    assert wtemp.shape == t2.shape


def test_sa_body_full_size_vs_composed_from_oracle_indices(oracle):
    """configs[1] end to end at 24000 / 6000 / 32 / [3,64,64,128]: FPS + ball query indices from the ORACLE, grouped
    tensor built with numpy, the same SharedMLP on torch (GPU fp32) + max -- against the module's fused HIP path."""
    from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(1609)
    sa = PointnetSAModuleVotes(mlp=[3, 64, 64, 128], npoint=6000, radius=0.1, nsample=32, use_xyz=True).to(DEV).eval()
    xyz = make_batch(1, 24000, start_index=0)[0]
    feats = np.random.default_rng(1609).standard_normal((1, 3, 24000)).astype(np.float32)
    with torch.no_grad():
        new_xyz, got, inds = sa(torch.from_numpy(xyz).to(DEV), torch.from_numpy(feats).to(DEV))
    w_inds = oracle.fps_dense(xyz, 6000, 512, True)
    assert np.array_equal(inds.cpu().numpy(), w_inds)
    c = np.take_along_axis(xyz, w_inds[..., None].astype(np.int64).repeat(3, -1), 1)
    assert np.array_equal(new_xyz.cpu().numpy(), c)
    idx = oracle.ball_query(c, xyz, 0.1, 32)
    gx = oracle.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx) - c.transpose(0, 2, 1)[..., None]
    gf = oracle.group_points(feats, idx)
    with torch.no_grad():
        y = sa.mlp_module(torch.from_numpy(np.concatenate([gx, gf], 1)).to(DEV))
        want = torch.nn.functional.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(-1)
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 2e-4 * scale       # MFMA accumulation order vs rocBLAS


def test_backbone_hotpath_step_b8_indices_vs_oracle(oracle):
    """workloads.backbone_hotpath_step's shapes at B = 8, N = 24 000: every index-producing launch of the step checked
    against the oracle (not `isfinite`), then the step itself runs forward + backward."""
    from geot_amd import workloads as wl
    from geot_amd.pointops.functions import pointops
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from geot_amd.knn_cuda import knn_sorted
    B = 8
    xyz = make_batch(B, 24000, start_index=0)[0]
    pts = torch.from_numpy(xyz).to(DEV)
    hot = wl.BackboneHotPath().to(DEV)
    nb, center, flat = hot.group(pts)
    w_c = oracle.fps_dense(xyz, 512, 512, True)
    c_np = np.take_along_axis(xyz, w_c[..., None].astype(np.int64).repeat(3, -1), 1)
    assert np.array_equal(center.cpu().numpy(), c_np)
    w_nn = oracle.knn_sorted(c_np, xyz, 32)[0]
    assert np.array_equal(flat.view(B, 512, 32).cpu().numpy() - (np.arange(B) * 24000)[:, None, None], w_nn)
    off = (np.arange(1, B + 1) * 24000).astype(np.int32)
    w8 = oracle.fps_offset(xyz.reshape(-1, 3), off, (np.arange(1, B + 1) * 8192).astype(np.int32)).reshape(B, 8192)
    with pointops.fps_prefix_scope():
        c8, c4 = pointops.fps(pts, 8192), pointops.fps(pts, 4096)
    flat_xyz = xyz.reshape(-1, 3)
    assert np.array_equal(c8.cpu().numpy(), flat_xyz[w8]) and np.array_equal(c4.cpu().numpy(), flat_xyz[w8[:, :4096]])
    c8n, c4n = c8.cpu().numpy(), c4.cpu().numpy()
    for unknown, known in ((c4n, c_np), (c8n, c_np), (xyz, c8n)):                 # propogation_2 / _1 / _0
        d, i3 = pu.three_nn(torch.from_numpy(unknown).to(DEV), torch.from_numpy(known).to(DEV))
        wd2, wi = oracle.three_nn(unknown, known)
        assert np.array_equal(i3.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), np.sqrt(wd2))
    for q, r in ((c4n, c_np), (c4n, c4n), (c8n, c4n), (c8n, c8n)):                # the four DGCNN kNN graphs, k = 4
        _, ki = knn_sorted(torch.from_numpy(q).to(DEV), torch.from_numpy(r).to(DEV), 4)
        assert np.array_equal(ki.cpu().numpy(), oracle.knn_sorted(q, r, 4)[0])
    tokens = torch.randn(B, wl.TRANS_DIM, wl.GROUPS, device=DEV)
    loss = wl.backbone_hotpath_step(hot, pts, tokens)
    assert torch.isfinite(loss)


def test_ntm_half_step_2x24000_vs_fp64_restatement(oracle):
    """The NTM block at configs[4]'s per-rank size (B_u = 2 clouds x 24 000 points): sig_t_mean, class transition,
    logit correction and threeD_space_loss(k = 32) forward values against oracle/np_ntm.py (fp64), kNN graph bit-exact."""
    from geot_amd import ntm
    from oracle import np_ntm
    C = 17
    xyz = make_batch(2, 24000, start_index=30)[0]
    pw_np, ps_np = make_logits(xyz, index=30), make_logits(xyz, index=31, sharp=3.0)
    pos, pw, ps = (torch.from_numpy(a).to(DEV) for a in (xyz, pw_np, ps_np))
    torch.manual_seed(4)
    pred = ntm.Ins_T_mean(nclasses=C).to(DEV)
    W = torch.stack([l.weight for l in pred.T_predictor.fc]).detach().cpu().numpy()
    cm = np.eye(C, dtype=np.float32) * 0.9 + 0.1 / C
    ema = np.eye(C, dtype=np.float32) * 0.9 + 0.1 / C
    sigma = np.full(C, 0.4, np.float32)

    def softmax(z):
        e = np.exp(z - z.max(1, keepdims=True))
        return (e / e.sum(1, keepdims=True)).astype(np.float32)
    eta_np, p_np = softmax(pw_np), softmax(ps_np)
    ema_corr, ema_next, class_T, prior_T = ntm.class_transition(torch.softmax(pw, 1), torch.from_numpy(sigma).to(DEV),
                                                                torch.from_numpy(ema).to(DEV))
    w = np_ntm.class_transition(eta_np, sigma, ema)
    w_corr, w_next, w_class, w_prior = w["ema_t_corr"], w["ema_t_next"], w["class_T"], w["prior_T"]
    def row_close(got, want, what):
        """north_star's 1e-5, relative to the scale of the element's ROW (a transition-matrix row sums to 1)."""
        got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
        ratio = np.abs(got - want) / np.maximum(np.abs(want).max(-1, keepdims=True), 1e-30)
        assert ratio.max() <= 1e-5, (what, float(ratio.max()))
    for got, want, what in ((ema_corr, w_corr, "ema_t_corr"), (ema_next, w_next, "ema_t_next"), (class_T, w_class, "class_T"),
                            (prior_T, w_prior, "prior_T")):
        row_close(got.cpu().numpy(), want, what)
    ins_t = pred(torch.softmax(ps, 1), torch.from_numpy(cm).to(DEV))
    w_ins = np_ntm.sig_t_mean(p_np, cm, W)
    row_close(ins_t.detach().cpu().numpy(), w_ins, "ins_T")
    corr = ntm.correct_logits(ps, ins_t, ema_corr, 0.9)
    w_newT, w_pred = np_ntm.correct_logits(ps_np, w_ins, w_corr, 0.9)
    # a corrected logit is a 17-term dot product logits_i . newT_i[:, c] with terms of both signs: 1e-5 relative to the sum of
    # the magnitudes of its terms (the bound of a backward-stable dot product), element by element
    cond = np_ntm.correct_logits(np.abs(ps_np), w_ins, w_corr, 0.9)[1]
    err = np.abs(corr.detach().cpu().numpy().astype(np.float64) - w_pred)
    assert (err <= 1e-5 * cond).all(), float((err / cond).max())
    crit = ntm.threeD_space_loss(k=32, sigma=1.0)
    nbr = crit.neighbours(pos)
    widx = oracle.knn_sorted(xyz, xyz, 33)[0][:, :, 1:]
    assert np.array_equal(nbr.cpu().numpy(), widx)
    labels = eta_np.argmax(1)
    loss = crit(pos, torch.from_numpy(labels).to(DEV), ins_t)
    want = np_ntm.threed_space_loss(xyz, labels, w_ins, widx, 1.0)[0]
    assert abs(loss.item() - want) <= 1e-5 * abs(want) + 1e-9, (loss.item(), want)


def test_fps_soak_pruned_equals_unpruned_on_full_size_adversarial_clouds():
    """tools/fps_soak.py, bounded: multi-commit pruned FPS == the unpruned kernel (indices and final temp, both tie
    rules) on 2 x 16 clouds of 24 000 points with 0-3 % duplicates, the second batch on quantised coordinates."""
    from geot_amd.ext import pointnet2_ext as p2, pointnet2_batch_cuda as p2b
    try:
        for chunk in range(2):
            B, n, m = 16, 24000, 6000
            xyz = np.stack([make_cloud(n, 5000 + chunk * 100 + i, dup_frac=0.01 * (i % 4))[0] for i in range(B)])
            if chunk % 2:
                xyz = ((xyz * 512).round() / 512).astype(np.float32)
            x = torch.from_numpy(xyz).to(DEV)
            res = {}
            for impl in ("multi", "basic"):
                os.environ["GEOT_FPS_IMPL"] = impl
                out = torch.empty(B, m, dtype=torch.int32, device=DEV)
                temp = torch.full((B, n), 1e10, device=DEV)
                p2b.furthest_point_sampling_wrapper(B, n, m, x, temp, out)
                res[impl] = (out.clone(), temp.clone(), p2.furthest_point_sampling(x, 2048).clone())
            assert all(torch.equal(a, b) for a, b in zip(res["multi"], res["basic"])), chunk
    finally:
        os.environ.pop("GEOT_FPS_IMPL", None)


def test_grid_soak_grid_search_equals_brute_force_on_adversarial_clouds():
    """tools/grid_soak.py, bounded: grid kNN / ball query == the brute-force wave kernels on flat, quantised, dense-core,
    few-location and blob-plus-outlier clouds (4 trials)."""
    from geot_amd.knn_cuda import knn_sorted
    from geot_amd.ext import pointnet2_ext as p2
    rng = np.random.default_rng(7)
    try:
        for trial in range(4):
            n = int(rng.choice([5000, 12000, 24000]))
            clouds = []
            for i in range(6):
                x = make_cloud(n, 7000 + trial * 10 + i, dup_frac=0.02 * (i % 2))[0]
                if i == 1:
                    x = x * np.array([1.0, 1.0, 0.02], np.float32)
                if i == 2:
                    x = (x * 64).round() / 64
                if i == 3:
                    x[: n // 3] *= 0.05
                if i == 4:
                    x = x[rng.integers(0, int(rng.integers(20, 400)), n)]
                if i == 5:
                    x = x * 1e-3
                    x[rng.integers(0, n, 5)] += 3.0
                clouds.append(np.ascontiguousarray(x, dtype=np.float32))
            ref = torch.from_numpy(np.stack(clouds)).to(DEV)
            q = torch.cat([ref[:, : n // 2], ref[:, :500] * 1.7 + 0.01], 1).contiguous()
            for k in (3, 33, 64):
                os.environ["GEOT_NN_IMPL"] = "grid"
                dg, ig = knn_sorted(q, ref, k)
                os.environ["GEOT_NN_IMPL"] = "wave"
                db, ib = knn_sorted(q, ref, k)
                assert torch.equal(ig, ib) and torch.equal(dg, db), (trial, k)
            for r, ns in ((0.05, 16), (0.25, 64)):
                os.environ["GEOT_NN_IMPL"] = "grid"
                a = p2.ball_query(q, ref, r, ns)
                os.environ["GEOT_NN_IMPL"] = "wave"
                assert torch.equal(a, p2.ball_query(q, ref, r, ns)), (trial, r, ns)
    finally:
        os.environ.pop("GEOT_NN_IMPL", None)


def test_gather_gradients_of_the_model_are_bit_reproducible():
    """The gradients the configured backbone takes through the hot path -- three_interpolate at prop0's shape (source
    rows cut into parts: the reverse-index gather loops them inside the workgroup, one writer per output) and the
    EdgeConv tail -- are identical from run to run (no float atomics on these paths)."""
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from geot_amd.openpoints.models.backbone.transformer_ops import edgeconv_tail
    xyz = make_batch(2, 24000, start_index=3)[0]
    pos = torch.from_numpy(xyz).to(DEV)
    known = pos[:, :8192].contiguous()
    dist, idx = pu.three_nn(pos, known)
    w = 1.0 / (dist + 1e-8)
    w = w / w.sum(2, keepdim=True)
    up = torch.randn(2, 96, 24000, device=DEV)
    grads = []
    for _ in range(3):
        f = torch.randn(2, 96, 8192, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1), requires_grad=True)
        (pu.three_interpolate(f, idx, w) * up).sum().backward()
        grads.append(f.grad.clone())
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    norm = torch.nn.GroupNorm(4, 64).to(DEV)
    nbr = torch.randint(0, 4096, (2, 8192, 4), device=DEV, dtype=torch.int32)
    outs = []
    for _ in range(2):
        g = torch.Generator(device=DEV).manual_seed(2)
        p = torch.randn(2, 64, 4096, device=DEV, generator=g, requires_grad=True)
        q = torch.randn(2, 64, 8192, device=DEV, generator=g, requires_grad=True)
        edgeconv_tail(p, q, nbr, norm, 0.2).square().sum().backward()
        outs.append((p.grad.clone(), q.grad.clone(), norm.weight.grad.clone()))
        norm.zero_grad()
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.parametrize("npts", [24000, 16000])
def test_fixmatch_iteration_at_the_configured_sizes(npts):
    """(16 000 points, B_l = B_u = 2: the sizes the authors train at, default.yaml:6, ...fixmatch_ntm.yaml:72-73.)
    BASELINE configs[4] as bench.py runs it: the full PointTransformer_seg_T (depth 12, 512 x 32 groups, targets
    8192 / 4096 / 2048) as student and frozen teacher, B_l = B_u = 2 clouds of 24 000 points, two FixMatch+NTM
    iterations: every loss finite, the 3-D loss non-negative, the EMA transition matrix a row-stochastic mixture, both
    optimisers stepped; and the same two iterations with the teacher on the main stream give the same losses."""
    import torch
    from geot_amd import train_step as ts
    from geot_amd.synth import make_batch
    dev = torch.device("cuda:0")
    xyz, lab = make_batch(2, npts, start_index=0)
    pos, target = torch.from_numpy(xyz).to(dev), torch.from_numpy(lab).long().to(dev)
    xu = torch.from_numpy(make_batch(2, npts, start_index=7)[0]).to(dev)
    xs = (xu * 1.05).contiguous()
    z = torch.zeros(2, 1, dtype=torch.long, device=dev)
    data = {"pos": pos, "x": pos.transpose(1, 2).contiguous(), "cls": z, "y": target}
    data_u = {"pos_w": xu, "x_w": xu.transpose(1, 2).contiguous(), "cls_w": z, "pos_s": xs,
              "x_s": xs.transpose(1, 2).contiguous(), "cls_s": z, "raw_pos": xu}
    runs = []
    for overlap in (True, False):
        torch.manual_seed(11)
        trainer = ts.build_fixmatch(dev, use_ddp=False)
        trainer.overlap_teacher = overlap
        w0 = [l.weight.detach().clone() for l in trainer.T_predictor.T_predictor.fc]
        torch.manual_seed(12)
        out = [trainer(data, data_u) for _ in range(2)]
        torch.cuda.synchronize()
        for o in out:
            assert all(bool(torch.isfinite(v)) for v in o.values()), o
            assert float(o["threed"]) >= 0
        rows = trainer.ema_t.sum(1)
        assert torch.allclose(rows, torch.ones_like(rows), atol=1e-3), rows
        assert any(not torch.equal(a, l.weight) for a, l in zip(w0, trainer.T_predictor.T_predictor.fc))
        runs.append([float(v) for o in out for v in o.values()])
    # the iteration is bit-reproducible (no float atomics left on its path since round 3: profiles/r03_determinism.txt), with
    # the teacher on its own stream or in line: both iterations give the same losses to the last bit
    for a, b in zip(*runs):
        assert a == b, runs


@pytest.mark.parametrize("n", [24576, 24577, 32768])
def test_fps_at_the_switch_between_the_pruned_and_the_streaming_kernel(n, oracle):
    """n <= 24 576 points live in the registers of one workgroup (bucket-pruned kernel); one point more and the
    unpruned / streaming kernels take over.  Both sides of the wall against the oracle: K1 (origin skip, 512-thread tie
    rule), K1' (1024) and the offset-batched K2, with duplicates."""
    from geot_amd.ext import pointnet2_ext as p2
    from geot_amd.pointops.functions import pointops
    from geot_amd.openpoints.models.layers import subsample
    xyz = make_batch(2, n, start_index=300, dup_frac=0.01)[0]
    x = torch.from_numpy(xyz).to(DEV)
    m = 1500
    assert np.array_equal(p2.furthest_point_sampling(x, m).cpu().numpy(), oracle.fps_dense(xyz, m, 512, True))
    assert np.array_equal(subsample.furthest_point_sample(x, m).cpu().numpy(), oracle.fps_dense(xyz, m, 1024, False))
    off = (np.arange(1, 3) * n).astype(np.int32)
    want = oracle.fps_offset(xyz.reshape(-1, 3), off, (np.arange(1, 3) * m).astype(np.int32)).reshape(2, m)
    assert np.array_equal(pointops.fps(x, m).cpu().numpy(), xyz.reshape(-1, 3)[want])


def test_validation_path_at_the_size_of_a_real_scan(oracle):
    """train.py:781-800 get_pred_whole at full size: logits of the 24 000 sampled points carried to the ~1e5 vertices of
    the whole scan by three_nn + inverse-distance interpolation -- neighbour ids and squared distances bit-exact against
    the oracle at 100 003 x 24 000, the predicted labels identical wherever the two best classes are not numerically tied."""
    from geot_amd.validation import get_pred_whole
    from geot_amd.ext import pointnet2_ext as p2
    rng = np.random.default_rng(21)
    n, m, c = 24000, 100003, 17
    pts = make_batch(1, n, start_index=55)[0]
    whole = (make_cloud(m, 56)[0] * np.float32(1.02)).astype(np.float32)[None]          # another scan of the same arch, denser
    d2, idx = p2.three_nn(torch.from_numpy(whole).to(DEV), torch.from_numpy(pts).to(DEV))
    wd2, widx = oracle.three_nn(whole, pts)
    assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(d2.cpu().numpy(), wd2)
    logits = (rng.normal(size=(1, c, n)) * 3).astype(np.float32)
    center, scale = [np.zeros((1, 3), np.float32)], [np.float32(1.0)]
    pred = get_pred_whole(torch.from_numpy(logits).to(DEV), torch.from_numpy(pts).to(DEV), [torch.from_numpy(whole[0])],
                          [torch.from_numpy(center[0])], [torch.tensor(scale[0])])[0][0].cpu().numpy()
    e = np.exp(logits - logits.max(1, keepdims=True))
    sm = (e / e.sum(1, keepdims=True)).astype(np.float32)
    r = 1.0 / (np.sqrt(wd2) + np.float32(1e-8))
    lw = oracle.three_interpolate(sm, widx, (r / r.sum(2, keepdims=True)).astype(np.float32))[0]
    top2 = np.sort(lw, 0)[-2:]
    clear = (top2[1] - top2[0]) > 1e-5
    assert pred.shape == (m,) and clear.mean() > 0.99
    assert np.array_equal(pred[clear], lw.argmax(0)[clear])


def test_supervised_step_at_the_bench_batch_size():
    """configs[2] as bench.py runs it -- 8 clouds x 24 000 points through the full PointTransformer_seg_T -- inside the suite
    (that step had only run in the bench).  Size-independent properties: (a) re-ordering the clouds of the
    batch re-orders the logits (every per-cloud op of the path -- sampling, grouping, kNN, interpolation, EdgeConv -- sees one
    cloud at a time; BatchNorm's batch statistics see the same multiset in another summation order); (b) three training
    steps with look-ahead between two alternating batches stay finite and reduce the loss on the first batch."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
    from geot_amd.train_step import SupervisedStep
    xyz, _ = make_batch(8, 24000, start_index=20)
    pos = torch.from_numpy(xyz).to(DEV)
    target = torch.from_numpy(region_labels(xyz)).to(DEV)
    cls = torch.zeros(8, 1, dtype=torch.long, device=DEV)
    torch.manual_seed(11)
    model = PointTransformer_seg_T(**dict(TOOTH_SEG_CFG, drop_path_rate=0.0)).to(DEV)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0                         # (a) compares two forward passes: no random numbers in them
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=DEV)
    with torch.no_grad():
        a = model(pos, pos.transpose(1, 2).contiguous(), cls)[0]
        b = model(pos[perm].contiguous(), pos[perm].transpose(1, 2).contiguous(), cls)[0]
    scale = float(a.abs().max())
    assert float((b - a[perm]).abs().max()) <= 2e-4 * scale, (float((b - a[perm]).abs().max()), scale)
    del a, b
    xyz2, _ = make_batch(8, 24000, start_index=40)
    pos2 = torch.from_numpy(xyz2).to(DEV)
    target2 = torch.from_numpy(region_labels(xyz2)).to(DEV)
    step = SupervisedStep(model, lr=1e-3)
    losses = []
    for it in range(5):
        cur, nxt = ((pos, target), (pos2, target2)) if it % 2 == 0 else ((pos2, target2), (pos, target))
        losses.append(float(step(cur[0], cls, cur[1], next_pos=nxt[0])))
    assert all(np.isfinite(losses)) and losses[4] < losses[0], losses
