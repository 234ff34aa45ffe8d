"""GPU tests of the runnable backbone and the training steps built on the HIP hot path (BASELINE configs[2]/[4]):
the model's index-producing steps against the C oracle at full size, the factored dense mode against the
reference op order, and one FixMatch+NTM iteration end to end."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SMALL = dict(trans_dim=384, depth=3, num_heads=4, group_size=32, num_group=128, encoder_dims=256, nclasses=17,
             drop_path_rate=0.0, downsample_targets=[2048, 1024, 512], extract_layers=[1, 2, 3])


def _batch(b, n, dev):
    from geot_amd.synth import make_batch, region_labels
    xyz, _ = make_batch(b, n)
    return xyz, torch.from_numpy(xyz).to(dev), torch.from_numpy(region_labels(xyz)).to(dev)


def test_factored_dense_equals_reference_order():
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    dev = torch.device("cuda:0")
    _, pos, target = _batch(2, 6000, dev)
    cls = torch.tensor([[0], [1]], device=dev)
    torch.manual_seed(0)
    ref = PointTransformer_seg_T(**SMALL, dense="reference").to(dev).train()
    fac = PointTransformer_seg_T(**SMALL, dense="factored").to(dev).train()
    fac.load_state_dict(ref.state_dict())
    for m in (ref, fac):
        m.seg_head[2].p = 0.0                       # the only random layer left (drop_path_rate is 0)
    outs, grads = [], []
    for m in (ref, fac):
        logit, corr, sigma, f_l0 = m(pos, pos.transpose(1, 2).contiguous(), cls, torch.eye(17, device=dev))
        assert logit.shape == (2, 17, 6000) and f_l0.shape == (2, 384, 6000) and corr.shape == (17, 17)
        torch.nn.functional.cross_entropy(logit, target).backward()
        outs.append((logit.detach(), f_l0.detach()))
        grads.append({n: p.grad.detach() for n, p in m.named_parameters() if p.grad is not None})
    for a, b in zip(outs[0], outs[1]):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-4 * float(a.abs().max())), float((a - b).abs().max())
    assert set(grads[0]) == set(grads[1])
    for k in grads[0]:
        if k in ("encoder.first_conv.0.bias", "encoder.first_conv.3.bias", "encoder.second_conv.0.bias", "seg_head.0.bias"):
            # a bias in front of a BatchNorm has a mathematically zero gradient (first_conv.3's shifts both halves of
            # the concatenation second_conv.0 -> BatchNorm sees): what is left is rounding noise
            wk = k[:-4] + "weight"
            assert float(grads[0][k].norm()) < 1e-3 * float(grads[0][wk].norm())
            assert float(grads[1][k].norm()) < 1e-3 * float(grads[1][wk].norm())
            continue
        err = float((grads[0][k] - grads[1][k]).norm() / (grads[0][k].norm() + 1e-20))
        assert err < 1e-2, (k, err)     # fp32 summation order + max ties after normalisation, through 30 layers


@pytest.mark.parametrize("npts", [24000, 16000])
def test_model_sampling_steps_match_oracle_full_size(oracle, npts):
    """The index-producing steps of one forward at the configs[2] shapes (B=2 of the 8 clouds to keep the oracle in
    seconds): Group's FPS(K1, origin-skip)+kNN, pointops.fps prefixes (K2), three_nn of every FP module, DGCNN kNN.
    16 000 points = the authors' own operating point (cfgs/tooth_semi/default.yaml:6 num_points)."""
    from geot_amd.openpoints.models.backbone import transformer as tr
    from geot_amd.pointops.functions import pointops
    from geot_amd.pointnet2 import pointnet2_utils as pu
    dev = torch.device("cuda:0")
    xyz_np, pos, _ = _batch(2, npts, dev)
    group = tr.Group(512, 32)
    neighborhood, center, flat = group(pos)
    want_c = oracle.fps_dense(xyz_np, 512, 512, True)
    c_np = np.take_along_axis(xyz_np, want_c[..., None].astype(np.int64).repeat(3, -1), 1)
    assert np.array_equal(center.cpu().numpy(), c_np)
    want_nn = oracle.knn_sorted(c_np, xyz_np, 32)[0]
    got_nn = flat.view(2, 512, 32).cpu().numpy() - (np.arange(2) * npts)[:, None, None]
    assert np.array_equal(got_nn, want_nn)
    off = (np.arange(1, 3) * npts).astype(np.int32)
    want8 = oracle.fps_offset(xyz_np.reshape(-1, 3), off, (np.arange(1, 3) * 8192).astype(np.int32)).reshape(2, 8192)
    c8 = pointops.fps(pos, 8192)
    c4 = pointops.fps(pos, 4096)
    flat_xyz = xyz_np.reshape(-1, 3)
    assert np.array_equal(c8.cpu().numpy(), flat_xyz[want8])
    assert np.array_equal(c4.cpu().numpy(), flat_xyz[want8[:, :4096]])            # prefix property (App. A.1)
    d, i3 = pu.three_nn(pos, c8)
    wd2, wi = oracle.three_nn(xyz_np, c8.cpu().numpy())
    assert np.array_equal(i3.cpu().numpy(), wi) and np.array_equal(d.cpu().numpy(), np.sqrt(wd2))
    idx = tr._knn_idx(c8.transpose(1, 2).contiguous(), c4.transpose(1, 2).contiguous(), 4)
    assert np.array_equal(idx.cpu().numpy(), oracle.knn_sorted(c8.cpu().numpy(), c4.cpu().numpy(), 4)[0])


def test_supervised_step_trains():
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd.train_step import SupervisedStep
    dev = torch.device("cuda:0")
    _, pos, target = _batch(2, 6000, dev)
    torch.manual_seed(1)
    model = PointTransformer_seg_T(**SMALL).to(dev)
    step = SupervisedStep(model, lr=1e-3)
    cls = torch.zeros(2, 1, dtype=torch.long, device=dev)
    losses = [float(step(pos, cls, target)) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def test_fixmatch_ntm_step_end_to_end():
    from geot_amd import train_step as ts
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    trainer = ts.build_fixmatch(dev, seg_cfg=SMALL, use_ddp=False)
    _, pos, target = _batch(2, 6000, dev)
    from geot_amd.synth import make_batch
    xu = torch.from_numpy(make_batch(2, 6000, start_index=50)[0]).to(dev)
    xs = (xu * 1.1).contiguous()
    z = torch.zeros(2, 1, dtype=torch.long, device=dev)
    data = {"pos": pos, "x": pos.transpose(1, 2).contiguous(), "cls": z, "y": target}
    data_u = {"pos_w": xu, "x_w": xu.transpose(1, 2).contiguous(), "cls_w": z, "pos_s": xs,
              "x_s": xs.transpose(1, 2).contiguous(), "cls_s": z, "raw_pos": xu}
    w0 = [l.weight.detach().clone() for l in trainer.T_predictor.T_predictor.fc]
    s0 = trainer.model.segmentor.sigma.detach().clone()
    out = [trainer(data, data_u) for _ in range(2)]
    for o in out:
        assert all(torch.isfinite(v) for v in o.values()), o
    assert float(out[0]["threed"]) >= 0
    assert not torch.equal(trainer.ema_t, torch.eye(17, device=dev))                    # EMA moved (train.py:556-557)
    assert any(not torch.equal(a, l.weight) for a, l in zip(w0, trainer.T_predictor.T_predictor.fc))   # T_optimizer stepped
    assert not torch.equal(s0, trainer.model.segmentor.sigma)                           # sigma learns through the prior
    assert all(not p.requires_grad for p in trainer.model_t.parameters())               # frozen teacher


def test_fixmatch_teacher_on_its_own_stream_gives_the_same_iteration():
    """FixMatchNTMStep queues the frozen teacher's forward on a stream of its own beside the student's forward: two
    iterations from the same initialisation and random state, with and without it, give the same losses."""
    from geot_amd import train_step as ts
    from geot_amd.synth import make_batch
    dev = torch.device("cuda:0")
    _, pos, target = _batch(2, 6000, dev)
    xu = torch.from_numpy(make_batch(2, 6000, start_index=50)[0]).to(dev)
    xs = (xu * 1.1).contiguous()
    z = torch.zeros(2, 1, dtype=torch.long, device=dev)
    data = {"pos": pos, "x": pos.transpose(1, 2).contiguous(), "cls": z, "y": target}
    data_u = {"pos_w": xu, "x_w": xu.transpose(1, 2).contiguous(), "cls_w": z, "pos_s": xs,
              "x_s": xs.transpose(1, 2).contiguous(), "cls_s": z, "raw_pos": xu}
    runs = []
    for overlap in (False, True):
        torch.manual_seed(4)
        trainer = ts.build_fixmatch(dev, seg_cfg=SMALL, use_ddp=False)
        trainer.overlap_teacher = overlap
        torch.manual_seed(5)
        out = [trainer(data, data_u) for _ in range(2)]
        torch.cuda.synchronize()
        runs.append([float(v) for o in out for v in o.values()])
    for a, b in zip(*runs):
        assert abs(a - b) <= 2e-5 * abs(a) + 1e-7, runs         # float atomics in the graph loss: last-bit differences


@pytest.mark.parametrize("b,c,nq,nk,k,groups", [(2, 64, 1000, 700, 4, 4), (1, 32, 333, 333, 5, 2), (2, 512, 4096, 512, 4, 4),
                                                 (1, 24, 2048, 9000, 3, 1), (1, 16, 3000, 41, 4, 2), (2, 6, 100, 1, 4, 1)])
def test_edgeconv_tail_matches_composed(b, c, nq, nk, k, groups):
    """Fused gather + GroupNorm + LeakyReLU + max (csrc/edgeconv.hip) vs the composed torch ops of
    transformer.py:366-379 on the same P, Q, idx -- forward and every gradient.  (Source counts 512, 41 and 1 under thousands
    of queries: the dP walk gives a target 4 / 8 lanes; 9000 sources under 2048 x 3 pairs: most lists are empty.)"""
    from geot_amd.openpoints.models.backbone.transformer_ops import edgeconv_tail
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(c + nq)
    p0 = torch.randn(b, c, nk, generator=g).to(dev)
    q0 = torch.randn(b, c, nq, generator=g).to(dev)
    idx = torch.randint(0, nk, (b, nq, k), generator=g).to(torch.int32).to(dev)
    norm = torch.nn.GroupNorm(groups, c).to(dev)
    with torch.no_grad():
        norm.weight.copy_(torch.randn(c, generator=g))          # both signs: max and min selection
        norm.bias.copy_(torch.randn(c, generator=g))
    up = torch.randn(b, c, nq, generator=g).to(dev)
    res = []
    for fused in (False, True):         # the composed reference in fp64: in fp32 two slots whose normalised values round
        dt = torch.float32 if fused else torch.float64   # to the same float tie, and torch's max may then pick the other one
        p, q = p0.clone().to(dt).requires_grad_(True), q0.clone().to(dt).requires_grad_(True)
        n2 = torch.nn.GroupNorm(groups, c).to(dev).to(dt)
        n2.load_state_dict(norm.state_dict())
        if fused:
            out = edgeconv_tail(p, q, idx, n2, 0.2)
        else:
            y = torch.gather(p, 2, idx.long().reshape(b, 1, nq * k).expand(-1, c, -1)).view(b, c, nq, k) + q.unsqueeze(-1)
            out = torch.nn.functional.leaky_relu(n2(y), 0.2).max(dim=-1)[0]
        (out * up.to(dt)).sum().backward()
        res.append([t.double() for t in (out.detach(), p.grad, q.grad, n2.weight.grad, n2.bias.grad)])
    for name, a, f in zip(("out", "dP", "dQ", "dgamma", "dbeta"), *res):
        scale = float(a.abs().max())
        assert float((a - f).abs().max()) <= 2e-5 * scale, (name, float((a - f).abs().max()), scale)


@pytest.mark.parametrize("b,c,l,relu,training", [(2, 37, 1000, True, True), (3, 8, 4099, False, True), (1, 64, 24000, True, True),
                                                 (2, 16, 513, True, False)])
def test_bn_act_equals_torch_batchnorm_relu(b, c, l, relu, training):
    """fused_norm.bn_act vs nn.BatchNorm1d (+ ReLU): outputs, every gradient, running statistics, eval mode."""
    from geot_amd.fused_norm import bn_act
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(b * 100 + c)
    x0 = (torch.randn(b, c, l, generator=g) * 2 + 0.5).to(dev)
    up = torch.randn(b, c, l, generator=g).to(dev)
    ref, ours = torch.nn.BatchNorm1d(c).to(dev).double(), torch.nn.BatchNorm1d(c).to(dev)   # fp64 reference: MIOpen's fp32
    with torch.no_grad():                                                                    # statistics are ~1e-4 off
        ours.weight.copy_(torch.randn(c, generator=g)); ours.bias.copy_(torch.randn(c, generator=g))
        ours.running_mean.uniform_(-0.5, 0.5); ours.running_var.uniform_(0.5, 2.0)
    ref.load_state_dict(ours.state_dict())
    ref.train(training); ours.train(training)
    res = []
    for mod, fused in ((ref, False), (ours, True)):
        x = (x0.double() if not fused else x0.clone()).requires_grad_(True)
        for _ in range(2):                                   # two steps: the running statistics move twice
            y = bn_act(mod, x, relu=relu) if fused else (torch.relu(mod(x)) if relu else mod(x))
        (y * up.to(y.dtype)).sum().backward()
        res.append([t.double() for t in (y.detach(), x.grad, mod.weight.grad, mod.bias.grad, mod.running_mean, mod.running_var)]
                   + [mod.num_batches_tracked.clone()])
    for name, a, f in zip(("out", "dx", "dgamma", "dbeta", "running_mean", "running_var"), *[r[:6] for r in res]):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - f).abs().max()) <= 2e-5 * scale + 1e-6, (name, float((a - f).abs().max()), scale)
    assert int(res[0][6]) == int(res[1][6])


@pytest.mark.parametrize("training", [True, False])
def test_bn_act_with_the_bias_of_the_layer_in_front(training):
    """bn_act(bn, y, pre_bias=b) == relu(bn(y + b)) without the add: output, dx, d gamma, d beta, the running
    statistics (the running mean sees the bias) and d b -- zero under batch statistics (the bias cancels), scale *
    sum g under running statistics."""
    from geot_amd.fused_norm import bn_act
    dev = torch.device("cuda:0")
    b, c, l = 3, 40, 777
    g = torch.Generator().manual_seed(3)
    y0 = (torch.randn(b, c, l, generator=g) * 2 + 0.5).to(dev)
    bias0 = torch.randn(c, generator=g).to(dev)
    up = torch.randn(b, c, l, generator=g).to(dev)
    ref, ours = torch.nn.BatchNorm1d(c).to(dev).double(), torch.nn.BatchNorm1d(c).to(dev)
    with torch.no_grad():
        ours.weight.copy_(torch.randn(c, generator=g)); ours.bias.copy_(torch.randn(c, generator=g))
        ours.running_mean.uniform_(-0.5, 0.5); ours.running_var.uniform_(0.5, 2.0)
    ref.load_state_dict(ours.state_dict())
    ref.train(training); ours.train(training)
    res = []
    for mod, fused in ((ref, False), (ours, True)):
        y = (y0.double() if not fused else y0.clone()).requires_grad_(True)
        bias = (bias0.double() if not fused else bias0.clone()).requires_grad_(True)
        out = bn_act(mod, y, relu=True, pre_bias=bias) if fused else torch.relu(mod(y + bias.view(1, -1, 1)))
        (out * up.to(out.dtype)).sum().backward()
        res.append([t.double() for t in (out.detach(), y.grad, mod.weight.grad, mod.bias.grad, bias.grad, mod.running_mean,
                                         mod.running_var)])
    for name, a, f in zip(("out", "dy", "dgamma", "dbeta", "dbias", "running_mean", "running_var"), *res):
        scale = float(res[0][1].abs().max() * l * b) if name == "dbias" and training else float(a.abs().max()) + 1e-12
        assert float((a - f).abs().max()) <= 2e-5 * scale + 1e-6, (name, float((a - f).abs().max()), scale)
    if training:
        assert float(res[1][4].abs().max()) == 0.0          # exact zero, where torch accumulates rounding noise


def test_add_last_broadcast_gradient_is_the_row_sum():
    from geot_amd.fused_norm import add_last_broadcast
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    a0, p0 = torch.randn(70, 333, 32, device=dev), torch.randn(70, 333, device=dev)
    up = torch.randn(70, 333, 32, device=dev)
    res = []
    for fused in (False, True):
        a, p = a0.clone().requires_grad_(True), p0.clone().requires_grad_(True)
        y = add_last_broadcast(a, p) if fused else a + p.unsqueeze(-1)
        (y * up).sum().backward()
        res.append((y.detach(), a.grad, p.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    want = up.double().sum(-1)
    assert float((res[1][2].double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("alpha,gamma", [(0.25, 2.0), (-1.0, 1.5), (0.6, 1.0)])
def test_fused_poly1_focal_losses_equal_the_torch_composition(alpha, gamma):
    """csrc/loss.hip (integer labels, no one-hot tensors) vs the reference's op chain in fp64: Poly1FocalLoss and the
    confidence-masked Poly1FocalLoss_U_corr, values and gradients; extreme logits included."""
    from geot_amd.openpoints.loss import Poly1FocalLoss, Poly1FocalLoss_U_corr
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    b, c, n = 3, 17, 1111
    logits0 = torch.randn(b, c, n, generator=g) * 4
    logits0[0, :, :5] = torch.tensor([-60.0, 60.0, 0.0, 30.0, -30.0])
    labels = torch.randint(0, c, (b, n), generator=g)
    conf = torch.rand(b, n, generator=g)
    for cls, extra in ((Poly1FocalLoss, ()), (Poly1FocalLoss_U_corr, (conf, 0.4))):
        crit = cls(alpha=alpha, gamma=gamma, epsilon=1.0)
        res = []
        for device, dt in ((torch.device("cpu"), torch.float64), (dev, torch.float32)):       # CPU: the torch composition
            x = logits0.to(device=device, dtype=dt).requires_grad_(True)
            args = [a.to(device) if torch.is_tensor(a) else a for a in extra]
            loss = crit(x, labels.to(device), *args)
            (loss * 3.0).backward()
            res.append((loss.detach().double().cpu(), x.grad.double().cpu()))
        assert abs(float(res[0][0] - res[1][0])) <= 2e-6 * abs(float(res[0][0])) + 1e-9
        scale = float(res[0][1].abs().max())
        assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-5 * scale


def test_thin_mm_weight_gradient():
    from geot_amd.fused_norm import thin_mm
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    for c, j, l in ((128, 3, 131072), (37, 8, 5001), (5, 1, 7)):
        w0, x0, up = torch.randn(c, j, device=dev), torch.randn(j, l, device=dev), torch.randn(c, l, device=dev)
        w = w0.clone().requires_grad_(True)
        y = thin_mm(w, x0)
        (y * up).sum().backward()
        assert torch.equal(y.detach(), torch.mm(w0, x0))
        want = up.double() @ x0.double().t()
        assert float((w.grad.double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_lean_linear_equals_nn_linear():
    from geot_amd.fused_norm import linear
    dev = torch.device("cuda:0")
    torch.manual_seed(6)
    for shape, cout in (((8, 512, 384), 1536), ((3, 7, 50), 33), ((4096, 1536), 384)):
        lin = torch.nn.Linear(shape[-1], cout).to(dev)
        x0, up = torch.randn(*shape, device=dev), torch.randn(*shape[:-1], cout, device=dev)
        res = []
        for fused in (False, True):
            lin.zero_grad()
            x = x0.clone().requires_grad_(True)
            y = linear(lin, x) if fused else lin(x)
            (y * up).sum().backward()
            res.append((y.detach(), x.grad, lin.weight.grad.clone(), lin.bias.grad.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
        want = up.double().reshape(-1, cout).sum(0)
        assert float((res[1][3].double() - want).abs().max()) <= 2e-5 * float(want.abs().max())


def test_res_ln_equals_add_then_layernorm():
    """fused_norm.res_ln (t = x + s*y + extra, z = LayerNorm(t)) vs the torch composition in fp64: both outputs, every
    gradient with both outputs used downstream; all optional inputs present / absent."""
    from geot_amd.fused_norm import res_ln
    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    b, n, c = 3, 50, 384
    x0, y0, e0 = (torch.randn(b, n, c, device=dev) for _ in range(3))
    s0 = (torch.rand(b, 1, 1, device=dev) > 0.3).float() / 0.7
    up_t, up_z = torch.randn(b, n, c, device=dev), torch.randn(b, n, c, device=dev)
    for use_y, use_s, use_e in ((True, True, True), (True, False, False), (False, False, True), (False, False, False)):
        res = []
        for fused in (False, True):
            dt = torch.float32 if fused else torch.float64
            ln = torch.nn.LayerNorm(c).to(dev).to(dt)
            with torch.no_grad():
                ln.weight.copy_(torch.linspace(0.5, 1.5, c)); ln.bias.copy_(torch.linspace(-1, 1, c))
            x = x0.detach().clone().to(dt).requires_grad_(True)
            y = y0.detach().clone().to(dt).requires_grad_(True) if use_y else None
            e = e0.detach().clone().to(dt).requires_grad_(True) if use_e else None
            s = s0.to(dt) if use_s else None
            if fused:
                t, z = res_ln(x, y, s, e, ln)
            else:
                t = x
                if y is not None:
                    t = t + (y if s is None else y * s)
                if e is not None:
                    t = t + e
                z = ln(t)
            ((t * up_t.to(dt)).sum() + (z * up_z.to(dt)).sum()).backward()
            res.append([v.double() for v in (t.detach(), z.detach(), x.grad, ln.weight.grad, ln.bias.grad)]
                       + ([y.grad.double()] if use_y else []) + ([e.grad.double()] if use_e else []))
        for a, f in zip(*res):
            assert float((a - f).abs().max()) <= 2e-5 * (float(a.abs().max()) + 1e-9), (use_y, use_s, use_e)


def test_qkv_split_and_its_gradient():
    from geot_amd.fused_norm import qkv_split
    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    b, n, h, d, scale = 3, 50, 4, 24, 0.37
    x0 = torch.randn(b, n, 3 * h * d, device=dev)
    ups = [torch.randn(b * h, n, d, device=dev) for _ in range(3)]
    res = []
    for fused in (False, True):
        x = x0.clone().requires_grad_(True)
        if fused:
            q, k, v = qkv_split(x, h, scale)
        else:
            q, k, v = x.view(b, n, 3, h, d).permute(2, 0, 3, 1, 4).contiguous().view(3, b * h, n, d).unbind(0)
            q = q * scale
        ((q * ups[0]).sum() + (k * ups[1]).sum() + (v * ups[2]).sum()).backward()
        res.append((q.detach(), k.detach(), v.detach(), x.grad))
    for a, f in zip(*res):
        assert torch.equal(a, f)
    x = x0.clone().requires_grad_(True)
    q, k, v = qkv_split(x, h, scale)
    (k * ups[1]).sum().backward()                               # q and v unused: their gradients are zeros
    want = torch.zeros(b, n, 3, h, d, device=dev)
    want[:, :, 1] = ups[1].view(b, h, n, d).permute(0, 2, 1, 3)
    assert torch.equal(x.grad, want.view(b, n, -1))


def test_softmax_last_gradient():
    from geot_amd.fused_norm import softmax_last
    dev = torch.device("cuda:0")
    torch.manual_seed(10)
    for shape in ((6, 70, 512), (5, 64), (3, 2, 1024)):
        x0, up = torch.randn(*shape, device=dev) * 3, torch.randn(*shape, device=dev)
        res = []
        for fused in (False, True):
            x = (x0.double() if not fused else x0.clone()).requires_grad_(True)
            y = softmax_last(x) if fused else x.softmax(dim=-1)
            (y * up.to(y.dtype)).sum().backward()
            res.append((y.detach().double(), x.grad.double()))
        assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-6
        assert float((res[0][1] - res[1][1]).abs().max()) <= 2e-6 * (float(res[0][1].abs().max()) + 1e-9) + 1e-7


def test_bn_relu_max_equals_batchnorm_relu_maxpool():
    """fused_norm.bn_relu_max (monotone form: relu(a sel + b), sel = max or min by the sign of a) vs BatchNorm -> ReLU ->
    max over the last n in fp64: output, dx, d gamma, d beta, running statistics; both signs of gamma, zero gamma."""
    from geot_amd.fused_norm import bn_relu_max
    dev = torch.device("cuda:0")
    b, c, g, n = 3, 24, 77, 32
    gen = torch.Generator().manual_seed(12)
    y0 = (torch.randn(b, c, g * n, generator=gen) * 2 + 0.3).to(dev)
    up = torch.randn(b, c, g, generator=gen).to(dev)
    ref, ours = torch.nn.BatchNorm1d(c).to(dev).double(), torch.nn.BatchNorm1d(c).to(dev)
    with torch.no_grad():
        w = torch.randn(c, generator=gen); w[3] = 0.0
        ours.weight.copy_(w); ours.bias.copy_(torch.randn(c, generator=gen))
    ref.load_state_dict(ours.state_dict())
    res = []
    for mod, fused in ((ref, False), (ours, True)):
        y = (y0.double() if not fused else y0.clone()).requires_grad_(True)
        out = bn_relu_max(mod, y, n) if fused else torch.relu(mod(y)).view(b, c, g, n).max(-1)[0]
        (out * up.to(out.dtype)).sum().backward()
        res.append([t.double() for t in (out.detach(), y.grad, mod.weight.grad, mod.bias.grad, mod.running_mean, mod.running_var)])
    for name, a, f in zip(("out", "dy", "dgamma", "dbeta", "running_mean", "running_var"), *res):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - f).abs().max()) <= 3e-5 * scale + 1e-6, (name, float((a - f).abs().max()), scale)


def test_fp_front_equals_interpolate_plus_skip_conv():
    """fused_norm.fp_front (interpolation + skip 1x1 conv + BatchNorm sums) vs three_interpolate + bmm, fwd and bwd."""
    from geot_amd.fused_norm import fp_front
    from geot_amd.pointnet2 import pointnet2_utils as pu
    dev = torch.device("cuda:0")
    _, pos, _ = _batch(2, 6000, dev)
    known = pos[:, :1500].contiguous()
    dist2, idx = pu._ext.three_nn(pos, known)
    weight = pu._ext.fp_weights(dist2)
    torch.manual_seed(0)
    a0 = torch.randn(2, 70, 1500, device=dev)
    skip = torch.randn(2, 5, 6000, device=dev)
    wb0 = torch.randn(70, 5, device=dev)
    up = torch.randn(2, 70, 6000, device=dev)
    res = []
    for fused in (False, True):
        a, wb = a0.clone().requires_grad_(True), wb0.clone().requires_grad_(True)
        if fused:
            y, partial = fp_front(a, idx, weight, skip, wb)
        else:
            y = pu.three_interpolate(a, idx, weight) + torch.bmm(wb.unsqueeze(0).expand(2, -1, -1), skip)
        (y * up).sum().backward()
        res.append((y.detach(), a.grad, wb.grad))
    for name, x, f in zip(("y", "dA", "dWb"), *res):
        assert float((x - f).abs().max()) <= 2e-5 * float(x.abs().max()), name
    # the statistics records (s1, s2, pivot, count) per (cloud, channel, slice): sum y = s1 + n p, sum y^2 = s2 + 2 p s1 + n p^2
    rec = partial.double()
    s1, s2, p, cnt = rec[..., 0], rec[..., 1], rec[..., 2], rec[..., 3]
    assert float(cnt.sum((0, 2)).min()) == float(cnt.sum((0, 2)).max()) == 2 * 6000
    sum_abs = res[1][0].double().abs().sum((0, 2))
    assert float((((s1 + cnt * p).sum((0, 2)) - res[1][0].double().sum((0, 2))).abs() / sum_abs).max()) <= 2e-6
    assert torch.allclose((s2 + 2 * p * s1 + cnt * p * p).sum((0, 2)), res[1][0].double().square().sum((0, 2)), rtol=1e-5)


def _sync_bn_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0")
    import torch.distributed as dist
    from geot_amd.fused_norm import bn_act
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    x_all = (torch.randn(4, 12, 3000, generator=g) * 1.5 + 0.3)
    up_all = torch.randn(4, 12, 3000, generator=g)
    lo, hi = (0, 1) if rank == 0 else (1, 4)                 # uneven shards: the element counts differ per rank
    res = []
    for fused in (False, True):
        torch.manual_seed(1)
        bn = torch.nn.SyncBatchNorm(12).to(dev)
        with torch.no_grad():
            bn.weight.uniform_(-1, 1); bn.bias.uniform_(-1, 1)
        x = x_all[lo:hi].to(dev).requires_grad_(True)
        y = bn_act(bn, x, relu=True) if fused else torch.relu(bn(x))
        (y * up_all[lo:hi].to(dev)).sum().backward()
        res.append([t.detach().cpu() for t in (y, x.grad, bn.weight.grad, bn.bias.grad, bn.running_mean, bn.running_var)])
    ok = all(torch.allclose(a, b, rtol=2e-4, atol=2e-4 * float(a.abs().max()) + 1e-6) for a, b in zip(*res))
    q.put((rank, ok, [float((a - b).abs().max()) for a, b in zip(*res)]))
    dist.barrier()
    dist.destroy_process_group()


def test_bn_act_under_sync_batchnorm_two_ranks():
    """fused_norm.bn_act with an nn.SyncBatchNorm module, 2 ranks sharing the GPU over gloo, uneven shards: output,
    gradients and running statistics equal torch's own SyncBatchNorm + ReLU."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sync_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in out), out


def test_full_config_factored_tracks_reference_order_over_training_steps():
    """The headline path at the CONFIGURED sizes (trans_dim 384, depth 12, 512 x 32 groups, targets 8192/4096/2048,
    24 000 points; 2 clouds to keep it short): three optimiser steps in the default `factored` mode and in the
    reference's op order from the same initialisation -- the loss sequences agree to fp32 noise, i.e. the fused /
    reordered kernels compute the reference's function, not something cheaper."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
    from geot_amd.train_step import SupervisedStep
    dev = torch.device("cuda:0")
    _, pos, target = _batch(2, 24000, dev)
    cls = torch.tensor([[0], [1]], device=dev)
    cfg = dict(TOOTH_SEG_CFG, drop_path_rate=0.0)
    losses = {}
    torch.manual_seed(5)
    init = PointTransformer_seg_T(**cfg).state_dict()
    for mode in ("reference", "factored"):
        model = PointTransformer_seg_T(**cfg, dense=mode).to(dev)
        model.load_state_dict(init)
        model.seg_head[2].p = 0.0
        step = SupervisedStep(model, lr=1e-3)
        losses[mode] = [float(step(pos, cls, target)) for _ in range(3)]
    for a, b in zip(losses["reference"], losses["factored"]):
        assert abs(a - b) <= 2e-3 * abs(a), losses
    assert losses["factored"][-1] < losses["factored"][0]


@pytest.mark.parametrize("shape", [(256, 4096, 32), (3, 5, 7, 16), (10, 64), (1000, 8), (33, 4), (2, 128, 500, 24)])
def test_max_last_equals_torch_max(shape):
    from geot_amd.fused_norm import max_last
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(sum(shape))
    x0 = torch.randn(*shape, generator=g)
    x0[..., 1] = x0[..., 0]                       # exact ties: the first maximum must win, as in torch
    x0 = x0.to(dev)
    up = torch.randn(*shape[:-1], generator=g).to(dev)
    res = []
    for fused in (False, True):
        x = x0.clone().requires_grad_(True)
        y = max_last(x) if fused else x.max(dim=-1)[0]
        (y * up).sum().backward()
        res.append((y.detach(), x.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_poly1_loss_contract_on_bad_labels_and_soft_masks(monkeypatch):
    """The reference's F.one_hot raises on a label outside [0, C); its mask is multiplied in by VALUE.  The fused kernel
    cannot raise: it returns NaN (never a silently different loss), GEOT_CHECK_LABELS=1 restores the exception, and a
    float mask takes the op-by-op path."""
    from geot_amd.openpoints.loss import Poly1FocalLoss, Poly1FocalLoss_U_corr
    DEV = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    logits = torch.randn(2, 17, 300, generator=g).to(DEV)
    labels = torch.randint(0, 17, (2, 300), generator=g).to(DEV)
    conf = torch.rand(2, 300, generator=g).to(DEV)
    assert torch.isfinite(Poly1FocalLoss()(logits, labels))
    for bad_value in (-1, 17, 255):
        bad = labels.clone()
        bad[1, 7] = bad_value
        assert torch.isnan(Poly1FocalLoss()(logits, bad))
        assert torch.isnan(Poly1FocalLoss_U_corr()(logits, bad, conf, thresh=0.0))
    monkeypatch.setenv("GEOT_CHECK_LABELS", "1")
    bad = labels.clone()
    bad[0, 0] = 17
    with pytest.raises(RuntimeError, match="smaller than num_classes"):
        Poly1FocalLoss()(logits, bad)
    bad[0, 0] = -1
    with pytest.raises(RuntimeError, match="non-negative"):
        Poly1FocalLoss_U_corr()(logits, bad, conf)
    monkeypatch.delenv("GEOT_CHECK_LABELS")
    soft = torch.rand(2, 300, generator=g).to(DEV)
    got = Poly1FocalLoss_U_corr()(logits, labels, conf, mask=soft)
    crit = Poly1FocalLoss_U_corr()
    want = Poly1FocalLoss_U_corr.forward(crit, logits.double(), labels, conf.double(), mask=soft.double())
    assert abs(got.item() - want.item()) <= 1e-5 * abs(want.item())
    hard = soft > 0.5
    assert abs(Poly1FocalLoss_U_corr()(logits, labels, conf, mask=hard).item()
               - Poly1FocalLoss_U_corr.forward(crit, logits.double(), labels, conf.double(), mask=hard).item()) <= 1e-5


def test_side_stream_index_plan_changes_nothing_but_the_schedule(monkeypatch):
    """The decoder's coordinate-only work (sampled clouds, three_nn + weights of the FP modules, the EdgeConv kNN graphs)
    runs on the side stream behind the long FPS (PointTransformer_seg_T._index_plan): same calls, same results -- logits
    and every gradient bit-identical to the in-line schedule; also under hipGraph-free repeated calls."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    dev = torch.device("cuda:0")
    _, pos, target = _batch(2, 6000, dev)
    cls = torch.tensor([[0], [1]], device=dev)
    torch.manual_seed(0)
    init = PointTransformer_seg_T(**SMALL).state_dict()
    res = {}
    for mode in ("side", "off", "side"):
        monkeypatch.setenv("GEOT_INDEX_PLAN", mode)
        m = PointTransformer_seg_T(**SMALL).to(dev).train()
        m.load_state_dict(init)
        m.seg_head[2].p = 0.0
        outs = []
        for _ in range(2):                       # twice: the second forward re-uses the side stream right after a backward
            m.zero_grad(set_to_none=True)
            logit = m(pos, pos.transpose(1, 2).contiguous(), cls, torch.eye(17, device=dev))[0]
            torch.nn.functional.cross_entropy(logit, target).backward()
            outs.append((logit.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}))
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(outs)
    a, b = res["side"][0], res["off"][0]
    for (la, ga), (lb, gb) in zip(a, b):
        assert torch.equal(la, lb)
        assert set(ga) == set(gb) and all(torch.equal(ga[k], gb[k]) for k in ga)
    for (la, ga), (lc, gc) in zip(a, res["side"][1]):
        assert torch.equal(la, lc) and all(torch.equal(ga[k], gc[k]) for k in ga)


def test_look_ahead_geometry_changes_nothing_but_the_schedule():
    """SupervisedStep(next_pos=...) / PointTransformer_seg_T.prefetch_geometry: the next batch's Group (FPS + kNN), 8192-sample
    FPS and index plan queued beside the current batch's backward -- a run of alternating batches gives bit-identical losses
    and parameters to the same run without look-ahead; a geometry of ANOTHER tensor, or one the tensor was edited after,
    is ignored."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd import train_step as ts
    dev = torch.device("cuda:0")
    from geot_amd.synth import make_batch, region_labels
    _, pos_a, tgt_a = _batch(2, 6000, dev)
    xyz_b = make_batch(2, 6000, start_index=40)[0]
    pos_b, tgt_b = torch.from_numpy(xyz_b).to(dev), torch.from_numpy(region_labels(xyz_b)).to(dev)
    cls = torch.tensor([[0], [1]], device=dev)
    torch.manual_seed(0)
    init = PointTransformer_seg_T(**SMALL).state_dict()
    batches = [(pos_a, cls, tgt_a), (pos_b, cls.flip(0).contiguous(), tgt_b)]
    runs = {}
    for look in (False, True):
        m = PointTransformer_seg_T(**SMALL).to(dev)
        m.load_state_dict(init)
        m.seg_head[2].p = 0.0
        step = ts.SupervisedStep(m)
        losses = []
        for i in range(4):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            losses.append(step(cur[0], cur[1], cur[2], next_pos=nxt[0] if look else None).clone())
            assert (step._geometry is not None) == look
        torch.cuda.synchronize()
        runs[look] = (losses, {k: v.detach().clone() for k, v in m.state_dict().items()})
    for a, b in zip(runs[False][0], runs[True][0]):
        assert torch.equal(a, b)
    for k, v in runs[False][1].items():
        assert torch.equal(v, runs[True][1][k]), k
    # a geometry that does not belong to the tensor passed in is not used
    m = PointTransformer_seg_T(**SMALL).to(dev).train()
    m.load_state_dict(init)
    m.seg_head[2].p = 0.0
    x = pos_a.transpose(1, 2).contiguous()
    want = m(pos_a, x, cls)[0]
    stale = m.prefetch_geometry(pos_b)
    assert torch.equal(m(pos_a, x, cls, geometry=stale)[0], want)
    edited = pos_a.clone()
    g = m.prefetch_geometry(edited)
    edited.mul_(1.0)                                      # version bump: the geometry no longer describes this tensor's history
    assert torch.equal(m(edited, x, cls, geometry=g)[0], want)
    assert torch.equal(m(pos_a, x, cls, geometry=m.prefetch_geometry(pos_a))[0], want)


def test_fixmatch_look_ahead_changes_nothing_but_the_schedule():
    """FixMatchNTMStep(next_batches=...): the next iteration's student batch (labelled + strong + weak views, concatenated)
    and teacher batch (weak view) have their sampling / grouping / index plan queued behind this iteration's student
    forward.  WholePartSeg with and without the prefetched geometry: bit-identical logits (student in training mode,
    teacher in eval mode); three alternating iterations: the first identical, the later ones within the run-to-run
    spread of the step (float atomics in the graph loss -> AdamW, see tests/test_fullsize_gpu.py); batches other than the
    announced ones are computed in line."""
    from geot_amd import train_step as ts
    from geot_amd.synth import make_batch, region_labels
    dev = torch.device("cuda:0")
    cfg = dict(ts.NTM_CFG, threed_k=8)

    def batch(seed):
        xl, xu = make_batch(2, 4096, start_index=seed)[0], make_batch(2, 4096, start_index=seed + 50)[0]
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
        lab, unl, strong = T(xl), T(xu), T(xu * np.float32(1.04))
        z = torch.zeros(2, 1, dtype=torch.long, device=dev)
        return ({"pos": lab, "x": lab.transpose(1, 2).contiguous(), "cls": z, "y": T(region_labels(xl))},
                {"pos_w": unl, "x_w": unl.transpose(1, 2).contiguous(), "cls_w": z, "pos_s": strong,
                 "x_s": strong.transpose(1, 2).contiguous(), "cls_s": z, "raw_pos": unl})
    batches = [batch(3), batch(400)]
    runs = {}
    for look in (False, True):
        torch.manual_seed(5)
        step = ts.build_fixmatch(dev, seg_cfg=SMALL, cfg=cfg, use_ddp=False)
        step.model.segmentor.seg_head[2].p = 0.0
        out = []
        for i in range(3):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            out.append({k: float(v) for k, v in step(cur[0], cur[1], next_batches=nxt if look else None).items()})
            assert (step._geometry[0] is not None) == look and (step._geometry[1] is not None) == look
        runs[look] = out
        if look:
            # forward with / without the queued geometry: the same logits, bit for bit
            d, u = batches[1]
            g_s, g_t = step._geometry
            with torch.no_grad():
                step.model.train()
                a = step.model(d, u0=dict(u, T=step.ema_t), fixmatch=True, geometry=g_s)[0]
                b = step.model(d, u0=dict(u, T=step.ema_t), fixmatch=True)[0]
                step.model_t.eval()
                # the teacher's geometry is the weak view's slice of the student's (slice_geometry), and the teacher takes it:
                # no Group, no index plan of its own
                assert g_t["pts"].data_ptr() == g_s["pts"][4:].data_ptr() and g_t["training"] is False
                seg_t, calls = step.model_t.segmentor, []
                plan0, group0 = seg_t._index_plan, seg_t.group_divider.forward
                seg_t._index_plan = lambda *a, **k: (calls.append("plan"), plan0(*a, **k))[1]
                seg_t.group_divider.forward = lambda *a, **k: (calls.append("group"), group0(*a, **k))[1]
                ta = step.model_t(u, if_teacher=True, geometry=g_t)[0]
                assert calls == [], calls
                tb = step.model_t(u, if_teacher=True)[0]
                assert sorted(calls) == ["group", "plan"]
                seg_t._index_plan, seg_t.group_divider.forward = plan0, group0
            assert torch.equal(ta, tb)
            assert torch.equal(a, b)
            # an iteration on batches that were NOT announced ignores the geometry
            other = batch(900)
            loss = step(other[0], other[1])
            assert all(bool(torch.isfinite(v)) for v in loss.values())
    for k in runs[False][0]:
        assert abs(runs[False][0][k] - runs[True][0][k]) <= 1e-6 * abs(runs[False][0][k]) + 1e-7, (k, runs)
    for i in (1, 2):
        for k in runs[False][i]:
            assert abs(runs[False][i][k] - runs[True][i][k]) <= 3e-4 * abs(runs[False][i][k]) + 1e-6, (i, k, runs)


def test_wholepartseg_geometry_is_keyed_to_its_source_tensors():
    """A queued geometry describes the tensors it was computed from, not a shape: WholePartSeg takes it for
    exactly those tensors, unedited -- another batch of the SAME shape, or the same tensor after an in-place edit, is computed
    in line (same logits as without any geometry)."""
    from geot_amd.openpoints.models.segmentation import WholePartSeg
    from geot_amd.synth import make_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = WholePartSeg(segmentor_args=dict(NAME="PointTransformer_seg_T", **SMALL)).to(dev).train()
    model.segmentor.seg_head[2].p = 0.0
    pos_a = torch.from_numpy(make_batch(2, 6000, start_index=0)[0]).to(dev)
    pos_b = torch.from_numpy(make_batch(2, 6000, start_index=70)[0]).to(dev)
    cls = torch.zeros(2, 1, dtype=torch.long, device=dev)
    with torch.no_grad():
        want_b = model(pos_b, pos_b.transpose(1, 2).contiguous(), cls)[0]
        g_a = model.prefetch_geometry(pos_a)
        assert g_a["src"][0][0] is pos_a
        got = model(pos_b, pos_b.transpose(1, 2).contiguous(), cls, geometry=g_a)[0]      # same shape, another batch
        assert torch.equal(got, want_b)
        g_b = model.prefetch_geometry(pos_b)
        assert torch.equal(model(pos_b, pos_b.transpose(1, 2).contiguous(), cls, geometry=g_b)[0], want_b)
        edited = pos_b.clone()
        g_e = model.prefetch_geometry(edited)
        edited.add_(0.0)                                  # version bump: the geometry no longer describes this tensor's history
        assert torch.equal(model(edited, edited.transpose(1, 2).contiguous(), cls, geometry=g_e)[0], want_b)
