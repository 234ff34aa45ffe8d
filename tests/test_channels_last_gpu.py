"""GPU suite: the point-major (B, N, C) form of the PointnetFPModule front end (csrc/channels_last.hip,
gather_group.hip gather_rows_csr_cl_kernel; pointnet2/pointnet2_modules.py:619-640 and interpolate_gpu.cu:88-146 are
the behaviour) against the channels-first kernels, which the oracle tests pin, and against float64.

* forward values: bit-identical to the channels-first kernel (same per-element arithmetic), with and without a row
  sequence, ragged channel counts, no skip tensor
* BatchNorm sums / passes: float64 referee at 1e-5, running statistics included
* gradient: the reverse-index row gather == the channels-first gradient (same summation order: bit-identical where the
  channels-first path is the reverse-index one), float64 referee, hub targets, empty targets, bit-reproducible
* the FP module and the whole model in both layouts
"""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _cloud(b, n, seed=0):
    from geot_amd.synth import make_batch
    return torch.from_numpy(make_batch(b, n, start_index=seed)[0]).to(DEV)


def _nn3(unknown, known):
    from geot_amd.pointnet2 import pointnet2_utils as pu
    d2, idx = pu._ext.three_nn(unknown.contiguous(), known.contiguous())
    return idx, pu._ext.fp_weights(d2)


def rel(got, want):
    want = want.detach().double().cpu()
    return float((got.detach().double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("b,n,m,c,cs", [(2, 6000, 1500, 256, 5), (3, 4099, 517, 260, 0), (1, 2048, 64, 1536, 3), (2, 777, 3, 512, 8)])
@pytest.mark.parametrize("ordered", [False, True])
def test_fp_front_cl_equals_the_channels_first_kernel(b, n, m, c, cs, ordered):
    from geot_amd import fused_norm as fn
    pos = _cloud(b, n, 1)
    known = pos[:, :m].contiguous()
    idx, w = _nn3(pos, known)
    torch.manual_seed(0)
    a = torch.randn(b, c, m, device=DEV)
    skip = torch.randn(b, cs, n, device=DEV) if cs else None
    wb = torch.randn(c, cs, device=DEV) if cs else None
    order = fn.local_spatial_order(pos) if ordered else None
    if order is not None:                                    # a permutation of every cloud's points
        assert torch.equal(order.long().sort(1)[0], torch.arange(n, device=DEV).expand(b, -1))
    y_cf, _ = fn.fp_front(a, idx, w, skip, wb)
    y_cl, partial = fn.fp_front_cl(a.transpose(1, 2).contiguous(), idx, w, skip, wb, order)
    assert y_cl.shape == (b, n, c)
    assert torch.equal(y_cl.transpose(1, 2), y_cf)
    # the statistics records that go to the BatchNorm: (tiles, 3, c) = (s1, s2, pivot) + tiles counts -> sum y, sum y^2
    from geot_amd import _lib
    from geot_amd.ext._common import call, ptr
    tiles = partial.numel() // (3 * c + 1)
    assert tiles == _lib.load().geot_fp_front_cl_tiles(b, c, n, cs)
    assert float(partial[tiles * 3 * c:].sum()) == b * n                     # every row counted once
    sums = torch.empty(c, 2, dtype=torch.float64, device=DEV)
    call("geot_bn_sums_shifted_cl", DEV, tiles, c, ptr(partial), ptr(sums))
    want = torch.stack([y_cf.double().sum((0, 2)), (y_cf.double() ** 2).sum((0, 2))], 1)
    scale = torch.stack([y_cf.double().abs().sum((0, 2)), want[:, 1]], 1)
    assert float(((sums - want).abs() / scale).max()) <= 1e-6


def test_fp_front_cl_rejects_what_it_does_not_cover():
    from geot_amd import _lib
    lib = _lib.load()
    assert lib.geot_fp_front_cl_tiles(2, 1538, 100, 3) == -1          # C % 4
    assert lib.geot_fp_front_cl_tiles(2, 4100, 100, 3) == -1          # C > 4096
    assert lib.geot_fp_front_cl_tiles(2, 512, 100, 9) == -1           # more than 8 skip channels
    assert lib.geot_cl_tiles(1, 100, 6) == -1
    assert lib.geot_fp_front_cl_tiles(2, 512, 100, 3) >= 1


def _grad_case(b, n, m, c, adversarial):
    pos = _cloud(b, n, 2)
    known = pos[:, :m].contiguous()
    idx, w = _nn3(pos, known)
    if adversarial:                                          # a hub (every third pair on target 0) and untouched targets
        idx = idx.clone()
        idx[:, ::3, 0] = 0
        idx[idx == m - 1] = 1
    torch.manual_seed(1)
    gy = torch.randn(b, c, n, device=DEV)
    return pos, known, idx.contiguous(), w, gy


@pytest.mark.parametrize("b,n,m,c", [(2, 6000, 1500, 256), (3, 4099, 517, 260), (1, 3000, 8, 1536)])
@pytest.mark.parametrize("adversarial", [False, True])
@pytest.mark.parametrize("ordered", [False, True])
def test_reverse_index_row_gather_is_the_interpolation_gradient(b, n, m, c, adversarial, ordered):
    from geot_amd import fused_norm as fn
    from geot_amd.ext import pointnet2_ext as p2
    pos, known, idx, w, gy = _grad_case(b, n, m, c, adversarial)
    want64 = torch.zeros(b, c, m, dtype=torch.float64)
    src = (gy.double().cpu().unsqueeze(-1) * w.double().cpu().unsqueeze(1)).reshape(b, c, n * 3)
    want64.scatter_add_(2, idx.long().cpu().reshape(b, 1, n * 3).expand(-1, c, -1), src)
    order = fn.local_spatial_order(known) if ordered else None
    gy_cl = gy.transpose(1, 2).contiguous()
    rix = fn.ReverseIndex(idx, w, m, order)
    got = rix.gather(gy_cl)
    assert got.shape == (b, m, c)
    assert rel(got.transpose(1, 2), want64) <= 1e-5 * (30 if adversarial else 1)      # a hub sums 2000 terms in fp32
    # the same index rebuilt, the same gather again: bit-identical (one writer per row, fixed summation order)
    again = fn.ReverseIndex(idx, w, m, order).gather(gy_cl)
    assert torch.equal(got, again)
    # untouched targets are written (zeros), not left as they were
    if adversarial:
        assert float(got[:, m - 1].abs().max()) == 0.0
    # and the channels-first gradient agrees
    cf = p2.three_interpolate_grad(gy, idx, w, m)
    # (two fp32 sums of 3 n / m terms per target in two different -- but now both FIXED -- orders: the channels-first gradient
    # has one writer per element at every shape since round 5, csrc/tile_scatter.hip; it is also bit-identical call to call)
    assert rel(got.transpose(1, 2), cf) <= 2e-6 * (30 if adversarial else 1)
    assert torch.equal(p2.three_interpolate_grad(gy, idx, w, m), cf)


def test_reverse_index_without_weights_and_single_slot():
    """nt = 1, unit weights: the gradient of a plain row gather (group / gather_operation in point-major form)."""
    from geot_amd import _lib
    from geot_amd.ext._common import call, ptr
    b, n, m, c = 2, 5000, 700, 128
    torch.manual_seed(4)
    idx = torch.randint(0, m, (b, n, 1), device=DEV, dtype=torch.int32)
    g = torch.randn(b, n, c, device=DEV)
    lib = _lib.load()
    ints = int(lib.geot_rix_ws_ints(b, n, m, 1))
    ws = torch.empty(ints, dtype=torch.int32, device=DEV)
    call("geot_rix_build", DEV, b, n, m, 1, ptr(idx), None, None, ptr(ws), ints)
    out = torch.full((b, m, c), float("nan"), device=DEV)
    call("geot_gather_rows_csr_cl", DEV, b, c, n, m, 1, ptr(g), ptr(ws), None, ptr(out))
    want = torch.zeros(b, m, c, dtype=torch.float64)
    want.scatter_add_(1, idx.long().cpu().expand(-1, -1, c), g.double().cpu())
    assert rel(out, want) <= 1e-6
    # too small a workspace is refused, not overrun
    with pytest.raises(RuntimeError):
        call("geot_rix_build", DEV, b, n, m, 1, ptr(idx), None, None, ptr(ws), ints - 1)


@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_cl_against_float64(relu, training):
    from geot_amd import fused_norm as fn
    b, l, c = 3, 2777, 260
    torch.manual_seed(5)
    x = (torch.randn(b, l, c, device=DEV) * 2 + 0.5)
    up = torch.randn(b, l, c, device=DEV)
    bn = torch.nn.BatchNorm1d(c).to(DEV)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5)
        bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    bn.train(training)
    ref = copy.deepcopy(bn).double().cpu()
    x64 = x.double().cpu().transpose(1, 2).contiguous().requires_grad_(True)          # (B, C, L) for the module
    y64 = ref(x64)
    y64 = torch.relu(y64) if relu else y64
    (y64 * up.double().cpu().transpose(1, 2)).sum().backward()
    xg = x.clone().requires_grad_(True)
    y = fn.bn_act_cl(bn, xg, relu=relu)
    (y * up).sum().backward()
    assert rel(y.transpose(1, 2), y64) <= 1e-5
    assert rel(xg.grad.transpose(1, 2), x64.grad) <= 2e-5
    assert rel(bn.weight.grad, ref.weight.grad) <= 2e-5 and rel(bn.bias.grad, ref.bias.grad) <= 2e-5
    assert rel(bn.running_mean, ref.running_mean) <= 1e-5 and rel(bn.running_var, ref.running_var) <= 1e-5
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)


@pytest.mark.parametrize("layout", ["cf", "cl"])
def test_batch_statistics_far_from_zero(layout):
    """|mean| = 1000 std (un-centred coordinates as features): the shifted sums keep the variance; plain fp32 sums of x
    and x^2 would lose all of its digits.  BatchNorm output and running statistics vs float64."""
    from geot_amd import fused_norm as fn
    b, l, c = 2, 3000, 260
    torch.manual_seed(7)
    x = torch.randn(b, c, l, device=DEV) * torch.linspace(0.5, 2.0, c, device=DEV).view(1, c, 1) + 1000.0
    bn = torch.nn.BatchNorm1d(c).to(DEV).train()
    ref = copy.deepcopy(bn).double().cpu()
    y64 = ref(x.double().cpu())
    if layout == "cf":
        y = fn.bn_act(bn, x, relu=False)
    else:
        y = fn.bn_act_cl(bn, x.transpose(1, 2).contiguous(), relu=False).transpose(1, 2)
    assert float((y.double().cpu() - y64).abs().max()) <= 2e-3               # |x| ~ 1e3 in fp32: 6e-5 absolute per element / std 0.5
    assert rel(bn.running_var, ref.running_var) <= 1e-4 and rel(bn.running_mean, ref.running_mean) <= 1e-6
    # and the FP front end's own records
    pos = _cloud(2, 3000, 4)
    known = pos[:, :600].contiguous()
    idx, w = _nn3(pos, known)
    a = torch.randn(2, c, 600, device=DEV) + 500.0
    bn2 = torch.nn.BatchNorm1d(c).to(DEV).train()
    if layout == "cf":
        yy, part = fn.fp_front(a, idx, w, None, None)
        fn.bn_act(bn2, yy, relu=False, partial=part)
        var64 = yy.double().transpose(0, 1).reshape(c, -1).var(1, unbiased=True)
    else:
        yy, part = fn.fp_front_cl(a.transpose(1, 2).contiguous(), idx, w, None, None)
        fn.bn_act_cl(bn2, yy, relu=False, partial=part)
        var64 = yy.double().reshape(-1, c).var(0, unbiased=True)
    want = 0.9 * 1.0 + 0.1 * var64
    assert rel(bn2.running_var, want) <= 1e-4


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("c_known,c_skip,widths", [(48, 5, [256, 32]), (96, 0, [512, 64])])
def test_fp_module_point_major_against_float64(c_known, c_skip, widths, fused, monkeypatch):
    """PointnetFPModule in training mode through _fp_factored(layout="cl") vs the reference op order in fp64 (as
    tests/test_fp64_referee_gpu.py does for the channels-first forms), and vs the channels-first factored form."""
    from geot_amd.pointnet2.pointnet2_modules import PointnetFPModule
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from geot_amd.openpoints.models.backbone.transformer import _fp_factored
    from geot_amd import fused_norm as fn
    monkeypatch.setenv("GEOT_FP_CL_FUSED", fused)        # "1": one node, gy never written; "0": fp_front_cl + bn_act_cl
    pos = _cloud(2, 1500, 3)      # (small: the fewer ReLU inputs, the sooner a seed without a kink)
    known = pos[:, ::5].contiguous()
    d2, idx = pu._ext.three_nn(pos, known)
    dist = torch.sqrt(d2.double().cpu())
    r = 1.0 / (dist + 1e-8)
    w64 = r / r.sum(2, keepdim=True)
    for seed in range(1, 60):
        # a ReLU input within fp32 rounding of zero flips its mask between any two fp32 evaluations and moves a whole
        # gradient element: the referee (fp64) looks for such kinks and the test takes the first seed without one
        torch.manual_seed(seed)
        fp = PointnetFPModule(mlp=[c_known + c_skip] + widths).to(DEV).train()
        kf = torch.randn(2, c_known, known.shape[1], device=DEV)
        sk = torch.randn(2, c_skip, 1500, device=DEV) if c_skip else None
        up = torch.randn(2, widths[-1], 1500, device=DEV)
        fp64 = copy.deepcopy(fp).double().cpu()
        kf64 = kf.double().cpu().requires_grad_(True)
        g = torch.gather(kf64, 2, idx.cpu().long().reshape(2, 1, -1).expand(-1, c_known, -1)).view(2, c_known, 1500, 3)
        x64 = (g * w64.unsqueeze(1)).sum(-1)
        if sk is not None:
            x64 = torch.cat([x64, sk.double().cpu()], 1)
        nearest = []
        hooks = [mod.register_forward_hook(lambda _m, _i, out: nearest.append(float(out.detach().abs().min())))
                 for mod in fp64.modules() if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm)]
        y64 = fp64.mlp(x64.unsqueeze(-1)).squeeze(-1)
        for h in hooks:
            h.remove()
        if min(nearest) > 1e-6:                  # fp32 evaluations of these values differ by ~2e-7
            break
    else:
        pytest.skip("no kink-free seed")
    (y64 * up.double().cpu()).sum().backward()
    p64 = dict(fp64.named_parameters())
    errs = {}
    for mode in ("cf", "cl", "cl+plan"):
        m = copy.deepcopy(fp)
        k = kf.clone().requires_grad_(True)
        nn3 = None
        if mode == "cl+plan":
            weight = pu._ext.fp_weights(d2)
            nn3 = (idx, weight, fn.local_spatial_order(pos), fn.ReverseIndex(idx, weight, known.shape[1], fn.local_spatial_order(known)))
        y = _fp_factored(m, pos, known, sk, k, nn3, layout=mode[:2])
        (y * up).sum().backward()
        errs[mode] = [rel(y, y64), rel(k.grad, kf64.grad)] + [rel(p.grad, p64[n].grad) for n, p in m.named_parameters()]
        for n, buf in m.named_buffers():
            if buf.dtype.is_floating_point:
                assert rel(buf, dict(fp64.named_buffers())[n]) <= 1e-5, (mode, n)
    for mode, e in errs.items():
        assert e[0] <= 1e-5, (mode, "forward", e)
        assert max(e[1:]) <= 5e-5, (mode, "gradients", e)
    assert errs["cl"][0] <= 3 * errs["cf"][0] + 2e-6 and max(errs["cl"][1:]) <= 3 * max(errs["cf"][1:]) + 5e-6, errs


def test_model_step_in_both_fp_layouts():
    """PointTransformer_seg_T (small config, training step): logits, every gradient and every buffer with
    fp_layout = "cl" against "cf" -- the same function, BatchNorm statistics summed in another order."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd.synth import make_batch, region_labels
    cfg = dict(trans_dim=384, depth=3, num_heads=4, group_size=32, num_group=128, encoder_dims=256, nclasses=17,
               drop_path_rate=0.0, downsample_targets=[2048, 1024, 512], extract_layers=[1, 2, 3])
    xyz = make_batch(2, 4096, start_index=5)[0]
    target = torch.from_numpy(region_labels(xyz)).to(DEV)
    cls = torch.tensor([[0], [1]], device=DEV)
    torch.manual_seed(3)
    init = PointTransformer_seg_T(**cfg).state_dict()
    res = {}
    for layout in ("cf", "cl"):
        m = PointTransformer_seg_T(**cfg)
        m.load_state_dict(init)
        m.fp_layout = layout
        m = m.to(DEV).train()
        m.seg_head[2].p = 0.0
        pos = torch.from_numpy(xyz).to(DEV)
        logit = m(pos, pos.transpose(1, 2).contiguous(), cls, torch.eye(17, device=DEV))[0]
        torch.nn.functional.cross_entropy(logit, target).backward()
        res[layout] = (logit.detach(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None},
                       {n: b.detach().clone() for n, b in m.named_buffers()})
        m.eval()                                             # and the eval forward (running statistics, no reverse index)
        with torch.no_grad():
            res[layout] += (m(pos, pos.transpose(1, 2).contiguous(), cls, torch.eye(17, device=DEV))[0],)
    (l0, g0, b0, e0), (l1, g1, b1, e1) = res["cf"], res["cl"]
    assert rel(l1, l0) <= 5e-5 and rel(e1, e0) <= 5e-5
    assert set(g0) == set(g1)
    zero_grad = ("encoder.first_conv.0.bias", "encoder.first_conv.3.bias", "encoder.second_conv.0.bias", "seg_head.0.bias")
    for k in g0:
        if k not in zero_grad:
            err = float((g0[k] - g1[k]).norm() / (g0[k].norm() + 1e-30))
            assert err <= 5e-3, (k, err)                     # fp32 through ~30 layers with max-pools (see test_fp64_referee)
    for k in b0:
        assert float((b0[k].float() - b1[k].float()).abs().max()) <= 1e-5 * max(1.0, float(b0[k].float().abs().max())), k


def test_edgeconv_gradient_with_the_reverse_index_built_ahead():
    """transformer_ops.edgeconv_reverse_index + edgeconv_tail(rix=...): the reverse index of the kNN graph built once, ahead
    (the model's index plan), gives the same gradients bit for bit as the one the gradient call builds for itself."""
    from geot_amd.openpoints.models.backbone.transformer_ops import edgeconv_tail, edgeconv_reverse_index
    from geot_amd.knn_cuda import knn_sorted
    b, c, nq, nk, k = 2, 384, 2048, 512, 4
    pos = _cloud(b, nq, 6)
    src = pos[:, :nk].contiguous()
    _, idx = knn_sorted(pos, src, k)                       # (query, reference) -> (b, nq, k) ids among the nk sources
    idx = idx.int().contiguous()
    assert idx.shape == (b, nq, k) and int(idx.max()) < nk
    torch.manual_seed(0)
    norm = torch.nn.GroupNorm(4, c).to(DEV)
    up = torch.randn(b, c, nq, device=DEV)
    res = []
    for ahead in (False, True):
        p = torch.randn(b, c, nk, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).requires_grad_(True)
        q = torch.randn(b, c, nq, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)).requires_grad_(True)
        norm.zero_grad(set_to_none=True)
        rix = edgeconv_reverse_index(idx, nk) if ahead else None
        out = edgeconv_tail(p, q, idx, norm, 0.2, rix)
        (out * up).sum().backward()
        res.append((out.detach(), p.grad.clone(), q.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone()))
    for x, y in zip(*res):
        assert torch.equal(x, y)


def test_spatial_order_is_the_same_on_every_call():
    """The Morton processing order decides in which order a tile sums its rows (BatchNorm statistics of the point-major FP
    stage), so it must not depend on the arrival order of the counting sort's atomics: inside a grid cell the points go by
    ascending index.  Clouds with many points per cell (duplicates, a cluster) make ties certain."""
    from geot_amd import fused_norm as fn
    from geot_amd.synth import make_batch
    xyz = make_batch(3, 6000, start_index=3, dup_frac=0.3)[0]
    xyz[1, 1000:3000] = xyz[1, 17] + 1e-4 * np.random.default_rng(0).standard_normal((2000, 3)).astype(np.float32)   # a cluster
    pos = torch.from_numpy(xyz).to(DEV)
    orders = [fn.local_spatial_order(pos).cpu().numpy() for _ in range(4)]
    for o in orders[1:]:
        assert np.array_equal(o, orders[0])
    for bi in range(3):
        assert np.array_equal(np.sort(orders[0][bi]), np.arange(6000))
    # equal points share a cell: they appear in ascending index
    o = orders[0][0]
    first = {}
    rank = np.empty(6000, np.int64)
    rank[o] = np.arange(6000)
    keys = [tuple(p) for p in xyz[0]]
    for i, k in enumerate(keys):
        if k in first:
            assert rank[first[k]] < rank[i]
        else:
            first[k] = i


def test_training_steps_are_bit_reproducible():
    """Two trainers built from the same seed, run independently (their own Morton orders, reverse indices, look-ahead): the
    same losses and the same parameters, bit for bit, after three steps -- no float atomics and no arrival-order dependence
    on the supervised path."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd.train_step import SupervisedStep
    from geot_amd.synth import make_batch, region_labels
    cfg = dict(trans_dim=384, depth=3, num_heads=4, group_size=32, num_group=128, encoder_dims=256, nclasses=17,
               drop_path_rate=0.1, downsample_targets=[4096, 2048, 1024], extract_layers=[1, 2, 3])
    xyz, xyz2 = make_batch(2, 12000, start_index=5)[0], make_batch(2, 12000, start_index=9)[0]
    batches = [(torch.from_numpy(x).to(DEV), torch.from_numpy(region_labels(x)).to(DEV)) for x in (xyz, xyz2)]
    cls = torch.zeros(2, 1, dtype=torch.long, device=DEV)
    runs = []
    for _ in range(2):
        torch.manual_seed(4)
        model = PointTransformer_seg_T(**cfg).to(DEV)
        step = SupervisedStep(model, lr=1e-3)
        losses = []
        for it in range(3):
            cur, nxt = batches[it % 2], batches[(it + 1) % 2]
            losses.append(step(cur[0].clone(), cls, cur[1], next_pos=None if it == 2 else nxt[0]).clone())
        runs.append((torch.stack(losses), [p.detach().clone() for p in model.parameters()]))
    assert torch.equal(runs[0][0], runs[1][0]), (runs[0][0], runs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))


def test_fixmatch_iterations_are_bit_reproducible():
    """The FixMatch+NTM iteration as well: two trainers from the same seed, two iterations each -- identical losses, student and
    predictor parameters and EMA matrix.  (Fixed-order reduces of the sig_t_mean / logit-correction partials, the graph loss's
    in-edges sorted by source before they are summed; hubs beyond the 64 in-edge slots would still go through float atomics.)"""
    from geot_amd import train_step as ts
    from geot_amd.synth import make_batch, region_labels
    small = dict(trans_dim=384, depth=3, num_heads=4, group_size=32, num_group=128, encoder_dims=256, nclasses=17,
                 drop_path_rate=0.0, downsample_targets=[2048, 1024, 512], extract_layers=[1, 2, 3])
    xyz = make_batch(2, 6000, start_index=0)[0]
    pos, target = torch.from_numpy(xyz).to(DEV), torch.from_numpy(region_labels(xyz)).to(DEV)
    xu = torch.from_numpy(make_batch(2, 6000, start_index=50)[0]).to(DEV)
    xs = (xu * 1.1).contiguous()
    z = torch.zeros(2, 1, dtype=torch.long, device=DEV)
    data = {"pos": pos, "x": pos.transpose(1, 2).contiguous(), "cls": z, "y": target}
    data_u = {"pos_w": xu, "x_w": xu.transpose(1, 2).contiguous(), "cls_w": z, "pos_s": xs,
              "x_s": xs.transpose(1, 2).contiguous(), "cls_s": z, "raw_pos": xu}
    runs = []
    for _ in range(2):
        torch.manual_seed(2)
        trainer = ts.build_fixmatch(DEV, seg_cfg=small, use_ddp=False)
        out = [trainer(data, data_u) for _ in range(2)]
        runs.append((torch.stack([v for o in out for v in o.values()]),
                     [p.detach().clone() for p in trainer.model.parameters()] + [p.detach().clone() for p in trainer.T_predictor.parameters()]
                     + [trainer.ema_t.clone()]))
    assert torch.equal(runs[0][0], runs[1][0]), (runs[0][0], runs[1][0])
    assert all(torch.equal(a, b) for a, b in zip(runs[0][1], runs[1][1]))


@pytest.mark.parametrize("shape", [(8, 5, 24000), (256, 4096), (3, 17, 1), (1, 1, 100003), (6, 384, 0)])
def test_rowsum_f64_and_the_bias_gradient_it_feeds(shape):
    """geot_rowsum_f64 (one workgroup per row, fp64 accumulation, fixed order) against torch's fp64 sum; add_channel_bias's
    forward is the torch expression bit for bit, its bias gradient the fp64 column sums rounded once -- closer to the fp64
    truth than aten::sum's fp32 tree, and the same bits on every call."""
    from geot_amd.fused_norm import rowsum_f64, add_channel_bias
    torch.manual_seed(3)
    x = torch.randn(*shape, device=DEV) * 3 + 0.5
    if x.numel():
        got = rowsum_f64(x)
        want = x.double().sum(-1)
        assert got.dtype == torch.float64 and got.shape == want.shape
        assert torch.allclose(got, want, rtol=1e-13, atol=1e-10)
        assert torch.equal(got, rowsum_f64(x))
    c = shape[-2]
    bias = torch.randn(c, device=DEV, requires_grad=True)
    y = x.clone().requires_grad_(True)
    out = add_channel_bias(y, bias)
    assert torch.equal(out, y + bias.view(-1, 1))
    g = torch.randn_like(out)
    gy, gb = torch.autograd.grad(out, (y, bias), g)
    assert torch.equal(gy, g)
    lead = tuple(range(g.dim() - 2)) + (g.dim() - 1,)
    assert torch.allclose(gb.double(), g.double().sum(lead), rtol=2e-7, atol=1e-30)


@pytest.mark.parametrize("b,n,m,c,kind", [
    (2, 24000, 8192, 64, "nn"),        # the model's first FP stage (8.8 pairs per target)
    (1, 6000, 1500, 256, "nn"),        # 12 pairs per target
    (2, 8192, 512, 64, "nn"),          # the second FP stage: 48 pairs per target -> the few-target form (sources stream once)
    (1, 4099, 300, 96, "nn"),          # the same form: a last chunk of 3 rows, fewer targets than slots
    (2, 1000, 100, 32, "random"),      # random ids, 30 pairs per target
    (1, 1500, 96, 32, "hub"),          # few targets, empty lists, one target with a third of all pairs
    (2, 3000, 700, 48, "hub"),         # the list walk with empty lists and a hub
])
@pytest.mark.parametrize("relu", [True, False])
def test_bn_row_gather_against_fp64_in_both_target_orders(b, n, m, c, kind, relu, monkeypatch):
    """geot_gather_rows_csr_bn_cl called directly (the BatchNorm(+ReLU) backward folded into the interpolation gradient):
    against fp64; with / without the Morton order of the targets (the order moves rows between workgroups, no sum: the same
    bits); and the few-target form (<= 512 targets, long lists: sums in registers, sources streamed once) against the list walk
    it replaces there (GEOT_GR_FORM=list) -- the same pairs in the same order through the same fma chain: the same bits."""
    from geot_amd import fused_norm as fn
    from geot_amd.ext import pointnet2_ext as p2
    from geot_amd.ext._common import call, ptr
    torch.manual_seed(n + m)
    pos = _cloud(b, n, 3)
    known = pos[:, torch.randperm(n, device=DEV)[:m]].contiguous()
    if kind == "nn":
        d2, idx = p2.three_nn(pos, known)
        w = p2.fp_weights(d2)
    else:
        idx = torch.randint(0, m - 1 if kind == "hub" else m, (b, n, 3), device=DEV, dtype=torch.int32)
        if kind == "hub":
            idx[:, ::3, 1] = 5
            idx[idx == 7] = 8                       # target 7 and m - 1 get no pairs
        w = torch.rand(b, n, 3, device=DEV)
        w = (w / w.sum(-1, keepdim=True)).contiguous()
    y = torch.randn(b, n, c, device=DEV)
    dz = torch.randn(b, n, c, device=DEV)
    scale, shift, mean = torch.randn(c, device=DEV), torch.randn(c, device=DEV), 0.1 * torch.randn(c, device=DEV)
    rstd, c1, c2 = torch.rand(c, device=DEV) + 0.5, 0.01 * torch.randn(c, device=DEV), 0.01 * torch.randn(c, device=DEV)
    outs = {}
    for ordered in (True, False):
        order = fn.local_spatial_order(known) if ordered else None
        rix = fn.ReverseIndex(idx, w, m, order)
        for form in ("default", "list"):
            monkeypatch.setenv("GEOT_GR_FORM", form)
            out = torch.full((b, m, c), float("nan"), device=DEV)
            call("geot_gather_rows_csr_bn_cl", DEV, b, c, n, m, 3, int(relu), ptr(y), ptr(dz), ptr(scale), ptr(shift), ptr(mean),
                 ptr(rstd), ptr(c1), ptr(c2), ptr(rix.ws), ptr(order), ptr(out))
            outs[(ordered, form)] = out
        assert torch.equal(outs[(ordered, "default")], outs[(ordered, "list")])
        outs[ordered] = outs[(ordered, "default")]
        # the plain row gather over the same index, both forms
        plain = {}
        for form in ("default", "list"):
            monkeypatch.setenv("GEOT_GR_FORM", form)
            plain[form] = rix.gather(dz)
        assert torch.equal(plain["default"], plain["list"])
    assert torch.equal(outs[True], outs[False])
    # fp64: gy = k0 (g - c1 - xhat c2), g = dz [relu: y k0 + shift > 0], then the interpolation gradient
    y64, dz64 = y.double().cpu(), dz.double().cpu()
    k0, mu, rs = scale.double().cpu(), mean.double().cpu(), rstd.double().cpu()
    mask = ((y.cpu() * scale.cpu() + shift.cpu()) > 0).double() if relu else 1.0     # (the mask as fp32 evaluates it)
    gy = k0 * (dz64 * mask - c1.double().cpu() - (y64 - mu) * rs * c2.double().cpu())
    want = torch.zeros(b, m, c, dtype=torch.float64)
    src = (gy.unsqueeze(2) * w.double().cpu().unsqueeze(-1)).reshape(b, n * 3, c)
    want.scatter_add_(1, idx.long().cpu().reshape(b, n * 3, 1).expand(-1, -1, c), src)
    got = outs[True]
    assert rel(got.transpose(1, 2), want.transpose(1, 2)) <= 2e-5 * (30 if kind == "hub" else 1)
    if kind == "hub":
        assert float(got[:, 7].abs().max()) == 0.0 and float(got[:, m - 1].abs().max()) == 0.0
