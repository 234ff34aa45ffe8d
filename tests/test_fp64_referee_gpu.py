"""GPU suite: the re-ordered ("factored") forms of the hot-path callers refereed in fp64.

The default model path moves the first 1x1 convolution of an FP / SA / EdgeConv / mini-PointNet stage in front of its
gather (geot_amd/openpoints/models/backbone/transformer.py, pointnet2_modules._sa_factored, fused_norm.fp_front).
"Same function, different fp32 summation order" is SHOWN here, not asserted against another fp32 run: every form is
compared with the reference's op order evaluated in fp64 on the CPU (indices from the GPU ops, which are exact), at
north_star's 1e-5 relative -- and the factored form may not sit farther from the fp64 result than the reference
order does (times a small factor for the luck of one draw)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
REL = 1e-5


def rel(got, want):
    want = want.detach().double().cpu()
    return float((got.detach().double().cpu() - want).abs().max() / want.abs().max().clamp_min(1e-30))


def _cloud(b, n, seed=0):
    from geot_amd.synth import make_batch
    return torch.from_numpy(make_batch(b, n, start_index=seed)[0]).to(DEV)


def _interp64(feats, idx, weight):
    """three_interpolate (pointnet2/_ext_src/src/interpolate_gpu.cu:101-102) in fp64 on the CPU."""
    b, c, m = feats.shape
    n = idx.shape[1]
    g = torch.gather(feats, 2, idx.long().reshape(b, 1, n * 3).expand(-1, c, -1)).view(b, c, n, 3)
    return (g * weight.unsqueeze(1)).sum(-1)


def _weights64(d2):
    dist = torch.sqrt(d2.double().cpu())
    r = 1.0 / (dist + 1e-8)
    return r / r.sum(2, keepdim=True)


def test_fp_front_against_fp64():
    """fused_norm.fp_front: interpolation + skip 1x1 conv in one kernel, forward and both gradients."""
    from geot_amd.fused_norm import fp_front
    from geot_amd.pointnet2 import pointnet2_utils as pu
    pos = _cloud(2, 6000)
    known = pos[:, :1500].contiguous()
    d2, idx = pu._ext.three_nn(pos, known)
    weight = pu._ext.fp_weights(d2)
    torch.manual_seed(0)
    a0, skip, wb0, up = (torch.randn(*s, device=DEV) for s in ((2, 70, 1500), (2, 5, 6000), (70, 5), (2, 70, 6000)))
    a64, wb64 = a0.double().cpu().requires_grad_(True), wb0.double().cpu().requires_grad_(True)
    w64 = weight.double().cpu()           # the kernel's own fp32 weights: the function under test starts behind them
    y64 = _interp64(a64, idx.cpu(), w64) + torch.matmul(wb64, skip.double().cpu())
    (y64 * up.double().cpu()).sum().backward()
    a, wb = a0.clone().requires_grad_(True), wb0.clone().requires_grad_(True)
    y, _ = fp_front(a, idx, weight, skip, wb)
    (y * up).sum().backward()
    assert rel(y, y64) <= REL and rel(a.grad, a64.grad) <= REL and rel(wb.grad, wb64.grad) <= REL
    # and the inverse-distance weights themselves against fp64 (pointnet2_modules.py:620-623)
    assert rel(weight, _weights64(d2)) <= REL


@pytest.mark.parametrize("c_known,c_skip,widths", [(48, 5, [64, 32]), (96, 0, [128, 64])])
def test_fp_module_factored_and_composed_against_fp64(c_known, c_skip, widths):
    """PointnetFPModule (pointnet2_modules.py:582-642, BatchNorm in training mode): the reference order on the HIP ops
    and transformer._fp_factored (first conv before the interpolation + fused front end) vs fp64."""
    from geot_amd.pointnet2.pointnet2_modules import PointnetFPModule
    from geot_amd.pointnet2 import pointnet2_utils as pu
    from geot_amd.openpoints.models.backbone.transformer import _fp_factored
    pos = _cloud(2, 5000, 3)
    known = pos[:, ::5].contiguous()
    d2, idx = pu._ext.three_nn(pos, known)
    for seed in range(1, 60):
        # a ReLU input within fp32 rounding of zero flips its mask between any two fp32 evaluations and moves a whole
        # gradient element: the referee (fp64) looks for such kinks and the test takes the first seed without one
        torch.manual_seed(seed)
        fp = PointnetFPModule(mlp=[c_known + c_skip] + widths).to(DEV).train()
        kf = torch.randn(2, c_known, known.shape[1], device=DEV)
        sk = torch.randn(2, c_skip, 5000, device=DEV) if c_skip else None
        up = torch.randn(2, widths[-1], 5000, device=DEV)
        # fp64 referee on the CPU: the reference's op chain with the GPU's (exact) neighbour ids
        fp64 = copy.deepcopy(fp).double().cpu()
        kf64 = kf.double().cpu().requires_grad_(True)
        x64 = _interp64(kf64, idx.cpu(), _weights64(d2))
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
    w64 = dict(fp64.named_parameters())
    errs = {}
    for mode in ("reference", "factored"):
        m = copy.deepcopy(fp)
        k = kf.clone().requires_grad_(True)
        y = m(pos, known, sk, k) if mode == "reference" else _fp_factored(m, pos, known, sk, k)
        (y * up).sum().backward()
        errs[mode] = [rel(y, y64), rel(k.grad, kf64.grad)] + [rel(p.grad, w64[n].grad) for n, p in m.named_parameters()]
        for n, b in m.named_buffers():          # running statistics after the step
            if b.dtype.is_floating_point:
                assert rel(b, dict(fp64.named_buffers())[n]) <= REL, (mode, n)
    for mode, e in errs.items():
        assert e[0] <= REL, (mode, "forward", e)
        assert max(e[1:]) <= 5e-5, (mode, "gradients", e)           # sums over 10 000 points behind a BatchNorm backward
    assert errs["factored"][0] <= 3 * errs["reference"][0] + 2e-6, errs
    assert max(errs["factored"][1:]) <= 3 * max(errs["reference"][1:]) + 5e-6, errs


def test_sa_module_training_factored_and_composed_against_fp64():
    """PointnetSAModuleVotes in training mode (pointnet2_modules.py:31-72): `_sa_factored` (first conv per point, then
    gathered) and the composed reference order vs fp64."""
    from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    from geot_amd.pointnet2 import pointnet2_utils as pu
    pos = _cloud(2, 4096, 9)
    torch.manual_seed(2)
    sa = PointnetSAModuleVotes(mlp=[6, 32, 32, 64], npoint=512, radius=0.15, nsample=16, use_xyz=True).to(DEV).train()
    feats = torch.randn(2, 6, 4096, device=DEV)
    up = torch.randn(2, 64, 512, device=DEV)
    inds = pu.furthest_point_sample(pos, 512)
    new_xyz = pu.gather_operation(pos.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = pu.ball_query(0.15, 16, pos, new_xyz).long().cpu()
    sa64 = copy.deepcopy(sa).double().cpu()
    f64 = feats.double().cpu().requires_grad_(True)
    p64 = pos.double().cpu().transpose(1, 2)

    def group(t):
        b, c, _ = t.shape
        return torch.gather(t, 2, idx.reshape(b, 1, -1).expand(-1, c, -1)).view(b, c, 512, 16)
    x64 = torch.cat([group(p64) - new_xyz.double().cpu().transpose(1, 2).unsqueeze(-1), group(f64)], 1)
    y64 = sa64.mlp_module(x64).max(-1)[0]
    (y64 * up.double().cpu()).sum().backward()
    w64 = dict(sa64.named_parameters())
    errs = {}
    for mode in ("reference", "factored"):
        m = copy.deepcopy(sa)
        m.factored_train = mode == "factored"
        f = feats.clone().requires_grad_(True)
        _, y, _ = m(pos, f, inds)
        (y * up).sum().backward()
        errs[mode] = [rel(y, y64), rel(f.grad, f64.grad)] + [rel(p.grad, w64[n].grad) for n, p in m.named_parameters()]
    for mode, e in errs.items():
        assert e[0] <= REL, (mode, "forward", e)
        assert max(e[1:]) <= 5e-5, (mode, "gradients", e)
    assert errs["factored"][0] <= 3 * errs["reference"][0] + 2e-6, errs
    assert max(errs["factored"][1:]) <= 3 * max(errs["reference"][1:]) + 5e-6, errs


def test_whole_model_both_orders_against_the_fp64_reference_order(oracle):
    """PointTransformer_seg_T (small config, 2 x 4096 points, training mode): logits and the gradient of every
    parameter in the factored and in the reference op order on the GPU vs the reference order in fp64 on the CPU (hot-path
    ops of the referee from the oracle: oracle/torch_cpu_ref.patched).  The errors of ~30 fp32 layers are reported, the
    factored order must not be worse than the reference order."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd.synth import make_batch, region_labels
    from oracle import torch_cpu_ref
    cfg = dict(trans_dim=384, depth=3, num_heads=4, group_size=32, num_group=128, encoder_dims=256, nclasses=17,
               drop_path_rate=0.0, downsample_targets=[2048, 1024, 512], extract_layers=[1, 2, 3])
    xyz = make_batch(2, 4096, start_index=5)[0]
    target = torch.from_numpy(region_labels(xyz))
    cls = torch.tensor([[0], [1]])
    torch.manual_seed(3)
    init = PointTransformer_seg_T(**cfg, dense="reference").state_dict()

    def run(mode, dev, dtype):
        m = PointTransformer_seg_T(**cfg, dense=mode, overlap=False)
        m.load_state_dict(init)
        m = m.to(dtype).to(dev).train()
        m.seg_head[2].p = 0.0
        pos = torch.from_numpy(xyz).to(dtype).to(dev)
        logit = m(pos, pos.transpose(1, 2).contiguous(), cls.to(dev), torch.eye(17, dtype=dtype, device=dev))[0]
        torch.nn.functional.cross_entropy(logit, target.to(dev)).backward()
        return logit.detach(), {n: p.grad.detach() for n, p in m.named_parameters() if p.grad is not None}
    with torch_cpu_ref.patched("port"):
        l64, g64 = run("reference", torch.device("cpu"), torch.float64)
    res = {mode: run(mode, DEV, torch.float32) for mode in ("reference", "factored")}
    zero_grad = ("encoder.first_conv.0.bias", "encoder.first_conv.3.bias", "encoder.second_conv.0.bias", "seg_head.0.bias")
    worst = {}
    for mode, (logit, grads) in res.items():
        e_l = rel(logit, l64)
        e_g = {k: float((grads[k].double().cpu() - g64[k]).norm() / (g64[k].norm() + 1e-30)) for k in g64 if k not in zero_grad}
        worst[mode] = (e_l, max(e_g.values()), max(e_g, key=e_g.get))
        assert set(grads) == set(g64)
    print("fp32 vs fp64 reference order: (logits max-rel, worst gradient norm-rel, which)", worst)
    # fp32 through ~30 layers with max-pools (a flipped arg-max moves a whole gradient row): 1e-4 on logits, 2e-3 on the
    # worst gradient norm -- for BOTH orders; and the factored order is not the worse one
    for mode in worst:
        assert worst[mode][0] <= 1e-4 and worst[mode][1] <= 5e-3, worst
    assert worst["factored"][0] <= 3 * worst["reference"][0] + 1e-6, worst
    assert worst["factored"][1] <= 3 * worst["reference"][1] + 1e-5, worst
