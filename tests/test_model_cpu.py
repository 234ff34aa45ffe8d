"""CPU-side tests of the model / training-step mirrors (no GPU compute): structure of PointTransformer_seg_T
against the reference's (parameter census from SURVEY.md section 2.4), the criteria and the filter_outlier anchors
against literal transcriptions of the reference loops, and -- world_size 2 over gloo -- that DDP's averaged gradients
of the data-parallel step equal the single-process ones (hot-path ops routed to the CPU oracle by the checker-side
patch in oracle/torch_cpu_ref.py; the product itself has no CPU path)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

SMALL = dict(trans_dim=384, depth=3, num_heads=4, group_size=16, num_group=32, encoder_dims=256, nclasses=17,
             drop_path_rate=0.0, downsample_targets=[256, 128, 64], extract_layers=[1, 2, 3])


def test_backbone_structure_matches_reference_census():
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
    m = PointTransformer_seg_T(**TOOTH_SEG_CFG)
    n_params = sum(p.numel() for p in m.parameters())
    assert abs(n_params - 27.05e6) < 0.01e6                  # "≈27.05 M fp32 params = 108 MB/step"
    keys = set(m.state_dict())
    for k in ("encoder.first_conv.0.weight", "encoder.second_conv.3.bias", "reduce_dim.weight", "pos_embed.2.weight",
              "blocks.blocks.11.attn.qkv.weight", "blocks.blocks.0.mlp.fc1.bias", "norm.weight",
              "propogation_0.mlp.layer0.conv.weight", "propogation_0.mlp.layer0.bn.bn.running_var",
              "propogation_2.mlp.layer1.conv.weight", "dgcnn_pro_1.layer1.0.weight", "dgcnn_pro_2.layer2.1.bias",
              "seg_head.0.weight", "seg_head.3.bias", "T_revision.weight", "T_linear.weight", "sigma"):
        assert k in keys, k
    assert tuple(m.propogation_0.mlp.layer0.conv.weight.shape) == (1536, 389, 1, 1)
    assert tuple(m.dgcnn_pro_1.layer2[0].weight.shape) == (384, 1024, 1, 1)
    assert not any("attn.qkv.bias" in k for k in keys)       # qkv_bias=False (transformer.py:45)
    assert float(m.sigma[0]) == pytest.approx(0.4) and float(m.T_linear.weight.abs().sum()) == 0.0


def test_poly1_losses_match_literal_transcription():
    from geot_amd.openpoints.loss import Poly1FocalLoss, Poly1FocalLoss_U_corr
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(3, 17, 40, generator=g, dtype=torch.float64)
    labels = torch.randint(0, 17, (3, 40), generator=g)
    conf = torch.rand(3, 40, generator=g, dtype=torch.float64)
    # openpoints/loss/build.py:215-258 / 841-884, written out
    onehot = F.one_hot(labels.unsqueeze(1), 17).transpose(1, -1).squeeze(-1).to(logits.dtype)
    p = torch.sigmoid(logits)
    ce = F.binary_cross_entropy_with_logits(logits, onehot, reduction="none")
    pt = onehot * p + (1 - onehot) * (1 - p)
    fl = (0.25 * onehot + 0.75 * (1 - onehot)) * ce * (1 - pt) ** 2.0
    poly1 = fl + 1.0 * torch.pow(1 - pt, 3.0)
    assert torch.allclose(Poly1FocalLoss()(logits, labels), poly1.mean(), rtol=1e-12)
    for thresh in (0.0, 0.5):
        mask = conf.ge(thresh).unsqueeze(1).repeat(1, 17, 1)
        want = torch.sum(poly1 * mask) / (mask.sum() + 0.001)
        assert torch.allclose(Poly1FocalLoss_U_corr()(logits, labels, conf, thresh=thresh), want, rtol=1e-12)


def test_filter_outlier_anchors_match_the_reference_loop():
    from geot_amd import ntm
    g = torch.Generator().manual_seed(11)
    eta = torch.softmax(3 * torch.randn(3, 17, 500, generator=g), dim=1)
    sigma = torch.full((17,), 0.4)
    ema_t = torch.eye(17)
    # train.py:505-526 with cfg.filter_outlier, literally (the in-place edit of eta_corr included)
    eta_corr = eta.clone()
    class_T = torch.empty(17, 17)
    for cc in range(17):
        thresh = eta_corr[:, cc, :].quantile(q=0.97)
        robust = eta_corr[:, cc, :]
        robust[robust >= thresh] = 0.0
        best = torch.argmax(robust.contiguous().view(-1))
        class_T[cc] = eta_corr[best // 500, :, best % 500]
    got = ntm.class_transition(eta, sigma, ema_t, filter_outlier=True)[2]
    assert torch.equal(got, class_T)
    plain = ntm.class_transition(eta, sigma, ema_t)[2]
    assert not torch.equal(plain, class_T)


def test_whole_part_seg_concatenates_views():
    from geot_amd.openpoints.models.segmentation import WholePartSeg

    class Probe(torch.nn.Module):
        def forward(self, p, f, c, T):
            self.seen = (p.shape, f.shape, c.shape, T)
            return p.sum(), None, None, None
    seg = WholePartSeg(segmentor_args=Probe())
    d = {"pos": torch.zeros(2, 8, 3), "x": torch.zeros(2, 3, 8), "cls": torch.zeros(2, 1, dtype=torch.long)}
    u = {k + s: v for s in ("_s", "_w") for k, v in d.items()}
    u["T"] = torch.eye(17)
    seg(d, u0=u, fixmatch=True)
    assert seg.segmentor.seen[0] == (6, 8, 3) and seg.segmentor.seen[3] is u["T"]
    seg(d, u0=u)
    assert seg.segmentor.seen[0] == (4, 8, 3)
    seg(u, if_teacher=True)
    assert seg.segmentor.seen[0] == (2, 8, 3) and seg.segmentor.seen[3] is None


# ---- data-parallel step: DDP gradients == single-process gradients (gloo, world_size 2) -----------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grads_of(model, pos, cls, target):
    from geot_amd.openpoints.loss import Poly1FocalLoss
    model.zero_grad(set_to_none=True)
    logits = model(pos, pos.transpose(1, 2).contiguous(), cls)[0]
    Poly1FocalLoss()(logits, target).backward()
    inner = model.module if hasattr(model, "module") else model
    return {n: p.grad.clone() for n, p in inner.named_parameters() if p.grad is not None}


def _make(seed=5):
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd.synth import make_batch, region_labels
    torch.manual_seed(seed)
    model = PointTransformer_seg_T(**SMALL, dense="reference", overlap=False).eval()   # eval: BN running stats, no dropout
    xyz, _ = make_batch(4, 1024)
    return model, torch.from_numpy(xyz), torch.zeros(4, 1, dtype=torch.long), torch.from_numpy(region_labels(xyz))


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from geot_amd import dist_utils, train_step
    from oracle import torch_cpu_ref
    dist_utils.init("gloo")
    with torch_cpu_ref.patched("port"):
        model, pos, cls, target = _make()
        net = train_step.ddp(model, torch.device("cpu"), sync_bn=False, unused=train_step.UNUSED_SUPERVISED)
        assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
        lo, hi = dist_utils.cloud_range(rank, 2)
        g1 = _grads_of(net, pos[lo:hi], cls[lo:hi], target[lo:hi])
        g2 = _grads_of(net, pos[lo:hi], cls[lo:hi], target[lo:hi])      # a second iteration: the reducer re-arms
    assert all(torch.equal(g1[k], g2[k]) for k in g1)
    q.put((rank, {k: v.numpy() for k, v in g1.items()}))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_ddp_gradients_equal_single_process():
    import torch.multiprocessing as mp
    from oracle import torch_cpu_ref
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    with torch_cpu_ref.patched("port"):
        model, pos, cls, target = _make()
        want = _grads_of(model, pos, cls, target)                            # all 4 clouds in one process
    assert set(res[0]) == set(want) and "sigma" not in want and "T_linear.weight" not in want
    for k, w in want.items():
        assert np.array_equal(res[0][k], res[1][k]), k                       # every rank holds the same average
        err = np.linalg.norm(res[0][k] - w.numpy()) / (float(w.norm()) + 1e-12)
        assert err <= 1e-3, (k, err)                 # fp32: the GEMM blocking differs with the batch size


def test_lean_transformer_block_is_the_same_function():
    """dense != "reference" runs the transformer blocks through fewer launches (one permuted copy of the qkv
    projection + unbind, the 1/sqrt(d) inside baddbmm, addcmul for drop-path + residual): the same function and the
    same random draws, checked in fp64 with and without stochastic depth."""
    from geot_amd.openpoints.models.backbone.transformer import Block
    torch.manual_seed(0)
    x = torch.randn(3, 50, 96, dtype=torch.double, requires_grad=True)
    for dp in (0.0, 0.3):
        blk = Block(96, 4, drop_path=dp).double().train()
        outs = []
        for lean in (False, True):
            blk.attn.lean = lean
            torch.manual_seed(7)
            y = blk(x)
            outs.append([y] + list(torch.autograd.grad(y.square().sum(), [x] + list(blk.parameters()))))
        for a, b in zip(*outs):
            assert float((a - b).abs().max()) < 1e-12


def test_mode_guard_reasserts_a_child_toggled_on_its_own():
    """train_step._mode skips the 180-module walk when the tree is in the wanted mode -- judged by the root AND by every
    BatchNorm / Dropout / DropPath child, so a child flipped on its own is put back."""
    import torch.nn as nn
    from geot_amd.train_step import _mode
    m = nn.Sequential(nn.Linear(3, 3), nn.BatchNorm1d(3), nn.Sequential(nn.Dropout(0.5)))
    _mode(m, True)
    assert m[1].training and m[2][0].training
    m[1].eval()                                   # the root still says "training"
    _mode(m, True)
    assert m[1].training
    m[2][0].eval()
    _mode(m, True)
    assert m[2][0].training
    _mode(m, False)
    assert not m.training and not m[1].training and not m[2][0].training
