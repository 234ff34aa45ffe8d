"""CPU suite: the oracle's NTM restatement (oracle/np_ntm.py) and the package's host-side torch logic against
fixtures PRODUCED BY THE REFERENCE'S OWN CODE (tests/golden/make_ntm_golden.py executes the class bodies /
statement ranges of /root/reference in place on the CPU, in fp32 and in fp64).  This is what pins
SURVEY.md section 8 rows a17-a19, the Poly-1 losses and the transformer / mini-PointNet blocks."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import np_ntm

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from det_init import det_state  # noqa: E402


def t64(a):
    return torch.from_numpy(np.asarray(a)).double()


# ------------------------------------------------------------------------------------------ oracle == reference (fp64)
@pytest.mark.parametrize("C", [17, 5, 20])
def test_np_sig_t_mean_equals_the_reference_class(golden, C):
    g = golden("ntm_ref_sig_t_mean.npz")
    t = "c%d_" % C
    p, cm, W, G = g[t + "p"], g[t + "cm"], g[t + "W"], g[t + "G"]
    out = np_ntm.sig_t_mean(p, cm, W)
    assert np.abs(out - g[t + "ins_T_f64"]).max() < 1e-13
    assert np.abs(out - g[t + "ins_T_f32"]).max() < 2e-6            # the reference's own fp32 rounding
    gW = np_ntm.sig_t_mean_grad_W(p, cm, W, G)
    assert np.abs(gW - g[t + "gW_f64"]).max() < 1e-10 * max(1.0, np.abs(gW).max())
    assert np.abs(gW - g[t + "gW_f32"]).max() < 2e-4 * np.abs(gW).max()


@pytest.mark.parametrize("tag,filt", [("c17_plain_", False), ("c17_filt_", True), ("c5_plain_", False)])
def test_np_transition_block_equals_the_reference_statements(golden, tag, filt):
    """train.py:505-557 executed in place vs oracle/np_ntm.py (class anchors, prior, EMA, correction, gradients)."""
    g = golden("ntm_ref_transition.npz")
    eta, sigma, ema_t, cm, W = (g[tag + k] for k in ("eta", "sigma", "ema_t", "cm", "W"))
    r = np_ntm.class_transition(eta, sigma, ema_t, 0.999, 0.999, filter_outlier=filt)
    for k in ("class_T", "prior_T", "new_T", "ema_t_corr"):
        assert np.abs(r[k] - g[tag + k + "_f64"]).max() < 1e-13, k
        assert np.abs(r[k] - g[tag + k + "_f32"]).max() < 1e-5 * max(1.0, np.abs(r[k]).max()), k
    assert np.abs(r["ema_t_next"] - g[tag + "ema_t_next_f64"]).max() < 1e-13
    strong = g[tag + "strong"]
    sm = np.exp(strong.astype(np.float64) - strong.max(1, keepdims=True))
    sm = sm / sm.sum(1, keepdims=True)
    insT = np_ntm.sig_t_mean(sm, cm, W)
    assert np.abs(insT - g[tag + "insT_f64"]).max() < 1e-12
    _, corr = np_ntm.correct_logits(strong, insT, r["ema_t_corr"], 0.9)
    assert np.abs(corr - g[tag + "pred_u_strong_corr_f64"]).max() < 1e-11
    assert np.abs(corr - g[tag + "pred_u_strong_corr_f32"]).max() < 1e-5 * np.abs(corr).max()
    gl, gi, _ = np_ntm.correct_logits_grads(strong, insT, r["ema_t_corr"], 0.9, g[tag + "G"])
    assert np.abs(gl - g[tag + "g_strong_f64"]).max() < 1e-11
    gW = np_ntm.sig_t_mean_grad_W(sm, cm, W, gi)
    assert np.abs(gW - g[tag + "g_W_f64"]).max() < 1e-9 * max(1.0, np.abs(gW).max())


@pytest.mark.parametrize("case", ["k32", "k7", "k7dup"])
def test_np_graph_losses_equal_the_reference_classes(golden, case):
    """utils/insT_loss.py executed in place; the neighbour lists are the reference's own (cdist + topk)."""
    g = golden("ntm_ref_losses.npz")
    xyz, labels, T, probs = (g[case + "_" + k] for k in ("xyz", "labels", "T", "probs"))
    loss, grad, _ = np_ntm.threed_space_loss(xyz, labels, T, g[case + "_threed_nbr_f64"], sigma=1.0)
    assert abs(loss - g[case + "_threed_loss_f64"]) < 1e-13 * max(1.0, abs(loss))
    loss32, grad32, _ = np_ntm.threed_space_loss(xyz, labels, T, g[case + "_threed_nbr_f32"], sigma=1.0)
    assert abs(loss32 - g[case + "_threed_loss_f32"]) < 2e-6 * abs(loss32)
    assert np.abs(grad32 - g[case + "_threed_grad_f32"]).max() < 1e-5 * np.abs(grad32).max()
    floss, fgrad, _ = np_ntm.feature_space_loss(probs, labels, T, g[case + "_feat_nbr_f64"], sigma=1.0)
    assert abs(floss - g[case + "_feat_loss_f64"]) < 1e-13 * max(1.0, abs(floss))
    if case == "k7":
        assert np.abs(grad - g[case + "_threed_grad_f64"]).max() < 1e-14
        assert np.abs(fgrad - g[case + "_feat_grad_f64"]).max() < 1e-14
    assert abs(np_ntm.identity_loss(T, np.eye(17)) - g[case + "_ident_loss_f64"]) < 1e-14


def test_reference_neighbours_are_the_exact_ones_up_to_near_ties(golden, oracle):
    """knn_point = torch.cdist (expanded form) + topk: its k nearest differ from the exact (d2, index) list only
    where two candidates are closer than the expansion's rounding.  Counted, not assumed."""
    g = golden("ntm_ref_losses.npz")
    for case, k in (("k32", 32), ("k7", 7)):
        xyz = g[case + "_xyz"]
        idx, _ = oracle.knn_sorted(xyz, xyz, k + 1)
        same = (idx[:, :, 1:] == g[case + "_threed_nbr_f32"]).mean()
        assert same > 0.995, (case, same)
        same64 = (idx[:, :, 1:] == g[case + "_threed_nbr_f64"]).mean()
        assert same64 > 0.9995, (case, same64)


# ------------------------------------------------------------------------------------------ host-side torch logic
@pytest.mark.parametrize("tag,filt", [("c17_plain_", False), ("c17_filt_", True), ("c5_plain_", False)])
def test_package_class_transition_torch_path_equals_the_reference(golden, tag, filt):
    """geot_amd.ntm.class_transition on CPU tensors (its op-by-op torch branch) in fp64 against the reference's
    fp64 run, including d/d sigma through the Gaussian prior -- the gradient the fused kernel must reproduce."""
    from geot_amd import ntm
    g = golden("ntm_ref_transition.npz")
    sigma = t64(g[tag + "sigma"]).requires_grad_(True)
    corr, nxt, class_T, prior = ntm.class_transition(t64(g[tag + "eta"]), sigma, t64(g[tag + "ema_t"]), 0.999, 0.999,
                                                     filter_outlier=filt)
    assert np.abs(corr.detach().numpy() - g[tag + "ema_t_corr_f64"]).max() < 1e-13
    assert np.abs(nxt.numpy() - g[tag + "ema_t_next_f64"]).max() < 1e-13
    assert np.array_equal(class_T.numpy(), g[tag + "class_T_f64"])
    assert np.abs(prior.detach().numpy() - g[tag + "prior_T_f64"]).max() < 1e-13
    insT = t64(g[tag + "insT_f64"])
    newT = torch.nn.functional.normalize(0.9 * corr + 0.1 * insT, p=1, dim=2)
    strong = t64(g[tag + "strong"])
    B, C, N = strong.shape
    pc = torch.bmm(strong.permute(0, 2, 1).reshape(-1, 1, C), newT).squeeze(1).view(B, N, C).permute(0, 2, 1)
    (pc * t64(g[tag + "G"])).sum().backward()
    assert np.abs(sigma.grad.numpy() - g[tag + "g_sigma_f64"]).max() < 1e-12 * max(1.0, np.abs(g[tag + "g_sigma_f64"]).max())


def test_package_poly1_losses_torch_path_equals_the_reference(golden):
    """The un-fused composition of geot_amd.openpoints.loss.build (what runs on CPU tensors) in fp64."""
    from geot_amd.openpoints.loss.build import Poly1FocalLoss, Poly1FocalLoss_U_corr
    g = golden("poly1_ref.npz")
    lab, conf, mask = torch.from_numpy(g["labels"]), t64(g["conf"]), torch.from_numpy(g["mask"])

    def run(fn, x, *a, **kw):
        xt = t64(x).requires_grad_(True)
        loss = fn(xt, *a, **kw)
        loss.backward()
        return loss.item(), xt.grad.numpy()
    cases = {
        "sup_mean": lambda: run(Poly1FocalLoss(), g["logits"], lab),
        "sup_sum": lambda: run(Poly1FocalLoss(reduction="sum"), g["logits"], lab),
        "sup_flat": lambda: run(Poly1FocalLoss(), g["flat_logits"], torch.from_numpy(g["flat_labels"])),
        "sup_eps2_a-1_g3": lambda: run(Poly1FocalLoss(epsilon=2.0, alpha=-1.0, gamma=3.0), g["logits"], lab),
        "u_t0": lambda: run(Poly1FocalLoss_U_corr(), g["logits"], lab, conf, thresh=0.0),
        "u_t095": lambda: run(Poly1FocalLoss_U_corr(), g["logits"], lab, conf, thresh=0.95),
        "u_t07": lambda: run(Poly1FocalLoss_U_corr(), g["logits"], lab, conf, thresh=0.7),
        "u_mask": lambda: run(Poly1FocalLoss_U_corr(), g["logits"], lab, conf, thresh=0.5, mask=mask),
    }
    for name, fn in cases.items():
        loss, grad = fn()
        want, wgrad = float(g[name + "_loss_f64"]), g[name + "_grad_f64"]
        assert abs(loss - want) < 1e-12 * max(1.0, abs(want)), name
        assert np.abs(grad - wgrad).max() < 1e-12 * max(1.0, np.abs(wgrad).max()), name


def _block_cases():
    from geot_amd.openpoints.models.backbone import transformer as T
    return {
        "mlp": (lambda D, H: T.Mlp(D, 4 * D), "blk.mlp.", ("x",)),
        "attn": (lambda D, H: T.Attention(D, num_heads=H), "blk.attn.", ("x",)),
        "attn_bias": (lambda D, H: T.Attention(D, num_heads=H, qkv_bias=True), "blk.attnb.", ("x",)),
        "block": (lambda D, H: T.Block(D, H), "blk.block.", ("x",)),
        "ench": (lambda D, H: T.TransformerEncoder_h(embed_dim=D, depth=3, num_heads=H, extract_layers=[1, 3]), "blk.ench.",
                 ("x", "pos")),
        "encoder_train": (lambda D, H: T.Encoder(64), "enc.", ("groups",)),
    }


def check_block_case(g, name, mod, in_names, dt, dev, rtol):
    """Shared with the GPU suite: module outputs, input gradients and the stored parameter gradients vs the fixture."""
    dn = "f64" if dt == torch.float64 else "f32"
    ins = [torch.from_numpy(g[k]).to(dt).to(dev).requires_grad_(True) for k in in_names]
    G = torch.from_numpy(g["Genc" if name.startswith("encoder") else "G"]).to(dt).to(dev)
    y = mod(*ins)
    ys = y if isinstance(y, (list, tuple)) else [y]
    sum((yy * G).sum() * (i + 1) for i, yy in enumerate(ys)).backward()
    for i, yy in enumerate(ys):
        want = g["%s_y%d_%s" % (name, i, dn)]
        assert np.abs(yy.detach().cpu().numpy() - want).max() <= rtol * np.abs(want).max(), (name, "y", i)
    for i, t in enumerate(ins):
        want = g["%s_gin%d_%s" % (name, i, dn)]
        assert np.abs(t.grad.cpu().numpy() - want).max() <= rtol * np.abs(want).max(), (name, "gin", i)
    params = dict(mod.named_parameters())
    pre = "%s_gw_" % name
    for key in g.files:
        if key.startswith(pre) and key.endswith("_" + dn):
            pname = key[len(pre):-len(dn) - 1].replace("__", ".")
            want = g[key]
            assert np.abs(params[pname].grad.cpu().numpy() - want).max() <= rtol * max(np.abs(want).max(), 1.0), (name, pname)   # (a bias in front of a BatchNorm: analytically 0)


@pytest.mark.parametrize("name", ["mlp", "attn", "attn_bias", "block", "ench", "encoder_train"])
def test_package_transformer_blocks_reference_order_equal_the_reference(golden, name):
    """Mirror modules (reference op order = what runs on CPU tensors) in fp64 vs the reference classes in fp64:
    same parameters by state_dict name (det_state fills both from the names), same outputs and gradients."""
    g = golden("blocks_ref.npz")
    B, L, D, H = (int(v) for v in g["dims"])
    make, prefix, in_names = _block_cases()[name]
    mod = det_state(make(D, H).double(), prefix).train()
    check_block_case(g, name, mod, in_names, torch.float64, "cpu", 1e-12)
    if name == "encoder_train":
        for n in ("first_conv.1.running_mean", "first_conv.1.running_var", "second_conv.1.running_mean",
                  "second_conv.1.running_var"):
            want = g["encoder_after_%s_f64" % n.replace(".", "__")]
            assert np.abs(mod.state_dict()[n].numpy() - want).max() < 1e-12
        det_state(mod, prefix)
        with torch.no_grad():
            y = mod.eval()(t64(g["groups"]))
        assert np.abs(y.numpy() - g["encoder_eval_y0_f64"]).max() < 1e-12


def test_optimizer_parameter_groups_equal_the_reference_factory(golden):
    """openpoints/optim/optim_factory.py:66-119 executed in place on the model mirrors -> train_step.parameter_groups:
    the same parameters in the no-decay (1-D, *.bias) and the decay group, in the same order, same weight decay."""
    from geot_amd.train_step import parameter_groups, make_optimizer
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd.ntm import sig_t_mean
    g = golden("param_groups_ref.npz")
    small = dict(trans_dim=384, depth=2, num_heads=4, group_size=16, num_group=32, encoder_dims=256, nclasses=17,
                 drop_path_rate=0.0, downsample_targets=[256, 128, 64], extract_layers=[1, 2])
    for tag, model in (("seg", PointTransformer_seg_T(**small)), ("pred", sig_t_mean(17))):
        ids = {id(p): n for n, p in model.named_parameters()}
        groups = parameter_groups(model, 1e-4)
        assert len(groups) == int(g[tag + "_n_groups"])
        for i, grp in enumerate(groups):
            assert [ids[id(p)] for p in grp["params"]] == [str(n) for n in g["%s_g%d_names" % (tag, i)]]
            assert grp["weight_decay"] == float(g["%s_g%d_weight_decay" % (tag, i)])
            assert grp["lr_scale"] == float(g["%s_g%d_lr_scale" % (tag, i)])
        opt = make_optimizer(model, 1e-3, 1e-4)
        decays = {ids[id(p)]: grp["weight_decay"] for grp in opt.param_groups for p in grp["params"]}
        assert len(decays) == len(ids)
        if tag == "seg":
            assert decays["sigma"] == 0.0 and decays["norm.weight"] == 0.0 and decays["seg_head.0.bias"] == 0.0
            assert decays["blocks.blocks.0.attn.qkv.weight"] == 1e-4 and decays["T_linear.weight"] == 1e-4
