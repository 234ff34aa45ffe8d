"""GPU suite: the HIP kernels of the NTM / loss / model half of the hot path against fixtures PRODUCED BY THE
REFERENCE'S OWN CODE (tests/golden/make_ntm_golden.py: the class bodies / statement ranges of /root/reference
executed in place on the CPU, fp32 and fp64).

Tolerance = north_star's 1e-5 relative, read two ways and never looser:
* `close`: every element within 1e-5 of the scale of ITS ROW (the last axis: a row of a transition matrix sums to 1,
  a row of logits has the magnitude of its logits) -- not of the whole tensor -- against the reference's fp32 AND fp64
  results, for everything computed in one pass;
* `referee`: for long fp32 reductions (weight / sigma gradients: hundreds of cancelling terms), where the reference's
  OWN fp32 result is farther than 1e-5 from its fp64 one, ours must be no farther from the fp64 result than the
  reference's fp32 run is (x 2: two different summation orders), or within 1e-5 of the row scale -- whichever is larger.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from det_init import det_state  # noqa: E402
from test_ref_fixtures_cpu import _block_cases, check_block_case  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL = 1e-5


def T(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


def _row_scale(want):
    """|want| maximum over the last axis (the row an element sits in), floored at 1 % of the tensor's maximum: a row of
    zeros (an unused class) is held to the rounding of its neighbours, not to 1e-5 of nothing."""
    want = np.abs(np.asarray(want, dtype=np.float64))
    shape = want.shape
    while want.ndim and want.shape[-1] == 1:       # (a 1x1-conv weight is (out, in, 1): its rows run along `in`)
        want = want[..., 0]
    if want.ndim == 0:
        return np.maximum(want, 1e-30).reshape((1,) * len(shape))
    top = want.max() if want.size else 0.0
    row = np.maximum(np.maximum(want.max(axis=-1, keepdims=True), 1e-2 * top), 1e-30)
    return row.reshape(row.shape + (1,) * (len(shape) - row.ndim))


def close(got, *wants, rel=REL, what=""):
    """|got - want| <= rel * (scale of the element's row), element-wise, for the reference's fp32 AND its fp64 result."""
    got = np.asarray(got, dtype=np.float64)
    for want in wants:
        want = np.asarray(want, dtype=np.float64)
        ratio = np.abs(got - want) / _row_scale(want)
        assert ratio.max() <= rel, (what, float(ratio.max()), float(np.abs(want).max()))


def referee(got, ref32, ref64, what="", factor=2.0, rel=REL):
    """Ours is no farther from the reference's fp64 result than the reference's own fp32 result is (x factor), or within
    rel of the row scale: the bound for long fp32 reductions whose fp32 reference is itself not 1e-5-accurate."""
    got, ref32, ref64 = (np.asarray(a, dtype=np.float64) for a in (got, ref32, ref64))
    scale = _row_scale(ref64)
    ours, theirs = np.abs(got - ref64) / scale, np.abs(ref32 - ref64) / scale
    bound = max(factor * float(theirs.max()), rel)
    assert float(ours.max()) <= bound, (what, "ours %.3g" % ours.max(), "reference fp32 %.3g" % theirs.max(), "bound %.3g" % bound)
    return float(ours.max()), float(theirs.max())


def _predictor(W):
    from geot_amd.ntm import sig_t_mean
    mod = sig_t_mean(W.shape[0]).to(DEV)
    with torch.no_grad():
        for kk, l in enumerate(mod.fc):
            l.weight.copy_(T(W[kk]))
    return mod


@pytest.mark.parametrize("C", [17, 5, 20])
def test_sig_t_mean_kernel_equals_the_reference_class(golden, C):
    """transformer.py:1099-1131 executed in place -> HIP sig_t_mean forward + fused weight gradient."""
    g = golden("ntm_ref_sig_t_mean.npz")
    t = "c%d_" % C
    mod = _predictor(g[t + "W"])
    out = mod(T(g[t + "p"]), T(g[t + "cm"]))
    close(host(out), g[t + "ins_T_f32"], g[t + "ins_T_f64"], what="ins_T")
    (out * T(g[t + "G"])).sum().backward()
    gW = host(torch.stack([l.weight.grad for l in mod.fc]))
    referee(gW, g[t + "gW_f32"], g[t + "gW_f64"], what="grad W")      # a (B*N)-term fp32 sum behind a 1 / row-norm


@pytest.mark.parametrize("tag,filt", [("c17_plain_", False), ("c17_filt_", True), ("c5_plain_", False)])
def test_transition_block_kernels_equal_the_reference_statements(golden, tag, filt):
    """train.py:505-557 executed in place -> class anchors + fused 17x17 block + predictor + fused correction,
    forward and the three gradients (sigma, strong-view logits, predictor weights)."""
    from geot_amd import ntm
    g = golden("ntm_ref_transition.npz")
    sigma = T(g[tag + "sigma"]).requires_grad_(True)
    strong = T(g[tag + "strong"]).requires_grad_(True)
    pred = _predictor(g[tag + "W"])
    corr, nxt, class_T, prior = ntm.class_transition(T(g[tag + "eta"]), sigma, T(g[tag + "ema_t"]), 0.999, 0.999,
                                                     filter_outlier=filt)
    assert np.array_equal(host(class_T), g[tag + "class_T_f32"])                  # copies of softmax rows: exact
    close(host(prior), g[tag + "prior_T_f32"], g[tag + "prior_T_f64"], what="prior_T")
    close(host(corr), g[tag + "ema_t_corr_f32"], g[tag + "ema_t_corr_f64"], what="ema_t_corr")
    close(host(nxt), g[tag + "ema_t_next_f32"], g[tag + "ema_t_next_f64"], what="ema_t_next")
    insT = pred(torch.softmax(strong, dim=1).detach(), T(g[tag + "cm"]))
    close(host(insT), g[tag + "insT_f32"], g[tag + "insT_f64"], what="insT")
    out = ntm.correct_logits(strong, insT, corr, 0.9)
    close(host(out), g[tag + "pred_u_strong_corr_f32"], g[tag + "pred_u_strong_corr_f64"], what="pred_corr")
    (out * T(g[tag + "G"])).sum().backward()
    close(host(strong.grad), g[tag + "g_strong_f32"], g[tag + "g_strong_f64"], what="d strong")
    referee(host(sigma.grad), g[tag + "g_sigma_f32"], g[tag + "g_sigma_f64"], what="d sigma")   # 2 x 96 x 17 cancelling terms
    gW = host(torch.stack([l.weight.grad for l in pred.fc]))
    referee(gW, g[tag + "g_W_f32"], g[tag + "g_W_f64"], what="d W")


@pytest.mark.parametrize("case,k", [("k32", 32), ("k7", 7), ("k7dup", 7)])
def test_graph_loss_kernels_equal_the_reference_classes(golden, case, k):
    """utils/insT_loss.py executed in place -> fused graph-loss kernels, given the reference's own neighbour lists
    (cdist + topk, whose tie order among duplicates is torch's) -- and, where the cloud has no duplicates, end to
    end through our exact kNN (the few near-tie swaps of cdist's expanded form move the loss by < 1e-5)."""
    from geot_amd import ntm
    g = golden("ntm_ref_losses.npz")
    xyz, labels, Tm, probs = (g[case + "_" + n] for n in ("xyz", "labels", "T", "probs"))
    for grad_mode in ("graph", "gather", "atomic"):
        os.environ["GEOT_NTM_GRAD"] = grad_mode
        try:
            Tt = T(Tm).requires_grad_(True)
            loss = ntm.threeD_space_loss(k=k, sigma=1.0)(T(xyz), T(labels, torch.int64), Tt, nbr=T(g[case + "_threed_nbr_f32"], torch.int32))
            loss.backward()
        finally:
            del os.environ["GEOT_NTM_GRAD"]
        close(loss.item(), g[case + "_threed_loss_f32"], what="threeD loss " + grad_mode)
        if grad_mode == "atomic":      # (the A/B scatter form: float atomics in arrival order)
            referee(host(Tt.grad), g[case + "_threed_grad_f32"], g[case + "_threed_grad_f64"], what="threeD grad atomic")
        else:
            close(host(Tt.grad), g[case + "_threed_grad_f32"], g[case + "_threed_grad_f64"], what="threeD grad " + grad_mode)
    Tt = T(Tm).requires_grad_(True)
    floss = ntm.feature_space_loss(k=k, sigma=1.0)(T(probs), T(labels, torch.int64), Tt, nbr=T(g[case + "_feat_nbr_f32"], torch.int64))
    floss.backward()
    close(floss.item(), g[case + "_feat_loss_f32"], what="feature loss")
    referee(host(Tt.grad), g[case + "_feat_grad_f32"], g[case + "_feat_grad_f64"], what="feature grad")     # float atomics
    Tt = T(Tm).requires_grad_(True)
    iloss = ntm.Idenyity_loss()(Tt, torch.eye(17, device=DEV))
    iloss.backward()
    close(iloss.item(), g[case + "_ident_loss_f32"], g[case + "_ident_loss_f64"], what="identity loss")
    close(host(Tt.grad), g[case + "_ident_grad_f32"], g[case + "_ident_grad_f64"], what="identity grad")
    if case != "k7dup":
        Tt = T(Tm).requires_grad_(True)
        loss = ntm.threeD_space_loss(k=k, sigma=1.0)(T(xyz), T(labels, torch.int64), Tt)
        close(loss.item(), g[case + "_threed_loss_f64"], rel=1e-4, what="threeD loss, own neighbours")
        floss = ntm.feature_space_loss(k=k, sigma=1.0)(T(probs), T(labels, torch.int64), Tt)
        close(floss.item(), g[case + "_feat_loss_f64"], rel=1e-3, what="feature loss, own neighbours")


def test_poly1_loss_kernels_equal_the_reference_classes(golden):
    """openpoints/loss/build.py:183-258, 799-892 executed in place -> fused Poly-1 focal kernels (fwd + bwd)."""
    from geot_amd.openpoints.loss.build import Poly1FocalLoss, Poly1FocalLoss_U_corr
    g = golden("poly1_ref.npz")
    lab, conf, mask = T(g["labels"], torch.int64), T(g["conf"]), T(g["mask"], torch.bool)

    def run(fn, x, *a, **kw):
        xt = T(x).requires_grad_(True)
        loss = fn(xt, *a, **kw)
        loss.backward()
        return loss.item(), host(xt.grad)
    cases = {
        "sup_mean": lambda: run(Poly1FocalLoss(), g["logits"], lab),
        "sup_sum": lambda: run(Poly1FocalLoss(reduction="sum"), g["logits"], lab),
        "sup_flat": lambda: run(Poly1FocalLoss(), g["flat_logits"], T(g["flat_labels"], torch.int64)),
        "sup_eps2_a-1_g3": lambda: run(Poly1FocalLoss(epsilon=2.0, alpha=-1.0, gamma=3.0), g["logits"], lab),
        "u_t0": lambda: run(Poly1FocalLoss_U_corr(), g["logits"], lab, conf, thresh=0.0),
        "u_t095": lambda: run(Poly1FocalLoss_U_corr(), g["logits"], lab, conf, thresh=0.95),
        "u_t07": lambda: run(Poly1FocalLoss_U_corr(), g["logits"], lab, conf, thresh=0.7),
        "u_mask": lambda: run(Poly1FocalLoss_U_corr(), g["logits"], lab, conf, thresh=0.5, mask=mask),
    }
    for name, fn in cases.items():
        loss, grad = fn()
        close(loss, g[name + "_loss_f32"], g[name + "_loss_f64"], what=name + " loss")
        close(grad, g[name + "_grad_f32"], g[name + "_grad_f64"], what=name + " grad")


@pytest.mark.parametrize("mode", ["reference", "lean"])
@pytest.mark.parametrize("name", ["mlp", "attn", "attn_bias", "block", "ench", "encoder_train"])
def test_transformer_blocks_equal_the_reference_classes(golden, name, mode):
    """transformer.py:16-136, 389-421 executed in place -> the mirror modules on the GPU in the reference op order
    and in the lean / factored order the model uses by default (fused LayerNorm / attention split / BatchNorm
    kernels), against the reference's fp64 run: same function, so the same 1e-5."""
    from geot_amd.openpoints.models.backbone import transformer as TR
    g = golden("blocks_ref.npz")
    B, L, D, H = (int(v) for v in g["dims"])
    make, prefix, in_names = _block_cases()[name]
    mod = make(D, H)
    if mode == "lean":
        if isinstance(mod, TR.Encoder):
            mod = TR.Encoder(64, factored=True)
        for m in mod.modules():
            if isinstance(m, (TR.Attention, TR.Mlp)):
                m.lean = True
    mod = det_state(mod, prefix).to(DEV).train()
    dt = torch.float32
    ins = [T(g[k]).requires_grad_(True) for k in in_names]
    G = T(g["Genc" if name.startswith("encoder") else "G"])
    y = mod(*ins)
    ys = y if isinstance(y, (list, tuple)) else [y]
    sum((yy * G).sum() * (i + 1) for i, yy in enumerate(ys)).backward()
    for i, yy in enumerate(ys):
        close(host(yy), g["%s_y%d_f64" % (name, i)], what="%s y%d" % (name, i))
    for i, t in enumerate(ins):
        referee(host(t.grad), g["%s_gin%d_f32" % (name, i)], g["%s_gin%d_f64" % (name, i)], what="%s gin%d" % (name, i))
    params = dict(mod.named_parameters())
    pre = "%s_gw_" % name
    for key in g.files:
        if key.startswith(pre) and key.endswith("_f64"):
            pname = key[len(pre):-4].replace("__", ".")
            want = g[key]
            if np.abs(want).max() < 1e-6:     # a bias in front of a BatchNorm: analytically zero, rounding noise at the scale of G
                assert np.abs(host(params[pname].grad)).max() <= 2e-5, (name, pname)
                continue
            referee(host(params[pname].grad), g[key[:-4] + "_f32"], want, what="%s %s" % (name, pname))
    if name == "encoder_train":
        for n in ("first_conv.1.running_mean", "first_conv.1.running_var", "second_conv.1.running_mean",
                  "second_conv.1.running_var"):
            close(host(mod.state_dict()[n]), g["encoder_after_%s_f64" % n.replace(".", "__")], what=n)
        det_state(mod, prefix)
        with torch.no_grad():
            close(host(mod.eval()(T(g["groups"]))), g["encoder_eval_y0_f64"], what="encoder eval")


@pytest.mark.parametrize("dense", ["reference", "factored"])
def test_dgcnn_propagation_equals_the_reference_class(golden, dense):
    """transformer.py:304-384 executed in place (knn_cuda.KNN substituted by the (d2, index) stand-in, see the
    fixture's meta) -> fused graph feature / EdgeConv tail kernels, both op orders."""
    from geot_amd.openpoints.models.backbone.transformer import DGCNN_Propagation
    g = golden("dgcnn_ref.npz")
    mod = det_state(DGCNN_Propagation(k=int(g["k"]), dense=dense), "dg.").to(DEV)
    f, f_q = T(g["f"]).requires_grad_(True), T(g["f_q"]).requires_grad_(True)
    y = mod(T(g["coor"]), f, T(g["coor_q"]), f_q)
    close(host(y), g["y_f32"], g["y_f64"], what="DGCNN y")
    (y * T(g["G"])).sum().backward()
    referee(host(f.grad), g["g_f_f32"], g["g_f_f64"], what="d f")
    referee(host(f_q.grad), g["g_fq_f32"], g["g_fq_f64"], what="d f_q")
    feat = mod.get_graph_feature(T(g["coor_q"]), f_q.detach(), T(g["coor"]), f.detach())
    assert np.array_equal(host(feat[:, ::37, ::7, :]), g["graph_feature_slice_f32"])          # differences of fp32: exact
    params = dict(mod.named_parameters())
    for n in ("layer1.0.weight", "layer1.1.weight", "layer2.1.bias"):
        key = "gw_%s_" % n.replace(".", "__")
        got = host(params[n].grad)
        got = got.reshape(-1)[::41] if got.size > 4096 else got
        referee(got, g[key + "f32"], g[key + "f64"], what=n)
