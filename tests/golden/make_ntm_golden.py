"""Regenerate the REFERENCE-EXECUTED fixtures of the Python half of the hot path (BUILD container only:
/root/reference does not exist on the GPU box).

    python tests/golden/make_ntm_golden.py

What runs is the reference's own code, in place: the class / function definitions (and, for the
class-transition block, the statement range inside ``train_one_epoch``) are taken out of the files under
/root/reference with ``ast``, compiled with the reference file as their filename and executed on the CPU.
Only what the package-level imports would have dragged in is supplied from outside (``torch``, ``nn``, ``F``,
``deepcopy``, the reference's own ``knn_point`` loaded by file path) and ``Tensor.cuda`` is the identity for
the duration of the run -- the reference hard-codes ``.cuda()`` (insT_loss.py:22 ..., transformer.py:1122).
Registry decorators (``@MODELS.register_module()``) are dropped.  No reference source text is written
anywhere: the fixtures hold seeded inputs, the outputs and the autograd gradients.

    ntm_ref_sig_t_mean.npz   sig_t_mean.forward + weight gradient     transformer.py:1099-1131   C = 17, 5, 20
    ntm_ref_transition.npz   class-T / prior / EMA / correction block train.py:505-557 (+48, 835-836)
    ntm_ref_losses.npz       threeD_space_loss, feature_space_loss, Idenyity_loss    utils/insT_loss.py:9-132
    poly1_ref.npz            Poly1FocalLoss, Poly1FocalLoss_U_corr    openpoints/loss/build.py:183-258, 799-892
    blocks_ref.npz           Mlp / Attention / Block / TransformerEncoder_h / Encoder   transformer.py:16-136, 389-421
    param_groups_ref.npz     get_parameter_groups (AdamW decay / no-decay split)   openpoints/optim/optim_factory.py:66-119
    dgcnn_ref.npz            DGCNN_Propagation                        transformer.py:304-384
                             (``knn_cuda.KNN`` is NOT in the reference tree; a stand-in with OUR tie rule,
                             exact fp32 (d2, index) order, is supplied and the fixture's metadata says so)

Every case is executed twice: in fp32 (what the reference computes) and with torch's default dtype set to
fp64 (the same statements, more digits) -- the fp64 outputs pin oracle/np_ntm.py to 1e-12, the fp32 ones are
what the HIP kernels are compared with at 1e-5.
"""
import ast
import contextlib
import importlib.util
import os
import sys
from copy import deepcopy
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from det_init import det_state, det_values  # noqa: E402
from geot_amd.synth import make_batch  # noqa: E402

REF = "/root/reference"
TRANSFORMER = "openpoints/models/backbone/transformer.py"
TRAIN = "examples/segmentation/train.py"
INST = "utils/insT_loss.py"
LOSS = "openpoints/loss/build.py"
PROVENANCE = {}


# --------------------------------------------------------------------------------------------- loading
def _load_by_path(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_defs(relpath, names, ns):
    """Execute the named top-level definitions (class / def / NAME = ...) of a reference file inside `ns`."""
    path = os.path.join(REF, relpath)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    picked = []
    for node in tree.body:
        name = getattr(node, "name", None)
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name):
            name = node.targets[0].id
        if name in names:
            if hasattr(node, "decorator_list"):
                node.decorator_list = []
            picked.append(node)
            PROVENANCE["%s::%s" % (relpath, name)] = "%d-%d" % (node.lineno, node.end_lineno)
    missing = set(names) - {getattr(n, "name", None) or n.targets[0].id for n in picked}
    assert not missing, "not found in %s: %s" % (relpath, sorted(missing))
    exec(compile(ast.Module(body=picked, type_ignores=[]), path, "exec"), ns)
    return ns


def ref_statements(relpath, func, first, last, ns_key):
    """The statement run [first .. last] (matched on their source text) of one block inside `func`, compiled
    as a code object that executes in a caller-supplied namespace."""
    path = os.path.join(REF, relpath)
    with open(path) as fh:
        src = fh.read()
    tree = ast.parse(src, filename=path)
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == func)
    seg = lambda s: " ".join((ast.get_source_segment(src, s) or "").split())   # noqa: E731
    for node in ast.walk(fn):
        for field in ("body", "orelse", "finalbody"):
            block = getattr(node, field, None)
            if not isinstance(block, list):
                continue
            texts = [seg(s) for s in block]
            if first in texts and last in texts[texts.index(first):]:
                i = texts.index(first)
                j = i + texts[i:].index(last)
                run = block[i:j + 1]
                PROVENANCE["%s::%s" % (relpath, ns_key)] = "%d-%d" % (run[0].lineno, run[-1].end_lineno)
                return compile(ast.Module(body=run, type_ignores=[]), path, "exec")
    raise AssertionError("statement range not found in %s:%s" % (relpath, func))


@contextlib.contextmanager
def on_cpu(dtype):
    """Tensor.cuda / Module.cuda are the identity; torch's default dtype is `dtype`."""
    t_cuda, m_cuda, old = torch.Tensor.cuda, nn.Module.cuda, torch.get_default_dtype()
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.set_default_dtype(dtype)
    try:
        yield
    finally:
        torch.Tensor.cuda, nn.Module.cuda = t_cuda, m_cuda
        torch.set_default_dtype(old)


class _KNNStandIn(nn.Module):
    """Stand-in for the absent third-party ``knn_cuda.KNN``: exact un-contracted fp32
    squared distances, neighbours by (d2, index) ascending, dist = sqrt(d2).  transpose_mode as upstream:
    True: ref (B,N,3), query (B,M,3) -> (B,M,k); False: ref (B,3,N), query (B,3,M) -> (B,k,M)."""

    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k, self.transpose_mode = k, transpose_mode

    def forward(self, ref, query):
        if not self.transpose_mode:
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        r = ref.detach().numpy().astype(np.float32)
        q = query.detach().numpy().astype(np.float32)
        d = q[:, :, None, :] - r[:, None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        order = np.lexsort((np.broadcast_to(np.arange(r.shape[1]), d2.shape), d2), axis=-1)[..., :self.k]
        dist = np.sqrt(np.take_along_axis(d2, order, -1))
        dist, idx = torch.from_numpy(dist).to(ref.dtype), torch.from_numpy(order.astype(np.int64))
        if not self.transpose_mode:
            dist, idx = dist.transpose(1, 2).contiguous(), idx.transpose(1, 2).contiguous()
        return dist, idx


class _DropPath(nn.Module):      # timm's DropPath at drop_prob 0 (the blocks use nn.Identity there anyway)
    def __init__(self, drop_prob=0.0):
        super().__init__()
        assert drop_prob == 0.0

    def forward(self, x):
        return x


def base_namespace():
    knn_mod = _load_by_path(os.path.join(REF, "openpoints/models/layers/knn.py"), "geot_ref_knn")
    PROVENANCE["openpoints/models/layers/knn.py::knn_point"] = "imported by file path"
    calls = []

    def knn_point(k, query, support=None):           # the reference's own, with its answer recorded
        dist, idx = knn_mod.knn_point(k, query, support)
        calls.append(idx.clone())
        return dist, idx

    return dict(torch=torch, nn=nn, F=F, np=np, deepcopy=deepcopy, knn_point=knn_point, KNN=_KNNStandIn,
                DropPath=_DropPath, _knn_calls=calls)


def npf(t):
    return t.detach().cpu().numpy().copy()      # a copy: .numpy() aliases the tensor, which may be rewritten later


def softmax_np(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return (e / e.sum(axis=axis, keepdims=True)).astype(np.float32)


def meta(extra=""):
    rows = ["%s lines %s" % kv for kv in sorted(PROVENANCE.items())]
    return np.array("executed from /root/reference (torch %s, CPU): %s. %s" % (torch.__version__, "; ".join(rows), extra))


DTYPES = (("f32", torch.float32), ("f64", torch.float64))


# --------------------------------------------------------------------------------------------- sig_t_mean
def _predictor(ns, C, prefix, dt):
    """The reference's sig_t_mean with fp32-valued weights (identical in the fp32 and the fp64 run): a predictor
    that has been trained towards I -- a mix of clamped and unclamped entries."""
    mod = ns["sig_t_mean"](C)
    with torch.no_grad():
        for kk, l in enumerate(mod.fc):
            w = det_values("%sfc.%d.weight" % (prefix, kk), (C, 2 * C), 2.0 * (3.0 / (2 * C)) ** 0.5) + np.float32(0.05)
            w[kk, kk] += np.float32(0.8)
            l.weight.copy_(torch.from_numpy(w).to(dt))
    return mod


def gen_sig_t_mean(ns):
    out = {}
    for C in (17, 5, 20):
        B, N = 2, 48
        p = softmax_np(det_values("sig.p%d" % C, (B, C, N), 6.0), 1)
        cm = softmax_np(det_values("sig.cm%d" % C, (C, C), 4.0), 1)
        G = det_values("sig.G%d" % C, (B * N, C, C), 2.0)
        tag = "c%d_" % C
        out.update({tag + "p": p, tag + "cm": cm, tag + "G": G})
        for dn, dt in DTYPES:
            with on_cpu(dt):
                mod = _predictor(ns, C, "sig_t_mean%d." % C, dt)
                ins_T = mod(torch.from_numpy(p).to(dt), torch.from_numpy(cm).to(dt))
                (ins_T * torch.from_numpy(G).to(dt)).sum().backward()
                W = torch.stack([l.weight for l in mod.fc])
                gW = torch.stack([l.weight.grad for l in mod.fc])
            if dn == "f32":
                out[tag + "W"] = npf(W)
            out[tag + "ins_T_" + dn] = npf(ins_T)
            out[tag + "gW_" + dn] = npf(gW)
        frac = float(((out[tag + "ins_T_f64"] * np.abs(out[tag + "ins_T_f64"]).sum(2, keepdims=True)) <= 1.0001e-5).mean())
        print("sig_t_mean C=%d: %.0f%% of the entries sit on the lower clamp" % (C, 100 * frac))
    np.savez_compressed(os.path.join(HERE, "ntm_ref_sig_t_mean.npz"), meta=meta(), **out)


# --------------------------------------------------------------------------------------------- train.py:505-557
def gen_transition(ns):
    block = ref_statements(TRAIN, "train_one_epoch", "c = cfg.num_classes", "ema_t = ema_t / torch.sum(ema_t, 1)",
                           "class-transition block")
    ref_defs(TRAIN, ["LABEL_PROJ", "gaussian"], ns)
    out = {}
    for C, filt in ((17, False), (17, True), (5, False)):
        B, N = 2, 96
        tag = "c%d_%s_" % (C, "filt" if filt else "plain")
        eta = softmax_np(det_values(tag + "eta", (B, C, N), 8.0), 1)
        logits_s = det_values(tag + "strong", (B, C, N), 6.0)
        sigma = det_values(tag + "sigma", (C,), 1.0, 1.2)
        ema_t = softmax_np(det_values(tag + "ema", (C, C), 5.0), 1)
        cm = softmax_np(det_values(tag + "cm", (C, C), 4.0), 1)
        G = det_values(tag + "G", (B, C, N), 2.0)
        out.update({tag + "eta": eta, tag + "strong": logits_s, tag + "sigma": sigma, tag + "ema_t": ema_t,
                    tag + "cm": cm, tag + "G": G})
        for dn, dt in DTYPES:
            with on_cpu(dt):
                pred = _predictor(ns, C, tag + "pred.", dt)
                env = dict(ns)
                env.update(cfg=SimpleNamespace(num_classes=C, filter_outlier=filt, geo_lambma=0.999, ema_t_decay=0.999,
                                               lambma=0.9),
                           pred_u=torch.from_numpy(eta).to(dt), batch_size_u=B, num_point=N,
                           sigma=torch.from_numpy(sigma).to(dt).requires_grad_(True),
                           ema_t=torch.from_numpy(ema_t).to(dt),
                           pred_u_strong=torch.from_numpy(logits_s).to(dt).requires_grad_(True),
                           T_predictor=pred, cm=torch.from_numpy(cm).to(dt))
                sigma_t, strong_t = env["sigma"], env["pred_u_strong"]
                exec(block, env)
                (env["pred_u_strong_corr"] * torch.from_numpy(G).to(dt)).sum().backward()
            if dn == "f32":
                out[tag + "W"] = npf(torch.stack([l.weight for l in pred.fc]))
            for k in ("class_T", "prior_T", "new_T", "ema_t_corr", "insT", "pred_u_strong_corr"):
                out[tag + k + "_" + dn] = npf(env[k])
            out[tag + "ema_t_next_" + dn] = npf(env["ema_t"])
            out[tag + "g_sigma_" + dn] = npf(sigma_t.grad)
            out[tag + "g_strong_" + dn] = npf(strong_t.grad)
            out[tag + "g_W_" + dn] = npf(torch.stack([l.weight.grad for l in pred.fc]))
    np.savez_compressed(os.path.join(HERE, "ntm_ref_transition.npz"),
                        meta=meta("cfg: geo_lambma 0.999, ema_t_decay 0.999, lambma 0.9 (cfgs/tooth_semi/*.yaml)"), **out)


# --------------------------------------------------------------------------------------------- insT_loss.py
def _coherent_labels(xyz, C):
    """Spatially coherent labels (teeth are compact): many equal-label neighbour pairs, some boundaries."""
    return np.clip(((xyz[..., 0] + 1.0) * 0.5 * C).astype(np.int64), 0, C - 1)


def gen_losses(ns):
    ref_defs(INST, ["feature_space_loss", "threeD_space_loss", "Idenyity_loss"], ns)
    out = {}
    C = 17
    for case, (B, N, k, dup) in {"k32": (2, 160, 32, 0.0), "k7": (2, 96, 7, 0.0), "k7dup": (1, 120, 7, 0.05)}.items():
        xyz, _ = make_batch(B, N, dup_frac=dup, start_index=40)
        labels = _coherent_labels(xyz, C)
        T = det_values(case + ".T", (B * N, C, C), 1.0, 0.5)
        T = (T / T.sum(2, keepdims=True)).astype(np.float32)
        probs = softmax_np(det_values(case + ".logits", (B, C, N), 4.0), 1)
        ident = np.eye(C, dtype=np.float32)
        out.update({case + "_xyz": xyz, case + "_labels": labels, case + "_T": T, case + "_probs": probs})
        for dn, dt in DTYPES:
            with on_cpu(dt):
                calls = ns["_knn_calls"]
                for name, cls, lead in (("threed", "threeD_space_loss", xyz), ("feat", "feature_space_loss", probs)):
                    del calls[:]
                    Tt = torch.from_numpy(T).to(dt).requires_grad_(True)
                    mod = ns[cls](k=k, sigma=1.0, num_classes=C)
                    loss = mod(torch.from_numpy(lead).to(dt), torch.from_numpy(labels), Tt)
                    loss.backward()
                    key = "%s_%s_" % (case, name)
                    out[key + "loss_" + dn] = npf(loss)
                    out[key + "grad_" + dn] = npf(Tt.grad)   # (fp64 too: the referee of the fp32 gradients' tolerance)
                    out[key + "nbr_" + dn] = npf(calls[0][:, :, 1:]).astype(np.int32)   # what the reference kept
                Tt = torch.from_numpy(T).to(dt).requires_grad_(True)
                loss = ns["Idenyity_loss"]()(Tt, torch.from_numpy(ident).to(dt))
                loss.backward()
                out[case + "_ident_loss_" + dn] = npf(loss)
                out[case + "_ident_grad_" + dn] = npf(Tt.grad)
    np.savez_compressed(os.path.join(HERE, "ntm_ref_losses.npz"), meta=meta("sigma 1.0; *_nbr_* = the reference's own "
                        "knn_point(k+1)[..., 1:] (torch.cdist + topk) in that precision"), **out)


# --------------------------------------------------------------------------------------------- Poly-1 focal losses
def gen_poly1(ns):
    ref_defs(LOSS, ["Poly1FocalLoss", "Poly1FocalLoss_U_corr"], ns)
    out = {}
    B, C, N = 2, 17, 200
    logits = det_values("poly.logits", (B, C, N), 8.0)
    labels = (det_values("poly.labels", (B, N), 1.0, 0.5) * C).astype(np.int64).clip(0, C - 1)
    conf = det_values("poly.conf", (B, N), 1.0, 0.5)
    mask = det_values("poly.mask", (B, N), 1.0, 0.5) > 0.4
    flat_logits = det_values("poly.flat", (300, C), 8.0)
    flat_labels = (det_values("poly.flatlab", (300,), 1.0, 0.5) * C).astype(np.int64).clip(0, C - 1)
    out.update(logits=logits, labels=labels, conf=conf, mask=mask, flat_logits=flat_logits, flat_labels=flat_labels)
    for dn, dt in DTYPES:
        with on_cpu(dt):
            def run(fn, x, *args, **kw):
                xt = torch.from_numpy(x).to(dt).requires_grad_(True)
                loss = fn(xt, *args, **kw)
                loss.backward()
                return npf(loss), npf(xt.grad)
            lab, cf, mk = torch.from_numpy(labels), torch.from_numpy(conf).to(dt), torch.from_numpy(mask)
            cases = {
                "sup_mean": lambda: run(ns["Poly1FocalLoss"](), logits, lab),
                "sup_sum": lambda: run(ns["Poly1FocalLoss"](reduction="sum"), logits, lab),
                "sup_flat": lambda: run(ns["Poly1FocalLoss"](), flat_logits, torch.from_numpy(flat_labels)),
                "sup_eps2_a-1_g3": lambda: run(ns["Poly1FocalLoss"](epsilon=2.0, alpha=-1.0, gamma=3.0), logits, lab),
                "u_t0": lambda: run(ns["Poly1FocalLoss_U_corr"](), logits, lab, cf, thresh=0.0),
                "u_t095": lambda: run(ns["Poly1FocalLoss_U_corr"](), logits, lab, cf, thresh=0.95),
                "u_t07": lambda: run(ns["Poly1FocalLoss_U_corr"](), logits, lab, cf, thresh=0.7),
                "u_mask": lambda: run(ns["Poly1FocalLoss_U_corr"](), logits, lab, cf, thresh=0.5, mask=mk),
            }
            for name, fn in cases.items():
                loss, grad = fn()
                out["%s_loss_%s" % (name, dn)] = loss
                out["%s_grad_%s" % (name, dn)] = grad
    np.savez_compressed(os.path.join(HERE, "poly1_ref.npz"), meta=meta(), **out)


# --------------------------------------------------------------------------------------------- transformer blocks
def _grads(module, names):
    sd = dict(module.named_parameters())
    return {n: npf(sd[n].grad) for n in names}


def gen_blocks(ns):
    ref_defs(TRANSFORMER, ["Mlp", "Attention", "Block", "TransformerEncoder_h", "Encoder"], ns)
    out = {}
    B, L, D, H = 2, 24, 48, 6
    x = det_values("blk.x", (B, L, D), 2.0)
    pos = det_values("blk.pos", (B, L, D), 1.0)
    G = det_values("blk.G", (B, L, D), 2.0)
    groups = det_values("enc.groups", (2, 10, 32, 3), 0.3)
    Genc = det_values("enc.G", (2, 10, 64), 2.0)
    out.update(x=x, pos=pos, G=G, groups=groups, Genc=Genc, dims=np.array([B, L, D, H], np.int32))
    for dn, dt in DTYPES:
        with on_cpu(dt):
            def run(mod, inputs, g, watch=()):
                ins = [torch.from_numpy(a).to(dt).requires_grad_(True) for a in inputs]
                y = mod(*ins)
                ys = y if isinstance(y, (list, tuple)) else [y]
                sum((yy * torch.from_numpy(g).to(dt)).sum() * (i + 1) for i, yy in enumerate(ys)).backward()
                return [npf(yy) for yy in ys], [npf(i.grad) for i in ins], _grads(mod, watch)
            mlp = det_state(ns["Mlp"](D, 4 * D), "blk.mlp.")
            attn = det_state(ns["Attention"](D, num_heads=H), "blk.attn.")
            attn_b = det_state(ns["Attention"](D, num_heads=H, qkv_bias=True), "blk.attnb.")
            block = det_state(ns["Block"](D, H), "blk.block.")
            enc_h = det_state(ns["TransformerEncoder_h"](embed_dim=D, depth=3, num_heads=H, extract_layers=[1, 3]), "blk.ench.")
            encoder = det_state(ns["Encoder"](64), "enc.")
            cases = {
                "mlp": (mlp, [x], G, ["fc1.weight", "fc2.bias"]),
                "attn": (attn, [x], G, ["qkv.weight", "proj.weight"]),
                "attn_bias": (attn_b, [x], G, ["qkv.bias"]),
                "block": (block, [x], G, ["norm1.weight", "norm2.bias", "attn.qkv.weight", "mlp.fc2.weight"]),
                "ench": (enc_h, [x, pos], G, ["blocks.0.attn.proj.weight", "blocks.2.norm2.weight"]),
                "encoder_train": (encoder.train(), [groups], Genc,
                                  ["first_conv.0.weight", "first_conv.1.weight", "first_conv.3.bias",
                                   "second_conv.1.bias", "second_conv.3.weight"]),
            }
            for name, (mod, inputs, g, watch) in cases.items():
                ys, gins, gws = run(mod, inputs, g, watch)
                for i, yy in enumerate(ys):
                    out["%s_y%d_%s" % (name, i, dn)] = yy
                for i, gi in enumerate(gins):
                    out["%s_gin%d_%s" % (name, i, dn)] = gi
                for n, gw in gws.items():
                    out["%s_gw_%s_%s" % (name, n.replace(".", "__"), dn)] = gw
            # BatchNorm running statistics after the one training step above, then the eval-mode forward
            for n in ("first_conv.1.running_mean", "first_conv.1.running_var", "second_conv.1.running_mean",
                      "second_conv.1.running_var"):
                out["encoder_after_%s_%s" % (n.replace(".", "__"), dn)] = npf(encoder.state_dict()[n])
            det_state(encoder, "enc.")
            with torch.no_grad():
                out["encoder_eval_y0_" + dn] = npf(encoder.eval()(torch.from_numpy(groups).to(dt)))
    np.savez_compressed(os.path.join(HERE, "blocks_ref.npz"),
                        meta=meta("parameters are NOT stored: det_init.det_state(module, prefix) rebuilds them "
                                  "(prefixes blk.mlp. blk.attn. blk.attnb. blk.block. blk.ench. enc.)"), **out)


def gen_dgcnn(ns):
    ref_defs(TRANSFORMER, ["DGCNN_Propagation"], ns)
    out = {}
    B, G_, N, Cc, k = 1, 48, 160, 384, 4
    xyz, _ = make_batch(B, N, start_index=70)
    coor_q = np.ascontiguousarray(xyz.transpose(0, 2, 1))
    coor = np.ascontiguousarray(coor_q[:, :, ::3][:, :, :G_])
    f = det_values("dg.f", (B, Cc, G_), 2.0)
    f_q = det_values("dg.fq", (B, Cc, N), 2.0)
    Gout = det_values("dg.G", (B, 384, N), 2.0)
    out.update(coor=coor, f=f, coor_q=coor_q, f_q=f_q, G=Gout, k=np.int32(k))
    for dn, dt in DTYPES:
        with on_cpu(dt):
            mod = det_state(ns["DGCNN_Propagation"](k=k), "dg.")
            ins = [torch.from_numpy(a).to(dt) for a in (coor, f, coor_q, f_q)]
            ins[1].requires_grad_(True)
            ins[3].requires_grad_(True)
            y = mod(*ins)
            (y * torch.from_numpy(Gout).to(dt)).sum().backward()
            feat = mod.get_graph_feature(ins[2], ins[3], ins[0], ins[1])
        out["y_" + dn] = npf(y)
        out["g_f_" + dn] = npf(ins[1].grad)                 # (fp64 too: the referee of the fp32 gradients' tolerance)
        out["g_fq_" + dn] = npf(ins[3].grad)
        if dn == "f32":
            out["graph_feature_slice_" + dn] = npf(feat[:, ::37, ::7, :])      # (B, 2C, N, k) subsampled
        for n in ("layer1.0.weight", "layer1.1.weight", "layer2.1.bias"):
            g = npf(dict(mod.named_parameters())[n].grad)
            out["gw_%s_%s" % (n.replace(".", "__"), dn)] = g.reshape(-1)[::41] if g.size > 4096 else g
    np.savez_compressed(os.path.join(HERE, "dgcnn_ref.npz"),
                        meta=meta("knn_cuda.KNN is absent from the reference tree: SUBSTITUTED by a stand-in with exact fp32 "
                                  "(d2, index) order (SURVEY.md App. A.5). Parameters: det_state(module, 'dg.'). Large weight "
                                  "gradients are stored as flat[::41]."), **out)


def gen_param_groups(ns):
    """openpoints/optim/optim_factory.py get_parameter_groups executed in place on the package's model mirrors (the
    function only walks named_parameters()): which parameter names land in the decay / no-decay AdamW groups."""
    import json
    import logging
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd.ntm import sig_t_mean
    env = dict(ns, json=json, logging=logging)
    ref_defs("openpoints/optim/optim_factory.py", ["get_parameter_groups"], env)
    small = dict(trans_dim=384, depth=2, num_heads=4, group_size=16, num_group=32, encoder_dims=256, nclasses=17,
                 drop_path_rate=0.0, downsample_targets=[256, 128, 64], extract_layers=[1, 2])
    out = {}
    for tag, model in (("seg", PointTransformer_seg_T(**small)), ("pred", sig_t_mean(17))):
        ids = {id(p): n for n, p in model.named_parameters()}
        groups = env["get_parameter_groups"](model, 1e-4, ())
        out[tag + "_n_groups"] = np.int32(len(groups))
        for i, g in enumerate(groups):
            out["%s_g%d_names" % (tag, i)] = np.array([ids[id(p)] for p in g["params"]])
            out["%s_g%d_weight_decay" % (tag, i)] = np.float64(g["weight_decay"])
            out["%s_g%d_lr_scale" % (tag, i)] = np.float64(g["lr_scale"])
    np.savez_compressed(os.path.join(HERE, "param_groups_ref.npz"), meta=meta("weight_decay 1e-4, empty skip list "
                        "(cfgs/tooth_semi/default.yaml:66-68; the model defines no no_weight_decay())"), **out)


if __name__ == "__main__":
    assert os.path.isdir(REF), "run in the build container"
    torch.manual_seed(0)
    space = base_namespace()
    ref_defs(TRANSFORMER, ["sig_t_mean"], space)
    gen_sig_t_mean(space)
    gen_transition(space)
    gen_losses(space)
    gen_poly1(space)
    gen_blocks(space)
    gen_dgcnn(space)
    gen_param_groups(space)
    for f in sorted(os.listdir(HERE)):
        if f.endswith("_ref.npz") or f.startswith("ntm_ref"):
            print("%8.1f KB  %s" % (os.path.getsize(os.path.join(HERE, f)) / 1024, f))
