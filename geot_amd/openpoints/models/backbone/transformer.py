"""The configured backbone ``PointTransformer_seg_T`` as a runnable model over the HIP hot path --
mirror of openpoints/models/backbone/transformer.py: Mlp :16-33, Attention :36-61, Block :64-83,
Encoder :106-136, fps :266-273, Group :275-303, DGCNN_Propagation :305-379, TransformerEncoder_h
:381-410, PointTransformer_seg_T :913-1068 (cfg: cfgs/tooth_semi/transformer_finetune_fixmatch_ntm.yaml:6-15).

Same constructor arguments, forward signature, returned tuple ``(logit, correction, sigma, f_l0)`` and
``state_dict`` keys as the reference, so its checkpoints load unchanged.  Every sampling / grouping /
interpolation step goes through the HIP operators (FPS K1 + gather + kNN in ``Group``; ``pointops.fps``
K2 once, sliced for the three targets; the fused FP front end; the fused EdgeConv graph feature); the
dense layers (1x1 convolutions, Linear, LayerNorm / BatchNorm / GroupNorm, attention) are stock PyTorch
-> rocBLAS / MIOpen, as they are stock PyTorch -> cuBLAS / cuDNN in the reference (out of scope,
SURVEY.md section 2.1 row 12).

Two things are scheduled differently from the reference, neither changes a value:
* the 8192-point FPS needs the coordinates only; it occupies one CU per cloud for milliseconds, so it
  is queued on a second HIP stream beside the patch embedding + the 12 transformer blocks;
* ``dense="factored"`` (default) uses the linearity of the first 1x1 convolution behind a gather:
  EdgeConv  W.[x_k[idx] - x_q ; x_q] = (W_d.x_k)[idx] + ((W_q - W_d).x_q)  (k x fewer GEMM flops, the
  (B,2C,Nq,k) operand is never built), FP module  W.[interp(f) ; skip] = interp(W_a.f) + W_b.skip
  (the interpolation weights sum to 1; the GEMM runs on the m known points instead of the n unknown), and the
  mini-PointNet's  W.[max-pooled ; per-point] = W_g.pooled + W_f.per-point  (the pooled half once per group).
  Same function, fp32 summation order differs (tests: 2e-5 relative); ``dense="reference"`` keeps the
  reference's op order literally.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ....pointops.functions import pointops
from ....pointnet2.pointnet2_modules import PointnetFPModule
from ....pointnet2 import pointnet2_utils as pt_utils
from ....pointnet2.pytorch_utils import PointwiseConv1d, PointwiseConv2d, pointwise, batch_norm_nd, shared_mlp_nd
from ....knn_cuda import KNN, knn_sorted
from .transformer_ops import (Group, fps, fps_downsample, graph_feature, get_graph_feature_unfused,  # noqa: F401
                              edgeconv_tail, edgeconv_tail_eligible, edgeconv_reverse_index)
from .... import streams
from ....ntm import sig_t_mean  # noqa: F401  (transformer.py:1099-1131 lives in ntm.py)
from ....fused_norm import bn_act, fp_front, fp_front_eligible, max_last, add_last_broadcast, thin_mm, add_channel_bias
from ....fused_norm import (fp_front_cl, fp_front_cl_eligible, bn_act_cl, fp_stage_cl, pointwise_to_cl, pointwise_from_cl,
                            local_spatial_order, ReverseIndex)
from ....fused_norm import linear as lean_linear, res_ln, res_ln_eligible, qkv_split, softmax_last


class DropPath(nn.Module):
    """timm.models.layers.DropPath (stochastic depth per sample), which the reference imports (:4)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def scale(self, x):
        """The per-sample factor (B, 1, ...): Bernoulli(keep) / keep, or None when nothing is dropped."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        return x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep).div_(keep)

    def forward(self, x):
        mask = self.scale(x)
        return x if mask is None else x * mask


def _residual(x, branch, drop_path):
    """x + drop_path(branch) with the mask multiply and the add as one launch (torch.addcmul)."""
    mask = drop_path.scale(branch) if isinstance(drop_path, DropPath) else None
    return x + branch if mask is None else torch.addcmul(x, branch, mask)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.lean = False            # set by PointTransformer_seg_T when dense != "reference"
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        if self.lean:
            return self.drop(lean_linear(self.fc2, self.drop(self.act(lean_linear(self.fc1, x)))))
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.fused = os.environ.get("GEOT_ATTN", "manual") == "sdpa"   # measured in the full step: sdpa 37.9 ms, manual 36.5
        self.lean = False            # set by PointTransformer_seg_T when dense != "reference"

    def forward(self, x):
        B, N, C = x.shape
        H, d = self.num_heads, C // self.num_heads
        if self.lean and self.fused and x.is_cuda and (self.attn_drop.p == 0.0 or not self.training):
            q, k, v = self.qkv(x).view(B, N, 3, H, d).permute(2, 0, 3, 1, 4).contiguous().unbind(0)
            x = F.scaled_dot_product_attention(q, k, v, scale=self.scale).transpose(1, 2).reshape(B, N, C)
            return self.proj_drop(lean_linear(self.proj, x))
        if self.lean and (self.attn_drop.p == 0.0 or not self.training):
            # The same function in fewer launches (26 -> 12 per block forward + backward at 512 tokens, where every
            # launch is ~5 us of a 40 us GEMM neighbourhood): q, k, v leave ONE permuted copy of the projection as
            # contiguous (B*H, N, d) views (their gradients come back through one stack instead of three
            # zero-fill + copy + add chains), and the 1/sqrt(d) rides in the GEMM (baddbmm's alpha) instead of an
            # element-wise pass over the (B, H, N, N) scores each way.
            qkv = self.qkv(x)
            split = qkv_split(qkv, H, self.scale)          # one launch each way, the softmax scale folded into q
            if split is not None:
                q, k, v = split
                attn = softmax_last(torch.bmm(q, k.transpose(1, 2)))
            else:
                q, k, v = qkv.view(B, N, 3, H, d).permute(2, 0, 3, 1, 4).contiguous().view(3, B * H, N, d).unbind(0)
                attn = torch.baddbmm(q.new_empty(()), q, k.transpose(1, 2), beta=0.0, alpha=self.scale).softmax(dim=-1)
            x = torch.bmm(attn, v).view(B, H, N, d).transpose(1, 2).reshape(B, N, C)
            return self.proj_drop(lean_linear(self.proj, x))
        qkv = self.qkv(x).reshape(B, N, 3, H, d).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        if self.fused and x.is_cuda and (self.attn_drop.p == 0.0 or not self.training):
            # softmax(q k^T scale) v as one kernel (torch's memory-efficient attention runs fp32 on gfx950): same
            # function (1e-6 apart); 425 -> 348 us per block fwd + bwd in isolation, but slower inside the step: opt-in
            x = F.scaled_dot_product_attention(q, k, v, scale=self.scale)
        else:
            attn = (q @ k.transpose(-2, -1)) * self.scale
            attn = self.attn_drop(attn.softmax(dim=-1))
            x = attn @ v
        x = x.transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)

    def forward(self, x):
        if self.attn.lean:
            x = _residual(x, self.attn(self.norm1(x)), self.drop_path)
            return _residual(x, self.mlp(self.norm2(x)), self.drop_path)
        x = x + self.drop_path(self.attn(self.norm1(x)))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class TransformerEncoder_h(nn.Module):
    """Plain (non-hierarchical) encoder returning the outputs of ``extract_layers`` (transformer.py:381-410)."""

    def __init__(self, embed_dim=768, depth=4, num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None,
                 drop_rate=0., attn_drop_rate=0., drop_path_rate=0., finetune=False, extract_layers=None):
        super().__init__()
        self.finetune = finetune
        self.extract_layers = extract_layers
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate,
                  drop_path=drop_path_rate[i] if isinstance(drop_path_rate, list) else drop_path_rate)
            for i in range(depth)])

    def forward(self, x, pos):
        if len(self.blocks) and self.blocks[0].attn.lean and res_ln_eligible(x, self.blocks[0].norm1):
            return self._forward_fused_norms(x, pos)
        inter_feats = []
        for i, block in enumerate(self.blocks):
            x = block(x + pos)
            if self.extract_layers is not None and i + 1 in self.extract_layers:
                inter_feats.append(x)
        return inter_feats if self.extract_layers is not None else x

    def _forward_fused_norms(self, x, pos):
        """The same blocks with every residual add fused into the LayerNorm behind it (fused_norm.res_ln): the
        position embedding and the previous block's MLP branch enter the next block's norm1 in one pass, the attention
        branch enters norm2 in one pass; a block's output is materialised only where it is extracted."""
        factors = self._drop_path_factors(x)
        inter_feats = []
        pending = None                                 # (a1, mlp branch, drop-path factors) of the previous block
        for i, blk in enumerate(self.blocks):
            f_att, f_mlp = factors[i]
            if pending is None:
                a0, n1 = res_ln(x, None, None, pos, blk.norm1)
            else:
                a0, n1 = res_ln(pending[0], pending[1], pending[2], pos, blk.norm1)
            att = blk.attn(n1)
            a1, n2 = res_ln(a0, att, f_att, None, blk.norm2)
            m = blk.mlp(n2)
            wanted = self.extract_layers is not None and i + 1 in self.extract_layers
            if wanted or i + 1 == len(self.blocks):
                x = a1 + m if f_mlp is None else torch.addcmul(a1, m, f_mlp)
                pending = None
                if wanted:
                    inter_feats.append(x)
            else:
                pending = (a1, m, f_mlp)
        return inter_feats if self.extract_layers is not None else x

    def _drop_path_factors(self, x):
        """[(attention branch's factor, MLP branch's factor)] per block: Bernoulli(keep) / keep per sample, (B, 1, 1), or None where
        nothing is dropped (eval, rate 0) -- what DropPath.scale draws, but ALL of a forward's factors in one bernoulli launch and
        one division instead of two tiny launches per branch (44 per forward at depth 12: 0.2 ms of launch latency on the main
        stream).  Same distribution per (branch, sample); the random stream is consumed in another order than branch by branch,
        which the reference-order path (Block.forward) keeps."""
        probs = [blk.drop_path.drop_prob if isinstance(blk.drop_path, DropPath) else 0.0 for blk in self.blocks]
        if not self.training or not any(probs):
            return [(None, None)] * len(self.blocks)
        live = [i for i, p in enumerate(probs) if p > 0.0]
        keep = self.__dict__.get("_dp_keep")
        if keep is None or keep.device != x.device or keep.shape[0] != 2 * len(live):
            keep = torch.tensor([1.0 - probs[i] for i in live for _ in (0, 1)], dtype=torch.float32, device=x.device).view(-1, 1)
            self.__dict__["_dp_keep"] = keep                       # (2 x live blocks, 1): built once per device, outside any capture
        f = (torch.bernoulli(keep.expand(-1, x.shape[0])) / keep).to(x.dtype)     # (2 x live, B)
        shape = (x.shape[0],) + (1,) * (x.dim() - 1)
        out = [(None, None)] * len(self.blocks)
        for j, i in enumerate(live):
            out[i] = (f[2 * j].view(shape), f[2 * j + 1].view(shape))
        return out


class Encoder(nn.Module):
    """Mini-PointNet over each group of points (transformer.py:106-136): (B,G,n,3) -> (B,G,C).

    The groups arrive channels-last, (B*G*n, 3); the reference transposes them to (B*G, 3, n) for Conv1d.  A 1x1
    convolution over (BG, C, n) is a Linear over the BG*n points and BatchNorm1d's statistics over (BG, n) are its
    statistics over those points, so the stack runs on 2-D tensors -- row-major (L, C) in the reference op order
    (``factored=False``), column-major (C, L) in the default mode (see _forward_channels_first) -- as four plain
    GEMMs without per-group batching; same modules, parameters and running statistics either way."""

    def __init__(self, encoder_channel, factored=False):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.factored = factored
        self.first_conv = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, self.encoder_channel, 1))

    @staticmethod
    def _rows(seq, x):
        for m in seq:
            x = F.linear(x, m.weight.squeeze(-1), m.bias) if isinstance(m, nn.Conv1d) else m(x)
        return x

    def _forward_channels_first(self, point_groups):
        """The same stack on (C, L) tensors, L = B*G*n columns: GEMMs W @ X, BatchNorm over the columns through
        the (1, C, L) spatial kernel (MIOpen: ~6 TB/s; torch's kernel for (L, C) rows reaches 0.5 TB/s here: 2.9 ms of
        the step), and  W.[pooled ; per-point] = W_g.pooled + W_f.per-point  (the pooled half once per group: half of
        that layer's GEMM, and the (L, 512) concatenation is never built)."""
        bs, g, n, _ = point_groups.shape
        L = bs * g * n

        def conv(m, x, w=None, bias=True):
            w = m.weight.squeeze(-1) if w is None else w
            y = thin_mm(w, x) if x.shape[0] <= 8 else torch.mm(w, x)
            return y if m.bias is None or not bias else y + m.bias.unsqueeze(1)

        def norm_act(seq, x, pre_bias):   # BatchNorm1d -> ReLU of a Sequential on (C, L); x arrives WITHOUT pre_bias,
            return bn_act(seq[1], x.unsqueeze(0), relu=True, pre_bias=pre_bias).squeeze(0)     # bn_act accounts for it

        # No bias is ever added to an (C, L) tensor: a per-channel constant commutes with the max over a group's points
        # (exactly: rounding is monotone), passes through the next convolution as the constant W b, and in front of a
        # BatchNorm goes through bn_act's pre_bias (cancels under batch statistics).  f0, h0 = the tensors without it.
        x = point_groups.reshape(L, 3).t()                                              # (3, L) view
        b1, b2 = self.first_conv[3].bias, self.second_conv[3].bias
        f0 = conv(self.first_conv[3], norm_act(self.first_conv, conv(self.first_conv[0], x, bias=False),
                                               self.first_conv[0].bias), bias=False)    # f = f0 + b1: (256, L)
        c1 = f0.shape[0]
        pooled = max_last(f0.view(c1, bs * g, n))                                       # (256, BG), + b1 below
        if b1 is not None:
            pooled = add_channel_bias(pooled, b1)
        c2 = self.second_conv[0]
        w = c2.weight.squeeze(-1)
        pre = c2.bias
        w_g, w_f = torch.split(w, [c1, w.shape[1] - c1], dim=1)  # (pooled | per-point) columns: one split, one concatenation back
        if b1 is not None:                                     # W_f (f0 + b1) = W_f f0 + W_f b1
            wb = torch.mv(w_f, b1)
            # (pre * 1.0: the two addends must not receive the SAME gradient tensor -- AccumulateGrad clones a shared one with a
            # device-to-device memcpy, a memcpy node when the step is captured; x1.0 is exact)
            pre = wb if pre is None else pre * 1.0 + wb
        h = add_last_broadcast(conv(c2, f0, w_f, bias=False).view(-1, bs * g, n), torch.mm(w_g, pooled))
        h0 = conv(self.second_conv[3], norm_act(self.second_conv, h.view(-1, L), pre), bias=False)  # (C_enc, L)
        out = max_last(h0.view(-1, bs * g, n))
        if b2 is not None:
            out = add_channel_bias(out, b2)
        return out.t().reshape(bs, g, self.encoder_channel)

    def forward(self, point_groups):
        if self.factored:
            return self._forward_channels_first(point_groups)
        bs, g, n, _ = point_groups.shape
        feature = self._rows(self.first_conv, point_groups.reshape(bs * g * n, 3)).view(bs * g, n, -1)
        feature_global = torch.max(feature, dim=1, keepdim=True)[0]                      # (BG, 1, 256)
        feature = torch.cat([feature_global.expand(-1, n, -1), feature], dim=2)          # (BG, n, 512)
        feature = self._rows(self.second_conv, feature.reshape(bs * g * n, -1)).view(bs * g, n, -1)
        return torch.max(feature, dim=1, keepdim=False)[0].reshape(bs, g, self.encoder_channel)


def _knn_idx(coor_q, coor_k, k):
    """coor (B,3,N) channel-first -> int32 (B,Nq,k) neighbour ids of every query among the keys."""
    with torch.no_grad():
        return knn_sorted(coor_q.transpose(1, 2).contiguous().float(), coor_k.transpose(1, 2).contiguous().float(), k)[1]


class DGCNN_Propagation(nn.Module):
    """EdgeConv up-sampling (transformer.py:305-379): two rounds of kNN graph feature -> 1x1 Conv2d ->
    GroupNorm -> LeakyReLU -> max over the k neighbours."""

    def __init__(self, k=16, dense="factored"):
        super().__init__()
        self.k = k
        self.knn = KNN(k=k, transpose_mode=False)
        self.dense = dense
        self.fused_tail = os.environ.get("GEOT_EDGE_TAIL", "fused") == "fused"
        self.layer1 = nn.Sequential(PointwiseConv2d(768, 512, kernel_size=1, bias=False), nn.GroupNorm(4, 512),
                                    nn.LeakyReLU(negative_slope=0.2))
        self.layer2 = nn.Sequential(PointwiseConv2d(1024, 384, kernel_size=1, bias=False), nn.GroupNorm(4, 384),
                                    nn.LeakyReLU(negative_slope=0.2))

    fps_downsample = staticmethod(fps_downsample)

    def get_graph_feature(self, coor_q, x_q, coor_k, x_k):
        """(B, 2C, Nq, k) = cat(x_k[nbr] - x_q, x_q), one fused kernel (transformer.py:343-364)."""
        return graph_feature(x_q, x_k, _knn_idx(coor_q, coor_k, self.k))

    def _edge(self, layer, coor_q, x_q, coor_k, x_k, idx=None):
        """idx: the kNN ids (B,Nq,k), or the pair (ids, reverse index of the ids) when the caller built both ahead."""
        rix = None
        if isinstance(idx, (tuple, list)):
            idx, rix = idx
        conv, norm, act = layer[0], layer[1], layer[2]
        if self.dense != "factored":
            y = conv(self.get_graph_feature(coor_q, x_q, coor_k, x_k))
        else:
            c = x_q.shape[1]
            w = conv.weight.view(conv.out_channels, 2 * c)
            w_d, w_q = torch.split(w, [c, c], dim=1)                     # (one split: its backward is one concatenation)
            if idx is None:      # (the model hands in the ids it searched on its side stream, see _index_plan)
                idx = _knn_idx(coor_q, coor_k, self.k)
            p = pointwise(w_d, x_k)                                      # (B, Cout, Nk)
            q = pointwise(w_q - w_d, x_q)                                # (B, Cout, Nq)
            if (self.fused_tail and p.is_cuda and isinstance(norm, nn.GroupNorm) and norm.affine and isinstance(act, nn.LeakyReLU)
                    and edgeconv_tail_eligible(p.shape[0], p.shape[1], q.shape[2], p.shape[2], self.k, norm.num_groups)):
                return edgeconv_tail(p, q, idx, norm, act.negative_slope, rix)   # gather + GN + LeakyReLU + max, fused
            y = pt_utils.grouping_operation(p.contiguous(), idx) + q.unsqueeze(-1)
        return act(norm(y)).max(dim=-1, keepdim=False)[0]

    def forward(self, coor, f, coor_q, f_q, idx=None):
        """coor, f: (B,3,G), (B,C,G) source; coor_q, f_q: (B,3,N), (B,C,N) target.  idx (optional): the two kNN id
        tensors (queries among the sources, queries among themselves), when the caller has searched them already."""
        i1, i2 = idx if idx is not None else (None, None)
        f_q = self._edge(self.layer1, coor_q, f_q, coor, f, i1)
        return self._edge(self.layer2, coor_q, f_q, coor_q, f_q, i2)


def _fp_factored(fp, unknown, known, unknow_feats, known_feats, nn3=None, layout="cf"):
    """forward of a PointnetFPModule (pointnet2_modules.py:597-642) with the first 1x1 convolution moved in
    front of the interpolation (see the module docstring); the parameters are ``fp``'s own.  nn3 (optional): the
    (idx, weight) pair of three_nn + the inverse-distance weights, when the caller has computed them already -- for the
    point-major layout optionally followed by (order, rix): the Morton sequence of the unknown points
    (fused_norm.local_spatial_order) and the ReverseIndex of (idx, weight) for the gradient.
    layout "cl": the first stage runs on point-major (B, n, C) activations where the layer is wide enough."""
    layers = list(fp.mlp.children())
    first = layers[0]
    conv = first.conv
    c = known_feats.shape[1]
    w = conv.weight.view(conv.out_channels, -1)
    # the two column blocks of the first convolution (known features | skip features) as ONE split: its backward is a single
    # concatenation, where two slices cost two zero-fills, two copies and an add per step
    w_known, w_skip = torch.split(w, [c, w.shape[1] - c], dim=1) if w.shape[1] > c else (w, None)
    order = rix = None
    if nn3 is None:
        dist2, idx = pt_utils._ext.three_nn(unknown.contiguous(), known.contiguous())
        weight = None
    else:
        idx, weight = nn3[:2]
        if len(nn3) > 2:
            order, rix = nn3[2:]
    has_bn = any(name == "bn" for name, _ in first.named_children())
    post_act = next(iter(first.named_children()))[0] == "conv"            # conv -> BatchNorm -> ReLU order
    if (layout == "cl" and conv.bias is None and has_bn and post_act and len(layers) > 1
            and [n for n, _ in first.named_children()][:2] == ["conv", "bn"]
            and all(n in ("conv", "bn") or isinstance(mod, nn.ReLU) for n, mod in first.named_children())
            and next(iter(layers[1].named_children()))[0] == "conv" and layers[1].conv.kernel_size in ((1,), (1, 1))
            and known_feats.is_cuda and known_feats.dtype == torch.float32
            and fp_front_cl_eligible(known_feats.new_empty((known_feats.shape[0], conv.out_channels, 0)), unknow_feats)):
        # the first stage on point-major activations (csrc/channels_last.hip): GEMM -> (B, m, C), interpolation + skip
        # + BatchNorm sums, BatchNorm + ReLU, and the second stage's convolution reads (B, n, C) as a transposed operand
        if weight is None:
            weight = pt_utils._ext.fp_weights(dist2)
        a_cl = pointwise_to_cl(w_known, known_feats)
        relu = any(isinstance(mod, nn.ReLU) for _, mod in first.named_children())
        wb = None if unknow_feats is None else w_skip
        if os.environ.get("GEOT_FP_CL_FUSED", "1") != "0":   # one node; its backward never writes the BatchNorm's input gradient
            z_cl = fp_stage_cl(first.bn.bn, a_cl, idx, weight, unknow_feats, wb, relu, order, rix)
        else:
            y_cl, partial = fp_front_cl(a_cl, idx, weight, unknow_feats, wb, order, rix)
            z_cl = bn_act_cl(first.bn.bn, y_cl, relu=relu, partial=partial)
        conv2 = layers[1].conv
        y2 = pointwise_from_cl(conv2.weight.view(conv2.out_channels, -1), z_cl)
        if conv2.bias is not None:
            y2 = y2 + conv2.bias.view(1, -1, 1)
        return shared_mlp_nd(layers[1:], y2, first_conv_done=True)
    a = pointwise(w_known, known_feats)                                  # (B, Cout, m): GEMM on the known points
    if conv.bias is None and has_bn and post_act and fp_front_eligible(a, unknow_feats):
        # interpolation + skip channels + the BatchNorm sums in one kernel, then BatchNorm + ReLU in one pass
        if weight is None:
            weight = pt_utils._ext.fp_weights(dist2)
        y, partial = fp_front(a, idx, weight, unknow_feats, None if unknow_feats is None else w_skip)
        relu = any(isinstance(mod, nn.ReLU) for _, mod in first.named_children())
        y = bn_act(first.bn.bn, y, relu=relu, partial=partial)
        for name, mod in first.named_children():
            if name not in ("conv", "bn") and not isinstance(mod, nn.ReLU):
                y = mod(y)
        return shared_mlp_nd(layers[1:], y)
    y = pt_utils.three_interpolate(a, idx, weight if weight is not None else pt_utils._ext.fp_weights(dist2))
    if unknow_feats is not None:
        y = y + pointwise(w_skip, unknow_feats)
    if conv.bias is not None:
        y = y + conv.bias.view(1, -1, 1)
    return _rest_of_stage(first, layers, y)


def _rest_of_stage(first, layers, y):
    for name, mod in first.named_children():
        if name != "conv":
            y = batch_norm_nd(mod.bn, y) if name == "bn" else mod(y)
    return shared_mlp_nd(layers[1:], y)


class PointTransformer_seg_T(nn.Module):
    def __init__(self, trans_dim, depth, drop_path_rate, nclasses, num_heads, group_size, num_group,
                 downsample_targets, extract_layers, encoder_dims, dense=None, overlap=True, **kwargs):
        super().__init__()
        self.trans_dim = trans_dim
        self.depth = depth
        self.drop_path_rate = drop_path_rate
        self.nclasses = nclasses
        self.num_heads = num_heads
        self.group_size = group_size
        self.num_group = num_group
        self.downsample_targets = downsample_targets
        self.dense = dense or os.environ.get("GEOT_DENSE", "factored")
        # layout of the wide first stage of the FP modules (dense == "factored" only): "cl" = point-major (B, N, C)
        # activations between two GEMMs (csrc/channels_last.hip), "cf" = the reference's (B, C, N) throughout
        self.fp_layout = os.environ.get("GEOT_FP_LAYOUT", "cl")
        self.overlap = overlap
        self._side = {}
        self.at_blocks_backward = None    # one-shot callback of the training step (set before forward, see _forward)
        self.cut_at_blocks = False        # detach the decoder from the blocks for a two-phase backward (see _forward, take_cut)
        self._cut = None

        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size)
        self.encoder_dims = encoder_dims
        self.encoder = Encoder(encoder_channel=self.encoder_dims, factored=self.dense == "factored")
        self.reduce_dim = nn.Identity()
        if self.encoder_dims != self.trans_dim:
            self.reduce_dim = nn.Linear(self.encoder_dims, self.trans_dim)
        self.extract_layers = extract_layers
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.trans_dim))
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder_h(embed_dim=self.trans_dim, depth=self.depth, drop_path_rate=dpr,
                                           num_heads=self.num_heads, finetune=True,
                                           extract_layers=self.extract_layers)
        self.norm = nn.LayerNorm(self.trans_dim)
        for blk in self.blocks.blocks:
            blk.attn.lean = blk.mlp.lean = self.dense != "reference"

        self.propogation_2 = PointnetFPModule([self.trans_dim + 3, self.trans_dim * 4, self.trans_dim])
        self.propogation_1 = PointnetFPModule([self.trans_dim + 3, self.trans_dim * 4, self.trans_dim])
        self.propogation_0 = PointnetFPModule([self.trans_dim + 3 + 2, self.trans_dim * 4, self.trans_dim])
        self.dgcnn_pro_1 = DGCNN_Propagation(k=4, dense=self.dense)
        self.dgcnn_pro_2 = DGCNN_Propagation(k=4, dense=self.dense)
        self.seg_head = nn.Sequential(PointwiseConv1d(self.trans_dim, 128, 1), nn.BatchNorm1d(128), nn.Dropout(0.5),
                                      PointwiseConv1d(128, self.nclasses, 1))
        self.apply(self._init_weights)

        self.T_revision = nn.Linear(self.nclasses, self.nclasses, False)
        nn.init.constant_(self.T_revision.weight, 0.0)
        self.T_linear = nn.Linear(self.nclasses, self.nclasses, False)
        nn.init.constant_(self.T_linear.weight, 0.0)
        self.sigma = nn.Parameter(torch.Tensor(self.nclasses), requires_grad=True)
        nn.init.constant_(self.sigma, 0.4)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)
        elif isinstance(m, (nn.Conv1d, nn.Conv2d)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def take_cut(self):
        """(the blocks' outputs, their detached copies the decoder consumed) of the last forward under cut_at_blocks, once."""
        cut, self._cut = self._cut, None
        return cut

    def _side_stream(self, device):
        key = str(device)
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device=device)
        return self._side[key]

    def _fp(self, module, unknown, known, unknow_feats, known_feats, nn3=None):
        if self.dense == "factored":
            return _fp_factored(module, unknown, known, unknow_feats, known_feats, nn3, layout=self.fp_layout)
        return module(unknown, known, unknow_feats, known_feats)

    @torch.no_grad()
    def _index_plan(self, pts, center):
        """Everything the decoder needs that depends on the COORDINATES only: the sampled clouds, the three_nn ids +
        inverse-distance weights of the three FP modules and the four kNN graphs of the EdgeConv stages (the same calls,
        in the same order, the modules would make themselves).  The model runs this on its side stream behind the long
        FPS, beside the transformer blocks: 0.73 ms of 8-to-100-CU kernels leave the critical path of the step."""
        center_pts = [pointops.fps(pts, t) for t in self.downsample_targets]
        trans = [pt.transpose(-1, -2).contiguous() for pt in center_pts]
        center_trans = center.transpose(-1, -2).contiguous()

        def nn3(unknown, known):
            dist2, idx = pt_utils._ext.three_nn(unknown.contiguous(), known.contiguous())
            weight = pt_utils._ext.fp_weights(dist2)
            if self.fp_layout != "cl":
                return idx, weight
            # point-major FP stages: in training the reverse index of the gradient; where the table of known points is
            # larger than an XCD's L2 (4 MB: m > 2048 rows at 384 channels in, 1536 out) also the Morton sequences in which
            # the forward takes its points and the gradient its targets -- a small table is L2-resident in any order
            # (prop1 / prop2: 80 us in memory order, 85 in Morton order), and the orders are not free: they run beside the
            # transformer blocks
            big = known.shape[1] > 2048
            # the gradient's targets (= the known points) in Morton order for every stage: its workgroups are dealt consecutive
            # lists inside an XCD (gather_group.hip gr_deal), which only pays when consecutive lists share source rows
            tgt_order = big or os.environ.get("GEOT_FP_RIX_ORDER", "all") == "all"
            rix = ReverseIndex(idx, weight, known.shape[1], local_spatial_order(known) if tgt_order else None) if self.training else None
            return idx, weight, local_spatial_order(unknown) if big else None, rix
        k2, k1 = self.dgcnn_pro_2.k, self.dgcnn_pro_1.k

        def graph(coor_q, coor_k, k):
            # the kNN ids of an EdgeConv layer and, in training, the reverse index its fused gradient walks (7 small launches
            # per layer that would otherwise sit on the backward's critical path)
            idx = _knn_idx(coor_q, coor_k, k)
            if not (self.training and self.dgcnn_pro_1.fused_tail):
                return idx
            return idx, edgeconv_reverse_index(idx, coor_k.shape[2])
        return {"center_pts": center_pts, "center_pts_trans": trans, "center_trans": center_trans,
                "fp2": nn3(center_pts[1], center), "fp1": nn3(center_pts[0], center), "fp0": nn3(pts, center_pts[0]),
                "dg2": (graph(trans[1], center_trans, k2), graph(trans[1], trans[1], k2)),
                "dg1": (graph(trans[0], trans[1], k1), graph(trans[0], trans[0], k1))}

    def forward(self, pts, x=None, cls_label=None, T=None, geometry=None):
        with pointops.fps_prefix_scope():       # the three pointops.fps targets are prefixes of one FPS run
            return self._forward(pts, x, cls_label, T, geometry)

    @torch.no_grad()
    def prefetch_geometry(self, pts, inline=False):
        """Queue everything forward() derives from the COORDINATES of a batch -- Group (512-sample FPS, kNN, the
        neighbourhoods), the 8192-sample FPS and the index plan -- on the side stream, for a batch that will be passed
        to forward(pts, ..., geometry=<the result>) later: a training loop calls this with batch k + 1 between the
        forward and the backward of batch k, so that ~6.5 ms of few-workgroup kernels run beside the GEMM-bound backward
        instead of at the head of the next step and beside its (shorter) encoder.  GEOT_LOOKAHEAD=group queues Group only.  Same kernels on the same inputs: the
        results are those forward() would compute itself.  None when the model cannot use it (CPU, overlap off)."""
        if inline:
            # the same work on the CURRENT stream, handed back as a geometry the caller vouches for ("static"): graph_step.py
            # captures it as a single-stream graph and replays that beside the training graph
            if not (pts.is_cuda and self.dense == "factored"):
                return None
            pts = pts.contiguous()
            with pointops.fps_prefix_scope():
                group = self.group_divider(pts)
                pointops.fps_indices(pts, max(self.downsample_targets))
                plan = self._index_plan(pts, group[1])
            return {"pts": pts, "version": None, "group": group, "grouped": None, "plan": plan, "training": self.training,
                    "fp_layout": self.fp_layout, "static": True}
        if not (pts.is_cuda and self.overlap and self.dense == "factored" and streams.may_fork(pts.device)):
            return None
        pts = pts.contiguous()
        dev = pts.device
        side, main = self._side_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(main)            # pts is ready; every side-stream allocation starts behind main's earlier uses
        with torch.cuda.stream(side), pointops.fps_prefix_scope():
            group = self.group_divider(pts)
            grouped = streams.event()
            grouped.record(side)
            plan = None
            if os.environ.get("GEOT_LOOKAHEAD", "all") == "all":
                # also the 8192-sample FPS and the index plan.  Beside the next batch's encoder + 12 blocks (~5 ms of main-stream
                # work) those ~6 ms of side-stream work ARE the critical path: the decoder waits for them (measured: queueing
                # only Group gains nothing, 33.94 vs 33.97 ms; queueing everything 33.64).  Beside the backward's GEMMs the
                # FPS launch itself takes 5.6 instead of 4.7 ms, but nothing waits for it
                pointops.fps_indices(pts, max(self.downsample_targets))
                plan = self._index_plan(pts, group[1])
            ready = streams.event()          # everything above is done: what a consumer OTHER than this model waits for
            ready.record(side)               # (this model's own forward waits for its side stream; slice_geometry's user cannot)
        return {"pts": pts, "version": pts._version, "group": group, "grouped": grouped, "plan": plan,
                "training": self.training, "fp_layout": self.fp_layout, "ready": ready}

    def _forward(self, pts, x, cls_label, T, geometry=None):
        B, N, _ = pts.shape
        pts = pts.contiguous()
        side = self._side_stream(pts.device) if (self.overlap and pts.is_cuda and streams.may_fork(pts.device)) else None
        top = max(self.downsample_targets)
        plan = ready = None
        forked = False                    # has this forward queued work on the side stream (and must wait for it)?
        if (geometry is not None and geometry["training"] == self.training and geometry["fp_layout"] == self.fp_layout
                and (geometry.get("static") or (side is not None and geometry["pts"] is pts and geometry["version"] == pts._version))):
            # the coordinate-only work of this batch was queued on the side stream earlier (prefetch_geometry); "static": the
            # caller holds the geometry in fixed buffers it refills itself and vouches for it (graph_step.py: a hipGraph replay
            # has no tensor identities to check -- the wrapper checks them on the host before it replays)
            main = torch.cuda.current_stream(pts.device)
            if geometry.get("grouped") is not None:
                main.wait_event(geometry["grouped"])
            ready = geometry.get("ready")
            forked = not geometry.get("static")   # a queued geometry's plan is still on the side stream; a static one is memory
            neighborhood, center, idx = geometry["group"]
            plan = geometry["plan"]
            if plan is None and side is not None:   # the long FPS and the index plan beside this batch's encoder, as without look-ahead
                forked = True
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    pointops.fps_indices(pts, top)
                    if os.environ.get("GEOT_INDEX_PLAN", "side") == "side":
                        plan = self._index_plan(pts, center)
        else:
            # the long FPS (largest target; the shorter ones are prefixes, pointops.fps_indices) beside the encoder
            if side is not None:
                forked = True
                main = torch.cuda.current_stream(pts.device)
                side.wait_stream(main)        # every side-stream allocation starts behind all earlier main-stream uses
                with torch.cuda.stream(side):
                    pointops.fps_indices(pts, top)
            neighborhood, center, idx = self.group_divider(pts)
            if side is not None and self.dense == "factored" and os.environ.get("GEOT_INDEX_PLAN", "side") == "side":
                grouped = streams.event()
                grouped.record(main)             # `center` (the 512 group centres) is the one input the plan needs from main
                with torch.cuda.stream(side):
                    side.wait_event(grouped)
                    plan = self._index_plan(pts, center)
        if self.dense != "reference":
            # the three Linear layers in front of the blocks through fused_norm.linear: the same GEMMs, the bias gradients as
            # fixed-order column sums (aten::sum over 4096 rows zeroes a semaphore buffer by memset: a memset node when captured)
            tokens = self.encoder(neighborhood)
            group_input_tokens = lean_linear(self.reduce_dim, tokens) if isinstance(self.reduce_dim, nn.Linear) else self.reduce_dim(tokens)
            pos = lean_linear(self.pos_embed[2], self.pos_embed[1](lean_linear(self.pos_embed[0], center)))
        else:
            group_input_tokens = self.reduce_dim(self.encoder(neighborhood))
            pos = self.pos_embed(center)
        inter_feats = self.blocks(group_input_tokens, pos)
        if self.at_blocks_backward is not None and inter_feats[-1].requires_grad:
            # look-ahead, timed: the training step's callback (it queues the next batch's coordinate-only work,
            # prefetch_geometry) runs when the backward REACHES the transformer blocks -- the gradient of their last output
            # exists, the decoder's backward is done.  Beside the decoder's wide GEMMs the 8192-sample FPS takes 5.6 ms and
            # slows them; beside the blocks' small GEMMs it takes its 4.7 ms (33.27 vs 33.61 ms per step)
            cb, self.at_blocks_backward = self.at_blocks_backward, None

            def run_callback(grad, cb=cb):
                cb()
                return grad
            inter_feats[-1].register_hook(run_callback)
        if self.cut_at_blocks and torch.is_grad_enabled() and inter_feats[-1].requires_grad:
            # two-phase backward (graph_step's split capture): the decoder works on detached copies of the blocks' outputs;
            # loss.backward() then stops there with the outputs' gradients, and take_cut() hands back what the second phase
            # needs -- torch.autograd.backward(outputs, gradients).  The decoder's contribution reaches each output first in
            # either form, so the sums are the same bits as in one backward.
            cut_src = list(inter_feats)
            inter_feats = [t.detach().requires_grad_(True) for t in cut_src]
            self._cut = (cut_src, inter_feats)
        inter_feats = [self.norm(t).transpose(-1, -2).contiguous() for t in inter_feats]
        cls_label_one_hot = F.one_hot(cls_label, 2).transpose(1, 2).float().repeat(1, 1, N)

        center_original = pts
        center_trans = center.transpose(-1, -2).contiguous()
        f_l0 = torch.cat([cls_label_one_hot, center_original.transpose(-1, -2).contiguous()], 1)

        assert len(inter_feats) == len(self.downsample_targets), \
            "the length of the cardinality and the features should be the same"
        if forked:
            main.wait_stream(side)
            if ready is not None:         # a geometry queued by ANOTHER model (slice_geometry): its producer's stream, not ours
                main.wait_event(ready)
        if plan is not None:
            center_pts, center_pts_trans = plan["center_pts"], plan["center_pts_trans"]
        else:
            plan = {}
            center_pts = [pointops.fps(pts, t) for t in self.downsample_targets]
            center_pts_trans = [pt.transpose(-1, -2).contiguous() for pt in center_pts]

        f_l3 = inter_feats[2]
        f_l2 = self._fp(self.propogation_2, center_pts[1], center, center_pts_trans[1], inter_feats[1], plan.get("fp2"))
        f_l1 = self._fp(self.propogation_1, center_pts[0], center, center_pts_trans[0], inter_feats[0], plan.get("fp1"))
        if plan.get("dg2") is not None:
            f_l2 = self.dgcnn_pro_2(center_trans, f_l3, center_pts_trans[1], f_l2, plan["dg2"])
            f_l1 = self.dgcnn_pro_1(center_pts_trans[1], f_l2, center_pts_trans[0], f_l1, plan["dg1"])
        else:
            f_l2 = self.dgcnn_pro_2(center_trans, f_l3, center_pts_trans[1], f_l2)
            f_l1 = self.dgcnn_pro_1(center_pts_trans[1], f_l2, center_pts_trans[0], f_l1)
        f_l0 = self._fp(self.propogation_0, center_original, center_pts[0], f_l0, f_l1, plan.get("fp0"))

        head = self.seg_head                     # conv -> BatchNorm1d -> Dropout -> conv; the BatchNorm as one fused op
        if self.dense != "reference" and isinstance(head[0], PointwiseConv1d):
            y = pointwise(head[0].weight.view(head[0].out_channels, -1), f_l0)       # the bias goes through bn_act
            logit = head[3](head[2](bn_act(head[1], y, relu=False, pre_bias=head[0].bias)))
        else:
            logit = head[3](head[2](bn_act(head[1], head[0](f_l0), relu=False)))
        correction = self.T_linear(T) if T is not None else None
        return logit, correction, self.sigma, f_l0


def slice_geometry(geometry, lo, hi):
    """The geometry (prefetch_geometry's result) of clouds [lo, hi) of the batch `geometry` describes, for a model in EVAL
    mode: every entry is per cloud -- sample ids, neighbour ids and Morton orders are cloud-local -- so the slices are exactly
    what prefetch_geometry(pts[lo:hi]) of an eval-mode model would compute; the training-only entries (the reverse indices of
    the gradients) are dropped.  FixMatch's frozen teacher sees the weak view, which is also the last third of the student's
    batch: its geometry -- an 8192-sample FPS, Group, the index plan -- need not be computed twice."""
    if geometry is None or geometry.get("plan") is None:
        return None
    pts = geometry["pts"]
    b, n = pts.shape[0], pts.shape[1]
    neighborhood, center, idx = geometry["group"]
    flat = None if idx is None else (idx.view(b, -1)[lo:hi] - lo * n).reshape(-1)
    plan = geometry["plan"]

    def cut(t):
        return None if t is None else t[lo:hi]

    def fp(entry):                   # (idx, weight) or (idx, weight, Morton order, reverse index)
        return tuple(cut(t) for t in entry[:3]) + ((None,) if len(entry) > 3 else ())

    def graph(entry):                # ids, or (ids, reverse index) in training
        return cut(entry[0] if isinstance(entry, tuple) else entry)
    new_plan = {"center_pts": [cut(t) for t in plan["center_pts"]], "center_pts_trans": [cut(t) for t in plan["center_pts_trans"]],
                "center_trans": cut(plan["center_trans"]), "fp2": fp(plan["fp2"]), "fp1": fp(plan["fp1"]), "fp0": fp(plan["fp0"]),
                "dg2": tuple(graph(e) for e in plan["dg2"]), "dg1": tuple(graph(e) for e in plan["dg1"])}
    out = {"pts": pts[lo:hi], "version": geometry["version"], "group": (neighborhood[lo:hi], center[lo:hi], flat),
           "grouped": geometry.get("grouped"), "plan": new_plan, "training": False, "fp_layout": geometry["fp_layout"]}
    if geometry.get("static"):
        out["static"] = True
    if geometry.get("ready") is not None:
        out["ready"] = geometry["ready"]
    return out


TOOTH_SEG_CFG = dict(trans_dim=384, depth=12, num_heads=4, group_size=32, num_group=512, encoder_dims=256,
                     nclasses=17, drop_path_rate=0.1, downsample_targets=[8192, 4096, 2048],
                     extract_layers=[4, 8, 12])   # cfgs/tooth_semi/transformer_finetune_fixmatch_ntm.yaml:6-15
