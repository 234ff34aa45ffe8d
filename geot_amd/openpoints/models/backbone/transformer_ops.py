"""The sampling / grouping callers of the configured backbone ``PointTransformer_seg_T``
(openpoints/models/backbone/transformer.py): fps :266-273, Group :275-303 (FPS + kNN patch
embedding front end), DGCNN_Propagation.fps_downsample / get_graph_feature :326-364.
Same signatures and outputs (incl. Group's batch-offset flat index); the dense layers around
them (Conv2d / GroupNorm / transformer blocks) are stock PyTorch and stay in the caller."""
import torch
import torch.nn as nn

from ....pointnet2 import pointnet2_utils as pt_utils
from ....knn_cuda import KNN
from ....ext._common import f32, i32, same_device, need, call, ptr, grad_workspace
from .... import _lib


def fps(data, number):
    """data (B,N,3) -> furthest-sampled points (B,number,3) (pointnet2._ext FPS: origin-skip quirk)."""
    fps_idx = pt_utils.furthest_point_sample(data, number)
    return pt_utils.gather_operation(data.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()


class Group(nn.Module):
    """FPS + kNN: forward(xyz (B,N,3)) -> (neighborhood (B,G,M,3) centred, center (B,G,3), flat idx (B*G*M,))."""

    def __init__(self, num_group, group_size):
        super().__init__()
        self.num_group = num_group
        self.group_size = group_size
        self.knn = KNN(k=self.group_size, transpose_mode=True)

    def forward(self, xyz):
        batch_size, num_points, _ = xyz.shape
        center = fps(xyz, self.num_group)
        _, idx = self.knn(xyz, center)
        assert idx.size(1) == self.num_group and idx.size(2) == self.group_size
        idx = (idx + torch.arange(0, batch_size, device=xyz.device).view(-1, 1, 1) * num_points).view(-1)
        neighborhood = xyz.reshape(batch_size * num_points, -1)[idx, :]
        neighborhood = neighborhood.view(batch_size, self.num_group, self.group_size, 3).contiguous()
        return neighborhood - center.unsqueeze(2), center, idx


def fps_downsample(coor, x, num_group):
    """coor (B,3,N), x (B,C,N) -> (new_coor (B,3,G), new_x (B,C,G))."""
    fps_idx = pt_utils.furthest_point_sample(coor.transpose(1, 2).contiguous(), num_group)
    combined = pt_utils.gather_operation(torch.cat([coor, x], dim=1).contiguous(), fps_idx)
    return combined[:, :3], combined[:, 3:]


class _GraphFeatureFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_q, x_k, idx):
        x_q, x_k = f32(x_q.contiguous(), "x_q", 3), f32(x_k.contiguous(), "x_k", 3)
        idx = i32(idx.contiguous(), "idx", 3)
        dev = same_device(x_q, x_k, idx)
        b, c, nq = x_q.shape
        nk, k = x_k.shape[2], idx.shape[2]
        need(x_k.shape[:2] == (b, c) and tuple(idx.shape[:2]) == (b, nq), "graph feature shape mismatch")
        out = torch.empty((b, 2 * c, nq, k), dtype=torch.float32, device=dev)
        call("geot_graph_feature", dev, b, c, nq, nk, k, ptr(x_q), ptr(x_k), ptr(idx), ptr(out))
        ctx.save_for_backward(idx)
        ctx.dims = (b, c, nq, nk, k)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, = ctx.saved_tensors
        b, c, nq, nk, k = ctx.dims
        g = grad_out.contiguous()
        gq = torch.zeros((b, c, nq), dtype=torch.float32, device=g.device)
        gk = torch.zeros((b, c, nk), dtype=torch.float32, device=g.device)
        ws = grad_workspace(g.device, b, c, nk, nq * k, 1)
        call("geot_graph_feature_grad", g.device, b, c, nq, nk, k, ptr(g), ptr(idx), ptr(gq), ptr(gk), ptr(ws))
        return gq, gk, None


def graph_feature(x_q, x_k, idx):
    """x_q (B,C,Nq), x_k (B,C,Nk), idx (B,Nq,k) int32 -> (B,2C,Nq,k) = cat(x_k[idx] - x_q, x_q), fused."""
    return _GraphFeatureFn.apply(x_q, x_k, idx)


class _EdgeConvTailFn(torch.autograd.Function):
    """max_j LeakyReLU(GroupNorm(P[:, idx] + Q[..., None])) without the (B,C,Nq,k) tensor (csrc/edgeconv.hip)."""

    @staticmethod
    def forward(ctx, p, q, idx, gamma, beta, groups, eps, slope, rix=None):
        p, q = f32(p.contiguous(), "P", 3), f32(q.contiguous(), "Q", 3)
        idx = i32(idx.contiguous(), "idx", 3)
        gamma, beta = f32(gamma.contiguous(), "weight", 1), f32(beta.contiguous(), "bias", 1)
        dev = same_device(p, q, idx, gamma, beta)
        b, c, nk = p.shape
        nq, k = q.shape[2], idx.shape[2]
        need(tuple(q.shape[:2]) == (b, c) and tuple(idx.shape[:2]) == (b, nq) and gamma.numel() == c and beta.numel() == c,
             "edgeconv tail shape mismatch")
        lib = _lib.load()
        need(lib.geot_edgeconv_eligible(b, c, nq, nk, k, int(groups)) == 1 and slope >= 0,
             "edgeconv tail: unsupported shape (see include/geot_hip.h)")
        nbytes = int(lib.geot_edgeconv_ws_bytes(b, c, nq, nk, k))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out, ysel, ysum = torch.empty((3, b, c, nq), dtype=torch.float32, device=dev).unbind(0)
        jsel = torch.empty((b, c, nq), dtype=torch.uint8, device=dev)
        stats = torch.empty((b, int(groups), 2), dtype=torch.float32, device=dev)
        call("geot_edgeconv_gn_max", dev, b, c, nq, nk, k, int(groups), float(eps), float(slope), ptr(p), ptr(q), ptr(idx),
             ptr(gamma), ptr(beta), ptr(out), ptr(ysel), ptr(ysum), ptr(jsel), ptr(stats), ptr(ws), nbytes)
        ctx.save_for_backward(p, q, idx, gamma, beta, ysel, ysum, jsel, stats)
        ctx.consts = (int(groups), float(slope), nbytes)
        ctx.rix = rix                  # the reverse index of idx built ahead (edgeconv_reverse_index), or None
        return out

    @staticmethod
    def backward(ctx, g):
        p, q, idx, gamma, beta, ysel, ysum, jsel, stats = ctx.saved_tensors
        groups, slope, nbytes = ctx.consts
        b, c, nk = p.shape
        nq, k = q.shape[2], idx.shape[2]
        g = g.contiguous()
        gp, gq = torch.empty_like(p), torch.empty_like(q)
        gg, gb = torch.empty_like(gamma), torch.empty_like(beta)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
        if ctx.rix is not None:
            call("geot_edgeconv_gn_max_grad_rix", p.device, b, c, nq, nk, k, groups, slope, ptr(p), ptr(q), ptr(ctx.rix), ptr(gamma),
                 ptr(beta), ptr(ysel), ptr(ysum), ptr(jsel), ptr(stats), ptr(g), ptr(gp), ptr(gq), ptr(gg), ptr(gb), ptr(ws),
                 nbytes)
        else:
            call("geot_edgeconv_gn_max_grad", p.device, b, c, nq, nk, k, groups, slope, ptr(p), ptr(q), ptr(idx), ptr(gamma),
                 ptr(beta), ptr(ysel), ptr(ysum), ptr(jsel), ptr(stats), ptr(g), ptr(gp), ptr(gq), ptr(gg), ptr(gb), ptr(ws),
                 nbytes)
        return gp, gq, None, gg, gb, None, None, None, None


def edgeconv_reverse_index(idx, nk):
    """idx (B,Nq,k) int32 neighbours among nk sources -> the reverse index the fused tail's gradient walks
    (geot_edgeconv_rix_build): depends on the graph alone, so the model builds it with its index plan."""
    idx = i32(idx.contiguous(), "idx", 3)
    b, nq, k = idx.shape
    ints = int(_lib.load().geot_edgeconv_rix_ints(b, nq, int(nk), k))
    rix = torch.empty(ints, dtype=torch.int32, device=idx.device)
    call("geot_edgeconv_rix_build", idx.device, b, nq, int(nk), k, ptr(idx), ptr(rix), ints)
    return rix


def edgeconv_tail(p, q, idx, norm, slope, rix=None):
    """p (B,C,Nk) = W_d x_k, q (B,C,Nq) = (W_q - W_d) x_q, idx (B,Nq,k) int32, norm = the layer's nn.GroupNorm ->
    (B,C,Nq) = max_j LeakyReLU(norm(p[:, idx] + q[..., None])), one fused forward / backward.  rix: the reverse index of
    idx when it was built ahead (edgeconv_reverse_index)."""
    return _EdgeConvTailFn.apply(p, q, idx, norm.weight, norm.bias, norm.num_groups, norm.eps, slope, rix)


def edgeconv_tail_eligible(b, c, nq, nk, k, groups):
    return _lib.load().geot_edgeconv_eligible(int(b), int(c), int(nq), int(nk), int(k), int(groups)) == 1


def get_graph_feature(knn, coor_q, x_q, coor_k, x_k):
    """EdgeConv features (B, 2C, Nq, k) = cat(x_k[nbr] - x_q, x_q); knn = KNN(k, transpose_mode=False).
    Same result as the reference's gather/permute/expand/cat chain, from one fused kernel (and the kNN
    indices are consumed as (B,Nq,k) int32 directly instead of being transposed to (B,k,Nq) int64)."""
    from ....knn_cuda import knn_sorted
    with torch.no_grad():
        _, idx = knn_sorted(coor_q.transpose(1, 2).contiguous().float(), coor_k.transpose(1, 2).contiguous().float(),
                            knn.k)
    return graph_feature(x_q, x_k, idx)


def get_graph_feature_unfused(knn, coor_q, x_q, coor_k, x_k):
    """The reference's own op sequence (kept for A/B tests)."""
    k = knn.k
    batch_size, num_dims, num_points_k = x_k.shape
    num_points_q = x_q.size(2)
    with torch.no_grad():
        _, idx = knn(coor_k, coor_q)                          # (B, k, Nq)
        assert idx.shape[1] == k
        idx = (idx + torch.arange(0, batch_size, device=x_q.device).view(-1, 1, 1) * num_points_k).view(-1)
    feature = x_k.transpose(2, 1).contiguous().view(batch_size * num_points_k, -1)[idx, :]
    feature = feature.view(batch_size, k, num_points_q, num_dims).permute(0, 3, 2, 1).contiguous()
    x_q = x_q.view(batch_size, num_dims, num_points_q, 1).expand(-1, -1, -1, k)
    return torch.cat((feature - x_q, x_q), dim=1)
