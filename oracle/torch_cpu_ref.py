"""TEST / BASELINE INFRASTRUCTURE -- never imported by the product (geot_amd/).

The reference's own CPU-capable path for the sampling / grouping operators, restated for timing next to the
GPU path (SURVEY.md section 8(d) "CPU baseline beside it"):

  square_distance / index_points / farthest_point_sample / query_ball_point / knn_point
      openpoints/models/backbone/pointmlp.py:45-143  (pure torch, device-agnostic; the file cannot be imported
      here -- relative imports into a package whose __init__ needs CUDA extensions -- so it is restated)
  knn_point (cdist + topk)   openpoints/models/layers/knn.py:7-20

These fallbacks are NOT bit-compatible with the CUDA ops (random FPS start :98, `>` radius test :124, expanded-form
distances :61-63, unsorted topk :142), so they are a TIMING baseline only -- parity is checked against
oracle/geot_oracle.c.  `cpu_model()` builds the mirrored PointTransformer_seg_T on the CPU and patches its hot-path
call sites to these fallbacks (kind "reference-fallback") or to the C/OpenMP oracle for the index-producing ops
(kind "port"); the dense layers are the same torch modules on torch-CPU either way.
"""
import contextlib
import time

import numpy as np
import torch


# ---- pointmlp.py:45-143 -------------------------------------------------------------------------------------
def square_distance(src, dst):
    B, N, _ = src.shape
    M = dst.shape[1]
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).view(B, N, 1)
    dist += torch.sum(dst ** 2, -1).view(B, 1, M)
    return dist


def index_points(points, idx):
    B = points.shape[0]
    view_shape = list(idx.shape)
    view_shape[1:] = [1] * (len(view_shape) - 1)
    repeat_shape = list(idx.shape)
    repeat_shape[0] = 1
    batch_indices = torch.arange(B, dtype=torch.long).view(view_shape).repeat(repeat_shape)
    return points[batch_indices, idx, :]


def farthest_point_sample(xyz, npoint):
    B, N, _ = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.ones(B, N) * 1e10
    farthest = torch.randint(0, N, (B,), dtype=torch.long)
    batch_indices = torch.arange(B, dtype=torch.long)
    for i in range(npoint):
        centroids[:, i] = farthest
        centroid = xyz[batch_indices, farthest, :].view(B, 1, 3)
        dist = torch.sum((xyz - centroid) ** 2, -1)
        distance = torch.min(distance, dist)
        farthest = torch.max(distance, -1)[1]
    return centroids


def query_ball_point(radius, nsample, xyz, new_xyz):
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    group_idx = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat([B, S, 1])
    sqrdists = square_distance(new_xyz, xyz)
    group_idx[sqrdists > radius ** 2] = N
    group_idx = group_idx.sort(dim=-1)[0][:, :, :nsample]
    group_first = group_idx[:, :, 0].view(B, S, 1).repeat([1, 1, nsample])
    mask = group_idx == N
    group_idx[mask] = group_first[mask]
    return group_idx


def knn_point_fallback(nsample, xyz, new_xyz):
    sqrdists = square_distance(new_xyz, xyz)
    return torch.topk(sqrdists, nsample, dim=-1, largest=False, sorted=False)[1]


def knn_point(k, query_xyz, support_xyz):
    """openpoints/models/layers/knn.py:7-20."""
    dist = torch.cdist(query_xyz, support_xyz)
    k_dist = dist.topk(k=k, dim=-1, largest=False, sorted=True)
    return k_dist[0], k_dist[1]


# ---- the hot-path call sites of the mirrored model, on the CPU ----------------------------------------------
class _CpuOps:
    """kind = "reference-fallback": every op from the torch fallbacks above; "port": index-producing ops from the
    C/OpenMP oracle (exact CUDA semantics), gathers as torch indexing."""

    def __init__(self, kind):
        self.kind = kind
        if kind == "port":
            from oracle import capi
            self.capi = capi

    def fps_idx(self, xyz, m, cap, skip_origin):
        if self.kind == "port":
            return torch.from_numpy(self.capi.fps_dense(xyz.numpy(), m, cap, skip_origin).astype(np.int64))
        return farthest_point_sample(xyz, m)

    def knn_idx(self, query, ref, k):
        if self.kind == "port":
            return torch.from_numpy(self.capi.knn_sorted(query.numpy(), ref.numpy(), k)[0].astype(np.int64))
        return knn_point(k, query, ref)[1]

    def three_nn(self, unknown, known):
        if self.kind == "port":
            d2, idx = self.capi.three_nn(unknown.numpy(), known.numpy())
            return torch.from_numpy(np.sqrt(d2)), torch.from_numpy(idx.astype(np.int64))
        d2, idx = torch.topk(square_distance(unknown, known), 3, dim=-1, largest=False, sorted=True)
        return torch.sqrt(d2.clamp_min(0)), idx


@contextlib.contextmanager
def patched(kind):
    """Route the mirrored model's hot-path call sites to CPU implementations for the duration of the block."""
    from geot_amd.openpoints.models.backbone import transformer as tr, transformer_ops as tro
    from geot_amd.pointnet2 import pointnet2_utils as pu, pointnet2_modules as pm
    from geot_amd.pointops.functions import pointops as pops
    ops = _CpuOps(kind)

    def furthest_point_sample(xyz, npoint):
        return ops.fps_idx(xyz.detach(), npoint, 512, True)

    def gather_operation(features, idx):                      # (B,C,N), (B,M) -> (B,C,M)
        return torch.gather(features, 2, idx.long().unsqueeze(1).expand(-1, features.shape[1], -1))

    def three_nn(unknown, known):
        return ops.three_nn(unknown.detach(), known.detach())

    def three_interpolate(features, idx, weight):             # (B,C,m), (B,n,3), (B,n,3) -> (B,C,n)
        b, c, m = features.shape
        n = idx.shape[1]
        g = torch.gather(features, 2, idx.long().reshape(b, 1, n * 3).expand(-1, c, -1)).view(b, c, n, 3)
        return (g * weight.unsqueeze(1)).sum(-1)

    def fp_interpolate_concat(unknown, known, unknow_feats, known_feats, skip_first=False):
        dist, idx = three_nn(unknown, known)
        r = 1.0 / (dist + 1e-8)
        interp = three_interpolate(known_feats, idx, r / torch.sum(r, dim=2, keepdim=True))
        if unknow_feats is None:
            return interp
        return torch.cat([unknow_feats, interp] if skip_first else [interp, unknow_feats], dim=1)

    def grouping_operation(features, idx):                    # (B,C,N), (B,np,ns) -> (B,C,np,ns)
        b, c, _ = features.shape
        return torch.gather(features, 2, idx.long().reshape(b, 1, -1).expand(-1, c, -1)).view(b, c, *idx.shape[1:])

    def fps_indices(x, k):
        b, n, _ = x.shape
        return ops.fps_idx(x.detach(), k, 1024, False) + torch.arange(b).view(b, 1) * n

    def knn_idx_cf(coor_q, coor_k, k):                        # channel-first (B,3,N) inputs -> (B,Nq,k)
        return ops.knn_idx(coor_q.transpose(1, 2).contiguous().detach(), coor_k.transpose(1, 2).contiguous().detach(), k)

    def graph_feature(x_q, x_k, idx):                         # transformer.py:343-364 with idx (B,Nq,k)
        g = grouping_operation(x_k, idx)
        xq = x_q.unsqueeze(-1).expand(-1, -1, -1, idx.shape[2])
        return torch.cat((g - xq, xq), dim=1)

    class KNN(torch.nn.Module):
        def __init__(self, k, transpose_mode=False):
            super().__init__()
            self.k, self._t = k, transpose_mode

        def forward(self, ref, query):
            if not self._t:
                ref, query = ref.transpose(1, 2), query.transpose(1, 2)
            idx = ops.knn_idx(query.contiguous().detach(), ref.contiguous().detach(), self.k)
            return None, idx if self._t else idx.transpose(1, 2).contiguous()

    saved = []

    def swap(owner, name, new):
        saved.append((owner, name, getattr(owner, name)))
        setattr(owner, name, new)

    for owner in (pu, tr.pt_utils):
        swap(owner, "furthest_point_sample", furthest_point_sample)
        swap(owner, "gather_operation", gather_operation)
        swap(owner, "three_nn", three_nn)
        swap(owner, "three_interpolate", three_interpolate)
        swap(owner, "fp_interpolate_concat", fp_interpolate_concat)
        swap(owner, "grouping_operation", grouping_operation)
    swap(pops, "fps_indices", fps_indices)
    swap(tr, "_knn_idx", knn_idx_cf)
    swap(tr, "graph_feature", graph_feature)
    swap(tro, "KNN", KNN)
    swap(tr, "KNN", KNN)
    try:
        yield
    finally:
        for owner, name, old in reversed(saved):
            setattr(owner, name, old)


def time_model_step(kind, clouds, n_points, cores, steps=1, seed=1609):
    """fwd + Poly1Focal loss + bwd + AdamW step of the mirrored PointTransformer_seg_T on torch-CPU with the
    hot-path ops routed as `kind` says.  Returns (clouds/s, seconds, description)."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
    from geot_amd.train_step import SupervisedStep
    from geot_amd.synth import make_batch, region_labels
    torch.set_num_threads(cores)
    if kind == "port":
        from oracle import capi
        capi.set_threads(cores)
    torch.manual_seed(seed)
    xyz_np, _ = make_batch(clouds, n_points)
    pos = torch.from_numpy(xyz_np)
    target = torch.from_numpy(region_labels(xyz_np))
    cls = torch.zeros(clouds, 1, dtype=torch.long)
    with patched(kind):
        model = PointTransformer_seg_T(**TOOTH_SEG_CFG, dense="reference", overlap=False)
        step = SupervisedStep(model)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = step(pos, cls, target)
        dt = time.perf_counter() - t0
    assert torch.isfinite(loss)
    return clouds * steps / dt, dt
