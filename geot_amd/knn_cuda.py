"""``knn_cuda.KNN`` -- the un-vendored third-party module the reference imports
(openpoints/models/backbone/transformer.py:11; call sites :280,293,313,353).

The package source is not in the reference tree and no version is pinned, so the
contract is taken from the call sites:
``KNN(k, transpose_mode)(ref, query) -> (dist f32, idx int64)``, neighbours in
ascending distance order; transpose_mode=True: ref (B,N,D), query (B,M,D) ->
(B,M,k); False: ref (B,D,N), query (B,D,M) -> (B,k,M).  Ties are broken by the
smaller index (our choice; parity against the real package is unpinned).
Only D == 3 is accelerated, which is all the reference uses.
"""
import torch
import torch.nn as nn

from .ext._common import f32, same_device, need, call, ptr, knn_workspace


def knn_sorted(query, ref, k):
    """query (B,Q,3), ref (B,R,3) contiguous f32 -> (dist2 (B,Q,k) f32, idx (B,Q,k) i32)."""
    f32(query, "query", 3); f32(ref, "ref", 3)
    dev = same_device(query, ref)
    b, nq, d = query.shape
    need(d == 3 and ref.shape[2] == 3 and ref.shape[0] == b, "knn expects (B, N, 3) coordinates")
    nr = ref.shape[1]
    idx = torch.empty((b, nq, k), dtype=torch.int32, device=dev)
    dist2 = torch.empty((b, nq, k), dtype=torch.float32, device=dev)
    wp, wb, _keep = knn_workspace(dev, b, nq, nr, int(k))   # grid search when the problem is big enough
    call("geot_knn_sorted_ws", dev, b, nq, nr, int(k), ptr(query), ptr(ref), ptr(idx), ptr(dist2), wp, wb)
    return dist2, idx


class KNN(nn.Module):
    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    @torch.no_grad()
    def forward(self, ref, query):
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        if not self._t:
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        dist2, idx = knn_sorted(query.contiguous().float(), ref.contiguous().float(), self.k)
        dist, idx = torch.sqrt(dist2), idx.long()
        if not self._t:
            dist, idx = dist.transpose(1, 2).contiguous(), idx.transpose(1, 2).contiguous()
        return dist, idx
