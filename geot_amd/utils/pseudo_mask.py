"""kNN-based pseudo-label refinement -- mirror of utils/pseudo_mask.py (get_neigbor_tensors :5-35,
pseudo_label_refine :38-53, pseudo_label_refine_margin :55-90, pseudo_label_refine_margin_v1 :92-170,
neigh_acc_count :174-196; the configured run keeps ``pseudo_refine: False``, yaml:75).

The neighbours come from ``pointops.knn`` (heap-order semantics kept, grid-accelerated) and the per-neighbour
``index_select`` loop is one grouping launch: (B,C,N) gathered at (B,N,n) -> (B,C,N,n).  Same signatures, same
returned tensors (sic: the reference's spellings are kept).  Everything runs under ``no_grad`` as there; the class
counters of ``neigh_acc_count`` stay on the device until ``acc_array`` is read (the reference adds 2 x num_class device
scalars to a numpy array per update = 34 host syncs).
"""
import numpy as np
import torch

from ..pointops.functions import pointops
from ..pointnet2 import pointnet2_utils as pt_utils

E_JOINT = [0.9698153347167245, 0.9595924029774019, 0.9596092881209647, 0.9617471101196512, 0.9662687092798028,
           0.9684095068416779, 0.9766432433032493, 0.9754884408811396, 0.9629032258064516, 0.9596091749248413,
           0.9584221215955251, 0.9619788870996601, 0.9666700999073025, 0.968204136476084, 0.9760611218051148,
           0.9746949382049295, 0.966996699669967]     # pseudo_mask.py:56-61


def get_neigbor_tensors(X, n, pos):
    """X (B,C,N), pos (B,N,3) -> (list of n tensors (B,C,N): X at each point's ii-th nearest OTHER point,
    top_dist (B,N,n))."""
    B, C, N = X.shape
    top_dist_index, top_dist = pointops.knn(pos, pos, n + 1)             # (B,N,n+1); slot 0 = the point itself
    top_dist = top_dist[:, :, 1:]
    idx = top_dist_index[:, :, 1:].to(torch.int32).contiguous()          # local ids within each cloud
    grouped = pt_utils.grouping_operation(X.contiguous().float(), idx)   # (B,C,N,n) in one launch
    return [grouped[..., ii].contiguous() for ii in range(n)], top_dist


def _blend(pred_t, pos, neigborhood_size, n_neigbors):
    neighbors, _ = get_neigbor_tensors(pred_t, n=neigborhood_size, pos=pos)
    k_neighbors, _ = torch.topk(torch.stack(neighbors), k=n_neigbors, dim=0)          # (n_neigbors,B,C,N)
    beta = float(np.exp(-0.5))
    for neighbor in k_neighbors:
        pred_t = pred_t + beta * neighbor - (pred_t * neighbor) * beta
    return pred_t


@torch.no_grad()
def pseudo_label_refine(pred_t, th, pos, neigborhood_size=4, n_neigbors=1):
    pred_t = _blend(pred_t, pos, neigborhood_size, n_neigbors)
    logits_u_aug, _ = torch.max(pred_t.detach(), dim=1)
    return logits_u_aug.ge(th).bool()


@torch.no_grad()
def pseudo_label_refine_margin(pred_t, th, pos, neigborhood_size=4, n_neigbors=1):
    pred_t = _blend(pred_t, pos, neigborhood_size, n_neigbors)
    _topk, _ = torch.topk(pred_t.detach(), 2, dim=1)
    _margin = _topk[:, 0, :] - _topk[:, 1, :]
    return _margin.ge(th).bool(), _margin


@torch.no_grad()
def pseudo_label_refine_margin_v1(pred_t, th, drop_percent, pos, neigborhood_size=4, n_neigbors=1):
    B, C, N = pred_t.shape
    E = torch.tensor(E_JOINT[:C], device=pred_t.device, dtype=pred_t.dtype).view(1, C, 1)
    neighbors, _ = get_neigbor_tensors(pred_t, n=neigborhood_size, pos=pos)
    k_neighbors, _ = torch.topk(torch.stack(neighbors), k=n_neigbors, dim=0)
    for neighbor in k_neighbors:
        upper_bound = E * pred_t / neighbor
        pred_t = pred_t + neighbor - (pred_t * upper_bound)
    _topk, _ = torch.topk(pred_t.detach(), 2, dim=1)
    _margin = _topk[:, 0, :] - _topk[:, 1, :]
    return _margin.ge(th).bool(), _margin, th


class neigh_acc_count:
    """Per-class agreement between a point's predicted label and its nearest neighbour's (first cloud of the batch
    only, as the reference: pseudo_mask.py:184-189)."""

    def __init__(self, num_class=17):
        self.num_class = num_class
        self._acc = None

    @torch.no_grad()
    def update(self, pred, pos, neigborhood_size=4, n_neigbors=1):
        top_dist_index, _ = pointops.knn(pos, pos, 2)
        nn_idx = top_dist_index[0, :, 1].long()
        p = pred[0]
        agree = (p == p[nn_idx])
        counts = torch.bincount(p, minlength=self.num_class)[:self.num_class]
        hits = torch.bincount(p[agree], minlength=self.num_class)[:self.num_class]
        upd = torch.stack([counts, hits], 1).to(torch.float64)
        self._acc = upd if self._acc is None else self._acc + upd

    @property
    def acc_array(self):
        return np.zeros((self.num_class, 2)) if self._acc is None else self._acc.cpu().numpy()
