"""numpy restatement of the dataloader-side ops -- TEST INFRASTRUCTURE ONLY.

* grid_subsampling: openpoints/cpp/subsampling/grid_subsampling/grid_subsampling.cpp:4-106 with
  SampledData (grid_subsampling.h:10-88) and PointXYZ (cpp_utils/cloud/cloud.h) arithmetic, op by op in
  fp32.  PINNED against the reference's own code compiled here (oracle/build_ref.py ->
  oracle/_ref/libgrid_subsampling_ref.so; tests/test_data_cpu.py) and against fixtures generated from it
  (tests/golden/grid_subsampling_ref.npz, generator tests/golden/make_grid_fixture.py).
  Two things the reference leaves to its hash map are fixed here, and only here:
    - output order: the reference emits cells in unordered_map iteration order; this restatement (and
      the HIP kernel) emits them by ascending cell key iX + NX*iY + NX*NY*iZ.  Same set of rows.
    - majority label on a tie: the reference takes the first maximum in hash-map iteration order; we
      take the smallest tied label.  Cells without a tie agree exactly.
* pc_norm / sample / class_weights: openpoints/dataset/tooth_semi/tooth_dataset.py:108-147 (numpy / torch
  on the CPU in the reference, so the reference IS numpy here: `pc_norm_numpy` below is that code's
  arithmetic verbatim in meaning -- fp32 mean, subtract, max row norm, divide -- kept beside an fp64 version
  used as ground truth for the tolerance test).
"""
import ctypes
import os

import numpy as np

F32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "_ref", "libgrid_subsampling_ref.so")


def grid_keys(points, dl):
    """-> (keys uint64 (N,), (NX, NY)) exactly as grid_subsampling.cpp:24-31, 52-56."""
    p = np.ascontiguousarray(points, dtype=F32)
    dl = F32(dl)
    mn, mx = p.min(axis=0), p.max(axis=0)
    inv = F32(1) / dl                                   # (1/sampleDl): int / float -> float
    origin = np.floor(mn * inv).astype(F32) * dl        # floor(minCorner * (1/sampleDl)) * sampleDl
    nx = np.uint64(np.floor((mx[0] - origin[0]) / dl)) + np.uint64(1)
    ny = np.uint64(np.floor((mx[1] - origin[1]) / dl)) + np.uint64(1)
    ijk = np.floor((p - origin[None, :]) / dl).astype(np.uint64)
    return ijk[:, 0] + nx * ijk[:, 1] + nx * ny * ijk[:, 2], (int(nx), int(ny))


def grid_subsampling(points, features=None, labels=None, dl=0.1):
    """-> dict(points (M,3), features (M,fdim)|None, labels (M,ldim)|None, keys (M,), tied (M,ldim) bool).
    Cells ascending by key; sums accumulate in input order (SampledData::update_*: `point += p`)."""
    p = np.ascontiguousarray(points, dtype=F32)
    keys, _ = grid_keys(p, dl)
    order = np.argsort(keys, kind="stable")
    ks = keys[order]
    head = np.ones(len(ks), dtype=bool)
    head[1:] = ks[1:] != ks[:-1]
    start = np.flatnonzero(head)
    count = np.diff(np.append(start, len(ks)))
    m = len(start)
    f = None if features is None else np.ascontiguousarray(features, dtype=F32).reshape(len(p), -1)
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32).reshape(len(p), -1)
    acc = np.zeros((m, 3), dtype=F32)
    facc = None if f is None else np.zeros((m, f.shape[1]), dtype=F32)
    for r in range(int(count.max())):                  # r-th member of every cell that has one: input order
        live = np.flatnonzero(count > r)
        src = order[start[live] + r]
        acc[live] = acc[live] + p[src]
        if f is not None:
            facc[live] = facc[live] + f[src]
    inv = (1.0 / count.astype(np.float64)).astype(F32)  # point * (1.0 / count): double -> float argument
    out = {"points": acc * inv[:, None], "keys": ks[start], "count": count,
           "features": None if f is None else facc / count.astype(F32)[:, None],   # f / (float)count
           "labels": None, "tied": None}
    if lab is not None:
        ol = np.zeros((m, lab.shape[1]), dtype=np.int32)
        tied = np.zeros((m, lab.shape[1]), dtype=bool)
        for c in range(m):
            rows = lab[order[start[c]:start[c] + count[c]]]
            for d in range(lab.shape[1]):
                vals, cnt = np.unique(rows[:, d], return_counts=True)
                best = cnt.max()
                ol[c, d] = vals[cnt == best].min()
                tied[c, d] = (cnt == best).sum() > 1
        out["labels"], out["tied"] = ol, tied
    return out


def have_reference():
    return os.path.exists(REF_SO)


def grid_subsampling_reference(points, features=None, labels=None, dl=0.1):
    """The reference's own compiled grid_subsampling() (oracle/_ref), rows in ITS order."""
    lib = ctypes.CDLL(REF_SO)
    p = np.ascontiguousarray(points, dtype=F32)
    n = len(p)
    f = None if features is None else np.ascontiguousarray(features, dtype=F32).reshape(n, -1)
    lab = None if labels is None else np.ascontiguousarray(labels, dtype=np.int32).reshape(n, -1)
    fdim = 0 if f is None else f.shape[1]
    ldim = 0 if lab is None else lab.shape[1]
    op = np.zeros((n, 3), dtype=F32)
    of = np.zeros((n, max(fdim, 1)), dtype=F32)
    ol = np.zeros((n, max(ldim, 1)), dtype=np.int32)
    P = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    lib.geot_gridref_subsample.restype = ctypes.c_int
    lib.geot_gridref_subsample.argtypes = [ctypes.c_int] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 3 + \
        [ctypes.c_int] + [ctypes.c_void_p] * 3
    m = lib.geot_gridref_subsample(n, fdim, ldim, float(dl), P(p), P(f), P(lab), n, P(op), P(of), P(ol))
    assert m >= 0
    return {"points": op[:m], "features": of[:m, :fdim] if fdim else None, "labels": ol[:m, :ldim] if ldim else None}


def row_order(a):
    """Permutation sorting (M,3) rows lexicographically (to compare outputs that differ only in order)."""
    return np.lexsort((a[:, 2], a[:, 1], a[:, 0]))


def pc_norm_numpy(pc):
    """tooth_dataset.py:108-114, as numpy executes it in fp32."""
    pc = np.asarray(pc, dtype=F32)
    centroid = np.mean(pc, axis=0)
    pc = pc - centroid
    m = np.max(np.sqrt(np.sum(pc ** 2, axis=1)))
    return pc / m, centroid, m


def pc_norm_f64(pc):
    pc = np.asarray(pc, dtype=np.float64)
    centroid = pc.mean(axis=0)
    q = pc - centroid
    m = np.sqrt((q ** 2).sum(axis=1)).max()
    with np.errstate(invalid="ignore", divide="ignore"):      # a one-point cloud has scale 0, as in the reference
        return q / m, centroid, m


def class_weights(sampled_labels, num_classes=17):
    """tooth_dataset.py:143-147: histogram over [0, C) in C bins, normalised; inf -> 0."""
    h = np.bincount(np.asarray(sampled_labels).astype(np.int64), minlength=num_classes)[:num_classes].astype(F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = h / h.sum(dtype=F32)
    return np.where(np.isinf(w), F32(0), w).astype(F32)
