"""Independent numpy restatement of the index-producing ops -- TEST INFRASTRUCTURE ONLY.

Where oracle/geot_oracle.c simulates the reference kernels' loops literally
(thread-strided scans, shared-memory tree reduction, heap sift-down), this file
derives the same answers from closed-form rules, so that agreement of the two
is evidence that the rules in SURVEY.md Appendix A are right:

* FPS winner   = arg-max of the running min-distance; ties broken by the key
                 (bitreverse_{log2 bs}(k mod bs), k div bs), smallest first
                 (pointnet2/_ext_src/src/sampling_gpu.cu:62-68, 92-175).
* ball query   = the first ``nsample`` support indices, ascending, with
                 d2 < r*r (strict); first hit pre-fills; no hit -> zeros
                 (pointnet2/_ext_src/src/ball_query_gpu.cu:27-45).
* three_nn     = 3 smallest by (d2, index) lexicographic
                 (pointnet2/_ext_src/src/interpolate_gpu.cu:31-53).
* sorted kNN   = k smallest by (d2, index) lexicographic.

All distances are un-contracted fp32: ((dx*dx) + (dy*dy)) + (dz*dz).
"""
import numpy as np

F32 = np.float32


def sqdist(a, b):
    """a (...,3) f32, b (...,3) f32 broadcastable -> fp32 squared distance, op by op."""
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    dx = a[..., 0] - b[..., 0]
    dy = a[..., 1] - b[..., 1]
    dz = a[..., 2] - b[..., 2]
    s = dx * dx
    s = s + dy * dy
    s = s + dz * dz
    return s.astype(F32)


def block_size(n, cap):
    p = int(np.log(float(n)) / np.log(2.0))
    return max(min(1 << p, cap), 1)


def _bitrev(x, bits):
    x = np.asarray(x, dtype=np.int64)
    r = np.zeros_like(x)
    for i in range(bits):
        r |= ((x >> i) & 1) << (bits - 1 - i)
    return r


def fps_keys(n, bs):
    """Tie-break key per local point index k (smaller wins)."""
    bits = int(np.log2(bs))
    k = np.arange(n, dtype=np.int64)
    rows = (n + bs - 1) // bs + 1
    return _bitrev(k % bs, bits) * rows + k // bs


def fps_one(pts, m, bs, skip_origin=False, weights=None):
    pts = np.asarray(pts, dtype=F32)
    n = pts.shape[0]
    temp = np.full(n, 1e10, dtype=F32)
    out = np.zeros(m, dtype=np.int32)
    if m <= 0:
        return out, temp
    valid = np.ones(n, dtype=bool)
    if skip_origin:
        mag = (pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]) + pts[:, 2] * pts[:, 2]
        valid = ~(mag.astype(np.float64) <= 1e-3)
    key = fps_keys(n, bs)
    old = 0
    for j in range(1, m):
        d = sqdist(pts, pts[old])
        if weights is not None:
            w = np.maximum(np.asarray(weights, dtype=np.float64), 1e-12)
            d = (d.astype(np.float64) * w).astype(F32)
        d2 = np.fmin(d, temp)
        temp = np.where(valid, d2, temp)
        if not valid.any():
            old = 0
        else:
            cand = np.where(valid, d2, F32(-np.inf))
            mx = cand.max()
            if not (mx > -1.0):  # nothing beats the per-thread init best=-1
                old = 0
            else:
                tie = np.nonzero(cand == mx)[0]
                old = int(tie[np.argmin(key[tie])])
        out[j] = old
    return out, temp


def fps_dense(xyz, m, cap=512, skip_origin=True):
    xyz = np.asarray(xyz, dtype=F32)
    bs = block_size(xyz.shape[1], cap)
    return np.stack([fps_one(c, m, bs, skip_origin)[0] for c in xyz])


def fps_offset(xyz, offset, new_offset, weights=None, n_max=None):
    xyz = np.asarray(xyz, dtype=F32)
    offset = np.asarray(offset)
    new_offset = np.asarray(new_offset)
    starts = np.concatenate([[0], offset[:-1]])
    mstarts = np.concatenate([[0], new_offset[:-1]])
    if n_max is None:
        n_max = int((offset - starts).max())
    bs = block_size(n_max, 1024)
    out = np.zeros(int(new_offset[-1]), dtype=np.int32)
    for s, e, ms, me in zip(starts, offset, mstarts, new_offset):
        if me <= ms or e <= s:
            continue
        w = None if weights is None else np.asarray(weights)[s:e]
        out[ms:me] = fps_one(xyz[s:e], int(me - ms), bs, False, w)[0] + s
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz = np.asarray(new_xyz, dtype=F32)
    xyz = np.asarray(xyz, dtype=F32)
    r2 = F32(radius) * F32(radius)
    b, m, _ = new_xyz.shape
    out = np.zeros((b, m, nsample), dtype=np.int32)
    for i in range(b):
        d2 = sqdist(new_xyz[i][:, None, :], xyz[i][None, :, :])  # (m, n)
        hit = d2 < r2
        for j in range(m):
            ks = np.nonzero(hit[j])[0][:nsample]
            if ks.size:
                out[i, j, :] = ks[0]
                out[i, j, :ks.size] = ks
    return out


def _lex_topk(d2, k):
    """rows of d2 (q, r): k smallest by (value, index); stable argsort does exactly that."""
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return order.astype(np.int32), np.take_along_axis(d2, order, axis=1)


def three_nn(unknown, known):
    unknown = np.asarray(unknown, dtype=F32)
    known = np.asarray(known, dtype=F32)
    b, n, _ = unknown.shape
    m = known.shape[1]
    idx = np.zeros((b, n, 3), dtype=np.int32)
    dist2 = np.full((b, n, 3), np.inf, dtype=F32)
    for i in range(b):
        d2 = sqdist(unknown[i][:, None, :], known[i][None, :, :])
        o, v = _lex_topk(d2, 3)
        idx[i, :, :min(3, m)] = o
        dist2[i, :, :min(3, m)] = v
    return dist2, idx


def knn_sorted(query, ref, k):
    query = np.asarray(query, dtype=F32)
    ref = np.asarray(ref, dtype=F32)
    b, nq, _ = query.shape
    nr = ref.shape[1]
    idx = np.zeros((b, nq, k), dtype=np.int32)
    dist2 = np.full((b, nq, k), np.inf, dtype=F32)
    for i in range(b):
        d2 = sqdist(query[i][:, None, :], ref[i][None, :, :])
        o, v = _lex_topk(d2, k)
        idx[i, :, :min(k, nr)] = o
        dist2[i, :, :min(k, nr)] = v
    return idx, dist2


def knn_sorted_nd(query, ref, k):
    """Sorted kNN in D dimensions, (d2, index) order; d2 accumulated over the dimensions in index order in
    fp32, one rounding per multiply and per add (the HIP kernel's un-contracted loop)."""
    query = np.asarray(query, dtype=F32)
    ref = np.asarray(ref, dtype=F32)
    b, nq, d = query.shape
    nr = ref.shape[1]
    idx = np.zeros((b, nq, k), dtype=np.int32)
    dist2 = np.full((b, nq, k), np.inf, dtype=F32)
    for i in range(b):
        d2 = np.zeros((nq, nr), dtype=F32)
        for t in range(d):
            df = (query[i][:, None, t] - ref[i][None, :, t]).astype(F32)
            d2 = (d2 + (df * df).astype(F32)).astype(F32)
        o, v = _lex_topk(d2, k)
        idx[i, :, :min(k, nr)] = o
        dist2[i, :, :min(k, nr)] = v
    return idx, dist2


def knnquery_heap_literal(nsample, xyz, new_xyz, offset, new_offset):
    """Pure-python transcription of the heap procedure (small inputs only)."""
    xyz = np.asarray(xyz, dtype=F32)
    new_xyz = np.asarray(new_xyz, dtype=F32)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    dist2 = np.zeros((m, nsample), dtype=F32)

    def sift(d, ix, k):
        root, child = 0, 1
        while child < k:
            if child + 1 < k and d[child + 1] > d[child]:
                child += 1
            if d[root] > d[child]:
                return
            d[root], d[child] = d[child], d[root]
            ix[root], ix[child] = ix[child], ix[root]
            root = child
            child = 2 * root + 1

    for p in range(m):
        bt = int(np.searchsorted(np.asarray(new_offset), p, side="right"))
        start = int(offset[bt - 1]) if bt else 0
        end = int(offset[bt])
        d = [F32(1e10)] * nsample
        ix = [start] * nsample
        dd = sqdist(new_xyz[p][None, :], xyz[start:end])
        for t in range(end - start):
            if dd[t] < d[0]:
                d[0] = dd[t]
                ix[0] = start + t
                sift(d, ix, nsample)
        for i in range(nsample - 1, 0, -1):
            d[0], d[i] = d[i], d[0]
            ix[0], ix[i] = ix[i], ix[0]
            sift(d, ix, i)
        idx[p] = ix
        dist2[p] = d
    return idx, dist2


# ---- float ops: plain numpy definitions ---------------------------------

def gather_points(points, idx):
    return np.take_along_axis(np.asarray(points, dtype=F32),
                              np.asarray(idx)[:, None, :].astype(np.int64), axis=2)


def group_points(points, idx):
    points = np.asarray(points, dtype=F32)
    idx = np.asarray(idx).astype(np.int64)
    b, c, n = points.shape
    _, npnt, ns = idx.shape
    flat = np.take_along_axis(points, idx.reshape(b, 1, npnt * ns).repeat(c, axis=1), axis=2)
    return flat.reshape(b, c, npnt, ns)


def three_interpolate(points, idx, weight):
    g = group_points(points, idx)  # (b,c,n,3)
    w = np.asarray(weight, dtype=F32)[:, None, :, :]
    return (g[..., 0] * w[..., 0] + g[..., 1] * w[..., 1] + g[..., 2] * w[..., 2]).astype(F32)
