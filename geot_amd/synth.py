"""Synthetic "tooth scan" clouds used by the tests and bench.py (no dataset ships offline).

The generator follows the measurement plan in SURVEY.md section 8(d): seed
``1609 + cloud_index`` (the reference's cfg seed,
cfgs/tooth_semi/transformer_finetune_fixmatch_ntm.yaml:67), N points on a bumpy
closed surface, then the reference's own normalisation -- subtract the centroid,
divide by the largest norm (openpoints/dataset/tooth_semi/tooth_dataset.py:108-114)
-- so every point has norm <= 1.  A few points are placed within |p|^2 <= 1e-3
to exercise the origin-skip quirk of pointnet2._ext's FPS, and ``dup_frac`` of the
points can be exact duplicates (the loader samples with replacement for small
scans, tooth_dataset.py:134-135) to exercise the tie rules.
"""
import numpy as np

BASE_SEED = 1609
NUM_CLASSES = 17


def make_cloud(n, index=0, dup_frac=0.0, origin_pts=4):
    """Return (xyz float32 (n,3), labels int64 (n,)) for cloud ``index``."""
    rng = np.random.default_rng(BASE_SEED + index)
    v = rng.standard_normal((n, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True) + 1e-12
    theta = np.arccos(np.clip(v[:, 2], -1, 1))
    phi = np.arctan2(v[:, 1], v[:, 0])
    r = 0.8 + 0.12 * np.sin(3 * theta) * np.cos(2 * phi) + 0.05 * np.cos(5 * phi)
    r = r + 0.004 * rng.standard_normal(n)
    p = v * r[:, None] * np.array([1.0, 0.8, 0.6])
    p = p - p.mean(axis=0, keepdims=True)
    p = p / max(np.sqrt((p ** 2).sum(axis=1)).max(), 1e-12)
    p = p.astype(np.float32)
    if origin_pts and n > 8 * origin_pts:
        where = rng.choice(np.arange(1, n), size=origin_pts, replace=False)
        p[where] = (rng.standard_normal((origin_pts, 3)) * 0.01).astype(np.float32)
    if dup_frac > 0:
        k = max(1, int(n * dup_frac))
        dst = rng.choice(np.arange(1, n), size=k, replace=False)
        src = rng.integers(0, n, size=k)
        p[dst] = p[src]
    labels = rng.integers(0, NUM_CLASSES, size=n).astype(np.int64)
    return p, labels


def make_batch(b, n, start_index=0, dup_frac=0.0, origin_pts=4):
    """Return (xyz float32 (b,n,3), labels int64 (b,n))."""
    clouds = [make_cloud(n, start_index + i, dup_frac, origin_pts) for i in range(b)]
    return np.stack([c[0] for c in clouds]), np.stack([c[1] for c in clouds])


def region_labels(xyz, num_classes=NUM_CLASSES):
    """Spatially coherent stand-in for tooth labels: class 0 ("gum") = the lower 40 % of the scan along z,
    classes 1..C-1 = equal angular sectors of the rest around the centroid, like teeth along an arch.
    xyz (..., n, 3) -> int64 (..., n).  Real scans have connected label regions; uniformly random labels
    would make the kNN-graph losses look ~C times cheaper than they are (edges only count within a class)."""
    z = xyz[..., 2]
    cut = np.quantile(z, 0.4, axis=-1, keepdims=True)
    ang = np.arctan2(xyz[..., 1], xyz[..., 0])
    sector = np.floor((ang + np.pi) / (2 * np.pi) * (num_classes - 1)).astype(np.int64)
    sector = np.clip(sector, 0, num_classes - 2) + 1
    return np.where(z < cut, 0, sector).astype(np.int64)


def make_logits(xyz, index=0, sharp=4.0, num_classes=NUM_CLASSES):
    """Per-point class logits (b, C, n) whose arg-max follows ``region_labels`` for ~9 points in 10
    (sharp * one-hot + unit Gaussian noise): what a half-trained segmentor outputs."""
    rng = np.random.default_rng(BASE_SEED + 7919 + index)
    lab = region_labels(xyz, num_classes)
    b, n = lab.shape
    logits = rng.standard_normal((b, num_classes, n)).astype(np.float32)
    np.put_along_axis(logits, lab[:, None, :], np.take_along_axis(logits, lab[:, None, :], 1) + np.float32(sharp), 1)
    return logits
