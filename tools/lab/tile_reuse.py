"""How often a source row is fetched if the targets of the FP stage's gradient are taken in tiles of TT consecutive
targets of their Morton sequence and every tile stages its DISTINCT sources once (CPU, numpy + scipy; bench cloud,
24000 unknown points, 8192 known = their farthest-point samples).  Printed: sum over tiles of distinct sources / sources."""
import sys
import os

import numpy as np
from scipy.spatial import cKDTree

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geot_amd.synth import make_batch  # noqa: E402


def fps(x, m):
    d = np.full(len(x), 1e10)
    idx = np.zeros(m, dtype=np.int64)
    cur = 0
    for i in range(m):
        idx[i] = cur
        d = np.minimum(d, ((x - x[cur]) ** 2).sum(1))
        cur = int(d.argmax())
    return idx


def morton(p, bits=10):
    q = ((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9) * (2 ** bits - 1)).astype(np.uint64)
    code = np.zeros(len(p), dtype=np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return code


def main():
    x = make_batch(1, 24000)[0][0]
    known = x[fps(x.astype(np.float64), 8192)]
    _, nn = cKDTree(known).query(x, k=3)
    order = np.argsort(morton(known), kind="stable")
    rank = np.empty(8192, dtype=np.int64)
    rank[order] = np.arange(8192)
    tr = rank[nn]
    for tt in (32, 64, 96, 128, 256, 512):
        st = np.sort(tr // tt, 1)
        first = np.ones_like(st, dtype=bool)
        first[:, 1:] = st[:, 1:] != st[:, :-1]
        nd = np.bincount(st[first], minlength=8192 // tt + 1)
        print("tiles of %3d targets: %.3f fetches per source row; distinct rows per tile max %d mean %d" % (tt, nd.sum() / 24000.0, nd.max(), nd.mean()))


if __name__ == "__main__":
    main()
