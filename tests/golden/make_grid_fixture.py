"""Regenerate tests/golden/grid_subsampling_ref.npz (run in the BUILD container only).

    python oracle/build_ref.py && python tests/golden/make_grid_fixture.py

Inputs: seeded clouds (with duplicate points and negative coordinates).  Expected outputs: what the
REFERENCE's own grid_subsampling() returns for them -- compiled from /root/reference by oracle/build_ref.py
into oracle/_ref/ -- in the reference's own row order.  The fixture holds inputs and outputs only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import np_data  # noqa: E402

CASES = [  # name, n, dl, fdim, ldim, scale
    ("a", 1500, 0.10, 0, 0, (1.0, 0.7, 0.4)),
    ("b", 2000, 0.05, 2, 1, (1.0, 1.0, 0.3)),
    ("c", 800, 0.30, 1, 2, (2.0, 0.5, 0.5)),
]


def main():
    assert np_data.have_reference(), "run oracle/build_ref.py first"
    out = {}
    for i, (name, n, dl, fdim, ldim, scale) in enumerate(CASES):
        rng = np.random.default_rng(4100 + i)
        p = (rng.standard_normal((n, 3)) * np.array(scale)).astype(np.float32)
        p[rng.integers(0, n, n // 40)] = p[rng.integers(0, n, n // 40)]
        f = rng.standard_normal((n, fdim)).astype(np.float32) if fdim else None
        lab = rng.integers(-1, 5, (n, ldim)).astype(np.int32) if ldim else None
        r = np_data.grid_subsampling_reference(p, f, lab, dl)
        out[name + "_points"], out[name + "_dl"] = p, np.float32(dl)
        out[name + "_ref_points"] = r["points"]
        if fdim:
            out[name + "_features"], out[name + "_ref_features"] = f, r["features"]
        if ldim:
            out[name + "_labels"], out[name + "_ref_labels"] = lab, r["labels"]
    np.savez_compressed(os.path.join(HERE, "grid_subsampling_ref.npz"), **out)
    print("wrote", sorted(out))


if __name__ == "__main__":
    main()
