"""AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU oracle ("ASan/UBSan on the CPU
restatement"; GPU ASan is not available on this pool).  Every entry point of oracle/geot_oracle.c is exercised on small
inputs incl. the edge cases the tests use (m > n, tiny clouds, duplicates).  Run by tests/test_oracle_sanitize_cpu.py:

    gcc -O1 -g -fsanitize=address,undefined -shared -o $TMP/libgeot_oracle_asan.so oracle/geot_oracle.c -fopenmp -lm
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python tools/oracle_sanitize.py $TMP/libgeot_oracle_asan.so
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import capi  # noqa: E402

capi._SO = sys.argv[1]
capi._lib = None
import numpy as np
from geot_amd.synth import make_batch
rng = np.random.default_rng(0)
xyz = make_batch(2, 1500, dup_frac=0.02)[0]
capi.set_threads(2)
idx = capi.fps_dense(xyz, 300, 512, True); capi.fps_dense(xyz, 300, 1024, False, return_temp=True)
off = np.array([1500, 3000], np.int32); noff = np.array([200, 500], np.int32)
capi.fps_offset(xyz.reshape(-1, 3), off, noff); capi.fps_offset(xyz.reshape(-1, 3), off, noff, weights=rng.random(3000).astype(np.float32))
c = np.take_along_axis(xyz, idx[..., None].astype(np.int64).repeat(3, -1), 1)
bq = capi.ball_query(c, xyz, 0.1, 16); capi.ballquery_offset(0.1, 16, xyz.reshape(-1, 3), c.reshape(-1, 3), off, np.array([300, 600], np.int32))
f = rng.standard_normal((2, 5, 1500)).astype(np.float32)
g = capi.group_points(f, bq); capi.group_points_grad(g, bq, 1500); ga = capi.gather_points(f, idx); capi.gather_points_grad(ga, idx, 1500)
d2, i3 = capi.three_nn(xyz, c); w = rng.random((2, 1500, 3)).astype(np.float32)
fk = rng.standard_normal((2, 5, 300)).astype(np.float32)
o = capi.three_interpolate(fk, i3, w); capi.three_interpolate_grad(o, i3, w, 300)
capi.knnquery_heap(7, xyz.reshape(-1, 3), c.reshape(-1, 3), off, np.array([300, 600], np.int32)); capi.knn_sorted(c, xyz, 9)
x_cl = rng.standard_normal((3000, 6)).astype(np.float32); nb = rng.integers(0, 3000, (3000, 4)).astype(np.int32)
gc = capi.grouping_cl(x_cl, nb); capi.grouping_cl_grad(gc, nb, 3000)
wt = rng.random((3000, 4)).astype(np.float32)
ic = capi.interpolation_cl(x_cl, nb, wt); capi.interpolation_cl_grad(ic, nb, wt, 3000)
sc = capi.subtraction_cl(x_cl, x_cl, nb); capi.subtraction_cl_grad(nb, sc)
pos = rng.standard_normal((3000, 4, 6)).astype(np.float32); w2 = rng.random((3000, 4, 3)).astype(np.float32)
ag = capi.aggregation_cl(x_cl, pos, w2, nb); capi.aggregation_cl_grad(x_cl, pos, w2, nb, ag)
# edge cases: m > n, tiny clouds, empty segments
capi.fps_dense(xyz[:, :3], 8, 512, True); capi.three_nn(xyz[:, :10], xyz[:, :2]); capi.knn_sorted(xyz[:, :5], xyz[:, :3], 6)
print("sanitizer run complete: every oracle entry point exercised")
