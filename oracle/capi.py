"""ctypes front end of oracle/libgeot_oracle.so (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Function names follow the
reference extension entry points they restate; each C function cites the
reference file:line it follows.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# the oracle follows the same GEOT_DISTANCE switch as the HIP library (exact | fma | fma_xy): one twin per mode
DISTANCE = os.environ.get("GEOT_DISTANCE", "exact")
_SO_NAME = {"exact": "libgeot_oracle.so", "fma": "libgeot_oracle_fma.so", "fma_xy": "libgeot_oracle_fma_xy.so"}[DISTANCE]
_SO = os.path.join(_HERE, _SO_NAME)
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "geot_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, _SO_NAME])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.geot_ref_block_size.restype = ctypes.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def block_size(n, cap):
    return int(lib().geot_ref_block_size(int(n), int(cap)))


def set_threads(n):
    """-> the number of OpenMP threads the C restatement will use from now on."""
    return int(lib().geot_ref_set_threads(int(n)))


def fps_dense(xyz, m, cap=512, skip_origin=True, return_temp=False):
    """xyz (B,N,3) -> idx (B,m) int32.  cap=512+skip: pointnet2._ext; cap=1024: pointnet2_batch."""
    xyz = _f32(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, dtype=np.float32)
    idx = np.zeros((b, m), dtype=np.int32)
    lib().geot_ref_fps_dense(b, n, m, _fp(xyz), _fp(temp), _ip(idx), int(cap), int(bool(skip_origin)))
    return (idx, temp) if return_temp else idx


def fps_offset(xyz, offset, new_offset, weights=None, n_max=None, return_temp=False):
    """xyz (n,3), offset (b), new_offset (b) -> idx (new_offset[-1]) int32 global indices."""
    xyz = _f32(xyz)
    offset, new_offset = _i32(offset), _i32(new_offset)
    b = offset.shape[0]
    if n_max is None:
        n_max = int(np.diff(np.concatenate([[0], offset])).max())
    tmp = np.full((xyz.shape[0],), 1e10, dtype=np.float32)
    idx = np.zeros((int(new_offset[-1]),), dtype=np.int32)
    w = None if weights is None else _f32(weights)
    lib().geot_ref_fps_offset(b, int(n_max), _fp(xyz), _ip(offset), _ip(new_offset),
                              _fp(w) if w is not None else None, _fp(tmp), _ip(idx))
    return (idx, tmp) if return_temp else idx


def gather_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), dtype=np.float32)
    lib().geot_ref_gather_points(b, c, n, m, _fp(points), _ip(idx), _fp(out))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, m = grad_out.shape
    g = np.zeros((b, c, n), dtype=np.float32)
    lib().geot_ref_gather_points_grad(b, c, n, m, _fp(grad_out), _ip(idx), _fp(g))
    return g


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, xyz = _f32(new_xyz), _f32(xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    idx = np.zeros((b, m, nsample), dtype=np.int32)
    lib().geot_ref_ball_query(b, n, m, ctypes.c_float(radius), nsample, _fp(new_xyz), _fp(xyz), _ip(idx))
    return idx


def ballquery_offset(radius, nsample, xyz, new_xyz, offset, new_offset):
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    offset, new_offset = _i32(offset), _i32(new_offset)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    lib().geot_ref_ballquery_offset(m, ctypes.c_float(radius), nsample, _fp(xyz), _fp(new_xyz),
                                    _ip(offset), _ip(new_offset), _ip(idx))
    return idx


def group_points(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, c, n = points.shape
    _, npnt, ns = idx.shape
    out = np.zeros((b, c, npnt, ns), dtype=np.float32)
    lib().geot_ref_group_points(b, c, n, npnt, ns, _fp(points), _ip(idx), _fp(out))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    b, c, npnt, ns = grad_out.shape
    g = np.zeros((b, c, n), dtype=np.float32)
    lib().geot_ref_group_points_grad(b, c, n, npnt, ns, _fp(grad_out), _ip(idx), _fp(g))
    return g


def three_nn(unknown, known):
    unknown, known = _f32(unknown), _f32(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), dtype=np.float32)
    idx = np.zeros((b, n, 3), dtype=np.int32)
    lib().geot_ref_three_nn(b, n, m, _fp(unknown), _fp(known), _fp(dist2), _ip(idx))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), dtype=np.float32)
    lib().geot_ref_three_interpolate(b, c, m, n, _fp(points), _ip(idx), _fp(weight), _fp(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    b, c, n = grad_out.shape
    g = np.zeros((b, c, m), dtype=np.float32)
    lib().geot_ref_three_interpolate_grad(b, c, n, m, _fp(grad_out), _ip(idx), _fp(weight), _fp(g))
    return g


def knnquery_heap(nsample, xyz, new_xyz, offset, new_offset):
    """pointops.knnquery_cuda semantics -> (idx (m,ns) int32 global, dist2 (m,ns))."""
    xyz, new_xyz = _f32(xyz), _f32(new_xyz)
    offset, new_offset = _i32(offset), _i32(new_offset)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    d2 = np.zeros((m, nsample), dtype=np.float32)
    lib().geot_ref_knnquery_heap(m, nsample, _fp(xyz), _fp(new_xyz), _ip(offset), _ip(new_offset),
                                 _ip(idx), _fp(d2))
    return idx, d2


def knn_sorted(query, ref, k):
    """query (B,Q,3), ref (B,R,3) -> (idx (B,Q,k) int32, dist2 (B,Q,k)) ascending (d2, index)."""
    query, ref = _f32(query), _f32(ref)
    b, nq, _ = query.shape
    nr = ref.shape[1]
    idx = np.zeros((b, nq, k), dtype=np.int32)
    d2 = np.zeros((b, nq, k), dtype=np.float32)
    lib().geot_ref_knn_sorted(b, nq, nr, k, _fp(query), _fp(ref), _ip(idx), _fp(d2))
    return idx, d2


def grouping_cl(inp, idx):
    inp, idx = _f32(inp), _i32(idx)
    m, ns = idx.shape
    c = inp.shape[1]
    out = np.zeros((m, ns, c), dtype=np.float32)
    lib().geot_ref_grouping_cl(m, ns, c, _fp(inp), _ip(idx), _fp(out))
    return out


def grouping_cl_grad(grad_out, idx, n):
    grad_out, idx = _f32(grad_out), _i32(idx)
    m, ns, c = grad_out.shape
    g = np.zeros((n, c), dtype=np.float32)
    lib().geot_ref_grouping_cl_grad(m, ns, c, _fp(grad_out), _ip(idx), _fp(g))
    return g


def interpolation_cl(inp, idx, weight):
    inp, idx, weight = _f32(inp), _i32(idx), _f32(weight)
    n, k = idx.shape
    c = inp.shape[1]
    out = np.zeros((n, c), dtype=np.float32)
    lib().geot_ref_interpolation_cl(n, c, k, _fp(inp), _ip(idx), _fp(weight), _fp(out))
    return out


def interpolation_cl_grad(grad_out, idx, weight, m):
    grad_out, idx, weight = _f32(grad_out), _i32(idx), _f32(weight)
    n, c = grad_out.shape
    k = idx.shape[1]
    g = np.zeros((m, c), dtype=np.float32)
    lib().geot_ref_interpolation_cl_grad(n, c, k, _fp(grad_out), _ip(idx), _fp(weight), _fp(g))
    return g


def subtraction_cl(in1, in2, idx):
    in1, in2, idx = _f32(in1), _f32(in2), _i32(idx)
    n, ns = idx.shape
    c = in1.shape[1]
    out = np.zeros((n, ns, c), dtype=np.float32)
    lib().geot_ref_subtraction_cl(n, ns, c, _fp(in1), _fp(in2), _ip(idx), _fp(out))
    return out


def subtraction_cl_grad(idx, grad_out):
    idx, grad_out = _i32(idx), _f32(grad_out)
    n, ns, c = grad_out.shape
    g1 = np.zeros((n, c), dtype=np.float32)
    g2 = np.zeros((n, c), dtype=np.float32)
    lib().geot_ref_subtraction_cl_grad(n, ns, c, _ip(idx), _fp(grad_out), _fp(g1), _fp(g2))
    return g1, g2


def aggregation_cl(inp, position, weight, idx):
    inp, position, weight, idx = _f32(inp), _f32(position), _f32(weight), _i32(idx)
    n, ns, c = position.shape
    w_c = weight.shape[2]
    out = np.zeros((n, c), dtype=np.float32)
    lib().geot_ref_aggregation_cl(n, ns, c, w_c, _fp(inp), _fp(position), _fp(weight), _ip(idx), _fp(out))
    return out


def aggregation_cl_grad(inp, position, weight, idx, grad_out):
    inp, position, weight, idx, grad_out = _f32(inp), _f32(position), _f32(weight), _i32(idx), _f32(grad_out)
    n, ns, c = position.shape
    w_c = weight.shape[2]
    g_in = np.zeros_like(inp)
    g_pos = np.zeros_like(position)
    g_w = np.zeros_like(weight)
    lib().geot_ref_aggregation_cl_grad(n, ns, c, w_c, _fp(inp), _fp(position), _fp(weight), _ip(idx),
                                       _fp(grad_out), _fp(g_in), _fp(g_pos), _fp(g_w))
    return g_in, g_pos, g_w
