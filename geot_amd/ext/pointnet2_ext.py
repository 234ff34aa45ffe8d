"""Drop-in for the reference's ``pointnet2._ext`` module.

Same nine callables, argument order and ownership as
pointnet2/_ext_src/src/bindings.cpp:10-21: outputs are allocated here, zero /
1e10-initialised exactly as the reference does (sampling.cpp:27-29,53-55,71-77;
ball_query.cpp:22-24; group_points.cpp:25-27,50-52; interpolate.cpp:26-31,58-60,
87-89), and returned.
"""
import torch

from ._common import (f32, i32, same_device, need, call, ptr, knn_workspace, ball_workspace, grad_workspace,
                      grad_needs_atomics, check_index)


def _fresh(shape, dtype, dev, source_len):
    """Output buffer of a forward op whose kernel writes every element: no zero-fill pass (the reference's
    torch.zeros costs a full extra write of, e.g., the 295 MB level-0 interpolation at 8 clouds).  An empty
    source (nothing to read from) keeps the zeros the reference would return."""
    return (torch.empty if source_len > 0 else torch.zeros)(shape, dtype=dtype, device=dev)


def gather_points(points, idx):
    f32(points, "points", 3); i32(idx, "idx", 2)
    dev = same_device(points, idx)
    b, c, n = points.shape
    need(idx.shape[0] == b, "idx batch mismatch")
    m = idx.shape[1]
    check_index(idx, n, "gather_points: idx")
    out = _fresh((b, c, m), torch.float32, dev, n)
    call("geot_gather_points", dev, b, c, n, m, ptr(points), ptr(idx), ptr(out))
    return out


def gather_points_grad(grad_out, idx, n):
    f32(grad_out, "grad_out", 3); i32(idx, "idx", 2)
    dev = same_device(grad_out, idx)
    b, c, m = grad_out.shape
    need(tuple(idx.shape) == (b, m), "idx shape mismatch")
    out = torch.zeros((b, c, int(n)), dtype=torch.float32, device=dev)
    ws = grad_workspace(dev, b, c, int(n), m, 1)
    call("geot_gather_points_grad_ws", dev, b, c, int(n), m, ptr(grad_out), ptr(idx), ptr(out), ptr(ws))
    return out


def furthest_point_sampling(points, nsamples):
    f32(points, "points", 3)
    need(points.shape[2] == 3, "points must be (B, N, 3)")
    dev = points.device
    b, n, _ = points.shape
    nsamples = int(nsamples)
    out = torch.zeros((b, nsamples), dtype=torch.int32, device=dev)
    tmp = torch.full((b, n), 1e10, dtype=torch.float32, device=dev)
    call("geot_furthest_point_sampling", dev, b, n, nsamples, ptr(points), ptr(tmp), ptr(out), 512, 1)
    return out


def three_nn(unknowns, knows):
    f32(unknowns, "unknowns", 3); f32(knows, "knows", 3)
    dev = same_device(unknowns, knows)
    b, n, _ = unknowns.shape
    need(knows.shape[0] == b and unknowns.shape[2] == 3 and knows.shape[2] == 3, "three_nn shape mismatch")
    m = knows.shape[1]
    idx = _fresh((b, n, 3), torch.int32, dev, m)
    dist2 = _fresh((b, n, 3), torch.float32, dev, m)
    wp, wb, _keep = knn_workspace(dev, b, n, m, 3)
    call("geot_three_nn_ws", dev, b, n, m, ptr(unknowns), ptr(knows), ptr(dist2), ptr(idx), wp, wb)
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    f32(points, "points", 3); i32(idx, "idx", 3); f32(weight, "weight", 3)
    dev = same_device(points, idx, weight)
    b, c, m = points.shape
    n = idx.shape[1]
    need(tuple(idx.shape) == (b, n, 3) and tuple(weight.shape) == (b, n, 3), "idx/weight must be (B, n, 3)")
    check_index(idx, m, "three_interpolate: idx")
    out = _fresh((b, c, n), torch.float32, dev, m)
    call("geot_three_interpolate", dev, b, c, m, n, ptr(points), ptr(idx), ptr(weight), ptr(out))
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    f32(grad_out, "grad_out", 3); i32(idx, "idx", 3); f32(weight, "weight", 3)
    dev = same_device(grad_out, idx, weight)
    b, c, n = grad_out.shape
    need(tuple(idx.shape) == (b, n, 3) and tuple(weight.shape) == (b, n, 3), "idx/weight must be (B, n, 3)")
    if c < 16 and grad_needs_atomics(b, c, int(m), n, 3):   # too few channels to fill a wave's 256-B atomic row: direct scatter
        out = torch.zeros((b, c, int(m)), dtype=torch.float32, device=dev)
        call("geot_three_interpolate_grad", dev, b, c, n, int(m), ptr(grad_out), ptr(idx), ptr(weight), ptr(out))
        return out
    out = torch.empty((b, c, int(m)), dtype=torch.float32, device=dev)      # every element is written by the call
    ws = grad_workspace(dev, b, c, int(m), n, 3)
    call("geot_three_interpolate_grad_out", dev, b, c, n, int(m), ptr(grad_out), ptr(idx), ptr(weight), ptr(out),
         ptr(ws))
    return out


def fp_weights(dist2):
    """three_nn's squared distances (B,n,3) -> the inverse-distance weights of pointnet2_modules.py:621-623."""
    f32(dist2, "dist2", 3)
    need(dist2.shape[2] == 3, "dist2 must be (B, n, 3)")
    w = torch.empty_like(dist2)
    call("geot_fp_weights", dist2.device, dist2.shape[0], dist2.shape[1], ptr(dist2), ptr(w))
    return w


def three_interpolate_into(points, idx, weight, out, ch_offset=0):
    """As three_interpolate, writing channels [ch_offset, ch_offset + c) of `out` (B, c + c_skip, n) in place."""
    f32(points, "points", 3); i32(idx, "idx", 3); f32(weight, "weight", 3); f32(out, "out", 3)
    dev = same_device(points, idx, weight, out)
    b, c, m = points.shape
    n = idx.shape[1]
    need(tuple(idx.shape) == (b, n, 3) and tuple(weight.shape) == (b, n, 3), "idx/weight must be (B, n, 3)")
    need(out.shape[0] == b and 0 <= ch_offset and ch_offset + c <= out.shape[1] and out.shape[2] == n,
         "out must be (B, >= ch_offset + c, n)")
    need(m > 0, "three_interpolate_into needs a non-empty source")
    check_index(idx, m, "three_interpolate_into: idx")
    call("geot_three_interpolate_into", dev, b, c, m, n, ptr(points), ptr(idx), ptr(weight),
         ptr(out) + 4 * int(ch_offset) * n, out.shape[1] * n)


def three_interpolate_grad_from(grad_wide, c, idx, weight, m, ch_offset=0):
    """Gradient of three_interpolate_into: reads channels [ch_offset, ch_offset + c) of grad_wide (B, c + c_skip, n)."""
    f32(grad_wide, "grad_wide", 3); i32(idx, "idx", 3); f32(weight, "weight", 3)
    dev = same_device(grad_wide, idx, weight)
    b, cw, n = grad_wide.shape
    need(c > 0 and ch_offset >= 0 and ch_offset + c <= cw and tuple(idx.shape) == (b, n, 3)
         and tuple(weight.shape) == (b, n, 3), "shape mismatch")
    out = torch.zeros((b, c, int(m)), dtype=torch.float32, device=dev)
    ws = grad_workspace(dev, b, c, int(m), n, 3)
    call("geot_three_interpolate_grad_from", dev, b, c, n, int(m), ptr(grad_wide) + 4 * int(ch_offset) * n, cw * n,
         ptr(idx), ptr(weight), ptr(out), ptr(ws))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    f32(new_xyz, "new_xyz", 3); f32(xyz, "xyz", 3)
    dev = same_device(new_xyz, xyz)
    b, m, _ = new_xyz.shape
    need(xyz.shape[0] == b and xyz.shape[2] == 3 and new_xyz.shape[2] == 3, "ball_query shape mismatch")
    n = xyz.shape[1]
    idx = torch.zeros((b, m, int(nsample)), dtype=torch.int32, device=dev)
    wp, wb, _keep = ball_workspace(dev, b, n, m, radius, nsample)
    call("geot_ball_query_ws", dev, b, n, m, float(radius), int(nsample), ptr(new_xyz), ptr(xyz), ptr(idx), wp, wb)
    return idx


def group_points(points, idx):
    f32(points, "points", 3); i32(idx, "idx", 3)
    dev = same_device(points, idx)
    b, c, n = points.shape
    need(idx.shape[0] == b, "idx batch mismatch")
    npoints, nsample = idx.shape[1], idx.shape[2]
    check_index(idx, n, "group_points: idx")
    out = _fresh((b, c, npoints, nsample), torch.float32, dev, n)
    call("geot_group_points", dev, b, c, n, npoints, nsample, ptr(points), ptr(idx), ptr(out))
    return out


def group_points_grad(grad_out, idx, n):
    f32(grad_out, "grad_out", 4); i32(idx, "idx", 3)
    dev = same_device(grad_out, idx)
    b, c, npoints, nsample = grad_out.shape
    need(tuple(idx.shape) == (b, npoints, nsample), "idx shape mismatch")
    out = torch.zeros((b, c, int(n)), dtype=torch.float32, device=dev)
    if c < 16 and grad_needs_atomics(b, c, int(n), npoints * nsample, 1):
        call("geot_group_points_grad", dev, b, c, int(n), npoints, nsample, ptr(grad_out), ptr(idx), ptr(out))
        return out
    ws = grad_workspace(dev, b, c, int(n), grad_out.shape[2] * grad_out.shape[3], 1)
    call("geot_group_points_grad_ws", dev, b, c, int(n), npoints, nsample, ptr(grad_out), ptr(idx), ptr(out),
         ptr(ws))
    return out
