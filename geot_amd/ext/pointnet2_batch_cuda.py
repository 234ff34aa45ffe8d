"""Drop-in for the reference's ``pointnet2_batch_cuda`` module
(openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24; re-exported as
``openpoints.cpp.pointnet2_batch.pointnet2_cuda``).

Explicit sizes + caller-allocated tensors, as the reference; unlike the
reference (which only checks ball_query and exit(-1)s) every call validates its
tensors and raises RuntimeError.  Outputs may be uninitialised
(torch.cuda.FloatTensor(...)); they are written in full.
"""
from ._common import (f32, i32, same_device, need, call, ptr, knn_workspace, ball_workspace, grad_workspace,
                      grad_needs_atomics)


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    f32(points, "points"); f32(temp, "temp"); i32(idx, "idx")
    dev = same_device(points, temp, idx)
    need(points.numel() == b * n * 3 and temp.numel() == b * n and idx.numel() == b * m, "fps size mismatch")
    call("geot_furthest_point_sampling", dev, b, n, m, ptr(points), ptr(temp), ptr(idx), 1024, 0)
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    f32(points, "points"); i32(idx, "idx"); f32(out, "out")
    dev = same_device(points, idx, out)
    need(points.numel() == b * c * n and idx.numel() == b * npoints and out.numel() == b * c * npoints,
         "gather size mismatch")
    call("geot_gather_points", dev, b, c, n, npoints, ptr(points), ptr(idx), ptr(out))
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    f32(grad_out, "grad_out"); i32(idx, "idx"); f32(grad_points, "grad_points")
    dev = same_device(grad_out, idx, grad_points)
    need(grad_out.numel() == b * c * npoints and idx.numel() == b * npoints and grad_points.numel() == b * c * n,
         "gather_grad size mismatch")
    ws = grad_workspace(dev, b, c, n, npoints, 1)
    call("geot_gather_points_grad_ws", dev, b, c, n, npoints, ptr(grad_out), ptr(idx), ptr(grad_points), ptr(ws))
    return 1


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    f32(new_xyz, "new_xyz"); f32(xyz, "xyz"); i32(idx, "idx")
    dev = same_device(new_xyz, xyz, idx)
    need(new_xyz.numel() == b * m * 3 and xyz.numel() == b * n * 3 and idx.numel() == b * m * nsample,
         "ball_query size mismatch")
    wp, wb, _keep = ball_workspace(dev, b, n, m, radius, nsample)
    call("geot_ball_query_ws", dev, b, n, m, float(radius), int(nsample), ptr(new_xyz), ptr(xyz), ptr(idx), wp, wb)
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    f32(points, "points"); i32(idx, "idx"); f32(out, "out")
    dev = same_device(points, idx, out)
    need(points.numel() == b * c * n and idx.numel() == b * npoints * nsample
         and out.numel() == b * c * npoints * nsample, "group size mismatch")
    call("geot_group_points", dev, b, c, n, npoints, nsample, ptr(points), ptr(idx), ptr(out))
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    f32(grad_out, "grad_out"); i32(idx, "idx"); f32(grad_points, "grad_points")
    dev = same_device(grad_out, idx, grad_points)
    need(grad_out.numel() == b * c * npoints * nsample and idx.numel() == b * npoints * nsample
         and grad_points.numel() == b * c * n, "group_grad size mismatch")
    if c < 16 and grad_needs_atomics(b, c, n, npoints * nsample, 1):
        call("geot_group_points_grad", dev, b, c, n, npoints, nsample, ptr(grad_out), ptr(idx), ptr(grad_points))
        return 1
    ws = grad_workspace(dev, b, c, n, npoints * nsample, 1)
    call("geot_group_points_grad_ws", dev, b, c, n, npoints, nsample, ptr(grad_out), ptr(idx), ptr(grad_points),
         ptr(ws))
    return 1


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    f32(unknown, "unknown"); f32(known, "known"); f32(dist2, "dist2"); i32(idx, "idx")
    dev = same_device(unknown, known, dist2, idx)
    need(unknown.numel() == b * n * 3 and known.numel() == b * m * 3 and dist2.numel() == b * n * 3
         and idx.numel() == b * n * 3, "three_nn size mismatch")
    wp, wb, _keep = knn_workspace(dev, b, n, m, 3)
    call("geot_three_nn_ws", dev, b, n, m, ptr(unknown), ptr(known), ptr(dist2), ptr(idx), wp, wb)


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    f32(points, "points"); i32(idx, "idx"); f32(weight, "weight"); f32(out, "out")
    dev = same_device(points, idx, weight, out)
    need(points.numel() == b * c * m and idx.numel() == b * n * 3 and weight.numel() == b * n * 3
         and out.numel() == b * c * n, "three_interpolate size mismatch")
    call("geot_three_interpolate", dev, b, c, m, n, ptr(points), ptr(idx), ptr(weight), ptr(out))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    f32(grad_out, "grad_out"); i32(idx, "idx"); f32(weight, "weight"); f32(grad_points, "grad_points")
    dev = same_device(grad_out, idx, weight, grad_points)
    need(grad_out.numel() == b * c * n and idx.numel() == b * n * 3 and weight.numel() == b * n * 3
         and grad_points.numel() == b * c * m, "three_interpolate_grad size mismatch")
    if c < 16 and grad_needs_atomics(b, c, m, n, 3):
        call("geot_three_interpolate_grad", dev, b, c, n, m, ptr(grad_out), ptr(idx), ptr(weight), ptr(grad_points))
        return
    ws = grad_workspace(dev, b, c, m, n, 3)
    call("geot_three_interpolate_grad_ws", dev, b, c, n, m, ptr(grad_out), ptr(idx), ptr(weight), ptr(grad_points),
         ptr(ws))
