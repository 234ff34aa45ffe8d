"""Drop-in for the reference's ``pointops_cuda`` module: the union of
pointops/src/pointops_api.cpp:8-12 and openpoints/cpp/pointops/src/pointops_api.cpp:13-25
(both trees build an extension of this name; the last installed one wins in the
reference).

Outputs are caller-allocated, as in the reference (pointops/functions/pointops.py:72-74,
111-113).  ``n_max`` may arrive as a Python int or a 0-dim tensor, as the
reference passes it (pointops.py:69-71).
"""
import torch

from ._common import f32, i32, same_device, need, call, ptr


def _as_int(v):
    return int(v.item()) if isinstance(v, torch.Tensor) else int(v)


def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    f32(xyz, "xyz", 2); f32(new_xyz, "new_xyz", 2); i32(offset, "offset", 1); i32(new_offset, "new_offset", 1)
    i32(idx, "idx"); f32(dist2, "dist2")
    dev = same_device(xyz, new_xyz, offset, new_offset, idx, dist2)
    m, nsample = _as_int(m), _as_int(nsample)
    b = offset.shape[0]
    need(new_offset.shape[0] == b, "offset/new_offset length mismatch")
    need(new_xyz.shape[0] >= m and idx.numel() == m * nsample and dist2.numel() == m * nsample,
         "knnquery size mismatch")
    call("geot_knnquery_heap", dev, b, m, nsample, ptr(xyz), ptr(new_xyz), ptr(offset), ptr(new_offset),
         ptr(idx), ptr(dist2))


def knnquery_uniform(b, n_per, m_per, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    """knnquery_cuda for batches of equal segments (b x n_per support, b x m_per queries): same output through
    the grid search + tie certification (not part of the reference module; used by pointops.knn)."""
    f32(xyz, "xyz", 2); f32(new_xyz, "new_xyz", 2); i32(offset, "offset", 1); i32(new_offset, "new_offset", 1)
    i32(idx, "idx"); f32(dist2, "dist2")
    dev = same_device(xyz, new_xyz, offset, new_offset, idx, dist2)
    need(xyz.shape[0] == b * n_per and new_xyz.shape[0] == b * m_per and idx.numel() == b * m_per * nsample
         and dist2.numel() == b * m_per * nsample and offset.shape[0] == b and new_offset.shape[0] == b,
         "knnquery_uniform size mismatch")
    import torch
    from .. import _lib
    nbytes = int(_lib.load().geot_knnquery_heap_ws_bytes(int(b), int(n_per), int(m_per), int(nsample)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    call("geot_knnquery_heap_ws", dev, int(b), int(n_per), int(m_per), int(nsample), ptr(xyz), ptr(new_xyz),
         ptr(offset), ptr(new_offset), ptr(idx), ptr(dist2), ptr(ws), nbytes)


def furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx):
    f32(xyz, "xyz", 2); i32(offset, "offset", 1); i32(new_offset, "new_offset", 1); f32(tmp, "tmp"); i32(idx, "idx")
    dev = same_device(xyz, offset, new_offset, tmp, idx)
    b, n_max = _as_int(b), _as_int(n_max)
    need(offset.shape[0] == b and new_offset.shape[0] == b, "offset length must equal b")
    need(tmp.numel() == xyz.shape[0], "tmp must have one entry per point")
    call("geot_furthestsampling_offset", dev, b, n_max, ptr(xyz), ptr(offset), ptr(new_offset), None,
         ptr(tmp), ptr(idx))


def furthestsampling_weights_cuda(b, n_max, xyz, offset, new_offset, weights, tmp, idx):
    f32(xyz, "xyz", 2); i32(offset, "offset", 1); i32(new_offset, "new_offset", 1); f32(weights, "weights")
    f32(tmp, "tmp"); i32(idx, "idx")
    dev = same_device(xyz, offset, new_offset, weights, tmp, idx)
    b, n_max = _as_int(b), _as_int(n_max)
    need(offset.shape[0] == b and new_offset.shape[0] == b, "offset length must equal b")
    need(tmp.numel() == xyz.shape[0] and weights.numel() == xyz.shape[0], "tmp/weights must have one entry per point")
    call("geot_furthestsampling_offset", dev, b, n_max, ptr(xyz), ptr(offset), ptr(new_offset), ptr(weights),
         ptr(tmp), ptr(idx))


def ballquery_cuda(m, radius, nsample, xyz, new_xyz, offset, new_offset, idx):
    f32(xyz, "xyz", 2); f32(new_xyz, "new_xyz", 2); i32(offset, "offset", 1); i32(new_offset, "new_offset", 1)
    i32(idx, "idx")
    dev = same_device(xyz, new_xyz, offset, new_offset, idx)
    m, nsample = _as_int(m), _as_int(nsample)
    b = offset.shape[0]
    need(new_offset.shape[0] == b and idx.numel() == m * nsample, "ballquery size mismatch")
    call("geot_ballquery_offset", dev, b, m, float(radius), nsample, ptr(xyz), ptr(new_xyz), ptr(offset),
         ptr(new_offset), ptr(idx))
    return 1


def grouping_forward_cuda(m, nsample, c, inp, idx, out):
    f32(inp, "input", 2); i32(idx, "idx"); f32(out, "output")
    dev = same_device(inp, idx, out)
    need(idx.numel() == m * nsample and out.numel() == m * nsample * c and inp.shape[1] == c, "grouping size mismatch")
    call("geot_grouping_cl", dev, m, nsample, c, ptr(inp), ptr(idx), ptr(out))


def grouping_backward_cuda(m, nsample, c, grad_out, idx, grad_in):
    f32(grad_out, "grad_output"); i32(idx, "idx"); f32(grad_in, "grad_input", 2)
    dev = same_device(grad_out, idx, grad_in)
    need(idx.numel() == m * nsample and grad_out.numel() == m * nsample * c and grad_in.shape[1] == c,
         "grouping_backward size mismatch")
    call("geot_grouping_cl_grad", dev, m, nsample, c, ptr(grad_out), ptr(idx), ptr(grad_in))


def interpolation_forward_cuda(n, c, k, inp, idx, weight, out):
    f32(inp, "input", 2); i32(idx, "idx"); f32(weight, "weight"); f32(out, "output")
    dev = same_device(inp, idx, weight, out)
    need(idx.numel() == n * k and weight.numel() == n * k and out.numel() == n * c and inp.shape[1] == c,
         "interpolation size mismatch")
    call("geot_interpolation_cl", dev, n, c, k, ptr(inp), ptr(idx), ptr(weight), ptr(out))


def interpolation_backward_cuda(n, c, k, grad_out, idx, weight, grad_in):
    f32(grad_out, "grad_output"); i32(idx, "idx"); f32(weight, "weight"); f32(grad_in, "grad_input", 2)
    dev = same_device(grad_out, idx, weight, grad_in)
    need(idx.numel() == n * k and weight.numel() == n * k and grad_out.numel() == n * c and grad_in.shape[1] == c,
         "interpolation_backward size mismatch")
    call("geot_interpolation_cl_grad", dev, n, c, k, ptr(grad_out), ptr(idx), ptr(weight), ptr(grad_in))


def subtraction_forward_cuda(n, nsample, c, in1, in2, idx, out):
    f32(in1, "input1", 2); f32(in2, "input2", 2); i32(idx, "idx"); f32(out, "output")
    dev = same_device(in1, in2, idx, out)
    need(in1.shape[0] >= n and in1.shape[1] == c and in2.shape[1] == c and idx.numel() == n * nsample
         and out.numel() == n * nsample * c, "subtraction size mismatch")
    call("geot_subtraction_cl", dev, n, nsample, c, ptr(in1), ptr(in2), ptr(idx), ptr(out))


def subtraction_backward_cuda(n, nsample, c, idx, grad_out, grad_in1, grad_in2):
    i32(idx, "idx"); f32(grad_out, "grad_output"); f32(grad_in1, "grad_input1", 2); f32(grad_in2, "grad_input2", 2)
    dev = same_device(idx, grad_out, grad_in1, grad_in2)
    need(idx.numel() == n * nsample and grad_out.numel() == n * nsample * c and grad_in1.shape[1] == c
         and grad_in2.shape[1] == c, "subtraction_backward size mismatch")
    call("geot_subtraction_cl_grad", dev, n, nsample, c, ptr(idx), ptr(grad_out), ptr(grad_in1), ptr(grad_in2))


def aggregation_forward_cuda(n, nsample, c, w_c, inp, position, weight, idx, out):
    f32(inp, "input", 2); f32(position, "position"); f32(weight, "weight"); i32(idx, "idx"); f32(out, "output")
    dev = same_device(inp, position, weight, idx, out)
    need(inp.shape[1] == c and position.numel() == n * nsample * c and weight.numel() == n * nsample * w_c
         and idx.numel() == n * nsample and out.numel() == n * c, "aggregation size mismatch")
    call("geot_aggregation_cl", dev, n, nsample, c, w_c, ptr(inp), ptr(position), ptr(weight), ptr(idx), ptr(out))


def aggregation_backward_cuda(n, nsample, c, w_c, inp, position, weight, idx, grad_out, grad_in, grad_pos, grad_w):
    f32(inp, "input", 2); f32(position, "position"); f32(weight, "weight"); i32(idx, "idx")
    f32(grad_out, "grad_output"); f32(grad_in, "grad_input"); f32(grad_pos, "grad_position"); f32(grad_w, "grad_weight")
    dev = same_device(inp, position, weight, idx, grad_out, grad_in, grad_pos, grad_w)
    need(inp.shape[1] == c and position.numel() == n * nsample * c and weight.numel() == n * nsample * w_c
         and idx.numel() == n * nsample and grad_out.numel() == n * c and grad_in.numel() == inp.numel()
         and grad_pos.numel() == position.numel() and grad_w.numel() == weight.numel(),
         "aggregation_backward size mismatch")
    call("geot_aggregation_cl_grad", dev, n, nsample, c, w_c, ptr(inp), ptr(position), ptr(weight), ptr(idx),
         ptr(grad_out), ptr(grad_in), ptr(grad_pos), ptr(grad_w))
