// pointnet2_batch_bindings.cpp -- host-only torch extension with the nine wrappers of the reference's
// `pointnet2_batch_cuda` module (openpoints/cpp/pointnet2_batch/src/pointnet2_api.cpp:10-24; wrapper signatures:
// sampling.cpp:12-48, ball_query.cpp:17-28, group_points.cpp:11-36, interpolate.cpp:16-56): explicit sizes and
// CALLER-allocated tensors (possibly uninitialised -- every output element is written; accumulating outputs arrive
// pre-filled), each a forwarder onto the C ABI of include/geot_hip.h on torch's current stream.  No device code.
// geot_amd/ext/pointnet2_batch_cuda.py makes the same calls through ctypes; tests/test_cpp_binding_gpu.py holds the two to
// identical outputs.  The reference only checks ball_query's tensors (and exit(-1)s); every call here validates its
// tensors and raises RuntimeError.
#include "binding_common.h"

using namespace geot_binding;

int furthest_point_sampling_wrapper(int b, int n, int m, at::Tensor points, at::Tensor temp, at::Tensor idx)
{
    check_f32(points, "points"); check_f32(temp, "temp"); check_i32(idx, "idx");
    same_device({&points, &temp, &idx});
    TORCH_CHECK(points.numel() == (int64_t)b * n * 3 && temp.numel() == (int64_t)b * n && idx.numel() == (int64_t)b * m,
                "fps size mismatch");
    c10::DeviceGuard guard(points.device());
    ok(geot_furthest_point_sampling(b, n, m, points.data_ptr<float>(), temp.data_ptr<float>(), idx.data_ptr<int>(),
                                    /*block_cap=*/1024, /*skip_origin=*/0, stream_of(points)),
       "furthest_point_sampling_wrapper");
    return 1;
}

int gather_points_wrapper(int b, int c, int n, int npoints, at::Tensor points, at::Tensor idx, at::Tensor out)
{
    check_f32(points, "points"); check_i32(idx, "idx"); check_f32(out, "out");
    same_device({&points, &idx, &out});
    TORCH_CHECK(points.numel() == (int64_t)b * c * n && idx.numel() == (int64_t)b * npoints && out.numel() == (int64_t)b * c * npoints,
                "gather size mismatch");
    c10::DeviceGuard guard(points.device());
    ok(geot_gather_points(b, c, n, npoints, points.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(), stream_of(points)),
       "gather_points_wrapper");
    return 1;
}

int gather_points_grad_wrapper(int b, int c, int n, int npoints, at::Tensor grad_out, at::Tensor idx, at::Tensor grad_points)
{
    check_f32(grad_out, "grad_out"); check_i32(idx, "idx"); check_f32(grad_points, "grad_points");
    same_device({&grad_out, &idx, &grad_points});
    TORCH_CHECK(grad_out.numel() == (int64_t)b * c * npoints && idx.numel() == (int64_t)b * npoints &&
                    grad_points.numel() == (int64_t)b * c * n,
                "gather_grad size mismatch");
    c10::DeviceGuard guard(grad_out.device());
    at::Tensor ws = grad_ws(grad_out, b, c, n, npoints, 1, 0);
    ok(geot_gather_points_grad_ws(b, c, n, npoints, grad_out.data_ptr<float>(), idx.data_ptr<int>(), grad_points.data_ptr<float>(),
                                  ws.data_ptr<float>(), stream_of(grad_out)),
       "gather_points_grad_wrapper");
    return 1;
}

int ball_query_wrapper(int b, int n, int m, float radius, int nsample, at::Tensor new_xyz, at::Tensor xyz, at::Tensor idx)
{
    check_f32(new_xyz, "new_xyz"); check_f32(xyz, "xyz"); check_i32(idx, "idx");
    same_device({&new_xyz, &xyz, &idx});
    TORCH_CHECK(new_xyz.numel() == (int64_t)b * m * 3 && xyz.numel() == (int64_t)b * n * 3 && idx.numel() == (int64_t)b * m * nsample,
                "ball_query size mismatch");
    c10::DeviceGuard guard(xyz.device());
    at::Tensor ws;
    long long ws_bytes = 0;
    if (geot_ball_grid_eligible(b, n, m, radius, nsample)) {
        ws_bytes = geot_knn_grid_ws_bytes(b, n);
        ws = at::empty({(int64_t)ws_bytes}, like(xyz, at::kByte));
    }
    ok(geot_ball_query_ws(b, n, m, radius, nsample, new_xyz.data_ptr<float>(), xyz.data_ptr<float>(), idx.data_ptr<int>(),
                          ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream_of(xyz)),
       "ball_query_wrapper");
    return 1;
}

int group_points_wrapper(int b, int c, int n, int npoints, int nsample, at::Tensor points, at::Tensor idx, at::Tensor out)
{
    check_f32(points, "points"); check_i32(idx, "idx"); check_f32(out, "out");
    same_device({&points, &idx, &out});
    TORCH_CHECK(points.numel() == (int64_t)b * c * n && idx.numel() == (int64_t)b * npoints * nsample &&
                    out.numel() == (int64_t)b * c * npoints * nsample,
                "group size mismatch");
    c10::DeviceGuard guard(points.device());
    ok(geot_group_points(b, c, n, npoints, nsample, points.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(),
                         stream_of(points)),
       "group_points_wrapper");
    return 1;
}

int group_points_grad_wrapper(int b, int c, int n, int npoints, int nsample, at::Tensor grad_out, at::Tensor idx,
                              at::Tensor grad_points)
{
    check_f32(grad_out, "grad_out"); check_i32(idx, "idx"); check_f32(grad_points, "grad_points");
    same_device({&grad_out, &idx, &grad_points});
    TORCH_CHECK(grad_out.numel() == (int64_t)b * c * npoints * nsample && idx.numel() == (int64_t)b * npoints * nsample &&
                    grad_points.numel() == (int64_t)b * c * n,
                "group_grad size mismatch");
    c10::DeviceGuard guard(grad_out.device());
    if (c < 16 && geot_grad_ws_needs_zero(b, c, n, (long long)npoints * nsample, 1)) {
        ok(geot_group_points_grad(b, c, n, npoints, nsample, grad_out.data_ptr<float>(), idx.data_ptr<int>(),
                                  grad_points.data_ptr<float>(), stream_of(grad_out)),
           "group_points_grad_wrapper");
        return 1;
    }
    at::Tensor ws = grad_ws(grad_out, b, c, n, (long long)npoints * nsample, 1, 0);
    ok(geot_group_points_grad_ws(b, c, n, npoints, nsample, grad_out.data_ptr<float>(), idx.data_ptr<int>(),
                                 grad_points.data_ptr<float>(), ws.data_ptr<float>(), stream_of(grad_out)),
       "group_points_grad_wrapper");
    return 1;
}

void three_nn_wrapper(int b, int n, int m, at::Tensor unknown, at::Tensor known, at::Tensor dist2, at::Tensor idx)
{
    check_f32(unknown, "unknown"); check_f32(known, "known"); check_f32(dist2, "dist2"); check_i32(idx, "idx");
    same_device({&unknown, &known, &dist2, &idx});
    TORCH_CHECK(unknown.numel() == (int64_t)b * n * 3 && known.numel() == (int64_t)b * m * 3 && dist2.numel() == (int64_t)b * n * 3 &&
                    idx.numel() == (int64_t)b * n * 3,
                "three_nn size mismatch");
    c10::DeviceGuard guard(unknown.device());
    at::Tensor ws;
    long long ws_bytes = 0;
    if (geot_knn_grid_eligible(b, n, m, 3)) {
        ws_bytes = geot_knn_grid_ws_bytes(b, m);
        ws = at::empty({(int64_t)ws_bytes}, like(unknown, at::kByte));
    }
    ok(geot_three_nn_ws(b, n, m, unknown.data_ptr<float>(), known.data_ptr<float>(), dist2.data_ptr<float>(), idx.data_ptr<int>(),
                        ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream_of(unknown)),
       "three_nn_wrapper");
}

void three_interpolate_wrapper(int b, int c, int m, int n, at::Tensor points, at::Tensor idx, at::Tensor weight, at::Tensor out)
{
    check_f32(points, "points"); check_i32(idx, "idx"); check_f32(weight, "weight"); check_f32(out, "out");
    same_device({&points, &idx, &weight, &out});
    TORCH_CHECK(points.numel() == (int64_t)b * c * m && idx.numel() == (int64_t)b * n * 3 && weight.numel() == (int64_t)b * n * 3 &&
                    out.numel() == (int64_t)b * c * n,
                "three_interpolate size mismatch");
    c10::DeviceGuard guard(points.device());
    ok(geot_three_interpolate(b, c, m, n, points.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                              out.data_ptr<float>(), stream_of(points)),
       "three_interpolate_wrapper");
}

void three_interpolate_grad_wrapper(int b, int c, int n, int m, at::Tensor grad_out, at::Tensor idx, at::Tensor weight,
                                    at::Tensor grad_points)
{
    check_f32(grad_out, "grad_out"); check_i32(idx, "idx"); check_f32(weight, "weight"); check_f32(grad_points, "grad_points");
    same_device({&grad_out, &idx, &weight, &grad_points});
    TORCH_CHECK(grad_out.numel() == (int64_t)b * c * n && idx.numel() == (int64_t)b * n * 3 && weight.numel() == (int64_t)b * n * 3 &&
                    grad_points.numel() == (int64_t)b * c * m,
                "three_interpolate_grad size mismatch");
    c10::DeviceGuard guard(grad_out.device());
    if (c < 16 && geot_grad_ws_needs_zero(b, c, m, (long long)n, 3)) {
        ok(geot_three_interpolate_grad(b, c, n, m, grad_out.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                                       grad_points.data_ptr<float>(), stream_of(grad_out)),
           "three_interpolate_grad_wrapper");
        return;
    }
    at::Tensor ws = grad_ws(grad_out, b, c, m, (long long)n, 3, 1);
    ok(geot_three_interpolate_grad_ws(b, c, n, m, grad_out.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                                      grad_points.data_ptr<float>(), ws.data_ptr<float>(), stream_of(grad_out)),
       "three_interpolate_grad_wrapper");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("furthest_point_sampling_wrapper", &furthest_point_sampling_wrapper);
    m.def("gather_points_wrapper", &gather_points_wrapper);
    m.def("gather_points_grad_wrapper", &gather_points_grad_wrapper);
    m.def("ball_query_wrapper", &ball_query_wrapper);
    m.def("group_points_wrapper", &group_points_wrapper);
    m.def("group_points_grad_wrapper", &group_points_grad_wrapper);
    m.def("three_nn_wrapper", &three_nn_wrapper);
    m.def("three_interpolate_wrapper", &three_interpolate_wrapper);
    m.def("three_interpolate_grad_wrapper", &three_interpolate_grad_wrapper);
}
