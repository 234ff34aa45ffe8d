// pointnet2_ext_bindings.cpp -- the binding BASELINE.json's north_star names ("a thin torch cpp_extension C-ABI"):
// a HOST-ONLY PyTorch extension with the nine functions of the reference's pointnet2._ext
// (pointnet2/_ext_src/src/bindings.cpp:9-22; argument order, returned tensors and their initial values as in
// sampling.cpp / ball_query.cpp / group_points.cpp / interpolate.cpp), each a forwarder onto the C ABI of
// include/geot_hip.h on torch's current stream.  No device code here (nothing for hipify to touch): g++ against the
// torch headers, linked with libgeot_hip.so.  The ctypes module geot_amd/ext/pointnet2_ext.py makes the same nine
// calls; tests/test_cpp_binding_gpu.py holds the two to identical outputs and times both.
#include "binding_common.h"

using namespace geot_binding;

at::Tensor gather_points(at::Tensor points, at::Tensor idx)
{
    check_f32(points, "points", 3);
    check_i32(idx, "idx", 2);
    TORCH_CHECK(idx.size(0) == points.size(0) && idx.device() == points.device(), "idx batch / device mismatch");
    c10::DeviceGuard guard(points.device());
    const int b = points.size(0), c = points.size(1), n = points.size(2), m = idx.size(1);
    at::Tensor out = n > 0 ? at::empty({b, c, m}, like(points, at::kFloat)) : at::zeros({b, c, m}, like(points, at::kFloat));
    ok(geot_gather_points(b, c, n, m, points.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(), stream_of(points)),
       "gather_points");
    return out;
}

at::Tensor gather_points_grad(at::Tensor grad_out, at::Tensor idx, const int n)
{
    check_f32(grad_out, "grad_out", 3);
    check_i32(idx, "idx", 2);
    TORCH_CHECK(idx.size(0) == grad_out.size(0) && idx.size(1) == grad_out.size(2), "idx shape mismatch");
    c10::DeviceGuard guard(grad_out.device());
    const int b = grad_out.size(0), c = grad_out.size(1), m = grad_out.size(2);
    at::Tensor out = at::zeros({b, c, n}, like(grad_out, at::kFloat));
    at::Tensor ws = grad_ws(grad_out, b, c, n, m, 1, 0);
    ok(geot_gather_points_grad_ws(b, c, n, m, grad_out.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(),
                                  ws.data_ptr<float>(), stream_of(grad_out)),
       "gather_points_grad");
    return out;
}

at::Tensor furthest_point_sampling(at::Tensor points, const int nsamples)
{
    check_f32(points, "points", 3);
    TORCH_CHECK(points.size(2) == 3, "points must be (B, N, 3)");
    c10::DeviceGuard guard(points.device());
    const int b = points.size(0), n = points.size(1);
    at::Tensor out = at::zeros({b, nsamples}, like(points, at::kInt));        // sampling.cpp:71-73
    at::Tensor tmp = at::full({b, n}, 1e10, like(points, at::kFloat));        // sampling.cpp:74-77
    ok(geot_furthest_point_sampling(b, n, nsamples, points.data_ptr<float>(), tmp.data_ptr<float>(), out.data_ptr<int>(),
                                    /*block_cap=*/512, /*skip_origin=*/1, stream_of(points)),
       "furthest_point_sampling");
    return out;
}

std::vector<at::Tensor> three_nn(at::Tensor unknowns, at::Tensor knows)
{
    check_f32(unknowns, "unknowns", 3);
    check_f32(knows, "knows", 3);
    TORCH_CHECK(knows.size(0) == unknowns.size(0) && unknowns.size(2) == 3 && knows.size(2) == 3 &&
                    knows.device() == unknowns.device(),
                "three_nn shape mismatch");
    c10::DeviceGuard guard(unknowns.device());
    const int b = unknowns.size(0), n = unknowns.size(1), m = knows.size(1);
    at::Tensor idx = m > 0 ? at::empty({b, n, 3}, like(unknowns, at::kInt)) : at::zeros({b, n, 3}, like(unknowns, at::kInt));
    at::Tensor dist2 = m > 0 ? at::empty({b, n, 3}, like(unknowns, at::kFloat)) : at::zeros({b, n, 3}, like(unknowns, at::kFloat));
    at::Tensor ws;
    long long ws_bytes = 0;
    if (geot_knn_grid_eligible(b, n, m, 3)) {          // the exact grid search needs scratch; same output either way
        ws_bytes = geot_knn_grid_ws_bytes(b, m);
        ws = at::empty({(int64_t)ws_bytes}, like(unknowns, at::kByte));
    }
    ok(geot_three_nn_ws(b, n, m, unknowns.data_ptr<float>(), knows.data_ptr<float>(), dist2.data_ptr<float>(),
                        idx.data_ptr<int>(), ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream_of(unknowns)),
       "three_nn");
    return {dist2, idx};
}

at::Tensor three_interpolate(at::Tensor points, at::Tensor idx, at::Tensor weight)
{
    check_f32(points, "points", 3);
    check_i32(idx, "idx", 3);
    check_f32(weight, "weight", 3);
    const int b = points.size(0), c = points.size(1), m = points.size(2), n = idx.size(1);
    TORCH_CHECK(idx.size(0) == b && idx.size(2) == 3 && weight.sizes() == idx.sizes(), "idx/weight must be (B, n, 3)");
    c10::DeviceGuard guard(points.device());
    at::Tensor out = m > 0 ? at::empty({b, c, n}, like(points, at::kFloat)) : at::zeros({b, c, n}, like(points, at::kFloat));
    ok(geot_three_interpolate(b, c, m, n, points.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                              out.data_ptr<float>(), stream_of(points)),
       "three_interpolate");
    return out;
}

at::Tensor three_interpolate_grad(at::Tensor grad_out, at::Tensor idx, at::Tensor weight, const int m)
{
    check_f32(grad_out, "grad_out", 3);
    check_i32(idx, "idx", 3);
    check_f32(weight, "weight", 3);
    const int b = grad_out.size(0), c = grad_out.size(1), n = grad_out.size(2);
    TORCH_CHECK(idx.size(0) == b && idx.size(1) == n && idx.size(2) == 3 && weight.sizes() == idx.sizes(),
                "idx/weight must be (B, n, 3)");
    c10::DeviceGuard guard(grad_out.device());
    if (c < 16 && geot_grad_ws_needs_zero(b, c, m, n, 3)) {     // the atomic fallback with too few channels for its 256-B rows: direct scatter
        at::Tensor out = at::zeros({b, c, m}, like(grad_out, at::kFloat));
        ok(geot_three_interpolate_grad(b, c, n, m, grad_out.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                                       out.data_ptr<float>(), stream_of(grad_out)),
           "three_interpolate_grad");
        return out;
    }
    at::Tensor out = at::empty({b, c, m}, like(grad_out, at::kFloat));        // every element is written by the call
    at::Tensor ws = grad_ws(grad_out, b, c, m, n, 3, 1);
    ok(geot_three_interpolate_grad_out(b, c, n, m, grad_out.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                                       out.data_ptr<float>(), ws.data_ptr<float>(), stream_of(grad_out)),
       "three_interpolate_grad");
    return out;
}

at::Tensor ball_query(at::Tensor new_xyz, at::Tensor xyz, const float radius, const int nsample)
{
    check_f32(new_xyz, "new_xyz", 3);
    check_f32(xyz, "xyz", 3);
    TORCH_CHECK(xyz.size(0) == new_xyz.size(0) && xyz.size(2) == 3 && new_xyz.size(2) == 3 && xyz.device() == new_xyz.device(),
                "ball_query shape mismatch");
    c10::DeviceGuard guard(xyz.device());
    const int b = new_xyz.size(0), m = new_xyz.size(1), n = xyz.size(1);
    at::Tensor idx = at::zeros({b, m, nsample}, like(xyz, at::kInt));          // ball_query.cpp:22-24
    at::Tensor ws;
    long long ws_bytes = 0;
    if (geot_ball_grid_eligible(b, n, m, radius, nsample)) {
        ws_bytes = geot_knn_grid_ws_bytes(b, n);
        ws = at::empty({(int64_t)ws_bytes}, like(xyz, at::kByte));
    }
    ok(geot_ball_query_ws(b, n, m, radius, nsample, new_xyz.data_ptr<float>(), xyz.data_ptr<float>(), idx.data_ptr<int>(),
                          ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, stream_of(xyz)),
       "ball_query");
    return idx;
}

at::Tensor group_points(at::Tensor points, at::Tensor idx)
{
    check_f32(points, "points", 3);
    check_i32(idx, "idx", 3);
    TORCH_CHECK(idx.size(0) == points.size(0) && idx.device() == points.device(), "idx batch / device mismatch");
    c10::DeviceGuard guard(points.device());
    const int b = points.size(0), c = points.size(1), n = points.size(2), np = idx.size(1), ns = idx.size(2);
    at::Tensor out = n > 0 ? at::empty({b, c, np, ns}, like(points, at::kFloat)) : at::zeros({b, c, np, ns}, like(points, at::kFloat));
    ok(geot_group_points(b, c, n, np, ns, points.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(), stream_of(points)),
       "group_points");
    return out;
}

at::Tensor group_points_grad(at::Tensor grad_out, at::Tensor idx, const int n)
{
    check_f32(grad_out, "grad_out", 4);
    check_i32(idx, "idx", 3);
    const int b = grad_out.size(0), c = grad_out.size(1), np = grad_out.size(2), ns = grad_out.size(3);
    TORCH_CHECK(idx.size(0) == b && idx.size(1) == np && idx.size(2) == ns, "idx shape mismatch");
    c10::DeviceGuard guard(grad_out.device());
    at::Tensor out = at::zeros({b, c, n}, like(grad_out, at::kFloat));        // group_points.cpp:50-52
    if (c < 16 && geot_grad_ws_needs_zero(b, c, n, (long long)np * ns, 1)) {
        ok(geot_group_points_grad(b, c, n, np, ns, grad_out.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(),
                                  stream_of(grad_out)),
           "group_points_grad");
        return out;
    }
    at::Tensor ws = grad_ws(grad_out, b, c, n, (long long)np * ns, 1, 0);
    ok(geot_group_points_grad_ws(b, c, n, np, ns, grad_out.data_ptr<float>(), idx.data_ptr<int>(), out.data_ptr<float>(),
                                 ws.data_ptr<float>(), stream_of(grad_out)),
       "group_points_grad");
    return out;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("gather_points", &gather_points);
    m.def("gather_points_grad", &gather_points_grad);
    m.def("furthest_point_sampling", &furthest_point_sampling);
    m.def("three_nn", &three_nn);
    m.def("three_interpolate", &three_interpolate);
    m.def("three_interpolate_grad", &three_interpolate_grad);
    m.def("ball_query", &ball_query);
    m.def("group_points", &group_points);
    m.def("group_points_grad", &group_points_grad);
}
