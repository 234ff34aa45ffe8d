// pointops_cuda_bindings.cpp -- host-only torch extension with the functions of the reference's `pointops_cuda` module:
// the UNION of pointops/src/pointops_api.cpp:8-12 (knnquery_cuda, furthestsampling_cuda, furthestsampling_weights_cuda)
// and openpoints/cpp/pointops/src/pointops_api.cpp:13-25 (knnquery, furthestsampling, ballquery, grouping, interpolation,
// subtraction, aggregation; forward + backward) -- both trees build an extension of this name.  Argument order as the
// reference's wrappers (e.g. knnquery_cuda.cpp:8-18, sampling_cuda.cpp:7-15, aggregation_cuda.cpp:8-33); outputs are
// CALLER-allocated; accumulating outputs (grad_*, tmp) arrive pre-filled.  Each function forwards onto the C ABI of
// include/geot_hip.h on torch's current stream; no device code.  geot_amd/ext/pointops_cuda.py is the ctypes twin,
// tests/test_cpp_binding_gpu.py holds the two to identical outputs.  Plus one extension (knnquery_uniform) that the
// package's own pointops.knn uses.  The reference checks nothing; every call here validates its tensors.
#include "binding_common.h"

using namespace geot_binding;

void knnquery_cuda(int m, int nsample, at::Tensor xyz, at::Tensor new_xyz, at::Tensor offset, at::Tensor new_offset,
                   at::Tensor idx, at::Tensor dist2)
{
    check_f32(xyz, "xyz", 2); check_f32(new_xyz, "new_xyz", 2); check_i32(offset, "offset", 1); check_i32(new_offset, "new_offset", 1);
    check_i32(idx, "idx"); check_f32(dist2, "dist2");
    same_device({&xyz, &new_xyz, &offset, &new_offset, &idx, &dist2});
    const int b = offset.size(0);
    TORCH_CHECK(new_offset.size(0) == b, "offset/new_offset length mismatch");
    TORCH_CHECK(new_xyz.size(0) >= m && idx.numel() == (int64_t)m * nsample && dist2.numel() == (int64_t)m * nsample,
                "knnquery size mismatch");
    c10::DeviceGuard guard(xyz.device());
    ok(geot_knnquery_heap(b, m, nsample, xyz.data_ptr<float>(), new_xyz.data_ptr<float>(), offset.data_ptr<int>(),
                          new_offset.data_ptr<int>(), idx.data_ptr<int>(), dist2.data_ptr<float>(), stream_of(xyz)),
       "knnquery_cuda");
}

void knnquery_uniform(int b, int n_per, int m_per, int nsample, at::Tensor xyz, at::Tensor new_xyz, at::Tensor offset,
                      at::Tensor new_offset, at::Tensor idx, at::Tensor dist2)
{
    check_f32(xyz, "xyz", 2); check_f32(new_xyz, "new_xyz", 2); check_i32(offset, "offset", 1); check_i32(new_offset, "new_offset", 1);
    check_i32(idx, "idx"); check_f32(dist2, "dist2");
    same_device({&xyz, &new_xyz, &offset, &new_offset, &idx, &dist2});
    TORCH_CHECK(xyz.size(0) == (int64_t)b * n_per && new_xyz.size(0) == (int64_t)b * m_per &&
                    idx.numel() == (int64_t)b * m_per * nsample && dist2.numel() == (int64_t)b * m_per * nsample &&
                    offset.size(0) == b && new_offset.size(0) == b,
                "knnquery_uniform size mismatch");
    c10::DeviceGuard guard(xyz.device());
    const long long ws_bytes = geot_knnquery_heap_ws_bytes(b, n_per, m_per, nsample);
    at::Tensor ws = at::empty({(int64_t)ws_bytes}, like(xyz, at::kByte));
    ok(geot_knnquery_heap_ws(b, n_per, m_per, nsample, xyz.data_ptr<float>(), new_xyz.data_ptr<float>(), offset.data_ptr<int>(),
                             new_offset.data_ptr<int>(), idx.data_ptr<int>(), dist2.data_ptr<float>(), ws.data_ptr(), ws_bytes,
                             stream_of(xyz)),
       "knnquery_uniform");
}

static void fps_offset(int b, int n_max, at::Tensor &xyz, at::Tensor &offset, at::Tensor &new_offset, const at::Tensor *weights,
                       at::Tensor &tmp, at::Tensor &idx, const char *what)
{
    check_f32(xyz, "xyz", 2); check_i32(offset, "offset", 1); check_i32(new_offset, "new_offset", 1); check_f32(tmp, "tmp");
    check_i32(idx, "idx");
    same_device({&xyz, &offset, &new_offset, &tmp, &idx});
    TORCH_CHECK(offset.size(0) == b && new_offset.size(0) == b, "offset length must equal b");
    TORCH_CHECK(tmp.numel() == xyz.size(0), "tmp must have one entry per point");
    if (weights) {
        check_f32(*weights, "weights");
        TORCH_CHECK(weights->device() == xyz.device() && weights->numel() == xyz.size(0), "weights must have one entry per point");
    }
    c10::DeviceGuard guard(xyz.device());
    ok(geot_furthestsampling_offset(b, n_max, xyz.data_ptr<float>(), offset.data_ptr<int>(), new_offset.data_ptr<int>(),
                                    weights ? weights->data_ptr<float>() : nullptr, tmp.data_ptr<float>(), idx.data_ptr<int>(),
                                    stream_of(xyz)),
       what);
}

void furthestsampling_cuda(int b, int n_max, at::Tensor xyz, at::Tensor offset, at::Tensor new_offset, at::Tensor tmp, at::Tensor idx)
{
    fps_offset(b, n_max, xyz, offset, new_offset, nullptr, tmp, idx, "furthestsampling_cuda");
}

void furthestsampling_weights_cuda(int b, int n_max, at::Tensor xyz, at::Tensor offset, at::Tensor new_offset, at::Tensor weights,
                                   at::Tensor tmp, at::Tensor idx)
{
    fps_offset(b, n_max, xyz, offset, new_offset, &weights, tmp, idx, "furthestsampling_weights_cuda");
}

int ballquery_cuda(int m, float radius, int nsample, at::Tensor xyz, at::Tensor new_xyz, at::Tensor offset, at::Tensor new_offset,
                   at::Tensor idx)
{
    check_f32(xyz, "xyz", 2); check_f32(new_xyz, "new_xyz", 2); check_i32(offset, "offset", 1); check_i32(new_offset, "new_offset", 1);
    check_i32(idx, "idx");
    same_device({&xyz, &new_xyz, &offset, &new_offset, &idx});
    const int b = offset.size(0);
    TORCH_CHECK(new_offset.size(0) == b && idx.numel() == (int64_t)m * nsample, "ballquery size mismatch");
    c10::DeviceGuard guard(xyz.device());
    ok(geot_ballquery_offset(b, m, radius, nsample, xyz.data_ptr<float>(), new_xyz.data_ptr<float>(), offset.data_ptr<int>(),
                             new_offset.data_ptr<int>(), idx.data_ptr<int>(), stream_of(xyz)),
       "ballquery_cuda");
    return 1;
}

void grouping_forward_cuda(int m, int nsample, int c, at::Tensor input, at::Tensor idx, at::Tensor output)
{
    check_f32(input, "input", 2); check_i32(idx, "idx"); check_f32(output, "output");
    same_device({&input, &idx, &output});
    TORCH_CHECK(idx.numel() == (int64_t)m * nsample && output.numel() == (int64_t)m * nsample * c && input.size(1) == c,
                "grouping size mismatch");
    c10::DeviceGuard guard(input.device());
    ok(geot_grouping_cl(m, nsample, c, input.data_ptr<float>(), idx.data_ptr<int>(), output.data_ptr<float>(), stream_of(input)),
       "grouping_forward_cuda");
}

void grouping_backward_cuda(int m, int nsample, int c, at::Tensor grad_output, at::Tensor idx, at::Tensor grad_input)
{
    check_f32(grad_output, "grad_output"); check_i32(idx, "idx"); check_f32(grad_input, "grad_input", 2);
    same_device({&grad_output, &idx, &grad_input});
    TORCH_CHECK(idx.numel() == (int64_t)m * nsample && grad_output.numel() == (int64_t)m * nsample * c && grad_input.size(1) == c,
                "grouping_backward size mismatch");
    c10::DeviceGuard guard(grad_output.device());
    ok(geot_grouping_cl_grad(m, nsample, c, grad_output.data_ptr<float>(), idx.data_ptr<int>(), grad_input.data_ptr<float>(),
                             stream_of(grad_output)),
       "grouping_backward_cuda");
}

void interpolation_forward_cuda(int n, int c, int k, at::Tensor input, at::Tensor idx, at::Tensor weight, at::Tensor output)
{
    check_f32(input, "input", 2); check_i32(idx, "idx"); check_f32(weight, "weight"); check_f32(output, "output");
    same_device({&input, &idx, &weight, &output});
    TORCH_CHECK(idx.numel() == (int64_t)n * k && weight.numel() == (int64_t)n * k && output.numel() == (int64_t)n * c &&
                    input.size(1) == c,
                "interpolation size mismatch");
    c10::DeviceGuard guard(input.device());
    ok(geot_interpolation_cl(n, c, k, input.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                             output.data_ptr<float>(), stream_of(input)),
       "interpolation_forward_cuda");
}

void interpolation_backward_cuda(int n, int c, int k, at::Tensor grad_output, at::Tensor idx, at::Tensor weight, at::Tensor grad_input)
{
    check_f32(grad_output, "grad_output"); check_i32(idx, "idx"); check_f32(weight, "weight"); check_f32(grad_input, "grad_input", 2);
    same_device({&grad_output, &idx, &weight, &grad_input});
    TORCH_CHECK(idx.numel() == (int64_t)n * k && weight.numel() == (int64_t)n * k && grad_output.numel() == (int64_t)n * c &&
                    grad_input.size(1) == c,
                "interpolation_backward size mismatch");
    c10::DeviceGuard guard(grad_output.device());
    ok(geot_interpolation_cl_grad(n, c, k, grad_output.data_ptr<float>(), idx.data_ptr<int>(), weight.data_ptr<float>(),
                                  grad_input.data_ptr<float>(), stream_of(grad_output)),
       "interpolation_backward_cuda");
}

void subtraction_forward_cuda(int n, int nsample, int c, at::Tensor input1, at::Tensor input2, at::Tensor idx, at::Tensor output)
{
    check_f32(input1, "input1", 2); check_f32(input2, "input2", 2); check_i32(idx, "idx"); check_f32(output, "output");
    same_device({&input1, &input2, &idx, &output});
    TORCH_CHECK(input1.size(0) >= n && input1.size(1) == c && input2.size(1) == c && idx.numel() == (int64_t)n * nsample &&
                    output.numel() == (int64_t)n * nsample * c,
                "subtraction size mismatch");
    c10::DeviceGuard guard(input1.device());
    ok(geot_subtraction_cl(n, nsample, c, input1.data_ptr<float>(), input2.data_ptr<float>(), idx.data_ptr<int>(),
                           output.data_ptr<float>(), stream_of(input1)),
       "subtraction_forward_cuda");
}

void subtraction_backward_cuda(int n, int nsample, int c, at::Tensor idx, at::Tensor grad_output, at::Tensor grad_input1,
                               at::Tensor grad_input2)
{
    check_i32(idx, "idx"); check_f32(grad_output, "grad_output"); check_f32(grad_input1, "grad_input1", 2);
    check_f32(grad_input2, "grad_input2", 2);
    same_device({&idx, &grad_output, &grad_input1, &grad_input2});
    TORCH_CHECK(idx.numel() == (int64_t)n * nsample && grad_output.numel() == (int64_t)n * nsample * c && grad_input1.size(1) == c &&
                    grad_input2.size(1) == c,
                "subtraction_backward size mismatch");
    c10::DeviceGuard guard(grad_output.device());
    ok(geot_subtraction_cl_grad(n, nsample, c, idx.data_ptr<int>(), grad_output.data_ptr<float>(), grad_input1.data_ptr<float>(),
                                grad_input2.data_ptr<float>(), stream_of(grad_output)),
       "subtraction_backward_cuda");
}

void aggregation_forward_cuda(int n, int nsample, int c, int w_c, at::Tensor input, at::Tensor position, at::Tensor weight,
                              at::Tensor idx, at::Tensor output)
{
    check_f32(input, "input", 2); check_f32(position, "position"); check_f32(weight, "weight"); check_i32(idx, "idx");
    check_f32(output, "output");
    same_device({&input, &position, &weight, &idx, &output});
    TORCH_CHECK(input.size(1) == c && position.numel() == (int64_t)n * nsample * c && weight.numel() == (int64_t)n * nsample * w_c &&
                    idx.numel() == (int64_t)n * nsample && output.numel() == (int64_t)n * c,
                "aggregation size mismatch");
    c10::DeviceGuard guard(input.device());
    ok(geot_aggregation_cl(n, nsample, c, w_c, input.data_ptr<float>(), position.data_ptr<float>(), weight.data_ptr<float>(),
                           idx.data_ptr<int>(), output.data_ptr<float>(), stream_of(input)),
       "aggregation_forward_cuda");
}

void aggregation_backward_cuda(int n, int nsample, int c, int w_c, at::Tensor input, at::Tensor position, at::Tensor weight,
                               at::Tensor idx, at::Tensor grad_output, at::Tensor grad_input, at::Tensor grad_position,
                               at::Tensor grad_weight)
{
    check_f32(input, "input", 2); check_f32(position, "position"); check_f32(weight, "weight"); check_i32(idx, "idx");
    check_f32(grad_output, "grad_output"); check_f32(grad_input, "grad_input"); check_f32(grad_position, "grad_position");
    check_f32(grad_weight, "grad_weight");
    same_device({&input, &position, &weight, &idx, &grad_output, &grad_input, &grad_position, &grad_weight});
    TORCH_CHECK(input.size(1) == c && position.numel() == (int64_t)n * nsample * c && weight.numel() == (int64_t)n * nsample * w_c &&
                    idx.numel() == (int64_t)n * nsample && grad_output.numel() == (int64_t)n * c &&
                    grad_input.numel() == input.numel() && grad_position.numel() == position.numel() &&
                    grad_weight.numel() == weight.numel(),
                "aggregation_backward size mismatch");
    c10::DeviceGuard guard(input.device());
    ok(geot_aggregation_cl_grad(n, nsample, c, w_c, input.data_ptr<float>(), position.data_ptr<float>(), weight.data_ptr<float>(),
                                idx.data_ptr<int>(), grad_output.data_ptr<float>(), grad_input.data_ptr<float>(),
                                grad_position.data_ptr<float>(), grad_weight.data_ptr<float>(), stream_of(input)),
       "aggregation_backward_cuda");
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m)
{
    m.def("knnquery_cuda", &knnquery_cuda);
    m.def("knnquery_uniform", &knnquery_uniform);
    m.def("furthestsampling_cuda", &furthestsampling_cuda);
    m.def("furthestsampling_weights_cuda", &furthestsampling_weights_cuda);
    m.def("ballquery_cuda", &ballquery_cuda);
    m.def("grouping_forward_cuda", &grouping_forward_cuda);
    m.def("grouping_backward_cuda", &grouping_backward_cuda);
    m.def("interpolation_forward_cuda", &interpolation_forward_cuda);
    m.def("interpolation_backward_cuda", &interpolation_backward_cuda);
    m.def("subtraction_forward_cuda", &subtraction_forward_cuda);
    m.def("subtraction_backward_cuda", &subtraction_backward_cuda);
    m.def("aggregation_forward_cuda", &aggregation_forward_cuda);
    m.def("aggregation_backward_cuda", &aggregation_backward_cuda);
}
