// binding_common.h -- checks and stream plumbing shared by the host-only torch extension modules of csrc_torch/.
// Error behaviour as geot_amd/ext/_common.py: every violated precondition is a RuntimeError (TORCH_CHECK), never an
// exit() (the reference's pointnet2_batch wrappers exit(-1), its pointops wrappers check nothing).
#pragma once
#include <algorithm>
#include <torch/extension.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <c10/core/DeviceGuard.h>
#include <vector>

#include "geot_hip.h"

namespace geot_binding {

inline void *stream_of(const at::Tensor &t)
{
    return (void *)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
}

inline void check(const at::Tensor &t, const char *name, at::ScalarType dtype, const char *dtype_name, int64_t ndim)
{
    TORCH_CHECK(t.is_cuda(), name, ": CPU not supported (tensor must live on the GPU)");
    TORCH_CHECK(t.scalar_type() == dtype, name, " must be ", dtype_name, " tensor");
    TORCH_CHECK(t.is_contiguous(), name, " must be a contiguous tensor");
    TORCH_CHECK(ndim < 0 || t.dim() == ndim, name, " must have ", ndim, " dimensions, got ", t.dim());
}

inline void check_f32(const at::Tensor &t, const char *name, int64_t ndim = -1) { check(t, name, at::kFloat, "a float", ndim); }
inline void check_i32(const at::Tensor &t, const char *name, int64_t ndim = -1) { check(t, name, at::kInt, "an int", ndim); }

inline void same_device(std::initializer_list<const at::Tensor *> ts)
{
    const at::Tensor *first = *ts.begin();
    for (const at::Tensor *t : ts)
        TORCH_CHECK(t->device() == first->device(), "all tensors must be on the same device (", first->device(), " vs ",
                    t->device(), ")");
}

inline void ok(int err, const char *what) { TORCH_CHECK(err == 0, what, ": ", geot_error_string(err)); }

inline at::TensorOptions like(const at::Tensor &t, at::ScalarType dtype) { return at::device(t.device()).dtype(dtype); }

// scratch of the atomic-free gradients (geot_scatter_grad_ws_floats); zero-filled only for the shapes that still
// accumulate in it (geot_grad_ws_needs_zero)
inline at::Tensor grad_ws(const at::Tensor &ref, int b, int c, int m, long long sources, int slots, int weighted)
{
    const long long floats = std::max<long long>(geot_scatter_grad_ws_floats(b, c, m, sources, slots, weighted), 1);
    return geot_grad_ws_needs_zero(b, c, m, sources, slots) ? at::zeros({(int64_t)floats}, like(ref, at::kFloat))
                                                            : at::empty({(int64_t)floats}, like(ref, at::kFloat));
}

} // namespace geot_binding
