// TEST INFRASTRUCTURE ONLY.  C-ABI harness around the reference's own grid_subsampling()
// (openpoints/cpp/subsampling/grid_subsampling/grid_subsampling.cpp:4-106), which is compiled from the
// sources where they lie under /root/reference by oracle/build_ref.py into oracle/_ref/ (never copied
// here).  It replaces only the numpy glue of the reference's CPython wrapper
// (openpoints/cpp/subsampling/wrapper.cpp:58-285): flat arrays in, flat arrays out, in the reference's own
// (hash-map iteration) output order.
#include "grid_subsampling/grid_subsampling.h"

extern "C" __attribute__((visibility("default"))) int geot_gridref_subsample(
    int n, int fdim, int ldim, float dl, const float *pts, const float *feats, const int *labels, int cap,
    float *out_pts, float *out_feats, int *out_labels)
{
    std::vector<PointXYZ> in(n), out;
    for (int i = 0; i < n; ++i) in[i] = PointXYZ(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    std::vector<float> f_in, f_out;
    std::vector<int> l_in, l_out;
    if (fdim > 0) f_in.assign(feats, feats + (size_t)n * fdim);
    if (ldim > 0) l_in.assign(labels, labels + (size_t)n * ldim);
    grid_subsampling(in, out, f_in, f_out, l_in, l_out, dl, 0);
    const int m = (int)out.size();
    if (m > cap) return -m;
    for (int i = 0; i < m; ++i) {
        out_pts[3 * i] = out[i].x;
        out_pts[3 * i + 1] = out[i].y;
        out_pts[3 * i + 2] = out[i].z;
    }
    if (fdim > 0) std::copy(f_out.begin(), f_out.end(), out_feats);
    if (ldim > 0) std::copy(l_out.begin(), l_out.end(), out_labels);
    return m;
}
