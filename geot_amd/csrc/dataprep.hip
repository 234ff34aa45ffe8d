// Dataloader-side ops moved onto the GPU so that the input pipeline does not starve the
// sampling / grouping path:
//
//   grid_subsampling  openpoints/cpp/subsampling/grid_subsampling/grid_subsampling.cpp:4-106
//                     (voxel barycentres of points and features, majority label per voxel)
//   pc_norm + random-choice gather + class weights
//                     openpoints/dataset/tooth_semi/tooth_dataset.py:108-147
//
// grid_subsampling is HBM/latency-bound integer + fp32 work.  The reference builds a hash map and adds each
// point into its voxel in input order; the sums are fp32, so the order is part of the result.  Here:
//   1. bounding box (wave min/max + ordered-int atomics), 2. 64-bit voxel key per point with the reference's
//   own fp32 expression, 3. stable radix sort of (key, index) -- rocPRIM, the one library primitive in this
//   file -- which keeps the input order inside a voxel, 4. head flags + exclusive scan = voxel ids and
//   extents, 5. one thread per voxel (per feature column) adds its points in that order: bit-identical
//   barycentres.  Majority labels: per label column a second sort by (voxel, label) and a linear run scan.
// Output rows come out by ascending voxel key (the reference's order is its hash map's; oracle/np_data.py).
//
// pc_norm: the centroid is a mean over ~1e5 rows; it is accumulated in fp64 in a fixed tree (deterministic,
// and closer to the exact mean than numpy's running fp32 sum), everything after it follows numpy's fp32
// expression order.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include "geot_common.h"
#include "geot_hip.h"

namespace geot {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// hdr: [0..2] min (ordered), [3..5] max (ordered), [6] voxel count
constexpr int GS_HDR = 8;

__global__ void gs_init_kernel(uint32_t *hdr)
{
    const int t = threadIdx.x;
    if (t < 3) hdr[t] = 0xffffffffu;
    else if (t < GS_HDR) hdr[t] = 0u;
}

__global__ __launch_bounds__(256) void gs_bbox_kernel(int n, const float *__restrict__ pts, uint32_t *__restrict__ hdr)
{
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[3 * (size_t)i + a];
            mn[a] = v < mn[a] ? v : mn[a];
            mx[a] = v > mx[a] ? v : mx[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t lo = wave_min_u32(f2ord(mn[a])), hi = wave_max_u32(f2ord(mx[a]));
        if (lane_id() == 0) {
            atomicMin(&hdr[a], lo);
            atomicMax(&hdr[3 + a], hi);
        }
    }
}

struct GsGrid {
    float ox, oy, oz, dl;
    u64 nx, nxy;
};

// grid_subsampling.cpp:24-31: originCorner = floor(minCorner * (1/sampleDl)) * sampleDl; NX, NY from maxCorner
__device__ __forceinline__ GsGrid gs_grid(const uint32_t *hdr, float dl)
{
    GsGrid g;
    const float inv = 1.0f / dl;
    g.dl = dl;
    g.ox = floorf(ord2f(hdr[0]) * inv) * dl;
    g.oy = floorf(ord2f(hdr[1]) * inv) * dl;
    g.oz = floorf(ord2f(hdr[2]) * inv) * dl;
    g.nx = (u64)floorf((ord2f(hdr[3]) - g.ox) / dl) + 1ull;
    const u64 ny = (u64)floorf((ord2f(hdr[4]) - g.oy) / dl) + 1ull;
    g.nxy = g.nx * ny;
    return g;
}

__global__ __launch_bounds__(256) void gs_key_kernel(int n, float dl, const float *__restrict__ pts,
                                                     const uint32_t *__restrict__ hdr, u64 *__restrict__ keys,
                                                     int *__restrict__ idx)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const GsGrid g = gs_grid(hdr, dl);
    // grid_subsampling.cpp:52-56
    const u64 ix = (u64)floorf((pts[3 * (size_t)i] - g.ox) / dl);
    const u64 iy = (u64)floorf((pts[3 * (size_t)i + 1] - g.oy) / dl);
    const u64 iz = (u64)floorf((pts[3 * (size_t)i + 2] - g.oz) / dl);
    keys[i] = ix + g.nx * iy + g.nxy * iz;
    idx[i] = i;
}

// head flag per sorted element (1 where a new voxel starts); scanned in place into voxel ids afterwards
__global__ __launch_bounds__(256) void gs_head_kernel(int n, const u64 *__restrict__ keys, int *__restrict__ seg)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    seg[i] = (i < n && i > 0 && keys[i] != keys[i - 1]) ? 1 : 0;   // inclusive ids after an exclusive scan shifted by one
}

// seg (after the scan) holds, at sorted position i, the number of voxel starts in (0, i): with the head of
// voxel 0 at i = 0 not counted, voxel id of element i = seg[i] + (flag(i) ? 1 : 0) ... written out here as
// cell[i] and start[cell] = i for heads; count = id of the last element + 1.
__global__ __launch_bounds__(256) void gs_cells_kernel(int n, const u64 *__restrict__ keys, const int *__restrict__ seg,
                                                       int *__restrict__ cell, int *__restrict__ start,
                                                       uint32_t *__restrict__ hdr, int *__restrict__ out_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool head = i == 0 || keys[i] != keys[i - 1];
    const int c = seg[i] + ((head && i > 0) ? 1 : 0);
    cell[i] = c;
    if (head) start[c] = i;
    if (i == n - 1) {
        start[c + 1] = n;
        hdr[6] = (uint32_t)(c + 1);
        if (out_count) *out_count = c + 1;
    }
}

// one thread per (voxel, column): column 0..2 = xyz, 3.. = features.  SampledData::update_* adds in input order
// (`point += p`), grid_subsampling.cpp:85-95 divides: points by `* (1.0 / count)` (a double that becomes the float
// argument of operator*), features by `/ (float)count`.
__global__ __launch_bounds__(256) void gs_reduce_kernel(int n, int fdim, const float *__restrict__ pts,
                                                        const float *__restrict__ feats, const int *__restrict__ order,
                                                        const int *__restrict__ start, const uint32_t *__restrict__ hdr,
                                                        float *__restrict__ out_pts, float *__restrict__ out_feats)
{
    const int cols = 3 + fdim;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(t / cols), col = (int)(t % cols);
    if (c >= (int)hdr[6]) return;
    const int s = start[c], e = start[c + 1];
    float acc = 0.f;
    if (col < 3) {
        for (int j = s; j < e; ++j) acc += pts[3 * (size_t)order[j] + col];
        out_pts[3 * (size_t)c + col] = acc * (float)(1.0 / (double)(e - s));
    } else {
        const int f = col - 3;
        for (int j = s; j < e; ++j) acc += feats[(size_t)order[j] * fdim + f];
        out_feats[(size_t)c * fdim + f] = acc / (float)(e - s);
    }
}

// key for the label pass: voxel id in the high word, label (order-preserving for signed ints) in the low word
__global__ __launch_bounds__(256) void gs_label_key_kernel(int n, int ldim, int d, const int *__restrict__ labels,
                                                           const int *__restrict__ order, const int *__restrict__ cell,
                                                           u64 *__restrict__ keys)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t lab = (uint32_t)labels[(size_t)order[i] * ldim + d] ^ 0x80000000u;
    keys[i] = ((u64)(uint32_t)cell[i] << 32) | lab;
}

// one thread per voxel walks its labels (ascending) and keeps the longest run; the first of equal runs wins
// = the smallest tied label (the reference: first maximum in hash-map order, grid_subsampling.cpp:98-100).
__global__ __launch_bounds__(256) void gs_label_vote_kernel(int ldim, int d, const u64 *__restrict__ keys,
                                                            const int *__restrict__ start,
                                                            const uint32_t *__restrict__ hdr, int *__restrict__ out_labels)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (int)hdr[6]) return;
    const int s = start[c], e = start[c + 1];
    uint32_t cur = (uint32_t)keys[s], best = cur;
    int run = 0, best_run = 0;
    for (int j = s; j < e; ++j) {
        const uint32_t v = (uint32_t)keys[j];
        if (v == cur) ++run;
        else {
            if (run > best_run) { best_run = run; best = cur; }
            cur = v;
            run = 1;
        }
    }
    if (run > best_run) best = cur;
    out_labels[(size_t)c * ldim + d] = (int)(best ^ 0x80000000u);
}

struct GsLayout {
    size_t hdr, keys_a, keys_b, idx_a, idx_b, seg, bsum, cell, start, sort_tmp, total, sort_bytes;
    bool ok;
};

// rocPRIM sizes its scratch for the current device: without one the query fails and the layout is invalid
static bool sort_temp_bytes(int n, size_t &bytes)
{
    size_t pairs = 0, keys = 0;
    hipError_t e1 = rocprim::radix_sort_pairs((void *)nullptr, pairs, (u64 *)nullptr, (u64 *)nullptr, (int *)nullptr, (int *)nullptr,
                                    (size_t)n, 0u, 64u, (hipStream_t)0);
    hipError_t e2 = rocprim::radix_sort_keys((void *)nullptr, keys, (u64 *)nullptr, (u64 *)nullptr, (size_t)n, 0u, 64u,
                                             (hipStream_t)0);
    bytes = pairs > keys ? pairs : keys;
    return e1 == hipSuccess && e2 == hipSuccess;
}

static GsLayout gs_layout(int n)
{
    GsLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.hdr = take(GS_HDR * 4);
    L.keys_a = take((size_t)n * 8);
    L.keys_b = take((size_t)n * 8);
    L.idx_a = take((size_t)n * 4);
    L.idx_b = take((size_t)n * 4);
    L.seg = take(((size_t)n + 1) * 4);
    L.bsum = take((size_t)scan_blocks(n) * 4);
    L.cell = take((size_t)n * 4);
    L.start = take(((size_t)n + 1) * 4);
    L.ok = sort_temp_bytes(n, L.sort_bytes);
    L.sort_tmp = take(L.sort_bytes);
    L.total = o;
    return L;
}

// ---- pc_norm / sample ---------------------------------------------------------------------------------------
constexpr int PN_BLOCKS = 256, PN_THREADS = 256;

__global__ __launch_bounds__(PN_THREADS) void pn_sum_kernel(int n, const float *__restrict__ pts, double *__restrict__ partial)
{
    __shared__ double red[3][PN_THREADS];
    double s[3] = {0, 0, 0};
    for (int i = blockIdx.x * PN_THREADS + threadIdx.x; i < n; i += PN_BLOCKS * PN_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) s[a] += (double)pts[3 * (size_t)i + a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) red[a][threadIdx.x] = s[a];
    __syncthreads();
    for (int w = PN_THREADS / 2; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) {
#pragma unroll
            for (int a = 0; a < 3; ++a) red[a][threadIdx.x] += red[a][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

// stats = (cx, cy, cz, scale); scale is filled in (as ordered bits; non-negative floats order like uints) by pn_max
__global__ void pn_centroid_kernel(int n, const double *__restrict__ partial, float *__restrict__ stats)
{
    const int a = threadIdx.x;
    if (a < 3) {
        double s = 0;
        for (int b = 0; b < PN_BLOCKS; ++b) s += partial[b * 3 + a];
        stats[a] = (float)(s / (double)n);
    } else if (a == 3) stats[3] = 0.f;
}

// tooth_dataset.py:112: m = max(sqrt(sum(pc**2, axis=1))) with pc already centred, fp32, (x^2 + y^2) + z^2
__global__ __launch_bounds__(256) void pn_max_kernel(int n, const float *__restrict__ pts, float *__restrict__ stats)
{
    const float cx = stats[0], cy = stats[1], cz = stats[2];
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = pts[3 * (size_t)i] - cx, y = pts[3 * (size_t)i + 1] - cy, z = pts[3 * (size_t)i + 2] - cz;
        const float r = sqrtf((x * x + y * y) + z * z);
        m = r > m ? r : m;
    }
    m = wave_max_f32(m);
    if (lane_id() == 0) atomicMax((uint32_t *)&stats[3], __float_as_uint(m));
}

// out[i] = (pts[sel[i]] - centroid) / scale; labels gathered to int64; histogram of the gathered labels
__global__ __launch_bounds__(256) void pn_sample_kernel(int n, int m, int num_classes, const float *__restrict__ pts,
                                                        const int *__restrict__ labels, const long long *__restrict__ sel,
                                                        const float *__restrict__ stats, float *__restrict__ out_pts,
                                                        long long *__restrict__ out_labels, int *__restrict__ hist,
                                                        int *__restrict__ bad)
{
    extern __shared__ int lh[];
    for (int c = threadIdx.x; c < num_classes; c += blockDim.x) lh[c] = 0;
    __syncthreads();
    const float cx = stats[0], cy = stats[1], cz = stats[2], sc = stats[3];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        long long s = sel ? sel[i] : (long long)i;
        if (s < 0 || s >= n) {           // numpy would raise IndexError: flag it, keep the launch safe
            atomicOr(bad, 1);
            s = 0;
        }
        out_pts[3 * (size_t)i] = (pts[3 * (size_t)s] - cx) / sc;
        out_pts[3 * (size_t)i + 1] = (pts[3 * (size_t)s + 1] - cy) / sc;
        out_pts[3 * (size_t)i + 2] = (pts[3 * (size_t)s + 2] - cz) / sc;
        if (labels) {
            const int l = labels[s];
            out_labels[i] = (long long)l;
            if (l >= 0 && l < num_classes) atomicAdd(&lh[l], 1);
        }
    }
    __syncthreads();
    if (labels)
        for (int c = threadIdx.x; c < num_classes; c += blockDim.x)
            if (lh[c]) atomicAdd(&hist[c], lh[c]);
}

// tooth_dataset.py:143-147: counts / sum(counts), inf -> 0
__global__ void pn_weights_kernel(int num_classes, const int *__restrict__ hist, float *__restrict__ w)
{
    float total = 0.f;
    for (int c = 0; c < num_classes; ++c) total += (float)hist[c];
    for (int c = threadIdx.x; c < num_classes; c += blockDim.x) {
        const float v = (float)hist[c] / total;
        w[c] = isinf(v) ? 0.f : v;
    }
}

} // namespace geot

using namespace geot;

GEOT_EXPORT long long geot_grid_subsampling_ws_bytes(int n)
{
    if (n < 1) return 0;
    const GsLayout L = gs_layout(n);
    return L.ok ? (long long)L.total : -1;
}

GEOT_EXPORT int geot_grid_subsampling(int n, int fdim, int ldim, float sample_dl, const float *points,
                                      const float *features, const int *labels, float *out_points,
                                      float *out_features, int *out_labels, int *out_count, void *ws,
                                      long long ws_bytes, void *stream)
{
    if (n < 1 || fdim < 0 || ldim < 0 || !(sample_dl > 0.f) || !points || !out_points || !out_count || !ws)
        return hipErrorInvalidValue;
    if ((fdim > 0 && (!features || !out_features)) || (ldim > 0 && (!labels || !out_labels))) return hipErrorInvalidValue;
    const GsLayout L = gs_layout(n);
    if (!L.ok) return hipErrorNoDevice;
    if (ws_bytes < (long long)L.total) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    char *base = (char *)ws;
    uint32_t *hdr = (uint32_t *)(base + L.hdr);
    u64 *ka = (u64 *)(base + L.keys_a), *kb = (u64 *)(base + L.keys_b);
    int *ia = (int *)(base + L.idx_a), *ib = (int *)(base + L.idx_b);
    int *seg = (int *)(base + L.seg), *bsum = (int *)(base + L.bsum), *cell = (int *)(base + L.cell),
        *start = (int *)(base + L.start);
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(gs_init_kernel, dim3(1), dim3(64), 0, s, hdr);
    hipLaunchKernelGGL(gs_bbox_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, n, points, hdr);
    hipLaunchKernelGGL(gs_key_kernel, dim3(nb), dim3(256), 0, s, n, sample_dl, points, hdr, ka, ia);
    size_t tmp = L.sort_bytes;
    hipError_t e = rocprim::radix_sort_pairs((void *)(base + L.sort_tmp), tmp, ka, kb, ia, ib, (size_t)n, 0u, 64u, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gs_head_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, s, n, kb, seg);
    exclusive_scan_i32(n, seg, bsum, nullptr, s);
    hipLaunchKernelGGL(gs_cells_kernel, dim3(nb), dim3(256), 0, s, n, kb, seg, cell, start, hdr, out_count);
    const long long threads = (long long)n * (3 + fdim);
    hipLaunchKernelGGL(gs_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, n, fdim, points, features,
                       ib, start, hdr, out_points, out_features);
    for (int d = 0; d < ldim; ++d) {
        hipLaunchKernelGGL(gs_label_key_kernel, dim3(nb), dim3(256), 0, s, n, ldim, d, labels, ib, cell, ka);
        tmp = L.sort_bytes;
        e = rocprim::radix_sort_keys((void *)(base + L.sort_tmp), tmp, ka, kb, (size_t)n, 0u, 64u, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gs_label_vote_kernel, dim3(nb), dim3(256), 0, s, ldim, d, kb, start, hdr, out_labels);
    }
    return hipGetLastError();
}

GEOT_EXPORT long long geot_pc_norm_ws_bytes(void) { return (long long)(PN_BLOCKS * 3 * sizeof(double)); }

GEOT_EXPORT int geot_pc_norm_stats(int n, const float *points, float *stats, void *ws, long long ws_bytes, void *stream)
{
    if (n < 1 || !points || !stats || !ws || ws_bytes < geot_pc_norm_ws_bytes()) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pn_sum_kernel, dim3(PN_BLOCKS), dim3(PN_THREADS), 0, s, n, points, (double *)ws);
    hipLaunchKernelGGL(pn_centroid_kernel, dim3(1), dim3(64), 0, s, n, (const double *)ws, stats);
    const int nb = (n + 255) / 256;
    hipLaunchKernelGGL(pn_max_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, n, points, stats);
    return hipGetLastError();
}

GEOT_EXPORT int geot_cloud_sample(int n, int m, int num_classes, const float *points, const int *labels,
                                  const long long *selected, const float *stats, float *out_points,
                                  long long *out_labels, float *class_weights, int *hist_ws, void *stream)
{
    if (n < 1 || m < 0 || !points || !stats || !out_points || !hist_ws) return hipErrorInvalidValue;
    if (labels && (!out_labels || !class_weights || num_classes < 1 || num_classes > 4096)) return hipErrorInvalidValue;
    if (!selected && m != n) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const int nc = labels ? num_classes : 0;
    hipError_t e = zero_words(hist_ws, (long long)nc + 1, s);
    if (e != hipSuccess) return e;
    if (m == 0) return hipSuccess;
    const int nb = (m + 255) / 256;
    hipLaunchKernelGGL(pn_sample_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), (size_t)(nc + 1) * sizeof(int), s, n, m, nc,
                       points, labels, selected, stats, out_points, out_labels, hist_ws, hist_ws + nc);
    if (labels) hipLaunchKernelGGL(pn_weights_kernel, dim3(1), dim3(64), 0, s, nc, hist_ws, class_weights);
    return hipGetLastError();
}
