// neighbors.hip -- radius / nearest-neighbour searches for gfx950 (MI355X).
//
// Replaces (behaviour, not code):
//   ball query : pointnet2/_ext_src/src/ball_query_gpu.cu:12-57,
//                openpoints/cpp/pointnet2_batch/src/ball_query_gpu.cu:15-74,
//                openpoints/cpp/pointops/src/ballquery/ballquery_cuda_kernel.cu:13-88
//   three_nn   : pointnet2/_ext_src/src/interpolate_gpu.cu:12-71,
//                openpoints/cpp/pointnet2_batch/src/interpolate_gpu.cu:16-72
//   kNN (heap) : pointops/src/knnquery/knnquery_cuda_kernel.cu:21-116
//   kNN (sorted): contract of knn_cuda.KNN / knn_point
//
// All of these are O(queries x support) scans of fp32 coordinates that fit in
// L2; they are VALU-bound, not HBM-bound. Support points are staged through
// LDS in tiles so that every lane of a wave reads the same address
// (broadcast, conflict-free) while each lane owns one query.
#include "geot_common.h"
#include "geot_hip.h"
#include <cstdlib>

namespace geot {

// ---------------------------------------------------------------------------
// Ball query: one wave per query, the 64 lanes test 64 consecutive support
// points per step, so "the first nsample hits in ascending index order" falls
// out of a ballot + prefix popcount, and the wave stops as soon as nsample
// hits are found (same early exit as the reference's serial scan).
// ---------------------------------------------------------------------------
constexpr int BQ_WAVES = 4;

__device__ __forceinline__ void ball_query_one(const float *__restrict__ P, int n_pts, int idx_base,
                                               float qx, float qy, float qz, float r2, int nsample,
                                               int *__restrict__ hits, int *__restrict__ out)
{
    const int lane = lane_id();
    int cnt = 0;
    for (int k0 = 0; k0 < n_pts && cnt < nsample; k0 += 64) {
        int k = k0 + lane;
        bool hit = false;
        if (k < n_pts) {
            float d2 = sqdist3(qx, qy, qz, P[k * 3 + 0], P[k * 3 + 1], P[k * 3 + 2]);
            hit = d2 < r2;
        }
        unsigned long long mask = __ballot(hit);
        if (mask) {
            int rank = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && rank < nsample) hits[rank] = idx_base + k;
            cnt += __popcll(mask);
        }
    }
    if (cnt > nsample) cnt = nsample;
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    int first = cnt ? hits[0] : 0;
    for (int l = lane; l < nsample; l += 64) out[l] = l < cnt ? hits[l] : first;
}

__global__ __launch_bounds__(BQ_WAVES * 64) void ball_query_kernel(
    int b, int n, int m, float radius, int nsample, const float *__restrict__ new_xyz,
    const float *__restrict__ xyz, int *__restrict__ idx)
{
    extern __shared__ int bq_hits[];
    const int wave = threadIdx.x >> 6;
    int *hits = bq_hits + wave * nsample;
    const float r2 = radius * radius;
    const long long total = (long long)b * m;
    for (long long q = (long long)blockIdx.x * BQ_WAVES + wave; q < total;
         q += (long long)gridDim.x * BQ_WAVES) {
        int bi = (int)(q / m);
        const float *Q = new_xyz + q * 3;
        ball_query_one(xyz + (size_t)bi * n * 3, n, 0, Q[0], Q[1], Q[2], r2, nsample, hits,
                       idx + q * nsample);
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ int segment_of(int pt, const int *__restrict__ off, int b)
{
    int i = 0;
    while (i < b - 1 && !(pt < off[i])) ++i;
    return i;
}

__global__ __launch_bounds__(BQ_WAVES * 64) void ballquery_offset_kernel(
    int b, int m, float radius, int nsample, const float *__restrict__ xyz,
    const float *__restrict__ new_xyz, const int *__restrict__ offset,
    const int *__restrict__ new_offset, int *__restrict__ idx)
{
    extern __shared__ int bq_hits[];
    const int wave = threadIdx.x >> 6;
    int *hits = bq_hits + wave * nsample;
    const float r2 = radius * radius;
    for (int q = blockIdx.x * BQ_WAVES + wave; q < m; q += gridDim.x * BQ_WAVES) {
        int bt = segment_of(q, new_offset, b);
        int start = bt ? offset[bt - 1] : 0, end = offset[bt];
        const float *Q = new_xyz + (size_t)q * 3;
        ball_query_one(xyz + (size_t)start * 3, end - start, start, Q[0], Q[1], Q[2], r2, nsample,
                       hits, idx + (size_t)q * nsample);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------
// three_nn: one lane per unknown point, known points streamed through LDS.
// Result = 3 smallest by (d2, index): strict '<' while scanning k ascending.
// The reference keeps the running bests as doubles initialised to 1e40; with
// fp32 candidates that is the same as fp32 bests initialised to +inf.
// ---------------------------------------------------------------------------
constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 1024;

__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(
    int n, int m, const float *__restrict__ unknown, const float *__restrict__ known,
    float *__restrict__ dist2, int *__restrict__ idx)
{
    __shared__ float tile[NN_TILE * 3];
    const int bi = blockIdx.y;
    const float *U = unknown + (size_t)bi * n * 3;
    const float *K = known + (size_t)bi * m * 3;
    const int j = blockIdx.x * NN_THREADS + threadIdx.x;
    const bool live = j < n;
    float ux = 0, uy = 0, uz = 0;
    if (live) { ux = U[j * 3]; uy = U[j * 3 + 1]; uz = U[j * 3 + 2]; }
    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    for (int k0 = 0; k0 < m; k0 += NN_TILE) {
        int cnt = min(NN_TILE, m - k0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += NN_THREADS) tile[t] = K[(size_t)k0 * 3 + t];
        __syncthreads();
        for (int t = 0; t < cnt; ++t) {
            float d = sqdist3(ux, uy, uz, tile[t * 3], tile[t * 3 + 1], tile[t * 3 + 2]);
            if (d < b3) {
                int k = k0 + t;
                if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
                else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
                else { b3 = d; i3 = k; }
            }
        }
    }
    if (live) {
        size_t o = ((size_t)bi * n + j) * 3;
        dist2[o] = b1; dist2[o + 1] = b2; dist2[o + 2] = b3;
        idx[o] = i1; idx[o + 1] = i2; idx[o + 2] = i3;
    }
}

// ---------------------------------------------------------------------------
// kNN, both flavours: one lane per query, its candidate list in LDS laid out
// [slot][lane] (bank = lane, so the common "all lanes touch slot s" access is
// conflict-free), support points read through the L1/L2 (same address across
// the wave except at segment boundaries).
// ---------------------------------------------------------------------------
constexpr int KNN_THREADS = 64;

struct LdsList {
    float *d;
    int *i;
    __device__ __forceinline__ float &D(int s) { return d[s * KNN_THREADS]; }
    __device__ __forceinline__ int &I(int s) { return i[s * KNN_THREADS]; }
};

// Max-heap sift-down with the reference's exact comparison structure
// (knnquery_cuda_kernel.cu:21-36): take the right child only if strictly
// larger; stop only if root strictly larger than that child.
__device__ __forceinline__ void heap_sift(LdsList h, int k)
{
    int root = 0, child = 1;
    while (child < k) {
        if (child + 1 < k && h.D(child + 1) > h.D(child)) ++child;
        float dr = h.D(root), dc = h.D(child);
        if (dr > dc) return;
        h.D(root) = dc; h.D(child) = dr;
        int ir = h.I(root); h.I(root) = h.I(child); h.I(child) = ir;
        root = child;
        child = 2 * root + 1;
    }
}

// qlist / qcount (nullable): process only the listed queries (the ones the fast path could not certify)
__global__ __launch_bounds__(KNN_THREADS) void knnquery_heap_kernel(
    int b, int m, int nsample, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const int *__restrict__ offset, const int *__restrict__ new_offset, const int *__restrict__ qlist,
    const int *__restrict__ qcount, int *__restrict__ idx, float *__restrict__ dist2)
{
    extern __shared__ float knn_lds[];
    LdsList h{knn_lds + threadIdx.x, (int *)(knn_lds + nsample * KNN_THREADS) + threadIdx.x};
    const int t = blockIdx.x * KNN_THREADS + threadIdx.x;
    if (t >= (qcount ? min(*qcount, m) : m)) return;
    const int p = qlist ? qlist[t] : t;
    int bt = segment_of(p, new_offset, b);
    int start = bt ? offset[bt - 1] : 0, end = offset[bt];
    float qx = new_xyz[p * 3], qy = new_xyz[p * 3 + 1], qz = new_xyz[p * 3 + 2];
    for (int s = 0; s < nsample; ++s) { h.D(s) = 1e10f; h.I(s) = start; }
    float root = 1e10f;
    for (int i = start; i < end; ++i) {
        float d2 = sqdist3(qx, qy, qz, xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]);
        if (d2 < root) {
            h.D(0) = d2; h.I(0) = i;
            heap_sift(h, nsample);
            root = h.D(0);
        }
    }
    for (int i = nsample - 1; i > 0; --i) {
        float t = h.D(0); h.D(0) = h.D(i); h.D(i) = t;
        int u = h.I(0); h.I(0) = h.I(i); h.I(i) = u;
        heap_sift(h, i);
    }
    for (int s = 0; s < nsample; ++s) {
        idx[(size_t)p * nsample + s] = h.I(s);
        dist2[(size_t)p * nsample + s] = h.D(s);
    }
}

// Sorted list, ascending by (d2, index): a candidate enters only if strictly
// smaller than the current k-th, and is inserted after all entries <= it.
__global__ __launch_bounds__(KNN_THREADS) void knn_sorted_kernel(
    int nq, int nr, int k, const float *__restrict__ query, const float *__restrict__ ref,
    int *__restrict__ idx, float *__restrict__ dist2)
{
    extern __shared__ float knn_lds[];
    LdsList h{knn_lds + threadIdx.x, (int *)(knn_lds + k * KNN_THREADS) + threadIdx.x};
    const int bi = blockIdx.y;
    const int j = blockIdx.x * KNN_THREADS + threadIdx.x;
    if (j >= nq) return;
    const float *Q = query + ((size_t)bi * nq + j) * 3;
    const float *R = ref + (size_t)bi * nr * 3;
    float qx = Q[0], qy = Q[1], qz = Q[2];
    for (int s = 0; s < k; ++s) { h.D(s) = INFINITY; h.I(s) = 0; }
    float kth = INFINITY;
    int filled = 0;
    for (int r = 0; r < nr; ++r) {
        float d = sqdist3(qx, qy, qz, R[r * 3], R[r * 3 + 1], R[r * 3 + 2]);
        if (filled == k && !(d < kth)) continue;
        int pos = filled < k ? filled : k - 1;
        while (pos > 0 && d < h.D(pos - 1)) {
            h.D(pos) = h.D(pos - 1); h.I(pos) = h.I(pos - 1);
            --pos;
        }
        h.D(pos) = d; h.I(pos) = r;
        if (filled < k) ++filled;
        kth = h.D(k - 1);
    }
    size_t o = ((size_t)bi * nq + j) * k;
    for (int s = 0; s < k; ++s) { idx[o + s] = h.I(s); dist2[o + s] = h.D(s); }
}

// ---------------------------------------------------------------------------
// Wave-cooperative sorted kNN (k <= 64): the 64 lanes each take one reference point per
// step, the wave serves QW queries whose running best-k lists live one entry per lane
// (lane i = i-th smallest, lanes >= k hold +inf).  A step costs ~10 VALU per query for all
// 64 pairs; a candidate beating the current k-th distance tau (wave-uniform) is inserted
// with one ballot (its rank = #entries <= it, i.e. after equal distances: the (d2, index)
// order, since lanes are visited in ascending index) and one wave_shr:1 DPP shift.
// Expected insertions per query are ~k ln(n/k), so the scan dominates.
// Result identical to knn_sorted_kernel / the reference contract; also serves three_nn
// (k = 3: "3 smallest by (d2, index)", interpolate_gpu.cu:31-53).
// ---------------------------------------------------------------------------
constexpr int KW_QW = 4;     // queries per wave
constexpr int KW_WAVES = 4;  // waves per block
constexpr int DPP_WAVE_SHR1 = 0x138;

__device__ __forceinline__ float dpp_wave_shr1(float v, float fill)
{
    // lane l receives lane l-1; lane 0 keeps `fill` (bound_ctrl off => old value)
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(fill), (int)__float_as_uint(v),
                                                                 DPP_WAVE_SHR1, 0xF, 0xF, false));
}
__device__ __forceinline__ int dpp_wave_shr1(int v, int fill)
{
    return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHR1, 0xF, 0xF, false);
}

__global__ __launch_bounds__(KW_WAVES * 64) void knn_wave_kernel(
    int nq, int nr, int k, const float *__restrict__ query, const float *__restrict__ ref,
    int *__restrict__ idx, float *__restrict__ dist2)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int bi = blockIdx.y;
    const int q0 = (blockIdx.x * KW_WAVES + wave) * KW_QW;
    if (q0 >= nq) return;
    const float *Q = query + (size_t)bi * nq * 3;
    const float *R = ref + (size_t)bi * nr * 3;
    float qx[KW_QW], qy[KW_QW], qz[KW_QW], ld[KW_QW], tau[KW_QW];
    int li[KW_QW];
#pragma unroll
    for (int q = 0; q < KW_QW; ++q) {
        int j = min(q0 + q, nq - 1); // tail waves recompute the last query; stores are guarded
        qx[q] = Q[j * 3]; qy[q] = Q[j * 3 + 1]; qz[q] = Q[j * 3 + 2];
        ld[q] = INFINITY; li[q] = 0; tau[q] = INFINITY;
    }
    for (int c0 = 0; c0 < nr; c0 += 64) {
        const int r = c0 + lane;
        const bool in = r < nr;
        float rx = 0.f, ry = 0.f, rz = 0.f;
        if (in) { rx = R[r * 3]; ry = R[r * 3 + 1]; rz = R[r * 3 + 2]; }
#pragma unroll
        for (int q = 0; q < KW_QW; ++q) {
            float d = sqdist3(qx[q], qy[q], qz[q], rx, ry, rz);
            unsigned long long mask = __ballot(in && d < tau[q]);
            while (mask) {
                int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                float dc = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), l));
                if (!(dc < tau[q])) continue; // tau may have dropped within this step
                int pos = __popcll(__ballot(ld[q] <= dc));
                float sd = dpp_wave_shr1(ld[q], ld[q]);
                int si = dpp_wave_shr1(li[q], li[q]);
                ld[q] = lane > pos ? sd : (lane == pos ? dc : ld[q]);
                li[q] = lane > pos ? si : (lane == pos ? c0 + l : li[q]);
                tau[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ld[q]), k - 1));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < KW_QW; ++q) {
        int j = q0 + q;
        if (j < nq && lane < k) {
            size_t o = ((size_t)bi * nq + j) * k + lane;
            idx[o] = li[q];
            dist2[o] = ld[q];
        }
    }
}

// ---- sorted kNN in D dimensions (feature space: feature_space_loss searches its neighbours among the
// 17-dimensional soft-max vectors, utils/insT_loss.py:19) ------------------------------------------------
// Same wave-cooperative scheme as knn_wave_kernel: 64 lanes = 64 reference points per step, KW_QW queries per
// wave with their coordinates in registers, best-k list one entry per lane, ties by smaller index.
// d2 = sum_j (q_j - r_j)^2 accumulated in index order, un-contracted.  D <= KN_DMAX.
constexpr int KN_DMAX = 32;

template <int DPAD>
__global__ __launch_bounds__(KW_WAVES * 64) void knn_wave_nd_kernel(
    int nq, int nr, int d, int k, const float *__restrict__ query, const float *__restrict__ ref,
    int *__restrict__ idx, float *__restrict__ dist2)
{
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int bi = blockIdx.y;
    const int q0 = (blockIdx.x * KW_WAVES + wave) * KW_QW;
    if (q0 >= nq) return;
    const float *Q = query + (size_t)bi * nq * d;
    const float *R = ref + (size_t)bi * nr * d;
    float qv[KW_QW][DPAD], ld[KW_QW], tau[KW_QW];
    int li[KW_QW];
#pragma unroll
    for (int q = 0; q < KW_QW; ++q) {
        const int j = min(q0 + q, nq - 1);
#pragma unroll
        for (int t = 0; t < DPAD; ++t) qv[q][t] = t < d ? Q[(size_t)j * d + t] : 0.f;
        ld[q] = INFINITY; li[q] = 0; tau[q] = INFINITY;
    }
    for (int c0 = 0; c0 < nr; c0 += 64) {
        const int r = c0 + lane;
        const bool in = r < nr;
        float rv[DPAD];
#pragma unroll
        for (int t = 0; t < DPAD; ++t) rv[t] = (in && t < d) ? R[(size_t)r * d + t] : 0.f;
#pragma unroll
        for (int q = 0; q < KW_QW; ++q) {
            float dd = 0.f;
#pragma unroll
            for (int t = 0; t < DPAD; ++t) {
                const float df = qv[q][t] - rv[t]; // padded dimensions contribute exactly 0
                dd = dd + df * df;
            }
            unsigned long long mask = __ballot(in && dd < tau[q]);
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                const float dc = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(dd), l));
                if (!(dc < tau[q])) continue;
                const int pos = __popcll(__ballot(ld[q] <= dc));
                const float sd = dpp_wave_shr1(ld[q], ld[q]);
                const int si = dpp_wave_shr1(li[q], li[q]);
                ld[q] = lane > pos ? sd : (lane == pos ? dc : ld[q]);
                li[q] = lane > pos ? si : (lane == pos ? c0 + l : li[q]);
                tau[q] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ld[q]), k - 1));
            }
        }
    }
#pragma unroll
    for (int q = 0; q < KW_QW; ++q) {
        const int j = q0 + q;
        if (j < nq && lane < k) {
            const size_t o = ((size_t)bi * nq + j) * k + lane;
            idx[o] = li[q];
            dist2[o] = ld[q];
        }
    }
}

static inline int grid_cap(long long want, int cap) { return (int)(want < cap ? (want < 1 ? 1 : want) : cap); }

} // namespace geot

using namespace geot;

GEOT_EXPORT int geot_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                const float *xyz, int *idx, void *stream)
{
    if (b < 0 || n < 0 || m < 0 || nsample < 0) return hipErrorInvalidValue;
    if (b == 0 || m == 0 || nsample == 0) return hipSuccess;
    size_t lds = (size_t)BQ_WAVES * nsample * sizeof(int);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    long long blocks = ((long long)b * m + BQ_WAVES - 1) / BQ_WAVES;
    hipLaunchKernelGGL(ball_query_kernel, dim3(grid_cap(blocks, 1 << 16)), dim3(BQ_WAVES * 64), lds,
                       (hipStream_t)stream, b, n, m, radius, nsample, new_xyz, xyz, idx);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ballquery_offset(int b, int m, float radius, int nsample, const float *xyz,
                                      const float *new_xyz, const int *offset, const int *new_offset,
                                      int *idx, void *stream)
{
    if (b < 0 || m < 0 || nsample < 0) return hipErrorInvalidValue;
    if (b == 0 || m == 0 || nsample == 0) return hipSuccess;
    size_t lds = (size_t)BQ_WAVES * nsample * sizeof(int);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    long long blocks = ((long long)m + BQ_WAVES - 1) / BQ_WAVES;
    hipLaunchKernelGGL(ballquery_offset_kernel, dim3(grid_cap(blocks, 1 << 16)), dim3(BQ_WAVES * 64),
                       lds, (hipStream_t)stream, b, m, radius, nsample, xyz, new_xyz, offset,
                       new_offset, idx);
    return hipGetLastError();
}

GEOT_EXPORT int geot_three_nn(int b, int n, int m, const float *unknown, const float *known,
                              float *dist2, int *idx, void *stream)
{
    if (b < 0 || n < 0 || m < 0) return hipErrorInvalidValue;
    if (b == 0 || n == 0) return hipSuccess;
    if (b > 65535) return hipErrorInvalidValue;
    const char *e = getenv("GEOT_NN_IMPL"); // "basic" = one-lane-per-query kernels (A/B testing)
    if (e && e[0] == 'b') {
        hipLaunchKernelGGL(three_nn_kernel, dim3((n + NN_THREADS - 1) / NN_THREADS, b), dim3(NN_THREADS), 0,
                           (hipStream_t)stream, n, m, unknown, known, dist2, idx);
    } else {
        int per_block = KW_WAVES * KW_QW;
        hipLaunchKernelGGL(knn_wave_kernel, dim3((n + per_block - 1) / per_block, b), dim3(KW_WAVES * 64), 0,
                           (hipStream_t)stream, n, m, 3, unknown, known, idx, dist2);
    }
    return hipGetLastError();
}

GEOT_EXPORT int geot_knnquery_heap(int b, int m, int nsample, const float *xyz, const float *new_xyz,
                                   const int *offset, const int *new_offset, int *idx, float *dist2,
                                   void *stream)
{
    if (b < 0 || m < 0 || nsample < 0 || nsample > 256) return hipErrorInvalidValue;
    if (b == 0 || m == 0 || nsample == 0) return hipSuccess;
    size_t lds = (size_t)nsample * KNN_THREADS * 8;
    hipLaunchKernelGGL(knnquery_heap_kernel, dim3((m + KNN_THREADS - 1) / KNN_THREADS),
                       dim3(KNN_THREADS), lds, (hipStream_t)stream, b, m, nsample, xyz, new_xyz, offset,
                       new_offset, nullptr, nullptr, idx, dist2);
    return hipGetLastError();
}

// ---- pointops kNN, fast path for uniform batches ------------------------------------------------
// The reference's max-heap + heap-sort (knnquery_cuda_kernel.cu:21-108) returns the k nearest in ascending
// order; only the order among EQUAL distances depends on the heap mechanics.  So: sorted (k+1)-NN from the
// exact grid search; a query whose first k+1 distances are strictly increasing has a unique answer, which
// is copied; the others (duplicates, lattice ties, fewer than k+1 candidates) are listed and go through the
// literal heap kernel.  Identical output to geot_knnquery_heap.
__global__ __launch_bounds__(256) void knn_heap_certify_kernel(int m_total, int m_per, int n_per, int k,
                                                               const int *__restrict__ tidx, const float *__restrict__ td2,
                                                               int *__restrict__ idx, float *__restrict__ dist2,
                                                               int *__restrict__ qlist, int *__restrict__ qcount)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= m_total) return;
    const float *d = td2 + (size_t)p * (k + 1);
    bool strict = d[k] < 1e10f; // the heap never takes distances >= its 1e10 initial root
    for (int s = 0; s < k; ++s) strict = strict && d[s] < d[s + 1];
    if (strict) {
        const int base = (p / m_per) * n_per; // global index of the batch's first support point
        for (int s = 0; s < k; ++s) {
            idx[(size_t)p * k + s] = base + tidx[(size_t)p * (k + 1) + s];
            dist2[(size_t)p * k + s] = d[s];
        }
    } else {
        qlist[atomicAdd(qcount, 1)] = p;
    }
}

extern "C" long long geot_knn_grid_ws_bytes(int b, int nr);
extern "C" int geot_knn_grid_eligible(int b, int nq, int nr, int k);
extern "C" int geot_knn_sorted_ws(int b, int nq, int nr, int k, const float *query, const float *ref, int *idx,
                                  float *dist2, void *workspace, long long ws_bytes, void *stream);

GEOT_EXPORT long long geot_knnquery_heap_ws_bytes(int b, int n_per, int m_per, int nsample)
{
    if (b < 0 || n_per < 0 || m_per < 0 || nsample < 0) return -1;
    const long long mt = (long long)b * m_per;
    long long bytes = geot_knn_grid_ws_bytes(b, n_per);
    bytes += 4 * (2 * mt * (nsample + 1) + mt + 4);
    return (bytes + 15) & ~15LL;
}

GEOT_EXPORT int geot_knnquery_heap_ws(int b, int n_per, int m_per, int nsample, const float *xyz,
                                      const float *new_xyz, const int *offset, const int *new_offset, int *idx,
                                      float *dist2, void *workspace, long long ws_bytes, void *stream)
{
    if (b < 0 || n_per < 0 || m_per < 0 || nsample < 0 || nsample > 256) return hipErrorInvalidValue;
    const long long mt = (long long)b * m_per;
    if (b == 0 || mt == 0 || nsample == 0) return hipSuccess;
    if (mt > 0x7fffffffLL) return hipErrorInvalidValue;
    if (!workspace || nsample > 63 || !geot_knn_grid_eligible(b, m_per, n_per, nsample + 1) ||
        ws_bytes < geot_knnquery_heap_ws_bytes(b, n_per, m_per, nsample) || ((uintptr_t)workspace & 15) != 0)
        return geot_knnquery_heap(b, (int)mt, nsample, xyz, new_xyz, offset, new_offset, idx, dist2, stream);
    hipStream_t s = (hipStream_t)stream;
    const long long gbytes = geot_knn_grid_ws_bytes(b, n_per);
    int *tidx = (int *)((char *)workspace + gbytes);
    float *td2 = (float *)(tidx + mt * (nsample + 1));
    int *qlist = (int *)(td2 + mt * (nsample + 1));
    int *qcount = qlist + mt;
    hipError_t e = zero_words(qcount, 1, s);
    if (e != hipSuccess) return e;
    int rc = geot_knn_sorted_ws(b, m_per, n_per, nsample + 1, new_xyz, xyz, tidx, td2, workspace, gbytes, stream);
    if (rc != 0) return rc;
    hipLaunchKernelGGL(knn_heap_certify_kernel, dim3((int)((mt + 255) / 256)), dim3(256), 0, s, (int)mt, m_per, n_per,
                       nsample, tidx, td2, idx, dist2, qlist, qcount);
    size_t lds = (size_t)nsample * KNN_THREADS * 8;
    hipLaunchKernelGGL(knnquery_heap_kernel, dim3((int)((mt + KNN_THREADS - 1) / KNN_THREADS)), dim3(KNN_THREADS), lds, s, b,
                       (int)mt, nsample, xyz, new_xyz, offset, new_offset, qlist, qcount, idx, dist2);
    return hipGetLastError();
}

GEOT_EXPORT int geot_knn_sorted(int b, int nq, int nr, int k, const float *query, const float *ref,
                                int *idx, float *dist2, void *stream)
{
    if (b < 0 || nq < 0 || nr < 0 || k < 0 || k > 256) return hipErrorInvalidValue;
    if (b == 0 || nq == 0 || k == 0) return hipSuccess;
    if (b > 65535) return hipErrorInvalidValue;
    const char *e = getenv("GEOT_NN_IMPL");
    if (k <= 64 && !(e && e[0] == 'b')) {
        int per_block = KW_WAVES * KW_QW;
        hipLaunchKernelGGL(knn_wave_kernel, dim3((nq + per_block - 1) / per_block, b), dim3(KW_WAVES * 64), 0,
                           (hipStream_t)stream, nq, nr, k, query, ref, idx, dist2);
        return hipGetLastError();
    }
    size_t lds = (size_t)k * KNN_THREADS * 8;
    hipLaunchKernelGGL(knn_sorted_kernel, dim3((nq + KNN_THREADS - 1) / KNN_THREADS, b),
                       dim3(KNN_THREADS), lds, (hipStream_t)stream, nq, nr, k, query, ref, idx, dist2);
    return hipGetLastError();
}

GEOT_EXPORT int geot_knn_sorted_nd(int b, int nq, int nr, int d, int k, const float *query, const float *ref,
                                   int *idx, float *dist2, void *stream)
{
    if (b < 0 || nq < 0 || nr < 0 || d < 1 || d > KN_DMAX || k < 1 || k > 64) return hipErrorInvalidValue;
    if (b == 0 || nq == 0) return hipSuccess;
    if (b > 65535) return hipErrorInvalidValue;
    const int per_block = KW_WAVES * KW_QW;
    const dim3 grid((nq + per_block - 1) / per_block, b);
    if (d <= 8)
        hipLaunchKernelGGL(knn_wave_nd_kernel<8>, grid, dim3(KW_WAVES * 64), 0, (hipStream_t)stream, nq, nr, d, k, query, ref,
                           idx, dist2);
    else if (d <= 20)
        hipLaunchKernelGGL(knn_wave_nd_kernel<20>, grid, dim3(KW_WAVES * 64), 0, (hipStream_t)stream, nq, nr, d, k, query,
                           ref, idx, dist2);
    else
        hipLaunchKernelGGL(knn_wave_nd_kernel<32>, grid, dim3(KW_WAVES * 64), 0, (hipStream_t)stream, nq, nr, d, k, query,
                           ref, idx, dist2);
    return hipGetLastError();
}
