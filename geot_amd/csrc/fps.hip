// fps.hip -- furthest point sampling for gfx950 (MI355X).
//
// Replaces (behaviour, not code):
//   pointnet2/_ext_src/src/sampling_gpu.cu:73-232          (dense, origin-skip, block<=512)
//   openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:101-260 (dense, block<=1024)
//   pointops/src/sampling/sampling_cuda_kernel.cu:15-171, 175-349 (offset-batched, weighted)
//
// Design (MI355X-first, see DESIGN.md "FPS"):
//   * one 1024-thread workgroup (16 wave64s, 4 per SIMD) per cloud; the cloud's
//     xyz and running min-distance live in VGPRs for the whole kernel (<=24
//     points per lane => n <= 24576), so a round touches no memory except the
//     16-entry LDS exchange and one scalar load of the winner's coordinates;
//   * the reference's block-size-dependent tie rule is reproduced with an
//     explicit key  bitreverse(k mod bs) : (k div bs)  instead of inheriting
//     whatever order our own reduction has, which frees the
//     launch geometry from the reference's;
//   * arg-max = wave DPP max of the fp32 bit pattern (non-negative floats order
//     like unsigned ints) + DPP min of the key among the maxima, then one
//     LDS hop + one barrier per round (double-buffered slots).
#include "geot_common.h"
#include "geot_hip.h"
#include <cmath>

namespace geot {

constexpr int FPS_THREADS = 1024;
constexpr int FPS_WAVES = FPS_THREADS / 64;
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t fps_key(uint32_t k, int L)
{
    uint32_t low = (1u << L) - 1u;
    return __builtin_bitreverse32(k & low) | (k >> L);
}
__device__ __forceinline__ uint32_t fps_key_decode(uint32_t key, int L)
{
    uint32_t hi = L ? (0xFFFFFFFFu << (32 - L)) : 0u;
    uint32_t t = __builtin_bitreverse32(key & hi);
    uint32_t row = key & ~hi;
    return (row << L) | t;
}

__device__ __forceinline__ bool origin_skipped(float x, float y, float z)
{
    float mag = (x * x) + (y * y) + (z * z);
    return (double)mag <= 1e-3; // fp32 magnitude against a double literal, as the reference
}

__device__ __forceinline__ float weighted(float d, float w)
{
    double ww = (double)w;
    if (!(ww > 1e-12)) ww = 1e-12;
    return (float)((double)d * ww);
}

// Block-wide arg-max exchange. Returns the winning local index (0 when no
// lane has a candidate). `bits` = fp32 pattern of the lane's best value,
// `key` = its tie key (KEY_NONE when the lane has no candidate).
__device__ __forceinline__ uint32_t fps_block_argmax(uint32_t bits, uint32_t key, uint2 *slot, int L)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t wM = wave_max_u32(bits);
    uint32_t wk = wave_min_u32(bits == wM ? key : KEY_NONE);
    if (lane == 0) slot[wave] = make_uint2(wM, wk);
    __syncthreads();
    uint2 e = slot[lane & (FPS_WAVES - 1)];
    uint32_t M = row16_max_u32(e.x);
    uint32_t kk = row16_min_u32(e.x == M ? e.y : KEY_NONE);
    kk = __builtin_amdgcn_readfirstlane(kk);
    return kk == KEY_NONE ? 0u : fps_key_decode(kk, L);
}

// PPT > 0: register-resident cloud (n <= PPT*1024). PPT == 0: streaming
// fallback for larger clouds (xyz / temp re-read from L2 every round).
template <int PPT, bool SKIP, bool WEIGHTED>
__global__ __launch_bounds__(FPS_THREADS) void fps_kernel(
    const float *__restrict__ xyz, const int *__restrict__ offset,
    const int *__restrict__ new_offset, int n_dense, int m_dense,
    const float *__restrict__ weights, float *__restrict__ temp, int *__restrict__ idxs, int L)
{
    __shared__ uint2 slots[2][FPS_WAVES];
    const int bid = blockIdx.x, tid = threadIdx.x;
    int start_n, n, start_m, m, base;
    if (offset) {
        start_n = bid ? offset[bid - 1] : 0;
        n = offset[bid] - start_n;
        start_m = bid ? new_offset[bid - 1] : 0;
        m = new_offset[bid] - start_m;
        base = start_n;
    } else {
        start_n = bid * n_dense; n = n_dense; start_m = bid * m_dense; m = m_dense; base = 0;
    }
    if (m <= 0 || n <= 0) return;
    const float *P = xyz + (size_t)start_n * 3;
    float *T = temp + start_n;
    const float *W = WEIGHTED ? weights + start_n : nullptr;
    int *out = idxs + start_m;

    uint32_t old = 0;
    if (tid == 0) out[0] = base;

    if constexpr (PPT > 0) {
        float px[PPT], py[PPT], pz[PPT], t[PPT], w[WEIGHTED ? PPT : 1];
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            int k = i * FPS_THREADS + tid;
            bool in = k < n;
            px[i] = in ? P[k * 3 + 0] : 0.f;
            py[i] = in ? P[k * 3 + 1] : 0.f;
            pz[i] = in ? P[k * 3 + 2] : 0.f;
            t[i] = in ? T[k] : -1.f;
            if (SKIP && in && origin_skipped(px[i], py[i], pz[i])) t[i] = -1.f;
            if (WEIGHTED) w[i] = in ? W[k] : 0.f;
        }
        for (int j = 1; j < m; ++j) {
            const float qx = P[old * 3 + 0], qy = P[old * 3 + 1], qz = P[old * 3 + 2];
            float best = -1.f;
            int besti = 0;
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                float d = sqdist3(px[i], py[i], pz[i], qx, qy, qz);
                if (WEIGHTED) d = weighted(d, w[i]);
                float d2 = fmin_raw(d, t[i]);
                t[i] = d2;
                if (d2 > best) { best = d2; besti = i; }
            }
            bool have = best >= 0.f;
            uint32_t bits = have ? __float_as_uint(best) : 0u;
            uint32_t key = have ? fps_key((uint32_t)(besti * FPS_THREADS + tid), L) : KEY_NONE;
            old = fps_block_argmax(bits, key, slots[j & 1], L);
            if (tid == 0) out[j] = base + (int)old;
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            int k = i * FPS_THREADS + tid;
            if (k < n && !(SKIP && origin_skipped(px[i], py[i], pz[i]))) T[k] = t[i];
        }
    } else {
        for (int j = 1; j < m; ++j) {
            const float qx = P[old * 3 + 0], qy = P[old * 3 + 1], qz = P[old * 3 + 2];
            float best = -1.f;
            int bestk = 0;
            for (int k = tid; k < n; k += FPS_THREADS) {
                float x = P[k * 3 + 0], y = P[k * 3 + 1], z = P[k * 3 + 2];
                if (SKIP && origin_skipped(x, y, z)) continue;
                float d = sqdist3(x, y, z, qx, qy, qz);
                if (WEIGHTED) d = weighted(d, W[k]);
                float d2 = fmin_raw(d, T[k]);
                T[k] = d2;
                if (d2 > best) { best = d2; bestk = k; }
            }
            bool have = best >= 0.f;
            uint32_t bits = have ? __float_as_uint(best) : 0u;
            uint32_t key = have ? fps_key((uint32_t)bestk, L) : KEY_NONE;
            old = fps_block_argmax(bits, key, slots[j & 1], L);
            if (tid == 0) out[j] = base + (int)old;
        }
    }
}

template <bool SKIP, bool WEIGHTED>
static hipError_t fps_launch(int b, int n_max, const float *xyz, const int *offset,
                             const int *new_offset, int n_dense, int m_dense, const float *weights,
                             float *temp, int *idxs, int L, hipStream_t s)
{
#define GEOT_FPS_CASE(P)                                                                          \
    hipLaunchKernelGGL((fps_kernel<P, SKIP, WEIGHTED>), dim3(b), dim3(FPS_THREADS), 0, s, xyz,    \
                       offset, new_offset, n_dense, m_dense, weights, temp, idxs, L)
    if (n_max <= 1 * FPS_THREADS) GEOT_FPS_CASE(1);
    else if (n_max <= 2 * FPS_THREADS) GEOT_FPS_CASE(2);
    else if (n_max <= 4 * FPS_THREADS) GEOT_FPS_CASE(4);
    else if (n_max <= 8 * FPS_THREADS) GEOT_FPS_CASE(8);
    else if (n_max <= 16 * FPS_THREADS) GEOT_FPS_CASE(16);
    else if (n_max <= 24 * FPS_THREADS) GEOT_FPS_CASE(24);
    else GEOT_FPS_CASE(0);
#undef GEOT_FPS_CASE
    return hipGetLastError();
}

// Reference block-size rule: pointnet2/_ext_src/include/cuda_utils.h:17-21,
// pointops/src/cuda_utils.h:11-14 (floor(log2 n) through double log, capped).
static int ref_log2_block(int work, int cap)
{
    int p = (int)(std::log((double)work) / std::log(2.0));
    int v = 1 << p;
    if (v > cap) v = cap;
    if (v < 1) v = 1;
    int L = 0;
    while ((1 << (L + 1)) <= v) ++L;
    return L;
}

} // namespace geot

GEOT_EXPORT int geot_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                             int *idxs, int block_cap, int skip_origin, void *stream)
{
    if (b < 0 || n < 0 || m < 0 || (block_cap != 512 && block_cap != 1024)) return hipErrorInvalidValue;
    if (b == 0 || n == 0 || m == 0) return hipSuccess;
    int L = geot::ref_log2_block(n, block_cap);
    hipStream_t s = (hipStream_t)stream;
    if (skip_origin)
        return geot::fps_launch<true, false>(b, n, xyz, nullptr, nullptr, n, m, nullptr, temp, idxs, L, s);
    return geot::fps_launch<false, false>(b, n, xyz, nullptr, nullptr, n, m, nullptr, temp, idxs, L, s);
}

GEOT_EXPORT int geot_furthestsampling_offset(int b, int n_max, const float *xyz, const int *offset,
                                             const int *new_offset, const float *weights, float *tmp,
                                             int *idx, void *stream)
{
    if (b < 0 || n_max < 0) return hipErrorInvalidValue;
    if (b == 0 || n_max == 0) return hipSuccess;
    int L = geot::ref_log2_block(n_max, 1024);
    hipStream_t s = (hipStream_t)stream;
    if (weights)
        return geot::fps_launch<false, true>(b, n_max, xyz, offset, new_offset, 0, 0, weights, tmp, idx, L, s);
    return geot::fps_launch<false, false>(b, n_max, xyz, offset, new_offset, 0, 0, nullptr, tmp, idx, L, s);
}
