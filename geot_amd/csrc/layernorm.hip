// Residual add + LayerNorm as one pass each way, for the transformer blocks' 512-token rows (transformer.py:41-104):
//   t = x + s[sample] * y + extra        (y, s, extra optional: drop-path scaled branch, position embedding)
//   z = LayerNorm(t) = (t - mean) * rstd * gamma + beta
// forward writes t and z (one launch instead of add / addcmul + layer_norm); backward takes the gradients of both and
// returns  g = g_t + LayerNorm'(g_z)  (the gradient of x and of extra), s * g (of y) and the per-block partial sums of
// d gamma / d beta (one launch + a finish instead of three LayerNorm kernels, a gradient add and a mask multiply).
// One wave per row, lane l holds columns l + 64 e; C = 64 * EPL.
#include "geot_common.h"
#include "geot_hip.h"

namespace geot {

constexpr int LN_ROWS_PER_WAVE = 2; // backward: rows a wave walks (4: 17 us + 6 us finish at 4096 x 384; 2: 11 + 4; 1: 9 + 8)

__device__ __forceinline__ float ln_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int EPL>
__global__ __launch_bounds__(256) void res_ln_fwd_kernel(int rows, int rows_per_sample, float eps, const float *__restrict__ x,
                                                         const float *__restrict__ y, const float *__restrict__ s,
                                                         const float *__restrict__ extra, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, float *__restrict__ t_out,
                                                         float *__restrict__ z_out, float *__restrict__ mean_out,
                                                         float *__restrict__ rstd_out)
{
    constexpr int C = 64 * EPL;
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * C + lane;
    const float sc = s ? s[row / rows_per_sample] : 1.f;
    float t[EPL], sum = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        float v = x[base + 64 * e];
        if (y) v = v + sc * y[base + 64 * e];
        if (extra) v = v + extra[base + 64 * e];
        t[e] = v;
        sum += v;
    }
    const float mean = ln_wave_sum(sum) * (1.f / C);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) sq = fmaf(t[e] - mean, t[e] - mean, sq);
    const float rstd = 1.f / sqrtf(ln_wave_sum(sq) * (1.f / C) + eps);
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        if (t_out) t_out[base + 64 * e] = t[e];
        z_out[base + 64 * e] = (t[e] - mean) * rstd * gamma[lane + 64 * e] + beta[lane + 64 * e];
    }
    if (lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

template <int EPL>
__global__ __launch_bounds__(256) void res_ln_bwd_kernel(int rows, int rows_per_sample, const float *__restrict__ gz,
                                                         const float *__restrict__ gt, const float *__restrict__ t,
                                                         const float *__restrict__ mean, const float *__restrict__ rstd,
                                                         const float *__restrict__ gamma, const float *__restrict__ s,
                                                         float *__restrict__ g_out, float *__restrict__ gy_out,
                                                         float *__restrict__ partial)
{
    constexpr int C = 64 * EPL;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float gam[EPL], ag[EPL], ab[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { gam[e] = gamma[lane + 64 * e]; ag[e] = 0.f; ab[e] = 0.f; }
    const int row0 = (blockIdx.x * 4 + wave) * LN_ROWS_PER_WAVE;
    // the wave's rows are independent: all their loads are issued before the first row's arithmetic
    float gzv[LN_ROWS_PER_WAVE][EPL], tv[LN_ROWS_PER_WAVE][EPL];
#pragma unroll
    for (int rr = 0; rr < LN_ROWS_PER_WAVE; ++rr) {
        const bool ok = row0 + rr < rows;
        const size_t base = (size_t)(ok ? row0 + rr : 0) * C + lane;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            gzv[rr][e] = (ok && gz) ? gz[base + 64 * e] : 0.f;
            tv[rr][e] = ok ? t[base + 64 * e] : 0.f;
        }
    }
#pragma unroll
    for (int rr = 0; rr < LN_ROWS_PER_WAVE; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) continue;
        const size_t base = (size_t)row * C + lane;
        const float mu = mean[row], rs = rstd[row];
        float a[EPL], xh[EPL], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const float g = gzv[rr][e];
            xh[e] = (tv[rr][e] - mu) * rs;
            a[e] = g * gam[e];
            s1 += a[e];
            s2 = fmaf(a[e], xh[e], s2);
            ag[e] = fmaf(g, xh[e], ag[e]);
            ab[e] += g;
        }
        const float c1 = ln_wave_sum(s1) * (1.f / C), c2 = ln_wave_sum(s2) * (1.f / C);
        const float sc = s ? s[row / rows_per_sample] : 1.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            float d = rs * (a[e] - c1 - xh[e] * c2);
            if (gt) d = d + gt[base + 64 * e];
            g_out[base + 64 * e] = d;
            if (gy_out) gy_out[base + 64 * e] = sc * d;
        }
    }
    __shared__ float sh[2][4][C];
#pragma unroll
    for (int e = 0; e < EPL; ++e) { sh[0][wave][lane + 64 * e] = ag[e]; sh[1][wave][lane + 64 * e] = ab[e]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        const int w = i / C, col = i - w * C;
        partial[((size_t)blockIdx.x * 2 + w) * C + col] = (sh[w][0][col] + sh[w][1][col]) + (sh[w][2][col] + sh[w][3][col]);
    }
}

// d gamma[col] = sum over blocks of partial[blk][0][col], d beta likewise: grid (C / 64), 16 waves take every 16th block
// (the partial rows are read one batch of 8 after the other: with 4 waves this pass was as long as the one before it)
__global__ __launch_bounds__(1024) void res_ln_finish_kernel(int nblk, int c, const float *__restrict__ partial,
                                                             float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = blockIdx.x * 64 + lane;
    float a = 0.f, b = 0.f;
    if (col < c) {
        int blk = wave;
        for (; blk + 7 * 16 < nblk; blk += 8 * 16) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = partial[((size_t)(blk + 16 * u) * 2) * c + col];
                vb[u] = partial[((size_t)(blk + 16 * u) * 2 + 1) * c + col];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { a += va[u]; b += vb[u]; }
        }
        for (; blk < nblk; blk += 16) {
            a += partial[((size_t)blk * 2) * c + col];
            b += partial[((size_t)blk * 2 + 1) * c + col];
        }
    }
    __shared__ float sh[2][16][64];
    sh[0][wave][lane] = a;
    sh[1][wave][lane] = b;
    __syncthreads();
    if (wave == 0 && col < c) {
        float ga = 0.f, gb = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) { ga += sh[0][w][lane]; gb += sh[1][w][lane]; }
        dgamma[col] = ga;
        dbeta[col] = gb;
    }
}

static inline bool ln_c_ok(int c) { return c == 128 || c == 256 || c == 384 || c == 512 || c == 768 || c == 1024; }
static inline int ln_bwd_blocks(int rows) { return (rows + 4 * LN_ROWS_PER_WAVE - 1) / (4 * LN_ROWS_PER_WAVE); }

} // namespace geot

using namespace geot;

GEOT_EXPORT int geot_res_ln_supported(int c) { return ln_c_ok(c) ? 1 : 0; }

GEOT_EXPORT long long geot_res_ln_ws_floats(int rows, int c)
{
    if (rows < 1 || !ln_c_ok(c)) return -1;
    return 2LL * ln_bwd_blocks(rows) * c;
}

#define GEOT_LN_DISPATCH(KERNEL, GRID, ...)                                                                     \
    switch (c / 64) {                                                                                           \
    case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;         \
    case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;         \
    case 6: hipLaunchKernelGGL(KERNEL<6>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;         \
    case 8: hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;         \
    case 12: hipLaunchKernelGGL(KERNEL<12>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;       \
    default: hipLaunchKernelGGL(KERNEL<16>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); break;       \
    }

GEOT_EXPORT int geot_res_ln(int rows, int c, int rows_per_sample, float eps, const float *x, const float *y, const float *s,
                            const float *extra, const float *gamma, const float *beta, float *t_out, float *z_out,
                            float *mean, float *rstd, void *stream)
{
    if (rows < 1 || !ln_c_ok(c) || rows_per_sample < 1 || !x || !gamma || !beta || !z_out || !mean || !rstd || (s && !y))
        return hipErrorInvalidValue;
    GEOT_LN_DISPATCH(res_ln_fwd_kernel, dim3((rows + 3) / 4), rows, rows_per_sample, eps, x, y, s, extra, gamma, beta, t_out,
                     z_out, mean, rstd)
    return hipGetLastError();
}

GEOT_EXPORT int geot_res_ln_grad(int rows, int c, int rows_per_sample, const float *gz, const float *gt, const float *t,
                                 const float *mean, const float *rstd, const float *gamma, const float *s, float *g_out,
                                 float *gy_out, float *dgamma, float *dbeta, float *workspace, void *stream)
{
    if (rows < 1 || !ln_c_ok(c) || rows_per_sample < 1 || !t || !mean || !rstd || !gamma || !g_out || !dgamma || !dbeta ||
        !workspace)
        return hipErrorInvalidValue;
    const int nblk = ln_bwd_blocks(rows);
    GEOT_LN_DISPATCH(res_ln_bwd_kernel, dim3(nblk), rows, rows_per_sample, gz, gt, t, mean, rstd, gamma, s, g_out, gy_out,
                     workspace)
    hipLaunchKernelGGL(res_ln_finish_kernel, dim3((c + 63) / 64), dim3(1024), 0, (hipStream_t)stream, nblk, c, workspace, dgamma,
                       dbeta);
    return hipGetLastError();
}

// ---- attention head split: (B, N, 3, H, d) projection -> q * scale, k, v as (3, B*H, N, d) ---------------------------
// (transformer.py:70-72: reshape + permute + three slices; the copy torch makes for the batched GEMMs, with the
// 1/sqrt(d) of the scores folded into q, and its gradient -- three (B*H, N, d) tensors back into one (B, N, 3, H, d)
// gradient, d q scaled -- instead of a stack + permuted copy + a pass over the (B, H, N, N) score gradient.)
namespace geot {
__global__ __launch_bounds__(256) void qkv_split_kernel(long long total4, int n, int h, int d4, float scale,
                                                        const float4 *__restrict__ qkv, float4 *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        long long r = i;                                   // out index (w, b, h, n, dd)
        const int dd = (int)(r % d4); r /= d4;
        const int nn = (int)(r % n); r /= n;
        const int hh = (int)(r % h); r /= h;
        const long long per_w = total4 / 3 / ((long long)h * n * d4);      // = B
        const int b = (int)(r % per_w), w = (int)(r / per_w);
        float4 v = qkv[((((long long)b * n + nn) * 3 + w) * h + hh) * d4 + dd];
        if (w == 0) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
        out[i] = v;
    }
}
__global__ __launch_bounds__(256) void qkv_merge_kernel(long long total4, int bsz, int n, int h, int d4, float scale,
                                                        const float4 *__restrict__ gq, const float4 *__restrict__ gk,
                                                        const float4 *__restrict__ gv, float4 *__restrict__ out)
{
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        long long r = i;                                   // out index (b, n, w, h, dd)
        const int dd = (int)(r % d4); r /= d4;
        const int hh = (int)(r % h); r /= h;
        const int w = (int)(r % 3); r /= 3;
        const int nn = (int)(r % n);
        const int b = (int)(r / n);
        const float4 *src = w == 0 ? gq : (w == 1 ? gk : gv);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (src) v = src[(((long long)b * h + hh) * n + nn) * d4 + dd];
        if (w == 0) { v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale; }
        out[i] = v;
    }
}
} // namespace geot

GEOT_EXPORT int geot_qkv_split(int b, int n, int h, int d, float scale, const float *qkv, float *out, void *stream)
{
    if (b < 1 || n < 1 || h < 1 || d < 4 || (d & 3) || !qkv || !out || (((uintptr_t)qkv | (uintptr_t)out) & 15)) return hipErrorInvalidValue;
    const long long total4 = 3LL * b * n * h * (d / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(qkv_split_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, total4, n, h, d / 4, scale,
                       (const float4 *)qkv, (float4 *)out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_qkv_split_grad(int b, int n, int h, int d, float scale, const float *gq, const float *gk, const float *gv,
                                    float *grad_qkv, void *stream)
{
    if (b < 1 || n < 1 || h < 1 || d < 4 || (d & 3) || !grad_qkv ||
        (((uintptr_t)gq | (uintptr_t)gk | (uintptr_t)gv | (uintptr_t)grad_qkv) & 15))
        return hipErrorInvalidValue;
    const long long total4 = 3LL * b * n * h * (d / 4);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(qkv_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, total4, b, n, h, d / 4, scale,
                       (const float4 *)gq, (const float4 *)gk, (const float4 *)gv, (float4 *)grad_qkv);
    return hipGetLastError();
}

// ---- row soft-max backward: gi = y * (g - sum_j g_j y_j), one wave per row, n <= 1024 a multiple of 64 -------------------
// (torch's backward is an element-wise g * y pass followed by its warp kernel)
namespace geot {
template <int EPL>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(long long rows, const float *__restrict__ g, const float *__restrict__ y,
                                                          float *__restrict__ gi)
{
    constexpr int N = 64 * EPL;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const size_t base = (size_t)row * N + lane;
    float gv[EPL], yv[EPL], dot = 0.f;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        gv[e] = g[base + 64 * e];
        yv[e] = y[base + 64 * e];
        dot = fmaf(gv[e], yv[e], dot);
    }
    dot = ln_wave_sum(dot);
#pragma unroll
    for (int e = 0; e < EPL; ++e) gi[base + 64 * e] = yv[e] * (gv[e] - dot);
}
} // namespace geot

GEOT_EXPORT int geot_softmax_grad(long long rows, int n, const float *grad, const float *y, float *grad_in, void *stream)
{
    if (rows < 0 || !(n == 64 || n == 128 || n == 256 || n == 512 || n == 1024) || !grad || !y || !grad_in) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const long long blocks = (rows + 3) / 4;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    switch (n / 64) {
    case 1: hipLaunchKernelGGL(softmax_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, rows, grad, y, grad_in); break;
    case 2: hipLaunchKernelGGL(softmax_bwd_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, rows, grad, y, grad_in); break;
    case 4: hipLaunchKernelGGL(softmax_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, rows, grad, y, grad_in); break;
    case 8: hipLaunchKernelGGL(softmax_bwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, rows, grad, y, grad_in); break;
    default: hipLaunchKernelGGL(softmax_bwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, rows, grad, y, grad_in); break;
    }
    return hipGetLastError();
}
