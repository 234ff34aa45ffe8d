// bnrelu.hip -- BatchNorm (+ ReLU) over channels-first (B, C, L) tensors as HBM-bound streaming kernels, and the
// PointnetFPModule front end fused with the statistics of the BatchNorm behind it (gfx950 / MI355X).
//
// Callers (behaviour, not code): the SharedMLP stages conv1x1 -> BatchNorm -> ReLU of pointnet2/pytorch_utils.py:8-117
// as PointnetFPModule (pointnet2/pointnet2_modules.py:597-642) and the mini-PointNet Encoder
// (openpoints/models/backbone/transformer.py:106-136) run them in training mode.  Stock PyTorch makes 5 passes over
// the tensor forward (statistics; normalise read + write; ReLU read + write) and 8 backward; at 8 clouds x 24 000
// points x 1536 channels one pass is 1.18 GB.  Here: forward = statistics pass (free when the producer is the FP
// front-end kernel below) + ONE apply pass (scale/shift/ReLU); backward = ONE reduce pass + ONE apply pass, the ReLU
// mask recomputed from x instead of stored.  The per-channel arithmetic between the passes (mean, variance, running
// statistics, SyncBatchNorm's all-reduce) is O(C) and stays in the Python wrapper (geot_amd/fused_norm.py).
//
// Algorithmic bytes: apply 8 B/element; backward reduce 8 B, backward apply 12 B; FP front end 4 (C n + C m) + 24 n + 4 Cs n.
#include "geot_common.h"
#include "geot_hip.h"

namespace geot {

constexpr int BN_THREADS = 256;
constexpr int BN_MAX_SLICES = 32;

__device__ __forceinline__ float bn_wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// two block-wide sums, written by thread 0 / 1 to dst[0], dst[1] (fixed order: deterministic)
__device__ __forceinline__ void bn_block_store2(float a, float b, float *__restrict__ dst)
{
    __shared__ float red[BN_THREADS / 64][2];
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x < 2) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < BN_THREADS / 64; ++w) t += red[w][threadIdx.x];
        dst[threadIdx.x] = t;
    }
}

// Batch statistics are accumulated SHIFTED: a slice sums (x - p) and (x - p)^2 around a pivot p = its own first
// element, in fp32, and hands over (s1, s2, p, count); the fp64 pass that adds the slices up rebuilds
// sum x = s1 + n p and sum x^2 = s2 + 2 p s1 + n p^2 exactly.  With plain fp32 sums of x and x^2 the variance
// E[x^2] - mean^2 loses (mean / std)^2 of its digits before any fp64 arithmetic sees them.
// grid (slices, C, B): partial[((b * C + c) * slices + s) * 4 + {0..3}] = s1, s2, pivot, count of the slice
__global__ __launch_bounds__(BN_THREADS) void bn_stats_kernel(int c, int l, const float *__restrict__ x,
                                                              float *__restrict__ partial)
{
    const int bi = blockIdx.z, cc = blockIdx.y;
    const float *row = x + ((size_t)bi * c + cc) * l;
    const int per = (((l + gridDim.x - 1) / gridDim.x) + 3) & ~3;
    const int e0 = blockIdx.x * per, e1 = min(l, e0 + per);
    float s = 0.f, ss = 0.f;
    const float p = e0 < e1 ? row[e0] : 0.f;
    if (e0 >= e1) {
        // empty slice (short rows): contributes zeros
    } else if ((((uintptr_t)row) & 15) == 0) {
        const int v0 = e0 >> 2, v1 = e1 >> 2;
        for (int v = v0 + threadIdx.x; v < v1; v += BN_THREADS) {
            float4 q = reinterpret_cast<const float4 *>(row)[v];
            q.x -= p; q.y -= p; q.z -= p; q.w -= p;
            s += (q.x + q.y) + (q.z + q.w);
            ss = fmaf(q.x, q.x, fmaf(q.y, q.y, fmaf(q.z, q.z, fmaf(q.w, q.w, ss))));
        }
        for (int e = (v1 << 2) + threadIdx.x; e < e1; e += BN_THREADS) { const float d = row[e] - p; s += d; ss = fmaf(d, d, ss); }
    } else {
        for (int e = e0 + threadIdx.x; e < e1; e += BN_THREADS) { const float d = row[e] - p; s += d; ss = fmaf(d, d, ss); }
    }
    float *dst = partial + (((size_t)bi * c + cc) * gridDim.x + blockIdx.x) * 4;
    bn_block_store2(s, ss, dst);
    if (threadIdx.x == 0) { dst[2] = p; dst[3] = (float)max(e1 - e0, 0); }
}

// out = x * scale[c] + shift[c], clamped at 0 when relu; grid (gx, C, B)
__global__ __launch_bounds__(BN_THREADS) void bn_apply_kernel(int c, int l, int relu, const float *__restrict__ x,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ shift, float *__restrict__ out)
{
    const int bi = blockIdx.z, cc = blockIdx.y;
    const size_t base = ((size_t)bi * c + cc) * l;
    const float a = scale[cc], b = shift[cc];
    const float lo = relu ? 0.f : -INFINITY;
    if ((((uintptr_t)(x + base) | (uintptr_t)(out + base)) & 15) == 0) {
        const int vec = l >> 2;
        for (int v = blockIdx.x * BN_THREADS + threadIdx.x; v < vec; v += gridDim.x * BN_THREADS) {
            float4 q = reinterpret_cast<const float4 *>(x + base)[v];
            q.x = fmaxf(fmaf(q.x, a, b), lo);
            q.y = fmaxf(fmaf(q.y, a, b), lo);
            q.z = fmaxf(fmaf(q.z, a, b), lo);
            q.w = fmaxf(fmaf(q.w, a, b), lo);
            reinterpret_cast<float4 *>(out + base)[v] = q;
        }
        for (int e = (vec << 2) + blockIdx.x * BN_THREADS + threadIdx.x; e < l; e += gridDim.x * BN_THREADS)
            out[base + e] = fmaxf(fmaf(x[base + e], a, b), lo);
    } else {
        for (int e = blockIdx.x * BN_THREADS + threadIdx.x; e < l; e += gridDim.x * BN_THREADS)
            out[base + e] = fmaxf(fmaf(x[base + e], a, b), lo);
    }
}

// backward reduce: g = dz * [x*scale + shift > 0 or !relu];  partial = (sum g, sum g * xhat),  xhat = (x - mean) rstd
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_reduce_kernel(int c, int l, int relu, const float *__restrict__ x,
                                                                   const float *__restrict__ dz,
                                                                   const float *__restrict__ scale,
                                                                   const float *__restrict__ shift,
                                                                   const float *__restrict__ mean,
                                                                   const float *__restrict__ rstd,
                                                                   float *__restrict__ partial)
{
    const int bi = blockIdx.z, cc = blockIdx.y;
    const size_t base = ((size_t)bi * c + cc) * l;
    const float a = scale[cc], b = shift[cc], mu = mean[cc], r = rstd[cc];
    const int per = (((l + gridDim.x - 1) / gridDim.x) + 3) & ~3;
    const int e0 = blockIdx.x * per, e1 = min(l, e0 + per);
    float s = 0.f, sx = 0.f;
    auto one = [&](float xv, float gv) {
        const float g = (!relu || fmaf(xv, a, b) > 0.f) ? gv : 0.f;
        s += g;
        sx = fmaf(g, (xv - mu) * r, sx);
    };
    if (e0 >= e1) {
        // empty slice
    } else if ((((uintptr_t)(x + base) | (uintptr_t)(dz + base)) & 15) == 0) {
        const int v0 = e0 >> 2, v1 = e1 >> 2;
        for (int v = v0 + threadIdx.x; v < v1; v += BN_THREADS) {
            const float4 q = reinterpret_cast<const float4 *>(x + base)[v];
            const float4 g = reinterpret_cast<const float4 *>(dz + base)[v];
            one(q.x, g.x); one(q.y, g.y); one(q.z, g.z); one(q.w, g.w);
        }
        for (int e = (v1 << 2) + threadIdx.x; e < e1; e += BN_THREADS) one(x[base + e], dz[base + e]);
    } else {
        for (int e = e0 + threadIdx.x; e < e1; e += BN_THREADS) one(x[base + e], dz[base + e]);
    }
    bn_block_store2(s, sx, partial + (((size_t)bi * c + cc) * gridDim.x + blockIdx.x) * 2);
}

// backward apply: dx = k0[c] * (g - c1[c] - xhat * c2[c])   (training: k0 = gamma rstd, c1 = mean(g), c2 = mean(g xhat);
//                                                           eval: k0 = gamma rstd_running, c1 = c2 = 0)
__global__ __launch_bounds__(BN_THREADS) void bn_bwd_apply_kernel(int c, int l, int relu, const float *__restrict__ x,
                                                                  const float *__restrict__ dz,
                                                                  const float *__restrict__ scale,
                                                                  const float *__restrict__ shift,
                                                                  const float *__restrict__ mean,
                                                                  const float *__restrict__ rstd,
                                                                  const float *__restrict__ k0,
                                                                  const float *__restrict__ c1,
                                                                  const float *__restrict__ c2, float *__restrict__ dx)
{
    const int bi = blockIdx.z, cc = blockIdx.y;
    const size_t base = ((size_t)bi * c + cc) * l;
    const float a = scale[cc], b = shift[cc], mu = mean[cc], r = rstd[cc], kk = k0[cc], m1 = c1[cc], m2 = c2[cc];
    auto one = [&](float xv, float gv) {
        const float g = (!relu || fmaf(xv, a, b) > 0.f) ? gv : 0.f;
        return kk * (g - m1 - (xv - mu) * r * m2);
    };
    if ((((uintptr_t)(x + base) | (uintptr_t)(dz + base) | (uintptr_t)(dx + base)) & 15) == 0) {
        const int vec = l >> 2;
        for (int v = blockIdx.x * BN_THREADS + threadIdx.x; v < vec; v += gridDim.x * BN_THREADS) {
            const float4 q = reinterpret_cast<const float4 *>(x + base)[v];
            const float4 g = reinterpret_cast<const float4 *>(dz + base)[v];
            reinterpret_cast<float4 *>(dx + base)[v] = make_float4(one(q.x, g.x), one(q.y, g.y), one(q.z, g.z), one(q.w, g.w));
        }
        for (int e = (vec << 2) + blockIdx.x * BN_THREADS + threadIdx.x; e < l; e += gridDim.x * BN_THREADS)
            dx[base + e] = one(x[base + e], dz[base + e]);
    } else {
        for (int e = blockIdx.x * BN_THREADS + threadIdx.x; e < l; e += gridDim.x * BN_THREADS)
            dx[base + e] = one(x[base + e], dz[base + e]);
    }
}

// ---- FP front end: y[b,c,e] = sum_t w[e,t] A[b,c,idx[e,t]] + sum_k Wb[c,k] skip[b,k,e], + its BatchNorm sums -----
// (the first SharedMLP stage of a PointnetFPModule with the 1x1 convolution moved in front of the interpolation:
// A = W_a known_feats; pointnet2_modules.py:619-640).  CH rows of A in LDS, queries streamed, 2 elements per thread in
// flight; per-(b, c, slice) partial sums of y and y^2 leave with the result, so the BatchNorm needs no pass of its own.
constexpr int FP_THREADS = 1024;
constexpr int FP_LDS_BYTES = 144 * 1024;
constexpr int FP_MAX_SKIP = 8;

template <int CH>
__global__ __launch_bounds__(FP_THREADS) void fp_front_kernel(int c, int m, int n, int cs, const float *__restrict__ A,
                                                              const int *__restrict__ idx, const float *__restrict__ w,
                                                              const float *__restrict__ skip, const float *__restrict__ Wb,
                                                              float *__restrict__ y, float *__restrict__ partial)
{
    extern __shared__ float fp_rows[]; // [CH][m]
    __shared__ float wb[CH][FP_MAX_SKIP];
    __shared__ float red[FP_THREADS / 64][CH][2];
    const int bi = blockIdx.z, c0 = blockIdx.y * CH, nch = min(CH, c - c0);
#ifndef GEOT_FP_LAB_NOTABLE
    {
        const float *src = A + ((size_t)bi * c + c0) * m;
        const int count = nch * m;
        if ((((uintptr_t)src) & 15) == 0) {
            const int vec = count >> 2;
            for (int v = threadIdx.x; v < vec; v += FP_THREADS) reinterpret_cast<float4 *>(fp_rows)[v] = reinterpret_cast<const float4 *>(src)[v];
            for (int e = (vec << 2) + threadIdx.x; e < count; e += FP_THREADS) fp_rows[e] = src[e];
        } else {
            for (int e = threadIdx.x; e < count; e += FP_THREADS) fp_rows[e] = src[e];
        }
    }
#endif
    if (threadIdx.x < CH * FP_MAX_SKIP) {
        const int l = threadIdx.x / FP_MAX_SKIP, k = threadIdx.x % FP_MAX_SKIP;
        wb[l][k] = (l < nch && k < cs) ? Wb[(size_t)(c0 + l) * cs + k] : 0.f;
    }
    __syncthreads();
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int e0 = blockIdx.x * per, e1 = min(n, e0 + per);
    float s[CH], ss[CH], piv[CH];
#pragma unroll
    for (int l = 0; l < CH; ++l) s[l] = ss[l] = piv[l] = 0.f;
    if (e0 < e1) {   // pivot of the shifted sums (see bn_stats_kernel): the slice's first value, formed by every thread
        int i3[3];
        float w3[3], sk0[FP_MAX_SKIP];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            i3[t] = idx[((size_t)bi * n + e0) * 3 + t];
            w3[t] = w[((size_t)bi * n + e0) * 3 + t];
        }
#pragma unroll
        for (int k = 0; k < FP_MAX_SKIP; ++k) sk0[k] = k < cs ? skip[((size_t)bi * cs + k) * n + e0] : 0.f;
#pragma unroll
        for (int l = 0; l < CH; ++l) {
            if (l < nch) {
                const float *R = fp_rows + (size_t)l * m;
                float v = R[i3[0]] * w3[0];
                v = v + R[i3[1]] * w3[1];
                v = v + R[i3[2]] * w3[2];
#pragma unroll
                for (int k = 0; k < FP_MAX_SKIP; ++k) v = fmaf(wb[l][k], sk0[k], v);
                piv[l] = v;
            }
        }
    }
#ifndef GEOT_FP_LAB_U
#define GEOT_FP_LAB_U 2
#endif
    constexpr int U = GEOT_FP_LAB_U;
    for (int eb = e0 + threadIdx.x; eb < e1; eb += U * FP_THREADS) {
        int ii[U][3];
        float ww[U][3], sk[U][FP_MAX_SKIP];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = eb + u * FP_THREADS;
            const bool ok = e < e1;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ii[u][t] = ok ? idx[((size_t)bi * n + e) * 3 + t] : 0;
                ww[u][t] = ok ? w[((size_t)bi * n + e) * 3 + t] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < FP_MAX_SKIP; ++k) sk[u][k] = (ok && k < cs) ? skip[((size_t)bi * cs + k) * n + e] : 0.f;
        }
#pragma unroll
        for (int l = 0; l < CH; ++l) {
            if (l < nch) {
                const float *R = fp_rows + (size_t)l * m;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = eb + u * FP_THREADS;
#ifdef GEOT_FP_LAB_NOLDS
                    float v = __int_as_float(ii[u][0]) * ww[u][0];
                    v = v + __int_as_float(ii[u][1]) * ww[u][1];
                    v = v + __int_as_float(ii[u][2]) * ww[u][2];
#else
                    float v = R[ii[u][0]] * ww[u][0];
                    v = v + R[ii[u][1]] * ww[u][1];
                    v = v + R[ii[u][2]] * ww[u][2];           // ((p0 w0 + p1 w1) + p2 w2): three_interpolate's order
#endif
#pragma unroll
                    for (int k = 0; k < FP_MAX_SKIP; ++k) v = fmaf(wb[l][k], sk[u][k], v);
                    if (e < e1) {
#ifndef GEOT_FP_LAB_NOSTORE
                        y[((size_t)bi * c + c0 + l) * n + e] = v;
#endif
                        const float d = v - piv[l];
                        s[l] += d;
                        ss[l] = fmaf(d, d, ss[l]);
                    }
                }
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int l = 0; l < CH; ++l) {
        const float a = bn_wave_sum(s[l]), b = bn_wave_sum(ss[l]);
        if (lane == 0) { red[wave][l][0] = a; red[wave][l][1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CH) {
        const int l = threadIdx.x >> 1, q = threadIdx.x & 1;
        if (l < nch) {
            float t = 0.f;
            for (int v = 0; v < FP_THREADS / 64; ++v) t += red[v][l][q];
            float *dst = partial + (((size_t)bi * c + c0 + l) * gridDim.x + blockIdx.x) * 4;
            dst[q] = t;
            dst[2 + q] = q == 0 ? piv[l] : (float)max(e1 - e0, 0);       // (s1, s2, pivot, count)
        }
    }
}

// ---- max over the n innermost (contiguous) elements of every row: (rows, n) -> (rows,) + the arg-max slot ----------
// (the "max over the group's points" of the mini-PointNet Encoder, transformer.py:127-134, and the "max over nsample" of
// the SetAbstraction modules, pointnet2_modules.py:62-66.)  torch's generic reduction reaches 0.5 TB/s on 32-element
// rows; this is a plain stream: LPR lanes share a row with one float4 each per step, first maximum wins (torch.max).
// (value, slot) a beats (value, slot) b: NaN beats every number (torch.max propagates NaN), the earlier slot wins ties
__device__ __forceinline__ bool seg_better(float a, int ja, float b, int jb)
{
    const bool an = a != a, bn = b != b;
    if (an || bn) return an && (!bn || ja < jb);
    return a > b || (a == b && ja < jb);
}
template <int LPR>
__global__ __launch_bounds__(256) void segment_max_kernel(long long rows, int n, const float *__restrict__ x,
                                                          float *__restrict__ out, uint8_t *__restrict__ arg)
{
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (row >= rows) return;
    const float4 *src = reinterpret_cast<const float4 *>(x + row * n);
    float best = -INFINITY;
    int bj = 0x7fffffff;
    for (int v = sub; v < (n >> 2); v += LPR) {
        const float4 q = src[v];
        const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * v + u;
            const bool take = seg_better(e[u], j, best, bj);
            best = take ? e[u] : best;
            bj = take ? j : bj;
        }
    }
#pragma unroll
    for (int d = 1; d < LPR; d <<= 1) {                       // the LPR lanes of a row are adjacent lanes of one wave
        const float ob = __shfl_xor(best, d);
        const int oj = __shfl_xor(bj, d);
        const bool take = seg_better(ob, oj, best, bj);
        best = take ? ob : best;
        bj = take ? oj : bj;
    }
    if (sub == 0) {
        out[row] = best;
        arg[row] = (uint8_t)(bj == 0x7fffffff ? 0 : bj);
    }
}
// out[row] = sum of the row's n elements: fixed order (lane sub-sums, then a butterfly), LPR adjacent lanes per row
template <int LPR>
__global__ __launch_bounds__(256) void segment_sum_kernel(long long rows, int n, const float *__restrict__ x,
                                                          float *__restrict__ out)
{
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (row >= rows) return;
    const float4 *src = reinterpret_cast<const float4 *>(x + row * n);
    float s = 0.f;
    for (int v = sub; v < (n >> 2); v += LPR) {
        const float4 q = src[v];
        s += (q.x + q.y) + (q.z + q.w);
    }
#pragma unroll
    for (int d = 1; d < LPR; d <<= 1) s += __shfl_xor(s, d);
    if (sub == 0) out[row] = s;
}
// partial[(row * S + slice) * J + j] = sum over the slice of a[row][l] * b[j][l], J <= 8 rows of b: the weight gradient of
// a 1x1 convolution with a handful of input channels (the Encoder's Conv1d(3, 128) over 131 072 columns is a
// 128 x 3 x 131072 GEMM to the libraries: 436 us; this streams the 67 MB once)
template <int J>
__global__ __launch_bounds__(256) void rowdot_small_kernel(int l, const float *__restrict__ a, const float *__restrict__ b,
                                                           float *__restrict__ partial)
{
    const int row = blockIdx.y, per = (((l + gridDim.x - 1) / gridDim.x) + 3) & ~3;
    const int e0 = blockIdx.x * per, e1 = min(l, e0 + per);
    const float *ar = a + (size_t)row * l;
    float acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = 0.f;
    for (int e = e0 + threadIdx.x; e < e1; e += 256) {
        const float v = ar[e];
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] = fmaf(v, b[(size_t)j * l + e], acc[j]);
    }
    __shared__ float sh[J][4];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        float v = acc[j];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) sh[j][threadIdx.x >> 6] = v;
    }
    __syncthreads();
    if (threadIdx.x < J)
        partial[((size_t)row * gridDim.x + blockIdx.x) * J + threadIdx.x] =
            (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}
// Column sums of a row-major (rows, cols) matrix -- the bias gradient of a Linear layer (grad_output.sum(0); torch's
// generic reduction takes 19 us + a 5 us fill for (4096, 384)).  grid (col blocks of 64, S row slices): a wave reads 64
// consecutive columns of a row (256 B), the block's 4 waves take every 4th row of the slice; partial (S, cols).
__global__ __launch_bounds__(256) void colsum_kernel(int rows, int cols, const float *__restrict__ x, float *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s0 = 0.f;
    if (col < cols) {
        int r = r0 + wave;
        for (; r + 28 < r1; r += 32) {                     // eight rows in flight per wave (the pass is latency-bound)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(r + 4 * u) * cols + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) s0 += v[u];
        }
        for (; r < r1; r += 4) s0 += x[(size_t)r * cols + col];
    }
    __shared__ float sh[4][64];
    sh[wave][lane] = s0;
    __syncthreads();
    if (wave == 0 && col < cols) partial[(size_t)blockIdx.y * cols + col] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
}
__global__ __launch_bounds__(256) void colsum_finish_kernel(int s, int cols, const float *__restrict__ partial, float *__restrict__ out)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = i < s ? partial[(size_t)i * cols + col] : 0.f;   // independent loads, then a fixed order
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) a += v[i];
    out[col] = a;
}
// ---- BatchNorm -> ReLU -> max over the n innermost elements, without materialising the normalised tensor ---------------
// max_j relu(a y_j + b) = relu(a sel + b) with sel = max_j y_j if a >= 0 else min_j y_j (the affine map is monotone): the
// last stage of a SetAbstraction module in training mode (pointnet2_modules.py:57-66: SharedMLP then max_pool2d over
// nsample).  Forward: rows = (b, c, group) of n elements; out = relu(a_c sel + b_c), sel, arg = slot of the first
// extremum.  Backward: the gradient of the (never built) normalised tensor is g at arg (if the pre-activation was
// positive) and 0 elsewhere, so  dx_j = k0_c ([j == arg] gm - c1_c - xhat_j c2_c)  in one pass over y.
template <int LPR>
__global__ __launch_bounds__(256) void bn_pool_kernel(long long rows, int n, int groups, int c, int relu,
                                                      const float *__restrict__ y, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, float *__restrict__ out,
                                                      float *__restrict__ sel_out, uint8_t *__restrict__ arg)
{
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (row >= rows) return;
    const int cc = (int)((row / groups) % c);
    const float a = scale[cc];
    const float sgn = a >= 0.f ? 1.f : -1.f;               // max of sgn * y: the maximum, or the minimum
    const float4 *src = reinterpret_cast<const float4 *>(y + row * n);
    float best = -INFINITY;
    int bj = 0x7fffffff;
    for (int v = sub; v < (n >> 2); v += LPR) {
        const float4 q = src[v];
        const float e[4] = {sgn * q.x, sgn * q.y, sgn * q.z, sgn * q.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = 4 * v + u;
            const bool take = seg_better(e[u], j, best, bj);
            best = take ? e[u] : best;
            bj = take ? j : bj;
        }
    }
#pragma unroll
    for (int d = 1; d < LPR; d <<= 1) {
        const float ob = __shfl_xor(best, d);
        const int oj = __shfl_xor(bj, d);
        const bool take = seg_better(ob, oj, best, bj);
        best = take ? ob : best;
        bj = take ? oj : bj;
    }
    if (sub == 0) {
        float sv = sgn * best;
        if (a == 0.f) {                                    // every slot ties at relu(b): the reference's max takes the first
            bj = 0;
            sv = y[row * n];
        }
        const float z = fmaf(sv, a, shift[cc]);
        out[row] = relu ? fmaxf(z, 0.f) : z;               // (NaN: fmaxf drops it; the reference's relu keeps it -- inputs are finite)
        sel_out[row] = sv;
        arg[row] = (uint8_t)(bj == 0x7fffffff ? 0 : bj);
    }
}
__global__ __launch_bounds__(256) void bn_pool_bwd_kernel(long long total4, int n4, int groups, int c,
                                                          const float *__restrict__ y, const float *__restrict__ gm,
                                                          const uint8_t *__restrict__ arg, const float *__restrict__ mean,
                                                          const float *__restrict__ rstd, const float *__restrict__ k0,
                                                          const float *__restrict__ c1, const float *__restrict__ c2,
                                                          float *__restrict__ dx)
{
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total4; v += (long long)gridDim.x * 256) {
        const long long row = v / n4;
        const int cc = (int)((row / groups) % c);
        const int j0 = (int)(v - row * n4) * 4, a = arg[row];
        const float g = gm[row], mu = mean[cc], r = rstd[cc], kk = k0[cc], m1 = c1[cc], m2 = c2[cc];
        const float4 q = reinterpret_cast<const float4 *>(y)[v];
        reinterpret_cast<float4 *>(dx)[v] =
            make_float4(kk * ((a == j0 ? g : 0.f) - m1 - (q.x - mu) * r * m2), kk * ((a == j0 + 1 ? g : 0.f) - m1 - (q.y - mu) * r * m2),
                        kk * ((a == j0 + 2 ? g : 0.f) - m1 - (q.z - mu) * r * m2), kk * ((a == j0 + 3 ? g : 0.f) - m1 - (q.w - mu) * r * m2));
    }
}
// dx (rows, n) = dy[row] at the arg-max slot, 0 elsewhere (written in full)
__global__ __launch_bounds__(256) void segment_max_grad_kernel(long long total4, int n4, const float *__restrict__ dy,
                                                               const uint8_t *__restrict__ arg, float *__restrict__ dx)
{
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < total4; v += (long long)gridDim.x * 256) {
        const long long row = v / n4;
        const int j0 = (int)(v - row * n4) * 4, a = arg[row];
        const float g = dy[row];
        reinterpret_cast<float4 *>(dx)[v] = make_float4(a == j0 ? g : 0.f, a == j0 + 1 ? g : 0.f, a == j0 + 2 ? g : 0.f,
                                                        a == j0 + 3 ? g : 0.f);
    }
}

static int bn_slices_for(int b, int c, int l)
{
    long long rows = (long long)b * c;
    long long sl = (1024 + rows - 1) / rows;       // >= 1024 workgroups when the rows alone do not give them
    const long long maxs = l / 4096;
    if (sl > maxs) sl = maxs;
    if (sl < 1) sl = 1;
    if (sl > BN_MAX_SLICES) sl = BN_MAX_SLICES;
    return (int)sl;
}
static int fp_ch_for(int m)
{
    const int fit = FP_LDS_BYTES / ((int)sizeof(float) * m);
    return fit >= 8 ? 8 : (fit >= 4 ? 4 : (fit >= 2 ? 2 : (fit >= 1 ? 1 : 0)));
}
static int fp_slices_for(int b, int c, int ch, int n)
{
    const long long chunks = ((long long)c + ch - 1) / ch * b;
    long long sl = (512 + chunks - 1) / chunks;
    const long long maxs = n / 4096;
    if (sl > maxs) sl = maxs;
    if (sl < 1) sl = 1;
    if (sl > BN_MAX_SLICES) sl = BN_MAX_SLICES;
    return (int)sl;
}
static bool bn_dims_ok(int b, int c, int l) { return b >= 1 && c >= 1 && l >= 1 && b <= 65535 && c <= 65535; }
static int bn_gx(int l)
{
    int gx = (l / 4 + BN_THREADS * 4 - 1) / (BN_THREADS * 4);
    return gx < 1 ? 1 : (gx > 64 ? 64 : gx);
}

// ---- the O(c) arithmetic between the passes --------------------------------------------------------------------
// One wave per channel sums the (b x S) partial pairs in fp64, lanes striding over them, then a fixed butterfly:
// the same order on every run.
__global__ __launch_bounds__(64) void bn_sums_kernel(int b, int c, int s, const float *__restrict__ partial,
                                                     double *__restrict__ sums)
{
    const int ch = blockIdx.x, lane = threadIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (int t = lane; t < b * s; t += 64) {
        const int bi = t / s, sl = t - bi * s;
        const float2 v = reinterpret_cast<const float2 *>(partial)[((size_t)bi * c + ch) * s + sl];
        a0 += (double)v.x;
        a1 += (double)v.y;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
    }
    if (lane == 0) {
        sums[2 * ch] = a0;
        sums[2 * ch + 1] = a1;
    }
}

// the same for the shifted statistics records (s1, s2, pivot, count) of bn_stats_kernel / fp_front_kernel: every record is
// turned back into (sum x, sum x^2) in fp64 before it is added
__global__ __launch_bounds__(64) void bn_sums_shifted_kernel(int b, int c, int s, const float *__restrict__ partial,
                                                             double *__restrict__ sums)
{
    const int ch = blockIdx.x, lane = threadIdx.x;
    double a0 = 0.0, a1 = 0.0;
    for (int t = lane; t < b * s; t += 64) {
        const int bi = t / s, sl = t - bi * s;
        const float4 v = reinterpret_cast<const float4 *>(partial)[((size_t)bi * c + ch) * s + sl];
        const double s1 = v.x, s2 = v.y, p = v.z, n = v.w;
        a0 += s1 + n * p;
        a1 += s2 + 2.0 * p * s1 + n * p * p;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a0 += __shfl_xor(a0, o);
        a1 += __shfl_xor(a1, o);
    }
    if (lane == 0) {
        sums[2 * ch] = a0;
        sums[2 * ch + 1] = a1;
    }
}

// mean / biased variance from (sum x, sum x^2) and the element count, everything BatchNorm derives from them:
// rstd, the affine pair of the apply pass, the running statistics (unbiased variance, factor eaf).
__global__ __launch_bounds__(256) void bn_finalize_kernel(int c, const double *__restrict__ sums, double count,
                                                          const double *__restrict__ count_dev, double eps, double eaf,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const float *__restrict__ pre_bias,
                                                          float *__restrict__ running_mean, float *__restrict__ running_var,
                                                          float *__restrict__ mean, float *__restrict__ rstd,
                                                          float *__restrict__ scale, float *__restrict__ shift)
{
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    const double n = count_dev ? *count_dev : count;
    const double m64 = sums[2 * ch] / n;
    double v64 = sums[2 * ch + 1] / n - m64 * m64;
    v64 = v64 > 0.0 ? v64 : 0.0;
    const float m = (float)m64, r = (float)(1.0 / sqrt(v64 + eps));
    const float sc = (gamma ? gamma[ch] : 1.f) * r;
    mean[ch] = m;
    rstd[ch] = r;
    scale[ch] = sc;
    shift[ch] = (beta ? beta[ch] : 0.f) - m * sc;
    // pre_bias: the statistics are of y, the layer normalises y + pre_bias (same output; the running mean sees the bias)
    if (running_mean)
        running_mean[ch] = running_mean[ch] * (float)(1.0 - eaf) + (float)eaf * (pre_bias ? (float)(m64 + (double)pre_bias[ch]) : m);
    if (running_var) {
        const double unbiased = v64 * (n / (n - 1.0 > 1.0 ? n - 1.0 : 1.0));
        running_var[ch] = running_var[ch] * (float)(1.0 - eaf) + (float)eaf * (float)unbiased;
    }
}

// backward: the parameter gradients from this rank's sums, the two means of the input gradient from the (all-reduced)
// sums; count == 0 (running statistics were used: constants) gives c1 = c2 = 0
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(int c, const double *__restrict__ local, const double *__restrict__ sums,
                                                          double count, const double *__restrict__ count_dev,
                                                          float *__restrict__ g_gamma, float *__restrict__ g_beta,
                                                          float *__restrict__ c1, float *__restrict__ c2)
{
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    const double n = count_dev ? *count_dev : count;
    g_beta[ch] = (float)local[2 * ch];
    g_gamma[ch] = (float)local[2 * ch + 1];
    c1[ch] = n > 0.0 ? (float)(sums[2 * ch] / n) : 0.f;
    c2[ch] = n > 0.0 ? (float)(sums[2 * ch + 1] / n) : 0.f;
}

} // namespace geot

using namespace geot;

GEOT_EXPORT int geot_bn_slices(int b, int c, int l) { return bn_dims_ok(b, c, l) ? bn_slices_for(b, c, l) : -1; }

GEOT_EXPORT int geot_bn_stats(int b, int c, int l, const float *x, float *partial, void *stream)
{
    if (!bn_dims_ok(b, c, l)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(bn_slices_for(b, c, l), c, b), dim3(BN_THREADS), 0, (hipStream_t)stream, c, l, x, partial);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_apply(int b, int c, int l, int relu, const float *x, const float *scale, const float *shift,
                              float *out, void *stream)
{
    if (!bn_dims_ok(b, c, l)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(bn_gx(l), c, b), dim3(BN_THREADS), 0, (hipStream_t)stream, c, l, relu, x, scale, shift, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_bwd_reduce(int b, int c, int l, int relu, const float *x, const float *dz, const float *scale,
                                   const float *shift, const float *mean, const float *rstd, float *partial, void *stream)
{
    if (!bn_dims_ok(b, c, l)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(bn_slices_for(b, c, l), c, b), dim3(BN_THREADS), 0, (hipStream_t)stream, c, l,
                       relu, x, dz, scale, shift, mean, rstd, partial);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_bwd_apply(int b, int c, int l, int relu, const float *x, const float *dz, const float *scale,
                                  const float *shift, const float *mean, const float *rstd, const float *k0,
                                  const float *c1, const float *c2, float *dx, void *stream)
{
    if (!bn_dims_ok(b, c, l)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(bn_gx(l), c, b), dim3(BN_THREADS), 0, (hipStream_t)stream, c, l, relu, x, dz,
                       scale, shift, mean, rstd, k0, c1, c2, dx);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_sums_shifted(int b, int c, int s, const float *partial, double *sums, void *stream)
{
    if (b < 0 || c < 0 || s < 0) return hipErrorInvalidValue;
    if (c == 0) return hipSuccess;
    hipLaunchKernelGGL(bn_sums_shifted_kernel, dim3(c), dim3(64), 0, (hipStream_t)stream, b, c, s, partial, sums);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_sums(int b, int c, int s, const float *partial, double *sums, void *stream)
{
    if (b < 1 || c < 1 || s < 1 || !partial || !sums) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_sums_kernel, dim3(c), dim3(64), 0, (hipStream_t)stream, b, c, s, partial, sums);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_finalize(int c, const double *sums, double count, const double *count_dev, double eps, double eaf,
                                 const float *gamma, const float *beta, const float *pre_bias, float *running_mean,
                                 float *running_var, float *mean, float *rstd, float *scale, float *shift, void *stream)
{
    if (c < 1 || !sums || !mean || !rstd || !scale || !shift || (!count_dev && !(count > 0.0))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, c, sums, count, count_dev,
                       eps, eaf, gamma, beta, pre_bias, running_mean, running_var, mean, rstd, scale, shift);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_bwd_coef(int c, const double *local_sums, const double *sums, double count, const double *count_dev,
                                 float *g_gamma, float *g_beta, float *c1, float *c2, void *stream)
{
    if (c < 1 || !local_sums || !sums || !g_gamma || !g_beta || !c1 || !c2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, c, local_sums, sums, count,
                       count_dev, g_gamma, g_beta, c1, c2);
    return hipGetLastError();
}

GEOT_EXPORT int geot_fp_front_slices(int b, int c, int m, int n)
{
    const int ch = fp_ch_for(m);
    return (bn_dims_ok(b, c, n) && m >= 1 && ch >= 1) ? fp_slices_for(b, c, ch, n) : -1;
}

GEOT_EXPORT int geot_fp_front(int b, int c, int m, int n, int cs, const float *A, const int *idx, const float *weight,
                              const float *skip, const float *Wb, float *y, float *partial, void *stream)
{
    const int ch = fp_ch_for(m);
    if (!bn_dims_ok(b, c, n) || m < 1 || ch < 1 || cs < 0 || cs > FP_MAX_SKIP || (cs > 0 && (!skip || !Wb)))
        return hipErrorInvalidValue;
    const int slices = fp_slices_for(b, c, ch, n);
    const size_t lds = (size_t)ch * m * sizeof(float);
    const dim3 grid(slices, (c + ch - 1) / ch, b);
    hipError_t e = hipSuccess;
#define GEOT_FP_LAUNCH(CHV)                                                                                       \
    {                                                                                                             \
        e = allow_big_lds((const void *)fp_front_kernel<CHV>, lds);                                               \
        if (e != hipSuccess) return e;                                                                            \
        hipLaunchKernelGGL((fp_front_kernel<CHV>), grid, dim3(FP_THREADS), lds, (hipStream_t)stream, c, m, n, cs, A, \
                           idx, weight, skip, Wb, y, partial);                                                    \
    }
    if (ch == 8) GEOT_FP_LAUNCH(8) else if (ch == 4) GEOT_FP_LAUNCH(4) else if (ch == 2) GEOT_FP_LAUNCH(2) else GEOT_FP_LAUNCH(1)
#undef GEOT_FP_LAUNCH
    return hipGetLastError();
}

GEOT_EXPORT int geot_segment_max(long long rows, int n, const float *x, float *out, unsigned char *arg, void *stream)
{
    if (rows < 0 || n < 4 || n > 256 || (n & 3) || (((uintptr_t)x) & 15)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const int lpr = n >= 32 ? 8 : (n >= 16 ? 4 : (n >= 8 ? 2 : 1));
    const long long blocks = (rows * lpr + 255) / 256;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (lpr == 8) hipLaunchKernelGGL(segment_max_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, x, out, arg);
    else if (lpr == 4) hipLaunchKernelGGL(segment_max_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, x, out, arg);
    else if (lpr == 2) hipLaunchKernelGGL(segment_max_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, x, out, arg);
    else hipLaunchKernelGGL(segment_max_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, x, out, arg);
    return hipGetLastError();
}

GEOT_EXPORT int geot_rowdot_small_slices(int rows, int l)
{
    if (rows < 1 || l < 1) return -1;
    long long s = (1024 + rows - 1) / rows;                  // >= 1024 workgroups, slices of >= 2048 elements
    const long long most = (l + 2047) / 2048;
    s = s > most ? most : s;
    return (int)(s < 1 ? 1 : (s > 64 ? 64 : s));
}

GEOT_EXPORT int geot_rowdot_small(int rows, int l, int j, const float *a, const float *b, float *partial, void *stream)
{
    if (rows < 1 || rows > 65535 || l < 1 || j < 1 || j > 8 || !a || !b || !partial) return hipErrorInvalidValue;
    const dim3 grid(geot_rowdot_small_slices(rows, l), rows);
#define GEOT_RD(JV) hipLaunchKernelGGL(rowdot_small_kernel<JV>, grid, dim3(256), 0, (hipStream_t)stream, l, a, b, partial)
    switch (j) {
    case 1: GEOT_RD(1); break;
    case 2: GEOT_RD(2); break;
    case 3: GEOT_RD(3); break;
    case 4: GEOT_RD(4); break;
    case 5: GEOT_RD(5); break;
    case 6: GEOT_RD(6); break;
    case 7: GEOT_RD(7); break;
    default: GEOT_RD(8); break;
    }
#undef GEOT_RD
    return hipGetLastError();
}

static int colsum_slices(int rows, int cols)
{
    const int cb = (cols + 63) / 64;
    long long s = (256 + cb - 1) / cb;                     // ~256 workgroups, slices of >= 32 rows, at most 16 slices: the
    const long long most = (rows + 31) / 32;               // finish pass reads them one after the other
    s = s > most ? most : s;
    return (int)(s < 1 ? 1 : (s > 16 ? 16 : s));
}
GEOT_EXPORT long long geot_colsum_ws_floats(int rows, int cols)
{
    if (rows < 1 || cols < 1) return -1;
    return (long long)colsum_slices(rows, cols) * cols;
}
GEOT_EXPORT int geot_colsum(int rows, int cols, const float *x, float *out, float *workspace, void *stream)
{
    if (rows < 1 || cols < 1 || !x || !out || !workspace) return hipErrorInvalidValue;
    const int s = colsum_slices(rows, cols);
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 63) / 64, s), dim3(256), 0, (hipStream_t)stream, rows, cols, x, workspace);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, s, cols, workspace, out);
    return hipGetLastError();
}

namespace geot {
// out[r] = sum of the n floats of row r, accumulated in fp64 in a fixed order (a lane's strided share ascending, then a
// fixed tree): one workgroup per row.  For the long, few-row sums whose torch form (aten::sum over 10^5 elements into a
// handful of outputs) zeroes a semaphore buffer with hipMemsetAsync -- a memset NODE once captured, which ROCm 7.0's graph
// packet capture does not keep ordered (geot_amd/graph_step.py): bias gradients of 1x1 convolutions over (B, C, N), the
// per-channel sums of the FP stage's skip input.
__global__ __launch_bounds__(256) void rowsum_f64_kernel(int n, const float *__restrict__ x, double *__restrict__ out)
{
    const float *row = x + (size_t)blockIdx.x * n;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += (double)row[i];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
} // namespace geot

GEOT_EXPORT int geot_rowsum_f64(long long rows, int n, const float *x, double *out, void *stream)
{
    if (rows < 0 || n < 0 || rows > 0x7fffffffLL) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(geot::rowsum_f64_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, n, x, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_segment_sum(long long rows, int n, const float *x, float *out, void *stream)
{
    if (rows < 0 || n < 4 || n > 256 || (n & 3) || (((uintptr_t)x) & 15)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const int lpr = n >= 32 ? 8 : (n >= 16 ? 4 : (n >= 8 ? 2 : 1));
    const long long blocks = (rows * lpr + 255) / 256;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (lpr == 8) hipLaunchKernelGGL(segment_sum_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, x, out);
    else if (lpr == 4) hipLaunchKernelGGL(segment_sum_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, x, out);
    else if (lpr == 2) hipLaunchKernelGGL(segment_sum_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, x, out);
    else hipLaunchKernelGGL(segment_sum_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, x, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_pool(int b, int c, int groups, int n, int relu, const float *y, const float *scale, const float *shift,
                             float *out, float *sel, unsigned char *arg, void *stream)
{
    if (b < 1 || c < 1 || groups < 1 || n < 4 || n > 256 || (n & 3) || !y || !scale || !shift || !out || !sel || !arg ||
        (((uintptr_t)y) & 15))
        return hipErrorInvalidValue;
    const long long rows = (long long)b * c * groups;
    const int lpr = n >= 32 ? 8 : (n >= 16 ? 4 : (n >= 8 ? 2 : 1));
    const long long blocks = (rows * lpr + 255) / 256;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
#define GEOT_BP(L) hipLaunchKernelGGL(bn_pool_kernel<L>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rows, n, groups, c, \
                                      relu, y, scale, shift, out, sel, arg)
    if (lpr == 8) GEOT_BP(8); else if (lpr == 4) GEOT_BP(4); else if (lpr == 2) GEOT_BP(2); else GEOT_BP(1);
#undef GEOT_BP
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_pool_grad(int b, int c, int groups, int n, const float *y, const float *gm, const unsigned char *arg,
                                  const float *mean, const float *rstd, const float *k0, const float *c1, const float *c2,
                                  float *dx, void *stream)
{
    if (b < 1 || c < 1 || groups < 1 || n < 4 || n > 256 || (n & 3) || !y || !gm || !arg || !mean || !rstd || !k0 || !c1 || !c2 ||
        !dx || ((((uintptr_t)y) | ((uintptr_t)dx)) & 15))
        return hipErrorInvalidValue;
    const long long total4 = (long long)b * c * groups * (n >> 2);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(bn_pool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, total4, n >> 2, groups, c, y, gm,
                       arg, mean, rstd, k0, c1, c2, dx);
    return hipGetLastError();
}

GEOT_EXPORT int geot_segment_max_grad(long long rows, int n, const float *dy, const unsigned char *arg, float *dx, void *stream)
{
    if (rows < 0 || n < 4 || n > 256 || (n & 3) || (((uintptr_t)dx) & 15)) return hipErrorInvalidValue;
    if (rows == 0) return hipSuccess;
    const long long total4 = rows * (n >> 2);
    long long blocks = (total4 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(segment_max_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, total4, n >> 2, dy, arg, dx);
    return hipGetLastError();
}
