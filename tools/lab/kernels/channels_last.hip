// channels_last.hip -- the PointnetFPModule front end and its BatchNorm on POINT-MAJOR (B, N, C) activations (gfx950).
//
// Why a second layout.  In the reference's (B, C, N) layout a gather "row e of the result = 3 weighted rows of a table"
// (three_interpolate, pointnet2/_ext_src/src/interpolate_gpu.cu:88-146) touches one 4-byte word per channel: a kernel
// has to own a few whole channel rows (LDS) and walk ALL the elements' neighbour ids, weights and skip values for them
// -- 44 B of side data per element for every 16 B of output (csrc/bnrelu.hip fp_front_kernel: 4.1 GB of it through L2 next
// to 1.59 GB of payload at 8 x 1536 x 24000; profiles/r03_fp_front_lab.txt), and the gradient walks the reverse index
// once per 4 channels (gather_group.hip: 23 % of HBM peak).  Point-major, a point's C channels are ONE contiguous row of
// 4 C bytes (6 KB at C = 1536): a workgroup takes a run of points, its lanes the channels; ids / weights / skip values
// are wave-uniform (scalar loads, read ONCE), every global access is a whole row.  The 1x1 convolutions on either side are
// GEMMs, which take either layout as a transpose flag -- so only this stage changes layout, nothing is ever transposed
// in memory (geot_amd/fused_norm.py: fp_front_cl, bn_act_cl, pointwise_to_cl / pointwise_from_cl).
//
// Kernels (R = B * N rows, C channels, c4 = C / 4; a thread owns one float4 of channels for the whole launch):
//   fp_front_cl        y[b,e,:] = sum_t w[b,e,t] A[b,idx[b,e,t],:] + Wb skip[b,:,e]   + per-tile (sum y, sum y^2)
//   bn_stats_cl        per-tile (sum x, sum x^2)                                        4 B / element
//   bn_apply_cl        out = max(x scale + shift, lo)                                   8 B
//   bn_bwd_reduce_cl   per-tile (sum g, sum g xhat), g = dz [x scale + shift > 0]       8 B
//   bn_bwd_apply_cl    dx = k0 (g - c1 - xhat c2)                                      12 B
//   bn_sums_cl         (tiles, 2, C) float partials -> (C, 2) double sums, fixed order
// Same arithmetic per element as the channels-first kernels of bnrelu.hip (bit-identical values; the statistics differ in
// summation order only).  The gradient of the interpolation lives in gather_group.hip (gather_rows_csr_cl_kernel).
#include "geot_common.h"
#include "geot_hip.h"

namespace geot {

typedef float cl_f4 __attribute__((ext_vector_type(4)));
// Row stores: the rows these kernels write are not read again before the whole tensor has gone by, so they should not
// take L2 capacity from the rows being gathered.  GEOT_CL_LAB_STORE (lab): 0 nt, 1 plain, 2 sc1, 3 sc0 sc1.
#ifndef GEOT_CL_LAB_STORE
#define GEOT_CL_LAB_STORE 0
#endif
__device__ __forceinline__ void cl_store(cl_f4 v, cl_f4 *p)
{
#if GEOT_CL_LAB_STORE == 1
    *p = v;
#elif GEOT_CL_LAB_STORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#elif GEOT_CL_LAB_STORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
#else
    __builtin_nontemporal_store(v, p);
#endif
}
#define CL_LD(p) __builtin_nontemporal_load(p)
#define CL_ST(v, p) cl_store(v, p)
constexpr int CL_MAX_SKIP = 8;
constexpr int CL_MAX_THREADS = 1024;

// Workgroups per launch: every row costs the same, so the launch is ONE round of co-resident workgroups -- CUs x the
// workgroups of cl_block(c4) threads a CU holds at <= 64 registers (8 waves per SIMD) -- each with one contiguous run of
// rows (a second, partly filled round is a ~20 % tail at ~1.6 rounds: measured).  At least 16 rows per workgroup.
static inline int cl_cus() { return device_cus(); }     // per device (geot_common.h)
static inline int cl_tiles_for(long long rows, int c)
{
    const int waves = (c / 4 + 63) / 64;
    int per_cu = 32 / waves;
    if (per_cu < 1) per_cu = 1;
    long long t = (long long)cl_cus() * per_cu;
    if (const char *mult = getenv("GEOT_CL_TILES_MULT")) t *= atoi(mult) > 0 ? atoi(mult) : 1;   // lab
    if (t > (rows + 15) / 16) t = (rows + 15) / 16;
    return (int)(t < 1 ? 1 : t);
}
static inline bool cl_dims_ok(long long rows, int c) { return rows > 0 && c >= 4 && c % 4 == 0 && c / 4 <= CL_MAX_THREADS; }
static inline int cl_block(int c4) { return (c4 + 63) & ~63; }

// One row of the result per step and wave-instruction stream: the side data of a run of CL_STAGE rows (table rows,
// weights, skip values, output row) is staged in LDS by the whole workgroup in one coalesced pass, then every row costs
// 3 uniform LDS reads, 3 row loads (scalar base + lane offset), ~24 packed-fp32 operations and one row store.
constexpr int CL_STAGE = 128;
#ifndef GEOT_CL_LAB_U
#define GEOT_CL_LAB_U 2
#endif
#ifndef GEOT_CL_LAB_STAGES
#define GEOT_CL_LAB_STAGES 2
#endif
constexpr int CL_FP_U = GEOT_CL_LAB_U, CL_FP_STAGES = GEOT_CL_LAB_STAGES;   // rows per group, groups in flight + 1

template <int U>
__device__ __forceinline__ void fp_front_cl_load(int i, int c4, int q, const cl_f4 *__restrict__ a, const int (*s_row)[3],
                                                 cl_f4 (&p)[U][3])
{
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int row = __builtin_amdgcn_readfirstlane(s_row[i + u][t]);
            p[u][t] = (a + (size_t)row * c4)[q];
        }
}
template <int CS, int U>
__device__ __forceinline__ void fp_front_cl_rows(int i, int cnt, int c4, int q, bool on, cl_f4 *__restrict__ y,
                                                 const cl_f4 (&p)[U][3], const float (*s_w)[3],
                                                 const float (*s_sk)[CS > 0 ? CS : 1], const int *s_out, const cl_f4 *wbr,
                                                 cl_f4 &s, cl_f4 &ss, cl_f4 &piv, int &seen)
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (i + u < cnt) {                                // uniform; the loads above are unconditional (padded rows)
            cl_f4 v = p[u][0] * s_w[i + u][0];
            v = v + p[u][1] * s_w[i + u][1];
            v = v + p[u][2] * s_w[i + u][2];              // ((p0 w0 + p1 w1) + p2 w2): three_interpolate's order
#pragma unroll
            for (int k = 0; k < CS; ++k) v = __builtin_elementwise_fma(wbr[k], (cl_f4)(s_sk[i + u][k]), v);
            if (on) {
                const int orow = __builtin_amdgcn_readfirstlane(s_out[i + u]);
                CL_ST(v, y + (size_t)orow * c4 + q);
            }
            if (seen == 0) piv = v;                       // pivot of the shifted sums: this workgroup's first row
            ++seen;
            const cl_f4 d = v - piv;
            s = s + d;
            ss = __builtin_elementwise_fma(d, d, ss);
        }
    }
}

// Which rows a workgroup takes.  Blocks b and b + 8 share an XCD (round-robin dispatch; speed only).  The R rows -- in
// the caller's `order`, a spatial order -- are cut into NX contiguous ranges, one per XCD, and inside a range dealt to the
// XCD's workgroups in granules of G rows: at any moment the XCD's workgroups sit in a narrow band of the sequence, so a
// table row gathered by one of them is still in the XCD's 4-MB L2 when its spatial neighbours ask for it (with one
// contiguous run per workgroup the 160 runs of an XCD touch ~50 MB between two uses of a row).
struct ClDeal {
    int a, b, l, lx, g;     // range [a, b) of this XCD, this workgroup's place l among its lx workgroups, granule
    __device__ __forceinline__ int row(int i) const { return a + (l + (i / g) * lx) * g + i % g; }   // i-th row of this workgroup
};
__device__ __forceinline__ ClDeal cl_deal(int R, int g)
{
    const int nx = gridDim.x >= 8 && gridDim.x % 8 == 0 ? 8 : 1;
    const int x = blockIdx.x % nx;
    ClDeal d;
    d.a = (int)((long long)R * x / nx);
    d.b = (int)((long long)R * (x + 1) / nx);
    d.l = blockIdx.x / nx;
    d.lx = gridDim.x / nx;
    d.g = g;
    return d;
}

template <int CS>
__global__ __launch_bounds__(CL_MAX_THREADS) void fp_front_cl_kernel(
    int c4, int m, int n, int R, int granule, const cl_f4 *__restrict__ a, const int *__restrict__ idx,
    const float *__restrict__ w, const float *__restrict__ skip, const float *__restrict__ wb,
    const int *__restrict__ order, cl_f4 *__restrict__ y, cl_f4 *__restrict__ partial)
{
    constexpr int U = CL_FP_U, S = CL_FP_STAGES;
    __shared__ int s_row[CL_STAGE + (2 * S - 1) * U][3];   // + the rows the software pipeline loads past the end
    __shared__ float s_w[CL_STAGE][3];
    __shared__ float s_sk[CL_STAGE][CS > 0 ? CS : 1];
    __shared__ int s_out[CL_STAGE];
    __shared__ int s_cnt;
    const ClDeal deal = cl_deal(R, granule);
    const bool on = (int)threadIdx.x < c4;
    const int q = on ? threadIdx.x : c4 - 1;              // idle lanes of the last wave shadow the last quad (loads only)
    cl_f4 wbr[CS > 0 ? CS : 1];
#pragma unroll
    for (int k = 0; k < CS; ++k) {
        wbr[k].x = wb[(size_t)(4 * q + 0) * CS + k];
        wbr[k].y = wb[(size_t)(4 * q + 1) * CS + k];
        wbr[k].z = wb[(size_t)(4 * q + 2) * CS + k];
        wbr[k].w = wb[(size_t)(4 * q + 3) * CS + k];
    }
    cl_f4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f}, piv = {0.f, 0.f, 0.f, 0.f};
    int seen = 0;
    for (int base = 0;; base += CL_STAGE) {
        __syncthreads();                                   // the previous run has been consumed
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < CL_STAGE; i += blockDim.x) {
            const int r = deal.row(base + i);
            if (r >= deal.b) continue;                     // rows ascend with i: the valid ones are a prefix
            atomicMax(&s_cnt, i + 1);
            const int bi = r / n, rr = r - bi * n;
            const int e = order ? order[r] : rr;           // order (b, n): per-cloud point ids
            const size_t g = (size_t)bi * n + e;
            s_out[i] = (int)g;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                s_row[i][t] = bi * m + idx[g * 3 + t];
                s_w[i][t] = w[g * 3 + t];
            }
#pragma unroll
            for (int k = 0; k < CS; ++k) s_sk[i][k] = skip[((size_t)bi * CS + k) * n + e];
        }
        __syncthreads();
        const int cnt = s_cnt;
        if (cnt == 0) break;
        if ((int)threadIdx.x < (2 * S - 1) * U * 3)        // padding: valid table rows, results unused
            s_row[cnt + threadIdx.x / 3][threadIdx.x % 3] = s_row[cnt - 1][0];
        __syncthreads();
        // S-stage software pipeline: the 3 U row loads of the next S - 1 groups of U rows are in flight under the arithmetic
        // and the stores of the current group (every load unconditional, so that the waits can be counted)
        cl_f4 buf[S][U][3];
#pragma unroll
        for (int st = 0; st < S - 1; ++st) fp_front_cl_load<U>(st * U, c4, q, a, s_row, buf[st]);
        for (int i = 0; i < cnt; i += S * U) {
#pragma unroll
            for (int st = 0; st < S; ++st) {
                fp_front_cl_load<U>(i + (st + S - 1) * U, c4, q, a, s_row, buf[(st + S - 1) % S]);
                fp_front_cl_rows<CS, U>(i + st * U, cnt, c4, q, on, y, buf[st], s_w, s_sk, s_out, wbr, s, ss, piv, seen);
            }
        }
        if (cnt < CL_STAGE) break;
    }
    if (on) {                                              // (s1, s2, pivot) rows of this tile + its row count
        cl_f4 *P = partial + (size_t)blockIdx.x * 3 * c4;
        P[q] = s;
        P[c4 + q] = ss;
        P[2 * c4 + q] = piv;
        if (q == 0) reinterpret_cast<float *>(partial + (size_t)gridDim.x * 3 * c4)[blockIdx.x] = (float)seen;
    }
}

// MODE 0: (sum x, sum x^2);  MODE 1: (sum g, sum g xhat) of the BatchNorm backward
template <int MODE>
__global__ __launch_bounds__(CL_MAX_THREADS) void bn_reduce_cl_kernel(
    int c4, long long R, int rows, int relu, const cl_f4 *__restrict__ x, const cl_f4 *__restrict__ dz,
    const cl_f4 *__restrict__ scale, const cl_f4 *__restrict__ shift, const cl_f4 *__restrict__ mean,
    const cl_f4 *__restrict__ rstd, cl_f4 *__restrict__ partial)
{
    const int q = threadIdx.x;
    if (q >= c4) return;
    const long long r0 = (long long)blockIdx.x * rows, r1 = min(R, r0 + rows);
    cl_f4 a = {0.f, 0.f, 0.f, 0.f}, b = a, mu = a, rs = a;
    if (MODE == 1) { a = scale[q]; b = shift[q]; mu = mean[q]; rs = rstd[q]; }
    float t0[4] = {0.f, 0.f, 0.f, 0.f}, t1[4] = {0.f, 0.f, 0.f, 0.f};
    cl_f4 piv = {0.f, 0.f, 0.f, 0.f};                      // MODE 0: pivot of the shifted sums = this tile's first row
    if (MODE == 0 && r0 < r1) piv = x[(size_t)r0 * c4 + q];
    const float pv[4] = {piv.x, piv.y, piv.z, piv.w};
    auto one = [&](int k, float xv, float gv, float av, float bv, float mv, float rv) {
        if (MODE == 0) {
            const float d = xv - pv[k];
            t0[k] += d;
            t1[k] = fmaf(d, d, t1[k]);
        } else {
            const float g = (!relu || fmaf(xv, av, bv) > 0.f) ? gv : 0.f;
            t0[k] += g;
            t1[k] = fmaf(g, (xv - mv) * rv, t1[k]);
        }
    };
    constexpr int U = 4;
    for (long long r = r0; r < r1; r += U) {
        cl_f4 xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = r + u < r1;
            xv[u] = ok ? CL_LD(x + (size_t)(r + u) * c4 + q) : (cl_f4){0.f, 0.f, 0.f, 0.f};
            gv[u] = (MODE == 1 && ok) ? CL_LD(dz + (size_t)(r + u) * c4 + q) : (cl_f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u < r1) {
                one(0, xv[u].x, gv[u].x, a.x, b.x, mu.x, rs.x);
                one(1, xv[u].y, gv[u].y, a.y, b.y, mu.y, rs.y);
                one(2, xv[u].z, gv[u].z, a.z, b.z, mu.z, rs.z);
                one(3, xv[u].w, gv[u].w, a.w, b.w, mu.w, rs.w);
            }
        }
    }
    const cl_f4 s0 = {t0[0], t0[1], t0[2], t0[3]}, s1 = {t1[0], t1[1], t1[2], t1[3]};
    if (MODE == 0) {                                       // (s1, s2, pivot) rows of this tile + its row count
        cl_f4 *P = partial + (size_t)blockIdx.x * 3 * c4;
        P[q] = s0;
        P[c4 + q] = s1;
        P[2 * c4 + q] = piv;
        if (q == 0) reinterpret_cast<float *>(partial + (size_t)gridDim.x * 3 * c4)[blockIdx.x] = (float)(r1 > r0 ? r1 - r0 : 0);
    } else {
        cl_f4 *P = partial + (size_t)blockIdx.x * 2 * c4;
        P[q] = s0;
        P[c4 + q] = s1;
    }
}

// MODE 0: out = max(x scale + shift, lo);  MODE 1: dx = k0 (g - c1 - xhat c2)
template <int MODE>
__global__ __launch_bounds__(CL_MAX_THREADS) void bn_apply_cl_kernel(
    int c4, long long R, int rows, int relu, const cl_f4 *__restrict__ x, const cl_f4 *__restrict__ dz,
    const cl_f4 *__restrict__ scale, const cl_f4 *__restrict__ shift, const cl_f4 *__restrict__ mean,
    const cl_f4 *__restrict__ rstd, const cl_f4 *__restrict__ k0, const cl_f4 *__restrict__ c1,
    const cl_f4 *__restrict__ c2, cl_f4 *__restrict__ out)
{
    const int q = threadIdx.x;
    if (q >= c4) return;
    const long long r0 = (long long)blockIdx.x * rows, r1 = min(R, r0 + rows);
    const cl_f4 a = scale[q], b = shift[q];
    cl_f4 mu = a, rs = a, kk = a, m1 = a, m2 = a;
    if (MODE == 1) { mu = mean[q]; rs = rstd[q]; kk = k0[q]; m1 = c1[q]; m2 = c2[q]; }
    const float lo = relu ? 0.f : -INFINITY;
    auto one = [&](float xv, float gv, float av, float bv, float mv, float rv, float kv, float p1, float p2) {
        if (MODE == 0) return fmaxf(fmaf(xv, av, bv), lo);
        const float g = (!relu || fmaf(xv, av, bv) > 0.f) ? gv : 0.f;
        return kv * (g - p1 - (xv - mv) * rv * p2);
    };
    constexpr int U = 4;
    for (long long r = r0; r < r1; r += U) {
        cl_f4 xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = r + u < r1;
            xv[u] = ok ? CL_LD(x + (size_t)(r + u) * c4 + q) : (cl_f4){0.f, 0.f, 0.f, 0.f};
            gv[u] = (MODE == 1 && ok) ? CL_LD(dz + (size_t)(r + u) * c4 + q) : (cl_f4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u < r1) {
                cl_f4 o;
                o.x = one(xv[u].x, gv[u].x, a.x, b.x, mu.x, rs.x, kk.x, m1.x, m2.x);
                o.y = one(xv[u].y, gv[u].y, a.y, b.y, mu.y, rs.y, kk.y, m1.y, m2.y);
                o.z = one(xv[u].z, gv[u].z, a.z, b.z, mu.z, rs.z, kk.z, m1.z, m2.z);
                o.w = one(xv[u].w, gv[u].w, a.w, b.w, mu.w, rs.w, kk.w, m1.w, m2.w);
                CL_ST(o, out + (size_t)(r + u) * c4 + q);
            }
        }
    }
}

// (tiles, K, C) float -> (C, K) double, K <= CL_MAX_SUMS.  One workgroup per (64 channels, one of the K sums): lanes =
// consecutive channels (coalesced rows), the tiles dealt to its 16 waves; every wave sums its tiles in ascending order
// with 4 tiles in flight, then the 16 wave sums are added in a fixed tree: the same result on every run.
constexpr int CL_MAX_SUMS = 2 + 2 * CL_MAX_SKIP;
__global__ __launch_bounds__(1024) void bn_sums_cl_kernel(int tiles, int c, int K, const float *__restrict__ partial,
                                                          double *__restrict__ sums)
{
    constexpr int PARTS = 16, U = 4;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6, k = blockIdx.y;
    const int ch = blockIdx.x * 64 + lane;
    __shared__ double red[PARTS][64];
    double acc = 0.0;
    if (ch < c) {
        int t = part;
        for (; t + (U - 1) * PARTS < tiles; t += U * PARTS) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = partial[((size_t)(t + u * PARTS) * K + k) * c + ch];
#pragma unroll
            for (int u = 0; u < U; ++u) acc += (double)v[u];
        }
        for (; t < tiles; t += PARTS) acc += (double)partial[((size_t)t * K + k) * c + ch];
    }
    red[part][lane] = acc;
    __syncthreads();
#pragma unroll
    for (int h = PARTS / 2; h >= 1; h >>= 1) {
        if (part < h) red[part][lane] += red[part + h][lane];
        __syncthreads();
    }
    if (part == 0 && ch < c) sums[(size_t)K * ch + k] = red[0][lane];
}

// the statistics records of fp_front_cl / bn_stats_cl: (tiles, 3, C) = (s1, s2, pivot) followed by `tiles` row counts.  Each
// record is turned back into (sum x, sum x^2) in fp64 before it is added (see bnrelu.hip bn_stats_kernel); same fixed order.
__global__ __launch_bounds__(1024) void bn_sums_shifted_cl_kernel(int tiles, int c, const float *__restrict__ partial,
                                                                  double *__restrict__ sums)
{
    constexpr int PARTS = 16;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + lane;
    const float *counts = partial + (size_t)tiles * 3 * c;
    __shared__ double red[PARTS][64][2];
    double a0 = 0.0, a1 = 0.0;
    if (ch < c) {
        for (int t = part; t < tiles; t += PARTS) {
            const double s1 = partial[((size_t)t * 3) * c + ch], s2 = partial[((size_t)t * 3 + 1) * c + ch];
            const double p = partial[((size_t)t * 3 + 2) * c + ch], n = counts[t];
            a0 += s1 + n * p;
            a1 += s2 + 2.0 * p * s1 + n * p * p;
        }
    }
    red[part][lane][0] = a0;
    red[part][lane][1] = a1;
    __syncthreads();
#pragma unroll
    for (int h = PARTS / 2; h >= 1; h >>= 1) {
        if (part < h) {
            red[part][lane][0] += red[part + h][lane][0];
            red[part][lane][1] += red[part + h][lane][1];
        }
        __syncthreads();
    }
    if (part == 0 && ch < c) {
        sums[2 * ch] = red[0][lane][0];
        sums[2 * ch + 1] = red[0][lane][1];
    }
}

// BatchNorm backward reduce of the FP front end with the skip-weight gradient riding along: per tile
//   [0] sum g   [1] sum g xhat   [2 + k] sum g skip_k   [2 + CS + k] sum xhat skip_k     (g = dz [x scale + shift > 0])
// -- everything grad_Wb = sum_e gy_e skip_e needs once the two means are known (gy = k0 (g - c1 - xhat c2)), so that gy
// itself is never written (the gather of the interpolation gradient forms it on the fly, gather_group.hip).
template <int CS>
__global__ __launch_bounds__(CL_MAX_THREADS) void bn_reduce_skip_cl_kernel(
    int c4, int R, int n, int rows, int relu, const cl_f4 *__restrict__ x, const cl_f4 *__restrict__ dz,
    const cl_f4 *__restrict__ scale, const cl_f4 *__restrict__ shift, const cl_f4 *__restrict__ mean,
    const cl_f4 *__restrict__ rstd, const float *__restrict__ skip, cl_f4 *__restrict__ partial)
{
    const int q = threadIdx.x;
    if (q >= c4) return;
    const int r0 = min(R, (int)blockIdx.x * rows), r1 = min(R, r0 + rows);
    const cl_f4 a = scale[q], b = shift[q], mu = mean[q], rs = rstd[q];
    const cl_f4 zero = {0.f, 0.f, 0.f, 0.f};
    cl_f4 t0 = zero, t1 = zero, s1[CS > 0 ? CS : 1], s3[CS > 0 ? CS : 1];
#pragma unroll
    for (int k = 0; k < CS; ++k) s1[k] = s3[k] = zero;
    constexpr int U = 2;
    for (int r = r0; r < r1; r += U) {
        cl_f4 xv[U], gv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = min(r + u, r1 - 1);
            xv[u] = CL_LD(x + (size_t)rr * c4 + q);
            gv[u] = CL_LD(dz + (size_t)rr * c4 + q);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (r + u < r1) {                                  // uniform
                const int bi = (r + u) / n, e = (r + u) - bi * n;
                cl_f4 g, xh;
                g.x = (!relu || fmaf(xv[u].x, a.x, b.x) > 0.f) ? gv[u].x : 0.f;
                g.y = (!relu || fmaf(xv[u].y, a.y, b.y) > 0.f) ? gv[u].y : 0.f;
                g.z = (!relu || fmaf(xv[u].z, a.z, b.z) > 0.f) ? gv[u].z : 0.f;
                g.w = (!relu || fmaf(xv[u].w, a.w, b.w) > 0.f) ? gv[u].w : 0.f;
                xh = (xv[u] - mu) * rs;
                t0 = t0 + g;
                t1 = __builtin_elementwise_fma(g, xh, t1);
#pragma unroll
                for (int k = 0; k < CS; ++k) {
                    const float sk = skip[((size_t)bi * CS + k) * n + e];   // wave-uniform
                    s1[k] = __builtin_elementwise_fma(g, (cl_f4)(sk), s1[k]);
                    s3[k] = __builtin_elementwise_fma(xh, (cl_f4)(sk), s3[k]);
                }
            }
        }
    }
    cl_f4 *P = partial + (size_t)blockIdx.x * (2 + 2 * CS) * c4;
    P[q] = t0;
    P[c4 + q] = t1;
#pragma unroll
    for (int k = 0; k < CS; ++k) {
        P[(size_t)(2 + k) * c4 + q] = s1[k];
        P[(size_t)(2 + CS + k) * c4 + q] = s3[k];
    }
}

// grad_wb[c, j] = scale_c (S1[c, j] - c1_c S2[j] - c2_c S3[c, j]) from the sums of bn_reduce_skip_cl_kernel (sums_k (C, 2 + 2 CS)
// fp64: [2 + j] = S1, [2 + CS + j] = S3), the means c1 / c2 of geot_bn_bwd_coef and S2[j] = sum over all points of skip_j
__global__ __launch_bounds__(256) void fp_skip_wgrad_cl_kernel(int c, int cs, const double *__restrict__ sums_k,
                                                               const float *__restrict__ scale, const float *__restrict__ c1,
                                                               const float *__restrict__ c2, const double *__restrict__ s2,
                                                               float *__restrict__ gwb)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= c * cs) return;
    const int ch = x / cs, j = x - ch * cs, K = 2 + 2 * cs;
    const double v = sums_k[(size_t)ch * K + 2 + j] - (double)c1[ch] * s2[j] - (double)c2[ch] * sums_k[(size_t)ch * K + 2 + cs + j];
    gwb[x] = (float)((double)scale[ch] * v);
}

} // namespace geot

using namespace geot;

// number of row tiles (= workgroups = rows of the (T, 2, c) partial-sum buffer) of a *_cl launch over batches x
// rows_per_batch rows; -1 when the kernels do not cover the shape (C % 4, C > 4096, more than 2^31 rows)
GEOT_EXPORT int geot_cl_tiles(int batches, long long rows_per_batch, int c)
{
    if (batches < 1 || !cl_dims_ok(rows_per_batch, c)) return -1;
    const long long rows = rows_per_batch * batches;
    if (rows > 0x7fffffffLL) return -1;
    return cl_tiles_for(rows, c);
}

// workgroups of the fp_front_cl launch = rows of its partial-sum buffer: one round of what the CUs hold of THIS
// instantiation (registers decide: 3 workgroups of 6 waves per CU at C = 1536)
template <int CS>
static int fp_front_cl_blocks_of(int block)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fp_front_cl_kernel<CS>, block, 0) != hipSuccess || nb < 1) nb = 1;
    return nb;
}
static int fp_front_cl_tiles(long long rows, int c, int cs)
{
    static int cache[GEOT_DEV_SLOTS][CL_MAX_SKIP + 1][CL_MAX_THREADS / 64 + 1];   // [device][cs][waves] -> workgroups per CU
    const int block = cl_block(c / 4), waves = block / 64;
    int &per_cu = cache[device_slot()][cs][waves];
    if (!per_cu) {
        switch (cs) {
        case 0: per_cu = fp_front_cl_blocks_of<0>(block); break;
        case 1: per_cu = fp_front_cl_blocks_of<1>(block); break;
        case 2: per_cu = fp_front_cl_blocks_of<2>(block); break;
        case 3: per_cu = fp_front_cl_blocks_of<3>(block); break;
        case 4: per_cu = fp_front_cl_blocks_of<4>(block); break;
        case 5: per_cu = fp_front_cl_blocks_of<5>(block); break;
        case 6: per_cu = fp_front_cl_blocks_of<6>(block); break;
        case 7: per_cu = fp_front_cl_blocks_of<7>(block); break;
        default: per_cu = fp_front_cl_blocks_of<8>(block); break;
        }
    }
    long long t = (long long)cl_cus() * per_cu;
    if (const char *mult = getenv("GEOT_CL_TILES_MULT")) t *= atoi(mult) > 0 ? atoi(mult) : 1;   // lab
    if (t > (rows + 15) / 16) t = (rows + 15) / 16;
    return (int)(t < 1 ? 1 : t);
}
GEOT_EXPORT int geot_fp_front_cl_tiles(int b, int c, int n, int cs)
{
    // (rows up to 2^30: the row deal of fp_front_cl_kernel steps past the end of its range by up to a stage of granules)
    if (b < 1 || cs < 0 || cs > CL_MAX_SKIP || !cl_dims_ok(n, c) || (long long)b * n > 0x3fffffffLL) return -1;
    return fp_front_cl_tiles((long long)b * n, c, cs);
}

GEOT_EXPORT int geot_fp_front_cl(int b, int c, int m, int n, int cs, const float *a_cl, const int *idx, const float *weight,
                                 const float *skip, const float *wb, const int *order, float *y_cl, float *partial,
                                 void *stream)
{
    if (b < 0 || c < 0 || m < 0 || n < 0 || cs < 0 || cs > CL_MAX_SKIP) return hipErrorInvalidValue;
    if (b == 0 || n == 0 || c == 0) return hipSuccess;
    const int tiles = geot_fp_front_cl_tiles(b, c, n, cs);
    if (tiles < 1 || m < 1 || (cs > 0 && (!skip || !wb)) || (long long)b * m > 0x7fffffffLL) return hipErrorInvalidValue;
    const int c4 = c / 4, R = b * n;
    int granule = getenv("GEOT_CL_GRANULE") ? atoi(getenv("GEOT_CL_GRANULE")) : 8;       // lab
    if (granule < 1) granule = (R + tiles - 1) / tiles;                                    // <= 0: one contiguous run each
#define GEOT_FPCL(CSV)                                                                                                      \
    hipLaunchKernelGGL(fp_front_cl_kernel<CSV>, dim3(tiles), dim3(cl_block(c4)), 0, (hipStream_t)stream, c4, m, n, R,       \
                       granule, (const cl_f4 *)a_cl, idx, weight, skip, wb, order, (cl_f4 *)y_cl, (cl_f4 *)partial)
    switch (cs) {
    case 0: GEOT_FPCL(0); break;
    case 1: GEOT_FPCL(1); break;
    case 2: GEOT_FPCL(2); break;
    case 3: GEOT_FPCL(3); break;
    case 4: GEOT_FPCL(4); break;
    case 5: GEOT_FPCL(5); break;
    case 6: GEOT_FPCL(6); break;
    case 7: GEOT_FPCL(7); break;
    default: GEOT_FPCL(8); break;
    }
#undef GEOT_FPCL
    return hipGetLastError();
}

static int cl_launch_dims(long long rows, int c, int *per, int *tiles)
{
    if (!cl_dims_ok(rows, c) || rows > 0x7fffffffLL) return hipErrorInvalidValue;
    *tiles = cl_tiles_for(rows, c);
    *per = (int)((rows + *tiles - 1) / *tiles);
    return hipSuccess;
}

GEOT_EXPORT int geot_bn_stats_cl(long long rows, int c, const float *x, float *partial, void *stream)
{
    if (rows == 0 || c == 0) return hipSuccess;
    int per, tiles;
    if (cl_launch_dims(rows, c, &per, &tiles)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_reduce_cl_kernel<0>, dim3(tiles), dim3(cl_block(c / 4)), 0, (hipStream_t)stream, c / 4, rows, per, 0,
                       (const cl_f4 *)x, nullptr, nullptr, nullptr, nullptr, nullptr, (cl_f4 *)partial);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_bwd_reduce_cl(long long rows, int c, int relu, const float *x, const float *dz, const float *scale,
                                      const float *shift, const float *mean, const float *rstd, float *partial, void *stream)
{
    if (rows == 0 || c == 0) return hipSuccess;
    int per, tiles;
    if (cl_launch_dims(rows, c, &per, &tiles)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_reduce_cl_kernel<1>, dim3(tiles), dim3(cl_block(c / 4)), 0, (hipStream_t)stream, c / 4, rows, per, relu,
                       (const cl_f4 *)x, (const cl_f4 *)dz, (const cl_f4 *)scale, (const cl_f4 *)shift, (const cl_f4 *)mean,
                       (const cl_f4 *)rstd, (cl_f4 *)partial);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_apply_cl(long long rows, int c, int relu, const float *x, const float *scale, const float *shift,
                                 float *out, void *stream)
{
    if (rows == 0 || c == 0) return hipSuccess;
    int per, tiles;
    if (cl_launch_dims(rows, c, &per, &tiles)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_apply_cl_kernel<0>, dim3(tiles), dim3(cl_block(c / 4)), 0, (hipStream_t)stream, c / 4, rows, per, relu,
                       (const cl_f4 *)x, nullptr, (const cl_f4 *)scale, (const cl_f4 *)shift, nullptr, nullptr, nullptr, nullptr,
                       nullptr, (cl_f4 *)out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_bwd_apply_cl(long long rows, int c, int relu, const float *x, const float *dz, const float *scale,
                                     const float *shift, const float *mean, const float *rstd, const float *k0, const float *c1,
                                     const float *c2, float *dx, void *stream)
{
    if (rows == 0 || c == 0) return hipSuccess;
    int per, tiles;
    if (cl_launch_dims(rows, c, &per, &tiles)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bn_apply_cl_kernel<1>, dim3(tiles), dim3(cl_block(c / 4)), 0, (hipStream_t)stream, c / 4, rows, per, relu,
                       (const cl_f4 *)x, (const cl_f4 *)dz, (const cl_f4 *)scale, (const cl_f4 *)shift, (const cl_f4 *)mean,
                       (const cl_f4 *)rstd, (const cl_f4 *)k0, (const cl_f4 *)c1, (const cl_f4 *)c2, (cl_f4 *)dx);
    return hipGetLastError();
}

GEOT_EXPORT int geot_bn_sums_cl(int tiles, int c, const float *partial, double *sums, void *stream)
{
    if (tiles < 0 || c < 0) return hipErrorInvalidValue;
    if (c == 0) return hipSuccess;
    hipLaunchKernelGGL(bn_sums_cl_kernel, dim3((c + 63) / 64, 2), dim3(1024), 0, (hipStream_t)stream, tiles, c, 2, partial, sums);
    return hipGetLastError();
}

// floats of the statistics buffer fp_front_cl / bn_stats_cl fill for `tiles` tiles: (tiles, 3, c) + tiles counts
GEOT_EXPORT long long geot_cl_stat_floats(int tiles, int c) { return tiles < 0 || c < 0 ? -1 : (long long)tiles * (3LL * c + 1); }

GEOT_EXPORT int geot_bn_sums_shifted_cl(int tiles, int c, const float *partial, double *sums, void *stream)
{
    if (tiles < 0 || c < 0) return hipErrorInvalidValue;
    if (c == 0) return hipSuccess;
    hipLaunchKernelGGL(bn_sums_shifted_cl_kernel, dim3((c + 63) / 64), dim3(1024), 0, (hipStream_t)stream, tiles, c, partial, sums);
    return hipGetLastError();
}

// partial (tiles, K, c) -> sums (c, K) fp64 for K = 2 + 2 cs (the reduce-with-skip pass below)
GEOT_EXPORT int geot_bn_sums_k_cl(int tiles, int c, int k, const float *partial, double *sums, void *stream)
{
    if (tiles < 0 || c < 0 || k < 1 || k > CL_MAX_SUMS) return hipErrorInvalidValue;
    if (c == 0) return hipSuccess;
    hipLaunchKernelGGL(bn_sums_cl_kernel, dim3((c + 63) / 64, k), dim3(1024), 0, (hipStream_t)stream, tiles, c, k, partial, sums);
    return hipGetLastError();
}

// BatchNorm backward reduce over x, dz (b, n, c) point-major with the sums of the skip-weight gradient: partial
// (geot_cl_tiles(1, b n, c), 2 + 2 cs, c); skip (b, cs, n) channels-first
GEOT_EXPORT int geot_bn_bwd_reduce_skip_cl(int b, int n, int c, int cs, int relu, const float *x, const float *dz,
                                           const float *scale, const float *shift, const float *mean, const float *rstd,
                                           const float *skip, float *partial, void *stream)
{
    if (b < 0 || n < 0 || c < 0 || cs < 0 || cs > CL_MAX_SKIP || (cs > 0 && !skip)) return hipErrorInvalidValue;
    if (b == 0 || n == 0 || c == 0) return hipSuccess;
    int per, tiles;
    if (cl_launch_dims((long long)b * n, c, &per, &tiles)) return hipErrorInvalidValue;
#define GEOT_RS(CSV)                                                                                                          \
    hipLaunchKernelGGL(bn_reduce_skip_cl_kernel<CSV>, dim3(tiles), dim3(cl_block(c / 4)), 0, (hipStream_t)stream, c / 4, b * n, n, \
                       per, relu, (const cl_f4 *)x, (const cl_f4 *)dz, (const cl_f4 *)scale, (const cl_f4 *)shift,             \
                       (const cl_f4 *)mean, (const cl_f4 *)rstd, skip, (cl_f4 *)partial)
    switch (cs) {
    case 0: GEOT_RS(0); break;
    case 1: GEOT_RS(1); break;
    case 2: GEOT_RS(2); break;
    case 3: GEOT_RS(3); break;
    case 4: GEOT_RS(4); break;
    case 5: GEOT_RS(5); break;
    case 6: GEOT_RS(6); break;
    case 7: GEOT_RS(7); break;
    default: GEOT_RS(8); break;
    }
#undef GEOT_RS
    return hipGetLastError();
}

// grad of the skip weights wb (c, cs) of fp_front_cl behind a BatchNorm, from the reduce-with-skip sums (see the kernel)
GEOT_EXPORT int geot_fp_skip_wgrad_cl(int c, int cs, const double *sums_k, const float *scale, const float *c1, const float *c2,
                                      const double *s2, float *gwb, void *stream)
{
    if (c < 0 || cs < 0 || cs > CL_MAX_SKIP) return hipErrorInvalidValue;
    if (c == 0 || cs == 0) return hipSuccess;
    hipLaunchKernelGGL(fp_skip_wgrad_cl_kernel, dim3((c * cs + 255) / 256), dim3(256), 0, (hipStream_t)stream, c, cs, sums_k, scale,
                       c1, c2, s2, gwb);
    return hipGetLastError();
}
