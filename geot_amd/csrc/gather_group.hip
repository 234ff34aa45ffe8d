// gather_group.hip -- index gathers / scatter-adds for gfx950 (MI355X).
//
// Replaces (behaviour, not code):
//   gather            : pointnet2/_ext_src/src/sampling_gpu.cu:11-60,
//                       openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:15-92
//   group             : pointnet2/_ext_src/src/group_points_gpu.cu:11-78,
//                       openpoints/cpp/pointnet2_batch/src/group_points_gpu.cu:14-93
//   three_interpolate : pointnet2/_ext_src/src/interpolate_gpu.cu:75-157,
//                       openpoints/cpp/pointnet2_batch/src/interpolate_gpu.cu:84-170
//   channels-last ops : openpoints/cpp/pointops/src/{grouping,interpolation,
//                       subtraction,aggregation}/*_cuda_kernel.cu
//
// These are HBM-bound copies. Layout rule used throughout: consecutive lanes
// walk the contiguous output dimension so stores coalesce, each lane loads its
// index (and weights) once and reuses them across a chunk of channels, and the
// random reads hit rows that are L2-resident (one channel row of a 24k-point
// cloud is 96 KB).
#include "geot_common.h"
#include "geot_hip.h"
#include "tile_scatter.h"
#include <cstdlib>

namespace geot {

constexpr int GG_THREADS = 256;
constexpr int GG_CCHUNK = 8; // channels handled per lane (index reuse)

// out[b,c,j] = points[b,c,idx[b,j]]
__global__ __launch_bounds__(GG_THREADS) void gather_points_kernel(
    int c, int n, int m, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out)
{
    const int bi = blockIdx.z, c0 = blockIdx.y * GG_CCHUNK;
    const int j = blockIdx.x * GG_THREADS + threadIdx.x;
    if (j >= m) return;
    const int a = idx[(size_t)bi * m + j];
    const int cend = min(c0 + GG_CCHUNK, c);
    for (int l = c0; l < cend; ++l)
        out[((size_t)bi * c + l) * m + j] = points[((size_t)bi * c + l) * n + a];
}

__global__ __launch_bounds__(GG_THREADS) void gather_points_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points)
{
    const int bi = blockIdx.z, c0 = blockIdx.y * GG_CCHUNK;
    const int j = blockIdx.x * GG_THREADS + threadIdx.x;
    if (j >= m) return;
    const int a = idx[(size_t)bi * m + j];
    const int cend = min(c0 + GG_CCHUNK, c);
    for (int l = c0; l < cend; ++l)
        atomicAdd(grad_points + ((size_t)bi * c + l) * n + a, grad_out[((size_t)bi * c + l) * m + j]);
}

// out[b,c,j,k] = points[b,c,idx[b,j,k]]; (j,k) flattened = the contiguous dim.
__global__ __launch_bounds__(GG_THREADS) void group_points_kernel(
    int c, int n, int npns, const float *__restrict__ points, const int *__restrict__ idx,
    float *__restrict__ out)
{
    const int bi = blockIdx.z, c0 = blockIdx.y * GG_CCHUNK;
    const int e = blockIdx.x * GG_THREADS + threadIdx.x;
    if (e >= npns) return;
    const int a = idx[(size_t)bi * npns + e];
    const int cend = min(c0 + GG_CCHUNK, c);
    for (int l = c0; l < cend; ++l)
        out[((size_t)bi * c + l) * npns + e] = points[((size_t)bi * c + l) * n + a];
}

__global__ __launch_bounds__(GG_THREADS) void group_points_grad_kernel(
    int c, int n, int npns, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_points)
{
    const int bi = blockIdx.z, c0 = blockIdx.y * GG_CCHUNK;
    const int e = blockIdx.x * GG_THREADS + threadIdx.x;
    if (e >= npns) return;
    const int a = idx[(size_t)bi * npns + e];
    const int cend = min(c0 + GG_CCHUNK, c);
    for (int l = c0; l < cend; ++l)
        atomicAdd(grad_points + ((size_t)bi * c + l) * n + a, grad_out[((size_t)bi * c + l) * npns + e]);
}

// out[b,c,j] = sum_t points[b,c,idx[b,j,t]] * weight[b,j,t]
// Evaluation order follows the reference expression p0*w0 + p1*w1 + p2*w2.
__global__ __launch_bounds__(GG_THREADS) void three_interpolate_kernel(
    int c, int m, int n, const float *__restrict__ points, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out, size_t out_bstride)
{
    const int bi = blockIdx.z, c0 = blockIdx.y * GG_CCHUNK;
    const int j = blockIdx.x * GG_THREADS + threadIdx.x;
    if (j >= n) return;
    const size_t o = ((size_t)bi * n + j) * 3;
    const int i0 = idx[o], i1 = idx[o + 1], i2 = idx[o + 2];
    const float w0 = weight[o], w1 = weight[o + 1], w2 = weight[o + 2];
    const int cend = min(c0 + GG_CCHUNK, c);
    for (int l = c0; l < cend; ++l) {
        const float *P = points + ((size_t)bi * c + l) * m;
        out[(size_t)bi * out_bstride + (size_t)l * n + j] = P[i0] * w0 + P[i1] * w1 + P[i2] * w2;
    }
}

__global__ __launch_bounds__(GG_THREADS) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points)
{
    const int bi = blockIdx.z, c0 = blockIdx.y * GG_CCHUNK;
    const int j = blockIdx.x * GG_THREADS + threadIdx.x;
    if (j >= n) return;
    const size_t o = ((size_t)bi * n + j) * 3;
    const int i0 = idx[o], i1 = idx[o + 1], i2 = idx[o + 2];
    const float w0 = weight[o], w1 = weight[o + 1], w2 = weight[o + 2];
    const int cend = min(c0 + GG_CCHUNK, c);
    for (int l = c0; l < cend; ++l) {
        const float g = grad_out[((size_t)bi * c + l) * n + j];
        float *G = grad_points + ((size_t)bi * c + l) * m;
        atomicAdd(G + i0, g * w0);
        atomicAdd(G + i1, g * w1);
        atomicAdd(G + i2, g * w2);
    }
}

// ---- channels-last (openpoints pointops) -----------------------------------
// One lane per output element with the channel index fastest, so both the
// gathered row reads and the stores are contiguous across the wave.
__global__ __launch_bounds__(GG_THREADS) void grouping_cl_kernel(
    long long total, int ns, int c, const float *__restrict__ input, const int *__restrict__ idx,
    float *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * GG_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * GG_THREADS) {
        int ch = (int)(e % c);
        long long ms = e / c; // m_idx*ns + s
        out[e] = input[(size_t)idx[ms] * c + ch];
    }
}

__global__ __launch_bounds__(GG_THREADS) void grouping_cl_grad_kernel(
    long long total, int ns, int c, const float *__restrict__ grad_out, const int *__restrict__ idx,
    float *__restrict__ grad_in)
{
    for (long long e = (long long)blockIdx.x * GG_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * GG_THREADS) {
        int ch = (int)(e % c);
        long long ms = e / c;
        atomicAdd(grad_in + (size_t)idx[ms] * c + ch, grad_out[e]);
    }
}

// The reference accumulates into a pre-zeroed output (output[index] += ...).
__global__ __launch_bounds__(GG_THREADS) void interpolation_cl_kernel(
    long long total, int c, int k, const float *__restrict__ input, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * GG_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * GG_THREADS) {
        int ch = (int)(e % c);
        long long ni = e / c;
        float acc = out[e];
        for (int t = 0; t < k; ++t)
            acc += input[(size_t)idx[ni * k + t] * c + ch] * weight[ni * k + t];
        out[e] = acc;
    }
}

__global__ __launch_bounds__(GG_THREADS) void interpolation_cl_grad_kernel(
    long long total, int c, int k, const float *__restrict__ grad_out, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_in)
{
    for (long long e = (long long)blockIdx.x * GG_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * GG_THREADS) {
        int ch = (int)(e % c);
        long long ni = e / c;
        float g = grad_out[e];
        for (int t = 0; t < k; ++t)
            atomicAdd(grad_in + (size_t)idx[ni * k + t] * c + ch, g * weight[ni * k + t]);
    }
}

__global__ __launch_bounds__(GG_THREADS) void subtraction_cl_kernel(
    long long total, int ns, int c, const float *__restrict__ in1, const float *__restrict__ in2,
    const int *__restrict__ idx, float *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * GG_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * GG_THREADS) {
        int ch = (int)(e % c);
        long long ms = e / c;
        long long ni = ms / ns;
        out[e] = in1[ni * c + ch] - in2[(size_t)idx[ms] * c + ch];
    }
}

__global__ __launch_bounds__(GG_THREADS) void subtraction_cl_grad_kernel(
    long long total, int ns, int c, const int *__restrict__ idx, const float *__restrict__ grad_out,
    float *__restrict__ g1, float *__restrict__ g2)
{
    for (long long e = (long long)blockIdx.x * GG_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * GG_THREADS) {
        int ch = (int)(e % c);
        long long ms = e / c;
        long long ni = ms / ns;
        float g = grad_out[e];
        atomicAdd(g1 + ni * c + ch, g);
        atomicAdd(g2 + (size_t)idx[ms] * c + ch, -g);
    }
}

__global__ __launch_bounds__(GG_THREADS) void aggregation_cl_kernel(
    long long total, int ns, int c, int w_c, const float *__restrict__ input,
    const float *__restrict__ position, const float *__restrict__ weight,
    const int *__restrict__ idx, float *__restrict__ out)
{
    for (long long e = (long long)blockIdx.x * GG_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * GG_THREADS) {
        int ch = (int)(e % c);
        long long ni = e / c;
        int wch = ch % w_c;
        float acc = out[e];
        for (int s = 0; s < ns; ++s) {
            long long ms = ni * ns + s;
            acc += (input[(size_t)idx[ms] * c + ch] + position[ms * c + ch]) * weight[ms * w_c + wch];
        }
        out[e] = acc;
    }
}

__global__ __launch_bounds__(GG_THREADS) void aggregation_cl_grad_kernel(
    long long total, int ns, int c, int w_c, const float *__restrict__ input,
    const float *__restrict__ position, const float *__restrict__ weight,
    const int *__restrict__ idx, const float *__restrict__ grad_out, float *__restrict__ g_in,
    float *__restrict__ g_pos, float *__restrict__ g_w)
{
    for (long long e = (long long)blockIdx.x * GG_THREADS + threadIdx.x; e < total;
         e += (long long)gridDim.x * GG_THREADS) {
        int ch = (int)(e % c);
        long long ni = e / c;
        int wch = ch % w_c;
        float g = grad_out[e];
        for (int s = 0; s < ns; ++s) {
            long long ms = ni * ns + s;
            size_t ii = (size_t)idx[ms] * c + ch;
            float w = weight[ms * w_c + wch];
            atomicAdd(g_in + ii, g * w);
            g_pos[ms * c + ch] = g * w;
            atomicAdd(g_w + ms * w_c + wch, g * (input[ii] + position[ms * c + ch]));
        }
    }
}

// ---- atomic-friendly backward: accumulate channels-last, then transpose ------------------------
// In the reference layout (B,C,N) a scatter-add "grad[b,c,idx] += g" puts the 64 lanes of a wave on
// 64 different rows; MI355X executes float atomics at the memory side, one request per touched
// 64-byte segment, so that shape runs ~17x below the atomic rate (MI355X_MICROARCH.md "Global float
// atomics").  Here the wave's lanes are 64 consecutive CHANNELS of one target point in a (B,M,C)
// workspace -- each atomic wave-instruction is one contiguous 256-B row segment -- and a tiled
// transpose-add folds the workspace into the caller's (B,C,M) gradient afterwards.
constexpr int SC_TILE = 64;

// src (B,C,L) channels-first; item e (< L) scatters src[b,:,e]*w[e,t] to ws[b, tgt[e,t], :], t < NT.
template <int NT, bool WEIGHTED>
__global__ __launch_bounds__(256) void scatter_rows_cl_kernel(
    int c, int L, int M, const float *__restrict__ src, size_t src_bstride, const int *__restrict__ tgt,
    const float *__restrict__ w, float *__restrict__ ws)
{
    __shared__ float tile[SC_TILE][SC_TILE + 1];
    const int b = blockIdx.z, c0 = blockIdx.y * SC_TILE, e0 = blockIdx.x * SC_TILE;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < SC_TILE / 4; ++r) {
        int cc = r * 4 + ty;
        tile[cc][tx] = (c0 + cc < c && e0 + tx < L) ? src[(size_t)b * src_bstride + (size_t)(c0 + cc) * L + e0 + tx] : 0.f;
    }
    __syncthreads();
    const int lane = tx, wave = ty;
    if (c0 + lane >= c) return;
    for (int q = wave; q < SC_TILE; q += 4) {
        int e = e0 + q;
        if (e >= L) break;
        float g = tile[lane][q];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int k = tgt[((size_t)b * L + e) * NT + t];
            float v = WEIGHTED ? g * w[((size_t)b * L + e) * NT + t] : g;
            atomicAdd(ws + ((size_t)b * M + k) * c + c0 + lane, v);
        }
    }
}

// dst[b,c,k] += ws[b,k,c]
__global__ __launch_bounds__(256) void transpose_add_kernel(int c, int M, const float *__restrict__ ws,
                                                            float *__restrict__ dst)
{
    __shared__ float tile[SC_TILE][SC_TILE + 1];
    const int b = blockIdx.z, c0 = blockIdx.y * SC_TILE, k0 = blockIdx.x * SC_TILE;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < SC_TILE / 4; ++r) {
        int kk = r * 4 + ty;
        tile[kk][tx] = (k0 + kk < M && c0 + tx < c) ? ws[((size_t)b * M + k0 + kk) * c + c0 + tx] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SC_TILE / 4; ++r) {
        int cc = r * 4 + ty;
        if (c0 + cc < c && k0 + tx < M) dst[((size_t)b * c + c0 + cc) * M + k0 + tx] += tile[tx][cc];
    }
}

// ---- EdgeConv graph feature (DGCNN_Propagation.get_graph_feature, transformer.py:343-364) -------
// out[b, c,     i, j] = x_k[b, c, idx[b,i,j]] - x_q[b, c, i]        (c < C)
// out[b, C + c, i, j] = x_q[b, c, i]
// One pass instead of the reference's transpose + fancy-index gather + permute + expand + cat chain.
// Lanes walk i (the contiguous dimension of x_q and, with j, of out); indices are loaded once per
// lane and reused over GG_CCHUNK channels; for k = 4, 8, 16 the neighbour ids are held in registers (int4).

typedef float gf_f4 __attribute__((ext_vector_type(4)));

template <int K4>   // K4 > 0: k == 4*K4 and rows are float4-aligned; 0: generic k
__global__ __launch_bounds__(GG_THREADS) void graph_feature_kernel(
    int c, int nq, int nk, int k, const float *__restrict__ x_q, const float *__restrict__ x_k,
    const int *__restrict__ idx, float *__restrict__ out)
{
    const int bi = blockIdx.z, c0 = blockIdx.y * GG_CCHUNK;
    const int i = blockIdx.x * GG_THREADS + threadIdx.x;
    if (i >= nq) return;
    const int *I = idx + ((size_t)bi * nq + i) * k;
    const int cend = min(c0 + GG_CCHUNK, c);
    if constexpr (K4 > 0) {
        int4 nb[K4];
#pragma unroll
        for (int q = 0; q < K4; ++q) nb[q] = reinterpret_cast<const int4 *>(I)[q];
        for (int l = c0; l < cend; ++l) {
            const float q = x_q[((size_t)bi * c + l) * nq + i];
            const float *K = x_k + ((size_t)bi * c + l) * nk;
            gf_f4 *o1 = reinterpret_cast<gf_f4 *>(out + (((size_t)bi * 2 * c + l) * nq + i) * k);
            gf_f4 *o2 = reinterpret_cast<gf_f4 *>(out + (((size_t)bi * 2 * c + c + l) * nq + i) * k);
#pragma unroll
            for (int t = 0; t < K4; ++t) {
                gf_f4 v = {K[nb[t].x] - q, K[nb[t].y] - q, K[nb[t].z] - q, K[nb[t].w] - q};
                gf_f4 w = {q, q, q, q};
                __builtin_nontemporal_store(v, o1 + t);
                __builtin_nontemporal_store(w, o2 + t);
            }
        }
    } else {
        for (int l = c0; l < cend; ++l) {
            const float q = x_q[((size_t)bi * c + l) * nq + i];
            const float *K = x_k + ((size_t)bi * c + l) * nk;
            float *o1 = out + (((size_t)bi * 2 * c + l) * nq + i) * k;
            float *o2 = out + (((size_t)bi * 2 * c + c + l) * nq + i) * k;
            for (int j = 0; j < k; ++j) {
                o1[j] = K[I[j]] - q;
                o2[j] = q;
            }
        }
    }
}

// grad_xq[b,c,i] += sum_j (g[b,C+c,i,j] - g[b,c,i,j]); the x_k part is a scatter of g[:, :C] (below)
template <int K4>
__global__ __launch_bounds__(GG_THREADS) void graph_feature_grad_q_kernel(
    int c, int nq, int k, const float *__restrict__ grad_out, float *__restrict__ grad_xq)
{
    const int bi = blockIdx.z, l = blockIdx.y;
    const int i = blockIdx.x * GG_THREADS + threadIdx.x;
    if (i >= nq) return;
    const float *g1 = grad_out + (((size_t)bi * 2 * c + l) * nq + i) * k;
    const float *g2 = grad_out + (((size_t)bi * 2 * c + c + l) * nq + i) * k;
    float acc = 0.f;
    if constexpr (K4 > 0) {
#pragma unroll
        for (int t = 0; t < K4; ++t) {   // same left-to-right order as the generic loop
            const gf_f4 a = __builtin_nontemporal_load(reinterpret_cast<const gf_f4 *>(g1) + t);
            const gf_f4 d = __builtin_nontemporal_load(reinterpret_cast<const gf_f4 *>(g2) + t);
            acc += d.x - a.x; acc += d.y - a.y; acc += d.z - a.z; acc += d.w - a.w;
        }
    } else {
        for (int j = 0; j < k; ++j) acc += g2[j] - g1[j];
    }
    grad_xq[((size_t)bi * c + l) * nq + i] += acc;
}

// ---- table-in-LDS gathers ------------------------------------------------------------------------
// out[b,c,e] = sum_t w[e,t] * table[b,c,idx[e,t]]  (three_interpolate: NT = 3; group / gather: NT = 1, w = 1).
// In the (B,C,N) layout each lane of a gather hits a different 4-byte word of a
// channel row: the vector-memory address path handles a few lanes per clock, which caps the kernels above
// at 0.4-1 TB/s.  A channel row is small (m floats), so a workgroup keeps `ch` whole rows of the table in
// LDS (up to 144 KB), where 64 random 4-byte reads are one ds_read_b32, and streams its slice of e with
// coalesced index loads and coalesced stores.  (The transposed scatter with LDS float atomics was measured
// too and loses to the channels-last workspace scatter below: 0.20 vs 0.13 ms for the prop0 gradient.)
constexpr int TLDS_THREADS = 1024;
constexpr int TLDS_FLOATS = 36 * 1024; // LDS budget for the table rows (144 KB of the CU's 160 KB)

// global -> LDS copy of `count` contiguous floats by the whole workgroup: 16-byte vectors when the source is
// aligned, four independent loads in flight per thread before the first LDS store (a plain loop makes every
// iteration wait for its own load: 24 serialised latencies for a 96 KB row)
__device__ __forceinline__ void tlds_load_rows(float *__restrict__ dst, const float *__restrict__ src, int count)
{
    const int tid = threadIdx.x;
    if ((((uintptr_t)src) & 15) == 0) {
        const int vec = count >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int e = tid; e < vec; e += 4 * TLDS_THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = e + u * TLDS_THREADS < vec ? s4[e + u * TLDS_THREADS] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u * TLDS_THREADS < vec) d4[e + u * TLDS_THREADS] = v[u];
        }
        for (int e = (vec << 2) + tid; e < count; e += TLDS_THREADS) dst[e] = src[e];
    } else {
        for (int e = tid; e < count; e += 4 * TLDS_THREADS) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = e + u * TLDS_THREADS < count ? src[e + u * TLDS_THREADS] : 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u * TLDS_THREADS < count) dst[e + u * TLDS_THREADS] = v[u];
        }
    }
}

template <int NT, bool WEIGHTED>
__global__ __launch_bounds__(TLDS_THREADS) void table_gather_lds_kernel(
    int c, int m, int L, int ch, const float *__restrict__ table, const int *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ out, size_t out_bstride)
{
    extern __shared__ float tlds_rows[]; // [ch][m]
    const int bi = blockIdx.z, c0 = blockIdx.y * ch, nch = min(ch, c - c0);
    const float *src = table + ((size_t)bi * c + c0) * m; // nch rows, contiguous
    tlds_load_rows(tlds_rows, src, nch * m);
    __syncthreads();
    const int per = (L + gridDim.x - 1) / gridDim.x;
    const int e0 = blockIdx.x * per, e1 = min(L, e0 + per);
    // U elements per thread and pass: U x 2 NT independent index / weight loads in flight (one workgroup per
    // CU -- the rows fill its LDS -- so memory-level parallelism has to come from within the thread)
    constexpr int U = 4;
    for (int eb = e0 + threadIdx.x; eb < e1; eb += U * TLDS_THREADS) {
        int ii[U][NT];
        float w[U][NT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = eb + u * TLDS_THREADS;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                ii[u][t] = e < e1 ? idx[((size_t)bi * L + e) * NT + t] : 0;
                w[u][t] = (WEIGHTED && e < e1) ? weight[((size_t)bi * L + e) * NT + t] : 1.f;
            }
        }
        for (int l = 0; l < nch; ++l) {
            const float *R = tlds_rows + l * m;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = eb + u * TLDS_THREADS;
                float v;
                if (WEIGHTED) {
                    v = R[ii[u][0]] * w[u][0];
#pragma unroll
                    for (int t = 1; t < NT; ++t) v = v + R[ii[u][t]] * w[u][t]; // ((p0*w0 + p1*w1) + p2*w2), un-contracted
                } else {
                    v = R[ii[u][0]];
                }
                if (e < e1) out[(size_t)bi * out_bstride + (size_t)(c0 + l) * L + e] = v;
            }
        }
    }
}

// launch geometry: (slices of e, channel chunks, batch); 0 rows -> the table does not fit, use the plain kernels
struct TldsPlan {
    int ch, slices;
    size_t lds;
};
static TldsPlan tlds_plan(int b, int c, int m, long long L, int min_ch)
{
    TldsPlan p{0, 1, 0};
    const char *env = getenv("GEOT_GATHER_IMPL"); // "plain" = the register-gather kernels (A/B tests)
    if ((env && env[0] == 'p') || m < 1 || m > TLDS_FLOATS || L * c < (1 << 16)) return p;
    int ch = TLDS_FLOATS / m;
    if (ch < min_ch) return p; // long rows: one or two channels per workgroup do not pay for loading them
    if (ch > 16) ch = 16;
    if (ch > c) ch = c;
    const int chunks = (c + ch - 1) / ch;
    // enough workgroups for 256 CUs (one per CU: the rows fill its LDS), but slices of >= 4096 elements so that
    // loading / flushing the rows stays a small part of a workgroup's work; the scatter prefers one owner per row
    long long want = (512 + (long long)chunks * b - 1) / ((long long)chunks * b);
    long long maxs = L / 4096;
    long long sl = want < 1 ? 1 : want;
    if (sl > maxs) sl = maxs < 1 ? 1 : maxs;
    if (sl > 64) sl = 64;
    p.ch = ch;
    p.slices = (int)sl;
    p.lds = (size_t)ch * m * sizeof(float);
    return p;
}

template <typename K>
static hipError_t tlds_set_lds(K kernel, size_t lds)
{
    return allow_big_lds((const void *)kernel, lds);
}

template <int NT, bool WEIGHTED>
static hipError_t tlds_gather(const TldsPlan &p, int b, int c, int m, int L, const float *table, const int *idx,
                              const float *weight, float *out, hipStream_t s, size_t out_bstride = 0)
{
    hipError_t e = tlds_set_lds(table_gather_lds_kernel<NT, WEIGHTED>, p.lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((table_gather_lds_kernel<NT, WEIGHTED>), dim3(p.slices, (c + p.ch - 1) / p.ch, b),
                       dim3(TLDS_THREADS), p.lds, s, c, m, L, p.ch, table, idx, weight, out,
                       out_bstride ? out_bstride : (size_t)c * L);
    return hipGetLastError();
}
// ---- gradients of the gathers as gathers: reverse index + source rows in LDS --------------------
// grad_table[b,c,j] += sum over the (e,t) with idx[b,e,t] == j of w[b,e,t] * grad_out[b,c,e].
// The pairs are grouped by target once per call (count with rank, exclusive scan, fill: three small
// kernels, no atomics in the fill because the count pass already handed out the ranks); then a workgroup
// keeps `ch` rows of grad_out (L floats each) in LDS and every thread sums one target's list from LDS:
// no float atomics at all, and grad_out is read exactly once.  Needs L <= TLDS_FLOATS (the prop0/1/2
// interpolations, the kNN graph features, the 512-group gather; not the 6000 x 32 SA grouping).
// The L sources of a batch are cut into Q parts of `partlen` (Q = 1: one part); pairs are grouped by
// (batch, part, target), so that a workgroup holding the rows of ONE part in LDS finds exactly its entries.
__global__ __launch_bounds__(256) void rix_count_kernel(long long total, long long per_batch, int m, int nt, int Q,
                                                        int partlen, const int *__restrict__ idx,
                                                        int *__restrict__ cnt, int *__restrict__ rank,
                                                        const int *__restrict__ remap = nullptr)
{   // remap (b, m), Q == 1 only: the number under which a target is filed (the point-major walk's target order)
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= total) return;
    const int bi = (int)(x / per_batch);
    const int e = (int)((x - (long long)bi * per_batch) / nt), part = e / partlen;
    const int j = remap ? remap[(size_t)bi * m + idx[x]] : idx[x];
    rank[x] = atomicAdd(&cnt[((size_t)bi * Q + part) * m + j], 1);
}
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void rix_fill_kernel(long long total, long long per_batch, int m, int nt, int Q,
                                                       int partlen, const int *__restrict__ idx,
                                                       const float *__restrict__ weight, const int *__restrict__ off,
                                                       const int *__restrict__ rank, int *__restrict__ rev,
                                                       float *__restrict__ revw, int *__restrict__ tmp,
                                                       const int *__restrict__ remap = nullptr)
{
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= total) return;
    const int bi = (int)(x / per_batch);
    const int e = (int)((x - (long long)bi * per_batch) / nt), part = e / partlen;
    const int j = remap ? remap[(size_t)bi * m + idx[x]] : idx[x];
    const int pos = off[((size_t)bi * Q + part) * m + j] + rank[x];
    if (tmp) { tmp[pos] = (int)x; return; } // reproducible build: pair ids first, placed by rix_place_kernel
    rev[pos] = e - part * partlen; // source element within its part
    if (WEIGHTED) revw[pos] = weight[x];
}
// second phase of the reproducible build: pair x goes to its position in ascending pair-id order within its list
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void rix_place_kernel(long long total, long long per_batch, int m, int nt, int Q,
                                                        int partlen, const int *__restrict__ idx,
                                                        const float *__restrict__ weight, const int *__restrict__ off,
                                                        const int *__restrict__ rank, const int *__restrict__ tmp,
                                                        int *__restrict__ rev, float *__restrict__ revw)
{
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= total) return;
    const int bi = (int)(x / per_batch);
    const int e = (int)((x - (long long)bi * per_batch) / nt), part = e / partlen;
    const size_t tgt = ((size_t)bi * Q + part) * m + idx[x];
    const int a = off[tgt], z = off[tgt + 1];
    const int pos = a + rix_sorted_position(tmp, a, z, (int)x, rank[x]);
    rev[pos] = e - part * partlen;
    if (WEIGHTED) revw[pos] = weight[x];
}

// LPT lanes per target (1, 4, 8 or 16 adjacent lanes): a layer with few targets and long lists (the FP modules that
// interpolate from the 512 group centres: 48 entries per target, half the workgroup idle at one thread per target)
// spreads every list over LPT lanes -- lane `sub` takes entries sub, sub + LPT, ... (consecutive lanes read consecutive
// entries) -- and sums the partial results inside the lane group.  LPT = 1 is the one-thread-per-target walk.
template <bool WEIGHTED, int CH, int LPT>
__global__ __launch_bounds__(TLDS_THREADS) void table_gather_csr_lds_kernel(
    int c, int m, int L, int Q, int partlen, const float *__restrict__ grad_out, size_t src_bstride,
    const int *__restrict__ off, const int *__restrict__ rev, const float *__restrict__ revw,
    float *__restrict__ grad_table, int set)
{
    extern __shared__ float tlds_rows[]; // [CH][plen] this part of CH rows of grad_out
    const int bq = blockIdx.z, bi = bq / Q, part = bq - bi * Q;
    const int c0 = blockIdx.y * CH, nch = min(CH, c - c0);
    const int p0 = part * partlen, plen = min(partlen, L - p0);
    const float *src = grad_out + (size_t)bi * src_bstride + (size_t)c0 * L + p0;
    if (Q == 1) tlds_load_rows(tlds_rows, src, nch * L); // whole rows are contiguous
    else
        for (int l = 0; l < nch; ++l) tlds_load_rows(tlds_rows + l * plen, src + (size_t)l * L, plen);
    __syncthreads();
    const int per = (m + gridDim.x - 1) / gridDim.x;
    const int j0 = blockIdx.x * per, j1 = min(m, j0 + per);
    // TP targets per lane group, their lists walked in lock step RU entries per lane at a time: TP x RU x 2 independent
    // loads in flight per thread (one workgroup per CU, so the memory-level parallelism must come from here)
    constexpr int RU = 4, TP = 4, SLOTS = TLDS_THREADS / LPT;
    const int sub = threadIdx.x & (LPT - 1), slot = threadIdx.x / LPT;
    for (int jb = j0 + slot; jb < j1; jb += TP * SLOTS) {
        int a[TP], z[TP];
        float acc[TP][CH];
#pragma unroll
        for (int p = 0; p < TP; ++p) {
            const int j = jb + p * SLOTS;
            a[p] = j < j1 ? off[(size_t)bq * m + j] : 0;
            z[p] = j < j1 ? off[(size_t)bq * m + j + 1] : 0;
#pragma unroll
            for (int l = 0; l < CH; ++l) acc[p][l] = 0.f;
        }
        int longest = 0; // entries per lane of the longest of the TP lists
#pragma unroll
        for (int p = 0; p < TP; ++p) longest = max(longest, (z[p] - a[p] + LPT - 1) / LPT);
        for (int it = 0; it < longest; it += RU) {
            int e[TP][RU];
            float w[TP][RU];
#pragma unroll
            for (int p = 0; p < TP; ++p)
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const int q = a[p] + (it + u) * LPT + sub;
                    const bool in = q < z[p];
                    e[p][u] = in ? rev[q] : 0;
                    w[p][u] = in ? (WEIGHTED ? revw[q] : 1.f) : 0.f;
                }
#pragma unroll
            for (int p = 0; p < TP; ++p)
#pragma unroll
                for (int l = 0; l < CH; ++l) {
                    if (l < nch) {
#pragma unroll
                        for (int u = 0; u < RU; ++u) acc[p][l] = fmaf(w[p][u], tlds_rows[l * plen + e[p][u]], acc[p][l]);
                    }
                }
        }
        if (LPT > 1) {
#pragma unroll
            for (int p = 0; p < TP; ++p)
#pragma unroll
                for (int l = 0; l < CH; ++l)
#pragma unroll
                    for (int o = LPT / 2; o >= 1; o >>= 1) acc[p][l] += __shfl_xor(acc[p][l], o);
        }
#pragma unroll
        for (int p = 0; p < TP; ++p) {
            const int j = jb + p * SLOTS;
            if (sub == 0 && j < j1 && (set || z[p] > a[p])) { // untouched targets keep what they had (the buffer is accumulated into)
#pragma unroll
                for (int l = 0; l < CH; ++l) {
                    if (l < nch) {
                        float *dst = grad_table + ((size_t)bi * c + c0 + l) * m + j;
                        if (set) *dst = acc[p][l];       // (Q == 1 only) sole writer, nothing to keep
                        else if (Q == 1) *dst += acc[p][l];   // sole writer of this element
                        else atomicAdd(dst, acc[p][l]);  // one contribution per part; lanes = consecutive targets
                    }
                }
            }
        }
    }
}

// The same gather with the Q source parts LOOPED INSIDE the workgroup: a thread owns TPT targets for the whole
// launch and keeps their CH partial sums in registers while the workgroup stages one part of its CH rows after the
// other.  Every output element has exactly one writer and one store: no float atomics, so the gradient is
// bit-reproducible from run to run (the per-part kernel above adds Q partial sums per output with memory-side
// atomics, in whatever order the parts finish), grad_out is read once, and L is unbounded.  Same speed as the atomic
// form for the prop0 gradient (340 vs 354 us at 8 clouds): the walk over the per-target lists is the cost of both.
template <bool WEIGHTED, int CH, int TPT, int TP = 4, int MINB = 1>
__global__ __launch_bounds__(TLDS_THREADS, MINB) void table_gather_csr_parts_kernel(
    int c, int m, int L, int Q, int partlen, const float *__restrict__ grad_out, size_t src_bstride,
    const int *__restrict__ off, const int *__restrict__ rev, const float *__restrict__ revw,
    float *__restrict__ grad_table, int set)
{
    extern __shared__ float tlds_rows[]; // [CH][partlen]
    const int bi = blockIdx.z, c0 = blockIdx.y * CH, nch = min(CH, c - c0);
    const int per = (m + gridDim.x - 1) / gridDim.x;
    const int j0 = blockIdx.x * per, j1 = min(m, j0 + per);
    float acc[TPT][CH];
#pragma unroll
    for (int p = 0; p < TPT; ++p)
#pragma unroll
        for (int l = 0; l < CH; ++l) acc[p][l] = 0.f;
    constexpr int RU = 4;        // RU 2 -> 4: 1229 -> 1181 us at (8, 1536, 24000 -> 8192); 8: 1269 (registers)
    for (int part = 0; part < Q; ++part) {
        const int p0 = part * partlen, plen = min(partlen, L - p0);
        if (part) __syncthreads(); // the previous part's rows have been read by everyone
        for (int l = 0; l < nch; ++l)
            tlds_load_rows(tlds_rows + (size_t)l * partlen, grad_out + (size_t)bi * src_bstride + (size_t)(c0 + l) * L + p0, plen);
        __syncthreads();
        const int *offp = off + ((size_t)bi * Q + part) * m;
#pragma unroll
        for (int g = 0; g < TPT; g += TP) {
            int a[TP], z[TP], longest = 0;
#pragma unroll
            for (int p = 0; p < TP; ++p) {
                const int j = j0 + threadIdx.x + (g + p) * TLDS_THREADS;
                a[p] = j < j1 ? offp[j] : 0;
                z[p] = j < j1 ? offp[j + 1] : 0;
                longest = max(longest, z[p] - a[p]);
            }
            for (int it = 0; it < longest; it += RU) {
                int e[TP][RU];
                float w[TP][RU];
#pragma unroll
                for (int p = 0; p < TP; ++p)
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int q = a[p] + it + u;
                        const bool in = q < z[p];
                        e[p][u] = in ? rev[q] : 0;
                        w[p][u] = in ? (WEIGHTED ? revw[q] : 1.f) : 0.f;
                    }
#pragma unroll
                for (int p = 0; p < TP; ++p)
#pragma unroll
                    for (int l = 0; l < CH; ++l) {
                        if (l < nch) {
#pragma unroll
                            for (int u = 0; u < RU; ++u)
                                acc[g + p][l] = fmaf(w[p][u], tlds_rows[(size_t)l * partlen + e[p][u]], acc[g + p][l]);
                        }
                    }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < TPT; ++p) {
        const int j = j0 + threadIdx.x + p * TLDS_THREADS;
        if (j < j1) {
#pragma unroll
            for (int l = 0; l < CH; ++l)
                if (l < nch) { // sole writer; the buffer is accumulated into unless the caller asked for a plain store
                    float *dst = grad_table + ((size_t)bi * c + c0 + l) * m + j;
                    *dst = set ? acc[p][l] : *dst + acc[p][l];
                }
        }
    }
}

// ---- the same gradient with BALANCED list walks: SELL-64 layout per (batch, part) --------------------------------
// Where the time of the kernel above goes (tools/gg_lab.py knock-outs at 8 x 1536 x 24000 -> 8192, profiles/
// r03_gg_lab_knockouts.txt): staging + stores 295 us, the list walk 575 us -- and the walk is lock-step waste, not
// arithmetic: a (target, part) list has 2.9 entries on average but the longest of the 256 lists a wave walks together
// has ~9, every lane pays for it (~25 % useful slots), each step costs two per-lane index gathers behind a per-lane
// offset load, and four 4-byte LDS reads.  Here the lists of one (batch, part) are SORTED BY LENGTH once per call
// (sell_build_kernel) and stored sliced-ELL: task = 64 lists of (nearly) equal length, entry k of lane l at
// base + 64 k + l.  A wave walks a task with a wave-uniform trip count, ONE coalesced 8-byte load per step (source
// id + weight) and ONE 16-byte LDS read (the CH = 4 channels of a source are interleaved in LDS: rows[e][4]).  The
// sorted order differs from part to part, so the per-target accumulators cannot follow the lists: every lane keeps the
// sums of its 8 lists of this part in registers, and after the walk the staging buffer -- free until the next part
// arrives -- carries them to the threads that own the targets (one 16-byte LDS write + read per target and part).
// RESULT (round 3, MI355X): the walk drops from 558 to 330 us, but the index build grows by 45 us and the exchange costs
// 85 us, and staging (300 us), walk and exchange still run one after the other (one 144-KB workgroup per CU): 897 us
// per call against 922 us.  The kernel is therefore NOT the default (GEOT_GATHER_IMPL=sell selects it).
// Lists longer than SELL_LMAX (never at the model's shapes; a hub target in adversarial input) are walked by a whole
// wave each, before the exchange.  One writer and one fixed summation order per output: bit-reproducible.
constexpr int SELL_LMAX = 32;         // longest list stored in the sliced layout (bounds the padding: < 64 * 32 entries)
constexpr int SELL_MAX_M = 8 * TLDS_THREADS;   // 128 tasks of 64 lists
constexpr int SELL_THREADS = 1024;    // the gather kernel: 16 waves x 8 tasks (512 threads x 16 tasks: 966 vs 781 us)
constexpr int SELL_LONG_CAP = 1024;   // long-list results per part kept in LDS (16 KB): >= partlen * nt / (SELL_LMAX + 1), checked by the caller
struct SellView {
    const uint2 *ent;   // [bq][cap]   (source id within the part, weight bits)
    const int *tgt;     // [bq][mp]    target of sorted position p (mp = m rounded up to 64; -1 = padding)
    const int *tbase;   // [bq][ntask] first entry of a task
    const int *tlen;    // [bq][ntask] entries per lane of a task
    const int *nlong;   // [bq]        lists longer than SELL_LMAX
    const int *longid;  // [bq][m]     their targets
    int cap, mp, ntask;
};

// one workgroup per (batch, part): lengths -> counting sort by length (descending) -> task table -> entries
template <bool WEIGHTED>
__global__ __launch_bounds__(TLDS_THREADS) void sell_build_kernel(int m, const int *__restrict__ off,
                                                                  const int *__restrict__ rev, const float *__restrict__ revw,
                                                                  uint2 *__restrict__ ent, int *__restrict__ tgt,
                                                                  int *__restrict__ tbase, int *__restrict__ tlen,
                                                                  int *__restrict__ nlong, int *__restrict__ longid, int cap,
                                                                  int mp, int ntask)
{
    __shared__ int hist[SELL_LMAX + 1], start[SELL_LMAX + 1], nl;
    __shared__ int tl[SELL_MAX_M / 64], tb[SELL_MAX_M / 64 + 1];
    __shared__ unsigned char cls_sorted[SELL_MAX_M];
    const int bq = blockIdx.x, tid = threadIdx.x;
    const int *offp = off + (size_t)bq * m;
    if (tid <= SELL_LMAX) hist[tid] = 0;
    if (tid == 0) nl = 0;
    for (int p = tid; p < mp; p += TLDS_THREADS) cls_sorted[p] = 0;
    __syncthreads();
    constexpr int TPT = SELL_MAX_M / TLDS_THREADS;
    int a[TPT], len[TPT], cls[TPT], rk[TPT];
#pragma unroll
    for (int s = 0; s < TPT; ++s) {
        const int j = tid + s * TLDS_THREADS;
        a[s] = j < m ? offp[j] : 0;
        len[s] = j < m ? offp[j + 1] - a[s] : 0;
        cls[s] = len[s] > SELL_LMAX ? 0 : len[s];
        rk[s] = -1;
        if (j < m) {
            rk[s] = atomicAdd(&hist[cls[s]], 1);                    // rank inside its length class (any order will do)
            if (len[s] > SELL_LMAX) longid[(size_t)bq * m + atomicAdd(&nl, 1)] = j;
        }
    }
    __syncthreads();
    if (tid == 0) {                                                 // longest class first
        int acc = 0;
        for (int c = SELL_LMAX; c >= 0; --c) { start[c] = acc; acc += hist[c]; }
        nlong[bq] = nl;
    }
    __syncthreads();
    int pos[TPT];
#pragma unroll
    for (int s = 0; s < TPT; ++s) {
        const int j = tid + s * TLDS_THREADS;
        pos[s] = j < m ? start[cls[s]] + rk[s] : -1;
        if (j < m) {
            tgt[(size_t)bq * mp + pos[s]] = j;
            cls_sorted[pos[s]] = (unsigned char)cls[s];
        }
    }
    for (int p = m + tid; p < mp; p += TLDS_THREADS) tgt[(size_t)bq * mp + p] = -1;    // padding lanes of the last task
    __syncthreads();
    if (tid < ntask) tl[tid] = cls_sorted[tid * 64];                // sorted descending: the first list of a task is its longest
    __syncthreads();
    if (tid == 0) {     // entry bases in WAVE-MAJOR order: the tasks of gather wave w (w, w + W, w + 2W, ...) are contiguous
        constexpr int W = SELL_THREADS / 64;
        int acc = 0;
        for (int w = 0; w < W; ++w)
            for (int t = w; t < ntask; t += W) { tb[t] = acc; acc += 64 * tl[t]; }
        tb[ntask] = acc;
    }
    __syncthreads();
    if (tid < ntask) {
        tbase[(size_t)bq * ntask + tid] = tb[tid];
        tlen[(size_t)bq * ntask + tid] = tb[ntask] <= cap ? tl[tid] : 0;   // (cannot happen: the padding is < 64 * SELL_LMAX)
    }
    if (tb[ntask] > cap) return;
    uint2 *dst = ent + (size_t)bq * cap;
#pragma unroll
    for (int s = 0; s < TPT; ++s) {
        if (pos[s] < 0) continue;
        const int task = pos[s] >> 6, lane = pos[s] & 63, L = tl[task];
        for (int k = 0; k < L; ++k) {
            const bool in = k < cls[s];
            dst[tb[task] + k * 64 + lane] = make_uint2(in ? (unsigned)rev[a[s] + k] : 0u,
                                                       in ? __float_as_uint(WEIGHTED ? revw[a[s] + k] : 1.f) : 0u);
        }
    }
}

template <bool WEIGHTED>
__global__ __launch_bounds__(SELL_THREADS) void table_gather_sell_kernel(
    int c, int m, int L, int Q, int partlen, const float *__restrict__ grad_out, size_t src_bstride,
    const int *__restrict__ off, const int *__restrict__ rev, const float *__restrict__ revw, SellView sv,
    float *__restrict__ grad_table, int set)
{
    constexpr int CH = 4, TPT = SELL_MAX_M / SELL_THREADS, WAVES = SELL_THREADS / 64;
    extern __shared__ float tlds_rows[];                 // [partlen][CH] staged sources; afterwards [mp][CH] list sums
    __shared__ float longres[SELL_LONG_CAP][CH];
    float4 *rows4 = reinterpret_cast<float4 *>(tlds_rows);
    const int bi = blockIdx.z, c0 = blockIdx.y * CH, nch = min(CH, c - c0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float acc[TPT][CH];
#pragma unroll
    for (int p = 0; p < TPT; ++p)
#pragma unroll
        for (int l = 0; l < CH; ++l) acc[p][l] = 0.f;
    for (int part = 0; part < Q; ++part) {
        const int bq = bi * Q + part, p0 = part * partlen, plen = min(partlen, L - p0);
        if (part) __syncthreads();                       // the exchange of the previous part has been read
        // stage CH rows of this part interleaved: rows4[e] = (g[c0][e], g[c0+1][e], g[c0+2][e], g[c0+3][e])
        {
            const float *g0 = grad_out + (size_t)bi * src_bstride + (size_t)c0 * L + p0;
            if (nch == CH && (L & 3) == 0 && (p0 & 3) == 0 && (((uintptr_t)g0) & 15) == 0) {
                // 16-byte loads from each of the four rows, a 4 x 4 transpose in registers, 16-byte LDS stores
                const int quads = plen >> 2;
                for (int q4 = tid; q4 < quads; q4 += SELL_THREADS) {
                    const float4 a0 = reinterpret_cast<const float4 *>(g0)[q4];
                    const float4 a1 = reinterpret_cast<const float4 *>(g0 + (size_t)L)[q4];
                    const float4 a2 = reinterpret_cast<const float4 *>(g0 + 2 * (size_t)L)[q4];
                    const float4 a3 = reinterpret_cast<const float4 *>(g0 + 3 * (size_t)L)[q4];
                    rows4[4 * q4 + 0] = make_float4(a0.x, a1.x, a2.x, a3.x);
                    rows4[4 * q4 + 1] = make_float4(a0.y, a1.y, a2.y, a3.y);
                    rows4[4 * q4 + 2] = make_float4(a0.z, a1.z, a2.z, a3.z);
                    rows4[4 * q4 + 3] = make_float4(a0.w, a1.w, a2.w, a3.w);
                }
                for (int e = (quads << 2) + tid; e < plen; e += SELL_THREADS)
                    rows4[e] = make_float4(g0[e], g0[(size_t)L + e], g0[2 * (size_t)L + e], g0[3 * (size_t)L + e]);
            } else {
                for (int e = tid; e < plen; e += SELL_THREADS) {
                    float4 v;
                    v.x = g0[e];
                    v.y = nch > 1 ? g0[(size_t)L + e] : 0.f;
                    v.z = nch > 2 ? g0[2 * (size_t)L + e] : 0.f;
                    v.w = nch > 3 ? g0[3 * (size_t)L + e] : 0.f;
                    rows4[e] = v;
                }
            }
        }
        __syncthreads();
        // Walk this wave's TPT tasks (task = wave + 16 s: the sorted order puts the long tasks first, so every wave gets a
        // mix): wave-uniform trip counts, one coalesced 8-byte load and one 16-byte LDS read per entry.
        // (Measured alternatives, profiles/r03_gg_lab_sell.txt: all tasks advanced together one entry per round, and one
        // flattened stream per wave with 8 loads in flight -- both slower: 24-50 spilled registers under the 128-register
        // budget of a 1024-thread workgroup, 512 threads x 16 tasks leave too few waves to hide the round trips.)
        float res[TPT][CH];
        const uint2 *ent = sv.ent + (size_t)bq * sv.cap;
#pragma unroll
        for (int s = 0; s < TPT; ++s) {
            const int task = wave + s * WAVES;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
            if (task < sv.ntask) {
                const int base = sv.tbase[(size_t)bq * sv.ntask + task];
                int len = sv.tlen[(size_t)bq * sv.ntask + task];
                const uint2 *q = ent + base + lane;
                int k = 0;
                for (; k + 4 <= len; k += 4) {           // 4 independent 8-byte loads + 4 LDS reads in flight
                    const uint2 e0 = q[(k + 0) * 64], e1 = q[(k + 1) * 64], e2 = q[(k + 2) * 64], e3 = q[(k + 3) * 64];
                    const float4 v0 = rows4[e0.x], v1 = rows4[e1.x], v2 = rows4[e2.x], v3 = rows4[e3.x];
                    float w = __uint_as_float(e0.y);
                    r0 = fmaf(w, v0.x, r0); r1 = fmaf(w, v0.y, r1); r2 = fmaf(w, v0.z, r2); r3 = fmaf(w, v0.w, r3);
                    w = __uint_as_float(e1.y);
                    r0 = fmaf(w, v1.x, r0); r1 = fmaf(w, v1.y, r1); r2 = fmaf(w, v1.z, r2); r3 = fmaf(w, v1.w, r3);
                    w = __uint_as_float(e2.y);
                    r0 = fmaf(w, v2.x, r0); r1 = fmaf(w, v2.y, r1); r2 = fmaf(w, v2.z, r2); r3 = fmaf(w, v2.w, r3);
                    w = __uint_as_float(e3.y);
                    r0 = fmaf(w, v3.x, r0); r1 = fmaf(w, v3.y, r1); r2 = fmaf(w, v3.z, r2); r3 = fmaf(w, v3.w, r3);
                }
                for (; k < len; ++k) {
                    const uint2 e0 = q[k * 64];
                    const float4 v0 = rows4[e0.x];
                    const float w = __uint_as_float(e0.y);
                    r0 = fmaf(w, v0.x, r0); r1 = fmaf(w, v0.y, r1); r2 = fmaf(w, v0.z, r2); r3 = fmaf(w, v0.w, r3);
                }
            }
            res[s][0] = r0; res[s][1] = r1; res[s][2] = r2; res[s][3] = r3;
        }
        // hub targets (more than SELL_LMAX sources in this part): one wave per list, straight from the CSR arrays
        const int nlong = sv.nlong[bq];
        for (int li = wave; li < nlong; li += WAVES) {
            const int j = sv.longid[(size_t)bq * m + li];
            const int a = off[(size_t)bq * m + j], z = off[(size_t)bq * m + j + 1];
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
            for (int q = a + lane; q < z; q += 64) {
                const float4 v = rows4[rev[q]];
                const float w = WEIGHTED ? revw[q] : 1.f;
                r0 = fmaf(w, v.x, r0); r1 = fmaf(w, v.y, r1); r2 = fmaf(w, v.z, r2); r3 = fmaf(w, v.w, r3);
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                r0 += __shfl_xor(r0, o); r1 += __shfl_xor(r1, o); r2 += __shfl_xor(r2, o); r3 += __shfl_xor(r3, o);
            }
            if (lane == 0 && li < SELL_LONG_CAP) {       // (li < cap always: the caller bounds partlen * nt)
                longres[li][0] = r0; longres[li][1] = r1; longres[li][2] = r2; longres[li][3] = r3;
            }
        }
        __syncthreads();                                 // every wave is done reading the staged rows
        // the exchange: sum of the list of target j -> rows4[j]
#pragma unroll
        for (int s = 0; s < TPT; ++s) {
            const int task = wave + s * WAVES;
            if (task < sv.ntask) {
                const int j = sv.tgt[(size_t)bq * sv.mp + task * 64 + lane];
                if (j >= 0) rows4[j] = make_float4(res[s][0], res[s][1], res[s][2], res[s][3]);
            }
        }
        __syncthreads();
        for (int li = tid; li < min(nlong, SELL_LONG_CAP); li += SELL_THREADS) {
            const int j = sv.longid[(size_t)bq * m + li];                    // its sliced list is empty: rows4[j] holds 0
            rows4[j] = make_float4(longres[li][0], longres[li][1], longres[li][2], longres[li][3]);
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < TPT; ++p) {
            const int j = tid + p * SELL_THREADS;
            if (j < m) {
                const float4 v = rows4[j];
                acc[p][0] += v.x; acc[p][1] += v.y; acc[p][2] += v.z; acc[p][3] += v.w;
            }
        }
    }
    // exactly one writer per output element
#pragma unroll
    for (int p = 0; p < TPT; ++p) {
        const int j = tid + p * SELL_THREADS;
        if (j < m) {
#pragma unroll
            for (int l = 0; l < CH; ++l)
                if (l < nch) {
                    float *dst = grad_table + ((size_t)bi * c + c0 + l) * m + j;
                    *dst = set ? acc[p][l] : *dst + acc[p][l];
                }
        }
    }
}

// ints needed in the workspace for the reverse index of (b, L, nt) pairs onto m targets per batch
// Parts: with whole rows in LDS only TLDS_FLOATS / L channels share a workgroup, and each of them re-reads the
// whole index (8 bytes per pair): at L = 24 000 that is one channel per workgroup and 4.4x more index than
// payload traffic.  Cutting the source range into Q parts lets >= 4 channels share one part's entries.
struct RixPlan {
    int Q, partlen, ch;
};
static RixPlan rix_plan(int c, long long L)
{
    RixPlan p{1, (int)L, 1};
    int ch = (int)(TLDS_FLOATS / L);
    if (ch < 4 && c >= 4) {
        p.Q = (int)((4 * L + TLDS_FLOATS - 1) / TLDS_FLOATS);
        p.partlen = (int)(((L + p.Q - 1) / p.Q + 3) & ~3LL); // 16-byte aligned parts
        p.Q = (int)((L + p.partlen - 1) / p.partlen);
        ch = TLDS_FLOATS / p.partlen;
    }
    p.ch = ch >= 8 ? 8 : (ch >= 4 ? 4 : (ch >= 2 ? 2 : 1)); // template instantiations
    while (p.ch > 1 && p.ch > c) p.ch >>= 1;
    return p;
}
static inline long long rix_ws_ints(int b, int c, int m, long long L, int nt)
{
    const long long t = (long long)b * rix_plan(c, L).Q * m, pairs = (long long)b * L * nt;
    return (t + 1) + scan_blocks(t) + 4 * pairs + 8;   // counts, scan scratch, rank, rev, revw, pair ids
}
// extra ints for the sliced (SELL) copy of the index: entries (8 bytes each, padded per (batch, part)), sorted targets,
// task table, hub lists
struct SellPlan {
    int cap, mp, ntask;
    long long ints;
};
static inline SellPlan sell_plan(int b, int Q, int m, int partlen, int nt)
{
    SellPlan p;
    p.cap = partlen * nt + 64 * SELL_LMAX;
    p.mp = (m + 63) & ~63;
    p.ntask = p.mp / 64;
    const long long bq = (long long)b * Q;
    p.ints = bq * (2LL * p.cap + p.mp + 2LL * p.ntask + 1 + m) + 16;
    return p;
}

static bool csr_applies(int b, int c, int m, long long L, int nt, long long ws_floats)
{
    const char *env = getenv("GEOT_GATHER_IMPL");
    if ((env && env[0] == 'p') || L < 1 || L > TLDS_FLOATS || L * c < (1 << 16)) return false;
    return ws_floats >= rix_ws_ints(b, c, m, L, nt) && (long long)b * L * nt <= 0x7fffffffLL &&
           (long long)b * rix_plan(c, L).Q * m <= 0x7ffffff0LL;
}

// Few targets with long lists (the FP modules that interpolate from the 512 group centres: 24-48 pairs per target):
// whole rows of grad_out fit LDS (Q = 1), one lane group per target walks its list -- one writer per element, fixed order,
// and 2x faster there than the sorted pair stream of tile_scatter.hip, which pays for a segmented sum per chunk when most
// pairs of a tile share their target (profiles/r05_tile_scatter.txt).  The pair stream takes everything else.
static bool csr_preferred(int b, int c, int m, long long L, int nt)
{
    const char *env = getenv("GEOT_GATHER_IMPL");
    if (env && env[0]) return env[0] != 't' && env[0] != 'p';       // forced: tiles / plain -> no; csr / sell / atomic -> as before
    return L >= 1 && m >= 1 && rix_plan(c, L).Q == 1 && (double)L * nt >= 12.0 * m &&
           csr_applies(b, c, m, L, nt, rix_ws_ints(b, c, m, L, nt));
}

// returns hipErrorNotSupported when this path does not apply (caller falls back)
template <int NT, bool WEIGHTED>
static hipError_t scatter_via_csr(int b, int c, int m, int L, size_t src_bstride, const float *grad_out,
                                  const int *idx, const float *weight, float *grad_table, float *workspace,
                                  long long ws_floats, hipStream_t s, bool overwrite = false)
{
    // overwrite: grad_table arrives uninitialised.  The one-writer-per-element forms store instead of adding (no
    // zero-fill, no read of the old value); every other form gets the buffer cleared first.
    if (!csr_applies(b, c, m, L, NT, ws_floats)) return hipErrorNotSupported;
    const RixPlan rp = rix_plan(c, L);
    const int ch = rp.ch, Q = rp.Q;
    const long long t = (long long)b * Q * m, pairs = (long long)b * L * NT;
    int *off = (int *)workspace;
    int *bsum = off + t + 1;
    int *rank = bsum + scan_blocks(t);
    int *rev = rank + pairs;
    float *revw = (float *)(rev + pairs);
    hipError_t e = zero_words(off, t + 1, s);
    if (e != hipSuccess) return e;
    const int pb = (int)((pairs + 255) / 256);
    hipLaunchKernelGGL(rix_count_kernel, dim3(pb), dim3(256), 0, s, pairs, (long long)L * NT, m, NT, Q, rp.partlen, idx,
                       off, rank);
    exclusive_scan_i32((int)t, off, bsum, nullptr, s);
    int *tmp = rix_reproducible() ? (int *)(revw + pairs) : nullptr;
    hipLaunchKernelGGL((rix_fill_kernel<WEIGHTED>), dim3(pb), dim3(256), 0, s, pairs, (long long)L * NT, m, NT, Q,
                       rp.partlen, idx, weight, off, rank, rev, revw, tmp);
    if (tmp)   // fixed list order = fixed summation order: reproducible gradients
        hipLaunchKernelGGL((rix_place_kernel<WEIGHTED>), dim3(pb), dim3(256), 0, s, pairs, (long long)L * NT, m, NT, Q,
                           rp.partlen, idx, weight, off, rank, tmp, rev, revw);
    const size_t lds = (size_t)ch * rp.partlen * sizeof(float);
    const int chunks = (c + ch - 1) / ch;
    const char *impl = getenv("GEOT_GATHER_IMPL");           // "atomic": the per-part kernel + float atomics (A/B runs)
    const bool parts_inside = Q > 1 && m <= 16 * TLDS_THREADS && b <= 65535 && !(impl && impl[0] == 'a');
    const int set = overwrite && (parts_inside || Q == 1) ? 1 : 0;
    if (overwrite && !set) {
        e = zero_words(grad_table, (long long)b * c * m, s);
        if (e != hipSuccess) return e;
    }
    if (parts_inside && ch == 4 && m <= SELL_MAX_M && impl && impl[0] == 's' &&
        (long long)rp.partlen * NT / (SELL_LMAX + 1) <= SELL_LONG_CAP) {
        // GEOT_GATHER_IMPL=sell: balanced walk over a sliced copy of the index (table_gather_sell_kernel), when the
        // workspace has room for it.  Opt-in: measured 897 us against 922 us for the per-target walk below at the model's
        // largest shape and 314 against 292 us at C = 384 (profiles/r03_gg_lab_sell.txt) -- not a win worth a second
        // index format; kept because it isolates what the walk costs once its lock-step waste is gone.
        const SellPlan sp = sell_plan(b, Q, m, rp.partlen, NT);
        const long long used = rix_ws_ints(b, c, m, L, NT);
        const size_t lds_sell = (size_t)(rp.partlen > sp.mp ? rp.partlen : sp.mp) * 4 * sizeof(float);
        if (ws_floats >= used + sp.ints && lds_sell + SELL_LONG_CAP * 4 * sizeof(float) + 64 <= 160 * 1024) {
            int *base = (int *)workspace + ((used + 1) & ~1LL);          // 8-byte aligned entries
            uint2 *ent = (uint2 *)base;
            int *tgt = base + 2LL * b * Q * sp.cap;
            int *tbase = tgt + (long long)b * Q * sp.mp;
            int *tlen = tbase + (long long)b * Q * sp.ntask;
            int *nlong = tlen + (long long)b * Q * sp.ntask;
            int *longid = nlong + (long long)b * Q;
            hipLaunchKernelGGL((sell_build_kernel<WEIGHTED>), dim3(b * Q), dim3(TLDS_THREADS), 0, s, m, off, rev, revw, ent, tgt,
                               tbase, tlen, nlong, longid, sp.cap, sp.mp, sp.ntask);
            e = tlds_set_lds(table_gather_sell_kernel<WEIGHTED>, lds_sell);
            if (e != hipSuccess) return e;
            const SellView sv{ent, tgt, tbase, tlen, nlong, longid, sp.cap, sp.mp, sp.ntask};
            hipLaunchKernelGGL((table_gather_sell_kernel<WEIGHTED>), dim3(1, chunks, b), dim3(SELL_THREADS), lds_sell, s, c, m, L,
                               Q, rp.partlen, grad_out, src_bstride, off, rev, revw, sv, grad_table, set);
            return hipGetLastError();
        }
    }
    if (parts_inside) {
        // parts looped inside the workgroup: one writer per output, no atomics, reproducible
        const int tpt = m <= 8 * TLDS_THREADS ? 8 : 16;
        const char *two = getenv("GEOT_GATHER_CH2");
        if (two && two[0] == '1' && ch == 4 && tpt == 8) {
            // LAB (measured, slower: 1177 vs 942 us, profiles/r03_gg_lab_sell.txt D): two channels per workgroup at the same
            // part length -> 64 KB of LDS, 64 registers: TWO workgroups per CU, one staging while the other walks; twice
            // the index walks cost more than the overlap buys
            const size_t lds2 = (size_t)2 * rp.partlen * sizeof(float);
            e = tlds_set_lds(table_gather_csr_parts_kernel<WEIGHTED, 2, 8, 2, 8>, lds2);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((table_gather_csr_parts_kernel<WEIGHTED, 2, 8, 2, 8>), dim3(1, (c + 1) / 2, b), dim3(TLDS_THREADS),
                               lds2, s, c, m, L, Q, rp.partlen, grad_out, src_bstride, off, rev, revw, grad_table, set);
            return hipGetLastError();
        }
        const dim3 grid(1, chunks, b);
#define GEOT_PARTS_LAUNCH(CHV, TPTV)                                                                                \
    {                                                                                                               \
        e = tlds_set_lds(table_gather_csr_parts_kernel<WEIGHTED, CHV, TPTV>, lds);                                  \
        if (e != hipSuccess) return e;                                                                              \
        hipLaunchKernelGGL((table_gather_csr_parts_kernel<WEIGHTED, CHV, TPTV>), grid, dim3(TLDS_THREADS), lds, s, c, m, \
                           L, Q, rp.partlen, grad_out, src_bstride, off, rev, revw, grad_table, set);               \
    }
        if (tpt == 8) {
            if (ch == 8) GEOT_PARTS_LAUNCH(8, 8) else if (ch == 4) GEOT_PARTS_LAUNCH(4, 8) else if (ch == 2) GEOT_PARTS_LAUNCH(2, 8) else GEOT_PARTS_LAUNCH(1, 8)
        } else {
            if (ch == 8) GEOT_PARTS_LAUNCH(8, 16) else if (ch == 4) GEOT_PARTS_LAUNCH(4, 16) else if (ch == 2) GEOT_PARTS_LAUNCH(2, 16) else GEOT_PARTS_LAUNCH(1, 16)
        }
#undef GEOT_PARTS_LAUNCH
        return hipGetLastError();
    }
    long long slices = (512 + (long long)chunks * b * Q - 1) / ((long long)chunks * b * Q);
    if (slices > m / 1024) slices = m / 1024;
    if (slices < 1) slices = 1;
    if ((long long)b * Q > 65535) return hipErrorNotSupported;
    const dim3 grid((int)slices, chunks, b * Q);
#define GEOT_CSR_LAUNCH2(CHV, LPTV)                                                                               \
    {                                                                                                            \
        e = tlds_set_lds(table_gather_csr_lds_kernel<WEIGHTED, CHV, LPTV>, lds);                                 \
        if (e != hipSuccess) return e;                                                                           \
        hipLaunchKernelGGL((table_gather_csr_lds_kernel<WEIGHTED, CHV, LPTV>), grid, dim3(TLDS_THREADS), lds, s, c, m, L, \
                           Q, rp.partlen, grad_out, src_bstride, off, rev, revw, grad_table, set);               \
    }
#define GEOT_CSR_LAUNCH(CHV)                                                                                     \
    {                                                                                                            \
        if (lpt == 16) GEOT_CSR_LAUNCH2(CHV, 16) else if (lpt == 8) GEOT_CSR_LAUNCH2(CHV, 8)                      \
        else if (lpt == 4) GEOT_CSR_LAUNCH2(CHV, 4) else GEOT_CSR_LAUNCH2(CHV, 1)                                 \
    }
    // lanes per target from the mean list length (entries per target and part)
    const double mean_len = (double)L * NT / ((double)Q * m);
    const int lpt = mean_len >= 32.0 ? 16 : (mean_len >= 16.0 ? 8 : (mean_len >= 8.0 ? 4 : 1));
    if (ch == 8) GEOT_CSR_LAUNCH(8) else if (ch == 4) GEOT_CSR_LAUNCH(4) else if (ch == 2) GEOT_CSR_LAUNCH(2) else GEOT_CSR_LAUNCH(1)
#undef GEOT_CSR_LAUNCH2
#undef GEOT_CSR_LAUNCH
    return hipGetLastError();
}

// ---- gradient of a POINT-MAJOR gather: whole rows gathered through the reverse index --------------------------------
// out[b, j, :] = sum over the pairs (e, t) with idx[b, e, t] == j of w[b, e, t] * g[b, e, :]   (g, out: (B, L, C), (B, m, C)).
// The channels-first kernels above walk the lists once per 4 channels (rev + revw = 8 B per pair against 16 B of
// payload); here the lanes of a workgroup are the channels: a list entry is wave-uniform and read ONCE, every source is
// one contiguous 4 C-byte row.  The index is built for this walk (geot_rix_build): targets are numbered in the order
// the kernel takes them (`order`: a spatial order keeps the sources of neighbouring targets in the XCD's L2), the
// pairs of all targets form ONE flat stream (source row, weight, target | last-of-list flag), and a workgroup takes an
// equal share of the STREAM, snapped to list boundaries -- lists of 0 to 30 pairs cost what they hold.  A run of the
// stream is staged in LDS, then walked with the loads of the next 4 pairs in flight under the arithmetic of the current
// 4.  One writer per output row, pairs in ascending pair order: bit-reproducible.  (csrc/channels_last.hip: the forward.)
typedef float cl_f4 __attribute__((ext_vector_type(4)));
constexpr int GR_CAP = 512;          // pairs staged per pass
constexpr int GR_U = 8;   // row loads in flight per half of the software pipeline
constexpr unsigned GR_LAST = 0x80000000u;

// placement for the point-major walk: pair x of target rank k goes to its slot (ascending pair id within the list) as
// (global source row, weight, k | last flag)
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void rix_place_cl_kernel(long long total, long long per_batch, int m, int nt, int L,
                                                           const int *__restrict__ idx, const float *__restrict__ weight,
                                                           const int *__restrict__ rank_of, const int *__restrict__ off,
                                                           const int *__restrict__ rank, const int *__restrict__ tmp,
                                                           int *__restrict__ rev, float *__restrict__ revw,
                                                           unsigned *__restrict__ rtgt)
{
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= total) return;
    const int bi = (int)(x / per_batch);
    const int e = (int)((x - (long long)bi * per_batch) / nt);
    const int j = idx[x];
    const size_t k = (size_t)bi * m + (rank_of ? rank_of[(size_t)bi * m + j] : j);
    const int a = off[k], z = off[k + 1];
    const int pos = a + rix_sorted_position(tmp, a, z, (int)x, rank[x]);
    rev[pos] = bi * L + e;
    revw[pos] = WEIGHTED ? weight[x] : 1.f;
    rtgt[pos] = (unsigned)k | (pos + 1 == z ? GR_LAST : 0u);
}
// rank_of[b, order[b, r]] = r
__global__ __launch_bounds__(256) void invert_order_kernel(long long total, int m, const int *__restrict__ order,
                                                           int *__restrict__ rank_of)
{
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= total) return;
    const long long bi = x / m;
    rank_of[bi * m + order[x]] = (int)(x - bi * m);
}

// ---- which pairs a workgroup of the row gathers walks ---------------------------------------------------------------------
// A source row is one of the three neighbours of THREE targets, and those targets are neighbours in the Morton order the
// lists are stored in.  With one contiguous share of the pair stream per workgroup (round 3, first form) the three uses of
// a row were three trips to memory: the uses sit tens of pairs apart, a CU's share of its XCD's 4-MB L2 is ~128 KB (ten
// pairs of 6-KB rows), and neighbouring shares ran on other XCDs -- rocprofv3 FETCH_SIZE 3.0x the algorithmic bytes.  So
// the targets are dealt like the rows of the forward (channels_last.hip): XCD x (blocks x, x + 8, ...) takes the x-th
// eighth of the target sequence, and inside it workgroup w of the XCD's nwg takes targets w, w + nwg, w + 2 nwg, ...: at
// any moment the XCD's workgroups walk ONE band of nwg consecutive lists, whose pair ids ascend the same way, so the
// second and third use of a row arrive while the first is in flight or still in that XCD's L2.  A list stays whole and
// keeps its order: results are bit-identical to the contiguous form.  List lengths vary (0 ... 30), but a workgroup's
// share is every nwg-th list of thousands: the sums differ by a few per cent.
// GR_SB target slots at a time: lengths -> LDS, one-wave scan, then chunks of GR_CAP pairs are staged by binary search.
constexpr bool GR_DEAL = true;
constexpr int GR_SB = 512;
constexpr int GR_GT = 2;   // consecutive targets a workgroup takes at a time
struct GrDeal {
    int tb, te, nwg, w, nloc;       // the XCD's target range, workgroups sharing it, this one's rank, its target slots
    __device__ __forceinline__ int target(int u) const { return tb + ((u / GR_GT) * nwg + w) * GR_GT + u % GR_GT; }
};
__device__ __forceinline__ GrDeal gr_deal(int T)
{
    const int nx = (gridDim.x % 8 == 0) ? 8 : 1;
    const int x = blockIdx.x % nx;
    GrDeal d;
    d.w = blockIdx.x / nx;
    d.nwg = gridDim.x / nx;
    d.tb = (int)((long long)T * x / nx);
    d.te = (int)((long long)T * (x + 1) / nx);
    const int granules = (d.te - d.tb + GR_GT - 1) / GR_GT;
    d.nloc = granules > d.w ? (granules - d.w + d.nwg - 1) / d.nwg * GR_GT : 0;     // (slots past the range's end are empty lists)
    return d;
}
// lengths and first pair of the slots [ub, ub + nb) of this workgroup; s_pre[u] = pairs in front of slot u (s_pre[nb] = all)
__device__ __forceinline__ void gr_scan_slots(const GrDeal &d, int ub, int nb, const int *__restrict__ off, int *s_pre, int *s_off)
{
    __syncthreads();                // the previous block's staging has read s_pre / s_off
    for (int u = threadIdx.x; u < nb; u += blockDim.x) {
        const int k = d.target(ub + u);
        const int a = k < d.te ? off[k] : 0;
        s_off[u] = a;
        s_pre[u + 1] = k < d.te ? off[k + 1] - a : 0;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int carry = 0;
        for (int piece = 0; piece < nb; piece += 64) {
            const int i = piece + lane;
            int v = i < nb ? s_pre[i + 1] : 0;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const int o = __shfl_up(v, sh);
                if (lane >= sh) v += o;
            }
            if (i < nb) s_pre[i + 1] = v + carry;
            carry += __shfl(v, 63);
        }
        if (lane == 0) s_pre[0] = 0;
    }
    __syncthreads();
}
// pairs [base, base + cnt) of the block's concatenated lists -> source row, weight, output row | last-of-list flag
__device__ __forceinline__ void gr_stage_pairs(const GrDeal &d, int ub, int nb, int base, int cnt, const int *s_pre, const int *s_off,
                                               const int *__restrict__ rev, const float *__restrict__ revw,
                                               const int *__restrict__ order, int m, int *s_src, float *s_w, unsigned *s_tgt)
{
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        const int v = base + i;
        int lo = 0, hi = nb;        // s_pre[lo] <= v < s_pre[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_pre[mid] <= v) lo = mid;
            else hi = mid;
        }
        const int pair = s_off[lo] + (v - s_pre[lo]);
        const int k = d.target(ub + lo);
        s_src[i] = rev[pair];
        s_w[i] = revw[pair];
        s_tgt[i] = (unsigned)(order ? (k / m) * m + order[k] : k) | (v + 1 == s_pre[lo + 1] ? GR_LAST : 0u);
    }
}

__global__ __launch_bounds__(1024) void gather_rows_csr_cl_kernel(int c4, int T, int P, const cl_f4 *__restrict__ g,
                                                                  const int *__restrict__ off, const int *__restrict__ rev,
                                                                  const float *__restrict__ revw,
                                                                  const unsigned *__restrict__ rtgt, const int *__restrict__ order,
                                                                  int m, cl_f4 *__restrict__ out)
{
    __shared__ int s_src[GR_CAP + 3 * GR_U];            // + the pairs the software pipeline loads past the end
    __shared__ float s_w[GR_CAP];
    __shared__ unsigned s_tgt[GR_CAP];       // output row | last flag
    __shared__ int s_empty[GR_CAP];
    __shared__ int s_nempty;
    // gridDim.y channel slabs: this workgroup's lanes are float4 columns [blockIdx.y * c4s, + c4s) of the rows
    const int c4s = c4 / (int)gridDim.y;
    const bool on = (int)threadIdx.x < c4s;
    const int q = (int)blockIdx.y * c4s + (on ? (int)threadIdx.x : c4s - 1);
    const cl_f4 zero = {0.f, 0.f, 0.f, 0.f};
    // rows of targets without pairs: this workgroup's share of the target range
    {
        const int per = (T + gridDim.x - 1) / gridDim.x;
        const int k0 = min(T, (int)blockIdx.x * per), k1 = min(T, k0 + per);
        for (int kb = k0; kb < k1; kb += GR_CAP) {
            if (threadIdx.x == 0) s_nempty = 0;
            __syncthreads();
            for (int k = kb + threadIdx.x; k < min(k1, kb + GR_CAP); k += blockDim.x)
                if (off[k] == off[k + 1]) s_empty[atomicAdd(&s_nempty, 1)] = order ? (k / m) * m + order[k] : k;
            __syncthreads();
            const int ne = s_nempty;
            for (int i = 0; i < ne; ++i)
                if (on) out[(size_t)s_empty[i] * c4 + q] = zero;
            __syncthreads();
        }
    }
    __shared__ int s_pre[GR_SB + 1], s_off[GR_SB];
    // contiguous form (lab): the lists that START in this workgroup's share [s0, s1) of the pair stream
    auto snap = [&](long long s) -> int {
        if (s <= 0) return 0;
        if (s >= P) return P;
        const int k = (int)(rtgt[s] & ~GR_LAST);
        return off[k] == (int)s ? (int)s : off[k + 1];
    };
    const GrDeal deal = gr_deal(T);
    const int p0 = GR_DEAL ? 0 : snap((long long)P * blockIdx.x / gridDim.x);
    const int p1 = GR_DEAL ? 0 : snap((long long)P * (blockIdx.x + 1) / gridDim.x);
    cl_f4 acc = zero;
    for (int ub = 0; ub < (GR_DEAL ? deal.nloc : 1); ub += GR_SB) {
    const int nb = min(GR_SB, deal.nloc - ub);
    if (GR_DEAL) gr_scan_slots(deal, ub, nb, off, s_pre, s_off);
    const int first = GR_DEAL ? 0 : p0, end = GR_DEAL ? s_pre[nb] : p1;
    for (int base = first; base < end; base += GR_CAP) {
        const int cnt = min(GR_CAP, end - base);
        __syncthreads();
        if (GR_DEAL) gr_stage_pairs(deal, ub, nb, base, cnt, s_pre, s_off, rev, revw, order, m, s_src, s_w, s_tgt);
        else
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const unsigned t = rtgt[base + i];
            const int k = (int)(t & ~GR_LAST);
            s_src[i] = rev[base + i];
            s_w[i] = revw[base + i];
            s_tgt[i] = (unsigned)(order ? (k / m) * m + order[k] : k) | (t & GR_LAST);
        }
        __syncthreads();
        if ((int)threadIdx.x < 3 * GR_U) s_src[cnt + threadIdx.x] = s_src[cnt - 1];   // padding: valid rows, results unused
        __syncthreads();
        // two-stage software pipeline over the flat pair stream; every load unconditional so that the waits can be counted
        auto load = [&](int i, cl_f4(&v)[GR_U]) {
#pragma unroll
            for (int u = 0; u < GR_U; ++u) v[u] = (g + (size_t)__builtin_amdgcn_readfirstlane(s_src[i + u]) * c4)[q];
        };
        auto walk = [&](int i, const cl_f4(&v)[GR_U]) {
#pragma unroll
            for (int u = 0; u < GR_U; ++u) {
                if (i + u < cnt) {
                    const float w = s_w[i + u];
                    acc.x = fmaf(w, v[u].x, acc.x);
                    acc.y = fmaf(w, v[u].y, acc.y);
                    acc.z = fmaf(w, v[u].z, acc.z);
                    acc.w = fmaf(w, v[u].w, acc.w);
                    const unsigned t = __builtin_amdgcn_readfirstlane(s_tgt[i + u]);
                    if (t & GR_LAST) {
                        if (on) __builtin_nontemporal_store(acc, out + (size_t)(t & ~GR_LAST) * c4 + q);
                        acc = zero;
                    }
                }
            }
        };
        cl_f4 va[GR_U], vb[GR_U];
        load(0, va);
        for (int i = 0; i < cnt; i += 2 * GR_U) {
            load(i + GR_U, vb);
            walk(i, va);
            load(i + 2 * GR_U, va);
            walk(i + GR_U, vb);
        }
    }
    }
}

// The same walk with the BatchNorm (+ ReLU) backward of the gathered tensor folded in: the rows that arrive are y (the
// BatchNorm's input) and dz (the gradient of its output); the gradient of y,
//     gy = k0 (g - c1 - xhat c2),   g = dz [y scale + shift > 0],   xhat = (y - mean) rstd        (bnrelu.hip bn_bwd_apply)
// is linear in (g, y - mean, 1), so a target's sum  sum_p w_p gy_p  =  k0 G + A Y - k0 c1 W  with  G = sum w g,
// Y = sum w (y - mean), W = sum w, A = -k0 rstd c2: three running sums per list, the constants applied once per target.
// gy is never written or read (1.18 GB each way at 8 x 24000 x 1536) and the bn_bwd_apply pass disappears.
constexpr int GRB_U = 4;    // pairs (2 row loads each) in flight per half of the software pipeline
__global__ __launch_bounds__(1024) void gather_rows_csr_bn_cl_kernel(
    int c4, int T, int P, int relu, const cl_f4 *__restrict__ y, const cl_f4 *__restrict__ dz, const cl_f4 *__restrict__ scale,
    const cl_f4 *__restrict__ shift, const cl_f4 *__restrict__ mean, const cl_f4 *__restrict__ rstd, const cl_f4 *__restrict__ c1,
    const cl_f4 *__restrict__ c2, const int *__restrict__ off, const int *__restrict__ rev, const float *__restrict__ revw,
    const unsigned *__restrict__ rtgt, const int *__restrict__ order, int m, cl_f4 *__restrict__ out)
{
    __shared__ int s_src[GR_CAP + 3 * GRB_U];
    __shared__ float s_w[GR_CAP];
    __shared__ unsigned s_tgt[GR_CAP];
    __shared__ int s_empty[GR_CAP];
    __shared__ int s_nempty;
    const int c4s = c4 / (int)gridDim.y;          // gridDim.y channel slabs (see gr_slabs)
    const bool on = (int)threadIdx.x < c4s;
    const int q = (int)blockIdx.y * c4s + (on ? (int)threadIdx.x : c4s - 1);
    const cl_f4 zero = {0.f, 0.f, 0.f, 0.f};
    const cl_f4 k0 = scale[q], sh = shift[q], mu = mean[q];
    const cl_f4 A = -(k0 * rstd[q] * c2[q]), k0c1 = k0 * c1[q];
    {   // targets without pairs: sum over nothing = 0
        const int per = (T + gridDim.x - 1) / gridDim.x;
        const int k0t = min(T, (int)blockIdx.x * per), k1t = min(T, k0t + per);
        for (int kb = k0t; kb < k1t; kb += GR_CAP) {
            if (threadIdx.x == 0) s_nempty = 0;
            __syncthreads();
            for (int k = kb + threadIdx.x; k < min(k1t, kb + GR_CAP); k += blockDim.x)
                if (off[k] == off[k + 1]) s_empty[atomicAdd(&s_nempty, 1)] = order ? (k / m) * m + order[k] : k;
            __syncthreads();
            const int ne = s_nempty;
            for (int i = 0; i < ne; ++i)
                if (on) out[(size_t)s_empty[i] * c4 + q] = zero;
            __syncthreads();
        }
    }
    __shared__ int s_pre[GR_SB + 1], s_off[GR_SB];
    auto snap = [&](long long s) -> int {
        if (s <= 0) return 0;
        if (s >= P) return P;
        const int k = (int)(rtgt[s] & ~GR_LAST);
        return off[k] == (int)s ? (int)s : off[k + 1];
    };
    const GrDeal deal = gr_deal(T);
    const int p0 = GR_DEAL ? 0 : snap((long long)P * blockIdx.x / gridDim.x);
    const int p1 = GR_DEAL ? 0 : snap((long long)P * (blockIdx.x + 1) / gridDim.x);
    cl_f4 G = zero, Y = zero;
    float W = 0.f;
    for (int ub = 0; ub < (GR_DEAL ? deal.nloc : 1); ub += GR_SB) {
    const int nb = min(GR_SB, deal.nloc - ub);
    if (GR_DEAL) gr_scan_slots(deal, ub, nb, off, s_pre, s_off);
    const int first = GR_DEAL ? 0 : p0, end = GR_DEAL ? s_pre[nb] : p1;
    for (int base = first; base < end; base += GR_CAP) {
        const int cnt = min(GR_CAP, end - base);
        __syncthreads();
        if (GR_DEAL) gr_stage_pairs(deal, ub, nb, base, cnt, s_pre, s_off, rev, revw, order, m, s_src, s_w, s_tgt);
        else
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const unsigned t = rtgt[base + i];
            const int k = (int)(t & ~GR_LAST);
            s_src[i] = rev[base + i];
            s_w[i] = revw[base + i];
            s_tgt[i] = (unsigned)(order ? (k / m) * m + order[k] : k) | (t & GR_LAST);
        }
        __syncthreads();
        if ((int)threadIdx.x < 3 * GRB_U) s_src[cnt + threadIdx.x] = s_src[cnt - 1];   // padding: valid rows, results unused
        __syncthreads();
        auto load = [&](int i, cl_f4(&vy)[GRB_U], cl_f4(&vg)[GRB_U]) {
#pragma unroll
            for (int u = 0; u < GRB_U; ++u) {
                const size_t row = (size_t)__builtin_amdgcn_readfirstlane(s_src[i + u]) * c4;
                vy[u] = (y + row)[q];
                vg[u] = (dz + row)[q];
            }
        };
        auto walk = [&](int i, const cl_f4(&vy)[GRB_U], const cl_f4(&vg)[GRB_U]) {
#pragma unroll
            for (int u = 0; u < GRB_U; ++u) {
                if (i + u < cnt) {
                    const float w = s_w[i + u];
                    cl_f4 g;
                    g.x = (!relu || fmaf(vy[u].x, k0.x, sh.x) > 0.f) ? vg[u].x : 0.f;
                    g.y = (!relu || fmaf(vy[u].y, k0.y, sh.y) > 0.f) ? vg[u].y : 0.f;
                    g.z = (!relu || fmaf(vy[u].z, k0.z, sh.z) > 0.f) ? vg[u].z : 0.f;
                    g.w = (!relu || fmaf(vy[u].w, k0.w, sh.w) > 0.f) ? vg[u].w : 0.f;
                    G = __builtin_elementwise_fma((cl_f4)(w), g, G);
                    Y = __builtin_elementwise_fma((cl_f4)(w), vy[u] - mu, Y);
                    W += w;
                    const unsigned t = __builtin_amdgcn_readfirstlane(s_tgt[i + u]);
                    if (t & GR_LAST) {
                        const cl_f4 r = __builtin_elementwise_fma(k0, G, __builtin_elementwise_fma(A, Y, -(k0c1 * W)));
                        if (on) __builtin_nontemporal_store(r, out + (size_t)(t & ~GR_LAST) * c4 + q);
                        G = Y = zero;
                        W = 0.f;
                    }
                }
            }
        };
        cl_f4 ya[GRB_U], ga[GRB_U], yb[GRB_U], gb[GRB_U];
        load(0, ya, ga);
        for (int i = 0; i < cnt; i += 2 * GRB_U) {
            load(i + GRB_U, yb, gb);
            walk(i, ya, ga);
            load(i + 2 * GRB_U, ya, ga);
            walk(i + GRB_U, yb, gb);
        }
    }
    }
}

// ---- the BatchNorm form for FEW targets: the sources stream, the targets' sums stay in registers ------------------------
// At the two 512-target stages (8192 <- 512 and 4096 <- 512: 48 and 24 pairs per target) the list walk reads every source
// row ~2.5 times (profiles/r04_gr_small_stages.txt): a band of lists wants far more rows than an XCD's L2 keeps.  With so few
// targets the roles turn: a workgroup takes (cloud, 128-byte slab of the rows) and ALL targets -- thread (target slot, column)
// keeps the three running sums of up to GC_TPT targets in registers -- and the sources stream past in memory order, GC_K rows
// at a time through a double-buffered LDS chunk (loads two chunks ahead, masked and centred on the way in): every row is
// fetched ONCE, by 128-byte coalesced pieces.  What a thread adds from a chunk is decided once per index (rix_chunks_kernel):
// the chunk's pairs grouped by target, ascending pair id inside a group, as (row within the chunk, weight) entries behind an
// offset table -- both copied into the chunk's buffer.  Chunks ascend, pair ids ascend inside a chunk: a target's sum is formed
// in the list walk's order with the list walk's fma chain -- the same bits; no atomics, one writer per output element.
constexpr int GC_K = 256;                  // source rows per chunk
constexpr int GC_S = 8;                    // float4 columns per slab (128 bytes of a row)
constexpr int GC_THREADS = 1024;           // GC_RPT (row, column) elements per thread while staging
constexpr int GC_RPT = GC_K * GC_S / GC_THREADS;
constexpr int GC_TPT = 4;                  // targets per thread: slots tid / GC_S + 128 i
constexpr int GC_MAXM = GC_TPT * GC_THREADS / GC_S;     // 512
constexpr int GC_MAXNT = 4;                // pairs per source (entries of a chunk: GC_K x nt <= 1024)

static inline bool gc_shape(long long L, int m, int nt) { return m <= GC_MAXM && nt <= GC_MAXNT && L * nt >= 16LL * m; }
static inline bool gc_applies(long long L, int m, int nt)
{
    const char *e = getenv("GEOT_GR_FORM");        // "list": the list walk everywhere (A/B runs)
    return !(e && e[0] == 'l') && gc_shape(L, m, nt);
}
static inline long long gc_chunks(long long L) { return (L + GC_K - 1) / GC_K; }

// per (cloud, chunk): coff[t] = first entry | entries << 16 of target t (the chunk's entries grouped by target), cent = (row within
// the chunk, weight) in ascending pair order inside a group
__global__ __launch_bounds__(GC_K * GC_MAXNT) void rix_chunks_kernel(int L, int m, int nt, long long nchunk, const int *__restrict__ idx,
                                                         const float *__restrict__ weight, int *__restrict__ coff,
                                                         uint2 *__restrict__ cent)
{
    __shared__ int cnt[GC_MAXM + 1], tg[GC_K * GC_MAXNT];
    const int tid = threadIdx.x;
    const long long bi = blockIdx.x / nchunk, ch = blockIdx.x - bi * nchunk;
    const int e0 = (int)ch * GC_K, rows = min(GC_K, L - e0), np = rows * nt;
    const long long x0 = (bi * L + e0) * nt;                 // first pair of the chunk
    for (int i = tid; i <= m; i += GC_K * GC_MAXNT) cnt[i] = 0;
    __syncthreads();
    const int j = tid < np ? idx[x0 + tid] : -1;
    if (tid < np) {
        tg[tid] = j;
        atomicAdd(&cnt[j + 1], 1);
    }
    __syncthreads();
    if (tid < 64) {                                           // inclusive scan of cnt[1 .. m] by one wave
        int carry = 0;
        for (int piece = 0; piece < m; piece += 64) {
            const int i = piece + tid;
            int v = i < m ? cnt[i + 1] : 0;
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                const int o = __shfl_up(v, sh);
                if (tid >= sh) v += o;
            }
            if (i < m) cnt[i + 1] = v + carry;
            carry += __shfl(v, 63);
        }
    }
    __syncthreads();
    int *co = coff + (size_t)blockIdx.x * (m + 1);
    for (int i = tid; i < m; i += GC_K * GC_MAXNT) co[i] = cnt[i] | ((cnt[i + 1] - cnt[i]) << 16);    // first entry | entries (both <= 1024)
    if (tid == 0) co[m] = cnt[m];
    if (tid < np) {
        int r = 0;                                            // pairs of the same target in front of this one
        for (int y = 0; y < tid; ++y) r += tg[y] == j ? 1 : 0;
        cent[(size_t)blockIdx.x * (GC_K * GC_MAXNT) + cnt[j] + r] = make_uint2((unsigned)(tid / nt), __float_as_uint(weight ? weight[x0 + tid] : 1.f));
    }
}

__global__ __launch_bounds__(GC_THREADS) void gather_rows_chunks_bn_cl_kernel(
    int c4, int L, int m, int nt, long long nchunk, int relu, const cl_f4 *__restrict__ y, const cl_f4 *__restrict__ dz,
    const cl_f4 *__restrict__ scale, const cl_f4 *__restrict__ shift, const cl_f4 *__restrict__ mean, const cl_f4 *__restrict__ rstd,
    const cl_f4 *__restrict__ c1, const cl_f4 *__restrict__ c2, const int *__restrict__ coff, const uint2 *__restrict__ cent,
    cl_f4 *__restrict__ out)
{
    // two buffers of: rows [GC_K][GC_S] masked gradient | [GC_K][GC_S] centred input | offsets [m + 1 .. padded to GC_MAXM + 4] | entries
    constexpr int ROWS = GC_K * GC_S, OFFS = GC_MAXM + 4, ENTS = GC_K * GC_MAXNT;
    static_assert(ENTS == GC_THREADS && GC_RPT * GC_THREADS == ROWS, "one entry per thread, GC_RPT row elements per thread");
    constexpr int BUF_BYTES = 2 * ROWS * 16 + OFFS * 4 + ENTS * 8;
    extern __shared__ char gc_lds[];
    const int tid = threadIdx.x, col = tid % GC_S, slot = tid / GC_S;
    const int slabs = c4 / GC_S, slab = blockIdx.x % slabs, bi = blockIdx.x / slabs;
    const int q = slab * GC_S + col;
    const cl_f4 k0s = scale[q], sh = shift[q], mu = mean[q];
    const cl_f4 zero = {0.f, 0.f, 0.f, 0.f};
    auto rowsG = [&](int b) { return reinterpret_cast<cl_f4 *>(gc_lds + (size_t)b * BUF_BYTES); };
    auto rowsY = [&](int b) { return rowsG(b) + ROWS; };
    auto offs = [&](int b) { return reinterpret_cast<int *>(gc_lds + (size_t)b * BUF_BYTES + 2 * ROWS * 16); };
    auto ents = [&](int b) { return reinterpret_cast<uint2 *>(gc_lds + (size_t)b * BUF_BYTES + 2 * ROWS * 16 + OFFS * 4); };
    const cl_f4 *yb = y + (size_t)bi * L * c4 + q, *gb = dz + (size_t)bi * L * c4 + q;
    const int *cob = coff + (size_t)bi * nchunk * (m + 1);
    const uint2 *ceb = cent + (size_t)bi * nchunk * ENTS;
    // loads unconditional, indices clamped (a clamped value is never used)
    auto load_rows = [&](long long ch, cl_f4 (&vy)[GC_RPT], cl_f4 (&vg)[GC_RPT]) {
#pragma unroll
        for (int u = 0; u < GC_RPT; ++u) {
            const long long e = min(min(ch, nchunk - 1) * GC_K + slot + u * (GC_THREADS / GC_S), (long long)L - 1);
            vy[u] = yb[e * c4];
            vg[u] = gb[e * c4];
        }
    };
    auto load_index = [&](long long ch, int &o, uint2 &en) {
        ch = min(ch, nchunk - 1);
        o = cob[ch * (m + 1) + min(tid, m)];
        en = ceb[ch * ENTS + tid];
    };
    cl_f4 G[GC_TPT], Y[GC_TPT];
    float W[GC_TPT];
#pragma unroll
    for (int i = 0; i < GC_TPT; ++i) {
        G[i] = Y[i] = zero;
        W[i] = 0.f;
    }
    cl_f4 py[2][GC_RPT], pg[2][GC_RPT];
    int po;
    uint2 pe;
    load_rows(0, py[0], pg[0]);
    load_rows(1, py[1], pg[1]);
    load_index(0, po, pe);
    for (long long ch = 0; ch < nchunk; ch += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long long cc = ch + b;
            if (cc < nchunk) {                               // (uniform)
#pragma unroll
                for (int u = 0; u < GC_RPT; ++u) {
                    const cl_f4 vy = py[b][u], vg = pg[b][u];
                    cl_f4 g;
                    g.x = (!relu || fmaf(vy.x, k0s.x, sh.x) > 0.f) ? vg.x : 0.f;
                    g.y = (!relu || fmaf(vy.y, k0s.y, sh.y) > 0.f) ? vg.y : 0.f;
                    g.z = (!relu || fmaf(vy.z, k0s.z, sh.z) > 0.f) ? vg.z : 0.f;
                    g.w = (!relu || fmaf(vy.w, k0s.w, sh.w) > 0.f) ? vg.w : 0.f;
                    rowsG(b)[tid + u * GC_THREADS] = g;      // (row slot + u 128, column): tid = slot * GC_S + col
                    rowsY(b)[tid + u * GC_THREADS] = vy - mu;
                }
                if (tid <= m) offs(b)[tid] = po;
                ents(b)[tid] = pe;                           // (ENTS = GC_THREADS)
                load_rows(cc + 2, py[b], pg[b]);
                load_index(cc + 1, po, pe);
                __syncthreads();                             // chunk cc is in buffer b; every thread has finished chunk cc - 1
                const cl_f4 *sG = rowsG(b), *sY = rowsY(b);
                const int *so = offs(b);
                const uint2 *se = ents(b);
#pragma unroll
                for (int i = 0; i < GC_TPT; ++i) {
                    const int t = slot + i * (GC_THREADS / GC_S);
                    if (t < m) {
                        const int az = so[t], a = az & 0xffff, z = a + (az >> 16);
                        for (int p = a; p < z; ++p) {
                            const uint2 en = se[p];
                            const float w = __uint_as_float(en.y);
                            G[i] = __builtin_elementwise_fma((cl_f4)(w), sG[en.x * GC_S + col], G[i]);
                            Y[i] = __builtin_elementwise_fma((cl_f4)(w), sY[en.x * GC_S + col], Y[i]);
                            W[i] += w;
                        }
                    }
                }
            }
        }
    }
    const cl_f4 A = -(k0s * rstd[q] * c2[q]), k0c1 = k0s * c1[q];
#pragma unroll
    for (int i = 0; i < GC_TPT; ++i) {
        const int t = slot + i * (GC_THREADS / GC_S);
        if (t < m) {
            // (a target without pairs: W = 0 and the list walk writes +0)
            const cl_f4 r = W[i] == 0.f && G[i].x == 0.f ? __builtin_elementwise_fma(k0s, G[i], __builtin_elementwise_fma(A, Y[i], zero))
                                                         : __builtin_elementwise_fma(k0s, G[i], __builtin_elementwise_fma(A, Y[i], -(k0c1 * W[i])));
            __builtin_nontemporal_store(r, out + ((size_t)bi * m + t) * c4 + q);
        }
    }
}

// the plain row gather (geot_gather_rows_csr_cl) over the same chunks: one tensor streams, one running sum per (target, column)
__global__ __launch_bounds__(GC_THREADS) void gather_rows_chunks_cl_kernel(int c4, int L, int m, long long nchunk, const cl_f4 *__restrict__ g,
                                                                          const int *__restrict__ coff, const uint2 *__restrict__ cent,
                                                                          cl_f4 *__restrict__ out)
{
    constexpr int ROWS = GC_K * GC_S, OFFS = GC_MAXM + 4, ENTS = GC_K * GC_MAXNT;
    constexpr int BUF_BYTES = ROWS * 16 + OFFS * 4 + ENTS * 8;
    extern __shared__ char gc_lds[];
    const int tid = threadIdx.x, col = tid % GC_S, slot = tid / GC_S;
    const int slabs = c4 / GC_S, slab = blockIdx.x % slabs, bi = blockIdx.x / slabs;
    const int q = slab * GC_S + col;
    const cl_f4 zero = {0.f, 0.f, 0.f, 0.f};
    auto rows = [&](int b) { return reinterpret_cast<cl_f4 *>(gc_lds + (size_t)b * BUF_BYTES); };
    auto offs = [&](int b) { return reinterpret_cast<int *>(gc_lds + (size_t)b * BUF_BYTES + ROWS * 16); };
    auto ents = [&](int b) { return reinterpret_cast<uint2 *>(gc_lds + (size_t)b * BUF_BYTES + ROWS * 16 + OFFS * 4); };
    const cl_f4 *gb = g + (size_t)bi * L * c4 + q;
    const int *cob = coff + (size_t)bi * nchunk * (m + 1);
    const uint2 *ceb = cent + (size_t)bi * nchunk * ENTS;
    auto load_rows = [&](long long ch, cl_f4 (&v)[GC_RPT]) {         // unconditional, indices clamped
#pragma unroll
        for (int u = 0; u < GC_RPT; ++u)
            v[u] = gb[min(min(ch, nchunk - 1) * GC_K + slot + u * (GC_THREADS / GC_S), (long long)L - 1) * c4];
    };
    auto load_index = [&](long long ch, int &o, uint2 &en) {
        ch = min(ch, nchunk - 1);
        o = cob[ch * (m + 1) + min(tid, m)];
        en = ceb[ch * ENTS + tid];
    };
    cl_f4 acc[GC_TPT];
#pragma unroll
    for (int i = 0; i < GC_TPT; ++i) acc[i] = zero;
    cl_f4 pv[2][GC_RPT];
    int po;
    uint2 pe;
    load_rows(0, pv[0]);
    load_rows(1, pv[1]);
    load_index(0, po, pe);
    for (long long ch = 0; ch < nchunk; ch += 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long long cc = ch + b;
            if (cc < nchunk) {                               // (uniform)
#pragma unroll
                for (int u = 0; u < GC_RPT; ++u) rows(b)[tid + u * GC_THREADS] = pv[b][u];
                if (tid <= m) offs(b)[tid] = po;
                ents(b)[tid] = pe;
                load_rows(cc + 2, pv[b]);
                load_index(cc + 1, po, pe);
                __syncthreads();
                const cl_f4 *sv = rows(b);
                const int *so = offs(b);
                const uint2 *se = ents(b);
#pragma unroll
                for (int i = 0; i < GC_TPT; ++i) {
                    const int t = slot + i * (GC_THREADS / GC_S);
                    if (t < m) {
                        const int az = so[t], a = az & 0xffff, z = a + (az >> 16);
                        for (int p = a; p < z; ++p) {
                            const uint2 en = se[p];
                            const float w = __uint_as_float(en.y);
                            const cl_f4 v = sv[en.x * GC_S + col];
                            acc[i].x = fmaf(w, v.x, acc[i].x);       // (the list walk's chain, component by component)
                            acc[i].y = fmaf(w, v.y, acc[i].y);
                            acc[i].z = fmaf(w, v.z, acc[i].z);
                            acc[i].w = fmaf(w, v.w, acc[i].w);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < GC_TPT; ++i) {
        const int t = slot + i * (GC_THREADS / GC_S);
        if (t < m) __builtin_nontemporal_store(acc[i], out + ((size_t)bi * m + t) * c4 + q);
    }
}

struct RixLayout {
    long long t, pairs, off, bsum, rank, rev, revw, rtgt, tmp, rank_of, coff, cent, ints;
};
static inline RixLayout rix_layout(int b, long long L, int m, int nt)
{
    RixLayout r;
    r.t = (long long)b * m;
    r.pairs = (long long)b * L * nt;
    r.off = 0;
    r.bsum = r.t + 1;
    r.rank = r.bsum + scan_blocks(r.t);
    r.rev = r.rank + r.pairs;
    r.revw = r.rev + r.pairs;
    r.rtgt = r.revw + r.pairs;
    r.tmp = r.rtgt + r.pairs;
    r.rank_of = r.tmp + r.pairs;
    // the few-target form's tables (gc_shape): per (cloud, chunk) an offset per target + 1, GC_K x GC_MAXNT entries of 2 words
    const long long chunks = gc_shape(L, m, nt) ? (long long)b * gc_chunks(L) : 0;
    r.coff = (r.rank_of + r.t + 1) & ~1LL;          // (entries are 8-byte words)
    r.cent = (r.coff + chunks * (m + 1) + 1) & ~1LL;
    r.ints = r.cent + chunks * GC_K * GC_MAXNT * 2 + 8;
    return r;
}

static inline dim3 grid3(long long inner, int c, int b)
{
    return dim3((unsigned)((inner + GG_THREADS - 1) / GG_THREADS), (unsigned)((c + GG_CCHUNK - 1) / GG_CCHUNK),
                (unsigned)b);
}
static inline dim3 grid1(long long total)
{
    long long blocks = (total + GG_THREADS - 1) / GG_THREADS;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks);
}

} // namespace geot

using namespace geot;

#define GEOT_CHECK_DIMS3(b, c) \
    if ((b) > 65535 || ((c) + GG_CCHUNK - 1) / GG_CCHUNK > 65535) return hipErrorInvalidValue

GEOT_EXPORT int geot_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                                   float *out, void *stream)
{
    if (b < 0 || c < 0 || n < 0 || m < 0) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || m == 0) return hipSuccess;
    GEOT_CHECK_DIMS3(b, c);
    const TldsPlan tp = tlds_plan(b, c, n, m, 1);
    if (tp.ch) return tlds_gather<1, false>(tp, b, c, n, m, points, idx, nullptr, out, (hipStream_t)stream);
    hipLaunchKernelGGL(gather_points_kernel, grid3(m, c, b), dim3(GG_THREADS), 0, (hipStream_t)stream, c,
                       n, m, points, idx, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                                        const int *idx, float *grad_points, void *stream)
{
    if (b < 0 || c < 0 || n < 0 || m < 0) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || m == 0) return hipSuccess;
    GEOT_CHECK_DIMS3(b, c);
    hipLaunchKernelGGL(gather_points_grad_kernel, grid3(m, c, b), dim3(GG_THREADS), 0,
                       (hipStream_t)stream, c, n, m, grad_out, idx, grad_points);
    return hipGetLastError();
}

// as geot_gather_points_grad without float atomics: workspace = geot_scatter_grad_ws_floats(b, c, n, m, 1, 0) floats
GEOT_EXPORT int geot_gather_points_grad_ws(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                           float *grad_points, float *workspace, void *stream)
{
    if (b < 0 || c < 0 || n < 0 || m < 0 || !workspace) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || m == 0 || n == 0) return hipSuccess;
    const long long wsf = geot_scatter_grad_ws_floats(b, c, n, m, 1, 0);
    hipError_t e = csr_preferred(b, c, n, m, 1) ? hipErrorNotSupported
                                                : scatter_via_tiles(b, c, n, m, 1, (size_t)c * m, grad_out, idx, nullptr, grad_points, workspace,
                                                                    wsf, (hipStream_t)stream, false);
    if (e != hipErrorNotSupported) return e;
    if (b <= 65535) {
        e = scatter_via_csr<1, false>(b, c, n, m, (size_t)c * m, grad_out, idx, nullptr, grad_points, workspace, wsf, (hipStream_t)stream);
        if (e != hipErrorNotSupported) return e;
    }
    return geot_gather_points_grad(b, c, n, m, grad_out, idx, grad_points, stream);
}

GEOT_EXPORT int geot_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                  const int *idx, float *out, void *stream)
{
    if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return hipErrorInvalidValue;
    long long npns = (long long)npoints * nsample;
    if (b == 0 || c == 0 || npns == 0) return hipSuccess;
    if (npns > 0x7fffffffLL) return hipErrorInvalidValue;
    GEOT_CHECK_DIMS3(b, c);
    const TldsPlan tp = tlds_plan(b, c, n, npns, 1);
    if (tp.ch) return tlds_gather<1, false>(tp, b, c, n, (int)npns, points, idx, nullptr, out, (hipStream_t)stream);
    hipLaunchKernelGGL(group_points_kernel, grid3(npns, c, b), dim3(GG_THREADS), 0, (hipStream_t)stream,
                       c, n, (int)npns, points, idx, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                       const float *grad_out, const int *idx, float *grad_points,
                                       void *stream)
{
    if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0) return hipErrorInvalidValue;
    long long npns = (long long)npoints * nsample;
    if (b == 0 || c == 0 || npns == 0) return hipSuccess;
    if (npns > 0x7fffffffLL) return hipErrorInvalidValue;
    GEOT_CHECK_DIMS3(b, c);
    hipLaunchKernelGGL(group_points_grad_kernel, grid3(npns, c, b), dim3(GG_THREADS), 0,
                       (hipStream_t)stream, c, n, (int)npns, grad_out, idx, grad_points);
    return hipGetLastError();
}

static int three_interpolate_launch(int b, int c, int m, int n, const float *points, const int *idx, const float *weight,
                                    float *out, size_t out_bstride, hipStream_t s)
{
    if (b < 0 || c < 0 || n < 0 || m < 0 || out_bstride < (size_t)c * n) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || n == 0) return hipSuccess;
    GEOT_CHECK_DIMS3(b, c);
    const TldsPlan tp = tlds_plan(b, c, m, n, 1);
    if (tp.ch) return tlds_gather<3, true>(tp, b, c, m, n, points, idx, weight, out, s, out_bstride);
    hipLaunchKernelGGL(three_interpolate_kernel, grid3(n, c, b), dim3(GG_THREADS), 0, s, c, m, n, points, idx, weight, out,
                       out_bstride);
    return hipGetLastError();
}

GEOT_EXPORT int geot_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                                       const float *weight, float *out, void *stream)
{
    return three_interpolate_launch(b, c, m, n, points, idx, weight, out, (size_t)(c > 0 ? c : 0) * (n > 0 ? n : 0),
                                    (hipStream_t)stream);
}

// FP-module front end without the concat copy: `out` is the first c channels of a wider (B, c + c_skip, n)
// buffer, out_bstride = (c + c_skip) * n floats between batches
GEOT_EXPORT int geot_three_interpolate_into(int b, int c, int m, int n, const float *points, const int *idx,
                                            const float *weight, float *out, long long out_bstride, void *stream)
{
    if (out_bstride < 0) return hipErrorInvalidValue;
    return three_interpolate_launch(b, c, m, n, points, idx, weight, out, (size_t)out_bstride, (hipStream_t)stream);
}

// weight[b,j,t] = r_t / ((r_0 + r_1) + r_2), r_t = 1 / (sqrt(dist2[b,j,t]) + 1e-8): pointnet2_modules.py:621-623
// on three_nn's squared distances, one kernel instead of sqrt / add / reciprocal / sum / div
__global__ __launch_bounds__(256) void fp_weights_kernel(long long rows, const float *__restrict__ dist2,
                                                         float *__restrict__ weight)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= rows) return;
    const float r0 = 1.0f / (sqrtf(dist2[3 * j]) + 1e-8f), r1 = 1.0f / (sqrtf(dist2[3 * j + 1]) + 1e-8f),
                r2 = 1.0f / (sqrtf(dist2[3 * j + 2]) + 1e-8f);
    const float norm = (r0 + r1) + r2;
    weight[3 * j] = r0 / norm;
    weight[3 * j + 1] = r1 / norm;
    weight[3 * j + 2] = r2 / norm;
}

GEOT_EXPORT int geot_fp_weights(int b, int n, const float *dist2, float *weight, void *stream)
{
    if (b < 0 || n < 0) return hipErrorInvalidValue;
    const long long rows = (long long)b * n;
    if (rows == 0) return hipSuccess;
    hipLaunchKernelGGL(fp_weights_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows, dist2,
                       weight);
    return hipGetLastError();
}

GEOT_EXPORT int geot_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                            const int *idx, const float *weight, float *grad_points,
                                            void *stream)
{
    if (b < 0 || c < 0 || n < 0 || m < 0) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || n == 0) return hipSuccess;
    GEOT_CHECK_DIMS3(b, c);
    hipLaunchKernelGGL(three_interpolate_grad_kernel, grid3(n, c, b), dim3(GG_THREADS), 0,
                       (hipStream_t)stream, c, n, m, grad_out, idx, weight, grad_points);
    return hipGetLastError();
}

// 1 if a *_grad_ws call with these sizes (targets m per batch, L source elements with nt slots each, c
// channels, workspace of b*m*c floats) accumulates in the workspace and needs it zero-filled; 0 if it
// only uses it as scratch (reverse-index path) and any contents will do.
GEOT_EXPORT int geot_grad_ws_needs_zero(int b, int c, int m, long long L, int nt)
{
    if (!csr_preferred(b, c, m, L, nt) && ts_ws_ints(b, c, m, L, nt, nt == 3) > 0) return 0;
    return csr_applies(b, c, m, L, nt, geot_scatter_grad_ws_floats(b, c, m, L, nt, nt == 3)) ? 0 : 1;
}

// floats of workspace the *_grad_ws / _grad_out / _grad_from entry points take for these sizes: b*m*c (the
// channels-last accumulator of the fallback) or the sorted pair stream of csrc/tile_scatter.hip, whichever is larger
GEOT_EXPORT long long geot_scatter_grad_ws_floats(int b, int c, int m, long long L, int nt, int weighted)
{
    if (b < 1 || c < 1 || m < 1) return 0;
    const long long base = (long long)b * m * c, tiles = L > 0 && nt > 0 ? ts_ws_ints(b, c, m, L, nt, weighted != 0) : 0;
    const long long rows = L > 0 && nt > 0 && L <= TLDS_FLOATS && rix_plan(c, L).Q == 1 && (double)L * nt >= 12.0 * m
                               ? rix_ws_ints(b, c, m, L, nt) : 0;           // the whole-rows gather (csr_preferred)
    return base > tiles ? (base > rows ? base : rows) : (tiles > rows ? tiles : rows);
}

static int three_interpolate_grad_launch(int b, int c, int n, int m, const float *grad_out, size_t grad_bstride,
                                         const int *idx, const float *weight, float *grad_points, float *workspace,
                                         hipStream_t s, bool overwrite = false)
{
    if (b < 0 || c < 0 || n < 0 || m < 0 || !workspace || grad_bstride < (size_t)c * n) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || m == 0) return hipSuccess;
    if (n == 0) return overwrite ? (int)zero_words(grad_points, (long long)b * c * m, s) : (int)hipSuccess;
    if (b > 65535) return hipErrorInvalidValue;
    {
        const long long wsf = geot_scatter_grad_ws_floats(b, c, m, n, 3, 1);
        hipError_t e = csr_preferred(b, c, m, n, 3) ? hipErrorNotSupported
                                                    : scatter_via_tiles(b, c, m, n, 3, grad_bstride, grad_out, idx, weight, grad_points,
                                                                        workspace, wsf, s, overwrite);
        if (e != hipErrorNotSupported) return e;
        e = scatter_via_csr<3, true>(b, c, m, n, grad_bstride, grad_out, idx, weight, grad_points, workspace,
                                                wsf, s, overwrite);
        if (e != hipErrorNotSupported) return e;
    }
    if (overwrite) {   // the channels-last scatter accumulates in the workspace and adds into grad_points
        hipError_t e = zero_words(grad_points, (long long)b * c * m, s);
        if (e == hipSuccess) e = zero_words(workspace, (long long)b * c * m, s);
        if (e != hipSuccess) return e;
    }
    dim3 g1((n + SC_TILE - 1) / SC_TILE, (c + SC_TILE - 1) / SC_TILE, b);
    hipLaunchKernelGGL((scatter_rows_cl_kernel<3, true>), g1, dim3(256), 0, s, c, n, m, grad_out, grad_bstride, idx, weight,
                       workspace);
    dim3 g2((m + SC_TILE - 1) / SC_TILE, (c + SC_TILE - 1) / SC_TILE, b);
    hipLaunchKernelGGL(transpose_add_kernel, g2, dim3(256), 0, s, c, m, workspace, grad_points);
    return hipGetLastError();
}

GEOT_EXPORT int geot_three_interpolate_grad_ws(int b, int c, int n, int m, const float *grad_out,
                                               const int *idx, const float *weight, float *grad_points,
                                               float *workspace, void *stream)
{
    return three_interpolate_grad_launch(b, c, n, m, grad_out, (size_t)(c > 0 ? c : 0) * (n > 0 ? n : 0), idx, weight,
                                         grad_points, workspace, (hipStream_t)stream);
}

// as _grad_ws, but grad_points and the workspace arrive UNINITIALISED and every element of grad_points is written
GEOT_EXPORT int geot_three_interpolate_grad_out(int b, int c, int n, int m, const float *grad_out, const int *idx,
                                                const float *weight, float *grad_points, float *workspace, void *stream)
{
    return three_interpolate_grad_launch(b, c, n, m, grad_out, (size_t)(c > 0 ? c : 0) * (n > 0 ? n : 0), idx, weight,
                                         grad_points, workspace, (hipStream_t)stream, true);
}

// as _grad_ws with grad_out being the first c channels of a wider (B, c + c_skip, n) gradient
GEOT_EXPORT int geot_three_interpolate_grad_from(int b, int c, int n, int m, const float *grad_out,
                                                 long long grad_bstride, const int *idx, const float *weight,
                                                 float *grad_points, float *workspace, void *stream)
{
    if (grad_bstride < 0) return hipErrorInvalidValue;
    return three_interpolate_grad_launch(b, c, n, m, grad_out, (size_t)grad_bstride, idx, weight, grad_points, workspace,
                                         (hipStream_t)stream);
}

// ---- reverse index as an object of its own + the point-major gradient over it ---------------------------------------
// The index depends on the neighbour ids alone: a caller that knows them early (the model's index plan, side stream)
// builds it once, off the critical path, and every gradient that needs it finds it ready.
// Workspace layout (ints): offsets [b m + 1] | scan scratch | ranks | sources | weights | pair ids.
GEOT_EXPORT long long geot_rix_ws_ints(int b, long long L, int m, int nt)
{
    if (b < 1 || L < 1 || m < 1 || nt < 1) return 0;
    return rix_layout(b, L, m, nt).ints;
}

GEOT_EXPORT int geot_rix_build(int b, int L, int m, int nt, const int *idx, const float *weight, const int *order, int *ws,
                               long long ws_ints, void *stream)
{
    if (b < 0 || L < 0 || m < 1 || nt < 1 || !ws) return hipErrorInvalidValue;
    const RixLayout r = rix_layout(b, L, m, nt);
    if (ws_ints < r.ints || r.pairs > 0x7ffffff0LL || r.t > 0x7ffffff0LL || (long long)b * L > 0x7fffffffLL)
        return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int *off = ws + r.off;
    hipError_t e = zero_words(off, r.t + 1, s);
    if (e != hipSuccess || r.pairs == 0) return e;
    int *rank_of = nullptr;
    if (order) {
        rank_of = ws + r.rank_of;
        hipLaunchKernelGGL(invert_order_kernel, dim3((unsigned)((r.t + 255) / 256)), dim3(256), 0, s, r.t, m, order, rank_of);
    }
    const int pb = (int)((r.pairs + 255) / 256);
    const long long pbatch = (long long)L * nt;
    hipLaunchKernelGGL(rix_count_kernel, dim3(pb), dim3(256), 0, s, r.pairs, pbatch, m, nt, 1, L, idx, off, ws + r.rank, rank_of);
    exclusive_scan_i32((int)r.t, off, ws + r.bsum, nullptr, s);
    float *revw = (float *)(ws + r.revw);
    // pair ids into the lists first (any order), then every pair to its slot in ascending pair order
    hipLaunchKernelGGL((rix_fill_kernel<false>), dim3(pb), dim3(256), 0, s, r.pairs, pbatch, m, nt, 1, L, idx, nullptr, off,
                       ws + r.rank, ws + r.rev, revw, ws + r.tmp, rank_of);
    if (weight)
        hipLaunchKernelGGL((rix_place_cl_kernel<true>), dim3(pb), dim3(256), 0, s, r.pairs, pbatch, m, nt, L, idx, weight, rank_of,
                           off, ws + r.rank, ws + r.tmp, ws + r.rev, revw, (unsigned *)(ws + r.rtgt));
    else
        hipLaunchKernelGGL((rix_place_cl_kernel<false>), dim3(pb), dim3(256), 0, s, r.pairs, pbatch, m, nt, L, idx, weight, rank_of,
                           off, ws + r.rank, ws + r.tmp, ws + r.rev, revw, (unsigned *)(ws + r.rtgt));
    if (gc_shape(L, m, nt))                         // (whatever GEOT_GR_FORM says now: the index may outlive the setting)
        hipLaunchKernelGGL(rix_chunks_kernel, dim3((unsigned)(b * gc_chunks(L))), dim3(GC_K * GC_MAXNT), 0, s, L, m, nt, gc_chunks(L), idx, weight,
                           ws + r.coff, (uint2 *)(ws + r.cent));
    return hipGetLastError();
}

// Channel slabs of the row gathers: the grid's y dimension cuts a row into `slabs` pieces of c4 / slabs float4 columns, and
// the workgroups are dispatched slab after slab (x runs fastest).  Why: a source row is used by three targets that are
// neighbours in the walk; between two uses the XCD's workgroups read (workgroups x pairs in flight) other rows, and at
// 6 KB per row (x 2 tensors in the BatchNorm form) that window does not fit the XCD's 4-MB L2 -- rocprofv3 FETCH_SIZE 2.2x
// the algorithmic bytes, L2 hit rate 29 %.  A 1-KB slab of the same rows does: at C = 1536 (one wave per workgroup, 6 slabs,
// 4 rounds of workgroups) FETCH_SIZE 5210 -> 3610 MB (1.5x) / 2263 -> 1334 MB (1.1x), L2 hits 29 -> 53 %, 800 -> 682 us and
// 492 -> 387 us (profiles/r04_gr_slabs.txt).  Narrow rows (C <= 1024) are best left whole: the per-workgroup staging of the
// pair stream is then shared by fewer lanes (C = 384: 96 -> 128 us with 3 slabs).  Lists, their order and the one-writer rule
// are untouched: bit-identical results.  GEOT_GR_SLABS / GEOT_CL_TILES_MULT override (lab).
static inline int gr_slabs(int c4)
{
    int slabs = (c4 >= 384 && c4 % 64 == 0) ? c4 / 64 : 1;
    if (const char *e = getenv("GEOT_GR_SLABS")) slabs = atoi(e);
    if (slabs < 1 || c4 % slabs) slabs = 1;
    return slabs;
}
static inline int gr_rounds(int c4)
{
    if (const char *mult = getenv("GEOT_CL_TILES_MULT")) return atoi(mult) > 0 ? atoi(mult) : 1;
    return c4 >= 256 ? 4 : 2;       // rounds of co-resident workgroups per slab: shorter shares keep a band of lists closer together
}

// g_cl (B, L, C) -> out_cl (B, m, C) through the index geot_rix_build left in `ws` (same b, L, m, nt and the SAME
// `order` it was built with)
GEOT_EXPORT int geot_gather_rows_csr_cl(int b, int c, int L, int m, int nt, const float *g_cl, const int *ws, const int *order,
                                        float *out_cl, void *stream)
{
    if (b < 0 || c < 0 || L < 0 || m < 0 || nt < 1 || !ws) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || m == 0) return hipSuccess;
    if (c % 4 || c / 4 > 1024) return hipErrorInvalidValue;
    const RixLayout r = rix_layout(b, L, m, nt);
    if (r.pairs > 0x7ffffff0LL || r.t > 0x7ffffff0LL) return hipErrorInvalidValue;
    if ((c / 4) % GC_S == 0 && gc_applies(L, m, nt) && (long long)b * (c / 4 / GC_S) <= 0x7fffffffLL) {       // few targets: see the BatchNorm form
        const size_t lds = 2 * (size_t)(GC_K * GC_S * 16 + (GC_MAXM + 4) * 4 + GC_K * GC_MAXNT * 8);
        hipError_t e = allow_big_lds((const void *)gather_rows_chunks_cl_kernel, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gather_rows_chunks_cl_kernel, dim3((unsigned)(b * (c / 4 / GC_S))), dim3(GC_THREADS), lds, (hipStream_t)stream, c / 4,
                           L, m, gc_chunks(L), (const cl_f4 *)g_cl, ws + r.coff, (const uint2 *)(ws + r.cent), (cl_f4 *)out_cl);
        return hipGetLastError();
    }
    const int c4 = c / 4, slabs = gr_slabs(c4), waves = (c4 / slabs + 63) / 64;
    const int cus = device_cus();
    static int per_cu_of[GEOT_DEV_SLOTS][1024 / 64 + 1];       // [device][waves] -> resident workgroups per CU
    int &per_cu = per_cu_of[device_slot()][waves];
    if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gather_rows_csr_cl_kernel, waves * 64, 0) != hipSuccess || per_cu < 1))
        per_cu = 1;
    long long grid = (long long)cus * per_cu * gr_rounds(c4);  // rounds of co-resident workgroups, equal shares of the targets
    if (grid > r.t) grid = r.t;
    hipLaunchKernelGGL(gather_rows_csr_cl_kernel, dim3((unsigned)grid, slabs), dim3(waves * 64), 0, (hipStream_t)stream, c4, (int)r.t,
                       (int)r.pairs, (const cl_f4 *)g_cl, ws + r.off, ws + r.rev, (const float *)(ws + r.revw),
                       (const unsigned *)(ws + r.rtgt), order, m, (cl_f4 *)out_cl);
    return hipGetLastError();
}

// out_cl (b, m, c) = the interpolation gradient of gy = BatchNorm(+ReLU)-backward(y_cl, dz_cl) without forming gy:
// scale = gamma rstd (also the k0 of geot_bn_bwd_apply), shift, mean, rstd of the forward, c1 / c2 from geot_bn_bwd_coef
GEOT_EXPORT int geot_gather_rows_csr_bn_cl(int b, int c, int L, int m, int nt, int relu, const float *y_cl, const float *dz_cl,
                                           const float *scale, const float *shift, const float *mean, const float *rstd,
                                           const float *c1, const float *c2, const int *ws, const int *order, float *out_cl,
                                           void *stream)
{
    if (b < 0 || c < 0 || L < 0 || m < 0 || nt < 1 || !ws) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || m == 0) return hipSuccess;
    if (c % 4 || c / 4 > 1024) return hipErrorInvalidValue;
    const RixLayout r = rix_layout(b, L, m, nt);
    if (r.pairs > 0x7ffffff0LL || r.t > 0x7ffffff0LL) return hipErrorInvalidValue;
    if ((c / 4) % GC_S == 0 && gc_applies(L, m, nt) && (long long)b * (c / 4 / GC_S) <= 0x7fffffffLL) {
        // few targets: the sources stream once, the sums stay in registers (the target order plays no part: rows are written by id)
        const size_t lds = 2 * (size_t)(2 * GC_K * GC_S * 16 + (GC_MAXM + 4) * 4 + GC_K * GC_MAXNT * 8);
        hipError_t e = allow_big_lds((const void *)gather_rows_chunks_bn_cl_kernel, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(gather_rows_chunks_bn_cl_kernel, dim3((unsigned)(b * (c / 4 / GC_S))), dim3(GC_THREADS), lds, (hipStream_t)stream,
                           c / 4, L, m, nt, gc_chunks(L), relu, (const cl_f4 *)y_cl, (const cl_f4 *)dz_cl, (const cl_f4 *)scale,
                           (const cl_f4 *)shift, (const cl_f4 *)mean, (const cl_f4 *)rstd, (const cl_f4 *)c1, (const cl_f4 *)c2,
                           ws + r.coff, (const uint2 *)(ws + r.cent), (cl_f4 *)out_cl);
        return hipGetLastError();
    }
    const int c4 = c / 4, slabs = gr_slabs(c4), waves = (c4 / slabs + 63) / 64;
    const int cus = device_cus();
    static int per_cu_of[GEOT_DEV_SLOTS][1024 / 64 + 1];
    int &per_cu = per_cu_of[device_slot()][waves];
    if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gather_rows_csr_bn_cl_kernel, waves * 64, 0) != hipSuccess || per_cu < 1))
        per_cu = 1;
    long long grid = (long long)cus * per_cu * gr_rounds(c4);
    if (grid > r.t) grid = r.t;
    hipLaunchKernelGGL(gather_rows_csr_bn_cl_kernel, dim3((unsigned)grid, slabs), dim3(waves * 64), 0, (hipStream_t)stream, c4, (int)r.t,
                       (int)r.pairs, relu, (const cl_f4 *)y_cl, (const cl_f4 *)dz_cl, (const cl_f4 *)scale, (const cl_f4 *)shift,
                       (const cl_f4 *)mean, (const cl_f4 *)rstd, (const cl_f4 *)c1, (const cl_f4 *)c2, ws + r.off, ws + r.rev,
                       (const float *)(ws + r.revw), (const unsigned *)(ws + r.rtgt), order, m, (cl_f4 *)out_cl);
    return hipGetLastError();
}

GEOT_EXPORT int geot_group_points_grad_ws(int b, int c, int n, int npoints, int nsample,
                                          const float *grad_out, const int *idx, float *grad_points,
                                          float *workspace, void *stream)
{
    if (b < 0 || c < 0 || n < 0 || npoints < 0 || nsample < 0 || !workspace) return hipErrorInvalidValue;
    long long npns = (long long)npoints * nsample;
    if (b == 0 || c == 0 || npns == 0 || n == 0) return hipSuccess;
    if (npns > 0x7fffffffLL || b > 65535) return hipErrorInvalidValue;
    {
        const long long wsf = geot_scatter_grad_ws_floats(b, c, n, npns, 1, 0);
        hipError_t e = csr_preferred(b, c, n, npns, 1) ? hipErrorNotSupported
                                                       : scatter_via_tiles(b, c, n, (int)npns, 1, (size_t)c * npns, grad_out, idx, nullptr,
                                                                           grad_points, workspace, wsf, (hipStream_t)stream, false);
        if (e != hipErrorNotSupported) return e;
        e = scatter_via_csr<1, false>(b, c, n, (int)npns, (size_t)c * npns, grad_out, idx, nullptr, grad_points, workspace,
                                                 wsf, (hipStream_t)stream);
        if (e != hipErrorNotSupported) return e;
    }
    dim3 g1((unsigned)((npns + SC_TILE - 1) / SC_TILE), (c + SC_TILE - 1) / SC_TILE, b);
    hipLaunchKernelGGL((scatter_rows_cl_kernel<1, false>), g1, dim3(256), 0, (hipStream_t)stream, c,
                       (int)npns, n, grad_out, (size_t)c * npns, idx, nullptr, workspace);
    dim3 g2((n + SC_TILE - 1) / SC_TILE, (c + SC_TILE - 1) / SC_TILE, b);
    hipLaunchKernelGGL(transpose_add_kernel, g2, dim3(256), 0, (hipStream_t)stream, c, n, workspace,
                       grad_points);
    return hipGetLastError();
}

GEOT_EXPORT int geot_graph_feature(int b, int c, int nq, int nk, int k, const float *x_q, const float *x_k,
                                   const int *idx, float *out, void *stream)
{
    if (b < 0 || c < 0 || nq < 0 || nk < 0 || k < 0) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || nq == 0 || k == 0) return hipSuccess;
    GEOT_CHECK_DIMS3(b, c);
    const bool al = ((uintptr_t)idx % 16 == 0) && ((uintptr_t)out % 16 == 0);
#define GF_LAUNCH(K4) hipLaunchKernelGGL(graph_feature_kernel<K4>, grid3(nq, c, b), dim3(GG_THREADS), 0, \
                                         (hipStream_t)stream, c, nq, nk, k, x_q, x_k, idx, out)
    if (al && k == 4) GF_LAUNCH(1);
    else if (al && k == 8) GF_LAUNCH(2);
    else if (al && k == 16) GF_LAUNCH(4);
    else GF_LAUNCH(0);
#undef GF_LAUNCH
    return hipGetLastError();
}

GEOT_EXPORT int geot_graph_feature_grad(int b, int c, int nq, int nk, int k, const float *grad_out,
                                        const int *idx, float *grad_xq, float *grad_xk, float *workspace,
                                        void *stream)
{
    if (b < 0 || c < 0 || nq < 0 || nk < 0 || k < 0 || !workspace) return hipErrorInvalidValue;
    if (b == 0 || c == 0 || nq == 0 || k == 0) return hipSuccess;
    if (b > 65535 || c > 65535 || (long long)nq * k > 0x7fffffffLL) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const dim3 gq((nq + GG_THREADS - 1) / GG_THREADS, c, b);
    const bool al = (uintptr_t)grad_out % 16 == 0;
    if (al && k == 4) hipLaunchKernelGGL(graph_feature_grad_q_kernel<1>, gq, dim3(GG_THREADS), 0, s, c, nq, k, grad_out, grad_xq);
    else if (al && k == 8) hipLaunchKernelGGL(graph_feature_grad_q_kernel<2>, gq, dim3(GG_THREADS), 0, s, c, nq, k, grad_out, grad_xq);
    else if (al && k == 16) hipLaunchKernelGGL(graph_feature_grad_q_kernel<4>, gq, dim3(GG_THREADS), 0, s, c, nq, k, grad_out, grad_xq);
    else hipLaunchKernelGGL(graph_feature_grad_q_kernel<0>, gq, dim3(GG_THREADS), 0, s, c, nq, k, grad_out, grad_xq);
    const int L = nq * k;
    {
        const long long wsf = geot_scatter_grad_ws_floats(b, c, nk, L, 1, 0);
        hipError_t e = csr_preferred(b, c, nk, L, 1) ? hipErrorNotSupported
                                                     : scatter_via_tiles(b, c, nk, L, 1, (size_t)2 * c * L, grad_out, idx, nullptr, grad_xk,
                                                                         workspace, wsf, s, false);
        if (e != hipErrorNotSupported) return e;
        e = scatter_via_csr<1, false>(b, c, nk, L, (size_t)2 * c * L, grad_out, idx, nullptr, grad_xk,
                                                 workspace, wsf, s);
        if (e != hipErrorNotSupported) return e;
    }
    dim3 g1((L + SC_TILE - 1) / SC_TILE, (c + SC_TILE - 1) / SC_TILE, b);
    hipLaunchKernelGGL((scatter_rows_cl_kernel<1, false>), g1, dim3(256), 0, s, c, L, nk, grad_out,
                       (size_t)2 * c * L, idx, nullptr, workspace);
    dim3 g2((nk + SC_TILE - 1) / SC_TILE, (c + SC_TILE - 1) / SC_TILE, b);
    hipLaunchKernelGGL(transpose_add_kernel, g2, dim3(256), 0, s, c, nk, workspace, grad_xk);
    return hipGetLastError();
}

GEOT_EXPORT int geot_grouping_cl(int m, int nsample, int c, const float *input, const int *idx,
                                 float *out, void *stream)
{
    long long total = (long long)m * nsample * c;
    if (m < 0 || nsample < 0 || c < 0) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(grouping_cl_kernel, grid1(total), dim3(GG_THREADS), 0, (hipStream_t)stream, total,
                       nsample, c, input, idx, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_grouping_cl_grad(int m, int nsample, int c, const float *grad_out, const int *idx,
                                      float *grad_in, void *stream)
{
    long long total = (long long)m * nsample * c;
    if (m < 0 || nsample < 0 || c < 0) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(grouping_cl_grad_kernel, grid1(total), dim3(GG_THREADS), 0, (hipStream_t)stream,
                       total, nsample, c, grad_out, idx, grad_in);
    return hipGetLastError();
}

GEOT_EXPORT int geot_interpolation_cl(int n, int c, int k, const float *input, const int *idx,
                                      const float *weight, float *out, void *stream)
{
    long long total = (long long)n * c;
    if (n < 0 || c < 0 || k < 0) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(interpolation_cl_kernel, grid1(total), dim3(GG_THREADS), 0, (hipStream_t)stream,
                       total, c, k, input, idx, weight, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_interpolation_cl_grad(int n, int c, int k, const float *grad_out, const int *idx,
                                           const float *weight, float *grad_in, void *stream)
{
    long long total = (long long)n * c;
    if (n < 0 || c < 0 || k < 0) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(interpolation_cl_grad_kernel, grid1(total), dim3(GG_THREADS), 0,
                       (hipStream_t)stream, total, c, k, grad_out, idx, weight, grad_in);
    return hipGetLastError();
}

GEOT_EXPORT int geot_subtraction_cl(int n, int nsample, int c, const float *in1, const float *in2,
                                    const int *idx, float *out, void *stream)
{
    long long total = (long long)n * nsample * c;
    if (n < 0 || nsample < 0 || c < 0) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(subtraction_cl_kernel, grid1(total), dim3(GG_THREADS), 0, (hipStream_t)stream,
                       total, nsample, c, in1, in2, idx, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_subtraction_cl_grad(int n, int nsample, int c, const int *idx,
                                         const float *grad_out, float *grad_in1, float *grad_in2,
                                         void *stream)
{
    long long total = (long long)n * nsample * c;
    if (n < 0 || nsample < 0 || c < 0) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(subtraction_cl_grad_kernel, grid1(total), dim3(GG_THREADS), 0,
                       (hipStream_t)stream, total, nsample, c, idx, grad_out, grad_in1, grad_in2);
    return hipGetLastError();
}

GEOT_EXPORT int geot_aggregation_cl(int n, int nsample, int c, int w_c, const float *input,
                                    const float *position, const float *weight, const int *idx,
                                    float *out, void *stream)
{
    long long total = (long long)n * c;
    if (n < 0 || nsample < 0 || c < 0 || w_c <= 0) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(aggregation_cl_kernel, grid1(total), dim3(GG_THREADS), 0, (hipStream_t)stream,
                       total, nsample, c, w_c, input, position, weight, idx, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_aggregation_cl_grad(int n, int nsample, int c, int w_c, const float *input,
                                         const float *position, const float *weight, const int *idx,
                                         const float *grad_out, float *grad_in, float *grad_position,
                                         float *grad_weight, void *stream)
{
    long long total = (long long)n * c;
    if (n < 0 || nsample < 0 || c < 0 || w_c <= 0) return hipErrorInvalidValue;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(aggregation_cl_grad_kernel, grid1(total), dim3(GG_THREADS), 0,
                       (hipStream_t)stream, total, nsample, c, w_c, input, position, weight, idx,
                       grad_out, grad_in, grad_position, grad_weight);
    return hipGetLastError();
}

GEOT_EXPORT int geot_abi_version(void) { return GEOT_ABI_VERSION; }

GEOT_EXPORT int geot_distance_mode(void) { return GEOT_DISTANCE_MODE; }

GEOT_EXPORT const char *geot_error_string(int e) { return hipGetErrorString((hipError_t)e); }
