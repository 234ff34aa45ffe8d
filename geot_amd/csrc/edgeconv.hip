// edgeconv.hip -- the tail of DGCNN_Propagation's EdgeConv layer as fused kernels for gfx950 (MI355X).
//
// Reference (behaviour, not code): openpoints/models/backbone/transformer.py:343-379
//     feature = cat(x_k[idx] - x_q, x_q)                       (B, 2C, Nq, k)      get_graph_feature
//     y       = Conv2d_1x1(feature)                            (B, Co, Nq, k)
//     out     = max_j LeakyReLU_0.2(GroupNorm_G(y))            (B, Co, Nq)
// The 1x1 convolution is linear, so  y[b,:,i,j] = P[b,:,idx[b,i,j]] + Q[b,:,i]  with the two small GEMMs
// P = W_d x_k (B,Co,Nk) and Q = (W_q - W_d) x_q (B,Co,Nq) done by the caller (rocBLAS).  What is left -- gather,
// add, GroupNorm statistics over (Co/G, Nq, k), normalise, LeakyReLU, max over k -- is HBM-bound work on a tensor
// k times larger than anything that has to exist: LeakyReLU and a positive scale are monotone, so
//     max_j LReLU(g (y_ij - mu) r + beta) = LReLU(g (sel_j y_ij - mu) r + beta),   sel = max if g >= 0 else min,
// and the statistics are sums over y that can be formed while the rows are gathered.  The (B,Co,Nq,k) tensor is
// never written: per (b, channel, query) the forward keeps the selected y, its slot j and the sum over j.
//
// Layout: channels-first, as the rest of the path.  A workgroup keeps CH whole rows of P (Nk floats each) in LDS
// (random 4-byte reads are one ds_read_b32 there; in global memory they are address-path bound) and streams its
// slice of the queries with coalesced loads / stores.  Backward: d/dQ is element-wise; d/dP is a gather over the
// reverse index of idx with the rows it needs staged in LDS (no float atomics, deterministic).
//
// Algorithmic bytes (forward): 4 (Co Nk + Co Nq) reads + 4 Nq k (indices) + Co Nq (4 + 4 + 1 + 4) writes per cloud.
#include "geot_common.h"
#include "geot_hip.h"

namespace geot {

constexpr int EC_THREADS = 1024;
constexpr int EC_LDS_BYTES = 150 * 1024; // of the CU's 160 KB

__device__ __forceinline__ void ec_load_row(float *__restrict__ dst, const float *__restrict__ src, int count)
{
    const int tid = threadIdx.x;
    if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0 && count >= 4) {
        const int vec = count >> 2;
        const float4 *s4 = reinterpret_cast<const float4 *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int e = tid; e < vec; e += 4 * EC_THREADS) {      // four loads in flight, unconditional (index clamped)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = s4[min(e + u * EC_THREADS, vec - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u * EC_THREADS < vec) d4[e + u * EC_THREADS] = v[u];
        }
        for (int e = (vec << 2) + tid; e < count; e += EC_THREADS) dst[e] = src[e];
    } else {
        for (int e = tid; e < count; e += EC_THREADS) dst[e] = src[e];
    }
}

__device__ __forceinline__ float ec_wave_sum(float v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// ---- forward 1: gather + add, select over k, sums for the statistics ---------------------------------------
// K4: k == 4 and the index rows are int4-aligned (the configured backbone); otherwise the generic k loop.
template <bool K4, int CH>
__global__ __launch_bounds__(EC_THREADS) void edge_fwd_kernel(
    int c, int nq, int nk, int k, const float *__restrict__ P, const float *__restrict__ Q,
    const int *__restrict__ idx, const float *__restrict__ gamma, float *__restrict__ ysel,
    float *__restrict__ ysum, uint8_t *__restrict__ jsel, float *__restrict__ partial)
{
    extern __shared__ float ec_rows[]; // [CH][nk]
    __shared__ float red[EC_THREADS / 64][CH][2];
    const int bi = blockIdx.z, c0 = blockIdx.y * CH, nch = min(CH, c - c0);
    ec_load_row(ec_rows, P + ((size_t)bi * c + c0) * nk, nch * nk);
    bool want_max[CH];
#pragma unroll
    for (int l = 0; l < CH; ++l) want_max[l] = l < nch ? gamma[c0 + l] >= 0.f : true;
    __syncthreads();
    const int per = (nq + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(nq, i0 + per);
    float s[CH], ss[CH];
#pragma unroll
    for (int l = 0; l < CH; ++l) s[l] = ss[l] = 0.f;
    if (K4 && nch == CH) {
        // The configured case, written for loads in flight: the indices and the CH values of Q of the NEXT query are requested
        // (unconditionally, index clamped) before this query's 3 CH stores are issued, so the wait at the top of the loop is a
        // counted vmcnt that leaves the stores in flight.  (The plain loop below loads Q[o] per channel behind the previous
        // channel's stores: every load waited for those stores to be acknowledged -- 2.8 TB/s.)
        const size_t qb = ((size_t)bi * c + c0) * nq;
        const int4 *idx4 = reinterpret_cast<const int4 *>(idx) + (size_t)bi * nq;
        int i = i0 + threadIdx.x;
        if (i < i1) {
            int4 n4 = idx4[i];
            float qv[CH];
#pragma unroll
            for (int l = 0; l < CH; ++l) qv[l] = Q[qb + (size_t)l * nq + i];
            // the first query's loads land before the loop: a pending load at the loop's entry would merge with the back edge's
            // pending stores into waits that drain the stores every iteration
            __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
            for (; i < i1; i += EC_THREADS) {
                const int in = min(i + EC_THREADS, i1 - 1);
                const int4 n4n = idx4[in];
                float qn[CH];
#pragma unroll
                for (int l = 0; l < CH; ++l) qn[l] = Q[qb + (size_t)l * nq + in];
                float pv[CH][4];
#pragma unroll
                for (int l = 0; l < CH; ++l) {
                    const float *R = ec_rows + l * nk;
                    pv[l][0] = R[n4.x];
                    pv[l][1] = R[n4.y];
                    pv[l][2] = R[n4.z];
                    pv[l][3] = R[n4.w];
                }
#pragma unroll
                for (int l = 0; l < CH; ++l) {
                    float best = 0.f, sum = 0.f;
                    int bj = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float y = pv[l][j] + qv[l];
                        sum += y;
                        ss[l] = fmaf(y, y, ss[l]);
                        const bool take = j == 0 || (want_max[l] ? y > best : y < best); // first extremum wins
                        best = take ? y : best;
                        bj = take ? j : bj;
                    }
                    s[l] += sum;
                    const size_t o = qb + (size_t)l * nq + i;
                    ysel[o] = best;
                    ysum[o] = sum;
                    jsel[o] = (uint8_t)bj;
                }
                n4 = n4n;
#pragma unroll
                for (int l = 0; l < CH; ++l) qv[l] = qn[l];
            }
        }
    } else
    for (int i = i0 + threadIdx.x; i < i1; i += EC_THREADS) {
        const size_t row = (size_t)bi * nq + i;
        if (K4) {
            const int4 n4 = *reinterpret_cast<const int4 *>(idx + row * 4);
            const int nn[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
            for (int l = 0; l < CH; ++l) {
                if (l < nch) {
                    const size_t o = ((size_t)bi * c + c0 + l) * nq + i;
                    const float q = Q[o];
                    const float *R = ec_rows + l * nk;
                    float best = 0.f, sum = 0.f;
                    int bj = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float y = R[nn[j]] + q;
                        sum += y;
                        ss[l] = fmaf(y, y, ss[l]);
                        const bool take = j == 0 || (want_max[l] ? y > best : y < best); // first extremum wins
                        best = take ? y : best;
                        bj = take ? j : bj;
                    }
                    s[l] += sum;
                    ysel[o] = best;
                    ysum[o] = sum;
                    jsel[o] = (uint8_t)bj;
                }
            }
        } else {
#pragma unroll
            for (int l = 0; l < CH; ++l) {
                if (l < nch) {
                    const size_t o = ((size_t)bi * c + c0 + l) * nq + i;
                    const float q = Q[o];
                    const float *R = ec_rows + l * nk;
                    float best = 0.f, sum = 0.f;
                    int bj = 0;
                    for (int j = 0; j < k; ++j) {
                        const float y = R[idx[row * k + j]] + q;
                        sum += y;
                        ss[l] = fmaf(y, y, ss[l]);
                        const bool take = j == 0 || (want_max[l] ? y > best : y < best);
                        best = take ? y : best;
                        bj = take ? j : bj;
                    }
                    s[l] += sum;
                    ysel[o] = best;
                    ysum[o] = sum;
                    jsel[o] = (uint8_t)bj;
                }
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int l = 0; l < CH; ++l) {
        const float a = ec_wave_sum(s[l]), b = ec_wave_sum(ss[l]);
        if (lane == 0) { red[wave][l][0] = a; red[wave][l][1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * CH) {
        const int l = threadIdx.x >> 1, w = threadIdx.x & 1;
        if (l < nch) {
            float t = 0.f;
            for (int v = 0; v < EC_THREADS / 64; ++v) t += red[v][l][w]; // fixed order: deterministic
            partial[(((size_t)bi * c + c0 + l) * gridDim.x + blockIdx.x) * 2 + w] = t;
        }
    }
}

// ---- forward 2: per (batch, group) mean and 1/sqrt(var + eps) from the per-(channel, slice) partial sums ----
__global__ __launch_bounds__(256) void edge_stats_kernel(int c, int groups, int slices, double count, float eps,
                                                         const float *__restrict__ partial, float *__restrict__ stats)
{
    __shared__ double red[256][2];
    const int bg = blockIdx.x, bi = bg / groups, g = bg - bi * groups, cpg = c / groups;
    const float *src = partial + ((size_t)bi * c + (size_t)g * cpg) * slices * 2;
    double a = 0.0, b = 0.0;
    for (int e = threadIdx.x; e < cpg * slices; e += 256) { a += src[2 * e]; b += src[2 * e + 1]; }
    red[threadIdx.x][0] = a;
    red[threadIdx.x][1] = b;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if (threadIdx.x < d) { red[threadIdx.x][0] += red[threadIdx.x + d][0]; red[threadIdx.x][1] += red[threadIdx.x + d][1]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean = red[0][0] / count;
        double var = red[0][1] / count - mean * mean;
        var = var > 0.0 ? var : 0.0;
        stats[2 * bg] = (float)mean;
        stats[2 * bg + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// ---- forward 3: out = LeakyReLU(gamma (ysel - mean) rstd + beta) --------------------------------------------
__global__ __launch_bounds__(256) void edge_out_kernel(int c, int nq, int groups, float slope,
                                                       const float *__restrict__ ysel, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const float *__restrict__ stats,
                                                       float *__restrict__ out)
{
    const int bi = blockIdx.z, cc = blockIdx.y, g = cc / (c / groups);
    const float mean = stats[2 * (bi * groups + g)], rstd = stats[2 * (bi * groups + g) + 1];
    const float a = gamma[cc] * rstd, b2 = beta[cc] - mean * a;
    const size_t base = ((size_t)bi * c + cc) * nq;
    if ((nq & 3) == 0 && ((((uintptr_t)ysel) | ((uintptr_t)out)) & 15) == 0) {       // 16-byte vectors
        const float4 *src = reinterpret_cast<const float4 *>(ysel + base);
        float4 *dst = reinterpret_cast<float4 *>(out + base);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < (nq >> 2); i += gridDim.x * 256) {
            const float4 y = src[i];
            float4 z = make_float4(fmaf(y.x, a, b2), fmaf(y.y, a, b2), fmaf(y.z, a, b2), fmaf(y.w, a, b2));
            z.x = z.x > 0.f ? z.x : slope * z.x;
            z.y = z.y > 0.f ? z.y : slope * z.y;
            z.z = z.z > 0.f ? z.z : slope * z.z;
            z.w = z.w > 0.f ? z.w : slope * z.w;
            dst[i] = z;
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nq; i += gridDim.x * 256) {
        const float z = fmaf(ysel[base + i], a, b2);
        out[base + i] = z > 0.f ? z : slope * z;
    }
}

// ---- backward 1: per (b, channel, slice): sum dz, sum dz * yhat_sel -------------------------------------------
__global__ __launch_bounds__(256) void edge_bwd_reduce_kernel(int c, int nq, int groups, float slope,
                                                              const float *__restrict__ ysel,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta,
                                                              const float *__restrict__ stats,
                                                              const float *__restrict__ grad_out,
                                                              float *__restrict__ bpart)
{
    __shared__ float red[4][2];
    const int bi = blockIdx.z, cc = blockIdx.y, g = cc / (c / groups);
    const float mean = stats[2 * (bi * groups + g)], rstd = stats[2 * (bi * groups + g) + 1];
    const float gm = gamma[cc], bt = beta[cc];
    const size_t base = ((size_t)bi * c + cc) * nq;
    const int per = (nq + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(nq, i0 + per);
    float a = 0.f, b = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float yh = (ysel[base + i] - mean) * rstd;
        const float z = fmaf(gm, yh, bt);
        const float dz = grad_out[base + i] * (z > 0.f ? 1.f : slope);
        a += dz;
        b = fmaf(dz, yh, b);
    }
    a = ec_wave_sum(a);
    b = ec_wave_sum(b);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x < 2) {
        const float t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
        bpart[(((size_t)bi * c + cc) * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = t;
    }
}

// ---- backward 2: per (b, group): s1 = mean(gamma dz), s2 = mean(gamma dz yhat) over the group's M elements;
// per channel: d gamma, d beta (summed over the batch) --------------------------------------------------------
__global__ __launch_bounds__(256) void edge_bwd_coef_kernel(int b, int c, int groups, int slices, double count,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ bpart, float *__restrict__ coef,
                                                            float *__restrict__ grad_gamma, float *__restrict__ grad_beta)
{
    __shared__ double red[256][2];
    const int cpg = c / groups;
    if ((int)blockIdx.x < b * groups) {
        const int bg = blockIdx.x, bi = bg / groups, g = bg - bi * groups;
        double a = 0.0, d = 0.0;
        for (int e = threadIdx.x; e < cpg * slices; e += 256) {
            const int cc = g * cpg + e / slices;
            const float *src = bpart + (((size_t)bi * c + cc) * slices + e % slices) * 2;
            a += (double)gamma[cc] * src[0];
            d += (double)gamma[cc] * src[1];
        }
        red[threadIdx.x][0] = a;
        red[threadIdx.x][1] = d;
        __syncthreads();
        for (int s = 128; s >= 1; s >>= 1) {
            if (threadIdx.x < s) { red[threadIdx.x][0] += red[threadIdx.x + s][0]; red[threadIdx.x][1] += red[threadIdx.x + s][1]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { coef[2 * bg] = (float)(red[0][0] / count); coef[2 * bg + 1] = (float)(red[0][1] / count); }
    } else {
        const int cc = (blockIdx.x - b * groups) * 256 + threadIdx.x;
        if (cc < c) {
            double a = 0.0, d = 0.0;
            for (int bi = 0; bi < b; ++bi)
                for (int s = 0; s < slices; ++s) {
                    const float *src = bpart + (((size_t)bi * c + cc) * slices + s) * 2;
                    a += src[0];
                    d += src[1];
                }
            grad_beta[cc] = (float)a;
            grad_gamma[cc] = (float)d;
        }
    }
}

// ---- backward 3: d/dQ[b,c,i] = sum_j dy_ij,  dy_ij = rstd (gamma dz_i [j == jsel] - s1 - yhat_ij s2)
//   = rstd (a_i - k s1 - s2 rstd (ysum_i - k mean)): element-wise over rows that backward 4 stages anyway, so it is written
//   there (one more read, one write in the staging loop instead of a pass that reloads ysel and grad_out).

// ---- reverse index of idx: pairs (i, j) grouped by (batch, target n) ------------------------------------------
__global__ __launch_bounds__(256) void edge_rix_count_kernel(long long total, long long per_batch, int nk,
                                                             const int *__restrict__ idx, int *__restrict__ cnt,
                                                             int *__restrict__ rank)
{
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= total) return;
    const int bi = (int)(x / per_batch);
    rank[x] = atomicAdd(&cnt[(size_t)bi * nk + idx[x]], 1);
}
__global__ __launch_bounds__(256) void edge_rix_fill_kernel(long long total, long long per_batch, int nk,
                                                            const int *__restrict__ idx, const int *__restrict__ off,
                                                            const int *__restrict__ rank, int *__restrict__ rev)
{
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= total) return;
    const int bi = (int)(x / per_batch);
    rev[off[(size_t)bi * nk + idx[x]] + rank[x]] = (int)(x - (long long)bi * per_batch); // pair id i*k + j within the batch
}
// reproducible order: the pair ids of a target ascending (geot_common.h rix_sorted_position); tmp = the first fill
__global__ __launch_bounds__(256) void edge_rix_place_kernel(long long total, long long per_batch, int nk,
                                                             const int *__restrict__ idx, const int *__restrict__ off,
                                                             const int *__restrict__ rank, const int *__restrict__ tmp,
                                                             int *__restrict__ rev)
{
    const long long x = (long long)blockIdx.x * 256 + threadIdx.x;
    if (x >= total) return;
    const int bi = (int)(x / per_batch);
    const size_t tgt = (size_t)bi * nk + idx[x];
    const int a = off[tgt], z = off[tgt + 1], mine = (int)(x - (long long)bi * per_batch);
    rev[a + rix_sorted_position(tmp, a, z, mine, rank[x])] = mine;
}

// ---- backward 4: d/dP[b,c,n] = sum over the pairs (i,j) with idx[b,i,j] == n of dy_ij ---------------------------
//   = rstd ( sum_pairs ( a_i [j == jsel_i] + u_i ) - cnt_n (s1 + s2 rstd (P_n - mean)) ),
//     a_i = gamma dz_i,  u_i = - s2 rstd Q_i        (y_ij = P_n + Q_i)
// The rows (a, u) (one float2 per query) and jsel (bytes) of CH channels live in LDS, computed while they are staged.
// (Interleaving the channels -- [query][channel], one ds_read_b128 + one u16 per pair for two channels -- measured slower:
// the strided staging stores cost more than the wider reads save, and CH = 4 spills.  Knock-outs at 8 clouds, c = 512,
// Nq = 8192, Nk = 4096: of the gradient call's 393 us the walk is 105 -- 55 its pair-id loads, 50 its LDS reads -- the staging
// 100; first-8 instead of first-4 pair ids, 2 or 8 slots per group, conflict-free addresses for masked reads: all within
// noise.  profiles/r03_edge_sweep.txt.)
// Both phases are written for loads in flight, because one 147-KB workgroup owns the CU and nothing else hides its latency:
// * the staging loop issues EC_SU x 4 row loads before its first LDS store;
// * the walk gives every target 2^lg adjacent lanes (host: from the mean list length Nq k / Nk, so that a lane's share is
//   a few times EC_E pairs at most -- 512 sources under 4096 queries x 4 make lists of 32), takes EC_TG (target, lane) slots per thread at
//   a time, and loads for all of them the list bounds, then the first EC_E pair ids of each share UNCONDITIONALLY (index
//   clamped into the list, value masked), then does all LDS reads of the group; only a share longer than EC_E runs the
//   dependent loop for its tail.  The first group's bounds and pair ids are requested BEFORE the staging loop, so two of
//   the walk's three dependent round trips overlap the row loads.
// A target's sum has a fixed order (its lanes' shares ascending, then an xor butterfly over the lanes): reproducible; with
// lg = 0 it is the ascending order of the plain walk.
constexpr int EC_TG = 4, EC_E = 4, EC_SU = 4;

template <bool K4, int CH>
__global__ __launch_bounds__(EC_THREADS) void edge_bwd_p_kernel(
    int c, int nq, int nk, int k, int groups, int lg, float slope, const float *__restrict__ P, const float *__restrict__ Q,
    const float *__restrict__ ysel, const uint8_t *__restrict__ jsel, const float *__restrict__ gamma,
    const float *__restrict__ beta, const float *__restrict__ stats, const float *__restrict__ coef,
    const float *__restrict__ grad_out, const int *__restrict__ off, const int *__restrict__ rev,
    const float *__restrict__ ysum, float *__restrict__ grad_p, float *__restrict__ grad_q)
{
    extern __shared__ float ec_rows[]; // [CH][nq] (a, u) | [CH][nq] jsel bytes
    float2 *AU = reinterpret_cast<float2 *>(ec_rows);
    uint8_t *J = reinterpret_cast<uint8_t *>(AU + (size_t)CH * nq);
    const int bi = blockIdx.z, c0 = blockIdx.y * CH, nch = min(CH, c - c0);
    const int per = (nk + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * per, n1 = min(nk, n0 + per);
    const int *offb = off + (size_t)bi * nk;
    const int step = 1 << lg, slot_end = n1 << lg;
    int slot = (n0 << lg) + threadIdx.x;          // (target, lane) slots: target = slot >> lg, lane of the target = slot & (step - 1)

    int a0[EC_TG], a1[EC_TG], pid[EC_TG][EC_E];
    float pv[EC_TG][CH], cnt[EC_TG];
    auto request = [&](int first) {
#pragma unroll
        for (int t = 0; t < EC_TG; ++t) {
            const int sl = first + t * EC_THREADS;
            const bool live = sl < slot_end;
            const int n = live ? sl >> lg : max(n1 - 1, 0);
            const int o0 = offb[n], o1 = offb[n + 1];
            a0[t] = o0 + (sl & (step - 1));
            a1[t] = live ? o1 : a0[t];              // a dead slot is an empty share
            cnt[t] = (float)(o1 - o0);
#pragma unroll
            for (int l = 0; l < CH; ++l) pv[t][l] = l < nch ? P[((size_t)bi * c + c0 + l) * nk + n] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < EC_TG; ++t) {
#pragma unroll
            for (int e = 0; e < EC_E; ++e) pid[t][e] = rev[max(min(a0[t] + (e << lg), a1[t] - 1), 0)];
        }
    };
    if (n1 > n0) request(slot);

    float mean[CH], rstd[CH], s1[CH], s2[CH];
    const float kf = (float)k;
#pragma unroll
    for (int l = 0; l < CH; ++l) {
        const int cc = min(c0 + l, c - 1), g = cc / (c / groups);
        mean[l] = stats[2 * (bi * groups + g)];
        rstd[l] = stats[2 * (bi * groups + g) + 1];
        s1[l] = coef[2 * (bi * groups + g)];
        s2[l] = coef[2 * (bi * groups + g) + 1];
    }
#pragma unroll
    for (int l = 0; l < CH; ++l) {
        if (l < nch) {
            const float gm = gamma[c0 + l], bt = beta[c0 + l], us = -s2[l] * rstd[l];
            const float kmean = kf * mean[l], ks1 = kf * s1[l];
            const size_t base = ((size_t)bi * c + c0 + l) * nq;
            for (int i0 = threadIdx.x; i0 < nq; i0 += EC_SU * EC_THREADS) {
                float ys[EC_SU], go[EC_SU], qq[EC_SU], sm[EC_SU];
                uint8_t jj[EC_SU];
#pragma unroll
                for (int u = 0; u < EC_SU; ++u) {
                    const int i = min(i0 + u * EC_THREADS, nq - 1);
                    ys[u] = ysel[base + i];
                    go[u] = grad_out[base + i];
                    qq[u] = Q[base + i];
                    jj[u] = jsel[base + i];
                    sm[u] = ysum[base + i];
                }
#pragma unroll
                for (int u = 0; u < EC_SU; ++u) {
                    const int i = i0 + u * EC_THREADS;
                    if (i < nq) {
                        const float yh = (ys[u] - mean[l]) * rstd[l];
                        const float z = fmaf(gm, yh, bt);
                        const float a = gm * (go[u] * (z > 0.f ? 1.f : slope));
                        AU[(size_t)l * nq + i] = make_float2(a, us * qq[u]);
                        J[(size_t)l * nq + i] = jj[u];
                        // d/dQ[b,c,i] = sum_j dy_ij = rstd (a_i - k s1 - s2 sum_j yhat_ij): the rows are here already
                        if (blockIdx.x == 0) grad_q[base + i] = rstd[l] * (a - ks1 - s2[l] * ((sm[u] - kmean) * rstd[l]));
                    }
                }
            }
        }
    }
    __syncthreads();
    if (n1 <= n0) return;
    while (true) {
        float acc[EC_TG][CH];
#pragma unroll
        for (int t = 0; t < EC_TG; ++t) {
#pragma unroll
            for (int l = 0; l < CH; ++l) acc[t][l] = 0.f;
#pragma unroll
            for (int e = 0; e < EC_E; ++e) {
                const int p = pid[t][e];
                const bool in = a0[t] + (e << lg) < a1[t];
                const int i = K4 ? (p >> 2) : (p / k), j = K4 ? (p & 3) : (p - i * k);
#pragma unroll
                for (int l = 0; l < CH; ++l) {
                    if (l < nch) {
                        const float2 au = AU[(size_t)l * nq + i];
                        const float v = (J[(size_t)l * nq + i] == (uint8_t)j ? au.x : 0.f) + au.y;
                        acc[t][l] = in ? acc[t][l] + v : acc[t][l];
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < EC_TG; ++t) {
            for (int e = a0[t] + (EC_E << lg); e < a1[t]; e += step) {          // the tail of a long share
                const int p = rev[e];
                const int i = K4 ? (p >> 2) : (p / k), j = K4 ? (p & 3) : (p - i * k);
#pragma unroll
                for (int l = 0; l < CH; ++l) {
                    if (l < nch) {
                        const float2 au = AU[(size_t)l * nq + i];
                        acc[t][l] += (J[(size_t)l * nq + i] == (uint8_t)j ? au.x : 0.f) + au.y;
                    }
                }
            }
            // the lanes of one target are adjacent and all of them are here (same target, same liveness)
            for (int d = 1; d < step; d <<= 1) {
#pragma unroll
                for (int l = 0; l < CH; ++l) acc[t][l] += __shfl_xor(acc[t][l], d);
            }
            const int sl = slot + t * EC_THREADS;
            if (sl < slot_end && (sl & (step - 1)) == 0) {
                const int n = sl >> lg;
#pragma unroll
                for (int l = 0; l < CH; ++l) {
                    if (l < nch)
                        grad_p[((size_t)bi * c + c0 + l) * nk + n] =
                            rstd[l] * (acc[t][l] - cnt[t] * (s1[l] + s2[l] * rstd[l] * (pv[t][l] - mean[l])));
                }
            }
        }
        slot += EC_TG * EC_THREADS;
        if (slot >= slot_end) break;
        request(slot);
    }
}

// ---- host side --------------------------------------------------------------------------------------------------
// channels per forward workgroup: rows of P within 64 KB of LDS so that two workgroups share a CU (one's staging under the
// other's streaming) and at most 4 (measured at 8 clouds, Nk = 512 / 4096 / 8192: 8 channels = 131 KB, one workgroup per
// CU, 492 us per step; <= 4: 466; <= 2: 448 -- profiles/r03_edge_sweep.txt); longer rows take what the CU's LDS holds
static int ec_fwd_ch(int nk)
{
    int fit = 64 * 1024 / ((int)sizeof(float) * nk);
    if (fit < 1) fit = EC_LDS_BYTES / ((int)sizeof(float) * nk);
    if (fit > 4) fit = 4;
    return fit >= 8 ? 8 : (fit >= 4 ? 4 : (fit >= 2 ? 2 : (fit >= 1 ? 1 : 0)));
}
static int ec_bwd_ch(int nq)
{
    int fit = EC_LDS_BYTES / (9 * nq);
    if (fit > 4) fit = 4;
    return fit >= 4 ? 4 : (fit >= 2 ? 2 : (fit >= 1 ? 1 : 0));
}
// lanes per target in the dP walk (log2): a lane's share of the mean list is about EC_E pairs
static int ec_lanes_log2(int nq, int nk, int k)
{
    const long long mean_len = ((long long)nq * k + nk - 1) / nk;
    int lg = 0;
    while (lg < 3 && (long long)(3 * EC_E) << lg < mean_len) ++lg;      // measured: a mean of 8 is best left to one lane
    return lg;
}
static int ec_slices(int b, int c, int ch, int n)
{
    const long long chunks = ((long long)c + ch - 1) / ch * b;
    long long sl = (512 + chunks - 1) / chunks;
    const long long maxs = n / 2048;
    if (sl > maxs) sl = maxs;
    if (sl < 1) sl = 1;
    if (sl > 32) sl = 32;
    return (int)sl;
}
static bool ec_ok(int b, int c, int nq, int nk, int k, int groups)
{
    return b >= 1 && c >= 1 && nq >= 1 && nk >= 1 && k >= 1 && k <= 255 && groups >= 1 && c % groups == 0 &&
           b <= 65535 && c <= 65535 * 8 && ec_fwd_ch(nk) >= 1 && ec_bwd_ch(nq) >= 1 &&
           (long long)b * nq * k <= 0x7fffffffLL && (long long)nq * k <= 0x7fffffffLL;
}
template <typename K>
static hipError_t ec_allow_lds(K kernel, size_t lds)
{
    return allow_big_lds((const void *)kernel, lds);
}

} // namespace geot

using namespace geot;

GEOT_EXPORT int geot_edgeconv_eligible(int b, int c, int nq, int nk, int k, int groups)
{
    return ec_ok(b, c, nq, nk, k, groups) ? 1 : 0;
}

// bytes of scratch both directions need (partials; reverse index)
GEOT_EXPORT long long geot_edgeconv_ws_bytes(int b, int c, int nq, int nk, int k)
{
    const long long part = (long long)b * c * 32 * 2 * (long long)sizeof(float);
    const long long t = (long long)b * nk, pairs = (long long)b * nq * k;
    const long long rix = ((t + 1) + scan_blocks(t) + 3 * pairs + 8) * (long long)sizeof(int);   // counts, scan, rank, rev, pair ids
    return part + rix + 256;
}

GEOT_EXPORT int geot_edgeconv_gn_max(int b, int c, int nq, int nk, int k, int groups, float eps, float slope,
                                     const float *P, const float *Q, const int *idx, const float *gamma,
                                     const float *beta, float *out, float *ysel, float *ysum, unsigned char *jsel,
                                     float *stats, void *workspace, long long ws_bytes, void *stream)
{
    if (!ec_ok(b, c, nq, nk, k, groups) || ws_bytes < geot_edgeconv_ws_bytes(b, c, nq, nk, k)) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    float *partial = (float *)workspace;
    const int ch = ec_fwd_ch(nk), slices = ec_slices(b, c, ch, nq);
    const size_t lds = (size_t)ch * nk * sizeof(float);
    const dim3 grid(slices, (c + ch - 1) / ch, b);
    const bool k4 = k == 4;
    hipError_t e = hipSuccess;
#define GEOT_EC_FWD(KV, CHV)                                                                                       \
    {                                                                                                              \
        e = ec_allow_lds(edge_fwd_kernel<KV, CHV>, lds);                                                           \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((edge_fwd_kernel<KV, CHV>), grid, dim3(EC_THREADS), lds, s, c, nq, nk, k, P, Q, idx,    \
                           gamma, ysel, ysum, jsel, partial);                                                      \
    }
    if (k4) {
        if (ch == 8) GEOT_EC_FWD(true, 8) else if (ch == 4) GEOT_EC_FWD(true, 4) else if (ch == 2) GEOT_EC_FWD(true, 2) else GEOT_EC_FWD(true, 1)
    } else {
        if (ch == 8) GEOT_EC_FWD(false, 8) else if (ch == 4) GEOT_EC_FWD(false, 4) else if (ch == 2) GEOT_EC_FWD(false, 2) else GEOT_EC_FWD(false, 1)
    }
#undef GEOT_EC_FWD
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const double count = (double)(c / groups) * nq * k;
    hipLaunchKernelGGL(edge_stats_kernel, dim3(b * groups), dim3(256), 0, s, c, groups, slices, count, eps, partial, stats);
    int gx = (nq + 2047) / 2048;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(edge_out_kernel, dim3(gx, c, b), dim3(256), 0, s, c, nq, groups, slope, ysel, gamma, beta, stats, out);
    return hipGetLastError();
}

// The reverse index of idx (pairs (i, j) grouped by (batch, target), ascending pair id inside a list) depends on the kNN
// graph alone: a caller that has the graph early (the model's index plan: side stream / look-ahead) builds it there with
// geot_edgeconv_rix_build and hands it to geot_edgeconv_gn_max_grad_rix; geot_edgeconv_gn_max_grad builds it in its
// workspace on the gradient's own stream (7 small launches per layer on the critical path of the backward).
struct EcRix {
    long long t, pairs, off, bsum, rank, rev, tmp, ints;
};
static inline EcRix ec_rix_layout(int b, int nq, int nk, int k)
{
    EcRix r;
    r.t = (long long)b * nk;
    r.pairs = (long long)b * nq * k;
    r.off = 0;
    r.bsum = r.t + 1;
    r.rank = r.bsum + scan_blocks(r.t);
    r.rev = r.rank + r.pairs;
    r.tmp = r.rev + r.pairs;
    r.ints = r.tmp + r.pairs + 8;
    return r;
}
static hipError_t ec_rix_build(int b, int nq, int nk, int k, const int *idx, int *ws, hipStream_t s)
{
    const EcRix r = ec_rix_layout(b, nq, nk, k);
    int *off = ws + r.off, *rank = ws + r.rank, *rev = ws + r.rev;
    hipError_t e = zero_words(off, r.t + 1, s);
    if (e != hipSuccess) return e;
    const int pb = (int)((r.pairs + 255) / 256);
    hipLaunchKernelGGL(edge_rix_count_kernel, dim3(pb), dim3(256), 0, s, r.pairs, (long long)nq * k, nk, idx, off, rank);
    exclusive_scan_i32((int)r.t, off, ws + r.bsum, nullptr, s);
    if (rix_reproducible()) {   // fill pair ids in arrival order, then place them in ascending order: fixed summation order
        int *tmp = ws + r.tmp;
        hipLaunchKernelGGL(edge_rix_fill_kernel, dim3(pb), dim3(256), 0, s, r.pairs, (long long)nq * k, nk, idx, off, rank, tmp);
        hipLaunchKernelGGL(edge_rix_place_kernel, dim3(pb), dim3(256), 0, s, r.pairs, (long long)nq * k, nk, idx, off, rank, tmp, rev);
    } else {
        hipLaunchKernelGGL(edge_rix_fill_kernel, dim3(pb), dim3(256), 0, s, r.pairs, (long long)nq * k, nk, idx, off, rank, rev);
    }
    return hipGetLastError();
}

GEOT_EXPORT long long geot_edgeconv_rix_ints(int b, int nq, int nk, int k)
{
    if (b < 1 || nq < 1 || nk < 1 || k < 1) return 0;
    return ec_rix_layout(b, nq, nk, k).ints;
}

GEOT_EXPORT int geot_edgeconv_rix_build(int b, int nq, int nk, int k, const int *idx, int *rix, long long rix_ints, void *stream)
{
    if (b < 1 || nq < 1 || nk < 1 || k < 1 || !idx || !rix || rix_ints < ec_rix_layout(b, nq, nk, k).ints ||
        (long long)b * nq * k > 0x7ffffff0LL || (long long)b * nk > 0x7ffffff0LL)
        return hipErrorInvalidValue;
    return ec_rix_build(b, nq, nk, k, idx, rix, (hipStream_t)stream);
}

static int ec_grad(int b, int c, int nq, int nk, int k, int groups, float slope, const float *P, const float *Q, const int *idx,
                   const float *gamma, const float *beta, const float *ysel, const float *ysum, const unsigned char *jsel,
                   const float *stats, const float *grad_out, float *grad_p, float *grad_q, float *grad_gamma, float *grad_beta,
                   void *workspace, long long ws_bytes, const int *rix, hipStream_t s)
{
    if (!ec_ok(b, c, nq, nk, k, groups) || ws_bytes < geot_edgeconv_ws_bytes(b, c, nq, nk, k)) return hipErrorInvalidValue;
    float *bpart = (float *)workspace;
    float *coef = bpart + (size_t)b * c * 32 * 2 - (size_t)b * groups * 2 - 8; // tail of the partials area (slices <= 16 used)
    int slices = ec_slices(b, c, 1, nq);
    if (slices > 16) slices = 16;
    const EcRix r = ec_rix_layout(b, nq, nk, k);
    hipError_t e;
    if (!rix) {                 // no index given: build it in the workspace, on this stream
        int *own = (int *)((char *)workspace + (size_t)b * c * 32 * 2 * sizeof(float));
        e = ec_rix_build(b, nq, nk, k, idx, own, s);
        if (e != hipSuccess) return e;
        rix = own;
    }
    const int *off = rix + r.off, *rev = rix + r.rev;

    hipLaunchKernelGGL(edge_bwd_reduce_kernel, dim3(slices, c, b), dim3(256), 0, s, c, nq, groups, slope, ysel, gamma, beta,
                       stats, grad_out, bpart);
    const double count = (double)(c / groups) * nq * k;
    hipLaunchKernelGGL(edge_bwd_coef_kernel, dim3(b * groups + (c + 255) / 256), dim3(256), 0, s, b, c, groups, slices, count,
                       gamma, bpart, coef, grad_gamma, grad_beta);
    const int ch = ec_bwd_ch(nq), pslices = ec_slices(b, c, ch, nk);
    const size_t lds = (size_t)ch * nq * 9;
    const dim3 grid(pslices, (c + ch - 1) / ch, b);
    const bool k4 = k == 4;
    const int lg = ec_lanes_log2(nq, nk, k);
#define GEOT_EC_BWD(KV, CHV)                                                                                       \
    {                                                                                                              \
        e = ec_allow_lds(edge_bwd_p_kernel<KV, CHV>, lds);                                                         \
        if (e != hipSuccess) return e;                                                                             \
        hipLaunchKernelGGL((edge_bwd_p_kernel<KV, CHV>), grid, dim3(EC_THREADS), lds, s, c, nq, nk, k, groups, lg, slope, \
                           P, Q, ysel, jsel, gamma, beta, stats, coef, grad_out, off, rev, ysum, grad_p, grad_q);   \
    }
    if (k4) {
        if (ch == 4) GEOT_EC_BWD(true, 4) else if (ch == 2) GEOT_EC_BWD(true, 2) else GEOT_EC_BWD(true, 1)
    } else {
        if (ch == 4) GEOT_EC_BWD(false, 4) else if (ch == 2) GEOT_EC_BWD(false, 2) else GEOT_EC_BWD(false, 1)
    }
#undef GEOT_EC_BWD
    return hipGetLastError();
}

GEOT_EXPORT int geot_edgeconv_gn_max_grad(int b, int c, int nq, int nk, int k, int groups, float slope, const float *P,
                                          const float *Q, const int *idx, const float *gamma, const float *beta,
                                          const float *ysel, const float *ysum, const unsigned char *jsel,
                                          const float *stats, const float *grad_out, float *grad_p, float *grad_q,
                                          float *grad_gamma, float *grad_beta, void *workspace, long long ws_bytes,
                                          void *stream)
{
    return ec_grad(b, c, nq, nk, k, groups, slope, P, Q, idx, gamma, beta, ysel, ysum, jsel, stats, grad_out, grad_p, grad_q,
                   grad_gamma, grad_beta, workspace, ws_bytes, nullptr, (hipStream_t)stream);
}

// as above with the reverse index of idx built beforehand (geot_edgeconv_rix_build, same b, nq, nk, k, idx)
GEOT_EXPORT int geot_edgeconv_gn_max_grad_rix(int b, int c, int nq, int nk, int k, int groups, float slope, const float *P,
                                              const float *Q, const int *rix, const float *gamma, const float *beta,
                                              const float *ysel, const float *ysum, const unsigned char *jsel,
                                              const float *stats, const float *grad_out, float *grad_p, float *grad_q,
                                              float *grad_gamma, float *grad_beta, void *workspace, long long ws_bytes,
                                              void *stream)
{
    if (!rix) return hipErrorInvalidValue;
    return ec_grad(b, c, nq, nk, k, groups, slope, P, Q, nullptr, gamma, beta, ysel, ysum, jsel, stats, grad_out, grad_p, grad_q,
                   grad_gamma, grad_beta, workspace, ws_bytes, rix, (hipStream_t)stream);
}
