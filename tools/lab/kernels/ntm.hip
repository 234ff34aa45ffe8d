// ntm.hip -- per-point noise-transition-matrix (NTM) block for gfx950 (MI355X).
//
// Replaces the Python/torch chains of the reference (behaviour, not code):
//   sig_t_mean.forward           openpoints/models/backbone/transformer.py:1120-1131
//       17 Linear(34->17) calls + cat/repeat temporaries + clamp + L1-normalise
//   logit correction             examples/segmentation/train.py:549-552
//       newT = normalize(lam*ema_t_corr + (1-lam)*insT); bmm((BN,1,C),(BN,C,C))
//   threeD_space_loss.forward    utils/insT_loss.py:68-110
//       32 index_select/cat rounds building (BN,32,289) twice, then the weighted distance
//
// All of these stream the (B*N, C, C) per-point matrices (1156 B per point at C=17): they are
// HBM / L2-bandwidth bound.  Every kernel moves those rows as whole contiguous tiles
// (64 points x C*C floats = 74 KB through LDS, or one 1156-B row per wave-instruction group)
// and fuses everything else, so each matrix is read or written exactly once per kernel.
// C is a template parameter (the reference fixes num_classes = 17).
#include "geot_common.h"
#include "geot_hip.h"
#include "ntm_generic.h"
#include <cstdlib>

namespace geot {

constexpr int NTM_THREADS = 256;
constexpr int NTM_TILE = 32; // points per LDS tile (32: 48-58 KB of LDS per block => 2-3 blocks per CU; 64 allowed one)
constexpr int NTM_GROUPS = NTM_THREADS / NTM_TILE; // row groups: thread = (point, group)
constexpr int NTM_TILE_SHIFT = 5;
static_assert((1 << NTM_TILE_SHIFT) == NTM_TILE, "tile shift");

template <int C>
struct NtmLds {
    static constexpr int CC = C * C;
    static constexpr int TILE_FLOATS = NTM_TILE * CC;
    static constexpr int STRIDE = CC | 1; // odd row stride: conflict-free per-point access
};

// Cooperative, fully coalesced copy of a [cnt][CC] tile between global (contiguous) and LDS
// (row stride STRIDE).
template <int C, bool TO_LDS>
__device__ __forceinline__ void ntm_tile_copy(float *__restrict__ g, float *__restrict__ lds, int cnt)
{
    constexpr int CC = C * C, STRIDE = NtmLds<C>::STRIDE;
    const int total = cnt * CC;
    if constexpr (STRIDE == CC) {
        // odd C: the LDS tile is the exact image of the global block -> 16-byte vectors, no index arithmetic
        // (tiles start at multiples of NTM_TILE points = multiples of 16 bytes)
        const int vec = total >> 2;
        for (int e = threadIdx.x; e < vec; e += NTM_THREADS) {
            if (TO_LDS) reinterpret_cast<float4 *>(lds)[e] = reinterpret_cast<const float4 *>(g)[e];
            else reinterpret_cast<float4 *>(g)[e] = reinterpret_cast<const float4 *>(lds)[e];
        }
        for (int e = (vec << 2) + threadIdx.x; e < total; e += NTM_THREADS) {
            if (TO_LDS) lds[e] = g[e];
            else g[e] = lds[e];
        }
    } else {
        for (int e = threadIdx.x; e < total; e += NTM_THREADS) {
            int pt = e / CC, w = e - pt * CC;
            if (TO_LDS) lds[pt * STRIDE + w] = g[e];
            else g[e] = lds[pt * STRIDE + w];
        }
    }
}

// ---- sig_t_mean forward -------------------------------------------------------------------
// raw[kk][o] = sum_j p_j W[kk][o][j] + sum_j cm[kk][j] W[kk][o][C+j]; clamp; L1-normalise over o.
template <int C, bool BACKWARD>
__global__ __launch_bounds__(NTM_THREADS) void sig_t_mean_kernel(
    int total_pts, int n, const float *__restrict__ p, const float *__restrict__ W,
    const float *__restrict__ cm, const float *__restrict__ grad_out, float *__restrict__ out)
{
    constexpr int CC = C * C, STRIDE = NtmLds<C>::STRIDE;
    extern __shared__ float ntm_lds[];
    float *Wa = ntm_lds;            // [kk][o][j]
    float *bias = Wa + C * CC;      // [kk][o]
    float *tile = ntm_lds + ((C * CC + CC + 3) & ~3); // [NTM_TILE][STRIDE], 16-byte aligned
    for (int e = threadIdx.x; e < C * CC; e += NTM_THREADS) {
        int kk = e / CC, r = e - kk * CC, o = r / C, j = r - o * C;
        Wa[e] = W[((size_t)kk * C + o) * 2 * C + j];
    }
    for (int e = threadIdx.x; e < CC; e += NTM_THREADS) {
        int kk = e / C, o = e - kk * C;
        float acc = 0.f;
        for (int j = 0; j < C; ++j) acc += cm[kk * C + j] * W[((size_t)kk * C + o) * 2 * C + C + j];
        bias[e] = acc;
    }
    __syncthreads();
    const int pt = threadIdx.x & (NTM_TILE - 1), grp = threadIdx.x >> NTM_TILE_SHIFT;
    for (int i0 = blockIdx.x * NTM_TILE; i0 < total_pts; i0 += gridDim.x * NTM_TILE) {
        const int cnt = min(NTM_TILE, total_pts - i0);
        if (BACKWARD) {
            ntm_tile_copy<C, true>(const_cast<float *>(grad_out) + (size_t)i0 * CC, tile, cnt);
            __syncthreads();
        }
        if (pt < cnt) {
            const int i = i0 + pt, b = i / n, ni = i - b * n;
            float pv[C];
#pragma unroll
            for (int j = 0; j < C; ++j) pv[j] = p[((size_t)b * C + j) * n + ni];
            for (int kk = grp; kk < C; kk += NTM_GROUPS) {
                float raw[C], s = 0.f;
#pragma unroll
                for (int o = 0; o < C; ++o) {
                    float acc = bias[kk * C + o];
#pragma unroll
                    for (int j = 0; j < C; ++j) acc = fmaf(pv[j], Wa[(kk * C + o) * C + j], acc);
                    raw[o] = acc;
                    s += fminf(fmaxf(acc, 1e-5f), 1.f - 1e-5f); // clamped values are positive
                }
                const float den = fmaxf(s, 1e-12f);
                float *row = tile + pt * STRIDE + kk * C;
                if (!BACKWARD) {
#pragma unroll
                    for (int o = 0; o < C; ++o) row[o] = fminf(fmaxf(raw[o], 1e-5f), 1.f - 1e-5f) / den;
                } else {
                    // d raw = [raw inside the clamp] * (g - sum_o g*tn) / den
                    float dot = 0.f;
#pragma unroll
                    for (int o = 0; o < C; ++o) dot += row[o] * (fminf(fmaxf(raw[o], 1e-5f), 1.f - 1e-5f) / den);
#pragma unroll
                    for (int o = 0; o < C; ++o) {
                        bool inside = raw[o] >= 1e-5f && raw[o] <= 1.f - 1e-5f;
                        row[o] = inside ? (row[o] - dot) / den : 0.f;
                    }
                }
            }
        }
        __syncthreads();
        ntm_tile_copy<C, false>(out + (size_t)i0 * CC, tile, cnt);
        __syncthreads();
    }
}

// ---- sig_t_mean on the matrix cores ---------------------------------------------------------
// The 17 Linear(34 -> 17) heads are one GEMM per tile of 32 points:
//     raw[pt][col] = sum_k A[pt][k] * Wt[k][col],   col = kk*17 + o (289, padded to 320),
//     A[pt] = (p_0 .. p_16, 1),  Wt[k<17][col] = W[kk][o][k],  Wt[17][col] = sum_j cm[kk][j] W[kk][o][17+j]
// i.e. K = 18 = 9 steps of v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation), 10 column
// tiles shared by the 4 waves of a block.  The 32 x 289 result lands in an LDS tile that is the exact
// image of the output block, rows are clamped + L1-normalised there by (point, row) threads, and the
// tile leaves as 16-byte vectors: the kernel is a pure stream of the (B*N, 17, 17) output.
// BACKWARD never materialises d raw: after recomputing raw it forms d raw in the tile and feeds it
// straight into a second GEMM  G[k][col] += sum_pt A[pt][k] * draw[pt][col]  (K = the 32 points), whose
// per-block partial sums are reduced by sig_t_mean_wgrad_reduce_kernel into the (17, 17, 34) weight
// gradient (columns 17..33 of a head see the constant cm row: G[17][col] * cm[kk][j]).
typedef float ntm_f32x16 __attribute__((ext_vector_type(16)));
constexpr int SM_PTS = 32;

template <int C>
struct SigMfma {
    static constexpr int CC = C * C;
    static constexpr int KP = C + 1;               // 18: inputs + the constant 1 that carries the cm part
    static constexpr int NCT = (CC + 31) / 32;     // 10 column tiles
    static constexpr int CP = NCT * 32;            // 320
    static constexpr int AT_STRIDE = KP + 1;       // 19 (odd)
    // the tile (+ the A^T copy and the slack its padded-column reads run into, backward only); the weights live in
    // registers, so four forward blocks (37 KB each) share a CU's 160 KB
    static constexpr int LDS_FLOATS_FWD = SM_PTS * CC;
    static constexpr int LDS_FLOATS_BWD = 2 * SM_PTS * CC + SM_PTS * AT_STRIDE + 64; // + the tile of incoming gradients
    static_assert(KP % 2 == 0, "K must be even for the 32x32x2 MFMA");
};

// The backward keeps the NEXT tile's incoming gradients in flight in registers through the whole tile (loads unconditional,
// indices clamped into the buffer, issued behind this tile's input loads so that the MFMAs wait with a counted vmcnt): with
// the copy at the top of the tile a block had one 37-KB tile in flight and waited for it.  183 -> 154 us at 8 clouds
// (1.5 TB/s).  What bounds it now is not HBM: per tile a wave issues 75 MFMAs of 64 cycles (the second GEMM uses 18 of its
// 32 rows), ~600 VALU instructions of row arithmetic in three passes (544 rows on 256 threads) and three barriers, at two
// waves per SIMD (229 registers, 77 KB of LDS per block).  512-thread blocks -- two column tiles per wave at most, the row
// phase in two passes -- measured SLOWER: 222 us at one block per CU (244 registers), 445 us held to 128 registers for two
// blocks per CU (181 spilled).  profiles/r03_sig_sweep.txt.
#ifndef GEOT_SIG_LAB_BWD_THREADS
#define GEOT_SIG_LAB_BWD_THREADS 256
#endif
constexpr int SIG_BWD_THREADS = GEOT_SIG_LAB_BWD_THREADS;
#ifndef GEOT_SIG_LAB_BWD_WPS
#define GEOT_SIG_LAB_BWD_WPS 2
#endif
constexpr int SIG_BWD_WPS = GEOT_SIG_LAB_BWD_WPS;

// GV float4 per thread of the gradient tile that starts at float4 index base4: loads unconditional, index clamped to the
// last whole float4 of the buffer (a clamped value is never used)
typedef float ntm_f32x4 __attribute__((ext_vector_type(4)));   // (HIP's float4 struct in an array stays in scratch memory)
template <int GV, int NT>
__device__ __forceinline__ void sig_fetch_tile(ntm_f32x4 (&pre)[GV], const float *__restrict__ grad_out, long long base4, long long last4,
                                               int tid)
{
    const ntm_f32x4 *src4 = reinterpret_cast<const ntm_f32x4 *>(grad_out);
#pragma unroll
    for (int v = 0; v < GV; ++v) {
        long long e = base4 + tid + v * NT;
        e = e < last4 ? e : last4;
        pre[v] = src4[e];
    }
}

template <int C, bool BACKWARD, int NT>
__global__ __launch_bounds__(NT, BACKWARD ? SIG_BWD_WPS : 4) void sig_t_mean_mfma_kernel(   // second figure: waves per SIMD
    int total_pts, int n, const float *__restrict__ p, const float *__restrict__ W,
    const float *__restrict__ cm, const float *__restrict__ grad_out, float *__restrict__ out,
    float *__restrict__ partial)
{
    using S = SigMfma<C>;
    constexpr int CC = S::CC, KP = S::KP, NCT = S::NCT, ATS = S::AT_STRIDE;
    constexpr int NW = NT / 64;
    constexpr int MYCT = (NCT + NW - 1) / NW; // column tiles per wave (4 waves: 3, 3, 2, 2; 8 waves: 2, 2, 1 ...)
    constexpr int GV = (SM_PTS * CC / 4 + NT - 1) / NT; // float4 of a full gradient tile per thread
    extern __shared__ float ntm_lds[];
    float *tile = ntm_lds;            // [SM_PTS][CC]  (contiguous = the global layout of the block)
    float *At = tile + SM_PTS * CC;   // [SM_PTS][ATS] (+ slack: the padded columns of the last row read past `tile`); BACKWARD only
    float *gtile = At + SM_PTS * ATS + 64; // [SM_PTS][CC] incoming gradients of the tile, copied as 16-byte vectors; BACKWARD only
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // B fragments of this wave's column tiles, once per block: lane (r, h) holds Wt[2*ks + h][ct*32 + r]
    float bw[KP / 2][MYCT];
#pragma unroll
    for (int t = 0; t < MYCT; ++t) {
        const int col = (wave + NW * t) * 32 + r;
        const bool live = wave + NW * t < NCT && col < CC;
#pragma unroll
        for (int ks = 0; ks < KP / 2; ++ks) {
            const int k = 2 * ks + h;
            float v = 0.f;
            if (live) {
                if (k < C) v = W[(size_t)col * 2 * C + k];
                else {
                    const int kk = col / C;
                    for (int j = 0; j < C; ++j) v += cm[kk * C + j] * W[(size_t)col * 2 * C + C + j];
                }
            }
            bw[ks][t] = v;
        }
    }
    if (BACKWARD) {
        for (int e = tid; e < SM_PTS * ATS + 64; e += NT) At[e] = 0.f;
    }
    ntm_f32x16 gacc[MYCT];
#pragma unroll
    for (int t = 0; t < MYCT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) gacc[t][e] = 0.f;

    // BACKWARD: the gradient tile of the block's next full tile, in registers (a partial last tile is copied directly)
    ntm_f32x4 pre[GV];
    const long long g_last4 = BACKWARD ? ((long long)total_pts * CC) / 4 - 1 : 0;   // last whole float4 of grad_out (>= 0: CC >= 4)
    auto fetch = [&](long long i0) { sig_fetch_tile<GV, NT>(pre, grad_out, i0 * CC / 4, g_last4, tid); };   // i0 % 32 == 0: whole float4s
    bool have_pre = false;
    if (BACKWARD) {
        const long long first = (long long)blockIdx.x * SM_PTS;
        have_pre = first + SM_PTS <= total_pts;
#ifdef GEOT_SIG_LAB_NOPREFETCH
        have_pre = false;
#endif
        fetch(have_pre ? first : 0);
    }
    __syncthreads();

    for (int i0 = blockIdx.x * SM_PTS; i0 < total_pts; i0 += gridDim.x * SM_PTS) {
        const int cnt = min(SM_PTS, total_pts - i0);
        if (BACKWARD) {
            if (have_pre) {
#pragma unroll
                for (int v = 0; v < GV; ++v) {
                    const int e = tid + v * NT;
                    if (e < SM_PTS * CC / 4) reinterpret_cast<ntm_f32x4 *>(gtile)[e] = pre[v];
                }
            } else {
                // (17 scalar loads per (point, row) thread straight from global touched 32 cache lines per instruction)
                const float *src = grad_out + (size_t)i0 * CC; // 16-byte aligned: i0 is a multiple of 32
                const int total = cnt * CC, vec = total >> 2;
                for (int e = tid; e < vec; e += NT)
                    reinterpret_cast<float4 *>(gtile)[e] = reinterpret_cast<const float4 *>(src)[e];
                for (int e = (vec << 2) + tid; e < total; e += NT) gtile[e] = src[e];
            }
        }
        // A fragment: lane (r = point, h) holds A[r][2*ks + h]  (loads unconditional: index clamped, value selected)
        float a[KP / 2];
        {
            const bool ok = r < cnt;
            const int i = min(i0 + r, total_pts - 1);
            const int b = i / n, ni = i - b * n;
#pragma unroll
            for (int ks = 0; ks < KP / 2; ++ks) {
                const int k = 2 * ks + h;
                const float v = p[((size_t)b * C + min(k, C - 1)) * n + ni];
                a[ks] = !ok ? 0.f : (k < C ? v : 1.f);
            }
        }
        if (BACKWARD) {
            const long long next = (long long)i0 + (long long)gridDim.x * SM_PTS;
            have_pre = next + SM_PTS <= total_pts;
#ifdef GEOT_SIG_LAB_NOPREFETCH
            have_pre = false;
#else
            fetch(have_pre ? next : 0);                   // stays in flight until the top of the next tile
#endif
            if (wave == 0) {
#pragma unroll
                for (int ks = 0; ks < KP / 2; ++ks) At[r * ATS + 2 * ks + h] = a[ks];
            }
        }
        ntm_f32x16 acc[MYCT];
#pragma unroll
        for (int t = 0; t < MYCT; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < KP / 2; ++ks) {
#pragma unroll
            for (int t = 0; t < MYCT; ++t) {
                const int ct = wave + NW * t;
                if (ct < NCT) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ks], bw[ks][t], acc[t], 0, 0, 0);
            }
        }
        // D layout of the 32x32 tile: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * h
#pragma unroll
        for (int t = 0; t < MYCT; ++t) {
            const int col = (wave + NW * t) * 32 + r;
            if (wave + NW * t < NCT && col < CC) {
#pragma unroll
                for (int e = 0; e < 16; ++e) tile[((e & 3) + 8 * (e >> 2) + 4 * h) * CC + col] = acc[t][e];
            }
        }
        __syncthreads();
        // (point, row) threads: clamp + L1-normalise (forward) or d raw (backward), in place
        for (int rr = tid; rr < SM_PTS * C; rr += NT) {
            const int pt = rr & (SM_PTS - 1), kk = rr >> 5;
            float *row = tile + pt * CC + kk * C;
            float raw[C], s = 0.f;
#pragma unroll
            for (int o = 0; o < C; ++o) {
                raw[o] = row[o];
                s += fminf(fmaxf(raw[o], 1e-5f), 1.f - 1e-5f); // clamped values are positive
            }
            const float den = fmaxf(s, 1e-12f);
            if (!BACKWARD) {
                // one division per row: 17 IEEE divisions were as many VALU cycles as the tile's HBM time (1 ulp apart)
                const float rden = 1.f / den;
#pragma unroll
                for (int o = 0; o < C; ++o) row[o] = fminf(fmaxf(raw[o], 1e-5f), 1.f - 1e-5f) * rden;
            } else if (pt < cnt) {
                const float *g = gtile + pt * CC + kk * C;
                const float rden = 1.f / den;
                float gv[C], dot = 0.f;
#pragma unroll
                for (int o = 0; o < C; ++o) {
                    gv[o] = g[o];
                    dot += gv[o] * (fminf(fmaxf(raw[o], 1e-5f), 1.f - 1e-5f) * rden);
                }
#pragma unroll
                for (int o = 0; o < C; ++o) {
                    const bool inside = raw[o] >= 1e-5f && raw[o] <= 1.f - 1e-5f;
                    row[o] = inside ? (gv[o] - dot) * rden : 0.f;
                }
            } else {
#pragma unroll
                for (int o = 0; o < C; ++o) row[o] = 0.f;
            }
        }
        __syncthreads();
        if (!BACKWARD) {
            float *dst = out + (size_t)i0 * CC; // 16-byte aligned: i0 is a multiple of 32
            const int total = cnt * CC, vec = total >> 2;
            for (int e = tid; e < vec; e += NT)
                reinterpret_cast<float4 *>(dst)[e] = reinterpret_cast<const float4 *>(tile)[e];
            for (int e = (vec << 2) + tid; e < total; e += NT) dst[e] = tile[e];
        } else {
            // G[k][col] += sum_pt A[pt][k] * draw[pt][col]:  M = k (rows r < 18 used), K = point
#pragma unroll 4
            for (int ks = 0; ks < SM_PTS / 2; ++ks) {
                const int pt = 2 * ks + h;
                const float av = r < KP ? At[pt * ATS + r] : 0.f;
#pragma unroll
                for (int t = 0; t < MYCT; ++t) {
                    const int ct = wave + NW * t;
                    if (ct < NCT) {
                        const float bv = tile[pt * CC + ct * 32 + r]; // cols >= 289: junk, discarded below
                        gacc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, gacc[t], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    if (BACKWARD) {
        float *P = partial + (size_t)blockIdx.x * KP * CC;
#pragma unroll
        for (int t = 0; t < MYCT; ++t) {
            const int col = (wave + NW * t) * 32 + r;
            if (wave + NW * t < NCT && col < CC) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (k < KP) P[k * CC + col] = gacc[t][e];
                }
            }
        }
    }
}

// grad_W[kk][o][j] += sum_blk partial[blk][j][col];  grad_W[kk][o][C+j] += cm[kk][j] * sum_blk partial[blk][C][col]
template <int C>
__global__ __launch_bounds__(1024) void sig_t_mean_wgrad_reduce_kernel(int nblk, const float *__restrict__ partial,
                                                                      const float *__restrict__ cm,
                                                                      float *__restrict__ grad_W)
{
    // one workgroup per 64 entries: its 16 waves take the block partials round-robin, each in ascending order, and add their
    // sums in a fixed tree; one writer per element of grad_W (pre-zeroed or carrying earlier gradients) -- the same bits on
    // every run (the first form sliced the blocks over grid.y and finished with float atomics: arrival order)
    constexpr int CC = C * C, KP = C + 1, PARTS = 16;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    __shared__ float red[PARTS][64];
    float s = 0.f;
    if (e < KP * CC)
        for (int bkt = part; bkt < nblk; bkt += PARTS) s += partial[(size_t)bkt * KP * CC + e];
    red[part][lane] = s;
    __syncthreads();
#pragma unroll
    for (int h = PARTS / 2; h >= 1; h >>= 1) {
        if (part < h) red[part][lane] += red[part + h][lane];
        __syncthreads();
    }
    if (part != 0 || e >= KP * CC) return;
    s = red[0][lane];
    const int k = e / CC, col = e - k * CC;
    if (k < C) grad_W[(size_t)col * 2 * C + k] += s;
    else {
        const int kk = col / C;
        for (int j = 0; j < C; ++j) grad_W[(size_t)col * 2 * C + C + j] += cm[kk * C + j] * s;
    }
}

// ---- logit correction ----------------------------------------------------------------------
// v = lam*E + (1-lam)*T_i; tn = v / max(sum_c |v|, eps); out[c] = sum_r logit[r] * tn[r][c].
// Backward: thread = (point, row group); rows grp, grp + 8, grp + 16 of the point's T_i (LDS tile, overwritten in
// place by d T_i and streamed back out), one reciprocal per row.
template <int C>
__global__ __launch_bounds__(NTM_THREADS, 4) void ntm_correct_bwd_kernel(
    int total_pts, int n, float lam, const float *__restrict__ logits, const float *__restrict__ insT,
    const float *__restrict__ E, const float *__restrict__ grad_out, float *__restrict__ grad_logits,
    float *__restrict__ grad_insT, float *__restrict__ grad_E, float *__restrict__ grad_E_partial)
{
    constexpr int CC = C * C, STRIDE = NtmLds<C>::STRIDE;
    extern __shared__ float ntm_lds[];
    float *El = ntm_lds;                         // [CC]
    float *tile = ntm_lds + ((CC + 3) & ~3);     // [NTM_TILE][STRIDE], 16-byte aligned
    for (int e = threadIdx.x; e < CC; e += NTM_THREADS) El[e] = E[e];
    const int pt = threadIdx.x & (NTM_TILE - 1), grp = threadIdx.x >> NTM_TILE_SHIFT;
    // this thread's share of grad_E (rows grp, grp + 8, grp + 16) stays in registers over all its tiles; 289 x points
    // LDS atomics on 289 addresses were the whole cost of the first version
    constexpr int ROWS_PT = (C + NTM_GROUPS - 1) / NTM_GROUPS;
    float eacc[ROWS_PT][C];
#pragma unroll
    for (int q = 0; q < ROWS_PT; ++q)
#pragma unroll
        for (int c = 0; c < C; ++c) eacc[q][c] = 0.f;
    for (int i0 = blockIdx.x * NTM_TILE; i0 < total_pts; i0 += gridDim.x * NTM_TILE) {
        const int cnt = min(NTM_TILE, total_pts - i0);
        ntm_tile_copy<C, true>(const_cast<float *>(insT) + (size_t)i0 * CC, tile, cnt);
        __syncthreads();
        if (pt < cnt) {
            const int i = i0 + pt, b = i / n, ni = i - b * n;
            float go[C];
#pragma unroll
            for (int c = 0; c < C; ++c) go[c] = grad_out[((size_t)b * C + c) * n + ni];
#pragma unroll
            for (int q = 0; q < ROWS_PT; ++q) {
                const int r = grp + q * NTM_GROUPS;
                if (r >= C) continue;
                float *row = tile + pt * STRIDE + r * C;
                const float l = logits[((size_t)b * C + r) * n + ni];
                float v[C], s = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    v[c] = lam * El[r * C + c] + (1.f - lam) * row[c];
                    s += fabsf(v[c]);
                }
                const float rden = 1.f / fmaxf(s, 1e-12f);
                float gl = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) gl += (v[c] * rden) * go[c];
                grad_logits[((size_t)b * C + r) * n + ni] = gl;
                const float dot = l * gl; // sum_c dtn*tn with dtn = l*go
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float sg = v[c] > 0.f ? 1.f : (v[c] < 0.f ? -1.f : 0.f);
                    float dv = s > 1e-12f ? (l * go[c] - sg * dot) * rden : l * go[c] * rden;
                    row[c] = (1.f - lam) * dv;
                    eacc[q][c] = fmaf(lam, dv, eacc[q][c]);
                }
            }
        }
        __syncthreads();
        ntm_tile_copy<C, false>(grad_insT + (size_t)i0 * CC, tile, cnt);
        __syncthreads();
    }
    // sum over the 32 points of the half-wave (one row group per half-wave), then one atomic per entry
#pragma unroll
    for (int q = 0; q < ROWS_PT; ++q) {
        const int r = grp + q * NTM_GROUPS;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float v = eacc[q][c];
#pragma unroll
            for (int o = NTM_TILE / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
            if (pt == 0 && r < C) {
                // per-block partials when the caller gave a workspace: 289 addresses shared by a thousand
                // blocks serialise in one or two L2 channels otherwise
                if (grad_E_partial) grad_E_partial[(size_t)blockIdx.x * CC + r * C + c] = v;
                else atomicAdd(grad_E + r * C + c, v);
            }
        }
    }
}

// Forward-only variant: 8 adjacent lanes per point (row group = lane & 7), the 8 partial outputs of a point are
// summed with DPP inside the wave -- no [groups][points][C] LDS buffer, so a block is the 37 KB tile + E and four of
// them share a CU (the two-block version stalled at 3.1 TB/s: the tile load of one block overlapped with the row
// arithmetic of only one other) -- and one division per row instead of 17 (the 17 IEEE divisions were as many
// VALU cycles as the tile's HBM time; scale = l / den, out += scale * v is 1 ulp from l * (v / den)).
template <int C>
__global__ __launch_bounds__(NTM_THREADS) void ntm_correct_fwd_kernel(
    int total_pts, int n, float lam, const float *__restrict__ logits, const float *__restrict__ insT,
    const float *__restrict__ E, float *__restrict__ out)
{
    constexpr int CC = C * C, STRIDE = NtmLds<C>::STRIDE;
    constexpr int LPP = NTM_THREADS / NTM_TILE;  // 8 lanes per point
    constexpr int ROWS_PT = (C + LPP - 1) / LPP; // rows grp, grp + 8, grp + 16
    static_assert(LPP == 8, "the DPP sum below covers 8 adjacent lanes");
    extern __shared__ float ntm_lds[];
    float *El = ntm_lds;                       // [CC]
    float *tile = ntm_lds + ((CC + 3) & ~3);   // [NTM_TILE][STRIDE], 16-byte aligned
    for (int e = threadIdx.x; e < CC; e += NTM_THREADS) El[e] = E[e];
    const int grp = threadIdx.x & (LPP - 1), pt = threadIdx.x / LPP;
    for (int i0 = blockIdx.x * NTM_TILE; i0 < total_pts; i0 += gridDim.x * NTM_TILE) {
        const int cnt = min(NTM_TILE, total_pts - i0);
        ntm_tile_copy<C, true>(const_cast<float *>(insT) + (size_t)i0 * CC, tile, cnt);
        __syncthreads();
        const int i = i0 + pt, b = pt < cnt ? i / n : 0, ni = pt < cnt ? i - b * n : 0;
        float acc[C];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0.f;
        if (pt < cnt) {
            float l[ROWS_PT];
#pragma unroll
            for (int q = 0; q < ROWS_PT; ++q) {
                const int r = grp + q * LPP;
                l[q] = r < C ? logits[((size_t)b * C + r) * n + ni] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < ROWS_PT; ++q) {
                const int r = grp + q * LPP;
                if (r >= C) continue;
                const float *row = tile + pt * STRIDE + r * C;
                float v[C], s = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    v[c] = lam * El[r * C + c] + (1.f - lam) * row[c];
                    s += fabsf(v[c]);
                }
                const float scale = l[q] / fmaxf(s, 1e-12f);
#pragma unroll
                for (int c = 0; c < C; ++c) acc[c] = fmaf(scale, v[c], acc[c]);
            }
        }
        // sum over the 8 lanes of a point (all lanes take part: idle points carry zeros)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float v = acc[c];
            v += __uint_as_float(dpp_mov<DPP_QUAD_XOR1>(__float_as_uint(v)));
            v += __uint_as_float(dpp_mov<DPP_QUAD_XOR2>(__float_as_uint(v)));
            v += __uint_as_float(dpp_mov<DPP_ROW_HALF_MIRROR>(__float_as_uint(v)));
            acc[c] = v;
        }
        if (pt < cnt) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                if ((c & (LPP - 1)) == grp) out[((size_t)b * C + c) * n + ni] = acc[c];
        }
        __syncthreads(); // the tile is overwritten by the next copy
    }
}

// ---- threeD_space_loss -----------------------------------------------------------------------
// One wave per point i: T_i held across lanes (element e = lane + 64*r), neighbour rows T_j read as
// contiguous 4*CC-byte rows; w_ij = [label_i == label_j] * exp(-|p_i-p_j|^2 / (2 sigma^2)).
// forward : per_point[i] = sum_j w_ij |T_i - T_j|^2 / (sum_j w_ij + 1e-3)
// backward: grad_T[i] += coef_i * sum_j w_ij (T_i - T_j);  grad_T[j] -= coef_i * w_ij (T_i - T_j)
//           with coef_i = 2 * gscale / (sum_j w_ij + 1e-3)     (gscale = upstream grad / (B*N))
// SIGNED (feature_space_loss, insT_loss.py:9-58): w_ij = (label_i == label_j ? +1 : -1) * exp(..) over
// pd-dimensional features and the per-point sum is NOT normalised (the caller divides by B*N*k).
// R = registers per lane for one row of C*C floats (ceil(CC / 64)); CC itself is a run-time argument, so one
// instantiation serves a range of class counts (R = 5: up to 17 classes, 2: up to 11, 8: up to 22, 16: up to 32).
template <int R, bool BACKWARD, bool SIGNED>
__global__ __launch_bounds__(256) void threed_loss_kernel(
    int total_pts, int n, int k, int pd, int CC, float inv2s2, float gscale, const float *__restrict__ pos,
    const int *__restrict__ labels, const float *__restrict__ T, const int *__restrict__ nbr,
    const int *__restrict__ order, float *__restrict__ per_point, float *__restrict__ grad_T)
{
    const int lane = lane_id();
    // XCD-aware walk of the (spatially sorted) point order: workgroups are dealt round-robin to the 8 XCDs,
    // so XCD x takes the x-th eighth of the order and its private L2 caches one region of the cloud instead of
    // all eight L2s caching the same rows.  (Placement is a performance assumption only; any mapping is correct.)
    const int xcd_chunk = (((total_pts + 7) >> 3) + 3) & ~3;
    for (int t = blockIdx.x >> 3; ; t += gridDim.x >> 3) {
        const int within = t * 4 + (threadIdx.x >> 6);
        if (within >= xcd_chunk) break;
        const int ii = (blockIdx.x & 7) * xcd_chunk + within;
        if (ii >= total_pts) break;
        const int i = order ? order[ii] : ii; // spatial processing order: neighbour rows are then found in L2
        const int b = i / n;
        const float *Ti = T + (size_t)i * CC;
        float ti[R];
#pragma unroll
        for (int r = 0; r < R; ++r) ti[r] = (lane + 64 * r < CC) ? Ti[lane + 64 * r] : 0.f;
        const int li = labels[i];
        // lanes 0..k-1 evaluate the k weights in parallel
        int j = -1;
        float w = 0.f;
        if (lane < k) {
            j = b * n + nbr[(size_t)i * k + lane];
            const float *pi = pos + (size_t)i * pd, *pj = pos + (size_t)j * pd;
            float d2 = 0.f;
            for (int d = 0; d < pd; ++d) {
                float dx = pi[d] - pj[d];
                d2 += dx * dx;
            }
            const float e = __expf(-d2 * inv2s2);
            w = labels[j] == li ? e : (SIGNED ? -e : 0.f);
        }
        float S = 1.f;
        if (!SIGNED) {
            S = w;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) S += __shfl_xor(S, o);
            S += 0.001f;
        }
        unsigned long long live = __ballot(w != 0.f);
        float acc = 0.f, gi[R];
#pragma unroll
        for (int r = 0; r < R; ++r) gi[r] = 0.f;
        const float coef = BACKWARD ? 2.f * gscale / S : 0.f;
        while (live) {
            int l = __builtin_ctzll(live);
            live &= live - 1;
            const int jj = __builtin_amdgcn_readlane(j, l);
            const float wj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w), l));
            const float *Tj = T + (size_t)jj * CC;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int e = lane + 64 * r;
                if (e < CC) {
                    float d = ti[r] - Tj[e];
                    if (!BACKWARD) acc = fmaf(wj * d, d, acc);
                    else {
                        float c = coef * wj * d;
                        gi[r] += c;
                        atomicAdd(grad_T + (size_t)jj * CC + e, -c);
                    }
                }
            }
        }
        if (!BACKWARD) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0) per_point[i] = acc / S;
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                int e = lane + 64 * r;
                if (e < CC) atomicAdd(grad_T + (size_t)i * CC + e, gi[r]);
            }
        }
    }
}

// ---- threeD_space_loss forward, G consecutive points of the spatial order per wave ------------
// Points that follow each other in Morton order share most of their neighbours.  The wave keeps the G
// neighbour lists in lanes, and every neighbour row it loads is applied to ALL of its points that list it
// (membership = one ballot per point), so a row shared by 3 of the 4 points is fetched once, not 3 times.
// The row gathers from L2 are what bounds this kernel; the arithmetic per (point, row) pair is unchanged.
// Graph arrays a forward pass can leave behind for its backward (see geot_ntm_threed_loss_fwd_graph):
// fixed-capacity reverse adjacency (TLG_CAP in-edges per point, the rest in an overflow list).
constexpr int TLG_CAP = 64;
struct TlGraph {
    float *S;     // [t]      per-point normaliser
    float *wout;  // [t*k]    out-edge weights
    int *cnt;     // [t]      in-degree (may exceed TLG_CAP; zero on entry)
    int *rev;     // [t*CAP]  in-edge sources
    float *revc;  // [t*CAP]  in-edge coefficients w / S_source
    int *ovf_cnt; // [1]      entries in the overflow list (zero on entry)
    int *ovf;     // [t*k*3]  (target, source, bits(coefficient))
};

template <int CC, int G, bool BUILD>
__global__ __launch_bounds__(256) void threed_loss_shared_kernel(
    int total_pts, int n, int k, float inv2s2, const float *__restrict__ pos, const int *__restrict__ labels,
    const float *__restrict__ T, const int *__restrict__ nbr, const int *__restrict__ order,
    float *__restrict__ per_point, TlGraph gr)
{
    constexpr int R = (CC + 63) / 64;
    const int lane = lane_id();
    const int xcd_chunk = (((total_pts + 7) >> 3) + 4 * G - 1) / (4 * G) * (4 * G);
    for (int t = blockIdx.x >> 3;; t += gridDim.x >> 3) {
        const int within = (t * 4 + (threadIdx.x >> 6)) * G;
        if (within >= xcd_chunk) break;
        const int ii0 = (blockIdx.x & 7) * xcd_chunk + within;
        if (ii0 >= total_pts) break;
        int pi[G], jl[G];
        float wl[G], S[G], ti[G][R], acc[G];
        unsigned long long live[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool ok = ii0 + g < total_pts && within + g < xcd_chunk;
            const int i = ok ? (order ? order[ii0 + g] : ii0 + g) : -1;
            pi[g] = i;
            jl[g] = -1; wl[g] = 0.f; acc[g] = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) ti[g][r] = (i >= 0 && lane + 64 * r < CC) ? T[(size_t)i * CC + lane + 64 * r] : 0.f;
            if (i >= 0 && lane < k) {
                const int j = (i / n) * n + nbr[(size_t)i * k + lane];
                jl[g] = j;
                if (labels[j] == labels[i]) {
                    const float dx = pos[(size_t)i * 3] - pos[(size_t)j * 3], dy = pos[(size_t)i * 3 + 1] - pos[(size_t)j * 3 + 1];
                    const float dz = pos[(size_t)i * 3 + 2] - pos[(size_t)j * 3 + 2];
                    float d2 = 0.f;
                    d2 += dx * dx; d2 += dy * dy; d2 += dz * dz;
                    wl[g] = __expf(-d2 * inv2s2);
                }
            }
            float s = wl[g];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            S[g] = s + 0.001f;
            live[g] = __ballot(wl[g] != 0.f);
            if (BUILD && i >= 0) {
                // leave the graph behind: weights, normaliser, and this point entered as an in-neighbour of each
                // of its live out-neighbours (the atomics overlap with the row gathers below)
                if (lane == 0) gr.S[i] = S[g];
                if (lane < k) {
                    gr.wout[(size_t)i * k + lane] = wl[g];
                    if (wl[g] != 0.f) {
                        const int slot = atomicAdd(&gr.cnt[jl[g]], 1);
                        const float cf = wl[g] / S[g];
                        if (slot < TLG_CAP) {
                            gr.rev[(size_t)jl[g] * TLG_CAP + slot] = i;
                            gr.revc[(size_t)jl[g] * TLG_CAP + slot] = cf;
                        } else {
                            const int o = atomicAdd(gr.ovf_cnt, 1); // < t*k by construction: one entry per edge at most
                            gr.ovf[3 * (size_t)o] = jl[g];
                            gr.ovf[3 * (size_t)o + 1] = i;
                            gr.ovf[3 * (size_t)o + 2] = __float_as_int(cf);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            while (live[g]) {
                const int l = __builtin_ctzll(live[g]);
                const int jj = __builtin_amdgcn_readlane(jl[g], l);
                const float *Tj = T + (size_t)jj * CC;
                float tj[R];
#pragma unroll
                for (int r = 0; r < R; ++r) tj[r] = (lane + 64 * r < CC) ? Tj[lane + 64 * r] : 0.f;
#pragma unroll
                for (int g2 = g; g2 < G; ++g2) { // point g finds its own entry the same way
                    const unsigned long long m = __ballot(jl[g2] == jj) & live[g2];
                    if (m) {
                        const float wj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wl[g2]), __builtin_ctzll(m)));
                        live[g2] &= ~m;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float d = ti[g2][r] - tj[r];
                            acc[g2] = fmaf(wj * d, d, acc[g2]); // padded lanes: ti = tj = 0
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float a = acc[g];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0 && pi[g] >= 0) per_point[pi[g]] = a / S[g];
        }
    }
}

// ---- threeD_space_loss backward without atomics: gather over the graph and its reverse --------
//   grad_T[i] = sum_{j in N(i)} c_i w_ij (T_i - T_j)  +  sum_{j : i in N(j)} c_j w_ij (T_i - T_j),
//   c_x = 2 * gscale / (sum_l w_xl + 1e-3),  w symmetric in the positions and 0 across labels,
// so every row is written once by the wave that owns it.  (The scatter form pushes 289 float atomics
// per live edge: with spatially coherent labels -- real scans -- that is 1.8e9 atomics per 8 clouds.)
// Pre-pass: per-point normaliser S and in-degree of the live edges; exclusive scan; fill of the
// reverse adjacency (order within a list is arbitrary; only the fp32 summation order depends on it).
// Edge-parallel pre-pass: lane = one (point, slot) edge, SEG = next power of two >= k lanes per point.
__device__ __forceinline__ float tl_edge_weight(const float *__restrict__ pos, const int *__restrict__ labels,
                                                int i, int j, float inv2s2)
{
    if (labels[j] != labels[i]) return 0.f;
    const float dx = pos[(size_t)i * 3] - pos[(size_t)j * 3], dy = pos[(size_t)i * 3 + 1] - pos[(size_t)j * 3 + 1];
    const float dz = pos[(size_t)i * 3 + 2] - pos[(size_t)j * 3 + 2];
    float d2 = 0.f;
    d2 += dx * dx; d2 += dy * dy; d2 += dz * dz;
    return __expf(-d2 * inv2s2);
}

__global__ __launch_bounds__(256) void tl_prep_kernel(int total_pts, int n, int k, int seg_shift, float inv2s2,
                                                      const float *__restrict__ pos, const int *__restrict__ labels,
                                                      const int *__restrict__ nbr, const int *__restrict__ order,
                                                      float *__restrict__ S, float *__restrict__ wout,
                                                      int *__restrict__ deg)
{
    const int seg = 1 << seg_shift;
    const long long gl = (long long)blockIdx.x * 256 + threadIdx.x;
    const int ii = (int)(gl >> seg_shift), l = (int)(gl & (seg - 1));
    const int i = ii < total_pts ? (order ? order[ii] : ii) : total_pts;
    float w = 0.f;
    if (i < total_pts && l < k) {
        const int j = (i / n) * n + nbr[(size_t)i * k + l];
        w = tl_edge_weight(pos, labels, i, j, inv2s2);
        wout[(size_t)i * k + l] = w;
        if (w != 0.f) atomicAdd(&deg[j], 1);
    }
    float s = w;
    for (int o = 1; o < seg; o <<= 1) s += __shfl_xor(s, o);
    if (i < total_pts && l == 0) S[i] = s + 0.001f;
}

// reverse adjacency: the in-edge (s -> i) is stored with its finished coefficient w_si / S_s
__global__ __launch_bounds__(256) void tl_fill_kernel(int total_pts, int n, int k, int seg_shift,
                                                      const int *__restrict__ nbr, const float *__restrict__ wout,
                                                      const float *__restrict__ S, const int *__restrict__ off,
                                                      const int *__restrict__ order, int *__restrict__ cursor,
                                                      int *__restrict__ rev, float *__restrict__ revc)
{
    const int seg = 1 << seg_shift;
    const long long gl = (long long)blockIdx.x * 256 + threadIdx.x;
    const int ii = (int)(gl >> seg_shift), l = (int)(gl & (seg - 1));
    if (ii >= total_pts || l >= k) return;
    const int i = order ? order[ii] : ii;
    const float w = wout[(size_t)i * k + l];
    if (w == 0.f) return;
    const int j = (i / n) * n + nbr[(size_t)i * k + l];
    const int slot = off[j] + atomicAdd(&cursor[j], 1);
    rev[slot] = i;
    revc[slot] = w / S[i];
}

// wave-wide bitonic sort of (source id, coefficient) by id, ascending; lanes without an edge (id < 0) end up behind the others
__device__ __forceinline__ void tl_sort_edges(int &id, float &cf, int lane)
{
    unsigned key = id < 0 ? 0xffffffffu : (unsigned)id;
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            const unsigned pk = (unsigned)__shfl_xor((int)key, j);
            const float pv = __shfl_xor(cf, j);
            const bool keep_min = ((lane & k2) == 0) == ((lane & j) == 0);
            const bool take = keep_min ? pk < key : pk > key;
            key = take ? pk : key;
            cf = take ? pv : cf;
        }
    }
    id = key == 0xffffffffu ? -1 : (int)key;
}

// Gather backward, G consecutive points of the spatial order per wave, neighbour rows shared: a row is
// loaded once and applied to every point of the wave that has it as an out- or an in-neighbour (most kNN
// edges are mutual, so even a single point usually meets each neighbour twice).
// FIXED: the reverse lists are the fixed-capacity ones a forward pass left (off = in-degree counts, list of
// point i at rev[i * TLG_CAP ..]); otherwise the CSR built by geot_ntm_threed_loss_grad_ws.
template <int CC, int G, bool FIXED>
__global__ __launch_bounds__(256) void tl_grad_gather_shared_kernel(
    int total_pts, int n, int k, float gscale, const float *__restrict__ T, const int *__restrict__ nbr,
    const float *__restrict__ wout, const float *__restrict__ S, const int *__restrict__ off,
    const int *__restrict__ rev, const float *__restrict__ revc, const int *__restrict__ order,
    const float *__restrict__ upstream, float *__restrict__ grad_T)
{
    constexpr int R = (CC + 63) / 64;
    const int lane = lane_id();
    // upstream: optional device scalar (d loss_total / d this loss), so the host never has to read it back
    const float two_g = 2.f * gscale * (upstream ? upstream[0] : 1.f);
    const int xcd_chunk = (((total_pts + 7) >> 3) + 4 * G - 1) / (4 * G) * (4 * G);
    for (int t = blockIdx.x >> 3;; t += gridDim.x >> 3) {
        const int within = (t * 4 + (threadIdx.x >> 6)) * G;
        if (within >= xcd_chunk) break;
        const int ii0 = (blockIdx.x & 7) * xcd_chunk + within;
        if (ii0 >= total_pts) break;
        int pi[G], jo[G], ji[G], r0[G], nin[G];
        float co[G], ci[G], ti[G][R], acc[G][R];
        unsigned long long lo[G], li[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool ok = ii0 + g < total_pts && within + g < xcd_chunk;
            const int i = ok ? (order ? order[ii0 + g] : ii0 + g) : -1;
            pi[g] = i;
            jo[g] = -1; ji[g] = -1; co[g] = 0.f; ci[g] = 0.f; r0[g] = 0; nin[g] = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                ti[g][r] = (i >= 0 && lane + 64 * r < CC) ? T[(size_t)i * CC + lane + 64 * r] : 0.f;
                acc[g][r] = 0.f;
            }
            if (i >= 0) {
                if (FIXED) { r0[g] = i * TLG_CAP; nin[g] = min(off[i], TLG_CAP); }
                else { r0[g] = off[i]; nin[g] = off[i + 1] - r0[g]; }
                if (lane < k) {
                    jo[g] = (i / n) * n + nbr[(size_t)i * k + lane];
                    co[g] = two_g * (wout[(size_t)i * k + lane] / S[i]);
                }
                if (lane < nin[g]) { ji[g] = rev[r0[g] + lane]; ci[g] = two_g * revc[r0[g] + lane]; }
            }
            // the in-edge slots were handed out by an atomic (arrival order): put them in ascending source id, so that a
            // point's sum has the same order on every run (empty lanes -- id -1 -- go last)
            tl_sort_edges(ji[g], ci[g], lane);
            lo[g] = __ballot(co[g] != 0.f);
            li[g] = __ballot(ci[g] != 0.f);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            while (lo[g] | li[g]) {
                int jj;
                if (lo[g]) jj = __builtin_amdgcn_readlane(jo[g], __builtin_ctzll(lo[g]));
                else jj = __builtin_amdgcn_readlane(ji[g], __builtin_ctzll(li[g]));
                const float *Tj = T + (size_t)jj * CC;
                float tj[R];
#pragma unroll
                for (int r = 0; r < R; ++r) tj[r] = (lane + 64 * r < CC) ? Tj[lane + 64 * r] : 0.f;
#pragma unroll
                for (int g2 = g; g2 < G; ++g2) {
                    const unsigned long long mo = __ballot(jo[g2] == jj) & lo[g2];
                    const unsigned long long mi = __ballot(ji[g2] == jj) & li[g2];
                    if (mo | mi) {
                        float cf = 0.f;
                        if (mo) cf += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(co[g2]), __builtin_ctzll(mo)));
                        if (mi) cf += __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ci[g2]), __builtin_ctzll(mi)));
                        lo[g2] &= ~mo;
                        li[g2] &= ~mi;
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[g2][r] = fmaf(cf, ti[g2][r] - tj[r], acc[g2][r]);
                    }
                }
            }
            // in-edges beyond the 64 kept in lanes (rare): unshared
            for (int e0 = 64; e0 < nin[g]; e0 += 64) {
                int j = 0;
                float cf = 0.f;
                if (e0 + lane < nin[g]) { j = rev[r0[g] + e0 + lane]; cf = two_g * revc[r0[g] + e0 + lane]; }
                unsigned long long live = __ballot(cf != 0.f);
                while (live) {
                    const int l = __builtin_ctzll(live);
                    live &= live - 1;
                    const int jj = __builtin_amdgcn_readlane(j, l);
                    const float c1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cf), l));
                    const float *Tj = T + (size_t)jj * CC;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int e = lane + 64 * r;
                        if (e < CC) acc[g][r] = fmaf(c1, ti[g][r] - Tj[e], acc[g][r]);
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (pi[g] < 0) continue;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int e = lane + 64 * r;
                if (e < CC) {
                    // FIXED (graph from the forward): every row is produced exactly once -> plain store, the
                    // buffer need not be zeroed; CSR entry point: accumulate into the caller's buffer (its contract)
                    if (FIXED) grad_T[(size_t)pi[g] * CC + e] = acc[g][r];
                    else grad_T[(size_t)pi[g] * CC + e] += acc[g][r];
                }
            }
        }
    }
}

// in-edges that did not fit a point's TLG_CAP slots (hubs; rare): grad_T[tgt] += 2 g c (T_tgt - T_src)
template <int CC>
__global__ __launch_bounds__(256) void tl_overflow_kernel(int cap, float gscale, const float *__restrict__ T,
                                                          const int *__restrict__ ovf_cnt, const int *__restrict__ ovf,
                                                          const float *__restrict__ upstream, float *__restrict__ grad_T)
{
    const int lane = lane_id();
    const int cnt = min(*ovf_cnt, cap);
    gscale *= upstream ? upstream[0] : 1.f;
    for (int e = blockIdx.x * 4 + (threadIdx.x >> 6); e < cnt; e += gridDim.x * 4) {
        const int tgt = ovf[3 * (size_t)e], src = ovf[3 * (size_t)e + 1];
        const float cf = 2.f * gscale * __int_as_float(ovf[3 * (size_t)e + 2]);
        for (int el = lane; el < CC; el += 64)
            atomicAdd(grad_T + (size_t)tgt * CC + el, cf * (T[(size_t)tgt * CC + el] - T[(size_t)src * CC + el]));
    }
}

// ---- class anchors (train.py:505-526): for every class cc the point with the largest eta[b, cc, n] -- the FIRST
// maximum in flattened (b, n) order, as torch.argmax / the reference's loop pick it -- and that point's whole
// soft-max row: class_T[cc][:] = eta[b*, :, n*].  One workgroup per class scans the B*N values (coalesced along n),
// replaces ~6 torch launches (argmax, gathers, advanced indexing).  v_star (nullable): the maxima themselves.
__global__ __launch_bounds__(1024) void class_anchor_kernel(int b, int n, int c, const float *__restrict__ eta,
                                                            float *__restrict__ class_T, float *__restrict__ v_star)
{
    __shared__ float sv[16];
    __shared__ long long si[16];
    __shared__ long long win;
    const int cc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long total = (long long)b * n;
    float bv = -INFINITY;
    long long bi = -1;          // flat (b, n) index; NaN never wins (torch.argmax would return a NaN's position:
                                // soft-max outputs of finite logits have none)
    for (long long f = tid; f < total; f += 1024) {
        const int bb = (int)(f / n);
        const float v = eta[((size_t)bb * c + cc) * n + (f - (long long)bb * n)];
        if (bi < 0 || v > bv) { bv = v; bi = f; }   // ascending f per thread: the first maximum is kept
    }
    // wave reduce by (value desc, index asc)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o);
        const long long oi = __shfl_xor(bi, o);
        const bool take = oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi));
        bv = take ? ov : bv;
        bi = take ? oi : bi;
    }
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        float v = sv[0];
        long long i = si[0];
        for (int w = 1; w < 16; ++w) {
            const bool take = si[w] >= 0 && (i < 0 || sv[w] > v || (sv[w] == v && si[w] < i));
            v = take ? sv[w] : v;
            i = take ? si[w] : i;
        }
        win = i;
        if (v_star) v_star[cc] = v;
    }
    __syncthreads();
    const long long w = win;
    if (tid < c && w >= 0) {
        const int bb = (int)(w / n);
        class_T[cc * c + tid] = eta[((size_t)bb * c + tid) * n + (w - (long long)bb * n)];
    }
}

// ---- class-level transition block (train.py:505-557 after the anchor rows are known) -----------
//   P0[r][k] = gauss(proj[k]; mu = proj[r], s = sigma[r]);  A = P0 with row 0 := e_0 and column 0 := 0 below it
//   B = A / rowsum(A)[k]        (the reference's `X / X.sum(1)`: column k is divided by ROW-sum k)
//   N = geo*class_T + (1-geo)*B, row 0 := class_T[0];   M = N / rowsum(N)[k]
//   E = dec*ema_t + (1-dec)*M;  ema_corr = E / rowsum(E)[k];   ema_next = likewise from class_T alone
// One workgroup, one thread per matrix entry: ~40 tiny torch kernels forward and as many backward become one
// launch each.  Backward (only sigma is learnable) uses forward-mode derivatives, one sigma component per
// workgroup: d/d sigma[c0] touches a single row of P0, the three quirky normalisations carry it everywhere.
constexpr int CT_MAXC = 32;

__device__ __forceinline__ float ct_rowsum(float v, int r, int k, int C, float *buf)
{
    // returns sum_j X[k][j] to thread (r, k): every thread deposits its entry, then reads across row k
    __syncthreads();
    buf[r * CT_MAXC + k] = v;
    __syncthreads();
    float s = 0.f;
    for (int j = 0; j < C; ++j) s += buf[k * CT_MAXC + j];
    return s;
}

__device__ __forceinline__ double ct_rowsum(double v, int r, int k, int C, double *buf)
{
    __syncthreads();
    buf[r * CT_MAXC + k] = v;
    __syncthreads();
    double s = 0.0;
    for (int j = 0; j < C; ++j) s += buf[k * CT_MAXC + j];
    return s;
}

// d(sum g_corr * ema_corr + sum g_prior * prior) / d sigma, in fp64: the derivative fields are differences of quotients
// (dA / RA - A dRA / RA^2, three normalisations deep) whose fp32 evaluation lost 2.6e-5 of the result against the reference's
// fp64 run -- 100 x the error of the reference's own fp32 autograd (tests/test_ref_fixtures_gpu.py, round 4).  32 x 32
// threads on 17 x 17 numbers: the precision is free.  The forward kernel below stays fp32, op for op what train.py:505-545
// computes.
__global__ __launch_bounds__(CT_MAXC * CT_MAXC) void class_transition_grad_kernel(
    int C, double geo, double dec, const float *__restrict__ class_T, const float *__restrict__ sigma,
    const float *__restrict__ ema_t, const float *__restrict__ proj, const float *__restrict__ g_corr,
    const float *__restrict__ g_prior, float *__restrict__ g_sigma)
{
    __shared__ double buf[CT_MAXC * CT_MAXC];
    __shared__ double red[CT_MAXC * CT_MAXC / 64];
    const int tid = threadIdx.x, r = tid / CT_MAXC, k = tid % CT_MAXC;
    const bool in = r < C && k < C;
    const double cT = in ? class_T[r * C + k] : 0.0, eT = in ? ema_t[r * C + k] : 0.0;
    const double sg = in ? sigma[r] : 1.0, delta = in ? (double)proj[k] - (double)proj[r] : 0.0;
    const double P0 = in ? (1.0 / (sg * 2.5066282746310002)) * exp(-(delta * delta) / (2.0 * sg * sg)) : 0.0;
    const double A = !in ? 0.0 : (r == 0 ? (k == 0 ? 1.0 : 0.0) : (k == 0 ? 0.0 : P0));
    const double RA = ct_rowsum(A, r, k, C, buf);
    const double Bv = in ? A / RA : 0.0;
    const double N = !in ? 0.0 : (r == 0 ? cT : geo * cT + (1.0 - geo) * Bv);
    const double SN = ct_rowsum(N, r, k, C, buf);
    const double M = in ? N / SN : 0.0;
    const double E = in ? dec * eT + (1.0 - dec) * M : 0.0;
    const double UE = ct_rowsum(E, r, k, C, buf);
    const double gc = (in && g_corr) ? g_corr[r * C + k] : 0.0, gp = (in && g_prior) ? g_prior[r * C + k] : 0.0;
    const int c0 = blockIdx.x;   // one workgroup per sigma component (the forward fields are recomputed: cheap)
    const double dP0 = (in && r == c0) ? P0 * (-1.0 / sg + delta * delta / (sg * sg * sg)) : 0.0;
    const double dA = (r >= 1 && k >= 1) ? dP0 : 0.0;
    const double dRA = ct_rowsum(dA, r, k, C, buf);
    const double dB = in ? dA / RA - A * dRA / (RA * RA) : 0.0;
    const double dN = (in && r >= 1) ? (1.0 - geo) * dB : 0.0;
    const double dSN = ct_rowsum(dN, r, k, C, buf);
    const double dM = in ? dN / SN - N * dSN / (SN * SN) : 0.0;
    const double dE = (1.0 - dec) * dM;
    const double dUE = ct_rowsum(dE, r, k, C, buf);
    const double dF = in ? dE / UE - E * dUE / (UE * UE) : 0.0;
    double contrib = gc * dF + gp * dB;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) contrib += __shfl_xor(contrib, o);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = contrib;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < CT_MAXC * CT_MAXC / 64; ++w) t += red[w];
        g_sigma[c0] = (float)t;   // sole writer: the caller's buffer needs no zero-fill
    }
}

__global__ __launch_bounds__(CT_MAXC * CT_MAXC) void class_transition_kernel(
    int C, float geo, float geo1, float dec, float dec1,       // geo1 = fl32(1 - geo_lambda) formed in double, as Python forms it
    const float *__restrict__ class_T, const float *__restrict__ sigma,
    const float *__restrict__ ema_t, const float *__restrict__ proj, float *__restrict__ ema_corr,
    float *__restrict__ ema_next, float *__restrict__ prior, float *__restrict__ ema_keep)
{
    __shared__ float buf[CT_MAXC * CT_MAXC];
    const int tid = threadIdx.x, r = tid / CT_MAXC, k = tid % CT_MAXC;
    const bool in = r < C && k < C;
    const float cT = in ? class_T[r * C + k] : 0.f, eT = in ? ema_t[r * C + k] : 0.f;
    const float sg = in ? sigma[r] : 1.f, delta = in ? proj[k] - proj[r] : 0.f;
    const float P0 = in ? (1.f / (sg * 2.5066282746310002f)) * expf(-(delta * delta) / (2.f * sg * sg)) : 0.f;
    const float A = !in ? 0.f : (r == 0 ? (k == 0 ? 1.f : 0.f) : (k == 0 ? 0.f : P0));
    const float RA = ct_rowsum(A, r, k, C, buf);
    const float Bv = in ? A / RA : 0.f;
    const float N = !in ? 0.f : (r == 0 ? cT : geo * cT + geo1 * Bv);
    const float SN = ct_rowsum(N, r, k, C, buf);
    const float M = in ? N / SN : 0.f;
    const float E = in ? dec * eT + dec1 * M : 0.f;
    const float UE = ct_rowsum(E, r, k, C, buf);
    const float X = in ? dec * eT + dec1 * cT : 0.f;
    const float UX = ct_rowsum(X, r, k, C, buf);
    if (in) {
        ema_corr[r * C + k] = E / UE;
        ema_next[r * C + k] = X / UX;
        prior[r * C + k] = Bv;
        if (ema_keep) ema_keep[r * C + k] = eT;
    }
}

// out[e] += sum over blocks of partial[blk][e], in a fixed order (one workgroup per 64 entries, 16 waves round-robin over the
// blocks, fixed tree, one writer per entry): reproducible
__global__ __launch_bounds__(1024) void ntm_partial_reduce_kernel(int nblk, int len, const float *__restrict__ partial,
                                                                  float *__restrict__ out)
{
    constexpr int PARTS = 16;
    const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    __shared__ float red[PARTS][64];
    float s = 0.f;
    if (e < len)
        for (int bkt = part; bkt < nblk; bkt += PARTS) s += partial[(size_t)bkt * len + e];
    red[part][lane] = s;
    __syncthreads();
#pragma unroll
    for (int h = PARTS / 2; h >= 1; h >>= 1) {
        if (part < h) red[part][lane] += red[part + h][lane];
        __syncthreads();
    }
    if (part == 0 && e < len) out[e] += red[0][lane];
}

template <typename K>
static hipError_t set_lds(K kernel, size_t lds)
{
    return allow_big_lds((const void *)kernel, lds);
}

static inline int ntm_blocks(int total_pts)
{
    int tiles = (total_pts + NTM_TILE - 1) / NTM_TILE;
    return tiles < 1 ? 1 : (tiles > 1024 ? 1024 : tiles); // persistent-style: block prologues/epilogues (weights, grad_E atomics) amortise
}

} // namespace geot

using namespace geot;

#define GEOT_NTM_C 17 /* the reference's num_classes (cfgs/tooth_semi/default.yaml:29): the specialised kernels */

// Any other class count 1..GEN_MAXC runs the run-time-C kernels of ntm_generic.hip / the R-bucketed graph kernel.
static inline bool ntm_generic_c(int c) { return c >= 1 && c <= GEN_MAXC && c != GEOT_NTM_C; }

template <bool BWD, bool SIGNED>
static hipError_t launch_threed_plain(int b, int n, int c, int k, int pd, float sigma, float gscale, const float *pos,
                                      const int *labels, const float *T, const int *nbr, const int *order,
                                      float *per_point, float *grad_T, hipStream_t s)
{
    int blocks = (b * n + 3) / 4;
    if (blocks > 16384) blocks = 16384;
    blocks = (blocks + 7) & ~7; // the XCD-chunked walk needs a multiple of 8 workgroups
    const int cc = c * c, r = (cc + 63) / 64;
    const float inv2s2 = 1.f / (2.f * sigma * sigma);
#define GEOT_TL_PLAIN(R)                                                                                            \
    hipLaunchKernelGGL((threed_loss_kernel<R, BWD, SIGNED>), dim3(blocks), dim3(256), 0, s, b * n, n, k, pd, cc, inv2s2, \
                       gscale, pos, labels, T, nbr, order, per_point, grad_T)
    if (r <= 2) GEOT_TL_PLAIN(2);
    else if (r <= 5) GEOT_TL_PLAIN(5);
    else if (r <= 8) GEOT_TL_PLAIN(8);
    else GEOT_TL_PLAIN(16);
#undef GEOT_TL_PLAIN
    return hipGetLastError();
}

static inline int sig_mfma_blocks(long long total_pts, int per_cu)
{
    long long tiles = (total_pts + SM_PTS - 1) / SM_PTS, cap = (long long)per_cu * 256; // blocks per CU x 256 CUs
    return (int)(tiles < 1 ? 1 : (tiles > cap ? cap : tiles));
}
constexpr int SIG_FWD_PER_CU = 4, SIG_BWD_PER_CU = 2; // forward: 37 KB of LDS, <= 128 registers; backward: 236 registers

GEOT_EXPORT int geot_ntm_sig_t_mean(int b, int n, int c, const float *p, const float *W, const float *cm,
                                    float *ins_T, void *stream)
{
    if ((c != GEOT_NTM_C && !ntm_generic_c(c)) || b < 0 || n < 0) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if (c != GEOT_NTM_C) return gen_sig_t_mean(false, b, n, c, p, W, cm, nullptr, ins_T, (hipStream_t)stream);
    constexpr int C = GEOT_NTM_C;
    size_t lds = (size_t)SigMfma<C>::LDS_FLOATS_FWD * sizeof(float);
    hipError_t e = set_lds(sig_t_mean_mfma_kernel<C, false, 256>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sig_t_mean_mfma_kernel<C, false, 256>), dim3(sig_mfma_blocks((long long)b * n, SIG_FWD_PER_CU)), dim3(256), lds,
                       (hipStream_t)stream, b * n, n, p, W, cm, nullptr, ins_T, nullptr);
    return hipGetLastError();
}

GEOT_EXPORT long long geot_ntm_sig_t_mean_ws_floats(int b, int n)
{
    if (b < 0 || n < 0) return -1;
    return (long long)sig_mfma_blocks((long long)b * n, SIG_BWD_PER_CU) * (GEOT_NTM_C + 1) * GEOT_NTM_C * GEOT_NTM_C;
}

GEOT_EXPORT int geot_ntm_sig_t_mean_grad_w(int b, int n, int c, const float *p, const float *W, const float *cm,
                                           const float *grad_ins_T, float *grad_W, float *workspace,
                                           void *stream)
{
    if (c != GEOT_NTM_C || b < 0 || n < 0 || !workspace) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    constexpr int C = GEOT_NTM_C;
    size_t lds = (size_t)SigMfma<C>::LDS_FLOATS_BWD * sizeof(float);
    hipError_t e = set_lds(sig_t_mean_mfma_kernel<C, true, SIG_BWD_THREADS>, lds);
    if (e != hipSuccess) return e;
    const int nblk = sig_mfma_blocks((long long)b * n, SIG_BWD_PER_CU);
    hipLaunchKernelGGL((sig_t_mean_mfma_kernel<C, true, SIG_BWD_THREADS>), dim3(nblk), dim3(SIG_BWD_THREADS), lds, (hipStream_t)stream,
                       b * n, n, p, W, cm, grad_ins_T, nullptr, workspace);
    hipLaunchKernelGGL((sig_t_mean_wgrad_reduce_kernel<C>), dim3(((C + 1) * C * C + 63) / 64), dim3(1024), 0,
                       (hipStream_t)stream, nblk, workspace, cm, grad_W);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_sig_t_mean_grad_raw(int b, int n, int c, const float *p, const float *W,
                                             const float *cm, const float *grad_ins_T, float *grad_raw,
                                             void *stream)
{
    if ((c != GEOT_NTM_C && !ntm_generic_c(c)) || b < 0 || n < 0) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if (c != GEOT_NTM_C) return gen_sig_t_mean(true, b, n, c, p, W, cm, grad_ins_T, grad_raw, (hipStream_t)stream);
    constexpr int C = GEOT_NTM_C;
    size_t lds = (size_t)(C * C * C + C * C + 4 + NTM_TILE * NtmLds<C>::STRIDE) * sizeof(float);
    hipError_t e = set_lds(sig_t_mean_kernel<C, true>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((sig_t_mean_kernel<C, true>), dim3(ntm_blocks(b * n)), dim3(NTM_THREADS), lds,
                       (hipStream_t)stream, b * n, n, p, W, cm, grad_ins_T, grad_raw);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_correct(int b, int n, int c, float lam, const float *logits, const float *ins_T,
                                 const float *ema_t, float *out, void *stream)
{
    if ((c != GEOT_NTM_C && !ntm_generic_c(c)) || b < 0 || n < 0) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if (c != GEOT_NTM_C) return gen_correct_fwd(b, n, c, lam, logits, ins_T, ema_t, out, (hipStream_t)stream);
    constexpr int C = GEOT_NTM_C;
    size_t lds = (size_t)(C * C + 4 + NTM_TILE * NtmLds<C>::STRIDE) * sizeof(float);
    hipError_t e = set_lds(ntm_correct_fwd_kernel<C>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ntm_correct_fwd_kernel<C>), dim3(ntm_blocks(b * n)), dim3(NTM_THREADS), lds,
                       (hipStream_t)stream, b * n, n, lam, logits, ins_T, ema_t, out);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_correct_grad(int b, int n, int c, float lam, const float *logits,
                                      const float *ins_T, const float *ema_t, const float *grad_out,
                                      float *grad_logits, float *grad_ins_T, float *grad_ema_t,
                                      void *stream)
{
    if ((c != GEOT_NTM_C && !ntm_generic_c(c)) || b < 0 || n < 0) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if (c != GEOT_NTM_C)
        return gen_correct_bwd(b, n, c, lam, logits, ins_T, ema_t, grad_out, grad_logits, grad_ins_T, grad_ema_t,
                               (hipStream_t)stream);
    constexpr int C = GEOT_NTM_C;
    size_t lds = (size_t)(C * C + 4 + NTM_TILE * NtmLds<C>::STRIDE) * sizeof(float);
    hipError_t e = set_lds(ntm_correct_bwd_kernel<C>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((ntm_correct_bwd_kernel<C>), dim3(ntm_blocks(b * n)), dim3(NTM_THREADS), lds,
                       (hipStream_t)stream, b * n, n, lam, logits, ins_T, ema_t, grad_out,
                       grad_logits, grad_ins_T, grad_ema_t, nullptr);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_class_anchors(int b, int n, int c, const float *eta, float *class_T, float *v_star, void *stream)
{
    if (b < 1 || n < 1 || c < 1 || c > 1024 || !eta || !class_T) return hipErrorInvalidValue;
    hipLaunchKernelGGL(class_anchor_kernel, dim3(c), dim3(1024), 0, (hipStream_t)stream, b, n, c, eta, class_T, v_star);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_class_transition(int c, double geo_lambda, double ema_decay, const float *class_T,
                                          const float *sigma, const float *ema_t, const float *proj, float *ema_t_corr,
                                          float *ema_t_next, float *prior_T, float *ema_t_keep, void *stream)
{
    if (c < 1 || c > CT_MAXC) return hipErrorInvalidValue;
    hipLaunchKernelGGL(class_transition_kernel, dim3(1), dim3(CT_MAXC * CT_MAXC), 0, (hipStream_t)stream, c,
                       (float)geo_lambda, (float)(1.0 - geo_lambda), (float)ema_decay, (float)(1.0 - ema_decay), class_T, sigma,
                       ema_t, proj, ema_t_corr, ema_t_next, prior_T, ema_t_keep);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_class_transition_grad(int c, double geo_lambda, double ema_decay, const float *class_T,
                                               const float *sigma, const float *ema_t, const float *proj,
                                               const float *grad_ema_t_corr, const float *grad_prior_T,
                                               float *grad_sigma, void *stream)
{
    if (c < 1 || c > CT_MAXC || !grad_sigma) return hipErrorInvalidValue;
    hipLaunchKernelGGL(class_transition_grad_kernel, dim3(c), dim3(CT_MAXC * CT_MAXC), 0, (hipStream_t)stream, c,
                       geo_lambda, ema_decay, class_T, sigma, ema_t, proj, grad_ema_t_corr, grad_prior_T, grad_sigma);
    return hipGetLastError();
}

GEOT_EXPORT long long geot_ntm_correct_ws_floats(int b, int n)
{
    if (b < 0 || n < 0) return -1;
    return (long long)ntm_blocks(b * n) * GEOT_NTM_C * GEOT_NTM_C;
}

// geot_ntm_correct_grad with the (c,c) grad_ema_t reduced through per-block partial sums in `workspace`
// (geot_ntm_correct_ws_floats(b, n) floats) instead of ~3e5 atomics on 289 addresses.
GEOT_EXPORT int geot_ntm_correct_grad_ws(int b, int n, int c, float lam, const float *logits,
                                         const float *ins_T, const float *ema_t, const float *grad_out,
                                         float *grad_logits, float *grad_ins_T, float *grad_ema_t,
                                         float *workspace, void *stream)
{
    if ((c != GEOT_NTM_C && !ntm_generic_c(c)) || b < 0 || n < 0) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if (!workspace || c != GEOT_NTM_C)
        return geot_ntm_correct_grad(b, n, c, lam, logits, ins_T, ema_t, grad_out, grad_logits, grad_ins_T, grad_ema_t,
                                     stream);
    constexpr int C = GEOT_NTM_C;
    size_t lds = (size_t)(C * C + 4 + NTM_TILE * NtmLds<C>::STRIDE) * sizeof(float);
    hipError_t e = set_lds(ntm_correct_bwd_kernel<C>, lds);
    if (e != hipSuccess) return e;
    const int nblk = ntm_blocks(b * n);
    hipLaunchKernelGGL((ntm_correct_bwd_kernel<C>), dim3(nblk), dim3(NTM_THREADS), lds, (hipStream_t)stream, b * n,
                       n, lam, logits, ins_T, ema_t, grad_out, grad_logits, grad_ins_T, grad_ema_t, workspace);
    hipLaunchKernelGGL(ntm_partial_reduce_kernel, dim3((C * C + 63) / 64), dim3(1024), 0,
                       (hipStream_t)stream, nblk, C * C, workspace, grad_ema_t);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_threed_loss(int b, int n, int c, int k, float sigma, const float *positions,
                                     const int *labels, const float *ins_T, const int *nbr,
                                     float *per_point, void *stream)
{
    if (c < 1 || c > GEN_MAXC || b < 0 || n < 0 || k < 1 || k > 64 || !(sigma > 0.f)) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    return launch_threed_plain<false, false>(b, n, c, k, 3, sigma, 0.f, positions, labels, ins_T, nbr, nullptr, per_point,
                                             nullptr, (hipStream_t)stream);
}

GEOT_EXPORT int geot_ntm_threed_loss_ord(int b, int n, int c, int k, float sigma, const float *positions,
                                         const int *labels, const float *ins_T, const int *nbr, const int *order,
                                         float *per_point, void *stream)
{
    if ((c != GEOT_NTM_C && !ntm_generic_c(c)) || b < 0 || n < 0 || k < 1 || k > 64 || !(sigma > 0.f)) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if (c != GEOT_NTM_C)   // the one-point-per-wave kernel walks the same order
        return launch_threed_plain<false, false>(b, n, c, k, 3, sigma, 0.f, positions, labels, ins_T, nbr, order, per_point,
                                                 nullptr, (hipStream_t)stream);
    constexpr int CC = GEOT_NTM_C * GEOT_NTM_C;
    const char *ge = getenv("GEOT_NTM_G");
    const int Gsel = ge ? atoi(ge) : 4;
#define GEOT_TL_FWD(G)                                                                                              \
    {                                                                                                               \
        int blocks = (b * n + 4 * G - 1) / (4 * G);                                                                 \
        if (blocks > 16384) blocks = 16384;                                                                         \
        blocks = (blocks + 7) & ~7;                                                                                 \
        hipLaunchKernelGGL((threed_loss_shared_kernel<CC, G, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, \
                           b * n, n, k, 1.f / (2.f * sigma * sigma), positions, labels, ins_T, nbr, order, per_point, \
                           TlGraph{});                                                                              \
    }
    if (Gsel == 2) GEOT_TL_FWD(2) else if (Gsel == 3) GEOT_TL_FWD(3) else if (Gsel == 6) GEOT_TL_FWD(6) else GEOT_TL_FWD(4)
#undef GEOT_TL_FWD
    return hipGetLastError();
}

GEOT_EXPORT long long geot_ntm_threed_loss_ws_bytes(int b, int n, int k)
{
    if (b < 0 || n < 0 || k < 0) return -1;
    const long long t = (long long)b * n;
    const long long nblk = (t + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
    // S, offsets, cursor, block sums, then per edge: wout, rev, revc
    return 4 * (t + (t + 1) + t + nblk + 3 * t * k) + 64;
}

// Same result as geot_ntm_threed_loss_grad (up to fp32 summation order) through the atomic-free gather
// over the graph and its reverse.  workspace: geot_ntm_threed_loss_ws_bytes(b, n, k) bytes.
GEOT_EXPORT int geot_ntm_threed_loss_grad_ws(int b, int n, int c, int k, float sigma, float grad_scale,
                                             const float *positions, const int *labels, const float *ins_T,
                                             const int *nbr, const int *order, float *grad_ins_T,
                                             void *workspace, long long ws_bytes, void *stream)
{
    if ((c != GEOT_NTM_C && !ntm_generic_c(c)) || b < 0 || n < 0 || k < 1 || k > 64 || !(sigma > 0.f)) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if ((long long)b * n * 64 > 0x7fffffffLL) return hipErrorInvalidValue;
    if (c != GEOT_NTM_C || !workspace || ws_bytes < geot_ntm_threed_loss_ws_bytes(b, n, k))
        return geot_ntm_threed_loss_grad(b, n, c, k, sigma, grad_scale, positions, labels, ins_T, nbr, grad_ins_T,
                                         stream);
    constexpr int CC = GEOT_NTM_C * GEOT_NTM_C;
    hipStream_t s = (hipStream_t)stream;
    const int t = b * n;
    const int nblk = (t + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK;
    float *S = (float *)workspace;
    int *off = (int *)(S + t);
    int *cursor = off + t + 1;
    int *bsum = cursor + t;
    float *wout = (float *)(bsum + nblk);
    int *rev = (int *)(wout + (size_t)t * k);
    float *revc = (float *)(rev + (size_t)t * k);
    const float inv2s2 = 1.f / (2.f * sigma * sigma);
    hipError_t e = zero_words(off, (long long)t + 1, s);
    if (e != hipSuccess) return e;
    int seg_shift = 0;
    while ((1 << seg_shift) < k) ++seg_shift;
    const long long lanes = (long long)t << seg_shift;
    const int eb = (int)((lanes + 255) / 256);
    hipLaunchKernelGGL(tl_prep_kernel, dim3(eb), dim3(256), 0, s, t, n, k, seg_shift, inv2s2, positions, labels, nbr,
                       order, S, wout, off);
    exclusive_scan_i32(t, off, bsum, cursor, s);
    hipLaunchKernelGGL(tl_fill_kernel, dim3(eb), dim3(256), 0, s, t, n, k, seg_shift, nbr, wout, S, off, order, cursor,
                       rev, revc);
    const char *ge = getenv("GEOT_NTM_G");
    const int Gsel = ge ? atoi(ge) : 4;
#define GEOT_TL_BWD(G)                                                                                              \
    {                                                                                                               \
        int blocks = (t + 4 * G - 1) / (4 * G);                                                                     \
        if (blocks > 16384) blocks = 16384;                                                                         \
        blocks = (blocks + 7) & ~7;                                                                                 \
        hipLaunchKernelGGL((tl_grad_gather_shared_kernel<CC, G, false>), dim3(blocks), dim3(256), 0, s, t, n, k, grad_scale, \
                           ins_T, nbr, wout, S, off, rev, revc, order, nullptr, grad_ins_T);                        \
    }
    if (Gsel == 2) GEOT_TL_BWD(2) else if (Gsel == 3) GEOT_TL_BWD(3) else if (Gsel == 6) GEOT_TL_BWD(6) else GEOT_TL_BWD(4)
#undef GEOT_TL_BWD
    return hipGetLastError();
}

static TlGraph tl_graph_views(void *workspace, long long t, int k)
{
    TlGraph g;
    g.S = (float *)workspace;
    g.wout = g.S + t;
    g.cnt = (int *)(g.wout + t * k);
    g.ovf_cnt = g.cnt + t; // adjacent to cnt: one memset clears both
    g.rev = g.ovf_cnt + 4;
    g.revc = (float *)(g.rev + t * TLG_CAP);
    g.ovf = (int *)(g.revc + t * TLG_CAP);
    return g;
}

GEOT_EXPORT long long geot_ntm_threed_graph_bytes(int b, int n, int k)
{
    if (b < 0 || n < 0 || k < 0) return -1;
    const long long t = (long long)b * n;
    return 4 * (t + t * k + t + 4 + 2 * t * TLG_CAP + 3 * t * k) + 64;
}

// Forward of threeD_space_loss that also leaves the kNN graph's reverse adjacency, edge weights and
// normalisers in `graph` (geot_ntm_threed_graph_bytes(b, n, k) bytes) for geot_ntm_threed_loss_grad_graph:
// the backward is then the gather alone, no graph building.  `order` may be NULL.
GEOT_EXPORT int geot_ntm_threed_loss_fwd_graph(int b, int n, int c, int k, float sigma, const float *positions,
                                               const int *labels, const float *ins_T, const int *nbr,
                                               const int *order, float *per_point, void *graph,
                                               long long graph_bytes, void *stream)
{
    if (c != GEOT_NTM_C || b < 0 || n < 0 || k < 1 || k > 64 || !(sigma > 0.f)) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if (!graph || graph_bytes < geot_ntm_threed_graph_bytes(b, n, k) || (long long)b * n * TLG_CAP > 0x7fffffffLL)
        return hipErrorInvalidValue;
    constexpr int CC = GEOT_NTM_C * GEOT_NTM_C;
    constexpr int G = 4;
    const long long t = (long long)b * n;
    TlGraph g = tl_graph_views(graph, t, k);
    hipError_t e = zero_words(g.cnt, t + 4, (hipStream_t)stream);
    if (e != hipSuccess) return e;
    int blocks = (int)((t + 4 * G - 1) / (4 * G));
    if (blocks > 16384) blocks = 16384;
    blocks = (blocks + 7) & ~7;
    hipLaunchKernelGGL((threed_loss_shared_kernel<CC, G, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, b * n,
                       n, k, 1.f / (2.f * sigma * sigma), positions, labels, ins_T, nbr, order, per_point, g);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_threed_loss_grad_graph(int b, int n, int c, int k, float grad_scale,
                                                const float *upstream, const float *ins_T, const int *nbr,
                                                const int *order, const void *graph, long long graph_bytes,
                                                float *grad_ins_T, void *stream)
{
    if (c != GEOT_NTM_C || b < 0 || n < 0 || k < 1 || k > 64) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    if (!graph || graph_bytes < geot_ntm_threed_graph_bytes(b, n, k)) return hipErrorInvalidValue;
    constexpr int CC = GEOT_NTM_C * GEOT_NTM_C;
    constexpr int G = 4;
    const long long t = (long long)b * n;
    const TlGraph g = tl_graph_views(const_cast<void *>(graph), t, k);
    int blocks = (int)((t + 4 * G - 1) / (4 * G));
    if (blocks > 16384) blocks = 16384;
    blocks = (blocks + 7) & ~7;
    hipLaunchKernelGGL((tl_grad_gather_shared_kernel<CC, G, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (int)t, n, k, grad_scale, ins_T, nbr, g.wout, g.S, g.cnt, g.rev, g.revc, order, upstream,
                       grad_ins_T);
    hipLaunchKernelGGL((tl_overflow_kernel<CC>), dim3(64), dim3(256), 0, (hipStream_t)stream, (int)(t * k), grad_scale,
                       ins_T, g.ovf_cnt, g.ovf, upstream, grad_ins_T);
    return hipGetLastError();
}

GEOT_EXPORT int geot_ntm_feature_loss(int b, int n, int c, int k, int feat_dim, float sigma, const float *feats,
                                      const int *labels, const float *ins_T, const int *nbr,
                                      float *per_point, void *stream)
{
    if (c < 1 || c > GEN_MAXC || b < 0 || n < 0 || k < 1 || k > 64 || feat_dim < 1 || !(sigma > 0.f))
        return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    return launch_threed_plain<false, true>(b, n, c, k, feat_dim, sigma, 0.f, feats, labels, ins_T, nbr, nullptr, per_point,
                                            nullptr, (hipStream_t)stream);
}

GEOT_EXPORT int geot_ntm_feature_loss_grad(int b, int n, int c, int k, int feat_dim, float sigma,
                                           float grad_scale, const float *feats, const int *labels,
                                           const float *ins_T, const int *nbr, float *grad_ins_T, void *stream)
{
    if (c < 1 || c > GEN_MAXC || b < 0 || n < 0 || k < 1 || k > 64 || feat_dim < 1 || !(sigma > 0.f))
        return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    return launch_threed_plain<true, true>(b, n, c, k, feat_dim, sigma, grad_scale, feats, labels, ins_T, nbr, nullptr,
                                           nullptr, grad_ins_T, (hipStream_t)stream);
}

GEOT_EXPORT int geot_ntm_threed_loss_grad(int b, int n, int c, int k, float sigma, float grad_scale,
                                          const float *positions, const int *labels, const float *ins_T,
                                          const int *nbr, float *grad_ins_T, void *stream)
{
    if (c < 1 || c > GEN_MAXC || b < 0 || n < 0 || k < 1 || k > 64 || !(sigma > 0.f)) return hipErrorInvalidValue;
    if ((long long)b * n == 0) return hipSuccess;
    return launch_threed_plain<true, false>(b, n, c, k, 3, sigma, grad_scale, positions, labels, ins_T, nbr, nullptr,
                                            nullptr, grad_ins_T, (hipStream_t)stream);
}
