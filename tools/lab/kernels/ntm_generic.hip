// LAB COPY (tools/lab/ntm_knockouts.sh: GEN_KO_STORE / GEN_KO_MFMA / GEN_KO_POST) of
// ntm_generic.hip -- the per-point transition-matrix kernels for ANY class count 2 <= C <= 32.
//
// The reference builds `nclasses` Linear(2C -> C) heads for whatever nclasses the config names
// (openpoints/models/backbone/transformer.py:1104-1110) and its losses take num_classes as an argument
// (utils/insT_loss.py:62-67).  ntm.hip holds the kernels specialised for the configured tooth label set
// (C = 17: MFMA tiles of 289 columns, LDS tiles of 32 x 289 floats); this file serves every other count with
// one mapping that needs no compile-time C:
//
//     one wave per point, a half-wave (32 lanes) per matrix row, lane = column  (C <= 32)
//
// so the L1 norms / dot products over a row are five xor-shuffles inside the half-wave, a row of T_i is one
// contiguous 4C-byte access of the half-wave (two adjacent rows per wave instruction), and nothing goes
// through shared memory except the read-only weights.  Lane efficiency is C/32; these are streaming kernels
// over (B*N, C, C) and stay HBM-bound from C ~ 12 up.  Same arithmetic, same operation order per element as the
// specialised kernels (one reciprocal per row).
#include "geot_common.h"
#include "ntm_generic.h"

namespace geot {

__device__ __forceinline__ float half_sum(float v)   // sum over the 32 lanes of this lane's half-wave
{
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ---- sig_t_mean forward / d raw ---------------------------------------------------------------------------
// raw[kk][o] = sum_j p_j W[kk][o][j] + sum_j cm[kk][j] W[kk][o][C+j];  clamp to [1e-5, 1-1e-5];  L1-normalise over o.
// BACKWARD: out = d raw = [raw inside the clamp] * (g - sum_o g tn) / den   (the d/dW GEMM is the caller's).
template <bool BACKWARD>
__global__ __launch_bounds__(256) void gen_sig_t_mean_kernel(int total_pts, int n, int c, const float *__restrict__ p,
                                                             const float *__restrict__ W, const float *__restrict__ cm,
                                                             const float *__restrict__ grad_out, float *__restrict__ out)
{
    extern __shared__ float gen_lds[];
    const int cc = c * c;
    float *Wl = gen_lds;          // [j][kk*c + o] = W[kk][o][j]   (lanes o read consecutive words)
    float *bias = Wl + c * cc;    // [kk*c + o]
    for (int e = threadIdx.x; e < c * cc; e += 256) {
        const int j = e / cc, col = e - j * cc;
        Wl[e] = W[(size_t)col * 2 * c + j];
    }
    for (int col = threadIdx.x; col < cc; col += 256) {
        const int kk = col / c;
        float acc = 0.f;
        for (int j = 0; j < c; ++j) acc += cm[kk * c + j] * W[(size_t)col * 2 * c + c + j];
        bias[col] = acc;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, o = lane & 31, h = lane >> 5;
    const bool live_o = o < c;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < total_pts; i += gridDim.x * 4) {
        const int b = i / n, ni = i - b * n;
        const float pl = lane < c ? p[((size_t)b * c + lane) * n + ni] : 0.f;   // lane j holds p_j
        for (int k0 = 0; k0 < c; k0 += 2) {
            const int kk = k0 + h;
            const bool live = live_o && kk < c;
            const int col = live ? kk * c + o : 0;
            float raw = live ? bias[col] : 0.f;
            for (int j = 0; j < c; ++j) {
                const float pj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(pl), j));
                raw = fmaf(pj, live ? Wl[j * cc + col] : 0.f, raw);
            }
            const float cl = live ? fminf(fmaxf(raw, 1e-5f), 1.f - 1e-5f) : 0.f;   // clamped values are positive
            const float rden = 1.f / fmaxf(half_sum(cl), 1e-12f);
            if (!BACKWARD) {
                if (live) out[(size_t)i * cc + col] = cl * rden;
            } else {
                const float g = live ? grad_out[(size_t)i * cc + col] : 0.f;
                const float dot = half_sum(g * (cl * rden));
                const bool inside = raw >= 1e-5f && raw <= 1.f - 1e-5f;
                if (live) out[(size_t)i * cc + col] = inside ? (g - dot) * rden : 0.f;
            }
        }
    }
}

// ---- sig_t_mean, second form: lane = point ------------------------------------------------------------------------
// The wave-per-point kernel above reads a weight from LDS for every multiply and leaves C/32 of its lanes idle
// (0.5 TB/s at C = 16).  Here a wave takes 64 consecutive points, one per lane: the point's probabilities sit in CP
// registers (CP = C rounded up to 8, a template parameter so that the arrays are statically indexed), the weights are
// wave-uniform and come through the SCALAR path (s_load), the clamp / L1 norm / dot product of a row are lane-local --
// no shuffles -- and a row of 64 x C results crosses a wave-private LDS tile so that global memory sees 4C-byte
// segments instead of one 4-byte store per lane.  d raw (BACKWARD) takes its incoming gradient through the same tile.
template <int CP, bool BACKWARD>
__global__ __launch_bounds__(256) void gen_sig_t_mean_rows_kernel(int total_pts, int n, int c, const float *__restrict__ p,
                                                                  const float *__restrict__ W, const float *__restrict__ cm,
                                                                  const float *__restrict__ grad_out, float *__restrict__ out)
{
    extern __shared__ float gen_lds[];
    const int cc = c * c, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *bias = gen_lds;                                           // [kk*c + o] = sum_j cm[kk][j] W[kk][o][c + j]
    float *tile = gen_lds + ((cc + 3) & ~3) + wave * 64 * (CP + 1);  // [point][CP + 1]: odd stride, conflict-free rows
    for (int col = threadIdx.x; col < cc; col += 256) {
        const int kk = col / c;
        float acc = 0.f;
        for (int j = 0; j < c; ++j) acc += cm[kk * c + j] * W[(size_t)col * 2 * c + c + j];
        bias[col] = acc;
    }
    __syncthreads();
    const int jmax = 2 * c - 1;                                      // last valid column of a weight row (padded reads stay inside it)
    for (long long i0 = ((long long)blockIdx.x * 4 + wave) * 64; i0 < total_pts; i0 += (long long)gridDim.x * 256) {
        const long long i = i0 + lane;
        const bool ok = i < total_pts;
        const int b = ok ? (int)(i / n) : 0, ni = ok ? (int)(i - (long long)b * n) : 0;
        float pv[CP];
#pragma unroll
        for (int j = 0; j < CP; ++j) pv[j] = (ok && j < c) ? p[((size_t)b * c + j) * n + ni] : 0.f;
        const int cnt = (int)min((long long)64, total_pts - i0);    // points of this wave's tile
        for (int kk = 0; kk < c; ++kk) {
            float *gdst = out + (size_t)i0 * cc + (size_t)kk * c;    // + point * cc + o
            float g[CP];
            if (BACKWARD) {
                // incoming gradient of row kk: coalesced segments -> tile -> one row per lane
                const float *gsrc = grad_out + (size_t)i0 * cc + (size_t)kk * c;
                int pt = lane / c, o = lane - pt * c;
                const int dp = 64 / c, dq = 64 - dp * c;
                for (int e = lane; e < cnt * c; e += 64) {
                    tile[pt * (CP + 1) + o] = gsrc[(size_t)pt * cc + o];
                    pt += dp; o += dq;
                    if (o >= c) { o -= c; ++pt; }
                }
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
#pragma unroll
                for (int o2 = 0; o2 < CP; ++o2) g[o2] = (o2 < c) ? tile[lane * (CP + 1) + o2] : 0.f;
                __builtin_amdgcn_wave_barrier();
                __threadfence_block();
            }
            float raw[CP], s = 0.f;
#pragma unroll
            for (int o = 0; o < CP; ++o) {
                raw[o] = 0.f;
                if (o < c) {                                         // wave-uniform
                    const float *wrow = W + ((size_t)kk * c + o) * 2 * c;
                    float acc = bias[kk * c + o];
#pragma unroll
                    for (int j = 0; j < CP; ++j) acc = fmaf(pv[j], wrow[min(j, jmax)], acc);   // pv[j] = 0 beyond c
                    raw[o] = acc;
                    s += fminf(fmaxf(acc, 1e-5f), 1.f - 1e-5f);      // clamped values are positive
                }
            }
            const float rden = 1.f / fmaxf(s, 1e-12f);
            if (!BACKWARD) {
#pragma unroll
                for (int o = 0; o < CP; ++o)
                    if (o < c) tile[lane * (CP + 1) + o] = fminf(fmaxf(raw[o], 1e-5f), 1.f - 1e-5f) * rden;
            } else {
                float dot = 0.f;
#pragma unroll
                for (int o = 0; o < CP; ++o)
                    if (o < c) dot += g[o] * (fminf(fmaxf(raw[o], 1e-5f), 1.f - 1e-5f) * rden);
#pragma unroll
                for (int o = 0; o < CP; ++o)
                    if (o < c) {
                        const bool inside = raw[o] >= 1e-5f && raw[o] <= 1.f - 1e-5f;
                        tile[lane * (CP + 1) + o] = inside ? (g[o] - dot) * rden : 0.f;
                    }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
            {
                int pt = lane / c, o = lane - pt * c;
                const int dp = 64 / c, dq = 64 - dp * c;
                for (int e = lane; e < cnt * c; e += 64) {
                    gdst[(size_t)pt * cc + o] = tile[pt * (CP + 1) + o];
                    pt += dp; o += dq;
                    if (o >= c) { o -= c; ++pt; }
                }
            }
            __builtin_amdgcn_wave_barrier();
            __threadfence_block();
        }
    }
}

// ---- sig_t_mean, third form: the heads on the matrix cores, any C <= 32 ---------------------------------------------
// The lane-per-point kernel above spends C^2 scalar-fed FMAs per point and head on the vector pipe and refetches 8 C^3
// bytes of weights through the scalar cache per wave tile (C = 20: 0.8 TB/s).  The heads are a GEMM with K = C:
//     raw[kk][o][pt] = bias[kk][o] + sum_j W[kk][o][j] p[pt][j]
// done as v_mfma_f32_32x32x2_f32 tiles with the WEIGHTS on the row side and 32 POINTS on the column side, so that in the
// result a lane owns one point (column = lane & 31) and 16 of a tile's 32 rows: clamp, L1 norm and the backward's dot
// product over o are 16 in-lane steps plus one exchange with lane ^ 32 -- no LDS tile, no barrier in the loop.
// A row tile holds 32 / CPAD heads of CPAD rows (CPAD = 8, 16, 32 >= C), the bias starts the accumulator, the weight
// fragments of every (row tile, 4 K steps) sit in LDS in fragment order (one ds_read_b128 per 4 MFMAs; 132 KB at
// C = 32, one workgroup of 16 waves per CU), each wave walks its own point tiles.  A lane writes its results as 16-byte
// pieces of its point's row (4-byte pieces if C is not a multiple of 4): lanes l and l ^ 32 fill 32 adjacent bytes, the C
// row tiles of a point complete its 4 C^2 bytes within the same wave.
typedef float gen_f32x16 __attribute__((ext_vector_type(16)));
typedef float gen_f32x4 __attribute__((ext_vector_type(4)));
typedef float gen_f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte piece at 4-byte alignment (one dwordx4 access)
constexpr int GEN_MT = 1024, GEN_MW = GEN_MT / 64;

// TAILS: C is not a multiple of 4, the last piece of a row is partial (element stores)
// Memory operations of a wave retire in order (one vmcnt counter for loads and stores), so a load issued behind a tile's stores
// waits for them: the forward fetches the NEXT unit's point fragments before the first store of this one, the backward the
// NEXT row tile's gradient pieces (a counted wait then leaves one row tile of stores in flight).
template <int CPAD, bool BACKWARD, bool TAILS>
__global__ __launch_bounds__(GEN_MT) void gen_sig_t_mean_mfma_kernel(int total_pts, int n, int c, int ksplit, const float *__restrict__ p,
                                                                     const float *__restrict__ W, const float *__restrict__ cm,
                                                                     const float *__restrict__ grad_out, float *__restrict__ out)
{
    constexpr int NG = 32 / CPAD, QPG = CPAD / 8, KSMAX = CPAD / 2;   // heads per row tile, 8-row blocks per head, K steps
    constexpr bool AHEAD = BACKWARD && !(TAILS && CPAD == 32);        // (that one instantiation has no registers for the second set)
    extern __shared__ float gen_lds[];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cc = c * c, nkt = (c + NG - 1) / NG, ng = (c + 7) / 8;   // row tiles; groups of 4 K steps (8 inputs)
    float *Af = gen_lds;                     // [nkt][ng][64][4]: lane (row, half), element e holds W[kk][o][2 (4 g + e) + half]
    float *Bt = Af + (size_t)nkt * ng * 256; // [nkt][32]: bias of the tile's rows
    for (int e = tid; e < nkt * ng * 256; e += GEN_MT) {
        const int l = (e >> 2) & 63, g = (e >> 8) % ng, kt = (e >> 8) / ng, row = l & 31;
        const int kk = kt * NG + row / CPAD, o = row % CPAD, j = 2 * (4 * g + (e & 3)) + (l >> 5);
        Af[e] = (kk < c && o < c && j < c) ? W[((size_t)kk * c + o) * 2 * c + j] : 0.f;
    }
    for (int e = tid; e < nkt * 32; e += GEN_MT) {                      // (ascending j, as the other two forms: same bits)
        const int row = e & 31, kk = (e >> 5) * NG + row / CPAD, o = row % CPAD;
        float acc = 0.f;
        if (kk < c && o < c) {
            const float *wr = W + ((size_t)kk * c + o) * 2 * c + c, *cr = cm + kk * c;
            for (int j = 0; j < c; ++j) acc += cr[j] * wr[j];
        }
        Bt[e] = acc;
    }
    __syncthreads();
    // work units: (tile of 32 points, one of ksplit ranges of row tiles)
    const long long units = (((long long)total_pts + 31) / 32) * ksplit, stride = (long long)gridDim.x * GEN_MW;
    long long u = (long long)blockIdx.x * GEN_MW + wave;
    if (u >= units) return;
    auto point = [&](long long unit, bool &ok, unsigned &row0, unsigned &pbase) {
        const long long i = unit / ksplit * 32 + r;
        ok = i < total_pts;
        const long long ic = ok ? i : (long long)total_pts - 1;
        const int b = (int)(ic / n), ni = (int)(ic - (long long)b * n);
        row0 = (unsigned)ic * (unsigned)cc;  // 32-bit element offsets (host: B N C^2 < 2^30): uniform base + lane offset
        pbase = (unsigned)(b * c * n + ni);
    };
    auto fetch_p = [&](float (&dst)[KSMAX], unsigned pbase) {           // column side: lane (point, half) holds p[point][2 s + half]
#pragma unroll
        for (int s = 0; s < KSMAX; ++s) dst[s] = p[pbase + (unsigned)(min(2 * s + h, c - 1) * n)];
    };
    auto fetch_g = [&](float (&dst)[16], unsigned row, int kt) {         // the incoming gradient in the result's own layout
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = kt * NG + q / QPG, o0 = 8 * (q % QPG) + 4 * h;
            const unsigned src = row + (unsigned)(min(kk, c - 1) * c);
            if (!TAILS || c >= 4) {
                const int at = max(min(o0, c - 4), 0);                  // a partial last piece is read (o0 - at) elements early
                const gen_f32x4u v = *reinterpret_cast<const gen_f32x4u *>(grad_out + (src + (unsigned)at));
                const int sh = TAILS ? o0 - at : 0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    dst[4 * q + e] = !TAILS || sh == 0 ? v[e] : (sh == 1 ? v[(e + 1) & 3] : (sh == 2 ? v[(e + 2) & 3] : v[3]));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[4 * q + e] = grad_out[src + (unsigned)min(o0 + e, c - 1)];
            }
        }
    };
    bool ok;
    unsigned row0, pbase;
    point(u, ok, row0, pbase);
    float bf[KSMAX], bfn[KSMAX], gv[16], gvn[16];
    fetch_p(bf, pbase);
    if (AHEAD) fetch_g(gv, row0, (int)((long long)nkt * (u % ksplit) / ksplit));
    while (true) {
        const int part = (int)(u % ksplit);
        const int kt0 = (int)((long long)nkt * part / ksplit), kt1 = (int)((long long)nkt * (part + 1) / ksplit);
        const long long un = u + stride;
        const bool more = un < units;
        bool okn;
        unsigned row0n, pbasen;
        point(more ? un : u, okn, row0n, pbasen);
        const int kt0n = more ? (int)((long long)nkt * (un % ksplit) / ksplit) : kt0;
        if (!BACKWARD) fetch_p(bfn, pbasen);
        for (int kt = kt0; kt < kt1; ++kt) {
            if (AHEAD) {
                const bool last = kt + 1 == kt1;
                fetch_g(gvn, last ? row0n : row0, last ? kt0n : kt + 1);
            } else if (BACKWARD) fetch_g(gv, row0, kt);
            gen_f32x16 acc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {    // result rows 8 q + 4 half + (0..3)
                const gen_f32x4 v = *reinterpret_cast<const gen_f32x4 *>(Bt + kt * 32 + 8 * q + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[4 * q + e] = v[e];
            }
            // the weight fragments of 4 K steps per 16-byte read, the next group's read in flight under this group's MFMAs
            // (a K step past C inside the last group multiplies zeros)
            const gen_f32x4 *af = reinterpret_cast<const gen_f32x4 *>(Af + (size_t)kt * ng * 256) + lane;
            gen_f32x4 a4 = af[0];
#pragma unroll
            for (int g = 0; g < KSMAX / 4; ++g) {
#ifdef GEN_KO_MFMA
                if (g < ng && n < 0) {
#else
                if (g < ng) {
#endif
                    gen_f32x4 nx = a4;
                    if (g + 1 < KSMAX / 4 && g + 1 < ng) nx = af[(g + 1) * 64];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], (ok && 2 * (4 * g + e) + h < c) ? bf[4 * g + e] : 0.f, acc, 0, 0, 0);
                    a4 = nx;
                }
            }
            // clamp in place; a value is inside the clamp iff the clamp left it alone
            float sum[NG], rden[NG], dot[NG];
            unsigned inside = 0;
#pragma unroll
            for (int g = 0; g < NG; ++g) sum[g] = dot[g] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#ifdef GEN_KO_POST
                if (8 * (q % QPG) < c && n < 0) {
#else
                if (8 * (q % QPG) < c) {     // (wave-uniform: 8-row blocks past C hold padding)
#endif
                    const int o0 = 8 * (q % QPG) + 4 * h;
                    float part4 = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = acc[4 * q + e], cl = fminf(fmaxf(a, 1e-5f), 1.f - 1e-5f);   // positive
                        if (BACKWARD) inside |= (a == cl ? 1u : 0u) << (4 * q + e);
                        acc[4 * q + e] = cl;
                        part4 += (!TAILS || o0 + e < c) ? cl : 0.f;
                    }
                    sum[q / QPG] += (TAILS || o0 < c) ? part4 : 0.f;
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                sum[g] += __shfl_xor(sum[g], 32);                      // the rows of the other half
                rden[g] = 1.f / fmaxf(sum[g], 1e-12f);
            }
            if (BACKWARD) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (8 * (q % QPG) < c) {
                        const int kk = kt * NG + q / QPG, o0 = 8 * (q % QPG) + 4 * h;
                        float part4 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            gv[4 * q + e] = (kk < c && o0 + e < c) ? gv[4 * q + e] : 0.f;   // padding carries no gradient
                            part4 += gv[4 * q + e] * (acc[4 * q + e] * rden[q / QPG]);
                        }
                        dot[q / QPG] += part4;
                    }
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) dot[g] += __shfl_xor(dot[g], 32);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = q / QPG, kk = kt * NG + g, o0 = 8 * (q % QPG) + 4 * h;
                if (8 * (q % QPG) < c) {
                    gen_f32x4u v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (!BACKWARD) v[e] = acc[4 * q + e] * rden[g];
                        else v[e] = ((inside >> (4 * q + e)) & 1u) ? (gv[4 * q + e] - dot[g]) * rden[g] : 0.f;
                    }
                    float *dst = out + (row0 + (unsigned)(kk * c + o0));
#ifdef GEN_KO_STORE
                    if (ok && kk < c && n < 0) {
#else
                    if (ok && kk < c) {
#endif
                        if (o0 + 4 <= c) *reinterpret_cast<gen_f32x4u *>(dst) = v;
                        else if (TAILS) {
#pragma unroll
                            for (int e = 0; e < 3; ++e)
                                if (o0 + e < c) dst[e] = v[e];
                        }
                    }
                }
            }
            if (AHEAD) {
#pragma unroll
                for (int e = 0; e < 16; ++e) gv[e] = gvn[e];
            }
        }
        if (!more) break;
        u = un;
        ok = okn;
        row0 = row0n;
        if (BACKWARD) fetch_p(bf, pbasen);
        else {
#pragma unroll
            for (int s = 0; s < KSMAX; ++s) bf[s] = bfn[s];
        }
    }
}

// ---- logit correction (train.py:549-552) --------------------------------------------------------------------
// v = lam*E + (1-lam)*T_i;  tn = v / max(sum_c |v|, eps);  out[c] = sum_r logit[r] * tn[r][c]
__global__ __launch_bounds__(256) void gen_correct_fwd_kernel(int total_pts, int n, int c, float lam,
                                                              const float *__restrict__ logits,
                                                              const float *__restrict__ insT, const float *__restrict__ E,
                                                              float *__restrict__ out)
{
    const int cc = c * c;
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < total_pts; i += gridDim.x * 4) {
        const int b = i / n, ni = i - b * n;
        float acc = 0.f;
        for (int r0 = 0; r0 < c; r0 += 2) {
            const int r = r0 + h;
            const bool live = col < c && r < c;
            const float v = live ? lam * E[r * c + col] + (1.f - lam) * insT[(size_t)i * cc + r * c + col] : 0.f;
            const float s = half_sum(fabsf(v));
            const float l = r < c ? logits[((size_t)b * c + r) * n + ni] : 0.f;
            acc = fmaf(l / fmaxf(s, 1e-12f), v, acc);
        }
        acc += __shfl_xor(acc, 32);     // even rows + odd rows
        if (lane < c) out[((size_t)b * c + lane) * n + ni] = acc;
    }
}

// Backward: grad_logits[r] = sum_c tn[r][c] go[c];  d v = (l go - sign(v) l gl) / den;  grad_T = (1-lam) d v;
// grad_E += lam * sum_points d v -- accumulated per wave in its own LDS copy (plain read-modify-write: a lane
// always meets the same address), folded into grad_E with one atomic per entry and workgroup.
__global__ __launch_bounds__(256) void gen_correct_bwd_kernel(int total_pts, int n, int c, float lam,
                                                              const float *__restrict__ logits,
                                                              const float *__restrict__ insT, const float *__restrict__ E,
                                                              const float *__restrict__ grad_out,
                                                              float *__restrict__ grad_logits, float *__restrict__ grad_insT,
                                                              float *__restrict__ grad_E)
{
    extern __shared__ float gen_lds[];
    const int cc = c * c;
    float *eacc = gen_lds + (threadIdx.x >> 6) * cc;      // [4 waves][cc]
    for (int e = threadIdx.x; e < 4 * cc; e += 256) gen_lds[e] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, col = lane & 31, h = lane >> 5;
    for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < total_pts; i += gridDim.x * 4) {
        const int b = i / n, ni = i - b * n;
        const float go = col < c ? grad_out[((size_t)b * c + col) * n + ni] : 0.f;
        for (int r0 = 0; r0 < c; r0 += 2) {
            const int r = r0 + h;
            const bool live = col < c && r < c;
            const float v = live ? lam * E[r * c + col] + (1.f - lam) * insT[(size_t)i * cc + r * c + col] : 0.f;
            const float s = half_sum(fabsf(v));
            const float rden = 1.f / fmaxf(s, 1e-12f);
            const float l = r < c ? logits[((size_t)b * c + r) * n + ni] : 0.f;
            const float gl = half_sum((v * rden) * go);
            if (col == 0 && r < c) grad_logits[((size_t)b * c + r) * n + ni] = gl;
            const float sg = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
            const float dv = s > 1e-12f ? (l * go - sg * (l * gl)) * rden : l * go * rden;
            if (live) {
                grad_insT[(size_t)i * cc + r * c + col] = (1.f - lam) * dv;
                eacc[r * c + col] += lam * dv;
            }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < cc; e += 256)
        atomicAdd(grad_E + e, (gen_lds[e] + gen_lds[cc + e]) + (gen_lds[2 * cc + e] + gen_lds[3 * cc + e]));
}

static inline int gen_blocks(long long total_pts)
{
    long long blocks = (total_pts + 3) / 4;
    return (int)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));   // 8 workgroups per CU, grid-stride beyond
}

template <int CP>
static hipError_t launch_sig_rows(bool backward, int b, int n, int c, const float *p, const float *W, const float *cm,
                                  const float *grad_out, float *out, hipStream_t s)
{
    const size_t lds = (size_t)(((c * c + 3) & ~3) + 4 * 64 * (CP + 1)) * sizeof(float);    // <= 38 KB
    long long blocks = ((long long)b * n + 255) / 256;
    const long long cap = 8LL * device_cus();
    if (blocks > cap) blocks = cap;
    if (backward)
        hipLaunchKernelGGL((gen_sig_t_mean_rows_kernel<CP, true>), dim3((int)blocks), dim3(256), lds, s, b * n, n, c, p, W, cm,
                           grad_out, out);
    else
        hipLaunchKernelGGL((gen_sig_t_mean_rows_kernel<CP, false>), dim3((int)blocks), dim3(256), lds, s, b * n, n, c, p, W, cm,
                           grad_out, out);
    return hipGetLastError();
}

template <int CPAD>
static hipError_t launch_sig_mfma(bool backward, int b, int n, int c, const float *p, const float *W, const float *cm,
                                  const float *grad_out, float *out, hipStream_t s)
{
    constexpr int NG = 32 / CPAD;
    const int nkt = (c + NG - 1) / NG, ng = (c + 7) / 8;
    const size_t lds = ((size_t)nkt * ng * 256 + (size_t)nkt * 32) * sizeof(float);         // 132 KB at C = 32
    const bool tails = (c & 3) != 0;
    const void *fn = backward ? (tails ? (const void *)gen_sig_t_mean_mfma_kernel<CPAD, true, true> : (const void *)gen_sig_t_mean_mfma_kernel<CPAD, true, false>)
                              : (tails ? (const void *)gen_sig_t_mean_mfma_kernel<CPAD, false, true> : (const void *)gen_sig_t_mean_mfma_kernel<CPAD, false, false>);
    hipError_t e = allow_big_lds(fn, lds);
    if (e != hipSuccess) return e;
    const long long tiles = ((long long)b * n + 31) / 32;
    int per_cu = 0;                          // workgroups of 16 waves that share a CU (registers and LDS)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, GEN_MT, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    const long long cap = (long long)per_cu * device_cus(), waves = cap * GEN_MW;
    // ranges of row tiles per point tile: the fewest that leave the last round of the resident waves >= 85 % full (every
    // unit boundary is a wait for the wave's stores in flight)
    int ksplit = 1;
    for (; ksplit < nkt; ++ksplit) {
        const long long units = tiles * ksplit, rounds = (units + waves - 1) / waves;
        if (units <= waves || units * 100 >= rounds * waves * 85) break;
    }
    long long blocks = (tiles * ksplit + GEN_MW - 1) / GEN_MW;
    if (blocks > cap) blocks = cap;
#define GEOT_GEN_SIG(BW, TL)                                                                                                   \
    hipLaunchKernelGGL((gen_sig_t_mean_mfma_kernel<CPAD, BW, TL>), dim3((int)blocks), dim3(GEN_MT), lds, s, b * n, n, c, ksplit, p, W, \
                       cm, grad_out, out)
    if (backward) { if (tails) GEOT_GEN_SIG(true, true); else GEOT_GEN_SIG(true, false); }
    else { if (tails) GEOT_GEN_SIG(false, true); else GEOT_GEN_SIG(false, false); }
#undef GEOT_GEN_SIG
    return hipGetLastError();
}

hipError_t gen_sig_t_mean(bool backward, int b, int n, int c, const float *p, const float *W, const float *cm,
                          const float *grad_out, float *out, hipStream_t s)
{
    const char *impl = getenv("GEOT_NTM_GENERIC");      // A/B tests: "rows" = the lane-per-point kernel, "wave" = the wave-per-point kernel
    const bool small = (long long)b * n * c * c < 0x7fffffffLL * 4LL;
    if ((!impl || (impl[0] != 'r' && impl[0] != 'w')) && (long long)b * n * c * c < (1LL << 30)) {
        if (c <= 8) return launch_sig_mfma<8>(backward, b, n, c, p, W, cm, grad_out, out, s);
        if (c <= 16) return launch_sig_mfma<16>(backward, b, n, c, p, W, cm, grad_out, out, s);
        return launch_sig_mfma<32>(backward, b, n, c, p, W, cm, grad_out, out, s);
    }
    if (!(impl && impl[0] == 'w') && small) {
        if (c <= 8) return launch_sig_rows<8>(backward, b, n, c, p, W, cm, grad_out, out, s);
        if (c <= 16) return launch_sig_rows<16>(backward, b, n, c, p, W, cm, grad_out, out, s);
        if (c <= 24) return launch_sig_rows<24>(backward, b, n, c, p, W, cm, grad_out, out, s);
        return launch_sig_rows<32>(backward, b, n, c, p, W, cm, grad_out, out, s);
    }
    const size_t lds = (size_t)(c + 1) * c * c * sizeof(float);        // 135 KB at C = 32
    const void *fn = backward ? (const void *)gen_sig_t_mean_kernel<true> : (const void *)gen_sig_t_mean_kernel<false>;
    hipError_t e = allow_big_lds(fn, lds);
    if (e != hipSuccess) return e;
    // persistent workgroups, as many per CU as the staged weights ((C+1) C^2 floats each) leave room for in LDS:
    // the inner loop is a chain of LDS reads, it needs waves to hide behind (2 per SIMD measured 0.23 TB/s at C = 16)
    long long blocks = ((long long)b * n + 3) / 4;
    long long per_cu = (160 * 1024) / (long long)(lds + 512);
    per_cu = per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu);
    const long long cap = per_cu * device_cus();
    if (blocks > cap) blocks = cap;
    if (backward)
        hipLaunchKernelGGL(gen_sig_t_mean_kernel<true>, dim3((int)blocks), dim3(256), lds, s, b * n, n, c, p, W, cm, grad_out, out);
    else
        hipLaunchKernelGGL(gen_sig_t_mean_kernel<false>, dim3((int)blocks), dim3(256), lds, s, b * n, n, c, p, W, cm, grad_out, out);
    return hipGetLastError();
}

hipError_t gen_correct_fwd(int b, int n, int c, float lam, const float *logits, const float *insT, const float *E,
                           float *out, hipStream_t s)
{
    hipLaunchKernelGGL(gen_correct_fwd_kernel, dim3(gen_blocks((long long)b * n)), dim3(256), 0, s, b * n, n, c, lam, logits,
                       insT, E, out);
    return hipGetLastError();
}

hipError_t gen_correct_bwd(int b, int n, int c, float lam, const float *logits, const float *insT, const float *E,
                           const float *grad_out, float *grad_logits, float *grad_insT, float *grad_E, hipStream_t s)
{
    hipLaunchKernelGGL(gen_correct_bwd_kernel, dim3(gen_blocks((long long)b * n)), dim3(256), (size_t)4 * c * c * sizeof(float), s,
                       b * n, n, c, lam, logits, insT, E, grad_out, grad_logits, grad_insT, grad_E);
    return hipGetLastError();
}

} // namespace geot
