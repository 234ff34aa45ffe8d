// fps.hip -- furthest point sampling for gfx950 (MI355X).
//
// Replaces (behaviour, not code):
//   pointnet2/_ext_src/src/sampling_gpu.cu:73-232          (dense, origin-skip, block<=512)
//   openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu:101-260 (dense, block<=1024)
//   pointops/src/sampling/sampling_cuda_kernel.cu:15-171, 175-349 (offset-batched, weighted)
//
// Design (MI355X-first; history: profiles/DESIGN_r01_r03.md 4.1):
//   * the reference's block-size-dependent tie rule is reproduced with an explicit key
//     bitreverse(k mod bs) : (k div bs)  instead of inheriting whatever order our own reduction
//     has, which frees the launch geometry from the reference's;
//   * three kernels, bit-identical outputs (tests run all three on every case):
//       fps_kernel                 unpruned: one 1024-thread workgroup per cloud, <= 24 points per
//                                  lane in VGPRs, DPP arg-max + one LDS hop + one barrier per round;
//                                  weighted FPS, n < 1024, n > 24576 (streaming variant);
//       fps_pruned_kernel<.., 1>   exact bucket pruning (Morton counting sort, per-slot boxes, cached
//                                  wave candidates), one sample per round   (GEOT_FPS_IMPL=single);
//       fps_pruned_kernel<.., 8>   the same with multi-commit rounds: up to 8 provably independent
//                                  samples per round, the certain first one applied while wave 0
//                                  ranks the candidates                     (default).
#include "geot_common.h"
#include "geot_hip.h"
#include <cmath>
#include <cstdlib>
#include <type_traits>

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


// ===========================================================================
// Pruned FPS (exact): the same greedy max-min selection, but a round only
// touches the 64-point buckets the new sample can possibly change.
//
//  * Prologue (once): counting sort of the cloud into a 16^3 Morton grid in LDS, so that
//    (wave, slot) = 64 consecutive sorted points is a spatially compact bucket; each lane
//    then keeps PPT points (xyz + running min-distance) in VGPRs and lane i of every wave
//    holds the bounding box and the current max min-distance ("smax") of the wave's slot i.
//  * Round: lanes test the new sample q against their slot's box: if the squared distance
//    from q to the box exceeds smax (with a 1e-5 relative safety margin covering fp32
//    rounding of both sides), then d(p,q) >= temp[p] for every point of the bucket and
//    min(d, temp) leaves it unchanged -- the bucket is skipped.  Only the surviving slots
//    (a ballot mask, typically 0-2 per wave) are updated and their smax re-reduced.
//  * Arg-max: wave max over the per-slot maxima, then the reference tie key among the
//    lanes of the winning slot(s).  The wave's candidate (max, key, x, y, z) is wave-uniform
//    (SGPRs), is recomputed only in rounds where the wave updated something, and is
//    published to LDS so the next round needs no memory access at all -- one LDS hop and
//    ONE barrier per round.
//  * The per-lane point arrays are ext_vector registers indexed with a wave-uniform runtime
//    slot (VGPR-index mode, s_set_gpr_idx_on), so the round loop is a few hundred
//    instructions.  (A first version dispatched to per-slot straight-line code; at 47 slots
//    that was 60 KB of loop body and every round missed the instruction cache.)
//  The temps, the selected indices and the tie-breaking are bit-identical to the unpruned
//  kernel (and to the reference): skipping is only ever a proven no-op.
// ===========================================================================
// ---- developer lab hooks (tools/fps_lab.py builds variants with -DGEOT_LAB_*) ----------
#if defined(GEOT_LAB_STATS) || defined(GEOT_LAB_STAMPS)
__device__ unsigned long long geot_fps_dbg[8];
#endif
constexpr int FP_CELLS = 4096;
constexpr int FP_MAX_N = 768 * 32;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));

// 16 or 32 floats per lane held in VGPRs; get/set take a WAVE-UNIFORM runtime index and are
// lowered to VGPR-index mode (s_set_gpr_idx_on + v_mov), branch-free.
template <int CAP> struct RegVec;
template <> struct RegVec<16> {
    f32x16 a;
    __device__ __forceinline__ float get(int i) const { return a[i]; }
    __device__ __forceinline__ void set(int i, float v) { a[i] = v; }
};
template <> struct RegVec<32> {
    f32x32 a;
    __device__ __forceinline__ float get(int i) const { return a[i]; }
    __device__ __forceinline__ void set(int i, float v) { a[i] = v; }
};

struct FpsEntry { // one wave's published candidate
    uint32_t key;
    float x, y, z;
};
struct FpsEntry2 { // multi-commit variant: sortable (hi:lo) = (bits(max)+1 : ~key), runner-up bound v2
    uint32_t lo, hi; // little-endian: the first 8 bytes read as one u64 give (hi << 32) | lo
    float x, y;
    float z;
    int v2;
    uint32_t pad[2];
};
#ifndef GEOT_FP_TMAX
#define GEOT_FP_TMAX 8
#endif
constexpr int FP_TMAX = GEOT_FP_TMAX; // samples committed per round at most (<= 8: the 8 x 8 conflict matrix is one wave)

// v[lane `lane`] = value, both wave-uniform.  v_writelane_b32 allows only one SGPR besides
// M0 on gfx9, so the lane select goes through M0.
__device__ __forceinline__ void set_lane(int &v, int value, int lane)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(value), "s"(lane) : "m0");
}
__device__ __forceinline__ void set_lane(float &v, float value, int lane)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(v) : "s"(value), "s"(lane) : "m0");
}
__device__ __forceinline__ float uniform(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ float read_lane(float v, int lane)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}

// Fused DPP reductions (the op reads its first source through the DPP network): half the
// instructions of a v_mov_dpp + op pair.  2 wait states between a VALU write and a DPP read.
#define GEOT_DPP_STEP(op, ctrl) op " %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
__device__ __forceinline__ int wave_max_i32_fast(int v)
{
    asm volatile("s_nop 1\n\t"
                 GEOT_DPP_STEP("v_max_i32_dpp", "quad_perm:[1,0,3,2]")
                 GEOT_DPP_STEP("v_max_i32_dpp", "quad_perm:[2,3,0,1]")
                 GEOT_DPP_STEP("v_max_i32_dpp", "row_half_mirror")
                 GEOT_DPP_STEP("v_max_i32_dpp", "row_mirror")
                 : "+v"(v));
    int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ uint32_t wave_min_u32_fast(uint32_t v)
{
    asm volatile("s_nop 1\n\t"
                 GEOT_DPP_STEP("v_min_u32_dpp", "quad_perm:[1,0,3,2]")
                 GEOT_DPP_STEP("v_min_u32_dpp", "quad_perm:[2,3,0,1]")
                 GEOT_DPP_STEP("v_min_u32_dpp", "row_half_mirror")
                 GEOT_DPP_STEP("v_min_u32_dpp", "row_mirror")
                 : "+v"(v));
    uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

// Multiset top-2 of one int per lane over the wave (t1 >= t2, both wave-uniform), optionally together with
// the min of a u32 key: ONE butterfly instead of two or three dependent ones.  Per step
//   n1 = max(m1, p1);  n2 = max(min(m1, p1), max(m2, p2))      (p = partner's pair; the sets are disjoint)
// with the three DPP ops independent of each other.  Register ping-pong keeps every DPP read two
// instructions behind the write it depends on (the required wait states), so no s_nop between steps.
#define GEOT_T2_STEP(A1, A2, B1, B2, CTRL)                                                   \
    "v_max_i32_dpp " B1 ", " A1 ", " A1 " " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
    "v_min_i32_dpp %[u], " A1 ", " A1 " " CTRL " row_mask:0xf bank_mask:0xf\n\t"             \
    "v_max_i32_dpp %[w], " A2 ", " A2 " " CTRL " row_mask:0xf bank_mask:0xf\n\t"             \
    "v_max_i32 " B2 ", %[u], %[w]\n\t"
#define GEOT_T2K_STEP(A1, A2, B1, B2, CTRL)                                                  \
    "v_max_i32_dpp " B1 ", " A1 ", " A1 " " CTRL " row_mask:0xf bank_mask:0xf\n\t"           \
    "v_min_i32_dpp %[u], " A1 ", " A1 " " CTRL " row_mask:0xf bank_mask:0xf\n\t"             \
    "v_max_i32_dpp %[w], " A2 ", " A2 " " CTRL " row_mask:0xf bank_mask:0xf\n\t"             \
    "v_min_u32_dpp %[k], %[k], %[k] " CTRL " row_mask:0xf bank_mask:0xf\n\t"                 \
    "v_max_i32 " B2 ", %[u], %[w]\n\t"
__device__ __forceinline__ void top2_merge_rows(int a1, int a2, int &t1, int &t2)
{
    int x1 = __builtin_amdgcn_readlane(a1, 0), x2 = __builtin_amdgcn_readlane(a2, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        int y1 = __builtin_amdgcn_readlane(a1, r), y2 = __builtin_amdgcn_readlane(a2, r);
        x2 = max(min(x1, y1), max(x2, y2));
        x1 = max(x1, y1);
    }
    t1 = x1; t2 = x2;
}
__device__ __forceinline__ void wave_top2_i32(int v, int &t1, int &t2)
{
    int a1 = v, a2 = (int)0x80000000, b1, b2, u, w;
    asm volatile("s_nop 1\n\t"
                 GEOT_T2_STEP("%[a1]", "%[a2]", "%[b1]", "%[b2]", "quad_perm:[1,0,3,2]")
                 GEOT_T2_STEP("%[b1]", "%[b2]", "%[a1]", "%[a2]", "quad_perm:[2,3,0,1]")
                 GEOT_T2_STEP("%[a1]", "%[a2]", "%[b1]", "%[b2]", "row_half_mirror")
                 GEOT_T2_STEP("%[b1]", "%[b2]", "%[a1]", "%[a2]", "row_mirror")
                 : [a1] "+v"(a1), [a2] "+v"(a2), [b1] "=&v"(b1), [b2] "=&v"(b2), [u] "=&v"(u), [w] "=&v"(w));
    top2_merge_rows(a1, a2, t1, t2);
}
__device__ __forceinline__ void wave_top2_i32_min_u32(int v, uint32_t key, int &t1, int &t2, uint32_t &kmin)
{
    int a1 = v, a2 = (int)0x80000000, b1, b2, u, w;
    uint32_t k = key;
    asm volatile("s_nop 1\n\t"
                 GEOT_T2K_STEP("%[a1]", "%[a2]", "%[b1]", "%[b2]", "quad_perm:[1,0,3,2]")
                 GEOT_T2K_STEP("%[b1]", "%[b2]", "%[a1]", "%[a2]", "quad_perm:[2,3,0,1]")
                 GEOT_T2K_STEP("%[a1]", "%[a2]", "%[b1]", "%[b2]", "row_half_mirror")
                 GEOT_T2K_STEP("%[b1]", "%[b2]", "%[a1]", "%[a2]", "row_mirror")
                 : [a1] "+v"(a1), [a2] "+v"(a2), [b1] "=&v"(b1), [b2] "=&v"(b2), [u] "=&v"(u), [w] "=&v"(w),
                   [k] "+v"(k));
    top2_merge_rows(a1, a2, t1, t2);
    uint32_t ka = __builtin_amdgcn_readlane(k, 0), kb = __builtin_amdgcn_readlane(k, 16);
    uint32_t kc = __builtin_amdgcn_readlane(k, 32), kd = __builtin_amdgcn_readlane(k, 48);
    kmin = min(min(ka, kb), min(kc, kd));
}

// LDS 64-bit max without return value (the atomic optimizer would wrap atomicMax in
// wave-election code although we are already down to one lane).
__device__ __forceinline__ void lds_max_u64(unsigned long long *addr, unsigned long long v)
{
    uint32_t a = (uint32_t)(uintptr_t)addr; // LDS pointers: low 32 bits are the LDS offset
    asm volatile("ds_max_u64 %0, %1" : : "v"(a), "v"(v) : "memory");
}

__device__ __forceinline__ uint32_t morton12(uint32_t cx, uint32_t cy, uint32_t cz)
{
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        m |= (((cx >> b) & 1u) << (3 * b)) | (((cy >> b) & 1u) << (3 * b + 1)) | (((cz >> b) & 1u) << (3 * b + 2));
    return m;
}

// NT threads (NW = NT/64 waves), PPT points per lane (<= CAP register slots; one slot per lane
// for the box test, so PPT <= 32 < 64).
template <int NT, int PPT, bool SKIP, int TMAX>
__global__ __launch_bounds__(NT) void fps_pruned_kernel(
    const float *__restrict__ xyz, const int *__restrict__ offset,
    const int *__restrict__ new_offset, int n_dense, int m_dense, float *__restrict__ temp,
    int *__restrict__ idxs, int L)
{
    constexpr int NW = NT / 64;
    constexpr int CAP = PPT <= 16 ? 16 : 32;
    static_assert(PPT <= CAP && NW <= 16, "geometry");
    __shared__ uint16_t perm[PPT * NT];
    __shared__ uint32_t cellcnt[FP_CELLS];
    __shared__ FpsEntry exch[2][16];
    __shared__ unsigned long long best[3];
    __shared__ float red[NW][6];
    __shared__ uint32_t wsum[8];

    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform for the compiler too (scalar branches)
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
    int *out = idxs + start_m;
    if (tid == 0) out[0] = base;
    if (m == 1) return;

    // ---- 1. bounding box of the cloud --------------------------------------------------
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = tid; k < n; k += NT) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = P[k * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = wave_min_f32(lo[a]), h = wave_max_f32(hi[a]);
        if (lane == 0) { red[wave][a] = l; red[wave][3 + a] = h; }
    }
    for (int c = tid; c < FP_CELLS; c += NT) cellcnt[c] = 0;
    if (tid < 3) best[tid] = 0ull;
    __syncthreads();
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[0][a], h = red[0][3 + a];
        for (int w = 1; w < NW; ++w) { l = fminf(l, red[w][a]); h = fmaxf(h, red[w][3 + a]); }
        lo[a] = l;
        float ext = h - l;
        inv[a] = (ext > 0.f && ext < INFINITY) ? 16.f / ext : 0.f;
    }

    // ---- 2. Morton cell histogram; remember (cell, rank-in-cell) per point -----------------
    RegVec<CAP> X, Y, Z, D; // D doubles as the (cell,rank) scratch of the sort, then holds temp
#pragma unroll 1
    for (int i = 0; i < PPT; ++i) {
        int k = i * NT + tid;
        uint32_t cr = 0;
        if (k < n) {
            uint32_t c[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float f = (P[k * 3 + a] - lo[a]) * inv[a];
                int ci = (f >= 0.f) ? (int)fminf(f, 15.f) : 0; // NaN -> 0
                c[a] = (uint32_t)ci;
            }
            uint32_t cell = morton12(c[0], c[1], c[2]);
            uint32_t r = atomicAdd(&cellcnt[cell], 1u);
            cr = (cell << 16) | r;
        }
        D.set(i, __uint_as_float(cr));
    }
    __syncthreads();
    // ---- 3. exclusive prefix over the 4096 cells (8 per thread, first 512 threads) -------------
    {
        uint32_t v[8], s = 0, inc = 0;
        if (tid < 512) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] = cellcnt[tid * 8 + e]; s += v[e]; }
            inc = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                uint32_t o = __shfl_up(inc, d);
                if (lane >= d) inc += o;
            }
            if (lane == 63) wsum[wave] = inc;
        }
        __syncthreads();
        if (tid < 512) {
            uint32_t woff = 0;
            for (int w = 0; w < wave; ++w) woff += wsum[w];
            uint32_t run = woff + inc - s;
#pragma unroll
            for (int e = 0; e < 8; ++e) { cellcnt[tid * 8 + e] = run; run += v[e]; }
        }
    }
    __syncthreads();
    // ---- 4. scatter: sorted position -> original (local) index ---------------------------------
#pragma unroll 1
    for (int i = 0; i < PPT; ++i) {
        int k = i * NT + tid;
        uint32_t cr = __float_as_uint(D.get(i));
        if (k < n) perm[cellcnt[cr >> 16] + (cr & 0xFFFFu)] = (uint16_t)k;
    }
    __syncthreads();

    // ---- 5. gather the sorted points; 6. per-slot boxes and maxima (lane i owns slot i) ---------
    // smax holds the fp32 BITS of the slot's max min-distance; slots without a valid point (and
    // lanes >= PPT) hold the bits of -1.0f: negative as an int, and never 'active' as a float.
    float bx0 = 0.f, by0 = 0.f, bz0 = 0.f, bx1 = 0.f, by1 = 0.f, bz1 = 0.f;
    int smax = __float_as_int(-1.f);
#pragma unroll 1
    for (int i = 0; i < PPT; ++i) {
        int pos = i * NT + tid;
        bool in = pos < n;
        int k = in ? (int)perm[pos] : 0;
        float x = in ? P[k * 3 + 0] : 0.f, y = in ? P[k * 3 + 1] : 0.f, z = in ? P[k * 3 + 2] : 0.f;
        float t = in ? T[k] : -1.f;
        if (SKIP && in && origin_skipped(x, y, z)) t = -1.f;
        X.set(i, x); Y.set(i, y); Z.set(i, z); D.set(i, t);
        bool valid = t >= 0.f;
        set_lane(bx0, uniform(wave_min_f32(valid ? x : INFINITY)), i);
        set_lane(bx1, uniform(wave_max_f32(valid ? x : -INFINITY)), i);
        set_lane(by0, uniform(wave_min_f32(valid ? y : INFINITY)), i);
        set_lane(by1, uniform(wave_max_f32(valid ? y : -INFINITY)), i);
        set_lane(bz0, uniform(wave_min_f32(valid ? z : INFINITY)), i);
        set_lane(bz1, uniform(wave_max_f32(valid ? z : -INFINITY)), i);
        set_lane(smax, __builtin_amdgcn_readfirstlane(wave_max_i32(__float_as_int(t))), i);
    }

    if constexpr (TMAX > 1) {
    // =====================================================================================
    // Multi-commit rounds.  FPS is sequential, but consecutive winners are usually far apart:
    // once c1 is chosen, the runner-up wave candidate c2 IS the next sample whenever
    //   (A) c1 does not lower c2's min-distance:  sqdist3(c2, c1) >= temp[c2]   (same fp32 expression
    //       the update evaluates, so this is exact), and
    //   (B) nothing left in c1's wave can beat c2:  V2(wave(c1)) < temp[c2], where V2 is an upper bound
    //       of the wave's second-largest min-distance (values only decrease, so a stale V2 stays valid);
    // every other wave's candidate already ranks below c2 in the (value desc, key asc) order.  By
    // induction the first Tn candidates of that order are the next Tn samples as long as every pair
    // (s < t) passes (A) and (B).  Each round therefore sorts the <= 16 wave candidates (rank by
    // counting), checks the 8 x 8 conflict matrix with one lane per pair, commits the longest
    // conflict-free prefix and applies all its updates before the next barrier.  Any doubt (ties, NaNs,
    // a stale bound) only shortens the prefix; the result is bit-identical to one-at-a-time selection.
    // =====================================================================================
    __shared__ FpsEntry2 exch2[2][16];
    __shared__ FpsEntry2 srt[TMAX];
    __shared__ int res_tn;
    int cM = -1, cslot = -1, cV2 = -1;
    uint32_t ckey = KEY_NONE;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    bool cand_ok = false;
    // committed samples of the current round: lane u holds sample u (read back with v_readlane)
    float qxv = P[0], qyv = P[1], qzv = P[2];
    int Tn = 1, par = 0;
#ifdef GEOT_LAB_STAMPS
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, t0, t1, rounds = 0;
    unsigned long long w0 = __builtin_readcyclecounter(), wredo = 0, nredo = 0;
    __shared__ unsigned int lab_max;
    if (tid == 0) lab_max = 0;
#define GEOT_STAMP(acc) do { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; } while (0)
#else
#define GEOT_STAMP(acc) do {} while (0)
#endif
#ifdef GEOT_LAB_STATS
#define GEOT_APPLY_STAT(mask) \
    if (lane == 0) { atomicAdd(&geot_fps_dbg[0], (unsigned long long)__popcll(mask)); atomicAdd(&geot_fps_dbg[1], 1ull); }
#else
#define GEOT_APPLY_STAT(mask)
#endif
    // one committed sample: box test by lanes (= slots), then update of the surviving slots
#define GEOT_APPLY(AX, AY, AZ, REDO)                                                                   \
    do {                                                                                               \
        float dx_ = fmaxf(fmaxf(bx0 - (AX), (AX) - bx1), 0.f);                                         \
        float dy_ = fmaxf(fmaxf(by0 - (AY), (AY) - by1), 0.f);                                         \
        float dz_ = fmaxf(fmaxf(bz0 - (AZ), (AZ) - bz1), 0.f);                                         \
        float lb2_ = dx_ * dx_ + dy_ * dy_ + dz_ * dz_;                                                \
        unsigned long long mask_ = __ballot(!(lb2_ > __int_as_float(smax) * 1.00001f));                \
        GEOT_APPLY_STAT(mask_)                                                                         \
        REDO = REDO || (cslot >= 0 && ((mask_ >> cslot) & 1ull));                                      \
        while (mask_) {                                                                                \
            int s_ = __builtin_ctzll(mask_);                                                           \
            mask_ &= mask_ - 1;                                                                        \
            float d_ = sqdist3(X.get(s_), Y.get(s_), Z.get(s_), (AX), (AY), (AZ));                     \
            float d2_ = fmin_raw(d_, D.get(s_));                                                       \
            D.set(s_, d2_);                                                                            \
            set_lane(smax, __builtin_amdgcn_readfirstlane(wave_max_i32_fast(__float_as_int(d2_))), s_); \
        }                                                                                              \
    } while (0)

    // The loop body alternates two phases around ONE apply site (LLVM keeps the 4 x 32 point registers
    // in place only if they are written at a single static location; a second site makes it copy
    // whole 32-register vectors around the branch):
    //   phase 0: [apply the rest of last round's samples] search, publish, barrier, then wave 0 ranks the
    //            candidates while the other waves fetch the top one -- the next sample for certain -- and
    //   phase 1: [apply it, overlapping wave 0's ranking] barrier, read the committed prefix.
    int j = 1, phase = 0, ua = 0, ub = 1;
    bool redo_acc = false, early = false, done = false;
    for (;;) {
#ifdef GEOT_LAB_STAMPS
        t0 = __builtin_readcyclecounter();
#endif
#pragma unroll 1
        for (int u = ua; u < ub; ++u) {
            const float ax = read_lane(qxv, u), ay = read_lane(qyv, u), az = read_lane(qzv, u);
            GEOT_APPLY(ax, ay, az, redo_acc);
        }
        if (done) break;
        if (phase == 0) {
            GEOT_STAMP(tA);
#ifdef GEOT_LAB_STAMPS
            ++rounds;
#endif
            // -- this wave's candidate + runner-up bound: recomputed only when its slot was touched
            if (!cand_ok || redo_acc) {
                cand_ok = true;
                int m2; // max and (multiset) runner-up of the per-slot maxima in one butterfly
                wave_top2_i32(smax, cM, m2);
                ckey = KEY_NONE;
                cslot = -1;
                cV2 = -1;
                int ins = (int)0x80000000;
                if (cM >= 0) {
                    unsigned long long cm = __ballot(smax == cM);
                    while (cm) {
                        int s = __builtin_ctzll(cm);
                        cm &= cm - 1;
                        const int dbits = __float_as_int(D.get(s));
                        bool hit = dbits == cM;
                        int tl = tid;
                        asm volatile("" : "+v"(tl)); // keep the perm address out of the loop-carried VGPRs
                        uint32_t key = hit ? fps_key(perm[s * NT + tl], L) : KEY_NONE;
                        int d1, d2;
                        uint32_t kmin; // tie key among the hits + runner-up inside the slot, one butterfly
                        wave_top2_i32_min_u32(dbits, key, d1, d2, kmin);
                        if (kmin < ckey) {
                            ckey = kmin;
                            cslot = s;
                            ins = d2;
                            const int owner = __builtin_ctzll(__ballot(key == kmin));
                            cx = read_lane(X.get(s), owner);
                            cy = read_lane(Y.get(s), owner);
                            cz = read_lane(Z.get(s), owner);
                        }
                    }
                }
                // runner-up bound: the best of the other slots' maxima (= m2: equal maxima count twice) and of
                // the candidate slot without its owner (= the slot's second value, again as a multiset)
                if (ckey == KEY_NONE) cM = -1;
                else cV2 = max(max(m2, ins), -1);
            }
#ifdef GEOT_LAB_STAMPS
            {
                unsigned long long work = __builtin_readcyclecounter() - w0;
                if (redo_acc) { wredo += work; ++nredo; }
                if (lane == 0) atomicMax(&lab_max, (unsigned int)work);
            }
#endif
            redo_acc = false;
            GEOT_STAMP(tB);
            if (lane == 0) {
                FpsEntry2 *mine = &exch2[par][wave];
                mine->hi = (uint32_t)(cM + 1); mine->lo = ~ckey;
                mine->x = cx; mine->y = cy; mine->z = cz; mine->v2 = cV2;
                lds_max_u64(&best[par], ((unsigned long long)(uint32_t)(cM + 1) << 32) | (uint32_t)~ckey);
            }
            __syncthreads();
            GEOT_STAMP(tC);
            const FpsEntry2 *ex = exch2[par];
            // Opaque copy of the lane id: everything derived from it below (LDS addresses, lane predicates)
            // is loop-invariant, and LLVM would hoist a dozen such values into VGPRs this kernel does not
            // have (128 of 168 hold the points); recomputing them costs a few VALU per round.
            int ln = lane;
            asm volatile("" : "+v"(ln));
            early = false;
            ua = 0; ub = 0;
            if (wave != 0) {
                if (j + 1 < m) { // otherwise the top candidate is the LAST sample, which is never applied
                    // keys are unique, so the low word of the arg-max identifies the winning wave
                    const unsigned long long bw = best[par];
                    const uint32_t elo = ex[ln & 15].lo;
                    const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(bw >> 32));
                    const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)bw);
                    if (bhi != 0u) {
                        const int wl = __builtin_ctzll(__ballot(elo == blo) & ((1ull << NW) - 1ull));
                        qxv = ex[wl].x; qyv = ex[wl].y; qzv = ex[wl].z; // every lane: only lane 0 is read back
                        early = true;
                        ub = 1;
                    }
                }
            } else {
                // -- wave 0 alone ranks the wave candidates and builds the conflict matrix (12 waves doing
                //    this redundantly would just fight over the VALU issue slots)
                FpsEntry2 e = ex[ln & 15];
                if ((ln & 15) >= NW) { e.hi = 0u; e.lo = 0u; }
                const unsigned long long mine64 = ((unsigned long long)e.hi << 32) | e.lo;
                int rank = 0; // (hi:lo) descending; rank = number of strictly better candidates
#pragma unroll
                for (int w = 0; w < NW; ++w) { // the other candidates come over v_readlane, not LDS
                    const unsigned long long o =
                        ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)e.hi, w) << 32) |
                        (uint32_t)__builtin_amdgcn_readlane((int)e.lo, w);
                    rank += o > mine64 ? 1 : 0;
                }
                const int nvalid = __popcll(__ballot(ln < NW && e.hi != 0u));
                // sort: lane w pushes its entry to lane rank(w) (ds_permute; lanes >= 16 hold copies and push
                // the same values to the same place; invalid entries all rank nvalid, outside the prefix)
                const int dst = rank << 2;
                const uint32_t s_hi = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)e.hi);
                const uint32_t s_lo = (uint32_t)__builtin_amdgcn_ds_permute(dst, (int)e.lo);
                const float s_x = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(e.x)));
                const float s_y = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(e.y)));
                const float s_z = __int_as_float(__builtin_amdgcn_ds_permute(dst, __float_as_int(e.z)));
                const int s_v2 = __builtin_amdgcn_ds_permute(dst, e.v2);
                if (ln < TMAX) { srt[ln].x = s_x; srt[ln].y = s_y; srt[ln].z = s_z; } // for the other waves
                // conflict matrix: lane p = (t = p / 8, s = p % 8), s < t; operands pulled with ds_bpermute
                const int ct = ln >> 3, cs = ln & 7;
                const float a_x = __int_as_float(__builtin_amdgcn_ds_bpermute(cs << 2, __float_as_int(s_x)));
                const float a_y = __int_as_float(__builtin_amdgcn_ds_bpermute(cs << 2, __float_as_int(s_y)));
                const float a_z = __int_as_float(__builtin_amdgcn_ds_bpermute(cs << 2, __float_as_int(s_z)));
                const int a_v2 = __builtin_amdgcn_ds_bpermute(cs << 2, s_v2);
                const float b_x = __int_as_float(__builtin_amdgcn_ds_bpermute(ct << 2, __float_as_int(s_x)));
                const float b_y = __int_as_float(__builtin_amdgcn_ds_bpermute(ct << 2, __float_as_int(s_y)));
                const float b_z = __int_as_float(__builtin_amdgcn_ds_bpermute(ct << 2, __float_as_int(s_z)));
                const int Mt = __builtin_amdgcn_ds_bpermute(ct << 2, (int)s_hi) - 1;
                bool conflict = false;
                if (cs < ct && ct < nvalid && ct < TMAX) {
                    const float d = sqdist3(b_x, b_y, b_z, a_x, a_y, a_z); // point first, sample second: as the update
                    // Mt <= 0: a candidate whose min-distance is exactly 0 (every valid point already sampled) ties with
                    // the sample committed before it, which keeps its smaller key and is picked AGAIN by the sequential
                    // algorithm -- so nothing may be committed behind it
                    conflict = !(__float_as_int(d) >= Mt && d >= 0.f) || !(a_v2 < Mt) || Mt <= 0;
                }
                const unsigned long long cmask = __ballot(conflict);
                int tn = 1;
                if (nvalid > 0) {
                    int lim = min(min(nvalid, TMAX), m - j);
                    // first row t >= 1 with any conflict bit: fold every byte of the matrix onto its bit 0
                    unsigned long long rows = cmask;
                    rows |= rows >> 4; rows |= rows >> 2; rows |= rows >> 1;
                    rows &= 0x0101010101010100ull;
                    int first_bad = rows ? (__builtin_ctzll(rows) >> 3) : 8;
                    tn = min(lim, first_bad);
                    if (ln < tn) out[j + ln] = base + (int)fps_key_decode(~s_lo, L);
                } else { // nobody has a candidate (e.g. every point origin-skipped): the sample is index 0
                    if (ln == 0) {
                        out[j] = base;
                        srt[0].x = P[0]; srt[0].y = P[1]; srt[0].z = P[2];
                    }
                }
                if (ln == 0) res_tn = tn;
            }
            phase = 1;
        } else {
            __syncthreads();
            if (tid == 0) best[par] = 0ull; // next used by the publish of round + 2, i.e. after the next barrier
            par ^= 1;
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const FpsEntry2 *cq = &srt[ln < TMAX ? ln : 0]; // lane u <- sample u (stale beyond Tn: unused)
            qxv = cq->x; qyv = cq->y; qzv = cq->z;
            Tn = __builtin_amdgcn_readfirstlane(res_tn);
#ifdef GEOT_LAB_STATS
            if (tid == 0) { atomicAdd(&geot_fps_dbg[2], (unsigned long long)Tn); atomicAdd(&geot_fps_dbg[3], 1ull); }
#endif
            j += Tn;
            // The reference's temp buffer ends as "min-distance to samples 0..m-2": the last pick of the
            // whole run is never applied.
            done = j >= m;
            ua = early ? 1 : 0;
            ub = done ? Tn - 1 : Tn;
            phase = 0;
            GEOT_STAMP(tD);
#ifdef GEOT_LAB_STAMPS
            if (tid == 0) { tE += lab_max; lab_max = 0; } // lab_max was final at barrier 1; next atomics come after this
            w0 = __builtin_readcyclecounter();
#endif
        }
    }
#undef GEOT_APPLY
#undef GEOT_APPLY_STAT
#ifdef GEOT_LAB_STAMPS
    if (lane == 0) {
        atomicAdd(&geot_fps_dbg[0], tA); atomicAdd(&geot_fps_dbg[1], tB); atomicAdd(&geot_fps_dbg[2], tC);
        atomicAdd(&geot_fps_dbg[3], tD); atomicAdd(&geot_fps_dbg[4], tE); atomicAdd(&geot_fps_dbg[5], (unsigned long long)(m - 1));
        atomicAdd(&geot_fps_dbg[7], (wredo << 24) | nredo); // lab only: both fit (cycles < 2^40, rounds < 2^24)
        if (tid == 0) atomicAdd(&geot_fps_dbg[6], rounds);
    }
#endif
#undef GEOT_STAMP
    } else {
    // This wave's candidate, wave-uniform, carried across rounds.  Min-distances only ever
    // decrease, so it stays the wave's arg-max until its own slot (cslot) is updated.
    int cM = -1, cslot = -1;
    uint32_t ckey = KEY_NONE;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    bool cand_ok = false;

#ifdef GEOT_LAB_STAMPS
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, t0, t1;
#define GEOT_STAMP(acc) do { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; } while (0)
#else
#define GEOT_STAMP(acc) do {} while (0)
#endif
    float qx = P[0], qy = P[1], qz = P[2];
    int r3 = 1; // j % 3
    for (int j = 1; j < m; ++j) {
#ifdef GEOT_LAB_STAMPS
        t0 = __builtin_readcyclecounter();
#endif
        // -- which of my wave's slots can the new sample change?
        float dx = fmaxf(fmaxf(bx0 - qx, qx - bx1), 0.f);
        float dy = fmaxf(fmaxf(by0 - qy, qy - by1), 0.f);
        float dz = fmaxf(fmaxf(bz0 - qz, qz - bz1), 0.f);
        float lb2 = dx * dx + dy * dy + dz * dz;
        bool act = !(lb2 > __int_as_float(smax) * 1.00001f);
        unsigned long long mask = __ballot(act);
#ifdef GEOT_LAB_STATS
        if (lane == 0) {
            atomicAdd(&geot_fps_dbg[0], (unsigned long long)__popcll(mask));
            atomicAdd(&geot_fps_dbg[1], 1ull);
            if (mask) atomicAdd(&geot_fps_dbg[2], 1ull);
            if (!cand_ok || (cslot >= 0 && ((mask >> cslot) & 1ull))) atomicAdd(&geot_fps_dbg[3], 1ull);
        }
#endif
        const bool redo = !cand_ok || (cslot >= 0 && ((mask >> cslot) & 1ull));
        GEOT_STAMP(tA);
        while (mask) {
            int s = __builtin_ctzll(mask);
            mask &= mask - 1;
            float d = sqdist3(X.get(s), Y.get(s), Z.get(s), qx, qy, qz);
            float d2 = fmin_raw(d, D.get(s));
            D.set(s, d2);
            set_lane(smax, __builtin_amdgcn_readfirstlane(wave_max_i32_fast(__float_as_int(d2))), s);
        }
        GEOT_STAMP(tB);
        // -- this wave's candidate: recomputed only when its slot was touched
        if (redo) {
            cand_ok = true;
            cM = __builtin_amdgcn_readfirstlane(wave_max_i32_fast(smax));
            ckey = KEY_NONE;
            cslot = -1;
            if (cM >= 0) {
                unsigned long long cm = __ballot(smax == cM);
                while (cm) {
                    int s = __builtin_ctzll(cm);
                    cm &= cm - 1;
                    bool hit = __float_as_int(D.get(s)) == cM;
                    uint32_t key = hit ? fps_key(perm[s * NT + tid], L) : KEY_NONE;
                    uint32_t kmin = __builtin_amdgcn_readfirstlane(wave_min_u32_fast(key));
                    if (kmin < ckey) {
                        ckey = kmin;
                        cslot = s;
                        int owner = __builtin_ctzll(__ballot(key == kmin));
                        cx = read_lane(X.get(s), owner);
                        cy = read_lane(Y.get(s), owner);
                        cz = read_lane(Z.get(s), owner);
                    }
                }
            }
            if (ckey == KEY_NONE) cM = -1;
        }
        GEOT_STAMP(tC);
        // -- publish: one LDS atomic max on (M+1 : ~key) does the block arg-max with the
        //    reference tie rule; the entry carries the coordinates for the next round
        if (lane == 0) {
            FpsEntry *mine = &exch[j & 1][wave];
            mine->key = ckey; mine->x = cx; mine->y = cy; mine->z = cz;
            unsigned long long packed = ((unsigned long long)(uint32_t)(cM + 1) << 32) | (uint32_t)~ckey;
            lds_max_u64(&best[r3], packed);
        }
        __syncthreads();
        GEOT_STAMP(tD);
        const unsigned long long bw = best[r3];
        const FpsEntry e = exch[j & 1][lane & 15];
        int rz = r3 + 2; rz = rz >= 3 ? rz - 3 : rz;
        if (tid == 0) best[rz] = 0ull; // slot of round j+2: nobody can touch it before barrier j+1
        r3 = r3 == 2 ? 0 : r3 + 1;
        const uint32_t bhi = __builtin_amdgcn_readfirstlane((uint32_t)(bw >> 32));
        const uint32_t kg = ~__builtin_amdgcn_readfirstlane((uint32_t)bw);
        uint32_t old;
        if (bhi == 0u) { // no wave has a candidate (e.g. every point origin-skipped)
            old = 0;
            qx = P[0]; qy = P[1]; qz = P[2];
        } else {
            old = fps_key_decode(kg, L);
            int wl = __builtin_ctzll(__ballot(e.key == kg) & ((1ull << NW) - 1ull));
            qx = read_lane(e.x, wl);
            qy = read_lane(e.y, wl);
            qz = read_lane(e.z, wl);
        }
        if (tid == 0) out[j] = base + (int)old;
        GEOT_STAMP(tE);
    }
#ifdef GEOT_LAB_STAMPS
    if (lane == 0) {
        atomicAdd(&geot_fps_dbg[0], tA); atomicAdd(&geot_fps_dbg[1], tB); atomicAdd(&geot_fps_dbg[2], tC);
        atomicAdd(&geot_fps_dbg[3], tD); atomicAdd(&geot_fps_dbg[4], tE); atomicAdd(&geot_fps_dbg[5], (unsigned long long)(m - 1));
    }
#endif
    }
#pragma unroll 1
    for (int i = 0; i < PPT; ++i) {
        int pos = i * NT + tid;
        float t = D.get(i);
        if (pos < n && t >= 0.f) T[perm[pos]] = t; // skipped / padded slots hold -1
    }
}

template <bool SKIP, int TMAX>
static hipError_t fps_pruned_launch(int b, int n_max, const float *xyz, const int *offset,
                                    const int *new_offset, int n_dense, int m_dense, float *temp,
                                    int *idxs, int L, hipStream_t s)
{
#define GEOT_FPP_CASE(NT, P)                                                                        \
    hipLaunchKernelGGL((fps_pruned_kernel<NT, P, SKIP, TMAX>), dim3(b), dim3(NT), 0, s, xyz, offset, \
                       new_offset, n_dense, m_dense, temp, idxs, L)
    if (n_max <= 8 * 512) GEOT_FPP_CASE(512, 8);
    else if (n_max <= 16 * 512) GEOT_FPP_CASE(512, 16);
    else if (n_max <= 32 * 512) GEOT_FPP_CASE(512, 32);
    else GEOT_FPP_CASE(768, 32);
#undef GEOT_FPP_CASE
    return hipGetLastError();
}

// GEOT_FPS_IMPL (read per call so tests can A/B the kernels): "basic" = unpruned, "single" = pruned,
// one sample per round; default = pruned with multi-commit rounds.  Returns 0 / 1 / 2.
static int fps_impl(int n_max, bool weighted)
{
    const char *e = getenv("GEOT_FPS_IMPL");
    if ((e && e[0] == 'b') || weighted || n_max < 1024 || n_max > FP_MAX_N) return 0;
    return (e && e[0] == 's') ? 1 : 2;
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

#if defined(GEOT_LAB_STATS) || defined(GEOT_LAB_STAMPS)
GEOT_EXPORT int geot_lab_read_stats(unsigned long long *out8, int reset)
{
    hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(geot::geot_fps_dbg), 8 * sizeof(unsigned long long));
    if (e == hipSuccess && reset) {
        unsigned long long z[8] = {0};
        e = hipMemcpyToSymbol(HIP_SYMBOL(geot::geot_fps_dbg), z, sizeof(z));
    }
    return e;
}
#endif

GEOT_EXPORT int geot_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                             int *idxs, int block_cap, int skip_origin, void *stream)
{
    if (b < 0 || n < 0 || m < 0 || (block_cap != 512 && block_cap != 1024)) return hipErrorInvalidValue;
    if (b == 0 || n == 0 || m == 0) return hipSuccess;
    int L = geot::ref_log2_block(n, block_cap);
    hipStream_t s = (hipStream_t)stream;
    const int impl = geot::fps_impl(n, false);
    if (impl == 2) {
        if (skip_origin)
            return geot::fps_pruned_launch<true, geot::FP_TMAX>(b, n, xyz, nullptr, nullptr, n, m, temp, idxs, L, s);
        return geot::fps_pruned_launch<false, geot::FP_TMAX>(b, n, xyz, nullptr, nullptr, n, m, temp, idxs, L, s);
    }
    if (impl == 1) {
        if (skip_origin)
            return geot::fps_pruned_launch<true, 1>(b, n, xyz, nullptr, nullptr, n, m, temp, idxs, L, s);
        return geot::fps_pruned_launch<false, 1>(b, n, xyz, nullptr, nullptr, n, m, temp, idxs, L, s);
    }
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
    const int impl = geot::fps_impl(n_max, weights != nullptr);
    if (impl == 2)
        return geot::fps_pruned_launch<false, geot::FP_TMAX>(b, n_max, xyz, offset, new_offset, 0, 0, tmp, idx, L, s);
    if (impl == 1)
        return geot::fps_pruned_launch<false, 1>(b, n_max, xyz, offset, new_offset, 0, 0, tmp, idx, L, s);
    if (weights)
        return geot::fps_launch<false, true>(b, n_max, xyz, offset, new_offset, 0, 0, weights, tmp, idx, L, s);
    return geot::fps_launch<false, false>(b, n_max, xyz, offset, new_offset, 0, 0, nullptr, tmp, idx, L, s);
}
