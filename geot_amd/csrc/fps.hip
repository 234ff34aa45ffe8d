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
//    lanes of the winning slot(s); the owning lane publishes (max, key, x, y, z) so the next
//    round needs no memory access at all -- one LDS hop and ONE barrier per round.
//  The temps, the selected indices and the tie-breaking are bit-identical to the unpruned
//  kernel (and to the reference): skipping is only ever a proven no-op.
// ===========================================================================
constexpr int FP_THREADS = 512;
constexpr int FP_WAVES = FP_THREADS / 64;
constexpr int FP_CELLS = 4096;
constexpr int FP_MAX_PPT = 47;

struct FpsExch {
    int M;
    uint32_t key;
    float x, y, z;
    float pad[3];
};

template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// Run f(slot) for every set bit of the wave-uniform `mask` with the slot as a COMPILE-TIME
// constant (register arrays cannot be indexed dynamically).  Two-level test: one scalar
// branch per group of 8 slots, then one per slot inside a non-empty group, so the common
// case (0-2 active slots out of PPT) costs ~PPT/8 + 8 scalar tests.  Straight-line,
// structured code: a 47-way switch made the register allocator spill.
template <int PPT, typename F>
__device__ __forceinline__ void fp_for_each_slot(unsigned long long mask, F &&f)
{
    static_for<0, (PPT + 7) / 8>([&](auto G) {
        constexpr int g = decltype(G)::value;
        if ((mask >> (8 * g)) & 0xFFull) {
            static_for<8 * g, (8 * g + 8 < PPT ? 8 * g + 8 : PPT)>([&](auto S) {
                constexpr int sl = decltype(S)::value;
                if ((mask >> sl) & 1ull) f(S);
            });
        }
    });
}

// v[lane LANE] = value (wave-uniform); v_writelane_b32 via asm (no builtin in this toolchain).
template <int LANE>
__device__ __forceinline__ void set_lane(int &v, int value)
{
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(value), "n"(LANE));
}
template <int LANE>
__device__ __forceinline__ void set_lane(float &v, float value)
{
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(value), "n"(LANE));
}
__device__ __forceinline__ float uniform(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

__device__ __forceinline__ uint32_t morton12(uint32_t cx, uint32_t cy, uint32_t cz)
{
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
        m |= (((cx >> b) & 1u) << (3 * b)) | (((cy >> b) & 1u) << (3 * b + 1)) | (((cz >> b) & 1u) << (3 * b + 2));
    return m;
}

template <int PPT, bool SKIP>
__global__ __launch_bounds__(FP_THREADS) void fps_pruned_kernel(
    const float *__restrict__ xyz, const int *__restrict__ offset,
    const int *__restrict__ new_offset, int n_dense, int m_dense, float *__restrict__ temp,
    int *__restrict__ idxs, int L)
{
    static_assert(PPT <= FP_MAX_PPT, "one slot per lane for the box test");
    __shared__ uint16_t perm[PPT * FP_THREADS];
    __shared__ uint32_t cellcnt[FP_CELLS];
    __shared__ FpsExch exch[2][FP_WAVES];
    __shared__ float red[FP_WAVES][6];
    __shared__ uint32_t wsum[FP_WAVES];

    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    for (int k = tid; k < n; k += FP_THREADS) {
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
    for (int c = tid; c < FP_CELLS; c += FP_THREADS) cellcnt[c] = 0;
    __syncthreads();
    float inv[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float l = red[0][a], h = red[0][3 + a];
        for (int w = 1; w < FP_WAVES; ++w) { l = fminf(l, red[w][a]); h = fmaxf(h, red[w][3 + a]); }
        lo[a] = l;
        float ext = h - l;
        inv[a] = (ext > 0.f && ext < INFINITY) ? 16.f / ext : 0.f;
    }

    // ---- 2. Morton cell histogram; remember (cell, rank-in-cell) per point -----------------
    uint32_t cr[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int k = i * FP_THREADS + tid;
        cr[i] = 0;
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
            cr[i] = (cell << 16) | r;
        }
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0); // bound look-ahead: VGPR pressure
    }
    __syncthreads();
    // ---- 3. exclusive prefix over the 4096 cells (8 per thread) ------------------------------
    {
        uint32_t v[8], s = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[e] = cellcnt[tid * 8 + e]; s += v[e]; }
        uint32_t inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        uint32_t run = woff + inc - s;
#pragma unroll
        for (int e = 0; e < 8; ++e) { cellcnt[tid * 8 + e] = run; run += v[e]; }
    }
    __syncthreads();
    // ---- 4. scatter: sorted position -> original (local) index ---------------------------------
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int k = i * FP_THREADS + tid;
        if (k < n) perm[cellcnt[cr[i] >> 16] + (cr[i] & 0xFFFFu)] = (uint16_t)k;
    }
    __syncthreads();

    // ---- 5. gather the sorted points into registers ----------------------------------------------
    float px[PPT], py[PPT], pz[PPT], t[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int pos = i * FP_THREADS + tid;
        bool in = pos < n;
        int k = in ? (int)perm[pos] : 0;
        px[i] = in ? P[k * 3 + 0] : 0.f;
        py[i] = in ? P[k * 3 + 1] : 0.f;
        pz[i] = in ? P[k * 3 + 2] : 0.f;
        t[i] = in ? T[k] : -1.f;
        if (SKIP && in && origin_skipped(px[i], py[i], pz[i])) t[i] = -1.f;
        if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0); // bound the scheduler's look-ahead (VGPR pressure)
    }
    // ---- 6. per-slot boxes and maxima: lane i owns slot i of its wave ------------------------------
    float bx0 = 0.f, by0 = 0.f, bz0 = 0.f, bx1 = 0.f, by1 = 0.f, bz1 = 0.f;
    int smax = -1;
    static_for<0, PPT>([&](auto I) {
        constexpr int i = decltype(I)::value;
        bool valid = t[i] >= 0.f;
        set_lane<i>(bx0, uniform(wave_min_f32(valid ? px[i] : INFINITY)));
        set_lane<i>(bx1, uniform(wave_max_f32(valid ? px[i] : -INFINITY)));
        set_lane<i>(by0, uniform(wave_min_f32(valid ? py[i] : INFINITY)));
        set_lane<i>(by1, uniform(wave_max_f32(valid ? py[i] : -INFINITY)));
        set_lane<i>(bz0, uniform(wave_min_f32(valid ? pz[i] : INFINITY)));
        set_lane<i>(bz1, uniform(wave_max_f32(valid ? pz[i] : -INFINITY)));
        set_lane<i>(smax, __builtin_amdgcn_readfirstlane(wave_max_i32(__float_as_int(t[i]))));
        __builtin_amdgcn_sched_barrier(0);
    });

    float qx = P[0], qy = P[1], qz = P[2];
    for (int j = 1; j < m; ++j) {
        // -- which of my wave's slots can the new sample change?
        float dx = fmaxf(fmaxf(bx0 - qx, qx - bx1), 0.f);
        float dy = fmaxf(fmaxf(by0 - qy, qy - by1), 0.f);
        float dz = fmaxf(fmaxf(bz0 - qz, qz - bz1), 0.f);
        float lb2 = dx * dx + dy * dy + dz * dz;
        bool act = lane < PPT && smax >= 0 && !(lb2 > __int_as_float(smax) * 1.00001f);
        unsigned long long mask = __ballot(act);
        {
            fp_for_each_slot<PPT>(mask, [&](auto I) {
                constexpr int s = decltype(I)::value;
                // opaque copies: without them LICM hoists all PPT distance evaluations out of
                // the mask loop (they only depend on q), which is exactly the work we prune
                float ax = qx, ay = qy, az = qz;
                asm volatile("" : "+s"(ax), "+s"(ay), "+s"(az));
                float d = sqdist3(px[s], py[s], pz[s], ax, ay, az);
                float d2 = fmin_raw(d, t[s]);
                t[s] = d2;
                int sm = __builtin_amdgcn_readfirstlane(wave_max_i32(__float_as_int(d2)));
                // lane s of every wave owns slot s: one v_writelane instead of a compare + select
                set_lane<s>(smax, sm);
            });
        }
        // -- this wave's candidate
        int wM = __builtin_amdgcn_readfirstlane(wave_max_i32(lane < PPT ? smax : -1));
        uint32_t wkey = KEY_NONE;
        // volatile: keeps each slot's publish store inside its own branch (merging the PPT
        // conditional stores into one costs ~3 VGPRs per slot in phi copies)
        volatile FpsExch *mine = &exch[j & 1][wave];
        if (wM >= 0) {
            unsigned long long cm = __ballot(lane < PPT && smax == wM);
            {
                fp_for_each_slot<PPT>(cm, [&](auto I) {
                    constexpr int s = decltype(I)::value;
                    int wMo = wM;
                    asm volatile("" : "+s"(wMo)); // keep the per-slot work inside the mask loop (see above)
                    bool hit = __float_as_int(t[s]) == wMo;
                    // `zero` is opaque so the (round-invariant) perm read + key computation is not
                    // hoisted out of the round loop for all PPT slots (+1 live VGPR per point)
                    int zero = 0;
                    asm volatile("" : "+s"(zero));
                    uint32_t key = KEY_NONE;
                    if (hit) key = fps_key(perm[s * FP_THREADS + tid + zero], L);
                    uint32_t kmin = wave_min_u32(key);
                    if (kmin < wkey) {
                        wkey = kmin;
                        if (key == kmin) {
                            mine->M = wM; mine->key = kmin;
                            mine->x = px[s]; mine->y = py[s]; mine->z = pz[s];
                        }
                    }
                });
            }
        }
        if (wkey == KEY_NONE && lane == 0) { mine->M = -1; mine->key = KEY_NONE; }
        __syncthreads();
        // -- block winner: max M, then min key; its coordinates become the next q
        const FpsExch e = exch[j & 1][lane & (FP_WAVES - 1)];
        int Mg = e.M;
        Mg = max(Mg, (int)dpp_mov<DPP_QUAD_XOR1>((uint32_t)Mg));
        Mg = max(Mg, (int)dpp_mov<DPP_QUAD_XOR2>((uint32_t)Mg));
        Mg = max(Mg, (int)dpp_mov<DPP_ROW_HALF_MIRROR>((uint32_t)Mg));
        uint32_t kg = e.M == Mg ? e.key : KEY_NONE;
        kg = min(kg, dpp_mov<DPP_QUAD_XOR1>(kg));
        kg = min(kg, dpp_mov<DPP_QUAD_XOR2>(kg));
        kg = min(kg, dpp_mov<DPP_ROW_HALF_MIRROR>(kg));
        kg = __builtin_amdgcn_readfirstlane(kg);
        uint32_t old;
        if (kg == KEY_NONE) {
            old = 0;
            qx = P[0]; qy = P[1]; qz = P[2];
        } else {
            old = fps_key_decode(kg, L);
            unsigned long long wm = __ballot(e.M == Mg && e.key == kg);
            int wl = __builtin_ctzll(wm);
            qx = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e.x), wl));
            qy = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e.y), wl));
            qz = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e.z), wl));
        }
        if (tid == 0) out[j] = base + (int)old;
    }
    // Re-read perm behind a compiler barrier: otherwise the gather's index and 64-bit address
    // per slot stay live across the whole round loop (+3 VGPRs per point).
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        int pos = i * FP_THREADS + tid;
        if (pos < n && t[i] >= 0.f) T[perm[pos]] = t[i]; // skipped / padded slots hold -1
    }
}

template <bool SKIP>
static hipError_t fps_pruned_launch(int b, int n_max, const float *xyz, const int *offset,
                                    const int *new_offset, int n_dense, int m_dense, float *temp,
                                    int *idxs, int L, hipStream_t s)
{
#define GEOT_FPP_CASE(P)                                                                            \
    hipLaunchKernelGGL((fps_pruned_kernel<P, SKIP>), dim3(b), dim3(FP_THREADS), 0, s, xyz, offset,  \
                       new_offset, n_dense, m_dense, temp, idxs, L)
    if (n_max <= 4 * FP_THREADS) GEOT_FPP_CASE(4);
    else if (n_max <= 8 * FP_THREADS) GEOT_FPP_CASE(8);
    else if (n_max <= 16 * FP_THREADS) GEOT_FPP_CASE(16);
    else if (n_max <= 24 * FP_THREADS) GEOT_FPP_CASE(24);
    else if (n_max <= 32 * FP_THREADS) GEOT_FPP_CASE(32);
    else GEOT_FPP_CASE(47);
#undef GEOT_FPP_CASE
    return hipGetLastError();
}

// GEOT_FPS_IMPL=basic forces the unpruned kernels (A/B testing); default = pruned when it applies.
static bool fps_use_pruned(int n_max, bool weighted)
{
    const char *e = getenv("GEOT_FPS_IMPL"); // read per call so tests can A/B both kernels
    bool pruned = !(e && e[0] == 'b');
    return pruned && !weighted && n_max >= 1024 && n_max <= FP_MAX_PPT * FP_THREADS;
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
    if (geot::fps_use_pruned(n, false)) {
        if (skip_origin)
            return geot::fps_pruned_launch<true>(b, n, xyz, nullptr, nullptr, n, m, temp, idxs, L, s);
        return geot::fps_pruned_launch<false>(b, n, xyz, nullptr, nullptr, n, m, temp, idxs, L, s);
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
    if (geot::fps_use_pruned(n_max, weights != nullptr))
        return geot::fps_pruned_launch<false>(b, n_max, xyz, offset, new_offset, 0, 0, tmp, idx, L, s);
    if (weights)
        return geot::fps_launch<false, true>(b, n_max, xyz, offset, new_offset, 0, 0, weights, tmp, idx, L, s);
    return geot::fps_launch<false, false>(b, n_max, xyz, offset, new_offset, 0, 0, nullptr, tmp, idx, L, s);
}
