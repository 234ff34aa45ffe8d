// geot_common.h -- shared device helpers for the gfx950 kernels.
// All translation units are built with -ffp-contract=off: squared distances
// must be un-contracted IEEE fp32 so integer results match the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <mutex>
#include <utility>
#include <vector>

#define GEOT_EXPORT extern "C" __attribute__((visibility("default")))
#define GEOT_WAVE 64

namespace geot {

// Squared distance, the ONE expression behind every index-producing comparison of the library.
// GEOT_DISTANCE_MODE (build-time; geot_amd/build.py variants, selected at load time by GEOT_DISTANCE):
//   0 "exact"  ((dx*dx) + (dy*dy)) + (dz*dz), every operation rounded -- the source semantics of the reference
//              kernels (sampling_gpu.cu:106-107, ball_query_gpu.cu:34-35, interpolate_gpu.cu:36, ...) and what a
//              CPU / numpy restatement computes; the default.
//   1 "fma"    fma(dz,dz, fma(dy,dy, dx*dx))  -- what nvcc's default -fmad=true most likely made of that source
//   2 "fma_xy" fma(dz,dz, fma(dx,dx, dy*dy))  -- the other contraction an LLVM-style combiner can choose
//              (the authors' binaries were built -O2 without --use_fast_math, pointnet2/build/.../build.ninja:8; which
//              form their compiler picked cannot be observed here: a maintainer holding those binaries picks the
//              mode whose indices match theirs).  The translation units are compiled with -ffp-contract=off, so
//              the explicit fmaf below is the only fusion.
#ifndef GEOT_DISTANCE_MODE
#define GEOT_DISTANCE_MODE 0
#endif
__device__ __forceinline__ float sqdist3(float ax, float ay, float az, float bx, float by, float bz)
{
    float dx = ax - bx, dy = ay - by, dz = az - bz;
#if GEOT_DISTANCE_MODE == 1
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
#elif GEOT_DISTANCE_MODE == 2
    return fmaf(dz, dz, fmaf(dx, dx, dy * dy));
#else
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
#endif
}

// fminf without the canonicalising v_max_f32 the compiler puts in front of
// llvm.minnum under IEEE mode. v_min_f32 returns the non-NaN operand, which is
// the fminf/CUDA min semantics the reference relies on.
__device__ __forceinline__ float fmin_raw(float a, float b)
{
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// ---- DPP cross-lane moves (no LDS traffic) --------------------------------
// dpp_ctrl encodings (GFX9): quad_perm 0x00-0xFF, row_mirror 0x140,
// row_half_mirror 0x141.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

constexpr int DPP_QUAD_XOR1 = 0xB1;       // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;       // quad_perm [2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;

// After these four steps every lane of a 16-lane row holds the row's result.
__device__ __forceinline__ uint32_t row16_max_u32(uint32_t v)
{
    v = max(v, dpp_mov<DPP_QUAD_XOR1>(v));
    v = max(v, dpp_mov<DPP_QUAD_XOR2>(v));
    v = max(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = max(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ uint32_t row16_min_u32(uint32_t v)
{
    v = min(v, dpp_mov<DPP_QUAD_XOR1>(v));
    v = min(v, dpp_mov<DPP_QUAD_XOR2>(v));
    v = min(v, dpp_mov<DPP_ROW_HALF_MIRROR>(v));
    v = min(v, dpp_mov<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float row16_min_f32(float v)
{
    v = fminf(v, __uint_as_float(dpp_mov<DPP_QUAD_XOR1>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<DPP_QUAD_XOR2>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<DPP_ROW_HALF_MIRROR>(__float_as_uint(v))));
    v = fminf(v, __uint_as_float(dpp_mov<DPP_ROW_MIRROR>(__float_as_uint(v))));
    return v;
}

// Wave-uniform results (returned in SGPRs via readlane).
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    v = row16_max_u32(v);
    uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    v = row16_min_u32(v);
    uint32_t a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    uint32_t c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

__device__ __forceinline__ int row16_max_i32(int v)
{
    v = max(v, (int)dpp_mov<DPP_QUAD_XOR1>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_QUAD_XOR2>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_ROW_HALF_MIRROR>((uint32_t)v));
    v = max(v, (int)dpp_mov<DPP_ROW_MIRROR>((uint32_t)v));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = row16_max_i32(v);
    int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
// float min / max over the wave (used in one-time prologues; the fminf/fmaxf
// canonicalisation cost does not matter there).
__device__ __forceinline__ float wave_min_f32(float v)
{
    v = row16_min_f32(v);
    float a = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 0));
    float b = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
    float c = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
    float d = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
    return fminf(fminf(a, b), fminf(c, d));
}
__device__ __forceinline__ float wave_max_f32(float v) { return -wave_min_f32(-v); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }


// ---- exclusive scan of int counters, in place, on a stream (used to turn per-target counts into CSR
// offsets): data[0..total] (total + 1 entries, the last one receives the grand total); `bsum` is scratch of
// scan_blocks(total) ints; `also_zero` (nullable) gets its first `total` entries cleared on the way.
// exclusive scan of deg[0..total] in three small kernels (block-local scan, scan of the block sums, add)
constexpr int SCAN_CHUNK = 4096;
static __global__ __launch_bounds__(1024) void scan_local_kernel(int total, int *__restrict__ deg, int *__restrict__ bsum)
{
    __shared__ int wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = blockIdx.x * SCAN_CHUNK + tid * 4;
    int v[4], s = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = base + e <= total ? deg[base + e] : 0; s += v[e]; }
    int inc = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + inc - s;
#pragma unroll
    for (int e = 0; e < 4; ++e) { if (base + e <= total) deg[base + e] = run; run += v[e]; }
    if (tid == 1023) bsum[blockIdx.x] = run;
}
static __global__ __launch_bounds__(1024) void scan_top_kernel(int nblk, int *__restrict__ bsum)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nblk; base += 1024) {
        const int v = base + tid < nblk ? bsum[base + tid] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int woff = carry;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (base + tid < nblk) bsum[base + tid] = woff + inc - v;
        __syncthreads();
        if (tid == 1023) carry = woff + inc;
        __syncthreads();
    }
}
static __global__ __launch_bounds__(1024) void scan_add_kernel(int total, int *__restrict__ deg, const int *__restrict__ bsum,
                                                           int *__restrict__ cursor)
{
    const int add = bsum[blockIdx.x];
    const int base = blockIdx.x * SCAN_CHUNK + threadIdx.x * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (base + e <= total) deg[base + e] += add;
        if (cursor && base + e < total) cursor[base + e] = 0;
    }
}

static inline int scan_blocks(long long total) { return (int)((total + 1 + SCAN_CHUNK - 1) / SCAN_CHUNK); }
static inline void exclusive_scan_i32(int total, int *data, int *bsum, int *also_zero, hipStream_t s)
{
    const int nblk = scan_blocks(total);
    hipLaunchKernelGGL(scan_local_kernel, dim3(nblk), dim3(1024), 0, s, total, data, bsum);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, s, nblk, bsum);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nblk), dim3(1024), 0, s, total, data, bsum, also_zero);
}

// ---- launch-side helpers that keep every entry point capturable into a hipGraph ---------------------------------
// (1) > 64 KB of dynamic LDS is opt-in per kernel and device.  The opt-in is raised ONCE, to the most the CU has,
// the first time a kernel needs it -- never re-set per call: a graph node recorded with a large LDS size must not
// find the function's limit lowered by a later call with a smaller one.  Keyed by the kernel's ADDRESS (two
// instantiations with the same signature are one C++ type).
static inline hipError_t allow_big_lds(const void *kernel, size_t lds, int max_bytes = 160 * 1024)
{
    if (lds <= 64 * 1024) return hipSuccess;
    if (lds > (size_t)max_bytes) return hipErrorInvalidValue;
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> guard(mu);
    for (const auto &d : done)
        if (d.first == kernel && d.second == dev) return hipSuccess;
    hipFuncAttributes attr;
    hipError_t e = hipFuncGetAttributes(&attr, kernel);     // the kernel's static LDS counts against the same 160 KB
    if (e != hipSuccess) return e;
    const int room = max_bytes - (int)attr.sharedSizeBytes;
    if ((long long)lds > room) return hipErrorInvalidValue;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, room);
    if (e == hipSuccess) done.emplace_back(kernel, dev);
    return e;
}
// (2) scratch counters are cleared by a kernel, not hipMemsetAsync: a plain kernel node replays identically in a
// graph on every ROCm release; memset nodes of odd sizes in the middle of an allocation have not always.
static __global__ __launch_bounds__(256) void zero_words_kernel(long long count, uint32_t *__restrict__ dst)
{
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long long)gridDim.x * 256) dst[e] = 0u;
}
static inline hipError_t zero_words(void *dst, long long words, hipStream_t s)
{
    if (words <= 0) return hipSuccess;
    long long blocks = (words + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, words, (uint32_t *)dst);
    return hipGetLastError();
}
// (2b) The reverse indices (CSR by target) are first filled through atomically handed-out ranks, so the order of a
// target's entries differs from run to run -- and with it the fp32 summation order of every gather over them.  The
// entry-parallel helper below gives every pair its position in the ascending-pair-id order of its list (a count over
// the list, L2-resident), so the final lists -- and the gathers -- are bit-reproducible.  Hub lists are counted in full
// too (every pair of the list is its own thread: a 30 000-entry list costs each of them 30 000 L2-resident reads, ~0.1 ms
// once per call); only beyond RIX_SORT_MAX entries does a list keep the arbitrary order: correct, not reproducible.
constexpr int RIX_SORT_MAX = 1 << 22;
__device__ __forceinline__ int rix_sorted_position(const int *__restrict__ tmp, int a, int z, int mine, int fallback)
{
    if (z - a > RIX_SORT_MAX) return fallback;
    int r = 0;
    for (int q = a; q < z; ++q) r += tmp[q] < mine ? 1 : 0;
    return r;
}
static inline bool rix_reproducible()
{
    const char *e = getenv("GEOT_REPRODUCIBLE");      // "0": keep the arbitrary (atomic) list order (A/B runs)
    return !(e && e[0] == '0');
}
// (3) CU count of the current device, looked up once per device
static inline int device_cus()
{
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] > 0) return cached[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cached[dev] = n;
    return n;
}

// slot of the current device in the launchers' small per-device caches (occupancy per block size, ...): results of
// hipOccupancy* / device properties are looked up once PER DEVICE, never shared across devices of a mixed node
constexpr int GEOT_DEV_SLOTS = 16;
static inline int device_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev % GEOT_DEV_SLOTS;
}

} // namespace geot
