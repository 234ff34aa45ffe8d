// knn_grid.hip -- exact sorted kNN (k <= 64) over a uniform grid, for gfx950 (MI355X).
//
// Same contract and the same bits as the brute-force kernels in neighbors.hip (knn_cuda.KNN / knn_point
// call sites: openpoints/models/backbone/transformer.py:280,293,313,353, utils/insT_loss.py:69; three_nn:
// pointnet2/_ext_src/src/interpolate_gpu.cu:12-71): the k reference points smallest by (d2, index), d2 the
// un-contracted fp32 ((dx*dx)+(dy*dy))+(dz*dz).  Only the set of pairs that is evaluated changes:
//
//   build (per cloud, 5 small kernels): bounding box -> cubic cells of edge h = max extent / G with
//     G ~ sqrt(3 n / (5 k)) (about k/3 points per occupied cell of a surface-like cloud) -> counting sort of
//     the reference points by cell into (x, y, z, original index) records, x-fastest cell order, so one grid
//     row of cells is one contiguous range of records;
//   query (one wave per query): the 3 x 3 x 3 block of cells around the query is 9 such ranges; the 64 lanes
//     take 64 records per step and candidates are inserted into the wave's sorted best-k list (one entry per
//     lane) exactly as in knn_wave_kernel, ordered by (d2, index) explicitly because records do not arrive
//     in index order.  After ring r every unvisited reference point lies outside the (2r+1)^3 block, i.e. at
//     least `bound` away, where bound is the distance from the query to the nearest block face that still
//     has cells behind it (minus a slack of h/1000 that dwarfs the fp32 rounding of the cell assignment).
//     The search stops once the k-th best d2 is STRICTLY below bound^2 -- ties keep it going, so the (d2,
//     index) order is never decided by what was skipped -- or when the block covers the grid; otherwise it
//     processes the next ring's shell.
//
// At 24 000 x 24 000, k = 33 this evaluates ~250 pairs per query instead of 24 000.
#include "geot_common.h"
#include "geot_hip.h"
#include <cstdlib>
#include <cmath>

namespace geot {

constexpr int KG_GMAX = 32;                           // cells per axis at most
constexpr int KG_CELLS = KG_GMAX * KG_GMAX * KG_GMAX; // counters per cloud (+1)
constexpr int KG_HDR = 16;                            // header words per cloud

// workspace per cloud: [header 16 words][cell_start KG_CELLS+1 ints][tmp nr x 2 ints][records nr x 4 words]
struct KgLayout {
    size_t per_cloud_words;
    size_t off_cells, off_tmp, off_rec;
};
static inline KgLayout kg_layout(int nr)
{
    KgLayout L;
    L.off_cells = KG_HDR;
    L.off_tmp = L.off_cells + (size_t)KG_CELLS + 1;
    L.off_tmp = (L.off_tmp + 3) & ~(size_t)3;
    L.off_rec = L.off_tmp + 2 * (size_t)nr;
    L.off_rec = (L.off_rec + 3) & ~(size_t)3; // 16-byte aligned records
    L.per_cloud_words = (L.off_rec + 4 * (size_t)nr + 3) & ~(size_t)3;
    return L;
}

// order-preserving float <-> uint (for atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t f2ord(float f)
{
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u)
{
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct KgGrid {
    float lo[3];
    float h, inv_h;
    int dim[3];
};

// header words: 0-2 min (ordered), 3-5 max (ordered); grid parameters are recomputed from them by everyone
// min_h > 0 (ball query): as many cells per axis as keep the cell edge >= min_h, instead of `gtarget`
__device__ __forceinline__ KgGrid kg_grid(const uint32_t *hdr, int gtarget, float min_h = 0.f)
{
    KgGrid g;
    float ext[3], mx = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        g.lo[a] = ord2f(hdr[a]);
        ext[a] = ord2f(hdr[3 + a]) - g.lo[a];
        if (!(ext[a] >= 0.f)) ext[a] = 0.f; // empty cloud / NaN
        mx = fmaxf(mx, ext[a]);
    }
    const bool ok = mx > 0.f && mx < INFINITY;
    if (min_h > 0.f) {
        const float f = ok ? mx / min_h : 1.f;
        gtarget = f >= (float)KG_GMAX ? KG_GMAX : (f >= 1.f ? (int)f : 1);
    }
    g.h = ok ? mx / (float)gtarget : INFINITY;
    g.inv_h = ok ? (float)gtarget / mx : 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int d = ok ? (int)(ext[a] * g.inv_h) + 1 : 1;
        g.dim[a] = d < 1 ? 1 : (d > gtarget ? gtarget : d);
    }
    return g;
}
__device__ __forceinline__ int kg_cell1(float p, float lo, float inv_h, int dim)
{
    float f = (p - lo) * inv_h;
    int c = (f >= 0.f) ? (int)fminf(f, (float)(dim - 1)) : 0; // NaN -> 0
    return c;
}

// Build = 5 small kernels (init, box, count, scan, scatter).  (A single-workgroup-per-cloud version with the
// histogram and the scan in LDS was measured too: one workgroup's serial passes over 24 000 points take
// ~65 us, the five launches below ~40 us including the gaps.)
__global__ __launch_bounds__(256) void kg_init_kernel(uint32_t *ws, size_t per_cloud)
{
    uint32_t *W = ws + (size_t)blockIdx.y * per_cloud;
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 3) W[i] = 0xFFFFFFFFu;       // min
    else if (i < KG_HDR) W[i] = 0u;      // max, spare
    if (i <= KG_CELLS) W[KG_HDR + i] = 0u;
}

__global__ __launch_bounds__(256) void kg_bbox_kernel(int nr, const float *__restrict__ ref, uint32_t *ws,
                                                      size_t per_cloud)
{
    const float *R = ref + (size_t)blockIdx.y * nr * 3;
    uint32_t *W = ws + (size_t)blockIdx.y * per_cloud;
    __shared__ uint32_t part[6][4];
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nr; i += gridDim.x * 256) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = R[i * 3 + a];
            if (v == v) { // NaNs do not take part in the box
                uint32_t o = f2ord(v);
                lo[a] = min(lo[a], o);
                hi[a] = max(hi[a], o);
            }
        }
    }
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        uint32_t l = wave_min_u32(lo[a]), h = wave_max_u32(hi[a]);
        if (lane_id() == 0) { part[a][wave] = l; part[3 + a][wave] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 6) { // one atomic per block and box component
        const int a = threadIdx.x;
        uint32_t v = part[a][0];
        for (int w = 1; w < 4; ++w) v = a < 3 ? min(v, part[a][w]) : max(v, part[a][w]);
        if (a < 3) atomicMin(&W[a], v);
        else atomicMax(&W[a], v);
    }
}

__device__ __forceinline__ uint32_t kg_morton15(uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t m = 0;
#pragma unroll
    for (int bit = 0; bit < 5; ++bit)
        m |= (((x >> bit) & 1u) << (3 * bit)) | (((y >> bit) & 1u) << (3 * bit + 1)) | (((z >> bit) & 1u) << (3 * bit + 2));
    return m;
}

// morton = 0: x-fastest linear cell ids (what the kNN query walks); 1: Morton ids (geot_spatial_order)
__global__ __launch_bounds__(256) void kg_count_kernel(int nr, int gtarget, int morton, const float *__restrict__ ref,
                                                       uint32_t *ws, size_t per_cloud, size_t off_tmp, float min_h)
{
    const float *R = ref + (size_t)blockIdx.y * nr * 3;
    uint32_t *W = ws + (size_t)blockIdx.y * per_cloud;
    const KgGrid g = kg_grid(W, gtarget, min_h);
    uint32_t *cnt = W + KG_HDR;
    uint32_t *tmp = W + off_tmp;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nr; i += gridDim.x * 256) {
        int cx = kg_cell1(R[i * 3], g.lo[0], g.inv_h, g.dim[0]);
        int cy = kg_cell1(R[i * 3 + 1], g.lo[1], g.inv_h, g.dim[1]);
        int cz = kg_cell1(R[i * 3 + 2], g.lo[2], g.inv_h, g.dim[2]);
        uint32_t cell = morton ? kg_morton15((uint32_t)cx, (uint32_t)cy, (uint32_t)cz)
                               : (uint32_t)((cz * g.dim[1] + cy) * g.dim[0] + cx);
        tmp[2 * i] = cell;
        tmp[2 * i + 1] = atomicAdd(&cnt[cell], 1u);
    }
}

// exclusive scan of the cell counters in place (one 1024-thread block per cloud, 32 cells per thread)
__global__ __launch_bounds__(1024) void kg_scan_kernel(uint32_t *ws, size_t per_cloud)
{
    uint32_t *cnt = ws + (size_t)blockIdx.x * per_cloud + KG_HDR;
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int PER = KG_CELLS / 1024;
    uint32_t v[PER], s = 0;
    const uint4 *c4 = reinterpret_cast<const uint4 *>(cnt + tid * PER); // KG_HDR and PER are multiples of 4 words
#pragma unroll
    for (int e = 0; e < PER; e += 4) {
        uint4 q = c4[e >> 2];
        v[e] = q.x; v[e + 1] = q.y; v[e + 2] = q.z; v[e + 3] = q.w;
        s += q.x + q.y + q.z + q.w;
    }
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
    uint4 *o4 = reinterpret_cast<uint4 *>(cnt + tid * PER);
#pragma unroll
    for (int e = 0; e < PER; e += 4) {
        uint4 q;
        q.x = run; run += v[e];
        q.y = run; run += v[e + 1];
        q.z = run; run += v[e + 2];
        q.w = run; run += v[e + 3];
        o4[e >> 2] = q;
    }
    if (tid == 1023) cnt[KG_CELLS] = run;
}

__global__ __launch_bounds__(256) void kg_scatter_kernel(int nr, const float *__restrict__ ref, uint32_t *ws,
                                                         size_t per_cloud, size_t off_tmp, size_t off_rec)
{
    const float *R = ref + (size_t)blockIdx.y * nr * 3;
    uint32_t *W = ws + (size_t)blockIdx.y * per_cloud;
    const uint32_t *start = W + KG_HDR;
    const uint32_t *tmp = W + off_tmp;
    float4 *rec = reinterpret_cast<float4 *>(W + off_rec);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nr; i += gridDim.x * 256) {
        uint32_t pos = start[tmp[2 * i]] + tmp[2 * i + 1];
        rec[pos] = make_float4(R[i * 3], R[i * 3 + 1], R[i * 3 + 2], __int_as_float(i));
    }
}

constexpr int KG_WAVES = 4;
constexpr int KG_SLOTS = 12;    // 64-record register slots of the select fast path
constexpr int KG_SEL_KMIN = 8;  // short lists are cheap to build by insertion
constexpr int KG_SEL_KMAX = 48; // beyond that the window k <= count <= 64 is too narrow to be worth probing
constexpr int KG_DPP_WAVE_SHR1 = 0x138;
__device__ __forceinline__ float kg_shr1(float v)
{
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v),
                                                                 KG_DPP_WAVE_SHR1, 0xF, 0xF, false));
}
__device__ __forceinline__ int kg_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, KG_DPP_WAVE_SHR1, 0xF, 0xF, false); }

__device__ __forceinline__ float read_lane_f(float v, int l)
{
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l));
}

struct KgBest { // the wave's best-k list: lane i = i-th smallest by (d2, index); lanes >= k stay (+inf, 0)
    float ld;
    int li;
    float tau; // entry k-1, wave-uniform
    int taui;
};

// records [s, e): 64 per step
__device__ __forceinline__ void kg_range(const float4 *__restrict__ rec, int s, int e, float qx, float qy, float qz,
                                         int k, KgBest &B)
{
    const int lane = lane_id();
    for (int c0 = s; c0 < e; c0 += 64) {
        const int r = c0 + lane;
        const bool in = r < e;
        float4 p = in ? rec[r] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float d = sqdist3(qx, qy, qz, p.x, p.y, p.z);
        const int pi = __float_as_int(p.w);
        unsigned long long mask = __ballot(in && (d < B.tau || (d == B.tau && pi < B.taui)));
        while (mask) {
            const int l = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float dc = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(d), l));
            const int ic = __builtin_amdgcn_readlane(pi, l);
            if (!(dc < B.tau || (dc == B.tau && ic < B.taui))) continue; // the threshold may have dropped
            const int pos = __popcll(__ballot(B.ld < dc || (B.ld == dc && B.li < ic)));
            const float sd = kg_shr1(B.ld);
            const int si = kg_shr1(B.li);
            B.ld = lane > pos ? sd : (lane == pos ? dc : B.ld);
            B.li = lane > pos ? si : (lane == pos ? ic : B.li);
            B.tau = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(B.ld), k - 1));
            B.taui = __builtin_amdgcn_readlane(B.li, k - 1);
        }
    }
}

__global__ __launch_bounds__(KG_WAVES * 64) void knn_grid_kernel(
    int nq, int nr, int k, int gtarget, const float *__restrict__ query, const uint32_t *__restrict__ ws,
    size_t per_cloud, size_t off_rec, int *__restrict__ idx, float *__restrict__ dist2)
{
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int j = blockIdx.x * KG_WAVES + wave;
    if (j >= nq) return;
    const uint32_t *W = ws + (size_t)bi * per_cloud;
    const KgGrid g = kg_grid(W, gtarget);
    const int *start = reinterpret_cast<const int *>(W + KG_HDR);
    const float4 *rec = reinterpret_cast<const float4 *>(W + off_rec);
    const float *Q = query + ((size_t)bi * nq + j) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2];
    const int cx = kg_cell1(qx, g.lo[0], g.inv_h, g.dim[0]);
    const int cy = kg_cell1(qy, g.lo[1], g.inv_h, g.dim[1]);
    const int cz = kg_cell1(qz, g.lo[2], g.inv_h, g.dim[2]);
    const int dx = g.dim[0], dy = g.dim[1], dz = g.dim[2];

    KgBest B;
    B.ld = INFINITY; B.li = 0; B.tau = INFINITY; B.taui = 0;
    // taui = 0 with tau = inf: "d == inf && index < 0" never holds, so infinite distances are never
    // inserted -- as in the brute-force kernel, whose test is d < tau.

    const int rmax = max(max(max(cx, dx - 1 - cx), max(cy, dy - 1 - cy)), max(cz, dz - 1 - cz));

    // ---- fast path: the whole 3 x 3 x 3 block in registers, threshold select + rank sort --------------
    // Inserting candidates one by one costs ~25 dependent instructions each and a query sees ~3 k of them
    // in arbitrary order (~k (1 + ln 3) insertions).  Instead: all distances of the block into <= KG_SLOTS
    // register slots; a few counting probes (compare + ballot + popcount per slot) find a threshold tau with
    // k <= #{d <= tau} <= 64 that is also STRICTLY inside the certified radius of the block (every unvisited
    // point is farther than tau); those <= 64 survivors are compacted through LDS, ranked by (d2, index)
    // with one readlane loop, and lanes with rank < k write the answer.  Anything unusual (rows too long,
    // no such tau: heavy ties or the k-th neighbour outside the block) falls through to the general loop.
    if (k >= KG_SEL_KMIN && k <= KG_SEL_KMAX) {
        __shared__ float kg_sd[KG_WAVES][64];
        __shared__ int kg_si[KG_WAVES][64];
        int rs = 0, re = 0;
        if (lane < 9) {
            const int y = cy + lane % 3 - 1, z = cz + lane / 3 - 1;
            if (y >= 0 && y < dy && z >= 0 && z < dz) {
                const int base = (z * dy + y) * dx;
                rs = start[base + max(cx - 1, 0)];
                re = start[base + min(cx + 1, dx - 1) + 1];
            }
        }
        int slots = (re - rs + 63) >> 6;
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) slots += __shfl_xor(slots, o, 16); // lanes 0..15 (9 rows + zeros)
        slots = __builtin_amdgcn_readfirstlane(slots);
        float b2 = 3.0e38f; // largest admissible threshold: finite, and strictly inside the certified radius
        if (rmax > 1) {
            float bound = INFINITY;
            if (cx - 1 > 0) bound = fminf(bound, qx - (g.lo[0] + (float)(cx - 1) * g.h));
            if (cx + 1 < dx - 1) bound = fminf(bound, (g.lo[0] + (float)(cx + 2) * g.h) - qx);
            if (cy - 1 > 0) bound = fminf(bound, qy - (g.lo[1] + (float)(cy - 1) * g.h));
            if (cy + 1 < dy - 1) bound = fminf(bound, (g.lo[1] + (float)(cy + 2) * g.h) - qy);
            if (cz - 1 > 0) bound = fminf(bound, qz - (g.lo[2] + (float)(cz - 1) * g.h));
            if (cz + 1 < dz - 1) bound = fminf(bound, (g.lo[2] + (float)(cz + 2) * g.h) - qz);
            bound = fmaxf(bound - g.h * 1e-3f, 0.f);
            b2 = fminf(bound * bound * 0.9999f, 3.0e38f); // NaN query -> NaN -> the probes below fail -> general loop
        }
        if (slots <= KG_SLOTS && b2 > 0.f) {
            float d[KG_SLOTS];
            int id[KG_SLOTS];
            int row = 0, a = __builtin_amdgcn_readlane(rs, 0), e = __builtin_amdgcn_readlane(re, 0);
#pragma unroll
            for (int sl = 0; sl < KG_SLOTS; ++sl) {
                while (row < 9 && a >= e) {
                    ++row;
                    if (row < 9) { a = __builtin_amdgcn_readlane(rs, row); e = __builtin_amdgcn_readlane(re, row); }
                }
                d[sl] = INFINITY;
                id[sl] = 0;
                if (row < 9) {
                    if (a + lane < e) {
                        const float4 pr = rec[a + lane];
                        d[sl] = sqdist3(qx, qy, qz, pr.x, pr.y, pr.z);
                        id[sl] = __float_as_int(pr.w);
                    }
                    a += 64;
                }
            }
            auto count_le = [&](float tau) {
                int c = 0;
#pragma unroll
                for (int sl = 0; sl < KG_SLOTS; ++sl) c += __popcll(__ballot(d[sl] <= tau));
                return c;
            };
            float tau = b2;
            int c = count_le(tau);
            bool found = c >= k && c <= 64;
            if (c > 64) { // bisect between a threshold with too few and one with too many survivors
                float lo_t = 0.f, hi_t = b2;
                for (int it = 0; it < 12 && !found; ++it) {
                    tau = 0.5f * (lo_t + hi_t);
                    c = count_le(tau);
                    if (c < k) lo_t = tau;
                    else if (c > 64) hi_t = tau;
                    else found = true;
                }
            }
            if (found) {
                int base = 0;
#pragma unroll
                for (int sl = 0; sl < KG_SLOTS; ++sl) {
                    const bool sel = d[sl] <= tau;
                    const unsigned long long m = __ballot(sel);
                    const int pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
                    if (sel) { kg_sd[wave][pos] = d[sl]; kg_si[wave][pos] = id[sl]; }
                    base += __popcll(m);
                }
                const float cd = lane < c ? kg_sd[wave][lane] : INFINITY;
                const int ci = lane < c ? kg_si[wave][lane] : 0x7fffffff;
                int rank = 0;
                for (int jx = 0; jx < c; ++jx) {
                    const float dj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cd), jx));
                    const int ij = __builtin_amdgcn_readlane(ci, jx);
                    rank += (dj < cd || (dj == cd && ij < ci)) ? 1 : 0;
                }
                if (lane < c && rank < k) {
                    const size_t o = ((size_t)bi * nq + j) * k + rank;
                    idx[o] = ci;
                    dist2[o] = cd;
                }
                return;
            }
        }
    }

    for (int r = 1;; ++r) {
        // rows (y, z) of the ring: r == 1 takes the whole 3x3x3 block (rings 0 and 1); r >= 2 only the shell
        const int side = 2 * r + 1, nrows = side * side;
        for (int row0 = 0; row0 < nrows; row0 += 64) {
            // lane = row.  For the first (3 x 3) block the rows are taken nearest-first (centre, the four
            // edge neighbours, the four corners) so that the k-th distance tightens before the far rows are
            // looked at; every row also carries a lower bound of the squared distance from the query to its
            // cells, and is skipped when that already exceeds the current k-th distance.
            int row = row0 + lane;
            if (r == 1) row = (int)((0xF862075314ull >> (4 * min(lane, 9))) & 15ull); // lanes >= 9 -> row 15: skipped below
            int s0 = 0, e0 = 0, s1 = 0, e1 = 0;
            float rlb = 0.f;
            if (row < nrows) {
                const int oy = row % side - r, oz = row / side - r;
                const int y = cy + oy, z = cz + oz;
                if (y >= 0 && y < dy && z >= 0 && z < dz) {
                    const int base = (z * dy + y) * dx;
                    const bool frame = r == 1 || abs(oy) == r || abs(oz) == r;
                    if (frame) {
                        const int x0 = max(cx - r, 0), x1 = min(cx + r, dx - 1);
                        s0 = start[base + x0];
                        e0 = start[base + x1 + 1];
                    } else {
                        if (cx - r >= 0) { s0 = start[base + cx - r]; e0 = start[base + cx - r + 1]; }
                        if (cx + r < dx) { s1 = start[base + cx + r]; e1 = start[base + cx + r + 1]; }
                    }
                    // distance from the query to the row's slab in y and z (x is not used: the row spans it)
                    const float y0 = g.lo[1] + (float)y * g.h, z0 = g.lo[2] + (float)z * g.h;
                    const float ey = fmaxf(fmaxf(y0 - qy, qy - (y0 + g.h)), 0.f);
                    const float ez = fmaxf(fmaxf(z0 - qz, qz - (z0 + g.h)), 0.f);
                    const float el = fmaxf(sqrtf(ey * ey + ez * ez) * 0.99999f - g.h * 1e-3f, 0.f);
                    rlb = el * el * 0.99999f;
                    if (!(rlb >= 0.f)) rlb = 0.f; // NaN query: never skip
                }
            }
            unsigned long long live = __ballot(e0 > s0 || e1 > s1);
            while (live) {
                const int l = __builtin_ctzll(live);
                live &= live - 1;
                if (read_lane_f(rlb, l) > B.tau) continue; // strictly farther than the k-th: cannot enter, not even as a tie
                const int a0 = __builtin_amdgcn_readlane(s0, l), b0 = __builtin_amdgcn_readlane(e0, l);
                kg_range(rec, a0, b0, qx, qy, qz, k, B);
                const int a1 = __builtin_amdgcn_readlane(s1, l), b1 = __builtin_amdgcn_readlane(e1, l);
                kg_range(rec, a1, b1, qx, qy, qz, k, B);
            }
        }
        if (r >= rmax) break; // the block covers the grid
        // nearest block face that still has cells behind it
        float bound = INFINITY;
        if (cx - r > 0) bound = fminf(bound, qx - (g.lo[0] + (float)(cx - r) * g.h));
        if (cx + r < dx - 1) bound = fminf(bound, (g.lo[0] + (float)(cx + r + 1) * g.h) - qx);
        if (cy - r > 0) bound = fminf(bound, qy - (g.lo[1] + (float)(cy - r) * g.h));
        if (cy + r < dy - 1) bound = fminf(bound, (g.lo[1] + (float)(cy + r + 1) * g.h) - qy);
        if (cz - r > 0) bound = fminf(bound, qz - (g.lo[2] + (float)(cz - r) * g.h));
        if (cz + r < dz - 1) bound = fminf(bound, (g.lo[2] + (float)(cz + r + 1) * g.h) - qz);
        bound = fmaxf(bound - g.h * 1e-3f, 0.f);
        if (B.tau < bound * bound * 0.99999f) break; // NaN bound (NaN query) never breaks early: full scan
    }
    if (lane < k) {
        size_t o = ((size_t)bi * nq + j) * k + lane;
        idx[o] = B.li;
        dist2[o] = B.ld;
    }
}

// The records of a cell sit in arrival order (kg_count_kernel ranks them with an atomic), which changes from launch to
// launch; a processing ORDER has to be the same every time -- per-tile BatchNorm sums are taken in that order, and their
// rounding reaches every weight -- so inside a cell the points go by ascending index: rank = members with a smaller index
// (cells hold a handful of points; a cell of more than RIX_SORT_MAX -- thousands of duplicates -- keeps its arrival order).
__global__ __launch_bounds__(256) void kg_order_kernel(int nr, const uint32_t *__restrict__ ws, size_t per_cloud,
                                                       size_t off_tmp, size_t off_rec, int *__restrict__ order)
{
    const uint32_t *W = ws + (size_t)blockIdx.y * per_cloud;
    const uint32_t *start = W + KG_HDR, *tmp = W + off_tmp;
    const float4 *rec = reinterpret_cast<const float4 *>(W + off_rec);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nr; i += gridDim.x * 256) {
        const int me = __float_as_int(rec[i].w);
        const uint32_t cell = tmp[2 * me];
        const int a = (int)start[cell], z = (int)start[cell + 1];
        int pos = i;
        if (z - a <= RIX_SORT_MAX) {
            pos = a;
            for (int j = a; j < z; ++j) pos += __float_as_int(rec[j].w) < me ? 1 : 0;
        }
        order[(size_t)blockIdx.y * nr + pos] = blockIdx.y * nr + me;
    }
}

// ---- ball query over the grid -------------------------------------------------------------------
// Same output as ball_query_kernel (neighbors.hip; pointnet2/_ext_src/src/ball_query_gpu.cu:12-47): the
// first nsample points, in index order, with d2 < r^2, the tail filled with the first hit (0 if none).
// "First nsample in index order" = the nsample SMALLEST indices among all hits, so the scan over half the
// cloud becomes: cells of edge >= 1.0001 r, the 27 around the query hold every hit; their records go into
// register slots; if there are more than 64 hits an integer bisection on the index finds a cut with
// nsample <= #{hits with index <= cut} <= 64; the survivors are compacted through LDS and ranked by index.
// Blocks too dense for the register slots are scanned row by row with an insertion list keyed by index.
__global__ __launch_bounds__(KG_WAVES * 64) void ball_grid_kernel(
    int nq, int nr, int nsample, float radius, float min_h, const float *__restrict__ query,
    const uint32_t *__restrict__ ws, size_t per_cloud, size_t off_rec, int *__restrict__ idx)
{
    __shared__ int bg_si[KG_WAVES][64];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int j = blockIdx.x * KG_WAVES + wave;
    if (j >= nq) return;
    const uint32_t *W = ws + (size_t)bi * per_cloud;
    const KgGrid g = kg_grid(W, 1, min_h);
    const int *start = reinterpret_cast<const int *>(W + KG_HDR);
    const float4 *rec = reinterpret_cast<const float4 *>(W + off_rec);
    const float *Q = query + ((size_t)bi * nq + j) * 3;
    const float qx = Q[0], qy = Q[1], qz = Q[2], r2 = radius * radius;
    const int cx = kg_cell1(qx, g.lo[0], g.inv_h, g.dim[0]);
    const int cy = kg_cell1(qy, g.lo[1], g.inv_h, g.dim[1]);
    const int cz = kg_cell1(qz, g.lo[2], g.inv_h, g.dim[2]);
    const int dx = g.dim[0], dy = g.dim[1], dz = g.dim[2];
    int *out = idx + ((size_t)bi * nq + j) * nsample;

    int rs = 0, re = 0;
    if (lane < 9) {
        const int y = cy + lane % 3 - 1, z = cz + lane / 3 - 1;
        if (y >= 0 && y < dy && z >= 0 && z < dz) {
            const int base = (z * dy + y) * dx;
            rs = start[base + max(cx - 1, 0)];
            re = start[base + min(cx + 1, dx - 1) + 1];
        }
    }
    int slots = (re - rs + 63) >> 6;
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) slots += __shfl_xor(slots, o, 16);
    slots = __builtin_amdgcn_readfirstlane(slots);

    if (slots <= KG_SLOTS) {
        int id[KG_SLOTS]; // index of a hit, INT_MAX otherwise
        int row = 0, a = __builtin_amdgcn_readlane(rs, 0), e = __builtin_amdgcn_readlane(re, 0), H = 0;
#pragma unroll
        for (int sl = 0; sl < KG_SLOTS; ++sl) {
            while (row < 9 && a >= e) {
                ++row;
                if (row < 9) { a = __builtin_amdgcn_readlane(rs, row); e = __builtin_amdgcn_readlane(re, row); }
            }
            id[sl] = 0x7fffffff;
            if (row < 9) {
                if (a + lane < e) {
                    const float4 pr = rec[a + lane];
                    if (sqdist3(qx, qy, qz, pr.x, pr.y, pr.z) < r2) id[sl] = __float_as_int(pr.w);
                }
                a += 64;
            }
            H += __popcll(__ballot(id[sl] != 0x7fffffff));
        }
        int cut = 0x7ffffffe, c = H; // keep hits with index <= cut
        bool ok = H <= 64;
        if (!ok) {
            int lo_i = -1, hi_i = nr - 1; // count(lo_i) < nsample <= ... ; count(hi_i) = H > 64
            for (int it = 0; it < 32 && !ok; ++it) {
                cut = lo_i + ((hi_i - lo_i) >> 1);
                c = 0;
#pragma unroll
                for (int sl = 0; sl < KG_SLOTS; ++sl) c += __popcll(__ballot(id[sl] <= cut));
                if (c < nsample) lo_i = cut;
                else if (c > 64) hi_i = cut;
                else ok = true;
                if (hi_i - lo_i <= 1) break;
            }
        }
        if (ok) {
            int base = 0;
#pragma unroll
            for (int sl = 0; sl < KG_SLOTS; ++sl) {
                const bool sel = id[sl] <= cut;
                const unsigned long long mk = __ballot(sel);
                const int pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0));
                if (sel) bg_si[wave][pos] = id[sl];
                base += __popcll(mk);
            }
            const int ci = lane < c ? bg_si[wave][lane] : 0x7fffffff;
            int rank = 0;
            for (int jx = 0; jx < c; ++jx) rank += __builtin_amdgcn_readlane(ci, jx) < ci ? 1 : 0;
            // smallest index of all = the reference's "first hit" used as filler
            int first = ci;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) first = min(first, __shfl_xor(first, o));
            if (c == 0) first = 0;
            if (lane < c && rank < nsample) out[rank] = ci;
            const int have = min(c, nsample);
            for (int l = have + lane; l < nsample; l += 64) out[l] = first;
            return;
        }
    }
    // dense block (or a pathological index distribution): rows one by one, list of the nsample smallest hit
    // indices (lane i = i-th smallest) built by insertion
    int li = 0x7fffffff, taui = 0x7fffffff;
    for (int row = 0; row < 9; ++row) {
        const int a0 = __builtin_amdgcn_readlane(rs, row), e0 = __builtin_amdgcn_readlane(re, row);
        for (int c0 = a0; c0 < e0; c0 += 64) {
            int cand = 0x7fffffff;
            if (c0 + lane < e0) {
                const float4 pr = rec[c0 + lane];
                if (sqdist3(qx, qy, qz, pr.x, pr.y, pr.z) < r2) cand = __float_as_int(pr.w);
            }
            unsigned long long mk = __ballot(cand < taui);
            while (mk) {
                const int l = __builtin_ctzll(mk);
                mk &= mk - 1;
                const int ic = __builtin_amdgcn_readlane(cand, l);
                if (!(ic < taui)) continue;
                const int pos = __popcll(__ballot(li < ic));
                const int sh = kg_shr1(li);
                li = lane > pos ? sh : (lane == pos ? ic : li);
                taui = __builtin_amdgcn_readlane(li, nsample - 1);
            }
        }
    }
    const int cnt = __popcll(__ballot(li != 0x7fffffff && lane < nsample));
    const int first = cnt ? __builtin_amdgcn_readlane(li, 0) : 0;
    if (lane < nsample) out[lane] = lane < cnt ? li : first;
}

static int kg_target(int nr, int k)
{
    double g = std::sqrt(3.0 * (double)nr / (5.0 * (double)(k < 1 ? 1 : k)));
    int G = (int)g;
    return G < 1 ? 1 : (G > KG_GMAX ? KG_GMAX : G);
}

} // namespace geot

using namespace geot;

GEOT_EXPORT long long geot_knn_grid_ws_bytes(int b, int nr)
{
    if (b < 0 || nr < 0) return -1;
    return (long long)(kg_layout(nr).per_cloud_words * 4) * (long long)(b < 1 ? 1 : b);
}

// 1 if geot_knn_sorted_ws / geot_three_nn_ws would take the grid path for these sizes.  The grid costs
// ~45 us of build kernels per call (independent of b: clouds build in parallel) and then beats the
// brute-force scan by 2-3x per pair-heavy query; measured break-even on MI355X: ~6.7e7 pairs per call, or
// long best-k lists (k >= 16) where the brute-force kernel is insertion-bound and short of waves.
GEOT_EXPORT int geot_knn_grid_eligible(int b, int nq, int nr, int k)
{
    const char *e = getenv("GEOT_NN_IMPL"); // "basic" / "wave" force the brute-force kernels, "grid" the grid
    if (e && (e[0] == 'b' || e[0] == 'w')) return 0;
    if (k < 1 || k > 64 || nr < 2048 || b < 1 || nq < 1) return 0;
    if (e && e[0] == 'g') return 1;
    const long long pairs = (long long)b * nq * nr;
    return (pairs >= (1ll << 26) || (k >= 16 && nr >= 8192 && pairs >= (1ll << 22))) ? 1 : 0;
}

GEOT_EXPORT int geot_knn_sorted_ws(int b, int nq, int nr, int k, const float *query, const float *ref, int *idx,
                                   float *dist2, void *workspace, long long ws_bytes, void *stream)
{
    if (b < 0 || nq < 0 || nr < 0 || k < 0 || k > 256) return hipErrorInvalidValue;
    if (b == 0 || nq == 0 || k == 0) return hipSuccess;
    if (b > 65535) return hipErrorInvalidValue;
    if (!workspace || !geot_knn_grid_eligible(b, nq, nr, k) || ws_bytes < geot_knn_grid_ws_bytes(b, nr))
        return geot_knn_sorted(b, nq, nr, k, query, ref, idx, dist2, stream);
    if (((uintptr_t)workspace & 15) != 0) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const KgLayout L = kg_layout(nr);
    uint32_t *ws = (uint32_t *)workspace;
    const int G = kg_target(nr, k);
    const int pb = (nr + 255) / 256 < 96 ? (nr + 255) / 256 : 96;
    hipLaunchKernelGGL(kg_init_kernel, dim3((KG_CELLS + 1 + 255) / 256, b), dim3(256), 0, s, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_bbox_kernel, dim3(pb, b), dim3(256), 0, s, nr, ref, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_count_kernel, dim3(pb, b), dim3(256), 0, s, nr, G, 0, ref, ws, L.per_cloud_words, L.off_tmp, 0.f);
    hipLaunchKernelGGL(kg_scan_kernel, dim3(b), dim3(1024), 0, s, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_scatter_kernel, dim3(pb, b), dim3(256), 0, s, nr, ref, ws, L.per_cloud_words, L.off_tmp,
                       L.off_rec);
    hipLaunchKernelGGL(knn_grid_kernel, dim3((nq + KG_WAVES - 1) / KG_WAVES, b), dim3(KG_WAVES * 64), 0, s, nq, nr,
                       k, G, query, ws, L.per_cloud_words, L.off_rec, idx, dist2);
    return hipGetLastError();
}

GEOT_EXPORT int geot_three_nn_ws(int b, int n, int m, const float *unknown, const float *known, float *dist2,
                                 int *idx, void *workspace, long long ws_bytes, void *stream)
{
    if (b < 0 || n < 0 || m < 0) return hipErrorInvalidValue;
    if (b == 0 || n == 0) return hipSuccess;
    if (!workspace || !geot_knn_grid_eligible(b, n, m, 3) || ws_bytes < geot_knn_grid_ws_bytes(b, m))
        return geot_three_nn(b, n, m, unknown, known, dist2, idx, stream);
    return geot_knn_sorted_ws(b, n, m, 3, unknown, known, idx, dist2, workspace, ws_bytes, stream);
}

// order (b*n) int32 <- the global point ids b*n + i sorted by (cloud, Morton cell of a 32^3 grid over the
// cloud's bounding box): a processing order in which consecutive points are spatial neighbours, so that
// kernels which gather per-point rows of a kNN graph find them in L2.  Ascending index within a cell: the same order on
// every call (cells of more than RIX_SORT_MAX points excepted).
GEOT_EXPORT int geot_spatial_order(int b, int n, const float *xyz, int *order, void *workspace, long long ws_bytes,
                                   void *stream)
{
    if (b < 0 || n < 0 || !order) return hipErrorInvalidValue;
    if (b == 0 || n == 0) return hipSuccess;
    if (b > 65535 || !workspace || ws_bytes < geot_knn_grid_ws_bytes(b, n) || ((uintptr_t)workspace & 15) != 0)
        return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const KgLayout L = kg_layout(n);
    uint32_t *ws = (uint32_t *)workspace;
    const int pb = (n + 255) / 256 < 96 ? (n + 255) / 256 : 96;
    hipLaunchKernelGGL(kg_init_kernel, dim3((KG_CELLS + 1 + 255) / 256, b), dim3(256), 0, s, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_bbox_kernel, dim3(pb, b), dim3(256), 0, s, n, xyz, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_count_kernel, dim3(pb, b), dim3(256), 0, s, n, KG_GMAX, 1, xyz, ws, L.per_cloud_words,
                       L.off_tmp, 0.f);
    hipLaunchKernelGGL(kg_scan_kernel, dim3(b), dim3(1024), 0, s, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_scatter_kernel, dim3(pb, b), dim3(256), 0, s, n, xyz, ws, L.per_cloud_words, L.off_tmp,
                       L.off_rec);
    hipLaunchKernelGGL(kg_order_kernel, dim3(pb, b), dim3(256), 0, s, n, ws, L.per_cloud_words, L.off_tmp, L.off_rec, order);
    return hipGetLastError();
}

// 1 if geot_ball_query_ws would take the grid path
GEOT_EXPORT int geot_ball_grid_eligible(int b, int n, int m, float radius, int nsample)
{
    const char *e = getenv("GEOT_NN_IMPL");
    if (e && (e[0] == 'b' || e[0] == 'w')) return 0;
    if (!(radius > 0.f) || nsample < 1 || nsample > 64 || n < 2048 || b < 1 || m < 1) return 0;
    if (e && e[0] == 'g') return 1;
    return (long long)b * m * n >= (1ll << 26) ? 1 : 0;
}

// geot_ball_query through the grid (identical output); workspace: geot_knn_grid_ws_bytes(b, n) bytes.
GEOT_EXPORT int geot_ball_query_ws(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                   const float *xyz, int *idx, void *workspace, long long ws_bytes, void *stream)
{
    if (b < 0 || n < 0 || m < 0 || nsample < 0) return hipErrorInvalidValue;
    if (b == 0 || m == 0 || nsample == 0) return hipSuccess;
    if (!workspace || !geot_ball_grid_eligible(b, n, m, radius, nsample) || ws_bytes < geot_knn_grid_ws_bytes(b, n) ||
        b > 65535 || ((uintptr_t)workspace & 15) != 0)
        return geot_ball_query(b, n, m, radius, nsample, new_xyz, xyz, idx, stream);
    hipStream_t s = (hipStream_t)stream;
    const KgLayout L = kg_layout(n);
    uint32_t *ws = (uint32_t *)workspace;
    const float min_h = radius * 1.0001f;
    const int pb = (n + 255) / 256 < 96 ? (n + 255) / 256 : 96;
    hipLaunchKernelGGL(kg_init_kernel, dim3((KG_CELLS + 1 + 255) / 256, b), dim3(256), 0, s, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_bbox_kernel, dim3(pb, b), dim3(256), 0, s, n, xyz, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_count_kernel, dim3(pb, b), dim3(256), 0, s, n, 1, 0, xyz, ws, L.per_cloud_words, L.off_tmp,
                       min_h);
    hipLaunchKernelGGL(kg_scan_kernel, dim3(b), dim3(1024), 0, s, ws, L.per_cloud_words);
    hipLaunchKernelGGL(kg_scatter_kernel, dim3(pb, b), dim3(256), 0, s, n, xyz, ws, L.per_cloud_words, L.off_tmp,
                       L.off_rec);
    hipLaunchKernelGGL(ball_grid_kernel, dim3((m + KG_WAVES - 1) / KG_WAVES, b), dim3(KG_WAVES * 64), 0, s, m, n, nsample,
                       radius, min_h, new_xyz, ws, L.per_cloud_words, L.off_rec, idx);
    return hipGetLastError();
}
