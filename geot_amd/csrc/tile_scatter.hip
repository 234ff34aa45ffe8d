// tile_scatter.hip -- channels-first scatter-add gradients for gfx950 (MI355X): the gradients of group_points,
// gather_points, three_interpolate and the kNN graph feature behind the reference's (B, C, N) layout.
//
// Replaces (behaviour, not code): pointnet2/_ext_src/src/interpolate_gpu.cu:119-146 (three_interpolate_grad),
// group_points_gpu.cu:46-67 (group_points_grad), sampling_gpu.cu:35-50 (gather_points_grad) and their pointnet2_batch
// twins -- one float atomicAdd per (pair, channel) there.
//
//   grad_table[b, ch, j] (+)= sum over the pairs (e, t) with idx[b, e, t] == j of w[b, e, t] * grad_out[b, ch, e]
//
// Shape of the problem: per channel a sparse (m x L) matrix with nt entries per column times a vector that lies
// contiguous along L; the sources are in the caller's (spatially random) order, so a target's sources are spread over
// the whole row.  One of the two sides has to be resident while the other streams.  Here the TARGETS are resident: a
// workgroup owns (batch, CH channels), keeps their m x CH sums in LDS (m = 8192, CH = 4: 128 KB) and streams the CH
// rows of grad_out through LDS tiles of `tl` sources in memory order -- every byte of grad_out is read once, by
// 16-byte coalesced loads issued four tiles ahead of their use (register-staged: 64 KB in flight per CU).
//
// What a wave does with a tile is decided once per call by ts_build_kernel (one workgroup per (batch, tile), a counting
// sort in LDS): the tile's pairs are dealt to the 16 waves by TARGET RANGE (equal pair counts, whole targets).  A wave's
// share is cut in two.  PLAIN chunks: the first pair of its first 64 P targets -- 64 lanes, 64 distinct targets: one
// coalesced 8-byte entry load (prefetched a tile ahead), one 16-byte LDS read of the source's channels, a multiply and
// one LDS read-add-write; no cross-lane step, ~10 vector instructions per 64 pairs.  The SORTED RUN: every other pair
// (the first pairs of the last < 64 targets and all second, third, ... pairs -- about a quarter of the pairs at the
// model's shapes, one chunk per wave and tile), ordered by (target, pair id): the lanes of a target are neighbours and
// a wave-shift DPP segmented sum leaves ONE read-add-write per target to the run's last lane.  The plain chunks come
// first (LDS is in-order per wave, so a target's later pairs add after its first), no two waves share a target inside
// a tile, tiles are separated by a barrier: no atomics of any kind, no padding beyond the run's last chunk, and ONE
// summation order per target (tile after tile; inside a tile ascending pair id): bit-reproducible.  Hub targets need
// nothing special: their pairs are one long run that the owning wave reduces 64 at a time.
//
// Against the per-target list walk this replaces (gather_group.hip table_gather_csr_parts_kernel, 1.4-1.7 TB/s): the index
// is read coalesced instead of through per-lane offsets, no lane idles on a neighbour's longer list, staging overlaps the
// walk.  Measurements: profiles/r05_tile_scatter.txt.
#include "geot_common.h"
#include "tile_scatter.h"

// phase stamps of the lab harness (tools/lab/ts_lab.hip compiles this file with them defined); nothing in the product
#ifndef TS_STAMP
#define TS_STAMP_DECL
#define TS_STAMP(slot)
#define TS_STAMP_FLUSH
#endif

namespace geot {

constexpr int TS_THREADS = 1024;
constexpr int TS_WAVES = TS_THREADS / 64;
constexpr int TS_EBITS = 12;              // entry key = target << 12 | source within its tile
constexpr int TS_TILE_FLOATS = 4096;      // floats of one staged tile at most (tl * CH): 16 KB
constexpr int TS_MAX_M = 32768;           // targets whose sums one CU's LDS can hold at CH = 1
constexpr int TS_MAX_Q = 1024;            // tiles per row (their 16-entry wave tables live in LDS)
constexpr int TS_LDS = 160 * 1024;
constexpr int TS_NPF = 4;                 // chunks of the next tile's entries a wave holds in registers
constexpr int TS_PMAX = 31;               // plain chunks per wave and tile at most (5-bit field; the rest joins the sorted run)
constexpr uint32_t TS_NONE = 0xffffffffu; // padding entry of a sorted run's last chunk

struct TsPlan {
    int ch, tl, q, ppp, cap;              // channels per workgroup, sources per tile, tiles, pairs per tile, entry slots per tile
    size_t lds, lds_build;
    long long ent_words, ints;
};

static inline long long ts_acc_floats(int m, int ch) { return ((long long)m * ch + 3) & ~3LL; }

static bool ts_plan(int b, int c, int m, long long L, int nt, bool weighted, TsPlan &p)
{
    const char *env = getenv("GEOT_GATHER_IMPL");   // plain | atomic | csr | sell: the older forms (A/B runs); unset | tiles: this one
    if (env && env[0] && env[0] != 't') return false;
    if (b < 1 || c < 1 || m < 1 || L < 1 || nt < 1 || m > TS_MAX_M || b > 65535 || L > 0x7fffffffLL / nt) return false;
    for (int ch = 4; ch >= 1; ch >>= 1) {
        if (ch > c) continue;
        const long long accb = ts_acc_floats(m, ch) * 4;
        long long tl = TS_TILE_FLOATS / ch;
        // the sort holds m + 1 counters and the tile's pair ids in LDS; its packed prefix sums are 16-bit
        long long sort_pairs = (TS_LDS - 2048 - ((long long)m + 1) * 4) / 4;
        if (sort_pairs > 0xfff0) sort_pairs = 0xfff0;
        if (tl * nt > sort_pairs) tl = (sort_pairs / nt) & ~3LL;
        if (tl > L) tl = (L + 3) & ~3LL;
        long long q = 0;
        bool fits = false;
        for (int it = 0; it < 6 && tl >= 4; ++it) {
            q = (L + tl - 1) / tl;
            const long long lds = accb + 2 * tl * ch * 4 + q * TS_WAVES * 4;
            if (lds <= TS_LDS) { fits = true; break; }
            tl = ((TS_LDS - accb - q * TS_WAVES * 4) / (2 * ch * 4)) & ~3LL;
        }
        if (!fits || tl < 4) continue;
        if (tl < 256 && tl < ((L + 3) & ~3LL)) continue;   // tiles this short are all barrier: fewer channels per workgroup
        if (q > TS_MAX_Q) continue;
        tl = ((L + q - 1) / q + 3) & ~3LL;                   // equal tiles
        // ... whose 16 shares are a few pairs short of whole chunks: a share of 64 j + (0 .. 3) pairs plus the spill of its last
        // target is j + 1 chunks with one or two entries in the last
        {
            long long t2 = tl;
            for (long long share = t2 * nt / TS_WAVES; t2 > 64 && share >= 64 && ((share & 63) >= 60 || (share & 63) < 2);
                 share = t2 * nt / TS_WAVES)
                t2 -= 4;
            if (((L + t2 - 1) / t2) * 20 <= q * 21) tl = t2;    // ... unless that costs more than 5 % more tiles (short rows)
        }
        q = (L + tl - 1) / tl;
        if (q > TS_MAX_Q) continue;
        const long long ppp = tl * nt;
        const long long cap = ((ppp + 64LL * TS_WAVES + 63) & ~63LL) + 64 * TS_NPF;
        if (cap > 0xffff) continue;                          // a wave's first slot is a 16-bit field of its table entry
        const long long ent = (long long)b * q * cap * (weighted ? 2 : 1);
        if (ent > 0x7ffffff0LL) continue;
        p.ch = ch;
        p.tl = (int)tl;
        p.q = (int)q;
        p.ppp = (int)ppp;
        p.cap = (int)cap;
        p.lds = (size_t)(accb + 2 * tl * ch * 4 + q * TS_WAVES * 4);
        p.lds_build = (size_t)(((long long)m + 1) * 4 + ppp * 4);
        p.ent_words = (ent + 1) & ~1LL;
        p.ints = p.ent_words + (long long)b * q * TS_WAVES + 8;
        return true;
    }
    return false;
}

long long ts_ws_ints(int b, int c, int m, long long L, int nt, bool weighted)
{
    TsPlan p;
    return ts_plan(b, c, m, L, nt, weighted, p) ? p.ints : 0;
}

// ---- the sort: one workgroup per (batch, tile) ------------------------------------------------------------------------------
// Output per tile: `cap` entry slots (key = target << 12 | source within the tile; weight) and one word per walking
// wave: first slot | plain chunks << 16 | chunks of the sorted run << 21.  A wave's slots: 64 P plain entries (the first
// pair of its first 64 P targets, ascending target), then the sorted run -- its other pairs by (target, pair id) --
// padded to whole chunks with TS_NONE.
// One LDS word per target carries two prefix sums at once: pairs before it (low half) and targets-with-pairs before it
// (high half); both stay below 65536 (plan).  Every slot follows from those and the pair's rank inside its target, so
// the layout does not depend on the order in which the LDS atomics of the counting sort arrive.
template <bool WEIGHTED>
__global__ __launch_bounds__(TS_THREADS) void ts_build_kernel(int m, int L, int nt, int tl, int q, int ppp, int cap,
                                                              const int *__restrict__ idx, const float *__restrict__ weight,
                                                              uint32_t *__restrict__ ent, int *__restrict__ wr)
{
    extern __shared__ int ts_i[];
    int *cnt = ts_i;                                          // [m + 1] (present << 16 | pairs): counts -> starts -> ends
    int *ids = cnt + m + 1;                                   // [ppp] pair ids grouped by target
    __shared__ int wsum[TS_WAVES];
    __shared__ int wfs[TS_WAVES + 1], wfp[TS_WAVES + 1], wfirst[TS_WAVES + 1], wplain[TS_WAVES];
    const int part = blockIdx.x, bi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p0 = part * tl, plen = min(tl, L - p0), np = plen * nt;
    const int *pidx = idx + ((size_t)bi * L + p0) * nt;
    for (int i = tid; i <= m; i += TS_THREADS) cnt[i] = 0;
    __syncthreads();
    for (int x = tid; x < np; x += TS_THREADS) {
        const int k = pidx[x];
        if ((unsigned)k < (unsigned)m) atomicAdd(&cnt[k], 1);
    }
    __syncthreads();
    {   // exclusive scan of (count > 0) << 16 | count over the targets
        const int per = (m + TS_THREADS - 1) / TS_THREADS, a0 = tid * per;
        int s = 0;
        for (int i = 0; i < per; ++i) {
            const int v = a0 + i < m ? cnt[a0 + i] : 0;
            s += v + (v ? 0x10000 : 0);
        }
        int inc = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int run = inc - s;
        for (int w = 0; w < wave; ++w) run += wsum[w];
        for (int i = 0; i < per; ++i)
            if (a0 + i < m) {
                const int v = cnt[a0 + i];
                cnt[a0 + i] = run;
                run += v + (v ? 0x10000 : 0);
            }
        if (tid == TS_THREADS - 1) cnt[m] = run;           // totals
    }
    __syncthreads();
    const int npv = cnt[m] & 0xffff;                       // pairs with a valid target
    const int per = max(1, (npv + TS_WAVES - 1) / TS_WAVES);
    if (tid <= TS_WAVES) {                                 // first target of walking wave w: the first whose pairs start at or behind w * per
        int lo = 0, hi = m;                                // (cnt[k] & 0xffff = pairs before k, non-decreasing; cnt[m] = all)
        const int want = min(npv, tid * per);
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((cnt[mid] & 0xffff) >= want) hi = mid;
            else lo = mid + 1;
        }
        if (tid == TS_WAVES) lo = m;
        wfs[tid] = cnt[lo] & 0xffff;
        wfp[tid] = (int)((unsigned)cnt[lo] >> 16);
    }
    __syncthreads();
    if (tid == 0) {
        int at = 0;
        for (int w = 0; w < TS_WAVES; ++w) {
            const int targets = wfp[w + 1] - wfp[w], pairs = wfs[w + 1] - wfs[w];
            const int P = min(TS_PMAX, targets >> 6), run_chunks = (pairs - 64 * P + 63) >> 6;
            wfirst[w] = at;
            wplain[w] = P;
            wr[((size_t)bi * q + part) * TS_WAVES + w] = at | (P << 16) | (run_chunks << 21);
            at += 64 * (P + run_chunks);
        }
        wfirst[TS_WAVES] = at;
    }
    __syncthreads();
    for (int x = tid; x < np; x += TS_THREADS) {          // place: afterwards the low half of cnt[k] = END of target k's group
        const int k = pidx[x];
        if ((unsigned)k < (unsigned)m) ids[atomicAdd(&cnt[k], 1) & 0xffff] = x;
    }
    __syncthreads();
    const size_t ebase = ((size_t)bi * q + part) * cap;
    auto put = [&](int slot, uint32_t key, float w) {
        if (WEIGHTED) reinterpret_cast<uint2 *>(ent)[ebase + slot] = make_uint2(key, __float_as_uint(w));
        else ent[ebase + slot] = key;
    };
    {   // padding of every wave's last run chunk
        const int w = wave, pairs = wfs[w + 1] - wfs[w], used = wfirst[w] + pairs, end = wfirst[w + 1];
        if (used + lane < end) put(used + lane, TS_NONE, 0.f);
    }
    for (int x = tid; x < np; x += TS_THREADS) {
        const int k = pidx[x];
        if ((unsigned)k >= (unsigned)m) continue;
        const int a = k ? cnt[k - 1] & 0xffff : 0, z = cnt[k] & 0xffff;
        int layer = 0;                                     // rank by pair id among the target's pairs
        for (int i = a; i < z; ++i) layer += ids[i] < x ? 1 : 0;
        const int w = min(TS_WAVES - 1, a / per);
        const int r = (int)((unsigned)cnt[k] >> 16) - wfp[w];              // rank of the target among the wave's targets with pairs
        const int plain = 64 * wplain[w];
        int slot;
        if (layer == 0 && r < plain) slot = wfirst[w] + r;
        else slot = wfirst[w] + plain + (a - wfs[w]) - min(r, plain) + layer - (r < plain ? 1 : 0);
        put(slot, ((uint32_t)k << TS_EBITS) | (uint32_t)(x / nt), WEIGHTED ? weight[((size_t)bi * L + p0) * nt + x] : 1.f);
    }
}

// ---- small vector helpers ------------------------------------------------------------------------------------------
typedef float ts_f4 __attribute__((ext_vector_type(4)));
typedef float ts_f2 __attribute__((ext_vector_type(2)));
template <int CH> struct TsVec;
template <> struct TsVec<4> { typedef ts_f4 type; };
template <> struct TsVec<2> { typedef ts_f2 type; };
template <> struct TsVec<1> { typedef float type; };

constexpr int TS_DPP_WAVE_SHR1 = 0x138;
__device__ __forceinline__ float ts_shr1(float v)
{   // lane l receives lane l - 1; lane 0 receives 0
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), TS_DPP_WAVE_SHR1, 0xF, 0xF, false));
}
__device__ __forceinline__ ts_f2 ts_shr1(ts_f2 v) { return ts_f2{ts_shr1(v.x), ts_shr1(v.y)}; }
__device__ __forceinline__ ts_f4 ts_shr1(ts_f4 v) { return ts_f4{ts_shr1(v.x), ts_shr1(v.y), ts_shr1(v.z), ts_shr1(v.w)}; }
__device__ __forceinline__ float ts_up(float v, int d) { return __shfl_up(v, d); }
__device__ __forceinline__ ts_f2 ts_up(ts_f2 v, int d) { return ts_f2{__shfl_up(v.x, d), __shfl_up(v.y, d)}; }
__device__ __forceinline__ ts_f4 ts_up(ts_f4 v, int d)
{
    return ts_f4{__shfl_up(v.x, d), __shfl_up(v.y, d), __shfl_up(v.z, d), __shfl_up(v.w, d)};
}
__device__ __forceinline__ float ts_get(float v, int) { return v; }
__device__ __forceinline__ float ts_get(ts_f2 v, int r) { return r ? v.y : v.x; }
__device__ __forceinline__ float ts_get(ts_f4 v, int r) { return r == 0 ? v.x : (r == 1 ? v.y : (r == 2 ? v.z : v.w)); }

// ---- the scatter ------------------------------------------------------------------------------------------------------
// Every global load of the kernel is UNCONDITIONAL (clamped addresses instead of predicates, tiles past the end re-load
// the last one) and every wave issues the same loads in the same order: the compiler's s_waitcnt vmcnt(N) then counts
// exactly, and a tile's loads stay in flight for four iterations instead of being drained at the next use of any load.
template <int CH, bool WEIGHTED, bool VEC>
__global__ __launch_bounds__(TS_THREADS) void ts_scatter_kernel(int c, int m, int L, int tl, int q, int cap,
                                                                const float *__restrict__ grad_out, size_t src_bstride,
                                                                const uint32_t *__restrict__ ent, const int *__restrict__ wr,
                                                                float *__restrict__ grad_table, int set)
{
    typedef typename TsVec<CH>::type vec;
    extern __shared__ float ts_f[];
    const int accn = (m * CH + 3) & ~3;
    vec *acc = reinterpret_cast<vec *>(ts_f);                    // [m] sums of this workgroup's CH channels
    float *buf0 = ts_f + accn, *buf1 = buf0 + tl * CH;           // two tiles [tl][CH]
    int *s_wr = reinterpret_cast<int *>(buf1 + tl * CH);         // [q][16]
    const int bi = blockIdx.y, c0 = blockIdx.x * CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // staging: thread <-> one float4 of the tile = 4 consecutive sources (quad qd) of row r; rows fastest over the lanes
    const int nf = CH * (tl >> 2);
    const bool stager = tid < nf;
    const int r = tid % CH, qd = min(tid, nf - 1) / CH;
    const float *grow = grad_out + (size_t)bi * src_bstride + (size_t)min(c0 + r, c - 1) * L;
    const uint32_t *pent = ent + (size_t)bi * q * cap * (WEIGHTED ? 2 : 1);

    TS_STAMP_DECL
    ts_f4 pre[4];
    const int qfull = 4 * min(qd, (tl >> 2) - 1), last_len = L - (q - 1) * tl;          // (all tiles but the last are tl long)
    const int qlast = 4 * min(qd, max(last_len >> 2, 1) - 1);
    auto load_tile = [&](int p, ts_f4 &dst) {
        p = min(p, q - 1);
        const int p0 = p * tl, plen = min(tl, L - p0);
        if (VEC) {
            dst = __builtin_nontemporal_load(reinterpret_cast<const ts_f4 *>(grow + p0 + (p == q - 1 ? qlast : qfull)));
        } else {
            const float *src = grow + p0;
            dst.x = src[min(4 * qd + 0, plen - 1)];
            dst.y = src[min(4 * qd + 1, plen - 1)];
            dst.z = src[min(4 * qd + 2, plen - 1)];
            dst.w = src[min(4 * qd + 3, plen - 1)];
        }
    };
    // transposed store [tl][CH]: a thread writes its 4 sources' values of row r as 4 dwords.  In source order the 32 lanes
    // of a store group would land on 8 banks (lanes = 4 rows x 8 quads, 64 B between quads); each lane therefore starts at
    // component (h + s) & 3 of its float4 in store s, h = a function of its quad: 32 lanes, 32 banks.
    const int h = CH == 4 ? (qd >> 1) & 3 : (qd >> 2) & 3;
    int soff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) soff[i] = (4 * qd + ((i + h) & 3)) * CH + r;
    auto store_tile = [&](int p, const ts_f4 &src) {
        float *buf = (p & 1) ? buf1 : buf0;
        if (!stager) return;
        if (CH == 1) {
            reinterpret_cast<ts_f4 *>(buf)[qd] = src;
        } else {
            const ts_f4 t = (h & 1) ? ts_f4{src.y, src.z, src.w, src.x} : src;       // t[i] = src[(i + (h & 1)) & 3]
            const ts_f4 u = (h & 2) ? ts_f4{t.z, t.w, t.x, t.y} : t;                 // u[i] = src[(i + h) & 3]
            buf[soff[0]] = u.x;
            buf[soff[1]] = u.y;
            buf[soff[2]] = u.z;
            buf[soff[3]] = u.w;
        }
    };

    // prologue: the first four tiles on their way, sums cleared, wave tables in LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) load_tile(j, pre[j]);
    for (int i = tid; i < accn / 4; i += TS_THREADS) reinterpret_cast<ts_f4 *>(ts_f)[i] = ts_f4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < q * TS_WAVES; i += TS_THREADS) s_wr[i] = wr[(size_t)bi * q * TS_WAVES + i];
    __syncthreads();

    // the entries of this wave's first TS_NPF chunks of a tile, fetched while the previous tile is walked: two register
    // sets, by the tile's parity (no copies); one address per tile, the chunks at immediate offsets
    uint32_t ek[2][TS_NPF];
    float ew[2][TS_NPF];
    auto entry = [&](int p, int slot, uint32_t &key, float &w) {
        const size_t at = (size_t)p * cap + min(slot, cap - 1);
        if (WEIGHTED) {
            const uint2 e2 = reinterpret_cast<const uint2 *>(pent)[at];
            key = e2.x;
            w = __uint_as_float(e2.y);
        } else {
            key = pent[at];
            w = 1.f;
        }
    };
    int desc_next = 0;                                          // descriptor of the tile whose entries were fetched last
    size_t ebase_next = 0;                                      // first entry slot of that tile
    auto fetch = [&](int p, uint32_t(&nk)[TS_NPF], float(&nw)[TS_NPF]) {
        if (p < q) {
            desc_next = __builtin_amdgcn_readfirstlane(s_wr[p * TS_WAVES + wave]);
            ebase_next = (size_t)p * cap;
        } else desc_next &= 0xffff;                             // past the end: no chunks (the loads below repeat the last tile's)
        const size_t at = ebase_next + (desc_next & 0xffff);    // (first + 64 TS_NPF <= cap: the plan's slack)
        if (WEIGHTED) {
            const uint2 *src = reinterpret_cast<const uint2 *>(pent) + at + lane;
#pragma unroll
            for (int j = 0; j < TS_NPF; ++j) {
                const uint2 e2 = src[64 * j];
                nk[j] = e2.x;
                nw[j] = __uint_as_float(e2.y);
            }
        } else {
            const uint32_t *src = pent + at + lane;
#pragma unroll
            for (int j = 0; j < TS_NPF; ++j) {
                nk[j] = src[64 * j];
                nw[j] = 1.f;
            }
        }
    };
    fetch(0, ek[0], ew[0]);
    TS_STAMP(0)

    // sorted runs: segmented sum over the lanes of one chunk -- on return every target's last lane holds the sum of the
    // target's lanes and `tail` marks it (padding lanes carry k = INT_MAX)
    // (the rounds beyond the first: s = the sums after one shift-and-add round, mask = ballot(same))
    auto scan_more = [&](int k, bool same, unsigned long long mask, vec v, vec s) -> vec {
        unsigned long long mm = mask & (mask << 1);
        if (mm) {                                           // three or more lanes of one target
            int R = 1;                                      // longest run of set bits = rounds needed in all
            for (; mm; mm &= mm << 1) ++R;
            if (R <= 6) {
                for (int rr = 1; rr < R; ++rr) {
                    const vec t = ts_shr1(s);
                    s = same ? v + t : v;
                }
            } else {                                        // log-step segmented scan (keys are sorted)
                s = v;
                for (int d = 1; d < 64; d <<= 1) {
                    const int kd = __shfl_up(k, d);
                    const vec t = ts_up(s, d);
                    if (lane >= d && kd == k) s = s + t;
                }
            }
        }
        return s;
    };
    auto scan = [&](int k, bool valid, vec v, bool &tail) -> vec {
        const int kp = __builtin_amdgcn_update_dpp(-1, k, TS_DPP_WAVE_SHR1, 0xF, 0xF, false);
        const bool same = valid && kp == k;                 // this lane continues the run of the lane below
        const unsigned long long mask = __ballot(same);
        vec s = v;
        if (mask) {
            const vec t1 = ts_shr1(v);
            s = scan_more(k, same, mask, v, same ? v + t1 : v);
        }
        tail = valid && !(lane < 63 && ((mask >> (lane + 1)) & 1ull));
        return s;
    };

    auto walk = [&](int p, const uint32_t(&ck)[TS_NPF], const float(&cw)[TS_NPF], uint32_t(&nk)[TS_NPF], float(&nw)[TS_NPF]) {
        const int desc = desc_next;                          // (set when this tile's entries were fetched)
        const int first = desc & 0xffff, np = (desc >> 16) & 31, ntot = np + (int)((unsigned)desc >> 21);
        fetch(p + 1, nk, nw);
        const vec *rows = reinterpret_cast<const vec *>((p & 1) ? buf1 : buf0);
        // The shapes of nearly every (wave, tile) of a large problem -- two or three full plain chunks and one chunk of sorted
        // run -- as straight-line code: no per-chunk branches, one shift-and-add round for the run (targets with three or more
        // lanes in the run fall through to the general rounds).
        auto fast = [&](auto npc) {
            constexpr int NP = decltype(npc)::value;
            int kk[NP];
            vec rr[NP], aa[NP];
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                kk[j] = (int)(ck[j] >> TS_EBITS);
                rr[j] = rows[ck[j] & ((1u << TS_EBITS) - 1)];
            }
            const bool ok = ck[NP] != TS_NONE;
            const int k2 = ok ? (int)(ck[NP] >> TS_EBITS) : 0x7fffffff;
            const vec r2 = rows[ok ? (int)(ck[NP] & ((1u << TS_EBITS) - 1)) : 0];
#pragma unroll
            for (int j = 0; j < NP; ++j) aa[j] = acc[kk[j]];
            const vec x2 = WEIGHTED ? r2 * cw[NP] : r2;
            const int kp = __builtin_amdgcn_update_dpp(-1, k2, TS_DPP_WAVE_SHR1, 0xF, 0xF, false);
            const bool same = ok && kp == k2;
            const unsigned long long mask = __ballot(same);
            const vec t1 = ts_shr1(x2);
            vec s2 = same ? x2 + t1 : x2;
#pragma unroll
            for (int j = 0; j < NP; ++j) acc[kk[j]] = WEIGHTED ? aa[j] + rr[j] * cw[j] : aa[j] + rr[j];
            s2 = scan_more(k2, same, mask, x2, s2);          // (targets with three or more lanes: further rounds)
            if (ok && !(lane < 63 && ((mask >> (lane + 1)) & 1ull))) acc[k2] = acc[k2] + s2;
        };
        if (np == 2 && ntot == 3) { fast(std::integral_constant<int, 2>{}); return; }
        if (np == 3 && ntot == 4) { fast(std::integral_constant<int, 3>{}); return; }
        // any other shape: the first TS_NPF chunks from the prefetched entries -- the source rows of all chunks first, the
        // plain chunks (distinct targets across all of them) as one batch of reads, adds and writes, then the run in order
        int k[TS_NPF];
        bool valid[TS_NPF];
        vec v[TS_NPF], a[TS_NPF];
#pragma unroll
        for (int j = 0; j < TS_NPF; ++j)
            if (j < ntot) {
                valid[j] = ck[j] != TS_NONE;
                k[j] = valid[j] ? (int)(ck[j] >> TS_EBITS) : 0x7fffffff;
                v[j] = rows[valid[j] ? (int)(ck[j] & ((1u << TS_EBITS) - 1)) : 0];
            }
#pragma unroll
        for (int j = 0; j < TS_NPF; ++j)
            if (j < np) a[j] = acc[k[j]];
#pragma unroll
        for (int j = 0; j < TS_NPF; ++j)
            if (j < np) acc[k[j]] = WEIGHTED ? a[j] + v[j] * cw[j] : a[j] + v[j];
#pragma unroll
        for (int j = 0; j < TS_NPF; ++j)
            if (j >= np && j < ntot) {
                bool tail;
                const vec sum = scan(k[j], valid[j], WEIGHTED ? v[j] * cw[j] : v[j], tail);
                if (tail) acc[k[j]] = acc[k[j]] + sum;      // a target's last lane: its sole writer in this tile
            }
        for (int j = TS_NPF; j < ntot; ++j) {                // more chunks than the prefetch holds (skewed tiles): one by one
            uint32_t key;
            float w;
            entry(p, first + 64 * j + lane, key, w);
            const bool ok = key != TS_NONE;
            const int kk = ok ? (int)(key >> TS_EBITS) : 0x7fffffff;
            vec vv = rows[ok ? (int)(key & ((1u << TS_EBITS) - 1)) : 0];
            if (WEIGHTED) vv = vv * w;
            if (j < np) acc[kk] = acc[kk] + vv;
            else {
                bool tail;
                vv = scan(kk, ok, vv, tail);
                if (tail) acc[kk] = acc[kk] + vv;
            }
        }
    };

    const int q4 = (q + 3) & ~3;
    for (int pb = 0; pb < q4; pb += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = pb + j;
            store_tile(p, pre[j]);
            load_tile(p + 4, pre[j]);
            TS_STAMP(1)
            __syncthreads();     // tile p is in LDS; every wave has finished tile p - 1
            TS_STAMP(2)
            walk(p, ek[j & 1], ew[j & 1], ek[(j + 1) & 1], ew[(j + 1) & 1]);
        }
    }
    __syncthreads();
    for (int k = tid; k < m; k += TS_THREADS) {
        const vec a = acc[k];
#pragma unroll
        for (int rr = 0; rr < CH; ++rr)
            if (c0 + rr < c) {
                float *dst = grad_table + ((size_t)bi * c + c0 + rr) * m + k;
                *dst = set ? ts_get(a, rr) : *dst + ts_get(a, rr);
            }
    }
    TS_STAMP(7)
    TS_STAMP_FLUSH
}

template <int CH, bool WEIGHTED>
static hipError_t ts_launch(const TsPlan &p, int b, int c, int m, int L, size_t src_bstride, const float *grad_out,
                            const uint32_t *ent, const int *wr, float *grad_table, int set, hipStream_t s)
{
    const bool vec_ok = (L & 3) == 0 && (src_bstride & 3) == 0 && (((uintptr_t)grad_out) & 15) == 0;
    const dim3 grid((c + CH - 1) / CH, b);
    hipError_t e;
    if (vec_ok) {
        e = allow_big_lds((const void *)ts_scatter_kernel<CH, WEIGHTED, true>, p.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ts_scatter_kernel<CH, WEIGHTED, true>), grid, dim3(TS_THREADS), p.lds, s, c, m, L, p.tl, p.q, p.cap,
                           grad_out, src_bstride, ent, wr, grad_table, set);
    } else {
        e = allow_big_lds((const void *)ts_scatter_kernel<CH, WEIGHTED, false>, p.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ts_scatter_kernel<CH, WEIGHTED, false>), grid, dim3(TS_THREADS), p.lds, s, c, m, L, p.tl, p.q, p.cap,
                           grad_out, src_bstride, ent, wr, grad_table, set);
    }
    return hipGetLastError();
}

hipError_t scatter_via_tiles(int b, int c, int m, int L, int nt, size_t src_bstride, const float *grad_out, const int *idx,
                             const float *weight, float *grad_table, void *workspace, long long ws_ints, hipStream_t s,
                             bool overwrite)
{
    TsPlan p;
    const bool weighted = weight != nullptr;
    if (!workspace || (((uintptr_t)workspace) & 7) || !ts_plan(b, c, m, L, nt, weighted, p) || ws_ints < p.ints)
        return hipErrorNotSupported;
    uint32_t *ent = (uint32_t *)workspace;
    int *wr = (int *)(ent + p.ent_words);
    hipError_t e;
    if (weighted) {
        e = allow_big_lds((const void *)ts_build_kernel<true>, p.lds_build);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ts_build_kernel<true>), dim3(p.q, b), dim3(TS_THREADS), p.lds_build, s, m, L, nt, p.tl, p.q, p.ppp, p.cap,
                           idx, weight, ent, wr);
    } else {
        e = allow_big_lds((const void *)ts_build_kernel<false>, p.lds_build);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((ts_build_kernel<false>), dim3(p.q, b), dim3(TS_THREADS), p.lds_build, s, m, L, nt, p.tl, p.q, p.ppp, p.cap,
                           idx, weight, ent, wr);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int set = overwrite ? 1 : 0;
#define TS_GO(CHV)                                                                                                         \
    return weighted ? ts_launch<CHV, true>(p, b, c, m, L, src_bstride, grad_out, ent, wr, grad_table, set, s)              \
                    : ts_launch<CHV, false>(p, b, c, m, L, src_bstride, grad_out, ent, wr, grad_table, set, s)
    if (p.ch == 4) { TS_GO(4); }
    if (p.ch == 2) { TS_GO(2); }
    TS_GO(1);
#undef TS_GO
}

} // namespace geot
