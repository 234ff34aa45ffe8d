// sa_mlp.hip -- fused SetAbstraction body for gfx950 (MI355X):
//   neighbourhood gather + centre subtraction + concat + shared MLP (1x1 conv stack with
//   folded BatchNorm + ReLU) + max over nsample, in one kernel.
//
// Replaces the reference chain (behaviour, not code)
//   QueryAndGroup.forward        pointnet2/pointnet2_utils.py:343-358  (2x grouping_operation, cat)
//   SharedMLP                    pointnet2/pytorch_utils.py:8-33       (Conv2d 1x1 + BN + ReLU)
//   F.max_pool2d over nsample    pointnet2/pointnet2_modules.py:360-363
// which materialises (B, 3+C, npoint, nsample) and every (B, C_l, npoint, nsample)
// activation in HBM.  Here the only HBM traffic is the gathered inputs and the
// (B, C_out, npoint) result; this is the one genuinely dense contraction on the hot
// path, so it runs on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32, fp32
// accumulate -- bit-for-bit an fmaf chain, so parity with an fp32 reference holds to
// rounding-order level).
//
// Mapping: one wave owns 32 consecutive rows (row = group*nsample + sample) through all
// layers; activations stay in a wave-private LDS tile [32][K+1] (odd stride => the
// A-fragment read, lane -> (row = lane&31, k = lane>>5), is bank-conflict free), weights
// for all layers sit in LDS as W^T [K][C] (B-fragment read is lane-contiguous).  No
// cross-wave synchronisation after the initial weight load.
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "geot_common.h"
#include "geot_hip.h"

namespace geot {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SA_WAVES = 12;  // at most (3 per SIMD); the launcher uses 8 or 4 when the activation tiles do not fit the LDS
constexpr int SA_MAX_LAYERS = 4;

struct SaDesc {
    int nlayers;
    int kp[SA_MAX_LAYERS];   // padded input width of layer l (even)
    int cp[SA_MAX_LAYERS];   // padded output width of layer l (32, 64, 128 or 256)
    int woff[SA_MAX_LAYERS]; // float offset of W^T [kp][cp] in the parameter block
    int boff[SA_MAX_LAYERS]; // float offset of bias [cp]
    int relu_mask;           // bit l = ReLU after layer l
    int total;               // floats in the parameter block
    int act_stride;          // floats per activation row (widest STORED activation + 1, odd; the last layer is pooled from registers)
};

template <int NCT>
__device__ __forceinline__ void sa_layer(const float *__restrict__ W, const float *__restrict__ bias,
                                         int kp, int cp, bool relu, float *__restrict__ act,
                                         int act_stride, f32x16 (&acc)[NCT])
{
    const int lane = lane_id(), r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        float bv = bias[ct * 32 + r];
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][e] = bv;
    }
    const float *arow = act + r * act_stride + h;
    const float *wrow = W + h * cp + r;
#ifdef GEOT_SA_LAB_NOMFMA
    kp = 2;
#endif
    for (int k0 = 0; k0 < kp; k0 += 2) {
        float a = arow[k0];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            float b = wrow[k0 * cp + ct * 32];
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ct], 0, 0, 0);
        }
    }
    if (relu) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][e] = fmaxf(acc[ct][e], 0.f);
    }
}

// D layout of the 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
template <int NCT>
__device__ __forceinline__ void sa_store_act(const f32x16 (&acc)[NCT], float *__restrict__ act, int act_stride)
{
    const int lane = lane_id(), c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            act[row * act_stride + ct * 32 + c] = acc[ct][e];
        }
}

// Max over the rows of each group inside the tile, merged into pool[g_local][col].
// gpt = groups per tile (1 when nsample >= 32, else 32 / nsample in {2, 4}).
template <int NCT>
__device__ __forceinline__ void sa_pool(const f32x16 (&acc)[NCT], float *__restrict__ pool, int cp, int gpt)
{
    const int lane = lane_id(), c = lane & 31;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        float m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = fmaxf(fmaxf(acc[ct][4 * j], acc[ct][4 * j + 1]), fmaxf(acc[ct][4 * j + 2], acc[ct][4 * j + 3]));
            m[j] = fmaxf(v, __shfl_xor(v, 32));
        }
        if (lane < 32) {
            if (gpt == 1) {
                float v = fmaxf(fmaxf(m[0], m[1]), fmaxf(m[2], m[3]));
                pool[ct * 32 + c] = fmaxf(pool[ct * 32 + c], v);
            } else if (gpt == 2) {
                pool[ct * 32 + c] = fmaxf(pool[ct * 32 + c], fmaxf(m[0], m[1]));
                pool[cp + ct * 32 + c] = fmaxf(pool[cp + ct * 32 + c], fmaxf(m[2], m[3]));
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) pool[j * cp + ct * 32 + c] = fmaxf(pool[j * cp + ct * 32 + c], m[j]);
            }
        }
    }
}

// nsample == 32: the tile IS the group, so the max over its 32 rows is final -- it stays in registers (no LDS
// pool): after the cross-half max both half-waves hold every column's result, half h keeps column tiles
// 2p + h, i.e. o[p] = column p*64 + h*32 + (lane & 31).
template <int NCT>
__device__ __forceinline__ void sa_pool_direct(const f32x16 (&acc)[NCT], float (&o)[4])
{
    const int h = lane_id() >> 5;
    float val[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        float v = acc[ct][0];
#pragma unroll
        for (int e = 1; e < 16; ++e) v = fmaxf(v, acc[ct][e]);
        val[ct] = fmaxf(v, __shfl_xor(v, 32));
    }
#pragma unroll
    for (int p2 = 0; p2 < 4; ++p2) {
        if (2 * p2 < NCT) {
            const float lo = val[2 * p2], hi = val[(2 * p2 + 1 < NCT) ? 2 * p2 + 1 : 2 * p2];
            o[p2] = h ? hi : lo;
        }
    }
}

// out[b, col, group] <- the register-pooled values of group g (column map: sa_pool_direct).  NP = padded width / 64
// store instructions, every lane active (the fast path requires c_out == padded width): no branch, so the
// compiler can count them when it places the s_waitcnt of the loads issued just before.
template <int NP>
__device__ __forceinline__ void sa_write_direct(const float (&o)[4], int g, int npoint, int c_out, float *__restrict__ out)
{
#ifdef GEOT_SA_LAB_NOOUT
    if (g != 0) return;
#endif
    const int lane = lane_id(), c = lane & 31, h = lane >> 5;
    const int bi = g / npoint, gi = g - bi * npoint;
    float *dst = out + ((size_t)bi * c_out + h * 32 + c) * npoint + gi;
#pragma unroll
    for (int p2 = 0; p2 < NP; ++p2) dst[(size_t)p2 * 64 * npoint] = o[p2];
}

// layer-0 operands of one tile row, in registers between the global loads and the LDS writes
struct SaRow {
    float p[3], q[3], f[4];
};

template <int NCT>
__device__ __forceinline__ void sa_run_layer(const SaDesc &d, int l, const float *__restrict__ P,
                                             float *__restrict__ act, float *__restrict__ pool, int gpt,
                                             bool direct, float (&o)[4])
{
    f32x16 acc[NCT];
    sa_layer<NCT>(P + d.woff[l], P + d.boff[l], d.kp[l], d.cp[l], (d.relu_mask >> l) & 1, act, d.act_stride, acc);
#if defined(GEOT_SA_LAB_NOSTORE) || defined(GEOT_SA_LAB_NOPOOL)
    {   // lab (tools/sa_lab.py): keep the accumulators alive while a phase is removed
        float keep = 0.f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) keep += acc[ct][0] + acc[ct][15];
        if (keep == 12345.678f) pool[0] = keep;
    }
#endif
#ifdef GEOT_SA_LAB_NOSTORE
    if (l + 1 < d.nlayers) return;
#endif
#ifdef GEOT_SA_LAB_NOPOOL
    if (l + 1 == d.nlayers) return;
#endif
    if (l + 1 < d.nlayers) sa_store_act<NCT>(acc, act, d.act_stride);
    else if (direct) sa_pool_direct<NCT>(acc, o);
    else sa_pool<NCT>(acc, pool, d.cp[l], gpt);
}

// MAXW = widest layer / 32 the variant supports: the 8-tile (256-wide) accumulators cost 128 VGPRs, which caps
// the occupancy at 2 waves per SIMD; networks up to 128 wide use the lean variant and run 3.
template <int MAXW>
__global__ __launch_bounds__(MAXW >= 8 ? 512 : SA_WAVES * 64) void sa_group_mlp_max_kernel(
    SaDesc d, int b, int n, int npoint, int nsample, int c_feat, int c_out,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ features, const int *__restrict__ idx, float xyz_scale,
    const float *__restrict__ params, float *__restrict__ out, int fast_np, int run_len)
{
    extern __shared__ float sa_lds[];
    float *P = sa_lds;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int cp_last = d.cp[d.nlayers - 1];
    const int gpt = nsample >= 32 ? 1 : 32 / nsample;  // groups per 32-row tile
    float *act = sa_lds + d.total + wave * (32 * d.act_stride + gpt * cp_last);
    float *pool = act + 32 * d.act_stride;
    const int nwaves = blockDim.x >> 6;

    const int tpg = nsample >= 32 ? nsample / 32 : 1;  // tiles per group
    const long long ngroups = (long long)b * npoint;
    const long long nunits = (ngroups + gpt - 1) / gpt;
    const int k_in = 3 + c_feat;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (fast_np) {
        // ---- nsample == 32, c_feat <= 8, no padded output columns, < 2^31 groups: the tile IS the group.  Pooled in
        // registers; software-pipelined so that the global loads of the NEXT group's rows are in flight before
        // this group's 128 scattered 4-byte stores are issued: vmcnt retires in order, and a gather issued behind
        // the stores would wait for their acknowledgement (the stores alone cost 12 % of the launch that way).
        const int r = lane & 31, h = lane >> 5;
        float *arow = act + r * d.act_stride;
        // a workgroup owns a contiguous run of groups and its waves walk it together: the 4-byte stores of one
        // output column then land next to each other in the SAME L2 within a few tiles and leave it as whole lines
        // (interleaving the groups over workgroups scattered every line over all eight L2s: partial-sector writes)
        auto load = [&](SaRow &R, int g) {
#ifdef GEOT_SA_LAB_NOGATHER
            for (int x = 0; x < 3; ++x) { R.p[x] = 0.01f * r; R.q[x] = 0.f; }
            for (int j = 0; j < 4; ++j) R.f[j] = 0.02f * g;
            return;
#endif
            const int bi = g / npoint;
            const int a = idx[(size_t)g * 32 + r];
            const float *pp = xyz + ((size_t)bi * n + a) * 3, *q = new_xyz + (size_t)g * 3;
#pragma unroll
            for (int x = 0; x < 3; ++x) { R.p[x] = pp[x]; R.q[x] = q[x]; }
            const float *f = features + (size_t)bi * c_feat * n + a;
#pragma unroll
            for (int j = 0; j < 4; ++j) R.f[j] = (h + 2 * j < c_feat) ? f[(size_t)(h + 2 * j) * n] : 0.f;
        };
        auto commit = [&](const SaRow &R) {
            if (h == 0) {
#pragma unroll
                for (int x = 0; x < 3; ++x) arow[x] = (R.p[x] - R.q[x]) * xyz_scale;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (h + 2 * j < c_feat) arow[3 + h + 2 * j] = R.f[j];
            for (int ch = k_in + h; ch < d.kp[0]; ch += 2) arow[ch] = 0.f;
        };
        // A wave walks RUNS of 8 consecutive groups and keeps their pooled columns in registers (hist): a lane then
        // owns, per column, 8 consecutive floats of out[b, col, :] = one aligned 32-byte sector, written as two
        // dwordx4 stores -- 4x fewer, 4x larger write requests than one 4-byte store per (group, column) (the
        // scattered 4-byte stores were request-rate-bound in the L2: 16 % of the launch).  The stores are
        // issued straight-line before the LDS writes of the rows loaded just before them (wait = vmcnt(2 NP)).
        float hist[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) hist[j][p2] = 0.f;
        auto store_run = [&](int g0, auto np_tag) {
            constexpr int NP = decltype(np_tag)::value;
            const int bi = g0 / npoint, gi0 = g0 - bi * npoint;
            float *dst = out + ((size_t)bi * c_out + h * 32 + (lane & 31)) * npoint + gi0;
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) {
                typedef float f32x4 __attribute__((ext_vector_type(4)));
                f32x4 lo = {hist[0][p2], hist[1][p2], hist[2][p2], hist[3][p2]};
                f32x4 hi = {hist[4][p2], hist[5][p2], hist[6][p2], hist[7][p2]};
                f32x4 *d4 = reinterpret_cast<f32x4 *>(dst + (size_t)p2 * 64 * npoint);
                d4[0] = lo;
                d4[1] = hi;
            }
        };
        // run length 1 (few groups per wave: keep every wave busy): the newest entry, 4-byte stores
        auto store_one = [&](int g, auto np_tag) {
            constexpr int NP = decltype(np_tag)::value;
            const int bi = g / npoint, gi = g - bi * npoint;
            float *dst = out + ((size_t)bi * c_out + h * 32 + (lane & 31)) * npoint + gi;
#pragma unroll
            for (int p2 = 0; p2 < NP; ++p2) dst[(size_t)p2 * 64 * npoint] = hist[7][p2];
        };
        auto store_then_commit = [&](int g0, bool do_store, const SaRow *Rn) {
            if (!do_store) { if (Rn) commit(*Rn); }
            else if (run_len == 8) {
                if (fast_np == 1) { store_run(g0, std::integral_constant<int, 1>()); if (Rn) commit(*Rn); }
                else if (fast_np == 2) { store_run(g0, std::integral_constant<int, 2>()); if (Rn) commit(*Rn); }
                else { store_run(g0, std::integral_constant<int, 4>()); if (Rn) commit(*Rn); }
            } else {
                if (fast_np == 1) { store_one(g0, std::integral_constant<int, 1>()); if (Rn) commit(*Rn); }
                else if (fast_np == 2) { store_one(g0, std::integral_constant<int, 2>()); if (Rn) commit(*Rn); }
                else { store_one(g0, std::integral_constant<int, 4>()); if (Rn) commit(*Rn); }
            }
        };
        // runs: a workgroup owns a contiguous range of runs (one L2 sees all of a line's sectors), wave w takes
        // runs first + w, first + w + nwaves, ...
        const int nruns = (int)(ngroups / run_len);      // run_len == 8 requires npoint % 8 == 0
        const int rchunk = (nruns + gridDim.x - 1) / gridDim.x;
        const int rend = min((int)(blockIdx.x + 1) * rchunk, nruns);
        int run = blockIdx.x * rchunk + wave, j = 0;
        SaRow R;
        if (run < rend) load(R, run * run_len);   // the first rows travel while the weights are staged
        for (int i = threadIdx.x; i < d.total; i += blockDim.x) P[i] = params[i];
        __syncthreads();
        if (run < rend) commit(R);
        while (run < rend) {
            for (int l = 0; l < d.nlayers; ++l) {
                const int w32 = d.cp[l] >> 5;
                if (w32 == 1) sa_run_layer<1>(d, l, P, act, pool, 1, true, o);
                else if (w32 == 2) sa_run_layer<2>(d, l, P, act, pool, 1, true, o);
                else if (MAXW >= 8 && w32 == 8) sa_run_layer<(MAXW >= 8 ? 8 : 4)>(d, l, P, act, pool, 1, true, o);
                else sa_run_layer<4>(d, l, P, act, pool, 1, true, o);
            }
#pragma unroll
            for (int jj = 0; jj < 7; ++jj)
#pragma unroll
                for (int p2 = 0; p2 < 4; ++p2) hist[jj][p2] = hist[jj + 1][p2];
#pragma unroll
            for (int p2 = 0; p2 < 4; ++p2) hist[7][p2] = o[p2];
            const bool full = j == run_len - 1;
            const int g0 = run * run_len;
            const int nrun = full ? run + nwaves : run, nj = full ? 0 : j + 1;
            if (nrun < rend) {
                load(R, nrun * run_len + nj);
                store_then_commit(g0, full, &R);
            } else {
                store_then_commit(g0, full, nullptr);
            }
            run = nrun;
            j = nj;
        }
        return;
    }
    for (int i = threadIdx.x; i < d.total; i += blockDim.x) P[i] = params[i];
    __syncthreads();
    const long long uchunk = (nunits + gridDim.x - 1) / gridDim.x, uend = min((long long)(blockIdx.x + 1) * uchunk, nunits);
    for (long long u = (long long)blockIdx.x * uchunk + wave; u < uend; u += nwaves) {   // contiguous run per workgroup
        for (int i = lane; i < gpt * cp_last; i += 64) pool[i] = -INFINITY;
        for (int t = 0; t < tpg; ++t) {
            // ---- layer-0 input: row r of the tile <- (xyz[idx]-centre)*scale, features[:, idx]
            {
                const int r = lane & 31, h = lane >> 5;
                long long g = u * gpt + (nsample >= 32 ? 0 : r / nsample);
                int s = nsample >= 32 ? t * 32 + r : r % nsample;
                float *arow = act + r * d.act_stride;
                if (g < ngroups) {
                    int bi = (int)(g / npoint);
                    int a = idx[g * nsample + s];
                    if (h == 0) {
                        const float *p = xyz + ((size_t)bi * n + a) * 3, *q = new_xyz + g * 3;
                        arow[0] = (p[0] - q[0]) * xyz_scale;
                        arow[1] = (p[1] - q[1]) * xyz_scale;
                        arow[2] = (p[2] - q[2]) * xyz_scale;
                    }
                    const float *f = features + (size_t)bi * c_feat * n + a;
                    for (int ch = h; ch < c_feat; ch += 2) arow[3 + ch] = f[(size_t)ch * n];
                    for (int ch = k_in + h; ch < d.kp[0]; ch += 2) arow[ch] = 0.f;
                } else {
                    for (int ch = h; ch < d.kp[0]; ch += 2) arow[ch] = 0.f;
                }
            }
            for (int l = 0; l < d.nlayers; ++l) {
                const int w32 = d.cp[l] >> 5;
                if (w32 == 1) sa_run_layer<1>(d, l, P, act, pool, gpt, false, o);
                else if (w32 == 2) sa_run_layer<2>(d, l, P, act, pool, gpt, false, o);
                else if (MAXW >= 8 && w32 == 8) sa_run_layer<(MAXW >= 8 ? 8 : 4)>(d, l, P, act, pool, gpt, false, o);
                else sa_run_layer<4>(d, l, P, act, pool, gpt, false, o);
            }
        }
        // ---- pooled result -> out[b, col, group]
        for (int i = lane; i < gpt * c_out; i += 64) {
            int gl = i / c_out, col = i - gl * c_out;
            long long g = u * gpt + gl;
            if (g < ngroups) {
                int bi = (int)(g / npoint);
                int gi = (int)(g - (long long)bi * npoint);
                out[((size_t)bi * c_out + col) * npoint + gi] = pool[gl * cp_last + col];
            }
        }
    }
}

static inline int pad_cols(int c) { return c <= 32 ? 32 : c <= 64 ? 64 : c <= 128 ? 128 : 256; }

} // namespace geot

using namespace geot;

GEOT_EXPORT int geot_sa_param_floats(int c_feat, int nlayers, const int *widths)
{
    if (nlayers < 1 || nlayers > SA_MAX_LAYERS || c_feat < 0) return -1;
    long long total = 0;
    int kp = (3 + c_feat + 1) & ~1;
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] < 1 || widths[l] > 256) return -1;
        int cp = pad_cols(widths[l]);
        total += (long long)kp * cp + cp;
        kp = cp;
    }
    return (int)total;
}

GEOT_EXPORT int geot_sa_group_mlp_max(int b, int n, int npoint, int nsample, int c_feat,
                                      const float *xyz, const float *new_xyz, const float *features,
                                      const int *idx, float xyz_scale, int nlayers, const int *widths,
                                      int relu_mask, const float *params, float *out, void *stream)
{
    if (b < 0 || n < 0 || npoint < 0 || nlayers < 1 || nlayers > SA_MAX_LAYERS || c_feat < 0)
        return hipErrorInvalidValue;
    if (!(nsample == 8 || nsample == 16 || (nsample >= 32 && nsample % 32 == 0))) return hipErrorInvalidValue;
    if (c_feat > 0 && !features) return hipErrorInvalidValue;
    if (b == 0 || npoint == 0) return hipSuccess;
    SaDesc d{};
    d.nlayers = nlayers;
    d.relu_mask = relu_mask;
    int kp = (3 + c_feat + 1) & ~1, off = 0, maxw = kp;
    for (int l = 0; l < nlayers; ++l) {
        if (widths[l] < 1 || widths[l] > 256) return hipErrorInvalidValue;
        int cp = pad_cols(widths[l]);
        d.kp[l] = kp; d.cp[l] = cp; d.woff[l] = off; off += kp * cp; d.boff[l] = off; off += cp;
        if (l + 1 < nlayers && cp > maxw) maxw = cp; // the last layer's output never goes to the activation tile
        kp = cp;
    }
    d.total = off;
    d.act_stride = maxw + 1;
    // As many waves per workgroup (= per CU: the weights + activation tiles fill its LDS) as fit next to the
    // weights, up to 3 per SIMD: one wave's gather, LDS round trips and accumulator hand-offs between layers
    // then overlap with the others' MFMA chains.
    int gpt = nsample >= 32 ? 1 : 32 / nsample;
    bool wide = false;
    for (int l = 0; l < nlayers; ++l) wide = wide || d.cp[l] > 128;
    const size_t per_wave = 32 * (size_t)d.act_stride + (size_t)gpt * d.cp[nlayers - 1];
    int waves = wide ? 8 : SA_WAVES;
    while (waves > 4 && ((size_t)d.total + waves * per_wave) * sizeof(float) > 160 * 1024) waves -= 4;
    const size_t lds = ((size_t)d.total + waves * per_wave) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    {   // > 64 KB of dynamic LDS is opt-in, per device and per kernel: raised once (geot_common.h allow_big_lds)
        hipError_t e = wide ? allow_big_lds((const void *)sa_group_mlp_max_kernel<8>, lds)
                            : allow_big_lds((const void *)sa_group_mlp_max_kernel<4>, lds);
        if (e != hipSuccess) return e;
    }
    long long nunits = ((long long)b * npoint + gpt - 1) / gpt;
    long long blocks = (nunits + waves - 1) / waves;
    // persistent workgroups: one per CU (the weights + activation tiles fill its LDS), each looping over its
    // share of the tiles, so the 52 KB of weights are staged once per CU and not once per 8 tiles
    const int n_cus = device_cus();
    if (blocks > n_cus) blocks = n_cus;
    // register-pooled, software-pipelined path (see the kernel): one 32-row tile per group, few feature channels,
    // no padded output columns
    const int c_last = widths[nlayers - 1];
    int fast_np = 0;
    if (nsample == 32 && c_feat <= 8 && c_last == d.cp[nlayers - 1] && c_last >= 64 && (long long)b * npoint < 0x7fffffffLL &&
        (long long)b * npoint * 32 < 0x7fffffffLL * 4)
        fast_np = c_last / 64;
    if (const char *fe = getenv("GEOT_SA_FAST")) fast_np = atoi(fe) ? fast_np : 0;
    // runs of 8 consecutive groups per wave (sector-sized output stores) once every wave still gets >= 2 runs
    int run_len = (fast_np && npoint % 8 == 0 && ((uintptr_t)out & 15) == 0 && nunits >= 16 * blocks * waves) ? 8 : 1;
    if (const char *re = getenv("GEOT_SA_RUN")) run_len = (atoi(re) == 8 && npoint % 8 == 0 && ((uintptr_t)out & 15) == 0) ? 8 : 1;
    if (wide)
        hipLaunchKernelGGL(sa_group_mlp_max_kernel<8>, dim3((unsigned)blocks), dim3(waves * 64), lds, (hipStream_t)stream,
                           d, b, n, npoint, nsample, c_feat, widths[nlayers - 1], xyz, new_xyz, features, idx, xyz_scale,
                           params, out, fast_np, run_len);
    else
        hipLaunchKernelGGL(sa_group_mlp_max_kernel<4>, dim3((unsigned)blocks), dim3(waves * 64), lds, (hipStream_t)stream,
                           d, b, n, npoint, nsample, c_feat, widths[nlayers - 1], xyz, new_xyz, features, idx, xyz_scale,
                           params, out, fast_np, run_len);
    return hipGetLastError();
}
