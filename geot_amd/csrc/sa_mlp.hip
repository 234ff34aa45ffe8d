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

template <int NCT>
__device__ __forceinline__ void sa_run_layer(const SaDesc &d, int l, const float *__restrict__ P,
                                             float *__restrict__ act, float *__restrict__ pool, int gpt)
{
    f32x16 acc[NCT];
    sa_layer<NCT>(P + d.woff[l], P + d.boff[l], d.kp[l], d.cp[l], (d.relu_mask >> l) & 1, act, d.act_stride, acc);
    if (l + 1 < d.nlayers) sa_store_act<NCT>(acc, act, d.act_stride);
    else sa_pool<NCT>(acc, pool, d.cp[l], gpt);
}

// MAXW = widest layer / 32 the variant supports: the 8-tile (256-wide) accumulators cost 128 VGPRs, which caps
// the occupancy at 2 waves per SIMD; networks up to 128 wide use the lean variant and run 3.
template <int MAXW>
__global__ __launch_bounds__(MAXW >= 8 ? 512 : SA_WAVES * 64) void sa_group_mlp_max_kernel(
    SaDesc d, int b, int n, int npoint, int nsample, int c_feat, int c_out,
    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
    const float *__restrict__ features, const int *__restrict__ idx, float xyz_scale,
    const float *__restrict__ params, float *__restrict__ out)
{
    extern __shared__ float sa_lds[];
    float *P = sa_lds;
    const int wave = threadIdx.x >> 6, lane = lane_id();
    const int cp_last = d.cp[d.nlayers - 1];
    const int gpt = nsample >= 32 ? 1 : 32 / nsample;  // groups per 32-row tile
    float *act = sa_lds + d.total + wave * (32 * d.act_stride + gpt * cp_last);
    float *pool = act + 32 * d.act_stride;
    const int nwaves = blockDim.x >> 6;
    for (int i = threadIdx.x; i < d.total; i += blockDim.x) P[i] = params[i];
    __syncthreads();

    const int tpg = nsample >= 32 ? nsample / 32 : 1;  // tiles per group
    const long long ngroups = (long long)b * npoint;
    const long long nunits = (ngroups + gpt - 1) / gpt;
    const int k_in = 3 + c_feat;
    for (long long u = (long long)blockIdx.x * nwaves + wave; u < nunits; u += (long long)gridDim.x * nwaves) {
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
                if (w32 == 1) sa_run_layer<1>(d, l, P, act, pool, gpt);
                else if (w32 == 2) sa_run_layer<2>(d, l, P, act, pool, gpt);
                else if (MAXW >= 8 && w32 == 8) sa_run_layer<(MAXW >= 8 ? 8 : 4)>(d, l, P, act, pool, gpt);
                else sa_run_layer<4>(d, l, P, act, pool, gpt);
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
    {   // > 64 KB of dynamic LDS is opt-in, per device and per kernel: set it on every call (cheap, and correct
        // when a process drives more than one GPU)
        hipError_t e = wide ? hipFuncSetAttribute((const void *)sa_group_mlp_max_kernel<8>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                            : hipFuncSetAttribute((const void *)sa_group_mlp_max_kernel<4>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    long long nunits = ((long long)b * npoint + gpt - 1) / gpt;
    long long blocks = (nunits + waves - 1) / waves;
    // persistent workgroups: one per CU (the weights + activation tiles fill its LDS), each looping over its
    // share of the tiles, so the 52 KB of weights are staged once per CU and not once per 8 tiles
    int dev = 0, n_cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cus < 1)
        n_cus = 256;
    if (blocks > n_cus) blocks = n_cus;
    if (wide)
        hipLaunchKernelGGL(sa_group_mlp_max_kernel<8>, dim3((unsigned)blocks), dim3(waves * 64), lds, (hipStream_t)stream,
                           d, b, n, npoint, nsample, c_feat, widths[nlayers - 1], xyz, new_xyz, features, idx, xyz_scale,
                           params, out);
    else
        hipLaunchKernelGGL(sa_group_mlp_max_kernel<4>, dim3((unsigned)blocks), dim3(waves * 64), lds, (hipStream_t)stream,
                           d, b, n, npoint, nsample, c_feat, widths[nlayers - 1], xyz, new_xyz, features, idx, xyz_scale,
                           params, out);
    return hipGetLastError();
}
