// Poly-1 focal loss on channels-first logits (B, C, N) with integer labels (B, N), forward + backward as streaming
// kernels (the reference composes ~15 element-wise torch ops each way on one-hot tensors:
// openpoints/loss/build.py:183-258 Poly1FocalLoss, :799-892 Poly1FocalLoss_U_corr).
//   y = [label == c],  p = sigmoid(x),  ce = BCE-with-logits(x, y),  pt = y p + (1 - y)(1 - p),  q = 1 - pt
//   l = at * ce * q^gamma + eps * q^(gamma + 1),   at = alpha y + (1 - alpha)(1 - y) if alpha >= 0 else 1
// mean form:   loss = sum l / (B C N);   masked form:  loss = sum l * keep[b, n] / (C * sum keep + 0.001).
#include "geot_common.h"
#include "geot_hip.h"

namespace geot {

constexpr int PL_THREADS = 256;

struct Poly1 {
    float alpha, gamma, eps;
    __device__ __forceinline__ float powg(float q, float g) const { return g == 2.f ? q * q : (g == 1.f ? q : powf(q, g)); }
    // value and d/dx for one logit
    __device__ __forceinline__ void eval(float x, bool pos, float &l, float &dl) const
    {
        const float p = 1.f / (1.f + expf(-x));
        const float ce = fmaxf(x, 0.f) - (pos ? x : 0.f) + log1pf(expf(-fabsf(x)));   // torch's stable form
        const float q = pos ? 1.f - p : p;                                               // 1 - pt
        const float at = alpha >= 0.f ? (pos ? alpha : 1.f - alpha) : 1.f;
        const float qg = powg(q, gamma);
        l = at * ce * qg + eps * qg * q;
        const float dq = (pos ? -1.f : 1.f) * p * (1.f - p);                             // d q / d x
        const float qg1 = gamma == 2.f ? q : (q > 0.f ? qg / q : 0.f);                   // q^(gamma - 1)
        dl = at * ((p - (pos ? 1.f : 0.f)) * qg + ce * gamma * qg1 * dq) + eps * (gamma + 1.f) * qg * dq;
    }
};

// grid (blocks over n, C, B): partial[block] = (sum l * keep, sum keep) in fp64 (keep counted once per point: c == 0)
__global__ __launch_bounds__(PL_THREADS) void poly1_fwd_kernel(int c, int n, Poly1 P, const float *__restrict__ logits,
                                                               const long long *__restrict__ labels,
                                                               const unsigned char *__restrict__ keep, double *__restrict__ partial)
{
    const int bi = blockIdx.z, cc = blockIdx.y;
    const float *row = logits + ((size_t)bi * c + cc) * n;
    const long long *lab = labels + (size_t)bi * n;
    const unsigned char *kp = keep ? keep + (size_t)bi * n : nullptr;
    double s = 0.0, k = 0.0;
    for (int i = blockIdx.x * PL_THREADS + threadIdx.x; i < n; i += gridDim.x * PL_THREADS) {
        float l, dl;
        P.eval(row[i], lab[i] == cc, l, dl);
        const float w = kp ? (kp[i] ? 1.f : 0.f) : 1.f;
        s += (double)(l * w);
        if (cc == 0) {
            k += (double)w;
            // the reference's F.one_hot raises on a label outside [0, C) (an ignore index of -1 / 255, say); a kernel
            // cannot raise, and treating the point as all-negative would be a silently different loss: poison it
            if (lab[i] < 0 || lab[i] >= c) s += (double)__int_as_float(0x7fc00000);
        }
    }
    __shared__ double sh[2][PL_THREADS / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); k += __shfl_xor(k, o); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = k; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < PL_THREADS / 64; ++w) { a += sh[0][w]; b += sh[1][w]; }
        double *dst = partial + 2 * (((size_t)bi * c + cc) * gridDim.x + blockIdx.x);
        dst[0] = a;
        dst[1] = b;
    }
}

// one block: out[0] = loss, out[1] = 1 / denominator (kept for the backward)
__global__ __launch_bounds__(PL_THREADS) void poly1_finish_kernel(int nparts, int c, int masked, double count,
                                                                  const double *__restrict__ partial, float *__restrict__ out)
{
    double s = 0.0, k = 0.0;
    for (int i = threadIdx.x; i < nparts; i += PL_THREADS) { s += partial[2 * i]; k += partial[2 * i + 1]; }
    __shared__ double sh[2][PL_THREADS / 64];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); k += __shfl_xor(k, o); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = k; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < PL_THREADS / 64; ++w) { a += sh[0][w]; b += sh[1][w]; }
        const double den = masked ? b * c + 0.001 : count;
        out[0] = (float)(a / den);
        out[1] = (float)(1.0 / den);
    }
}

// grad_logits = upstream * inv_den * keep * dl/dx
__global__ __launch_bounds__(PL_THREADS) void poly1_bwd_kernel(int c, int n, Poly1 P, const float *__restrict__ logits,
                                                               const long long *__restrict__ labels,
                                                               const unsigned char *__restrict__ keep,
                                                               const float *__restrict__ fin, const float *__restrict__ upstream,
                                                               float *__restrict__ grad)
{
    const int bi = blockIdx.z, cc = blockIdx.y;
    const size_t base = ((size_t)bi * c + cc) * n;
    const long long *lab = labels + (size_t)bi * n;
    const unsigned char *kp = keep ? keep + (size_t)bi * n : nullptr;
    const float g = upstream[0] * fin[1];
    for (int i = blockIdx.x * PL_THREADS + threadIdx.x; i < n; i += gridDim.x * PL_THREADS) {
        float l, dl;
        P.eval(logits[base + i], lab[i] == cc, l, dl);
        grad[base + i] = (kp && !kp[i]) ? 0.f : g * dl;
    }
}

static int pl_gx(int n)
{
    int gx = (n + PL_THREADS * 4 - 1) / (PL_THREADS * 4);
    return gx < 1 ? 1 : (gx > 32 ? 32 : gx);
}

} // namespace geot

using namespace geot;

GEOT_EXPORT long long geot_poly1_focal_ws_doubles(int b, int c, int n)
{
    if (b < 1 || c < 1 || n < 1) return -1;
    return 2LL * b * c * pl_gx(n);
}

GEOT_EXPORT int geot_poly1_focal(int b, int c, int n, float alpha, float gamma, float epsilon, const float *logits,
                                 const long long *labels, const unsigned char *keep, double *workspace, float *out2,
                                 void *stream)
{
    if (b < 1 || c < 1 || n < 1 || b > 65535 || c > 65535 || !logits || !labels || !workspace || !out2) return hipErrorInvalidValue;
    const int gx = pl_gx(n);
    const Poly1 P{alpha, gamma, epsilon};
    hipLaunchKernelGGL(poly1_fwd_kernel, dim3(gx, c, b), dim3(PL_THREADS), 0, (hipStream_t)stream, c, n, P, logits, labels, keep,
                       workspace);
    hipLaunchKernelGGL(poly1_finish_kernel, dim3(1), dim3(PL_THREADS), 0, (hipStream_t)stream, b * c * gx, c, keep ? 1 : 0,
                       (double)b * c * n, workspace, out2);
    return hipGetLastError();
}

GEOT_EXPORT int geot_poly1_focal_grad(int b, int c, int n, float alpha, float gamma, float epsilon, const float *logits,
                                      const long long *labels, const unsigned char *keep, const float *out2,
                                      const float *upstream, float *grad_logits, void *stream)
{
    if (b < 1 || c < 1 || n < 1 || b > 65535 || c > 65535 || !logits || !labels || !out2 || !upstream || !grad_logits)
        return hipErrorInvalidValue;
    const Poly1 P{alpha, gamma, epsilon};
    hipLaunchKernelGGL(poly1_bwd_kernel, dim3(pl_gx(n), c, b), dim3(PL_THREADS), 0, (hipStream_t)stream, c, n, P, logits, labels,
                       keep, out2, upstream, grad_logits);
    return hipGetLastError();
}
