// Lab: gradient of a gather as a scatter into LDS-resident OUTPUT rows with native LDS float atomics.
//   grad_table[b,c,j] += sum_{(e,t): idx[b,e,t]==j} w[b,e,t] * grad_out[b,c,e]
// A workgroup owns CH output rows (m floats each) in LDS, streams the L source elements (coalesced reads of
// grad_out rows and of idx / w), ds_add_f32 per (element, slot, channel), then writes the rows out once.
// Shapes: interpolation gradient (C=384 / 1536, L=24000, m=8192, NT=3) and SA grouping gradient (C=64, L=192000,
// m=24000, NT=1).   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lab/lds_scatter_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NT, bool WEIGHTED, int CH, int THREADS>
__global__ __launch_bounds__(THREADS) void lds_scatter_kernel(int c, int m, int L, const float *__restrict__ g,
                                                             const int *__restrict__ idx, const float *__restrict__ w,
                                                             float *__restrict__ out)
{
    extern __shared__ float rows[]; // [CH][m]
    const int bi = blockIdx.z, c0 = blockIdx.y * CH, nch = min(CH, c - c0);
    for (int e = threadIdx.x; e < CH * m; e += THREADS) rows[e] = 0.f;
    __syncthreads();
    const int per = (L + gridDim.x - 1) / gridDim.x;
    const int e0 = blockIdx.x * per, e1 = min(L, e0 + per);
    constexpr int U = 4;
    for (int eb = e0 + threadIdx.x; eb < e1; eb += U * THREADS) {
        int ii[U][NT];
        float ww[U][NT], gv[U][CH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = eb + u * THREADS;
            const bool ok = e < e1;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                ii[u][t] = ok ? idx[((size_t)bi * L + e) * NT + t] : 0;
                ww[u][t] = ok ? (WEIGHTED ? w[((size_t)bi * L + e) * NT + t] : 1.f) : 0.f;
            }
#pragma unroll
            for (int l = 0; l < CH; ++l) gv[u][l] = (ok && l < nch) ? g[((size_t)bi * c + c0 + l) * L + e] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (eb + u * THREADS < e1) {
#pragma unroll
                for (int l = 0; l < CH; ++l)
#pragma unroll
                    for (int t = 0; t < NT; ++t) unsafeAtomicAdd(&rows[l * m + ii[u][t]], ww[u][t] * gv[u][l]);
            }
        }
    }
    __syncthreads();
    for (int l = 0; l < nch; ++l)
        for (int j = threadIdx.x; j < m; j += THREADS) {
            float *dst = out + ((size_t)bi * c + c0 + l) * m + j;
            if (gridDim.x == 1) *dst += rows[l * m + j];
            else unsafeAtomicAdd(dst, rows[l * m + j]);
        }
}

template <int NT, bool WEIGHTED, int CH, int THREADS>
static void run(const char *name, int B, int C, int L, int m, int slices)
{
    std::vector<float> hg((size_t)B * C * L), hw((size_t)B * L * NT);
    std::vector<int> hi((size_t)B * L * NT);
    srand(1);
    for (auto &v : hg) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto &v : hw) v = (rand() % 1000) / 1000.f;
    // neighbour-like indices: element e points near e * m / L (as three_nn / ball query produce), +- 8
    for (int b = 0; b < B; ++b)
        for (int e = 0; e < L; ++e)
            for (int t = 0; t < NT; ++t) {
                long long j = (long long)e * m / L + (rand() % 17) - 8;
                hi[((size_t)b * L + e) * NT + t] = (int)((j % m + m) % m);
            }
    float *g, *w, *out;
    int *idx;
    CHECK(hipMalloc(&g, hg.size() * 4)); CHECK(hipMalloc(&w, hw.size() * 4)); CHECK(hipMalloc(&idx, hi.size() * 4));
    CHECK(hipMalloc(&out, (size_t)B * C * m * 4));
    CHECK(hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(idx, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    const size_t lds = (size_t)CH * m * 4;
    auto kern = lds_scatter_kernel<NT, WEIGHTED, CH, THREADS>;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid(slices, (C + CH - 1) / CH, B);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(out, 0, (size_t)B * C * m * 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, grid, dim3(THREADS), lds, 0, C, m, L, g, idx, w, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // check batch 0, channel 1 against the host
    std::vector<float> ho(m), ref(m, 0.f);
    CHECK(hipMemcpy(ho.data(), out + (size_t)(0 * C + 1) * m, m * 4, hipMemcpyDeviceToHost));
    for (int e = 0; e < L; ++e)
        for (int t = 0; t < NT; ++t)
            ref[hi[(size_t)e * NT + t]] += (WEIGHTED ? hw[(size_t)e * NT + t] : 1.f) * hg[(size_t)1 * L + e];
    double err = 0, mag = 0;
    for (int j = 0; j < m; ++j) { err = fmax(err, fabs(ho[j] - ref[j])); mag = fmax(mag, fabs(ref[j])); }
    const double bytes = 4.0 * B * ((double)C * L + (double)C * m) + (WEIGHTED ? 8.0 : 4.0) * B * L * NT;
    printf("%-44s CH=%d thr=%d slices=%d: %8.1f us  %6.2f TB/s algorithmic  (max err %.2e of %.2f)\n", name, CH, THREADS, slices,
           best * 1e3, bytes / (best * 1e-3) / 1e12, err, mag);
    fflush(stdout);
    CHECK(hipFree(g)); CHECK(hipFree(w)); CHECK(hipFree(idx)); CHECK(hipFree(out));
}

int main()
{
    run<3, true, 4, 1024>("interp grad C=384 L=24000 m=8192", 8, 384, 24000, 8192, 1);
    run<3, true, 2, 512>("interp grad C=384 L=24000 m=8192", 8, 384, 24000, 8192, 1);
    run<3, true, 2, 1024>("interp grad C=384 L=24000 m=8192", 8, 384, 24000, 8192, 1);
    run<3, true, 4, 1024>("interp grad C=1536 L=24000 m=8192", 8, 1536, 24000, 8192, 1);
    run<1, false, 1, 1024>("group grad C=64 L=192000 m=24000", 8, 64, 192000, 24000, 1);
    run<1, false, 1, 1024>("group grad C=64 L=192000 m=24000", 8, 64, 192000, 24000, 2);
    run<1, false, 1, 512>("group grad C=64 L=192000 m=24000", 8, 64, 192000, 24000, 4);
    return 0;
}
