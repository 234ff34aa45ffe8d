// Lab harness for csrc/tile_scatter.hip: the product source compiled with phase stamps (s_memtime accumulated by
// lane 0 of wave 0 and wave 7 of every workgroup), driven from a plain HIP main -- no torch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Igeot_amd/csrc tools/lab/ts_lab.hip -o tools/lab/ts_lab
//   tools/lab/ts_lab [b c L m nt iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#define TS_NSTAMP 8
#ifndef TS_NO_STAMPS
__device__ unsigned long long ts_stamp_buf[2][TS_NSTAMP];
#define TS_STAMP_DECL unsigned long long ts_t0 = __builtin_amdgcn_s_memtime(); unsigned long long ts_acc[TS_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TS_STAMP(slot)                                                         \
    {                                                                          \
        const unsigned long long ts_t1 = __builtin_amdgcn_s_memtime();         \
        ts_acc[slot] += ts_t1 - ts_t0;                                         \
        ts_t0 = ts_t1;                                                         \
    }
#define TS_STAMP_FLUSH                                                                                                  \
    if ((threadIdx.x == 0 || threadIdx.x == 7 * 64) && blockIdx.x < 64)                                                 \
        for (int i = 0; i < TS_NSTAMP; ++i) atomicAdd(&ts_stamp_buf[threadIdx.x ? 1 : 0][i], ts_acc[i]);
#else
__device__ unsigned long long ts_stamp_buf[2][TS_NSTAMP];
#endif
#include "../../geot_amd/csrc/tile_scatter.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    int b = argc > 1 ? atoi(argv[1]) : 8, c = argc > 2 ? atoi(argv[2]) : 1536, L = argc > 3 ? atoi(argv[3]) : 24000;
    int m = argc > 4 ? atoi(argv[4]) : 8192, nt = argc > 5 ? atoi(argv[5]) : 3, iters = argc > 6 ? atoi(argv[6]) : 10;
    const bool weighted = nt == 3;
    std::mt19937 rng(1);
    std::vector<int> idx((size_t)b * L * nt);
    std::vector<float> w((size_t)b * L * nt), g((size_t)b * c * L);
    for (auto &x : idx) x = rng() % m;
    for (auto &x : w) x = (rng() % 1000) / 1000.f;
    for (auto &x : g) x = (int)(rng() % 2001 - 1000) / 1000.f;
    int *d_idx; float *d_w, *d_g, *d_out; void *d_ws;
    const long long ws_ints = geot::ts_ws_ints(b, c, m, L, nt, weighted);
    if (!ws_ints) { printf("shape not taken by the tile path\n"); return 1; }
    CK(hipMalloc(&d_idx, idx.size() * 4)); CK(hipMalloc(&d_w, w.size() * 4)); CK(hipMalloc(&d_g, g.size() * 4));
    CK(hipMalloc(&d_out, (size_t)b * c * m * 4)); CK(hipMalloc(&d_ws, ws_ints * 4));
    CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, w.data(), w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_g, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    geot::TsPlan p;
    geot::ts_plan(b, c, m, L, nt, weighted, p);
    printf("plan: ch %d tl %d q %d ppp %d cap %d lds %zu build lds %zu\n", p.ch, p.tl, p.q, p.ppp, p.cap, p.lds, p.lds_build);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int it = 0; it < 3; ++it)
        CK(geot::scatter_via_tiles(b, c, m, L, nt, (size_t)c * L, d_g, d_idx, weighted ? d_w : nullptr, d_out, d_ws, ws_ints, 0, true));
    CK(hipDeviceSynchronize());
    unsigned long long zero[2][TS_NSTAMP] = {};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(ts_stamp_buf), zero, sizeof(zero)));
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; ++it)
        CK(geot::scatter_via_tiles(b, c, m, L, nt, (size_t)c * L, d_g, d_idx, weighted ? d_w : nullptr, d_out, d_ws, ws_ints, 0, true));
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 4.0 * b * c * ((double)L + m) + 8.0 * b * L * nt;
    printf("b %d c %d L %d m %d nt %d: %.1f us per call, %.2f TB/s algorithmic\n", b, c, L, m, nt, ms / iters * 1e3, bytes / (ms / iters * 1e-3) / 1e12);
#ifndef TS_NO_STAMPS
    unsigned long long st[2][TS_NSTAMP];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(ts_stamp_buf), sizeof(st)));
    const char *names[TS_NSTAMP] = {"prologue", "stage(store+load)", "barrier", "stage A (rows of p+1)", "stage B (sums of p)", "-", "-", "epilogue"};
    const int wgs = std::min(64, (c + p.ch - 1) / p.ch) * b;   // blockIdx.x < 64 of every batch
    for (int wv = 0; wv < 2; ++wv) {
        unsigned long long tot = 0;
        for (int i = 0; i < TS_NSTAMP; ++i) tot += st[wv][i];
        printf("wave %d: memtime ticks per workgroup (100 MHz), share:\n", wv ? 7 : 0);
        for (int i = 0; i < TS_NSTAMP; ++i)
            printf("   %-20s %10.1f  %5.1f %%\n", names[i], (double)st[wv][i] / iters / wgs, 100.0 * st[wv][i] / (double)tot);
        printf("   total %.1f ticks = %.1f us per workgroup\n", (double)tot / iters / wgs, (double)tot / iters / wgs / 100.0);
    }
#endif
    // check against a double scatter-add on the host (first batch, first 8 channels)
    std::vector<float> out((size_t)b * c * m);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int ch = 0; ch < std::min(c, 8); ++ch) {
        std::vector<double> ref(m, 0.0);
        for (int e = 0; e < L; ++e)
            for (int t = 0; t < nt; ++t)
                ref[idx[(size_t)e * nt + t]] += (double)(weighted ? w[(size_t)e * nt + t] : 1.f) * g[(size_t)ch * L + e];
        for (int j = 0; j < m; ++j) worst = std::max(worst, std::abs(ref[j] - out[(size_t)ch * m + j]));
    }
    printf("max abs error vs double (batch 0, 8 channels): %.3g\n", worst);
    return 0;
}
