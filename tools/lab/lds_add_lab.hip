// Lab: cost of LDS float adds (ds_add_f32, no return) against plain LDS reads, per wave-instruction, under
// controlled address patterns: distinct words / runs of R lanes on the same word / random words.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lab/lds_add_lab.hip -o tools/lab/lds_add_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int WORDS = 8192, ITERS = 256, U = 8;

template <int MODE> // 0: ds_add_f32, 1: ds_read_b32 (gather), 2: plain read-modify-write (non-atomic)
__global__ __launch_bounds__(1024) void lab_kernel(const int *__restrict__ addr, float *__restrict__ out, long long *cycles)
{
    extern __shared__ float lds[];
    for (int e = threadIdx.x; e < WORDS; e += blockDim.x) lds[e] = 0.f;
    __syncthreads();
    int a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = addr[u * blockDim.x + threadIdx.x];
    float acc = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int w = (a[u] + it * 64) & (WORDS - 1);
            if (MODE == 0) unsafeAtomicAdd(&lds[w], 1.0f);
            else if (MODE == 1) acc += lds[w];
            else lds[w] = lds[w] + 1.0f;
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}

static void run(const char *name, int run_len, bool random_words, int threads)
{
    int *ha = (int *)malloc(sizeof(int) * U * threads);
    srand(3);
    for (int u = 0; u < U; ++u)
        for (int t = 0; t < threads; ++t) {
            int lane = t & 63, wave = t >> 6;
            int word = random_words ? rand() % WORDS : (wave * 512 + u * 64 + lane / run_len);   // runs of run_len lanes share a word
            ha[u * threads + t] = word;
        }
    int *addr; float *out; long long *cyc;
    const int blocks = 256;
    CHECK(hipMalloc(&addr, sizeof(int) * U * threads)); CHECK(hipMalloc(&out, sizeof(float) * blocks * threads));
    CHECK(hipMalloc(&cyc, sizeof(long long) * blocks));
    CHECK(hipMemcpy(addr, ha, sizeof(int) * U * threads, hipMemcpyHostToDevice));
    const char *modes[3] = {"ds_add_f32", "ds_read_b32", "plain rmw"};
    for (int mode = 0; mode < 3; ++mode) {
        void (*k)(const int *, float *, long long *) = mode == 0 ? lab_kernel<0> : (mode == 1 ? lab_kernel<1> : lab_kernel<2>);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), WORDS * 4, 0, addr, out, cyc);
            CHECK(hipDeviceSynchronize());
        }
        long long hc[256];
        CHECK(hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost));
        double avg = 0;
        for (int b = 0; b < blocks; ++b) avg += hc[b];
        avg /= blocks;
        const double instr = (double)ITERS * U * (threads / 64);   // wave-instructions per workgroup (= per CU)
        printf("%-34s %-12s threads=%4d: %7.1f cycles per wave-instruction (per CU)\n", name, modes[mode], threads, avg / instr);
    }
    fflush(stdout);
    free(ha); CHECK(hipFree(addr)); CHECK(hipFree(out)); CHECK(hipFree(cyc));
}

int main()
{
    run("distinct consecutive words", 1, false, 1024);
    run("runs of 2 lanes per word", 2, false, 1024);
    run("runs of 3 lanes per word", 3, false, 1024);
    run("runs of 4 lanes per word", 4, false, 1024);
    run("runs of 8 lanes per word", 8, false, 1024);
    run("random words", 1, true, 1024);
    run("distinct consecutive words", 1, false, 256);
    run("random words", 1, true, 256);
    return 0;
}
