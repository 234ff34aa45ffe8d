// Lab: what does ONE all-to-all candidate exchange between the workgroups of a split FPS round cost on MI355X?
//
// FPS is sequential: sample i+1 needs the arg-max after sample i.  Today one cloud = one workgroup = one CU and a
// round (3.8 committed samples on tooth clouds) takes ~2.1 us.  Splitting a cloud over W workgroups divides the
// per-round local work by ~W but adds one exchange per round: every workgroup publishes its candidate
// (value, key, xyz = 20 bytes) and needs everybody else's before it can go on.  This program measures that exchange
// alone -- W workgroups on the same XCD (blocks b, b+8, b+16, ... share one under the observed round-robin
// placement; checked with XCC_ID), `teams` independent teams running side by side (one per cloud), ROUNDS rounds,
// data-tagged 8-byte granules written with sc1 stores and polled with sc1 loads (MI355X_MICROARCH.md,
// handoff-1to1 / "Valid forms": one relaxed poll loop, no fences needed for a single self-tagged granule).
//
//   hipcc --offload-arch=gfx950 -O3 -o xcd_exchange tools/lab/xcd_exchange.hip && ./xcd_exchange
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// slots[team][member][parity][4] granules of {payload 32 bit, tag 32 bit}; tag = round + 1.  Two parities: a
// workgroup may be one round ahead of a partner that has not read its previous granule yet.
__global__ __launch_bounds__(256) void exchange_kernel(int W, int rounds, unsigned long long *slots, unsigned *xcc_out,
                                                       unsigned long long *sink)
{
    // grid = teams * W blocks laid out so that a team's members are 8 apart: block = member * (8 * lanes) ...
    // block b: team = (b % 8) + 8 * (b / (8 * W)), member = (b / 8) % W   -> members of a team share b % 8
    const int b = blockIdx.x;
    const int team = (b % 8) + 8 * (b / (8 * W)), member = (b / 8) % W;
    unsigned long long *mine = slots + ((size_t)team * W + member) * 8;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc_out[b] = xcc & 0xf;
    }
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; ++r) {
        // "local work": nothing -- the exchange alone.  Publish 3 granules (value+key, x+y, z) from lane 0..2
        if (threadIdx.x < 3) {
            const unsigned long long g = ((unsigned long long)(unsigned)(r + 1) << 32) | (unsigned)(b * 131 + r + threadIdx.x);
            __hip_atomic_store(mine + (r & 1) * 4 + threadIdx.x, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // every wave-0 lane < 3 * W polls one granule of one member (its own included: trivially ready)
        if (threadIdx.x < 3 * W) {
            const int m = threadIdx.x / 3, q = threadIdx.x % 3;
            const unsigned long long *src = slots + ((size_t)team * W + m) * 8 + (r & 1) * 4 + q;
            unsigned long long v;
            int spin = 0;       // bounded: a partner that is not resident must not hang the GPU (result is then garbage)
            do {
                v = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((unsigned)(v >> 32) != (unsigned)(r + 1) && ++spin < (1 << 16));
            acc += v & 0xffffffffu;
        }
        __syncthreads(); // the rest of the workgroup waits for wave 0 (as the real kernel's waves wait for the winner)
    }
    if (threadIdx.x < 3 * W) sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

int main()
{
    const int rounds = 4096;
    for (int W : {2, 4, 8}) {
        for (int teams : {8, 64}) {             // clouds per launch (8 = the bench's batch); teams % 8 == 0
            const int blocks = teams * W;
            unsigned long long *slots, *sink;
            unsigned *xcc;
            CHECK(hipMalloc(&slots, (size_t)teams * W * 8 * 8));
            CHECK(hipMemset(slots, 0, (size_t)teams * W * 8 * 8));
            CHECK(hipMalloc(&sink, (size_t)blocks * 64 * 8));
            CHECK(hipMalloc(&xcc, blocks * 4));
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0));
            CHECK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(slots, 0, (size_t)teams * W * 8 * 8));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(exchange_kernel, dim3(blocks), dim3(256), 0, 0, W, rounds, slots, xcc, sink);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            std::vector<unsigned> hx(blocks);
            CHECK(hipMemcpy(hx.data(), xcc, blocks * 4, hipMemcpyDeviceToHost));
            int same = 0;
            for (int t = 0; t < teams; ++t) {
                bool ok = true;
                const int base = (t % 8) + 8 * W * (t / 8);
                for (int m = 1; m < W; ++m) ok = ok && hx[base + 8 * m] == hx[base];
                same += ok;
            }
            printf("W=%d workgroups per cloud, %2d clouds: %.3f us per exchange round (%d rounds, %.2f ms); teams on one XCD: %d/%d\n",
                   W, teams, best * 1e3f / rounds, rounds, best, same, teams);
            fflush(stdout);
            CHECK(hipFree(slots)); CHECK(hipFree(sink)); CHECK(hipFree(xcc));
        }
    }
    return 0;
}
