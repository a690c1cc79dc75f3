// Measurement tool (not part of the library): what limits the pixel-operand stream of the 256-pixel tile kernels.
// One block of 512 threads per CU walks 256-pixel tiles of an NHWC slab; every K-step each thread requests two 16-byte
// chunks D steps ahead (register ring, hand-counted vmcnt), burns a fixed number of clocks standing in for the step's
// MFMA / prologue work, then consumes the chunks of the step after this one.  HALF: a step covers 32 channels (64 bytes
// of each of 256 rows, the wide-tile kernels); otherwise 64 channels of 128 rows (full 128-byte lines).
//   hipcc --offload-arch=gfx950 -O3 -o prefetch_depth_bench tools/prefetch_depth_bench.hip && ./prefetch_depth_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

template <int D, bool HALF>
__global__ __launch_bounds__(512, 2) void walk(const unsigned char* __restrict__ x, const int M, const int ld_bytes, const int ksteps,
                                                const int burn, uint32_t* __restrict__ out) {
    extern __shared__ unsigned char smem[];
    const int t = threadIdx.x;
    const int ntile = M / 256;
    u32x4 buf[D][2];
    uint32_t acc = 0;
    // request stream: (tile, step) pairs of this block in order
    int r_tile = blockIdx.x, r_k = 0;
    auto issue = [&](u32x4 (&b)[2]) {
        const int tl = r_tile < ntile ? r_tile : blockIdx.x;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint32_t off;
            if (HALF) off = (uint32_t)(tl * 256 + (t >> 2) + i * 128) * (uint32_t)ld_bytes + (uint32_t)r_k * 64u + (uint32_t)(t & 3) * 16u;
            else off = (uint32_t)(tl * 256 + (r_k & 1) * 128 + (t >> 3) + i * 64) * (uint32_t)ld_bytes + (uint32_t)(r_k >> 1) * 128u + (uint32_t)(t & 7) * 16u;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[i]) : "v"(off), "s"(x) : "memory");
        }
        if (++r_k == ksteps) { r_k = 0; r_tile += gridDim.x; }
    };
    auto consume = [&](u32x4 (&b)[2]) {
        asm volatile("" : "+v"(b[0]), "+v"(b[1]));
        acc ^= b[0].x ^ b[0].y ^ b[0].z ^ b[0].w ^ b[1].x ^ b[1].y ^ b[1].z ^ b[1].w;
    };
    auto tie_wait = [&]() {
        if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(2)" : "+v"(buf[0][0]), "+v"(buf[0][1]), "+v"(buf[1][0]), "+v"(buf[1][1]) : : "memory");
        if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(4)" : "+v"(buf[0][0]), "+v"(buf[0][1]), "+v"(buf[1][0]), "+v"(buf[1][1]), "+v"(buf[2][0]), "+v"(buf[2][1]) : : "memory");
        if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(6)" : "+v"(buf[0][0]), "+v"(buf[0][1]), "+v"(buf[1][0]), "+v"(buf[1][1]), "+v"(buf[2][0]), "+v"(buf[2][1]), "+v"(buf[3][0]), "+v"(buf[3][1]) : : "memory");
    };
    int total = 0;
    for (int tl = blockIdx.x; tl < ntile; tl += gridDim.x) total += ksteps;
#pragma unroll
    for (int d = 0; d < D; ++d) issue(buf[d]);
    for (int g = 0; g < total; g += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            // step g + j: its chunks are in buf[j]; everything younger (D - 1 steps' worth) may stay in flight
            const long t0 = __builtin_readcyclecounter();
            while (__builtin_readcyclecounter() - t0 < burn) __builtin_amdgcn_s_sleep(1);
            // every register of the ring is an operand of the wait: the compiler must neither read a landed buffer before
            // it nor move one that is still in flight
            tie_wait();
            consume(buf[j]);
            issue(buf[j]);
            __builtin_amdgcn_s_barrier();
        }
    }
    if constexpr (D == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf[0][0]), "+v"(buf[0][1]), "+v"(buf[1][0]), "+v"(buf[1][1]) : : "memory");
    if constexpr (D == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf[0][0]), "+v"(buf[0][1]), "+v"(buf[1][0]), "+v"(buf[1][1]), "+v"(buf[2][0]), "+v"(buf[2][1]) : : "memory");
    if constexpr (D == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(buf[0][0]), "+v"(buf[0][1]), "+v"(buf[1][0]), "+v"(buf[1][1]), "+v"(buf[2][0]), "+v"(buf[2][1]), "+v"(buf[3][0]), "+v"(buf[3][1]) : : "memory");
#pragma unroll
    for (int d = 0; d < D; ++d) consume(buf[d]);
    if (acc == 0x12345678u) out[blockIdx.x * 512 + t] = acc + smem[t];
}

template <int D, bool HALF>
void run(const unsigned char* x, int M, int C, int cin, int burn, uint32_t* out, hipEvent_t a, hipEvent_t b) {
    const int ksteps = HALF ? cin / 32 : cin / 32;       // same number of steps and bytes per step either way (16 KiB)
    CHECK(hipFuncSetAttribute((const void*)walk<D, HALF>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((walk<D, HALF>), dim3(256), dim3(512), 100 * 1024, 0, x, M, C * 2, ksteps, burn, out);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double moved = (double)M * cin * 2;
    const double steps = (double)(M / 256 / 256) * ksteps;
    printf("%-5s depth %d burn %5d clk: %8.1f us  %.2f TB/s  %.0f ns per step\n", HALF ? "64B" : "128B", D, burn, best * 1e3, moved / best * 1e-9, best * 1e6 / steps);
}

int main() {
    const int M = 524288, C = 2112, cin = 1248;
    unsigned char* x;
    uint32_t* out;
    CHECK(hipMalloc(&x, (size_t)M * C * 2));
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    CHECK(hipMemset(x, 1, (size_t)M * C * 2));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int burns[] = {0, 1000, 2000, 2800};
    for (int bi = 0; bi < 4; ++bi) {
        const int burn = burns[bi];
        run<2, true>(x, M, C, cin, burn, out, a, b);
        run<3, true>(x, M, C, cin, burn, out, a, b);
        run<4, true>(x, M, C, cin, burn, out, a, b);
        run<2, false>(x, M, C, cin, burn, out, a, b);
        run<3, false>(x, M, C, cin, burn, out, a, b);
        run<4, false>(x, M, C, cin, burn, out, a, b);
    }
    return 0;
}
