// Measurement tool (not part of the library): HBM read rate of the activation-operand access pattern of the
// 1x1 implicit-GEMM kernel -- a block walks a 128-pixel strip of an NHWC slab 64 channels (128 bytes per pixel
// row) at a time -- against the same bytes read from a channel-blocked layout [C/64][M][64], with an optional
// delay per K-step standing in for the MFMA work of the real kernel.
//   hipcc --offload-arch=gfx950 -O3 -o stream_pattern tools/stream_pattern.hip && ./stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool BLOCKED>
__global__ __launch_bounds__(512) void walk(const uint4* __restrict__ x, const long M, const int ld_chunks, const int ksteps,
                                            const int delay, uint4* __restrict__ out) {
    const int t = threadIdx.x;
    const int chunk = t & 7, r0 = t >> 3;            // 64 rows per pass, 2 passes = 128 rows
    uint4 acc = make_uint4(0, 0, 0, 0);
    const long strips = M / 128;
    for (long strip = blockIdx.x; strip < strips; strip += gridDim.x) {
        const long row0 = strip * 128;
        for (int k = 0; k < ksteps; ++k) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const long row = row0 + r0 + p * 64;
                const long idx = BLOCKED ? ((long)k * M + row) * 8 + chunk : row * ld_chunks + (long)k * 8 + chunk;
                const uint4 v = x[idx];
                acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
            }
            for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(32);
        }
    }
    if (acc.x == 0x12345678u) out[blockIdx.x * 512 + t] = acc;      // keep the loads alive
}

int main(int argc, char** argv) {
    const long M = 524288;
    const int C = 2208, ksteps = 24;                  // 1536 of 2208 channels, like a mid-block dense layer
    const size_t bytes = (size_t)M * C * 2;
    uint4 *x, *out;
    CHECK(hipMalloc(&x, bytes));
    CHECK(hipMalloc(&out, 4096 * 512 * 16));
    CHECK(hipMemset(x, 1, bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const double moved = (double)M * ksteps * 128;
    const int grids[] = {512, 1024, 4096};
    const int delays[] = {0, 4, 16};
    for (int gi = 0; gi < 3; ++gi)
        for (int di = 0; di < 3; ++di)
            for (int blocked = 0; blocked < 2; ++blocked) {
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    CHECK(hipEventRecord(a));
                    if (blocked) hipLaunchKernelGGL(walk<true>, dim3(grids[gi]), dim3(512), 0, 0, x, M, C / 8, ksteps, delays[di], out);
                    else hipLaunchKernelGGL(walk<false>, dim3(grids[gi]), dim3(512), 0, 0, x, M, C / 8, ksteps, delays[di], out);
                    CHECK(hipEventRecord(b));
                    CHECK(hipEventSynchronize(b));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                printf("grid %4d delay %2d %-8s %.3f ms  %.2f TB/s\n", grids[gi], delays[di], blocked ? "blocked" : "nhwc", best, moved / best * 1e-9);
            }
    return 0;
}
