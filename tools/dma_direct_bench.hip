// Measurement tool (not part of the library): the K-step of a DMA-fed GEMM kernel when one operand bypasses LDS.
// 8 waves; per step the pixel half of a 32 KiB stage (16 KiB) arrives by LDS-DMA (2 global_load_lds_dwordx4 per wave) and
// each wave loads its own filter fragments global -> registers (NDIR global_load_dwordx4 per wave, 16 rows x 64 bytes
// each), two steps ahead, with one counted wait per step; NRD fragment reads from LDS and NMF MFMAs per wave and step.
// Compare with tools/dma_fill_bench.hip (both halves by DMA, 12 reads).
//   hipcc --offload-arch=gfx950 -O3 -o dma_direct_bench tools/dma_direct_bench.hip && ./dma_direct_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int STAGE = 16 * 1024;      // the DMA-fed half
constexpr int DEPTH = 2, NS = 3;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

template <int NRD, int NMF, int NDIR>
__global__ __launch_bounds__(512, 2) void kstep_bench(const unsigned char* __restrict__ src, const size_t window, const int steps3, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ring_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)smem);
    size_t cur = (size_t)blockIdx.x * 3 * 32768;
    u32x4 dir[NS][NDIR > 0 ? NDIR : 1];
    f32x4 acc[4] = {};
    // one step's requests into ring slot SLOT: the DMA half, then this wave's direct fragment loads
    auto issue = [&](auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            size_t off = (cur + (size_t)(i * 8 + wave) * 1024 + lane * 16) % window;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         : : "s"(ring_l + SLOT * STAGE + (i * 8 + wave) * 1024), "v"(src + off) : "memory");
        }
#pragma unroll
        for (int d = 0; d < NDIR; ++d) {
            // 16 rows of a 4224-byte pitch, 64 bytes of each: the filter rows of this wave's fragment d
            size_t off = (cur + 16384 + (size_t)((wave >> 1) * NDIR + d) * 16 * 4224 + (size_t)(lane & 15) * 4224 + (lane >> 4) * 16) % window;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dir[SLOT][d]) : "v"(src + off) : "memory");
        }
        cur += 32768;
    };
    auto step = [&](auto slot_c, auto next_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        wait_vm<(DEPTH - 1) * (2 + NDIR)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(next_c);
        // the direct loads of this step have landed: from here on their registers are ordinary values
#pragma unroll
        for (int d = 0; d < NDIR; ++d) asm volatile("" : "+v"(dir[SLOT][d]));
        const unsigned char* st = smem + SLOT * STAGE;
        bf16x8 fr[NRD > 0 ? NRD : 1];
#pragma unroll
        for (int r = 0; r < NRD; ++r) fr[r] = *reinterpret_cast<const bf16x8*>(st + ((r * 16 + (lane & 15)) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) << 4)) % STAGE);
#pragma unroll
        for (int m = 0; m < NMF; ++m) {
            const bf16x8 a = NDIR > 0 ? __builtin_bit_cast(bf16x8, dir[SLOT][m % (NDIR > 0 ? NDIR : 1)]) : fr[(m + 1) % (NRD > 0 ? NRD : 1)];
            acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, fr[NRD > 0 ? m % NRD : 0], acc[m & 3], 0, 0, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    issue(I0{});
    issue(I1{});
    for (int s = 0; s < steps3; ++s) {
        step(I0{}, I2{});
        step(I1{}, I0{});
        step(I2{}, I1{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // nothing of this block may land in LDS after it has gone
    float t = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
#pragma unroll
    for (int d = 0; d < NDIR; ++d) t += __builtin_bit_cast(f32x4, dir[0][d])[0] + __builtin_bit_cast(f32x4, dir[1][d])[0];
    if (t == 12345.678f) out[blockIdx.x * 512 + tid] = t;
}

template <int NRD, int NMF, int NDIR>
void run(const char* name, const unsigned char* src, size_t window, int steps3, float* out, int blocks) {
    auto kern = kstep_bench<NRD, NMF, NDIR>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = (size_t)NS * STAGE + 65536;      // one block per CU, as in the kernels
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, src, window, steps3, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, src, window, steps3, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 3, steps = 3.0 * steps3;
    printf("%-44s reads %2d mfma %2d direct %d: %8.1f us  %6.0f clk per step (2.4 GHz)  %6.0f TFLOP/s of MFMA\n", name, NRD, NMF, NDIR, us,
           us * 1e-6 * 2.4e9 / steps, (double)blocks * steps * 8 * NMF * 16384.0 / us / 1e6);
}

int main() {
    const int blocks = 256, steps3 = 700;
    const size_t win = (size_t)2 << 20;
    unsigned char* src;
    float* out;
    CHECK(hipMalloc(&src, win + 4096));
    CHECK(hipMemset(src, 0x3c, win + 4096));
    CHECK(hipMalloc(&out, (size_t)blocks * 512 * 4));
    // pixel half by DMA only (what the LDS side costs without the filter half)
    run<8, 32, 0>("16 KiB DMA, 8 reads (no filter operand)", src, win, steps3, out, blocks);
    // 256 x 256 x 32 step: 128 x 64 wave tiles, 8 pixel fragments from LDS, 4 filter fragments direct
    run<8, 32, 4>("16 KiB DMA + 4 direct fragments", src, win, steps3, out, blocks);
    // 128 x 128 x 64 step of the grouped kernels: 64 x 32 wave tiles, 8 pixel fragment reads, 4 filter fragments direct
    run<8, 16, 4>("16 KiB DMA + 4 direct fragments", src, win, steps3, out, blocks);
    run<8, 0, 4>("16 KiB DMA + 4 direct fragments", src, win, steps3, out, blocks);
    return 0;
}
