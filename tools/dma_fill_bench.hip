// Measurement tool (not part of the library): how fast can a CU fill LDS by DMA (global_load_lds_dwordx4) in the
// shape the DMA-fed GEMM kernels use -- 8 waves, a ring of 32 KiB stages, 4 DMA instructions per wave and stage,
// one counted wait and one barrier per step -- from an L2-resident window or from an HBM stream, alone and beside
// the LDS fragment reads and MFMAs of a K-step.
//   hipcc --offload-arch=gfx950 -O3 -o dma_fill_bench tools/dma_fill_bench.hip && ./dma_fill_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#ifndef DMA_MOD
#define DMA_MOD ""        // cache-policy modifiers of the DMA instruction for experiments: -DDMA_MOD='" sc0"', '" sc1"', '" nt"', '" sc0 sc1"'
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int STAGE = 32 * 1024;

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// DEPTH stages in flight, ring of DEPTH + 1 slots.  row_stride: bytes between the 256 rows of a stage (128 = the stage
// is one contiguous 32 KiB block); a stage advances step_stride bytes per step for `ksteps` steps, then the next 256
// rows follow.  window: the source wraps at this many bytes (small: L2 resident; 0: no wrap, HBM stream).
// SEG: bytes a stage takes from each row (128: 8 rows per DMA instruction, 64: 16 rows of half a cache line each)
template <int DEPTH, bool BAR, int NRD, int NMF, int SEG = 128>
__global__ __launch_bounds__(512, 2) void fill(const unsigned char* __restrict__ src, const size_t window, const size_t block_span,
                                               const int row_stride, const int step_stride, const int ksteps, const int steps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ring_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)smem);
    const size_t base = (size_t)blockIdx.x * block_span;
    int c_step = 0, c_slot = 0;
    auto issue = [&]() {
        const int tile = c_step / ksteps, kk = c_step - tile * ksteps;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            constexpr int LPR = SEG / 16, RPI = 64 / LPR;              // lanes per row, rows per instruction
            const int row = (i * 8 + wave) * RPI + lane / LPR;
            size_t off = base + ((size_t)tile * (32 * RPI) + row) * row_stride + (size_t)kk * step_stride + (lane % LPR) * 16;
            if (window) off %= window;
            const unsigned char* p = src + off;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" DMA_MOD
                         : : "s"(ring_l + c_slot * STAGE + (i * 8 + wave) * 1024), "v"(p) : "memory");
        }
        ++c_step;
        c_slot = c_slot == DEPTH ? 0 : c_slot + 1;
    };
    for (int i = 0; i < DEPTH; ++i) issue();
    f32x4 acc[4] = {};
    int r_slot = 0;
    for (int s = 0; s < steps; ++s) {
        wait_vm<4 * (DEPTH - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (BAR) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        const unsigned char* st = smem + r_slot * STAGE;
        bf16x8 fr[NRD > 0 ? NRD : 1];
#pragma unroll
        for (int r = 0; r < NRD; ++r) fr[r] = *reinterpret_cast<const bf16x8*>(st + ((r * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4)) % STAGE);
#pragma unroll
        for (int m = 0; m < NMF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[NRD > 0 ? m % NRD : 0], fr[NRD > 0 ? (m + 1) % NRD : 0], acc[m & 3], 0, 0, 0);
        if (NMF == 0 && NRD > 0) {
#pragma unroll
            for (int r = 0; r < NRD; ++r) acc[r & 3][0] += __builtin_bit_cast(f32x4, fr[r])[0];
        }
        r_slot = r_slot == DEPTH ? 0 : r_slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // nothing of this block may land in LDS after it has gone
    const float t = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
    if (t == 12345.678f) out[blockIdx.x * 512 + tid] = t;
}

// The same ring with the K-step software-pipelined across the barrier: the step's barrier sits between the two halves of
// its MFMAs, and the fragment reads of the NEXT stage are issued right behind it, so that the barrier skew and the LDS
// latency hide under the second half (fragments double-buffered in registers).
template <int DEPTH, int NRD, int NMF>
__global__ __launch_bounds__(512, 2) void fill_pipe(const unsigned char* __restrict__ src, const size_t window, const size_t block_span,
                                                    const int steps, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ring_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)smem);
    const size_t base = (size_t)blockIdx.x * block_span;
    int c_step = 0, c_slot = 0;
    auto issue = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            size_t off = (base + (size_t)c_step * STAGE + (size_t)(i * 8 + wave) * 1024 + lane * 16) % window;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                         : : "s"(ring_l + c_slot * STAGE + (i * 8 + wave) * 1024), "v"(src + off) : "memory");
        }
        ++c_step;
        c_slot = c_slot == DEPTH ? 0 : c_slot + 1;
    };
    auto read_frags = [&](int slot, bf16x8 (&fr)[NRD]) {
        const unsigned char* st = smem + slot * STAGE;
#pragma unroll
        for (int r = 0; r < NRD; ++r) fr[r] = *reinterpret_cast<const bf16x8*>(st + ((r * 16 + (lane & 15)) * 128 + (((lane >> 4) ^ ((lane >> 1) & 7)) << 4)) % STAGE);
    };
    for (int i = 0; i < DEPTH; ++i) issue();
    f32x4 acc[4] = {};
    bf16x8 fa[NRD], fb[NRD];
    // stage 0: plain start
    wait_vm<4 * (DEPTH - 1)>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue();
    read_frags(0, fa);
    int r_slot = 1;
    auto half = [&](const bf16x8 (&fr)[NRD], int h) {
#pragma unroll
        for (int m = 0; m < NMF / 2; ++m) {
            const int mm = h * (NMF / 2) + m;
            acc[mm & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[mm % NRD], fr[(mm + 1) % NRD], acc[mm & 3], 0, 0, 0);
        }
    };
    auto step = [&](bf16x8 (&cur)[NRD], bf16x8 (&nxt)[NRD]) {
        half(cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        wait_vm<4 * (DEPTH - 1)>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue();
        read_frags(r_slot, nxt);
        r_slot = r_slot == DEPTH ? 0 : r_slot + 1;
        __builtin_amdgcn_sched_barrier(0);
        half(cur, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < steps / 2; ++s) {
        step(fa, fb);
        step(fb, fa);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const float t = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
    if (t == 12345.678f) out[blockIdx.x * 512 + tid] = t;
}

template <int DEPTH, int NRD, int NMF>
void run_pipe(const char* name, const unsigned char* src, size_t window, size_t block_span, int steps, float* out, int blocks) {
    auto kern = fill_pipe<DEPTH, NRD, NMF>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = (size_t)(DEPTH + 1) * STAGE;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, src, window, block_span, steps, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, src, window, block_span, steps, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 3, bytes = (double)blocks * steps * STAGE;
    printf("%-46s depth %d pipelined reads %2d mfma %2d: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU  %6.0f TFLOP/s of MFMA\n", name, DEPTH, NRD, NMF, us,
           bytes / us / 1e6, bytes / blocks / (us * 1e-6 * 2.4e9), (double)blocks * steps * 8 * NMF * 16384.0 / us / 1e6);
}

template <int DEPTH, bool BAR, int NRD, int NMF, int SEG = 128>
void run(const char* name, const unsigned char* src, size_t window, size_t block_span, int row_stride, int step_stride, int ksteps, int steps, float* out, int blocks) {
    auto kern = fill<DEPTH, BAR, NRD, NMF, SEG>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t lds = (size_t)(DEPTH + 1) * STAGE;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, src, window, block_span, row_stride, step_stride, ksteps, steps, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, src, window, block_span, row_stride, step_stride, ksteps, steps, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / 3, bytes = (double)blocks * steps * STAGE;
    printf("%-46s depth %d bar %d reads %2d mfma %2d: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, %d CUs)  %6.0f TFLOP/s of MFMA\n", name, DEPTH, (int)BAR, NRD,
           NMF, us, bytes / us / 1e6, bytes / blocks / (us * 1e-6 * 2.4e9), blocks, (double)blocks * steps * 8 * NMF * 16384.0 / us / 1e6);
}

int main() {
    const int blocks = 256, steps = 2048;
    const size_t big = (size_t)6 << 30;
    unsigned char* src;
    float* out;
    CHECK(hipMalloc(&src, big));
    CHECK(hipMemset(src, 0x3c, big));
    CHECK(hipMalloc(&out, (size_t)blocks * 512 * 4));
    const size_t span_c = (size_t)steps * STAGE;                       // contiguous stream: each block its own 64 MiB
    // ---- L2-resident window (2 MiB shared by all blocks), contiguous stages
#define L2C(D, B, R, M) run<D, B, R, M>("L2 window 2 MiB, contiguous stages", src, (size_t)2 << 20, (size_t)STAGE * 3, 128, STAGE, 1 << 30, steps, out, blocks)
    L2C(3, true, 0, 0);
    L2C(3, false, 0, 0);
    L2C(2, true, 0, 0);
    L2C(3, true, 12, 0);
    L2C(3, true, 12, 16);
    L2C(3, false, 12, 16);
    // the same stage and the same twelve fragment reads feeding twice the MFMAs: a 256 x 256 x 32 step (128 x 64 wave tiles)
    L2C(3, true, 12, 32);
    L2C(3, true, 12, 48);
    // software-pipelined across the barrier
    run_pipe<3, 12, 16>("L2 window 2 MiB, contiguous stages", src, (size_t)2 << 20, (size_t)STAGE * 3, steps, out, blocks);
    run_pipe<3, 12, 32>("L2 window 2 MiB, contiguous stages", src, (size_t)2 << 20, (size_t)STAGE * 3, steps, out, blocks);
    run_pipe<2, 12, 32>("L2 window 2 MiB, contiguous stages", src, (size_t)2 << 20, (size_t)STAGE * 3, steps, out, blocks);
    run_pipe<3, 12, 48>("L2 window 2 MiB, contiguous stages", src, (size_t)2 << 20, (size_t)STAGE * 3, steps, out, blocks);
    // ---- L2-resident, rows of a 4224-byte pitch (a slab), 16 K-steps per 256-row tile
#define L2S(D, B, R, M) run<D, B, R, M>("L2 window 2 MiB, 4224-byte row pitch", src, (size_t)2 << 20, (size_t)256 * 4224, 4224, 128, 16, steps, out, blocks)
    L2S(3, true, 0, 0);
    L2S(3, true, 12, 16);
    L2S(3, true, 12, 32);
    // ---- the same with 64 bytes per row and step (a 512-row stage of a 256 x 256 x 32 step): half cache lines
#define L2H(D, B, R, M) run<D, B, R, M, 64>("L2 window 2 MiB, 4224-byte pitch, 64 B per row", src, (size_t)2 << 20, (size_t)512 * 4224, 4224, 64, 32, steps, out, blocks)
    L2H(3, true, 0, 0);
    L2H(3, true, 12, 32);
    // ---- HBM stream, contiguous
#define HBC(D, B, R, M) run<D, B, R, M>("HBM stream, contiguous stages", src, big, span_c / 4, 128, STAGE, 1 << 30, steps / 4, out, blocks)
    HBC(3, true, 0, 0);
    HBC(3, true, 12, 16);
    // ---- HBM stream, slab rows: every block walks its own tiles
#define HBS(D, B, R, M) run<D, B, R, M>("HBM stream, 4224-byte row pitch", src, big, (size_t)(steps / 4 / 16) * 256 * 4224, 4224, 128, 16, steps / 4, out, blocks)
    HBS(3, true, 0, 0);
    HBS(3, true, 12, 16);
#define HBH(D, B, R, M) run<D, B, R, M, 64>("HBM stream, 4224-byte pitch, 64 B per row", src, big, (size_t)(steps / 4 / 32) * 512 * 4224, 4224, 64, 32, steps / 4, out, blocks)
    HBH(3, true, 0, 0);
    HBH(3, true, 12, 32);
    return 0;
}
