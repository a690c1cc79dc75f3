// Which LDS row strides are free of bank conflicts for the transposing fragment read of the weight-gradient kernels?
// Lane (g = lane / 16, i = lane % 16) reads 8 bytes at row g * 8 + i / 4 (+4 for the second half), byte column (i % 4) * 8 of a
// 16-channel block -- the address pattern of tr_frag() in wgrad_group.hip / wgrad_mfma.hip.  Eight waves of one block issue the
// read back to back; the time per read against the row stride shows the conflict-free strides (and the same for ds_read_b128
// of the forward kernels with --b128: lane (fr = lane % 16, fc = lane / 16) reads 16 bytes at row fr, byte column fc * 16).
//   hipcc --offload-arch=gfx950 -O3 -o tools/lds_tr_stride_probe tools/lds_tr_stride_probe.hip && ./tools/lds_tr_stride_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

__global__ __launch_bounds__(512) void probe(const int stride, const int mode, const int iters, unsigned long long* out) {
    extern __shared__ unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 40960 / 4; i += 512) reinterpret_cast<int*>(lds)[i] = i;
    __syncthreads();
    int addr;
    if (mode == 0) addr = ((lane >> 4) * 8 + ((lane & 15) >> 2)) * stride + (lane & 3) * 8;
    else addr = (lane & 15) * stride + (lane >> 4) * 16;
    addr += wave * 32;                                   // the waves of a block read different 16-channel blocks
    typedef __attribute__((address_space(3))) unsigned char* lp;
    const unsigned base = (unsigned)(unsigned long long)(lp)lds + addr;
    unsigned long long acc = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (mode == 0) {
            unsigned long long v0, v1, v2, v3;
            asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:512\n\tds_read_b64_tr_b16 %2, %4 offset:1024\n\t"
                         "ds_read_b64_tr_b16 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(base) : "memory");
            acc += v0 ^ v1 ^ v2 ^ v3;
        } else {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 v0, v1, v2, v3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:512\n\tds_read_b128 %2, %4 offset:1024\n\t"
                         "ds_read_b128 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "v"(base) : "memory");
            acc += v0.x ^ v1.y ^ v2.z ^ v3.w;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x123456789ull) out[1] = acc;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 && !strcmp(argv[1], "--b128") ? 1 : 0;
    unsigned long long* out;
    hipMalloc(&out, 16);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    const int iters = 2000;
    printf("%s: clocks (100 MHz ticks x 1000) per wave-read, 8 waves, by row stride in bytes\n", mode ? "ds_read_b128" : "ds_read_b64_tr_b16");
    for (int stride = 32; stride <= 1056; stride += 16) {
        if (31 * stride + 2048 > 60 * 1024) break;
        unsigned long long h[2] = {0, 0};
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(512), 64 * 1024, 0, stride, mode, iters, out);
            hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        }
        printf("%5d %7.1f%s", stride, 1000.0 * h[0] / (4.0 * iters * 8), ((stride / 16) % 6 == 1) ? "\n" : "   ");
    }
    printf("\n");
    return 0;
}
