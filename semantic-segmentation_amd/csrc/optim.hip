// Parameter-side kernels over the flat parameter arena: fused Adam (coupled L2, as
// torch.optim.Adam(weight_decay=...)) that also refreshes the bf16 shadow weights, the
// bf16 cast, the batched weight transposes used by the data-gradient convolutions, and
// small strided copies.  All HBM-bound, 16 bytes per lane where the layout allows.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(const dsaspp_adam_desc d, const float bc1, const float bc2) {
    const int64_t n4 = d.n >> 2;
    u16* w16 = reinterpret_cast<u16*>(d.w16);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 p = reinterpret_cast<float4*>(d.p)[i];
        const float4 g4 = reinterpret_cast<const float4*>(d.g)[i];
        float4 m = reinterpret_cast<float4*>(d.m)[i];
        float4 v = reinterpret_cast<float4*>(d.v)[i];
        float pp[4] = {p.x, p.y, p.z, p.w}, gg[4] = {g4.x, g4.y, g4.z, g4.w};
        float mm[4] = {m.x, m.y, m.z, m.w}, vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g = gg[k] * d.grad_scale + d.weight_decay * pp[k];
            mm[k] = d.beta1 * mm[k] + (1.f - d.beta1) * g;
            vv[k] = d.beta2 * vv[k] + (1.f - d.beta2) * g * g;
            const float denom = sqrtf(vv[k]) / sqrtf(bc2) + d.eps;
            pp[k] -= (d.lr / bc1) * (mm[k] / denom);
        }
        reinterpret_cast<float4*>(d.p)[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
        reinterpret_cast<float4*>(d.m)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
        reinterpret_cast<float4*>(d.v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
        if (w16) {
            uint2 o;
            o.x = pack_bf2(pp[0], pp[1]);
            o.y = pack_bf2(pp[2], pp[3]);
            reinterpret_cast<uint2*>(w16)[i] = o;
        }
    }
}

__global__ __launch_bounds__(256) void cast_kernel(const float* src, u16* dst, const int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 p = reinterpret_cast<const float4*>(src)[i];
        uint2 o;
        o.x = pack_bf2(p.x, p.y);
        o.y = pack_bf2(p.z, p.w);
        reinterpret_cast<uint2*>(dst)[i] = o;
    }
}

// dst[c][taps-1-t][n] = src[n][t][c]; one block row (blockIdx.y) per table entry
template <typename T>
__global__ __launch_bounds__(256) void wtrans_kernel(const T* src, T* dst, const dsaspp_wt_entry* table) {
    const dsaspp_wt_entry e = table[blockIdx.y];
    const int64_t total = (int64_t)e.cout * e.cin * e.taps;
    const T* s = src + e.src_off;
    T* o = dst + e.dst_off;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // i indexes dst: [c][t'][n]
        const int n = (int)(i % e.cout);
        const int64_t r = i / e.cout;
        const int tp = (int)(r % e.taps);
        const int c = (int)(r / e.taps);
        const int t = e.taps - 1 - tp;
        o[i] = s[((int64_t)n * e.taps + t) * e.cin + c];
    }
}

template <bool TO_BF16>
__global__ __launch_bounds__(256) void copy2d_kernel(const float* src, int lds, void* dst, int ldd, int rows, int cols,
                                                     int dst_cols, int accumulate) {
    const int64_t total = (int64_t)rows * dst_cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / dst_cols), c = (int)(i % dst_cols);
        const float v = c < cols ? src[(int64_t)r * lds + c] : 0.f;
        if (TO_BF16) {
            reinterpret_cast<u16*>(dst)[(int64_t)r * ldd + c] = (u16)f2bf_bits(v);
        } else {
            float* o = reinterpret_cast<float*>(dst) + (int64_t)r * ldd + c;
            *o = accumulate ? *o + v : v;
        }
    }
}

inline unsigned stream_blocks(int64_t items) {
    int64_t b = (items + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int dsaspp_adam_step(const dsaspp_adam_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n > 0 && d->n % 4 == 0, "adam_step: n must be a positive multiple of 4");
    DS_REQUIRE(d->p && d->g && d->m && d->v && d->step >= 1, "adam_step: bad descriptor");
    DS_REQUIRE((((uintptr_t)d->p | (uintptr_t)d->g | (uintptr_t)d->m | (uintptr_t)d->v) & 15) == 0, "adam_step: buffers must be 16-byte aligned");
    const float bc1 = 1.f - powf(d->beta1, (float)d->step);
    const float bc2 = 1.f - powf(d->beta2, (float)d->step);
    hipLaunchKernelGGL(adam_kernel, dim3(stream_blocks(d->n / 4)), dim3(256), 0, (hipStream_t)stream, *d, bc1, bc2);
    DS_LAUNCH_CHECK("adam_step");
    return 0;
}

extern "C" int dsaspp_cast_bf16(const float* src, void* dst, int64_t n, dsaspp_stream_t stream) {
    DS_REQUIRE(src && dst && n > 0 && n % 4 == 0, "cast_bf16: n must be a positive multiple of 4");
    hipLaunchKernelGGL(cast_kernel, dim3(stream_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, src, reinterpret_cast<u16*>(dst), n);
    DS_LAUNCH_CHECK("cast_bf16");
    return 0;
}

extern "C" int dsaspp_weight_transpose(const void* src, void* dst, const dsaspp_wt_entry* table_dev, int32_t n_entries,
                                       int64_t max_elems, dsaspp_stream_t stream) {
    DS_REQUIRE(src && dst && table_dev && n_entries > 0 && max_elems > 0, "weight_transpose: bad arguments");
    DS_REQUIRE(n_entries <= 65535, "weight_transpose: too many entries");
    int64_t bx = (max_elems + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(wtrans_kernel<u16>, dim3((unsigned)bx, (unsigned)n_entries), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const u16*>(src), reinterpret_cast<u16*>(dst), table_dev);
    DS_LAUNCH_CHECK("weight_transpose");
    return 0;
}

extern "C" int dsaspp_weight_transpose_f32(const float* src, float* dst, const dsaspp_wt_entry* table_dev, int32_t n_entries,
                                           int64_t max_elems, dsaspp_stream_t stream) {
    DS_REQUIRE(src && dst && table_dev && n_entries > 0, "weight_transpose_f32: bad arguments");
    int64_t bx = (max_elems + 255) / 256;
    if (bx > 256) bx = 256;
    hipLaunchKernelGGL(wtrans_kernel<float>, dim3((unsigned)bx, (unsigned)n_entries), dim3(256), 0, (hipStream_t)stream, src, dst,
                       table_dev);
    DS_LAUNCH_CHECK("weight_transpose_f32");
    return 0;
}

extern "C" int dsaspp_copy2d_f32(const float* src, int32_t lds, float* dst, int32_t ldd, int32_t rows, int32_t cols,
                                 int32_t dst_cols, int32_t accumulate, dsaspp_stream_t stream) {
    DS_REQUIRE(src && dst && rows > 0 && cols > 0 && dst_cols >= cols && ldd >= dst_cols && lds >= cols, "copy2d_f32: bad arguments");
    hipLaunchKernelGGL(copy2d_kernel<false>, dim3(stream_blocks((int64_t)rows * dst_cols)), dim3(256), 0, (hipStream_t)stream,
                       src, lds, (void*)dst, ldd, rows, cols, dst_cols, accumulate);
    DS_LAUNCH_CHECK("copy2d_f32");
    return 0;
}

extern "C" int dsaspp_copy2d_f32_bf16(const float* src, int32_t lds, void* dst, int32_t ldd, int32_t rows, int32_t cols,
                                      int32_t dst_cols, dsaspp_stream_t stream) {
    DS_REQUIRE(src && dst && rows > 0 && cols > 0 && dst_cols >= cols && ldd >= dst_cols && lds >= cols, "copy2d_f32_bf16: bad arguments");
    hipLaunchKernelGGL(copy2d_kernel<true>, dim3(stream_blocks((int64_t)rows * dst_cols)), dim3(256), 0, (hipStream_t)stream,
                       src, lds, dst, ldd, rows, cols, dst_cols, 0);
    DS_LAUNCH_CHECK("copy2d_f32_bf16");
    return 0;
}

extern "C" int dsaspp_memset_async(void* p, int32_t value, int64_t bytes, dsaspp_stream_t stream) {
    DS_REQUIRE(p && bytes >= 0, "memset_async: bad arguments");
    if (bytes == 0) return 0;
    hipError_t e = hipMemsetAsync(p, value, (size_t)bytes, (hipStream_t)stream);
    DS_REQUIRE(e == hipSuccess, "memset_async: %s", hipGetErrorString(e));
    return 0;
}
