// Shared device helpers for the gfx950 kernels of libdsaspp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dsaspp.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short u16;

#define DS_R DSASPP_STAT_REPLICAS

int dsaspp_set_error(const char* fmt, ...);

// Per-device one-time set-up (program.hip).  HIP keeps hipFuncAttributeMaxDynamicSharedMemorySize per device function and
// any host thread may drive any device (nn.DataParallel runs one thread per device), so neither a process-wide nor a
// thread-local flag says whether the CURRENT device has been set up: both helpers key on hipGetDevice under a mutex.
// Raises the dynamic-LDS limit of kernel `fn` to `bytes` on the current device, once per (function, device).
void dsaspp_lds_limit(const void* fn, int bytes);
// Compute units of the current device that a persistent (one block per CU) kernel may occupy: the device's CU count
// minus what dsaspp_set_comm_cus reserved, never fewer than 8.
int dsaspp_persistent_cus(void);

// persistent grouped 1x1 data gradient (dgrad_group.hip): -1 = descriptor not of its form, 0 = launched
int dsaspp_dgrad_group_launch(const dsaspp_conv_desc& d, hipStream_t s);
// single-source slab data gradient with a long K on a 256 x 256 tile (dgrad_wide.hip): same return convention
int dsaspp_dgrad_wide_launch(const dsaspp_conv_desc& d, hipStream_t s);
// BN-ReLU-1x1 forward with many output channels and a long K on the same tile (conv_wide.hip): same return convention
int dsaspp_conv_wide_launch(const dsaspp_conv_desc& d, hipStream_t s, bool narrow_only = false);
// halo-patch 3x3 convolution with few output channels (conv3x3_halo.hip): same return convention
int dsaspp_conv3x3_halo_launch(const dsaspp_conv_desc& d, hipStream_t s);

// 3x3 data gradient of a dense layer fused with the gradient-slab read-out (dgrad3x3_fused.hip); 0 = launched, else an error code
int dsaspp_dgrad3x3_fused_launch(const dsaspp_conv_desc& d, hipStream_t s);

// grouped 1x1 weight gradient of layers that share their input (wgrad_group.hip); 0 = launched, else an error code
int dsaspp_wgrad_group_launch(const dsaspp_wgrad_desc& d, hipStream_t s);

// 1x1 weight gradient with many input and output channels on a 256 x 256 tile (wgrad_wide.hip): -1 = not of its form, 0 = launched
int dsaspp_wgrad_wide_launch(const dsaspp_wgrad_desc& d, hipStream_t s);
// 3x3 weight gradient of a dense layer, every activation row drawn once for all nine taps (wgrad3x3_strip.hip): same return convention
int dsaspp_wgrad3x3_strip_launch(const dsaspp_wgrad_desc& d, hipStream_t s);
// records the kernel family of the calling thread's most recent convolution / weight-gradient op (dsaspp_conv_last_route)
void dsaspp_note_route(const char* name);

// streaming BN-ReLU-1x1 forward for cout <= 192 (conv1x1_stream.hip): same return convention
int dsaspp_conv1x1_stream_launch(const dsaspp_conv_desc& d, hipStream_t s);

// fp32 forms (descriptor dtype == 1) of the eval-mode forward kernels (fp32_path.hip); called by the public entry
// points after their argument checks
int dsaspp_f32_conv(const dsaspp_conv_desc& d, hipStream_t s);
int dsaspp_f32_bn_act(const dsaspp_bnact_desc& d, hipStream_t s);
int dsaspp_f32_avgpool(const dsaspp_pool_desc& d, hipStream_t s);
int dsaspp_f32_im2col(const dsaspp_im2col_desc& d, hipStream_t s);
int dsaspp_f32_maxpool(const dsaspp_maxpool_desc& d, hipStream_t s);
int dsaspp_f32_stem3(const dsaspp_stem3_desc& d, hipStream_t s);
int dsaspp_f32_dwconv(const dsaspp_dw_desc& d, hipStream_t s);
// training side of the fp32 form (DenseNet trunks)
int dsaspp_f32_wgrad(const dsaspp_wgrad_desc& d, hipStream_t s);
int dsaspp_f32_grad_extract(const dsaspp_gradx_desc& d, hipStream_t s);
int dsaspp_f32_bn_bwd_apply(const dsaspp_bnapply_desc& d, hipStream_t s);
int dsaspp_f32_avgpool_bwd(const dsaspp_pool_desc& d, hipStream_t s);
int dsaspp_f32_maxpool_bwd(const dsaspp_maxpool_desc& d, hipStream_t s);

#define DS_REQUIRE(cond, ...)                                  \
    do {                                                       \
        if (!(cond)) return dsaspp_set_error(__VA_ARGS__);     \
    } while (0)

#define DS_LAUNCH_CHECK(name)                                                          \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) return dsaspp_set_error("%s: %s", name, hipGetErrorString(e__)); \
    } while (0)

__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t f2bf_bits(float f) {
    __bf16 b = (__bf16)f;  // round-to-nearest-even, NaN preserving (v_cvt_pk_bf16_f32)
    return (uint32_t)__builtin_bit_cast(u16, b);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
// two floats -> packed bf16 pair, round-to-nearest-even: ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// ReLU on a packed bf16 pair as a signed 16-bit max (v_pk_max_i16): bf16 values with the sign bit set are
// negative as int16 too.  floor_bits = 0 applies the ReLU, 0x80008000 (INT16_MIN twice) leaves the pair as it is,
// so the choice costs no branch.  Rounding is monotonic and keeps the sign: relu(round(x)) == round(relu(x)).
__device__ __forceinline__ uint32_t relu_bf2(uint32_t v, uint32_t floor_bits) {
    const i16x2_t a = __builtin_bit_cast(i16x2_t, v), f = __builtin_bit_cast(i16x2_t, floor_bits);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(a, f));
}
__device__ __forceinline__ uint4 relu_bf8(uint4 v, uint32_t floor_bits) {
    v.x = relu_bf2(v.x, floor_bits); v.y = relu_bf2(v.y, floor_bits);
    v.z = relu_bf2(v.z, floor_bits); v.w = relu_bf2(v.w, floor_bits);
    return v;
}
__device__ __forceinline__ float bf_round(float f) { return __uint_as_float(f2bf_bits(f) << 16); }

// 16-byte global accesses with the non-temporal hint: for tensors that are streamed through once (hundreds of MB
// against 32 MB of L2 and 256 MB of memory-side cache), so that they do not push out what IS used again -- weights, halos,
// the rows a neighbouring tile shares.  NT = false gives the plain access (one template for both, chosen per call site).
// Measured in the training step (batch 32, same box): the BatchNorm-backward apply with hinted loads and stores 448.9 ->
// 444.7 ms, the other row-wise kernels with hinted loads 450.2 -> 446.8 (hinted stores add nothing there).  NOT for tensors
// the next kernel reads at once: hinted stores of the convolutions' outputs cost 7 ms, the hint on the grouped data
// gradient's slab traffic 3 ms.
typedef uint32_t ds_u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ uint4 ld16(const void* p) {
    if (!NT) return *reinterpret_cast<const uint4*>(p);
    const ds_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ds_u32x4*>(p));
    return uint4{v.x, v.y, v.z, v.w};
}
template <bool NT>
__device__ __forceinline__ void st16(void* p, const uint4 v) {
    if (!NT) *reinterpret_cast<uint4*>(p) = v;
    else __builtin_nontemporal_store(ds_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<ds_u32x4*>(p));
}
// 8 bf16 (one 16-byte chunk) <-> 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = bf_lo(v.x); f[1] = bf_hi(v.x); f[2] = bf_lo(v.y); f[3] = bf_hi(v.y);
    f[4] = bf_lo(v.z); f[5] = bf_hi(v.z); f[6] = bf_lo(v.w); f[7] = bf_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    uint4 v;
    v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
    v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
    return v;
}
__device__ __forceinline__ void load8f(const float* p, float (&f)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ void store8f(float* p, const float (&f)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

// Sum over the 16 lanes of a DPP row (lanes 16k..16k+15); every lane of the row ends up with the total.
// Four VALU adds with DPP operand swizzles (quad permutes, then the half-row and row mirrors) instead of
// four LDS-pipe permutes.
__device__ __forceinline__ float row16_sum(float v) {
    int x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));   // row_half_mirror
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

// Bijective XCD-aware block remap: blocks b and b+8 share an XCD (and its L2), so give each
// XCD group a contiguous range of logical tile ids.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Row-wise streaming kernels: a block is CGB channel groups (8 channels each) wide and
// 256/CGB rows tall.  Reduce per-thread column partials over the rows of the block and add
// them to a replica row of an accumulator [R][NS*C] (sum s of channel c at [rep][s*C + c]).
template <int CGB, int NS>
__device__ __forceinline__ void block_colsum(float (&part)[NS][8], float* lds /* 256*NS*8 floats */,
                                             float* acc, int C, int cg0, int rep_stride, int rep) {
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) lds[(ry * CGB + cx) * (NS * 8) + s * 8 + e] = part[s][e];
    __syncthreads();
    for (int o = threadIdx.x; o < CGB * NS * 8; o += 256) {
        const int ocx = o / (NS * 8), rest = o % (NS * 8);
        const int s = rest / 8, e = rest % 8;
        float t = 0.f;
        for (int r = 0; r < RB; ++r) t += lds[(r * CGB + ocx) * (NS * 8) + rest];
        const int c = (cg0 + ocx) * 8 + e;
        if (c < C) atomicAdd(acc + (size_t)rep * rep_stride + s * C + c, t);
    }
}
