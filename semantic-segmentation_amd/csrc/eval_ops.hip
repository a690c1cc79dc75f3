// Kernels either side of the forward path: the evaluation metric (upsample + arg-max + confusion counts)
// and the input pre-processing (bytes -> normalised NCHW fp32, label table lookup).  Both are streaming
// kernels bound by HBM traffic: 8 B/pixel of labels in, and 4 B in + 20 B out per pixel respectively.
#include "common.h"

namespace {

// bilinear source coordinates, align_corners=False, scale 8 (same arithmetic as the loss kernels)
__device__ __forceinline__ void src_index8(int o, int n, int& i0, int& i1, float& l1) {
    float s = 0.125f * ((float)o + 0.5f) - 0.5f;
    s = fmaxf(s, 0.f);
    i0 = (int)s;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

constexpr int CONF_MAX = 33 * 32;

__global__ __launch_bounds__(256) void up_confusion_kernel(const dsaspp_conf_desc d) {
    __shared__ unsigned int hist[CONF_MAX];
    const int nc = d.n_class, ncell = (nc + 1) * nc;
    for (int i = threadIdx.x; i < ncell; i += 256) hist[i] = 0u;
    __syncthreads();
    const int H = 8 * d.h, W = 8 * d.w;
    const int64_t total = (int64_t)d.n_img * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int X = (int)(idx % W);
        const int64_t t = idx / W;
        const int Y = (int)(t % H);
        const int n = (int)(t / H);
        int x0, x1, y0, y1;
        float lx, ly;
        src_index8(X, d.w, x0, x1, lx);
        src_index8(Y, d.h, y0, y1, ly);
        const float hx = 1.f - lx, hy = 1.f - ly;
        const float* L = d.logits + (int64_t)n * d.h * d.w * d.ldl;
        const float* p00 = L + ((int64_t)y0 * d.w + x0) * d.ldl;
        const float* p01 = L + ((int64_t)y0 * d.w + x1) * d.ldl;
        const float* p10 = L + ((int64_t)y1 * d.w + x0) * d.ldl;
        const float* p11 = L + ((int64_t)y1 * d.w + x1) * d.ldl;
        float best = -3.0e38f;
        int bi = 0;
        for (int c = 0; c < nc; ++c) {
            const float v = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
            if (v > best) { best = v; bi = c; }
        }
        if (d.argmax) d.argmax[idx] = bi;
        const int64_t lab = d.labels[idx];
        const int row = (lab >= 0 && lab < nc) ? (int)lab : nc;
        atomicAdd(&hist[row * nc + bi], 1u);        // a block sees at most 2^32 pixels: the grid has >= 1 block per 2^24
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncell; i += 256) {
        const unsigned int v = hist[i];
        if (v) atomicAdd(reinterpret_cast<unsigned long long*>(d.conf) + i, (unsigned long long)v);
    }
}

// One thread per 4 consecutive pixels of a row: reads 12 image bytes (three 4-byte words when aligned),
// writes one float4 per colour plane and, when labels are present, four int64 labels.
__global__ __launch_bounds__(256) void preprocess_kernel(const dsaspp_prep_desc d, const int64_t total) {
    const int W = d.W, W4 = W / 4;
    const int64_t plane = (int64_t)d.H * W;
    const float mean[3] = {d.mean0, d.mean1, d.mean2}, sd[3] = {d.std0, d.std1, d.std2};
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int xq = (int)(idx % W4);
        const int64_t row = idx / W4;                 // n * H + y
        const int64_t n = row / d.H;
        const int y = (int)(row - n * d.H);
        const int64_t src = row * W + xq * 4;         // first source pixel
        const uint32_t* w3 = reinterpret_cast<const uint32_t*>(d.img + src * 3);   // 12 bytes, 4-byte aligned (W % 4 == 0)
        const uint32_t b0 = w3[0], b1 = w3[1], b2 = w3[2];
        const uint8_t px[12] = {(uint8_t)b0, (uint8_t)(b0 >> 8), (uint8_t)(b0 >> 16), (uint8_t)(b0 >> 24),
                                (uint8_t)b1, (uint8_t)(b1 >> 8), (uint8_t)(b1 >> 16), (uint8_t)(b1 >> 24),
                                (uint8_t)b2, (uint8_t)(b2 >> 8), (uint8_t)(b2 >> 16), (uint8_t)(b2 >> 24)};
        const int xd = d.hflip ? W - 4 - xq * 4 : xq * 4;     // first destination column of the group
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = ((float)px[k * 3 + c] / 255.0f - mean[c]) / sd[c];
            const float4 o = d.hflip ? make_float4(v[3], v[2], v[1], v[0]) : make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(d.x + (n * 3 + c) * plane + (int64_t)y * W + xd) = o;
        }
        if (d.label_ids) {
            const uint32_t lw = *reinterpret_cast<const uint32_t*>(d.label_ids + src);
            int64_t t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] = (int64_t)d.lut[(lw >> (8 * k)) & 255u];
            int64_t* o = d.labels + row * W + xd;
            if (d.hflip) { o[0] = t[3]; o[1] = t[2]; o[2] = t[1]; o[3] = t[0]; }
            else { o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = t[3]; }
        }
    }
}

}  // namespace

extern "C" int dsaspp_upsample8_confusion(const dsaspp_conf_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->h > 0 && d->w > 0, "upsample8_confusion: bad geometry");
    DS_REQUIRE(d->n_class > 0 && d->n_class <= 32 && d->ldl >= d->n_class, "upsample8_confusion: n_class must be in [1,32] and ldl >= n_class");
    DS_REQUIRE(d->logits && d->labels && d->conf, "upsample8_confusion: null tensor");
    const int64_t total = (int64_t)d->n_img * 64 * d->h * d->w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks * (1ll << 24) < total) blocks = (total >> 24) + 1;     // keeps every block's 32-bit LDS counters exact
    hipLaunchKernelGGL(up_confusion_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d);
    DS_LAUNCH_CHECK("upsample8_confusion");
    return 0;
}

extern "C" int dsaspp_preprocess(const dsaspp_prep_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->H > 0 && d->W > 0 && d->W % 4 == 0, "preprocess: width must be a positive multiple of 4 (got %d x %d)", d ? d->H : 0, d ? d->W : 0);
    DS_REQUIRE(d->img && d->x, "preprocess: null image tensor");
    DS_REQUIRE(((uintptr_t)d->img & 3) == 0 && ((uintptr_t)d->x & 15) == 0, "preprocess: image must be 4-byte and output 16-byte aligned");
    DS_REQUIRE(d->std0 != 0.f && d->std1 != 0.f && d->std2 != 0.f, "preprocess: zero std");
    if (d->label_ids) DS_REQUIRE(d->lut && d->labels && ((uintptr_t)d->label_ids & 3) == 0, "preprocess: label table / output missing or labels misaligned");
    const int64_t total = (int64_t)d->n_img * d->H * (d->W / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(preprocess_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d, total);
    DS_LAUNCH_CHECK("preprocess");
    return 0;
}
