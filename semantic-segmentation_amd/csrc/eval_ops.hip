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

// ---------------------------------------------------------------- bilinear x2 upsample of a feature map (output stride 16)
namespace {

// source rows/columns and weights of output index o (align_corners=False, scale 2): even o = 2k reads (k-1, k) with
// weights (0.25, 0.75), odd o = 2k+1 reads (k, k+1) with (0.75, 0.25); indices clamp at the border.
__device__ __forceinline__ void src2(int o, int n, int& i0, int& i1, float& w1) {
    float s = 0.5f * ((float)o + 0.5f) - 0.5f;
    s = fmaxf(s, 0.f);
    i0 = (int)s;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    w1 = s - (float)i0;
}

template <bool F32>
__device__ __forceinline__ void ld8(const void* base, int64_t elem, float (&f)[8]) {
    if (F32) {
        load8f(reinterpret_cast<const float*>(base) + elem, f);
    } else {
        unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(base) + elem), f);
    }
}

template <int CGB, bool F32>
__global__ __launch_bounds__(256) void up2_fwd_kernel(const dsaspp_up2_desc d) {
    __shared__ float red[256 * 16];
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int c = (blockIdx.x * CGB + cx) * 8;
    const bool act = c < d.C;
    const int H = 2 * d.h, W = 2 * d.w;
    const int64_t Mo = (int64_t)d.n_img * H * W;
    float sc[8], sh[8], part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; part[0][e] = 0.f; part[1][e] = 0.f; }
    if (act && d.scale) { load8f(d.scale + c, sc); load8f(d.shift + c, sh); }
    const float lo = d.relu ? 0.f : -3.0e38f;
    if (act) {
        for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < Mo; m += (int64_t)gridDim.y * RB) {
            const int n = (int)(m / ((int64_t)H * W));
            const int rem = (int)(m - (int64_t)n * H * W);
            const int Y = rem / W, X = rem - Y * W;
            int y0, y1, x0, x1;
            float ly, lx;
            src2(Y, d.h, y0, y1, ly);
            src2(X, d.w, x0, x1, lx);
            const int64_t b = (int64_t)n * d.h * d.w;
            float a[8], bq[8], cq[8], dq[8], o[8];
            ld8<F32>(d.x, (b + (int64_t)y0 * d.w + x0) * d.ldx + c, a);
            ld8<F32>(d.x, (b + (int64_t)y0 * d.w + x1) * d.ldx + c, bq);
            ld8<F32>(d.x, (b + (int64_t)y1 * d.w + x0) * d.ldx + c, cq);
            ld8<F32>(d.x, (b + (int64_t)y1 * d.w + x1) * d.ldx + c, dq);
            const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = hy * (hx * a[e] + lx * bq[e]) + ly * (hx * cq[e] + lx * dq[e]);
                o[e] = fmaxf(fmaf(v, sc[e], sh[e]), lo);
            }
            if (F32) {
                store8f(reinterpret_cast<float*>(d.y) + m * d.ldy + c, o);
            } else {
                const uint4 ov = pack8(o);
                *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(d.y) + m * d.ldy + c) = ov;
                if (d.stat_acc) {
                    float g[8];
                    unpack8(ov, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { part[0][e] += g[e]; part[1][e] += g[e] * g[e]; }
                }
            }
        }
    }
    if (!F32 && d.stat_acc) block_colsum<CGB, 2>(part, red, d.stat_acc, d.C, blockIdx.x * CGB, 2 * d.C, blockIdx.y % DS_R);
}

// adjoint: a low-resolution pixel (y, x) is read by output rows 2y-1 .. 2y+2 (weights 0.25, 0.75, 0.75, 0.25), plus,
// at the borders, the clamped reads: row 0 also takes the weight that would have gone to row -1 and row h-1 the
// one for row h.  Implemented as a gather over the <= 4x4 outputs with the weight each gives to (y, x).
template <int CGB>
__global__ __launch_bounds__(256) void up2_bwd_kernel(const dsaspp_up2_desc d) {
    __shared__ float red[256 * 16];
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int c = (blockIdx.x * CGB + cx) * 8;
    const bool act = c < d.C;
    const int H = 2 * d.h, W = 2 * d.w;
    const int64_t Mi = (int64_t)d.n_img * d.h * d.w;
    float mu[8], rs[8], part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mu[e] = 0.f; rs[e] = 1.f; part[0][e] = 0.f; part[1][e] = 0.f; }
    if (act && d.s_acc) { load8f(d.mean + c, mu); load8f(d.rstd + c, rs); }
    const u16* dyg = reinterpret_cast<const u16*>(d.y);
    if (act) {
        for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < Mi; m += (int64_t)gridDim.y * RB) {
            const int n = (int)(m / ((int64_t)d.h * d.w));
            const int rem = (int)(m - (int64_t)n * d.h * d.w);
            const int yl = rem / d.w, xl = rem - yl * d.w;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            const int Y0 = max(2 * yl - 1, 0), Y1 = min(2 * yl + 2, H - 1);
            const int X0 = max(2 * xl - 1, 0), X1 = min(2 * xl + 2, W - 1);
            for (int Y = Y0; Y <= Y1; ++Y) {
                int y0, y1;
                float ly;
                src2(Y, d.h, y0, y1, ly);
                float wy = 0.f;
                if (y0 == yl) wy += 1.f - ly;
                if (y1 == yl) wy += ly;
                if (wy == 0.f) continue;
                for (int X = X0; X <= X1; ++X) {
                    int x0, x1;
                    float lx;
                    src2(X, d.w, x0, x1, lx);
                    float wx = 0.f;
                    if (x0 == xl) wx += 1.f - lx;
                    if (x1 == xl) wx += lx;
                    if (wx == 0.f) continue;
                    float g[8];
                    unpack8(*reinterpret_cast<const uint4*>(dyg + (((int64_t)n * H + Y) * W + X) * d.ldy + c), g);
                    const float wgt = wy * wx;
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
                }
            }
            if (d.dx_f32) store8f(reinterpret_cast<float*>(d.dx) + m * d.lddx + c, acc);
            else *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(d.dx) + m * d.lddx + c) = pack8(acc);
            if (d.s_acc) {
                float yv[8];
                unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(d.x) + m * d.ldx + c), yv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    part[0][e] += acc[e];
                    part[1][e] += acc[e] * ((yv[e] - mu[e]) * rs[e]);
                }
            }
        }
    }
    if (d.s_acc) block_colsum<CGB, 2>(part, red, d.s_acc, d.C, blockIdx.x * CGB, 2 * d.C, blockIdx.y % DS_R);
}

dim3 up2_grid(int C, int64_t M, int cgb) {
    const int gx = ((C + 7) / 8 + cgb - 1) / cgb;
    const int rb = 256 / cgb;
    int64_t gy = (M + rb - 1) / rb;
    const int64_t cap = (4096 + gx - 1) / gx;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    return dim3(gx, (unsigned)gy);
}

int up2_check(const dsaspp_up2_desc* d, const char* who) {
    DS_REQUIRE(d && d->n_img > 0 && d->h > 0 && d->w > 0 && d->C > 0 && d->C % 8 == 0, "%s: bad geometry", who);
    DS_REQUIRE(d->y && d->ldy % 8 == 0 && d->ldy >= d->C, "%s: bad high-resolution tensor", who);
    return 0;
}

}  // namespace

extern "C" int dsaspp_upsample2_fwd(const dsaspp_up2_desc* d, dsaspp_stream_t stream) {
    if (int rc = up2_check(d, "upsample2_fwd")) return rc;
    DS_REQUIRE(d->x && d->ldx % 8 == 0 && d->ldx >= d->C, "upsample2_fwd: bad input");
    DS_REQUIRE((d->scale == nullptr) == (d->shift == nullptr), "upsample2_fwd: scale/shift must come together");
    DS_REQUIRE(d->dtype == 0 || (d->dtype == 1 && !d->stat_acc), "upsample2_fwd: the fp32 form has no statistics");
    const int64_t Mo = (int64_t)d->n_img * 4 * d->h * d->w;
    const int cgb = d->C <= 64 ? 8 : 32;
    const dim3 grid = up2_grid(d->C, Mo, cgb);
    hipStream_t s = (hipStream_t)stream;
    if (d->dtype == 1) {
        if (cgb == 8) hipLaunchKernelGGL((up2_fwd_kernel<8, true>), grid, dim3(256), 0, s, *d);
        else hipLaunchKernelGGL((up2_fwd_kernel<32, true>), grid, dim3(256), 0, s, *d);
    } else {
        if (cgb == 8) hipLaunchKernelGGL((up2_fwd_kernel<8, false>), grid, dim3(256), 0, s, *d);
        else hipLaunchKernelGGL((up2_fwd_kernel<32, false>), grid, dim3(256), 0, s, *d);
    }
    DS_LAUNCH_CHECK("upsample2_fwd");
    return 0;
}

extern "C" int dsaspp_upsample2_bwd(const dsaspp_up2_desc* d, dsaspp_stream_t stream) {
    if (int rc = up2_check(d, "upsample2_bwd")) return rc;
    DS_REQUIRE(d->dx && d->lddx % 8 == 0 && d->lddx >= d->C, "upsample2_bwd: bad output");
    if (d->s_acc) DS_REQUIRE(d->x && d->ldx % 8 == 0 && d->mean && d->rstd, "upsample2_bwd: s_acc needs yref (x), mean and rstd");
    const int64_t Mi = (int64_t)d->n_img * d->h * d->w;
    const int cgb = d->C <= 64 ? 8 : 32;
    const dim3 grid = up2_grid(d->C, Mi, cgb);
    if (cgb == 8) hipLaunchKernelGGL(up2_bwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    else hipLaunchKernelGGL(up2_bwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, *d);
    DS_LAUNCH_CHECK("upsample2_bwd");
    return 0;
}
