// HBM-bound row-wise kernels around the convolutions: BatchNorm bookkeeping, slab gradient
// read-out, BN backward application, pooling and the stem im2col.  NHWC bf16, one 16-byte
// chunk (8 channels) per lane per row; per-channel sums are reduced in registers, then across
// the block through LDS, then added to one of DS_R replica rows with fp32 atomics.
#include "common.h"

#ifndef EW_NT
#define EW_NT 1                  // row-wise kernels: bit 0 loads, bit 1 stores with the non-temporal hint
#endif
constexpr bool EWL = (EW_NT & 1) != 0, EWS = (EW_NT & 2) != 0;

namespace {

// ---------------------------------------------------------------- BN finalize (forward)
__global__ void bn_finalize_kernel(const dsaspp_bn_desc d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0 && !d.use_running && d.num_batches_tracked) *d.num_batches_tracked += 1;
    if (c >= d.C) return;
    float mean, var;
    if (d.use_running) {
        mean = d.running_mean[c];
        var = d.running_var[c];
    } else {
        const int a = c - d.acc_c0;
        if (d.acc && a >= 0 && a < d.acc_cn) {
            float s = 0.f, q = 0.f;
            for (int r = 0; r < DS_R; ++r) {
                float* row = d.acc + (size_t)r * 2 * d.acc_cn;
                s += row[a];
                q += row[d.acc_cn + a];
                row[a] = 0.f;
                row[d.acc_cn + a] = 0.f;
            }
            mean = s / d.count;
            var = fmaxf(q / d.count - mean * mean, 0.f);
            d.mean[c] = mean;
            d.var[c] = var;
            d.rstd[c] = rsqrtf(var + d.eps);
        } else {
            mean = d.mean[c];
            var = d.var[c];
        }
        if (d.running_mean) {
            const float unbiased = d.count > 1.f ? var * d.count / (d.count - 1.f) : var;
            d.running_mean[c] = (1.f - d.momentum) * d.running_mean[c] + d.momentum * mean;
            d.running_var[c] = (1.f - d.momentum) * d.running_var[c] + d.momentum * unbiased;
        }
    }
    if (d.scale) {
        const float rstd = rsqrtf(var + d.eps);
        const float g = d.gamma ? d.gamma[c] : 1.f, b = d.beta ? d.beta[c] : 0.f;
        const float sc = g * rstd;
        d.scale[c] = sc;
        d.shift[c] = b - mean * sc;
    }
}

// ---------------------------------------------------------------- BN finalize (backward)
__global__ void bn_bwd_finalize_kernel(const dsaspp_bn_bwd_desc d) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.C) return;
    float s1 = 0.f, s2 = 0.f;
    const int ald = d.acc_ld > 0 ? d.acc_ld : d.C;
    for (int r = 0; r < DS_R; ++r) {
        float* row = d.acc + (size_t)r * 2 * ald;
        s1 += row[c];
        s2 += row[ald + c];
        row[c] = 0.f;
        row[ald + c] = 0.f;
    }
    if (d.dbeta) d.dbeta[c] += s1;
    if (d.dgamma) d.dgamma[c] += s2;
    const float sc = d.scale[c];
    const float inv = 1.f / d.count;
    if (d.mode == 0) {
        const float B = -sc * d.rstd[c] * s2 * inv;
        d.coefA[c] = sc;
        d.coefB[c] = B;
        d.coefC[c] = -sc * s1 * inv - B * d.mean[c];
    } else {
        d.corr1[c] += sc * s1 * inv;
        d.corr2[c] += sc * s2 * inv;
    }
}

// ---------------------------------------------------------------- y = relu(x*scale+shift) (+stats)
template <int CGB>
__global__ __launch_bounds__(256) void bn_act_kernel(const dsaspp_bnact_desc d) {
    __shared__ float red[256 * 16];
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int cg = blockIdx.x * CGB + cx;
    const int c = cg * 8;
    const bool act = c < d.C;
    float sc[8], sh[8], part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; part[0][e] = 0.f; part[1][e] = 0.f; }
    if (act && d.scale) { load8f(d.scale + c, sc); load8f(d.shift + c, sh); }
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    u16* yg = reinterpret_cast<u16*>(d.y);
    const u16* rg = reinterpret_cast<const u16*>(d.res);
    const float lo = d.relu ? 0.f : -3.0e38f, hi = (d.relu == 2) ? 6.f : 3.0e38f;
    if (act) {
        for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < d.M; m += (int64_t)gridDim.y * RB) {
            const uint4 v = ld16<EWL>(xg + m * d.ldx + c);
            float f[8];
            unpack8(v, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
            if (rg) {
                float r[8];
                unpack8(ld16<EWL>(rg + m * d.ldres + c), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += r[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e], lo), hi);
            const uint4 o = pack8(f);
            st16<EWS>(yg + m * d.ldy + c, o);
            if (d.stat_acc) {
                float g[8];
                unpack8(o, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) { part[0][e] += g[e]; part[1][e] += g[e] * g[e]; }
            }
        }
    }
    if (d.stat_acc) block_colsum<CGB, 2>(part, red, d.stat_acc, d.C, blockIdx.x * CGB, 2 * d.C, blockIdx.y % DS_R);
}

// ---------------------------------------------------------------- slab gradient read-out
template <int CGB>
__global__ __launch_bounds__(256) void grad_extract_kernel(const dsaspp_gradx_desc d) {
    __shared__ float red[256 * 16];
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int cg = blockIdx.x * CGB + cx;
    const int c = cg * 8;
    const bool act = c < d.C;
    float k1[8], k2[8], sm[8], sr[8], rs[8], rh[8], rm[8], rr[8], part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        k1[e] = 0.f; k2[e] = 0.f; sm[e] = 0.f; sr[e] = 1.f; rs[e] = 1.f; rh[e] = 0.f; rm[e] = 0.f; rr[e] = 1.f;
        part[0][e] = 0.f; part[1][e] = 0.f;
    }
    if (act) {
        if (d.corr1) { load8f(d.corr1 + c, k1); load8f(d.corr2 + c, k2); load8f(d.s_mean + c, sm); load8f(d.s_rstd + c, sr); }
        if (d.r_scale) { load8f(d.r_scale + c, rs); load8f(d.r_shift + c, rh); }
        if (d.s_acc) { load8f(d.r_mean + c, rm); load8f(d.r_rstd + c, rr); }
    }
    const u16* xs = reinterpret_cast<const u16*>(d.xs);
    const u16* yr = reinterpret_cast<const u16*>(d.yref);
    u16* dz = reinterpret_cast<u16*>(d.dz);
    if (act) {
        for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < d.M; m += (int64_t)gridDim.y * RB) {
            float v[8];
            if (d.g_bf16) unpack8(ld16<EWL>(reinterpret_cast<const u16*>(d.g) + m * d.ldg + c), v);
            else load8f(reinterpret_cast<const float*>(d.g) + m * d.ldg + c, v);
            if (d.corr1) {
                float x[8];
                unpack8(ld16<EWL>(xs + m * d.ldxs + c), x);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] -= k1[e] + ((x[e] - sm[e]) * sr[e]) * k2[e];
            }
            if (d.mask) {
                float mk[8];
                load8f(d.mask + (m / d.hw) * d.mask_ld + c, mk);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= mk[e];
            }
            float y[8];
            if (d.relu || d.s_acc) unpack8(ld16<EWL>(yr + m * d.ldyref + c), y);
            if (d.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pre = d.r_scale ? fmaf(y[e], rs[e], rh[e]) : y[e];
                    if (!(pre > 0.f) || (d.relu == 2 && pre >= 6.f)) v[e] = 0.f;
                }
            }
            if (d.dz_f32) store8f(reinterpret_cast<float*>(d.dz) + m * d.lddz + c, v);
            else st16<EWS>(dz + m * d.lddz + c, pack8(v));
            if (d.s_acc) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { part[0][e] += v[e]; part[1][e] += v[e] * ((y[e] - rm[e]) * rr[e]); }
            } else if (d.colsum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) part[0][e] += v[e];
            }
        }
    }
    if (d.s_acc) block_colsum<CGB, 2>(part, red, d.s_acc, d.C, blockIdx.x * CGB, 2 * d.C, blockIdx.y % DS_R);
    else if (d.colsum) {
        float p1[1][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) p1[0][e] = part[0][e];
        block_colsum<CGB, 1>(p1, red, d.colsum, d.C, blockIdx.x * CGB, 0, 0);
    }
}

// ---------------------------------------------------------------- dy = A*dr + B*y + C
template <int CGB>
__global__ __launch_bounds__(256) void bn_apply_kernel(const dsaspp_bnapply_desc d) {
    __shared__ float red[256 * 8];
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int cg = blockIdx.x * CGB + cx;
    const int c = cg * 8;
    const bool act = c < d.C;
    float A[8], B[8], Cc[8], part[1][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { A[e] = 0.f; B[e] = 0.f; Cc[e] = 0.f; part[0][e] = 0.f; }
    if (act) { load8f(d.coefA + c, A); load8f(d.coefB + c, B); load8f(d.coefC + c, Cc); }
    const u16* dr = reinterpret_cast<const u16*>(d.dr);
    const u16* yy = reinterpret_cast<const u16*>(d.y);
    u16* dy = reinterpret_cast<u16*>(d.dy);
    if (act) {
        for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < d.M; m += (int64_t)gridDim.y * RB) {
            float a[8], y[8], o[8];
            if (d.dr_f32) load8f(reinterpret_cast<const float*>(d.dr) + m * d.lddr + c, a);
            else unpack8(ld16<EWL>(dr + m * d.lddr + c), a);
            unpack8(ld16<EWL>(yy + m * d.ldy + c), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) { o[e] = fmaf(A[e], a[e], fmaf(B[e], y[e], Cc[e])); part[0][e] += o[e]; }
            st16<EWS>(dy + m * d.lddy + c, pack8(o));
        }
    }
    if (d.colsum) block_colsum<CGB, 1>(part, red, d.colsum, d.C, blockIdx.x * CGB, 0, 0);
}

// Contiguous form (all three tensors [M][C] with no padding, no column sums): the tensors are one long array of 16-byte
// chunks.  The grid stride is a multiple of the chunks per row, so a thread keeps its channel group (coefficients loaded
// once) and every lane is busy whatever C is (192 channels are 24 chunks: the row-block form leaves 8 of 32 lanes idle).
template <bool DRF32>
__global__ __launch_bounds__(256) void bn_apply_flat_kernel(const dsaspp_bnapply_desc d, const int CG, const int64_t nchunks) {
    const int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
    const int c = (int)(g0 % CG) * 8;
    float A[8], B[8], Cc[8];
    load8f(d.coefA + c, A); load8f(d.coefB + c, B); load8f(d.coefC + c, Cc);
    const uint4* yy = reinterpret_cast<const uint4*>(d.y);
    uint4* dy = reinterpret_cast<uint4*>(d.dy);
#ifndef BNAF_U
#define BNAF_U 8
#endif
#ifndef BNAF_NT
#define BNAF_NT 3                // bit 0: loads, bit 1: stores carry the non-temporal hint
#endif
    constexpr int U = BNAF_U;
    for (int64_t g = g0; g < nchunks; g += U * stride) {
        float a[U][8];
        uint4 yv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t gu = g + u * stride;
            const int64_t gc = gu < nchunks ? gu : g;
            if (DRF32) load8f(reinterpret_cast<const float*>(d.dr) + gc * 8, a[u]);
            else unpack8(ld16<(BNAF_NT & 1) != 0>(reinterpret_cast<const uint4*>(d.dr) + gc), a[u]);
            yv[u] = ld16<(BNAF_NT & 1) != 0>(yy + gc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t gu = g + u * stride;
            float y[8], o[8];
            unpack8(yv[u], y);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(A[e], a[u][e], fmaf(B[e], y[e], Cc[e]));
            if (gu < nchunks) st16<(BNAF_NT & 2) != 0>(dy + gu, pack8(o));
        }
    }
}

// ---------------------------------------------------------------- 2x2 average pooling
template <int CGB>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const dsaspp_pool_desc d) {
    __shared__ float red[256 * 16];
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int c = (blockIdx.x * CGB + cx) * 8;
    const bool act = c < d.C;
    const int Ho = d.H / 2, Wo = d.W / 2;
    const int64_t Mo = (int64_t)d.n_img * Ho * Wo;
    float part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[0][e] = 0.f; part[1][e] = 0.f; }
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    u16* yg = reinterpret_cast<u16*>(d.y);
    if (act) {
        for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < Mo; m += (int64_t)gridDim.y * RB) {
            const int n = (int)(m / (Ho * Wo));
            const int rem = (int)(m - (int64_t)n * Ho * Wo);
            const int ho = rem / Wo, wo = rem - ho * Wo;
            const int64_t p00 = ((int64_t)n * d.H + 2 * ho) * d.W + 2 * wo;
            float a[8], b[8], e0[8], e1[8], o[8];
            unpack8(*reinterpret_cast<const uint4*>(xg + p00 * d.ldx + c), a);
            unpack8(*reinterpret_cast<const uint4*>(xg + (p00 + 1) * d.ldx + c), b);
            unpack8(*reinterpret_cast<const uint4*>(xg + (p00 + d.W) * d.ldx + c), e0);
            unpack8(*reinterpret_cast<const uint4*>(xg + (p00 + d.W + 1) * d.ldx + c), e1);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.25f * ((a[e] + b[e]) + (e0[e] + e1[e]));
            const uint4 ov = pack8(o);
            *reinterpret_cast<uint4*>(yg + m * d.ldy + c) = ov;
            if (d.stat_acc) {
                float g[8];
                unpack8(ov, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) { part[0][e] += g[e]; part[1][e] += g[e] * g[e]; }
            }
        }
    }
    if (d.stat_acc) block_colsum<CGB, 2>(part, red, d.stat_acc, d.C, blockIdx.x * CGB, 2 * d.C, blockIdx.y % DS_R);
}

// dx (full size, d.x) <- dy (half size, d.y) / 4
template <int CGB>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const dsaspp_pool_desc d) {
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int c = (blockIdx.x * CGB + cx) * 8;
    if (c >= d.C) return;
    const int Ho = d.H / 2, Wo = d.W / 2;
    const int64_t Mi = (int64_t)d.n_img * d.H * d.W;
    u16* xg = reinterpret_cast<u16*>(const_cast<void*>(d.x));
    const u16* yg = reinterpret_cast<const u16*>(d.y);
    for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < Mi; m += (int64_t)gridDim.y * RB) {
        const int n = (int)(m / (d.H * d.W));
        const int rem = (int)(m - (int64_t)n * d.H * d.W);
        const int h = rem / d.W, w = rem - h * d.W;
        const int64_t po = ((int64_t)n * Ho + (h >> 1)) * Wo + (w >> 1);
        float a[8];
        unpack8(*reinterpret_cast<const uint4*>(yg + po * d.ldy + c), a);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= 0.25f;
        *reinterpret_cast<uint4*>(xg + m * d.ldx + c) = pack8(a);
    }
}

// ---------------------------------------------------------------- stem im2col (7x7, stride 2, pad 3)
// Column k = (ky*7+kx)*3 + ci.  A block owns IM2COL_PX consecutive output pixels of one output row: the 3 x 7 input row
// segments they draw from are staged in LDS once (contiguous fp32 reads, zero outside the image), and the block writes
// its part of `col` as one contiguous run of 16-byte chunks.  No 64-bit division, 32-bit ones by compile-time constants
// only (a thread-per-chunk gather with two 64-bit divisions per thread took 1.9 ms at 16 x 1024 x 2048).
constexpr int IM2COL_PX = 64;
constexpr int IM2COL_SEG = 2 * IM2COL_PX + 5;
__global__ __launch_bounds__(256) void im2col_kernel(const dsaspp_im2col_desc d, const int wtiles) {
    __shared__ float patch[3 * 7 * IM2COL_SEG];
    const int Ho = d.H / 2, Wo = d.W / 2;
    const int wt = blockIdx.x % wtiles;
    const int rowi = blockIdx.x / wtiles;                  // n * Ho + ho
    const int n = rowi / Ho, ho = rowi - n * Ho;
    const int wo0 = wt * IM2COL_PX;
    const int npx = min(IM2COL_PX, Wo - wo0);
    const int64_t plane = (int64_t)d.H * d.W;
    const float* img = d.x + (int64_t)n * 3 * plane;
    for (int i = threadIdx.x; i < 3 * 7 * IM2COL_SEG; i += 256) {
        const int r = i / IM2COL_SEG, xx = i - r * IM2COL_SEG;     // r = ci * 7 + ky
        const int ci = r / 7, ky = r - ci * 7;
        const int hi = 2 * ho - 3 + ky, wi = 2 * wo0 - 3 + xx;
        float v = 0.f;
        if ((unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W) v = img[ci * plane + (int64_t)hi * d.W + wi];
        patch[i] = v;
    }
    __syncthreads();
    const int nchunk = d.ldcol / 8;
    u16* col = reinterpret_cast<u16*>(d.col) + ((int64_t)rowi * Wo + wo0) * d.ldcol;
    for (int q = threadIdx.x; q < npx * nchunk; q += 256) {
        const int px = q / nchunk, ch = q - px * nchunk;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            const int t = k / 3, ci = k - t * 3;
            const int ky = t / 7, kx = t - ky * 7;
            f[e] = k < 147 ? patch[(ci * 7 + ky) * IM2COL_SEG + 2 * px + kx] : 0.f;
        }
        *reinterpret_cast<uint4*>(col + (int64_t)px * d.ldcol + ch * 8) = pack8(f);
    }
}

// ---------------------------------------------------------------- adjoint of the stem im2col (input gradient)
// One thread per input pixel and all three channels: the taps whose parity matches gather from at most 4 x 4 (ks 7) or
// 2 x 2 (ks 3) output positions, three consecutive column entries each, in a fixed order.
__device__ __forceinline__ float col_ld(const u16* p) { return bf_lo((uint32_t)*p); }
__device__ __forceinline__ float col_ld(const float* p) { return *p; }
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const dsaspp_col2im_desc d, const int pad) {
    const int H = d.H, W = d.W, Ho = H / 2, Wo = W / 2, ks = d.ks;
    const int64_t total = (int64_t)d.n_img * H * W;
    const T* col = reinterpret_cast<const T*>(d.dcol);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int wi = (int)(idx % W);
        const int64_t t = idx / W;
        const int hi = (int)(t % H);
        const int n = (int)(t / H);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int ky = (hi + pad) & 1; ky < ks; ky += 2) {
            const int h2 = hi + pad - ky;
            if (h2 < 0 || (h2 >> 1) >= Ho) continue;
            for (int kx = (wi + pad) & 1; kx < ks; kx += 2) {
                const int w2 = wi + pad - kx;
                if (w2 < 0 || (w2 >> 1) >= Wo) continue;
                const T* p = col + (((int64_t)n * Ho + (h2 >> 1)) * Wo + (w2 >> 1)) * d.ldcol + (ky * ks + kx) * 3;
                a0 += col_ld(p);
                a1 += col_ld(p + 1);
                a2 += col_ld(p + 2);
            }
        }
        const int64_t plane = (int64_t)H * W;
        float* o = d.dx + (int64_t)n * 3 * plane + (int64_t)hi * W + wi;
        if (d.accumulate) {
            o[0] += a0;
            o[plane] += a1;
            o[2 * plane] += a2;
        } else {
            o[0] = a0;
            o[plane] = a1;
            o[2 * plane] = a2;
        }
    }
}

// ---------------------------------------------------------------- BN+ReLU -> maxpool 3x3 stride 2 pad 1
template <int CGB>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const dsaspp_maxpool_desc d) {
    __shared__ float red[256 * 16];
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int c = (blockIdx.x * CGB + cx) * 8;
    const bool act = c < d.C;
    const int Ho = d.H / 2, Wo = d.W / 2;
    const int64_t Mo = (int64_t)d.n_img * Ho * Wo;
    float sc[8], sh[8], part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; part[0][e] = 0.f; part[1][e] = 0.f; }
    if (act) { load8f(d.scale + c, sc); load8f(d.shift + c, sh); }
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    u16* yg = reinterpret_cast<u16*>(d.y);
    if (act) {
        for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < Mo; m += (int64_t)gridDim.y * RB) {
            const int n = (int)(m / (Ho * Wo));
            const int rem = (int)(m - (int64_t)n * Ho * Wo);
            const int ho = rem / Wo, wo = rem - ho * Wo;
            float best[8];
            int arg[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { best[e] = -1.f; arg[e] = 0; }   // relu output >= 0 > -1
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hi = 2 * ho - 1 + t / 3, wi = 2 * wo - 1 + t % 3;
                if ((unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W) {
                    float f[8];
                    unpack8(*reinterpret_cast<const uint4*>(xg + (((int64_t)n * d.H + hi) * d.W + wi) * d.ldx + c), f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float v = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);
                        if (v > best[e]) { best[e] = v; arg[e] = t; }
                    }
                }
            }
            const uint4 ov = pack8(best);
            *reinterpret_cast<uint4*>(yg + m * d.ldy + c) = ov;
            if (d.argmax) {
                uint2 a;
                a.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
                a.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
                *reinterpret_cast<uint2*>(d.argmax + m * d.C + c) = a;
            }
            if (d.stat_acc) {
                float g[8];
                unpack8(ov, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) { part[0][e] += g[e]; part[1][e] += g[e] * g[e]; }
            }
        }
    }
    if (d.stat_acc) block_colsum<CGB, 2>(part, red, d.stat_acc, d.C, blockIdx.x * CGB, 2 * d.C, blockIdx.y % DS_R);
}

// Gather form of the max-pool backward fused with the ReLU mask of norm0 and the S1/S2 sums
// of its BatchNorm backward.  One thread per (conv0-output pixel, 8 channels).
template <int CGB>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const dsaspp_maxpool_desc d) {
    __shared__ float red[256 * 16];
    constexpr int RB = 256 / CGB;
    const int cx = threadIdx.x % CGB, ry = threadIdx.x / CGB;
    const int c = (blockIdx.x * CGB + cx) * 8;
    const bool act = c < d.C;
    const int Ho = d.H / 2, Wo = d.W / 2;
    const int64_t Mi = (int64_t)d.n_img * d.H * d.W;
    float sc[8], sh[8], mu[8], rs[8], part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; part[0][e] = 0.f; part[1][e] = 0.f; }
    if (act) { load8f(d.scale + c, sc); load8f(d.shift + c, sh); load8f(d.mean + c, mu); load8f(d.rstd + c, rs); }
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    const u16* dyg = reinterpret_cast<const u16*>(d.dy);
    u16* drg = reinterpret_cast<u16*>(d.dr);
    if (act) {
        for (int64_t m = (int64_t)blockIdx.y * RB + ry; m < Mi; m += (int64_t)gridDim.y * RB) {
            const int n = (int)(m / (d.H * d.W));
            const int rem = (int)(m - (int64_t)n * d.H * d.W);
            const int h = rem / d.W, w = rem - h * d.W;
            float g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) g[e] = 0.f;
            // windows (ho, wo) that contain (h, w): 2*ho-1 <= h <= 2*ho+1
            const int ho0 = h >> 1, ho1 = (h + 1) >> 1;
            const int wo0 = w >> 1, wo1 = (w + 1) >> 1;
            for (int ho = ho0; ho <= ho1; ++ho) {
                if (ho >= Ho) continue;
                for (int wo = wo0; wo <= wo1; ++wo) {
                    if (wo >= Wo) continue;
                    const int t = (h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1));
                    const int64_t mo = ((int64_t)n * Ho + ho) * Wo + wo;
                    const uint2 a = *reinterpret_cast<const uint2*>(d.argmax + mo * d.C + c);
                    float dyv[8];
                    unpack8(*reinterpret_cast<const uint4*>(dyg + mo * d.lddy + c), dyv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int ae = ((e < 4 ? a.x : a.y) >> (8 * (e & 3))) & 0xff;
                        if (ae == t) g[e] += dyv[e];
                    }
                }
            }
            float x[8];
            unpack8(*reinterpret_cast<const uint4*>(xg + m * d.ldx + c), x);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pre = fmaf(x[e], sc[e], sh[e]);
                if (!(pre > 0.f)) g[e] = 0.f;
                part[0][e] += g[e];
                part[1][e] += g[e] * ((x[e] - mu[e]) * rs[e]);
            }
            if (d.dr_f32) store8f(reinterpret_cast<float*>(d.dr) + m * d.lddr + c, g);
            else *reinterpret_cast<uint4*>(drg + m * d.lddr + c) = pack8(g);
        }
    }
    if (d.s_acc) block_colsum<CGB, 2>(part, red, d.s_acc, d.C, blockIdx.x * CGB, 2 * d.C, blockIdx.y % DS_R);
}

// Flat forms of the two stem max-pool kernels (bf16, fewer than 2^31 (pixel, chunk) pairs): the (pixel, 16-byte chunk) pairs are
// one long index space and the grid stride is a multiple of the chunks per pixel, so that a thread keeps its channel group
// (coefficients loaded once, per-thread column sums) and every lane is busy whatever C is -- 96 channels are 12 chunks, the
// row-block form leaves 4 of 16 lanes idle -- all loads of a pixel issued before the first is used, 32-bit index arithmetic
// (the row-block forms divide 64-bit pixel indices).  Column sums: per-thread partials, LDS atomics once per thread at the end, one
// set of global atomics per block.
template <int NS>
__device__ __forceinline__ void flat_colsum(const float (&part)[NS][8], float* red /* NS * C floats, zeroed */, float* acc, const int C,
                                            const int c, const int rep_stride, const int rep) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(red + s * C + c + e, part[s][e]);
    __syncthreads();
    for (int i = threadIdx.x; i < NS * C; i += 256) atomicAdd(acc + (size_t)rep * rep_stride + i, red[i]);
}

__global__ __launch_bounds__(256) void maxpool_fwd_flat_kernel(const dsaspp_maxpool_desc d, const int CG, const uint32_t npairs) {
    extern __shared__ float red[];
    const int Ho = d.H / 2, Wo = d.W / 2;
    const uint32_t g0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u, dm = stride / CG;
    const int c = (int)(g0 % CG) * 8;
    float sc[8], sh[8], part[2][8];
    load8f(d.scale + c, sc); load8f(d.shift + c, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[0][e] = 0.f; part[1][e] = 0.f; }
    if (d.stat_acc) {
        for (int i = threadIdx.x; i < 2 * d.C; i += 256) red[i] = 0.f;
        __syncthreads();
    }
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    u16* yg = reinterpret_cast<u16*>(d.y);
    const uint32_t Mo = npairs / CG;
    for (uint32_t m = g0 / CG; m < Mo; m += dm) {
        const uint32_t n = m / (uint32_t)(Ho * Wo), rem = m - n * (uint32_t)(Ho * Wo);
        const int ho = (int)(rem / (uint32_t)Wo), wo = (int)(rem - (uint32_t)ho * Wo);
        uint4 v[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hi = 2 * ho - 1 + t / 3, wi = 2 * wo - 1 + t % 3;
            const bool ok = (unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W;
            v[t] = ld16<false>(xg + (((int64_t)n * d.H + (ok ? hi : 2 * ho)) * d.W + (ok ? wi : 2 * wo)) * d.ldx + c);   // outside: the centre again
        }
        float best[8];
        int arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -1.f; arg[e] = 0; }   // relu output >= 0 > -1
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int hi = 2 * ho - 1 + t / 3, wi = 2 * wo - 1 + t % 3;
            if ((unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W) {
                float f[8];
                unpack8(v[t], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = fmaxf(fmaf(f[e], sc[e], sh[e]), 0.f);
                    if (x > best[e]) { best[e] = x; arg[e] = t; }
                }
            }
        }
        const uint4 ov = pack8(best);
        *reinterpret_cast<uint4*>(yg + (int64_t)m * d.ldy + c) = ov;
        if (d.argmax) {
            uint2 a;
            a.x = arg[0] | (arg[1] << 8) | (arg[2] << 16) | (arg[3] << 24);
            a.y = arg[4] | (arg[5] << 8) | (arg[6] << 16) | (arg[7] << 24);
            *reinterpret_cast<uint2*>(d.argmax + (int64_t)m * d.C + c) = a;
        }
        if (d.stat_acc) {
            float g[8];
            unpack8(ov, g);
#pragma unroll
            for (int e = 0; e < 8; ++e) { part[0][e] += g[e]; part[1][e] += g[e] * g[e]; }
        }
    }
    if (d.stat_acc) flat_colsum<2>(part, red, d.stat_acc, d.C, c, 2 * d.C, blockIdx.x % DS_R);
}

__global__ __launch_bounds__(256) void maxpool_bwd_flat_kernel(const dsaspp_maxpool_desc d, const int CG, const uint32_t npairs) {
    extern __shared__ float red[];
    const int Ho = d.H / 2, Wo = d.W / 2;
    const uint32_t g0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u, dm = stride / CG;
    const int c = (int)(g0 % CG) * 8;
    float sc[8], sh[8], mu[8], rs[8], part[2][8];
    load8f(d.scale + c, sc); load8f(d.shift + c, sh); load8f(d.mean + c, mu); load8f(d.rstd + c, rs);
#pragma unroll
    for (int e = 0; e < 8; ++e) { part[0][e] = 0.f; part[1][e] = 0.f; }
    if (d.s_acc) {
        for (int i = threadIdx.x; i < 2 * d.C; i += 256) red[i] = 0.f;
        __syncthreads();
    }
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    const u16* dyg = reinterpret_cast<const u16*>(d.dy);
    u16* drg = reinterpret_cast<u16*>(d.dr);
    const uint32_t Mi = npairs / CG, HW = (uint32_t)(d.H * d.W);
    for (uint32_t m = g0 / CG; m < Mi; m += dm) {
        const uint32_t n = m / HW, rem = m - n * HW;
        const int h = (int)(rem / (uint32_t)d.W), w = (int)(rem - (uint32_t)h * d.W);
        // windows (ho, wo) that contain (h, w): 2*ho-1 <= h <= 2*ho+1; a window that does not exist is replaced by one that
        // does and its contribution dropped, so that the four loads are unconditional
        const int hoA = h >> 1, hoB = (h + 1) >> 1, woA = w >> 1, woB = (w + 1) >> 1;
        const uint4 xv = ld16<true>(xg + (int64_t)m * d.ldx + c);
        uint2 av[4];
        uint4 dv[4];
        int tt[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ho = (k & 2) ? hoB : hoA, wo = (k & 1) ? woB : woA;
            const bool ok = ho < Ho && wo < Wo && !((k & 2) && hoB == hoA) && !((k & 1) && woB == woA);
            const int hoc = ok ? ho : hoA, woc = ok ? wo : woA;
            tt[k] = ok ? (h - (2 * ho - 1)) * 3 + (w - (2 * wo - 1)) : 255;
            const int64_t mo = ((int64_t)n * Ho + hoc) * Wo + woc;
            av[k] = *reinterpret_cast<const uint2*>(d.argmax + mo * d.C + c);
            dv[k] = ld16<false>(dyg + mo * d.lddy + c);
        }
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float dyv[8];
            unpack8(dv[k], dyv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ae = ((e < 4 ? av[k].x : av[k].y) >> (8 * (e & 3))) & 0xff;
                if (ae == tt[k]) g[e] += dyv[e];
            }
        }
        float x[8];
        unpack8(xv, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float pre = fmaf(x[e], sc[e], sh[e]);
            if (!(pre > 0.f)) g[e] = 0.f;
            part[0][e] += g[e];
            part[1][e] += g[e] * ((x[e] - mu[e]) * rs[e]);
        }
        st16<true>(drg + (int64_t)m * d.lddr + c, pack8(g));
    }
    if (d.s_acc) flat_colsum<2>(part, red, d.s_acc, d.C, c, 2 * d.C, blockIdx.x % DS_R);
}

// grid of a flat kernel: a multiple of q = CG / gcd(256, CG) blocks, so that the grid stride is a multiple of CG
inline int flat_blocks(const int CG, const int64_t npairs, const int per_thread, const int cap) {
    int q = CG, r = 256;
    while (r) { const int t = q % r; q = r; r = t; }
    q = CG / q;
    int64_t blocks = (npairs + (int64_t)per_thread * 256 - 1) / ((int64_t)per_thread * 256);
    if (blocks > cap) blocks = cap;
    return (int)((blocks + q - 1) / q * q);
}

inline dim3 row_grid(int C, int64_t M, int cgb) {
    const int cg = (C + 7) / 8;
    const int gx = (cg + cgb - 1) / cgb;
    const int rb = 256 / cgb;
    int64_t gy = (M + rb - 1) / rb;
    const int64_t cap = (4096 + gx - 1) / gx;
    if (gy > cap) gy = cap;
    if (gy < 1) gy = 1;
    return dim3(gx, (unsigned)gy);
}

#define ROW_DISPATCH(KERN, C, M, desc, s)                                                   \
    do {                                                                                    \
        if ((C) <= 64) {                                                                    \
            hipLaunchKernelGGL(KERN<8>, row_grid((C), (M), 8), dim3(256), 0, s, desc);      \
        } else if ((C) <= 128) { /* 96 channels: 12 of 16 lanes busy instead of 12 of 32 */ \
            hipLaunchKernelGGL(KERN<16>, row_grid((C), (M), 16), dim3(256), 0, s, desc);    \
        } else {                                                                            \
            hipLaunchKernelGGL(KERN<32>, row_grid((C), (M), 32), dim3(256), 0, s, desc);    \
        }                                                                                   \
    } while (0)

}  // namespace

extern "C" int dsaspp_bn_finalize(const dsaspp_bn_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->C > 0, "bn_finalize: bad descriptor");
    if (d->use_running) DS_REQUIRE(d->running_mean && d->running_var, "bn_finalize: running stats missing");
    else DS_REQUIRE(d->mean && d->var && d->rstd && d->count > 0.f, "bn_finalize: batch statistic buffers missing");
    if (d->scale) DS_REQUIRE(d->shift != nullptr, "bn_finalize: shift missing");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((d->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d);
    DS_LAUNCH_CHECK("bn_finalize");
    return 0;
}

extern "C" int dsaspp_bn_bwd_finalize(const dsaspp_bn_bwd_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->C > 0 && d->acc && d->scale && d->count > 0.f, "bn_bwd_finalize: bad descriptor");
    if (d->mode == 0) DS_REQUIRE(d->coefA && d->coefB && d->coefC && d->mean && d->rstd, "bn_bwd_finalize: coefficient buffers missing");
    else DS_REQUIRE(d->corr1 && d->corr2, "bn_bwd_finalize: correction buffers missing");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((d->C + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d);
    DS_LAUNCH_CHECK("bn_bwd_finalize");
    return 0;
}

extern "C" int dsaspp_bn_act(const dsaspp_bnact_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->M > 0 && d->C > 0 && d->C % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0, "bn_act: bad descriptor");
    DS_REQUIRE(d->x && d->y, "bn_act: null tensor");
    DS_REQUIRE(d->relu >= 0 && d->relu <= 2, "bn_act: relu must be 0, 1 or 2");
    if (d->res) DS_REQUIRE(d->ldres % 8 == 0 && d->ldres >= d->C && ((uintptr_t)d->res & 15) == 0, "bn_act: bad residual stride/alignment");
    if (d->dtype == 1) return dsaspp_f32_bn_act(*d, (hipStream_t)stream);
    ROW_DISPATCH(bn_act_kernel, d->C, d->M, *d, (hipStream_t)stream);
    DS_LAUNCH_CHECK("bn_act");
    return 0;
}

extern "C" int dsaspp_grad_extract(const dsaspp_gradx_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->M > 0 && d->C > 0 && d->C % 8 == 0 && d->ldg % 8 == 0 && d->lddz % 8 == 0, "grad_extract: bad descriptor");
    DS_REQUIRE(d->g && d->dz, "grad_extract: null tensor");
    if (d->corr1) DS_REQUIRE(d->corr2 && d->xs && d->s_mean && d->s_rstd && d->ldxs % 8 == 0, "grad_extract: correction inputs missing");
    if (d->relu || d->s_acc) DS_REQUIRE(d->yref && d->ldyref % 8 == 0, "grad_extract: yref missing");
    if (d->s_acc) DS_REQUIRE(d->r_mean && d->r_rstd, "grad_extract: r_mean/r_rstd missing");
    if (d->mask) DS_REQUIRE(d->hw > 0 && d->mask_ld % 4 == 0, "grad_extract: mask needs hw");
    if (d->dtype == 1) return dsaspp_f32_grad_extract(*d, (hipStream_t)stream);
    ROW_DISPATCH(grad_extract_kernel, d->C, d->M, *d, (hipStream_t)stream);
    DS_LAUNCH_CHECK("grad_extract");
    return 0;
}

extern "C" int dsaspp_bn_bwd_apply(const dsaspp_bnapply_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->M > 0 && d->C > 0 && d->C % 8 == 0 && d->lddr % 8 == 0 && d->ldy % 8 == 0 && d->lddy % 8 == 0, "bn_bwd_apply: bad descriptor");
    DS_REQUIRE(d->dr && d->y && d->dy && d->coefA && d->coefB && d->coefC, "bn_bwd_apply: null tensor");
    if (d->dtype == 1) return dsaspp_f32_bn_bwd_apply(*d, (hipStream_t)stream);
    if (!d->colsum && d->lddr == d->C && d->ldy == d->C && d->lddy == d->C &&
        (((uintptr_t)d->dr | (uintptr_t)d->y | (uintptr_t)d->dy) & 15) == 0) {
        const int CG = d->C / 8;
        int q = CG, r = 256;                      // q = CG / gcd(256, CG): grid sizes that keep a thread on one channel group
        while (r) { const int t = q % r; q = r; r = t; }
        q = CG / q;
        const int64_t nchunks = d->M * CG;
#ifndef BNAF_BLOCKS
#define BNAF_BLOCKS 2048
#endif
        int64_t blocks = (nchunks + 4 * 256 - 1) / (4 * 256);
        if (blocks > BNAF_BLOCKS) blocks = BNAF_BLOCKS;
        blocks = (blocks + q - 1) / q * q;
        if (d->dr_f32) hipLaunchKernelGGL(bn_apply_flat_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d, CG, nchunks);
        else hipLaunchKernelGGL(bn_apply_flat_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d, CG, nchunks);
        DS_LAUNCH_CHECK("bn_bwd_apply");
        return 0;
    }
    ROW_DISPATCH(bn_apply_kernel, d->C, d->M, *d, (hipStream_t)stream);
    DS_LAUNCH_CHECK("bn_bwd_apply");
    return 0;
}

extern "C" int dsaspp_avgpool2_fwd(const dsaspp_pool_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "avgpool2_fwd: H and W must be even");
    DS_REQUIRE(d->C > 0 && d->C % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->x && d->y, "avgpool2_fwd: bad descriptor");
    if (d->dtype == 1) return dsaspp_f32_avgpool(*d, (hipStream_t)stream);
    const int64_t Mo = (int64_t)d->n_img * (d->H / 2) * (d->W / 2);
    ROW_DISPATCH(avgpool_fwd_kernel, d->C, Mo, *d, (hipStream_t)stream);
    DS_LAUNCH_CHECK("avgpool2_fwd");
    return 0;
}

extern "C" int dsaspp_avgpool2_bwd(const dsaspp_pool_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "avgpool2_bwd: H and W must be even");
    DS_REQUIRE(d->C > 0 && d->C % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->x && d->y, "avgpool2_bwd: bad descriptor");
    const int64_t Mi = (int64_t)d->n_img * d->H * d->W;
    if (d->dtype == 1) return dsaspp_f32_avgpool_bwd(*d, (hipStream_t)stream);
    ROW_DISPATCH(avgpool_bwd_kernel, d->C, Mi, *d, (hipStream_t)stream);
    DS_LAUNCH_CHECK("avgpool2_bwd");
    return 0;
}

extern "C" int dsaspp_stem_im2col(const dsaspp_im2col_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "stem_im2col: H and W must be even");
    DS_REQUIRE(d->x && d->col && d->ldcol >= 152 && d->ldcol % 8 == 0, "stem_im2col: ldcol must be >= 152 and a multiple of 8");
    if (d->dtype == 1) return dsaspp_f32_im2col(*d, (hipStream_t)stream);
    const int wtiles = (d->W / 2 + IM2COL_PX - 1) / IM2COL_PX;
    const int64_t blocks = (int64_t)d->n_img * (d->H / 2) * wtiles;
    DS_REQUIRE(blocks < (1ll << 31), "stem_im2col: too many output row segments (%lld)", (long long)blocks);
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d, wtiles);
    DS_LAUNCH_CHECK("stem_im2col");
    return 0;
}

extern "C" int dsaspp_stem_col2im(const dsaspp_col2im_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "stem_col2im: H and W must be even");
    DS_REQUIRE(d->ks == 7 || d->ks == 3, "stem_col2im: ks must be 7 or 3 (got %d)", d->ks);
    DS_REQUIRE(d->dcol && d->dx && d->ldcol >= d->ks * d->ks * 3, "stem_col2im: ldcol must be at least ks*ks*3 (got %d)", d->ldcol);
    DS_REQUIRE(d->dtype == 0 || d->dtype == 1, "stem_col2im: dtype must be 0 (bf16) or 1 (fp32)");
    const int64_t total = (int64_t)d->n_img * d->H * d->W;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const int pad = d->ks == 7 ? 3 : 1;
    if (d->dtype == 1) hipLaunchKernelGGL(col2im_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d, pad);
    else hipLaunchKernelGGL(col2im_kernel<u16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d, pad);
    DS_LAUNCH_CHECK("stem_col2im");
    return 0;
}

extern "C" int dsaspp_maxpool3s2_fwd(const dsaspp_maxpool_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "maxpool3s2_fwd: H and W must be even");
    DS_REQUIRE(d->C > 0 && d->C % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && d->x && d->y && d->scale && d->shift, "maxpool3s2_fwd: bad descriptor");
    if (d->dtype == 1) return dsaspp_f32_maxpool(*d, (hipStream_t)stream);
    const int64_t Mo = (int64_t)d->n_img * (d->H / 2) * (d->W / 2);
    if (Mo * (d->C / 8) < (1ll << 31) && (int64_t)d->n_img * d->H * d->W < (1ll << 31)) {
        const int CG = d->C / 8;
        const int64_t npairs = Mo * CG;
        hipLaunchKernelGGL(maxpool_fwd_flat_kernel, dim3(flat_blocks(CG, npairs, 2, 4096)), dim3(256), (size_t)2 * d->C * sizeof(float),
                           (hipStream_t)stream, *d, CG, (uint32_t)npairs);
        DS_LAUNCH_CHECK("maxpool3s2_fwd");
        return 0;
    }
    ROW_DISPATCH(maxpool_fwd_kernel, d->C, Mo, *d, (hipStream_t)stream);
    DS_LAUNCH_CHECK("maxpool3s2_fwd");
    return 0;
}

extern "C" int dsaspp_maxpool3s2_bwd(const dsaspp_maxpool_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "maxpool3s2_bwd: H and W must be even");
    DS_REQUIRE(d->C > 0 && d->C % 8 == 0 && d->ldx % 8 == 0 && d->lddy % 8 == 0 && d->lddr % 8 == 0, "maxpool3s2_bwd: bad strides");
    DS_REQUIRE(d->x && d->dy && d->dr && d->argmax && d->scale && d->shift && d->mean && d->rstd, "maxpool3s2_bwd: null tensor");
    const int64_t Mi = (int64_t)d->n_img * d->H * d->W;
    if (d->dtype == 1) return dsaspp_f32_maxpool_bwd(*d, (hipStream_t)stream);
    if (!d->dr_f32 && Mi * (d->C / 8) < (1ll << 31)) {
        const int CG = d->C / 8;
        const int64_t npairs = Mi * CG;
        hipLaunchKernelGGL(maxpool_bwd_flat_kernel, dim3(flat_blocks(CG, npairs, 4, 4096)), dim3(256), (size_t)2 * d->C * sizeof(float),
                           (hipStream_t)stream, *d, CG, (uint32_t)npairs);
        DS_LAUNCH_CHECK("maxpool3s2_bwd");
        return 0;
    }
    ROW_DISPATCH(maxpool_bwd_kernel, d->C, Mi, *d, (hipStream_t)stream);
    DS_LAUNCH_CHECK("maxpool3s2_bwd");
    return 0;
}
