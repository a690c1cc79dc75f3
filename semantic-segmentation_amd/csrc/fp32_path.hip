// fp32 form of the eval-mode forward kernels (descriptor dtype = 1), gfx950.
//
// Activations and weights are fp32 in memory, products run on the fp32 MFMA (16x16x4) with fp32
// accumulation, so the only difference to an fp32 reference is summation order.  This path exists for
// checking the network against an fp32 reference (1e-4 relative on the logits); it is a plain LDS-tiled
// implicit GEMM without the tuning of the bf16 kernels, and covers forward inference only.
#include "common.h"

namespace {

constexpr int FP = 64, FQ = 64, FK = 16;     // block tile: 64 pixels x 64 output channels, K-step 16
constexpr int LDT = FK + 1;                   // padded LDS row (floats)

__device__ __forceinline__ float act_clamp(float v, int relu) {
    if (relu) v = fmaxf(v, 0.f);
    if (relu == 2) v = fminf(v, 6.f);
    return v;
}

// 4 waves as 2 x 2; a wave owns 32 pixels x 32 channels = 2 x 2 MFMA tiles.  Weights are the A operand
// (rows = channels), pixels the B operand: a lane's four accumulator values are 4 consecutive channels of
// one pixel, stored as one float4.
__global__ __launch_bounds__(256) void conv_f32_kernel(const dsaspp_conv_desc d, const int M, const int nQ) {
    __shared__ float As[FP * LDT];
    __shared__ float Ws[FQ * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pt = blockIdx.x / nQ, qt = blockIdx.x - pt * nQ;
    const int p_base = pt * FP, q_base = qt * FQ;
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout, taps = d.taps, dil = d.dil, hw = H * W;
    const int Ktot = taps * cin;
    const float* xg = reinterpret_cast<const float*>(d.x);
    const float* wg = reinterpret_cast<const float*>(d.w);
    // staging coordinates: row = tid / 4 (pixel or output channel), 4-float chunk = tid % 4
    const int srow = tid >> 2, sch = (tid & 3) * 4;
    const int m = p_base + srow;
    const bool m_ok = m < M;
    const int mm = m_ok ? m : 0;
    const int n_img = mm / hw, rem = mm - n_img * hw, ph = rem / W, pw = rem - ph * W;
    const bool q_ok = q_base + srow < cout;
    const float* wrow = wg + (int64_t)(q_ok ? q_base + srow : 0) * Ktot;

    const int fr = lane & 15, fk = lane >> 4;
    const int wp = wave & 1, wq = wave >> 1;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tap = 0; tap < taps; ++tap) {
        int dy = 0, dx = 0;
        if (taps == 9) {
            const int ty = tap / 3;
            dy = (ty - 1) * dil;
            dx = (tap - ty * 3 - 1) * dil;
        }
        const int hh = ph + dy, ww = pw + dx;
        const bool pix_ok = m_ok && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
        const float* xrow = xg + (pix_ok ? (((int64_t)n_img * H + hh) * W + ww) * d.ldx : 0);
        for (int c0 = 0; c0 < cin; c0 += FK) {
            const int c = c0 + sch;
            const bool c_ok = c < cin;          // cin is a multiple of 8: a chunk of 4 is all in or all out
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), w4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix_ok && c_ok) {
                a = *reinterpret_cast<const float4*>(xrow + c);
                if (d.in_scale) {
                    const float4 sc = *reinterpret_cast<const float4*>(d.in_scale + c);
                    const float4 sh = *reinterpret_cast<const float4*>(d.in_shift + c);
                    a.x = fmaf(a.x, sc.x, sh.x); a.y = fmaf(a.y, sc.y, sh.y); a.z = fmaf(a.z, sc.z, sh.z); a.w = fmaf(a.w, sc.w, sh.w);
                }
                a.x = act_clamp(a.x, d.in_relu); a.y = act_clamp(a.y, d.in_relu);
                a.z = act_clamp(a.z, d.in_relu); a.w = act_clamp(a.w, d.in_relu);
                if (d.in_mask) {
                    const float4 mk = *reinterpret_cast<const float4*>(d.in_mask + (int64_t)n_img * d.in_mask_ld + c);
                    a.x *= mk.x; a.y *= mk.y; a.z *= mk.z; a.w *= mk.w;
                }
            }
            if (q_ok && c_ok) w4 = *reinterpret_cast<const float4*>(wrow + tap * cin + c);
            __syncthreads();                    // previous K-step's fragment reads are done
            As[srow * LDT + sch + 0] = a.x; As[srow * LDT + sch + 1] = a.y; As[srow * LDT + sch + 2] = a.z; As[srow * LDT + sch + 3] = a.w;
            Ws[srow * LDT + sch + 0] = w4.x; Ws[srow * LDT + sch + 1] = w4.y; Ws[srow * LDT + sch + 2] = w4.z; Ws[srow * LDT + sch + 3] = w4.w;
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < FK / 4; ++kk) {
                float wf[2], af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[i] = Ws[(wq * 32 + i * 16 + fr) * LDT + kk * 4 + fk];
#pragma unroll
                for (int j = 0; j < 2; ++j) af[j] = As[(wp * 32 + j * 16 + fr) * LDT + kk * 4 + fk];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], af[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    float* yg = reinterpret_cast<float*>(d.y);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = q_base + wq * 32 + i * 16 + fk * 4;
        if (q >= cout) continue;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.bias) b = *reinterpret_cast<const float4*>(d.bias + q);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = p_base + wp * 32 + j * 16 + fr;
            if (p >= M) continue;
            float4 v = make_float4(acc[i][j][0] + b.x, acc[i][j][1] + b.y, acc[i][j][2] + b.z, acc[i][j][3] + b.w);
            if (d.out_mask) {
                const float4 mk = *reinterpret_cast<const float4*>(d.out_mask + (int64_t)(p / hw) * d.out_mask_ld + q);
                v.x *= mk.x; v.y *= mk.y; v.z *= mk.z; v.w *= mk.w;
            }
            *reinterpret_cast<float4*>(yg + (int64_t)p * d.ldy + q) = v;
        }
    }
}

// ---------------------------------------------------------------- row-wise fp32 kernels: a lane owns 4 channels of a pixel
__global__ __launch_bounds__(256) void bn_act_f32_kernel(const dsaspp_bnact_desc d, const int cg, const int64_t total) {
    const float* xg = reinterpret_cast<const float*>(d.x);
    const float* rg = reinterpret_cast<const float*>(d.res);
    float* yg = reinterpret_cast<float*>(d.y);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / cg;
        const int c = (int)(idx - m * cg) * 4;
        float4 v = *reinterpret_cast<const float4*>(xg + m * d.ldx + c);
        if (d.scale) {
            const float4 sc = *reinterpret_cast<const float4*>(d.scale + c), sh = *reinterpret_cast<const float4*>(d.shift + c);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
        if (rg) {
            const float4 r = *reinterpret_cast<const float4*>(rg + m * d.ldres + c);
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        v.x = act_clamp(v.x, d.relu); v.y = act_clamp(v.y, d.relu); v.z = act_clamp(v.z, d.relu); v.w = act_clamp(v.w, d.relu);
        *reinterpret_cast<float4*>(yg + m * d.ldy + c) = v;
    }
}

__global__ __launch_bounds__(256) void avgpool_f32_kernel(const dsaspp_pool_desc d, const int cg, const int64_t total) {
    const int Ho = d.H / 2, Wo = d.W / 2;
    const float* xg = reinterpret_cast<const float*>(d.x);
    float* yg = reinterpret_cast<float*>(d.y);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / cg;
        const int c = (int)(idx - m * cg) * 4;
        const int n = (int)(m / (Ho * Wo));
        const int rem = (int)(m - (int64_t)n * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const int64_t p00 = ((int64_t)n * d.H + 2 * ho) * d.W + 2 * wo;
        const float4 a = *reinterpret_cast<const float4*>(xg + p00 * d.ldx + c);
        const float4 b = *reinterpret_cast<const float4*>(xg + (p00 + 1) * d.ldx + c);
        const float4 e = *reinterpret_cast<const float4*>(xg + (p00 + d.W) * d.ldx + c);
        const float4 f = *reinterpret_cast<const float4*>(xg + (p00 + d.W + 1) * d.ldx + c);
        *reinterpret_cast<float4*>(yg + m * d.ldy + c) =
            make_float4(0.25f * ((a.x + b.x) + (e.x + f.x)), 0.25f * ((a.y + b.y) + (e.y + f.y)),
                        0.25f * ((a.z + b.z) + (e.z + f.z)), 0.25f * ((a.w + b.w) + (e.w + f.w)));
    }
}

__global__ __launch_bounds__(256) void im2col_f32_kernel(const dsaspp_im2col_desc d, const int64_t total) {
    const int Ho = d.H / 2, Wo = d.W / 2, nchunk = d.ldcol / 4;
    const int64_t plane = (int64_t)d.H * d.W;
    float* col = reinterpret_cast<float*>(d.col);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int ch = (int)(idx % nchunk);
        const int64_t m = idx / nchunk;
        const int n = (int)(m / (Ho * Wo));
        const int rem = (int)(m - (int64_t)n * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = ch * 4 + e;
            float v = 0.f;
            if (k < 147) {
                const int t = k / 3, ci = k - t * 3;
                const int ky = t / 7, kx = t - ky * 7;
                const int hi = 2 * ho - 3 + ky, wi = 2 * wo - 3 + kx;
                if ((unsigned)hi < (unsigned)d.H && (unsigned)wi < (unsigned)d.W)
                    v = d.x[((int64_t)n * 3 + ci) * plane + (int64_t)hi * d.W + wi];
            }
            f[e] = v;
        }
        *reinterpret_cast<float4*>(col + m * d.ldcol + ch * 4) = make_float4(f[0], f[1], f[2], f[3]);
    }
}

__global__ __launch_bounds__(256) void maxpool_f32_kernel(const dsaspp_maxpool_desc d, const int cg, const int64_t total) {
    const int Ho = d.H / 2, Wo = d.W / 2;
    const float* xg = reinterpret_cast<const float*>(d.x);
    float* yg = reinterpret_cast<float*>(d.y);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / cg;
        const int c = (int)(idx - m * cg) * 4;
        const int n = (int)(m / (Ho * Wo));
        const int rem = (int)(m - (int64_t)n * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const float4 sc = *reinterpret_cast<const float4*>(d.scale + c), sh = *reinterpret_cast<const float4*>(d.shift + c);
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f);     // ReLU output is >= 0 and every window holds a valid pixel
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hi = 2 * ho - 1 + ky;
            if ((unsigned)hi >= (unsigned)d.H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int wi = 2 * wo - 1 + kx;
                if ((unsigned)wi >= (unsigned)d.W) continue;
                const float4 v = *reinterpret_cast<const float4*>(xg + (((int64_t)n * d.H + hi) * d.W + wi) * d.ldx + c);
                best.x = fmaxf(best.x, fmaf(v.x, sc.x, sh.x)); best.y = fmaxf(best.y, fmaf(v.y, sc.y, sh.y));
                best.z = fmaxf(best.z, fmaf(v.z, sc.z, sh.z)); best.w = fmaxf(best.w, fmaf(v.w, sc.w, sh.w));
            }
        }
        *reinterpret_cast<float4*>(yg + m * d.ldy + c) = best;
    }
}

__global__ __launch_bounds__(256) void stem3_f32_kernel(const dsaspp_stem3_desc d, const int cg, const int64_t total) {
    const int Ho = d.H / 2, Wo = d.W / 2;
    const int64_t plane = (int64_t)d.H * d.W;
    const float* wg = reinterpret_cast<const float*>(d.w);
    float* yg = reinterpret_cast<float*>(d.y);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / cg;
        const int c = (int)(idx - m * cg) * 4;
        const int n = (int)(m / (Ho * Wo));
        const int rem = (int)(m - (int64_t)n * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < 3; ++ky) {
            const int hi = 2 * ho - 1 + ky;
            if ((unsigned)hi >= (unsigned)d.H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int wi = 2 * wo - 1 + kx;
                if ((unsigned)wi >= (unsigned)d.W) continue;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float v = d.x[((int64_t)n * 3 + ci) * plane + (int64_t)hi * d.W + wi];
                    const int k = (ky * 3 + kx) * 3 + ci;
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(v, wg[(c + e) * 27 + k], acc[e]);
                }
            }
        }
        *reinterpret_cast<float4*>(yg + m * d.ldy + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

__global__ __launch_bounds__(256) void dwconv_f32_kernel(const dsaspp_dw_desc d, const int Ho, const int Wo, const int cg,
                                                         const int64_t total) {
    const float* xg = reinterpret_cast<const float*>(d.x);
    const float* wg = reinterpret_cast<const float*>(d.w);
    float* yg = reinterpret_cast<float*>(d.y);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / cg;
        const int c = (int)(idx - m * cg) * 4;
        const int n = (int)(m / (Ho * Wo));
        const int rem = (int)(m - (int64_t)n * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (d.in_scale) { sc = *reinterpret_cast<const float4*>(d.in_scale + c); sh = *reinterpret_cast<const float4*>(d.in_shift + c); }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hi = ho * d.stride + (ky - 1) * d.dil;
            if ((unsigned)hi >= (unsigned)d.H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int wi = wo * d.stride + (kx - 1) * d.dil;
                if ((unsigned)wi >= (unsigned)d.W) continue;
                const float4 v = *reinterpret_cast<const float4*>(xg + (((int64_t)n * d.H + hi) * d.W + wi) * d.ldx + c);
                const float a[4] = {act_clamp(fmaf(v.x, sc.x, sh.x), d.in_relu), act_clamp(fmaf(v.y, sc.y, sh.y), d.in_relu),
                                    act_clamp(fmaf(v.z, sc.z, sh.z), d.in_relu), act_clamp(fmaf(v.w, sc.w, sh.w), d.in_relu)};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(a[e], wg[(c + e) * 9 + ky * 3 + kx], acc[e]);
            }
        }
        *reinterpret_cast<float4*>(yg + m * d.ldy + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

unsigned blocks_for(int64_t total) {
    int64_t nb = (total + 255) / 256;
    if (nb > 32768) nb = 32768;
    if (nb < 1) nb = 1;
    return (unsigned)nb;
}

}  // namespace

int dsaspp_f32_conv(const dsaspp_conv_desc& d, hipStream_t s) {
    DS_REQUIRE(!d.g && !d.epi_relu && !d.s_acc && !d.stat_acc && d.n_src <= 1,
               "conv_fwd(fp32): only the eval-mode forward form is implemented (no statistics, no gradient epilogues)");
    DS_REQUIRE(d.y && d.ldy % 4 == 0 && d.ldy >= d.cout && ((uintptr_t)d.y & 15) == 0, "conv_fwd(fp32): bad output stride/alignment");
    DS_REQUIRE(d.ldx % 4 == 0, "conv_fwd(fp32): ldx must be a multiple of 4");
    const int M = d.n_img * d.H * d.W;
    const int nP = (M + FP - 1) / FP, nQ = (d.cout + FQ - 1) / FQ;
    hipLaunchKernelGGL(conv_f32_kernel, dim3(nP * nQ), dim3(256), 0, s, d, M, nQ);
    DS_LAUNCH_CHECK("conv_fwd(fp32)");
    return 0;
}

int dsaspp_f32_bn_act(const dsaspp_bnact_desc& d, hipStream_t s) {
    DS_REQUIRE(!d.stat_acc, "bn_act(fp32): statistics are not implemented");
    const int cg = d.C / 4;
    const int64_t total = d.M * cg;
    hipLaunchKernelGGL(bn_act_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, d, cg, total);
    DS_LAUNCH_CHECK("bn_act(fp32)");
    return 0;
}

int dsaspp_f32_avgpool(const dsaspp_pool_desc& d, hipStream_t s) {
    DS_REQUIRE(!d.stat_acc, "avgpool2_fwd(fp32): statistics are not implemented");
    const int cg = d.C / 4;
    const int64_t total = (int64_t)d.n_img * (d.H / 2) * (d.W / 2) * cg;
    hipLaunchKernelGGL(avgpool_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, d, cg, total);
    DS_LAUNCH_CHECK("avgpool2_fwd(fp32)");
    return 0;
}

int dsaspp_f32_im2col(const dsaspp_im2col_desc& d, hipStream_t s) {
    const int64_t total = (int64_t)d.n_img * (d.H / 2) * (d.W / 2) * (d.ldcol / 4);
    hipLaunchKernelGGL(im2col_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, d, total);
    DS_LAUNCH_CHECK("stem_im2col(fp32)");
    return 0;
}

int dsaspp_f32_maxpool(const dsaspp_maxpool_desc& d, hipStream_t s) {
    DS_REQUIRE(!d.stat_acc && !d.argmax, "maxpool3s2_fwd(fp32): statistics / argmax are not implemented");
    const int cg = d.C / 4;
    const int64_t total = (int64_t)d.n_img * (d.H / 2) * (d.W / 2) * cg;
    hipLaunchKernelGGL(maxpool_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, d, cg, total);
    DS_LAUNCH_CHECK("maxpool3s2_fwd(fp32)");
    return 0;
}

int dsaspp_f32_stem3(const dsaspp_stem3_desc& d, hipStream_t s) {
    const int cg = d.cout / 4;
    const int64_t total = (int64_t)d.n_img * (d.H / 2) * (d.W / 2) * cg;
    hipLaunchKernelGGL(stem3_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, d, cg, total);
    DS_LAUNCH_CHECK("stem3x3s2(fp32)");
    return 0;
}

int dsaspp_f32_dwconv(const dsaspp_dw_desc& d, hipStream_t s) {
    const int Ho = (d.H - 1) / d.stride + 1, Wo = (d.W - 1) / d.stride + 1, cg = d.C / 4;
    const int64_t total = (int64_t)d.n_img * Ho * Wo * cg;
    hipLaunchKernelGGL(dwconv_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, d, Ho, Wo, cg, total);
    DS_LAUNCH_CHECK("dwconv3x3(fp32)");
    return 0;
}
