// fp32 form of the kernels (descriptor dtype = 1), gfx950.
//
// Activations and weights are fp32 in memory, products run on the fp32 MFMA (16x16x4) or plain fp32 FMAs with
// fp32 accumulation, so the only difference to an fp32 reference is summation order.  This path exists for
// checking the network against an fp32 reference (1e-4 relative on the logits, 1e-3 on the gradients); it is
// made of plain LDS-tiled kernels without the tuning of the bf16 ones.  It covers the forward pass of both
// trunks and the training pass of the DenseNet trunks: the convolution has the full epilogue of the bf16 kernels
// (reference-masked data gradients, BatchNorm-backward sums, gradient-slab accumulation, output statistics).
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
    // ---- epilogue (the one documented at dsaspp_conv_desc): a lane holds channels q..q+3 of pixel p
    float* yg = reinterpret_cast<float*>(d.y);
    float* gg = reinterpret_cast<float*>(d.g);
    const float* rg = reinterpret_cast<const float*>(d.ref);
    const int rep = blockIdx.x % DS_R;
    const int sld = d.s_ld > 0 ? d.s_ld : cout;
    const bool need_ref = d.epi_relu || d.s_acc;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = q_base + wq * 32 + i * 16 + fk * 4;
        const bool q_in = q < cout;                   // cout is a multiple of 4 here: all four channels in or out
        const int qc = q_in ? q : 0;
        float b[4] = {0.f, 0.f, 0.f, 0.f}, es[4] = {1.f, 1.f, 1.f, 1.f}, eh[4] = {0.f, 0.f, 0.f, 0.f};
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (d.bias) b[e] = d.bias[qc + e];
            if (d.e_scale) { es[e] = d.e_scale[qc + e]; eh[e] = d.e_shift[qc + e]; }
            if (d.s_acc) { mu[e] = d.e_mean[qc + e]; rs[e] = d.e_rstd[qc + e]; }
        }
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = p_base + wp * 32 + j * 16 + fr;
            const bool ok = q_in && p < M;
            const int64_t pc = ok ? p : 0;
            float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b[e];
            if (d.out_mask) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= d.out_mask[(pc / hw) * d.out_mask_ld + qc + e];
            }
            if (need_ref) {
                const float4 r4 = *reinterpret_cast<const float4*>(rg + pc * d.ldref + qc);
                r[0] = r4.x; r[1] = r4.y; r[2] = r4.z; r[3] = r4.w;
            }
            if (d.epi_relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pre = d.e_scale ? fmaf(r[e], es[e], eh[e]) : r[e];
                    if (!(pre > 0.f) || (d.epi_relu == 2 && pre >= 6.f)) v[e] = 0.f;
                }
            }
            if (!ok) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = 0.f;
            }
            if (d.s_acc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], (r[e] - mu[e]) * rs[e], s2[e]); }
            }
            if (gg) {
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = d.e_scale ? v[e] * es[e] : v[e];
                if (ok) {
                    float4* gp = reinterpret_cast<float4*>(gg + pc * d.ldg + qc);
                    float4 o = make_float4(t[0], t[1], t[2], t[3]);
                    if (d.g_accumulate) { const float4 old = *gp; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                    *gp = o;
                }
            } else {
                if (ok) *reinterpret_cast<float4*>(yg + pc * d.ldy + qc) = make_float4(v[0], v[1], v[2], v[3]);
                if (d.stat_acc) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
                }
            }
        }
        // per-channel sums: over the 16 pixels of the DPP row, then one atomic per channel and row of lanes
        float* sums = d.s_acc ? d.s_acc : (gg ? nullptr : d.stat_acc);
        const int half = d.s_acc ? sld : cout;
        if (sums) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a1 = row16_sum(s1[e]), a2 = row16_sum(s2[e]);
                if (fr == 0 && q_in) {
                    atomicAdd(sums + (size_t)rep * 2 * half + q + e, a1);
                    atomicAdd(sums + (size_t)rep * 2 * half + half + q + e, a2);
                }
            }
        }
    }
}

// ---------------------------------------------------------------- weight gradient, fp32
// dw[q][tap][c] += sum_m dz[m][q] * A[m + off(tap)][c]; a block owns 64 x 64 outputs of one tap over a pixel range,
// a thread a 4 x 4 patch; tiles are staged in [pixel][channel] layout and read as float4 broadcasts.
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const dsaspp_wgrad_desc d, const int M, const int nQ, const int nC,
                                                        const int chunk) {
    __shared__ float Zs[64 * 64];
    __shared__ float Xs[64 * 64];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    const int ct = bid % nC; bid /= nC;
    const int qt = bid % nQ; bid /= nQ;
    const int tap = bid % d.taps;
    const int split = bid / d.taps;
    const int q_base = qt * 64, c_base = ct * 64;
    const int H = d.H, W = d.W, hw = H * W, cin = d.cin, cout = d.cout;
    const int m_begin = split * chunk, m_end = min(M, m_begin + chunk);
    int dy = 0, dx = 0;
    if (d.taps == 9) {
        const int ty = tap / 3;
        dy = (ty - 1) * d.dil;
        dx = (tap - ty * 3 - 1) * d.dil;
    }
    const float* xg = reinterpret_cast<const float*>(d.x);
    const float* zg = reinterpret_cast<const float*>(d.dz);
    const int srow = tid >> 4, sch = (tid & 15) * 4;        // staging: 16 rows per pass, float4 column
    const int tq = tid >> 4, tc = tid & 15;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int m0 = m_begin; m0 < m_end; m0 += 64) {
        __syncthreads();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = srow + 16 * pass;
            const int m = m0 + row;
            float4 z = make_float4(0.f, 0.f, 0.f, 0.f), a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m < m_end) {
                if (q_base + sch < cout) z = *reinterpret_cast<const float4*>(zg + (int64_t)m * d.lddz + q_base + sch);
                const int n = m / hw, rem = m - n * hw, ph = rem / W, pw = rem - ph * W;
                const int hh = ph + dy, ww = pw + dx;
                const int c = c_base + sch;
                if (c < cin && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) {
                    a = *reinterpret_cast<const float4*>(xg + (((int64_t)n * H + hh) * W + ww) * d.ldx + c);
                    if (d.in_scale) {
                        const float4 sc = *reinterpret_cast<const float4*>(d.in_scale + c);
                        const float4 sh = *reinterpret_cast<const float4*>(d.in_shift + c);
                        a.x = fmaf(a.x, sc.x, sh.x); a.y = fmaf(a.y, sc.y, sh.y); a.z = fmaf(a.z, sc.z, sh.z); a.w = fmaf(a.w, sc.w, sh.w);
                    }
                    a.x = act_clamp(a.x, d.in_relu); a.y = act_clamp(a.y, d.in_relu);
                    a.z = act_clamp(a.z, d.in_relu); a.w = act_clamp(a.w, d.in_relu);
                    if (d.in_mask) {
                        const float4 mk = *reinterpret_cast<const float4*>(d.in_mask + (int64_t)n * d.in_mask_ld + c);
                        a.x *= mk.x; a.y *= mk.y; a.z *= mk.z; a.w *= mk.w;
                    }
                }
            }
            *reinterpret_cast<float4*>(Zs + row * 64 + sch) = z;
            *reinterpret_cast<float4*>(Xs + row * 64 + sch) = a;
        }
        __syncthreads();
#pragma unroll 8
        for (int p = 0; p < 64; ++p) {
            const float4 z = *reinterpret_cast<const float4*>(Zs + p * 64 + tq * 4);
            const float4 x = *reinterpret_cast<const float4*>(Xs + p * 64 + tc * 4);
            const float zz[4] = {z.x, z.y, z.z, z.w}, xx[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(zz[i], xx[j], acc[i][j]);
        }
    }
    const int Ktot = d.taps * cin;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q_base + tq * 4 + i;
        if (q >= cout) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c_base + tc * 4 + j;
            if (c < cin) atomicAdd(d.dw + (int64_t)q * Ktot + (int64_t)tap * cin + c, acc[i][j]);
        }
    }
}

// ---------------------------------------------------------------- row-wise fp32 kernels: a lane owns 4 channels of a pixel
// statistics of a stored float4 (sum, sum of squares) into replica row `rep` of an accumulator [R][2*C]: one atomic per
// element -- this path is for checking, not for speed
__device__ __forceinline__ void stat_add4(float* acc, const int C, const int rep, const int c, const float4 v) {
    float* a = acc + (size_t)rep * 2 * C + c;
    atomicAdd(a + 0, v.x); atomicAdd(a + 1, v.y); atomicAdd(a + 2, v.z); atomicAdd(a + 3, v.w);
    atomicAdd(a + C + 0, v.x * v.x); atomicAdd(a + C + 1, v.y * v.y); atomicAdd(a + C + 2, v.z * v.z); atomicAdd(a + C + 3, v.w * v.w);
}

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
        if (d.stat_acc) stat_add4(d.stat_acc, d.C, (int)(m % DS_R), c, v);
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
        const float4 o = make_float4(0.25f * ((a.x + b.x) + (e.x + f.x)), 0.25f * ((a.y + b.y) + (e.y + f.y)),
                                     0.25f * ((a.z + b.z) + (e.z + f.z)), 0.25f * ((a.w + b.w) + (e.w + f.w)));
        *reinterpret_cast<float4*>(yg + m * d.ldy + c) = o;
        if (d.stat_acc) stat_add4(d.stat_acc, d.C, (int)(m % DS_R), c, o);
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
        // first maximum in window scan order, as F.max_pool2d picks it (ties are common: ReLU zeros)
        float best[4] = {-3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f};
        int pos[4] = {0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hi = 2 * ho - 1 + ky;
            if ((unsigned)hi >= (unsigned)d.H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int wi = 2 * wo - 1 + kx;
                if ((unsigned)wi >= (unsigned)d.W) continue;
                const float4 v = *reinterpret_cast<const float4*>(xg + (((int64_t)n * d.H + hi) * d.W + wi) * d.ldx + c);
                const float a[4] = {fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f),
                                    fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), fmaxf(fmaf(v.w, sc.w, sh.w), 0.f)};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (a[e] > best[e]) { best[e] = a[e]; pos[e] = ky * 3 + kx; }
            }
        }
        const float4 o = make_float4(best[0], best[1], best[2], best[3]);
        *reinterpret_cast<float4*>(yg + m * d.ldy + c) = o;
        if (d.argmax) {
#pragma unroll
            for (int e = 0; e < 4; ++e) d.argmax[m * d.C + c + e] = (uint8_t)pos[e];
        }
        if (d.stat_acc) stat_add4(d.stat_acc, d.C, (int)(m % DS_R), c, o);
    }
}

// ---------------------------------------------------------------- training-side row-wise kernels, fp32
// Thread mapping: a thread owns one group of 4 channels and walks pixel rows with a fixed stride, so per-channel sums
// are per-thread partials that leave as one atomic each.
struct RowWalk {
    int c;             // first channel of this thread's group
    int64_t r0, step;  // first row and row stride
    bool act;
};
__device__ __forceinline__ RowWalk row_walk(const int C) {
    const int cg = C / 4;
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    RowWalk w;
    w.step = ((int64_t)gridDim.x * 256) / cg;
    w.r0 = gtid / cg;
    w.c = (int)(gtid - w.r0 * cg) * 4;
    w.act = w.r0 < w.step;
    return w;
}
__device__ __forceinline__ void ld4(const float* p, float (&f)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}

__global__ __launch_bounds__(256) void grad_extract_f32_kernel(const dsaspp_gradx_desc d) {
    const RowWalk w = row_walk(d.C);
    if (!w.act) return;
    const int c = w.c;
    const float* g = reinterpret_cast<const float*>(d.g);
    const float* xs = reinterpret_cast<const float*>(d.xs);
    const float* yr = reinterpret_cast<const float*>(d.yref);
    float* dz = reinterpret_cast<float*>(d.dz);
    float k1[4] = {0.f, 0.f, 0.f, 0.f}, k2[4] = {0.f, 0.f, 0.f, 0.f}, sm[4] = {0.f, 0.f, 0.f, 0.f}, sr[4] = {1.f, 1.f, 1.f, 1.f};
    float rs[4] = {1.f, 1.f, 1.f, 1.f}, rh[4] = {0.f, 0.f, 0.f, 0.f}, rm[4] = {0.f, 0.f, 0.f, 0.f}, rr[4] = {1.f, 1.f, 1.f, 1.f};
    if (d.corr1) { ld4(d.corr1 + c, k1); ld4(d.corr2 + c, k2); ld4(d.s_mean + c, sm); ld4(d.s_rstd + c, sr); }
    if (d.r_scale) { ld4(d.r_scale + c, rs); ld4(d.r_shift + c, rh); }
    if (d.s_acc) { ld4(d.r_mean + c, rm); ld4(d.r_rstd + c, rr); }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t m = w.r0; m < d.M; m += w.step) {
        float v[4], y[4] = {0.f, 0.f, 0.f, 0.f};
        ld4(g + m * d.ldg + c, v);
        if (d.corr1) {
            float x[4];
            ld4(xs + m * d.ldxs + c, x);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] -= k1[e] + ((x[e] - sm[e]) * sr[e]) * k2[e];
        }
        if (d.mask) {
            float mk[4];
            ld4(d.mask + (m / d.hw) * d.mask_ld + c, mk);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mk[e];
        }
        if (d.relu || d.s_acc) ld4(yr + m * d.ldyref + c, y);
        if (d.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pre = d.r_scale ? fmaf(y[e], rs[e], rh[e]) : y[e];
                if (!(pre > 0.f) || (d.relu == 2 && pre >= 6.f)) v[e] = 0.f;
            }
        }
        *reinterpret_cast<float4*>(dz + m * d.lddz + c) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], (y[e] - rm[e]) * rr[e], s2[e]); }
    }
    const int rep = (int)(w.r0 % DS_R);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (d.s_acc) {
            atomicAdd(d.s_acc + (size_t)rep * 2 * d.C + c + e, s1[e]);
            atomicAdd(d.s_acc + (size_t)rep * 2 * d.C + d.C + c + e, s2[e]);
        } else if (d.colsum) {
            atomicAdd(d.colsum + c + e, s1[e]);
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_f32_kernel(const dsaspp_bnapply_desc d) {
    const RowWalk w = row_walk(d.C);
    if (!w.act) return;
    const int c = w.c;
    const float* dr = reinterpret_cast<const float*>(d.dr);
    const float* y = reinterpret_cast<const float*>(d.y);
    float* dy = reinterpret_cast<float*>(d.dy);
    float A[4], B[4], Cc[4], s[4] = {0.f, 0.f, 0.f, 0.f};
    ld4(d.coefA + c, A); ld4(d.coefB + c, B); ld4(d.coefC + c, Cc);
    for (int64_t m = w.r0; m < d.M; m += w.step) {
        float r[4], yy[4], o[4];
        ld4(dr + m * d.lddr + c, r);
        ld4(y + m * d.ldy + c, yy);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = fmaf(A[e], r[e], fmaf(B[e], yy[e], Cc[e])); s[e] += o[e]; }
        *reinterpret_cast<float4*>(dy + m * d.lddy + c) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (d.colsum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(d.colsum + c + e, s[e]);
    }
}

// dx[n][h][w][c] = dy[n][h/2][w/2][c] / 4 ; d.x is the full-size OUTPUT, d.y the pooled-size INPUT
__global__ __launch_bounds__(256) void avgpool_bwd_f32_kernel(const dsaspp_pool_desc d, const int cg, const int64_t total) {
    const int Ho = d.H / 2, Wo = d.W / 2;
    float* xg = reinterpret_cast<float*>(const_cast<void*>(d.x));
    const float* yg = reinterpret_cast<const float*>(d.y);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / cg;
        const int c = (int)(idx - m * cg) * 4;
        const int n = (int)(m / ((int64_t)d.H * d.W));
        const int rem = (int)(m - (int64_t)n * d.H * d.W);
        const int h = rem / d.W, w = rem - h * d.W;
        const float4 v = *reinterpret_cast<const float4*>(yg + (((int64_t)n * Ho + h / 2) * Wo + w / 2) * d.ldy + c);
        *reinterpret_cast<float4*>(xg + m * d.ldx + c) = make_float4(0.25f * v.x, 0.25f * v.y, 0.25f * v.z, 0.25f * v.w);
    }
}

// gather form of the max-pooling backward: an input pixel collects the gradient of every window whose recorded
// maximum position is that pixel, then the ReLU mask of norm0 and that BatchNorm's backward sums
__global__ __launch_bounds__(256) void maxpool_bwd_f32_kernel(const dsaspp_maxpool_desc d) {
    const RowWalk w = row_walk(d.C);
    if (!w.act) return;
    const int c = w.c, H = d.H, W = d.W, Ho = H / 2, Wo = W / 2;
    const float* xg = reinterpret_cast<const float*>(d.x);
    const float* dyg = reinterpret_cast<const float*>(d.dy);
    float* dr = reinterpret_cast<float*>(d.dr);
    float sc[4], sh[4], mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {1.f, 1.f, 1.f, 1.f};
    ld4(d.scale + c, sc); ld4(d.shift + c, sh);
    if (d.s_acc) { ld4(d.mean + c, mu); ld4(d.rstd + c, rs); }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int64_t M = (int64_t)d.n_img * H * W;
    for (int64_t m = w.r0; m < M; m += w.step) {
        const int n = (int)(m / ((int64_t)H * W));
        const int rem = (int)(m - (int64_t)n * H * W);
        const int hi = rem / W, wi = rem - hi * W;
        float gsum[4] = {0.f, 0.f, 0.f, 0.f};
        for (int ho = (hi + 1) / 2 - 1; ho <= (hi + 1) / 2; ++ho) {
            const int ky = hi - (2 * ho - 1);
            if (ho < 0 || ho >= Ho || ky < 0 || ky > 2) continue;
            for (int wo = (wi + 1) / 2 - 1; wo <= (wi + 1) / 2; ++wo) {
                const int kx = wi - (2 * wo - 1);
                if (wo < 0 || wo >= Wo || kx < 0 || kx > 2) continue;
                const int64_t mo = ((int64_t)n * Ho + ho) * Wo + wo;
                float gy[4];
                ld4(dyg + mo * d.lddy + c, gy);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (d.argmax[mo * d.C + c + e] == ky * 3 + kx) gsum[e] += gy[e];
            }
        }
        float x[4];
        ld4(xg + m * d.ldx + c, x);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (!(fmaf(x[e], sc[e], sh[e]) > 0.f)) gsum[e] = 0.f;
            s1[e] += gsum[e];
            s2[e] = fmaf(gsum[e], (x[e] - mu[e]) * rs[e], s2[e]);
        }
        *reinterpret_cast<float4*>(dr + m * d.lddr + c) = make_float4(gsum[0], gsum[1], gsum[2], gsum[3]);
    }
    if (d.s_acc) {
        const int rep = (int)(w.r0 % DS_R);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(d.s_acc + (size_t)rep * 2 * d.C + c + e, s1[e]);
            atomicAdd(d.s_acc + (size_t)rep * 2 * d.C + d.C + c + e, s2[e]);
        }
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

// rows walked per thread column: enough threads to fill the chip, a whole number of channel groups
static unsigned walk_blocks(const int C, const int64_t M) {
    const int cg = C / 4;
    int64_t rows = M < 8192 ? M : 8192;
    if (rows < 1) rows = 1;
    return (unsigned)((rows * cg + 255) / 256);
}

static int conv_f32_one(const dsaspp_conv_desc& d, hipStream_t s) {
    const int M = d.n_img * d.H * d.W;
    const int nP = (M + FP - 1) / FP, nQ = (d.cout + FQ - 1) / FQ;
    hipLaunchKernelGGL(conv_f32_kernel, dim3(nP * nQ), dim3(256), 0, s, d, M, nQ);
    DS_LAUNCH_CHECK("conv_fwd(fp32)");
    return 0;
}

int dsaspp_f32_conv(const dsaspp_conv_desc& d, hipStream_t s) {
    DS_REQUIRE(d.cout % 4 == 0 && d.ldx % 4 == 0, "conv_fwd(fp32): cout and ldx must be multiples of 4");
    if (d.g) {
        DS_REQUIRE(!d.g_bf16 && d.ldg % 4 == 0 && ((uintptr_t)d.g & 15) == 0, "conv_fwd(fp32): the gradient slab must be fp32, 16-byte aligned");
    } else {
        DS_REQUIRE(d.y && d.ldy % 4 == 0 && d.ldy >= d.cout && ((uintptr_t)d.y & 15) == 0, "conv_fwd(fp32): bad output stride/alignment");
    }
    if (d.epi_relu || d.s_acc) DS_REQUIRE(d.ref && d.ldref % 4 == 0, "conv_fwd(fp32): reference tensor missing");
    if (d.s_acc) DS_REQUIRE(d.e_mean && d.e_rstd, "conv_fwd(fp32): e_mean/e_rstd missing");
    const int nsrc = d.n_src > 1 ? d.n_src : 1;
    if (nsrc == 1) return conv_f32_one(d, s);
    // several sources into the gradient slab: one launch each, every launch after the first accumulates
    DS_REQUIRE(d.g && d.taps == 1 && nsrc <= 4, "conv_fwd(fp32): extra sources need the gradient-slab form");
    const void* xs[4] = {d.x, d.x1, d.x2, d.x3};
    const void* ws[4] = {d.w, d.w1, d.w2, d.w3};
    const float* ess[4] = {d.e_scale, d.es1, d.es2, d.es3};
    const float* ehs[4] = {d.e_shift, d.eh1, d.eh2, d.eh3};
    float* sas[4] = {d.s_acc, d.sa1, d.sa2, d.sa3};
    const int sls[4] = {d.s_ld, d.sl1, d.sl2, d.sl3};
    for (int k = 0; k < nsrc; ++k) {
        dsaspp_conv_desc e = d;
        e.n_src = 1;
        e.x = xs[k]; e.w = ws[k]; e.e_scale = ess[k]; e.e_shift = ehs[k]; e.s_acc = sas[k]; e.s_ld = sls[k];
        if (k > 0) e.g_accumulate = 1;
        DS_REQUIRE(e.x && e.w && e.e_scale && e.e_shift, "conv_fwd(fp32): source %d incomplete", k);
        if (int rc = conv_f32_one(e, s)) return rc;
    }
    return 0;
}

int dsaspp_f32_wgrad(const dsaspp_wgrad_desc& d, hipStream_t s) {
    DS_REQUIRE(d.ldx % 4 == 0 && d.lddz % 4 == 0 && d.cin % 4 == 0 && d.cout % 4 == 0, "conv_wgrad(fp32): strides and widths must be multiples of 4");
    const int nsrc = d.n_src > 1 ? d.n_src : 1;
    DS_REQUIRE(nsrc <= 4, "conv_wgrad(fp32): at most 4 sources");
    const void* dzs[4] = {d.dz, d.dz1, d.dz2, d.dz3};
    float* dws[4] = {d.dw, d.dw1, d.dw2, d.dw3};
    const float* scs[4] = {d.in_scale, d.sc1, d.sc2, d.sc3};
    const float* shs[4] = {d.in_shift, d.sh1, d.sh2, d.sh3};
    const int cins[4] = {d.cin, d.cin1, d.cin2, d.cin3};
    const int M = d.n_img * d.H * d.W;
    for (int k = 0; k < nsrc; ++k) {
        dsaspp_wgrad_desc e = d;
        e.n_src = 0;
        e.dz = dzs[k]; e.dw = dws[k]; e.in_scale = scs[k]; e.in_shift = shs[k]; e.cin = cins[k];
        DS_REQUIRE(e.dz && e.dw && e.cin > 0 && e.cin % 4 == 0, "conv_wgrad(fp32): source %d incomplete", k);
        const int nQ = (e.cout + 63) / 64, nC = (e.cin + 63) / 64;
        int ksplit = 1024 / (nQ * nC * e.taps);
        const int max_split = (M + 255) / 256;
        if (ksplit > max_split) ksplit = max_split;
        if (ksplit < 1) ksplit = 1;
        int chunk = (M + ksplit - 1) / ksplit;
        chunk = (chunk + 63) / 64 * 64;
        ksplit = (M + chunk - 1) / chunk;
        hipLaunchKernelGGL(wgrad_f32_kernel, dim3(nQ * nC * e.taps * ksplit), dim3(256), 0, s, e, M, nQ, nC, chunk);
        DS_LAUNCH_CHECK("conv_wgrad(fp32)");
    }
    return 0;
}

int dsaspp_f32_grad_extract(const dsaspp_gradx_desc& d, hipStream_t s) {
    DS_REQUIRE(!d.g_bf16 && d.C % 4 == 0 && d.ldg % 4 == 0 && d.lddz % 4 == 0, "grad_extract(fp32): fp32 slab and strides that are multiples of 4");
    hipLaunchKernelGGL(grad_extract_f32_kernel, dim3(walk_blocks(d.C, d.M)), dim3(256), 0, s, d);
    DS_LAUNCH_CHECK("grad_extract(fp32)");
    return 0;
}

int dsaspp_f32_bn_bwd_apply(const dsaspp_bnapply_desc& d, hipStream_t s) {
    DS_REQUIRE(d.C % 4 == 0 && d.lddr % 4 == 0 && d.ldy % 4 == 0 && d.lddy % 4 == 0, "bn_bwd_apply(fp32): strides must be multiples of 4");
    hipLaunchKernelGGL(bn_bwd_apply_f32_kernel, dim3(walk_blocks(d.C, d.M)), dim3(256), 0, s, d);
    DS_LAUNCH_CHECK("bn_bwd_apply(fp32)");
    return 0;
}

int dsaspp_f32_avgpool_bwd(const dsaspp_pool_desc& d, hipStream_t s) {
    const int cg = d.C / 4;
    const int64_t total = (int64_t)d.n_img * d.H * d.W * cg;
    hipLaunchKernelGGL(avgpool_bwd_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, d, cg, total);
    DS_LAUNCH_CHECK("avgpool2_bwd(fp32)");
    return 0;
}

int dsaspp_f32_maxpool_bwd(const dsaspp_maxpool_desc& d, hipStream_t s) {
    DS_REQUIRE(d.argmax && d.dy && d.dr && d.scale && d.shift, "maxpool3s2_bwd(fp32): null tensor");
    hipLaunchKernelGGL(maxpool_bwd_f32_kernel, dim3(walk_blocks(d.C, (int64_t)d.n_img * d.H * d.W)), dim3(256), 0, s, d);
    DS_LAUNCH_CHECK("maxpool3s2_bwd(fp32)");
    return 0;
}

int dsaspp_f32_bn_act(const dsaspp_bnact_desc& d, hipStream_t s) {
    const int cg = d.C / 4;
    const int64_t total = d.M * cg;
    hipLaunchKernelGGL(bn_act_f32_kernel, dim3(blocks_for(total)), dim3(256), 0, s, d, cg, total);
    DS_LAUNCH_CHECK("bn_act(fp32)");
    return 0;
}

int dsaspp_f32_avgpool(const dsaspp_pool_desc& d, hipStream_t s) {
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
