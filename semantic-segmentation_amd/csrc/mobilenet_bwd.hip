// Backward kernels of the MobileNetV2 trunk that are not GEMM shaped: the depthwise 3x3 data and weight
// gradients and the 3x3/2 im2col that turns the stem's weight gradient into a GEMM.  All three are
// HBM-bound streaming kernels over a flat (pixel, 8-channel chunk) index: a lane owns one 16-byte NHWC chunk,
// consecutive lanes own consecutive chunks, per-channel sums are kept in registers, reduced per block in an
// LDS table with LDS atomics and added to global memory with one fp32 atomic per entry per block.
#include "common.h"

namespace {

int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// Number of 256-thread blocks for the flat (pixel, channel-chunk) kernels: a multiple of `unit` so that the
// grid-wide thread count is a multiple of the chunk count cg -- a lane then keeps the same 8 channels (and
// their weights, tables and partial sums in registers) for its whole loop, while consecutive lanes still
// touch consecutive 16-byte chunks of memory whatever cg is.
int flat_blocks(int64_t total, int cg, int cap) {
    const int unit = cg / gcd_i(cg, 256);
    int64_t want = (total + 255) / 256;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    return (int)((want + unit - 1) / unit * unit);
}

// Block-level reduction for the flat kernels: every lane adds its NS x 8 channel partials into an LDS table
// [NS][C] with LDS atomics, then the block adds the table to global memory (one atomic per entry per block).
template <int NS>
__device__ __forceinline__ void flat_colsum(float (&part)[NS][8], float* lds, float* out, int C, int c, int out_stride) {
    for (int i = threadIdx.x; i < NS * C; i += 256) lds[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(lds + s * C + c + e, part[s][e]);
    __syncthreads();
    for (int i = threadIdx.x; i < NS * C; i += 256) {
        const int s = i / C, cc = i - s * C;
        atomicAdd(out + (int64_t)s * out_stride + cc, lds[i]);
    }
}

// ---------------------------------------------------------------- depthwise 3x3: data gradient
// Gather form: an input pixel (h, w) collects, for each tap, the output pixel whose window touched it through
// that tap (for stride 2 only positions of the right parity have one).  The activation mask of the producer's
// BatchNorm + ReLU(6) and that BatchNorm's backward sums are applied on the way out, so the raw input x is
// read once here instead of in a separate masking pass.
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const dsaspp_dwbwd_desc d, const int Ho, const int Wo, const int cg,
                                                          const int total) {
    extern __shared__ float red[];
    const int tid0 = blockIdx.x * 256 + threadIdx.x;
    const int nthreads = gridDim.x * 256;
    const int c = (tid0 % cg) * 8;
    float w[9][8], sc[8], sh[8], mu[8], rs[8], part[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; part[0][e] = 0.f; part[1][e] = 0.f; }
    {
        const u16* wg = reinterpret_cast<const u16*>(d.w) + (int64_t)c * 9;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t][e] = __uint_as_float((uint32_t)wg[e * 9 + t] << 16);
    }
    if (d.in_scale) { load8f(d.in_scale + c, sc); load8f(d.in_shift + c, sh); }
    if (d.s_acc) { load8f(d.e_mean + c, mu); load8f(d.e_rstd + c, rs); }
    const int H = d.H, W = d.W, st = d.stride, dil = d.dil;
    const u16* dyg = reinterpret_cast<const u16*>(d.dy) + c;
    const u16* xg = reinterpret_cast<const u16*>(d.x) + c;
    const bool need_x = d.in_relu || d.s_acc;
    for (int idx = tid0; idx < total; idx += nthreads) {
        const int m = idx / cg;
        const int n = m / (H * W);
        const int rem = m - n * (H * W);
        const int h = rem / W, wq = rem - h * W;
        uint4 v[9];
        bool ok[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hn = h - (ky - 1) * dil;
            const int ho = hn / st;
            const bool hok = hn >= 0 && ho * st == hn && ho < Ho;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int wn = wq - (kx - 1) * dil;
                const int wo = wn / st;
                const bool o = hok && wn >= 0 && wo * st == wn && wo < Wo;
                ok[ky * 3 + kx] = o;
                const int64_t pix = o ? ((int64_t)n * Ho + ho) * Wo + wo : 0;
                v[ky * 3 + kx] = *reinterpret_cast<const uint4*>(dyg + pix * d.lddy);   // always a mapped address
            }
        }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float f[8];
            unpack8(v[t], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(ok[t] ? f[e] : 0.f, w[t][e], acc[e]);
        }
        float x[8];
        if (need_x) unpack8(*reinterpret_cast<const uint4*>(xg + (int64_t)m * d.ldx), x);
        if (d.in_relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pre = fmaf(x[e], sc[e], sh[e]);
                if (!(pre > 0.f) || (d.in_relu == 2 && pre >= 6.f)) acc[e] = 0.f;
            }
        }
        if (d.dx_f32) store8f(reinterpret_cast<float*>(d.dx) + (int64_t)m * d.lddx + c, acc);
        else *reinterpret_cast<uint4*>(reinterpret_cast<u16*>(d.dx) + (int64_t)m * d.lddx + c) = pack8(acc);
        if (d.s_acc) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { part[0][e] += acc[e]; part[1][e] += acc[e] * ((x[e] - mu[e]) * rs[e]); }
        }
    }
    if (d.s_acc) flat_colsum<2>(part, red, d.s_acc + (size_t)(blockIdx.x % DS_R) * 2 * d.C, d.C, c, d.C);
}

// ---------------------------------------------------------------- depthwise 3x3: weight gradient
// A lane accumulates the 9 x 8 products of its channel chunk over a strided set of output pixels (the nine
// activated input values are recomputed from the raw input with the forward prologue); the block reduces them
// in an LDS table [9][C] and adds that to dw [C][9].
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const dsaspp_dwbwd_desc d, const int Ho, const int Wo, const int cg,
                                                       const int total) {
    extern __shared__ float red[];
    const int tid0 = blockIdx.x * 256 + threadIdx.x;
    const int nthreads = gridDim.x * 256;
    const int c = (tid0 % cg) * 8;
    float sc[8], sh[8], part[9][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = 1.f; sh[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) part[t][e] = 0.f;
    }
    if (d.in_scale) { load8f(d.in_scale + c, sc); load8f(d.in_shift + c, sh); }
    const bool pro = d.in_scale != nullptr || d.in_relu;
    const float lo = d.in_relu ? 0.f : -3.0e38f, hi = (d.in_relu == 2) ? 6.f : 3.0e38f;
    const int H = d.H, W = d.W, st = d.stride, dil = d.dil;
    const u16* dyg = reinterpret_cast<const u16*>(d.dy) + c;
    const u16* xg = reinterpret_cast<const u16*>(d.x) + c;
    for (int idx = tid0; idx < total; idx += nthreads) {
        const int m = idx / cg;
        const int n = m / (Ho * Wo);
        const int rem = m - n * (Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        float g[8];
        unpack8(*reinterpret_cast<const uint4*>(dyg + (int64_t)m * d.lddy), g);
        uint4 v[9];
        bool ok[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hi_ = ho * st + (ky - 1) * dil;
            const bool hok = (unsigned)hi_ < (unsigned)H;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int wi = wo * st + (kx - 1) * dil;
                const bool o = hok && (unsigned)wi < (unsigned)W;
                ok[ky * 3 + kx] = o;
                const int64_t pix = o ? ((int64_t)n * H + hi_) * W + wi : 0;
                v[ky * 3 + kx] = *reinterpret_cast<const uint4*>(xg + pix * d.ldx);   // always a mapped address
            }
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float f[8];
            unpack8(v[t], f);
            if (pro) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bf_round(fminf(fmaxf(fmaf(f[e], sc[e], sh[e]), lo), hi));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) part[t][e] = fmaf(g[e], ok[t] ? f[e] : 0.f, part[t][e]);
        }
    }
    // LDS table [9][C]; dw is [C][9]
    for (int i = threadIdx.x; i < 9 * d.C; i += 256) red[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(red + t * d.C + c + e, part[t][e]);
    __syncthreads();
    for (int i = threadIdx.x; i < 9 * d.C; i += 256) {
        const int cc = i / 9, t = i - cc * 9;
        atomicAdd(d.dw + i, red[t * d.C + cc]);
    }
}

// ---------------------------------------------------------------- stem 3x3/2 im2col
// One lane writes one 16-byte chunk (8 columns) of a row; columns k = (ky*3 + kx)*3 + ci, zero for k >= 27
// and for taps outside the image.
__global__ __launch_bounds__(256) void stem3_im2col_kernel(const dsaspp_im2col_desc d, const int Ho, const int Wo, const int cg,
                                                           const int64_t total) {
    const int H = d.H, W = d.W;
    const int64_t plane = (int64_t)H * W;
    u16* col = reinterpret_cast<u16*>(d.col);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t m = idx / cg;
        const int g = (int)(idx - m * cg);
        const int n = (int)(m / ((int64_t)Ho * Wo));
        const int rem = (int)(m - (int64_t)n * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const float* xn = d.x + (int64_t)n * 3 * plane;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = g * 8 + e;
            const int tap = k / 3, ci = k - tap * 3;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int hi = 2 * ho - 1 + ky, wi = 2 * wo - 1 + kx;
            const bool ok = k < 27 && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
            v[e] = ok ? xn[ci * plane + (int64_t)hi * W + wi] : 0.f;
        }
        *reinterpret_cast<uint4*>(col + m * d.ldcol + g * 8) = pack8(v);
    }
}

int check_dw(const dsaspp_dwbwd_desc& d, const char* who) {
    DS_REQUIRE(d.n_img > 0 && d.H > 0 && d.W > 0, "%s: bad geometry %d x %d x %d", who, d.n_img, d.H, d.W);
    DS_REQUIRE(d.C > 0 && d.C % 8 == 0, "%s: C must be a positive multiple of 8 (got %d)", who, d.C);
    DS_REQUIRE(d.stride == 1 || d.stride == 2, "%s: stride must be 1 or 2 (got %d)", who, d.stride);
    DS_REQUIRE(d.dil >= 1, "%s: dilation must be >= 1 (got %d)", who, d.dil);
    DS_REQUIRE(d.dy && d.lddy % 8 == 0 && d.lddy >= d.C && ((uintptr_t)d.dy & 15) == 0, "%s: output-gradient tensor missing/misaligned", who);
    DS_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr), "%s: in_scale/in_shift must come together", who);
    DS_REQUIRE(d.in_relu >= 0 && d.in_relu <= 2, "%s: in_relu must be 0, 1 or 2", who);
    DS_REQUIRE((int64_t)d.n_img * d.H * d.W < (1ll << 31), "%s: too many pixels", who);
    return 0;
}

}  // namespace

extern "C" int dsaspp_dwconv3x3_bwd_data(const dsaspp_dwbwd_desc* dp, dsaspp_stream_t stream) {
    DS_REQUIRE(dp != nullptr, "dwconv3x3_bwd_data: null descriptor");
    const dsaspp_dwbwd_desc& d = *dp;
    if (const int rc = check_dw(d, "dwconv3x3_bwd_data")) return rc;
    DS_REQUIRE(d.w && d.dx && d.lddx % 8 == 0 && d.lddx >= d.C && ((uintptr_t)d.dx & 15) == 0, "dwconv3x3_bwd_data: weights / result tensor missing or misaligned");
    if (d.in_relu || d.s_acc) DS_REQUIRE(d.x && d.ldx % 8 == 0 && d.ldx >= d.C && ((uintptr_t)d.x & 15) == 0, "dwconv3x3_bwd_data: forward input missing/misaligned");
    if (d.s_acc) DS_REQUIRE(d.e_mean && d.e_rstd, "dwconv3x3_bwd_data: s_acc needs e_mean/e_rstd");
    const int Ho = (d.H - 1) / d.stride + 1, Wo = (d.W - 1) / d.stride + 1, cg = d.C / 8;
    const int64_t total = (int64_t)d.n_img * d.H * d.W * cg;
    DS_REQUIRE(total < (1ll << 31) - (1 << 22), "dwconv3x3_bwd_data: too many chunks");
    DS_REQUIRE(d.C <= 4096, "dwconv3x3_bwd_data: at most 4096 channels (got %d)", d.C);
    const int nb = flat_blocks(total, cg, 4096);
    hipLaunchKernelGGL(dw_bwd_data_kernel, dim3(nb), dim3(256), (size_t)2 * d.C * sizeof(float), (hipStream_t)stream, d, Ho, Wo, cg, (int)total);
    DS_LAUNCH_CHECK("dwconv3x3_bwd_data");
    return 0;
}

extern "C" int dsaspp_dwconv3x3_wgrad(const dsaspp_dwbwd_desc* dp, dsaspp_stream_t stream) {
    DS_REQUIRE(dp != nullptr, "dwconv3x3_wgrad: null descriptor");
    const dsaspp_dwbwd_desc& d = *dp;
    if (const int rc = check_dw(d, "dwconv3x3_wgrad")) return rc;
    DS_REQUIRE(d.x && d.ldx % 8 == 0 && d.ldx >= d.C && ((uintptr_t)d.x & 15) == 0, "dwconv3x3_wgrad: forward input missing/misaligned");
    DS_REQUIRE(d.dw != nullptr, "dwconv3x3_wgrad: null result");
    const int Ho = (d.H - 1) / d.stride + 1, Wo = (d.W - 1) / d.stride + 1, cg = d.C / 8;
    const int64_t total = (int64_t)d.n_img * Ho * Wo * cg;
    DS_REQUIRE(total < (1ll << 31) - (1 << 22), "dwconv3x3_wgrad: too many chunks");
    DS_REQUIRE(d.C <= 1536, "dwconv3x3_wgrad: at most 1536 channels (got %d)", d.C);
    const int nb = flat_blocks(total, cg, 1024);
    hipLaunchKernelGGL(dw_wgrad_kernel, dim3(nb), dim3(256), (size_t)9 * d.C * sizeof(float), (hipStream_t)stream, d, Ho, Wo, cg, (int)total);
    DS_LAUNCH_CHECK("dwconv3x3_wgrad");
    return 0;
}

extern "C" int dsaspp_stem3_im2col(const dsaspp_im2col_desc* dp, dsaspp_stream_t stream) {
    DS_REQUIRE(dp != nullptr, "stem3_im2col: null descriptor");
    const dsaspp_im2col_desc& d = *dp;
    DS_REQUIRE(d.n_img > 0 && d.H > 0 && d.W > 0 && d.H % 2 == 0 && d.W % 2 == 0, "stem3_im2col: image size must be even (got %d x %d)", d.H, d.W);
    DS_REQUIRE(d.x && d.col && d.ldcol >= 32 && d.ldcol % 8 == 0 && ((uintptr_t)d.col & 15) == 0, "stem3_im2col: ldcol must be a multiple of 8, at least 32");
    DS_REQUIRE(d.dtype == 0, "stem3_im2col: bf16 rows only");
    const int Ho = d.H / 2, Wo = d.W / 2, cg = d.ldcol / 8;
    const int64_t total = (int64_t)d.n_img * Ho * Wo * cg;
    int64_t nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(stem3_im2col_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, d, Ho, Wo, cg, total);
    DS_LAUNCH_CHECK("stem3_im2col");
    return 0;
}
