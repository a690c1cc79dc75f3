// MobileNetV2-trunk kernels that are not GEMM shaped: the 3x3/2 stem on the 3-channel image and the
// depthwise (dilated, optionally strided) 3x3 convolution.  Both are bandwidth-bound streaming kernels:
// a lane owns 8 consecutive channels of one output pixel (one 16-byte NHWC chunk), consecutive lanes own
// consecutive chunks of the same pixel and then the next pixel, so every store instruction of a wave covers
// one contiguous run of memory and the nine tap loads are contiguous runs shifted by the tap offset.
#include <cstdlib>

#include "common.h"

namespace {

// ---------------------------------------------------------------- stem: 3x3 stride 2, 3 -> cout channels
// Weights (cout x 27, bf16) are expanded to fp32 in LDS as [27][cout]; a lane reads the 27 input values of
// its pixel (shared by the cout/8 lanes of that pixel: broadcast loads) and 8 output channels of weights.
__global__ __launch_bounds__(256) void stem3_kernel(const dsaspp_stem3_desc d, const int Ho, const int Wo, const int cg,
                                                    const int total) {
    __shared__ float ws[27 * 64];
    const int cout = d.cout;
    const u16* wg = reinterpret_cast<const u16*>(d.w);
    for (int i = threadIdx.x; i < 27 * cout; i += 256) {
        const int k = i / cout, c = i - k * cout;
        ws[i] = __uint_as_float((uint32_t)wg[c * 27 + k] << 16);
    }
    __syncthreads();
    const int H = d.H, W = d.W;
    const int64_t plane = (int64_t)H * W;
    u16* yg = reinterpret_cast<u16*>(d.y);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int m = idx / cg, g = idx - m * cg;
        const int n = m / (Ho * Wo);
        const int rem = m - n * (Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const float* xn = d.x + (int64_t)n * 3 * plane;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int hi = 2 * ho - 1 + ky;
            const bool hok = (unsigned)hi < (unsigned)H;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int wi = 2 * wo - 1 + kx;
                const bool ok = hok && (unsigned)wi < (unsigned)W;
                const int64_t off = ok ? (int64_t)hi * W + wi : 0;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    // the network input is rounded to bf16 like every other MFMA/VALU operand of the path
                    const float v = ok ? bf_round(xn[ci * plane + off]) : 0.f;
                    float w8[8];
                    load8f(ws + ((ky * 3 + kx) * 3 + ci) * cout + g * 8, w8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(v, w8[e], acc[e]);
                }
            }
        }
        *reinterpret_cast<uint4*>(yg + (int64_t)m * d.ldy + g * 8) = pack8(acc);
    }
}

// ---------------------------------------------------------------- depthwise 3x3
// The grid is sized so that (threads in the grid) is a multiple of the channel-group count: a lane keeps the
// same 8 channels for its whole pixel loop and holds their 72 weights and the prologue tables in registers.
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const dsaspp_dw_desc d, const int Ho, const int Wo, const int cg,
                                                        const int total) {
    const int tid0 = blockIdx.x * 256 + threadIdx.x;
    const int nthreads = gridDim.x * 256;
    const int g = tid0 % cg, c = g * 8;
    float w[9][8], sc[8], sh[8];
    {
        const u16* wg = reinterpret_cast<const u16*>(d.w) + (int64_t)c * 9;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t][e] = __uint_as_float((uint32_t)wg[e * 9 + t] << 16);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    const bool affine = d.in_scale != nullptr;
    if (affine) { load8f(d.in_scale + c, sc); load8f(d.in_shift + c, sh); }
    const bool pro = affine || d.in_relu;
    const float lo = d.in_relu ? 0.f : -3.0e38f, hi = (d.in_relu == 2) ? 6.f : 3.0e38f;
    const int H = d.H, W = d.W, st = d.stride, dil = d.dil;
    const u16* xg = reinterpret_cast<const u16*>(d.x) + c;
    u16* yg = reinterpret_cast<u16*>(d.y) + c;
    for (int idx = tid0; idx < total; idx += nthreads) {
        const int m = idx / cg;
        const int n = m / (Ho * Wo);
        const int rem = m - n * (Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
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
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float f[8];
            unpack8(v[t], f);
            if (pro) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bf_round(fminf(fmaxf(fmaf(f[e], sc[e], sh[e]), lo), hi));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(ok[t] ? f[e] : 0.f, w[t][e], acc[e]);
        }
        *reinterpret_cast<uint4*>(yg + (int64_t)m * d.ldy) = pack8(acc);
    }
}

// ---------------------------------------------------------------- stem, 32 output channels: a lane owns a pixel
// A lane computes all 32 output channels of one output pixel: 27 input loads (lanes = consecutive output
// columns, so a load instruction covers a contiguous run of the image row) and 864 FMAs whose weight operands
// are wave-uniform: the weights sit in LDS as fp32 [27][32] and every lane reads the same 16 bytes (broadcast).
__global__ __launch_bounds__(256) void stem3_c32_kernel(const dsaspp_stem3_desc d, const int Ho, const int Wo, const int total) {
    constexpr int CO = 32;
    __shared__ __attribute__((aligned(16))) float ws[27 * CO];
    {
        const u16* wg = reinterpret_cast<const u16*>(d.w);
        for (int i = threadIdx.x; i < 27 * CO; i += 256) {
            const int k = i / CO, c = i - k * CO;
            ws[i] = __uint_as_float((uint32_t)wg[c * 27 + k] << 16);
        }
    }
    __syncthreads();
    const int H = d.H, W = d.W;
    const int64_t plane = (int64_t)H * W;
    u16* yg = reinterpret_cast<u16*>(d.y);
    // one pixel per lane and no loop around it: in a loop the 216 weight reads are loop invariant, get hoisted into
    // registers and spill
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m < total) {
        const int n = m / (Ho * Wo);
        const int rem = m - n * (Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const float* xn = d.x + (int64_t)n * 3 * plane;
        float acc[CO];
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[c] = 0.f;
        // the tap loop stays a loop: fully unrolled, the scheduler pulls all 216 weight reads to the front and spills
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - ky * 3;
            {
                const int hi = 2 * ho - 1 + ky, wi = 2 * wo - 1 + kx;
                const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
                const int64_t off = ok ? (int64_t)hi * W + wi : 0;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float v = ok ? bf_round(xn[ci * plane + off]) : 0.f;
                    const float* wk = ws + (t * 3 + ci) * CO;
#pragma unroll
                    for (int c4 = 0; c4 < CO / 4; ++c4) {
                        const float4 w4 = *reinterpret_cast<const float4*>(wk + c4 * 4);
                        acc[c4 * 4 + 0] = fmaf(v, w4.x, acc[c4 * 4 + 0]);
                        acc[c4 * 4 + 1] = fmaf(v, w4.y, acc[c4 * 4 + 1]);
                        acc[c4 * 4 + 2] = fmaf(v, w4.z, acc[c4 * 4 + 2]);
                        acc[c4 * 4 + 3] = fmaf(v, w4.w, acc[c4 * 4 + 3]);
                    }
                }
            }
        }
        u16* yo = yg + (int64_t)m * d.ldy;
#pragma unroll
        for (int c8 = 0; c8 < CO / 8; ++c8) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = acc[c8 * 8 + e];
            *reinterpret_cast<uint4*>(yo + c8 * 8) = pack8(f);
        }
    }
}

// ---------------------------------------------------------------- depthwise 3x3 through an LDS tile
// A dilated depthwise convolution is d*d independent dense 3x3 convolutions, one per residue class
// (h mod d, w mod d): output pixel (h, w) only reads input pixels of its own class.  A block takes one image,
// one group of CCH 16-byte channel chunks, one residue class and one TH x TW tile of that class's sub-grid;
// it stages the (TH-1)*stride+3 by (TW-1)*stride+3 input pixels the tile needs in LDS -- BatchNorm + activation
// applied ONCE per element on the way in, zero padding after it -- and every lane then reads its nine taps from
// LDS.  Against the direct kernel this cuts the prologue arithmetic 9x -> ~1.4x and turns the nine strided
// global reads per output (which miss L2 for large dilations) into one pass over the input.
template <int CCH, int TH, int TW, int ST>
__global__ __launch_bounds__(32 * CCH) void dwconv3x3_tile_kernel(const dsaspp_dw_desc d, const int Ho, const int Wo, const int tiles_w,
                                                                  const int tiles_per_class) {
    constexpr int ITH = (TH - 1) * ST + 3, ITW = (TW - 1) * ST + 3;
    constexpr int ROWB = CCH * 16;                 // bytes of one staged pixel
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, ck = tid % CCH, pl = tid / CCH;      // chunk lane, pixel lane (0..31)
    const int dil = d.dil, H = d.H, W = d.W;
    const int cls = blockIdx.x / tiles_per_class, tile = blockIdx.x - cls * tiles_per_class;
    const int ca = cls / dil, cb = cls - ca * dil;                    // residue class (row, column); dil == 1 when ST == 2
    const int ti = tile / tiles_w, tj = tile - ti * tiles_w;
    const int n = blockIdx.z;
    const int c = (blockIdx.y * CCH + ck) * 8;
    // sub-grid coordinates: output (oi, oj) of the class <-> output pixel (ca + oi*dil, cb + oj*dil);
    // input sub-coordinate of tap (ky, kx): (oi*ST + ky - 1, oj*ST + kx - 1) <-> pixel (ca + (.)*dil, cb + (.)*dil)
    const int oi0 = ti * TH, oj0 = tj * TW;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    const bool affine = d.in_scale != nullptr;
    if (affine) { load8f(d.in_scale + c, sc); load8f(d.in_shift + c, sh); }
    const bool pro = affine || d.in_relu;
    const float lo = d.in_relu ? 0.f : -3.0e38f, hi = (d.in_relu == 2) ? 6.f : 3.0e38f;
    const u16* xg = reinterpret_cast<const u16*>(d.x) + c;
    // ---- stage the input tile
    for (int p = pl; p < ITH * ITW; p += 32) {
        const int ii = p / ITW, jj = p - ii * ITW;
        const int hi_ = ca + (oi0 * ST + ii - 1) * dil, wi = cb + (oj0 * ST + jj - 1) * dil;
        const bool ok = (unsigned)hi_ < (unsigned)H && (unsigned)wi < (unsigned)W;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (ok) {
            v = *reinterpret_cast<const uint4*>(xg + (((int64_t)n * H + hi_) * W + wi) * d.ldx);
            if (pro) {
                float f[8];
                unpack8(v, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(fmaf(f[e], sc[e], sh[e]), lo), hi);
                v = pack8(f);
            }
        }
        *reinterpret_cast<uint4*>(smem + p * ROWB + ck * 16) = v;
    }
    float w[9][8];
    {
        const u16* wg = reinterpret_cast<const u16*>(d.w) + (int64_t)c * 9;
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t][e] = __uint_as_float((uint32_t)wg[e * 9 + t] << 16);
    }
    __syncthreads();
    // ---- nine taps from LDS
    u16* yg = reinterpret_cast<u16*>(d.y) + c;
#pragma unroll
    for (int q = 0; q < TH * TW / 32; ++q) {
        const int po = q * 32 + pl;
        const int oi = po / TW, oj = po - oi * TW;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(smem + ((oi * ST + ky) * ITW + oj * ST + kx) * ROWB + ck * 16), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(f[e], w[ky * 3 + kx][e], acc[e]);
            }
        const int ho = ca + (oi0 + oi) * dil, wo = cb + (oj0 + oj) * dil;     // ST == 2 implies dil == 1, ca == cb == 0
        if (ho < Ho && wo < Wo) *reinterpret_cast<uint4*>(yg + (((int64_t)n * Ho + ho) * Wo + wo) * d.ldy) = pack8(acc);
    }
}

template <int CCH, int TH, int TW, int ST>
int launch_dw_tile(const dsaspp_dw_desc& d, int Ho, int Wo, hipStream_t s) {
    constexpr int ITH = (TH - 1) * ST + 3, ITW = (TW - 1) * ST + 3;
    const int dil = d.dil;
    // sub-grid of a residue class: ceil(Ho / dil) x ceil(Wo / dil) outputs (classes near the end hold one row/column less)
    const int sh_ = (Ho + dil - 1) / dil, sw_ = (Wo + dil - 1) / dil;
    const int tiles_h = (sh_ + TH - 1) / TH, tiles_w = (sw_ + TW - 1) / TW;
    const int64_t gx = (int64_t)dil * dil * tiles_h * tiles_w;
    DS_REQUIRE(gx < (1ll << 31) && d.n_img <= 65535, "dwconv3x3: grid too large");
    const size_t lds = (size_t)ITH * ITW * CCH * 16;
    auto kern = dwconv3x3_tile_kernel<CCH, TH, TW, ST>;
    if (lds > 64 * 1024) dsaspp_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3((unsigned)gx, (d.C / 8) / CCH, d.n_img), dim3(32 * CCH), lds, s, d, Ho, Wo, tiles_w, tiles_h * tiles_w);
    DS_LAUNCH_CHECK("dwconv3x3(tile)");
    return 0;
}

// -1: shape not covered by the tile kernel (the direct kernel takes it)
int launch_dw_tiled(const dsaspp_dw_desc& d, int Ho, int Wo, hipStream_t s) {
    const int cg = d.C / 8;
    if (d.stride == 2 && d.dil != 1) return -1;
    const int cch = (cg % 8 == 0) ? 8 : (cg % 6 == 0) ? 6 : (cg % 4 == 0) ? 4 : 0;
    if (!cch) return -1;
    if (d.stride == 1) {
        if (cch == 8) return launch_dw_tile<8, 8, 16, 1>(d, Ho, Wo, s);
        if (cch == 6) return launch_dw_tile<6, 8, 16, 1>(d, Ho, Wo, s);
        return launch_dw_tile<4, 8, 16, 1>(d, Ho, Wo, s);
    }
    if (cch == 8) return launch_dw_tile<8, 4, 16, 2>(d, Ho, Wo, s);
    if (cch == 6) return launch_dw_tile<6, 4, 16, 2>(d, Ho, Wo, s);
    return launch_dw_tile<4, 4, 16, 2>(d, Ho, Wo, s);
}

int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

// Number of 256-thread blocks: enough to fill the chip several times over, a multiple of `unit` so that
// the grid-wide thread count is a multiple of the channel-group count.
int grid_blocks(int64_t total, int cg) {
    const int unit = cg / gcd_i(cg, 256);
    int64_t want = (total + 255) / 256;
    if (want > 8192) want = 8192;
    if (want < 1) want = 1;
    return (int)((want + unit - 1) / unit * unit);
}

}  // namespace

extern "C" int dsaspp_stem3x3s2(const dsaspp_stem3_desc* dp, dsaspp_stream_t stream) {
    DS_REQUIRE(dp != nullptr, "stem3x3s2: null descriptor");
    const dsaspp_stem3_desc& d = *dp;
    DS_REQUIRE(d.n_img > 0 && d.H > 0 && d.W > 0 && d.H % 2 == 0 && d.W % 2 == 0, "stem3x3s2: image size must be even (got %d x %d)", d.H, d.W);
    DS_REQUIRE(d.cout > 0 && d.cout % 8 == 0 && d.cout <= 64, "stem3x3s2: cout must be a multiple of 8, at most 64 (got %d)", d.cout);
    DS_REQUIRE(d.x && d.w && d.y, "stem3x3s2: null tensor");
    DS_REQUIRE(d.ldy % 8 == 0 && d.ldy >= d.cout && ((uintptr_t)d.y & 15) == 0, "stem3x3s2: output stride/alignment");
    if (d.dtype == 1) return dsaspp_f32_stem3(d, (hipStream_t)stream);
    const int Ho = d.H / 2, Wo = d.W / 2, cg = d.cout / 8;
    const int64_t total = (int64_t)d.n_img * Ho * Wo * cg;
    DS_REQUIRE(total < (1ll << 31) - (1 << 23), "stem3x3s2: too many output chunks");
    if (d.cout == 32) {      // one lane per pixel, weights through scalar registers
        const int64_t npix = (int64_t)d.n_img * Ho * Wo;
        const int64_t nbp = (npix + 255) / 256;
        hipLaunchKernelGGL(stem3_c32_kernel, dim3((unsigned)nbp), dim3(256), 0, (hipStream_t)stream, d, Ho, Wo, (int)npix);
        DS_LAUNCH_CHECK("stem3x3s2");
        return 0;
    }
    int64_t nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(stem3_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, d, Ho, Wo, cg, (int)total);
    DS_LAUNCH_CHECK("stem3x3s2");
    return 0;
}

extern "C" int dsaspp_dwconv3x3(const dsaspp_dw_desc* dp, dsaspp_stream_t stream) {
    DS_REQUIRE(dp != nullptr, "dwconv3x3: null descriptor");
    const dsaspp_dw_desc& d = *dp;
    DS_REQUIRE(d.n_img > 0 && d.H > 0 && d.W > 0, "dwconv3x3: bad geometry %d x %d x %d", d.n_img, d.H, d.W);
    DS_REQUIRE(d.C > 0 && d.C % 8 == 0, "dwconv3x3: C must be a positive multiple of 8 (got %d)", d.C);
    DS_REQUIRE(d.stride == 1 || d.stride == 2, "dwconv3x3: stride must be 1 or 2 (got %d)", d.stride);
    DS_REQUIRE(d.dil >= 1, "dwconv3x3: dilation must be >= 1 (got %d)", d.dil);
    DS_REQUIRE(d.x && d.w && d.y, "dwconv3x3: null tensor");
    DS_REQUIRE(d.ldx % 8 == 0 && d.ldx >= d.C && d.ldy % 8 == 0 && d.ldy >= d.C, "dwconv3x3: strides must be multiples of 8 and >= C");
    DS_REQUIRE((((uintptr_t)d.x | (uintptr_t)d.y) & 15) == 0, "dwconv3x3: tensors must be 16-byte aligned");
    DS_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr), "dwconv3x3: in_scale/in_shift must come together");
    DS_REQUIRE(d.in_relu >= 0 && d.in_relu <= 2, "dwconv3x3: in_relu must be 0, 1 or 2");
    if (d.dtype == 1) return dsaspp_f32_dwconv(d, (hipStream_t)stream);
    const int Ho = (d.H - 1) / d.stride + 1, Wo = (d.W - 1) / d.stride + 1, cg = d.C / 8;
    const int64_t total = (int64_t)d.n_img * Ho * Wo * cg;
    DS_REQUIRE(total < (1ll << 31) - (1 << 22), "dwconv3x3: too many output chunks");
    {   // LDS-tiled form (DSASPP_DW_DIRECT=1 forces the direct kernel: measurement / fallback check)
        static const int direct = [] { const char* e = getenv("DSASPP_DW_DIRECT"); return e ? atoi(e) : 0; }();
        if (!direct) {
            const int rc = launch_dw_tiled(d, Ho, Wo, (hipStream_t)stream);
            if (rc >= 0) {
                dsaspp_note_route("dwconv_tile");
                return rc;
            }
        }
    }
    dsaspp_note_route("dwconv_direct");
    const int nb = grid_blocks(total, cg);
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, d, Ho, Wo, cg, (int)total);
    DS_LAUNCH_CHECK("dwconv3x3");
    return 0;
}
