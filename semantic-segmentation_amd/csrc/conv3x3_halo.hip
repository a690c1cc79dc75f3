// BN-ReLU -> 3x3 dilated convolution of the dense layers (wide bottleneck in, growth-rate channels out),
// gfx950.  The generic implicit-GEMM kernel stages the shifted input tile once per filter tap, i.e. nine
// times; here a block owns an 8 x 32 pixel patch of one image and stages the patch plus its dilation halo
// ONCE per 64-channel chunk (BatchNorm scale/shift, ReLU and the zero padding applied on the way into
// LDS), then runs all nine taps out of LDS by shifting the fragment base address.  LDS rows are one pixel
// (64 channels = 128 B) padded to 160 B, which makes any 16 consecutive pixels conflict-free for
// ds_read_b128 whatever the shift.  The few output channels make the filter small: its MFMA fragments are
// read straight from global memory (L1/L2 resident), one tap ahead of use, and never touch LDS.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32, ROWB = 160, NT = 256;

template <int MQ>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo_kernel(const dsaspp_conv_desc d, const int tiles_x, const int tiles_y,
                                                            const int nblocks) {
    constexpr int MP = 4;                               // per wave: 2 patch rows x 2 half rows of 16 pixels
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout, dil = d.dil;
    const int HW2 = TW + 2 * dil, HH2 = TH + 2 * dil, NPIX = HW2 * HH2;
    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int per_img = tiles_x * tiles_y;
    const int n = bid / per_img, rem = bid - n * per_img;
    const int tyi = rem / tiles_x;
    const int ty0 = tyi * TH, tx0 = (rem - tyi * tiles_x) * TW;
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    const u16* wg = reinterpret_cast<const u16*>(d.w);
    const float relu_floor = d.in_relu ? 0.f : -3.0e38f;
    const bool has_scale = d.in_scale != nullptr;

    // staging: item = tid + 256*it covers chunk column (tid & 7) of halo pixel item >> 3
    const int ck = tid & 7;
    const int nit = (NPIX * 8 + NT - 1) / NT;

    // fragment coordinates
    const int fr = lane & 15, fc = lane >> 4;
    int rbase[MP];
#pragma unroll
    for (int j = 0; j < MP; ++j) {
        const int ty = 2 * wave + (j >> 1), tx = (j & 1) * 16 + fr;
        rbase[j] = ((ty + dil) * HW2 + tx + dil) * ROWB + fc * 16;
    }
    f32x4 acc[MQ][MP];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int Ktot = 9 * cin;
    const u16* wrow[MQ];
    bool wok[MQ];
#pragma unroll
    for (int i = 0; i < MQ; ++i) {
        const int co = i * 16 + fr;
        wok[i] = co < cout;
        wrow[i] = wg + (int64_t)(wok[i] ? co : 0) * Ktot + fc * 8;
    }

    constexpr int WD = (MQ == 2) ? 4 : (MQ == 3 ? 3 : 2);   // filter prefetch distance in taps
    constexpr int SB = (MQ == 4) ? 8 : 16;              // staging loads in flight per thread
    auto wload = [&](bf16x8 (&dst)[MQ][2], int tap, int c0) {
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                uint4 t = *reinterpret_cast<const uint4*>(wrow[i] + tap * cin + c0 + ks * 32);
                if (!wok[i]) t = make_uint4(0u, 0u, 0u, 0u);
                dst[i][ks] = __builtin_bit_cast(bf16x8, t);
            }
    };

    for (int c0 = 0; c0 < cin; c0 += 64) {
        // ---- stage the halo patch of channels [c0, c0+64)
        float sc[8], sh[8];
        if (has_scale) {
            load8f(d.in_scale + c0 + ck * 8, sc);
            load8f(d.in_shift + c0 + ck * 8, sh);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        }
        if (c0) __syncthreads();                        // every wave is done with the previous chunk's patch
        int hx = tid >> 3, hy = 0;
        while (hx >= HW2) { hx -= HW2; ++hy; }
        // filter fragments of the chunk's first taps: in flight while the patch is staged
        bf16x8 wb[WD + 1][MQ][2];
#pragma unroll
        for (int t = 0; t < WD; ++t) wload(wb[t], t, c0);
        for (int it0 = 0; it0 < nit; it0 += SB) {
            uint4 v[SB];
            bool ok[SB];
            int lds_off[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int pix = (tid >> 3) + (it0 + u) * (NT / 8);
                const int iy = ty0 - dil + hy, ix = tx0 - dil + hx;
                const bool inb = pix < NPIX;
                ok[u] = inb && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                lds_off[u] = inb ? pix * ROWB + ck * 16 : -1;
                const int64_t src = ok[u] ? (((int64_t)n * H + iy) * W + ix) * d.ldx + c0 + ck * 8 : 0;
                v[u] = *reinterpret_cast<const uint4*>(xg + src);
                hx += NT / 8;
                if (hx >= HW2) { hx -= HW2; ++hy; }
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                float f[8];
                unpack8(v[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(fmaf(f[e], sc[e], sh[e]), relu_floor);
                uint4 o = pack8(f);
                const bool k = ok[u];                   // zero padding comes after BatchNorm + ReLU
                o.x = k ? o.x : 0u; o.y = k ? o.y : 0u; o.z = k ? o.z : 0u; o.w = k ? o.w : 0u;
                if (lds_off[u] >= 0) *reinterpret_cast<uint4*>(smem + lds_off[u]) = o;
            }
        }
        __syncthreads();

        // ---- nine taps out of LDS; filter fragments WD taps ahead, straight from global memory
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + WD < 9) wload(wb[(tap + WD) % (WD + 1)], tap + WD, c0);
            const int tyy = tap / 3, txx = tap - tyy * 3;
            const int toff = ((tyy - 1) * dil * HW2 + (txx - 1) * dil) * ROWB;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[MP];
#pragma unroll
                for (int j = 0; j < MP; ++j) af[j] = *reinterpret_cast<const bf16x8*>(smem + rbase[j] + toff + ks * 64);
#pragma unroll
                for (int i = 0; i < MQ; ++i)
#pragma unroll
                    for (int j = 0; j < MP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[tap % (WD + 1)][i][ks], af[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bf16 NHWC store into the slab slice, statistics of the stored values
    const bool want_stats = d.stat_acc != nullptr;
    const int rep = blockIdx.x % DS_R;
    u16* yg = reinterpret_cast<u16*>(d.y);
#pragma unroll
    for (int i = 0; i < MQ; ++i) {
        const int co = i * 16 + fc * 4;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MP; ++j) {
            const int oy = ty0 + 2 * wave + (j >> 1), ox = tx0 + (j & 1) * 16 + fr;
            if (oy < H && ox < W && co < cout) {
                uint2 o;
                o.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
                o.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
                *reinterpret_cast<uint2*>(yg + (((int64_t)n * H + oy) * W + ox) * d.ldy + co) = o;
                const float t0 = bf_lo(o.x), t1 = bf_hi(o.x), t2 = bf_lo(o.y), t3 = bf_hi(o.y);
                s1[0] += t0; s1[1] += t1; s1[2] += t2; s1[3] += t3;
                s2[0] += t0 * t0; s2[1] += t1 * t1; s2[2] += t2 * t2; s2[3] += t3 * t3;
            }
        }
        if (want_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] = row16_sum(s1[e]);
                s2[e] = row16_sum(s2[e]);
            }
            if (fr == 0 && co < cout) {
                float* base = d.stat_acc + (size_t)rep * 2 * cout;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(base + co + e, s1[e]);
                    atomicAdd(base + cout + co + e, s2[e]);
                }
            }
        }
    }
}

template <int MQ>
int launch_halo(const dsaspp_conv_desc& d, hipStream_t s) {
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    const int nblocks = d.n_img * tiles_x * tiles_y;
    const size_t lds = (size_t)(TW + 2 * d.dil) * (TH + 2 * d.dil) * ROWB;
    auto kern = conv3x3_halo_kernel<MQ>;
    static thread_local bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), lds, s, d, tiles_x, tiles_y, nblocks);
    DS_LAUNCH_CHECK("conv_fwd(3x3 halo)");
    return 0;
}

}  // namespace

// -1: not of this kernel's form (the caller falls back to the generic kernel); 0: launched.
int dsaspp_conv3x3_halo_launch(const dsaspp_conv_desc& d, hipStream_t s) {
    const bool form = d.taps == 9 && d.dil >= 1 && d.dil <= 4 && d.cout <= 64 && d.cin % 64 == 0 && !d.in_mask && !d.bias && !d.out_mask &&
                      !d.y_f32 && !d.epi_relu && !d.s_acc && !d.g && d.y;
    if (!form) return -1;
    if (d.cout <= 32) return launch_halo<2>(d, s);
    if (d.cout <= 48) return launch_halo<3>(d, s);
    return launch_halo<4>(d, s);
}
