// BN-ReLU -> 3x3 dilated convolution of the dense layers (wide bottleneck in, growth-rate channels out),
// gfx950.  The generic implicit-GEMM kernel stages the shifted input tile once per filter tap, i.e. nine
// times; here a block owns an 8 x 32 pixel patch of one image and stages the patch plus its dilation halo
// ONCE per 64-channel chunk (BatchNorm scale/shift, ReLU and the zero padding applied on the way into
// LDS), then runs all nine taps out of LDS by shifting the fragment base address.  LDS rows are one pixel
// (64 channels = 128 B) padded to 160 B, which makes any 16 consecutive pixels conflict-free for
// ds_read_b128 whatever the shift.  The chunk's filter slice (nine taps x few output channels) sits beside it
// in XOR-swizzled 128-byte rows.  Patch and filter of the next chunk are loaded into registers while the nine
// taps of the current one run.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32, ROWB = 160;

constexpr int NT = 512;

template <int MQ, int DIL>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo_kernel(const dsaspp_conv_desc d, const int tiles_x, const int tiles_y,
                                                            const int nblocks) {
    constexpr int MP = 2;                               // per wave: one patch row = 2 half rows of 16 pixels
    constexpr int HW2 = TW + 2 * DIL, HH2 = TH + 2 * DIL, NPIX = HW2 * HH2;
    constexpr int NLD = (NPIX * 8 + NT - 1) / NT;       // staged 16-byte pieces of the patch per thread and chunk
    constexpr int WROWS = 9 * MQ * 16;                  // filter rows [tap][channel] of one chunk, 128 B each
    constexpr int NWL = (WROWS * 8 + NT - 1) / NT;
    constexpr int XBYTES = NPIX * ROWB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wlds = smem + XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout;
    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int per_img = tiles_x * tiles_y;
    const int n = bid / per_img, rem = bid - n * per_img;
    const int tyi = rem / tiles_x;
    const int ty0 = tyi * TH, tx0 = (rem - tyi * tiles_x) * TW;
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    const u16* wg = reinterpret_cast<const u16*>(d.w);
    const float relu_floor = d.in_relu ? 0.f : -3.0e38f;
    const bool has_scale = d.in_scale != nullptr;

    // staging: piece u of this thread is chunk column (tid & 7) of halo pixel / filter row (tid >> 3) + 64 u
    const int ck = tid & 7;
    int src_off[NLD], wsrc_off[NWL];
    unsigned okmask = 0, wokmask = 0;
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int pix = (tid >> 3) + u * (NT / 8);
        const int hy = pix / HW2, hx = pix - hy * HW2;
        const int iy = ty0 - DIL + hy, ix = tx0 - DIL + hx;
        const bool ok = pix < NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        okmask |= ok ? (1u << u) : 0u;
        src_off[u] = ok ? ((n * H + iy) * W + ix) * d.ldx + ck * 8 : 0;
    }
#pragma unroll
    for (int u = 0; u < NWL; ++u) {
        const int r = (tid >> 3) + u * (NT / 8);
        const int tap = r / (MQ * 16), co = r - tap * (MQ * 16);
        const bool ok = r < WROWS && co < cout;
        wokmask |= ok ? (1u << u) : 0u;
        wsrc_off[u] = ok ? (co * 9 + tap) * cin + ck * 8 : 0;
    }
    uint4 v[NLD], wv[NWL];
    auto load_raw = [&](int c0) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) v[u] = *reinterpret_cast<const uint4*>(xg + src_off[u] + c0);
#pragma unroll
        for (int u = 0; u < NWL; ++u) wv[u] = *reinterpret_cast<const uint4*>(wg + wsrc_off[u] + c0);
    };

    // fragment coordinates
    const int fr = lane & 15, fc = lane >> 4;
    int rbase[MP];
#pragma unroll
    for (int j = 0; j < MP; ++j) rbase[j] = ((wave + DIL) * HW2 + j * 16 + fr + DIL) * ROWB + fc * 16;
    const int woff0 = fr * 128 + ((fc ^ (fr & 7)) << 4), woff1 = fr * 128 + (((4 + fc) ^ (fr & 7)) << 4);
    f32x4 acc[MQ][MP];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_raw(0);
    for (int c0 = 0; c0 < cin; c0 += 64) {
        // ---- BatchNorm + ReLU + zero padding on the way into LDS: channels [c0, c0+64) of the halo patch
        float sc[8], sh[8];
        if (has_scale) {
            load8f(d.in_scale + c0 + ck * 8, sc);
            load8f(d.in_shift + c0 + ck * 8, sh);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        }
        if (c0) __syncthreads();                        // every wave is done with the previous chunk
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            float f[8];
            unpack8(v[u], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
            uint4 o = relu_bf8(pack8(f), d.in_relu ? 0u : 0x80008000u);     // ReLU after the rounding (packed 16-bit max)
            const bool k = (okmask >> u) & 1u;          // zero padding comes after BatchNorm + ReLU
            o.x = k ? o.x : 0u; o.y = k ? o.y : 0u; o.z = k ? o.z : 0u; o.w = k ? o.w : 0u;
            const int pix = (tid >> 3) + u * (NT / 8);
            if (pix < NPIX) *reinterpret_cast<uint4*>(smem + pix * ROWB + ck * 16) = o;
        }
#pragma unroll
        for (int u = 0; u < NWL; ++u) {
            uint4 o = wv[u];
            const bool k = (wokmask >> u) & 1u;
            o.x = k ? o.x : 0u; o.y = k ? o.y : 0u; o.z = k ? o.z : 0u; o.w = k ? o.w : 0u;
            const int r = (tid >> 3) + u * (NT / 8);
            if (r < WROWS) *reinterpret_cast<uint4*>(wlds + r * 128 + ((ck ^ (r & 7)) << 4)) = o;
        }
        __syncthreads();
        if (c0 + 64 < cin) load_raw(c0 + 64);           // next chunk: in flight during the nine taps

        // ---- nine taps out of LDS.  The fragments of a (tap, K half) group are read one group ahead of the MFMAs
        //      that use them (explicit double buffer, order pinned with scheduling barriers): left to itself the
        //      compiler reads each group right before its MFMAs and the wave waits out an LDS round trip per group.
        bf16x8 afb[2][MP], wfb[2][MQ];
        auto read_group = [&](const int t) {
            const int tap = t >> 1, ks = t & 1;
            const int tyy = tap / 3, txx = tap - tyy * 3;
            const int toff = ((tyy - 1) * DIL * HW2 + (txx - 1) * DIL) * ROWB;
#pragma unroll
            for (int j = 0; j < MP; ++j) afb[t & 1][j] = *reinterpret_cast<const bf16x8*>(smem + rbase[j] + toff + ks * 64);
#pragma unroll
            for (int i = 0; i < MQ; ++i)
                wfb[t & 1][i] = *reinterpret_cast<const bf16x8*>(wlds + (tap * MQ + i) * 2048 + (ks ? woff1 : woff0));
        };
        read_group(0);
#pragma unroll
        for (int t = 0; t < 18; ++t) {
            if (t + 1 < 18) read_group(t + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MQ; ++i)
#pragma unroll
                for (int j = 0; j < MP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfb[t & 1][i], afb[t & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: bf16 NHWC store into the slab slice, statistics of the stored values
    // Statistics go through one LDS reduction per block and leave as a few full-width atomic instructions:
    // float atomics execute at the memory side, an instruction with 4 active lanes costs what a full one does.
    const bool want_stats = d.stat_acc != nullptr;
    const int rep = blockIdx.x % DS_R;
    float* red = reinterpret_cast<float*>(smem);        // [2][MQ*16]
    if (want_stats) {
        __syncthreads();                                // the patch is dead
        if (tid < 2 * MQ * 16) red[tid] = 0.f;
        __syncthreads();
    }
    u16* yg = reinterpret_cast<u16*>(d.y);
#pragma unroll
    for (int i = 0; i < MQ; ++i) {
        const int co = i * 16 + fc * 4;
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < MP; ++j) {
            const int oy = ty0 + wave, ox = tx0 + j * 16 + fr;
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
            if (fr == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(red + co + e, s1[e]);
                    atomicAdd(red + MQ * 16 + co + e, s2[e]);
                }
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        if (tid < 2 * MQ * 16) {
            const int half = tid / (MQ * 16), c = tid - half * (MQ * 16);
            if (c < cout) atomicAdd(d.stat_acc + (size_t)rep * 2 * cout + half * cout + c, red[tid]);
        }
    }
}

template <int MQ, int DIL>
int launch_halo(const dsaspp_conv_desc& d, hipStream_t s) {
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    const int nblocks = d.n_img * tiles_x * tiles_y;
    const size_t lds = (size_t)(TW + 2 * DIL) * (TH + 2 * DIL) * ROWB + (size_t)9 * MQ * 16 * 128;
    if (lds > 160 * 1024) return -1;                    // patch + filter slice do not fit: generic kernel
    auto kern = conv3x3_halo_kernel<MQ, DIL>;
    dsaspp_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(NT), lds, s, d, tiles_x, tiles_y, nblocks);
    DS_LAUNCH_CHECK("conv_fwd(3x3 halo)");
    return 0;
}

template <int MQ>
int launch_dil(const dsaspp_conv_desc& d, hipStream_t s) {
    switch (d.dil) {
        case 1: return launch_halo<MQ, 1>(d, s);
        case 2: return launch_halo<MQ, 2>(d, s);
        case 3: return launch_halo<MQ, 3>(d, s);
        default: return launch_halo<MQ, 4>(d, s);
    }
}

}  // namespace

// -1: not of this kernel's form (the caller falls back to the generic kernel); 0: launched.
int dsaspp_conv3x3_halo_launch(const dsaspp_conv_desc& d, hipStream_t s) {
    const bool form = d.taps == 9 && d.dil >= 1 && d.dil <= 4 && d.cout <= 64 && d.cin % 64 == 0 && !d.in_mask && !d.bias && !d.out_mask &&
                      !d.y_f32 && !d.epi_relu && !d.s_acc && !d.g && d.y;
    if (!form) return -1;
    if ((int64_t)d.n_img * d.H * d.W * d.ldx >= (1ll << 31)) return -1;      // 32-bit element offsets inside
    if (d.cout <= 32) return launch_dil<2>(d, s);
    if (d.cout <= 48) return launch_dil<3>(d, s);
    return launch_dil<4>(d, s);
}
