// Weight gradient of the fused BN-ReLU -> conv layer on MFMA, gfx950.
//
// dw[q][tap][c] += sum_m dz[m][q] * A[m + off(tap)][c],  A = mask * relu(x*scale+shift) recomputed
// here from the slab (the normalised activations are never stored).  GEMM view: rows = output
// channels q, cols = input channels c, K = pixels.  NHWC data is channel-contiguous while both MFMA
// operands need 8 consecutive K (pixels) per lane.  The tiles are therefore staged in their natural
// [pixel][channel] layout -- plain 16-byte chunk copies, the prologue applied on the way exactly as in
// the forward kernel -- and the fragments are fetched with the CDNA4 transposing LDS read
// (ds_read_b64_tr_b16): each 16-lane group reads a 4-pixel x 16-channel block and every lane receives
// the 4 pixels of its own channel.  One block owns a (BQ x BC) tile of one filter tap over a pixel
// range; the partial sums of the pixel ranges are combined with fp32 atomics.  For the narrow 3x3
// layers a block instead owns the three taps of one filter row (RT = 3): a K-step of 64 pixels lies
// inside one image row, so the three horizontally shifted input windows are one staged tile of
// 64 + 2*dilation pixels read at three row offsets, and the dz tile is shared as well.
// The fp32 atomics of the epilogue execute at the memory side (about 1.3 TB/s chip-wide) and their volume is
// (pixel splits) x (weight bytes).  KG = 2 halves it at equal occupancy: a block is two wave groups with private
// staging buffers that take alternate K-steps of the block's pixel range; the second group's accumulators meet
// the first group's through LDS and only one set goes out.
#include "common.h"

namespace {

constexpr int BKP = 64;   // pixels per K-step
constexpr int PADE = 8;   // row padding in elements (16 B): spreads the 4 rows of a transposed read over the banks

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const int half_stride) {
    // two transposed reads: pixels k..k+3 and k+4..k+7 of this lane's channel
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0 + half_stride));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

constexpr int RT_MAXDIL = 8;   // largest dilation of the shared-row form (tile rows = BKP + 2*dilation)

template <int BQ, int BC, int WQ, int WC, int PRO, int RT, int KG>
__global__ __launch_bounds__(64 * WQ * WC * KG, (WQ * WC == 8) ? 4 : 2) void wgrad_kernel(const dsaspp_wgrad_desc d, const int M,
                                                                                   const int nQ, const int nC,
                                                                                   const int chunk) {
    constexpr int NW = WQ * WC, NT = 64 * NW;        // waves / threads of one K group
    constexpr int NTB = NT * KG;                     // threads of the block
    constexpr int ZS = BQ + PADE, XS = BC + PADE;               // LDS row strides in elements
    constexpr int XROWS = (RT == 3) ? BKP + 2 * RT_MAXDIL : BKP;   // staged input rows (upper bound)
    constexpr int Z_BYTES = BKP * ZS * 2, X_BYTES = XROWS * XS * 2;
    constexpr int ZCH = BQ / 8, XCH = BC / 8;                   // 16-byte chunks per pixel row
    constexpr int ZLPR = (ZCH <= 4) ? 4 : (ZCH <= 8) ? 8 : (ZCH <= 16) ? 16 : 32;    // lanes per row (pow2 >= chunks)
    constexpr int XLPR = (XCH <= 4) ? 4 : (XCH <= 8) ? 8 : (XCH <= 16) ? 16 : 32;
    constexpr int ZRPP = NT / ZLPR, XRPP = NT / XLPR;           // rows per pass
    constexpr int ZR = (BKP + ZRPP - 1) / ZRPP, XR = (XROWS + XRPP - 1) / XRPP;
    constexpr int TQ = BQ / WQ, TC = BC / WC, MQ = TQ / 16, MC = TC / 16;
    static_assert(TQ % 16 == 0 && TC % 16 == 0, "wave tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int grp = (KG > 1) ? (int)(threadIdx.x / NT) : 0;            // K group of this wave
    const int tid = (KG > 1) ? (int)(threadIdx.x % NT) : (int)threadIdx.x;   // thread index inside the group
    unsigned char* Zs = smem + grp * (Z_BYTES + X_BYTES);
    unsigned char* Xs = Zs + Z_BYTES;
    float* tab = reinterpret_cast<float*>(smem + KG * (Z_BYTES + X_BYTES));   // scale[BC], shift[BC]

    const int lane = tid & 63, wave = tid >> 6;
    // blocks that share a pixel range (and so the dz tile) get neighbouring logical ids -> same XCD / L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % nC; bid /= nC;
    const int qt = bid % nQ; bid /= nQ;
    const int ntapblk = d.taps / RT;
    const int tap = (bid % ntapblk) * RT;      // first tap of this block
    const int split = bid / ntapblk;
    const int q_base = qt * BQ, c_base = ct * BC;
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout;
    const int hw = H * W;
    const int m_begin = split * chunk;
    const int m_end = min(M, m_begin + chunk);
    const int ksteps = (m_end - m_begin + BKP - 1) / BKP;
    const bool has_scale = d.in_scale != nullptr;
    const bool need_nhw = (d.taps == 9) || (PRO == 2 && d.in_mask != nullptr);
    const float relu_floor = d.in_relu ? 0.f : -3.0e38f;

    if (PRO != 0 && has_scale) {
        for (int c = threadIdx.x; c < BC; c += NTB) {
            const bool ok = c_base + c < cin;
            tab[c] = ok ? d.in_scale[c_base + c] : 0.f;
            tab[BC + c] = ok ? d.in_shift[c_base + c] : 0.f;
        }
    }
    int dy = 0, dx = 0;
    if (d.taps == 9) {
        const int ty = tap / 3;
        dy = (ty - 1) * d.dil;
        dx = (tap - ty * 3 - 1) * d.dil;      // RT == 3: -dil, the leftmost window; tile row r is pixel m0 + dx + r
    }
    const int xrows = (RT == 3) ? BKP + 2 * d.dil : BKP;

    // ---- staging coordinates
    const int zc = tid % ZLPR, zr0 = tid / ZLPR;       // chunk column / first row of the dz tile
    const int xc = tid % XLPR, xr0 = tid / XLPR;
    const bool zc_ok = zc < ZCH && (q_base + zc * 8) < cout;
    const bool xc_ok = xc < XCH && (c_base + xc * 8) < cin;
    const u16* zg = reinterpret_cast<const u16*>(d.dz) + (zc_ok ? q_base + zc * 8 : 0);
    const u16* xg = reinterpret_cast<const u16*>(d.x) + (xc_ok ? c_base + xc * 8 : 0);

    uint4 zreg[ZR], xreg[XR];
    bool xval[XR];
    int ximg[XR];

    auto gload = [&](const int step) {
        const int m0 = m_begin + step * BKP;
#pragma unroll
        for (int i = 0; i < ZR; ++i) {
            const int row = zr0 + ZRPP * i;
            const int m = m0 + row;
            const bool ok = zc_ok && row < BKP && m < m_end;
            uint4 v = *reinterpret_cast<const uint4*>(zg + (int64_t)(ok ? m : m_begin) * d.lddz);
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            zreg[i] = v;
        }
        int n0 = 0, h0 = 0, w0 = 0;
        if (need_nhw) {
            const int mm = min(m0, M - 1);
            n0 = mm / hw;
            const int rem = mm - n0 * hw;
            h0 = rem / W;
            w0 = rem - h0 * W;
        }
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int row = xr0 + XRPP * i;
            const int m = m0 + row;
            bool ok = xc_ok && row < xrows && (RT == 3 || m < m_end);
            int64_t pix = m;
            if (need_nhw) {
                int n = n0, h = h0, w = w0 + row;
                if (RT != 3) {            // RT == 3: the whole K-step lies in image row h0
                    while (w >= W) { w -= W; ++h; }
                    while (h >= H) { h -= H; ++n; }
                }
                const int hh = h + dy, ww = w + dx;
                ok = ok && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
                pix = ((int64_t)n * H + hh) * W + ww;
                ximg[i] = n;
            }
            xreg[i] = *reinterpret_cast<const uint4*>(xg + (ok ? pix : 0) * d.ldx);
            xval[i] = ok;
        }
    };

    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < ZR; ++i) {
            const int row = zr0 + ZRPP * i;
            if (zc < ZCH && row < BKP) *reinterpret_cast<uint4*>(Zs + (row * ZS + zc * 8) * 2) = zreg[i];
        }
        float sc[8], sh[8];
        if (PRO == 1 || (PRO == 2 && has_scale)) {
            load8f(tab + (xc < XCH ? xc : 0) * 8, sc);
            load8f(tab + BC + (xc < XCH ? xc : 0) * 8, sh);
        }
#pragma unroll
        for (int i = 0; i < XR; ++i) {
            const int row = xr0 + XRPP * i;
            uint4 v = xreg[i];
            if (PRO != 0) {
                float f[8];
                unpack8(v, f);
                if (PRO == 1 || has_scale) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                }
                if (PRO == 2 && d.in_mask) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], relu_floor);
                    float mk[8];
                    load8f(d.in_mask + (int64_t)ximg[i] * d.in_mask_ld + (xc_ok ? c_base + xc * 8 : 0), mk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= mk[e];
                    v = pack8(f);
                } else {
                    if (d.in_relu == 2) {      // ReLU6: the ceiling in fp32 (6 is exact in bf16), the floor below
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fminf(f[e], 6.f);
                    }
                    v = relu_bf8(pack8(f), d.in_relu ? 0u : 0x80008000u);     // ReLU after the rounding (packed 16-bit max)
                }
            }
            const bool ok = xval[i];       // zero padding comes after the prologue
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            if (xc < XCH && row < xrows) *reinterpret_cast<uint4*>(Xs + (row * XS + xc * 8) * 2) = v;
        }
    };

    // ---- fragment addressing: lane (g = lane/16, i16 = lane%16) supplies row g*8 + i16/4 (+4 for the
    // second half), columns 4*(i16%4).. of a 16-channel block and receives channel i16, pixels g*8..g*8+7.
    const int g16 = lane >> 4, i16 = lane & 15;
    const int wq = wave % WQ, wc = wave / WQ;
    const int zfrag = ((g16 * 8 + (i16 >> 2)) * ZS + wq * TQ + (i16 & 3) * 4) * 2;
    const int xfrag = ((g16 * 8 + (i16 >> 2)) * XS + wc * TC + (i16 & 3) * 4) * 2;
    f32x4 acc[RT][MQ][MC];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MC; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tstride = (RT == 3) ? d.dil * XS * 2 : 0;     // byte offset between the windows of neighbouring taps

    // group g takes K-steps g, g+KG, ...; barriers are block-wide, so both groups run the same trip count
    {
        const int trips = (ksteps + KG - 1) / KG;
        if (grp < ksteps) gload(grp);
        __syncthreads();      // scale/shift table visible
        for (int it = 0; it < trips; ++it) {
            const int s = it * KG + grp;
            const bool active = s < ksteps;
            if (active) sstore();
            __syncthreads();
            if (s + KG < ksteps) gload(s + KG);
            if (active) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 zf[MQ];
#pragma unroll
                    for (int i = 0; i < MQ; ++i) zf[i] = tr_frag(Zs + zfrag + (ks * 32 * ZS + i * 16) * 2, 4 * ZS * 2);
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        bf16x8 xf[MC];
#pragma unroll
                        for (int j = 0; j < MC; ++j) xf[j] = tr_frag(Xs + xfrag + t * tstride + (ks * 32 * XS + j * 16) * 2, 4 * XS * 2);
#pragma unroll
                        for (int i = 0; i < MQ; ++i)
#pragma unroll
                            for (int j = 0; j < MC; ++j)
                                acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[i], xf[j], acc[t][i][j], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    }
    if (KG > 1) {
        // the second group's partial sums meet the first group's in LDS (same thread index = same outputs)
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        if (grp == 1) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int i = 0; i < MQ; ++i)
#pragma unroll
                    for (int j = 0; j < MC; ++j) red[((t * MQ + i) * MC + j) * NT + tid] = acc[t][i][j];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < MQ; ++i)
#pragma unroll
                for (int j = 0; j < MC; ++j) acc[t][i][j] += red[((t * MQ + i) * MC + j) * NT + tid];
    }

    // ---- epilogue: acc[i][j][r] is (q = q0 + i*16 + fc*4 + r, c = c0 + j*16 + fr)
    const int fr = lane & 15, fc = lane >> 4;
    const int Ktot = d.taps * cin;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#pragma unroll
        for (int i = 0; i < MQ; ++i) {
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                const int c = c_base + wc * TC + j * 16 + fr;
                if (c >= cin) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = q_base + wq * TQ + i * 16 + fc * 4 + r;
                    if (q < cout) atomicAdd(d.dw + (int64_t)q * Ktot + (int64_t)(tap + t) * cin + c, acc[t][i][j][r]);
                }
            }
        }
    }
}

template <int BQ, int BC, int WQ, int WC, int PRO, int RT>
int launch_rt(const dsaspp_wgrad_desc& d, hipStream_t s) {
    constexpr int KG = (RT == 3) ? 2 : 1;      // two lock-stepped 8-wave groups lose more overlap than the atomics cost
    constexpr int MQ = BQ / WQ / 16, MC = BC / WC / 16;
    const int M = d.n_img * d.H * d.W;
    const int nQ = (d.cout + BQ - 1) / BQ, nC = (d.cin + BC - 1) / BC;
    const int base_blocks = nQ * nC * (d.taps / RT);
    int ksplit = d.ksplit;
    const int max_split = (M + 4 * BKP - 1) / (4 * BKP);   // at least 4 K-steps per block
    if (ksplit <= 0) ksplit = (512 / KG) / base_blocks;    // one round of resident blocks: the fewest atomic epilogues
    if (ksplit > max_split) ksplit = max_split;
    if (ksplit < 1) ksplit = 1;
    int chunk = (M + ksplit - 1) / ksplit;
    chunk = (chunk + BKP - 1) / BKP * BKP;
    ksplit = (M + chunk - 1) / chunk;
    size_t lds = KG * ((size_t)BKP * (BQ + PADE) * 2 + (size_t)(RT == 3 ? BKP + 2 * RT_MAXDIL : BKP) * (BC + PADE) * 2) +
                 ((PRO != 0 && d.in_scale) ? (size_t)BC * 8 : 0);
    const size_t red = (KG > 1) ? (size_t)RT * MQ * MC * 16 * 64 * WQ * WC : 0;      // accumulator exchange
    if (red > lds) lds = red;
    DS_REQUIRE(lds <= 160 * 1024, "conv_wgrad: LDS request %zu too large", lds);
    auto kern = wgrad_kernel<BQ, BC, WQ, WC, PRO, RT, KG>;
    if (lds > 64 * 1024) dsaspp_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(base_blocks * ksplit), dim3(64 * WQ * WC * KG), lds, s, d, M, nQ, nC, chunk);
    DS_LAUNCH_CHECK("conv_wgrad");
    return 0;
}

template <int BQ, int BC, int WQ, int WC, int PRO>
int launch_pro(const dsaspp_wgrad_desc& d, hipStream_t s) {
    // shared-row form: narrow 3x3 layers whose 64-pixel K-steps never straddle an image row
    if (BQ <= 64 && d.taps == 9 && d.W % BKP == 0 && d.dil <= RT_MAXDIL) return launch_rt<BQ, BC, WQ, WC, PRO, (BQ <= 64 ? 3 : 1)>(d, s);
    return launch_rt<BQ, BC, WQ, WC, PRO, 1>(d, s);
}

template <int BQ, int BC, int WQ, int WC>
int launch(const dsaspp_wgrad_desc& d, hipStream_t s) {
    if (d.in_mask) return launch_pro<BQ, BC, WQ, WC, 2>(d, s);
    if (d.in_scale) return launch_pro<BQ, BC, WQ, WC, 1>(d, s);
    if (d.in_relu) return launch_pro<BQ, BC, WQ, WC, 2>(d, s);
    return launch_pro<BQ, BC, WQ, WC, 0>(d, s);
}

}  // namespace

extern "C" int dsaspp_conv_wgrad(const dsaspp_wgrad_desc* dp, dsaspp_stream_t stream) {
    DS_REQUIRE(dp != nullptr, "conv_wgrad: null descriptor");
    const dsaspp_wgrad_desc& d = *dp;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DS_REQUIRE(d.n_img > 0 && d.H > 0 && d.W > 0, "conv_wgrad: bad geometry");
    DS_REQUIRE((int64_t)d.n_img * d.H * d.W < (1ll << 31), "conv_wgrad: too many pixels");
    DS_REQUIRE(d.taps == 1 || d.taps == 9, "conv_wgrad: taps must be 1 or 9 (got %d)", d.taps);
    DS_REQUIRE(d.cin > 0 && d.cin % 8 == 0 && d.cout > 0 && d.cout % 8 == 0, "conv_wgrad: cin/cout must be positive multiples of 8 (got %d, %d)", d.cin, d.cout);
    DS_REQUIRE(d.ldx % 8 == 0 && d.ldx >= d.cin && d.lddz % 8 == 0 && d.lddz >= d.cout, "conv_wgrad: bad strides");
    DS_REQUIRE(d.x && d.dz && d.dw, "conv_wgrad: null operand");
    DS_REQUIRE(((uintptr_t)d.x & 15) == 0 && ((uintptr_t)d.dz & 15) == 0, "conv_wgrad: operands must be 16-byte aligned");
    DS_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr), "conv_wgrad: in_scale/in_shift must come together");
    if (d.in_mask) DS_REQUIRE(d.in_mask_ld % 4 == 0, "conv_wgrad: in_mask_ld must be a multiple of 4");
    DS_REQUIRE(d.dtype == 0 || d.dtype == 1, "conv_wgrad: dtype must be 0 (bf16) or 1 (fp32)");
    if (d.dtype == 1) {
        dsaspp_note_route("wgrad_f32");
        return dsaspp_f32_wgrad(d, s);
    }
    if (d.n_src >= 2) {
        dsaspp_note_route("wgrad_group");
        return dsaspp_wgrad_group_launch(d, s);
    }
    if (d.taps == 1) {
        const int rc = dsaspp_wgrad_wide_launch(d, s);      // many input and output channels (transitions, ASPP 1x1): 256 x 256 tile
        if (rc >= 0) {
            dsaspp_note_route("wgrad_wide");
            return rc;
        }
    }
    if (d.taps == 9) {
        const int rc = dsaspp_wgrad3x3_strip_launch(d, s);  // dense-layer 3x3 (cout 32 / 48, dilation <= 4, 32 | W): column strips, all nine taps per activation row
        if (rc >= 0) {
            dsaspp_note_route("wgrad3x3_strip");
            return rc;
        }
    }
    if (d.taps == 9) {
        const int rc = dsaspp_wgrad_wide_launch(d, s);      // many input channels (ASPP 3x3): 256 x 256 tile over nine shifted copies of dz
        if (rc >= 0) {
            dsaspp_note_route("wgrad_wide9");
            return rc;
        }
    }
    dsaspp_note_route("wgrad_tile");
    const int co = d.cout;
    if (co <= 32) return launch<32, 192, 1, 4>(d, s);
    if (co <= 48) return launch<48, 192, 1, 4>(d, s);
    if (co <= 64) return launch<64, 192, 1, 4>(d, s);
    if (co > 128 && (co + 191) / 192 * 192 <= (co + 127) / 128 * 128) return launch<192, 128, 4, 2>(d, s);
    return launch<128, 128, 2, 4>(d, s);
}
