// Weight gradient of the fused BN-ReLU -> conv layer on MFMA, gfx950.
//
// dw[q][tap][c] += sum_m dz[m][q] * A[m + off(tap)][c],  A = mask * relu(x*scale+shift) recomputed
// here from the slab (the normalised activations are never stored).  GEMM view: rows = output
// channels q, cols = input channels c, K = pixels.  NHWC data is channel-contiguous but the MFMA
// fragments need 8 consecutive K (pixels) per lane, so each thread loads an 8-pixel x 8-channel
// patch (eight 16-byte loads), transposes it in registers while applying the prologue, and writes
// eight [channel][8 pixels] chunks into a [channel][64 pixel] LDS image (128-byte rows, XOR
// swizzled).  One block owns a (BQ x BC) tile of one tap over a pixel range; partial sums of the
// pixel ranges are combined with fp32 atomics.
#include "common.h"

namespace {

constexpr int BKP = 64;  // pixels per K-step

__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

template <int BQ, int BC, int WQ, int WC>
__global__ __launch_bounds__(256) void wgrad_kernel(const dsaspp_wgrad_desc d, const int M, const int nQ,
                                                    const int nC, const int chunk) {
    static_assert(BQ + BC == 256, "one 8x8 patch per thread");
    static_assert(WQ * WC == 4, "4 waves");
    constexpr int Z_BYTES = BQ * BKP * 2, X_BYTES = BC * BKP * 2;
    constexpr int TQ = BQ / WQ, TC = BC / WC, MQ = TQ / 16, MC = TC / 16;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Zs = smem;
    unsigned char* Xs = smem + 2 * Z_BYTES;
    float* tab = reinterpret_cast<float*>(smem + 2 * Z_BYTES + 2 * X_BYTES);  // scale[BC], shift[BC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // blocks that share a pixel range (and so the dz tile) get neighbouring logical ids -> same XCD / L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % nC; bid /= nC;
    const int qt = bid % nQ; bid /= nQ;
    const int tap = bid % d.taps;
    const int split = bid / d.taps;
    const int q_base = qt * BQ, c_base = ct * BC;
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout;
    const int hw = H * W;
    const int m_begin = split * chunk;
    const int m_end = min(M, m_begin + chunk);
    const int ksteps = (m_end - m_begin + BKP - 1) / BKP;
    const bool has_scale = d.in_scale != nullptr;
    const bool has_pro = has_scale || d.in_relu || d.in_mask != nullptr;
    const bool need_nhw = (d.taps == 9) || (d.in_mask != nullptr);

    if (has_scale) {
        for (int c = tid; c < BC; c += 256) {
            const bool ok = c_base + c < cin;
            tab[c] = ok ? d.in_scale[c_base + c] : 0.f;
            tab[BC + c] = ok ? d.in_shift[c_base + c] : 0.f;
        }
    }

    int dy = 0, dx = 0;
    if (d.taps == 9) {
        const int ty = tap / 3;
        dy = (ty - 1) * d.dil;
        dx = (tap - ty * 3 - 1) * d.dil;
    }

    // ---- staging role: threads [0,BQ) transpose dz, threads [BQ,256) transpose the activations
    const bool is_z = tid < BQ;
    const int t2 = is_z ? tid : tid - BQ;
    const int po = t2 & 7;              // pixel octet within the 64-pixel step
    const int cg = t2 >> 3;             // channel group (8 channels)
    const int ch0 = (is_z ? q_base : c_base) + cg * 8;
    const bool ch_ok = ch0 < (is_z ? cout : cin);
    const u16* zg = reinterpret_cast<const u16*>(d.dz);
    const u16* xg = reinterpret_cast<const u16*>(d.x);

    uint4 reg[8];
    unsigned okmask = 0;     // bit j: pixel j of the patch is a real (in-bounds) sample
    int img[8];              // image index per pixel (only kept for in_mask)

    auto gload = [&](int step) {
        const int m0 = m_begin + step * BKP + po * 8;
        okmask = 0;
        if (is_z) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = m0 + j;
                if (ch_ok && m < m_end) {
                    reg[j] = *reinterpret_cast<const uint4*>(zg + (int64_t)m * d.lddz + ch0);
                } else {
                    reg[j] = make_uint4(0, 0, 0, 0);
                }
            }
        } else {
            int n = 0, h = 0, w = 0;
            if (need_nhw) {
                const int mm = min(m0, M - 1);
                n = mm / hw;
                const int rem = mm - n * hw;
                h = rem / W;
                w = rem - h * W;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int m = m0 + j;
                bool ok = ch_ok && m < m_end;
                int64_t pix = m;
                if (need_nhw) {
                    const int hh = h + dy, ww = w + dx;
                    ok = ok && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
                    pix = ((int64_t)n * H + hh) * W + ww;
                    img[j] = n;
                    // advance (n, h, w) to the next pixel
                    if (++w == W) { w = 0; if (++h == H) { h = 0; ++n; } }
                }
                if (ok) {
                    reg[j] = *reinterpret_cast<const uint4*>(xg + pix * d.ldx + ch0);
                    okmask |= 1u << j;
                } else {
                    reg[j] = make_uint4(0, 0, 0, 0);
                }
            }
        }
    };

    auto sstore = [&](int buf) {
        unsigned char* base = is_z ? (Zs + buf * Z_BYTES) : (Xs + buf * X_BYTES);
        uint32_t out[8][4];   // [channel e][pixel pair k]
        if (is_z || !has_pro) {
            const uint32_t* r32 = reinterpret_cast<const uint32_t*>(reg);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t a = r32[(2 * k) * 4 + (e >> 1)], b = r32[(2 * k + 1) * 4 + (e >> 1)];
                    out[e][k] = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
                }
            }
        } else {
            float sc[8], sh[8];
            if (has_scale) {
                load8f(tab + cg * 8, sc);
                load8f(tab + BC + cg * 8, sh);
            }
            float f[8][8];   // [pixel j][channel e]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unpack8(reg[j], f[j]);
                const bool ok = (okmask >> j) & 1;
                float mk[8];
                if (d.in_mask && ok) load8f(d.in_mask + (int64_t)img[j] * d.in_mask_ld + ch0, mk);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = f[j][e];
                    if (has_scale) v = fmaf(v, sc[e], sh[e]);
                    if (d.in_relu) v = fmaxf(v, 0.f);
                    if (d.in_mask && ok) v *= mk[e];
                    f[j][e] = ok ? v : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int k = 0; k < 4; ++k) out[e][k] = pack_bf2(f[2 * k][e], f[2 * k + 1][e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int row = cg * 8 + e;
            *reinterpret_cast<uint4*>(base + row * 128 + ((po ^ swz8(row)) << 4)) =
                make_uint4(out[e][0], out[e][1], out[e][2], out[e][3]);
        }
    };

    const int fr = lane & 15, fc = lane >> 4;
    const int wq = wave % WQ, wc = wave / WQ;
    f32x4 acc[MQ][MC];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < MC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (ksteps > 0) {
        gload(0);
        __syncthreads();
        sstore(0);
        __syncthreads();
        for (int s = 0; s < ksteps; ++s) {
            const int buf = s & 1;
            if (s + 1 < ksteps) gload(s + 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 zf[MQ], xf[MC];
                const int chunkk = ks * 4 + fc;
#pragma unroll
                for (int i = 0; i < MQ; ++i) {
                    const int row = wq * TQ + i * 16 + fr;
                    zf[i] = *reinterpret_cast<const bf16x8*>(Zs + buf * Z_BYTES + row * 128 + ((chunkk ^ swz8(row)) << 4));
                }
#pragma unroll
                for (int j = 0; j < MC; ++j) {
                    const int row = wc * TC + j * 16 + fr;
                    xf[j] = *reinterpret_cast<const bf16x8*>(Xs + buf * X_BYTES + row * 128 + ((chunkk ^ swz8(row)) << 4));
                }
#pragma unroll
                for (int i = 0; i < MQ; ++i)
#pragma unroll
                    for (int j = 0; j < MC; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[i], xf[j], acc[i][j], 0, 0, 0);
            }
            if (s + 1 < ksteps) sstore(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: acc[i][j][r] is (q = q0 + i*16 + fc*4 + r, c = c0 + j*16 + fr)
    const int Ktot = d.taps * cin;
#pragma unroll
    for (int i = 0; i < MQ; ++i) {
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            const int c = c_base + wc * TC + j * 16 + fr;
            if (c >= cin) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q_base + wq * TQ + i * 16 + fc * 4 + r;
                if (q < cout) atomicAdd(d.dw + (int64_t)q * Ktot + (int64_t)tap * cin + c, acc[i][j][r]);
            }
        }
    }
}

template <int BQ, int BC, int WQ, int WC>
int launch(const dsaspp_wgrad_desc& d, hipStream_t s) {
    const int M = d.n_img * d.H * d.W;
    const int nQ = (d.cout + BQ - 1) / BQ, nC = (d.cin + BC - 1) / BC;
    const int base_blocks = nQ * nC * d.taps;
    int ksplit = d.ksplit;
    const int max_split = (M + 4 * BKP - 1) / (4 * BKP);   // at least 4 K-steps per block
    if (ksplit <= 0) ksplit = 1024 / base_blocks;      // at most 1024 blocks: two full rounds at 2 blocks per CU
    if (ksplit > max_split) ksplit = max_split;
    if (ksplit < 1) ksplit = 1;
    int chunk = (M + ksplit - 1) / ksplit;
    chunk = (chunk + BKP - 1) / BKP * BKP;
    ksplit = (M + chunk - 1) / chunk;
    const size_t lds = 2 * (size_t)(BQ + BC) * BKP * 2 + (d.in_scale ? (size_t)BC * 8 : 0);
    auto kern = wgrad_kernel<BQ, BC, WQ, WC>;
    static thread_local bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
    hipLaunchKernelGGL(kern, dim3(base_blocks * ksplit), dim3(256), lds, s, d, M, nQ, nC, chunk);
    DS_LAUNCH_CHECK("conv_wgrad");
    return 0;
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
    if (d.cout <= 64) return launch<64, 192, 1, 4>(d, s);
    if (d.cout > 128 && (d.cout + 191) / 192 * 192 <= (d.cout + 127) / 128 * 128) return launch<192, 64, 2, 2>(d, s);
    return launch<128, 128, 2, 2>(d, s);
}
