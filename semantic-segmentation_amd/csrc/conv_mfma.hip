// Fused BN-ReLU -> convolution (1x1 / 3x3 dilated) implicit GEMM on MFMA, gfx950.
//
// GEMM view: P = pixels (N*H*W), Q = output channels, K = taps*cin (tap-major, channel-minor).
// A block computes a BP x BQ tile with 4 waves; operands are staged through LDS in K-steps of
// 32 as [row][32] bf16 rows (64 B) with an XOR chunk swizzle that makes the ds_read_b128
// fragment reads conflict-free.  The activation tile passes through registers on its way to
// LDS, where the consumer's BatchNorm scale/shift, the ReLU and the zero padding of the
// dilated halo are applied, so neither the normalised tensor nor the concatenated input ever
// exists in memory.  Weights are the MFMA A operand (rows = channels) and pixels the B operand,
// so every lane ends up holding 4 consecutive channels of one pixel: 8-byte NHWC stores into
// the channel slice of the feature slab replace torch.cat.
#include "common.h"

namespace {

constexpr int BK = 32;

__device__ __forceinline__ int swz4(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

template <int BP, int BQ, int WP, int WQ>
__global__ __launch_bounds__(256) void conv_kernel(const dsaspp_conv_desc d, const int M, const int nQ,
                                                   const int nblocks) {
    static_assert(WP * WQ == 4, "4 waves");
    constexpr int A_BYTES = BP * BK * 2, W_BYTES = BQ * BK * 2;
    constexpr int AR = BP / 64;                   // activation chunks per thread
    constexpr int WCH = BQ * 4;                   // weight chunks per tile
    constexpr int WR = (WCH + 255) / 256;
    constexpr int TP = BP / WP, TQ = BQ / WQ, MP = TP / 16, MQ = TQ / 16;
    static_assert(TP % 16 == 0 && TQ % 16 == 0, "wave tile");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Ws = smem + 2 * A_BYTES;
    float* tab = reinterpret_cast<float*>(smem + 2 * A_BYTES + 2 * W_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int pt = bid / nQ, qt = bid - pt * nQ;
    const int p_base = pt * BP, q_base = qt * BQ;
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout, taps = d.taps, dil = d.dil;
    const int hw = H * W;
    const int Ktot = taps * cin;
    const int ksteps = (Ktot + BK - 1) / BK;
    const bool has_scale = d.in_scale != nullptr;
    const bool has_pro = has_scale || d.in_relu || d.in_mask != nullptr;

    if (has_scale) {
        for (int c = tid; c < cin; c += 256) {
            tab[c] = d.in_scale[c];
            tab[cin + c] = d.in_shift[c];
        }
    }

    // ---- per-thread staging coordinates
    const int ck = tid & 3;
    int a_n[AR], a_h[AR], a_w[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = p_base + (tid >> 2) + 64 * i;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        a_n[i] = mm / hw;
        const int rem = mm - a_n[i] * hw;
        a_h[i] = rem / W;
        a_w[i] = rem - a_h[i] * W;
    }
    bool w_ok[WR];
    int w_row[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int idx = tid + 256 * i;
        w_row[i] = idx >> 2;
        w_ok[i] = (idx < WCH) && (q_base + w_row[i] < cout);
    }

    const u16* xg = reinterpret_cast<const u16*>(d.x);
    const u16* wg = reinterpret_cast<const u16*>(d.w);

    uint4 areg[AR], wreg[WR];
    bool aval[AR];
    int tap = 0, c = ck * 8;            // (tap, channel) of this thread's chunk in the current K-step
    while (c >= cin) { c -= cin; ++tap; }
    int lc = c;                         // channel of the chunk held in areg (for the prologue)

    auto gload = [&]() {
        int dy = 0, dx = 0;
        if (taps == 9) {
            const int ty = tap / 3;
            dy = (ty - 1) * dil;
            dx = (tap - ty * 3 - 1) * dil;
        }
        const bool kvalid = tap < taps;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int hh = a_h[i] + dy, ww = a_w[i] + dx;
            const bool ok = a_ok[i] && kvalid && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            aval[i] = ok;
            if (ok) {
                const int64_t pix = ((int64_t)a_n[i] * H + hh) * W + ww;
                areg[i] = *reinterpret_cast<const uint4*>(xg + pix * d.ldx + c);
            } else {
                areg[i] = make_uint4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            if (w_ok[i] && kvalid) {
                wreg[i] = *reinterpret_cast<const uint4*>(wg + (int64_t)(q_base + w_row[i]) * Ktot + (int64_t)tap * cin + c);
            } else {
                wreg[i] = make_uint4(0, 0, 0, 0);
            }
        }
        lc = c;
        c += BK;
        while (c >= cin) { c -= cin; ++tap; }
    };

    auto sstore = [&](int buf) {
        float sc[8], sh[8];
        if (has_scale && lc < cin) {
            load8f(tab + lc, sc);
            load8f(tab + cin + lc, sh);
        }
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            uint4 v = areg[i];
            if (has_pro && aval[i]) {
                float f[8];
                unpack8(v, f);
                if (has_scale) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                }
                if (d.in_relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                }
                if (d.in_mask) {
                    float mk[8];
                    load8f(d.in_mask + (int64_t)a_n[i] * d.in_mask_ld + lc, mk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] *= mk[e];
                }
                v = pack8(f);
            }
            const int row = (tid >> 2) + 64 * i;
            *reinterpret_cast<uint4*>(As + buf * A_BYTES + row * 64 + ((ck ^ swz4(row)) << 4)) = v;
        }
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            if (tid + 256 * i < WCH) {
                const int row = w_row[i];
                *reinterpret_cast<uint4*>(Ws + buf * W_BYTES + row * 64 + ((ck ^ swz4(row)) << 4)) = wreg[i];
            }
        }
    };

    // ---- MFMA fragment coordinates
    const int fr = lane & 15, fc = lane >> 4;
    const int wp = wave % WP, wq = wave / WP;
    const int frag_off = fr * 64 + ((fc ^ swz4(fr)) << 4);
    f32x4 acc[MQ][MP];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload();
    __syncthreads();   // scale/shift table visible
    sstore(0);
    __syncthreads();

    for (int s = 0; s < ksteps; ++s) {
        const int buf = s & 1;
        if (s + 1 < ksteps) gload();
        bf16x8 af[MP], wf[MQ];
#pragma unroll
        for (int j = 0; j < MP; ++j)
            af[j] = *reinterpret_cast<const bf16x8*>(As + buf * A_BYTES + (wp * TP + j * 16) * 64 + frag_off);
#pragma unroll
        for (int i = 0; i < MQ; ++i)
            wf[i] = *reinterpret_cast<const bf16x8*>(Ws + buf * W_BYTES + (wq * TQ + i * 16) * 64 + frag_off);
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        if (s + 1 < ksteps) sstore(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane holds channels q..q+3 of pixel p for each (i, j)
    const int rep = blockIdx.x % DS_R;
    const bool want_stats = (d.stat_acc != nullptr) && (d.g == nullptr);
    const bool want_s = d.s_acc != nullptr;
#pragma unroll
    for (int i = 0; i < MQ; ++i) {
        const int q = q_base + wq * TQ + i * 16 + fc * 4;
        const bool q_ok = q < cout;
        float st_sum[4] = {0, 0, 0, 0}, st_sq[4] = {0, 0, 0, 0};   // stats or S1/S2 partials
        float bias4[4] = {0, 0, 0, 0}, es[4] = {1, 1, 1, 1}, eh[4] = {0, 0, 0, 0}, em[4] = {0, 0, 0, 0},
              er[4] = {1, 1, 1, 1};
        if (q_ok) {
            if (d.bias) { const float4 b = *reinterpret_cast<const float4*>(d.bias + q); bias4[0] = b.x; bias4[1] = b.y; bias4[2] = b.z; bias4[3] = b.w; }
            if (d.e_scale) {
                const float4 a = *reinterpret_cast<const float4*>(d.e_scale + q); es[0] = a.x; es[1] = a.y; es[2] = a.z; es[3] = a.w;
                const float4 b = *reinterpret_cast<const float4*>(d.e_shift + q); eh[0] = b.x; eh[1] = b.y; eh[2] = b.z; eh[3] = b.w;
            }
            if (want_s) {
                const float4 a = *reinterpret_cast<const float4*>(d.e_mean + q); em[0] = a.x; em[1] = a.y; em[2] = a.z; em[3] = a.w;
                const float4 b = *reinterpret_cast<const float4*>(d.e_rstd + q); er[0] = b.x; er[1] = b.y; er[2] = b.z; er[3] = b.w;
            }
        }
#pragma unroll
        for (int j = 0; j < MP; ++j) {
            const int p = p_base + wp * TP + j * 16 + fr;
            if (!(q_ok && p < M)) continue;
            float v[4] = {acc[i][j][0] + bias4[0], acc[i][j][1] + bias4[1], acc[i][j][2] + bias4[2], acc[i][j][3] + bias4[3]};
            if (d.out_mask) {
                const int n = p / hw;
                const float4 mk = *reinterpret_cast<const float4*>(d.out_mask + (int64_t)n * d.out_mask_ld + q);
                v[0] *= mk.x; v[1] *= mk.y; v[2] *= mk.z; v[3] *= mk.w;
            }
            float rf[4] = {0, 0, 0, 0};
            if (d.epi_relu || want_s) {
                const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(d.ref) + (int64_t)p * d.ldref + q);
                rf[0] = bf_lo(rr.x); rf[1] = bf_hi(rr.x); rf[2] = bf_lo(rr.y); rf[3] = bf_hi(rr.y);
            }
            if (d.epi_relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pre = d.e_scale ? fmaf(rf[r], es[r], eh[r]) : rf[r];
                    if (!(pre > 0.f)) v[r] = 0.f;
                }
            }
            if (want_s) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    st_sum[r] += v[r];
                    st_sq[r] += v[r] * ((rf[r] - em[r]) * er[r]);
                }
            }
            if (d.g) {
                float* gp = d.g + (int64_t)p * d.ldg + q;
                float4 t = make_float4(v[0] * es[0], v[1] * es[1], v[2] * es[2], v[3] * es[3]);
                if (d.g_accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(gp);
                    t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
                }
                *reinterpret_cast<float4*>(gp) = t;
            } else if (d.y_f32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.y) + (int64_t)p * d.ldy + q) = make_float4(v[0], v[1], v[2], v[3]);
                if (want_stats) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { st_sum[r] += v[r]; st_sq[r] += v[r] * v[r]; }
                }
            } else {
                uint2 o;
                o.x = pack_bf2(v[0], v[1]);
                o.y = pack_bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(d.y) + (int64_t)p * d.ldy + q) = o;
                if (want_stats) {
                    const float s0 = bf_lo(o.x), s1 = bf_hi(o.x), s2 = bf_lo(o.y), s3 = bf_hi(o.y);
                    st_sum[0] += s0; st_sum[1] += s1; st_sum[2] += s2; st_sum[3] += s3;
                    st_sq[0] += s0 * s0; st_sq[1] += s1 * s1; st_sq[2] += s2 * s2; st_sq[3] += s3 * s3;
                }
            }
        }
        if (want_stats || want_s) {
            // reduce over the 16 lanes that share fc (pixels), then one atomic per channel per wave
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    st_sum[r] += __shfl_xor(st_sum[r], o, 64);
                    st_sq[r] += __shfl_xor(st_sq[r], o, 64);
                }
            }
            if (fr == 0 && q_ok) {
                float* base = (want_s ? d.s_acc : d.stat_acc) + (size_t)rep * 2 * cout;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    atomicAdd(base + q + r, st_sum[r]);
                    atomicAdd(base + cout + q + r, st_sq[r]);
                }
            }
        }
    }
}

template <int BP, int BQ, int WP, int WQ>
int launch(const dsaspp_conv_desc& d, hipStream_t s) {
    const int64_t M64 = (int64_t)d.n_img * d.H * d.W;
    const int M = (int)M64;
    const int nP = (M + BP - 1) / BP, nQ = (d.cout + BQ - 1) / BQ;
    const int nblocks = nP * nQ;
    const size_t lds = 2 * (size_t)(BP + BQ) * BK * 2 + (d.in_scale ? (size_t)d.cin * 8 : 0);
    DS_REQUIRE(lds <= 160 * 1024, "conv_fwd: LDS request %zu too large", lds);
    auto kern = conv_kernel<BP, BQ, WP, WQ>;
    if (lds > 64 * 1024) {
        static thread_local bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), lds, s, d, M, nQ, nblocks);
    DS_LAUNCH_CHECK("conv_fwd");
    return 0;
}

}  // namespace

extern "C" int dsaspp_conv_fwd(const dsaspp_conv_desc* dp, dsaspp_stream_t stream) {
    DS_REQUIRE(dp != nullptr, "conv_fwd: null descriptor");
    const dsaspp_conv_desc& d = *dp;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    DS_REQUIRE(d.n_img > 0 && d.H > 0 && d.W > 0, "conv_fwd: bad geometry %d x %d x %d", d.n_img, d.H, d.W);
    DS_REQUIRE((int64_t)d.n_img * d.H * d.W < (1ll << 31), "conv_fwd: too many pixels");
    DS_REQUIRE(d.taps == 1 || d.taps == 9, "conv_fwd: taps must be 1 or 9 (got %d)", d.taps);
    DS_REQUIRE(d.cin > 0 && d.cin % 8 == 0 && d.cout > 0 && d.cout % 8 == 0, "conv_fwd: cin/cout must be positive multiples of 8 (got %d, %d)", d.cin, d.cout);
    DS_REQUIRE(d.ldx % 8 == 0 && d.ldx >= d.cin, "conv_fwd: ldx %d invalid for cin %d", d.ldx, d.cin);
    DS_REQUIRE(d.x && d.w, "conv_fwd: null operand");
    DS_REQUIRE(((uintptr_t)d.x & 15) == 0 && ((uintptr_t)d.w & 15) == 0, "conv_fwd: operands must be 16-byte aligned");
    DS_REQUIRE((d.in_scale == nullptr) == (d.in_shift == nullptr), "conv_fwd: in_scale/in_shift must come together");
    if (d.g) {
        DS_REQUIRE(d.ldg % 4 == 0 && d.ldg >= d.cout && ((uintptr_t)d.g & 15) == 0, "conv_fwd: bad gradient slab stride/alignment");
    } else {
        DS_REQUIRE(d.y != nullptr && d.ldy % 4 == 0 && d.ldy >= d.cout, "conv_fwd: bad output stride");
        DS_REQUIRE(((uintptr_t)d.y & 7) == 0, "conv_fwd: output must be 8-byte aligned");
    }
    if (d.epi_relu || d.s_acc) {
        DS_REQUIRE(d.ref != nullptr && d.ldref % 4 == 0 && ((uintptr_t)d.ref & 7) == 0, "conv_fwd: epilogue reference tensor missing/misaligned");
    }
    if (d.s_acc) DS_REQUIRE(d.e_mean && d.e_rstd, "conv_fwd: s_acc needs e_mean/e_rstd");
    if (d.e_scale) DS_REQUIRE(d.e_shift != nullptr, "conv_fwd: e_scale needs e_shift");
    if (d.in_mask) DS_REQUIRE(d.in_mask_ld % 4 == 0, "conv_fwd: in_mask_ld must be a multiple of 4");
    if (d.out_mask) DS_REQUIRE(d.out_mask_ld % 4 == 0, "conv_fwd: out_mask_ld must be a multiple of 4");

    const int co = d.cout;
    if (co <= 32) return launch<128, 32, 4, 1>(d, s);
    if (co <= 48) return launch<256, 48, 4, 1>(d, s);
    if (co <= 64) return launch<128, 64, 2, 2>(d, s);
    const int pad192 = (co + 191) / 192 * 192, pad128 = (co + 127) / 128 * 128;
    if (pad192 <= pad128) return launch<128, 192, 2, 2>(d, s);
    return launch<128, 128, 2, 2>(d, s);
}
