// Data gradient of a dense layer's 3x3 convolution (growth-rate channels in, bottleneck channels out) fused with
// the read-out of its input from the gradient slab, gfx950.
//
//   v[p][c]  = g[p][c] - corr1[c] - ((xs[p][c] - mean[c]) * rstd[c]) * corr2[c]          (c < CIN; rounded to bf16)
//   dr[p][q] = [ref[p][q] * e_scale[q] + e_shift[q] > 0] * sum_{tap, c} v[p + off(tap)][c] * w[q][tap][c]
//   s_acc   += { sum_p dr[p][q], sum_p dr[p][q] * (ref[p][q] - e_mean[q]) * e_rstd[q] }
//
// The input has only 32 or 48 channels, so all nine taps of a pixel patch come from ONE staged copy of the patch
// and its dilation halo: a block owns 8 x 32 pixels of one image, reads the halo patch of the two slab slices
// (gradient and activations) once, applies the read-out on the way into LDS and writes the patch's own pixels
// out as the contiguous bf16 tensor the weight gradient of the same layer reads.  The filter is streamed in three
// phases of one filter row (three taps, K = 3*CIN) each; within a phase K runs across the taps in steps of 32, every
// 8-channel chunk of a step finding its own tap (CIN = 48: 4.5 steps per phase instead of 6).  A wave owns one
// patch row (32 pixels) and all output channels.
#include "common.h"

#ifndef D3F_ABL
#define D3F_ABL 0     // diagnostic builds (timing only, results wrong): 1 no slab loads for the patch, 2 no MFMA, 3 no epilogue loads / stores
#endif

#ifndef D3F_STAMP
#define D3F_STAMP 0   // diagnostic build: wave 0 of every block records the shader clock at its phase boundaries (tools/d3f_stamps.py)
#endif
#if D3F_STAMP
__device__ unsigned long long d3f_stamps[4096 * 16];
extern "C" int dsaspp_debug_read_stamps(void* host, size_t bytes) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(d3f_stamps), bytes, 0, hipMemcpyDeviceToHost);
}
#define STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) d3f_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

namespace {

constexpr int TH = 8, TW = 32, NT = 512;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int CIN, int COUT, int DIL>
struct Geo {
    static constexpr int HW2 = TW + 2 * DIL, HH2 = TH + 2 * DIL, NPIX = HW2 * HH2;
    static constexpr int CPP = CIN / 8;                    // 16-byte chunks per pixel
    // LDS row lengths in 16-byte slots are == 2 (mod 4).  A ds_read_b128 is served in groups of 16 lanes made of 8
    // fragment rows at K chunk c and the other 8 rows of the 16-row tile at chunk c + 1 ({0-3, 12-15} with {4-11}, or
    // the reverse): with such a row length the first eight land on the even slots of the 256-byte bank window and
    // the other eight on the odd ones, whatever c is.
    static constexpr int PSL = CPP + ((2 - CPP % 4) + 4) % 4;
    static constexpr int PROWB = PSL * 16;                 // LDS bytes per patch pixel
    static constexpr int XBYTES = NPIX * PROWB;
    static constexpr int KPH = 3 * CIN;                    // K of one phase (one filter row)
    static constexpr int NS = (KPH + 31) / 32;             // 32-wide K-steps per phase
    static constexpr int WCPR = KPH / 8;                   // 16-byte chunks per filter row and phase
    static constexpr int WSL = WCPR + ((2 - WCPR % 4) + 4) % 4;
    static constexpr int WROWB = WSL * 16;
    static constexpr int WBYTES = COUT * WROWB;
    static constexpr int TABF = 4 * COUT + 4 * CIN;        // e_scale, e_shift, e_mean, e_rstd | corr1, corr2, mean, rstd
    static constexpr int LDS = XBYTES + WBYTES + TABF * 4;
    static_assert(PSL % 4 == 2 && WSL % 4 == 2 && PSL >= CPP && WSL >= WCPR, "slot counts");
    static_assert(NT / 64 * 2 * COUT * 4 <= XBYTES, "the per-wave sums reuse the patch");
    static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int CIN, int COUT, int DIL>
__global__ __launch_bounds__(NT, 2) void dgrad3x3_fused_kernel(const dsaspp_conv_desc d, const int tiles_x, const int tiles_y, const int nblocks) {
    typedef Geo<CIN, COUT, DIL> G;
    constexpr int MQ = COUT / 16, MP = 2;
    constexpr int NWL = (COUT * G::WCPR + NT - 1) / NT;           // staged filter pieces per thread and phase
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* wlds = smem + G::XBYTES;
    float* tab = reinterpret_cast<float*>(smem + G::XBYTES + G::WBYTES);
    float* gtab = tab + 4 * COUT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = d.H, W = d.W;
    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int per_img = tiles_x * tiles_y;
    const int n = bid / per_img, rem = bid - n * per_img;
    const int tyi = rem / tiles_x;
    const int ty0 = tyi * TH, tx0 = (rem - tyi * tiles_x) * TW;

    for (int c = tid; c < COUT; c += NT) {
        tab[c] = d.e_scale[c];
        tab[COUT + c] = d.e_shift[c];
        tab[2 * COUT + c] = d.e_mean[c];
        tab[3 * COUT + c] = d.e_rstd[c];
    }
    for (int c = tid; c < CIN; c += NT) {
        gtab[c] = d.gx_corr1[c];
        gtab[CIN + c] = d.gx_corr2[c];
        gtab[2 * CIN + c] = d.gx_mean[c];
        gtab[3 * CIN + c] = d.gx_rstd[c];
    }

    // ---- filter pieces of a phase: [COUT][3 taps][CIN] of w[COUT][9][CIN], through registers
    const u16* wg = reinterpret_cast<const u16*>(d.w);
    u32x4 wv[NWL];
    auto load_w = [&](const int ph) {
#pragma unroll
        for (int u = 0; u < NWL; ++u) {
            const int idx = tid + u * NT;
            const int row = idx / G::WCPR, c = idx - row * G::WCPR;
            const bool ok = idx < COUT * G::WCPR;
            wv[u] = *reinterpret_cast<const u32x4*>(wg + (ok ? (size_t)row * (9 * CIN) + ph * G::KPH + c * 8 : 0));
        }
    };
    // Output channel q sits in LDS row tile(q) * 16 + r(q) with q = 32 P + 8 gq + 4 h + e  ->  tile 2 P + h, row 4 gq + e: a lane of
    // the accumulator layout (rows 4 g .. 4 g + 3 of every tile) then holds channels 32 P + 8 g .. + 7 in tiles 2 P and 2 P + 1 --
    // sixteen consecutive bytes per pixel, so the epilogue loads and stores 16-byte pieces (64 consecutive bytes per pixel and
    // instruction) instead of 8-byte ones.
    auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < NWL; ++u) {
            const int idx = tid + u * NT;
            const int q = idx / G::WCPR, c = idx - q * G::WCPR;
            const int row = ((q >> 5) * 2 + ((q >> 2) & 1)) * 16 + ((q >> 3) & 3) * 4 + (q & 3);
            if (idx < COUT * G::WCPR) *reinterpret_cast<u32x4*>(wlds + row * G::WROWB + c * 16) = wv[u];
        }
    };
    STAMP(0);
    load_w(0);
    __syncthreads();                                    // read-out tables visible
    STAMP(1);

    // ---- the halo patch: read-out applied on the way into LDS, zero padding after it; own pixels also go to gx_dz.
    //      Every load of the patch is issued before the first one is used.  A thread keeps one 8-channel chunk
    //      position for all its pieces (the last NT % CPP threads idle), so the read-out coefficients are loaded once.
    {
        constexpr int NTS = NT / G::CPP * G::CPP, PPP = NTS / G::CPP;          // staging threads, pixels per pass
        constexpr int NU = (G::NPIX + PPP - 1) / PPP;
        const u16* gs = reinterpret_cast<const u16*>(d.x);
        const u16* xs = reinterpret_cast<const u16*>(d.gx_xs);
        u16* dz = reinterpret_cast<u16*>(d.gx_dz);
        const bool stg = tid < NTS;
        const int ck = tid % G::CPP, pix0 = tid / G::CPP;
        u32x4 gv[NU], xv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int pix = pix0 + u * PPP;
            const int hy = pix / G::HW2, hx = pix - hy * G::HW2;
            const int iy = ty0 - DIL + hy, ix = tx0 - DIL + hx;
            const bool ok = stg && pix < G::NPIX && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && D3F_ABL != 1;
            const int64_t p = ok ? ((int64_t)n * H + iy) * W + ix : 0;       // always a mapped address
            gv[u] = *reinterpret_cast<const u32x4*>(gs + p * d.ldx + (ok ? ck * 8 : 0));
            xv[u] = *reinterpret_cast<const u32x4*>(xs + p * d.gx_ldxs + (ok ? ck * 8 : 0));
        }
        // v = g - corr1 - ((x - mean) * rstd) * corr2 = g + x * al + be, with al = -rstd * corr2, be = mean * rstd * corr2 - corr1
        float al[8], be[8];
        {
            float k1[8], k2[8], sm[8], sr[8];
            load8f(gtab + ck * 8, k1);
            load8f(gtab + CIN + ck * 8, k2);
            load8f(gtab + 2 * CIN + ck * 8, sm);
            load8f(gtab + 3 * CIN + ck * 8, sr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                al[e] = -sr[e] * k2[e];
                be[e] = fmaf(-sm[e], al[e], -k1[e]);
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int pix = pix0 + u * PPP;
            const int hy = pix / G::HW2, hx = pix - hy * G::HW2;
            const int iy = ty0 - DIL + hy, ix = tx0 - DIL + hx;
            const bool inr = stg && pix < G::NPIX;
            const bool ok = inr && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            float v[8], x[8];
            unpack8(uint4{gv[u].x, gv[u].y, gv[u].z, gv[u].w}, v);
            unpack8(uint4{xv[u].x, xv[u].y, xv[u].z, xv[u].w}, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += fmaf(x[e], al[e], be[e]);
            uint4 o = pack8(v);
            o.x = ok ? o.x : 0u; o.y = ok ? o.y : 0u; o.z = ok ? o.z : 0u; o.w = ok ? o.w : 0u;
            const bool own = ok && hy >= DIL && hy < DIL + TH && hx >= DIL && hx < DIL + TW;
            if (own && dz) *reinterpret_cast<uint4*>(dz + (((int64_t)n * H + iy) * W + ix) * d.gx_lddz + ck * 8) = o;
            if (inr) *reinterpret_cast<uint4*>(smem + pix * G::PROWB + ck * 16) = o;
        }
    }

    STAMP(2);                                           // patch staged (this wave's share)
    // ---- fragment coordinates: lane (fr, g) = row fr of a 16-row tile, K chunk g of a 32-wide step
    const int fr = lane & 15, g = lane >> 4;
    f32x4 acc[MQ][MP];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pbase = ((wave + DIL) * G::HW2 + fr + DIL) * G::PROWB;      // this lane's pixel of row tile 0, no shift

    // Reference activations of the epilogue: the first half of the channel tiles is requested before the last phase
    // (whose filter prefetch registers are free), the second half after it
    const u16* rg = reinterpret_cast<const u16*>(d.ref);
    const int oy = ty0 + wave;
    int64_t poff[MP];
    bool pok[MP];
#pragma unroll
    for (int j = 0; j < MP; ++j) {
        const int ox = tx0 + j * 16 + fr;
        pok[j] = oy < H && ox < W;
        poff[j] = pok[j] ? ((int64_t)n * H + oy) * W + ox : 0;
    }
    constexpr int NPR = MQ / 2;                          // pairs of channel tiles
    static_assert(MQ % 2 == 0 && COUT % 32 == 0, "channel tiles are paired");
    uint4 rr[NPR][MP];
    auto load_ref = [&](const int p0, const int p1) {
#pragma unroll
        for (int pr = 0; pr < NPR; ++pr)
#pragma unroll
            for (int j = 0; j < MP; ++j)
                if (pr >= p0 && pr < p1)
                    rr[pr][j] = D3F_ABL == 3 ? make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u)
                                             : *reinterpret_cast<const uint4*>(rg + poff[j] * d.ldref + pr * 32 + g * 8);
    };

    // Fragments are read in groups (the step's two pixel fragments, then WGN filter fragments at a time), one group
    // ahead of the MFMAs that use them: explicit double buffer, order pinned with scheduling barriers, so that a wave
    // has LDS reads in flight while its matrix pipe is busy.
    constexpr int WGN = 4, NG = MQ / WGN, NGT = G::NS * NG;
    static_assert(MQ % WGN == 0, "filter fragment groups");
    const int g8 = 8 * g;
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        if (ph) __syncthreads();                        // every wave has left the previous phase's filter rows
        STAMP(3 + 3 * ph);
        store_w();
        __syncthreads();                                // filter rows (and, the first time, the patch) visible
        STAMP(4 + 3 * ph);
        if (ph < 2) load_w(ph + 1);                     // in flight during this phase's MFMAs
        if (ph == 2) load_ref(0, NPR / 2);
        bf16x8 afb[2][MP], wfb[2][WGN];
        auto read_group = [&](const int t) {
            const int s = t / NG, gi = t - s * NG;
            const int k0 = 32 * s + g8;                 // this lane's position along the phase's K = (tap, channel)
            const int k0c = k0 < G::KPH ? k0 : 0;       // past the phase's K: read something, the pixel fragment is zeroed
            if (gi == 0) {
                const int tx = k0c >= 2 * CIN ? 2 : (k0c >= CIN ? 1 : 0);
                const int ch = k0c - tx * CIN;
                const int shift = ((ph - 1) * DIL * G::HW2 + (tx - 1) * DIL) * G::PROWB + ch * 2;
#pragma unroll
                for (int j = 0; j < MP; ++j) afb[s & 1][j] = *reinterpret_cast<const bf16x8*>(smem + pbase + j * 16 * G::PROWB + shift);
            }
#pragma unroll
            for (int i = 0; i < WGN; ++i)
                wfb[t & 1][i] = *reinterpret_cast<const bf16x8*>(wlds + ((gi * WGN + i) * 16 + fr) * G::WROWB + k0c * 2);
        };
        read_group(0);
#pragma unroll
        for (int t = 0; t < NGT; ++t) {
            if (t + 1 < NGT) read_group(t + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int s = t / NG, gi = t - s * NG;
            if (gi == 0 && G::KPH % 32 != 0 && s == G::NS - 1) {
                const bool valid = 32 * s + g8 < G::KPH;
#pragma unroll
                for (int j = 0; j < MP; ++j) afb[s & 1][j] = valid ? afb[s & 1][j] : bf16x8{};
            }
#pragma unroll
            for (int i = 0; i < WGN; ++i)
#pragma unroll
                for (int j = 0; j < MP; ++j) {
                    if (D3F_ABL == 2) continue;
                    acc[gi * WGN + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfb[t & 1][i], afb[s & 1][j], acc[gi * WGN + i][j], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(5 + 3 * ph);
    }

    // ---- epilogue: lane holds output channels q = i*16 + g*4 .. +3 of pixels (ty0 + wave, tx0 + j*16 + fr)
    u16* yg = reinterpret_cast<u16*>(d.y);
    load_ref(NPR / 2, NPR);
    // patch and filter rows are dead: the patch becomes the table of sums.  The barrier waits for this wave's LDS
    // operations only; __syncthreads() would also wait for the reference loads just issued
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    STAMP(12);
    float* sred = reinterpret_cast<float*>(smem) + wave * (2 * COUT);
    // The second sum is taken against the raw reference value, sum(v * ref), and corrected per channel when the block's
    // totals are formed: sum(v * xhat) = rstd * (sum(v * ref) - mean * sum(v)).
    const bool full = ty0 + TH <= H && tx0 + TW <= W;      // every pixel of the patch is inside the image (block-uniform)
#pragma unroll
    for (int pr = 0; pr < NPR; ++pr) {
        const int q = pr * 32 + g * 8;                  // this lane's eight channels: tile 2 pr holds the first four, 2 pr + 1 the rest
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
        // ReLU mask as a number: clamp(2^64 * (ref * scale + shift)) is 1 for every positive pre-activation of at least
        // 2^-64, 0 for zero, negatives and NaN: one packed fma with the clamp modifier per two elements
        constexpr float BIG = 18446744073709551616.f;
        f32x2_t scb[4], shb[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 es = *reinterpret_cast<const f32x4*>(tab + q + 4 * h), eh = *reinterpret_cast<const f32x4*>(tab + COUT + q + 4 * h);
            scb[2 * h] = f32x2_t{es[0] * BIG, es[1] * BIG}; scb[2 * h + 1] = f32x2_t{es[2] * BIG, es[3] * BIG};
            shb[2 * h] = f32x2_t{eh[0] * BIG, eh[1] * BIG}; shb[2 * h + 1] = f32x2_t{eh[2] * BIG, eh[3] * BIG};
        }
#pragma unroll
        for (int j = 0; j < MP; ++j) {
            const uint32_t rw[4] = {rr[pr][j].x, rr[pr][j].y, rr[pr][j].z, rr[pr][j].w};
            uint32_t ow[4];
#pragma unroll
            for (int h2 = 0; h2 < 4; ++h2) {            // channel pair 2 h2, 2 h2 + 1 of the eight: tile 2 pr + h2 / 2, elements 2 (h2 & 1), + 1
                const f32x2_t rf = {bf_lo(rw[h2]), bf_hi(rw[h2])};
                f32x2_t m;
                asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m) : "v"(rf), "v"(scb[h2]), "v"(shb[h2]));
                float v[2];
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    v[e2] = acc[2 * pr + (h2 >> 1)][j][2 * (h2 & 1) + e2] * m[e2];
                    if (!full) v[e2] = pok[j] ? v[e2] : 0.f;      // rows past the image edge were computed from padding
                    s1[2 * h2 + e2] += v[e2];
                    s2[2 * h2 + e2] = fmaf(v[e2], rf[e2], s2[2 * h2 + e2]);
                }
                ow[h2] = pack_bf2(v[0], v[1]);
            }
            if (pok[j] && (D3F_ABL != 3 || ow[0] == 0x12345678u)) *reinterpret_cast<uint4*>(yg + poff[j] * d.ldy + q) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[e] = row16_sum(s1[e]);
            s2[e] = row16_sum(s2[e]);
        }
        if (fr == 0) {                                  // one plain store per total into the wave's own row of the table
            *reinterpret_cast<f32x4*>(sred + q) = f32x4{s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<f32x4*>(sred + q + 4) = f32x4{s1[4], s1[5], s1[6], s1[7]};
            *reinterpret_cast<f32x4*>(sred + COUT + q) = f32x4{s2[0], s2[1], s2[2], s2[3]};
            *reinterpret_cast<f32x4*>(sred + COUT + q + 4) = f32x4{s2[4], s2[5], s2[6], s2[7]};
        }
    }
    STAMP(13);
    __syncthreads();
    STAMP(14);
    if (tid < 2 * COUT) {
        const float* all = reinterpret_cast<const float*>(smem);
        const int half = tid / COUT, q = tid - half * COUT;
        float t = 0.f, t1 = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < NT / 64; ++w8) {
            t += all[w8 * 2 * COUT + tid];
            t1 += all[w8 * 2 * COUT + q];              // sum(v) of the same channel
        }
        if (half) t = tab[3 * COUT + q] * fmaf(-tab[2 * COUT + q], t1, t);
        const int sld = d.s_ld > 0 ? d.s_ld : COUT;
        atomicAdd(d.s_acc + (size_t)(blockIdx.x % DS_R) * 2 * sld + half * sld + q, t);
    }
}

template <int CIN, int COUT, int DIL>
int launch_fused(const dsaspp_conv_desc& d, hipStream_t s) {
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    const int64_t nb = (int64_t)d.n_img * tiles_x * tiles_y;
    DS_REQUIRE(nb < (1ll << 31), "conv_fwd(3x3 data gradient with read-out): grid too large");
    auto kern = dgrad3x3_fused_kernel<CIN, COUT, DIL>;
    dsaspp_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024);
    constexpr size_t lds = Geo<CIN, COUT, DIL>::LDS;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(NT), lds, s, d, tiles_x, tiles_y, (int)nb);
    DS_LAUNCH_CHECK("conv_fwd(3x3 data gradient with read-out)");
    return 0;
}

template <int CIN, int COUT>
int launch_dil(const dsaspp_conv_desc& d, hipStream_t s) {
    switch (d.dil) {
        case 1: return launch_fused<CIN, COUT, 1>(d, s);
        case 2: return launch_fused<CIN, COUT, 2>(d, s);
        case 4: return launch_fused<CIN, COUT, 4>(d, s);
        default: return dsaspp_set_error("conv_fwd(3x3 data gradient with read-out): dilation %d not built (1, 2, 4)", d.dil);
    }
}

}  // namespace

// The gradient read-out form of dsaspp_conv_fwd (gx_corr1 != NULL).  0 = launched, else an error code: there is no
// other kernel for this form, the caller records the read-out and the convolution separately for other shapes.
int dsaspp_dgrad3x3_fused_launch(const dsaspp_conv_desc& d, hipStream_t s) {
    DS_REQUIRE(d.taps == 9 && d.dtype == 0 && d.n_src <= 1 && !d.g && d.y && !d.y_f32 && !d.in_scale && !d.in_shift && !d.in_relu && !d.in_mask &&
                   !d.bias && !d.out_mask && !d.stat_acc,
               "conv_fwd: the gradient read-out prologue goes with the 3x3 data-gradient form only (bf16 output, no other prologue)");
    DS_REQUIRE(d.epi_relu == 1 && d.ref && d.e_scale && d.e_shift && d.e_mean && d.e_rstd && d.s_acc,
               "conv_fwd: the gradient read-out form needs the ReLU-mask epilogue with its BatchNorm-backward sums");
    DS_REQUIRE(d.gx_corr1 && d.gx_corr2 && d.gx_xs && d.gx_mean && d.gx_rstd, "conv_fwd: gradient read-out prologue incomplete");
    DS_REQUIRE(d.gx_ldxs % 8 == 0 && d.gx_ldxs >= d.cin && ((uintptr_t)d.gx_xs & 15) == 0, "conv_fwd: gx_xs stride/alignment");
    DS_REQUIRE(!d.gx_dz || (d.gx_lddz % 8 == 0 && d.gx_lddz >= d.cin && ((uintptr_t)d.gx_dz & 15) == 0), "conv_fwd: gx_dz stride/alignment");
    if (d.cin == 48 && d.cout == 192) return launch_dil<48, 192>(d, s);
    if (d.cin == 32 && d.cout == 128) return launch_dil<32, 128>(d, s);
    return dsaspp_set_error("conv_fwd: the gradient read-out form is built for (cin, cout) = (48, 192) and (32, 128), got (%d, %d)", d.cin, d.cout);
}
