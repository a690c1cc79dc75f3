// Grouped 1x1 data gradient into the gradient slab, gfx950.
//
//   g[p][c] (+)= sum_s  e_scale_s[c] * [ref[p][c] * e_scale_s[c] + e_shift_s[c] > 0] * (x_s[p][:] . w_s[c][:])
//
// for up to four consumer layers s of the same slab prefix, with each layer's BatchNorm-backward
// sums as a by-product.  K (the bottleneck width) is short, so a launch is a long stream of short
// GEMMs; the kernel is built around keeping that stream's operand traffic in flight:
//   * persistent blocks, one per CU, walk (pixel tile, channel tile) pairs; tiles that run at
//     the same time on one XCD share their pixel tile (and its L2 lines);
//   * operands go global -> LDS by DMA (global_load_lds, 16 B per lane) into a ring of 4 stages of
//     one 64-wide K-step each; the XOR chunk swizzle lives in the per-lane source address, the LDS
//     image is lane-linear.  Three stages are in flight across the per-step barrier and across
//     sources, tiles and the slab read-modify-write (counted vmcnt, raw s_barrier);
//   * each source's product is masked / summed / scaled in accumulator layout into a running total;
//     the slab tile is read and written once per tile, through a wave-private LDS patch so that the
//     accesses are full 128-byte lines.
#include "common.h"

namespace {

constexpr int BP = 128, BQ = 128, BK = 64;
constexpr int NW = 8, NT = 512, WP = 2, WQ = 4;
constexpr int TP = BP / WP, TQ = BQ / WQ, MP = TP / 16, MQ = TQ / 16;   // 64 x 32 wave tile
constexpr int STAGE = (BP + BQ) * BK * 2;                                // 32 KiB
constexpr int NS = 4, DEPTH = 3;                                         // ring stages, stages in flight
constexpr int ROWF = TQ + 4;
constexpr int PATCH_BYTES = NW * 16 * ROWF * 4;
constexpr int PAR_ROWS = 10;                                             // es[4], eh[4], mean, rstd
constexpr int PAR_BYTES = PAR_ROWS * BQ * 4;
constexpr int SRED_BYTES = 4 * 2 * BQ * 4;
constexpr int LDS_BYTES = NS * STAGE + PATCH_BYTES + PAR_BYTES + SRED_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

struct Src {
    const u16* x;
    const u16* w;
    const float* es;
    const float* eh;
    float* sa;
    int sl;
};

__device__ __forceinline__ Src get_src(const dsaspp_conv_desc& d, int s) {
    Src r;
    if (s == 0) { r.x = (const u16*)d.x; r.w = (const u16*)d.w; r.es = d.e_scale; r.eh = d.e_shift; r.sa = d.s_acc; r.sl = d.s_ld; }
    else if (s == 1) { r.x = (const u16*)d.x1; r.w = (const u16*)d.w1; r.es = d.es1; r.eh = d.eh1; r.sa = d.sa1; r.sl = d.sl1; }
    else if (s == 2) { r.x = (const u16*)d.x2; r.w = (const u16*)d.w2; r.es = d.es2; r.eh = d.eh2; r.sa = d.sa2; r.sl = d.sl2; }
    else { r.x = (const u16*)d.x3; r.w = (const u16*)d.w3; r.es = d.es3; r.eh = d.eh3; r.sa = d.sa3; r.sl = d.sl3; }
    if (r.sl <= 0) r.sl = d.cout;
    return r;
}

__device__ __forceinline__ void wait_vm(int younger_ops) {
    // wait until at most `younger_ops` vector-memory operations are outstanding (rounded down: safe)
    // (LDS operations of this wave are retired too, so that what it wrote is visible after the barrier)
    if (younger_ops >= 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    else if (younger_ops >= 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    else if (younger_ops >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if (younger_ops >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (younger_ops >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (younger_ops >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <bool ACCG, bool GBF16>
__global__ __launch_bounds__(NT, 2) void dgrad_group_kernel(const dsaspp_conv_desc d, const int M, const int nP, const int nQ) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* patch = reinterpret_cast<float*>(smem + NS * STAGE);
    float* par = reinterpret_cast<float*>(smem + NS * STAGE + PATCH_BYTES);
    float* sred = reinterpret_cast<float*>(smem + NS * STAGE + PATCH_BYTES + PAR_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = d.cin, cout = d.cout, ldx = d.ldx;
    const int ksteps = cin / BK;
    const int nsrc = d.n_src > 1 ? d.n_src : 1;
    const int nstep = nsrc * ksteps;                 // K-steps per tile

    // ---- this block's tile sequence: XCD x = blockIdx % 8 owns the pixel tiles pt = x (mod 8)
    const int xcd = blockIdx.x & 7, jblk = blockIdx.x >> 3, g8 = gridDim.x >> 3;
    const int nPx = xcd < nP ? (nP - xcd + 7) >> 3 : 0;
    const int Lx = nPx * nQ;
    const int ntile = jblk < Lx ? (Lx - jblk + g8 - 1) / g8 : 0;
    if (ntile == 0) return;
    const int total_steps = ntile * nstep;
    auto tile_coords = [&](int k, int& p_base, int& q_base) {
        const int r = k * g8 + jblk;
        const int ptl = r / nQ;
        p_base = (ptl * 8 + xcd) * BP;
        q_base = (r - ptl * nQ) * BQ;
    };

    // ---- DMA issue cursor
    const int drow = lane >> 3, dchunk = lane & 7;
    int c_k = 0, c_s = 0, c_kk = 0, c_g = 0;
    int a_off[2], w_off[2];
    auto cursor_tile = [&]() {
        int pb, qb;
        tile_coords(c_k, pb, qb);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = (i * 8 + wave) * 8 + drow;
            const int lc = dchunk ^ swz8(row);
            int p = pb + row;
            p = p < M ? p : M - 1;
            int q = qb + row;
            q = q < cout ? q : cout - 1;
            a_off[i] = p * ldx + lc * 8;
            w_off[i] = q * cin + lc * 8;
        }
    };
    auto issue = [&]() {
        if (c_k < ntile) {
            const Src s = get_src(d, c_s);
            unsigned char* slot = ring + (c_g % NS) * STAGE;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(s.x + a_off[i] + c_kk * BK), (lds_ptr_t)(slot + (i * 8 + wave) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(s.w + w_off[i] + c_kk * BK), (lds_ptr_t)(slot + BP * 128 + (i * 8 + wave) * 1024), 16, 0, 0);
            ++c_g;
            if (++c_kk == ksteps) {
                c_kk = 0;
                if (++c_s == nsrc) {
                    c_s = 0;
                    if (++c_k < ntile) cursor_tile();
                }
            }
        }
    };

    auto load_par = [&](int q_base) {
        for (int c = tid; c < PAR_ROWS * BQ; c += NT) {
            const int row = c / BQ, cc = c - row * BQ;
            int q = q_base + cc;
            q = q < cout ? q : cout - 1;
            float v;
            if (row < 8) {
                const int s = row & 3;
                if (s < nsrc) {
                    const Src sr = get_src(d, s);
                    v = row < 4 ? sr.es[q] : sr.eh[q];
                } else {
                    v = 0.f;
                }
            } else {
                v = row == 8 ? d.e_mean[q] : d.e_rstd[q];
            }
            par[c] = v;
        }
    };

    // ---- fragment coordinates
    const int fr = lane & 15, fc = lane >> 4;
    const int wp = wave % WP, wq = wave / WP;
    const int frag_off0 = fr * 128 + ((fc ^ swz8(fr)) << 4);
    const int frag_off1 = fr * 128 + (((4 + fc) ^ swz8(fr)) << 4);
    const int rep = blockIdx.x % DS_R;
    constexpr bool acc_g = ACCG;
    float* stage = patch + wave * (16 * ROWF);
    const int er = lane >> 3, ec = lane & 7;

    for (int c = tid; c < 4 * 2 * BQ; c += NT) sred[c] = 0.f;
    {
        int pb, qb;
        tile_coords(0, pb, qb);
        load_par(qb);
    }
    cursor_tile();
    for (int i = 0; i < DEPTH; ++i) issue();

    int g = 0;
    for (int k = 0; k < ntile; ++k) {
        int p_base, q_base;
        tile_coords(k, p_base, q_base);
        // reference activations of this tile in accumulator layout
        uint2 rfm[MQ][MP];
        bool pok[MP], qok[MQ];
#pragma unroll
        for (int j = 0; j < MP; ++j) pok[j] = p_base + wp * TP + j * 16 + fr < M;
#pragma unroll
        for (int i = 0; i < MQ; ++i) qok[i] = q_base + wq * TQ + i * 16 + fc * 4 < cout;
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) {
                const int q = q_base + wq * TQ + i * 16 + fc * 4;
                const int64_t p = p_base + wp * TP + j * 16 + fr;
                const bool ok = pok[j] && qok[i];
                rfm[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(d.ref) + (ok ? p * d.ldref + q : 0));
            }
        // old slab values of this tile, in the row layout of the final read-modify-write
        constexpr int RPP = 8, NRP = 2;          // 8 lanes cover 32 channels of a pixel; 8 pixel rows per pass
        const int q = q_base + wq * TQ + ec * 4;
        float4 gold[MP][NRP];
        bool okk[MP][NRP];
#pragma unroll
        for (int j = 0; j < MP; ++j)
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                const int64_t p = p_base + wp * TP + j * 16 + rp * RPP + er;
                const bool ok = p < M && q < cout;
                okk[j][rp] = ok;
                if (acc_g) {
                    const int64_t off = ok ? p * d.ldg + q : 0;
                    if (GBF16) {
                        const uint2 gb = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(d.g) + off);
                        gold[j][rp] = make_float4(bf_lo(gb.x), bf_hi(gb.x), bf_lo(gb.y), bf_hi(gb.y));
                    } else {
                        gold[j][rp] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(d.g) + off);
                    }
                } else {
                    gold[j][rp] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        // These loads (MQ*MP reference loads, MP*NRP slab loads) sit, in issue order, after the DMAs of the
        // tile's first DEPTH steps: while one of those steps is awaited they may stay outstanding too.
        const int n_ord = MQ * MP + (acc_g ? MP * NRP : 0);
        const int g0 = g;
        f32x4 tot[MQ][MP], xh[MQ][MP];
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        for (int s = 0; s < nsrc; ++s) {
            f32x4 acc[MQ][MP];
#pragma unroll
            for (int i = 0; i < MQ; ++i)
#pragma unroll
                for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int kk = 0; kk < ksteps; ++kk, ++g) {
                // stage g landed (this wave's DMAs), then everybody's; the barrier also says every wave
                // is done reading stage g-1, whose ring slot the next DMA overwrites
                const int younger = total_steps - 1 - g;
                wait_vm(4 * (younger < DEPTH - 1 ? younger : DEPTH - 1) + (g - g0 < DEPTH ? n_ord : 0));
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue();
                const unsigned char* As = ring + (g % NS) * STAGE;
                const unsigned char* Ws = As + BP * 128;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int fo = ks ? frag_off1 : frag_off0;
                    bf16x8 af[MP], wf[MQ];
#pragma unroll
                    for (int j = 0; j < MP; ++j)
                        af[j] = *reinterpret_cast<const bf16x8*>(As + (wp * TP + j * 16) * 128 + fo);
#pragma unroll
                    for (int i = 0; i < MQ; ++i)
                        wf[i] = *reinterpret_cast<const bf16x8*>(Ws + (wq * TQ + i * 16) * 128 + fo);
#pragma unroll
                    for (int i = 0; i < MQ; ++i)
#pragma unroll
                        for (int j = 0; j < MP; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
                }
            }
            if (s == 0) {
                // normalised reference activations, shared by every source's sums
#pragma unroll
                for (int i = 0; i < MQ; ++i) {
                    const int cl = wq * TQ + i * 16 + fc * 4;
                    const f32x4 mu4 = *reinterpret_cast<const f32x4*>(par + 8 * BQ + cl);
                    const f32x4 rs4 = *reinterpret_cast<const f32x4*>(par + 9 * BQ + cl);
#pragma unroll
                    for (int j = 0; j < MP; ++j) {
                        const uint2 rr = rfm[i][j];
                        xh[i][j] = f32x4{(bf_lo(rr.x) - mu4[0]) * rs4[0], (bf_hi(rr.x) - mu4[1]) * rs4[1],
                                         (bf_lo(rr.y) - mu4[2]) * rs4[2], (bf_hi(rr.y) - mu4[3]) * rs4[3]};
                    }
                }
            }
            // ---- this source's product: ReLU mask, BatchNorm-backward sums, scaled add into the total
            const bool want_s = get_src(d, s).sa != nullptr;
#pragma unroll
            for (int i = 0; i < MQ; ++i) {
                const int cl = wq * TQ + i * 16 + fc * 4;
                const f32x4 sc4 = *reinterpret_cast<const f32x4*>(par + s * BQ + cl);
                const f32x4 sh4 = *reinterpret_cast<const f32x4*>(par + (4 + s) * BQ + cl);
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < MP; ++j) {
                    const uint2 rr = rfm[i][j];
                    const float rf[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                    const bool ok = pok[j] && qok[i];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pre = fmaf(rf[e], sc4[e], sh4[e]);
                        const float v = (ok && pre > 0.f) ? acc[i][j][e] : 0.f;
                        s1[e] += v;
                        s2[e] = fmaf(v, xh[i][j][e], s2[e]);
                        tot[i][j][e] = fmaf(v, sc4[e], tot[i][j][e]);
                    }
                }
                if (want_s) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1[e] = row16_sum(s1[e]);
                        s2[e] = row16_sum(s2[e]);
                    }
                    if (fr == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            atomicAdd(sred + (s * 2) * BQ + cl + e, s1[e]);
                            atomicAdd(sred + (s * 2 + 1) * BQ + cl + e, s2[e]);
                        }
                    }
                }
            }
        }

        // ---- tile end: all waves are done with the parameter rows and the partial sums are complete
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        for (int c = tid; c < nsrc * 2 * BQ; c += NT) {
            const int s = c / (2 * BQ), rest = c - s * 2 * BQ;
            const int half = rest / BQ, cc = rest - half * BQ;
            const Src sr = get_src(d, s);
            if (sr.sa && q_base + cc < cout) atomicAdd(sr.sa + (size_t)rep * 2 * sr.sl + half * sr.sl + q_base + cc, sred[c]);
            sred[c] = 0.f;
        }
        if (k + 1 < ntile) {
            int pb, qb;
            tile_coords(k + 1, pb, qb);
            load_par(qb);
        }

        // ---- slab read-modify-write through the wave-private patch, 16 pixels at a time
#pragma unroll
        for (int j = 0; j < MP; ++j) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MQ; ++i) *reinterpret_cast<f32x4*>(stage + fr * ROWF + i * 16 + fc * 4) = tot[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(stage + (rp * RPP + er) * ROWF + ec * 4);
                if (!okk[j][rp]) continue;
                const int64_t p = p_base + wp * TP + j * 16 + rp * RPP + er;
                const float4 t = make_float4(a4[0] + gold[j][rp].x, a4[1] + gold[j][rp].y, a4[2] + gold[j][rp].z, a4[3] + gold[j][rp].w);
                if (GBF16) {
                    uint2 o;
                    o.x = pack_bf2(t.x, t.y);
                    o.y = pack_bf2(t.z, t.w);
                    *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(d.g) + p * d.ldg + q) = o;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.g) + p * d.ldg + q) = t;
                }
            }
        }
    }
}

}  // namespace

// Form check + launch.  Returns -1 when the descriptor is not of the form this kernel handles (the caller
// then uses the general convolution kernel), 0 on success, a positive error code otherwise.
int dsaspp_dgrad_group_launch(const dsaspp_conv_desc& d, hipStream_t s) {
    const bool form = d.g && d.taps == 1 && d.epi_relu && d.ref && d.e_scale && d.e_shift && d.e_mean && d.e_rstd && !d.in_scale &&
                      !d.in_mask && !d.in_relu && !d.bias && !d.out_mask && d.cin % BK == 0 && d.cin <= 1024;
    if (!form) return -1;
    const int64_t M64 = (int64_t)d.n_img * d.H * d.W;
    DS_REQUIRE(M64 * d.ldx < (1ll << 31) && (int64_t)d.cout * d.cin < (1ll << 31), "conv_fwd: operand too large for 32-bit offsets");
    const int M = (int)M64;
    const int nP = (M + BP - 1) / BP, nQ = (d.cout + BQ - 1) / BQ;
    static thread_local int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0)
            n_cu = 256;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_group_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_group_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_group_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dgrad_group_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    }
    int64_t tiles = (int64_t)nP * nQ;
    int grid = (int)((tiles + 7) / 8 * 8);
    const int cap = n_cu / 8 * 8;
    if (grid > cap) grid = cap;
    if (grid < 8) grid = 8;
    auto kern = d.g_accumulate ? (d.g_bf16 ? dgrad_group_kernel<true, true> : dgrad_group_kernel<true, false>)
                               : (d.g_bf16 ? dgrad_group_kernel<false, true> : dgrad_group_kernel<false, false>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS_BYTES, s, d, M, nP, nQ);
    DS_LAUNCH_CHECK("conv_fwd(grouped data gradient)");
    return 0;
}
