// Fused BN-ReLU -> convolution (1x1 / 3x3 dilated) implicit GEMM on MFMA, gfx950.
//
// GEMM view: P = pixels (N*H*W), Q = output channels, K = taps*cin (tap-major, channel-minor).
// A block computes a BP x BQ tile with 4 waves; operands are staged through LDS in K-steps of
// 64 channels as [row][64] bf16 rows (128 B, one full cache line per pixel) with an XOR chunk
// swizzle that makes the ds_read_b128 fragment reads conflict-free.  The K loop keeps the
// instruction count per MFMA low: row pointers are rebuilt only when the filter tap changes,
// loads are branch-free (clamped address + select), and the prologue variant is a template
// parameter.  The activation tile passes through registers on its way to
// LDS, where the consumer's BatchNorm scale/shift, the ReLU and the zero padding of the
// dilated halo are applied, so neither the normalised tensor nor the concatenated input ever
// exists in memory.  Weights are the MFMA A operand (rows = channels) and pixels the B operand,
// so every lane ends up holding 4 consecutive channels of one pixel: 8-byte NHWC stores into
// the channel slice of the feature slab replace torch.cat.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int BK = 64;            // K-step: 64 channels = one full 128-byte line per pixel row

// 128-byte LDS rows hold 8 chunks of 16 B; XOR the chunk index with (row/2)%8 so that the
// ds_read_b128 fragment reads (16 rows x one chunk column per 16-lane group) are conflict-free.
__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

// PRO selects the prologue compiled into the activation staging path:
//   0 = none (raw copy), 1 = per-channel scale/shift (+ReLU when in_relu), 2 = general
//   (optional scale/shift, optional ReLU, per-image channel mask)
// EPI selects the epilogue: 0 = plain store (+bias, +dropout mask, +statistics), 1 = store masked by a
// reference pre-activation with BatchNorm-backward sums (3x3 data gradient), 2 = gradient-slab store/accumulate.
template <int BP, int BQ, int WP, int WQ, int PRO, int EPI>
__global__ __launch_bounds__(64 * WP * WQ, (WP * WQ == 8 && BP < 256) ? 4 : 2) void conv_kernel(const dsaspp_conv_desc d, const int M,
                                                                                  const int nQ, const int nblocks) {
    constexpr int NW = WP * WQ, NT = 64 * NW;     // waves / threads per block
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    constexpr int A_BYTES = BP * BK * 2, W_BYTES = BQ * BK * 2;
    constexpr int RSTEP = NT / 8;                 // rows covered by one pass of the block's threads
    static_assert(BP % RSTEP == 0, "activation tile rows");
    constexpr int AR = BP / RSTEP;                // activation chunks per thread per K-step
    constexpr int WCH = BQ * 8;                   // weight chunks per tile
    constexpr int WR = (WCH + NT - 1) / NT;
    constexpr int TP = BP / WP, TQ = BQ / WQ, MP = TP / 16, MQ = TQ / 16;
    static_assert(TP % 16 == 0 && TQ % 16 == 0, "wave tile");

    // 8-wave tiles: the weight tile needs no prologue, so it goes global -> LDS by DMA (global_load_lds) into a
    // ring of two buffers, one K-step ahead; that takes three of the five 16-byte register->LDS stores per lane
    // and K-step (the slowest LDS operation there is) and their staging registers out of the loop.
#ifndef DSASPP_DMAW
#define DSASPP_DMAW 1
#endif
    // BIG: 256-pixel tile, one block per CU with 256 registers per lane.  Its operands all travel through
    // registers (no DMA) so that the compiler's vmcnt bookkeeping stays exact -- an outstanding LDS-DMA makes it
    // wait for vmcnt(0) at the next use of any loaded register -- and the activation loads run two K-steps ahead.
    constexpr bool BIG = (BP >= 256) && (NW == 8);
    constexpr bool DMAW = (NW == 8) && !BIG && (DSASPP_DMAW != 0);
    constexpr int NWB = DMAW ? 2 : 1;             // weight buffers
    static_assert(!DMAW || BQ % 64 == 0, "DMA weight tile: 8 waves x 8 rows per pass");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Ws = smem + A_BYTES;
    float* tab = reinterpret_cast<float*>(smem + A_BYTES + NWB * W_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int pt = bid / nQ, qt = bid - pt * nQ;
    const int p_base = pt * BP, q_base = qt * BQ;
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout, taps = d.taps, dil = d.dil;
    const int hw = H * W;
    const int Ktot = taps * cin;
    // Merged-K form of a 3x3 convolution with fewer input channels than one K-step (the data gradient of a dense
    // layer's 3x3: 48 or 32 channels): stepping tap by tap would leave a quarter or half of every 64-wide K-step
    // empty, so the nine taps are laid end to end (K = 9*cin, exactly how the weights are stored) and every 16-byte
    // chunk of a K-step finds its own tap: 7 instead of 9 steps for 48 channels, 5 instead of 9 for 32.
    const bool merged = (PRO == 0) && taps == 9 && cin < BK;
    const int cpt = cin >> 3;                          // chunks per tap
    const int csteps = (cin + BK - 1) / BK;
    const int ksteps = merged ? (Ktot + BK - 1) / BK : taps * csteps;
    const int cinW = merged ? Ktot : cin;              // the weight cursor sees a 1x1 convolution over K = 9*cin
    const bool has_scale = d.in_scale != nullptr;

    if (PRO != 0 && has_scale) {
        for (int c = tid; c < cin; c += NT) {
            tab[c] = d.in_scale[c];
            tab[cin + c] = d.in_shift[c];
        }
    }
    const float relu_floor = d.in_relu ? 0.f : -3.0e38f;
    const float relu_ceil = (d.in_relu == 2) ? 6.f : 3.0e38f;     // ReLU6: general prologue (PRO == 2) only
    const uint32_t relu_bits = d.in_relu ? 0u : 0x80008000u;      // see relu_bf2

    // ---- per-thread staging coordinates: chunk column ck of rows (tid>>3) + RSTEP*i
    const int ck = tid & 7;
    const int arow0 = tid >> 3;
    int a_n[AR], a_h[AR], a_w[AR];
    bool a_ok[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        const int m = p_base + arow0 + RSTEP * i;
        a_ok[i] = m < M;
        const int mm = a_ok[i] ? m : 0;
        a_n[i] = mm / hw;
        const int rem = mm - a_n[i] * hw;
        a_h[i] = rem / W;
        a_w[i] = rem - a_h[i] * W;
    }
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    const u16* wg = reinterpret_cast<const u16*>(d.w);
    const u16* w_ptr[WR];
    bool w_ok[WR];
    int wd_chunk[WR];                   // DMA form: source chunk of this lane (the XOR swizzle lives in the source address)
    auto w_setup = [&]() {
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            if (DMAW) {
                // wave-instruction i of this wave fills LDS rows (i*8 + wave)*8 .. +7, lane-linear: lane -> (row, slot)
                const int row = (i * 8 + wave) * 8 + (lane >> 3);
                int q = q_base + row;
                q = q < cout ? q : cout - 1;          // rows past cout: any finite data, their results are never stored
                w_ptr[i] = wg + (int64_t)q * Ktot;
                wd_chunk[i] = ((lane & 7) ^ swz8(row)) * 8;
                w_ok[i] = true;
            } else {
                const int idx = tid + NT * i;
                const int row = idx >> 3;
                w_ok[i] = (idx < WCH) && (q_base + row < cout);
                w_ptr[i] = wg + (w_ok[i] ? (int64_t)(q_base + row) * Ktot : 0);
            }
        }
    };
    w_setup();
    int wbuf = 0;                       // ring slot the next weight DMA fills

    // per-tap row pointers (recomputed only when the tap changes)
    const u16* a_ptr[AR];
    bool a_val[AR];
    auto tap_setup = [&](const int tap) {
        int dy = 0, dx = 0;
        if (taps == 9) {
            const int ty = tap / 3;
            dy = (ty - 1) * dil;
            dx = (tap - ty * 3 - 1) * dil;
        }
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int hh = a_h[i] + dy, ww = a_w[i] + dx;
            const bool ok = a_ok[i] && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W;
            a_val[i] = ok;
            const int64_t pix = ((int64_t)a_n[i] * H + hh) * W + ww;
            a_ptr[i] = xg + (ok ? pix * d.ldx : 0);
        }
    };

    // Activation chunks travel through registers (the prologue is applied on their way to LDS), PD K-steps ahead.
    constexpr int PD = BIG ? 2 : 1;
    const u16* a_ctr[AR];               // merged-K form: the row's own pixel, and which of its nine taps are inside the image
    unsigned a_tapok[AR];
    if (merged) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            a_ctr[i] = xg + (((int64_t)a_n[i] * H + a_h[i]) * W + a_w[i]) * d.ldx;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int hh = a_h[i] + (t / 3 - 1) * dil, ww = a_w[i] + (t % 3 - 1) * dil;
                if (a_ok[i] && (unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) mk |= 1u << t;
            }
            a_tapok[i] = mk;
        }
    }
    int kstep_a = 0;                    // merged-K form: index of the next K-step the activation cursor loads
    uint4 areg[PD][AR], wreg[WR];
    bool lval[PD][AR], wval[WR];        // validity of the chunks held in areg / wreg
    int lc[PD];                         // channel of the chunks held in areg (for the prologue tables)
    int tap = 0, c0 = 0;                // activation cursor: position of the next K-step to load
    int tapw = 0, c0w = 0;              // weight cursor

    auto loadA = [&](const int st) {
        if (merged) {
            const int kc = kstep_a * 8 + ck;               // chunk index along K = (tap, 8-channel group)
            const int tp = kc / cpt, c8 = kc - tp * cpt;
            const bool kvalid = tp < 9;
            const int ty = tp / 3, tx = tp - ty * 3;
            const int64_t toff = (int64_t)((ty - 1) * W + (tx - 1)) * dil * d.ldx + c8 * 8;
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const bool ok = kvalid && ((a_tapok[i] >> tp) & 1u);
                areg[st][i] = *reinterpret_cast<const uint4*>(a_ctr[i] + (ok ? toff : 0));   // always a mapped address
                lval[st][i] = ok;
            }
            lc[st] = 0;
            ++kstep_a;
            return;
        }
        const int c = c0 + ck * 8;
        const bool cvalid = c < cin;
        const int co = cvalid ? c : 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            areg[st][i] = *reinterpret_cast<const uint4*>(a_ptr[i] + co);     // always a mapped address
            lval[st][i] = a_val[i] && cvalid;
        }
        lc[st] = co;
        c0 += BK;
        if (c0 >= cin) {
            c0 = 0;
            ++tap;
            if (tap < taps) tap_setup(tap);
        }
    };

    auto loadW = [&]() {
        if (DMAW) {
            typedef __attribute__((address_space(3))) void* lds_p;
            typedef const __attribute__((address_space(1))) void* glb_p;
            unsigned char* slot = Ws + wbuf * W_BYTES;
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                // K tail: the activation chunk is zeroed, so the weight chunk only has to be finite -> channel 0
                const int cc = c0w + wd_chunk[i];
                const int kq = cc < cinW ? tapw * cin + cc : 0;
                __builtin_amdgcn_global_load_lds((glb_p)(w_ptr[i] + kq), (lds_p)(slot + (i * 8 + wave) * 1024), 16, 0, 0);
            }
            wbuf ^= 1;
        } else {
            const int c = c0w + ck * 8;
            const bool cvalid = c < cinW;
            const int ko = cvalid ? tapw * cin + c : 0;
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                // raw load only: touching the value here would make the wave wait for it before the MFMAs of the
                // step in progress; out-of-range rows / the K tail are zeroed when the chunk goes to LDS
                wreg[i] = *reinterpret_cast<const uint4*>(w_ptr[i] + ko);
                wval[i] = w_ok[i] && cvalid;
            }
        }
        c0w += BK;
        if (c0w >= cinW) {
            c0w = 0;
            ++tapw;
        }
    };

    auto sstore = [&](const int st) {
        float sc[8], sh[8];
        if (PRO == 1 || (PRO == 2 && has_scale)) {
            load8f(tab + lc[st], sc);
            load8f(tab + cin + lc[st], sh);
        }
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            uint4 v = areg[st][i];
            if (PRO != 0) {
                float f[8];
                unpack8(v, f);
                if (PRO == 1 || has_scale) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                }
                if (PRO == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fminf(fmaxf(f[e], relu_floor), relu_ceil);
                    if (d.in_mask) {
                        float mk[8];
                        load8f(d.in_mask + (int64_t)a_n[i] * d.in_mask_ld + lc[st], mk);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] *= mk[e];
                    }
                    v = pack8(f);
                } else {
                    v = relu_bf8(pack8(f), relu_bits);     // ReLU after the rounding, four packed 16-bit max
                }
            }
            const bool ok = lval[st][i];  // zero padding (halo, K tail, rows past M) comes AFTER the prologue
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            const int row = arow0 + RSTEP * i;
            *reinterpret_cast<uint4*>(As + row * 128 + ((ck ^ swz8(row)) << 4)) = v;
        }
#pragma unroll
        for (int i = 0; i < (DMAW ? 0 : WR); ++i) {
            const int idx = tid + NT * i;
            if (idx < WCH) {
                const int row = idx >> 3;
                uint4 v = wreg[i];
                const bool ok = wval[i];
                v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
                *reinterpret_cast<uint4*>(Ws + row * 128 + ((ck ^ swz8(row)) << 4)) = v;
            }
        }
    };

    // ---- MFMA fragment coordinates
    const int fr = lane & 15, fc = lane >> 4;
    const int wp = wave % WP, wq = wave / WP;
    const int frag_off0 = fr * 128 + ((fc ^ swz8(fr)) << 4);
    const int frag_off1 = fr * 128 + (((4 + fc) ^ swz8(fr)) << 4);
    f32x4 acc[MQ][MP];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // block barrier that waits for this wave's LDS operations only: __syncthreads() also drains vmcnt, i.e.
    // it would wait for the prefetches that are meant to stay in flight
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    auto run_k = [&]() {
        tap = 0; c0 = 0; tapw = 0; c0w = 0;
        kstep_a = 0;
        wbuf = 0;
        if (!merged) tap_setup(0);
        loadW();                                   // weights of step 0
        loadA(0);
        if (PD > 1 && ksteps > 1) loadA(1);
        __syncthreads();   // scale/shift table visible
        for (int s = 0; s < ksteps; s += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int ss = s + u;
                if (ss < ksteps) {
                    sstore(u);                     // K-step ss: registers -> LDS (prologue applied here)
                    if (DMAW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's weight DMAs of step ss have landed
                    lds_barrier();
                    // weights one step ahead (a DMA fills the slot last read in step ss-1, which every wave left
                    // before the barrier that ended that step), activations PD steps ahead
                    if (ss + 1 < ksteps) loadW();
                    if (ss + PD < ksteps) loadA(u);
                    const unsigned char* Wc = Ws + (DMAW ? (ss & 1) * W_BYTES : 0);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int fo = ks ? frag_off1 : frag_off0;
                        bf16x8 af[MP], wf[MQ];
#pragma unroll
                        for (int j = 0; j < MP; ++j)
                            af[j] = *reinterpret_cast<const bf16x8*>(As + (wp * TP + j * 16) * 128 + fo);
#pragma unroll
                        for (int i = 0; i < MQ; ++i)
                            wf[i] = *reinterpret_cast<const bf16x8*>(Wc + (wq * TQ + i * 16) * 128 + fo);
#pragma unroll
                        for (int i = 0; i < MQ; ++i)
#pragma unroll
                            for (int j = 0; j < MP; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
                    }
                    lds_barrier();                 // every wave is done reading before the next store
                }
            }
        }
    };
    const int rep = blockIdx.x % DS_R;

    if (EPI != 3) {
        run_k();
    } else {
        // Several consumer layers of the same slab prefix in one pass: each source's product is masked
        // by that layer's ReLU, feeds that layer's BatchNorm-backward sums and is added, scaled, to a
        // running total in accumulator layout; the slab is then read and written once for all of them.
        float* sred = tab;                         // [2][BQ] block-level partial sums of one source
        for (int c = tid; c < 2 * BQ; c += NT) sred[c] = 0.f;
        uint2 rfm[MQ][MP];
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) {
                const int q = q_base + wq * TQ + i * 16 + fc * 4;
                const int64_t p = p_base + wp * TP + j * 16 + fr;
                const bool ok = p < M && q < cout;
                rfm[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(d.ref) + (ok ? p * d.ldref + q : 0));
            }
        f32x4 tot[MQ][MP];
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nsrc = d.n_src;
        for (int sidx = 0; sidx < nsrc; ++sidx) {
            const float *es, *eh;
            float* sa;
            int sl;
            if (sidx == 0) { xg = reinterpret_cast<const u16*>(d.x); wg = reinterpret_cast<const u16*>(d.w); es = d.e_scale; eh = d.e_shift; sa = d.s_acc; sl = d.s_ld; }
            else if (sidx == 1) { xg = reinterpret_cast<const u16*>(d.x1); wg = reinterpret_cast<const u16*>(d.w1); es = d.es1; eh = d.eh1; sa = d.sa1; sl = d.sl1; }
            else if (sidx == 2) { xg = reinterpret_cast<const u16*>(d.x2); wg = reinterpret_cast<const u16*>(d.w2); es = d.es2; eh = d.eh2; sa = d.sa2; sl = d.sl2; }
            else { xg = reinterpret_cast<const u16*>(d.x3); wg = reinterpret_cast<const u16*>(d.w3); es = d.es3; eh = d.eh3; sa = d.sa3; sl = d.sl3; }
            if (sl <= 0) sl = cout;
            w_setup();
#pragma unroll
            for (int i = 0; i < MQ; ++i)
#pragma unroll
                for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            run_k();
#pragma unroll
            for (int i = 0; i < MQ; ++i) {
                const int cl = wq * TQ + i * 16 + fc * 4;
                const int q = q_base + cl;
                const bool qok = q < cout;
                const int qc = qok ? q : 0;
                const float4 e_s = *reinterpret_cast<const float4*>(es + qc);
                const float4 e_h = *reinterpret_cast<const float4*>(eh + qc);
                const float4 e_m = *reinterpret_cast<const float4*>(d.e_mean + qc);
                const float4 e_r = *reinterpret_cast<const float4*>(d.e_rstd + qc);
                const float sc4[4] = {e_s.x, e_s.y, e_s.z, e_s.w}, sh4[4] = {e_h.x, e_h.y, e_h.z, e_h.w};
                const float mu4[4] = {e_m.x, e_m.y, e_m.z, e_m.w}, rs4[4] = {e_r.x, e_r.y, e_r.z, e_r.w};
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < MP; ++j) {
                    const uint2 rr = rfm[i][j];
                    const float rf[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pre = fmaf(rf[e], sc4[e], sh4[e]);
                        const float v = (pre > 0.f) ? acc[i][j][e] : 0.f;   // rows past M and channels past cout are exact zeros
                        s1[e] += v;
                        s2[e] += v * ((rf[e] - mu4[e]) * rs4[e]);
                        tot[i][j][e] = fmaf(v, sc4[e], tot[i][j][e]);
                    }
                }
                if (sa) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) {
                            s1[e] += __shfl_xor(s1[e], o, 64);
                            s2[e] += __shfl_xor(s2[e], o, 64);
                        }
                    }
                    if (fr == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            atomicAdd(sred + cl + e, s1[e]);
                            atomicAdd(sred + BQ + cl + e, s2[e]);
                        }
                    }
                }
            }
            if (sa) {
                __syncthreads();
                for (int c = tid; c < 2 * BQ; c += NT) {
                    const int half = c / BQ, cc = c - half * BQ;
                    if (q_base + cc < cout) atomicAdd(sa + (size_t)rep * 2 * sl + half * sl + q_base + cc, sred[c]);
                    sred[c] = 0.f;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) acc[i][j] = tot[i][j];
    }

    // ---- epilogue.  The accumulators hold 4 consecutive channels of one pixel per lane; going
    // straight to memory from there touches 16 different pixel rows per instruction.  Each wave
    // instead passes its tile through a private LDS patch, 16 pixels at a time, and re-reads it so
    // that LPR consecutive lanes cover one contiguous run of CBW channels of a pixel (128 B in
    // fp32): the gradient-slab read-modify-write and the NHWC stores become full-line accesses.
    constexpr int CBW = (TQ % 32 == 0) ? 32 : 16;   // channels per column block
    constexpr int LPR = CBW / 4;                    // lanes per pixel row (one float4 each)
    constexpr int RPP = 64 / LPR;                   // pixel rows per pass
    constexpr int NCB = TQ / CBW;
    constexpr int NRP = 16 / RPP;
    constexpr int ROWF = TQ + 4;                    // padded row stride in floats (bank spread)
    static_assert(NW * 16 * ROWF * 4 + 7 * BQ * 4 <= A_BYTES + NWB * W_BYTES, "epilogue LDS fits in the staging buffers");
    float* stage = reinterpret_cast<float*>(smem) + wave * (16 * ROWF);
    float* epar = reinterpret_cast<float*>(smem) + NW * 16 * ROWF;     // [5][BQ]: bias, scale, shift, mean, rstd
    float* red = epar + 5 * BQ;                                        // [2][BQ]: block-level sums
    const bool want_stats = (EPI < 2) && (d.stat_acc != nullptr);
    const bool want_s = (EPI == 1 || EPI == 2) && d.s_acc != nullptr;
    const bool need_ref = (EPI == 1 || EPI == 2) && (d.epi_relu || want_s);
    const bool acc_g = (EPI >= 2) && d.g_accumulate;
    for (int c = tid; c < BQ; c += NT) {
        const int q = q_base + c;
        const bool ok = q < cout;
        epar[c] = (ok && d.bias) ? d.bias[q] : 0.f;
        epar[BQ + c] = (ok && d.e_scale && EPI != 3) ? d.e_scale[q] : 1.f;
        epar[2 * BQ + c] = (ok && d.e_scale) ? d.e_shift[q] : 0.f;
        epar[3 * BQ + c] = (ok && want_s) ? d.e_mean[q] : 0.f;
        epar[4 * BQ + c] = (ok && want_s) ? d.e_rstd[q] : 1.f;
        red[c] = 0.f;
        red[BQ + c] = 0.f;
    }
    const int er = lane / LPR, ec = lane % LPR;
    float st_a[NCB][4], st_b[NCB][4];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) { st_a[cb][e] = 0.f; st_b[cb][e] = 0.f; }

    __syncthreads();              // K-loop LDS reads finished in every wave; epar visible
    // phase 0: every global load the epilogue needs (reference activations, old gradient-slab values,
    // dropout mask) is issued here for the whole wave tile, so the block pays one memory round trip
    // instead of one per 16-pixel group
    constexpr int NI = NCB * NRP;
    bool ok_[MP][NI];
    float4 gold_[(EPI >= 2) ? MP : 1][NI];
    uint2 rr_[(EPI == 1 || EPI == 2) ? MP : 1][NI];
#pragma unroll
    for (int j = 0; j < MP; ++j) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                const int it = cb * NRP + rp;
                const int q = q_base + wq * TQ + cb * CBW + ec * 4;
                const int p = p_base + wp * TP + j * 16 + rp * RPP + er;
                const bool ok = p < M && q < cout;
                ok_[j][it] = ok;
                const int64_t pc = ok ? p : 0;
                const int qc = ok ? q : 0;
                if ((EPI == 1 || EPI == 2) && need_ref) rr_[j][it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(d.ref) + pc * d.ldref + qc);
                if (EPI >= 2 && acc_g) {
                    if (d.g_bf16) {
                        const uint2 gb = *reinterpret_cast<const uint2*>(reinterpret_cast<const u16*>(d.g) + pc * d.ldg + qc);
                        gold_[j][it] = make_float4(bf_lo(gb.x), bf_hi(gb.x), bf_lo(gb.y), bf_hi(gb.y));
                    } else {
                        gold_[j][it] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(d.g) + pc * d.ldg + qc);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < MP; ++j) {
        // The patch is private to the wave and LDS executes a wave's accesses in order, so no
        // block barrier is needed between the write and the read-back; only the compiler must not
        // reorder them.
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < MQ; ++i)
            *reinterpret_cast<f32x4*>(stage + fr * ROWF + i * 16 + fc * 4) = acc[i][j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 a4_[NI];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp)
                a4_[cb * NRP + rp] = *reinterpret_cast<const f32x4*>(stage + (rp * RPP + er) * ROWF + cb * CBW + ec * 4);
        // phase 2: arithmetic and stores
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                const int it = cb * NRP + rp;
                if (!ok_[j][it]) continue;
                const int cl = wq * TQ + cb * CBW + ec * 4;      // channel within the block tile
                const int q = q_base + cl;
                const int64_t p = p_base + wp * TP + j * 16 + rp * RPP + er;
                float v[4] = {a4_[it][0] + epar[cl], a4_[it][1] + epar[cl + 1], a4_[it][2] + epar[cl + 2], a4_[it][3] + epar[cl + 3]};
                if (d.out_mask) {
                    const float4 mk = *reinterpret_cast<const float4*>(d.out_mask + (p / hw) * d.out_mask_ld + q);
                    v[0] *= mk.x; v[1] *= mk.y; v[2] *= mk.z; v[3] *= mk.w;
                }
                float rf[4] = {0, 0, 0, 0};
                if ((EPI == 1 || EPI == 2) && need_ref) {
                    const uint2 rr = rr_[j][it];
                    rf[0] = bf_lo(rr.x); rf[1] = bf_hi(rr.x); rf[2] = bf_lo(rr.y); rf[3] = bf_hi(rr.y);
                }
                if ((EPI == 1 || EPI == 2) && d.epi_relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pre = d.e_scale ? fmaf(rf[e], epar[BQ + cl + e], epar[2 * BQ + cl + e]) : rf[e];
                        if (!(pre > 0.f) || (d.epi_relu == 2 && pre >= 6.f)) v[e] = 0.f;
                    }
                }
                if ((EPI == 1 || EPI == 2) && want_s) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        st_a[cb][e] += v[e];
                        st_b[cb][e] += v[e] * ((rf[e] - epar[3 * BQ + cl + e]) * epar[4 * BQ + cl + e]);
                    }
                }
                if (EPI >= 2) {
                    float4 t = make_float4(v[0] * epar[BQ + cl], v[1] * epar[BQ + cl + 1], v[2] * epar[BQ + cl + 2],
                                           v[3] * epar[BQ + cl + 3]);
                    if (acc_g) { t.x += gold_[j][it].x; t.y += gold_[j][it].y; t.z += gold_[j][it].z; t.w += gold_[j][it].w; }
                    if (d.g_bf16) {
                        uint2 o;
                        o.x = pack_bf2(t.x, t.y);
                        o.y = pack_bf2(t.z, t.w);
                        *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(d.g) + p * d.ldg + q) = o;
                    } else {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.g) + p * d.ldg + q) = t;
                    }
                } else if (d.y_f32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(d.y) + p * d.ldy + q) = make_float4(v[0], v[1], v[2], v[3]);
                    if (want_stats) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { st_a[cb][e] += v[e]; st_b[cb][e] += v[e] * v[e]; }
                    }
                } else {
                    uint2 o;
                    o.x = pack_bf2(v[0], v[1]);
                    o.y = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(d.y) + p * d.ldy + q) = o;
                    if (want_stats) {
                        const float s0 = bf_lo(o.x), s1 = bf_hi(o.x), s2 = bf_lo(o.y), s3 = bf_hi(o.y);
                        st_a[cb][0] += s0; st_a[cb][1] += s1; st_a[cb][2] += s2; st_a[cb][3] += s3;
                        st_b[cb][0] += s0 * s0; st_b[cb][1] += s1 * s1; st_b[cb][2] += s2 * s2; st_b[cb][3] += s3 * s3;
                    }
                }
            }
        }
    }
    if (want_stats || want_s) {
        // Block-level reduction in LDS, then one full-width atomic instruction per 64 channels: float atomics
        // execute at the memory side and an instruction with a few active lanes costs what a full one does.
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // lanes that share ec (same channels) differ in er: reduce over them
#pragma unroll
                for (int o = LPR; o < 64; o <<= 1) {
                    st_a[cb][e] += __shfl_xor(st_a[cb][e], o, 64);
                    st_b[cb][e] += __shfl_xor(st_b[cb][e], o, 64);
                }
            }
            const int cl = wq * TQ + cb * CBW + ec * 4;
            if (er == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    atomicAdd(red + cl + e, st_a[cb][e]);
                    atomicAdd(red + BQ + cl + e, st_b[cb][e]);
                }
            }
        }
        __syncthreads();
        const int sld = (want_s && d.s_ld > 0) ? d.s_ld : cout;
        float* base = (want_s ? d.s_acc : d.stat_acc) + (size_t)rep * 2 * sld;
        for (int c = tid; c < 2 * BQ; c += NT) {
            const int half = c / BQ, cc = c - half * BQ;
            if (q_base + cc < cout) atomicAdd(base + half * sld + q_base + cc, red[c]);
        }
    }
}

template <int BP, int BQ, int WP, int WQ, int PRO, int EPI>
int launch_mode(const dsaspp_conv_desc& d, hipStream_t s) {
    const int64_t M64 = (int64_t)d.n_img * d.H * d.W;
    const int M = (int)M64;
    const int nP = (M + BP - 1) / BP, nQ = (d.cout + BQ - 1) / BQ;
    const int nblocks = nP * nQ;
    const size_t lds = (size_t)(BP + BQ * ((WP * WQ == 8 && BP < 256 && DSASPP_DMAW != 0) ? 2 : 1)) * BK * 2 + ((PRO != 0 && d.in_scale) ? (size_t)d.cin * 8 : 0) + (EPI == 3 ? (size_t)BQ * 8 : 0);
    DS_REQUIRE(lds <= 160 * 1024, "conv_fwd: LDS request %zu too large", lds);
    auto kern = conv_kernel<BP, BQ, WP, WQ, PRO, EPI>;
    if (lds > 64 * 1024) dsaspp_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * WP * WQ), lds, s, d, M, nQ, nblocks);
    DS_LAUNCH_CHECK("conv_fwd");
    return 0;
}

template <int BP, int BQ, int WP, int WQ>
int launch(const dsaspp_conv_desc& d, hipStream_t s) {
    const int pro = (d.in_mask || d.in_relu == 2) ? 2 : d.in_scale ? 1 : d.in_relu ? 2 : 0;
    const int epi = d.g ? 2 : ((d.epi_relu || d.s_acc) ? 1 : 0);
    if (epi == 0) {
        if (pro == 0) return launch_mode<BP, BQ, WP, WQ, 0, 0>(d, s);
        if (pro == 1) return launch_mode<BP, BQ, WP, WQ, 1, 0>(d, s);
        return launch_mode<BP, BQ, WP, WQ, 2, 0>(d, s);
    }
    DS_REQUIRE(pro == 0, "conv_fwd: the reference-masked and gradient-slab epilogues take a raw input (no prologue)");
    if (epi == 1) return launch_mode<BP, BQ, WP, WQ, 0, 1>(d, s);
    return launch_mode<BP, BQ, WP, WQ, 0, 2>(d, s);
}

int launch_multi(const dsaspp_conv_desc& d, hipStream_t s) { return launch_mode<128, 128, 2, 4, 0, 3>(d, s); }

}  // namespace

// name of the kernel family the calling thread's most recent convolution or weight gradient was routed to
static thread_local const char* conv_route = "";

void dsaspp_note_route(const char* name) { conv_route = name; }

extern "C" const char* dsaspp_conv_last_route(void) { return conv_route; }

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
    DS_REQUIRE(d.in_relu >= 0 && d.in_relu <= 2 && (d.in_relu != 2 || d.taps == 1), "conv_fwd: in_relu must be 0, 1, or 2 (ReLU6, 1x1 only)");
    if (d.g) {
        DS_REQUIRE(d.ldg % 8 == 0 && d.ldg >= d.cout && ((uintptr_t)d.g & 15) == 0, "conv_fwd: bad gradient slab stride/alignment");
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

    DS_REQUIRE(d.dtype == 0 || d.dtype == 1, "conv_fwd: dtype must be 0 (bf16) or 1 (fp32)");
    if (d.gx_corr1) {
        DS_REQUIRE(d.dtype == 0, "conv_fwd: the gradient read-out prologue is a bf16 form");
        conv_route = "dgrad3x3_fused";
        return dsaspp_dgrad3x3_fused_launch(d, s);
    }
    if (d.dtype == 1) {
        conv_route = "f32";
        return dsaspp_f32_conv(d, s);
    }
    if (d.n_src > 1) {
        DS_REQUIRE(d.n_src <= 4, "conv_fwd: at most 4 sources (got %d)", d.n_src);
        DS_REQUIRE(d.g && d.epi_relu && d.taps == 1 && d.ref && d.e_scale && d.e_shift && d.e_mean && d.e_rstd && !d.in_scale && !d.in_mask && !d.in_relu && !d.bias && !d.out_mask,
                   "conv_fwd: several sources need the 1x1 gradient-slab form (g, ref, e_scale/e_shift, e_mean/e_rstd, raw inputs)");
        DS_REQUIRE(d.x1 && d.w1 && d.es1 && d.eh1 && (d.n_src < 3 || (d.x2 && d.w2 && d.es2 && d.eh2)) && (d.n_src < 4 || (d.x3 && d.w3 && d.es3 && d.eh3)),
                   "conv_fwd: null operand in an extra source");
        DS_REQUIRE((((uintptr_t)d.x1 | (uintptr_t)d.w1 | (uintptr_t)d.x2 | (uintptr_t)d.w2 | (uintptr_t)d.x3 | (uintptr_t)d.w3) & 15) == 0,
                   "conv_fwd: operands must be 16-byte aligned");
        const int rc = dsaspp_dgrad_group_launch(d, s);
        conv_route = rc >= 0 ? "dgrad_group" : "multi_source";
        return rc >= 0 ? rc : launch_multi(d, s);
    }
    // slab data gradients of the masked form: persistent DMA-fed kernel.  The narrow ones too (48..144 channels a group of
    // dense layers produced itself): they are bound by the slab traffic, which that kernel keeps in flight across tiles
    // (DSASPP_DGG_MIN_COUT overrides the threshold for A/B measurements).
    static const int dgg_min_cout = [] { const char* e = getenv("DSASPP_DGG_MIN_COUT"); return e ? atoi(e) : 48; }();
    if (d.g) {           // wide single-source slab data gradients (transitions, ASPP 1x1): 256 x 256 tile
        conv_route = "dgrad_wide";
        const int rc = dsaspp_dgrad_wide_launch(d, s);
        if (rc >= 0) return rc;
    }
    if (d.g && d.cout >= dgg_min_cout) {
        conv_route = "dgrad_group";
        const int rc = dsaspp_dgrad_group_launch(d, s);
        if (rc >= 0) return rc;
    }
    if (d.taps == 9 && !d.g && d.epi_relu == 1 && d.cout >= 448) {      // 3x3 data gradient with many output channels (ASPP): 256 x 256 tile
        const int rc = dsaspp_dgrad_wide_launch(d, s);
        if (rc >= 0) {
            conv_route = "dgrad_wide9";
            return rc;
        }
    }
    if (d.taps == 9) {   // narrow-output 3x3 with a small dilation: halo-patch kernel (DSASPP_NO_HALO=1 disables it)
        static const int off = [] { const char* e = getenv("DSASPP_NO_HALO"); return e ? atoi(e) : 0; }();
        if (!off) {
            conv_route = "conv3x3_halo";
            const int rc = dsaspp_conv3x3_halo_launch(d, s);
            if (rc >= 0) return rc;
        }
    }
    if (d.taps == 1) {   // BN-ReLU-1x1 forward with few output channels: streaming kernel (DSASPP_STREAM1X1=0 disables it)
        if (d.cout > 64 && d.cout <= 192) {          // ... 65..192 of them and a K in the range where it wins: 256 x 192 / 256 x 128 tile
            conv_route = "conv_wide_narrow";
            const int rcn = dsaspp_conv_wide_launch(d, s, true);
            if (rcn >= 0) return rcn;
        }
        conv_route = "conv1x1_stream";
        const int rc = dsaspp_conv1x1_stream_launch(d, s);
        if (rc >= 0) return rc;
        conv_route = "conv_wide";                    // many output channels, long K (transitions, ASPP 1x1): 256 x 256 tile
        const int rcw = dsaspp_conv_wide_launch(d, s);
        if (rcw >= 0) return rcw;
    }
    conv_route = "conv_tile";
    const int co = d.cout;
    {   // experiment hook: DSASPP_CONV_TILE=<id> forces a tile for launches with cout > 64
        static const int forced = [] { const char* e = getenv("DSASPP_CONV_TILE"); return e ? atoi(e) : 0; }();
        if (forced && co > 64) {
            switch (forced) {
                case 1: return launch<128, 192, 2, 4>(d, s);
                case 2: return launch<128, 128, 2, 4>(d, s);
                case 5: return launch<64, 128, 1, 4>(d, s);
                case 6: return launch<64, 192, 1, 4>(d, s);
                case 8: return launch<128, 64, 2, 2>(d, s);
                case 9: return launch<256, 192, 4, 2>(d, s);
                default: break;
            }
        }
    }
    if (co <= 32) return launch<128, 32, 4, 1>(d, s);
    if (co <= 48) return launch<256, 48, 4, 1>(d, s);
    if (co <= 64) return launch<128, 64, 2, 2>(d, s);
    const int pad192 = (co + 191) / 192 * 192, pad128 = (co + 127) / 128 * 128;
    if (pad192 <= pad128) return launch<128, 192, 2, 4>(d, s);
    return launch<128, 128, 2, 4>(d, s);
}
