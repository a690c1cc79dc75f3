// BN-ReLU -> 1x1 convolution forward with many output channels and a long K (transitions, ASPP 1x1 layers), gfx950:
//
//   y[p][q] = bias[q] + sum_k bf16(relu(x[p][k] * in_scale[k] + in_shift[k])) * w[q][k]        (+ per-channel sums of y, y^2)
//
// on the 256 x 256 x 32 step of dgrad_wide.hip (8 waves as 2 x 4, wave tile 128 pixels x 64 channels, 128 accumulator
// registers, persistent blocks walking (pixel tile, channel tile) pairs XCD by XCD, ring of three 32 KiB stages with
// 64-byte rows and the XOR chunk swizzle).  The filter half of a stage arrives by LDS-DMA.  The pixel half cannot -- the
// prologue has to be applied on the way -- so it takes the register path: a thread owns two 16-byte chunks of the
// stage's 256 pixel rows, requests them global -> registers TWO steps ahead (inline assembly, so that the compiler does
// not wait for them where it pleases), and after the MFMAs of step t applies scale / shift / rounding / ReLU to the chunks
// of step t + 1 and writes them into that stage's slot; the barrier of step t + 1 publishes them together with the
// filter rows.  Scale and shift of the whole K live in LDS, zero-padded to a multiple of the K-step, so that a chunk past
// the last channel (fetched from the row's first chunk: finite values) becomes zero by itself.
// Queue per wave and step: [2 filter DMAs][2 pixel-chunk loads]; the one counted wait of a step, vmcnt(4), sits in front
// of the prologue work and covers the chunks of step t + 1 and, being older, the filter rows of step t + 1.
// No pixel-chunk load is in flight across a tile's epilogue (a register that is the target of a load in flight must not be
// touched, and an epilogue needs every register): the last step of a tile requests its chunks after the epilogue.
#include <stdlib.h>

#include "common.h"

// -DCW_ABL_X=1: diagnostic build in which every block reads the same 1024 pixel rows (cache-resident pixel operand)
#ifndef CW_ABL_X
#define CW_ABL_X 0
#endif

namespace {

constexpr int BP = 256, BK = 32;
constexpr int NW = 8, NT = 512, WP = 2, WQ = 4;
constexpr int TP = BP / WP, MP = TP / 16;                                 // a wave owns 128 pixels: 8 MFMA tiles
constexpr int NS = 3;                                                     // ring stages
constexpr int NGRP = MP, NRP = 4;                                         // 16-pixel groups of the wave tile, 4-pixel passes per group
// Per output-channel width of the block tile.  256: wave tile 128 x 64 (transitions, ASPP 1x1).  192: wave tile 128 x 48
// (the dense layers' bottleneck convolution, whose 192 output channels would leave a quarter of the 256-wide tile empty):
// the filter half of a stage is 12 row groups of 16, so waves 4..7 issue one filter DMA per step instead of two.
template <int BQ>
struct Geo {
    static constexpr int TQ = BQ / WQ, MQ = TQ / 16;
    static constexpr int STAGE = (BP + BQ) * BK * 2;                      // pixel half, then filter half (32 / 28 KiB)
    static constexpr int ROWF = TQ + 4;                                   // patch row in floats (bank spread)
    static constexpr int PATCH_BYTES = NW * 16 * ROWF * 4;
    static constexpr int SRED_BYTES = WP * 2 * BQ * 4;                    // [pixel half][which sum][channel]
    static constexpr int FIXED_LDS = NS * STAGE + PATCH_BYTES + SRED_BYTES;
    static constexpr int NDMA_HI = BQ / 16 - NW;                          // waves below this index issue a second filter DMA
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct CWArgs {
    const u16* x;
    const u16* w;
    const float* sc;
    const float* sh;
    const float* bias;
    float* stat;
    u16* y;
    int ldx, ldy, cin, cinp, cout, M, nP, nQ;
    uint32_t relu_bits;
};

// chunk swizzle of a 64-byte stage row: position of K chunk c in row r is c ^ swz(r) (see dgrad_wide.hip)
__device__ __forceinline__ int swz(int row) { return (0 - (row >> 2)) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// ODD: the number of K-steps per tile is odd (the steps alternate between two register buffers; a kernel-wide parity
// keeps the end of a tile straight-line code instead of two paths that meet in front of the epilogue)
template <int BQ, bool STATS, bool ODD>
__global__ __launch_bounds__(NT, 2) void conv_wide_kernel(const CWArgs a) {
    constexpr int TQ = Geo<BQ>::TQ, MQ = Geo<BQ>::MQ, STAGE = Geo<BQ>::STAGE, ROWF = Geo<BQ>::ROWF;
    constexpr int PATCH_BYTES = Geo<BQ>::PATCH_BYTES, FIXED_LDS = Geo<BQ>::FIXED_LDS, NDMA_HI = Geo<BQ>::NDMA_HI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* patch = reinterpret_cast<float*>(smem + NS * STAGE);
    float* sred = reinterpret_cast<float*>(smem + NS * STAGE + PATCH_BYTES);
    float* tab = reinterpret_cast<float*>(smem + FIXED_LDS);               // [scale cinp | shift cinp]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = a.cin, cinp = a.cinp, cout = a.cout, M = a.M, nQ = a.nQ;
    const int ksteps = cinp / BK;

    // ---- this block's tile sequence: XCD x = blockIdx % 8 owns the pixel tiles pt = x (mod 8)
    const int xcd = blockIdx.x & 7, jblk = blockIdx.x >> 3, g8 = gridDim.x >> 3;
    const int nPx = xcd < a.nP ? (a.nP - xcd + 7) >> 3 : 0;
    const int Lx = nPx * nQ;
    const int ntile = jblk < Lx ? (Lx - jblk + g8 - 1) / g8 : 0;
    if (ntile == 0) return;
    auto tile_coords = [&](int k, int& p_base, int& q_base) {
        const int r = k * g8 + jblk;
        const int ptl = r / nQ;
        p_base = (ptl * 8 + xcd) * BP;
        q_base = (r - ptl * nQ) * BQ;
    };

    // (without a scale / shift pair the table holds 1 and 0: x * 1 + 0 is exact, the layer is ReLU -> conv or a plain GEMM)
    for (int c = tid; c < cinp; c += NT) {
        tab[c] = c < cin ? (a.sc ? a.sc[c] : 1.f) : 0.f;
        tab[cinp + c] = (c < cin && a.sh) ? a.sh[c] : 0.f;
    }

    // ---- operand cursor (two steps ahead of the arithmetic).  Filter rows: DMA instruction i of a wave fills stage rows
    //      256 + (i*8 + wave)*16 .. +15, a lane brings the chunk that belongs at its lane-linear LDS position.  Pixel
    //      rows: thread t owns chunk position t & 3 of rows t >> 2 and (t >> 2) + 128 (both rows have the same swizzle).
    const uint32_t ring_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)ring);
    int c_k = 0, c_kk = 0, c_slot = 0;
    uint32_t w_off[2], x_off[2];                      // byte offsets of this lane's rows in w and, from the tile's first row, in x
    const u16* x_tile = a.x;                          // first pixel row of the cursor's tile (uniform; x may be larger than 4 GiB)
    // (lane- and thread-derived coordinates are formed where they are used, from an opaque copy of the thread id: kept as
    // values that live through the whole kernel they are spilled in favour of the epilogue, and a scratch reload inside
    // the K-loop is waited for with vmcnt(0), which drains the DMA queue)
    auto opaque_tid = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        return t;
    };
    auto cursor_tile = [&]() {
        int pb, qb;
        tile_coords(c_k, pb, qb);
        const int t = opaque_tid();
        const int lrow = (t & 63) >> 2, xrow = t >> 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int q = qb + (i * 8 + wave) * 16 + lrow;
            q = q < cout ? q : cout - 1;
            w_off[i] = (uint32_t)q * (uint32_t)cin * 2u;
            int p = pb + xrow + i * 128;
            p = p < M ? p : M - 1;
#if CW_ABL_X
            p &= 1023;                                 // diagnostic build: every block reads the same cache-resident rows
#endif
#if CW_ABL_X
            x_off[i] = (uint32_t)p * (uint32_t)a.ldx * 2u;
#else
            x_off[i] = (uint32_t)(p - pb) * (uint32_t)a.ldx * 2u;
#endif
        }
#if !CW_ABL_X
        x_tile = a.x + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane(pb) * a.ldx;
#endif
    };
    uint32_t xo_n[2];                                 // pixel-chunk offsets of the stage whose filter rows were requested last
    const u16* xb_n = a.x;                            // ... and the tile base they are relative to
    int kk_n = 0;                                     // ... and its K-step
    auto issue_w = [&]() {
        const uint32_t slot_l = ring_l + c_slot * STAGE + BP * 64;
        // a chunk past the last channel: the row's first chunk instead (finite values of the same tensor; the pixel side
        // of that chunk is zero)
        const int t = opaque_tid();
        const int lchunk = (t & 3) ^ swz((t & 63) >> 2), xkc = (t & 3) ^ swz(t >> 2);      // the K chunks that belong at this lane's positions
        const uint32_t kbw = (c_kk * BK + lchunk * 8 < cin) ? (uint32_t)(c_kk * BK + lchunk * 8) * 2u : 0u;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t off = w_off[i] + kbw;
            if (i == 0 || NDMA_HI >= NW || wave < NDMA_HI)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             : : "s"(slot_l + (i * 8 + wave) * 1024), "v"(off), "s"(a.w) : "memory");
        }
        const uint32_t kbx = (c_kk * BK + xkc * 8 < cin) ? (uint32_t)(c_kk * BK + xkc * 8) * 2u : 0u;
        xo_n[0] = x_off[0] + kbx;
        xo_n[1] = x_off[1] + kbx;
        xb_n = x_tile;
        kk_n = c_kk;
        c_slot = c_slot + 1 < NS ? c_slot + 1 : 0;
        if (c_k < ntile) {
            if (++c_kk == ksteps) {
                if (c_k + 1 < ntile) {
                    c_kk = 0;
                    ++c_k;
                    cursor_tile();
                } else {
                    c_k = ntile;                      // frozen on the last stage: it is requested again into a free slot
                    --c_kk;
                }
            }
        }
    };
    // pixel chunks of the stage issue_w() was last called for, into one of two register buffers
    // the one counted wait of a step: everything but this step's own requests has landed (a wave that issues one filter
    // DMA per step has one request fewer in flight; `last`: the step requested no pixel chunks)
    auto wait_landed = [&](const bool last) {
        if (NDMA_HI >= NW || wave < NDMA_HI) {
            if (last) wait_vm<2>();
            else wait_vm<4>();
        } else {
            if (last) wait_vm<1>();
            else wait_vm<3>();
        }
    };
    auto issue_x = [&](u32x4 (&buf)[2], int& kk_buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(buf[i]) : "v"(xo_n[i]), "s"(xb_n) : "memory");
        kk_buf = kk_n;
    };
    // prologue of a landed buffer into the slot of its stage (the ring slot after the one being read)
    auto consume = [&](u32x4 (&buf)[2], const int kk_buf, const int slot) {
        asm volatile("" : "+v"(buf[0]), "+v"(buf[1]));        // from here on the registers of the loads are ordinary values
        const int t = opaque_tid();
        const int xrow = t >> 2, xpos = t & 3, xkc = xpos ^ swz(xrow);
        float sc[8], sh[8];
        load8f(tab + kk_buf * BK + xkc * 8, sc);
        load8f(tab + cinp + kk_buf * BK + xkc * 8, sh);
        unsigned char* dst = ring + slot * STAGE + xrow * 64 + xpos * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float f[8];
            unpack8(__builtin_bit_cast(uint4, buf[i]), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
            const uint4 v = relu_bf8(pack8(f), a.relu_bits);          // ReLU after the rounding (packed 16-bit max)
            *reinterpret_cast<uint4*>(dst + i * 128 * 64) = v;
        }
    };

    // ---- fragment coordinates
    const int wp = wave % WP, wq = wave / WP;
    const int rep = blockIdx.x % DS_R;

    // ---- start: stages 0 and 1 requested, stage 0's pixel half written
    u32x4 xa[2], xb[2];
    int kk_a = 0, kk_b = 0;
    cursor_tile();
    issue_w();
    issue_x(xb, kk_b);
    issue_w();
    issue_x(xa, kk_a);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                     // tables visible
    asm volatile("" ::: "memory");
    wait_landed(false);
    consume(xb, kk_b, 0);
    int r_slot = 0;

    for (int k = 0; k < ntile; ++k) {
        int p_base, q_base;
        tile_coords(k, p_base, q_base);
        f32x4 acc[MQ][MP];
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (the fragment offset is formed per tile from an opaque copy of the lane id: see dgrad_wide.hip)
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        const int fr = lane_k & 15;
        const int frag_off = fr * 64 + (((lane_k >> 4) ^ swz(fr)) << 4);

        // one K-step: stage r_slot is complete after the barrier (this wave's filter DMAs were waited for in front of the
        // previous step's prologue work, everybody's pixel chunks are written); request stage t + 2, MFMAs of stage t,
        // then the pixel half of stage t + 1 from the buffer that has landed
        auto mfmas = [&]() {
            const unsigned char* As = ring + r_slot * STAGE;
            const unsigned char* Ws = As + BP * 64;
            bf16x8 wf[MQ];
#pragma unroll
            for (int i = 0; i < MQ; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(Ws + (wq * TQ + i * 16) * 64 + frag_off);
#pragma unroll
            for (int jh = 0; jh < MP; jh += 4) {
                bf16x8 af[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const bf16x8*>(As + (wp * TP + (jh + j) * 16) * 64 + frag_off);
#pragma unroll
                for (int i = 0; i < MQ; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][jh + j], 0, 0, 0);
            }
        };
        auto step = [&](u32x4 (&landed)[2], const int kk_landed, u32x4 (&next)[2], int& kk_next, const bool last) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_w();
            if (!last) issue_x(next, kk_next);
            mfmas();
            wait_landed(last);
            const int w_slot = r_slot + 1 < NS ? r_slot + 1 : 0;
            consume(landed, kk_landed, w_slot);
            r_slot = w_slot;
        };
        for (int s = 0; s + 2 < ksteps; s += 2) {
            step(xa, kk_a, xb, kk_b, false);
            step(xb, kk_b, xa, kk_a, false);
        }
        // one or two steps are left; the last one requests no pixel chunks
        if constexpr (ODD) {
            step(xa, kk_a, xb, kk_b, true);
        } else {
            step(xa, kk_a, xb, kk_b, false);
            step(xb, kk_b, xa, kk_a, true);
        }

        // ---- epilogue, 16 pixels of the wave tile at a time, in row layout (a pixel's 64 channels = 128 contiguous
        //      bytes across 16 lanes): bias, rounding, store, sums of the rounded values.  Rows past M and channel quads
        //      past cout are neither stored nor summed.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int er = lane_e >> 4, ec = lane_e & 15;        // row layout: pixel er of a 4-pixel pass, channel quad ec
        const int fr_e = lane_e & 15, g_e = lane_e >> 4;     // accumulator layout
        float* stage = patch + wave * (16 * ROWF);
        const bool qin = ec * 4 < TQ;                       // (a 48-channel wave tile: quads 12..15 of a pass are idle)
        const int ecc = qin ? ec : 0;
        const int qw = q_base + wq * TQ + ecc * 4;          // this lane's channel quad (cout is a multiple of 4)
        const bool qok = qin && qw < cout;
        const int cl = wq * TQ + ecc * 4;
        const int p_w = wp * TP + er;                       // counted from the tile's first row
        const int m_rel = M - p_base;
        const uint32_t qoff = (uint32_t)(qok ? qw : 0) * 2u, ldy2 = (uint32_t)a.ldy * 2u;
        unsigned char* yb = reinterpret_cast<unsigned char*>(a.y) + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane(p_base) * ldy2;      // the tile's first row: uniform 64-bit base, 32-bit offsets inside the tile
        f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (a.bias && qok) b4 = *reinterpret_cast<const f32x4*>(a.bias + qw);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NGRP; ++j) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MQ; ++i) *reinterpret_cast<f32x4*>(stage + fr_e * ROWF + i * 16 + g_e * 4) = acc[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(stage + (rp * 4 + er) * ROWF + ecc * 4);
                int p = p_w + j * 16 + rp * 4;
                asm volatile("" : "+v"(p));
                const bool ok = qok && p < m_rel;
                const u32x2 o = u32x2{pack_bf2(a4[0] + b4[0], a4[1] + b4[1]), pack_bf2(a4[2] + b4[2], a4[3] + b4[3])};
                if (STATS) {
                    const float r[4] = {bf_lo(o.x), bf_hi(o.x), bf_lo(o.y), bf_hi(o.y)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = ok ? r[e] : 0.f;
                        s1[e] += v;
                        s2[e] = fmaf(v, v, s2[e]);
                    }
                    asm volatile("" : "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3]), "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]));
                }
                if (ok) *reinterpret_cast<u32x2*>(yb + (size_t)((uint32_t)p * ldy2 + qoff)) = o;
            }
        }
        if (STATS) {
            // sums over the wave's 128 pixels: the four lanes that share a channel quad, then one plain store per wave
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[e] += __shfl_xor(s1[e], 16, 64);
                s1[e] += __shfl_xor(s1[e], 32, 64);
                s2[e] += __shfl_xor(s2[e], 16, 64);
                s2[e] += __shfl_xor(s2[e], 32, 64);
            }
            if (er == 0 && qin) {
                *reinterpret_cast<f32x4*>(sred + (wp * 2 + 0) * BQ + cl) = f32x4{s1[0], s1[1], s1[2], s1[3]};
                *reinterpret_cast<f32x4*>(sred + (wp * 2 + 1) * BQ + cl) = f32x4{s2[0], s2[1], s2[2], s2[3]};
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int which = tid / BQ, cc = tid - which * BQ;       // 2 sums x BQ channels on the first 2 * BQ threads
            if (which < 2 && q_base + cc < cout)
                atomicAdd(a.stat + (size_t)rep * 2 * cout + which * cout + q_base + cc, sred[which * BQ + cc] + sred[(2 + which) * BQ + cc]);
        }
        // the pixel chunks of the next tile's second stage (requested late: nothing in flight across the epilogue)
        issue_x(xa, kk_a);
    }
    // nothing of this block may land in LDS or in a register after it has gone
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xb[0]), "+v"(xb[1]));
}

}  // namespace

// Form check + launch.  Returns -1 when the descriptor is not of the form this kernel is for (the caller then uses the
// generic kernel), 0 on success, a positive error code otherwise.  129..192 output channels take the 192-wide tile
// (`narrow_only`: the caller asks for that form alone, in front of the streaming kernel).
template <int BQ>
static int conv_wide_launch_t(const dsaspp_conv_desc& d, hipStream_t s) {
    const int cinp = (d.cin + BK - 1) / BK * BK;
    const size_t lds = (size_t)Geo<BQ>::FIXED_LDS + (size_t)cinp * 8;
    if (lds > 160 * 1024) return -1;
    const int64_t M64 = (int64_t)d.n_img * d.H * d.W;
    // operand and output offsets are 32-bit byte offsets from the tensor base
    // pixel rows are addressed from a uniform 64-bit tile base (tensors may exceed 4 GiB); inside a tile and in the filter,
    // offsets are 32-bit
    if ((int64_t)d.cout * d.cin * 2 >= (1ll << 32) || M64 >= (1ll << 31) - BP || (int64_t)BP * d.ldx * 2 >= (1ll << 31) || (int64_t)BP * d.ldy * 2 >= (1ll << 31)) return -1;
    if ((((uintptr_t)d.x | (uintptr_t)d.w) & 15) != 0 || ((uintptr_t)d.y & 7) != 0 || (d.bias && ((uintptr_t)d.bias & 15) != 0)) return -1;
    CWArgs a;
    a.x = reinterpret_cast<const u16*>(d.x);
    a.w = reinterpret_cast<const u16*>(d.w);
    a.sc = d.in_scale;
    a.sh = d.in_shift;
    a.bias = d.bias;
    a.stat = d.stat_acc;
    a.y = reinterpret_cast<u16*>(d.y);
    a.ldx = d.ldx;
    a.ldy = d.ldy;
    a.cin = d.cin;
    a.cinp = cinp;
    a.cout = d.cout;
    a.M = (int)M64;
    a.nP = (a.M + BP - 1) / BP;
    a.nQ = (d.cout + BQ - 1) / BQ;
    a.relu_bits = d.in_relu ? 0u : 0x80008000u;
    const int n_cu = dsaspp_persistent_cus();
    {
        const void* kernels[] = {(const void*)conv_wide_kernel<BQ, true, true>, (const void*)conv_wide_kernel<BQ, true, false>,
                                 (const void*)conv_wide_kernel<BQ, false, true>, (const void*)conv_wide_kernel<BQ, false, false>};
        for (const void* f : kernels) dsaspp_lds_limit(f, 160 * 1024);
    }
    const int64_t tiles = (int64_t)a.nP * a.nQ;
    int grid = (int)((tiles + 7) / 8 * 8);
    int cap = n_cu / 8 * 8;
    if (const char* e = getenv("DSASPP_DGG_GRID")) {      // test hook: fewer blocks, so that small cases walk several tiles per block
        const int v = atoi(e);
        if (v >= 8) cap = v / 8 * 8;
    }
    if (grid > cap) grid = cap;
    if (grid < 8) grid = 8;
    const bool odd = ((cinp / BK) & 1) != 0;
    void (*kern)(const CWArgs) = d.stat_acc ? (odd ? conv_wide_kernel<BQ, true, true> : conv_wide_kernel<BQ, true, false>)
                                            : (odd ? conv_wide_kernel<BQ, false, true> : conv_wide_kernel<BQ, false, false>);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, a);
    DS_LAUNCH_CHECK("conv_fwd(wide forward 1x1)");
    return 0;
}

int dsaspp_conv_wide_launch(const dsaspp_conv_desc& d, hipStream_t s, bool narrow_only) {
    static const int enabled = [] { const char* e = getenv("DSASPP_CONV_WIDE"); return e ? atoi(e) : 1; }();
    if (!enabled) return -1;
    const bool form = d.taps == 1 && d.y && !d.y_f32 && !d.g && !d.epi_relu && !d.s_acc && ((d.in_scale != nullptr) == (d.in_shift != nullptr)) && !d.in_mask &&
                      (d.in_relu == 0 || d.in_relu == 1) && !d.out_mask && d.n_src <= 1 && d.dtype == 0 && !d.gx_corr1 && d.cin % 8 == 0 &&
                      d.ldx % 8 == 0 && d.cout % 4 == 0 && d.ldy % 4 == 0;
    if (!form) return -1;
    // the dense layers' bottleneck convolution (DSASPP_CONV_WIDE192=0: left to the streaming kernel)
    static const int en192 = [] { const char* e = getenv("DSASPP_CONV_WIDE192"); return e ? atoi(e) : 1; }();
    // (measured at 16 x 128 x 256 pixels against the streaming kernel: 90 vs 128 us at cin 96, 229 vs 256 at 480, 466 vs 474
    // at 1248, 546 vs 540 at 1488, 707 vs 686 at 2064: the long-K layers stay with the streaming kernel)
    static const int min_cin192 = [] { const char* e = getenv("DSASPP_CONV_WIDE192_MIN_CIN"); return e ? atoi(e) : 64; }();
    static const int max_cin192 = [] { const char* e = getenv("DSASPP_CONV_WIDE192_MAX_CIN"); return e ? atoi(e) : 1392; }();
    if (en192 && d.cout > 128 && d.cout <= 192 && d.cin >= min_cin192 && d.cin <= max_cin192) return conv_wide_launch_t<192>(d, s);
    // 65..128 output channels (the bottleneck convolution of the growth-32 trunks): the same kernel on a 256 x 128 tile
    // (these launches are bound by the activation stream; against the streaming kernel at 16 x 128 x 256 pixels: 80 vs 92 us
    // at cin 128, 116 vs 127 at 256, 196 vs 199 at 512, 373 vs 340 at 1024)
    static const int en128 = [] { const char* e = getenv("DSASPP_CONV_WIDE128"); return e ? atoi(e) : 1; }();
    static const int max_cin128 = [] { const char* e = getenv("DSASPP_CONV_WIDE128_MAX_CIN"); return e ? atoi(e) : 512; }();
    if (en128 && d.cout > 64 && d.cout <= 128 && d.cin >= 64 && d.cin <= max_cin128) return conv_wide_launch_t<128>(d, s);
    if (narrow_only) return -1;
    static const int min_cin = [] { const char* e = getenv("DSASPP_CONV_WIDE_MIN_CIN"); return e ? atoi(e) : 512; }();
    static const int min_cout = [] { const char* e = getenv("DSASPP_CONV_WIDE_MIN_COUT"); return e ? atoi(e) : 448; }();
    if (d.cin < min_cin || d.cout < min_cout) return -1;
    return conv_wide_launch_t<256>(d, s);
}
