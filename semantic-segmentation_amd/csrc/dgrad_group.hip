// Grouped 1x1 data gradient into the gradient slab, gfx950.
//
//   g[p][c] (+)= sum_s  e_scale_s[c] * [ref[p][c] * e_scale_s[c] + e_shift_s[c] > 0] * (x_s[p][:] . w_s[c][:])
//
// for up to four consumer layers s of the same slab prefix, with each layer's BatchNorm-backward
// sums as a by-product.  K (the bottleneck width) is short, so a launch is a long stream of short
// GEMMs; the kernel is built around keeping that stream's memory traffic in flight across every
// phase change (K-step, source, tile):
//   * persistent blocks, one per CU, walk (pixel tile, channel tile) pairs; tiles that run at
//     the same time on one XCD share their pixel tile (and its L2 lines);
//   * operands go global -> LDS by DMA (global_load_lds, 16 B per lane) into a ring of 4 stages of
//     one 64-wide K-step each; the XOR chunk swizzle lives in the per-lane source address, the LDS
//     image is lane-linear.  Three stages are in flight across the per-step barrier and across
//     sources and tiles;
//   * the next tile's reference activations, old slab values (registers) and per-channel
//     parameters (DMA into LDS) are requested at the end of the current tile and land while the
//     next tile's first K-steps run;
//   * EVERY vector-memory instruction of the loop is issued from inline assembly and waited for by
//     a hand-counted s_waitcnt: the compiler, left to itself, drains the whole queue (vmcnt(0))
//     at the first use of any loaded register and before any LDS write while an LDS-DMA is
//     outstanding -- three full pipeline drains per tile in the first version of this kernel;
//   * each source's product is masked / summed / scaled in accumulator layout into a running total;
//     the slab tile is read and written once per tile, through a wave-private LDS patch so that the
//     accesses are full 128-byte lines.
//
// Counting rule.  Per wave the queue is, in issue order, [DMA of step g: 4 ops][what step g issued
// besides].  Step g waits for its own stage, issued three steps earlier, so the number of younger
// operations is 8 (two DMA stages) + the extra operations of the last three steps; lower bounds of
// those are tracked in three scalars.  A wait count below the true number only waits longer.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef DGG_NT
#define DGG_NT 0      // 1: reference loads, slab loads and slab stores carry the non-temporal hint (streamed once: keep the weights in L2)
#endif
#if DGG_NT
#define DGG_NTS " nt"
#else
#define DGG_NTS ""
#endif
#ifndef DGG_ABL
#define DGG_ABL 0     // diagnostic builds (timing only, results wrong): 1 no fragment reads / MFMA, 2 no per-source mask / sums, 3 no operand DMA,
                      // 4 no side loads and no slab stores, 6 no per-step barrier
#endif

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
constexpr int SRED_BYTES = WP * 4 * 2 * BQ * 4;                        // [pixel half of the tile][source][which sum][channel]
constexpr int LDS_BYTES = NS * STAGE + PATCH_BYTES + PAR_BYTES + SRED_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(16 * 72 <= 16 * ROWF * 4, "reference patch");
static_assert(MQ == 2 && MP == 4, "register tie statements are written for a 64 x 32 wave tile");
constexpr int N_PAR = 3;                  // parameter DMAs per wave and tile (20 half rows over 8 waves, padded)
constexpr int N_REF = MQ * MP;            // reference loads per wave and tile
constexpr int RPP = 8, NRP = 2;           // slab rows per pass of the read-modify-write, passes per 16 rows
constexpr int N_G = MP * NRP;             // slab loads / stores per wave and tile

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct GArgs {
    const u16* x[4];
    const u16* w[4];
    const float* es[4];
    const float* eh[4];
    float* sa[4];
    int sl[4];
    const float* mean;
    const float* rstd;
    const u16* ref;
    void* g;
    int ldx, ldref, ldg, cin, cout, nsrc, M, nP, nQ;
};

__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)p);
}

// wait until at most c vector-memory operations are outstanding (rounded down: safe); LDS operations of
// this wave are retired too, so that what it wrote is visible after the barrier that follows
__device__ __forceinline__ void wait_vm(int c) {
    if (c == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (c >= 62) asm volatile("s_waitcnt vmcnt(62) lgkmcnt(0)" ::: "memory");
    else if (c >= 54) asm volatile("s_waitcnt vmcnt(54) lgkmcnt(0)" ::: "memory");
    else if (c >= 46) asm volatile("s_waitcnt vmcnt(46) lgkmcnt(0)" ::: "memory");
    else if (c >= 35) asm volatile("s_waitcnt vmcnt(35) lgkmcnt(0)" ::: "memory");
    else if (c >= 27) asm volatile("s_waitcnt vmcnt(27) lgkmcnt(0)" ::: "memory");
    else if (c >= 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
    else if (c >= 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    else if (c >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if (c >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (c >= 11) asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory");
    else if (c >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (c >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <bool ACCG, bool GBF16, int KS>
__global__ __launch_bounds__(NT, 2) void dgrad_group_kernel(const GArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* patch = reinterpret_cast<float*>(smem + NS * STAGE);
    float* par = reinterpret_cast<float*>(smem + NS * STAGE + PATCH_BYTES);
    float* sred = reinterpret_cast<float*>(smem + NS * STAGE + PATCH_BYTES + PAR_BYTES);
    typedef typename std::conditional<GBF16, u32x2, f32x4>::type gvec_t;      // 4 slab values of a lane
    constexpr int GB = GBF16 ? 2 : 4;                                        // bytes per slab element

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = a.cin, cout = a.cout, M = a.M, nQ = a.nQ;
    const int ksteps = (cin + BK - 1) / BK;
    const int rem = cin - (ksteps - 1) * BK;         // channels of a source's last K-step (a multiple of 8)
    const bool has_tail = rem < BK;
    const int nsrc = a.nsrc;

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

    // ---- operand DMA cursor (runs DEPTH steps ahead of the arithmetic)
    const uint32_t ring_l = lds_addr(ring);
    const int drow = lane >> 3, dchunk = lane & 7;
    int c_k = 0, c_s = 0, c_kk = 0, c_slot = 0;
    uint32_t a_off[2], w_off[2];                     // byte offsets of this lane's two rows
    const uint32_t lcb = (uint32_t)((dchunk ^ swz8(wave * 8 + drow)) * 16);      // this lane's (swizzled) chunk inside the 128-byte K window
    const u16* cur_x = a.x[0];
    const u16* cur_w = a.w[0];
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
            a_off[i] = ((uint32_t)p * (uint32_t)a.ldx + lc * 8) * 2u;
            w_off[i] = ((uint32_t)q * (uint32_t)cin + lc * 8) * 2u;
        }
    };
    // one stage = 4 DMA instructions per wave; past the last stage the previous one is requested again
    // (same bytes into a free slot), so that every step issues exactly four
    auto issue = [&]() {
        const uint32_t slot_l = ring_l + c_slot * STAGE;
        const uint32_t kb = c_kk * (BK * 2);
        uint32_t ao[2] = {a_off[0] + kb, a_off[1] + kb}, wo[2] = {w_off[0] + kb, w_off[1] + kb};
        if (has_tail && c_kk == ksteps - 1 && kb + lcb >= (uint32_t)cin * 2u) {
            // a chunk past the last channel: fetch the row's first chunk instead (finite values of the same tensor);
            // the step's MFMAs see zeros in its place
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ao[i] = a_off[i] - lcb;
                wo[i] = w_off[i] - lcb;
            }
        }
        if (DGG_ABL != 3)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         : : "s"(slot_l + (i * 8 + wave) * 1024), "v"(ao[i]), "s"(cur_x) : "memory");
        if (DGG_ABL != 3)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         : : "s"(slot_l + BP * 128 + (i * 8 + wave) * 1024), "v"(wo[i]), "s"(cur_w) : "memory");
        c_slot = (c_slot + 1) & (NS - 1);
        if (c_k < ntile) {
            if (++c_kk == ksteps) {
                if (++c_s == nsrc) {
                    if (c_k + 1 < ntile) {
                        c_kk = 0;
                        c_s = 0;
                        ++c_k;
                        cursor_tile();
                    } else {
                        c_k = ntile;               // frozen on the last stage
                        --c_kk;
                        --c_s;
                    }
                } else {
                    c_kk = 0;
                }
                cur_x = a.x[c_s];
                cur_w = a.w[c_s];
            }
        }
    };

    // ---- fragment coordinates
    const int fr = lane & 15, fc = lane >> 4;
    const int wp = wave % WP, wq = wave / WP;
    const int frag_off0 = fr * 128 + ((fc ^ swz8(fr)) << 4);
    const int frag_off1 = fr * 128 + (((4 + fc) ^ swz8(fr)) << 4);
    const int rep = blockIdx.x % DS_R;
    float* stage = patch + wave * (16 * ROWF);
    const int er = lane >> 3, ec = lane & 7;
    const bool b0 = fr & 1, b1 = fr & 2;
    const int sred_lane = ((fr >> 2) & 1) * BQ + wq * TQ + (fr >> 3) * 16 + fc * 4 + (fr & 3);      // [which sum][channel] of the total this lane keeps

    // ---- per-tile side loads
    // parameters: 20 half rows (10 rows x 2 x 64 channels), one DMA instruction each, 4 B per lane
    auto issue_par = [&](int qb) {
        const uint32_t par_l = lds_addr(par);
#pragma unroll
        for (int t = 0; t < N_PAR; ++t) {
            int idx = wave + 8 * t;
            idx = idx < 2 * PAR_ROWS ? idx : 2 * PAR_ROWS - 1;
            const int row = idx >> 1, half = idx & 1;
            const int sidx = (row & 3) < nsrc ? (row & 3) : 0;
            const float* base = row < 4 ? a.es[sidx] : row < 8 ? a.eh[sidx] : row == 8 ? a.mean : a.rstd;
            int q = qb + half * 64 + lane;
            q = q < cout ? q : cout - 1;
            const float* src = base + q;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(par_l + idx * 256), "v"(src) : "memory");
        }
    };
    // Reference activations of a tile.  They are fetched in the row layout of the slab read-modify-write (8 lanes cover 64
    // contiguous bytes of a pixel; in accumulator layout a load instruction touches 16 rows for 32 bytes each and the
    // tile's eight loads cost 11 % of the kernel) and reach the accumulator layout through the wave's LDS patch.
    f32x4 rfm[MQ][MP];      // unpacked to fp32 once per tile: every source's mask and second sum read them
    u32x2 rfp[MP][NRP];
    auto issue_ref = [&](int pb, int qb) {
        const u16* base = a.ref + (int64_t)pb * a.ldref;
#pragma unroll
        for (int j = 0; j < MP; ++j)
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                int p = pb + wp * TP + j * 16 + rp * RPP + er;
                p = (p < M ? p : M - 1) - pb;
                int q = qb + wq * TQ + ec * 4;
                q = q < cout - 4 ? q : cout - 4;
                const uint32_t off = ((uint32_t)p * (uint32_t)a.ldref + q) * 2u;
                asm volatile("global_load_dwordx2 %0, %1, %2" DGG_NTS : "=v"(rfp[j][rp]) : "v"(off), "s"(base) : "memory");
            }
    };
    gvec_t gold[MP][NRP];
    auto slab_off = [&](int pb, int qb, int j, int rp) -> uint32_t {
        int p = pb + wp * TP + j * 16 + rp * RPP + er;
        p = (p < M ? p : M - 1) - pb;
        int q = qb + wq * TQ + ec * 4;
        q = q < cout - 4 ? q : cout - 4;
        return ((uint32_t)p * (uint32_t)a.ldg + q) * (uint32_t)GB;
    };
    auto issue_gold = [&](int pb, int qb) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(a.g) + (int64_t)pb * a.ldg * GB;
#pragma unroll
        for (int j = 0; j < MP; ++j)
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                const uint32_t off = slab_off(pb, qb, j, rp);
                if (GBF16) asm volatile("global_load_dwordx2 %0, %1, %2" DGG_NTS : "=v"(gold[j][rp]) : "v"(off), "s"(base) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, %2" DGG_NTS : "=v"(gold[j][rp]) : "v"(off), "s"(base) : "memory");
            }
    };

    // ---- prologue: the first tile's side loads, then DEPTH operand stages (the side loads are older than every
    //      stage, so the first step's wait covers them)
    {
        int pb, qb;
        tile_coords(0, pb, qb);
        issue_par(qb);
        issue_ref(pb, qb);
        if (!ACCG) {
#pragma unroll
            for (int j = 0; j < MP; ++j)
#pragma unroll
                for (int rp = 0; rp < NRP; ++rp) gold[j][rp] = gvec_t{};
        }
    }
    cursor_tile();
    for (int i = 0; i < DEPTH; ++i) issue();

    int x1 = 0, x2 = 0, x3 = 0;        // lower bounds of the non-DMA operations issued in the last three steps
    int e1_bound = 0;                  // operations certainly younger than the side loads of the current tile, DMA stages aside

    for (int k = 0; k < ntile; ++k) {
        int p_base, q_base;
        tile_coords(k, p_base, q_base);
        const bool ptail = p_base + BP > M;
        const bool full = !ptail && q_base + BQ <= cout;
        f32x4 tot[MQ][MP];
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) tot[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- one K-step: wait for its stage, barrier, request the stage three steps ahead, 12 fragment reads, 16 MFMAs
        int u = 0;
        const unsigned char* As;
        auto kstep_head = [&]() {
            // stage of this step landed (this wave's DMAs), then everybody's; the barrier also says every wave
            // is done reading the previous stage, whose ring slot the next DMA overwrites.  The last step of the
            // first source also waits for the tile's side loads (parameters visible to all waves after its barrier).
            int c = 8 + x1 + x2 + x3;
            if (k > 0 && u == ksteps - 1) {
                const int cb = e1_bound + 4 * u;
                c = c < cb ? c : cb;
            }
            wait_vm(c);
            if (DGG_ABL != 6) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            x3 = x2;
            x2 = x1;
            x1 = 0;
            ++u;
            As = ring + ((c_slot + NS - DEPTH) & (NS - 1)) * STAGE;
            issue();
        };
        auto kstep_mma = [&](f32x4 (&acc)[MQ][MP], const bool tail) {
            const unsigned char* Ws = As + BP * 128;
#pragma unroll
            for (int ks = 0; ks < (DGG_ABL == 1 ? 0 : 2); ++ks) {
                const int fo = ks ? frag_off1 : frag_off0;
                bf16x8 af[MP], wf[MQ];
#pragma unroll
                for (int j = 0; j < MP; ++j)
                    af[j] = *reinterpret_cast<const bf16x8*>(As + (wp * TP + j * 16) * 128 + fo);
#pragma unroll
                for (int i = 0; i < MQ; ++i)
                    wf[i] = *reinterpret_cast<const bf16x8*>(Ws + (wq * TQ + i * 16) * 128 + fo);
                if (tail && ks * 32 + fc * 8 >= rem) {
                    // this lane's 8 channels of the pixel operand lie past the last channel
#pragma unroll
                    for (int j = 0; j < MP; ++j) af[j] = bf16x8{};
                }
#pragma unroll
                for (int i = 0; i < MQ; ++i)
#pragma unroll
                    for (int j = 0; j < MP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
            }
        };
        auto kstep = [&](f32x4 (&acc)[MQ][MP], const bool tail) {
            kstep_head();
            kstep_mma(acc, tail);
        };
        auto zero = [&](f32x4 (&acc)[MQ][MP]) {
#pragma unroll
            for (int i = 0; i < MQ; ++i)
#pragma unroll
                for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // the reference loads have landed (waited for in the first source's last step); from here on their registers
        // are ordinary values
        auto tie_ref = [&]() {
            asm volatile("" : "+v"(rfp[0][0]), "+v"(rfp[0][1]), "+v"(rfp[1][0]), "+v"(rfp[1][1]),
                              "+v"(rfp[2][0]), "+v"(rfp[2][1]), "+v"(rfp[3][0]), "+v"(rfp[3][1]));
            unsigned char* st = reinterpret_cast<unsigned char*>(stage);      // 16 rows x 32 bf16, 72-byte rows
#pragma unroll
            for (int j = 0; j < MP; ++j) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int rp = 0; rp < NRP; ++rp) *reinterpret_cast<u32x2*>(st + (rp * RPP + er) * 72 + ec * 8) = rfp[j][rp];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int i = 0; i < MQ; ++i) {
                    const u32x2 rr = *reinterpret_cast<const u32x2*>(st + fr * 72 + i * 32 + fc * 8);
                    rfm[i][j] = f32x4{bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        // a finished product leaves the MFMA accumulators; rows past the last pixel were computed from a repeated row
        // and must not reach the sums
        auto take = [&](f32x4 (&dst)[MQ][MP], const f32x4 (&acc)[MQ][MP]) {
            if (ptail) {
#pragma unroll
                for (int j = 0; j < MP; ++j) {
                    const bool pok = p_base + wp * TP + j * 16 + fr < M;
#pragma unroll
                    for (int i = 0; i < MQ; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[i][j][e] = pok ? acc[i][j][e] : 0.f;
                }
            } else {
#pragma unroll
                for (int i = 0; i < MQ; ++i)
#pragma unroll
                    for (int j = 0; j < MP; ++j) dst[i][j] = acc[i][j];
            }
        };
        // ---- a source's product: ReLU mask, BatchNorm-backward sums, scaled add into the total (fragment row i).
        //      Channels past cout carry finite garbage through their own columns only; nothing of them is stored.
        // The sums are sum(v) and sum(v * xhat), xhat = (ref - mean) * rstd: the second one is accumulated against the
        // raw reference value and corrected per channel afterwards, rstd * (sum(v * ref) - mean * sum(v)).
        auto epi_main = [&](const int i, const int sp, const f32x4 (&acc)[MQ][MP], f32x4 (&s1)[MQ], f32x4 (&s2)[MQ]) {
            const int cl = wq * TQ + i * 16 + fc * 4;
            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(par + sp * BQ + cl);
            const f32x4 sh4 = *reinterpret_cast<const f32x4*>(par + (4 + sp) * BQ + cl);
            s1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            s2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (DGG_ABL == 2) {
#pragma unroll
                for (int j = 0; j < MP; ++j) tot[i][j] += acc[i][j];
                return;
            }
            // ReLU mask as a number: clamp(2^64 * (ref * scale + shift)) is 1 for every positive pre-activation of at least
            // 2^-64, 0 for zero, negatives and NaN -- one packed fma with the clamp modifier per two elements instead of a
            // compare and a select per element (the power of two scales exactly, the sign is that of the unscaled value).
            constexpr float BIG = 18446744073709551616.f;
            const f32x2_t scb[2] = {{sc4[0] * BIG, sc4[1] * BIG}, {sc4[2] * BIG, sc4[3] * BIG}};
            const f32x2_t shb[2] = {{sh4[0] * BIG, sh4[1] * BIG}, {sh4[2] * BIG, sh4[3] * BIG}};
#pragma unroll
            for (int j = 0; j < MP; ++j) {
                const f32x2_t rf[2] = {{rfm[i][j][0], rfm[i][j][1]}, {rfm[i][j][2], rfm[i][j][3]}};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2_t m;
                    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(m) : "v"(rf[h]), "v"(scb[h]), "v"(shb[h]));
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const int e = h * 2 + e2;
                        const float v = acc[i][j][e] * m[e2];
                        s1[i][e] += v;
                        s2[i][e] = fmaf(v, rf[h][e2], s2[i][e]);
                        tot[i][j][e] = fmaf(v, sc4[e], tot[i][j][e]);
                    }
                }
            }
        };
        auto epi_reduce = [&](const int sp, f32x4 (&s1)[MQ], f32x4 (&s2)[MQ]) {
            if (DGG_ABL == 2) return;
            // 16 partial sums per lane, V[t] with t = fragment * 8 + which sum * 4 + channel, to be added over the 16 lanes
            // (pixels) of a DPP row.  Reduce-scatter instead of 16 full butterflies: every exchange step halves the values a
            // lane still carries (16 + 8 + 6 + 3 vector instructions instead of 64 and a 15-select tree), and lane fr ends
            // up with the row total of value number fr.
            float x[8], w[8];
#pragma unroll
            for (int i = 0; i < MQ; ++i) {
                const int cl = wq * TQ + i * 16 + fc * 4;
                const f32x4 mu4 = *reinterpret_cast<const f32x4*>(par + 8 * BQ + cl);
                const f32x4 rs4 = *reinterpret_cast<const f32x4*>(par + 9 * BQ + cl);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c2 = rs4[e] * fmaf(-mu4[e], s1[i][e], s2[i][e]);
                    if (i == 0) x[e] = s1[i][e], x[4 + e] = c2;
                    else w[e] = s1[i][e], w[4 + e] = c2;
                }
            }
            // lanes fr and fr ^ 8: lanes 0..7 of a row keep fragment 0 (x), lanes 8..15 (banks 2, 3) fragment 1 (w).
            // (s_nop: a DPP operand must not be read within two cycles of the instruction that wrote it; the compiler does
            // not see DPP reads inside inline assembly)
#define DGG_RS_A(n) "v_add_f32_dpp %" #n ", %" #n ", %" #n " row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
#define DGG_RS_A2(n, m) "v_add_f32_dpp %" #n ", %" #m ", %" #m " row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            asm volatile("s_nop 1\n\t"
                         DGG_RS_A(0) DGG_RS_A(1) DGG_RS_A(2) DGG_RS_A(3) DGG_RS_A(4) DGG_RS_A(5) DGG_RS_A(6) DGG_RS_A(7)
                         DGG_RS_A2(0, 8) DGG_RS_A2(1, 9) DGG_RS_A2(2, 10) DGG_RS_A2(3, 11) DGG_RS_A2(4, 12) DGG_RS_A2(5, 13) DGG_RS_A2(6, 14) DGG_RS_A2(7, 15)
                         "s_nop 1"
                         : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
                         : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]));
#undef DGG_RS_A
#undef DGG_RS_A2
            // lanes fr and fr ^ 4: banks 0 and 2 keep the first sum (x[0..3], partner four lanes up), banks 1 and 3 the
            // second (x[4..7], partner four lanes down)
            float y[4];
#define DGG_RS_B(n, lo, hi)                                                                        \
    "v_add_f32_dpp %" #n ", %" #lo ", %" #lo " row_shl:4 row_mask:0xf bank_mask:0x5\n\t"          \
    "v_add_f32_dpp %" #n ", %" #hi ", %" #hi " row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            asm volatile(DGG_RS_B(0, 4, 8) DGG_RS_B(1, 5, 9) DGG_RS_B(2, 6, 10) DGG_RS_B(3, 7, 11) "s_nop 1"
                         : "=&v"(y[0]), "=&v"(y[1]), "=&v"(y[2]), "=&v"(y[3])
                         : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
#undef DGG_RS_B
            // lanes fr ^ 2, then fr ^ 1, inside the quad: keep one half, send the other
            float z[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const float keep = b1 ? y[2 + m] : y[m], send = b1 ? y[m] : y[2 + m];
                z[m] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xF, 0xF, true));
            }
            const float keep = b0 ? z[1] : z[0], send = b0 ? z[0] : z[1];
            const float total = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));
            // one plain store per wave and source into the wave's own slots of the table (nobody else writes them; LDS float
            // atomics cost as much per lane as sixteen four-lane instructions did)
            sred[(wp * 4 + sp) * 2 * BQ + sred_lane] = total;
        };

        f32x4 acc[MQ][MP], prod[MQ][MP];
        f32x4 s1[MQ], s2[MQ];
        if constexpr (KS > 0) {
            // Overlapped form (KS K-steps per source, known at compile time): the product of source s-1 is masked, summed and
            // added while the MFMAs of source s run -- the same wave issues both, so its vector ALU work hides behind its
            // matrix work; every wave of the block being in the same phase, nothing else would.
            zero(acc);
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) kstep(acc, false);
            tie_ref();
            for (int s = 1; s < nsrc; ++s) {
                take(prod, acc);
                zero(acc);
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    // The step's MFMAs are issued first; the matrix pipe works through them while the same wave masks, sums
                    // and adds the previous source's product.  (Measured: MFMAs first on every wave 2345 us; the two waves of
                    // a SIMD in opposite orders 2497-2600; the epilogue after the source's last step, no overlap, 2439.)
                    kstep_head();
                    kstep_mma(acc, false);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kk == 0) epi_main(0, s - 1, prod, s1, s2);
                    if (kk == 1) epi_main(1, s - 1, prod, s1, s2);
                    if (kk == KS - 1) epi_reduce(s - 1, s1, s2);
                }
            }
            take(prod, acc);
            if (ACCG) {
                if (DGG_ABL != 4) issue_gold(p_base, q_base);
                x1 += N_G;
            }
            epi_main(0, nsrc - 1, prod, s1, s2);
            epi_main(1, nsrc - 1, prod, s1, s2);
            epi_reduce(nsrc - 1, s1, s2);
        } else {
            for (int s = 0; s < nsrc; ++s) {
                zero(acc);
                for (int kk = 0; kk < ksteps; ++kk) kstep(acc, has_tail && kk == ksteps - 1);
                if (s == 0) tie_ref();
                take(prod, acc);
                if (ACCG && s == nsrc - 1) {
                    issue_gold(p_base, q_base);
                    x1 += N_G;
                }
                epi_main(0, s, prod, s1, s2);
                epi_main(1, s, prod, s1, s2);
                epi_reduce(s, s1, s2);
            }
        }

        // ---- tile end: all waves are done with the parameter rows and the partial sums are complete
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // two sources per pass, one per half block: the source index is wave-uniform, so its pointers are scalar loads
        for (int s0 = 0; s0 < nsrc; s0 += 2) {
            const int s = s0 + (wave >> 2);
            if (s < nsrc) {
                const int c = s * 2 * BQ + (tid & 255);
                const int half = (tid >> 7) & 1, cc = tid & (BQ - 1);
                float* sa = a.sa[s];
                const int sl = a.sl[s];
                if (sa && q_base + cc < cout) atomicAdd(sa + (size_t)rep * 2 * sl + half * sl + q_base + cc, sred[c] + sred[4 * 2 * BQ + c]);
            }
        }
        const bool has_next = k + 1 < ntile;
        int pb_n = 0, qb_n = 0;
        if (has_next) {
            tile_coords(k + 1, pb_n, qb_n);
            if (DGG_ABL != 4) {
                issue_par(qb_n);
                issue_ref(pb_n, qb_n);
            }
        }

        // ---- slab read-modify-write through the wave-private patch, 16 pixels at a time.  The old values were
        //      requested before the last source's sums; only the next tile's side loads are younger.
        if (ACCG) {
            wait_vm(has_next ? N_PAR + N_REF : 0);
            asm volatile("" : "+v"(gold[0][0]), "+v"(gold[0][1]), "+v"(gold[1][0]), "+v"(gold[1][1]),
                              "+v"(gold[2][0]), "+v"(gold[2][1]), "+v"(gold[3][0]), "+v"(gold[3][1]));
        }
        unsigned char* gbase = reinterpret_cast<unsigned char*>(a.g) + (int64_t)p_base * a.ldg * GB;
#pragma unroll
        for (int j = 0; j < MP; ++j) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MQ; ++i) *reinterpret_cast<f32x4*>(stage + fr * ROWF + i * 16 + fc * 4) = tot[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(stage + (rp * RPP + er) * ROWF + ec * 4);
                f32x4 t = a4;
                if (ACCG) {
                    if constexpr (GBF16) {
                        const u32x2 gb = gold[j][rp];
                        t += f32x4{bf_lo(gb.x), bf_hi(gb.x), bf_lo(gb.y), bf_hi(gb.y)};
                    } else {
                        t += gold[j][rp];
                    }
                }
                const uint32_t off = slab_off(p_base, q_base, j, rp);
                if (DGG_ABL == 4 && a4[0] != 12345.678f) continue;
                if constexpr (GBF16) {
                    u32x2 o;
                    o.x = pack_bf2(t[0], t[1]);
                    o.y = pack_bf2(t[2], t[3]);
                    if (full) {
                        asm volatile("global_store_dwordx2 %0, %1, %2" DGG_NTS : : "v"(off), "v"(o), "s"(gbase) : "memory");
                    } else if (p_base + wp * TP + j * 16 + rp * RPP + er < M && q_base + wq * TQ + ec * 4 < cout) {
                        *reinterpret_cast<u32x2*>(gbase + off) = o;
                    }
                } else {
                    if (full) {
                        asm volatile("global_store_dwordx4 %0, %1, %2" DGG_NTS : : "v"(off), "v"(t), "s"(gbase) : "memory");
                    } else if (p_base + wp * TP + j * 16 + rp * RPP + er < M && q_base + wq * TQ + ec * 4 < cout) {
                        *reinterpret_cast<f32x4*>(gbase + off) = t;
                    }
                }
            }
        }
        // what this tile end certainly issued after the last step's DMA, and how much of it is younger than the side loads
        e1_bound = full ? N_G : 0;
        x1 += has_next ? N_PAR + N_REF + e1_bound : 0;
        if (DGG_ABL == 4) x1 = 0, e1_bound = 0;
    }
    // nothing of this block may land in LDS after it has gone
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace

// Form check + launch.  Returns -1 when the descriptor is not of the form this kernel handles (the caller
// then uses the general convolution kernel), 0 on success, a positive error code otherwise.
int dsaspp_dgrad_group_launch(const dsaspp_conv_desc& d, hipStream_t s) {
    const bool form = d.g && d.taps == 1 && d.epi_relu == 1 && d.ref && d.e_scale && d.e_shift && d.e_mean && d.e_rstd && !d.in_scale &&
                      !d.in_mask && !d.in_relu && !d.bias && !d.out_mask && d.cin % 8 == 0 && d.cin >= BK && d.cin <= 4096 && d.ldx % 8 == 0 && d.cout >= 4 &&
                      d.cout % 4 == 0 && d.n_src <= 4;
    if (!form) return -1;
    const int64_t M64 = (int64_t)d.n_img * d.H * d.W;
    // operand offsets are 32-bit byte offsets from the tensor base; slab offsets are relative to the tile's first row
    if (M64 * d.ldx * 2 >= (1ll << 32) || (int64_t)d.cout * d.cin * 2 >= (1ll << 32) || M64 >= (1ll << 31) ||
        (int64_t)BP * d.ldg * 4 >= (1ll << 32) || (int64_t)BP * d.ldref * 2 >= (1ll << 32))
        return -1;
    GArgs a;
    const int nsrc = d.n_src > 1 ? d.n_src : 1;
    const void* xs[4] = {d.x, d.x1, d.x2, d.x3};
    const void* ws[4] = {d.w, d.w1, d.w2, d.w3};
    const float* ess[4] = {d.e_scale, d.es1, d.es2, d.es3};
    const float* ehs[4] = {d.e_shift, d.eh1, d.eh2, d.eh3};
    float* sas[4] = {d.s_acc, d.sa1, d.sa2, d.sa3};
    const int sls[4] = {d.s_ld, d.sl1, d.sl2, d.sl3};
    for (int i = 0; i < 4; ++i) {
        const int k = i < nsrc ? i : 0;
        DS_REQUIRE(xs[k] && ws[k] && ess[k] && ehs[k], "conv_fwd(grouped data gradient): source %d incomplete", k);
        a.x[i] = reinterpret_cast<const u16*>(xs[k]);
        a.w[i] = reinterpret_cast<const u16*>(ws[k]);
        a.es[i] = ess[k];
        a.eh[i] = ehs[k];
        a.sa[i] = i < nsrc ? sas[k] : nullptr;
        a.sl[i] = sls[k] > 0 ? sls[k] : d.cout;
    }
    a.mean = d.e_mean;
    a.rstd = d.e_rstd;
    a.ref = reinterpret_cast<const u16*>(d.ref);
    a.g = d.g;
    a.ldx = d.ldx;
    a.ldref = d.ldref;
    a.ldg = d.ldg;
    a.cin = d.cin;
    a.cout = d.cout;
    a.nsrc = nsrc;
    a.M = (int)M64;
    a.nP = (a.M + BP - 1) / BP;
    a.nQ = (d.cout + BQ - 1) / BQ;
    const int n_cu = dsaspp_persistent_cus();
    {
        const void* kernels[] = {
            (const void*)dgrad_group_kernel<false, false, 0>, (const void*)dgrad_group_kernel<false, true, 0>, (const void*)dgrad_group_kernel<true, false, 0>,
            (const void*)dgrad_group_kernel<true, true, 0>,   (const void*)dgrad_group_kernel<false, false, 2>, (const void*)dgrad_group_kernel<false, true, 2>,
            (const void*)dgrad_group_kernel<true, false, 2>,  (const void*)dgrad_group_kernel<true, true, 2>,   (const void*)dgrad_group_kernel<false, false, 3>,
            (const void*)dgrad_group_kernel<false, true, 3>,  (const void*)dgrad_group_kernel<true, false, 3>,  (const void*)dgrad_group_kernel<true, true, 3>};
        for (const void* f : kernels) dsaspp_lds_limit(f, LDS_BYTES);
    }
    int64_t tiles = (int64_t)a.nP * a.nQ;
    int grid = (int)((tiles + 7) / 8 * 8);
    int cap = n_cu / 8 * 8;
    if (const char* e = getenv("DSASPP_DGG_GRID")) {      // test hook: fewer blocks, so that small cases walk several tiles per block
        const int v = atoi(e);
        if (v >= 8) cap = v / 8 * 8;
    }
    if (grid > cap) grid = cap;
    if (grid < 8) grid = 8;
    // K-steps per source: 2 and 3 (bottleneck widths 128 and 192) have the overlapped form
    const int ks = d.cin % BK == 0 ? d.cin / BK : 0;
    const bool no_overlap = getenv("DSASPP_DGG_NO_OVERLAP") != nullptr;      // A/B measurement hook
    void (*kern)(const GArgs);
#define DGG_PICK(KS_)                                                                                              \
    kern = d.g_accumulate ? (d.g_bf16 ? dgrad_group_kernel<true, true, KS_> : dgrad_group_kernel<true, false, KS_>) \
                          : (d.g_bf16 ? dgrad_group_kernel<false, true, KS_> : dgrad_group_kernel<false, false, KS_>)
    if (ks == 3 && !no_overlap) DGG_PICK(3);
    else if (ks == 2 && !no_overlap) DGG_PICK(2);
    else DGG_PICK(0);
#undef DGG_PICK
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS_BYTES, s, a);
    DS_LAUNCH_CHECK("conv_fwd(grouped data gradient)");
    return 0;
}
