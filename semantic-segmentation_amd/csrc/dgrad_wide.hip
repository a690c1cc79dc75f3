// Single-source 1x1 data gradient into the gradient slab with a long K (transitions, ASPP 1x1 layers), gfx950:
//
//   g[p][c] (+)= e_scale[c] * [ref[p][c] * e_scale[c] + e_shift[c] > 0] * (x[p][:] . w[c][:])
//
// with the consumer's BatchNorm-backward sums as a by-product -- what dgrad_group.hip computes for one source, on a
// 256 x 256 block tile.  The DMA ring that feeds a 128 x 128 x 64 step (32 KiB per stage, twelve fragment reads per
// wave) saturates at about 9 TB/s of LDS fill beside those reads, which caps that tile near 580 TFLOP/s; a
// 256 x 256 x 32 step moves the same 32 KiB and reads the same twelve fragments per wave for twice the MFMAs.
//   * 8 waves as 2 (pixel halves) x 4 (channel quarters): a wave owns 128 pixels x 64 channels = 8 x 4 accumulator
//     tiles of mfma_f32_16x16x32_bf16 (128 registers);
//   * persistent blocks, one per CU, walking (pixel tile, channel tile) pairs XCD by XCD; operands go global -> LDS by
//     DMA (global_load_lds, 16 B per lane) into a ring of 3 stages, 2 in flight, one counted wait and one LDS-only
//     barrier per step; stage rows are 64 bytes (32 channels) with an XOR chunk swizzle that lives in the source
//     address, the LDS image is lane-linear;
//   * the epilogue takes 16 pixels of the wave tile at a time through a wave-private LDS patch into row layout
//     (16 lanes cover the 128 bytes of a pixel's 64 channels), where the reference values and the old slab values are
//     loaded one group ahead, the product is masked, summed, scaled and stored.
// The K-loop's DMAs are issued from inline assembly and waited for by a counted s_waitcnt; everything else is left to
// the compiler, whose own wait counts do not know about the DMAs in the queue and therefore only wait longer.
//
// T9: the dilated 3x3 data gradient with many output channels (the ASPP 3x3 layers, 128 -> 512, dilation 3 .. 24),
//
//   y[p][c] = [ref[p][c] * e_scale[c] + e_shift[c] > 0] * sum_t x[p + off_t][:] . w[c][t][:]        (x = 0 outside the image)
//
// as the same GEMM with K = 9 * cin (taps side by side, the order the transposed filter is stored in): only the source
// address of the pixel rows changes -- a lane fetches its row shifted by the tap its 8-channel chunk belongs to, or, where
// that falls outside the image, a row of zeros (64-bit per-lane address).  The product is stored unscaled and
// overwrites y (the form the generic kernel writes for this layer).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BP = 256, BQ = 256, BK = 32;
constexpr int NW = 8, NT = 512, WP = 2, WQ = 4;
constexpr int TP = BP / WP, TQ = BQ / WQ, MP = TP / 16, MQ = TQ / 16;    // 128 x 64 wave tile: 8 x 4 MFMA tiles
constexpr int STAGE = (BP + BQ) * BK * 2;                                 // 32 KiB
constexpr int NS = 3, DEPTH = 2;                                          // ring stages, stages in flight
constexpr int ROWF = TQ + 4;                                              // patch row in floats (bank spread)
constexpr int PATCH_BYTES = NW * 16 * ROWF * 4;
constexpr int PAR_BYTES = 4 * BQ * 4;                                     // e_scale, e_shift, e_mean, e_rstd of the channel tile
constexpr int SRED_BYTES = WP * 2 * BQ * 4;                               // [pixel half][which sum][channel]
constexpr int LDS_BYTES = NS * STAGE + PATCH_BYTES + PAR_BYTES + SRED_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int NGRP = MP;                                                  // 16-pixel groups of the wave tile
constexpr int NRP = 4;                                                    // 4-pixel passes per group (64 lanes = 4 pixels x 16 channel quads)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct WArgs {
    const u16* x;
    const u16* w;
    const float* es;
    const float* eh;
    const float* mean;
    const float* rstd;
    float* sa;
    const u16* ref;
    u16* g;
    int ldx, ldref, ldg, cin, cout, sld, M, nP, nQ, accumulate;      // cin: K of the GEMM (9 * cin_real in the 3x3 form)
    // 3x3 form
    const unsigned char* zero;                       // >= 16 bytes of zeros
    int H, W, dil, cin_real;
};

// zeros for the pixel-row DMA of the 3x3 form (a static device array is zero-initialised; one copy per GPU)
__device__ __attribute__((aligned(256))) unsigned int g_zero_row_dw[64];

// chunk swizzle of a 64-byte stage row: position of K chunk c in row r is c ^ swz(r).  With it the sixteen lanes of every
// ds_read_b128 lane group (eight fragment rows at chunk c, the other eight rows of the 16-row tile at chunk c + 1) fall
// on sixteen different 16-byte slots of the 256-byte bank window
__device__ __forceinline__ int swz(int row) { return (0 - (row >> 2)) & 3; }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// T9: 0 = 1x1 form; 1 = 3x3 form with 32 | cin (a K-step lies inside one tap: the tap is a scalar); 2 = 3x3 form, any cin >= 32
template <bool ACCG, int T9>
__global__ __launch_bounds__(NT, 2) void dgrad_wide_kernel(const WArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* patch = reinterpret_cast<float*>(smem + NS * STAGE);
    float* par = reinterpret_cast<float*>(smem + NS * STAGE + PATCH_BYTES);
    float* sred = reinterpret_cast<float*>(smem + NS * STAGE + PATCH_BYTES + PAR_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cin = a.cin, cout = a.cout, M = a.M, nQ = a.nQ;
    const int ksteps = (cin + BK - 1) / BK;

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

    // ---- operand DMA cursor (runs DEPTH steps ahead of the arithmetic).  Instruction i of a wave fills stage rows
    //      (i*8 + wave)*16 .. +15: i = 0, 1 are pixel rows, i = 2, 3 filter rows; a lane brings the chunk that belongs at
    //      its lane-linear LDS position
    const uint32_t ring_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)ring);
    int c_k = 0, c_kk = 0, c_slot = 0;
    uint32_t r_off[4];                                // byte offset of this lane's row: pixel rows from the tile's first row, filter rows in w
    const u16* x_tile = a.x;                          // first pixel row of the cursor's tile (uniform; the operand may be larger than 4 GiB)
    int phw[2];                                       // 3x3 form: image coordinates (h << 16 | w) of this lane's two pixel rows; rows past the end: far outside
    const int lrow = lane >> 2;
    auto cursor_tile = [&]() {
        int pb, qb;
        tile_coords(c_k, pb, qb);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = ((i & 1) * 8 + wave) * 16 + lrow;      // row inside the pixel half (i < 2) or the filter half of the stage
            if (i < 2) {
                int p = pb + row;
                if (T9) {
                    const int rem = (p < M ? p : 0) % (a.H * a.W);
                    const int h = rem / a.W;
                    phw[i] = p < M ? (h << 16) | (rem - h * a.W) : 0x7fff7fff;
                }
                p = p < M ? p : M - 1;
                r_off[i] = (uint32_t)(p - pb) * (uint32_t)a.ldx * 2u;
            } else {
                int q = qb + row;
                q = q < cout ? q : cout - 1;
                r_off[i] = (uint32_t)q * (uint32_t)cin * 2u;
            }
        }
        x_tile = a.x + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane(pb) * a.ldx;
    };
    const int lchunk = (lane & 3) ^ swz(lrow);        // the K chunk this lane fetches ((i*8 + wave)*16 is a multiple of 16: swz depends on lrow only)
    auto issue = [&]() {
        const uint32_t slot_l = ring_l + c_slot * STAGE;
        // a chunk past the last channel: fetch the row's first chunk instead (finite values of the same tensor); the step's
        // MFMAs see zeros in the pixel operand there
        const uint32_t kb = (c_kk * BK + lchunk * 8 < cin) ? (uint32_t)(c_kk * BK + lchunk * 8) * 2u : 0u;
        // 3x3 form: the tap of this lane's chunk and its channel inside the tap.  A K-step of 32 channels meets at most two
        // taps (cin >= 32): the step's first tap is a scalar, a chunk past its end belongs to the next one; a chunk past
        // the last tap (K tail of the last step) reads zeros
        int dy = 0, dx = 0;
        uint32_t kx_b = 0;
        bool tap_ok = true;
        if (T9) {
            const int t0 = (c_kk * BK) / a.cin_real;
            int kin = c_kk * BK - t0 * a.cin_real + lchunk * 8;
            int t = t0;
            if (T9 == 2 && kin >= a.cin_real) {
                kin -= a.cin_real;
                ++t;
            }
            tap_ok = t < 9;
            const int ky = t / 3;
            dy = (ky - 1) * a.dil;
            dx = (t - ky * 3 - 1) * a.dil;
            kx_b = (uint32_t)kin * 2u;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (T9 && i < 2) {
                const bool ok = tap_ok && (unsigned)((phw[i] >> 16) + dy) < (unsigned)a.H && (unsigned)((phw[i] & 0xffff) + dx) < (unsigned)a.W;
                // (the row's own byte offset is the one of the 1x1 form; the tap's shift in bytes fits 32 bits: launcher)
                const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(x_tile) + r_off[i] + (int64_t)((dy * a.W + dx) * (a.ldx * 2)) + kx_b : a.zero;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             : : "s"(slot_l + (i * 8 + wave) * 1024), "v"(src) : "memory");
                continue;
            }
            const uint32_t off = r_off[i] + kb;
            const u16* base = i < 2 ? x_tile : a.w;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         : : "s"(slot_l + (i * 8 + wave) * 1024), "v"(off), "s"(base) : "memory");
        }
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

    // ---- fragment coordinates
    const int fr = lane & 15, g = lane >> 4;
    const int wp = wave % WP, wq = wave / WP;
    const int frag_off = fr * 64 + ((g ^ swz(fr)) << 4);       // tile rows start at multiples of 16: swz depends on fr only
    const int rep = blockIdx.x % DS_R;

    cursor_tile();
    for (int i = 0; i < DEPTH; ++i) issue();
    int r_slot = 0;

    for (int k = 0; k < ntile; ++k) {
        int p_base, q_base;
        tile_coords(k, p_base, q_base);
        // per-channel parameters of this channel tile (the previous tile's readers have passed its last barrier)
        for (int c = tid; c < 4 * BQ; c += NT) {
            const int row = c / BQ, cc = c - row * BQ;
            int q = q_base + cc;
            q = q < cout ? q : cout - 1;
            const float* src = row == 0 ? a.es : row == 1 ? a.eh : row == 2 ? a.mean : a.rstd;
            par[c] = src[q];
        }
        f32x4 acc[MQ][MP];
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        auto kstep = [&](const int kk, const bool last) {
            // this wave's DMAs of the step's stage have landed, then everybody's; the barrier also says every wave is done
            // reading the previous step's stage, whose ring slot the next DMA overwrites
            wait_vm<4 * (DEPTH - 1)>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue();
            const unsigned char* As = ring + r_slot * STAGE;
            const unsigned char* Ws = As + BP * 64;
            r_slot = r_slot + 1 < NS ? r_slot + 1 : 0;
            // the filter fragments once, the pixel fragments in two halves of four (registers)
            bf16x8 wf[MQ];
#pragma unroll
            for (int i = 0; i < MQ; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(Ws + (wq * TQ + i * 16) * 64 + frag_off);
            const bool ktail = last && kk * BK + g * 8 >= cin;     // this lane's 8 channels lie past the last channel
#pragma unroll
            for (int jh = 0; jh < MP; jh += 4) {
                bf16x8 af[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const bf16x8*>(As + (wp * TP + (jh + j) * 16) * 64 + frag_off);
                if (ktail) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) af[j] = bf16x8{};
                }
#pragma unroll
                for (int i = 0; i < MQ; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][jh + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][jh + j], 0, 0, 0);
            }
        };

        for (int kk = 0; kk < ksteps - 1; ++kk) kstep(kk, false);

        // ---- epilogue, 16 pixels of the wave tile at a time, in row layout (a pixel's 64 channels = 128 contiguous
        //      bytes across 16 lanes).  The slab traffic of a tile (reference values in, old gradients in, new gradients
        //      out: 384 KiB) is what the epilogue waits for, so the loads run far ahead: the first EARLY groups are
        //      requested before the last K-step, then as many as the registers freed by the K-step's fragments hold, and
        //      every finished group (16 accumulator registers free) requests one more.  Rows past M load the last row (a
        //      mapped address) and store nothing; lanes whose channel quad lies past cout use quad 0 and store nothing.
        // (the epilogue's lane coordinates are derived from an opaque copy of the lane id, so that they are computed here
        // and do not occupy registers across the K loop)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int er = lane_e >> 4, ec = lane_e & 15;        // row layout: pixel er of a 4-pixel pass, channel quad ec
        const int fr_e = lane_e & 15, g_e = lane_e >> 4;     // accumulator layout
        float* stage = patch + wave * (16 * ROWF);
        const int qw = q_base + wq * TQ + ec * 4;           // this lane's channel quad (cout is a multiple of 4)
        const bool qok = qw < cout;
        const int cl = wq * TQ + ec * 4;                     // ... inside the channel tile
        const int p_w = wp * TP + er;                        // this lane's pixel of group 0, pass 0, counted from the tile's first row
        const int m_rel = M - p_base;                        // rows of the tensor from the tile's first row on (> 0)
        // slab addresses are a uniform base plus a 32-bit byte offset (the launcher checks that they fit)
        const uint32_t qoff = (uint32_t)(qok ? qw : 0) * 2u, ldref2 = (uint32_t)a.ldref * 2u, ldg2 = (uint32_t)a.ldg * 2u;
        // (the tile's first row: slabs may be larger than 4 GiB)
        const uint32_t pb_s = __builtin_amdgcn_readfirstlane(p_base);        // (scalar arithmetic for the two bases)
        const unsigned char* refb = reinterpret_cast<const unsigned char*>(a.ref) + (size_t)pb_s * ldref2;
        unsigned char* gb = reinterpret_cast<unsigned char*>(a.g) + (size_t)pb_s * ldg2;
        // groups requested before the last K-step / in flight at most.  Accumulating (two loads per pass), three groups are
        // what fits beside the accumulators without a spill; with five, 17 registers went to scratch, every reload was
        // waited for with vmcnt(0) -- which also waits for the whole prefetch -- and the launches took 6-8 % longer
        constexpr int EARLY = 2, AHEAD = ACCG ? 3 : (T9 ? 6 : NGRP);      // (3x3 form: the lane's row coordinates live across the tile; all 8 groups in flight spilled 3 registers)
        u32x2 rfv[NGRP][NRP], gov[ACCG ? NGRP : 1][NRP];
        auto load_group = [&](const int j) {
            int pj = p_w + j * 16;
            asm volatile("" : "+v"(pj));                      // (keeps the 32 row addresses of the tile from being formed up front)
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                int p = pj + rp * 4;
                p = p < m_rel ? p : m_rel - 1;
                rfv[j][rp] = *reinterpret_cast<const u32x2*>(refb + (size_t)((uint32_t)p * ldref2 + qoff));
                if constexpr (ACCG) gov[j][rp] = *reinterpret_cast<const u32x2*>(gb + (size_t)((uint32_t)p * ldg2 + qoff));
            }
        };
#pragma unroll
        for (int j = 0; j < EARLY; ++j) load_group(j);
        asm volatile("" ::: "memory");
        kstep(ksteps - 1, true);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = EARLY; j < AHEAD; ++j) load_group(j);
        const f32x4 es = *reinterpret_cast<const f32x4*>(par + cl), eh = *reinterpret_cast<const f32x4*>(par + BQ + cl);
        float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NGRP; ++j) {
            // accumulator layout -> the wave's patch -> row layout.  The patch is private to the wave and LDS executes a
            // wave's accesses in order; only the compiler must not reorder them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MQ; ++i) *reinterpret_cast<f32x4*>(stage + fr_e * ROWF + i * 16 + g_e * 4) = acc[i][j];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (j + AHEAD < NGRP) load_group(j + AHEAD);
#pragma unroll
            for (int rp = 0; rp < NRP; ++rp) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(stage + (rp * 4 + er) * ROWF + ec * 4);
                int p = p_w + j * 16 + rp * 4;
                asm volatile("" : "+v"(p));
                const bool ok = qok && p < m_rel;
                const u32x2 rr = rfv[j][rp];
                const float rf[4] = {bf_lo(rr.x), bf_hi(rr.x), bf_lo(rr.y), bf_hi(rr.y)};
                float t[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pre = fmaf(rf[e], es[e], eh[e]);
                    const float v = (ok && pre > 0.f) ? a4[e] : 0.f;     // rows past M were computed from a repeated row
                    s1[e] += v;
                    s2[e] = fmaf(v, rf[e], s2[e]);
                    t[e] = T9 ? v : v * es[e];
                }
                // (the sums pass through an empty asm statement: the store below sits in a branch, and the compiler otherwise
                // sinks the 256 additions of the tile below the last branch, with their operands kept alive until then)
                asm volatile("" : "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3]), "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]));
                if constexpr (ACCG) {
                    const u32x2 go = gov[j][rp];
                    t[0] += bf_lo(go.x); t[1] += bf_hi(go.x); t[2] += bf_lo(go.y); t[3] += bf_hi(go.y);
                }
                if (ok) *reinterpret_cast<u32x2*>(gb + (size_t)((uint32_t)p * ldg2 + qoff)) = u32x2{pack_bf2(t[0], t[1]), pack_bf2(t[2], t[3])};
            }
        }
        // sums over the wave's 128 pixels: the four lanes that share a channel quad (er = 0..3), then one plain store per
        // wave into its own slots of the table.  The second sum was taken against the raw reference value.
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1[e] += __shfl_xor(s1[e], 16, 64);
            s1[e] += __shfl_xor(s1[e], 32, 64);
            s2[e] += __shfl_xor(s2[e], 16, 64);
            s2[e] += __shfl_xor(s2[e], 32, 64);
        }
        if (er == 0) {
            *reinterpret_cast<f32x4*>(sred + (wp * 2 + 0) * BQ + cl) = f32x4{s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<f32x4*>(sred + (wp * 2 + 1) * BQ + cl) = f32x4{s2[0], s2[1], s2[2], s2[3]};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (a.sa) {
            const int which = tid / BQ, cc = tid - which * BQ;       // 512 threads = 2 sums x 256 channels
            if (q_base + cc < cout) {
                float tsum = sred[which * BQ + cc] + sred[(2 + which) * BQ + cc];
                if (which) {
                    const float t1 = sred[cc] + sred[2 * BQ + cc];
                    tsum = par[3 * BQ + cc] * fmaf(-par[2 * BQ + cc], t1, tsum);      // rstd * (sum(v * ref) - mean * sum(v))
                }
                atomicAdd(a.sa + (size_t)rep * 2 * a.sld + which * a.sld + q_base + cc, tsum);
            }
        }
        // the table and the parameter rows are rewritten by the next tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    // nothing of this block may land in LDS after it has gone
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

}  // namespace

// Form check + launch.  Returns -1 when the descriptor is not of the form this kernel is for (the caller then uses the
// grouped kernel), 0 on success, a positive error code otherwise.
int dsaspp_dgrad_wide_launch(const dsaspp_conv_desc& d, hipStream_t s) {
    static const int enabled = [] { const char* e = getenv("DSASPP_DGRAD_WIDE"); return e ? atoi(e) : 1; }();
    if (!enabled) return -1;
    const bool common = d.epi_relu == 1 && d.ref && d.e_scale && d.e_shift && d.e_mean && d.e_rstd && !d.in_scale && !d.in_mask && !d.in_relu && !d.bias &&
                        !d.out_mask && d.n_src <= 1 && d.dtype == 0 && d.cin % 8 == 0 && d.ldx % 8 == 0 && d.cout % 4 == 0 && d.ldref % 4 == 0 && !d.gx_corr1;
    const bool form1 = common && d.g && d.g_bf16 && d.taps == 1 && d.ldg % 4 == 0;
    // the 3x3 form writes y (bf16, unscaled, overwritten): the ASPP 3x3 data gradient
    const bool t9 = common && !d.g && d.y && !d.y_f32 && d.taps == 9 && d.cin >= BK && d.ldy % 4 == 0 && d.dil >= 1 && !d.stat_acc;
    if (!form1 && !t9) return -1;
    static const int min_cin = [] { const char* e = getenv("DSASPP_DGRAD_WIDE_MIN_CIN"); return e ? atoi(e) : 256; }();
    static const int min_cout = [] { const char* e = getenv("DSASPP_DGRAD_WIDE_MIN_COUT"); return e ? atoi(e) : 512; }();
    const int kdim = t9 ? 9 * d.cin : d.cin;
    if (kdim < min_cin || d.cout < (t9 ? 448 : min_cout)) return -1;
    const int64_t M64 = (int64_t)d.n_img * d.H * d.W;
    void* outp = t9 ? d.y : d.g;
    const int ldo = t9 ? d.ldy : d.ldg;
    // pixel rows of every tensor are addressed from a uniform 64-bit tile base (slabs may exceed 4 GiB); inside a tile and
    // in the filter, offsets are 32-bit
    if ((int64_t)d.cout * kdim * 2 >= (1ll << 32) || M64 >= (1ll << 31) - BP || (int64_t)BP * d.ldx * 2 >= (1ll << 31) ||
        (int64_t)BP * d.ldref * 2 >= (1ll << 31) || (int64_t)BP * ldo * 2 >= (1ll << 31))
        return -1;
    if ((((uintptr_t)d.ref | (uintptr_t)outp) & 7) != 0) return -1;
    if (t9 && ((int64_t)d.dil * (d.W + 1) * d.ldx * 2 >= (1ll << 31) || d.H + d.dil >= 32000 || d.W + d.dil >= 32000)) return -1;
    // the row of zeros of the 3x3 form: one copy per GPU, its address looked up once per device and thread
    static thread_local int zero_dev = -1;
    static thread_local const unsigned char* zero_ptr = nullptr;
    if (t9) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return -1;
        if (dev != zero_dev) {
            void* zp = nullptr;
            if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_row_dw)) != hipSuccess || !zp) return -1;
            zero_ptr = static_cast<const unsigned char*>(zp);
            zero_dev = dev;
        }
    }
    WArgs a;
    a.x = reinterpret_cast<const u16*>(d.x);
    a.w = reinterpret_cast<const u16*>(d.w);
    a.es = d.e_scale;
    a.eh = d.e_shift;
    a.mean = d.e_mean;
    a.rstd = d.e_rstd;
    a.sa = d.s_acc;
    a.ref = reinterpret_cast<const u16*>(d.ref);
    a.g = reinterpret_cast<u16*>(outp);
    a.ldx = d.ldx;
    a.ldref = d.ldref;
    a.ldg = ldo;
    a.cin = kdim;
    a.zero = zero_ptr;
    a.H = d.H;
    a.W = d.W;
    a.dil = d.dil;
    a.cin_real = d.cin;
    a.cout = d.cout;
    a.sld = d.s_ld > 0 ? d.s_ld : d.cout;
    a.M = (int)M64;
    a.nP = (a.M + BP - 1) / BP;
    a.nQ = (d.cout + BQ - 1) / BQ;
    a.accumulate = t9 ? 0 : d.g_accumulate;
    const int n_cu = dsaspp_persistent_cus();
    dsaspp_lds_limit(reinterpret_cast<const void*>(dgrad_wide_kernel<true, 0>), LDS_BYTES);
    dsaspp_lds_limit(reinterpret_cast<const void*>(dgrad_wide_kernel<false, 0>), LDS_BYTES);
    dsaspp_lds_limit(reinterpret_cast<const void*>(dgrad_wide_kernel<false, 1>), LDS_BYTES);
    dsaspp_lds_limit(reinterpret_cast<const void*>(dgrad_wide_kernel<false, 2>), LDS_BYTES);
    const int64_t tiles = (int64_t)a.nP * a.nQ;
    int grid = (int)((tiles + 7) / 8 * 8);
    int cap = n_cu / 8 * 8;
    if (const char* e = getenv("DSASPP_DGG_GRID")) {      // test hook: fewer blocks, so that small cases walk several tiles per block
        const int v = atoi(e);
        if (v >= 8) cap = v / 8 * 8;
    }
    if (grid > cap) grid = cap;
    if (grid < 8) grid = 8;
    if (t9 && d.cin % BK == 0) hipLaunchKernelGGL((dgrad_wide_kernel<false, 1>), dim3(grid), dim3(NT), LDS_BYTES, s, a);
    else if (t9) hipLaunchKernelGGL((dgrad_wide_kernel<false, 2>), dim3(grid), dim3(NT), LDS_BYTES, s, a);
    else if (d.g_accumulate) hipLaunchKernelGGL((dgrad_wide_kernel<true, 0>), dim3(grid), dim3(NT), LDS_BYTES, s, a);
    else hipLaunchKernelGGL((dgrad_wide_kernel<false, 0>), dim3(grid), dim3(NT), LDS_BYTES, s, a);
    DS_LAUNCH_CHECK("conv_fwd(wide data gradient)");
    return 0;
}
