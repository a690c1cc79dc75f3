// BN-ReLU -> 3x3 dilated convolution of the dense layers (wide bottleneck in, growth-rate channels out),
// gfx950.  The generic implicit-GEMM kernel stages the shifted input tile once per filter tap, i.e. nine
// times; here a block owns an 8 x 32 pixel patch of one image and stages the patch plus its dilation halo
// ONCE per 32-channel chunk (BatchNorm scale/shift, ReLU and the zero padding applied on the way into
// LDS), then runs all nine taps out of LDS by shifting the fragment base address.  LDS rows are one pixel
// (32 channels = 64 B) padded to 96 B: a stride of 2 (mod 4) 16-byte slots makes any 16 consecutive pixels
// conflict-free for ds_read_b128 whatever the shift.  The chunk's filter slice (nine taps x few output channels,
// 64-byte rows, XOR-swizzled) sits beside it.
//
// Patch and filter slice take most of the LDS of a CU, so one block is resident per CU, and a block whose waves all
// load, then convert, then multiply, then store uses the memory path, the vector ALU, the LDS and the matrix pipe one
// after the other (measured: the four parts add up).  Therefore:
//   * the waves are specialised.  Waves 4..7 (one per SIMD) PRODUCE: they request the raw rows, apply the prologue and
//     write patch and filter slice into LDS.  Waves 0..3 CONSUME: fragment reads and MFMAs (two patch rows = 4 pixel
//     fragments per wave, so a filter fragment is used four times), output stores and statistics.  Patch (and, where it
//     fits, filter) are double-buffered: during step q the consumers multiply item q out of one buffer while the
//     producers stage item q+1 into the other; one block barrier per step, which waits for LDS traffic only;
//   * the unit of work is an item = (tile, 32-channel chunk); blocks are persistent (one per CU, each walks a strided
//     list of tiles) and the producers' requests run RD items ahead, across tile boundaries; the output stores of a tile
//     drain while the next tile is multiplied;
//   * past the last item the last one is requested and staged again: the number of loads in flight never depends on a
//     branch, and the wait counts the compiler derives stay exact.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

#ifndef C3H_ABL
#define C3H_ABL 0                // diagnostic builds (timing only, results wrong): 1 no fragment reads / MFMA, 3 no stores / statistics,
                                 // 4 no global loads, 5 no prologue arithmetic / LDS stores
#endif
#ifndef C3H_RD
#define C3H_RD 2                 // register sets of raw rows in flight per producer thread
#endif

namespace {

constexpr int TH = 8, TW = 32, ROWB = 96, WROWB = 64, KC = 32;

constexpr int NT = 512, NPROD = 256;                    // threads per block, producer threads
constexpr int LDS_MAX = 160 * 1024;

template <int MQ, int DIL>
struct HaloShape {
    static constexpr int HW2 = TW + 2 * DIL, HH2 = TH + 2 * DIL, NPIX = HW2 * HH2;
    static constexpr int WROWS = 9 * MQ * 16;
    static constexpr int PBYTES = NPIX * ROWB, FBYTES = WROWS * WROWB;
    static constexpr int TAIL = 2 * MQ * 16 * 4;                         // statistics area; the scale/shift table follows
    static constexpr bool FDB = 2 * PBYTES + 2 * FBYTES + TAIL + 2 * 256 * 4 <= LDS_MAX;   // filter double-buffered too
    static constexpr int LDS0 = 2 * PBYTES + (FDB ? 2 : 1) * FBYTES + TAIL;
};

template <int N>
using slot_t = std::integral_constant<int, N>;

// block barrier that waits for this wave's LDS operations only (__syncthreads() would also wait for the loads that are
// meant to stay in flight)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <int MQ, int DIL>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo_kernel(const dsaspp_conv_desc d, const int tiles_x, const int tiles_y,
                                                            const int ntiles) {
    using S = HaloShape<MQ, DIL>;
    constexpr int MP = 4;                               // per consumer wave: two patch rows = 4 half rows of 16 pixels
    constexpr int RD = C3H_RD;
    constexpr int HW2 = S::HW2, NPIX = S::NPIX, WROWS = S::WROWS;
    constexpr int PBYTES = S::PBYTES, FBYTES = S::FBYTES;
    constexpr bool FDB = S::FDB;
    static_assert(RD >= 1 && RD <= 3, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const flds = smem + 2 * PBYTES;
    float* red = reinterpret_cast<float*>(smem + 2 * PBYTES + (FDB ? 2 : 1) * FBYTES);      // [2][MQ*16] statistics of this block's tiles
    float* tab = red + 2 * MQ * 16;                                                           // [scale cin | shift cin]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout;
    const int per_img = tiles_x * tiles_y;
    const int nchunk = cin / KC;
    const int G = gridDim.x;                            // a multiple of 8 whenever a block has more than one tile
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int nitems = my_tiles * nchunk;
    const int nsteps = (nitems + RD - 1) / RD * RD;     // both roles run the same number of steps (= barriers)
    const bool want_stats = d.stat_acc != nullptr;

    for (int c = tid; c < cin; c += NT) {
        tab[c] = d.in_scale ? d.in_scale[c] : 1.f;
        tab[cin + c] = d.in_scale ? d.in_shift[c] : 0.f;
    }
    if (tid < 2 * MQ * 16) red[tid] = 0.f;
    __syncthreads();

    // tile k of this block: blocks of one XCD (blockIdx % 8) walk one contiguous range of tiles side by side
    auto tile_coords = [&](const int k, int& n, int& ty0, int& tx0) __attribute__((always_inline)) {
        const int t = xcd_remap((int)blockIdx.x + k * G, ntiles);
        n = t / per_img;
        const int rem = t - n * per_img;
        const int tyi = rem / tiles_x;
        ty0 = tyi * TH;
        tx0 = (rem - tyi * tiles_x) * TW;
    };

    if (wave >= 4) {
        // =========================================================================== producers
        constexpr int NLD = (NPIX * 4 + NPROD - 1) / NPROD;   // 16-byte pieces of the patch per thread and item
        constexpr int NWL = (WROWS * 4 + NPROD - 1) / NPROD;
        const int pt = tid - NPROD;
        const unsigned char* xg = reinterpret_cast<const unsigned char*>(d.x);
        const unsigned char* wg = reinterpret_cast<const unsigned char*>(d.w);
        const uint32_t relu_floor = d.in_relu ? 0u : 0x80008000u;
        // piece u of this thread is 16-byte column (pt & 3) of halo pixel / filter row (pt >> 2) + 64 u
        const int ck = pt & 3;
        int hyx[NLD];                                   // halo row / column of a piece, packed; they do not depend on the tile
        uint32_t wsrc_off[NWL];
        unsigned wokmask = 0;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int pix = (pt >> 2) + u * (NPROD / 4);
            const int hy = pix / HW2, hx = pix - hy * HW2;
            hyx[u] = (hy << 8) | hx;
        }
        const bool last_piece = (pt >> 2) + (NLD - 1) * (NPROD / 4) < NPIX;   // only the last piece can lie past the patch
#pragma unroll
        for (int u = 0; u < NWL; ++u) {
            const int r = (pt >> 2) + u * (NPROD / 4);
            const int tap = r / (MQ * 16), co = r - tap * (MQ * 16);
            const bool ok = r < WROWS && co < cout;
            wokmask |= ok ? (1u << u) : 0u;
            wsrc_off[u] = ok ? (uint32_t)(((co * 9 + tap) * cin + ck * 8) * 2) : 0u;
        }

        uint4 v[RD][NLD], wv[RD][NWL];
        unsigned okm[RD];
        // load cursor: the item requested next
        int ld_k = 0, ld_c = 0, ld_n, ld_ty0, ld_tx0;
        tile_coords(0, ld_n, ld_ty0, ld_tx0);
        auto issue = [&](auto slot_tag) __attribute__((always_inline)) {
            constexpr int P = decltype(slot_tag)::value;
            unsigned m = 0;
            const uint32_t cb = (uint32_t)(ld_c * KC * 2);
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int iy = ld_ty0 - DIL + (hyx[u] >> 8), ix = ld_tx0 - DIL + (hyx[u] & 255);
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && (u < NLD - 1 || last_piece);
                m |= ok ? (1u << u) : 0u;
                const uint32_t off = ok ? (uint32_t)(((ld_n * H + iy) * W + ix) * d.ldx + ck * 8) * 2u + cb : 0u;
                if (C3H_ABL == 4) v[P][u] = uint4{off, 0u, 0u, 0u};
                else v[P][u] = *reinterpret_cast<const uint4*>(xg + (size_t)off);
            }
#pragma unroll
            for (int u = 0; u < NWL; ++u) {
                if (C3H_ABL == 4) wv[P][u] = uint4{wsrc_off[u] + cb, 0u, 0u, 0u};
                else wv[P][u] = *reinterpret_cast<const uint4*>(wg + (size_t)(wsrc_off[u] + cb));
            }
            okm[P] = m;
            // advance; past the last item the cursor stays on it
            if (ld_c + 1 < nchunk) ++ld_c;
            else if (ld_k + 1 < my_tiles) {
                ld_c = 0;
                ++ld_k;
                tile_coords(ld_k, ld_n, ld_ty0, ld_tx0);
            }
        };
        // ---- BatchNorm + ReLU + zero padding on the way into LDS: the 32 channels of staging cursor st_c, into buffer `buf`
        int st_c = 0;
        auto stage_patch = [&](auto slot_tag, const int buf) __attribute__((always_inline)) {
            constexpr int P = decltype(slot_tag)::value;
            float sc[8], sh[8];
            load8f(tab + st_c * KC + ck * 8, sc);
            load8f(tab + cin + st_c * KC + ck * 8, sh);
            unsigned char* dst = smem + buf * PBYTES + ck * 16;
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                if (C3H_ABL == 5) { if (v[P][u].x == 0x12345u) dst[0] = 1; continue; }
                float f[8];
                unpack8(v[P][u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                uint4 o = relu_bf8(pack8(f), relu_floor);   // ReLU after the rounding (packed 16-bit max)
                const bool k = (okm[P] >> u) & 1u;          // zero padding comes after BatchNorm + ReLU
                o.x = k ? o.x : 0u; o.y = k ? o.y : 0u; o.z = k ? o.z : 0u; o.w = k ? o.w : 0u;
                const int pix = (pt >> 2) + u * (NPROD / 4);
                if (u < NLD - 1 || last_piece) *reinterpret_cast<uint4*>(dst + pix * ROWB) = o;
            }
            st_c = st_c + 1 < nchunk ? st_c + 1 : 0;
        };
        auto stage_filter = [&](auto slot_tag, const int buf) __attribute__((always_inline)) {
            constexpr int P = decltype(slot_tag)::value;
#pragma unroll
            for (int u = 0; u < NWL; ++u) {
                if (C3H_ABL == 5) { if (wv[P][u].x == 0x12345u) flds[0] = 1; continue; }
                uint4 o = wv[P][u];
                const bool k = (wokmask >> u) & 1u;
                o.x = k ? o.x : 0u; o.y = k ? o.y : 0u; o.z = k ? o.z : 0u; o.w = k ? o.w : 0u;
                const int r = (pt >> 2) + u * (NPROD / 4);
                if (r < WROWS) *reinterpret_cast<uint4*>(flds + buf * FBYTES + r * WROWB + ((ck ^ (((r >> 2) & 1) << 1)) << 4)) = o;
            }
        };
        // step q: item q+1 (register set `slot_tag`) goes into the buffer the consumers are not reading, the set is refilled
        // with item q+1+RD
        auto step = [&](auto slot_tag, const int q) __attribute__((always_inline)) {
            const int buf = (q & 1) ^ 1;
            stage_patch(slot_tag, buf);
            if (FDB) stage_filter(slot_tag, buf);
            else {
                lds_barrier();                          // the consumers are done with the one filter buffer
                stage_filter(slot_tag, 0);
            }
            issue(slot_tag);
            lds_barrier();
        };

        // in flight from here on, oldest first: items 0 .. RD-1 (item RD once item 0 is staged)
        issue(slot_t<0>{});
        if constexpr (RD > 1) issue(slot_t<1>{});
        if constexpr (RD > 2) issue(slot_t<2>{});
        stage_patch(slot_t<0>{}, 0);
        stage_filter(slot_t<0>{}, 0);
        issue(slot_t<0>{});
        lds_barrier();
        for (int q = 0; q < nsteps; q += RD) {
            step(slot_t<1 % RD>{}, q);
            if constexpr (RD > 1) step(slot_t<2 % RD>{}, q + 1);
            if constexpr (RD > 2) step(slot_t<0>{}, q + 2);
        }
    } else {
        // =========================================================================== consumers
        // fragment coordinates (bases at the top-left tap: every tap offset is a non-negative immediate)
        const int fr = lane & 15, fc = lane >> 4;
        int rbase[MP];
#pragma unroll
        for (int j = 0; j < MP; ++j) rbase[j] = ((2 * wave + (j >> 1)) * HW2 + (j & 1) * 16 + fr) * ROWB + fc * 16;
        // filter rows are 64 B with the two 32-byte halves exchanged in rows 4..7 (mod 8): a ds_read_b128 serves lanes
        // 0..7 and 16..23 together (eight rows x two adjacent 16-byte slots), and 64-byte rows laid end to end would put rows r
        // and r+4 on the same banks
        const int wbase = fr * WROWB + ((fc ^ (((fr >> 2) & 1) << 1)) << 4);
        f32x4 acc[MQ][MP];
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u16* yg = reinterpret_cast<u16*>(d.y);
        int cur_k = 0, cur_c = 0, cn, cty0, ctx0;
        tile_coords(0, cn, cty0, ctx0);

        // ---- nine taps of an item out of LDS buffer `buf`.  The fragments of a tap are read one tap ahead of the MFMAs that use
        //      them (explicit double buffer, order pinned with scheduling barriers): left to itself the compiler reads each
        //      group right before its MFMAs and the wave -- the only one of its SIMD that multiplies -- waits out an LDS round
        //      trip per group.
        auto multiply = [&](const int buf) __attribute__((always_inline)) {
            const unsigned char* pa = smem + buf * PBYTES;
            const unsigned char* pw = flds + (FDB ? buf : 0) * FBYTES + wbase;
            bf16x8 afb[2][MP], wfb[2][MQ];
            auto read_group = [&](const int tap) __attribute__((always_inline)) {
                const int tyy = tap / 3, txx = tap - tyy * 3;
                const int toff = (tyy * DIL * HW2 + txx * DIL) * ROWB;
#pragma unroll
                for (int j = 0; j < MP; ++j) afb[tap & 1][j] = *reinterpret_cast<const bf16x8*>(pa + rbase[j] + toff);
#pragma unroll
                for (int i = 0; i < MQ; ++i) wfb[tap & 1][i] = *reinterpret_cast<const bf16x8*>(pw + (tap * MQ + i) * 16 * WROWB);
            };
            if (C3H_ABL == 1) return;
            read_group(0);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                if (t + 1 < 9) read_group(t + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MQ; ++i)
#pragma unroll
                    for (int j = 0; j < MP; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfb[t & 1][i], afb[t & 1][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        // ---- a complete tile: bf16 NHWC store into the slab slice, statistics of the stored values.
        // Statistics are summed in LDS over all tiles of the block (an area of their own: no barrier) and leave at the end of
        // the block as a few full-width atomic instructions: float atomics execute at the memory side, an instruction
        // with 4 active lanes costs what a full one does.
        auto finish_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MQ; ++i) {
                const int co = i * 16 + fc * 4;
                float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < MP; ++j) {
                    const int oy = cty0 + 2 * wave + (j >> 1), ox = ctx0 + (j & 1) * 16 + fr;
                    if (C3H_ABL != 3 && oy < H && ox < W && co < cout) {
                        uint2 o;
                        o.x = pack_bf2(acc[i][j][0], acc[i][j][1]);
                        o.y = pack_bf2(acc[i][j][2], acc[i][j][3]);
                        *reinterpret_cast<uint2*>(yg + (((int64_t)cn * H + oy) * W + ox) * d.ldy + co) = o;
                        const float t0 = bf_lo(o.x), t1 = bf_hi(o.x), t2 = bf_lo(o.y), t3 = bf_hi(o.y);
                        s1[0] += t0; s1[1] += t1; s1[2] += t2; s1[3] += t3;
                        s2[0] += t0 * t0; s2[1] += t1 * t1; s2[2] += t2 * t2; s2[3] += t3 * t3;
                    }
                    if (C3H_ABL == 3) s1[0] += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
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
        };

        lds_barrier();                                  // item 0 is staged
        for (int q = 0; q < nsteps; ++q) {
            if (q < nitems) {
                multiply(q & 1);
                if (cur_c + 1 < nchunk) ++cur_c;
                else {
                    finish_tile();
                    cur_c = 0;
                    ++cur_k;
                    if (cur_k < my_tiles) tile_coords(cur_k, cn, cty0, ctx0);
                }
            }
            if (!FDB) lds_barrier();
            lds_barrier();
        }
    }

    if (want_stats) {
        __syncthreads();
        if (tid < 2 * MQ * 16) {
            const int rep = blockIdx.x % DS_R;
            const int half = tid / (MQ * 16), c = tid - half * (MQ * 16);
            if (c < cout) atomicAdd(d.stat_acc + (size_t)rep * 2 * cout + half * cout + c, red[tid]);
        }
    }
}

template <int MQ, int DIL>
int launch_halo(const dsaspp_conv_desc& d, hipStream_t s) {
    using S = HaloShape<MQ, DIL>;
    const int tiles_x = (d.W + TW - 1) / TW, tiles_y = (d.H + TH - 1) / TH;
    const int ntiles = d.n_img * tiles_x * tiles_y;
    const size_t lds = (size_t)S::LDS0 + (size_t)2 * d.cin * 4;
    if (lds > (size_t)LDS_MAX) return -1;               // buffers + scale/shift table do not fit: generic kernel
    auto kern = conv3x3_halo_kernel<MQ, DIL>;
    dsaspp_lds_limit(reinterpret_cast<const void*>(kern), LDS_MAX);
    // one block per CU; a block with more than one tile needs a grid that is a multiple of 8 (tiles of one XCD stay together)
    int cus = dsaspp_persistent_cus() & ~7;
    if (const char* e = getenv("DSASPP_DGG_GRID")) {      // test hook: fewer blocks, so that small cases walk several tiles per block
        const int v = atoi(e);
        if (v >= 8) cus = v / 8 * 8;
    }
    const int grid = ntiles <= cus ? ntiles : cus;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, s, d, tiles_x, tiles_y, ntiles);
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
