// BN-ReLU -> 3x3 dilated convolution of the dense layers (wide bottleneck in, growth-rate channels out),
// gfx950.  The generic implicit-GEMM kernel stages the shifted input tile once per filter tap, i.e. nine
// times; here a block owns an 8 x 32 pixel patch of one image and stages the patch plus its dilation halo
// ONCE per 32-channel chunk, then runs all nine taps out of LDS by shifting the fragment base address.
//
// What bounds the kernel is what a CU can draw from L2 (the halo makes the patch 1.3-2.5 times the tile, and every tile
// needs the whole filter): about 12 B/clk through registers, about 21 B/clk by LDS-DMA.  Patch and filter slice take
// most of the LDS of a CU, so one block is resident per CU, and a block whose waves all load, then convert, then
// multiply, then store uses the memory path, the vector ALU, the LDS and the matrix pipe one after the other
// (measured: the four parts add up).  Therefore:
//   * the waves are specialised.  Waves 4..7 (one per SIMD) PRODUCE: they bring raw patch rows and the filter slice into
//     LDS by DMA (global_load_lds, no registers) and then apply BatchNorm scale/shift, ReLU and the zero padding to
//     the patch IN PLACE -- each lane converts exactly the 16-byte pieces its own DMA requests brought, so the landing
//     needs no barrier.  Waves 0..3 CONSUME: fragment reads and MFMAs (two patch rows = 4 pixel fragments per wave, so
//     a filter fragment is used four times), output stores and statistics.  One block barrier per step, which waits
//     for LDS traffic only;
//   * the unit of work is an item = (tile, 32-channel chunk); blocks are persistent (one per CU, each walks a strided
//     list of tiles).  With three patch buffers (NPB = 3, where they fit) the DMA of item q+2 lands while item q+1 is
//     converted and item q multiplied; with two, an item is requested and converted within the step before its use.
//     The filter slice has two buffers.  The ring runs on across tile boundaries, and the output stores of a tile drain
//     while the next tile is multiplied;
//   * past the last item the last one is requested again: the number of DMAs in flight is the same in every step, and
//     the hand-counted waits stay exact.
// LDS layout: rows of 64 B (one pixel or one filter row, 32 channels) laid end to end, patch rows 40 pixels apart
// whatever the dilation.  A ds_read_b128 serves lanes 0..7 and 16..23 together (eight rows x two adjacent 16-byte
// slots), and rows r and r+4 of such a layout share their banks: the two 32-byte halves of a row are exchanged where
// bit 2 of the row index is set.  With a row pitch of 40 the bit depends on the column only, so a tap's row offset is
// an immediate and its column offset one of three precomputed bases.  The DMA writes LDS lane-linearly; the exchange
// lives in the global address a lane requests.
#include "common.h"
#include <stdlib.h>

#ifndef C3H_ABL
#define C3H_ABL 0                // diagnostic builds (timing only, results wrong): 1 no fragment reads / MFMA, 3 no stores / statistics,
                                 // 4 no DMA, 5 no prologue pass
#endif

namespace {

constexpr int TH = 8, TW = 32, HWP = 40, KC = 32, ROWB = 64;

constexpr int NT = 512;
constexpr int LDS_MAX = 160 * 1024;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MQ, int DIL>
struct HaloShape {
    static constexpr int HH2 = TH + 2 * DIL;
    static constexpr int PBYTES = (HWP * HH2 * ROWB + 4095) / 4096 * 4096;      // a whole number of DMA instructions per producer wave
    static constexpr int WROWS = 9 * MQ * 16;
    static constexpr int FBYTES = (WROWS * ROWB + 4095) / 4096 * 4096;
    static constexpr int TAIL = 2 * MQ * 16 * 4;                         // statistics area; the scale/shift table follows
    static constexpr int NPB = 3 * PBYTES + 2 * FBYTES + TAIL + 2 * 256 * 4 <= LDS_MAX ? 3 : 2;
    static constexpr int LDS0 = NPB * PBYTES + 2 * FBYTES + TAIL;
};

// block barrier that waits for this wave's LDS operations only (__syncthreads() would also wait for the DMAs that are
// meant to stay in flight)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// wait until at most N vector-memory operations of this wave are outstanding
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}

// 64 lanes x 16 B from the lanes' global addresses to LDS [lds_base, +1024), lane-linear
__device__ __forceinline__ void dma16(const uint32_t lds_base, const unsigned char* p) {
    if (C3H_ABL == 4) asm volatile("" : : "s"(lds_base), "v"(p) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(lds_base), "v"(p) : "memory");
}

template <int MQ, int DIL>
__global__ __launch_bounds__(NT, 2) void conv3x3_halo_kernel(const dsaspp_conv_desc d, const int tiles_x, const int tiles_y,
                                                            const int ntiles) {
    using S = HaloShape<MQ, DIL>;
    constexpr int MP = 4;                               // per consumer wave: two patch rows = 4 half rows of 16 pixels
    constexpr int HH2 = S::HH2, WROWS = S::WROWS, NPB = S::NPB;
    constexpr int PBYTES = S::PBYTES, FBYTES = S::FBYTES;
    constexpr int NP = PBYTES / 4096, NF = FBYTES / 4096;       // DMA instructions per producer wave and item: patch, filter
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const flds = smem + NPB * PBYTES;
    float* red = reinterpret_cast<float*>(smem + NPB * PBYTES + 2 * FBYTES);      // [2][MQ*16] statistics of this block's tiles
    float* tab = red + 2 * MQ * 16;                                               // [scale cin | shift cin]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.H, W = d.W, cin = d.cin, cout = d.cout;
    const int per_img = tiles_x * tiles_y;
    const int nchunk = cin / KC;
    const int G = gridDim.x;                            // a multiple of 8 whenever a block has more than one tile
    const int my_tiles = (ntiles - (int)blockIdx.x + G - 1) / G;
    const int nitems = my_tiles * nchunk;               // = steps (and barriers) of both roles
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
        // DMA instruction i of this wave covers LDS bytes [(4 i + pw) * 1024, +1024) of a buffer: lane L is 16-byte slot
        // (L & 3) of row (4 i + pw) * 16 + (L >> 2), which holds channel piece (L & 3) ^ 2 * bit 2 of the row -- the same
        // piece for every i, since rows of one lane differ by multiples of 8
        const int pw = wave - 4;
        const int piece = (lane & 3) ^ (((lane >> 4) & 1) << 1);
        const unsigned char* xg = reinterpret_cast<const unsigned char*>(d.x);
        const unsigned char* wg = reinterpret_cast<const unsigned char*>(d.w);
        const uint32_t relu_floor = d.in_relu ? 0u : 0x80008000u;
        const uint32_t smem_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)smem);
        const uint32_t flds_l = smem_l + NPB * PBYTES;
        int hyx[NP];                                    // halo row / column of the lane's patch rows, packed; tile-independent
        uint32_t wsrc_off[NF];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pix = (4 * i + pw) * 16 + (lane >> 2);
            const int hy = pix / HWP, hx = pix - hy * HWP;
            hyx[i] = (hy << 8) | hx;                    // rows past the patch (hy >= HH2) and the pad columns read as outside the image
        }
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            int r = (4 * i + pw) * 16 + (lane >> 2);
            r = r < WROWS ? r : WROWS - 1;              // the padding of the buffer: any row
            const int tap = r / (MQ * 16);
            int co = r - tap * (MQ * 16);
            co = co < cout ? co : cout - 1;             // rows past cout: any finite data, their results are never stored
            wsrc_off[i] = (uint32_t)(((co * 9 + tap) * cin + piece * 8) * 2);
        }

        // load cursor: the item requested next
        int ld_k = 0, ld_c = 0, ld_n, ld_ty0, ld_tx0;
        tile_coords(0, ld_n, ld_ty0, ld_tx0);
        // requests the patch of the cursor's item into patch buffer `buf`; returns which of the lane's rows lie in the image
        auto issue_patch = [&](const int buf) __attribute__((always_inline)) {
            unsigned m = 0;
            const uint32_t cb = (uint32_t)(ld_c * KC * 2) + piece * 16;
            const uint32_t base_l = smem_l + buf * PBYTES + pw * 1024;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int hy = hyx[i] >> 8, hx = hyx[i] & 255;
                const int iy = ld_ty0 - DIL + hy, ix = ld_tx0 - DIL + hx;
                const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && hy < HH2 && hx < TW + 2 * DIL;
                m |= ok ? (1u << i) : 0u;
                const uint32_t off = ok ? (uint32_t)(((ld_n * H + iy) * W + ix) * d.ldx) * 2u + cb : 0u;
                dma16(base_l + i * 4096, xg + (size_t)off);
            }
            // advance; past the last item the cursor stays on it
            if (ld_c + 1 < nchunk) ++ld_c;
            else if (ld_k + 1 < my_tiles) {
                ld_c = 0;
                ++ld_k;
                tile_coords(ld_k, ld_n, ld_ty0, ld_tx0);
            }
            return m;
        };
        int fl_c = 0;                                   // chunk of the filter slice requested next
        auto issue_filter = [&](const int buf) __attribute__((always_inline)) {
            const uint32_t base_l = flds_l + buf * FBYTES + pw * 1024;
#pragma unroll
            for (int i = 0; i < NF; ++i) dma16(base_l + i * 4096, wg + (size_t)(wsrc_off[i] + (uint32_t)(fl_c * KC * 2)));
            fl_c = fl_c + 1 < nchunk ? fl_c + 1 : 0;
        };
        // ---- BatchNorm + ReLU + zero padding, in place: the 32 channels of conversion cursor st_c in patch buffer `buf`
        int st_c = 0;
        auto convert = [&](const int buf, const unsigned okm) __attribute__((always_inline)) {
            float sc[8], sh[8];
            load8f(tab + st_c * KC + piece * 8, sc);
            load8f(tab + cin + st_c * KC + piece * 8, sh);
            unsigned char* p = smem + buf * PBYTES + pw * 1024 + lane * 16;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                if (C3H_ABL == 5) continue;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(p + i * 4096), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                uint4 o = relu_bf8(pack8(f), relu_floor);   // ReLU after the rounding (packed 16-bit max)
                const bool k = (okm >> i) & 1u;             // zero padding comes after BatchNorm + ReLU
                o.x = k ? o.x : 0u; o.y = k ? o.y : 0u; o.z = k ? o.z : 0u; o.w = k ? o.w : 0u;
                *reinterpret_cast<uint4*>(p + i * 4096) = o;
            }
            st_c = st_c + 1 < nchunk ? st_c + 1 : 0;
        };

        if (NPB == 3) {
            // in flight at the first wait of step q, oldest first: [patch q+1][filter q+1][patch q+2]
            issue_filter(0);
            unsigned m_cur = issue_patch(0);
            unsigned m_next = issue_patch(1);
            wait_vm<NP>();                              // filter 0 and patch 0 have landed
            convert(0, m_cur);
            lds_barrier();
            int b1 = 1, b2 = 2;                         // patch buffers of items q+1, q+2
            for (int q = 0; q < nitems; ++q) {
                issue_filter((q + 1) & 1);
                const unsigned m_new = issue_patch(b2);
                wait_vm<NF + NP>();                     // patch q+1 (requested a step ago) has landed
                convert(b1, m_next);
                m_next = m_new;
                wait_vm<NP>();                          // filter q+1 has landed
                lds_barrier();
                b1 = b2;
                b2 = b2 == 2 ? 0 : b2 + 1;
            }
        } else {
            issue_filter(0);
            unsigned m = issue_patch(0);
            wait_vm<0>();
            convert(0, m);
            lds_barrier();
            for (int q = 0; q < nitems; ++q) {
                const int b = (q + 1) & 1;
                issue_filter(b);
                m = issue_patch(b);
                wait_vm<0>();
                convert(b, m);
                lds_barrier();
            }
        }
    } else {
        // =========================================================================== consumers
        // fragment coordinates (bases at the top-left tap: a tap's row offset is a non-negative immediate, its column offset
        // selects one of three bases because it can flip the half-row exchange)
        const int fr = lane & 15, fc = lane >> 4;
        int rbase[MP][3];
#pragma unroll
        for (int j = 0; j < MP; ++j)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) {
                const int pix = (2 * wave + (j >> 1)) * HWP + (j & 1) * 16 + fr + tx * DIL;
                rbase[j][tx] = pix * ROWB + ((fc ^ (((pix >> 2) & 1) << 1)) << 4);
            }
        const int wbase = fr * ROWB + ((fc ^ (((fr >> 2) & 1) << 1)) << 4);
        f32x4 acc[MQ][MP];
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u16* yg = reinterpret_cast<u16*>(d.y);
        int cur_k = 0, cur_c = 0, cn, cty0, ctx0;
        tile_coords(0, cn, cty0, ctx0);

        // ---- nine taps of an item out of patch buffer pb / filter buffer fb.  The fragments of a tap are read one tap ahead of
        //      the MFMAs that use them (explicit double buffer, order pinned with scheduling barriers): left to itself the
        //      compiler reads each group right before its MFMAs and the wave -- the only one of its SIMD that multiplies --
        //      waits out an LDS round trip per group.
        auto multiply = [&](const int pb, const int fb) __attribute__((always_inline)) {
            const unsigned char* pa = smem + pb * PBYTES;
            const unsigned char* pwf = flds + fb * FBYTES + wbase;
            bf16x8 afb[2][MP], wfb[2][MQ];
            auto read_group = [&](const int tap) __attribute__((always_inline)) {
                const int tyy = tap / 3, txx = tap - tyy * 3;
                const int toff = tyy * DIL * HWP * ROWB;
#pragma unroll
                for (int j = 0; j < MP; ++j) afb[tap & 1][j] = *reinterpret_cast<const bf16x8*>(pa + rbase[j][txx] + toff);
#pragma unroll
                for (int i = 0; i < MQ; ++i) wfb[tap & 1][i] = *reinterpret_cast<const bf16x8*>(pwf + (tap * MQ + i) * 16 * ROWB);
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
        int pb = 0;
        for (int q = 0; q < nitems; ++q) {
            multiply(pb, q & 1);
            if (cur_c + 1 < nchunk) ++cur_c;
            else {
                finish_tile();
                cur_c = 0;
                ++cur_k;
                if (cur_k < my_tiles) tile_coords(cur_k, cn, cty0, ctx0);
            }
            lds_barrier();
            pb = pb + 1 < NPB ? pb + 1 : 0;
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
