// Weight gradient of the BN-ReLU -> 3x3 convolution of a dense layer (few output channels, dilation <= 4), gfx950:
//
//   dw[q][ky][kx][c] += sum_{n,h,w} dz[n][h][w][q] * A[n][h + (ky-1)*dil][w + (kx-1)*dil][c],   A = bf16(relu(x*scale+shift)), 0 outside
//
// organised so that every activation row is drawn from L2 ONCE for all nine taps.  A block owns 64 input channels and a
// contiguous range of (image, 32-pixel column strip, image row) triples and walks it row by row.  One step = one
// activation row segment of 32 + 2*dil pixels (prologue applied, zero outside the image) against the dz row segments of
// the image rows h + dil, h and h - dil, which wait in a ring of 16 row slots in LDS: the 3 x 3 taps are three row
// offsets into the activation tile (kx) times three ring slots (ky).  K = 32 pixels per step.  Four waves, a wave owns
// 16 input channels and all 9 taps x cout: 27 accumulator tiles (108 registers).  Both tiles keep the natural
// [pixel][channel] layout and are read with the transposing LDS read.  dz rows arrive by LDS-DMA (one instruction of
// 16*cout/16 lanes per wave and step), activation chunks through registers (two loads per thread and step), both
// requested LAT = 4 steps before their use, so that a step is: barrier, requests, up to 27 MFMAs per wave, one counted
// wait (vmcnt(9): three steps of three requests are younger), prologue of the next row into the other activation stage.
// Requests outside the block's rows are made anyway, from clamped addresses into ring slots and registers nobody reads:
// every step issues the same three vector-memory operations, which keeps the wait count a constant.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int SW = 32;                       // strip width in pixels = K of a step
constexpr int MAXD = 4;                      // largest dilation
constexpr int XROWS = SW + 2 * MAXD;         // rows of an activation tile
constexpr int XS = 144;                      // its row stride in bytes: 64 channels + 16 (bank spread of the transposed read)
constexpr int XSTAGE = XROWS * XS;
constexpr int NR = 16;                       // dz row slots: rows v - dil .. v + dil + LAT live at step v
constexpr int LAT = 4;                       // steps between a request and its use
constexpr int NT = 256;
#ifndef DSASPP_WS_ABL
#define DSASPP_WS_ABL 0      // diagnostic builds: 1 no atomic epilogue, 2 no MFMA, 3 no activation loads, 4 no prologue / stage writes
#endif

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

struct SArgs {
    const u16* x;
    const u16* dz;
    const float* sc;
    const float* sh;
    float* dw;
    int ldx, lddz, cin, cout, H, W, d, nstrip, nthird;
    long long S;                             // strip rows in all: n_img * nstrip * H
    uint32_t relu_bits;
};

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// pixels k..k+3 and k+4..k+7 (four rows further) of this lane's channel
__device__ __forceinline__ bf16x8 tr_frag(const uint32_t addr, const uint32_t four_rows) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)addr);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(addr + four_rows));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// NQB: 16-channel blocks of the output (cout = 16 * NQB)
template <int NQB>
__global__ __launch_bounds__(NT, 2) void wgrad3x3_strip_kernel(const SArgs a) {
    constexpr int ZROW = NQB * 32;           // bytes of one pixel of a dz row
    constexpr int ZSLOT = SW * ZROW;
    constexpr int ZLANES = 16 * NQB;         // lanes of a wave's dz DMA: 4 waves x ZLANES chunks = one row segment
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const uint32_t xst_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)smem);     // two activation stages
    const uint32_t zring_l = xst_l + 2 * XSTAGE;                                                       // NR dz row slots

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W, d = a.d;
    const int third = blockIdx.x % a.nthird, gidx = blockIdx.x / a.nthird;
    const int G = ((int)gridDim.x - third + a.nthird - 1) / a.nthird;       // blocks that own this channel range
    long long f0 = a.S * gidx / G;
    const long long f1 = a.S * (gidx + 1) / G;
    const int cbase = third * 64;
    const uint32_t ldx2 = (uint32_t)a.ldx * 2u, lddz2 = (uint32_t)a.lddz * 2u;

    // ---- this thread's share of an activation tile: chunk xch of tile rows xr0 and 32 + xr0 (the latter only below 32 + 2*dil)
    const int xch = tid & 7, xr0 = tid >> 3;
    const bool has2 = xr0 < 2 * d;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = 1.f;
        sh[e] = 0.f;
    }
    if (a.sc) {
        load8f(a.sc + cbase + xch * 8, sc);
        load8f(a.sh + cbase + xch * 8, sh);
    }
    // (the values are used here once, so that the compiler's wait for these loads stands in front of the loop and not, as
    // vmcnt(0), inside it)
#pragma unroll
    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(sc[e]), "+v"(sh[e]));
    const uint32_t x_dst = xr0 * XS + xch * 16;

    // ---- this lane's share of a dz row segment: chunk zk of the segment (pixel zk / (2*NQB), 16-byte piece zk % (2*NQB))
    const int zk = wave * ZLANES + (lane < ZLANES ? lane : 0);
    const int zpx = zk / (2 * NQB), zpc = zk - zpx * (2 * NQB);

    // ---- fragment coordinates: lane (g = lane/16, i16 = lane%16) supplies tile row g*8 + i16/4 (+4 for the second read),
    //      bytes 8*(i16%4).. of a 16-channel block, and receives channel i16, pixels g*8..g*8+7
    const int g16 = lane >> 4, i16 = lane & 15;
    const int frow = g16 * 8 + (i16 >> 2);
    const uint32_t zfrag = frow * ZROW + (i16 & 3) * 8;
    const uint32_t xfrag = frow * XS + wave * 32 + (i16 & 3) * 8;

    f32x4 acc[3][3][NQB];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb) acc[ky][kx][qb] = f32x4{0.f, 0.f, 0.f, 0.f};

    u32x4 b0[2], b1[2], b2[2], b3[2];

    while (f0 < f1) {
        // ---- one segment: rows h0 .. h1-1 of column strip `col`
        const int col = (int)(f0 / H);
        const int h0 = (int)(f0 - (long long)col * H);
        const int h1 = (long long)(H - h0) < f1 - f0 ? H : h0 + (int)(f1 - f0);
        f0 += h1 - h0;
        const int n = col / a.nstrip;
        const int w0 = (col - n * a.nstrip) * SW;
        const int r_lo = h0 - d > 0 ? h0 - d : 0, r_hi = h1 + d < H ? h1 + d : H;      // activation rows that meet a dz row of the segment
        const int v_start = h0 - d - LAT;
        const int nsteps = (r_hi - v_start + 3) & ~3;
        // image bases as scalars; per-lane parts are 32-bit byte offsets inside the image
        const unsigned char* xn = reinterpret_cast<const unsigned char*>(a.x + (long long)n * H * W * a.ldx + cbase);
        const unsigned char* zn = reinterpret_cast<const unsigned char*>(a.dz + (long long)n * H * W * a.lddz);
        const int wp0 = w0 - d + xr0, wp1 = wp0 + SW;
        const bool ok0 = wp0 >= 0 && wp0 < W, ok1 = has2 && wp1 >= 0 && wp1 < W;
        const uint32_t xo0 = (uint32_t)(ok0 ? wp0 : w0) * ldx2 + xch * 16;
        const uint32_t xo1 = (uint32_t)(ok1 ? wp1 : w0) * ldx2 + xch * 16;
        const uint32_t zo = (uint32_t)(w0 + zpx) * lddz2 + zpc * 16;
        const uint32_t xrow2 = (uint32_t)W * ldx2, zrow2 = (uint32_t)W * lddz2;

        auto issue_z = [&](const int hrow) {
            int hh = hrow < h0 ? h0 : hrow;
            hh = hh < h1 ? hh : h1 - 1;
            const uint32_t off = (uint32_t)hh * zrow2 + zo;
            const uint32_t m0v = zring_l + (hrow & (NR - 1)) * ZSLOT + wave * (ZLANES * 16);
            if (lane < ZLANES)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m0v), "v"(off), "s"(zn) : "memory");
        };
        auto issue_x = [&](u32x4 (&buf)[2], const int r) {
            int rr = r < 0 ? 0 : r;
            rr = rr < H ? rr : H - 1;
            const uint32_t ro = (uint32_t)rr * xrow2;
            if (DSASPP_WS_ABL == 3) {
                asm volatile("" : "=v"(buf[0]), "=v"(buf[1]) : "v"(ro + xo0), "v"(ro + xo1));
                return;
            }
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(buf[0]) : "v"(ro + xo0), "s"(xn) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(buf[1]) : "v"(ro + xo1), "s"(xn) : "memory");
        };
        // prologue of a landed row into an activation stage; rows outside the segment's range and pixels outside the image are zero
        auto consume = [&](u32x4 (&buf)[2], const bool row_ok, const int stage) {
            asm volatile("" : "+v"(buf[0]), "+v"(buf[1]));        // from here on the registers of the loads are ordinary values
            if (DSASPP_WS_ABL == 4) return;
            const uint32_t dst = xst_l + stage * XSTAGE + x_dst;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float f[8];
                unpack8(__builtin_bit_cast(uint4, buf[i]), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                uint4 v = relu_bf8(pack8(f), a.relu_bits);            // ReLU after the rounding (packed 16-bit max)
                const bool ok = row_ok && (i == 0 ? ok0 : ok1);
                v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
                if (i == 0 || has2) *(__attribute__((address_space(3))) u32x4*)(dst + i * SW * XS) = __builtin_bit_cast(u32x4, v);
            }
        };
        auto step = [&](const int k, u32x4 (&bi)[2], u32x4 (&bc)[2]) {
            const int v = v_start + k;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_z(v + d + LAT);
            issue_x(bi, v + LAT);
            if (v >= r_lo && v < r_hi && (DSASPP_WS_ABL != 2 || a.cin < 0)) {
                const uint32_t xb = xst_l + (k & 1) * XSTAGE + xfrag;
                bf16x8 xf[3];
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) xf[kx] = tr_frag(xb + kx * d * XS, 4 * XS);
                // (requesting the dz fragments of all three filter rows up front instead was measured: no faster, 24 more registers)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int h = v - (ky - 1) * d;
                    if (h >= h0 && h < h1) {
                        const uint32_t zb = zring_l + (h & (NR - 1)) * ZSLOT + zfrag;
                        bf16x8 zf[NQB];
#pragma unroll
                        for (int qb = 0; qb < NQB; ++qb) zf[qb] = tr_frag(zb + qb * 32, 4 * ZROW);
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int qb = 0; qb < NQB; ++qb)
                                acc[ky][kx][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[qb], xf[kx], acc[ky][kx][qb], 0, 0, 0);
                    }
                }
            }
            wait_vm<3 * (LAT - 1)>();
            consume(bc, v + 1 >= r_lo && v + 1 < r_hi, (k + 1) & 1);
        };
        for (int k = 0; k < nsteps; k += 4) {
            step(k, b0, b1);
            step(k + 1, b1, b2);
            step(k + 2, b2, b3);
            step(k + 3, b3, b0);
        }
        // the requests past the segment's end land in slots and registers nobody reads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("" : "+v"(b0[0]), "+v"(b0[1]), "+v"(b1[0]), "+v"(b1[1]), "+v"(b2[0]), "+v"(b2[1]), "+v"(b3[0]), "+v"(b3[1]));
    }

    // ---- epilogue: acc[ky][kx][qb][r] is (q = qb*16 + fc*4 + r, tap = ky*3 + kx, c = cbase + wave*16 + fr)
    const int fr = lane & 15, fc = lane >> 4;
    const int c = cbase + wave * 16 + fr;
    if (DSASPP_WS_ABL == 1 && a.cin > 0) return;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qb * 16 + fc * 4 + r;
                    atomicAdd(a.dw + ((long long)q * 9 + ky * 3 + kx) * a.cin + c, acc[ky][kx][qb][r]);
                }
}

}  // namespace

// Form check + launch.  Returns -1 when the descriptor is not of the form this kernel is for (the caller then uses the
// generic kernel), 0 on success, a positive error code otherwise.
int dsaspp_wgrad3x3_strip_launch(const dsaspp_wgrad_desc& d, hipStream_t s) {
    static const int enabled = [] { const char* e = getenv("DSASPP_WGRAD3_STRIP"); return e ? atoi(e) : 1; }();
    if (!enabled) return -1;
    const bool form = d.taps == 9 && d.n_src <= 1 && d.dtype == 0 && !d.in_mask && (d.in_relu == 0 || d.in_relu == 1) &&
                      ((d.in_scale != nullptr) == (d.in_shift != nullptr)) && d.dil >= 1 && d.dil <= MAXD && d.W % SW == 0 && d.cin % 64 == 0 &&
                      (d.cout == 32 || d.cout == 48) && d.ldx % 8 == 0 && d.lddz % 8 == 0;
    if (!form) return -1;
    // per-lane operand offsets are 32-bit byte offsets inside one image
    const int64_t hw = (int64_t)d.H * d.W;
    if (hw * d.ldx * 2 >= (1ll << 32) || hw * d.lddz * 2 >= (1ll << 32)) return -1;
    if ((((uintptr_t)d.x | (uintptr_t)d.dz) & 15) != 0 || (d.in_scale && ((((uintptr_t)d.in_scale | (uintptr_t)d.in_shift) & 15) != 0))) return -1;
    const int n_cu = dsaspp_persistent_cus();
    SArgs a;
    a.x = reinterpret_cast<const u16*>(d.x);
    a.dz = reinterpret_cast<const u16*>(d.dz);
    a.sc = d.in_scale;
    a.sh = d.in_shift;
    a.dw = d.dw;
    a.ldx = d.ldx;
    a.lddz = d.lddz;
    a.cin = d.cin;
    a.cout = d.cout;
    a.H = d.H;
    a.W = d.W;
    a.d = d.dil;
    a.nstrip = d.W / SW;
    a.nthird = d.cin / 64;
    a.S = (long long)d.n_img * a.nstrip * d.H;
    a.relu_bits = d.in_relu ? 0u : 0x80008000u;
    // two resident blocks per CU; every block gets at least one strip row
    long long grid = 2ll * n_cu;
    if (const char* e = getenv("DSASPP_DGG_GRID")) {      // test hook: fewer blocks, so that small cases walk several segments per block
        const int v = atoi(e);
        if (v >= 1) grid = v;
    }
    if (grid > a.S * a.nthird) grid = a.S * a.nthird;
    grid = grid / a.nthird * a.nthird;
    if (grid < a.nthird) return -1;
    const size_t lds = 2 * XSTAGE + (size_t)NR * SW * (d.cout * 2);
    if (d.cout == 48) hipLaunchKernelGGL(wgrad3x3_strip_kernel<3>, dim3((unsigned)grid), dim3(NT), lds, s, a);
    else hipLaunchKernelGGL(wgrad3x3_strip_kernel<2>, dim3((unsigned)grid), dim3(NT), lds, s, a);
    DS_LAUNCH_CHECK("conv_wgrad(3x3 strips)");
    return 0;
}
