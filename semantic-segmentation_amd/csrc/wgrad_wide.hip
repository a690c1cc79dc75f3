// Weight gradient of a BN-ReLU -> 1x1 convolution with many input and output channels (transitions, ASPP 1x1 layers), gfx950:
//
//   dw[q][c] += sum_m dz[m][q] * bf16(relu(x[m][c] * in_scale[c] + in_shift[c]))
//
// on a 256 (output channels) x 256 (input channels) x 32 (pixels) step: 8 waves as 2 x 4, wave tile 128 x 64 = 128
// accumulator registers, one block per (channel tile pair, pixel range), partial sums of the pixel ranges combined with
// fp32 atomics.  K = pixels is the strided dimension of both NHWC operands, so the stage keeps the natural
// [pixel][channel] layout (32 rows of 512 bytes per operand) and the fragments come from the transposing LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads 4 pixels x 16 channels, every lane receives the 4 pixels of its channel).
// Rows are not padded; the 16-channel blocks of row r sit at block position b ^ swz(r), which spreads the four rows of
// one transposed read and the two lane groups served together over all banks and keeps every DMA a lane-linear 1 KiB
// image.  The dz half of a stage arrives by LDS-DMA (the swizzle is in the source address).  The x half takes the
// register path because the prologue is applied on the way: a thread owns one 16-byte chunk position of two rows, its 8
// channels (so its scale / shift values, kept in registers) are the same at every step; it requests the chunks
// global -> registers two steps ahead into alternating buffers (inline assembly: the compiler must not wait for them
// where it pleases) and after the MFMAs of step t writes the chunks of step t + 1 into that stage's slot.
// Queue per wave and step: [2 dz DMAs][2 x chunk loads]; one counted wait, vmcnt(4), in front of the prologue work.
// Pixel rows past the end contribute zero through the x side (their dz rows are fetched from the last valid row).
//
// T9: the dilated 3x3 form with many input channels (the ASPP 3x3 layers, 512 -> 128, dilation 3 .. 24),
//
//   dw[q][ky][kx][c] += sum_{n,h,w} dz[n][h - (ky-1)*dil][w - (kx-1)*dil][q] * A[n][h][w][c]        (dz = 0 outside the image)
//
// as the same GEMM over a VIRTUAL dz operand of 9 * cout channels: virtual channel t * cout + q of pixel row p is channel q
// of the dz row shifted by tap t.  Only the source address of the dz DMA changes: every lane owns one 16-byte chunk
// position of a stage row, i.e. a fixed (tap, channel) pair, follows the (h, w) coordinates of its row from step to step
// (rows advance by 32 pixels: at most one wrap when W >= 32) and fetches from the shifted row, or -- outside the image,
// past the last virtual channel or past the last pixel -- from a row of zeros; the address is a 64-bit per-lane one for
// that reason.  The x side and the K-loop are those of the 1x1 form; 256 virtual channels are two taps of 128.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BQ = 256, BC = 256, BK = 32;
constexpr int NT = 512, WQ = 2, WC = 4;
constexpr int TQ = BQ / WQ, TC = BC / WC, MQ = TQ / 16, MC = TC / 16;    // 128 x 64 wave tile: 8 x 4 MFMA tiles
constexpr int ROWB = 512;                                                 // stage row: 256 channels
constexpr int HALF = BK * ROWB;                                           // 16 KiB: dz rows, then x rows
constexpr int STAGE = 2 * HALF;
constexpr int NS = 3;                                                     // ring stages

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

struct WWArgs {
    const u16* x;
    const u16* dz;
    const float* sc;
    const float* sh;
    float* dw;
    int ldx, lddz, cin, cout, M, nQ, nC, chunk;      // cout: channels of the (virtual) dz operand
    uint32_t relu_bits;
    // 3x3 form
    const unsigned char* zero;                       // >= 16 bytes of zeros
    int H, W, dil, qreal;                            // qreal: output channels of the layer (cout = 9 * qreal)
};

// zeros for the dz DMA of the 3x3 form (a static device array is zero-initialised; one copy per GPU)
__device__ __attribute__((aligned(256))) unsigned int g_zero_row[64];

// block swizzle of a stage row: rows r..r+3 of one transposed read differ in the low two bits, the two lane groups that
// are served together (rows r and r + 8) in the third
__device__ __forceinline__ int swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory"); }

// pixels k..k+3 and k+4..k+7 (four rows = 2 KiB further) of this lane's channel
__device__ __forceinline__ bf16x8 tr_frag(const uint32_t addr) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)addr);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(addr + 4 * ROWB));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <bool T9>
__global__ __launch_bounds__(NT, 2) void wgrad_wide_kernel(const WWArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    unsigned char* ring = smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks that share a pixel range (and so both operand streams) get neighbouring logical ids -> same XCD / L2
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % a.nC;
    bid /= a.nC;
    const int qt = bid % a.nQ;
    const int split = bid / a.nQ;
    const int q_base = qt * BQ, c_base = ct * BC;
    const int cin = a.cin, cout = a.cout;
    const int m_begin = split * a.chunk;
    const int ksteps = a.chunk / BK;                  // even (the launcher rounds the range to 64 pixels)
    const int m_last = a.M - 1 - m_begin;             // last valid row, relative to the range (>= 0)
    // operand bases of this block as scalars; per-lane parts are 32-bit byte offsets
    const unsigned char* zbase = reinterpret_cast<const unsigned char*>(a.dz + (int64_t)m_begin * a.lddz + q_base);
    const unsigned char* zimg = reinterpret_cast<const unsigned char*>(a.dz);
    const unsigned char* xbase = reinterpret_cast<const unsigned char*>(a.x + (int64_t)m_begin * a.ldx + c_base);
    const uint32_t lddz2 = (uint32_t)a.lddz * 2u, ldx2 = (uint32_t)a.ldx * 2u;
    const uint32_t ring_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)ring);

    // ---- this thread's share of the x half: chunk position xpos of rows xrow and xrow + 16 (same swizzle)
    const int xpos = tid & 31, xrow = tid >> 5;
    const bool xc_ok = c_base + xpos * 8 < cin;
    const uint32_t xc_off = xc_ok ? (uint32_t)xpos * 16u : 0u;      // a chunk past the last channel: the first chunk, its result is never stored
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e] = 1.f;
        sh[e] = 0.f;
    }
    if (a.sc && xc_ok) {
        load8f(a.sc + c_base + xpos * 8, sc);
        load8f(a.sh + c_base + xpos * 8, sh);
    }
    const uint32_t x_dst = HALF + xrow * ROWB + (((((xpos >> 1) ^ swz(xrow)) << 1) | (xpos & 1)) << 4);

    // ---- this lane's share of the dz half: DMA instruction n of a wave fills rows wave*4 + n*2 + lane/32, the lane brings
    //      the chunk that belongs at its lane-linear position
    int zrow[2];
    uint32_t zc_off[2];
    // 3x3 form: this lane's tap offsets, whether its virtual channel exists, and the image coordinates of its row at the
    // stage that is requested next
    int zdy[2], zdx[2], zh[2], zw[2];
    bool zvc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        zrow[n] = wave * 4 + n * 2 + (lane >> 5);
        const int p = lane & 31;
        const int gc = (((p >> 1) ^ swz(zrow[n])) << 1) | (p & 1);
        if (T9) {
            const int Qc = q_base + gc * 8;
            zvc[n] = Qc < cout;
            const int t = zvc[n] ? Qc / a.qreal : 0;
            const int ky = t / 3, kx = t - ky * 3;
            zdy[n] = (ky - 1) * a.dil;
            zdx[n] = (kx - 1) * a.dil;
            zc_off[n] = (uint32_t)(Qc - t * a.qreal) * 2u;
            const int p0 = (m_begin + zrow[n]) % (a.H * a.W);
            zh[n] = p0 / a.W;
            zw[n] = p0 - zh[n] * a.W;
        } else {
            zc_off[n] = (q_base + gc * 8 < cout) ? (uint32_t)gc * 16u : 0u;
        }
    }

    int c_slot = 0;
    auto issue_z = [&](const int s) {
        const uint32_t slot_l = ring_l + c_slot * STAGE;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (T9) {
                // (stages are requested in order, so the coordinates advance by one stage per call)
                const int rel = s * BK + zrow[n];
                const int hs = zh[n] - zdy[n], ws = zw[n] - zdx[n];
                const bool ok = zvc[n] && rel <= m_last && (unsigned)hs < (unsigned)a.H && (unsigned)ws < (unsigned)a.W;
                const unsigned char* src = ok ? zimg + ((int64_t)(m_begin + rel) - (int64_t)zdy[n] * a.W - zdx[n]) * (int64_t)lddz2 + zc_off[n] : a.zero;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                             : : "s"(slot_l + (wave * 2 + n) * 1024), "v"(src) : "memory");
                zw[n] += BK;
                if (zw[n] >= a.W) {
                    zw[n] -= a.W;
                    zh[n] = zh[n] + 1 < a.H ? zh[n] + 1 : 0;
                }
            } else {
                int rel = s * BK + zrow[n];
                rel = rel < m_last ? rel : m_last;
                const uint32_t off = (uint32_t)rel * lddz2 + zc_off[n];
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             : : "s"(slot_l + (wave * 2 + n) * 1024), "v"(off), "s"(zbase) : "memory");
            }
        }
        c_slot = c_slot + 1 < NS ? c_slot + 1 : 0;
    };
    auto issue_x = [&](u32x4 (&buf)[2], int& m_buf, const int s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int rel = s * BK + xrow + i * 16;
            rel = rel < m_last ? rel : m_last;
            const uint32_t off = (uint32_t)rel * ldx2 + xc_off;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(buf[i]) : "v"(off), "s"(xbase) : "memory");
        }
        m_buf = s * BK;
    };
    // prologue of a landed buffer into the x half of its stage's slot
    auto consume = [&](u32x4 (&buf)[2], const int m_buf, const int slot) {
        asm volatile("" : "+v"(buf[0]), "+v"(buf[1]));        // from here on the registers of the loads are ordinary values
        unsigned char* dst = ring + slot * STAGE + x_dst;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float f[8];
            unpack8(__builtin_bit_cast(uint4, buf[i]), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
            uint4 v = relu_bf8(pack8(f), a.relu_bits);            // ReLU after the rounding (packed 16-bit max)
            const bool ok = m_buf + xrow + i * 16 <= m_last;      // zero padding comes after the prologue
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            *reinterpret_cast<uint4*>(dst + i * 16 * ROWB) = v;
        }
    };

    // ---- fragment addressing: lane (g = lane/16, i16 = lane%16) supplies row g*8 + i16/4 (+4 for the second read),
    //      bytes 8*(i16%4).. of a 16-channel block and receives channel i16, pixels g*8..g*8+7.  Block b of the wave's
    //      range sits at (b ^ swz(row)): the swizzle is in bits 5..7 of the offset, so block i is one XOR away.
    const int g16 = lane >> 4, i16 = lane & 15;
    const int wq = wave % WQ, wc = wave / WQ;
    const int frow = g16 * 8 + (i16 >> 2), fs = swz(frow);
    const uint32_t zb = frow * ROWB + wq * (TQ * 2) + (fs << 5) + (i16 & 3) * 8;
    const uint32_t xb = HALF + frow * ROWB + (((wc * MC) ^ fs) << 5) + (i16 & 3) * 8;

    f32x4 acc[MQ][MC];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < MC; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- start: stages 0 and 1 requested, stage 0's x half written
    u32x4 xa[2], xb_[2];
    int m_a = 0, m_b = 0;
    issue_z(0);
    issue_x(xb_, m_b, 0);
    issue_z(1);
    issue_x(xa, m_a, 1);
    wait_vm<4>();
    consume(xb_, m_b, 0);
    int r_slot = 0;

    // one K-step: stage r_slot is complete after the barrier (every wave waited for its dz DMAs in front of the previous
    // step's prologue work and wrote its x chunks); request stage s + 2, MFMAs of stage s, then the x half of stage s + 1
    auto step = [&](const int s, u32x4 (&landed)[2], const int m_landed, u32x4 (&next)[2], int& m_next) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_z(s + 2);
        issue_x(next, m_next, s + 2);
        const uint32_t so = ring_l + r_slot * STAGE;
        bf16x8 xf[MC];
#pragma unroll
        for (int j = 0; j < MC; ++j) xf[j] = tr_frag(so + (xb ^ (j << 5)));
#pragma unroll
        for (int ih = 0; ih < MQ; ih += 4) {
            bf16x8 zf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) zf[i] = tr_frag(so + (zb ^ ((ih + i) << 5)));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < MC; ++j) acc[ih + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[i], xf[j], acc[ih + i][j], 0, 0, 0);
        }
        wait_vm<4>();
        const int w_slot = r_slot + 1 < NS ? r_slot + 1 : 0;
        consume(landed, m_landed, w_slot);
        r_slot = w_slot;
    };
    for (int s = 0; s < ksteps; s += 2) {
        step(s, xa, m_a, xb_, m_b);
        step(s + 1, xb_, m_b, xa, m_a);
    }
    // the two stages requested past the end land in free slots and registers nobody reads
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(xa[0]), "+v"(xa[1]), "+v"(xb_[0]), "+v"(xb_[1]));

    // ---- epilogue: acc[i][j][r] is (q = q0 + i*16 + fc*4 + r, c = c0 + j*16 + fr); 16 lanes = 64 contiguous bytes of a dw row
    const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
    for (int j = 0; j < MC; ++j) {
        const int c = c_base + wc * TC + j * 16 + fr;
        if (c >= cin) continue;
#pragma unroll
        for (int i = 0; i < MQ; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = q_base + wq * TQ + i * 16 + fc * 4 + r;
                if (q >= cout) continue;
                if (T9) {       // virtual channel t * qreal + qq -> dw[qq][t][c]
                    const int t = q / a.qreal, qq = q - t * a.qreal;
                    atomicAdd(a.dw + ((int64_t)qq * 9 + t) * cin + c, acc[i][j][r]);
                } else {
                    atomicAdd(a.dw + (int64_t)q * cin + c, acc[i][j][r]);
                }
            }
        }
    }
}

}  // namespace

// Form check + launch.  Returns -1 when the descriptor is not of the form this kernel is for (the caller then uses the
// generic kernel), 0 on success, a positive error code otherwise.
int dsaspp_wgrad_wide_launch(const dsaspp_wgrad_desc& d, hipStream_t s) {
    static const int enabled = [] { const char* e = getenv("DSASPP_WGRAD_WIDE"); return e ? atoi(e) : 1; }();
    if (!enabled) return -1;
    const bool t9 = d.taps == 9;
    const bool form = (d.taps == 1 || (t9 && d.W >= BK && d.dil >= 1)) && d.n_src <= 1 && d.dtype == 0 && !d.in_mask && (d.in_relu == 0 || d.in_relu == 1) &&
                      ((d.in_scale != nullptr) == (d.in_shift != nullptr)) && d.cin % 8 == 0 && d.cout % 8 == 0 && d.ldx % 8 == 0 && d.lddz % 8 == 0;
    if (!form) return -1;
    static const int min_cin = [] { const char* e = getenv("DSASPP_WGRAD_WIDE_MIN_CIN"); return e ? atoi(e) : 512; }();
    static const int min_cout = [] { const char* e = getenv("DSASPP_WGRAD_WIDE_MIN_COUT"); return e ? atoi(e) : 448; }();
    const int cout_v = t9 ? 9 * d.cout : d.cout;     // channels of the (virtual) dz operand
    if (d.cin < (t9 ? 448 : min_cin) || cout_v < min_cout) return -1;
    if (t9 && (int64_t)d.dil * (d.W + 1) >= (1ll << 30)) return -1;
    const int64_t M64 = (int64_t)d.n_img * d.H * d.W;
    if (M64 >= (1ll << 31)) return -1;
    if ((((uintptr_t)d.x | (uintptr_t)d.dz) & 15) != 0 || (d.in_scale && ((((uintptr_t)d.in_scale | (uintptr_t)d.in_shift) & 15) != 0))) return -1;
    const int n_cu = dsaspp_persistent_cus();
    dsaspp_lds_limit((const void*)wgrad_wide_kernel<false>, 160 * 1024);
    dsaspp_lds_limit((const void*)wgrad_wide_kernel<true>, 160 * 1024);
    // the row of zeros of the 3x3 form: one copy per GPU, its address looked up once per device and thread
    static thread_local int zero_dev = -1;
    static thread_local const unsigned char* zero_ptr = nullptr;
    if (t9) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return -1;
        if (dev != zero_dev) {
            void* zp = nullptr;
            if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_row)) != hipSuccess || !zp) return -1;
            zero_ptr = static_cast<const unsigned char*>(zp);
            zero_dev = dev;
        }
    }
    WWArgs a;
    a.M = (int)M64;
    a.nQ = (cout_v + BQ - 1) / BQ;
    a.nC = (d.cin + BC - 1) / BC;
    const int tiles = a.nQ * a.nC;
    // one round of resident blocks (one per CU): the fewest atomic epilogues
    int ksplit = d.ksplit > 0 ? d.ksplit : n_cu / tiles;
    if (ksplit < 1) ksplit = 1;
    // per-lane operand offsets are 32-bit byte offsets from the block's first row (a 64-bit scalar): a pixel range whose
    // rows span 4 GiB or more is cut into more, shorter ranges
    const int64_t ldmax = d.ldx > d.lddz ? d.ldx : d.lddz;
    int64_t chunk;
    for (;;) {
        chunk = (M64 + ksplit - 1) / ksplit;
        chunk = (chunk + 2 * BK - 1) / (2 * BK) * (2 * BK);
        if (chunk < 4 * BK) chunk = 4 * BK;
        if ((chunk + 2 * BK) * ldmax * 2 < (1ll << 32) || chunk == 4 * BK) break;
        ksplit *= 2;
    }
    ksplit = (int)((M64 + chunk - 1) / chunk);
    if ((chunk + 2 * BK) * ldmax * 2 >= (1ll << 32) || (int64_t)tiles * ksplit >= (1ll << 31)) return -1;
    a.x = reinterpret_cast<const u16*>(d.x);
    a.dz = reinterpret_cast<const u16*>(d.dz);
    a.sc = d.in_scale;
    a.sh = d.in_shift;
    a.dw = d.dw;
    a.ldx = d.ldx;
    a.lddz = d.lddz;
    a.cin = d.cin;
    a.cout = cout_v;
    a.chunk = (int)chunk;
    a.zero = zero_ptr;
    a.H = d.H;
    a.W = d.W;
    a.dil = d.dil;
    a.qreal = d.cout;
    a.relu_bits = d.in_relu ? 0u : 0x80008000u;
    if (t9) hipLaunchKernelGGL(wgrad_wide_kernel<true>, dim3(tiles * ksplit), dim3(NT), (size_t)NS * STAGE, s, a);
    else hipLaunchKernelGGL(wgrad_wide_kernel<false>, dim3(tiles * ksplit), dim3(NT), (size_t)NS * STAGE, s, a);
    DS_LAUNCH_CHECK("conv_wgrad(wide tile)");
    return 0;
}
