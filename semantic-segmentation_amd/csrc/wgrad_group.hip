// Grouped weight gradient of 1x1 layers that read the same input tensor, gfx950.
//
//   dw_s[q][c] += sum_m dz_s[m][q] * relu(x[m][c] * sc_s[c] + sh_s[c])      s = 0 .. n_src-1,  c < cin_s
//
// The dense layers of one block all read the block's feature slab: layer L the channel prefix [0, cin_L), each through
// its own BatchNorm.  Taken one layer at a time the slab prefix is streamed from HBM once per layer and that stream
// bounds the kernel; here a block owns a 128-channel column of the slab over a pixel range and keeps the accumulators
// of up to four layers, so a raw slab tile is fetched ONCE per K-step, held in registers, and pushed through each
// layer's scale/shift/ReLU on its way into LDS.
//
// Pipeline: a "stage" is one (K-step of 64 pixels, source) pair.
//   * dz tiles go global -> LDS by DMA (global_load_lds, 16 B per lane, lane-linear image with 400-byte rows: the
//     25th chunk of a row is padding that keeps the transposing fragment reads off each other's banks) into a ring of
//     four slots, three stages ahead of the arithmetic;
//   * the raw x tile of the next K-step is requested at the first stage of the current one, also by DMA: every wave
//     fetches exactly the 2 KiB its own lanes will read back (no register is the target of a load in flight, so
//     nothing the compiler does with registers between request and use can go wrong, and no barrier is needed);
//   * while the MFMAs of stage i run, the same wave prepares the x operand of stage i+1 (prologue of the next source)
//     in the other of two LDS buffers; one barrier per stage, which waits for this wave's DMA of the stage and for
//     LDS traffic only.
// Every vector-memory instruction of the loop is a DMA issued from inline assembly and waited for by a counted
// s_waitcnt (the compiler does not see the DMAs).
// Counting rule.  Every stage head issues one dz tile (ZMIN or ZMIN + 1 DMA instructions per wave; past the last stage
// the last tile is requested again, so the number never drops) and the first stage of a K-step two x DMAs after it.
// A stage waits for its own tile, issued three stages earlier: younger than it are two tiles (>= 2 * ZMIN operations)
// and, in the stages 1..3 of a K-step, the two x loads.  A wait count below the true number only waits longer.
// Fragments are fetched with the transposing LDS read (ds_read_b64_tr_b16) from tiles kept in their natural
// [pixel][channel] layout, as in the single-layer kernel.  Sources are ordered by decreasing cin; a column tile past
// a source's cin skips that source altogether.  Pixel rows past the block's range are zeroed on the x side (the DMA
// fetches a repeated valid row there).
#include "common.h"
#include <stdlib.h>

#ifndef WGG_ABL
#define WGG_ABL 0                // diagnostic builds (timing only, results wrong): 1 no fragment reads / MFMA, 2 no x prologue / LDS
                                 // store, 3 no dz DMA, 4 no x DMA, 5 no barrier
#endif
#ifndef WGG_XPOLICY
#define WGG_XPOLICY ""           // cache policy suffix of the x stream (read once per launch): "" or " nt"
#endif

namespace {

constexpr int BKP = 64;          // pixels per K-step
constexpr int BC = 128;          // slab channels per block
constexpr int PADE = 8;          // LDS row padding in elements
constexpr int NT = 512, NW = 8, WQ = 4, WC = 2;
constexpr int TC = BC / WC, MC = TC / 16;
constexpr int XS = BC + PADE;
constexpr int NZ = 4, DEPTH = 3;  // dz ring slots, stages in flight

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p0, const int half_stride) {
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p0 + half_stride));
    const s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    return __builtin_bit_cast(bf16x8, v);
}

struct WGArgs {
    const u16* x;
    const u16* dz[4];
    const float* sc[4];
    const float* sh[4];
    float* dw[4];
    int cin[4];
    int ldx, lddz, cout, relu, M, nC, chunk, nsrc;
};

// wait until at most N vector-memory operations of this wave are outstanding; its LDS operations are retired too
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" : : "n"(N) : "memory");
}

template <int BQ, int NSRC>
__global__ __launch_bounds__(NT, 2) void wgrad_group_kernel(const WGArgs a) {
    constexpr int ZCH = BQ / 8;                       // data chunks per dz row
    constexpr int ZRC = ZCH + 1;                      // chunks per LDS row (one of padding)
    constexpr int ZS = ZRC * 8;                       // LDS row stride in elements
    constexpr int Z_CHUNKS = BKP * ZRC;               // 1600 (BQ 192) or 1088 (BQ 128): multiples of 64
    constexpr int Z_INSTR = Z_CHUNKS / 64;            // DMA wave-instructions per tile
    constexpr int Z_PER_WAVE = (Z_INSTR + NW - 1) / NW;
    constexpr int Z_BYTES = Z_CHUNKS * 16, X_BYTES = BKP * XS * 2;
    constexpr int TQ = BQ / WQ, MQ = TQ / 16;
    static_assert(Z_CHUNKS % 64 == 0 && TQ % 16 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* zring = smem;
    unsigned char* xbuf = smem + NZ * Z_BYTES;
    float* tab = reinterpret_cast<float*>(smem + NZ * Z_BYTES + 2 * X_BYTES);      // [NSRC][scale 128 | shift 128]
    unsigned char* xraw = smem + NZ * Z_BYTES + 2 * X_BYTES + NSRC * 2 * BC * 4;   // [wave][2][64 lanes x 16 B] raw x rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int ct = bid % a.nC;
    const int split = bid / a.nC;
    const int c_base = ct * BC;
    const int M = a.M;
    const int m_begin = split * a.chunk;
    const int m_end = min(M, m_begin + a.chunk);
    const int ksteps = (m_end - m_begin + BKP - 1) / BKP;
    int nact = 0;
#pragma unroll
    for (int s = 0; s < NSRC; ++s) nact += (s < a.nsrc && c_base < a.cin[s]) ? 1 : 0;
    if (nact == 0 || ksteps <= 0) return;
    const int nstage = ksteps * nact;

    for (int i = tid; i < NSRC * BC; i += NT) {
        const int s = i / BC, c = i - s * BC;
        const bool ok = s < a.nsrc && c_base + c < a.cin[s];
        tab[s * 2 * BC + c] = ok ? a.sc[s][c_base + c] : 0.f;
        tab[s * 2 * BC + BC + c] = ok ? a.sh[s][c_base + c] : 0.f;
    }

    // ---- dz DMA: wave w issues instructions t*8 + w (t < Z_PER_WAVE) of the tile's Z_INSTR; LDS chunk p = instr*64 + lane
    //      holds (row p / ZRC, data chunk p % ZRC), the padding chunk repeats the row's last data chunk
    const uint32_t zring_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)zring);
    //      Row and chunk of a lane's Z_PER_WAVE instructions are recomputed at every request (a division by a constant,
    //      from an opaque copy of the lane id so that the compiler does not hoist them back out of the K-loop) instead of
    //      being kept in eight registers: with four sources of 192 output channels the accumulators alone take 192, one
    //      accumulator tile went to scratch, and its reload inside the K-loop -- waited for with vmcnt(0) -- drained the DMA
    //      queue at every stage.  tools/check_isa.py reports scratch accesses and vmcnt(0) waits inside MFMA loops.
    const int zch_ok = (a.cout + 7) / 8;              // data chunks that exist in a dz row
    constexpr int ZMIN = Z_INSTR / NW;                   // DMA instructions every wave issues per tile (some issue one more)
    int d_stage = 0, d_k = 0, d_s = 0;                   // DMA cursor; frozen on the last tile once it is reached
    auto issue_z = [&]() {
        // (rows are counted from the block's first row, whose address is a 64-bit scalar: the dz tensors may exceed 4 GiB)
        const u16* base = a.dz[d_s] + (size_t)(uint32_t)m_begin * a.lddz;
        const int m0 = d_k * BKP;
        const int m_lim = M - 1 - m_begin;
        const uint32_t slot_l = zring_l + (uint32_t)(d_stage % NZ) * Z_BYTES;
        int lane_z = lane;                            // opaque copy: otherwise the rows and chunks below are hoisted out of the
        asm volatile("" : "+v"(lane_z));              // K-loop as eight loop-invariant registers
#pragma unroll
        for (int t = 0; t < Z_PER_WAVE; ++t) {
            if (WGG_ABL != 3 && t * NW + wave < Z_INSTR) {
                const int p = (t * NW + wave) * 64 + lane_z;
                int row = p / ZRC;
                int cc = p - row * ZRC;
                cc = cc < zch_ok ? cc : zch_ok - 1;
                row = row < BKP ? row : BKP - 1;
                int m = m0 + row;
                m = m < m_lim ? m : m_lim;
                const uint32_t off = (uint32_t)m * (uint32_t)a.lddz * 2u + (uint32_t)cc * 16u;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                             : : "s"(slot_l + (uint32_t)(t * NW + wave) * 1024u), "v"(off), "s"(base) : "memory");
            }
        }
        ++d_stage;
        if (d_stage < nstage) {
            if (++d_s == nact) {
                d_s = 0;
                ++d_k;
            }
        }
    };

    // ---- x staging: 16 chunks per row, rows xr0 and xr0 + 32 of the K-step; this wave's lanes own rows 4w..4w+3 and
    //      32+4w..32+4w+3, fetched by two DMA instructions into the wave's private 2 KiB
    const int xc = tid & 15, xr0 = tid >> 4;
    const bool xc_ok = c_base + xc * 8 < a.cin[0];
    const u16* xg = a.x + (xc_ok ? c_base + xc * 8 : 0);
    const uint32_t relu_floor = a.relu ? 0u : 0x80008000u;
    unsigned char* xraw_w = xraw + wave * 2048;
    const uint32_t xraw_l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(lds_ptr_t)xraw_w);
    auto issue_x = [&](const int step) {
        const int m0 = m_begin + (step < ksteps ? step : ksteps - 1) * BKP;
        int ma = m0 + xr0, mb = m0 + xr0 + 32;
        ma = ma < M ? ma : M - 1;
        mb = mb < M ? mb : M - 1;
        const u16* pa = xg + (int64_t)ma * a.ldx;
        const u16* pb = xg + (int64_t)mb * a.ldx;
        if (WGG_ABL == 4) return;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" WGG_XPOLICY : : "s"(xraw_l), "v"(pa) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" WGG_XPOLICY : : "s"(xraw_l + 1024u), "v"(pb) : "memory");
    };
    uint4 xc0, xc1;                                      // raw x rows of the current K-step
    auto store_x = [&](const int s, const int step, const uint4& x0, const uint4& x1, unsigned char* Xs) {
        if (WGG_ABL == 2) return;
        float sc[8], sh[8];
        load8f(tab + s * 2 * BC + xc * 8, sc);
        load8f(tab + s * 2 * BC + BC + xc * 8, sh);
        const int m0 = m_begin + step * BKP;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float f[8];
            unpack8(i ? x1 : x0, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
            uint4 v = relu_bf8(pack8(f), relu_floor);             // ReLU after the rounding (packed 16-bit max)
            const bool ok = m0 + xr0 + 32 * i < m_end;            // rows past the range: zero (their dz rows are repeats)
            v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
            *reinterpret_cast<uint4*>(Xs + ((xr0 + 32 * i) * XS + xc * 8) * 2) = v;
        }
    };

    // ---- fragment addressing (see wgrad_mfma.hip): lane (g = lane/16, i16 = lane%16) supplies row g*8 + i16/4
    // (+4 for the second half), columns 4*(i16%4).. of a 16-channel block and receives channel i16, pixels g*8..g*8+7
    const int g16 = lane >> 4, i16 = lane & 15;
    const int wq = wave % WQ, wc = wave / WQ;
    const int zfrag = ((g16 * 8 + (i16 >> 2)) * ZS + wq * TQ + (i16 & 3) * 4) * 2;
    const int xfrag = ((g16 * 8 + (i16 >> 2)) * XS + wc * TC + (i16 & 3) * 4) * 2;
    f32x4 acc[NSRC][MQ][MC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int j = 0; j < MC; ++j) acc[s][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto mma = [&](f32x4 (&ac)[MQ][MC], const unsigned char* Zs, const unsigned char* Xs) {
#pragma unroll
        for (int ks = 0; ks < (WGG_ABL == 1 ? 0 : 2); ++ks) {
            bf16x8 zf[MQ];
#pragma unroll
            for (int i = 0; i < MQ; ++i) zf[i] = tr_frag(Zs + zfrag + (ks * 32 * ZS + i * 16) * 2, 4 * ZS * 2);
            // the x fragment of column block j + 1 is requested before the MFMAs of block j are issued
            bf16x8 xf = tr_frag(Xs + xfrag + (ks * 32 * XS) * 2, 4 * XS * 2);
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                bf16x8 xn = xf;
                if (j + 1 < MC) xn = tr_frag(Xs + xfrag + (ks * 32 * XS + (j + 1) * 16) * 2, 4 * XS * 2);
#pragma unroll
                for (int i = 0; i < MQ; ++i)
                    ac[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[i], xf, ac[i][j], 0, 0, 0);
                xf = xn;
            }
        }
    };

    // ---- prologue: x of the first K-step, DEPTH dz tiles in flight
    issue_x(0);
    for (int i = 0; i < DEPTH; ++i) issue_z();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // tables visible
    asm volatile("" ::: "memory");

    int stage = 0;
    for (int k = 0; k < ksteps; ++k) {
        // x of this K-step: requested at the first stage of the previous one (first one: in the prologue, before three
        // tiles); the tiles of that K-step's further stages are younger
        if (nact >= 4) wait_vm<3 * ZMIN>();
        else if (nact == 3) wait_vm<2 * ZMIN>();
        else if (nact == 2) wait_vm<ZMIN>();
        else wait_vm<0>();
        xc0 = *reinterpret_cast<const uint4*>(xraw_w + lane * 16);
        xc1 = *reinterpret_cast<const uint4*>(xraw_w + 1024 + lane * 16);
        store_x(0, k, xc0, xc1, xbuf + (stage & 1) * X_BYTES);
        // ---- first source.  Stage head: this wave's share of the stage's dz tile has landed and its x operand is
        // written; after the barrier that holds for every wave, and every wave is done reading the previous stage's
        // buffers, so the ring slot of the previous stage takes the tile three stages ahead
        wait_vm<2 * ZMIN>();
        if (WGG_ABL != 5) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_z();
        issue_x(k + 1);                                          // past the last K-step: the last one again, never used
        mma(acc[0], zring + (stage % NZ) * Z_BYTES, xbuf + (stage & 1) * X_BYTES);
        if (1 < nact) store_x(1, k, xc0, xc1, xbuf + ((stage + 1) & 1) * X_BYTES);
        ++stage;
        // ---- further sources
#pragma unroll
        for (int s = 1; s < NSRC; ++s) {
            if (s < nact) {
                wait_vm<2 * ZMIN + 2>();
                if (WGG_ABL != 5) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_z();
                mma(acc[s], zring + (stage % NZ) * Z_BYTES, xbuf + (stage & 1) * X_BYTES);
                if (s + 1 < nact) store_x(s + 1, k, xc0, xc1, xbuf + ((stage + 1) & 1) * X_BYTES);
                ++stage;
            }
        }
    }
    // nothing of this block may land in LDS after it has gone (the cursor never runs past the last stage, so this
    // only retires what the last stages used)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- epilogue: acc[s][i][j][r] is (q = q0 + i*16 + fc*4 + r, c = c0 + j*16 + fr)
    const int fr = lane & 15, fc = lane >> 4;
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        if (s < nact) {
            const int cin = a.cin[s];
            float* dw = a.dw[s];
#pragma unroll
            for (int i = 0; i < MQ; ++i) {
#pragma unroll
                for (int j = 0; j < MC; ++j) {
                    const int c = c_base + wc * TC + j * 16 + fr;
                    if (c >= cin) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int q = wq * TQ + i * 16 + fc * 4 + r;
                        if (q < a.cout) atomicAdd(dw + (int64_t)q * cin + c, acc[s][i][j][r]);
                    }
                }
            }
        }
    }
}

template <int BQ, int NSRC>
int launch_group(const WGArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)NZ * BKP * (BQ / 8 + 1) * 16 + 2 * (size_t)BKP * XS * 2 + (size_t)NSRC * 2 * BC * 4 + (size_t)NW * 2048;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = wgrad_group_kernel<BQ, NSRC>;
    dsaspp_lds_limit(reinterpret_cast<const void*>(kern), (int)lds);
    const int ksplit = (a.M + a.chunk - 1) / a.chunk;
    hipLaunchKernelGGL(kern, dim3(a.nC * ksplit), dim3(NT), lds, s, a);
    DS_LAUNCH_CHECK("conv_wgrad(grouped)");
    return 0;
}

}  // namespace

// Called by dsaspp_conv_wgrad for descriptors with n_src >= 2 (after the common argument checks).
int dsaspp_wgrad_group_launch(const dsaspp_wgrad_desc& d, hipStream_t s) {
    const int n = d.n_src;
    DS_REQUIRE(n >= 1 && n <= 4, "conv_wgrad(grouped): n_src must be in 1..4 (got %d)", n);
    DS_REQUIRE(d.taps == 1 && !d.in_mask, "conv_wgrad(grouped): 1x1 layers without a channel mask only");
    DS_REQUIRE(d.cout <= 192, "conv_wgrad(grouped): cout must be at most 192 (got %d)", d.cout);
    DS_REQUIRE(d.in_relu == 0 || d.in_relu == 1, "conv_wgrad(grouped): in_relu must be 0 or 1");
    const void* dzs[4] = {d.dz, d.dz1, d.dz2, d.dz3};
    float* dws[4] = {d.dw, d.dw1, d.dw2, d.dw3};
    const float* scs[4] = {d.in_scale, d.sc1, d.sc2, d.sc3};
    const float* shs[4] = {d.in_shift, d.sh1, d.sh2, d.sh3};
    const int cins[4] = {d.cin, d.cin1, d.cin2, d.cin3};
    int order[4] = {0, 1, 2, 3};
    for (int i = 0; i < n; ++i) {
        DS_REQUIRE(dzs[i] && dws[i] && scs[i] && shs[i], "conv_wgrad(grouped): source %d incomplete", i);
        DS_REQUIRE(cins[i] > 0 && cins[i] % 8 == 0 && cins[i] <= d.ldx, "conv_wgrad(grouped): source %d: bad cin %d", i, cins[i]);
        DS_REQUIRE(((uintptr_t)dzs[i] & 15) == 0, "conv_wgrad(grouped): operands must be 16-byte aligned");
    }
    for (int i = 1; i < n; ++i)                      // widest source first
        for (int j = i; j > 0 && cins[order[j]] > cins[order[j - 1]]; --j) {
            const int t = order[j];
            order[j] = order[j - 1];
            order[j - 1] = t;
        }
    WGArgs a;
    for (int i = 0; i < 4; ++i) {
        const int k = order[i < n ? i : 0];
        a.dz[i] = reinterpret_cast<const u16*>(dzs[k]);
        a.dw[i] = dws[k];
        a.sc[i] = scs[k];
        a.sh[i] = shs[k];
        a.cin[i] = i < n ? cins[k] : 0;
    }
    a.x = reinterpret_cast<const u16*>(d.x);
    a.ldx = d.ldx;
    a.lddz = d.lddz;
    a.cout = d.cout;
    a.relu = d.in_relu;
    const int64_t M64 = (int64_t)d.n_img * d.H * d.W;
    DS_REQUIRE(M64 < (1ll << 31) - 4096, "conv_wgrad(grouped): too many pixels");
    a.M = (int)M64;
    a.nC = (a.cin[0] + BC - 1) / BC;
    a.nsrc = n;
    const int n_cu = dsaspp_persistent_cus();
    // one resident block per CU: as many pixel ranges as fit in one round, each at least 4 K-steps long
    int ksplit = d.ksplit > 0 ? d.ksplit : n_cu / a.nC;
    const int max_split = (a.M + 4 * BKP - 1) / (4 * BKP);
    if (ksplit > max_split) ksplit = max_split;
    if (ksplit < 1) ksplit = 1;
    int chunk = (a.M + ksplit - 1) / ksplit;
    a.chunk = (chunk + BKP - 1) / BKP * BKP;
    // dz rows are addressed by 32-bit byte offsets from the block's first row
    DS_REQUIRE(((int64_t)a.chunk + BKP) * d.lddz * 2 < (1ll << 32), "conv_wgrad(grouped): pixel range of a block too long (%d rows)", a.chunk);
    const bool wide = d.cout > 128;
#define WGG_PICK(N_)                                                 \
    return wide ? launch_group<192, N_>(a, s) : launch_group<128, N_>(a, s)
    if (n == 1) WGG_PICK(1);
    if (n == 2) WGG_PICK(2);
    if (n == 3) WGG_PICK(3);
    WGG_PICK(4);
#undef WGG_PICK
}
