// Streaming form of the BN-ReLU-1x1 convolution of a dense layer (forward): y[M][cout] = relu(x*scale+shift) W^T
// with few output channels (cout <= 192) and a long K (the dense block's channel prefix).
//
// With N <= 192 this GEMM is bound by streaming the activation operand from HBM (192 flop per byte of x), so the
// kernel is organised around that stream instead of around a shared activation tile:
//   * a wave owns 32 pixel rows and ALL output channels; its activation fragments go global -> registers directly
//     in the MFMA operand layout (lane = pixel row + 8 consecutive channels = one 16-byte load; the four lanes that
//     share a row cover 64 contiguous bytes), the BatchNorm + ReLU prologue is applied in registers, once per
//     element, and no activation byte ever touches LDS;
//   * only the weight tile (BQ x 64 per K-step), which every wave needs, goes through LDS: global -> LDS by DMA
//     into a three-slot ring, two K-steps ahead, one block barrier per K-step;
//   * activation loads run two K-steps ahead.  Both they and the DMAs are issued from inline assembly and waited
//     for by hand: the compiler guards every LDS read that follows a DMA it knows about with vmcnt(0) and loses
//     count of register prefetches across a loop back edge, either of which drains the stream.  Every K-step issues
//     the same number of vector-memory operations (dummy ones past the end of K), in the order [weight DMAs of step
//     k+2][activations of step k+2], so at the top of step k `s_waitcnt vmcnt(VM_PER_STEP)` is exactly "the operands
//     of step k have landed".
#include <cstdlib>

#include "common.h"

namespace {

constexpr int BK = 64;                  // channels per K-step: one 128-byte line per pixel row
#ifndef DSASPP_STREAM_NWAVE
#define DSASPP_STREAM_NWAVE 8
#endif
#ifndef DSASPP_STREAM_NSLOT
#define DSASPP_STREAM_NSLOT 3
#endif
constexpr int NWAVE = DSASPP_STREAM_NWAVE;
constexpr int NT = 64 * NWAVE;
constexpr int WAVE_ROWS = 32;           // pixel rows per wave (two MFMA row tiles)
constexpr int BP = NWAVE * WAVE_ROWS;   // 256 pixel rows per block
constexpr int MI = WAVE_ROWS / 16;
constexpr int NSLOT = DSASPP_STREAM_NSLOT;     // weight ring; the DMA runs NSLOT-1 K-steps ahead
constexpr int WLEAD = NSLOT - 1;
#ifndef DSASPP_ABL
#define DSASPP_ABL 0      // diagnostic builds: 1 no MFMA, 2 no prologue arithmetic, 3 no activation loads, 4 no weight DMA / barrier
#endif
#ifndef DSASPP_STREAM_WG
#define DSASPP_STREAM_WG 2
#endif

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ int swz8(int row) { return (row >> 1) & 7; }

template <int BQ>
__global__ __launch_bounds__(NT) void conv1x1_stream_kernel(const dsaspp_conv_desc d, const int M, const int nblocks) {
    constexpr int NJ = BQ / 16;                    // MFMA column tiles (output channels)
    constexpr int W_BYTES = BQ * 128;
    constexpr int WR = BQ / (8 * NWAVE);           // weight DMA instructions per wave and K-step
    static_assert(BQ % (8 * NWAVE) == 0, "weight rows per DMA pass");
    // vector-memory operations younger than the operands of the current step when it starts
    constexpr int VM_PER_STEP = (WLEAD - 1) * WR + 2 * MI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ws = smem;
    float* tab = reinterpret_cast<float*>(smem + NSLOT * W_BYTES);      // [2][cin] scale | shift, then [2][BQ] sums

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, nblocks);
    const int q_base = blockIdx.y * BQ;
    const int cin = d.cin, cout = d.cout;
    const int ksteps = (cin + BK - 1) / BK;
    for (int c = tid; c < cin; c += NT) {
        tab[c] = d.in_scale[c];
        tab[cin + c] = d.in_shift[c];
    }
    float* sums = tab + 2 * cin;
    for (int c = tid; c < 2 * BQ; c += NT) sums[c] = 0.f;

    // ---- activation fragments: lane (fr = lane % 16, g = lane / 16) holds row fr of a 16-row tile, channels g*8..+7
    const int fr = lane & 15, g = lane >> 4;
    const int p0 = bid * BP + wave * WAVE_ROWS;
    const u16* xg = reinterpret_cast<const u16*>(d.x);
    const u16* rowp[MI];
    bool rok[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = p0 + i * 16 + fr;
        rok[i] = m < M;
        rowp[i] = xg + (int64_t)(rok[i] ? m : 0) * d.ldx + g * 8;
    }
    // ---- weight DMA: wave-instruction i of this wave fills LDS rows (i*8 + wave)*8 .. +7, lane-linear
    const u16* wg = reinterpret_cast<const u16*>(d.w);
    const u16* w_ptr[WR];
    int wd_chunk[WR];
#pragma unroll
    for (int i = 0; i < WR; ++i) {
        const int row = (i * NWAVE + wave) * 8 + (lane >> 3);
        // LDS row (tile 2 P + h, row 4 gq + e) holds output channel 32 P + 8 gq + 4 h + e: a lane of the accumulator layout then
        // owns eight consecutive channels in two tiles, and the epilogue stores 16-byte pieces (64 consecutive bytes per pixel
        // and instruction instead of 32)
        const int tile = row >> 4, rw = row & 15;
        int q = q_base + (tile >> 1) * 32 + (rw >> 2) * 8 + (tile & 1) * 4 + (rw & 3);
        q = q < cout ? q : cout - 1;               // rows past cout: any finite data, their results are never stored
        w_ptr[i] = wg + (int64_t)q * cin;
        wd_chunk[i] = ((lane & 7) ^ swz8(row)) * 8;   // the XOR swizzle lives in the source address
    }

    u32x4 areg[2][MI][2];                          // [stage][row tile][k half]
    auto issueA = [&](const int st, const int k0) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = k0 + h * 32 + g * 8;
                const u16* p = rowp[i] + (c < cin ? k0 + h * 32 : -g * 8);      // K tail / dummy stage: channel 0 of the row
                if (DSASPP_ABL == 3) { areg[st][i][h] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(areg[st][i][h]) : "v"(p)); }
                else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(areg[st][i][h]) : "v"(p));
            }
    };
    auto issueW = [&](const int slot, const int k0) {
        typedef __attribute__((address_space(3))) void* lds_p;
#pragma unroll
        for (int i = 0; i < WR; ++i) {
            const int cc = k0 + wd_chunk[i];
            const u16* p = w_ptr[i] + (cc < cin ? cc : 0);       // K tail: the activation chunk is zeroed, any finite weight will do
            const uint32_t la = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(lds_p)(Ws + slot * W_BYTES + (i * NWAVE + wave) * 1024));
            if (DSASPP_ABL == 4) asm volatile("" : : "s"(la), "v"(p) : "memory");
            else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(la), "v"(p) : "memory");
        }
    };
    static_assert(NSLOT == 2 || NSLOT == 3, "ring depth");
    static_assert(MI == 2, "the wait statement ties four staged registers");
    auto await_stage = [&](const int st) {
        asm volatile("s_waitcnt vmcnt(%4)"
                     : "+v"(areg[st][0][0]), "+v"(areg[st][0][1]), "+v"(areg[st][1][0]), "+v"(areg[st][1][1])
                     : "n"(DSASPP_ABL == 3 ? WR : DSASPP_ABL == 4 ? 2 * MI : VM_PER_STEP)
                     : "memory");
    };

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prologue: steps 0 and 1 in flight (step 1 may be a dummy)
    if (WLEAD == 1) {          // in-flight order the loop maintains: [A(k)][W(k)][A(k+1)]
        issueA(0, 0);
        issueW(0, 0);
        issueA(1, BK);
    } else {                   // [W(k)][A(k)][W(k+1)][A(k+1)]
        issueW(0, 0);
        issueA(0, 0);
        issueW(1, BK);
        issueA(1, BK);
    }
    __syncthreads();                               // scale/shift table visible

    const int wfrag0 = fr * 128, sw = swz8(fr);
    int slot = 0;
    for (int s = 0; s < ksteps; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int ss = s + u;
            if (ss < ksteps) {
                const int k0 = ss * BK;
                await_stage(u);                    // this wave's activations and weight DMAs of step ss have landed
                // every wave's DMAs of step ss have landed, and every wave has left step ss-1, whose slot is refilled now
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (DSASPP_ABL != 4) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const int slot2 = slot >= 1 ? slot - 1 : NSLOT - 1;      // (slot + WLEAD) % NSLOT
                issueW(slot2, k0 + WLEAD * BK);
                // ---- BatchNorm + ReLU on the staged fragments, in registers
                bf16x8 af[MI][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = k0 + h * 32 + g * 8;
                    const bool cvalid = c < cin;
                    float sc[8], sh[8];
                    load8f(tab + (cvalid ? c : 0), sc);
                    load8f(tab + cin + (cvalid ? c : 0), sh);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const u32x4 r = areg[u][i][h];
                        float f[8];
                        unpack8(uint4{r.x, r.y, r.z, r.w}, f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
                        uint4 v = relu_bf8(pack8(f), 0u);
                        if (DSASPP_ABL == 2) v = uint4{r.x, r.y, r.z, r.w};
                        const bool ok = cvalid && rok[i];   // zero padding (K tail, rows past M) comes after the prologue
                        v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u; v.z = ok ? v.z : 0u; v.w = ok ? v.w : 0u;
                        af[i][h] = __builtin_bit_cast(bf16x8, v);
                    }
                }
                issueA(u, k0 + 2 * BK);            // the stage's registers are free again: step ss+2
                // ---- MFMAs: weight fragments from the LDS slot of this step
                const unsigned char* Wc = Ws + slot * W_BYTES;
                // Weight fragments are read in groups of WG, one group ahead of the MFMAs that use them (explicit
                // double buffer, order pinned with scheduling barriers): a wave then has LDS reads in flight while
                // its matrix pipe is busy -- two waves per SIMD cannot hide an LDS round trip per pair of MFMAs.
                constexpr int WG = DSASPP_STREAM_WG;
                constexpr int NG = 2 * NJ / WG;                   // groups per K-step (both k halves)
                static_assert(NJ % WG == 0, "fragment group size");
                bf16x8 wf[2][WG];
                auto read_group = [&](const int n, bf16x8 (&dst)[WG]) {
                    const int h = n / (NJ / WG), j0 = (n % (NJ / WG)) * WG;
#pragma unroll
                    for (int j = 0; j < WG; ++j)
                        dst[j] = *reinterpret_cast<const bf16x8*>(Wc + (j0 + j) * 16 * 128 + wfrag0 + (((h * 4 + g) ^ sw) << 4));
                };
                read_group(0, wf[0]);
#pragma unroll
                for (int n = 0; n < NG; ++n) {
                    if (n + 1 < NG) read_group(n + 1, wf[(n + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    const int h = n / (NJ / WG), j0 = (n % (NJ / WG)) * WG;
#pragma unroll
                    for (int j = 0; j < WG; ++j)
#pragma unroll
                        for (int i = 0; i < MI; ++i)
                            if (DSASPP_ABL == 1) acc[i][j0 + j][0] += __builtin_bit_cast(f32x4, wf[n & 1][j])[i] * __builtin_bit_cast(f32x4, af[i][h])[0];
                            else acc[i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n & 1][j], af[i][h], acc[i][j0 + j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                slot = slot + 1 < NSLOT ? slot + 1 : 0;
            }
        }
    }
    // The stages issued past the end of K are still in flight.  Their destination registers must stay reserved
    // until they have landed -- a register the compiler considers dead would be handed to the epilogue and then
    // overwritten by the late load -- so every staged register is an operand of the draining wait.
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(areg[0][0][0]), "+v"(areg[0][0][1]), "+v"(areg[0][1][0]), "+v"(areg[0][1][1]),
                   "+v"(areg[1][0][0]), "+v"(areg[1][0][1]), "+v"(areg[1][1][0]), "+v"(areg[1][1][1])
                 :
                 : "memory");

    // ---- epilogue: lane holds output channels q_base + 32 P + 8 g .. + 7 (tiles 2 P and 2 P + 1) of pixel rows i*16 + fr
    u16* yg = reinterpret_cast<u16*>(d.y);
    const bool want_stats = d.stat_acc != nullptr;
    static_assert(NJ % 2 == 0, "channel tiles are paired");
#pragma unroll
    for (int pr = 0; pr < NJ / 2; ++pr) {
        const int qo = pr * 32 + g * 8, q = q_base + qo;
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = p0 + i * 16 + fr;
            uint32_t o[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) o[h] = pack_bf2(acc[i][2 * pr + (h >> 1)][2 * (h & 1)], acc[i][2 * pr + (h >> 1)][2 * (h & 1) + 1]);
            if (rok[i] && q < cout) *reinterpret_cast<uint4*>(yg + (int64_t)m * d.ldy + q) = make_uint4(o[0], o[1], o[2], o[3]);
            if (want_stats) {      // statistics of the values as stored (rows past M contribute exact zeros)
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const float v0 = bf_lo(o[h]), v1 = bf_hi(o[h]);
                    s1[2 * h] += v0; s1[2 * h + 1] += v1;
                    s2[2 * h] += v0 * v0; s2[2 * h + 1] += v1 * v1;
                }
            }
        }
        if (want_stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t1 = row16_sum(s1[e]), t2 = row16_sum(s2[e]);
                if (fr == 0) {
                    atomicAdd(sums + qo + e, t1);
                    atomicAdd(sums + BQ + qo + e, t2);
                }
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        float* row = d.stat_acc + (size_t)(blockIdx.x % DS_R) * 2 * cout;
        for (int c = tid; c < 2 * BQ; c += NT) {
            const int sidx = c / BQ, q = q_base + (c - sidx * BQ);
            if (q < cout) atomicAdd(row + sidx * cout + q, sums[c]);
        }
    }
}

template <int BQ>
int launch_stream(const dsaspp_conv_desc& d, hipStream_t s) {
    const int64_t M = (int64_t)d.n_img * d.H * d.W;
    const int nblocks = (int)((M + BP - 1) / BP);
    const int nq = (d.cout + BQ - 1) / BQ;
    const size_t lds = (size_t)NSLOT * BQ * 128 + (size_t)(2 * d.cin + 2 * BQ) * sizeof(float);
    auto kern = conv1x1_stream_kernel<BQ>;
    dsaspp_lds_limit(reinterpret_cast<const void*>(kern), 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(nblocks, nq), dim3(NT), lds, s, d, (int)M, nblocks);
    DS_LAUNCH_CHECK("conv1x1_stream");
    return 0;
}

}  // namespace

// -1 = the descriptor is not of this kernel's form (the caller falls back to the generic kernel), 0 = launched
int dsaspp_conv1x1_stream_launch(const dsaspp_conv_desc& d, hipStream_t s) {
    static const int enabled = [] { const char* e = getenv("DSASPP_STREAM1X1"); return e ? atoi(e) : 1; }();
    if (!enabled) return -1;
    const bool form = d.taps == 1 && d.dtype == 0 && d.in_scale && d.in_shift && d.in_relu == 1 && !d.in_mask && !d.bias &&
                      !d.out_mask && !d.y_f32 && !d.epi_relu && !d.s_acc && !d.g && d.y && d.n_src <= 1;
    if (!form) return -1;
    const int64_t M = (int64_t)d.n_img * d.H * d.W;
    if (d.cout > 192 || d.cout < 96 || d.cout % 8 || d.cin % 8 || d.cin < 64 || M >= (1ll << 31)) return -1;
    if (d.ldx % 8 || d.ldy % 8 || ((uintptr_t)d.x & 15) || ((uintptr_t)d.y & 15) || ((uintptr_t)d.w & 15)) return -1;    // 16-byte loads and stores
    const size_t lds192 = (size_t)NSLOT * 192 * 128 + (size_t)(2 * d.cin + 2 * 192) * sizeof(float);
    if (lds192 > 150 * 1024) return -1;
    if (d.cout <= 128) return launch_stream<128>(d, s);
    return launch_stream<192>(d, s);
}
