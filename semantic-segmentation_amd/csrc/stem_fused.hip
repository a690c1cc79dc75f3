// Stem of the DenseNet trunks (conv0: 7x7, stride 2, pad 3, 3 -> 64 / 96 channels) without a column matrix in HBM, gfx950.
//
// The im2col + GEMM form writes the [N*Ho*Wo][160] bf16 column matrix once (2.7 GB for sixteen 1024x2048 images) and
// reads it twice, in the forward GEMM and in the weight gradient; both passes are bound by those bytes.  Here a block
// builds the 64 x 160 column tile of one output-row segment in LDS from the 3 x 7 input row segments it draws from
// (staged once, fp32 -> bf16 on the way, zero outside the image) and feeds the MFMAs from there:
//   forward          y[64 px][cout] = col . w^T, rounded to bf16, with the batch-statistics sums of the rounded values;
//   weight gradient  dw[cout][147] += dy^T . col with dy = bf16(A*dr + B*y + C) formed on the fly (the BatchNorm-backward
//                    apply of norm0), so neither the column matrix nor dy exists in memory.
// Blocks are persistent over the output-row segments; the weight gradient keeps its [cout][160] accumulators in
// registers across all its segments and leaves them as one partial per block in a workspace that a second kernel adds up
// (no atomics: a block's output is the whole weight tensor, atomics would be grid x 15360 adds on the same addresses).
#include "common.h"

namespace {

constexpr int SPX = 64;                  // output pixels per segment
constexpr int SSEG = 2 * SPX + 5;        // input columns a segment draws from
constexpr int NROW = 21;                 // 3 channels x 7 kernel rows
constexpr int KP = 160;                  // padded K (147 = 7*7*3 real columns, order (ky, kx, ci))
constexpr int KREAL = 147;
// LDS row lengths in 16-byte slots are == 2 (mod 4): see dgrad3x3_fused.hip (lane groups of ds_read_b128)
constexpr int ROWB = 352;                // bytes per [.][160] row: 22 slots
constexpr int PROWB = 160;               // bytes per [.][64 px] row: 10 slots
constexpr int NT = 256;

constexpr int NPV = (NROW * SSEG + NT - 1) / NT;        // patch elements per thread (11)

// Coordinates of a segment
struct Seg {
    int n, ho, wo0, npx;
    int64_t rowi;
};
__device__ __forceinline__ Seg seg_of(const int64_t item, const int wtiles, const int Ho, const int Wo) {
    Seg s;
    const int wt = (int)(item % wtiles);
    s.rowi = item / wtiles;                                 // n * Ho + ho
    s.n = (int)(s.rowi / Ho);
    s.ho = (int)(s.rowi - (int64_t)s.n * Ho);
    s.wo0 = wt * SPX;
    s.npx = min(SPX, Wo - s.wo0);
    return s;
}

// The 3 x 7 input row segments of a segment, [ci*7 + ky][SSEG] fp32 with zeros outside the image: every thread requests its
// NPV elements at once (registers), so that they are in flight together and can land behind other work
__device__ __forceinline__ void load_patch(float (&pv)[NPV], const float* x, const int64_t plane, const int H, const int W, const Seg& s) {
    const float* img = x + (int64_t)s.n * 3 * plane;
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
        const int i = threadIdx.x + u * NT;
        const int r = i / SSEG, xx = i - r * SSEG;
        const int ci = r / 7, ky = r - ci * 7;
        const int hi = 2 * s.ho - 3 + ky, wi = 2 * s.wo0 - 3 + xx;
        const bool ok = i < NROW * SSEG && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const float v = img[ok ? ci * plane + (int64_t)hi * W + wi : 0];
        pv[u] = ok ? v : 0.f;
    }
}
__device__ __forceinline__ void store_patch(float* patch, const float (&pv)[NPV]) {
#pragma unroll
    for (int u = 0; u < NPV; ++u) {
        const int i = threadIdx.x + u * NT;
        if (i < NROW * SSEG) patch[i] = pv[u];
    }
}

// offset in the patch of column k for output pixel 0 (pixel px adds 2*px); -1 for the zero columns past 147
__device__ __forceinline__ int col_base(const int k) {
    const int t = k / 3, ci = k - t * 3;
    const int ky = t / 7, kx = t - ky * 7;
    return k < KREAL ? (ci * 7 + ky) * SSEG + kx : -1;
}

template <int CS>
__global__ __launch_bounds__(NT, 2) void stem7_fwd_kernel(const dsaspp_stem7_desc d, const int wtiles, const int64_t nitems) {
    constexpr int MQ = CS / 16, NTASK = SPX * (KP / 8) / NT;       // 5 column-tile chunks per thread
    static_assert(SPX * (KP / 8) % NT == 0, "chunks per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* patch = reinterpret_cast<float*>(smem);
    unsigned char* colt = smem + ((NROW * SSEG * 4 + 15) & ~15);
    unsigned char* wlds = colt + SPX * ROWB;
    float* sred = reinterpret_cast<float*>(wlds + CS * ROWB);        // [4 waves][2][CS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int H = d.H, W = d.W, Ho = H / 2, Wo = W / 2;
    const int64_t plane = (int64_t)H * W;

    // filter rows [CS][160] bf16 -> LDS, once per block
    {
        const u16* wg = reinterpret_cast<const u16*>(d.w);
        for (int i = tid; i < CS * (KP / 8); i += NT) {
            const int row = i / (KP / 8), c = i - row * (KP / 8);
            *reinterpret_cast<uint4*>(wlds + row * ROWB + c * 16) = *reinterpret_cast<const uint4*>(wg + (size_t)row * d.ldw + c * 8);
        }
    }
    // this thread's chunks of a column tile: (pixel, 8 columns) pairs that are the same for every segment
    int t_dst[NTASK], t_src[NTASK][8];
#pragma unroll
    for (int t = 0; t < NTASK; ++t) {
        const int q = tid + t * NT;
        const int px = q / (KP / 8), ch = q - px * (KP / 8);
        t_dst[t] = px * ROWB + ch * 16;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int b = col_base(ch * 8 + e);
            t_src[t][e] = b < 0 ? -1 : b + 2 * px;
        }
    }
    float s1[MQ][4], s2[MQ][4];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[i][r] = 0.f; s2[i][r] = 0.f; }
    u16* yg = reinterpret_cast<u16*>(d.y);

    float pv[NPV];
    Seg cur = seg_of(blockIdx.x, wtiles, Ho, Wo);
    load_patch(pv, d.x, plane, H, W, cur);
    store_patch(patch, pv);
    __syncthreads();
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        // column tile of this segment from the staged rows
#pragma unroll
        for (int t = 0; t < NTASK; ++t) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = t_src[t][e] < 0 ? 0.f : patch[t_src[t][e]];
            *reinterpret_cast<uint4*>(colt + t_dst[t]) = pack8(f);
        }
        __syncthreads();                                    // tile complete; the staged rows are free
        // the next segment's rows are requested now and land behind the MFMAs and the stores
        const int64_t nxt = item + gridDim.x;
        const bool more = nxt < nitems;
        Seg nseg = cur;
        if (more) {
            nseg = seg_of(nxt, wtiles, Ho, Wo);
            load_patch(pv, d.x, plane, H, W, nseg);
        }
        f32x4 acc[MQ];
#pragma unroll
        for (int i = 0; i < MQ; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KP / 32; ++kk) {
            const bf16x8 pf = *reinterpret_cast<const bf16x8*>(colt + (wave * 16 + fr) * ROWB + kk * 64 + g * 16);
#pragma unroll
            for (int i = 0; i < MQ; ++i) {
                const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wlds + (i * 16 + fr) * ROWB + kk * 64 + g * 16);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, pf, acc[i], 0, 0, 0);
            }
        }
        // lane (fr, g) holds channels i*16 + g*4 .. +3 of pixel wave*16 + fr
        const int px = wave * 16 + fr;
        const bool ok = px < cur.npx;
        u16* yrow = yg + ((int64_t)cur.rowi * Wo + cur.wo0 + px) * d.ldy;
#pragma unroll
        for (int i = 0; i < MQ; ++i) {
            const uint2 o = make_uint2(pack_bf2(acc[i][0], acc[i][1]), pack_bf2(acc[i][2], acc[i][3]));
            if (ok) *reinterpret_cast<uint2*>(yrow + i * 16 + g * 4) = o;
            const float r[4] = {bf_lo(o.x), bf_hi(o.x), bf_lo(o.y), bf_hi(o.y)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = ok ? r[e] : 0.f;
                s1[i][e] += v;
                s2[i][e] = fmaf(v, v, s2[i][e]);
            }
        }
        if (more) store_patch(patch, pv);
        cur = nseg;
        __syncthreads();                                    // next rows staged; every wave is done reading this tile
    }
    if (d.stat_acc) {
        // sums over the 16 pixels of the lane's row group, then over the block's four waves
#pragma unroll
        for (int i = 0; i < MQ; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1[i][e] = row16_sum(s1[i][e]);
                s2[i][e] = row16_sum(s2[i][e]);
            }
        if (fr == 0) {
#pragma unroll
            for (int i = 0; i < MQ; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sred[(wave * 2 + 0) * CS + i * 16 + g * 4 + e] = s1[i][e];
                    sred[(wave * 2 + 1) * CS + i * 16 + g * 4 + e] = s2[i][e];
                }
        }
        __syncthreads();
        if (tid < 2 * CS) {
            const int which = tid / CS, c = tid - which * CS;
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += sred[(w * 2 + which) * CS + c];
            atomicAdd(d.stat_acc + (size_t)(blockIdx.x % DS_R) * 2 * CS + which * CS + c, t);
        }
    }
}

// Weight gradient.  Per segment: the column tile TRANSPOSED ([160 k][64 px]: K of this GEMM is the pixel axis) and
// dy^T ([CS][64 px]) in LDS; a wave owns the column tiles j = wave, wave + 4, wave + 8 (< 10) of the [CS][160] result.
// The next segment's input rows and its dr / y chunks are requested before the MFMAs of the current one.
template <int CS, bool DRF32>
__global__ __launch_bounds__(NT, 2) void stem7_wgrad_kernel(const dsaspp_stem7_desc d, const int wtiles, const int64_t nitems) {
    constexpr int MQ = CS / 16, NJ = 3, CG = CS / 8;
    constexpr int NDY = SPX * CG / NT;                      // (pixel, 8 channels) chunks of dy per thread: 3 (96) or 2 (64)
    constexpr int NCT = KP * (SPX / 8) / NT;                // (column, 8 pixels) chunks of the transposed tile per thread: 5
    static_assert(SPX * CG % NT == 0 && KP * (SPX / 8) % NT == 0, "chunks per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* patch = reinterpret_cast<float*>(smem);
    unsigned char* colT = smem + ((NROW * SSEG * 4 + 15) & ~15);
    unsigned char* dyT = colT + KP * PROWB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, g = lane >> 4;
    const int H = d.H, W = d.W, Ho = H / 2, Wo = W / 2;
    const int64_t plane = (int64_t)H * W;
    const u16* yy = reinterpret_cast<const u16*>(d.yref);

    f32x4 acc[MQ][NJ];
#pragma unroll
    for (int i = 0; i < MQ; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's chunks, the same for every segment
    int ct_dst[NCT], ct_src[NCT];
#pragma unroll
    for (int t = 0; t < NCT; ++t) {
        const int q = tid + t * NT;
        const int k = q / (SPX / 8), p8 = (q - k * (SPX / 8)) * 8;
        const int b = col_base(k);
        ct_dst[t] = k * PROWB + p8 * 2;
        ct_src[t] = b < 0 ? -1 : b + 2 * p8;
    }
    // raw dr / y chunks of a segment (pixels past the end of the output row: not loaded, dy = 0)
    uint4 ya[NDY];
    uint4 dra[NDY][DRF32 ? 2 : 1];
    auto load_dy = [&](const Seg& s) {
        const int64_t m0 = (int64_t)s.rowi * Wo + s.wo0;
#pragma unroll
        for (int t = 0; t < NDY; ++t) {
            const int q = tid + t * NT;
            const int px = q / CG, c = (q - px * CG) * 8;
            const int64_t m = m0 + (px < s.npx ? px : 0);
            if (DRF32) {
                const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(d.dr) + m * d.lddr + c);
                dra[t][0] = p[0];
                dra[t][DRF32 ? 1 : 0] = p[1];
            } else {
                dra[t][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const u16*>(d.dr) + m * d.lddr + c);
            }
            ya[t] = *reinterpret_cast<const uint4*>(yy + m * d.ldyref + c);
        }
    };
    auto store_dy = [&](const Seg& s) {
#pragma unroll
        for (int t = 0; t < NDY; ++t) {
            const int q = tid + t * NT;
            const int px = q / CG, c = (q - px * CG) * 8;
            float a[8], y[8], A[8], B[8], Cc[8], o[8];
            load8f(d.coefA + c, A); load8f(d.coefB + c, B); load8f(d.coefC + c, Cc);
            if (DRF32) {
                const uint4 lo = dra[t][0], hi = dra[t][DRF32 ? 1 : 0];
                a[0] = __uint_as_float(lo.x); a[1] = __uint_as_float(lo.y); a[2] = __uint_as_float(lo.z); a[3] = __uint_as_float(lo.w);
                a[4] = __uint_as_float(hi.x); a[5] = __uint_as_float(hi.y); a[6] = __uint_as_float(hi.z); a[7] = __uint_as_float(hi.w);
            } else {
                unpack8(dra[t][0], a);
            }
            unpack8(ya[t], y);
            const bool ok = px < s.npx;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = ok ? fmaf(A[e], a[e], fmaf(B[e], y[e], Cc[e])) : 0.f;
            const uint4 v = pack8(o);
            const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 8; ++e)
                *reinterpret_cast<u16*>(dyT + (c + e) * PROWB + px * 2) = (u16)(e & 1 ? w4[e >> 1] >> 16 : w4[e >> 1] & 0xffffu);
        }
    };

    float pv[NPV];
    Seg cur = seg_of(blockIdx.x, wtiles, Ho, Wo);
    load_patch(pv, d.x, plane, H, W, cur);
    load_dy(cur);
    store_patch(patch, pv);
    store_dy(cur);
    __syncthreads();
    for (int64_t item = blockIdx.x; item < nitems; item += gridDim.x) {
        // column tile, transposed: a thread takes column k and 8 consecutive pixels
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = ct_src[t] < 0 ? 0.f : patch[ct_src[t] + 2 * e];
            *reinterpret_cast<uint4*>(colT + ct_dst[t]) = pack8(f);
        }
        __syncthreads();                                    // both tiles complete; the staged rows are free
        const int64_t nxt = item + gridDim.x;
        const bool more = nxt < nitems;
        Seg nseg = cur;
        if (more) {
            nseg = seg_of(nxt, wtiles, Ho, Wo);
            load_patch(pv, d.x, plane, H, W, nseg);
            load_dy(nseg);
        }
#pragma unroll
        for (int kk = 0; kk < SPX / 32; ++kk) {
            bf16x8 zf[MQ];
#pragma unroll
            for (int i = 0; i < MQ; ++i) zf[i] = *reinterpret_cast<const bf16x8*>(dyT + (i * 16 + fr) * PROWB + kk * 64 + g * 16);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int jt = wave + 4 * j;                 // wave-uniform
                if (jt < KP / 16) {
                    const bf16x8 cf = *reinterpret_cast<const bf16x8*>(colT + (jt * 16 + fr) * PROWB + kk * 64 + g * 16);
#pragma unroll
                    for (int i = 0; i < MQ; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf[i], cf, acc[i][j], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                    // every wave is done reading dy^T and the column tile
        if (more) {
            store_patch(patch, pv);
            store_dy(nseg);
        }
        cur = nseg;
        __syncthreads();                                    // next segment staged
    }
    // acc[i][j][r]: output channel i*16 + g*4 + r, column (wave + 4j)*16 + fr
    float* part = d.ws + (size_t)blockIdx.x * CS * KP;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int jt = wave + 4 * j;
        if (jt < KP / 16) {
#pragma unroll
            for (int i = 0; i < MQ; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(i * 16 + g * 4 + r) * KP + jt * 16 + fr] = acc[i][j][r];
        }
    }
}

// dw[q][k] += sum over the blocks' partials, k < 147
__global__ __launch_bounds__(256) void stem7_wgrad_reduce_kernel(const float* ws, const int nblocks, const int cs, float* dw, const int lddw) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cs * KP) return;
    const int q = idx / KP, k = idx - q * KP;
    if (k >= KREAL) return;
    float t = 0.f;
    for (int b = 0; b < nblocks; ++b) t += ws[(size_t)b * cs * KP + idx];
    dw[(size_t)q * lddw + k] += t;
}

int check_common(const dsaspp_stem7_desc* d, const char* who) {
    DS_REQUIRE(d && d->n_img > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, "%s: H and W must be even", who);
    DS_REQUIRE(d->cout == 64 || d->cout == 96, "%s: cout must be 64 or 96 (got %d)", who, d->cout);
    DS_REQUIRE(d->x != nullptr, "%s: null input", who);
    return 0;
}

// persistent grid: a few blocks per compute unit, never more than there are segments
int grid_for(const int64_t nitems, const int per_cu) {
    int64_t g = (int64_t)dsaspp_persistent_cus() * per_cu;
    if (g > nitems) g = nitems;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int64_t dsaspp_stem7_wgrad_workspace_floats(int32_t cout) {
    // one [cout][160] partial per block of the largest grid stem7_wgrad launches (256 compute units at most)
    return (int64_t)256 * 2 * cout * KP;
}

extern "C" int dsaspp_stem7_fwd(const dsaspp_stem7_desc* d, dsaspp_stream_t stream) {
    if (int rc = check_common(d, "stem7_fwd")) return rc;
    DS_REQUIRE(d->w && d->y && d->ldw >= KP && d->ldw % 8 == 0 && (((uintptr_t)d->w | (uintptr_t)d->y) & 15) == 0, "stem7_fwd: filter rows must be 160 columns or more (16-byte aligned)");
    DS_REQUIRE(d->ldy >= d->cout && d->ldy % 4 == 0, "stem7_fwd: bad output stride");
    const int wtiles = (d->W / 2 + SPX - 1) / SPX;
    const int64_t nitems = (int64_t)d->n_img * (d->H / 2) * wtiles;
    const size_t lds = ((NROW * SSEG * 4 + 15) & ~15) + (size_t)SPX * ROWB + (size_t)d->cout * ROWB + 8 * d->cout * 4;
    const int grid = grid_for(nitems, 2);
    if (d->cout == 96) {
        dsaspp_lds_limit(reinterpret_cast<const void*>(stem7_fwd_kernel<96>), 160 * 1024);
        hipLaunchKernelGGL(stem7_fwd_kernel<96>, dim3(grid), dim3(NT), lds, (hipStream_t)stream, *d, wtiles, nitems);
    } else {
        dsaspp_lds_limit(reinterpret_cast<const void*>(stem7_fwd_kernel<64>), 160 * 1024);
        hipLaunchKernelGGL(stem7_fwd_kernel<64>, dim3(grid), dim3(NT), lds, (hipStream_t)stream, *d, wtiles, nitems);
    }
    DS_LAUNCH_CHECK("stem7_fwd");
    return 0;
}

extern "C" int dsaspp_stem7_wgrad(const dsaspp_stem7_desc* d, dsaspp_stream_t stream) {
    if (int rc = check_common(d, "stem7_wgrad")) return rc;
    DS_REQUIRE(d->dr && d->yref && d->coefA && d->coefB && d->coefC && d->dw && d->ws, "stem7_wgrad: null tensor");
    DS_REQUIRE(d->lddr >= d->cout && d->lddr % 8 == 0 && d->ldyref >= d->cout && d->ldyref % 8 == 0 && d->lddw >= KREAL, "stem7_wgrad: bad strides");
    DS_REQUIRE((((uintptr_t)d->dr | (uintptr_t)d->yref | (uintptr_t)d->coefA | (uintptr_t)d->coefB | (uintptr_t)d->coefC) & 15) == 0, "stem7_wgrad: tensors must be 16-byte aligned");
    const int wtiles = (d->W / 2 + SPX - 1) / SPX;
    const int64_t nitems = (int64_t)d->n_img * (d->H / 2) * wtiles;
    const int grid = grid_for(nitems, 2);      // two resident blocks per compute unit (250 registers, 4 waves): one round
    DS_REQUIRE(d->ws_floats >= (int64_t)grid * d->cout * KP, "stem7_wgrad: workspace of %lld floats, %lld needed (dsaspp_stem7_wgrad_workspace_floats)",
               (long long)d->ws_floats, (long long)grid * d->cout * KP);
    const size_t lds = ((NROW * SSEG * 4 + 15) & ~15) + (size_t)KP * PROWB + (size_t)d->cout * PROWB;
    hipStream_t s = (hipStream_t)stream;
#define STEM7_WG(CS_, F32_) hipLaunchKernelGGL((stem7_wgrad_kernel<CS_, F32_>), dim3(grid), dim3(NT), lds, s, *d, wtiles, nitems)
    if (d->cout == 96) { if (d->dr_f32) STEM7_WG(96, true); else STEM7_WG(96, false); }
    else { if (d->dr_f32) STEM7_WG(64, true); else STEM7_WG(64, false); }
#undef STEM7_WG
    DS_LAUNCH_CHECK("stem7_wgrad");
    const int n = d->cout * KP;
    hipLaunchKernelGGL(stem7_wgrad_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d->ws, grid, d->cout, d->dw, d->lddw);
    DS_LAUNCH_CHECK("stem7_wgrad(reduce)");
    return 0;
}
