// Bilinear x8 upsample (align_corners=False) fused with softmax cross-entropy, gfx950.
//
// The full-resolution logits (19 x H x W fp32, 159 MB per 1024x2048 image) are never written:
// a lane owns one full-resolution column of one "cell" -- the 8x8 block of output pixels that
// interpolates between the same four low-resolution corners -- walks its 8 rows, and keeps
// the x-interpolated corner rows and the per-corner-row gradient sums in registers.  The 8
// lanes of a cell are combined with DPP adds; the cells of a block meet in an LDS tile of the
// low-resolution gradient, which leaves as contiguous fp32 atomics.  HBM traffic is the
// low-resolution logits, the int64 labels and the low-resolution gradient.
#include "common.h"
#include <mutex>

namespace {

constexpr int NC_MAX = 32;

// Source coordinate of output index o (scale 8, align_corners=False), as PyTorch computes it:
// src = max(0.125*(o+0.5)-0.5, 0); i0 = floor(src); i1 = i0 + (i0 < n-1); l1 = src-i0.
__device__ __forceinline__ void src_index(int o, int n, int& i0, int& i1, float& l1) {
    float s = 0.125f * ((float)o + 0.5f) - 0.5f;
    s = fmaxf(s, 0.f);
    i0 = (int)s;
    i1 = i0 + (i0 < n - 1 ? 1 : 0);
    l1 = s - (float)i0;
}

// Cells: cell row j in [0, h] covers output rows [8j-4, 8j+4) ; same for columns.  All rows of
// a cell share (i0, i1).  A block owns a tile of UP_TCX x UP_TCY cells; the corner gradients of its
// cells are summed in an LDS copy of the (UP_TCX+1) x (UP_TCY+1) low-resolution pixels the tile
// touches and leave as one pass of contiguous fp32 atomics (pixels on the tile border are shared
// with the neighbouring blocks).
constexpr int UP_TCX = 8, UP_TCY = 4;

__device__ __forceinline__ float row8_sum(float v) {
    // sum over the 8 lanes of a cell (aligned 8-lane groups inside a DPP row), result in every lane
    int x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));   // row_half_mirror
    return v;
}

// EXACT: n_class == NC, no per-class range checks (they are scalar conditions: the compiler turns each into a
// branch and the unrolled class loops into thousands of lines of duplicated code)
template <int NC, bool EXACT>
__global__ __launch_bounds__(256) void upce_kernel(const dsaspp_ce_desc d, const int tiles_x, const int tiles_y) {
    constexpr int PX = UP_TCX + 1, PY = UP_TCY + 1, CS = NC + 1;     // LDS tile [PY][PX][CS]
    __shared__ float gt[PY * PX * CS];
    const int h = d.h, w = d.w, H = 8 * h, W = 8 * w;
    const int cells_x = w + 1, cells_y = h + 1;
    const int tid = threadIdx.x, lane8 = tid & 7;
    int b = blockIdx.x;
    const int bx = b % tiles_x; b /= tiles_x;
    const int by = b % tiles_y;
    const int n = b / tiles_y;
    const int ci = bx * UP_TCX + ((tid >> 3) & (UP_TCX - 1)), cj = by * UP_TCY + (tid >> 6);
    const bool cell_ok = ci < cells_x && cj < cells_y;
    for (int i = tid; i < PY * PX * CS; i += 256) gt[i] = 0.f;
    const int X = 8 * ci - 4 + lane8;
    const bool col_ok = cell_ok && X >= 0 && X < W;
    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
    float lx = 0.f, ly_dummy = 0.f;
    src_index(max(X, 0), w, x0, x1, lx);
    // corner rows of this cell (identical for all its rows)
    src_index(max(8 * cj - 4, 0), h, y0, y1, ly_dummy);
    const float hx = 1.f - lx;
    const float* L = d.logits + (int64_t)n * h * w * d.ldl;
    float r0[NC], r1[NC], g0[NC], g1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { r0[c] = 0.f; r1[c] = 0.f; g0[c] = 0.f; g1[c] = 0.f; }
    float loss = 0.f, cnt = 0.f;
    if (col_ok) {
        const float* p00 = L + ((int64_t)y0 * w + x0) * d.ldl;
        const float* p01 = L + ((int64_t)y0 * w + x1) * d.ldl;
        const float* p10 = L + ((int64_t)y1 * w + x0) * d.ldl;
        const float* p11 = L + ((int64_t)y1 * w + x1) * d.ldl;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // classes past n_class sit at -1e30: they never win the max and their exp() is exactly 0, so the row
            // loop needs no per-class range checks
            const bool cv = EXACT || c < d.n_class;
            r0[c] = cv ? hx * p00[cv ? c : 0] + lx * p01[cv ? c : 0] : -1.0e30f;
            r1[c] = cv ? hx * p10[cv ? c : 0] + lx * p11[cv ? c : 0] : -1.0e30f;
        }
        const int64_t* lab = d.labels + (int64_t)n * H * W;
        int tl[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {              // the 8 labels of this lane's column: loads in flight together
            const int Y = 8 * cj - 4 + r;
            tl[r] = (Y >= 0 && Y < H) ? (int)lab[(int64_t)Y * W + X] : d.ignore_index;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int Y = 8 * cj - 4 + r;
            const bool row_ok = Y >= 0 && Y < H;
            int yy0, yy1;
            float ly;
            src_index(row_ok ? Y : 0, h, yy0, yy1, ly);
            const float hy = 1.f - ly;
            const int t = tl[r];
            const float live = (t != d.ignore_index) ? 1.f : 0.f;
            // one-hot of the target as a bit mask: keeps the per-class selects plain arithmetic (a chain of
            // (c == t) selects gets lowered to a jump table over t with one copy of the row body per class)
            const unsigned tm = ((unsigned)t < 32u) ? (1u << t) : 0u;
            float z[NC];
            float mx = -3.0e38f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                z[c] = hy * r0[c] + ly * r1[c];
                mx = fmaxf(mx, z[c]);
            }
            float se = 0.f, zt = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                zt = fmaf((float)((tm >> c) & 1u), z[c], zt);
                const float e = __expf(z[c] - mx);
                z[c] = e;
                se += e;
            }
            const float inv = live / se;
            const float lse = mx + __logf(se);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float gcl = fmaf(z[c], inv, -live * (float)((tm >> c) & 1u));
                g0[c] = fmaf(hy, gcl, g0[c]);
                g1[c] = fmaf(ly, gcl, g1[c]);
            }
            loss += live * (lse - zt);
            cnt += live;
        }
    }
    // combine the 8 columns of the cell: corner (y, x0) gets the hx-weighted, (y, x1) the
    // lx-weighted sum.  Within a cell every in-range lane names the same two low-res columns
    // (lanes left of output column 4 all clamp to (0, 1) with lx = 0), taken here from the
    // cell's first in-range column.
    int cx0, cx1;
    {
        float dummy;
        src_index(max(8 * ci - 4, 0), w, cx0, cx1, dummy);
    }
    // tile-local pixel coordinates: the tile's first low-resolution pixel is (by*UP_TCY - 1, bx*UP_TCX - 1)
    const int oy = by * UP_TCY - 1, ox = bx * UP_TCX - 1;
    const int l00 = ((y0 - oy) * PX + (cx0 - ox)) * CS, l01 = ((y0 - oy) * PX + (cx1 - ox)) * CS;
    const int l10 = ((y1 - oy) * PX + (cx0 - ox)) * CS, l11 = ((y1 - oy) * PX + (cx1 - ox)) * CS;
    __syncthreads();                               // tile zeroed
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float a00 = row8_sum(hx * g0[c]), a01 = row8_sum(lx * g0[c]);
        const float a10 = row8_sum(hx * g1[c]), a11 = row8_sum(lx * g1[c]);
        // the 4*n_class additions of the cell, spread over its 8 lanes
        if (cell_ok && (EXACT || c < d.n_class) && (c & 7) == lane8) {
            atomicAdd(gt + l00 + c, a00);
            atomicAdd(gt + l01 + c, a01);
            atomicAdd(gt + l10 + c, a10);
            atomicAdd(gt + l11 + c, a11);
        }
    }
    __syncthreads();
    float* G = d.dlogits + (int64_t)n * h * w * d.ldl;
    for (int i = tid; i < PY * PX * 32; i += 256) {
        const int c = i & 31, pix = i >> 5;
        const int py = pix / PX, px = pix - py * PX;
        const int gy = oy + py, gx = ox + px;
        if (c < NC && (EXACT || c < d.n_class) && gy >= 0 && gy < h && gx >= 0 && gx < w) {
            const float v = gt[pix * CS + c];
            if (v != 0.f) atomicAdd(G + ((int64_t)gy * w + gx) * d.ldl + c, v);
        }
    }
    // loss / count: wave reduce, one atomic pair per wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        loss += __shfl_xor(loss, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if ((threadIdx.x & 63) == 0 && cnt > 0.f) {
        atomicAdd(d.accum, loss);
        atomicAdd(d.accum + 1, cnt);
    }
}

// dlogits /= count (x grad_scale); loss = sum/count; bf16 copy; per-class column sums -> dbias
__global__ __launch_bounds__(256) void ce_finalize_kernel(const dsaspp_ce_desc d) {
    const int64_t M = (int64_t)d.n_img * d.h * d.w;
    const float cnt = d.accum[1];
    const float inv = cnt > 0.f ? d.grad_scale / cnt : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.loss) d.loss[0] = cnt > 0.f ? d.accum[0] / cnt : 0.f;
    const int c = threadIdx.x & 31;
    const int rsub = threadIdx.x >> 5;     // 8 rows per pass
    float colsum = 0.f;
    u16* ob = reinterpret_cast<u16*>(d.dlogits_bf16);
    for (int64_t m = (int64_t)blockIdx.x * 8 + rsub; m < M; m += (int64_t)gridDim.x * 8) {
        float v = 0.f;
        if (c < d.ldl) {
            v = (c < d.n_class) ? d.dlogits[m * d.ldl + c] * inv : 0.f;
            d.dlogits[m * d.ldl + c] = v;
        }
        if (ob && c < d.lddb) ob[m * d.lddb + c] = (u16)f2bf_bits(v);
        colsum += v;
    }
    if (d.dbias) {
        __shared__ float red[256];
        red[threadIdx.x] = colsum;
        __syncthreads();
        if (threadIdx.x < 32) {
            float t = 0.f;
            for (int r = 0; r < 8; ++r) t += red[r * 32 + threadIdx.x];
            if (threadIdx.x < d.n_class) atomicAdd(d.dbias + threadIdx.x, t);
        }
    }
}

// ---------------------------------------------------------------- materialised upsample (forward() API)
// One thread per output pixel column strip: thread = (n, Y, X); loops classes; NCHW fp32 writes
// are contiguous in X across lanes.
__global__ __launch_bounds__(256) void up_fwd_kernel(const dsaspp_up_desc d) {
    const int H = 8 * d.h, W = 8 * d.w;
    const int64_t total = (int64_t)d.n_img * H * W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int X = (int)(idx % W);
        const int64_t t = idx / W;
        const int Y = (int)(t % H);
        const int n = (int)(t / H);
        int x0, x1, y0, y1;
        float lx, ly;
        src_index(X, d.w, x0, x1, lx);
        src_index(Y, d.h, y0, y1, ly);
        const float hx = 1.f - lx, hy = 1.f - ly;
        const float* L = d.logits + (int64_t)n * d.h * d.w * d.ldl;
        const float* p00 = L + ((int64_t)y0 * d.w + x0) * d.ldl;
        const float* p01 = L + ((int64_t)y0 * d.w + x1) * d.ldl;
        const float* p10 = L + ((int64_t)y1 * d.w + x0) * d.ldl;
        const float* p11 = L + ((int64_t)y1 * d.w + x1) * d.ldl;
        if (d.out) {
            float* o = d.out + (int64_t)n * d.n_class * H * W + (int64_t)Y * W + X;
            for (int c = 0; c < d.n_class; ++c)
                o[(int64_t)c * H * W] = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
        }
        if (d.argmax) {
            float best = -3.0e38f;
            int bi = 0;
            for (int c = 0; c < d.n_class; ++c) {
                const float v = hy * (hx * p00[c] + lx * p01[c]) + ly * (hx * p10[c] + lx * p11[c]);
                if (v > best) { best = v; bi = c; }
            }
            d.argmax[idx] = bi;
        }
    }
}

// Adjoint of the upsample: one thread per (low-res pixel, class) gathers its <= 16x16 footprint.
__global__ __launch_bounds__(256) void up_bwd_kernel(const dsaspp_up_desc d) {
    const int H = 8 * d.h, W = 8 * d.w;
    const int64_t total = (int64_t)d.n_img * d.h * d.w * d.n_class;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        // x fastest so that neighbouring lanes read neighbouring full-res columns
        const int xl = (int)(idx % d.w);
        int64_t t = idx / d.w;
        const int yl = (int)(t % d.h);
        t /= d.h;
        const int c = (int)(t % d.n_class);
        const int n = (int)(t / d.n_class);
        const float* gsrc = d.out + ((int64_t)n * d.n_class + c) * H * W;
        float acc = 0.f;
        const int Y0 = max(8 * yl - 4, 0), Y1 = min(8 * yl + 12, H);
        const int X0 = max(8 * xl - 4, 0), X1 = min(8 * xl + 12, W);
        for (int Y = Y0; Y < Y1; ++Y) {
            int y0, y1;
            float ly;
            src_index(Y, d.h, y0, y1, ly);
            float wy = 0.f;
            if (y0 == yl) wy += 1.f - ly;
            if (y1 == yl) wy += ly;
            if (wy == 0.f) continue;
            float rowacc = 0.f;
            for (int X = X0; X < X1; ++X) {
                int x0, x1;
                float lx;
                src_index(X, d.w, x0, x1, lx);
                float wx = 0.f;
                if (x0 == xl) wx += 1.f - lx;
                if (x1 == xl) wx += lx;
                if (wx != 0.f) rowacc += wx * gsrc[(int64_t)Y * W + X];
            }
            acc += wy * rowacc;
        }
        const_cast<float*>(d.logits)[(((int64_t)n * d.h + yl) * d.w + xl) * d.ldl + c] = acc;
    }
}

int check_ce(const dsaspp_ce_desc* d, const char* who) {
    DS_REQUIRE(d && d->n_img > 0 && d->h > 0 && d->w > 0, "%s: bad geometry", who);
    DS_REQUIRE(d->n_class > 0 && d->n_class <= NC_MAX && d->ldl >= d->n_class, "%s: n_class must be in [1,32] and ldl >= n_class", who);
    DS_REQUIRE(d->dlogits && d->accum, "%s: null tensor", who);
    return 0;
}

}  // namespace

extern "C" int dsaspp_upsample8_ce(const dsaspp_ce_desc* d, dsaspp_stream_t stream) {
    if (int rc = check_ce(d, "upsample8_ce")) return rc;
    DS_REQUIRE(d->logits && d->labels, "upsample8_ce: null tensor");
    const int tiles_x = (d->w + 1 + UP_TCX - 1) / UP_TCX, tiles_y = (d->h + 1 + UP_TCY - 1) / UP_TCY;
    const int64_t blocks = (int64_t)d->n_img * tiles_x * tiles_y;
    DS_REQUIRE(blocks < (1ll << 31), "upsample8_ce: grid too large");
    if (d->n_class == 19) hipLaunchKernelGGL((upce_kernel<19, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d, tiles_x, tiles_y);
    else hipLaunchKernelGGL((upce_kernel<32, false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d, tiles_x, tiles_y);
    DS_LAUNCH_CHECK("upsample8_ce");
    return 0;
}

extern "C" int dsaspp_ce_finalize(const dsaspp_ce_desc* d, dsaspp_stream_t stream) {
    if (int rc = check_ce(d, "ce_finalize")) return rc;
    DS_REQUIRE(d->ldl <= 32 && (!d->dlogits_bf16 || d->lddb <= 32), "ce_finalize: ldl/lddb must be <= 32");
    const int64_t M = (int64_t)d->n_img * d->h * d->w;
    int64_t blocks = (M + 7) / 8;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ce_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d);
    DS_LAUNCH_CHECK("ce_finalize");
    return 0;
}

// ---------------------------------------------------------------- allocation-free fused loss
// Working memory of dsaspp_upsample8_ce_loss_noalloc: the two loss accumulators (first 64 floats) and the fp32
// low-resolution gradient the cells add into.  One copy per GPU, resident from the moment the code object is loaded on
// that GPU until the process ends; every call on a GPU uses that GPU's copy.
#define NOALLOC_MAX_PIXELS DSASPP_NOALLOC_MAX_PIXELS   /* low-resolution pixels per call (four 1024x2048 images) */
#define NOALLOC_LD 32                        /* floats per pixel row of the gradient staging */
#define NOALLOC_HEAD 64                      /* floats in front of the gradient staging */
static __device__ __align__(256) float scratch[NOALLOC_HEAD + (size_t)NOALLOC_MAX_PIXELS * NOALLOC_LD];

constexpr int NOALLOC_MAX_DEVICES = 64;
static float* noalloc_ptr[NOALLOC_MAX_DEVICES];       // device address of `scratch` on each GPU, resolved once
static std::mutex noalloc_mutex;

// Device address of this GPU's copy of `scratch` (the array's name is only a host-side handle: kernels and
// hipMemsetAsync must be given the address hipGetSymbolAddress returns, looked up with that GPU current).
static int noalloc_scratch(float** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    DS_REQUIRE(e == hipSuccess, "upsample8_ce_loss_noalloc: hipGetDevice: %s", hipGetErrorString(e));
    DS_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "upsample8_ce_loss_noalloc: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(noalloc_mutex);
    if (!noalloc_ptr[dev]) {
        float* p = nullptr;
        e = hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch));
        DS_REQUIRE(e == hipSuccess && p != nullptr, "upsample8_ce_loss_noalloc: hipGetSymbolAddress: %s", hipGetErrorString(e));
        noalloc_ptr[dev] = p;
    }
    *out = noalloc_ptr[dev];
    return 0;
}

extern "C" int dsaspp_upsample8_ce_loss_noalloc(const dsaspp_ce_desc* d, dsaspp_stream_t stream) {
    DS_REQUIRE(d && d->n_img > 0 && d->h > 0 && d->w > 0, "upsample8_ce_loss_noalloc: bad geometry");
    DS_REQUIRE(d->n_class > 0 && d->n_class <= NC_MAX && d->ldl >= d->n_class,
               "upsample8_ce_loss_noalloc: n_class must be in [1,32] and ldl >= n_class");
    DS_REQUIRE(d->logits && d->labels && d->loss, "upsample8_ce_loss_noalloc: null tensor");
    DS_REQUIRE(d->ldl <= NOALLOC_LD && (!d->dlogits_bf16 || d->lddb <= 32), "upsample8_ce_loss_noalloc: ldl/lddb must be <= 32");
    const int64_t M = (int64_t)d->n_img * d->h * d->w;
    DS_REQUIRE(M <= NOALLOC_MAX_PIXELS,
               "upsample8_ce_loss_noalloc: %lld low-resolution pixels exceed the built-in working memory (%d); "
               "use dsaspp_upsample8_ce + dsaspp_ce_finalize with caller-provided buffers",
               (long long)M, NOALLOC_MAX_PIXELS);
    float* ws = nullptr;
    if (int rc = noalloc_scratch(&ws)) return rc;
    hipStream_t s = (hipStream_t)stream;
    // the array keeps the previous call's sums: clear exactly what this call adds into
    hipError_t e = hipMemsetAsync(ws, 0, (NOALLOC_HEAD + (size_t)M * d->ldl) * sizeof(float), s);
    DS_REQUIRE(e == hipSuccess, "upsample8_ce_loss_noalloc: hipMemsetAsync: %s", hipGetErrorString(e));
    dsaspp_ce_desc c = *d;
    c.accum = ws;
    c.dlogits = ws + NOALLOC_HEAD;
    if (int rc = dsaspp_upsample8_ce(&c, stream)) return rc;
    return dsaspp_ce_finalize(&c, stream);
}

static int up_common(const dsaspp_up_desc* d, const char* who) {
    DS_REQUIRE(d && d->n_img > 0 && d->h > 0 && d->w > 0 && d->n_class > 0 && d->ldl >= d->n_class && d->logits, "%s: bad descriptor", who);
    return 0;
}

extern "C" int dsaspp_upsample8_fwd(const dsaspp_up_desc* d, dsaspp_stream_t stream) {
    if (int rc = up_common(d, "upsample8_fwd")) return rc;
    DS_REQUIRE(d->out != nullptr, "upsample8_fwd: null output");
    dsaspp_up_desc e = *d;
    e.argmax = nullptr;
    const int64_t total = (int64_t)d->n_img * 64 * d->h * d->w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(up_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, e);
    DS_LAUNCH_CHECK("upsample8_fwd");
    return 0;
}

extern "C" int dsaspp_upsample8_argmax(const dsaspp_up_desc* d, dsaspp_stream_t stream) {
    if (int rc = up_common(d, "upsample8_argmax")) return rc;
    DS_REQUIRE(d->argmax != nullptr, "upsample8_argmax: null output");
    dsaspp_up_desc e = *d;
    e.out = nullptr;
    const int64_t total = (int64_t)d->n_img * 64 * d->h * d->w;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(up_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, e);
    DS_LAUNCH_CHECK("upsample8_argmax");
    return 0;
}

extern "C" int dsaspp_upsample8_bwd(const dsaspp_up_desc* d, dsaspp_stream_t stream) {
    if (int rc = up_common(d, "upsample8_bwd")) return rc;
    DS_REQUIRE(d->out != nullptr, "upsample8_bwd: null gradient input");
    const int64_t total = (int64_t)d->n_img * d->h * d->w * d->n_class;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 262144) blocks = 262144;
    hipLaunchKernelGGL(up_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *d);
    DS_LAUNCH_CHECK("upsample8_bwd");
    return 0;
}
