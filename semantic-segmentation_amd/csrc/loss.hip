// Bilinear x8 upsample (align_corners=False) fused with softmax cross-entropy, gfx950.
//
// The full-resolution logits (19 x H x W fp32, 159 MB per 1024x2048 image) are never written:
// a lane owns one full-resolution column of one "cell" -- the 8x8 block of output pixels that
// interpolates between the same four low-resolution corners -- walks its 8 rows, and keeps
// the x-interpolated corner rows and the per-corner-row gradient sums in registers.  The 8
// lanes of a cell are combined with DPP adds; each cell leaves its four corner sums in LDS (plain
// stores: LDS float atomics retire about one lane per 2.5 cycles), the block adds up the cells
// that meet in each low-resolution pixel and sends the result out as contiguous fp32 atomics.
// HBM traffic is the low-resolution logits, the int64 labels and the low-resolution gradient.
#include "common.h"
#include <mutex>

#ifndef UPCE_WAVES
#define UPCE_WAVES 3     // waves per SIMD the register budget allows; measured flat from 2 to 4 (the row loop is ALU work)
#endif
#ifndef UPCE_UNROLL
#define UPCE_UNROLL 2
#endif
#ifndef UPCE_ABL
#define UPCE_ABL 0       // diagnostic builds (timing only): 1 no per-cell LDS adds, 2 no global adds of the tile, 3 no softmax rows,
                         // 4 no loss/count atomics
#endif

// The loss and the pixel count are summed in ACC_SLOTS (loss, count) pairs, one pair per 128-byte line, block b into
// pair b % ACC_SLOTS: atomics on one address are executed one after the other at the memory side (about 13 ns each;
// 139k of them from the waves of a batch of sixteen 1024x2048 images were 1.7 of this kernel's 1.8 ms).
#ifndef UPCE_TILES_PER_BLOCK
#define UPCE_TILES_PER_BLOCK 1
#endif
#define ACC_SLOTS DSASPP_CE_ACCUM_SLOTS
#define ACC_STRIDE DSASPP_CE_ACCUM_STRIDE

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
// branch and the unrolled class loops into thousands of lines of duplicated code).
//
// The kernel is bound by vector-ALU issue in the row loop (counters: the vector ALU of every SIMD is busy for three
// quarters of the kernel), so the row body is cut down to what the softmax itself needs, per class: one multiply-add
// for the interpolated logit, the exponential, one add into the sum, two multiply-adds into the two corner-row
// gradient sums.
//  - Logits are pre-scaled by log2(e): exp and log are the bare base-2 hardware instructions.
//  - No per-row maximum: each corner row is shifted by its own maximum once per tile (r0 - M0, r1 - M1), so the
//    interpolated logit z - (hy*M0 + ly*M1) is never positive.  A row whose largest term still underflows (the two
//    corner rows peak at different classes, hundreds of units apart) is redone with its true maximum.
//  - Nothing is a select over the target class t.  The row adds its weights hy, ly to lane-private LDS sums
//    oh0[t], oh1[t]; after the rows, g -= oh gives the -onehot(t) half of the gradient and sum_c oh[c] * r[c] the
//    target-logit half of the loss.
// A block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and adds its loss and pixel count once, at the end.
template <int NC, bool EXACT>
__global__ __launch_bounds__(256, EXACT ? UPCE_WAVES : 1) void upce_kernel(const dsaspp_ce_desc d, const int tiles_x, const int tiles_y,
                                                               const int n_tiles, const int vec4) {
    constexpr int PX = UP_TCX + 1, PY = UP_TCY + 1, CS = NC + 1;
    constexpr int NCELL = UP_TCX * UP_TCY, KS = NCELL * CS;          // ct[corner][cell][CS]
    constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f, NEG = -1.0e30f;
    __shared__ float ct[4 * KS];
    __shared__ float oh[2 * NC * 256];                               // [corner row][class][lane]: lane-private columns
    __shared__ float part[8];
    const int h = d.h, w = d.w, H = 8 * h, W = 8 * w;
    const int cells_x = w + 1, cells_y = h + 1;
    const int tid = threadIdx.x, lane8 = tid & 7;
    const int n_class = EXACT ? NC : d.n_class;
#pragma unroll
    for (int c = 0; c < 2 * NC; ++c) oh[c * 256 + tid] = 0.f;       // each lane zeroes, reads and re-zeroes its own column
    float loss = 0.f, cnt = 0.f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int b = tile;
        const int bx = b % tiles_x; b /= tiles_x;
        const int by = b % tiles_y;
        const int n = b / tiles_y;
        const int ci = bx * UP_TCX + ((tid >> 3) & (UP_TCX - 1)), cj = by * UP_TCY + (tid >> 6);
        const bool cell_ok = ci < cells_x && cj < cells_y;
        const int X = 8 * ci - 4 + lane8;
        const bool col_ok = cell_ok && X >= 0 && X < W;
        int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        float lx = 0.f, ly_dummy = 0.f;
        src_index(max(X, 0), w, x0, x1, lx);
        // corner rows of this cell (identical for all its rows)
        src_index(max(8 * cj - 4, 0), h, y0, y1, ly_dummy);
        const float hx = 1.f - lx;
        // first low-resolution pixel the tile's cells touch, before clamping (see the write-out below)
        const int oy = by * UP_TCY - 1, ox = bx * UP_TCX - 1;
        float r0[NC], dd[NC], g0[NC], g1[NC];      // r0 - M0;  (r1 - M1) - (r0 - M0);  gradient sums of the two corner rows
#pragma unroll
        for (int c = 0; c < NC; ++c) { r0[c] = NEG; dd[c] = 0.f; g0[c] = 0.f; g1[c] = 0.f; }
        if (col_ok) {
            const float* L = d.logits + (int64_t)n * h * w * d.ldl;
            const float* p00 = L + ((int64_t)y0 * w + x0) * d.ldl;
            const float* p01 = L + ((int64_t)y0 * w + x1) * d.ldl;
            const float* p10 = L + ((int64_t)y1 * w + x0) * d.ldl;
            const float* p11 = L + ((int64_t)y1 * w + x1) * d.ldl;
            const int64_t* lab = d.labels + (int64_t)n * H * W;
            int tl[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {              // the 8 labels of this lane's column: loads in flight together
                const int Y = 8 * cj - 4 + r;
                tl[r] = (Y >= 0 && Y < H) ? (int)lab[(int64_t)Y * W + X] : d.ignore_index;
            }
            // x-interpolated corner rows in log2 units; classes past n_class stay at -1e30: their exp2() is exactly 0
            float r1[NC];
            if (EXACT && vec4) {
                const float4* q00 = reinterpret_cast<const float4*>(p00);
                const float4* q01 = reinterpret_cast<const float4*>(p01);
                const float4* q10 = reinterpret_cast<const float4*>(p10);
                const float4* q11 = reinterpret_cast<const float4*>(p11);
#pragma unroll
                for (int q = 0; q < (NC + 3) / 4; ++q) {
                    const float4 a = q00[q], bq = q01[q], cq = q10[q], e = q11[q];
                    const float va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {bq.x, bq.y, bq.z, bq.w};
                    const float vc[4] = {cq.x, cq.y, cq.z, cq.w}, ve[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = 4 * q + k;
                        if (c < NC) {
                            r0[c] = LOG2E * (hx * va[k] + lx * vb[k]);
                            r1[c] = LOG2E * (hx * vc[k] + lx * ve[k]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const bool cv = EXACT || c < d.n_class;
                    r0[c] = cv ? LOG2E * (hx * p00[cv ? c : 0] + lx * p01[cv ? c : 0]) : NEG;
                    r1[c] = cv ? LOG2E * (hx * p10[cv ? c : 0] + lx * p11[cv ? c : 0]) : NEG;
                }
            }
            float M0 = r0[0], M1 = r1[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) { M0 = fmaxf(M0, r0[c]); M1 = fmaxf(M1, r1[c]); }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                r0[c] -= M0;
                dd[c] = (r1[c] - M1) - r0[c];
            }
            const float dM = M1 - M0;
#pragma unroll UPCE_UNROLL
            for (int r = 0; r < (UPCE_ABL == 3 ? 0 : 8); ++r) {
                const int Y = 8 * cj - 4 + r;
                const bool row_ok = Y >= 0 && Y < H;
                int yy0, yy1;
                float ly;
                src_index(row_ok ? Y : 0, h, yy0, yy1, ly);
                const float hy = 1.f - ly;
                const int t = tl[r];
                const float live = (t != d.ignore_index) ? 1.f : 0.f;
                // a live label outside [0, n_class) has no class to subtract (the reference raises on it)
                const bool tv = (t != d.ignore_index) && (unsigned)t < (unsigned)n_class;
                float* o0 = oh + (tv ? t : 0) * 256 + tid;
                const float s0 = o0[0], s1 = o0[NC * 256];      // read now, needed after the exponentials
                float z[NC];
                float se = 0.f;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    z[c] = __builtin_amdgcn_exp2f(fmaf(ly, dd[c], r0[c]));
                    se += z[c];
                }
                const float up = fmaf(ly, dM, M0);     // what the row's logits were shifted by
                float more_up = 0.f;
                if (se < 1.0e-30f) {                   // even the largest term underflowed: shift by the row's true maximum
                    float mx = -3.0e38f;
#pragma unroll
                    for (int c = 0; c < NC; ++c) mx = fmaxf(mx, fmaf(ly, dd[c], r0[c]));
                    se = 0.f;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        z[c] = __builtin_amdgcn_exp2f(fmaf(ly, dd[c], r0[c]) - mx);
                        se += z[c];
                    }
                    more_up = mx;
                }
                const float inv = live / se;
                const float a0 = hy * inv, a1 = ly * inv;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    g0[c] = fmaf(a0, z[c], g0[c]);
                    g1[c] = fmaf(a1, z[c], g1[c]);
                }
                // log-sum-exp of the row, in the shifted frame for a valid target (the same shift is missing from the
                // target-logit sum below), in the true frame otherwise
                loss += live * (__builtin_amdgcn_logf(se) + more_up + (tv ? 0.f : up));
                cnt += live;
                o0[0] = s0 + (tv ? hy : 0.f);
                o0[NC * 256] = s1 + (tv ? ly : 0.f);
            }
            // -onehot(t) half of the gradient and target-logit half of the loss, from the per-class weight sums
            float zt = 0.f;
            const bool more = tile + (int)gridDim.x < n_tiles;      // the sums start the next tile at zero
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float w0 = oh[c * 256 + tid], w1 = oh[(NC + c) * 256 + tid];
                if (more) {
                    oh[c * 256 + tid] = 0.f;
                    oh[(NC + c) * 256 + tid] = 0.f;
                }
                g0[c] -= w0;
                g1[c] -= w1;
                zt = fmaf(w0, r0[c], zt);
                zt = fmaf(w1, r0[c] + dd[c], zt);
            }
            loss -= zt;
        }
        // Combine the 8 columns of the cell: its corner (y, x0) gets the hx-weighted, (y, x1) the lx-weighted sum.
        // Each cell stores its four corner sums in its own LDS row (plain stores, spread over its 8 lanes); cell
        // (a, b) of the tile stands for the pixels (oy + a + {0,1}, ox + b + {0,1}) BEFORE clamping: on the image
        // border the interpolation clamps both corners into the image, and whatever weight the outside corner would
        // carry belongs to the nearest inside pixel (its interpolation weight there is 0, or both corners coincide).
        const int cell = tid >> 3;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float a00 = row8_sum(hx * g0[c]), a01 = row8_sum(lx * g0[c]);
            const float a10 = row8_sum(hx * g1[c]), a11 = row8_sum(lx * g1[c]);
            if (UPCE_ABL != 1 && (c & 7) == lane8) {            // cells outside the image store their zeros
                ct[0 * KS + cell * CS + c] = a00;
                ct[1 * KS + cell * CS + c] = a01;
                ct[2 * KS + cell * CS + c] = a10;
                ct[3 * KS + cell * CS + c] = a11;
            }
        }
        __syncthreads();
        float* G = d.dlogits + (int64_t)n * h * w * d.ldl;
        for (int i = tid; i < PY * PX * 32; i += 256) {
            const int c = i & 31, pix = i >> 5;
            const int py = pix / PX, px = pix - py * PX;
            // pixel (py, px) of the tile is corner 00 of cell (py, px), 01 of (py, px-1), 10 of (py-1, px), 11 of (py-1, px-1)
            const int gy = min(max(oy + py, 0), h - 1), gx = min(max(ox + px, 0), w - 1);
            if (c < NC && (EXACT || c < d.n_class)) {
                float v = 0.f;
                if (py < UP_TCY && px < UP_TCX) v += ct[0 * KS + (py * UP_TCX + px) * CS + c];
                if (py < UP_TCY && px > 0) v += ct[1 * KS + (py * UP_TCX + px - 1) * CS + c];
                if (py > 0 && px < UP_TCX) v += ct[2 * KS + ((py - 1) * UP_TCX + px) * CS + c];
                if (py > 0 && px > 0) v += ct[3 * KS + ((py - 1) * UP_TCX + px - 1) * CS + c];
                if (UPCE_ABL != 2 && v != 0.f) atomicAdd(G + ((int64_t)gy * w + gx) * d.ldl + c, v);
            }
        }
        __syncthreads();                               // the corner sums are overwritten in the next round
    }
    // loss / count: wave reduce, block reduce through LDS, one atomic pair per block into the block's slot
    loss *= LN2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        loss += __shfl_xor(loss, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    if ((tid & 63) == 0) { part[(tid >> 6) * 2] = loss; part[(tid >> 6) * 2 + 1] = cnt; }
    __syncthreads();
    if (tid == 0) {
        const float ls = (part[0] + part[2]) + (part[4] + part[6]), cs = (part[1] + part[3]) + (part[5] + part[7]);
        if (UPCE_ABL != 4 && cs > 0.f) {
            float* slot = d.accum + (blockIdx.x % ACC_SLOTS) * ACC_STRIDE;
            atomicAdd(slot, ls);
            atomicAdd(slot + 1, cs);
        }
    }
}

// dlogits /= count (x grad_scale); loss = sum/count; bf16 copy; per-class column sums -> dbias
__global__ __launch_bounds__(256) void ce_finalize_kernel(const dsaspp_ce_desc d) {
    const int64_t M = (int64_t)d.n_img * d.h * d.w;
    const int c = threadIdx.x & 31;
    // lane c of each 32-lane half reads pair c; the butterfly leaves the same sums, added in the same order, in every lane
    static_assert(ACC_SLOTS == 32, "one pair per lane of a 32-lane half");
    float lsum = d.accum[c * ACC_STRIDE], cnt = d.accum[c * ACC_STRIDE + 1];
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
        lsum += __shfl_xor(lsum, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    const float inv = cnt > 0.f ? d.grad_scale / cnt : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.loss) d.loss[0] = cnt > 0.f ? lsum / cnt : 0.f;
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
    // a block walks up to UPCE_TILES_PER_BLOCK tiles (fewer tile set-ups and loss atomics); small inputs keep one tile per block
    const int64_t grid = blocks <= 2048 ? blocks : (blocks + UPCE_TILES_PER_BLOCK - 1) / UPCE_TILES_PER_BLOCK;
    // 16-byte loads of the corner pixels: rows of 4-float groups, the last of which may reach into the row padding
    const int vec4 = ((reinterpret_cast<uintptr_t>(d->logits) & 15) == 0 && d->ldl % 4 == 0 && d->ldl >= 20) ? 1 : 0;
    // Any other class count runs the instantiation for the next multiple of 8: the per-lane weight columns in LDS are
    // 2 * NC * 256 floats (16 KiB at 8 classes, 64 KiB at 32) and the row body is NC exponentials whatever n_class is.
#define UPCE_LAUNCH(NC, EX, V4) \
    hipLaunchKernelGGL((upce_kernel<NC, EX>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, *d, tiles_x, tiles_y, (int)blocks, V4)
    if (d->n_class == 19) UPCE_LAUNCH(19, true, vec4);
    else if (d->n_class <= 8) UPCE_LAUNCH(8, false, 0);
    else if (d->n_class <= 16) UPCE_LAUNCH(16, false, 0);
    else if (d->n_class <= 24) UPCE_LAUNCH(24, false, 0);
    else UPCE_LAUNCH(32, false, 0);
#undef UPCE_LAUNCH
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
// Working memory of dsaspp_upsample8_ce_loss_noalloc: the loss accumulators (first NOALLOC_HEAD floats) and the fp32
// low-resolution gradient the cells add into.  One copy per GPU, resident from the moment the code object is loaded on
// that GPU until the process ends; every call on a GPU uses that GPU's copy.
#define NOALLOC_MAX_PIXELS DSASPP_NOALLOC_MAX_PIXELS   /* low-resolution pixels per call (four 1024x2048 images) */
#define NOALLOC_LD 32                        /* floats per pixel row of the gradient staging */
#define NOALLOC_HEAD DSASPP_CE_ACCUM_FLOATS   /* floats in front of the gradient staging */
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
