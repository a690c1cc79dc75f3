// Error reporting and the recorded-program runner of libdsaspp (host side only).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

static thread_local char g_err[512] = "";

int dsaspp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// ---------------------------------------------------------------- per-device set-up
#include <atomic>
#include <mutex>
#include <set>
#include <utility>
namespace {
std::mutex g_setup_mutex;
std::set<std::pair<const void*, int>> g_lds_done;     // (kernel, device) pairs whose LDS limit has been raised
int g_cu_count[64];                                   // CU count per device, 0 = not looked up yet
std::atomic<int> g_comm_cus{0};
}  // namespace

void dsaspp_lds_limit(const void* fn, int bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_setup_mutex);
    if (g_lds_done.insert(std::make_pair(fn, dev)).second)
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

int dsaspp_persistent_cus(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lock(g_setup_mutex);
        if (g_cu_count[dev] == 0) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
            g_cu_count[dev] = v;
        }
        n = g_cu_count[dev];
    } else {
        n = 256;
    }
    n -= g_comm_cus.load(std::memory_order_relaxed);
    return n < 8 ? 8 : n;
}

extern "C" int dsaspp_set_comm_cus(int32_t n) {
    DS_REQUIRE(n >= 0 && n <= 128, "set_comm_cus: the number of reserved compute units must be in [0, 128] (got %d)", n);
    g_comm_cus.store(n, std::memory_order_relaxed);
    return 0;
}
extern "C" int dsaspp_get_comm_cus(void) { return g_comm_cus.load(std::memory_order_relaxed); }

extern "C" int dsaspp_version(void) { return 2; }
#ifndef DSASPP_SOURCE_HASH
#define DSASPP_SOURCE_HASH "unknown"
#endif
// the marker in front makes the hash findable in the file without loading it
static const char g_source_hash[] = "DSASPP_SRC=" DSASPP_SOURCE_HASH;
extern "C" const char* dsaspp_source_hash(void) { return g_source_hash + 11; }
extern "C" const char* dsaspp_last_error(void) { return g_err; }

namespace {
union AnyDesc {
    dsaspp_conv_desc conv;
    dsaspp_wgrad_desc wgrad;
    dsaspp_bn_desc bn;
    dsaspp_bn_bwd_desc bnb;
    dsaspp_bnact_desc bnact;
    dsaspp_gradx_desc gradx;
    dsaspp_bnapply_desc bnapply;
    dsaspp_pool_desc pool;
    dsaspp_im2col_desc im2col;
    dsaspp_maxpool_desc maxpool;
    dsaspp_ce_desc ce;
    dsaspp_up_desc up;
    dsaspp_generic_desc gen;
    dsaspp_stem3_desc stem3;
    dsaspp_dw_desc dw;
    dsaspp_up2_desc up2;
    dsaspp_dwbwd_desc dwb;
    dsaspp_col2im_desc c2i;
    dsaspp_stem7_desc stem7;
};
struct Op {
    int32_t kind;
    int32_t tag;
    int32_t side;
    AnyDesc d;
};
struct Stamp {
    int32_t tag;
    hipEvent_t a, b;
};
size_t desc_size(int32_t op) {
    switch (op) {
        case DSASPP_OP_CONV: return sizeof(dsaspp_conv_desc);
        case DSASPP_OP_WGRAD: return sizeof(dsaspp_wgrad_desc);
        case DSASPP_OP_BN_FINALIZE: return sizeof(dsaspp_bn_desc);
        case DSASPP_OP_BN_BWD_FINALIZE: return sizeof(dsaspp_bn_bwd_desc);
        case DSASPP_OP_BN_ACT: return sizeof(dsaspp_bnact_desc);
        case DSASPP_OP_GRAD_EXTRACT: return sizeof(dsaspp_gradx_desc);
        case DSASPP_OP_BN_BWD_APPLY: return sizeof(dsaspp_bnapply_desc);
        case DSASPP_OP_AVGPOOL_FWD:
        case DSASPP_OP_AVGPOOL_BWD: return sizeof(dsaspp_pool_desc);
        case DSASPP_OP_IM2COL: return sizeof(dsaspp_im2col_desc);
        case DSASPP_OP_MAXPOOL_FWD:
        case DSASPP_OP_MAXPOOL_BWD: return sizeof(dsaspp_maxpool_desc);
        case DSASPP_OP_UPCE:
        case DSASPP_OP_CE_FINALIZE: return sizeof(dsaspp_ce_desc);
        case DSASPP_OP_UP_FWD:
        case DSASPP_OP_UP_BWD:
        case DSASPP_OP_UP_ARGMAX: return sizeof(dsaspp_up_desc);
        case DSASPP_OP_MEMSET:
        case DSASPP_OP_COPY2D:
        case DSASPP_OP_COPY2D_BF16: return sizeof(dsaspp_generic_desc);
        case DSASPP_OP_STEM3: return sizeof(dsaspp_stem3_desc);
        case DSASPP_OP_DWCONV: return sizeof(dsaspp_dw_desc);
        case DSASPP_OP_UP2_FWD:
        case DSASPP_OP_UP2_BWD: return sizeof(dsaspp_up2_desc);
        case DSASPP_OP_DW_BWD_DATA:
        case DSASPP_OP_DW_WGRAD: return sizeof(dsaspp_dwbwd_desc);
        case DSASPP_OP_STEM3_COL: return sizeof(dsaspp_im2col_desc);
        case DSASPP_OP_COL2IM: return sizeof(dsaspp_col2im_desc);
        case DSASPP_OP_STEM7_FWD:
        case DSASPP_OP_STEM7_WGRAD: return sizeof(dsaspp_stem7_desc);
        default: return 0;
    }
}
}  // namespace

struct dsaspp_program {
    std::vector<Op> ops;
    std::vector<Stamp> stamps;
    bool any_tag = false;
    bool any_side = false;
    hipStream_t side = nullptr;       // created on first use, on the device current at that time
    hipEvent_t fork = nullptr, join = nullptr;
    int side_device = -1;
};

extern "C" dsaspp_program* dsaspp_program_create(void) { return new (std::nothrow) dsaspp_program(); }
extern "C" void dsaspp_program_destroy(dsaspp_program* p) {
    if (!p) return;
    for (auto& st : p->stamps) {
        (void)hipEventDestroy(st.a);
        (void)hipEventDestroy(st.b);
    }
    if (p->fork) (void)hipEventDestroy(p->fork);
    if (p->join) (void)hipEventDestroy(p->join);
    if (p->side) (void)hipStreamDestroy(p->side);
    delete p;
}

extern "C" int dsaspp_program_set_side(dsaspp_program* p, int32_t op_index, int32_t side) {
    DS_REQUIRE(p && op_index >= 0 && op_index < (int)p->ops.size(), "program_set_side: bad arguments");
    DS_REQUIRE(side >= 0 && side <= 2, "program_set_side: side must be 0 (caller's stream), 1 (side stream) or 2 (join before)");
    p->ops[op_index].side = side;
    p->any_side = false;
    for (auto& o : p->ops) p->any_side = p->any_side || o.side != 0;
    return 0;
}

extern "C" int dsaspp_program_set_tag(dsaspp_program* p, int32_t op_index, int32_t tag) {
    DS_REQUIRE(p && op_index >= 0 && op_index < (int)p->ops.size() && tag >= 0, "program_set_tag: bad arguments");
    p->ops[op_index].tag = tag;
    p->any_tag = false;
    for (auto& o : p->ops) p->any_tag = p->any_tag || o.tag != 0;
    return 0;
}

extern "C" int dsaspp_program_read_timing(dsaspp_program* p, int32_t n_tags, double* ms, int64_t* launches) {
    DS_REQUIRE(p && ms && launches && n_tags > 0, "program_read_timing: bad arguments");
    int rc = 0;
    for (auto& st : p->stamps) {
        float t = 0.f;
        hipError_t e = hipEventSynchronize(st.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&t, st.a, st.b);
        if (e != hipSuccess) rc = dsaspp_set_error("program_read_timing: %s", hipGetErrorString(e));
        else if (st.tag < n_tags) {
            ms[st.tag] += (double)t;
            launches[st.tag] += 1;
        }
        (void)hipEventDestroy(st.a);
        (void)hipEventDestroy(st.b);
    }
    p->stamps.clear();
    return rc;
}
extern "C" int dsaspp_program_size(const dsaspp_program* p) { return p ? (int)p->ops.size() : 0; }

extern "C" int dsaspp_program_add(dsaspp_program* p, int32_t op, const void* desc, int64_t desc_bytes) {
    DS_REQUIRE(p && desc, "program_add: null argument");
    const size_t want = desc_size(op);
    DS_REQUIRE(want != 0, "program_add: unknown op %d", op);
    DS_REQUIRE((size_t)desc_bytes == want, "program_add: op %d descriptor is %lld bytes, library expects %zu (header mismatch)", op, (long long)desc_bytes, want);
    Op o;
    o.kind = op;
    o.tag = 0;
    o.side = 0;
    memset(&o.d, 0, sizeof(o.d));
    memcpy(&o.d, desc, want);
    try {
        p->ops.push_back(o);
    } catch (...) {
        return dsaspp_set_error("program_add: out of memory");
    }
    return 0;
}

extern "C" int dsaspp_program_run(dsaspp_program* p, int32_t first, int32_t last, dsaspp_stream_t s) {
    DS_REQUIRE(p, "program_run: null program");
    const int n = (int)p->ops.size();
    if (last < 0 || last > n) last = n;
    DS_REQUIRE(first >= 0 && first <= last, "program_run: bad range [%d, %d)", first, last);
    const dsaspp_stream_t main_s = s;
    bool side_used = false;
    if (p->any_side) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return dsaspp_set_error("program_run: hipGetDevice failed");
        if (p->side && p->side_device != dev) return dsaspp_set_error("program_run: the program's side stream belongs to device %d, current device is %d", p->side_device, dev);
        if (!p->side) {
            if (hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&p->fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&p->join, hipEventDisableTiming) != hipSuccess)
                return dsaspp_set_error("program_run: could not create the side stream");
            p->side_device = dev;
        }
    }
    for (int i = first; i < last; ++i) {
        const Op& o = p->ops[i];
        int rc = 0;
        Stamp st;
        const bool timed = o.tag != 0;
        s = main_s;
        if (o.side == 2 && p->side && side_used) {
            // this op overwrites what a side op reads (or reads what it writes): the caller's stream first waits for
            // everything the side stream has been given so far
            if (hipEventRecord(p->join, p->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)main_s, p->join, 0) != hipSuccess)
                return dsaspp_set_error("program_run: join of the side stream failed");
        }
        if (o.side == 1 && p->side) {
            // the side op sees everything enqueued on the caller's stream so far
            if (hipEventRecord(p->fork, (hipStream_t)main_s) != hipSuccess || hipStreamWaitEvent(p->side, p->fork, 0) != hipSuccess)
                return dsaspp_set_error("program_run: fork to the side stream failed");
            s = (dsaspp_stream_t)p->side;
            side_used = true;
        }
        if (timed) {
            st.tag = o.tag;
            if (hipEventCreate(&st.a) != hipSuccess || hipEventCreate(&st.b) != hipSuccess)
                return dsaspp_set_error("program_run: hipEventCreate failed");
            (void)hipEventRecord(st.a, (hipStream_t)s);
        }
        switch (o.kind) {
            case DSASPP_OP_CONV: rc = dsaspp_conv_fwd(&o.d.conv, s); break;
            case DSASPP_OP_WGRAD: rc = dsaspp_conv_wgrad(&o.d.wgrad, s); break;
            case DSASPP_OP_BN_FINALIZE: rc = dsaspp_bn_finalize(&o.d.bn, s); break;
            case DSASPP_OP_BN_BWD_FINALIZE: rc = dsaspp_bn_bwd_finalize(&o.d.bnb, s); break;
            case DSASPP_OP_BN_ACT: rc = dsaspp_bn_act(&o.d.bnact, s); break;
            case DSASPP_OP_GRAD_EXTRACT: rc = dsaspp_grad_extract(&o.d.gradx, s); break;
            case DSASPP_OP_BN_BWD_APPLY: rc = dsaspp_bn_bwd_apply(&o.d.bnapply, s); break;
            case DSASPP_OP_AVGPOOL_FWD: rc = dsaspp_avgpool2_fwd(&o.d.pool, s); break;
            case DSASPP_OP_AVGPOOL_BWD: rc = dsaspp_avgpool2_bwd(&o.d.pool, s); break;
            case DSASPP_OP_IM2COL: rc = dsaspp_stem_im2col(&o.d.im2col, s); break;
            case DSASPP_OP_MAXPOOL_FWD: rc = dsaspp_maxpool3s2_fwd(&o.d.maxpool, s); break;
            case DSASPP_OP_MAXPOOL_BWD: rc = dsaspp_maxpool3s2_bwd(&o.d.maxpool, s); break;
            case DSASPP_OP_UPCE: rc = dsaspp_upsample8_ce(&o.d.ce, s); break;
            case DSASPP_OP_CE_FINALIZE: rc = dsaspp_ce_finalize(&o.d.ce, s); break;
            case DSASPP_OP_UP_FWD: rc = dsaspp_upsample8_fwd(&o.d.up, s); break;
            case DSASPP_OP_UP_BWD: rc = dsaspp_upsample8_bwd(&o.d.up, s); break;
            case DSASPP_OP_UP_ARGMAX: rc = dsaspp_upsample8_argmax(&o.d.up, s); break;
            case DSASPP_OP_MEMSET: rc = dsaspp_memset_async(o.d.gen.dst, o.d.gen.value, o.d.gen.n, s); break;
            case DSASPP_OP_COPY2D:
                rc = dsaspp_copy2d_f32((const float*)o.d.gen.src, o.d.gen.lds, (float*)o.d.gen.dst, o.d.gen.ldd, o.d.gen.rows,
                                       o.d.gen.cols, o.d.gen.dst_cols, o.d.gen.accumulate, s);
                break;
            case DSASPP_OP_COPY2D_BF16:
                rc = dsaspp_copy2d_f32_bf16((const float*)o.d.gen.src, o.d.gen.lds, o.d.gen.dst, o.d.gen.ldd, o.d.gen.rows,
                                            o.d.gen.cols, o.d.gen.dst_cols, s);
                break;
            case DSASPP_OP_STEM3: rc = dsaspp_stem3x3s2(&o.d.stem3, s); break;
            case DSASPP_OP_DWCONV: rc = dsaspp_dwconv3x3(&o.d.dw, s); break;
            case DSASPP_OP_UP2_FWD: rc = dsaspp_upsample2_fwd(&o.d.up2, s); break;
            case DSASPP_OP_UP2_BWD: rc = dsaspp_upsample2_bwd(&o.d.up2, s); break;
            case DSASPP_OP_DW_BWD_DATA: rc = dsaspp_dwconv3x3_bwd_data(&o.d.dwb, s); break;
            case DSASPP_OP_DW_WGRAD: rc = dsaspp_dwconv3x3_wgrad(&o.d.dwb, s); break;
            case DSASPP_OP_STEM3_COL: rc = dsaspp_stem3_im2col(&o.d.im2col, s); break;
            case DSASPP_OP_COL2IM: rc = dsaspp_stem_col2im(&o.d.c2i, s); break;
            case DSASPP_OP_STEM7_FWD: rc = dsaspp_stem7_fwd(&o.d.stem7, s); break;
            case DSASPP_OP_STEM7_WGRAD: rc = dsaspp_stem7_wgrad(&o.d.stem7, s); break;
            default: rc = dsaspp_set_error("program_run: unknown op %d at %d", o.kind, i);
        }
        if (timed) {
            (void)hipEventRecord(st.b, (hipStream_t)s);
            p->stamps.push_back(st);
        }
        if (rc != 0) {
            char tmp[400];
            snprintf(tmp, sizeof(tmp), "%s", g_err);
            if (side_used) {
                (void)hipEventRecord(p->join, p->side);
                (void)hipStreamWaitEvent((hipStream_t)main_s, p->join, 0);
            }
            return dsaspp_set_error("program op %d (kind %d): %s", i, o.kind, tmp);
        }
    }
    if (side_used) {
        // the caller's stream continues only after the side work of this run
        if (hipEventRecord(p->join, p->side) != hipSuccess || hipStreamWaitEvent((hipStream_t)main_s, p->join, 0) != hipSuccess)
            return dsaspp_set_error("program_run: join of the side stream failed");
    }
    return 0;
}
