// Context, error reporting and workspace management of libradarml_hip.so.
#include "rml_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <new>

namespace {
thread_local char g_err[512] = "";
}

void rml_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int rml_hip_fail(hipError_t e, const char* what, const char* file, int line) {
    rml_set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return RML_ERR_HIP;
}

// outgrown workspace blocks whose retirement point has passed (all of them when `all`: the caller has synchronised)
static void ws_release_retired(rml_ctx* ctx, bool all) {
    size_t k = 0;
    for (size_t i = 0; i < ctx->ws_retired.size(); ++i) {
        rml_ws_retired& r = ctx->ws_retired[i];
        if (all || hipEventQuery(r.ev) == hipSuccess) {
            (void)hipFree(r.p);
            (void)hipEventDestroy(r.ev);
        } else {
            ctx->ws_retired[k++] = r;
        }
    }
    (void)hipGetLastError();            // hipErrorNotReady of a pending event is not an error
    ctx->ws_retired.resize(k);
}

static int ws_grow(rml_ctx* ctx, size_t bytes, hipStream_t st, bool retire_on_stream) {
    void* nw = nullptr;
    size_t want = bytes + (bytes >> 3);
    hipError_t e = hipMalloc(&nw, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ws_release_retired(ctx, false);
        want = bytes;
        e = hipMalloc(&nw, want);
    }
    if (e != hipSuccess) {
        rml_set_error("workspace allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return RML_ERR_NOMEM;
    }
    if (ctx->ws) {
        // The old block may still be in use by work queued before this call.  Every user ran under rml_ctx_guard and this call's
        // stream already waits for the last of them (ev_last), so an event recorded on it NOW completes after all of them: the block
        // is parked behind that event and freed when a later growth (or rml_ctx_reserve_workspace / rml_ctx_destroy) finds it done.
        // No hipDeviceSynchronize, no hipFree on this path (round 5 did both here, inside "asynchronous" entry points).
        rml_ws_retired r{ctx->ws, nullptr};
        if (retire_on_stream && hipEventCreateWithFlags(&r.ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(r.ev, st) == hipSuccess) {
            ctx->ws_retired.push_back(r);
        } else {
            (void)hipGetLastError();
            if (r.ev) (void)hipEventDestroy(r.ev);
            (void)hipDeviceSynchronize();
            (void)hipFree(ctx->ws);
        }
    }
    ctx->ws = nw;
    ctx->ws_bytes = want;
    return RML_OK;
}

int rml_ws_reserve(rml_ctx* ctx, size_t bytes, void** out, hipStream_t st) {
    if (bytes > ctx->ws_bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
        RML_REQUIRE(cs == hipStreamCaptureStatusNone, RML_ERR_INVALID,
                    "the context's workspace would have to grow from %zu to %zu bytes inside a stream capture: call rml_ctx_reserve_workspace "
                    "(or run the same call once) before capturing", ctx->ws_bytes, bytes);
        ws_release_retired(ctx, false);
        const int rc = ws_grow(ctx, bytes, st, true);
        if (rc) return rc;
    }
    *out = ctx->ws;
    return RML_OK;
}

extern "C" int rml_ctx_reserve_workspace(rml_ctx* ctx, int64_t bytes) {
    RML_REQUIRE(ctx != nullptr && bytes >= 0, RML_ERR_INVALID, "rml_ctx_reserve_workspace: bad arguments");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    int prev = -1;
    RML_HIP(hipGetDevice(&prev));
    RML_HIP(hipSetDevice(ctx->device));
    RML_HIP(hipDeviceSynchronize());
    ws_release_retired(ctx, true);
    int rc = RML_OK;
    if ((size_t)bytes > ctx->ws_bytes) {
        if (ctx->ws) { (void)hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
        rc = ws_grow(ctx, (size_t)bytes, nullptr, false);
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return rc;
}

extern "C" int64_t rml_ctx_workspace_bytes(const rml_ctx* ctx) { return ctx ? (int64_t)ctx->ws_bytes : 0; }

extern "C" const char* rml_version(void) { return "radarml-hip 0.1 (gfx950)"; }
extern "C" const char* rml_last_error(void) { return g_err; }

extern "C" int rml_ctx_create(int device, rml_ctx** out) {
    RML_REQUIRE(out != nullptr, RML_ERR_INVALID, "rml_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    RML_HIP(hipGetDeviceCount(&n));
    RML_REQUIRE(device >= 0 && device < n, RML_ERR_INVALID, "rml_ctx_create: device %d out of range (%d devices)", device, n);
    // the caller's current device is the caller's: streams and events are created on `device` and the previous one is restored
    struct restore_device {
        int prev = -1;
        restore_device() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
        ~restore_device() { if (prev >= 0) (void)hipSetDevice(prev); }
    } restore;
    RML_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RML_HIP(hipGetDeviceProperties(&prop, device));
    rml_ctx* c = new (std::nothrow) rml_ctx();
    RML_REQUIRE(c != nullptr, RML_ERR_NOMEM, "rml_ctx_create: out of host memory");
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    // (CU-masked streams for a GEMM / projection partition lived here in rounds 2-4 -- mask bit i selects local CU i/8 of XCD
    // i%8, tools/exp/exp_cumask.hip -- and never won a measurement: DESIGN.md 3.3.)
    hipError_t e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_last, hipEventDisableTiming);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->ev_proj[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        delete c;
        return rml_hip_fail(e, "stream/event creation", __FILE__, __LINE__);
    }
    *out = c;
    return RML_OK;
}

extern "C" int rml_ctx_destroy(rml_ctx* ctx) {
    if (!ctx) return RML_OK;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ws_release_retired(ctx, true);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_last) (void)hipEventDestroy(ctx->ev_last);
    for (int i = 0; i < 2; ++i) {
        if (ctx->ev_proj[i]) (void)hipEventDestroy(ctx->ev_proj[i]);
        if (ctx->ev_done[i]) (void)hipEventDestroy(ctx->ev_done[i]);
    }
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof_ev_g) (void)hipEventDestroy(e);
    for (const rml_resize_tab& t : ctx->resize_tabs) (void)hipFree(const_cast<double*>(t.kk));
    for (const rml_pre_tab& t : ctx->pre_tabs) (void)hipFree(t.dev);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    delete ctx;
    if (prev >= 0) (void)hipSetDevice(prev);
    return RML_OK;
}

// (timing events are not recorded into a stream capture: they could not be read back)
static bool stream_is_capturing(hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
    return cs != hipStreamCaptureStatusNone;
}

void rml_prof_mark(rml_ctx* ctx, hipStream_t st) {
    if (!ctx->profiling || stream_is_capturing(st)) return;
    if (ctx->prof_used == ctx->prof_ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return; }
        ctx->prof_ev.push_back(e);
    }
    (void)hipEventRecord(ctx->prof_ev[ctx->prof_used++], st);
}

void rml_prof_mark_gemm(rml_ctx* ctx, hipStream_t st) {
    if (!ctx->profiling || stream_is_capturing(st)) return;
    if (ctx->prof_used_g == ctx->prof_ev_g.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return; }
        ctx->prof_ev_g.push_back(e);
    }
    (void)hipEventRecord(ctx->prof_ev_g[ctx->prof_used_g++], st);
}

extern "C" int rml_profile_read_gemm(rml_ctx* ctx, int64_t* launches, double* total_ms, double* ops) {
    RML_REQUIRE(ctx != nullptr, RML_ERR_INVALID, "rml_profile_read_gemm: ctx is NULL");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);          // the marks are recorded by guarded entry points: the same lock
    RML_HIP(hipSetDevice(ctx->device));
    double tot = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < ctx->prof_used_g; i += 2) {
        RML_HIP(hipEventSynchronize(ctx->prof_ev_g[i + 1]));
        float ms = 0.0f;
        RML_HIP(hipEventElapsedTime(&ms, ctx->prof_ev_g[i], ctx->prof_ev_g[i + 1]));
        tot += ms;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    if (ops) *ops = ctx->prof_ops_g;
    ctx->prof_used_g = 0;
    ctx->prof_ops_g = 0.0;
    return RML_OK;
}

extern "C" int rml_profile_enable(rml_ctx* ctx, int on) {
    RML_REQUIRE(ctx != nullptr, RML_ERR_INVALID, "rml_profile_enable: ctx is NULL");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->profiling = on != 0;
    ctx->prof_used = 0;
    ctx->prof_frames = 0;
    ctx->prof_used_g = 0;
    ctx->prof_ops_g = 0.0;
    return RML_OK;
}

extern "C" int rml_profile_read(rml_ctx* ctx, int64_t* launches, double* total_ms, int64_t* frames) {
    RML_REQUIRE(ctx != nullptr, RML_ERR_INVALID, "rml_profile_read: ctx is NULL");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    RML_HIP(hipSetDevice(ctx->device));
    double tot = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < ctx->prof_used; i += 2) {
        RML_HIP(hipEventSynchronize(ctx->prof_ev[i + 1]));
        float ms = 0.0f;
        RML_HIP(hipEventElapsedTime(&ms, ctx->prof_ev[i], ctx->prof_ev[i + 1]));
        tot += ms;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = tot;
    if (frames) *frames = ctx->prof_frames;
    ctx->prof_used = 0;
    ctx->prof_frames = 0;
    return RML_OK;
}

extern "C" int rml_ctx_device(const rml_ctx* ctx) { return ctx ? ctx->device : RML_ERR_INVALID; }

static int* opt_slot(rml_opts& o, int option) {
    switch (option) {
        case RML_OPT_PROJECT_SHARE_CU: return &o.project_share_cu;
        case RML_OPT_WAVEFRAME: return &o.waveframe;
        case RML_OPT_LINPLANE: return &o.linplane;
        case RML_OPT_STAGE_CODES: return &o.stage_codes;
        case RML_OPT_SLICE_WAVE: return &o.slice_wave;
        case RML_OPT_DERIVE_FUSED: return &o.derive_fused;
        case RML_OPT_CODE_RMW: return &o.code_rmw;
        case RML_OPT_GEMM_BIG: return &o.gemm_big;
        case RML_OPT_C1_PK: return &o.c1_pk;
        default: return nullptr;
    }
}

extern "C" int rml_ctx_set_option(rml_ctx* ctx, int option, int value) {
    RML_REQUIRE(ctx != nullptr, RML_ERR_INVALID, "rml_ctx_set_option: ctx is NULL");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    if (option == RML_OPT_CHUNK) {
        RML_REQUIRE(value == 0 || value >= 128, RML_ERR_INVALID, "rml_ctx_set_option: RML_OPT_CHUNK is 0 (automatic) or >= 128 rows");
        ctx->opt.chunk = value;
        return RML_OK;
    }
    int* slot = opt_slot(ctx->opt, option);
    RML_REQUIRE(slot != nullptr, RML_ERR_INVALID, "rml_ctx_set_option: unknown option %d", option);
    switch (option) {
        case RML_OPT_WAVEFRAME: RML_REQUIRE(value >= 0 && value <= 3, RML_ERR_INVALID, "rml_ctx_set_option: RML_OPT_WAVEFRAME is 0..3"); *slot = value; break;
        case RML_OPT_CODE_RMW: case RML_OPT_GEMM_BIG: *slot = value < 0 ? -1 : (value ? 1 : 0); break;
        default: *slot = value ? 1 : 0;
    }
    return RML_OK;
}

extern "C" int rml_ctx_get_option(const rml_ctx* ctx, int option, int* value) {
    RML_REQUIRE(ctx != nullptr && value != nullptr, RML_ERR_INVALID, "rml_ctx_get_option: NULL argument");
    rml_ctx* c = const_cast<rml_ctx*>(ctx);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (option == RML_OPT_CHUNK) { *value = (int)c->opt.chunk; return RML_OK; }
    const int* slot = opt_slot(c->opt, option);
    RML_REQUIRE(slot != nullptr, RML_ERR_INVALID, "rml_ctx_get_option: unknown option %d", option);
    *value = *slot;
    return RML_OK;
}

// ---- rml_probe_stream: what a pure streaming READ reaches on this device ------------------------------------------------------------
// The "measured" denominator beside the 8 TB/s specification: a persistent kernel, 2 x 256 threads per CU, every
// lane keeps 16 non-temporal 16-byte loads in flight over workgroup-contiguous 64 KB pieces and folds them with a max (the
// access pattern and the arithmetic of a projection without its epilogue); nothing is written unless a sentinel matches.
namespace {
typedef float probe_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_probe_stream(const probe_v4* __restrict__ src, int64_t n16, float* sink) {
    constexpr int PF = 16;
    const int64_t piece = (int64_t)256 * PF;                 // 16-byte words per workgroup step (64 KB)
    probe_v4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int64_t base = (int64_t)blockIdx.x * piece; base < n16; base += (int64_t)gridDim.x * piece) {
        probe_v4 v[PF];
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            int64_t q = base + (int64_t)u * 256 + threadIdx.x;
            q = q < n16 ? q : n16 - 1;
            v[u] = __builtin_nontemporal_load(src + q);
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            m.x = fmaxf(m.x, v[u].x); m.y = fmaxf(m.y, v[u].y); m.z = fmaxf(m.z, v[u].z); m.w = fmaxf(m.w, v[u].w);
        }
    }
    const float r = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));
    if (r == 12345.678f) sink[blockIdx.x] = r;               // never true for the caller's data in practice: keeps the loads alive
}
}  // namespace

extern "C" int rml_probe_stream(rml_ctx* ctx, const void* buf, int64_t bytes, int reps, double* gb_per_s, void* stream) {
    RML_REQUIRE(ctx && buf && gb_per_s, RML_ERR_INVALID, "rml_probe_stream: NULL argument");
    RML_REQUIRE(bytes >= (1 << 20) && reps >= 1 && (reinterpret_cast<uintptr_t>(buf) & 15) == 0, RML_ERR_INVALID,
                "rml_probe_stream: needs a 16-byte aligned buffer of at least 1 MiB and reps >= 1");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    rml_ctx_guard guard(ctx, st);           // a workspace user like the others: serialised on the context
    void* sink = nullptr;
    int rc = rml_ws_reserve(ctx, (size_t)4 * 2 * ctx->num_cu, &sink, st);
    if (rc) return rc;
    const int64_t n16 = bytes / 16;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    RML_HIP(hipEventCreate(&e0));
    {
        const hipError_t ec = hipEventCreate(&e1);
        if (ec != hipSuccess) { (void)hipEventDestroy(e0); RML_HIP(ec); }
    }
    const dim3 grid((unsigned)(2 * ctx->num_cu)), block(256);
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL(k_probe_stream, grid, block, 0, st, static_cast<const probe_v4*>(buf), n16, static_cast<float*>(sink));
    hipError_t e = hipGetLastError();                       // a failed warm-up launch shows here, before anything is timed
    if (e == hipSuccess) e = hipEventRecord(e0, st);
    for (int i = 0; i < reps && e == hipSuccess; ++i)
        hipLaunchKernelGGL(k_probe_stream, grid, block, 0, st, static_cast<const probe_v4*>(buf), n16, static_cast<float*>(sink));
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    RML_HIP(e);
    RML_HIP(hipGetLastError());
    *gb_per_s = ms > 0.0f ? (double)reps * (double)(n16 * 16) / ((double)ms * 1e-3) / 1e9 : 0.0;
    return RML_OK;
}

// the pipelines' default for read-compare-write code stores (rml_internal.h rml_code_rmw; RML_OPT_CODE_RMW overrides it per context): bench.py reports it
extern "C" int rml_code_rmw_default(int64_t D, int64_t frame_bytes, int derive, int u8) {
    return rml_code_rmw(D, frame_bytes, derive != 0, u8 != 0);
}
