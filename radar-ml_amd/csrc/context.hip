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

int rml_ws_reserve(rml_ctx* ctx, size_t bytes, void** out) {
    if (bytes > ctx->ws_bytes) {
        if (ctx->ws) {
            // the old block may still be in use by queued work on any stream
            RML_HIP(hipDeviceSynchronize());
            RML_HIP(hipFree(ctx->ws));
            ctx->ws = nullptr;
            ctx->ws_bytes = 0;
        }
        size_t want = bytes + (bytes >> 3);
        hipError_t e = hipMalloc(&ctx->ws, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            want = bytes;
            e = hipMalloc(&ctx->ws, want);
        }
        if (e != hipSuccess) {
            rml_set_error("workspace allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
            (void)hipGetLastError();
            return RML_ERR_NOMEM;
        }
        ctx->ws_bytes = want;
    }
    *out = ctx->ws;
    return RML_OK;
}

extern "C" const char* rml_version(void) { return "radarml-hip 0.1 (gfx950)"; }
extern "C" const char* rml_last_error(void) { return g_err; }

extern "C" int rml_ctx_create(int device, rml_ctx** out) {
    RML_REQUIRE(out != nullptr, RML_ERR_INVALID, "rml_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    RML_HIP(hipGetDeviceCount(&n));
    RML_REQUIRE(device >= 0 && device < n, RML_ERR_INVALID, "rml_ctx_create: device %d out of range (%d devices)", device, n);
    RML_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RML_HIP(hipGetDeviceProperties(&prop, device));
    rml_ctx* c = new (std::nothrow) rml_ctx();
    RML_REQUIRE(c != nullptr, RML_ERR_NOMEM, "rml_ctx_create: out of host memory");
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    // CU masks on MI355X (measured, tools/exp/exp_cumask.hip): mask bit i selects local CU i/8 of XCD i%8; an XCD
    // left without any CU makes the runtime ignore the mask for that XCD.
    const char* env = getenv("RML_GEMM_CUS");
    int g = env ? atoi(env) : 0;
    hipError_t e = hipSuccess;
    if (g > 0 && g < 32 && prop.multiProcessorCount == 256) {
        uint32_t mg[8] = {0}, mp[8] = {0};
        for (int bit = 0; bit < 256; ++bit) ((bit / 8) < g ? mg : mp)[bit / 32] |= 1u << (bit % 32);
        e = hipExtStreamCreateWithCUMask(&c->aux_stream, 8, mg);
        if (e == hipSuccess) e = hipExtStreamCreateWithCUMask(&c->proj_stream, 8, mp);
        c->gemm_cus_per_xcd = g;
    } else {
        e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_last, hipEventDisableTiming);
    for (int i = 0; i < 3 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->ev_proj[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_flags[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_gemm[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return rml_hip_fail(e, "stream/event creation", __FILE__, __LINE__);
    }
    *out = c;
    return RML_OK;
}

extern "C" int rml_ctx_destroy(rml_ctx* ctx) {
    if (!ctx) return RML_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->ev_last) (void)hipEventDestroy(ctx->ev_last);
    for (int i = 0; i < 3; ++i) {
        if (ctx->ev_proj[i]) (void)hipEventDestroy(ctx->ev_proj[i]);
        if (ctx->ev_done[i]) (void)hipEventDestroy(ctx->ev_done[i]);
        if (ctx->ev_flags[i]) (void)hipEventDestroy(ctx->ev_flags[i]);
        if (ctx->ev_gemm[i]) (void)hipEventDestroy(ctx->ev_gemm[i]);
    }
    if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
    for (hipEvent_t e : ctx->prof_ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof_ev_g) (void)hipEventDestroy(e);
    for (const rml_resize_tab& t : ctx->resize_tabs) (void)hipFree(const_cast<double*>(t.kk));
    for (const rml_pre_tab& t : ctx->pre_tabs) (void)hipFree(t.dev);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->proj_stream) (void)hipStreamDestroy(ctx->proj_stream);
    delete ctx;
    return RML_OK;
}

void rml_prof_mark(rml_ctx* ctx, hipStream_t st) {
    if (!ctx->profiling) return;
    if (ctx->prof_used == ctx->prof_ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return; }
        ctx->prof_ev.push_back(e);
    }
    (void)hipEventRecord(ctx->prof_ev[ctx->prof_used++], st);
}

void rml_prof_mark_gemm(rml_ctx* ctx, hipStream_t st) {
    if (!ctx->profiling) return;
    if (ctx->prof_used_g == ctx->prof_ev_g.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return; }
        ctx->prof_ev_g.push_back(e);
    }
    (void)hipEventRecord(ctx->prof_ev_g[ctx->prof_used_g++], st);
}

extern "C" int rml_profile_read_gemm(rml_ctx* ctx, int64_t* launches, double* total_ms, double* ops) {
    RML_REQUIRE(ctx != nullptr, RML_ERR_INVALID, "rml_profile_read_gemm: ctx is NULL");
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
    ctx->profiling = on != 0;
    ctx->prof_used = 0;
    ctx->prof_frames = 0;
    ctx->prof_used_g = 0;
    ctx->prof_ops_g = 0.0;
    return RML_OK;
}

extern "C" int rml_profile_read(rml_ctx* ctx, int64_t* launches, double* total_ms, int64_t* frames) {
    RML_REQUIRE(ctx != nullptr, RML_ERR_INVALID, "rml_profile_read: ctx is NULL");
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

extern "C" int rml_ctx_set_option(rml_ctx* ctx, int option, int value) {
    RML_REQUIRE(ctx != nullptr, RML_ERR_INVALID, "rml_ctx_set_option: ctx is NULL");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    switch (option) {
        case RML_OPT_PROJECT_SHARE_CU: ctx->opt_project_share_cu = value ? 1 : 0; return RML_OK;
        default: RML_REQUIRE(false, RML_ERR_INVALID, "rml_ctx_set_option: unknown option %d", option);
    }
}
