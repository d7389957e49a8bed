// Internal declarations shared by the HIP translation units of libradarml_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/radarml.h"

// device copy of Pillow's precompute_coeffs() table for one (in, out) size pair (resize.hip)
struct rml_resize_tab {
    int in, out, ksize;
    const int* bounds;      // [out][2] (points into the allocation that starts at kk)
    const double* kk;       // [out][ksize]
};

// device tables of the CNN preprocessing kernel for one (grid, output size) pair (preprocess.hip)
struct rml_pre_tab {
    int X, Y, Z, OH, OW;
    void* dev;              // one allocation; the six tables start at off[]
    size_t off[6];
};

// rml_ctx_set_option values.  The defaults are the tuned configuration; the others exist for A/B measurements and for the tests
// that pin one kernel family against another (they used to be environment variables read with getenv inside the launch paths:
// a data race beside a caller's setenv, and invisible in the API).
struct rml_opts {
    int project_share_cu = 0;   // RML_OPT_PROJECT_SHARE_CU
    int waveframe = 1;          // RML_OPT_WAVEFRAME: 0 off, 1 on, 2 quarter-plane buffers everywhere, 3 also short rows stand-alone
    int linplane = 1;           // RML_OPT_LINPLANE: k_project_lin for rows of 40 / 44 / 48 / 56 quads
    int stage_codes = 1;        // RML_OPT_STAGE_CODES: per-wave LDS code stage of the wave-per-frame kernels
    int slice_wave = 1;         // RML_OPT_SLICE_WAVE: k_slice_rows (0: the round-1 workgroup-per-row kernel)
    int derive_fused = 1;       // RML_OPT_DERIVE_FUSED: k_derive_slice (0: sum planes + top-k + slices)
    int code_rmw = -1;          // RML_OPT_CODE_RMW: -1 = the measured rule (rml_code_rmw), 0 / 1 forced
    int gemm_big = -1;          // RML_OPT_GEMM_BIG: -1 = whole-round rule (use_big_gemm), 0 never, 1 for every n >= 256
    int c1_pk = 1;              // RML_OPT_C1_PK: packed first-layer kernels of the SGAN branches
    int64_t chunk = 0;          // RML_OPT_CHUNK: rows per chunk of the chunked front doors, 0 = chosen per batch
};

struct rml_ws_retired {         // a workspace block that was outgrown: freed once the work queued before its retirement is done
    void* p;
    hipEvent_t ev;
};

struct rml_ctx {
    int device = 0;
    int num_cu = 256;
    // grow-on-demand device workspace for the fused front doors (owned by the ctx)
    void* ws = nullptr;
    size_t ws_bytes = 0;
    std::vector<rml_ws_retired> ws_retired;
    rml_opts opt;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_proj[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};   // chunk pipeline of rml_project_svm (two workspaces)
    hipStream_t aux_stream = nullptr;   // second stream for overlapping GEMM with projection
    // optional in-situ timing of the projection launches issued by rml_project_svm
    bool profiling = false;
    std::vector<hipEvent_t> prof_ev;    // start/stop pairs
    size_t prof_used = 0;
    int64_t prof_frames = 0;
    std::vector<hipEvent_t> prof_ev_g;  // the same for the GEMM + finish of every chunk (aux stream)
    size_t prof_used_g = 0;
    double prof_ops_g = 0.0;
    std::vector<rml_resize_tab> resize_tabs;    // owned; freed with the context
    std::vector<rml_pre_tab> pre_tabs;          // owned; freed with the context
    // Entry points that use the shared workspace / events / caches take `mu` for the duration of the call and order
    // their stream behind the previous user's work (ev_last), so calls from several host threads or on several
    // streams cannot interleave on the workspace (rml_ctx_guard below).
    std::recursive_mutex mu;
    hipEvent_t ev_last = nullptr;
    bool ev_last_valid = false;
};

// RAII serialisation of one entry point on a context: locks ctx->mu, makes `stream` wait for the work the previous
// guarded call queued (on whatever stream), and on destruction records the new "last use" point on `stream`.
struct rml_ctx_guard {
    rml_ctx* ctx;
    hipStream_t st;
    bool capturing = false;
    explicit rml_ctx_guard(rml_ctx* c, hipStream_t stream) : ctx(c), st(stream) {
        ctx->mu.lock();
        // Inside a stream capture (a caller recording the call into a HIP graph) the cross-call ordering is the caller's: an event
        // recorded by an earlier, un-captured call cannot be waited for from a captured stream on every runtime, and an event recorded
        // INSIDE the capture must not become the "last use" that later un-captured calls wait for.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) (void)hipGetLastError();
        capturing = cs != hipStreamCaptureStatusNone;
        if (!capturing && ctx->ev_last_valid) (void)hipStreamWaitEvent(st, ctx->ev_last, 0);
    }
    ~rml_ctx_guard() {
        if (!capturing && ctx->ev_last && hipEventRecord(ctx->ev_last, st) == hipSuccess) ctx->ev_last_valid = true;
        ctx->mu.unlock();
    }
    rml_ctx_guard(const rml_ctx_guard&) = delete;
    rml_ctx_guard& operator=(const rml_ctx_guard&) = delete;
};

// records an event on st when profiling is on (no-op otherwise)
void rml_prof_mark(rml_ctx* ctx, hipStream_t st);
void rml_prof_mark_gemm(rml_ctx* ctx, hipStream_t st);

// "done once per device" for hipFuncSetAttribute: one process may drive several devices (one context each), and the
// attribute belongs to the device's copy of the kernel
struct rml_once_per_device {
    std::atomic<uint64_t> mask{0};
    bool first(int dev) { const uint64_t bit = 1ull << (dev & 63); return !(mask.fetch_or(bit) & bit); }
};
// allow `bytes` of dynamic LDS for kernel (the variadic part: a template-id may hold commas) on the current device
#define RML_MAX_DYN_LDS(bytes, ...)                                                                                     \
    do {                                                                                                                \
        static rml_once_per_device _once;                                                                               \
        int _dev = 0;                                                                                                   \
        (void)hipGetDevice(&_dev);                                                                                      \
        if (_once.first(_dev))                                                                                          \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
    } while (0)

void rml_set_error(const char* fmt, ...);
int rml_hip_fail(hipError_t e, const char* what, const char* file, int line);

#define RML_HIP(call)                                                        \
    do {                                                                     \
        hipError_t _e = (call);                                              \
        if (_e != hipSuccess) return rml_hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

#define RML_REQUIRE(cond, status, ...)      \
    do {                                    \
        if (!(cond)) {                      \
            rml_set_error(__VA_ARGS__);     \
            return (status);                \
        }                                   \
    } while (0)

// workspace of at least `bytes` (inside an rml_ctx_guard on `st`); growing allocates -- not inside a stream capture: context.hip
int rml_ws_reserve(rml_ctx* ctx, size_t bytes, void** out, hipStream_t st);

// ---- projection (project.hip) -------------------------------------------------------------
struct ProjOut {
    // destination of each plane for frame b: p[pl] + b*stride[pl] (float) ; NULL = plane not wanted
    float* p[3];
    int64_t stride[3];
    uint32_t sel;      // bit pl set: plane pl is part of the row (stored and counted in the statistics)
    // uint8 codes of the same values (before scaling); NULL = not wanted
    uint8_t* q[3];
    int64_t qstride;
    uint8_t* qrow;     // base of the code rows (pad columns [qD, qstride) are zeroed), or NULL
    int64_t qD;
    int32_t* row_isum;
    int64_t* row_isq;
    int32_t* row_flags;
    float scale_div;   // 0/1 => none
    // float row bookkeeping for the SVM front door: zero the pad columns [pD, stride) of the
    // float row at prow + b*pstride, and the float64 squared norm of the (scaled) row
    float* prow;
    int64_t pD, pstride;
    double* row_nsq;
    // predication: the whole launch is a no-op when *skip_if_set != 0 (device flag)
    const int32_t* skip_if_set;
    // set by the fused pipeline: an SVM GEMM runs concurrently on another stream, so a persistent projection kernel
    // should leave every CU the registers / LDS for one of its workgroups (k_project_wave: quarter-plane buffers, one
    // workgroup per CU) -- measured +5 % end to end on the Walabot grid against filling the CUs with projection waves
    int share_cu;
    // the predicated (usually skipped) second pass of the fused pipeline: launch without the LDS pad of share_cu, so that its
    // workgroups can start -- and exit at once -- on CUs whose LDS a resident projection workgroup of the next chunk holds
    int no_pad;
    // Read-compare-write of the code rows: read the old 4 / 16 bytes and store only what changed (always correct: whatever the row
    // held before, it holds the new codes afterwards).  Radar projections are mostly background -- 17 % / 9 % of the codes of a
    // Walabot / 64x64x128 frame are non-zero -- and the rows land in buffers that are re-used batch after batch (the chunk workspaces
    // of the fused pipelines, a caller's code buffer), so most of a row's HBM WRITE traffic -- what the streaming kernels pay
    // 3-15 % for, DESIGN.md 3.1b' -- becomes reads.  The kernels put the old words of a region in flight together (Emitter::old_word).
    int q_rmw;
};
// rml_code_rmw: the default of ProjOut::q_rmw in the fused pipelines, for frames of frame_bytes volume bytes and D codes.  The
// read-back costs its bytes (and, in a per-frame epilogue, its latency) whatever the rows hold; the stores it saves depend on the
// data.  So the rule is what was measured (tools/exp/README.md round 4; synthetic frames with the sparsity above; sessions r4be /
// r4bf: this library against the library of the commit before, three interleaved whole-round runs on one box):
//   64x64x128 uint8 (rows 3.9 % of the frame): 7.41 -> 7.86 M frames/s (+6 %; +11 % on another box)                        on
//   64x64x128 derive -> slice (rows 1 %): 2.193 -> 2.264 M (+3.2 %)                                                          on
//   Walabot grid, uint8 (rows 8.3 %): -2.5 %                                                                                off
//   Walabot grid, derive -> slice (rows 2.1 %): per store 7.94 -> 7.89 M (kernel in situ -7 %); with the old words loaded
//     alongside the gather's batches (slice_emit<., ., true>, session r4bs) 8.33-8.35 -> 8.33-8.39 M, kernel 0.576 -> 0.585;
//     64x64x128 in that form 2.24 -> 2.26 M, kernel 0.618 -> 0.634                                                          on
//   float32 max projections: +0.7 / +1.1 % end to end at the Walabot grid and +-0.5 % at 64x64x128 against the same library
//     without -- but the prefetched old words and the test in front of every store cost k_project_lin 9 % in situ (0.735 -> 0.665)
//     and the headline 1 % against the library before: those kernels store plainly again (Emitter::rmw = false)              off
// The CNN chain's first pass keeps plain stores (5.98 / 5.97 M frames/s without, 5.87 / 5.96 with).
// rml_ctx_set_option(RML_OPT_CODE_RMW, 0 / 1) forces it off / on where a kernel has it (-1: this rule).
inline int rml_code_rmw(int64_t D, int64_t frame_bytes, bool derive, bool u8) {
    // round 5, FRESH frames (bench.py's sliding windows: no workspace row meets its own frame's old codes; sessions r5c / r5d):
    // 64x64x128 uint8 +2.7 % (the +6-10 % above came with a batch re-run 20 times), derive -> slice +1.6 % / +1.3 % -- below the
    // 2 % bar: the derive pipelines store plainly
    if (derive) return 0;
    return u8 && D * 64 >= frame_bytes && D * 16 <= frame_bytes;
}

// true when rml_launch_project would use the persistent wave-per-frame kernel for this shape (the fused pipeline then
// pairs it with the 128x128 GEMM, whose workgroups fit beside it on a CU)
bool rml_project_uses_wave_kernel(const rml_ctx* ctx, int vdtype, int mode, int X, int Y, int Z, bool share_cu, int64_t B);

int rml_launch_project(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode,
                       const int32_t* ijk, const ProjOut& o, hipStream_t st, int targets_per_frame = 1);

// single observations: mode MAX with every frame split into S pieces along x (project.hip); S = rml_project_split_pieces (0: not taken)
constexpr int RML_SMALL_FRAMES = 8;
int rml_project_split_pieces(int X, int Y, int Z);
size_t rml_project_split_scratch_bytes(int64_t B, int X, int Y, int Z, int S);
int rml_launch_project_split(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, const ProjOut& o, float* scratch, int S,
                             hipStream_t st);

// k_derive_slice (project_slice.hip): DerivedTarget.get_derived_targets (common.py:49-80) and the slices at the derived (i,j,k) in one
// pass over the frames; o.sel == 0: derive only.  RML_ERR_UNSUPPORTED (no message set) when the shape has no fused kernel.
int rml_launch_derive_slice(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int num_targets,
                            int32_t* ijk_out, float* profiles, const ProjOut& o, hipStream_t st);

// ---- SVM (svm.hip) ------------------------------------------------------------------------
struct rml_svm {
    int64_t M = 0, Mpad = 0, D = 0;
    int64_t Kq = 0;               // code K extent (bytes)   = round_up(D, 128)
    int64_t Kf = 0;               // float K extent (floats) = round_up(D, 32)
    // Row strides: the K extent, plus one 128-byte granule when that makes the stride an ODD multiple of 128 B.
    // With an even multiple (D = 20 480: 160 x 128 B) the same K-slice of every row of a tile maps to the same
    // L2 channel and the LDS-DMA staging of the GEMM serialises on it.
    int64_t Dq = 0;               // code row stride (bytes)
    int64_t Df = 0;               // float row stride (floats)
    int C = 0, P = 0, PT = 0, kernel = 0;
    double gamma = 0, code_scale = 1;
    bool exact = false, has_calib = false;
    // device buffers
    float* sv_f32 = nullptr;      // Mpad x Df, zero padded
    double* sv_nsq = nullptr;     // Mpad  ||sv||^2 (float64)
    uint8_t* sv_q = nullptr;      // Mpad x Dq codes XOR 0x80 (= int8 code-128); pad = 0
    double* sv_term_q = nullptr;  // Mpad  exact-path per-SV term (see svm.hip)
    double* W = nullptr;          // PT x Mpad per-pair SV weights (zero padded)
    double* intercept = nullptr;  // P
    double* calib = nullptr;      // 2*C (a then b)
    double* platt = nullptr;      // libsvm probA | probB (2 x kMaxP), set by rml_svm_set_platt
    // digit path (general rows on the int8 matrix cores, svm.hip "multi-digit"): every value v is the 32-bit fixed-point
    // number u = (v - dig_c0) / dig_s in [-1, 1), split into four balanced int8 digits; plane i of row r lives at
    // sv_dig + (i * Mpad + r) * Dq (plane-major: the row stride stays the odd multiple of 128 B)
    bool dig_ok = false;
    double dig_c0 = 0, dig_s = 1;
    int8_t* sv_dig = nullptr;     // 4 x Mpad x Dq
    double* sv_dig_nsq = nullptr; // Mpad  ||u_sv||^2 of the quantised values
    uint32_t mask_hint = 0;
};

// host-side packed operands of a model (rml_svm_pack_host fills them, rml_svm_load uploads them)
struct rml_svm_pack {
    std::vector<double> W, nsq, term, dnsq;
    std::vector<float> svf;
    std::vector<uint8_t> svq;
    std::vector<int8_t> svd;
};
int rml_svm_pack_host(const double* sv, int64_t M, int64_t D, const double* dual_coef, const int32_t* n_support,
                      int n_classes, int kernel, double gamma, double code_scale, bool has_calib, rml_svm* m, rml_svm_pack* pk);

struct rml_linear {
    int64_t D = 0;
    int C = 0;
    bool has_calib = false;
    double* coef = nullptr;       // C x D
    double* intercept = nullptr;  // C
    double* calib = nullptr;      // 2*C
};
