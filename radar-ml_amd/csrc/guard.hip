// Device side of the CNN row's margin guard (radar-ml_amd/dnn.py Classifier._guard; reference: the labels of model.predict,
// dnn.py:373-381): the top-2 gap of every probability row, and the bookkeeping of a re-scoring round -- replace the rows, the
// largest change seen, which of the new rows are still near a tie -- each in ONE launch.  Round 5 ran this as ~20 element-wise
// PyTorch launches and five host round trips per 256 rows.
#include "rml_internal.h"
#include <math.h>

namespace {

constexpr int kMaxClasses = 16;

// gap[r] = largest - second largest of row r; 0 ("a tie") when the row holds a non-finite value
__device__ __forceinline__ float top2_gap(const float* __restrict__ p, int C) {
    float m1 = -INFINITY, m2 = -INFINITY;
    bool fin = true;
    for (int c = 0; c < C; ++c) {
        const float v = p[c];
        fin = fin && isfinite(v);
        if (v > m1) { m2 = m1; m1 = v; } else if (v > m2) m2 = v;
    }
    return fin ? m1 - m2 : 0.0f;
}

__global__ __launch_bounds__(256) void k_top2_gap(const float* __restrict__ proba, int64_t ld, int64_t N, int C, float* __restrict__ gap) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < N) gap[r] = top2_gap(proba + r * ld, C);
}

// rows[i] of proba <- fresh[i]; stats[0] = max over the finite rows of |old - new| (float bits: non-negative floats order like
// unsigned integers), stats[1] += rows whose new gap is below thr_close; close[i] = that test; gap[rows[i]] = +inf (re-scored: never
// a candidate again)
__global__ __launch_bounds__(256) void k_guard_apply(float* __restrict__ proba, int64_t ld, int C, const int64_t* __restrict__ rows, int64_t n,
                                                     const float* __restrict__ fresh, float thr_close, float* __restrict__ gap,
                                                     uint32_t* __restrict__ stats, uint8_t* __restrict__ close) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float d = 0.0f;
    int cl = 0;
    if (i < n) {
        const int64_t r = rows[i];
        float* p = proba + r * ld;
        const float* f = fresh + i * C;
        bool fin = true;
        for (int c = 0; c < C; ++c) {
            const float o = p[c], v = f[c];
            fin = fin && isfinite(o) && isfinite(v);
            d = fmaxf(d, fabsf(o - v));
            p[c] = v;
        }
        if (!fin) d = 0.0f;                     // a NaN row (mode "max_nan") says nothing about the chain's error
        cl = top2_gap(f, C) < thr_close;
        close[i] = (uint8_t)cl;
        if (gap) gap[r] = INFINITY;
    }
    // one atomic pair per wave
    for (int o = 32; o > 0; o >>= 1) {
        d = fmaxf(d, __shfl_xor(d, o));
        cl += __shfl_xor(cl, o);
    }
    if ((threadIdx.x & 63) == 0) {
        if (d > 0.0f) atomicMax(stats, __float_as_uint(d));
        if (cl) atomicAdd(stats + 1, (uint32_t)cl);
    }
}

}  // namespace

extern "C" int rml_dnn_top2_gap(rml_ctx* ctx, const float* proba, int64_t ld, int64_t N, int C, float* gap, void* stream) {
    RML_REQUIRE(ctx && N >= 0 && C >= 2 && C <= kMaxClasses && ld >= C, RML_ERR_INVALID, "rml_dnn_top2_gap: bad arguments (2 <= C <= %d)", kMaxClasses);
    if (N == 0) return RML_OK;
    RML_REQUIRE(proba && gap, RML_ERR_INVALID, "rml_dnn_top2_gap: NULL argument");
    RML_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_top2_gap, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), proba, ld, N, C, gap);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

extern "C" int rml_dnn_guard_apply(rml_ctx* ctx, float* proba, int64_t ld, int C, const int64_t* rows, int64_t n, const float* fresh,
                                   float thr_close, float* gap, uint32_t* stats, uint8_t* close, void* stream) {
    RML_REQUIRE(ctx && n >= 0 && C >= 2 && C <= kMaxClasses && ld >= C, RML_ERR_INVALID, "rml_dnn_guard_apply: bad arguments (2 <= C <= %d)", kMaxClasses);
    if (n == 0) return RML_OK;
    RML_REQUIRE(proba && rows && fresh && stats && close, RML_ERR_INVALID, "rml_dnn_guard_apply: NULL argument");
    RML_HIP(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_guard_apply, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), proba, ld, C, rows, n,
                       fresh, thr_close, gap, stats, close);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
