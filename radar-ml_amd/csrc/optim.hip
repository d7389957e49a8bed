// Adam update of ALL parameters of the SGAN discriminator in one pass (sgan.py:206, 214: Adam(lr=0.0002, beta_1=0.5) inside
// train_on_batch, sgan.py:525-532), with the loss-scale bookkeeping of a half-precision step on the device.
//
// torch's fused Adam walks its tensor lists one 64 K-element chunk per workgroup: the discriminator's 1.86 M parameters in 53
// tensors are 29 workgroups on 256 CUs, twice per update (the list splits), 70 us each -- latency, not bandwidth -- plus a
// multi-tensor pass that looks for non-finite gradients (27 us) and a handful of one-element kernels of the loss scaler:
// profiles/r04_stats_sgan.txt, 0.5 ms of the 8.7 ms three-update step.  Here: three launches per update --
//   k_adam_check   every gradient once, 1 024 elements per workgroup over the virtual concatenation of the tensors: non-finite -> found
//   k_adam_decide  one thread: skip this step? loss-scale backoff / growth (torch.amp.GradScaler's rule), step count, 1 / scale
//   k_adam_apply   the update (torch.optim.Adam's formula, no weight decay, no amsgrad), unscaling the gradient on the fly
// The table of (param, grad, exp_avg, exp_avg_sq, start) lives on the device; tensors only need to be dense with the same
// strides for all four (a channels_last convolution kernel is its storage order).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rml_internal.h"

namespace {

struct AdamEntry {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t start;          // first element of this tensor in the concatenation; entry n holds the total
};

constexpr int kAT = 256;    // threads
constexpr int kPer = 4;     // elements per thread

// entry e with start[e] <= idx < start[e + 1]
__device__ __forceinline__ int find_entry(const AdamEntry* __restrict__ tab, int n, int64_t idx) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].start <= idx) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// The table in LDS (up to kMaxLds tensors; more: searched in global memory): the search is six dependent loads per element
constexpr int kMaxLds = 127;
__device__ __forceinline__ const AdamEntry* stage_table(const AdamEntry* __restrict__ tab, int n, AdamEntry* lds) {
    if (n > kMaxLds) return tab;
    for (int i = threadIdx.x; i <= n; i += kAT) lds[i] = tab[i];
    __syncthreads();
    return lds;
}

// state: int32 [0] found (non-finite gradients seen, reset by k_adam_decide), [1] growth tracker, [2] skip flag of this step
__global__ __launch_bounds__(kAT) void k_adam_check(const AdamEntry* tab, int n, int64_t total, int* state) {
    // element j of a thread sits kAT elements behind element j - 1: neighbouring lanes read neighbouring floats
    __shared__ AdamEntry ltab[kMaxLds + 1];
    tab = stage_table(tab, n, ltab);
    const int64_t base = (int64_t)blockIdx.x * kAT * kPer + threadIdx.x;
    bool bad = false;
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int64_t idx = base + (int64_t)i * kAT;
        if (idx < total) {
            const int e = find_entry(tab, n, idx);
            const float g = tab[e].g[idx - tab[e].start];
            bad |= !(fabsf(g) <= 3.4028234663852886e38f);     // NaN or infinity
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicAdd(state, 1);
}

__global__ void k_adam_decide(int* state, float* scale, float* step, float* inv_scale, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool skip = state[0] != 0;
    state[0] = 0;
    state[2] = skip ? 1 : 0;
    float s = scale ? *scale : 1.0f;
    *inv_scale = 1.0f / s;                               // of the scale THIS backward ran with
    if (scale) {
        if (skip) { s *= backoff; state[1] = 0; }
        else if (++state[1] >= interval) { s *= growth; state[1] = 0; }
        *scale = s;
    }
    if (!skip) *step += 1.0f;
}

__global__ __launch_bounds__(kAT) void k_adam_apply(const AdamEntry* tab, int n, int64_t total, float lr, float beta1, float beta2,
                                                    float eps, const float* __restrict__ step, const float* __restrict__ inv_scale,
                                                    const int* __restrict__ state) {
    if (state[2]) return;                                // a non-finite gradient somewhere: the whole step is skipped (uniform)
    __shared__ AdamEntry ltab[kMaxLds + 1];
    tab = stage_table(tab, n, ltab);
    const int64_t base = (int64_t)blockIdx.x * kAT * kPer + threadIdx.x;
    const float t = *step, inv = *inv_scale;
    // torch.optim.Adam: step_size = lr / (1 - beta1^t), denom = sqrt(v) / sqrt(1 - beta2^t) + eps
    const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t);
    const float step_size = lr / bc1, rsq_bc2 = 1.0f / sqrtf(bc2);
#pragma unroll
    for (int i = 0; i < kPer; ++i) {
        const int64_t idx = base + (int64_t)i * kAT;
        if (idx >= total) break;                       // no barrier below
        const int e = find_entry(tab, n, idx);
        const int64_t o = idx - tab[e].start;
        const float g = tab[e].g[o] * inv;
        const float m = beta1 * tab[e].m[o] + (1.0f - beta1) * g;
        const float v = beta2 * tab[e].v[o] + (1.0f - beta2) * g * g;
        tab[e].m[o] = m;
        tab[e].v[o] = v;
        tab[e].p[o] -= step_size * (m / (sqrtf(v) * rsq_bc2 + eps));
    }
}

}  // namespace

extern "C" int rml_adam_entry_bytes(void) { return (int)sizeof(AdamEntry); }

extern "C" int rml_adam_step(rml_ctx* ctx, const void* table, int n_tensors, int64_t total, float lr, float beta1, float beta2, float eps,
                             float* step, float* scale, int32_t* state, float* inv_scale, int check, float growth, float backoff,
                             int growth_interval, void* stream) {
    RML_REQUIRE(ctx && table && step && state && inv_scale && n_tensors > 0 && total > 0, RML_ERR_INVALID, "rml_adam_step: bad arguments");
    RML_REQUIRE(lr > 0.0f && beta1 >= 0.0f && beta1 < 1.0f && beta2 >= 0.0f && beta2 < 1.0f && eps >= 0.0f, RML_ERR_INVALID,
                "rml_adam_step: bad hyper-parameters");
    RML_REQUIRE(!scale || (growth >= 1.0f && backoff > 0.0f && backoff <= 1.0f && growth_interval > 0), RML_ERR_INVALID,
                "rml_adam_step: bad loss-scale rule");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const AdamEntry* tab = static_cast<const AdamEntry*>(table);
    const unsigned blocks = (unsigned)((total + (int64_t)kAT * kPer - 1) / ((int64_t)kAT * kPer));
    if (check) hipLaunchKernelGGL(k_adam_check, dim3(blocks), dim3(kAT), 0, st, tab, n_tensors, total, state);
    hipLaunchKernelGGL(k_adam_decide, dim3(1), dim3(64), 0, st, state, scale, step, inv_scale, growth, backoff, growth_interval);
    hipLaunchKernelGGL(k_adam_apply, dim3(blocks), dim3(kAT), 0, st, tab, n_tensors, total, lr, beta1, beta2, eps, step, inv_scale, state);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
