// Synthetic radar return volumes generated directly in HBM (bench / large-scale tests).
//
// Model of the real data described in SURVEY.md §4/§8d (the 491 real samples in the
// reference's ground_truth_samples.log): background exactly 0, 1-3 separable Gaussian blobs
// per frame, peak amplitude U[76,255], values rounded to integers, values below 13 set to 0,
// clipped to 255; blob size grows with the class index so a classifier has signal.
// Counter-based hashing (seed, frame, stream) makes every frame independent of the batch it
// is generated in, so ranks can generate disjoint frame slabs of one global data set.
#include "rml_internal.h"

namespace {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float u01(uint32_t seed, uint32_t frame, uint32_t stream) {
    uint32_t h = mix32(seed ^ mix32(frame * 0x9E3779B1u + stream * 0x85EBCA77u + 0x165667B1u));
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// Per-frame blob parameters -> three separable profiles per blob in LDS (amplitude folded
// into the x profile).  Returns the frame's class.
__device__ int fill_profiles(float* prof, uint32_t seed_lo, uint32_t seed_hi, int64_t gframe, int X, int Y, int Z, int n_classes) {
    const uint32_t frame = (uint32_t)gframe;
    const uint32_t seed = seed_lo ^ mix32(seed_hi + 0x27D4EB2Fu) ^ mix32((uint32_t)(gframe >> 32));
    const int cls = (int)(mix32(seed ^ mix32(frame * 0x9E3779B1u + 0x51ED27u)) % (uint32_t)n_classes);
    const int nblob = 1 + (cls % 3);
    const float size = 0.6f + 0.4f * ((float)cls + 1.0f);
    const int L = X + Y + Z;
    for (int t = threadIdx.x; t < 3 * L; t += 256) {
        int bl = t / L, r = t - bl * L;
        float c, s;
        int idx;
        if (r < X) { idx = r; c = u01(seed, frame, 16 * bl + 0) * (X - 1); s = (1.0f + u01(seed, frame, 16 * bl + 3) * 1.5f) * size; }
        else if (r < X + Y) { idx = r - X; c = u01(seed, frame, 16 * bl + 1) * (Y - 1); s = (1.0f + u01(seed, frame, 16 * bl + 4) * 1.5f) * size; }
        else { idx = r - X - Y; c = u01(seed, frame, 16 * bl + 2) * (Z - 1); s = (3.0f + u01(seed, frame, 16 * bl + 5) * 7.0f) * size; }
        float d = ((float)idx - c) / s;
        float g = __expf(-0.5f * d * d);
        if (r < X) {
            float amp = floorf(76.0f + u01(seed, frame, 16 * bl + 6) * 179.0f);
            g *= (bl < nblob) ? amp : 0.0f;
        }
        prof[t] = g;
    }
    __syncthreads();
    return cls;
}

__global__ __launch_bounds__(256) void k_synth(uint32_t seed_lo, uint32_t seed_hi, int64_t frame0, int X, int Y, int Z,
                                               int n_classes, float* V, int32_t* cls_out) {
    extern __shared__ float prof[];   // 3 blobs * (X + Y + Z)
    const int64_t b = blockIdx.x;
    const int L = X + Y + Z;
    const int cls = fill_profiles(prof, seed_lo, seed_hi, frame0 + b, X, Y, Z, n_classes);
    if (threadIdx.x == 0 && cls_out) cls_out[b] = cls;
    float4* Vb = reinterpret_cast<float4*>(V + b * (int64_t)X * Y * Z);
    const int ZQ = Z >> 2;
    const int64_t total = (int64_t)X * Y * ZQ;
    for (int64_t t = threadIdx.x; t < total; t += 256) {
        int kq = (int)(t % ZQ);
        int64_t row = t / ZQ;
        int j = (int)(row % Y), i = (int)(row / Y);
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int kk = kq * 4 + c;
            float v = 0.0f;
#pragma unroll
            for (int bl = 0; bl < 3; ++bl) {
                const float* p = prof + bl * L;
                v = fmaxf(v, p[i] * p[X + j] * p[X + Y + kk]);
            }
            v = floorf(v + 0.5f);
            v = fminf(v, 255.0f);
            o[c] = v < 13.0f ? 0.0f : v;
        }
        Vb[t] = make_float4(o[0], o[1], o[2], o[3]);
    }
    // tail when Z % 4 != 0 is handled by the scalar kernel below
}

__global__ __launch_bounds__(256) void k_synth_scalar(uint32_t seed_lo, uint32_t seed_hi, int64_t frame0, int X, int Y, int Z,
                                                      int n_classes, float* V, int32_t* cls_out) {
    extern __shared__ float prof[];
    const int64_t b = blockIdx.x;
    const int L = X + Y + Z;
    const int cls = fill_profiles(prof, seed_lo, seed_hi, frame0 + b, X, Y, Z, n_classes);
    if (threadIdx.x == 0 && cls_out) cls_out[b] = cls;
    float* Vb = V + b * (int64_t)X * Y * Z;
    const int64_t total = (int64_t)X * Y * Z;
    for (int64_t t = threadIdx.x; t < total; t += 256) {
        int kk = (int)(t % Z);
        int64_t row = t / Z;
        int j = (int)(row % Y), i = (int)(row / Y);
        float v = 0.0f;
        for (int bl = 0; bl < 3; ++bl) {
            const float* p = prof + bl * L;
            v = fmaxf(v, p[i] * p[X + j] * p[X + Y + kk]);
        }
        v = floorf(v + 0.5f);
        v = fminf(v, 255.0f);
        Vb[t] = v < 13.0f ? 0.0f : v;
    }
}

}  // namespace

extern "C" int rml_synth_volumes(rml_ctx* ctx, uint64_t seed, int64_t frame0, int64_t B, int X, int Y, int Z,
                                 int n_classes, float* V, int32_t* cls, void* stream) {
    if (ctx && B == 0) return RML_OK;
    RML_REQUIRE(ctx && V && B >= 0 && X > 0 && Y > 0 && Z > 0 && n_classes > 0, RML_ERR_INVALID, "rml_synth_volumes: bad arguments");
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_synth_volumes: B too large for one launch");
    RML_HIP(hipSetDevice(ctx->device));
    if (B == 0) return RML_OK;
    size_t lds = (size_t)3 * (X + Y + Z) * sizeof(float);
    RML_REQUIRE(lds <= 60 * 1024, RML_ERR_UNSUPPORTED, "rml_synth_volumes: grid too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    bool vec = (Z % 4 == 0) && ((reinterpret_cast<uintptr_t>(V) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(k_synth, dim3((unsigned)B), dim3(256), lds, st, (uint32_t)seed, (uint32_t)(seed >> 32), frame0, X, Y, Z, n_classes, V, cls);
    else
        hipLaunchKernelGGL(k_synth_scalar, dim3((unsigned)B), dim3(256), lds, st, (uint32_t)seed, (uint32_t)(seed >> 32), frame0, X, Y, Z, n_classes, V, cls);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
