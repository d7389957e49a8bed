// Experiment harness (not part of the product): do wave-private frames whose size is a power of two (64 x 64 x 128 f32 = 2 MiB, uint8 =
// 512 KiB) alias in the memory system when every wave starts its frame at offset 0 and the waves advance in lock step?  A persistent
// grid of 256 x 4 waves streams frames with flat 1 KB non-temporal loads, 8 in flight per lane, either from offset 0 or from a
// per-frame start plane (wrapping), for frame strides of exactly 2 MiB / 512 KiB and for the same plus 4 KiB.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_alias.hip -o tools/exp/exp_alias && tools/exp/exp_alias
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <math.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float4* p) { v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x,b.x), fmaxf(a.y,b.y), fmaxf(a.z,b.z), fmaxf(a.w,b.w)); }

// nl = 1 KB loads per frame, strideq = quads between frame starts, planes = start positions (nl % planes == 0)
template <int PF>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ V, int64_t B, int nl, int64_t strideq, int planes, int stagger, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const int64_t nunits = (int64_t)gridDim.x * 4;
    const int lpp = nl / planes;
    for (int64_t f = (int64_t)blockIdx.x * 4 + wave; f < B; f += nunits) {
        const float4* Vb = V + f * strideq;
        int t = stagger ? (int)((f * 5) % planes) * lpp : 0;
        for (int n = 0; n < nl; n += PF) {
            float4 v[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                v[u] = ldnt(Vb + (int64_t)t * 64 + lane);
                ++t; t = t == nl ? 0 : t;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) m = max4(m, v[u]);
        }
    }
    if (m.x + m.y + m.z + m.w == 12345.678f) out[0] = m.x;
}

int main() {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    float* out; CK(hipMalloc(&out, 64));
    struct Case { const char* name; int64_t frame_bytes; int64_t stride_bytes; int planes; };
    const Case cases[] = {{"2 MiB frames, stride 2 MiB", 2 << 20, 2 << 20, 64}, {"2 MiB frames, stride 2 MiB + 4 KiB", 2 << 20, (2 << 20) + 4096, 64},
                          {"512 KiB frames, stride 512 KiB", 512 << 10, 512 << 10, 64}, {"512 KiB frames, stride 512 KiB + 4 KiB", 512 << 10, (512 << 10) + 4096, 64}};
    for (const Case& c : cases) {
        const int64_t B = (int64_t)(8ll << 30) / c.stride_bytes;        // 8 GiB of frames
        float4* V; CK(hipMalloc(&V, (size_t)B * c.stride_bytes + 4096));
        CK(hipMemset(V, 0, (size_t)B * c.stride_bytes));
        const int nl = (int)(c.frame_bytes / 1024);
        for (int stag = 0; stag < 2; ++stag) {
            std::vector<float> ms;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 7; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_stream<8>, dim3(cus), dim3(256), 0, 0, V, B, nl, c.stride_bytes / 16, c.planes, stag, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("%-40s start %-9s %8.3f ms  %7.1f GB/s (median of 7; min %.3f)\n", c.name, stag ? "staggered" : "offset 0", ms[3],
                   (double)B * c.frame_bytes / ms[3] / 1e6, ms[0]);
        }
        CK(hipFree(V));
    }
    return 0;
}
