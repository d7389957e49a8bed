// Experiment harness (not part of the product): which ACCESS PATTERN lets a persistent streaming kernel reach the HBM
// ceiling on small frames (Walabot arena: 22 x 31 x 176 f32 = 480 128 B per frame)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_stream.hip -o /tmp/exp_stream && /tmp/exp_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <math.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float4* p) { v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 ldpl(const float4* p) { return *p; }
__device__ __forceinline__ float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x,b.x), fmaxf(a.y,b.y), fmaxf(a.z,b.z), fmaxf(a.w,b.w)); }
constexpr int FQ = 22 * 31 * 44;        // quads per frame (30 008)

// MODE 0: wave-private frame, 44-lane row loads (704 B); MODE 1: wave-private frame, flat 64-lane loads (1 KB);
// MODE 2: workgroup-shared frame, wave w reads chunk 4t + w (4 KB per workgroup step); MODE 3: CU-pair ... (unused)
// PF = loads in flight per lane (register ring), NT = non-temporal
template <int MODE, int PF, bool NT>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ V, int64_t B, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    const int64_t nunits = MODE == 2 ? gridDim.x : (int64_t)gridDim.x * 4;
    int64_t f = MODE == 2 ? blockIdx.x : (int64_t)blockIdx.x * 4 + wave;
    // per-frame list of load offsets (in quads): n loads, lane offset
    const int nl = MODE == 0 ? 22 * 31 : (MODE == 1 ? (FQ + 63) / 64 : (FQ + 255) / 256);
    for (; f < B; f += nunits) {
        const float4* Vb = V + f * (int64_t)FQ;
        for (int t0 = 0; t0 < nl; t0 += PF) {
            float4 v[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                int t = t0 + u; t = t < nl ? t : nl - 1;
                int q;
                if (MODE == 0) q = t * 44 + (lane < 44 ? lane : 43);
                else if (MODE == 1) q = t * 64 + lane;
                else q = (t * 4 + wave) * 64 + lane;
                q = q < FQ ? q : FQ - 1;
                v[u] = NT ? ldnt(Vb + q) : ldpl(Vb + q);
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) m = max4(m, v[u]);
        }
    }
    float r = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));
    if (r == 12345.678f) out[blockIdx.x] = r;
}

template <int MODE, int PF, bool NT>
void run(const char* name, const float4* V, int64_t B, float* out, int grid) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_stream<MODE, PF, NT>), dim3(grid), dim3(256), 0, 0, V, B, out);
    std::vector<float> ts;
    for (int i = 0; i < 7; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream<MODE, PF, NT>), dim3(grid), dim3(256), 0, 0, V, B, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    printf("%-44s grid %5d  %.4f ms  %.2f TB/s\n", name, grid, ts[3], (double)B * FQ * 16 / ts[3] / 1e9);
}

int main() {
    const int64_t B = 16384;
    float4* V; float* out;
    CK(hipMalloc(&V, B * FQ * 16)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(V, 0, B * FQ * 16));
    for (int grid : {256, 512, 768, 1024}) {
        run<0, 8, true>("rows 704B, wave-private, PF8 nt", V, B, out, grid);
        run<0, 16, true>("rows 704B, wave-private, PF16 nt", V, B, out, grid);
        run<0, 31, true>("rows 704B, wave-private, PF31 nt", V, B, out, grid);
        run<1, 8, true>("flat 1KB, wave-private, PF8 nt", V, B, out, grid);
        run<1, 16, true>("flat 1KB, wave-private, PF16 nt", V, B, out, grid);
        run<1, 32, true>("flat 1KB, wave-private, PF32 nt", V, B, out, grid);
        run<2, 8, true>("flat 4KB/WG, WG-shared frame, PF8 nt", V, B, out, grid);
        run<2, 16, true>("flat 4KB/WG, WG-shared frame, PF16 nt", V, B, out, grid);
        run<2, 16, false>("flat 4KB/WG, WG-shared frame, PF16 plain", V, B, out, grid);
        run<1, 16, false>("flat 1KB, wave-private, PF16 plain", V, B, out, grid);
    }
    return 0;
}
