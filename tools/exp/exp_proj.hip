// Experiment harness: ablations of the max-projection streaming kernel (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_proj.hip -o /tmp/exp_proj && /tmp/exp_proj
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <math.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int X = 64, Y = 64, Z = 128, ZQ = 32;

typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float4* p) { v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float4 max4(float4 a, float4 b) { return make_float4(fmaxf(a.x,b.x), fmaxf(a.y,b.y), fmaxf(a.z,b.z), fmaxf(a.w,b.w)); }

// plain streaming read: grid-stride float4 max into one value per block
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ V, size_t n4, float* out) {
    float4 m = make_float4(-INFINITY,-INFINITY,-INFINITY,-INFINITY);
    size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = ldnt(V + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) m = max4(m, v[u]);
    }
    float r = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));
    if (r == 12345.678f) out[blockIdx.x] = r;
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// max over the 32 lanes of a half-wave; lanes 16..31 (48..63) end with the result
__device__ __forceinline__ float rowmax32_dpp(float r) {
    r = fmaxf(r, dppf<0xB1>(r));        // quad_perm [1,0,3,2]
    r = fmaxf(r, dppf<0x4E>(r));        // quad_perm [2,3,0,1]
    r = fmaxf(r, dppf<0x141>(r));       // row_half_mirror
    r = fmaxf(r, dppf<0x140>(r));       // row_mirror
    r = fmaxf(r, dppf<0x142, 0xA>(r));  // row_bcast15 into rows 1 and 3
    return r;
}

// frame-per-block streaming, contiguous per block (the product kernel's access pattern), no reductions kept
template <int NM, int PF, bool DO_YZ, bool DO_XZ, bool DO_XY, int WAVES, bool DPP = false>
__global__ __launch_bounds__(WAVES * 64) void k_frame(const float4* __restrict__ V, float* out, float* outxy) {
    extern __shared__ float lds[];
    float* xz_lds = lds; float* xy_lds = lds + X * Z;
    constexpr int T = WAVES * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NSLOT = WAVES * 2;
    const int slot = wave * 2 + (lane >> 5), kq = lane & 31;
    const float4* Vb = V + (size_t)blockIdx.x * X * Y * ZQ;
    if (DO_XZ) { for (int i = tid; i < X * Z; i += T) xz_lds[i] = -INFINITY; __syncthreads(); }
    float4 yz[NM];
    int roff[NM];
    const float4 id4 = make_float4(-INFINITY,-INFINITY,-INFINITY,-INFINITY);
#pragma unroll
    for (int m = 0; m < NM; ++m) { yz[m] = id4; roff[m] = (slot + NSLOT * m) * ZQ + kq; }
    constexpr int plane = Y * ZQ;
    float4 buf[PF][NM];
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)
#pragma unroll
        for (int m = 0; m < NM; ++m) buf[p][m] = ldnt(Vb + (size_t)p * plane + roff[m]);
    float4 acc = id4;
#pragma unroll 1
    for (int i0 = 0; i0 < X; i0 += PF) {
#pragma unroll
        for (int pp = 0; pp < PF; ++pp) {
            const int i = i0 + pp;
            const int nxt = i + PF - 1;
            if (nxt < X) {
#pragma unroll
                for (int m = 0; m < NM; ++m) buf[(pp + PF - 1) % PF][m] = ldnt(Vb + (size_t)nxt * plane + roff[m]);
            }
            float4 p = id4;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                float4 c = buf[pp][m];
                if (DO_YZ) yz[m] = max4(yz[m], c);
                p = max4(p, c);
                if (DO_XY) {
                    float r = fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, c.w));
                    if (DPP) {
                        r = rowmax32_dpp(r);
                        if (kq == 31) xy_lds[i * Y + slot + NSLOT * m] = r;
                    } else {
#pragma unroll
                        for (int off = 16; off >= 1; off >>= 1) r = fmaxf(r, __shfl_xor(r, off));
                        if (kq == 0) xy_lds[i * Y + slot + NSLOT * m] = r;
                    }
                }
            }
            if (DO_XZ && DPP) {
                // v_permlane32_swap: upper half of a <-> lower half of b; one swap + one max combines two components
                auto sw = [](float a, float b) {
                    auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                    return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1]));
                };
                float xy_ = sw(p.x, p.y);     // lanes 0-31: x combined, lanes 32-63: y combined
                float zw_ = sw(p.z, p.w);     // lanes 0-31: z combined, lanes 32-63: w combined
                float* d = xz_lds + i * Z + 4 * kq + (lane >> 5);
                __hip_atomic_fetch_max(d + 0, xy_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_max(d + 2, zw_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (DO_XZ) {
                float4 q;
                q.x = __shfl_xor(p.x, 32); q.y = __shfl_xor(p.y, 32); q.z = __shfl_xor(p.z, 32); q.w = __shfl_xor(p.w, 32);
                p = max4(p, q);
                if (lane < 32) {
                    float* d = xz_lds + i * Z + 4 * kq;
                    __hip_atomic_fetch_max(d + 0, p.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(d + 1, p.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(d + 2, p.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_max(d + 3, p.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            } else {
                acc = max4(acc, p);
            }
        }
    }
    // outputs: keep everything live, write like the product does
    float* o = out + (size_t)blockIdx.x * (X * Z + Y * Z + X * Y);
    if (DO_YZ) {
#pragma unroll
        for (int m = 0; m < NM; ++m) *reinterpret_cast<float4*>(o + X * Z + (slot + NSLOT * m) * Z + 4 * kq) = yz[m];
    }
    if (DO_XZ || DO_XY) __syncthreads();
    if (DO_XZ) for (int i = tid; i < X * Z / 4; i += T) *reinterpret_cast<float4*>(o + i * 4) = *reinterpret_cast<float4*>(xz_lds + i * 4);
    else if (acc.x == 12345.6f) o[tid] = acc.x + acc.y + acc.z + acc.w;
    if (DO_XY) for (int i = tid; i < X * Y / 4; i += T) *reinterpret_cast<float4*>(o + X * Z + Y * Z + i * 4) = *reinterpret_cast<float4*>(xy_lds + i * 4);
}

template <typename F> float bench(F f, int iters = 10) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 2; ++i) f();
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < iters; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t n = (size_t)B * X * Y * Z;
    float* V; CK(hipMalloc(&V, n * 4));
    CK(hipMemset(V, 0, n * 4));
    { std::vector<float> h(1 << 22); for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 24) & 255) * ((i % 61) == 0);
      for (size_t o = 0; o < n; o += h.size()) CK(hipMemcpy(V + o, h.data(), std::min(h.size(), n - o) * 4, hipMemcpyHostToDevice)); }
    float* out; CK(hipMalloc(&out, (size_t)B * 20480 * 4));
    const double gb = n * 4 / 1e9;
    const size_t lds = (X * Z + X * Y) * 4;
#define RUN_STREAM(U, G) { float ms = bench([&] { hipLaunchKernelGGL((k_stream<U>), dim3(G), dim3(256), 0, 0, (const float4*)V, n / 4, out); }); \
        printf("stream unroll=%d grid=%d : %.3f ms %.1f GB/s\n", U, G, ms, gb / ms * 1e3); }
    RUN_STREAM(4, 2048) RUN_STREAM(8, 2048) RUN_STREAM(8, 4096) RUN_STREAM(16, 2048) RUN_STREAM(8, 8192) RUN_STREAM(4, 16384)
#define RUN_FRAME(NM, PF, YZ, XZ, XY, W, ...) { \
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_frame<NM, PF, YZ, XZ, XY, W, ##__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
        float ms = bench([&] { hipLaunchKernelGGL((k_frame<NM, PF, YZ, XZ, XY, W, ##__VA_ARGS__>), dim3(B), dim3(W * 64), lds, 0, (const float4*)V, out, out); }); \
        printf("frame NM=%d PF=%d yz=%d xz=%d xy=%d waves=%d %s: %.3f ms %.1f GB/s (read only)\n", NM, PF, YZ, XZ, XY, W, #__VA_ARGS__, ms, gb / ms * 1e3); }
    RUN_FRAME(8, 1, false, false, false, 4)
    RUN_FRAME(8, 2, false, false, false, 4)
    RUN_FRAME(8, 1, true, false, false, 4)
    RUN_FRAME(8, 1, true, true, false, 4)
    RUN_FRAME(8, 1, true, false, true, 4)
    RUN_FRAME(8, 1, true, true, true, 4)
    RUN_FRAME(8, 2, true, true, true, 4)
    RUN_FRAME(8, 1, true, true, true, 4, true)
    RUN_FRAME(4, 1, true, true, true, 8)
    RUN_FRAME(4, 2, true, true, true, 8)
    RUN_FRAME(4, 1, true, true, true, 8, true)
    RUN_FRAME(2, 2, true, true, true, 16)
    return 0;
}
