// Experiment: what does hipExtStreamCreateWithCUMask select on MI355X, and does partitioning the CUs between an
// HBM-streaming kernel and an MFMA kernel beat letting them fight?   (not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void k_where(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        out[blockIdx.x] = (xcc << 16) | ((hw >> 8) & 0xFF);      // cu_id[3:0], sh_id, se_id
        // keep the CU busy a little so blocks spread
        for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
    }
}

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_mfma(int* out, int iters) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    v16i acc0 = {0}, acc1 = {0}, acc2 = {0}, acc3 = {0};
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc3, 0, 0, 0);
    }
    if (acc0[0] + acc1[1] + acc2[2] + acc3[3] == 123456789) out[0] = 1;
}

__global__ __launch_bounds__(256) void k_stream(const float4* __restrict__ V, size_t per_block4, float* out) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f* p = reinterpret_cast<const v4f*>(V) + (size_t)blockIdx.x * per_block4;
    v4f m = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < per_block4; i += 256 * 8) {
        v4f v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + i + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) m = __builtin_elementwise_max(m, v[u]);
    }
    if (m.x + m.y + m.z + m.w == 12345.f) out[blockIdx.x] = m.x;
}

float elapsed(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    unsigned* d; CK(hipMalloc(&d, 8192 * 4));
    auto census = [&](hipStream_t st, const char* label) {
        CK(hipMemset(d, 0xFF, 8192 * 4));
        hipLaunchKernelGGL(k_where, dim3(4096), dim3(64), 0, st, d);
        CK(hipStreamSynchronize(st));
        std::vector<unsigned> h(4096); CK(hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost));
        std::map<unsigned, int> per_xcc; std::map<unsigned, int> cus;
        for (unsigned v : h) { per_xcc[v >> 16]++; cus[v]++; }
        printf("%s: distinct (xcc,se,sh,cu) = %zu; per xcc:", label, cus.size());
        for (auto& kv : per_xcc) { int n = 0; for (auto& c : cus) if ((c.first >> 16) == kv.first) ++n; printf(" x%u:%dCU", kv.first, n); }
        printf("\n");
    };
    census(0, "unmasked");
    auto mk = [&](std::vector<uint32_t> m) { hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data())); return s; };
    hipStream_t s32 = mk({0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0});
    census(s32, "mask word0 = all ones (32 bits)");
    hipStream_t s8 = mk({0xFFu, 0, 0, 0, 0, 0, 0, 0});
    census(s8, "mask bits 0..7");
    hipStream_t sEach = mk({0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu, 0x0F0F0F0Fu});
    census(sEach, "mask 0x0F0F0F0F x8 (128 bits)");
    hipStream_t sHi = mk({0, 0, 0, 0, 0, 0, 0, 0xFFFFFFFFu});
    census(sHi, "mask word7 = all ones");
    hipStream_t s1w = mk({0x0000FFFFu});
    census(s1w, "mask size 1 word 0x0000FFFF");

    // ---- contention experiment: streaming (HBM) + MFMA, shared vs partitioned ----------------------
    const size_t B = 2048, frame4 = 64 * 64 * 32;      // 4 GiB
    float4* V; CK(hipMalloc(&V, B * frame4 * 16)); CK(hipMemset(V, 0, B * frame4 * 16));
    float* o; CK(hipMalloc(&o, B * 4));
    int* oi; CK(hipMalloc(&oi, 64));
    hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    auto run = [&](hipStream_t sa, hipStream_t sb, int mf_blocks, int mf_iters, const char* label) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, sa));
            hipLaunchKernelGGL(k_stream, dim3(B), dim3(256), 0, sa, V, frame4, o);
            CK(hipEventRecord(e1, sa));
            CK(hipEventRecord(e2, sb));
            if (mf_blocks) hipLaunchKernelGGL(k_mfma, dim3(mf_blocks), dim3(256), 0, sb, oi, mf_iters);
            CK(hipEventRecord(e3, sb));
            CK(hipDeviceSynchronize());
            if (rep == 2) printf("%-48s stream %.3f ms (%.0f GB/s)  mfma %.3f ms\n", label, elapsed(e0, e1), B * frame4 * 16 / elapsed(e0, e1) / 1e6, elapsed(e2, e3));
        }
    };
    hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
    run(a, b, 0, 0, "stream alone");
    run(a, b, 2048, 20000, "stream + mfma(2048 blk), unpartitioned");
    // partition: per-XCD split is what we want; try low 48 CUs-per-... using the interleaved hypothesis
    std::vector<uint32_t> mg(8, 0), mp(8, 0);
    for (int bit = 0; bit < 256; ++bit) { bool g = (bit % 32) < 6; (g ? mg : mp)[bit / 32] |= 1u << (bit % 32); }
    hipStream_t sg = mk(mg), sp = mk(mp);
    census(sg, "gemm mask (bit%32 < 6)"); census(sp, "proj mask (complement)");
    run(sp, sg, 2048, 20000 * 48 / 256, "partitioned: stream on 208, mfma on 48 (scaled work)");
    run(sp, sg, 0, 0, "stream alone on 208 CUs");
    run(a, b, 2048, 20000 * 48 / 256, "unpartitioned, same reduced mfma work");
    return 0;
}
