// Experiment: per-CU bandwidth of streaming L2-resident data into registers / LDS (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

// each block repeatedly streams `span` bytes (L2 resident when small) ; UN loads in flight per lane
template <int UN, bool NT>
__global__ __launch_bounds__(256) void k_regs(const v4i* __restrict__ src, size_t span16, int iters, int* out) {
    v4i acc = {0, 0, 0, 0};
    const size_t base = ((size_t)blockIdx.x * 7919) % (span16 / (256 * UN)) * (256 * UN);
    for (int it = 0; it < iters; ++it) {
        size_t o = (base + (size_t)it * 256 * UN) % span16;
        v4i v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = NT ? __builtin_nontemporal_load(src + o + u * 256 + threadIdx.x) : src[o + u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < UN; ++u) acc ^= v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 0x12345678) out[0] = 1;
}

// LDS-DMA version: STAGES x 16 KiB ring per block, barrier per stage (like the GEMM)
template <int STAGES, int KB>
__global__ __launch_bounds__(256) void k_glds(const unsigned char* __restrict__ src, size_t span, int iters, int* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER_WAVE = KB / 4;        // 1 KiB instructions per wave per stage (KB KiB per stage, 4 waves)
    const size_t nblk = span / (KB * 1024);
    size_t blk = ((size_t)blockIdx.x * 7919) % nblk;
    auto stage = [&](int buf) {
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) {
            const unsigned char* g = src + blk * (KB * 1024) + ((wave * PER_WAVE + q) * 64 + lane) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                (__attribute__((address_space(3))) void*)(smem + buf * KB * 1024 + (wave * PER_WAVE + q) * 1024), 16, 0, 0);
        }
        blk = blk + 1 >= nblk ? 0 : blk + 1;
    };
    for (int s = 0; s < STAGES - 1; ++s) stage(s);
    int acc = 0, buf = 0;
    for (int it = 0; it < iters; ++it) {
        if (STAGES == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (STAGES == 3) { if (PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else { if (PER_WAVE == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        int nb = buf + STAGES - 1; nb = nb >= STAGES ? nb - STAGES : nb;
        stage(nb);
        acc += *reinterpret_cast<const int*>(smem + buf * KB * 1024 + threadIdx.x * 16);
        buf = buf + 1 >= STAGES ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678) out[0] = 1;
}

template <typename F> float bench(F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int i = 0; i < 5; ++i) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end()); return t[2];
}

int main() {
    const size_t total = 1ull << 30;
    unsigned char* buf; CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total));
    int* out; CK(hipMalloc(&out, 64));
    for (size_t span : {(size_t)16 << 20, (size_t)128 << 20, (size_t)1 << 30}) {
        printf("---- span %zu MiB\n", span >> 20);
        const int blocks = 512, iters = 2000;
#define RR(UN, NT) { float ms = bench([&] { hipLaunchKernelGGL((k_regs<UN, NT>), dim3(blocks), dim3(256), 0, 0, (const v4i*)buf, span / 16, iters / UN * 4, out); }); \
            double bytes = (double)blocks * (iters / UN * 4) * 256 * UN * 16; printf("regs UN=%d nt=%d: %.3f ms  %.1f TB/s  %.1f GB/s/CU\n", UN, NT, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256); }
        RR(4, false) RR(8, false) RR(16, false) RR(8, true)
#define RG(ST, KB) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_glds<ST, KB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            float ms = bench([&] { hipLaunchKernelGGL((k_glds<ST, KB>), dim3(blocks), dim3(256), ST * KB * 1024, 0, buf, span, iters, out); }); \
            double bytes = (double)blocks * iters * KB * 1024; printf("glds stages=%d x %d KiB: %.3f ms  %.1f TB/s  %.1f GB/s/CU\n", ST, KB, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256); }
        RG(2, 16) RG(3, 16) RG(4, 16) RG(2, 32) RG(3, 32)
    }
    return 0;
}
