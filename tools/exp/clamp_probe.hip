// probe: does gfx950 honour the VOP3 clamp bit on v_cvt_pk_bf16_f32 (float result clamped to [0, 1])?
#include <hip/hip_runtime.h>
__global__ void k_probe(const float* in, unsigned* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a = in[2 * i], b = in[2 * i + 1];
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    out[i] = r;
}
extern "C" int clamp_probe(const float* in, unsigned* out, int n, void* stream) {
    hipLaunchKernelGGL(k_probe, dim3((n / 2 + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, out, n);
    return (int)hipGetLastError();
}
