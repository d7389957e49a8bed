// Fused training-mode BatchNorm + LeakyReLU + TensorFlow-'same' zero pad for the SGAN discriminator branches
// (sgan.py:137-158: Conv2D -> BatchNormalization -> LeakyReLU(0.2), three times per branch) on NHWC half tensors.
//
// In PyTorch these are separate passes over the layer activation (268 MB for layer 1 at batch 256): batch-norm
// statistics, normalise, LeakyReLU, the pad copy in front of the next stride-2 convolution, and the mirror images of all
// of them in backward -- 12 of the step's 19 ms of GPU time.  Here the forward is two passes (per-channel sums; then
// normalise + LeakyReLU written straight into the padded layout the next convolution reads) and the backward two
// (per-channel sums of g and g*x_hat with g = dy * leaky'(z); then dx), all HBM-bound streaming over 16-byte
// (8-channel) lanes.  Statistics are float32 partial sums per workgroup, combined in a fixed order in float64:
// deterministic.  Algorithmic bytes per element (2-byte activations): forward 2 + 2 + 2, backward 2*(2 + 2) + 2.
//
// Semantics = torch.nn.BatchNorm2d(training) (biased variance for the normalisation, unbiased for the running
// estimate, momentum m: running = (1-m)*running + m*batch) followed by LeakyReLU(slope) and F.pad(0, pw, 0, ph).
#include "rml_internal.h"

namespace {

constexpr int kT = 256;

template <bool BF> __device__ __forceinline__ float h2f(uint16_t h);
template <> __device__ __forceinline__ float h2f<true>(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
template <> __device__ __forceinline__ float h2f<false>(uint16_t h) {
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
}
template <bool BF> __device__ __forceinline__ uint16_t f2h(float f);
template <> __device__ __forceinline__ uint16_t f2h<true>(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
template <> __device__ __forceinline__ uint16_t f2h<false>(float f) {
    _Float16 v = (_Float16)f;
    uint16_t h;
    __builtin_memcpy(&h, &v, 2);
    return h;
}

template <bool BF> __device__ __forceinline__ void unpack8(const uint4& q, float (&v)[8]) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = h2f<BF>((uint16_t)(w[i] & 0xFFFFu)); v[2 * i + 1] = h2f<BF>((uint16_t)(w[i] >> 16)); }
}
template <bool BF> __device__ __forceinline__ uint4 pack8(const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (uint32_t)f2h<BF>(v[2 * i]) | ((uint32_t)f2h<BF>(v[2 * i + 1]) << 16);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// workgroup-level sum of per-thread [2][8] partials over the threads that share a channel group -> part[blockIdx][2][C]
__device__ __forceinline__ void block_channel_sums(const float (&a)[8], const float (&b)[8], int CG, int C, float* part, float* lds) {
    const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG, PL = kT / CG;
    float* mine = lds + (size_t)tid * 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) { mine[i] = a[i]; mine[8 + i] = b[i]; }
    __syncthreads();
    // thread (cg, j) for j < 16 sums column j of its channel group over the PL pixel lanes
    for (int item = tid; item < CG * 16; item += kT) {
        const int g = item / 16, j = item - g * 16;
        float s = 0.0f;
        for (int p = 0; p < PL; ++p) s += lds[(size_t)(p * CG + g) * 16 + j];
        part[(size_t)blockIdx.x * 2 * C + (j >> 3) * C + g * 8 + (j & 7)] = s;
    }
    (void)cg; (void)pl;
}

// ---- forward pass 1: per-channel sum and sum of squares ---------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(kT) void k_bn_stats(const uint16_t* __restrict__ x, int64_t M, int C, float* part) {
    __shared__ float lds[kT * 16];
    const int CG = C >> 3, PL = kT / CG;
    const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.0f; q[i] = 0.0f; }
    if (pl < PL) {
        for (int64_t p = (int64_t)blockIdx.x * PL + pl; p < M; p += (int64_t)gridDim.x * PL) {
            float v[8];
            unpack8<BF>(*reinterpret_cast<const uint4*>(x + p * C + cg * 8), v);
#pragma unroll
            for (int i = 0; i < 8; ++i) { s[i] += v[i]; q[i] = fmaf(v[i], v[i], q[i]); }
        }
    }
    block_channel_sums(s, q, CG, C, part, lds);
}

// ---- combine the workgroup partials in a fixed order (float64) -----------------------------------------------------
// mode 0 (forward): a = sum x, b = sum x^2 -> mean, rstd, running statistics.  mode 1 (backward): a = sum g,
// b = sum g*x_hat -> dbeta, dgamma, c1 = a/M, c2 = b/M.
__global__ __launch_bounds__(kT) void k_bn_finalize(const float* part, int G, int C, double M, int mode, float eps, float momentum,
                                                    float* o0, float* o1, float* o2, float* o3) {
    // one workgroup per channel: strided float64 partial sums, then a fixed-shape tree
    __shared__ double sa[kT], sb[kT];
    const int c = blockIdx.x, t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int g = t; g < G; g += kT) { a += (double)part[(size_t)g * 2 * C + c]; b += (double)part[(size_t)g * 2 * C + C + c]; }
    sa[t] = a; sb[t] = b;
    __syncthreads();
    for (int off = kT / 2; off >= 1; off >>= 1) {
        if (t < off) { sa[t] += sa[t + off]; sb[t] += sb[t + off]; }
        __syncthreads();
    }
    if (t != 0) return;
    a = sa[0]; b = sb[0];
    if (mode == 0) {
        const double mean = a / M;
        double var = b / M - mean * mean;
        var = var > 0.0 ? var : 0.0;
        o0[c] = (float)mean;
        o1[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (o2) o2[c] = (float)((1.0 - (double)momentum) * (double)o2[c] + (double)momentum * mean);                     // running_mean
        if (o3) o3[c] = (float)((1.0 - (double)momentum) * (double)o3[c] + (double)momentum * var * (M > 1.0 ? M / (M - 1.0) : 1.0));
    } else {
        o0[c] = (float)a;            // dbeta
        o1[c] = (float)b;            // dgamma
        o2[c] = (float)(a / M);
        o3[c] = (float)(b / M);
    }
}

// ---- forward pass 2: normalise + LeakyReLU, written into the (H+ph) x (W+pw) padded layout ---------------------------
template <bool BF>
__global__ __launch_bounds__(kT) void k_bn_apply_pad(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int H, int W, int C,
                                                     int ph, int pw, const float* mean, const float* rstd, const float* gamma,
                                                     const float* beta, float slope) {
    const int CG = C >> 3, Wp = W + pw, Hp = H + ph;
    const int64_t row = blockIdx.x;                 // n * Hp + h
    const int64_t n = row / Hp;
    const int h = (int)(row - n * Hp);
    uint16_t* yr = y + row * (int64_t)Wp * C;
    const uint16_t* xr = x + (n * H + h) * (int64_t)W * C;
    for (int item = threadIdx.x; item < Wp * CG; item += kT) {
        const int w = item / CG, cg = item - w * CG;
        uint4 out = make_uint4(0, 0, 0, 0);
        if (h < H && w < W) {
            float v[8], r[8];
            unpack8<BF>(*reinterpret_cast<const uint4*>(xr + (int64_t)w * C + cg * 8), v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = cg * 8 + i;
                const float z = (v[i] - mean[c]) * rstd[c] * gamma[c] + beta[c];
                r[i] = z > 0.0f ? z : z * slope;
            }
            out = pack8<BF>(r);
        }
        *reinterpret_cast<uint4*>(yr + (int64_t)w * C + cg * 8) = out;
    }
}

// ---- backward pass 1: sum g and sum g * x_hat, g = dy * leaky'(z) --------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(kT) void k_bn_bwd_reduce(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, int64_t N, int H, int W,
                                                      int C, int ph, int pw, const float* mean, const float* rstd, const float* gamma,
                                                      const float* beta, float slope, float* part) {
    __shared__ float lds[kT * 16];
    const int CG = C >> 3, PL = kT / CG, Wp = W + pw, Hp = H + ph;
    const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
    const int64_t M = N * H * W;
    float s[8], q[8], mu[8], rs[8], ga[8], be[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s[i] = 0.0f; q[i] = 0.0f;
        const int c = cg * 8 + i;
        mu[i] = mean[c]; rs[i] = rstd[c]; ga[i] = gamma[c]; be[i] = beta[c];
    }
    if (pl < PL) {
        for (int64_t p = (int64_t)blockIdx.x * PL + pl; p < M; p += (int64_t)gridDim.x * PL) {
            const int64_t nh = p / W;
            const int w = (int)(p - nh * W);
            const int64_t n = nh / H;
            const int h = (int)(nh - n * H);
            float v[8], d[8];
            unpack8<BF>(*reinterpret_cast<const uint4*>(x + p * C + cg * 8), v);
            unpack8<BF>(*reinterpret_cast<const uint4*>(dy + ((n * Hp + h) * (int64_t)Wp + w) * C + cg * 8), d);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (v[i] - mu[i]) * rs[i];
                const float z = xh * ga[i] + be[i];
                const float g = z > 0.0f ? d[i] : d[i] * slope;
                s[i] += g;
                q[i] = fmaf(g, xh, q[i]);
            }
        }
    }
    block_channel_sums(s, q, CG, C, part, lds);
}

// ---- backward pass 2: dx = gamma * rstd * (g - c1 - x_hat * c2) ----------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(kT) void k_bn_bwd_apply(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy, uint16_t* __restrict__ dx,
                                                     int H, int W, int C, int ph, int pw, const float* mean, const float* rstd,
                                                     const float* gamma, const float* beta, const float* c1, const float* c2, float slope) {
    const int CG = C >> 3, Wp = W + pw, Hp = H + ph;
    const int64_t row = blockIdx.x;                 // n * H + h
    const int64_t n = row / H;
    const int h = (int)(row - n * H);
    const uint16_t* xr = x + row * (int64_t)W * C;
    const uint16_t* dr = dy + (n * Hp + h) * (int64_t)Wp * C;
    uint16_t* or_ = dx + row * (int64_t)W * C;
    for (int item = threadIdx.x; item < W * CG; item += kT) {
        const int w = item / CG, cg = item - w * CG;
        float v[8], d[8], r[8];
        unpack8<BF>(*reinterpret_cast<const uint4*>(xr + (int64_t)w * C + cg * 8), v);
        unpack8<BF>(*reinterpret_cast<const uint4*>(dr + (int64_t)w * C + cg * 8), d);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = cg * 8 + i;
            const float xh = (v[i] - mean[c]) * rstd[c];
            const float z = xh * gamma[c] + beta[c];
            const float g = z > 0.0f ? d[i] : d[i] * slope;
            r[i] = gamma[c] * rstd[c] * (g - c1[c] - xh * c2[c]);
        }
        *reinterpret_cast<uint4*>(or_ + (int64_t)w * C + cg * 8) = pack8<BF>(r);
    }
}

// ---- the first layer with its convolution folded in: z is never stored ---------------------------------------------
// z[n,h,w,c] = sum_k wk[k][c] * img[n, 2h+ky, 2w+kx] (3x3, stride 2, 1 input channel, bias-free) costs 9 FMAs per element,
// far less than reading it back from HBM: every pass below recomputes it from the (tiny, cache-resident) image.  A
// thread keeps the 9 x 8 weights of its channel group in registers.
template <bool BF>
struct Conv1 {
    float wk[9][8];
    const uint16_t* img;
    int IH, IW;
    __device__ __forceinline__ void init(const float* wgt /* [9][C] */, int C, int cg, const uint16_t* image, int H, int W) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) wk[k][i] = wgt[k * C + cg * 8 + i];
        img = image; IH = 2 * H + 1; IW = 2 * W + 1;
    }
    __device__ __forceinline__ void taps(int64_t n, int h, int w, float (&t)[9]) const {
        const uint16_t* ip = img + (n * IH + 2 * h) * (int64_t)IW + 2 * w;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) t[ky * 3 + kx] = h2f<BF>(ip[ky * IW + kx]);
    }
    __device__ __forceinline__ void z(const float (&t)[9], float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float a = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) a = fmaf(wk[k][i], t[k], a);
            v[i] = a;
        }
    }
};

template <bool BF>
__global__ __launch_bounds__(kT) void k_c1_apply_pad(const uint16_t* __restrict__ image, const float* __restrict__ wgt, uint16_t* __restrict__ y,
                                                     int64_t nrows, int H, int W, int C, int ph, int pw, const float* mean, const float* rstd,
                                                     const float* gamma, const float* beta, float slope) {
    const int CG = C >> 3, Wp = W + pw, Hp = H + ph;
    // threads keep their channel group (item = w * CG + cg with kT % CG == 0) and the workgroup walks many rows, so the
    // 72 weights and the batch-norm scale/shift of a thread are loaded once
    const int cg = threadIdx.x % CG;
    Conv1<BF> cv;
    cv.init(wgt, C, cg, image, H, W);
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = cg * 8 + i;
        sc[i] = rstd[c] * gamma[c];
        sh[i] = beta[c] - mean[c] * sc[i];
    }
    for (int64_t row = blockIdx.x; row < nrows; row += gridDim.x) {       // row = n * Hp + h
        const int64_t n = row / Hp;
        const int h = (int)(row - n * Hp);
        uint16_t* yr = y + row * (int64_t)Wp * C;
        for (int w = threadIdx.x / CG; w < Wp; w += kT / CG) {
            uint4 out = make_uint4(0, 0, 0, 0);
            if (h < H && w < W) {
                float t[9], v[8], r[8];
                cv.taps(n, h, w, t);
                cv.z(t, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float zz = fmaf(v[i], sc[i], sh[i]);
                    r[i] = zz > 0.0f ? zz : zz * slope;
                }
                out = pack8<BF>(r);
            }
            *reinterpret_cast<uint4*>(yr + (int64_t)w * C + cg * 8) = out;
        }
    }
}

// packed float32 FMAs with one operand broadcast by the instruction (op_sel), not by a register copy
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_mul_lo(f32x2 a, f32x2 p) {       // a * p.lo
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(p));
    return r;
}
__device__ __forceinline__ void pk_fma_lo(f32x2& acc, f32x2 a, f32x2 p) {     // acc += a * p.lo
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(p));
}
__device__ __forceinline__ void pk_fma_hi(f32x2& acc, f32x2 a, f32x2 p) {     // acc += a * p.hi
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(p));
}

// k_c1_apply_pad_pk<BF, NB>: the same pass for C = 128 and rows of W = 8 * NB pixels.  The general kernel above waits for nine
// dependent image loads per pixel and thread with nothing else in flight (3.7-3.9 TB/s of output: latency, not bandwidth); here
// the three image rows of an output row go through LDS (the next row's are loaded while this one is computed), a pixel's taps are
// aligned pairs feeding v_pk_fma_f32 through op_sel, a thread owns four channels (an 8-byte store; two pixels of a wave are 512
// contiguous bytes).  Same products in the same order: bit-identical output.
template <bool BF, int NB>
__global__ __launch_bounds__(kT) void k_c1_apply_pad_pk(const uint16_t* __restrict__ image, const float* __restrict__ wgt, uint16_t* __restrict__ y,
                                                        int64_t nrows, int H, int ph, int pw, const float* mean, const float* rstd,
                                                        const float* gamma, const float* beta, float slope) {
    constexpr int C = 128, CPT = 4, CG = 32, PL = 8, W = NB * PL;
    constexpr int IW = 2 * W + 1, IWp = IW + 1;
    constexpr int NS = (3 * IW + kT - 1) / kT;
    constexpr int BS = 3 * IWp + 2;
    __shared__ __align__(16) float lds[2 * BS];
    const int Wp = W + pw, Hp = H + ph, IH = 2 * H + 1;
    const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
    f32x2 wk[9][2], sc[2], sh[2];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) wk[k][i] = f32x2{wgt[k * C + cg * CPT + 2 * i], wgt[k * C + cg * CPT + 2 * i + 1]};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = cg * CPT + 2 * i;
        sc[i] = f32x2{rstd[c] * gamma[c], rstd[c + 1] * gamma[c + 1]};
        sh[i] = f32x2{beta[c] - mean[c] * sc[i].x, beta[c + 1] - mean[c + 1] * sc[i].y};
    }
    uint16_t img[NS];
    auto load_img = [&](int64_t row) __attribute__((always_inline)) {
        const int64_t n = row / Hp;
        const int h0 = (int)(row - n * Hp), h = h0 < H ? h0 : H - 1;      // a pad row loads a real one (unused)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i0 = tid + s * kT, i = i0 < 3 * IW ? i0 : 3 * IW - 1, ky = i / IW, xx = i - ky * IW;
            img[s] = image[(n * IH + 2 * h + ky) * (int64_t)IW + xx];
        }
    };
    auto stage = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + s * kT, ky = i / IW, xx = i - ky * IW;
            buf[i < 3 * IW ? ky * IWp + xx : 3 * IWp] = h2f<BF>(img[s]);
        }
    };
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    int64_t row = blockIdx.x;
    int cur = 0;
    if (row < nrows) {
        load_img(row);
        stage(lds);
        if (tid < 3) { lds[tid * IWp + IW] = 0.0f; lds[BS + tid * IWp + IW] = 0.0f; }
    }
    __syncthreads();
    for (; row < nrows; row += gridDim.x) {
        const int64_t nxt = row + gridDim.x < nrows ? row + gridDim.x : row;
        load_img(nxt);
        __builtin_amdgcn_sched_barrier(0);
        const int64_t n = row / Hp;
        const int h = (int)(row - n * Hp);
        uint16_t* yr = y + row * (int64_t)Wp * C + cg * CPT;
        if (h < H) {
            const float* buf = lds + cur * BS + 2 * pl;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                f32x2 P[3][2];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    P[ky][0] = *reinterpret_cast<const f32x2*>(buf + ky * IWp + 2 * u * PL);
                    P[ky][1] = *reinterpret_cast<const f32x2*>(buf + ky * IWp + 2 * u * PL + 2);
                }
                u32x2 out;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    f32x2 v = pk_mul_lo(wk[0][i], P[0][0]);
                    pk_fma_hi(v, wk[1][i], P[0][0]); pk_fma_lo(v, wk[2][i], P[0][1]);
                    pk_fma_lo(v, wk[3][i], P[1][0]); pk_fma_hi(v, wk[4][i], P[1][0]); pk_fma_lo(v, wk[5][i], P[1][1]);
                    pk_fma_lo(v, wk[6][i], P[2][0]); pk_fma_hi(v, wk[7][i], P[2][0]); pk_fma_lo(v, wk[8][i], P[2][1]);
                    const f32x2 zz = __builtin_elementwise_fma(v, sc[i], sh[i]);
                    const float r0 = zz.x > 0.0f ? zz.x : zz.x * slope, r1 = zz.y > 0.0f ? zz.y : zz.y * slope;
                    out[i] = (uint32_t)f2h<BF>(r0) | ((uint32_t)f2h<BF>(r1) << 16);
                }
                *reinterpret_cast<u32x2*>(yr + (int64_t)(pl + u * PL) * C) = out;
            }
        } else {
#pragma unroll
            for (int u = 0; u < NB; ++u) *reinterpret_cast<u32x2*>(yr + (int64_t)(pl + u * PL) * C) = u32x2{0u, 0u};
        }
        for (int w = W + pl; w < Wp; w += PL) *reinterpret_cast<u32x2*>(yr + (int64_t)w * C) = u32x2{0u, 0u};     // the pad columns
        __builtin_amdgcn_sched_barrier(0);
        stage(lds + (cur ^ 1) * BS);
        cur ^= 1;
        __syncthreads();
    }
}

// ---- single-pass backward of the first layer -------------------------------------------------------------------------
// dW[k][c] = sum_p dz * t_k with dz = gamma*rstd*(g - c1 - x_hat*c2) expands to
//     gamma*rstd * ( A[k][c] - c1[c]*B[k] - c2[c]*D[k][c] ),
// A[k][c] = sum g*t_k, B[k] = sum t_k, D[k][c] = sum x_hat*t_k = rstd*(sum_j w[j][c]*T2[j][k] - mean*B[k]), T2 = sum t_j*t_k.
// B and T2 depend on the image only (k_c1_imgstats, 8.5 MB), so ONE pass over dy accumulates A, sum g and sum g*x_hat
// (k_c1_bwd1) and k_c1_wgrad_combine finishes in float64 -- dy is read once instead of twice.
// sum over the 64 lanes of a wave on DPP moves (fixed order; the result is wave-uniform) -- see wave_sum in dense.hip
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov_f(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, BOUND));
}
__device__ __forceinline__ float wave_sum_dpp(float r) {
    r += dpp_mov_f<0xB1, 0xF, true>(0.0f, r);     // quad_perm [1,0,3,2]
    r += dpp_mov_f<0x4E, 0xF, true>(0.0f, r);     // quad_perm [2,3,0,1]
    r += dpp_mov_f<0x141, 0xF, true>(0.0f, r);    // row_half_mirror
    r += dpp_mov_f<0x140, 0xF, true>(0.0f, r);    // row_mirror
    r += dpp_mov_f<0x142, 0xA, false>(0.0f, r);   // row_bcast15 into rows 1 and 3
    r += dpp_mov_f<0x143, 0xC, false>(0.0f, r);   // row_bcast31 into rows 2 and 3: row 3 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 63));
}

template <bool BF>
__global__ __launch_bounds__(kT) void k_c1_imgstats(const uint16_t* __restrict__ image, int64_t N, int H, int W, float* part /* [G][54] */) {
    const int IH = 2 * H + 1, IW = 2 * W + 1;
    const int64_t M = N * H * W;
    float b[9], t2[45];
#pragma unroll
    for (int k = 0; k < 9; ++k) b[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 45; ++k) t2[k] = 0.0f;
    for (int64_t p = (int64_t)blockIdx.x * kT + threadIdx.x; p < M; p += (int64_t)gridDim.x * kT) {
        const int64_t nh = p / W;
        const int w = (int)(p - nh * W);
        const int64_t n = nh / H;
        const int h = (int)(nh - n * H);
        const uint16_t* ip = image + (n * IH + 2 * h) * (int64_t)IW + 2 * w;
        float t[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) t[ky * 3 + kx] = h2f<BF>(ip[ky * IW + kx]);
        int q = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            b[j] += t[j];
#pragma unroll
            for (int k = j; k < 9; ++k) { t2[q] = fmaf(t[j], t[k], t2[q]); ++q; }
        }
    }
    // 54 block sums: each wave's on DPP moves (six VALU steps per value; the __shfl_xor butterfly was 324 ds_bpermute round trips
    // per wave -- most of this kernel), then the four waves through LDS (fixed order)
    __shared__ float wsum[kT / 64][54];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int v = 0; v < 54; ++v) {
        const float x = wave_sum_dpp(v < 9 ? b[v] : t2[v - 9]);
        if (lane == 0) wsum[wave][v] = x;
    }
    __syncthreads();
    if (threadIdx.x < 54) {
        float x = 0.0f;
        for (int w = 0; w < kT / 64; ++w) x += wsum[w][threadIdx.x];
        part[(size_t)blockIdx.x * 54 + threadIdx.x] = x;
    }
}

// CPT = channels per thread.  8 (round 2): 250 registers -- 88 accumulators, 72 weights, 32 batch-norm constants -- two waves per
// SIMD, and every output row starts with a dependent global load (the image rows) in front of a barrier: the kernel ran at a third
// of what its arithmetic needs.  4 (round 3): half the accumulators, weights and constants per thread, twice the threads' worth of
// waves per SIMD to hide the loads behind; a pixel's 8-byte gradient loads of 32 lanes are the same contiguous 256 B.  By itself that
// changed nothing (154 vs 153 us per branch); with the row's eight gradient loads issued before the image rows are staged 136 us.
template <bool BF, int CPT>
__global__ __launch_bounds__(kT) void k_c1_bwd1(const uint16_t* __restrict__ image, const float* __restrict__ wgt, const uint16_t* __restrict__ dy,
                                                int64_t N, int H, int W, int C, int ph, int pw, const float* mean, const float* rstd,
                                                const float* gamma, const float* beta, float slope, float* part /* [G][11][C] */) {
    __shared__ float lds[kT * 8];
    const int CG = C / CPT, PL = kT / CG, Wp = W + pw, Hp = H + ph;
    const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
    float wk[9][CPT];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < CPT; ++i) wk[k][i] = wgt[k * C + cg * CPT + i];
    float acc[11][CPT];                 // 0..8: A[k], 9: sum g, 10: sum g*x_hat
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
        for (int i = 0; i < CPT; ++i) acc[k][i] = 0.0f;
    float mu[CPT], rs[CPT], ga[CPT], be[CPT];
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
        const int c = cg * CPT + i;
        mu[i] = mean[c]; rs[i] = rstd[c]; ga[i] = gamma[c]; be[i] = beta[c];
    }
    // one output row per iteration: its three image rows go through LDS (as float), so the nine taps of a pixel are LDS
    // broadcasts instead of nine dependent global loads
    float* rows_s = lds;                             // [3][IW], reused for the reductions afterwards
    const int IW = 2 * W + 1, IH = 2 * H + 1;
    for (int64_t row = blockIdx.x; row < N * H; row += gridDim.x) {
        const int64_t n = row / H;
        const int h = (int)(row - n * H);
        const uint16_t* dyr = dy + ((n * Hp + h) * (int64_t)Wp) * C + cg * CPT;
        // the row's gradient loads go out first (up to NB per thread in flight), then the image rows are staged: the barrier and the
        // staging's own global loads hide under them.  (Issued at their use, each of them was a full memory latency per pixel.)
        constexpr int NB = CPT == 8 ? 1 : 8;        // CPT = 8 has no registers to spare (292 with four loads ahead: one wave per SIMD)
        typedef uint32_t dq_t __attribute__((ext_vector_type(CPT / 2)));
        dq_t dq[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int w = pl + u * PL;
            dq[u] = *reinterpret_cast<const dq_t*>(dyr + (int64_t)(w < W ? w : W - 1) * C);
        }
        __syncthreads();
        for (int i = tid; i < 3 * IW; i += kT) {
            const int ky = i / IW, xx = i - ky * IW;
            rows_s[i] = h2f<BF>(image[(n * IH + 2 * h + ky) * (int64_t)IW + xx]);
        }
        __syncthreads();
        for (int w0 = pl, u0 = 0; w0 < W; w0 += NB * PL, ++u0) {
            if (u0 > 0) {               // rows longer than NB * PL pixels: the next batch of loads (not prefetched)
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int w = w0 + u * PL;
                    dq[u] = *reinterpret_cast<const dq_t*>(dyr + (int64_t)(w < W ? w : W - 1) * C);
                }
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
            const int w = w0 + u * PL;
            if (w >= W) break;
            float t[9], d[CPT];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) t[ky * 3 + kx] = rows_s[ky * IW + 2 * w + kx];
#pragma unroll
            for (int i = 0; i < CPT / 2; ++i) {
                const uint32_t q = dq[u][i];
                d[2 * i] = h2f<BF>((uint16_t)(q & 0xFFFFu)); d[2 * i + 1] = h2f<BF>((uint16_t)(q >> 16));
            }
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                float v = 0.0f;
#pragma unroll
                for (int k = 0; k < 9; ++k) v = fmaf(wk[k][i], t[k], v);
                const float xh = (v - mu[i]) * rs[i];
                const float zz = xh * ga[i] + be[i];
                const float g = zz > 0.0f ? d[i] : d[i] * slope;
                acc[9][i] += g;
                acc[10][i] = fmaf(g, xh, acc[10][i]);
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k][i] = fmaf(g, t[k], acc[k][i]);
            }
            }
        }
    }
    for (int k = 0; k < 11; ++k) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < CPT; ++i) lds[tid * CPT + i] = acc[k][i];
        __syncthreads();
        for (int c = tid; c < C; c += kT) {
            float s = 0.0f;
            for (int q = 0; q < PL; ++q) s += lds[(q * CG + (c / CPT)) * CPT + (c % CPT)];
            part[((size_t)blockIdx.x * 11 + k) * C + c] = s;
        }
    }
}

// k_c1_bwd1_pk<BF, NB>: the same pass for C = 128 and rows of W = 8 * NB pixels, written for the VALU it is bound by.  k_c1_bwd1<., 4>
// issues ~160 vector instructions per pixel and thread of which 36 are the packed FMAs the sums need (rocprof: 134 us per branch
// at configs[4] = the kernel's own instruction count at 2 waves per SIMD; the 268 MB of dy would take 42 us): address arithmetic of
// the nine LDS reads, register copies that duplicate a tap into both halves of a packed operand, row-end tests.  Here a pixel's taps
// arrive as three pairs + three singles (ds_read_b64 on an even row stride) and feed v_pk_fma_f32 through op_sel -- the tap is
// broadcast by the instruction, not by a copy --, the pixel loop has no bounds, and the next row's gradients and image rows are
// in flight (registers / the other LDS buffer) while this row is summed: one barrier per row.
template <bool BF, int NB>
__global__ __launch_bounds__(kT) void k_c1_bwd1_pk(const uint16_t* __restrict__ image, const float* __restrict__ wgt, const uint16_t* __restrict__ dy,
                                                   int64_t N, int H, int ph, int pw, const float* mean, const float* rstd,
                                                   const float* gamma, const float* beta, float slope, float* part /* [G][11][128] */) {
    constexpr int C = 128, CPT = 4, CG = 32, PL = 8, W = NB * PL;
    constexpr int IW = 2 * W + 1, IWp = IW + 1;         // even row stride: a pixel's taps (2w, 2w+1), (2w+2, -) are aligned pairs
    constexpr int NS = (3 * IW + kT - 1) / kT;          // staged image values per thread and row
    __shared__ __align__(16) float lds[kT * 8];         // two image buffers [3][IWp]; the reductions afterwards
    constexpr int BS = 3 * IWp + 2;                     // buffer stride; [3 * IWp] is a slot for the staging threads past the rows
    static_assert(2 * BS <= kT * 8, "image rows do not fit");
    const int Wp = W + pw, Hp = H + ph, IH = 2 * H + 1;
    const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
    f32x2 wk[9][2], acc[11][2], mu[2], rs[2], ga[2], be[2];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) wk[k][i] = f32x2{wgt[k * C + cg * CPT + 2 * i], wgt[k * C + cg * CPT + 2 * i + 1]};
#pragma unroll
    for (int k = 0; k < 11; ++k) acc[k][0] = acc[k][1] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = cg * CPT + 2 * i;
        mu[i] = f32x2{mean[c], mean[c + 1]}; rs[i] = f32x2{rstd[c], rstd[c + 1]};
        ga[i] = f32x2{gamma[c], gamma[c + 1]}; be[i] = f32x2{beta[c], beta[c + 1]};
    }
    const f32x2 sl2 = f32x2{slope, slope};
    const int64_t rows = N * H;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 dq[NB], dqn[NB];
    uint16_t img[NS];
    auto load_row = [&](int64_t row, u32x2 (&d)[NB]) __attribute__((always_inline)) {
        const int64_t n = row / H;
        const int h = (int)(row - n * H);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i0 = tid + s * kT, i = i0 < 3 * IW ? i0 : 3 * IW - 1, ky = i / IW, xx = i - ky * IW;      // unconditional (clamped):
            img[s] = image[(n * IH + 2 * h + ky) * (int64_t)IW + xx];                          // a branch here drains the queue
        }
        const uint16_t* dyr = dy + ((n * Hp + h) * (int64_t)Wp) * C + cg * CPT;
#pragma unroll
        for (int u = 0; u < NB; ++u) d[u] = *reinterpret_cast<const u32x2*>(dyr + (int64_t)(pl + u * PL) * C);
    };
    auto stage = [&](float* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + s * kT, ky = i / IW, xx = i - ky * IW;
            buf[i < 3 * IW ? ky * IWp + xx : 3 * IWp] = h2f<BF>(img[s]);        // unconditional: hipcc sinks the LOAD into a branch here
        }
    };
    int64_t row = blockIdx.x;
    int cur = 0;
    if (row < rows) {
        load_row(row, dq);
        stage(lds);
        if (tid < 3) { lds[tid * IWp + IW] = 0.0f; lds[BS + tid * IWp + IW] = 0.0f; }     // the unused half of the last pair
    }
    __syncthreads();
    for (; row < rows; row += gridDim.x) {
        const int64_t nxt = row + gridDim.x < rows ? row + gridDim.x : row;     // past the end: this row again (never used)
        load_row(nxt, dqn);
        __builtin_amdgcn_sched_barrier(0);              // the loads stay here (left alone hipcc sinks them below the sums: a full latency per row)
        const float* buf = lds + cur * BS + 2 * pl;
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            f32x2 P[3][2];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                P[ky][0] = *reinterpret_cast<const f32x2*>(buf + ky * IWp + 2 * u * PL);
                P[ky][1] = *reinterpret_cast<const f32x2*>(buf + ky * IWp + 2 * u * PL + 2);
            }
            f32x2 d[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t q = dq[u][i];
                d[i] = f32x2{h2f<BF>((uint16_t)(q & 0xFFFFu)), h2f<BF>((uint16_t)(q >> 16))};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                // the convolution output again, in the tap order of the forward pass (k = 0 first: fma(w, t, 0) is the product)
                f32x2 v = pk_mul_lo(wk[0][i], P[0][0]);
                pk_fma_hi(v, wk[1][i], P[0][0]); pk_fma_lo(v, wk[2][i], P[0][1]);
                pk_fma_lo(v, wk[3][i], P[1][0]); pk_fma_hi(v, wk[4][i], P[1][0]); pk_fma_lo(v, wk[5][i], P[1][1]);
                pk_fma_lo(v, wk[6][i], P[2][0]); pk_fma_hi(v, wk[7][i], P[2][0]); pk_fma_lo(v, wk[8][i], P[2][1]);
                const f32x2 xh = (v - mu[i]) * rs[i];
                const f32x2 zz = xh * ga[i] + be[i];
                const f32x2 ds = d[i] * sl2;
                const f32x2 g = f32x2{zz.x > 0.0f ? d[i].x : ds.x, zz.y > 0.0f ? d[i].y : ds.y};
                acc[9][i] += g;
                acc[10][i] = __builtin_elementwise_fma(g, xh, acc[10][i]);
                pk_fma_lo(acc[0][i], g, P[0][0]); pk_fma_hi(acc[1][i], g, P[0][0]); pk_fma_lo(acc[2][i], g, P[0][1]);
                pk_fma_lo(acc[3][i], g, P[1][0]); pk_fma_hi(acc[4][i], g, P[1][0]); pk_fma_lo(acc[5][i], g, P[1][1]);
                pk_fma_lo(acc[6][i], g, P[2][0]); pk_fma_hi(acc[7][i], g, P[2][0]); pk_fma_lo(acc[8][i], g, P[2][1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        stage(lds + (cur ^ 1) * BS);               // the next row's image rows: their loads went out before this row's sums
#pragma unroll
        for (int u = 0; u < NB; ++u) dq[u] = dqn[u];
        cur ^= 1;
        __syncthreads();
    }
    for (int k = 0; k < 11; ++k) {
        __syncthreads();
        lds[tid * CPT + 0] = acc[k][0].x; lds[tid * CPT + 1] = acc[k][0].y; lds[tid * CPT + 2] = acc[k][1].x; lds[tid * CPT + 3] = acc[k][1].y;
        __syncthreads();
        for (int c = tid; c < C; c += kT) {
            float s = 0.0f;
            for (int q = 0; q < PL; ++q) s += lds[(q * CG + (c / CPT)) * CPT + (c % CPT)];
            part[((size_t)blockIdx.x * 11 + k) * C + c] = s;
        }
    }
}

// sums [11][C] (A, sum g, sum g*x_hat) and img [54] (B, packed upper triangle of T2) -> dW[9][C], dgamma, dbeta
__global__ void k_c1_wgrad_combine(const float* sums, const float* img, const float* wgt, const float* mean, const float* rstd,
                                   const float* gamma, int C, double M, float* dweight, float* dgamma, float* dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double sg = sums[9 * C + c], sgx = sums[10 * C + c];
    dbeta[c] = (float)sg;
    dgamma[c] = (float)sgx;
    const double c1 = sg / M, c2 = sgx / M, rs = rstd[c], mu = mean[c], ga = gamma[c];
    for (int k = 0; k < 9; ++k) {
        double wt2 = 0.0;
        for (int j = 0; j < 9; ++j) {
            const int a = j < k ? j : k, b = j < k ? k : j;                 // packed index of (a <= b)
            const int q = a * 9 - a * (a - 1) / 2 + (b - a);
            wt2 += (double)wgt[j * C + c] * (double)img[9 + q];
        }
        const double Bk = img[k];
        const double D = rs * (wt2 - mu * Bk);
        dweight[k * C + c] = (float)(ga * rs * ((double)sums[k * C + c] - c1 * Bk - c2 * D));
    }
}

// batch statistics of the first layer from the image statistics alone: mean_c = sum_j w[j][c]*B[j] / M,
// E[z^2]_c = sum_jk w[j][c]*w[k][c]*T2[j][k] / M -- no pass over the (virtual) convolution output at all
__global__ void k_c1_stats_from_img(const float* img, const float* wgt, int C, double M, float eps, float momentum,
                                    float* mean_o, float* rstd_o, float* running_mean, float* running_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double m1 = 0.0, m2 = 0.0;
    for (int j = 0; j < 9; ++j) {
        const double wj = wgt[j * C + c];
        m1 += wj * (double)img[j];
        for (int k = 0; k < 9; ++k) {
            const int a = j < k ? j : k, b = j < k ? k : j;
            m2 += wj * (double)wgt[k * C + c] * (double)img[9 + a * 9 - a * (a - 1) / 2 + (b - a)];
        }
    }
    const double mean = m1 / M;
    double var = m2 / M - mean * mean;
    var = var > 0.0 ? var : 0.0;
    mean_o[c] = (float)mean;
    rstd_o[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
    if (running_var) running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * var * (M > 1.0 ? M / (M - 1.0) : 1.0));
}

// out[j] = sum over g of part[g][j] (float64, strided partial sums + fixed tree): one workgroup per j
__global__ __launch_bounds__(kT) void k_sum_partials(const float* part, int G, int L, float* out) {
    __shared__ double sa[kT];
    const int j = blockIdx.x, t = threadIdx.x;
    double a = 0.0;
    for (int g = t; g < G; g += kT) a += (double)part[(size_t)g * L + j];
    sa[t] = a;
    __syncthreads();
    for (int off = kT / 2; off >= 1; off >>= 1) {
        if (t < off) sa[t] += sa[t + off];
        __syncthreads();
    }
    if (t == 0) out[j] = (float)sa[0];
}

int stats_grid(rml_ctx* ctx, int64_t M, int C) {
    const int PL = kT / (C >> 3);
    int64_t g = (M + PL - 1) / PL;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

int check_common(const char* who, rml_ctx* ctx, int64_t N, int H, int W, int C, int ph, int pw, int dtype) {
    RML_REQUIRE(ctx && N >= 0 && H > 0 && W > 0 && C > 0, RML_ERR_INVALID, "%s: bad arguments", who);
    RML_REQUIRE(C % 8 == 0 && C <= 2048 && kT % (C >> 3) == 0, RML_ERR_UNSUPPORTED, "%s: C = %d (multiples of 8 that divide 2048)", who, C);
    RML_REQUIRE((ph == 0 || ph == 1) && (pw == 0 || pw == 1), RML_ERR_INVALID, "%s: pad must be 0 or 1", who);
    RML_REQUIRE(dtype == 0 || dtype == 1, RML_ERR_INVALID, "%s: dtype 0 = float16, 1 = bfloat16", who);
    RML_REQUIRE(N * (int64_t)(H + ph) < ((int64_t)1 << 31), RML_ERR_UNSUPPORTED, "%s: too many rows for one launch", who);
    return RML_OK;
}

}  // namespace

extern "C" int64_t rml_bn_workspace_floats(rml_ctx* ctx, int C) {
    // sized for the first-layer backward: [G][11][C] partials + [11][C] sums + image statistics (256*54 + 54)
    return ctx ? (int64_t)ctx->num_cu * 8 * 11 * C + (int64_t)11 * C + 256 * 54 + 54 : 0;
}

extern "C" int rml_bn_lrelu_pad_forward(rml_ctx* ctx, const void* x, int dtype, int64_t N, int H, int W, int C, int pad_h, int pad_w,
                                        const float* gamma, const float* beta, float eps, float momentum, float slope,
                                        float* running_mean, float* running_var, float* save_mean, float* save_rstd,
                                        float* workspace, void* y, void* stream) {
    int rc = check_common("rml_bn_lrelu_pad_forward", ctx, N, H, W, C, pad_h, pad_w, dtype);
    if (rc) return rc;
    if (N == 0) return RML_OK;
    RML_REQUIRE(x && y && gamma && beta && save_mean && save_rstd && workspace, RML_ERR_INVALID, "rml_bn_lrelu_pad_forward: NULL argument");
    RML_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, RML_ERR_INVALID,
                "rml_bn_lrelu_pad_forward: x and y must be 16-byte aligned");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t M = N * H * W;
    const int G = stats_grid(ctx, M, C);
    const uint16_t* xs = static_cast<const uint16_t*>(x);
    uint16_t* ys = static_cast<uint16_t*>(y);
    if (dtype) hipLaunchKernelGGL(k_bn_stats<true>, dim3(G), dim3(kT), 0, st, xs, M, C, workspace);
    else hipLaunchKernelGGL(k_bn_stats<false>, dim3(G), dim3(kT), 0, st, xs, M, C, workspace);
    hipLaunchKernelGGL(k_bn_finalize, dim3(C), dim3(kT), 0, st, workspace, G, C, (double)M, 0, eps, momentum,
                       save_mean, save_rstd, running_mean, running_var);
    const unsigned rows = (unsigned)(N * (H + pad_h));
    if (dtype) hipLaunchKernelGGL(k_bn_apply_pad<true>, dim3(rows), dim3(kT), 0, st, xs, ys, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope);
    else hipLaunchKernelGGL(k_bn_apply_pad<false>, dim3(rows), dim3(kT), 0, st, xs, ys, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

extern "C" int rml_bn_lrelu_pad_backward(rml_ctx* ctx, const void* x, const void* dy, int dtype, int64_t N, int H, int W, int C,
                                         int pad_h, int pad_w, const float* gamma, const float* beta, const float* save_mean,
                                         const float* save_rstd, float slope, float* workspace, void* dx, float* dgamma, float* dbeta,
                                         void* stream) {
    int rc = check_common("rml_bn_lrelu_pad_backward", ctx, N, H, W, C, pad_h, pad_w, dtype);
    if (rc) return rc;
    if (N == 0) return RML_OK;
    RML_REQUIRE(x && dy && dx && gamma && beta && save_mean && save_rstd && workspace && dgamma && dbeta, RML_ERR_INVALID,
                "rml_bn_lrelu_pad_backward: NULL argument");
    RML_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0, RML_ERR_INVALID,
                "rml_bn_lrelu_pad_backward: x, dy and dx must be 16-byte aligned");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t M = N * H * W;
    const int G = stats_grid(ctx, M, C);
    const uint16_t* xs = static_cast<const uint16_t*>(x);
    const uint16_t* ds = static_cast<const uint16_t*>(dy);
    uint16_t* os = static_cast<uint16_t*>(dx);
    float* c1 = workspace + (size_t)G * 2 * C;          // the two per-channel means live behind the partials
    float* c2 = c1 + C;
    if (dtype) hipLaunchKernelGGL(k_bn_bwd_reduce<true>, dim3(G), dim3(kT), 0, st, xs, ds, N, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope, workspace);
    else hipLaunchKernelGGL(k_bn_bwd_reduce<false>, dim3(G), dim3(kT), 0, st, xs, ds, N, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope, workspace);
    hipLaunchKernelGGL(k_bn_finalize, dim3(C), dim3(kT), 0, st, workspace, G, C, (double)M, 1, 0.0f, 0.0f, dbeta, dgamma, c1, c2);
    const unsigned rows = (unsigned)(N * H);
    if (dtype) hipLaunchKernelGGL(k_bn_bwd_apply<true>, dim3(rows), dim3(kT), 0, st, xs, ds, os, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, c1, c2, slope);
    else hipLaunchKernelGGL(k_bn_bwd_apply<false>, dim3(rows), dim3(kT), 0, st, xs, ds, os, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, c1, c2, slope);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

// ---- first layer, convolution folded in: image (N x (2H+1) x (2W+1), half) + weights [9][C] float32 -> y ------------
extern "C" int rml_conv1_bn_lrelu_pad_forward(rml_ctx* ctx, const void* image, const float* weight, int dtype, int64_t N, int H, int W,
                                              int C, int pad_h, int pad_w, const float* gamma, const float* beta, float eps,
                                              float momentum, float slope, float* running_mean, float* running_var, float* save_mean,
                                              float* save_rstd, float* img_stats, float* workspace, void* y, void* stream) {
    int rc = check_common("rml_conv1_bn_lrelu_pad_forward", ctx, N, H, W, C, pad_h, pad_w, dtype);
    if (rc) return rc;
    if (N == 0) return RML_OK;
    RML_REQUIRE(image && weight && y && gamma && beta && save_mean && save_rstd && img_stats && workspace, RML_ERR_INVALID,
                "rml_conv1_bn_lrelu_pad_forward: NULL argument");
    RML_REQUIRE((reinterpret_cast<uintptr_t>(y) & 15) == 0, RML_ERR_INVALID, "rml_conv1_bn_lrelu_pad_forward: y must be 16-byte aligned");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t M = N * H * W;
    // image-statistics workgroups: eight per CU (with 256 -- one per CU, 16 pixels per thread, each waiting for its own nine loads --
    // the pass took 22 us for an 8.5 MB image); the partials fit the workspace for every C >= 8 (num_cu * 8 * 54 <= num_cu * 8 * 11 * C)
    const int GI = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)ctx->num_cu * 8, (M + kT - 1) / kT));
    const uint16_t* is = static_cast<const uint16_t*>(image);
    uint16_t* ys = static_cast<uint16_t*>(y);
    // statistics of the image (B[9], T2 packed [45]) -> batch statistics of the convolution output, no pass over it
    if (dtype) hipLaunchKernelGGL(k_c1_imgstats<true>, dim3(GI), dim3(kT), 0, st, is, N, H, W, workspace);
    else hipLaunchKernelGGL(k_c1_imgstats<false>, dim3(GI), dim3(kT), 0, st, is, N, H, W, workspace);
    hipLaunchKernelGGL(k_sum_partials, dim3(54), dim3(kT), 0, st, workspace, GI, 54, img_stats);
    hipLaunchKernelGGL(k_c1_stats_from_img, dim3((C + 63) / 64), dim3(64), 0, st, img_stats, weight, C, (double)M, eps, momentum,
                       save_mean, save_rstd, running_mean, running_var);
    const unsigned rows = (unsigned)(N * (H + pad_h));
    const unsigned ga = rows < (unsigned)(ctx->num_cu * 8) ? rows : (unsigned)(ctx->num_cu * 8);
    if (ctx->opt.c1_pk && C == 128 && (W == 16 || W == 32 || W == 64)) {
        auto go = [&](auto bf, auto nb) {
            constexpr bool BFV = decltype(bf)::value;
            constexpr int NBV = decltype(nb)::value;
            static const int per_cu = [] { int n = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_c1_apply_pad_pk<BFV, NBV>, kT, 0) == hipSuccess && n > 0 ? n : 4; }();
            const unsigned g = std::min<unsigned>(rows, (unsigned)(per_cu * ctx->num_cu));
            hipLaunchKernelGGL((k_c1_apply_pad_pk<BFV, NBV>), dim3(g), dim3(kT), 0, st, is, weight, ys, (int64_t)rows, H, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope);
        };
        using T = std::true_type; using F = std::false_type;
        if (dtype) { if (W == 64) go(T{}, std::integral_constant<int, 8>{}); else if (W == 32) go(T{}, std::integral_constant<int, 4>{}); else go(T{}, std::integral_constant<int, 2>{}); }
        else { if (W == 64) go(F{}, std::integral_constant<int, 8>{}); else if (W == 32) go(F{}, std::integral_constant<int, 4>{}); else go(F{}, std::integral_constant<int, 2>{}); }
    } else if (dtype) hipLaunchKernelGGL(k_c1_apply_pad<true>, dim3(ga), dim3(kT), 0, st, is, weight, ys, (int64_t)rows, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope);
    else hipLaunchKernelGGL(k_c1_apply_pad<false>, dim3(ga), dim3(kT), 0, st, is, weight, ys, (int64_t)rows, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

extern "C" int rml_conv1_bn_lrelu_pad_backward(rml_ctx* ctx, const void* image, const float* weight, const void* dy, int dtype, int64_t N,
                                               int H, int W, int C, int pad_h, int pad_w, const float* gamma, const float* beta,
                                               const float* save_mean, const float* save_rstd, const float* img_stats, float slope,
                                               float* workspace, float* dweight, float* dgamma, float* dbeta, void* stream) {
    int rc = check_common("rml_conv1_bn_lrelu_pad_backward", ctx, N, H, W, C, pad_h, pad_w, dtype);
    if (rc) return rc;
    if (N == 0) return RML_OK;
    RML_REQUIRE(image && weight && dy && gamma && beta && save_mean && save_rstd && img_stats && workspace && dweight && dgamma && dbeta,
                RML_ERR_INVALID, "rml_conv1_bn_lrelu_pad_backward: NULL argument");
    RML_REQUIRE((reinterpret_cast<uintptr_t>(dy) & 15) == 0, RML_ERR_INVALID, "rml_conv1_bn_lrelu_pad_backward: dy must be 16-byte aligned");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t M = N * H * W;
    int G = stats_grid(ctx, M, C);
    const uint16_t* is = static_cast<const uint16_t*>(image);
    const uint16_t* ds = static_cast<const uint16_t*>(dy);
    // workspace: [G][11][C] partials | sums [11][C]
    float* part = workspace;
    float* sums = part + (size_t)G * 11 * C;
    const float* img = img_stats;
    // four channels per thread where the channel groups divide the workgroup (C = 128: 32 groups x 8 pixels)
    const bool four = C % 4 == 0 && kT % (C / 4) == 0 && C / 4 <= kT;
    // C = 128 and rows of 16 / 32 / 64 pixels: the packed kernel (RML_OPT_C1_PK = 0: the general one)
    const bool pk_on = ctx->opt.c1_pk != 0;
    const bool pk = pk_on && C == 128 && (W == 16 || W == 32 || W == 64);
    if (pk) {
        // one round of resident workgroups (166 registers at W = 64: three per CU), rows strided over them
        auto go = [&](auto bf, auto nb) {
            constexpr bool BFV = decltype(bf)::value;
            constexpr int NBV = decltype(nb)::value;
            static const int per_cu = [] { int n = 0; return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_c1_bwd1_pk<BFV, NBV>, kT, 0) == hipSuccess && n > 0 ? n : 2; }();
            G = std::min(G, per_cu * ctx->num_cu);
            hipLaunchKernelGGL((k_c1_bwd1_pk<BFV, NBV>), dim3(G), dim3(kT), 0, st, is, weight, ds, N, H, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope, part);
        };
        using T = std::true_type; using F = std::false_type;
        if (dtype) { if (W == 64) go(T{}, std::integral_constant<int, 8>{}); else if (W == 32) go(T{}, std::integral_constant<int, 4>{}); else go(T{}, std::integral_constant<int, 2>{}); }
        else { if (W == 64) go(F{}, std::integral_constant<int, 8>{}); else if (W == 32) go(F{}, std::integral_constant<int, 4>{}); else go(F{}, std::integral_constant<int, 2>{}); }
    } else if (four) {
        if (dtype) hipLaunchKernelGGL((k_c1_bwd1<true, 4>), dim3(G), dim3(kT), 0, st, is, weight, ds, N, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope, part);
        else hipLaunchKernelGGL((k_c1_bwd1<false, 4>), dim3(G), dim3(kT), 0, st, is, weight, ds, N, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope, part);
    } else {
        if (dtype) hipLaunchKernelGGL((k_c1_bwd1<true, 8>), dim3(G), dim3(kT), 0, st, is, weight, ds, N, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope, part);
        else hipLaunchKernelGGL((k_c1_bwd1<false, 8>), dim3(G), dim3(kT), 0, st, is, weight, ds, N, H, W, C, pad_h, pad_w, save_mean, save_rstd, gamma, beta, slope, part);
    }
    hipLaunchKernelGGL(k_sum_partials, dim3(11 * C), dim3(kT), 0, st, part, G, 11 * C, sums);
    hipLaunchKernelGGL(k_c1_wgrad_combine, dim3((C + 63) / 64), dim3(64), 0, st, sums, img, weight, save_mean, save_rstd, gamma, C, (double)M,
                       dweight, dgamma, dbeta);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
