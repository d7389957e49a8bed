// Pillow's bicubic resize of float32 ('F' mode) images for gfx950 -- the step between the projections and the
// dnn / sgan classifiers: Image.fromarray(p).resize(RESCALE, resample=Image.BICUBIC) (dnn.py:240-245,
// sgan.py:676-681) after the [-1,1] scaling (dnn.py:202-205).
//
// The algorithm is Pillow's src/libImaging/Resample.c (Pillow is a pinned dependency of the reference,
// requirements.txt:41; not vendored): per axis a table of windows and normalised double weights
// (precompute_coeffs: antialiased, the support grows with the downscale factor), a horizontal pass into a float32
// intermediate and a vertical pass; every output is a double-precision sum in tap order rounded to float32.  The
// weights are computed on the host exactly as Pillow does; the passes run with fp contraction off (separate
// multiply and add, no fma), so the result is bit-identical to Pillow's (tests/golden/pil_resize.npz).
//
// Persistent workgroups: the weight tables are staged in LDS once, then the workgroup walks its planes; the
// (scaled) plane and the intermediate live in LDS, the next plane is prefetched into registers meanwhile, input and
// output are touched once.  In the horizontal pass a thread owns an output column and keeps its weights in registers.  Bytes per plane: 4*H*W in, 4 (or 2, bf16 for the conv trunk)*OH*OW out.
#include "rml_internal.h"
#include "resize_tables.h"

namespace {

using rmlresize::AxisTable;
using rmlresize::precompute;

struct ResizeArgs {
    const float* in;
    int64_t in_stride, B;
    int H, W, OH, OW;
    float sub, div;
    const int* bh; const double* kh; int ksh;       // horizontal tables (nullptr: width unchanged)
    const int* bv; const double* kv; int ksv;       // vertical tables (nullptr: height unchanged)
    void* out;
    int out_bf16;
};

__device__ __forceinline__ uint16_t f2bf_rne(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

constexpr int KPAD = 16;        // zeroed floats behind the plane: a register-resident window may overrun its row
constexpr int PF = 24;          // floats of the next plane a thread prefetches (planes up to 6144 pixels)

// LDS-only barrier: __syncthreads() would also wait for the prefetch of the next plane (vmcnt)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Horizontal pass with the window in registers: a thread owns output column xx (its KS weights never change) and
// walks the rows; taps past the window have weight 0 and read finite values, which leaves the double sum bit-identical.
template <int KS>
__device__ __forceinline__ void horizontal_fixed(const float* in_s, int H, int W, int OW, int first, const double (&k)[KS],
                                                 int g, int G, int xx, float* tmp_s) {
#pragma clang fp contract(off)
    if (g >= G) return;
    // two rows at a time: two independent float64 chains per thread (a single chain of KS dependent adds left the VALU idle most
    // of the time: three waves per SIMD cannot cover an 8-cycle add latency per tap); each sum keeps its own tap order
    int y = g;
    for (; y + G < H; y += 2 * G) {
        const float* s = in_s + y * W + first;
        const float* s2 = s + G * W;
        double ss = 0.0, ss2 = 0.0;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            ss = ss + (double)s[t] * k[t];
            ss2 = ss2 + (double)s2[t] * k[t];
        }
        tmp_s[y * OW + xx] = (float)ss;                 // Pillow's float32 intermediate image
        tmp_s[(y + G) * OW + xx] = (float)ss2;
    }
    if (y < H) {
        const float* s = in_s + y * W + first;
        double ss = 0.0;
#pragma unroll
        for (int t = 0; t < KS; ++t) ss = ss + (double)s[t] * k[t];
        tmp_s[y * OW + xx] = (float)ss;
    }
}

// KS = 0: weights from LDS (any window size)
template <int KS>
__global__ __launch_bounds__(256) void k_resize(ResizeArgs a) {
#pragma clang fp contract(off)      // HIP contracts a*b+c into an fma by default; Pillow multiplies, then adds
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = a.H, W = a.W, OH = a.OH, OW = a.OW;
    const int tid = threadIdx.x;
    const bool horiz = a.kh != nullptr, vert = a.kv != nullptr;
    // LDS: weight tables (doubles first: 8-byte alignment), windows, the plane, the intermediate
    double* kh_s = reinterpret_cast<double*>(smem);
    double* kv_s = kh_s + (horiz ? OW * a.ksh : 0);
    int* bh_s = reinterpret_cast<int*>(kv_s + (vert ? OH * a.ksv : 0));
    int* bv_s = bh_s + (horiz ? 2 * OW : 0);
    float* in_s = reinterpret_cast<float*>(bv_s + (vert ? 2 * OH : 0));      // [H*W + KPAD]
    float* tmp_s = in_s + H * W + KPAD;                                      // [H][OW] (only when both passes run)

    // persistent workgroup: the tables are staged once, planes blockIdx.x, +gridDim.x, ... follow, and while one
    // plane is resampled the next one is already in flight into registers
    if (horiz) {
        for (int i = tid; i < OW * a.ksh; i += 256) kh_s[i] = a.kh[i];
        for (int i = tid; i < 2 * OW; i += 256) bh_s[i] = a.bh[i];
    }
    if (vert) {
        for (int i = tid; i < OH * a.ksv; i += 256) kv_s[i] = a.kv[i];
        for (int i = tid; i < 2 * OH; i += 256) bv_s[i] = a.bv[i];
    }
    if (tid < KPAD) in_s[H * W + tid] = 0.0f;
    __syncthreads();
    // the thread's column of the horizontal pass
    const int G = OW <= 256 ? 256 / OW : 0;
    const int hg = G ? tid / OW : 0, hx = tid - hg * OW;
    constexpr int KR = KS > 0 ? KS : 1;
    double kreg[KR];
    int hfirst = 0;
    if (KS > 0 && horiz && hg < G) {
        hfirst = bh_s[2 * hx];
#pragma unroll
        for (int t = 0; t < KR; ++t) kreg[t] = t < a.ksh ? kh_s[hx * a.ksh + t] : 0.0;
    }
    const int npix = H * W;
    const bool prefetch = npix <= 256 * PF;
    const bool scaled = a.div != 0.0f;
    float pv[PF];
    auto issue = [&](int64_t bb) {
        const float* __restrict__ src = a.in + bb * a.in_stride;
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int i = u * 256 + tid;
            pv[u] = src[i < npix ? i : npix - 1];
        }
    };
    if (prefetch) issue(blockIdx.x);

    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        if (prefetch) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int i = u * 256 + tid;
                if (i < npix) in_s[i] = scaled ? __fdiv_rn(pv[u] - a.sub, a.div) : pv[u];
            }
            const int64_t nb = b + gridDim.x;
            issue(nb < a.B ? nb : b);
        } else {
            const float* __restrict__ src = a.in + b * a.in_stride;
            for (int i0 = 0; i0 < npix; i0 += 256 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * 256 + tid;
                    v[u] = src[i < npix ? i : npix - 1];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * 256 + tid;
                    if (i < npix) in_s[i] = scaled ? __fdiv_rn(v[u] - a.sub, a.div) : v[u];
                }
            }
        }
        lds_barrier();

        float* outf = reinterpret_cast<float*>(a.out) + b * (int64_t)OH * OW;
        uint16_t* outh = reinterpret_cast<uint16_t*>(a.out) + b * (int64_t)OH * OW;
        auto emit = [&](int i, float v) {
            if (a.out_bf16) outh[i] = f2bf_rne(v);
            else outf[i] = v;
        };
        const float* cur = in_s;
        if (horiz) {
            if (KS > 0 && vert) {
                horizontal_fixed<KR>(in_s, H, W, OW, hfirst, kreg, hg, G, hx, tmp_s);
            } else {
                const int dq = 256 / OW, dr = 256 - dq * OW;
                int y = tid / OW, xx = tid - y * OW;
                for (int i = tid; i < H * OW; i += 256) {
                    const int first = bh_s[2 * xx], n = bh_s[2 * xx + 1];
                    const float* s = in_s + y * W + first;
                    const double* k = kh_s + xx * a.ksh;
                    double ss = 0.0;
                    for (int t = 0; t < n; ++t) ss = ss + (double)s[t] * k[t];
                    if (vert) tmp_s[i] = (float)ss;
                    else emit(i, (float)ss);            // width only: straight to the output
                    y += dq; xx += dr;
                    if (xx >= OW) { xx -= OW; ++y; }
                }
            }
            if (vert) lds_barrier();
            cur = tmp_s;
        }
        if (!vert) {
            if (!horiz)             // no pass at all: Image.resize returns a copy
                for (int i = tid; i < OH * OW; i += 256) emit(i, cur[i]);
        } else if ((OW & 3) == 0) {
            // vertical pass: a thread produces the four outputs (yy, xx + q * OW/4) from one read of the row's weights -- four
            // independent float64 chains (two left the VALU waiting on the add latency), each in its own tap order
            const int HW4 = OW >> 2;
            const int dq = 256 / HW4, dr = 256 - dq * HW4;
            int yy = tid / HW4, xx = tid - yy * HW4;
            for (int i = tid; i < OH * HW4; i += 256) {
                const int first = bv_s[2 * yy], n = bv_s[2 * yy + 1];
                const float* s = cur + first * OW + xx;
                const double* k = kv_s + yy * a.ksv;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                for (int t = 0; t < n; ++t) {
                    const double kt = k[t];
                    const float* r = s + t * OW;
                    s0 = s0 + (double)r[0] * kt;
                    s1 = s1 + (double)r[HW4] * kt;
                    s2 = s2 + (double)r[2 * HW4] * kt;
                    s3 = s3 + (double)r[3 * HW4] * kt;
                }
                emit(yy * OW + xx, (float)s0);
                emit(yy * OW + xx + HW4, (float)s1);
                emit(yy * OW + xx + 2 * HW4, (float)s2);
                emit(yy * OW + xx + 3 * HW4, (float)s3);
                yy += dq; xx += dr;
                while (xx >= HW4) { xx -= HW4; ++yy; }
            }
        } else if ((OW & 1) == 0) {
            // vertical pass: a thread produces (yy, xx) and (yy, xx + OW/2) from one read of the row's weights
            const int HW2 = OW >> 1;
            const int dq = 256 / HW2, dr = 256 - dq * HW2;
            int yy = tid / HW2, xx = tid - yy * HW2;
            for (int i = tid; i < OH * HW2; i += 256) {
                const int first = bv_s[2 * yy], n = bv_s[2 * yy + 1];
                const float* s = cur + first * OW + xx;
                const double* k = kv_s + yy * a.ksv;
                double s0 = 0.0, s1 = 0.0;
                for (int t = 0; t < n; ++t) {
                    const double kt = k[t];
                    s0 = s0 + (double)s[t * OW] * kt;
                    s1 = s1 + (double)s[t * OW + HW2] * kt;
                }
                emit(yy * OW + xx, (float)s0);
                emit(yy * OW + xx + HW2, (float)s1);
                yy += dq; xx += dr;
                if (xx >= HW2) { xx -= HW2; ++yy; }
            }
        } else {
            const int dq = 256 / OW, dr = 256 - dq * OW;
            int yy = tid / OW, xx = tid - yy * OW;
            for (int i = tid; i < OH * OW; i += 256) {
                const int first = bv_s[2 * yy], n = bv_s[2 * yy + 1];
                const float* s = cur + first * OW + xx;
                const double* k = kv_s + yy * a.ksv;
                double ss = 0.0;
                for (int t = 0; t < n; ++t) ss = ss + (double)s[t * OW] * k[t];
                emit(i, (float)ss);
                yy += dq; xx += dr;
                if (xx >= OW) { xx -= OW; ++yy; }
            }
        }
        lds_barrier();          // the plane and the intermediate are free for the next sample
    }
}

template <int KS>
void launch_resize(const ResizeArgs& a, size_t lds, int num_cu, hipStream_t st) {
    RML_MAX_DYN_LDS(160 * 1024, &k_resize<KS>);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_resize<KS>, 256, lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const int64_t slots = (int64_t)per_cu * num_cu;
    hipLaunchKernelGGL(k_resize<KS>, dim3((unsigned)(a.B < slots ? a.B : slots)), dim3(256), lds, st, a);
}

// device copy of an axis table, cached in the context (a handful of (in, out) pairs per process)
int axis_table(rml_ctx* ctx, int in_size, int out_size, const int** bounds, const double** kk, int* ksize) {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);    // the table cache is shared by every thread using the context
    for (const auto& e : ctx->resize_tabs)
        if (e.in == in_size && e.out == out_size) { *bounds = e.bounds; *kk = e.kk; *ksize = e.ksize; return RML_OK; }
    AxisTable t;
    precompute(in_size, out_size, t);
    rml_resize_tab e{};
    e.in = in_size; e.out = out_size; e.ksize = t.ksize;
    void* p = nullptr;
    const size_t kb = t.kk.size() * sizeof(double), bb = t.bounds.size() * sizeof(int);
    RML_HIP(hipMalloc(&p, kb + bb));
    hipError_t err = hipMemcpy(p, t.kk.data(), kb, hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(static_cast<char*>(p) + kb, t.bounds.data(), bb, hipMemcpyHostToDevice);
    if (err != hipSuccess) { (void)hipFree(p); RML_HIP(err); }
    e.kk = static_cast<const double*>(p);
    e.bounds = reinterpret_cast<const int*>(static_cast<char*>(p) + kb);
    ctx->resize_tabs.push_back(e);
    *bounds = e.bounds; *kk = e.kk; *ksize = e.ksize;
    return RML_OK;
}

}  // namespace

extern "C" int rml_resize_bicubic(rml_ctx* ctx, const float* in, int64_t in_stride, int64_t B, int H, int W, int out_h,
                                  int out_w, float sub, float div, void* out, int out_bf16, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && H > 0 && W > 0 && out_h > 0 && out_w > 0, RML_ERR_INVALID, "rml_resize_bicubic: bad arguments");
    if (B == 0) return RML_OK;
    RML_REQUIRE(in && out, RML_ERR_INVALID, "rml_resize_bicubic: NULL argument");
    RML_REQUIRE(in_stride >= (int64_t)H * W, RML_ERR_INVALID, "rml_resize_bicubic: in_stride %lld < H*W", (long long)in_stride);
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_resize_bicubic: B too large");
    RML_HIP(hipSetDevice(ctx->device));
    ResizeArgs a{};
    a.in = in; a.in_stride = in_stride; a.B = B; a.H = H; a.W = W; a.OH = out_h; a.OW = out_w;
    a.sub = sub; a.div = div; a.out = out; a.out_bf16 = out_bf16 ? 1 : 0;
    size_t lds = ((size_t)H * W + KPAD) * 4;
    if (out_w != W) {
        int rc = axis_table(ctx, W, out_w, &a.bh, &a.kh, &a.ksh);
        if (rc != RML_OK) return rc;
        lds += (size_t)out_w * a.ksh * 8 + (size_t)out_w * 8;
    }
    if (out_h != H) {
        int rc = axis_table(ctx, H, out_h, &a.bv, &a.kv, &a.ksv);
        if (rc != RML_OK) return rc;
        lds += (size_t)out_h * a.ksv * 8 + (size_t)out_h * 8;
    }
    if (out_w != W && out_h != H) lds += (size_t)H * out_w * 4;
    RML_REQUIRE(lds <= 150 * 1024, RML_ERR_UNSUPPORTED, "rml_resize_bicubic: %dx%d -> %dx%d does not fit the LDS-resident resize", H, W, out_h, out_w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fixed = out_w != W && out_h != H && out_w <= 256;
    if (fixed && a.ksh <= 6) launch_resize<6>(a, lds, ctx->num_cu, st);
    else if (fixed && a.ksh <= 12) launch_resize<12>(a, lds, ctx->num_cu, st);
    else if (fixed && a.ksh <= KPAD) launch_resize<KPAD>(a, lds, ctx->num_cu, st);
    else launch_resize<0>(a, lds, ctx->num_cu, st);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
