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
// One workgroup per plane: the (scaled) plane, the intermediate and both weight tables live in LDS; input and output
// are touched once.  Bytes per plane: 4*H*W in, 4 (or 2, bf16 for the conv trunk)*OH*OW out.
#include "rml_internal.h"
#include <math.h>
#include <vector>

namespace {

struct AxisTable {          // host copy of precompute_coeffs() for one (in, out) pair
    int in = 0, out = 0, ksize = 0;
    std::vector<int> bounds;        // [out][2]: first input index, tap count
    std::vector<double> kk;         // [out][ksize]
};

double bicubic_filter(double x) {
#pragma clang fp contract(off)
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

void precompute(int in_size, int out_size, AxisTable& t) {
#pragma clang fp contract(off)
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    t.in = in_size; t.out = out_size;
    t.ksize = (int)ceil(support) * 2 + 1;
    t.bounds.assign((size_t)out_size * 2, 0);
    t.kk.assign((size_t)out_size * t.ksize, 0.0);
    const double ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double* k = &t.kk[(size_t)xx * t.ksize];
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) {
            const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; ++x)
            if (ww != 0.0) k[x] /= ww;
        t.bounds[2 * xx] = xmin;
        t.bounds[2 * xx + 1] = xmax;
    }
}

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

// one pass of Resample.c over LDS: dst[o][j] (vertical) or dst[j][o] (horizontal) = (float) sum_t src[..] * k[o][t]
template <bool HORIZ>
__device__ __forceinline__ float resample_at(const float* src, int src_stride, int line, int o, const int* bnd,
                                             const double* kk, int ksize) {
#pragma clang fp contract(off)      // HIP contracts a*b+c into an fma by default, and __dmul_rn/__dadd_rn are plain * and +
    const int first = bnd[2 * o], n = bnd[2 * o + 1];
    const double* k = kk + o * ksize;
    const float* s = HORIZ ? src + line * src_stride + first : src + first * src_stride + line;
    double ss = 0.0;
    for (int t = 0; t < n; ++t) ss = ss + (double)s[HORIZ ? t : t * src_stride] * k[t];
    return (float)ss;
}

__global__ __launch_bounds__(256) void k_resize(ResizeArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = a.H, W = a.W, OH = a.OH, OW = a.OW;
    const int tid = threadIdx.x;
    const bool horiz = a.kh != nullptr, vert = a.kv != nullptr;
    // LDS: weight tables (doubles first: 8-byte alignment), windows, the plane, the intermediate
    double* kh_s = reinterpret_cast<double*>(smem);
    double* kv_s = kh_s + (horiz ? OW * a.ksh : 0);
    int* bh_s = reinterpret_cast<int*>(kv_s + (vert ? OH * a.ksv : 0));
    int* bv_s = bh_s + (horiz ? 2 * OW : 0);
    float* in_s = reinterpret_cast<float*>(bv_s + (vert ? 2 * OH : 0));
    float* tmp_s = in_s + H * W;                    // [H][OW] (only when both passes run)

    if (horiz) {
        for (int i = tid; i < OW * a.ksh; i += 256) kh_s[i] = a.kh[i];
        for (int i = tid; i < 2 * OW; i += 256) bh_s[i] = a.bh[i];
    }
    if (vert) {
        for (int i = tid; i < OH * a.ksv; i += 256) kv_s[i] = a.kv[i];
        for (int i = tid; i < 2 * OH; i += 256) bv_s[i] = a.bv[i];
    }
    const int64_t b = blockIdx.x;
    const float* __restrict__ src = a.in + b * a.in_stride;
    const bool scaled = a.div != 0.0f;
    for (int i = tid; i < H * W; i += 256) {
        const float v = src[i];
        in_s[i] = scaled ? __fdiv_rn(v - a.sub, a.div) : v;
    }
    __syncthreads();
    const float* cur = in_s;
    int cur_w = W;
    const float inv_ow = 1.0f / (float)OW;
    if (horiz) {
        float* dst = tmp_s;
        for (int i = tid; i < H * OW; i += 256) {
            int y = (int)(((float)i + 0.5f) * inv_ow);
            y = y * OW > i ? y - 1 : ((y + 1) * OW <= i ? y + 1 : y);
            const int xx = i - y * OW;
            dst[i] = resample_at<true>(in_s, W, y, xx, bh_s, kh_s, a.ksh);
        }
        __syncthreads();
        cur = dst;
        cur_w = OW;
    }
    float* outf = reinterpret_cast<float*>(a.out) + b * (int64_t)OH * OW;
    uint16_t* outh = reinterpret_cast<uint16_t*>(a.out) + b * (int64_t)OH * OW;
    for (int i = tid; i < OH * OW; i += 256) {
        float v;
        if (vert) {
            int yy = (int)(((float)i + 0.5f) * inv_ow);
            yy = yy * OW > i ? yy - 1 : ((yy + 1) * OW <= i ? yy + 1 : yy);
            const int xx = i - yy * OW;
            v = resample_at<false>(cur, cur_w, xx, yy, bv_s, kv_s, a.ksv);
        } else {
            v = cur[i];
        }
        if (a.out_bf16) outh[i] = f2bf_rne(v);
        else outf[i] = v;
    }
}

// device copy of an axis table, cached in the context (a handful of (in, out) pairs per process)
int axis_table(rml_ctx* ctx, int in_size, int out_size, const int** bounds, const double** kk, int* ksize) {
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
    size_t lds = (size_t)H * W * 4;
    if (out_w != W) {
        int rc = axis_table(ctx, W, out_w, &a.bh, &a.kh, &a.ksh);
        if (rc != RML_OK) return rc;
        lds += (size_t)out_w * a.ksh * 8 + (size_t)out_w * 8 + (size_t)H * out_w * 4;
    }
    if (out_h != H) {
        int rc = axis_table(ctx, H, out_h, &a.bv, &a.kv, &a.ksv);
        if (rc != RML_OK) return rc;
        lds += (size_t)out_h * a.ksv * 8 + (size_t)out_h * 8;
    }
    RML_REQUIRE(lds <= 150 * 1024, RML_ERR_UNSUPPORTED, "rml_resize_bicubic: %dx%d -> %dx%d does not fit the LDS-resident resize", H, W, out_h, out_w);
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_resize), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done = true; }
    hipLaunchKernelGGL(k_resize, dim3((unsigned)B), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
