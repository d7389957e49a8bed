// Non-unit projection zoom: scipy.ndimage.zoom(p, zoom) with SciPy's defaults (order-3 B-spline, mode
// 'constant', prefilter) as called by common.process_samples (common.py:143) when the predict arena differs
// from the training arena (predict.py:34-54,109-116).
//
// SciPy's algorithm (scipy/ndimage/_interpolation.py zoom -> spline_filter -> NI_ZoomShift), restated:
//   1. prefilter: separable cubic B-spline IIR, pole z = sqrt(3)-2, gain (1-z)(1-1/z) = 6, applied along axis 0
//      then axis 1 in float64; mode 'constant' uses the MIRROR boundary initialisation (exact finite sum for the
//      causal start, closed form for the anti-causal start);
//   2. resample: output shape = round(in*zoom) (Python banker's rounding -- computed by the caller), coordinate
//      of output index o = o * (in-1)/(out-1), 4x4 cubic B-spline taps around floor(coordinate), taps outside
//      the array mirrored about 0 and n-1; result cast to float32, then the optional float32 "/ RADAR_MAX".
// One workgroup per (sample, plane); the plane lives in LDS as float64 (<= 90 KB at the Walabot arena).
#include "rml_internal.h"
#include "spline_dev.h"
#include <math.h>
#include <algorithm>

namespace {

using namespace rml_spline;

struct ZoomPlane {
    const float* src; int H, W, OH, OW;
    int64_t out_off;      // offset of this plane inside the feature row
};
struct ZoomArgs {
    ZoomPlane pl[3];
    int npl;
    float* feat; int64_t ld;
    float scale_div;
};

__global__ __launch_bounds__(256) void k_zoom(ZoomArgs a) {
    extern __shared__ __align__(16) double coef[];
    const int64_t b = blockIdx.x;
    const ZoomPlane P = a.pl[blockIdx.y];
    const int H = P.H, W = P.W;
    const float* src = P.src + b * (int64_t)H * W;
    for (int i = threadIdx.x; i < H * W; i += 256) coef[i] = (double)src[i];
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += 256) prefilter_line(coef + w, H, W);       // axis 0
    __syncthreads();
    for (int h = threadIdx.x; h < H; h += 256) prefilter_line(coef + (int64_t)h * W, W, 1);   // axis 1
    __syncthreads();
    const double z0 = P.OH > 1 ? (double)(H - 1) / (double)(P.OH - 1) : 1.0;
    const double z1 = P.OW > 1 ? (double)(W - 1) / (double)(P.OW - 1) : 1.0;
    float* dst = a.feat + b * a.ld + P.out_off;
    for (int o = threadIdx.x; o < P.OH * P.OW; o += 256) {
        const int o0 = o / P.OW, o1 = o - o0 * P.OW;
        const double c0 = o0 * z0, c1 = o1 * z1;
        float r = 0.0f;                                    // cval for out-of-range coordinates (mode 'constant')
        if (c0 >= 0.0 && c0 <= (double)(H - 1) && c1 >= 0.0 && c1 <= (double)(W - 1)) {
            const int f0 = (int)floor(c0), f1 = (int)floor(c1);
            double w0[4], w1[4];
            bspline3(c0 - f0, w0);
            bspline3(c1 - f1, w1);
            double s = 0.0;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const double* row = coef + (int64_t)mirror_idx(f0 - 1 + p, H) * W;
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) t += w1[q] * row[mirror_idx(f1 - 1 + q, W)];
                s += w0[p] * t;
            }
            r = (float)s;
        }
        dst[o] = a.scale_div > 1.0f ? __fdiv_rn(r, a.scale_div) : r;
    }
}

}  // namespace

extern "C" int rml_zoom_features(rml_ctx* ctx, const float* xz, const float* yz, const float* xy,
                                 int64_t B, int X, int Y, int Z, const int32_t* out_shape /* host, 6 */,
                                 float scale_div, uint32_t mask, float* feat, int64_t ld_feat, void* stream) {
    RML_REQUIRE(ctx && out_shape && B >= 0 && X > 0 && Y > 0 && Z > 0, RML_ERR_INVALID, "rml_zoom_features: bad arguments");
    RML_REQUIRE((mask & RML_MASK_ALL) != 0, RML_ERR_INVALID, "rml_zoom_features: empty mask");
    if (B == 0) return RML_OK;
    RML_REQUIRE(feat != nullptr, RML_ERR_INVALID, "rml_zoom_features: feat is NULL");
    const float* src[3] = {xz, yz, xy};
    const int inH[3] = {X, Y, X}, inW[3] = {Z, Z, Y};
    ZoomArgs a{};
    int64_t off = 0;
    size_t lds = 0;
    for (int pl = 0; pl < 3; ++pl) {
        if (!(mask & (1u << pl))) continue;
        RML_REQUIRE(src[pl] != nullptr, RML_ERR_INVALID, "rml_zoom_features: selected plane %d is NULL", pl);
        const int OH = out_shape[2 * pl], OW = out_shape[2 * pl + 1];
        RML_REQUIRE(OH > 0 && OW > 0, RML_ERR_INVALID, "rml_zoom_features: empty output plane %d", pl);
        a.pl[a.npl++] = ZoomPlane{src[pl], inH[pl], inW[pl], OH, OW, off};
        off += (int64_t)OH * OW;
        lds = std::max(lds, (size_t)inH[pl] * inW[pl] * sizeof(double));
    }
    RML_REQUIRE(ld_feat >= off, RML_ERR_INVALID, "rml_zoom_features: ld_feat < D");
    RML_REQUIRE(lds <= 150 * 1024, RML_ERR_UNSUPPORTED, "rml_zoom_features: plane too large for the LDS-resident spline filter");
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_zoom_features: B too large");
    RML_HIP(hipSetDevice(ctx->device));
    a.feat = feat; a.ld = ld_feat; a.scale_div = scale_div;
    RML_MAX_DYN_LDS(160 * 1024, &k_zoom);
    hipLaunchKernelGGL(k_zoom, dim3((unsigned)B, (unsigned)a.npl), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
