// Input preprocessing of the multi-view CNN (dnn.py:200-254: p -> (p - 127.5) / 127.5, then
// Image.fromarray(p).resize((80, 80), Image.BICUBIC) per projection) for the bf16 conv trunk (dnn.hip): all three projections of a
// feature row [xz | yz | xy] in ONE launch, bf16 out.
//
// rml_resize_bicubic (resize.hip) reproduces Pillow bit for bit -- float64 multiply and add per tap, tap order kept -- and is
// VALU-bound on that arithmetic (0.36 ms per 8 192 Walabot samples, three launches).  The trunk rounds its input to bf16 (8 bits of
// significand), so this kernel computes the same resize -- Pillow's windows and normalised weights, from the same host tables, rounded
// to float32 -- with float32 fused multiply-adds and partial sums: within ~1e-6 of the exact value before the bf16 rounding, i.e. the
// bf16 result differs from the rounded exact one by at most one bf16 ulp on a small fraction of the pixels (tests/test_nn_gpu.py
// measures both).  It is NOT the parity surface: rml_resize_bicubic stays the Pillow-exact entry point.
//
// Input per row, chosen by a per-row flag, one kernel instantiation per kind: the biased uint8 code row the projection kernels write
// (k_pre3<true>: a quarter of the float row's bytes; exact whenever the projections are integers 0..255, which the flag says) or the
// float32 feature row (k_pre3<false>: the rows whose flag is clear; predicated on the device in rml_dnn_preprocess_volumes).
//
// One 256-thread workgroup per sample, persistent, the next sample's code row prefetched into registers:
//  * staging: the raw codes as float16 (exact; float rows: float32) images in LDS, row strides an odd number of 16-byte slots; the
//    small xy image linear;
//  * horizontal pass: a thread owns an output column (its <= 16 window weights, shifted to the 8- / 16-byte grid and zero padded,
//    stay in registers for the whole launch) and walks the rows: the window is four ds_read_b64 (float32 images: ds_read_b128) and
//    v_fma_mix_f32 taps -- lanes of a wave start 0.4-2.2 elements apart, so a lane group touches distinct or identical (broadcast)
//    slots: conflict-free; (p - 127.5) / 127.5 is applied to the result (the weights of a window sum to 1);
//  * vertical pass: a thread produces four adjacent outputs of a row from four ds_read_b128 (the row's weights from a 16-byte record
//    in LDS) with v_pk_fma_f32 and stores them as one 8-byte piece -- consecutive threads write consecutive bytes.
// Where its time goes, and what was tried on top (matrix-core horizontal pass, projection-by-projection intermediate):
// tools/exp/README.md, round 4.
#include "rml_internal.h"
#include "resize_tables.h"
#include <type_traits>
#include <array>
#include <map>
#include <mutex>

namespace {

using rmlresize::AxisTable;
using rmlresize::precompute;

constexpr int PU = 5;       // 16-element units of a row a thread holds (rows up to 20 480 elements)
constexpr int VT = 4;       // taps of the vertical pass (Pillow's bicubic window when the height does not shrink)

struct PreArgs {
    const float* rows; int64_t ld;          // float32 feature rows (k_pre3<false>)
    const uint8_t* codes; int64_t ldq;      // biased uint8 code rows, byte = code ^ 0x80 (k_pre3<true>)
    const int32_t* flags;                   // per row: != 0 -> the code row is valid (nullptr: every row)
    const int32_t* skip_if_set;             // k_pre3<false>: the whole launch is a no-op when *skip_if_set != 0
    int64_t B;
    int X, Y, Z, OH, OW;
    int SZ, TW;                             // LDS row strides in elements: [xz | yz] image, intermediate
    const int* hz_a0; const float* hz_w;    // Z -> OW: aligned window start per output column, [OW][WZ] weights
    const int* hy_a0; const float* hy_w;    // Y -> OW: window start, [OW][WY] weights (unshifted)
    const float* vw; const int* vfirst;     // [2][OH][VT] weights, [2][OH] first rows: table 0 = X -> OH (xz, xy), 1 = Y -> OH (yz)
    uint16_t* out[3];                       // bf16 (B, OH, OW) per projection
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {     // v_cvt_pk_bf16_f32 (round to nearest even)
    bf16x2 b = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
    return *reinterpret_cast<uint32_t*>(&b);
}
// LDS-only barrier: __syncthreads() would also wait for the prefetch of the next sample (vmcnt)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the reference's (p - 127.5) / 127.5 (dnn.py:202-205), applied to the horizontal pass's result: the weights of a window sum to 1
__device__ __forceinline__ float unit_range(float p) { return fmaf(p, 1.0f / 127.5f, -1.0f); }

// four consecutive image values (8- / 16-byte aligned) as floats
__device__ __forceinline__ float4 quad(const _Float16* s) {
    const f16x4 v = *reinterpret_cast<const f16x4*>(s);
    return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
}
__device__ __forceinline__ float4 quad(const float* s) { return *reinterpret_cast<const float4*>(s); }

template <int WIN, typename T>
__device__ __forceinline__ float window_dot(const T* s, const float (&w)[WIN]) {
    float ax = 0.0f, ay = 0.0f, az = 0.0f, aw = 0.0f;
#pragma unroll
    for (int q = 0; q < WIN / 4; ++q) {
        const float4 v = quad(s + 4 * q);
        ax = fmaf(v.x, w[4 * q], ax);
        ay = fmaf(v.y, w[4 * q + 1], ay);
        az = fmaf(v.z, w[4 * q + 2], az);
        aw = fmaf(v.w, w[4 * q + 3], aw);
    }
    return (ax + ay) + (az + aw);
}
// the same on an unaligned window, element by element (the small xy image, kept linear)
template <int WIN, typename T>
__device__ __forceinline__ float window_dot1(const T* s, const float (&w)[WIN]) {
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int t = 0; t + 1 < WIN; t += 2) {
        a0 = fmaf((float)s[t], w[t], a0);
        a1 = fmaf((float)s[t + 1], w[t + 1], a1);
    }
    if (WIN & 1) a0 = fmaf((float)s[WIN - 1], w[WIN - 1], a0);
    return a0 + a1;
}

// One sample per workgroup iteration.  CODES: code rows in, raw codes as float16 in LDS (integers 0..255 are exact in float16: half
// the LDS bytes of the float path, ds_read_b64 windows, v_fma_mix_f32 taps); !CODES: float32 rows in (the rows whose flag is NOT
// set: projections that left the code grid), float32 images.
template <bool CODES, int WZ, int WY>
__global__ __launch_bounds__(256) void k_pre3(PreArgs a) {
    typedef typename std::conditional<CODES, _Float16, float>::type T;
    if (!CODES && a.skip_if_set && *a.skip_if_set) return;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int X = a.X, Y = a.Y, Z = a.Z, OH = a.OH, OW = a.OW, SZ = a.SZ, TW = a.TW;
    const int R1 = X + Y, XY = X * Y, D = R1 * Z + XY;
    const int nimg = R1 * SZ + 16 + ((XY + 15) & ~15) + 32;     // image elements: [R1][SZ] + 16 pad, then xy linear + pad
    float* tmp = reinterpret_cast<float*>(smem);    // [R1 + X + 3][TW]: xz, yz, xy rows after the horizontal pass, 3 pad rows
    float* vw_s = tmp + (R1 + X + 3) * TW;          // [2][OH][VT]
    int* vf_s = reinterpret_cast<int*>(vw_s + 2 * OH * VT);     // [2][OH]
    T* inA = reinterpret_cast<T*>(vf_s + ((2 * OH + 3) & ~3));  // [R1][SZ] + 16: xz rows, then yz rows
    T* inB = inA + R1 * SZ + 16;                    // [XY] linear + pad
    // every pad (row tails, the elements behind each image, the pad rows of the intermediate) is read under a zero weight and must
    // be finite: zeroed once, never written again
    for (int i = tid; i < (R1 + X + 3) * TW; i += 256) tmp[i] = 0.0f;
    for (int i = tid; i < nimg; i += 256) inA[i] = (T)0.0f;
    for (int i = tid; i < 2 * OH * VT; i += 256) vw_s[i] = a.vw[i];
    for (int i = tid; i < 2 * OH; i += 256) vf_s[i] = a.vfirst[i];

    // horizontal pass: thread (g, xx) with its window weights in registers
    const int G = 256 / OW;
    const int hg = tid / OW, hx = tid - hg * OW;
    const bool hact = hg < G;
    float wz[WZ], wy[WY];
    int az = 0, ay = 0;
#pragma unroll
    for (int t = 0; t < WZ; ++t) wz[t] = 0.0f;
#pragma unroll
    for (int t = 0; t < WY; ++t) wy[t] = 0.0f;
    if (hact) {
        az = a.hz_a0[hx]; ay = a.hy_a0[hx];
#pragma unroll
        for (int t = 0; t < WZ; ++t) wz[t] = a.hz_w[hx * WZ + t];
#pragma unroll
        for (int t = 0; t < WY; ++t) wy[t] = a.hy_w[hx * WY + t];
    }

    // staging map of this thread (fixed for the launch): 16-element units of the row [xz | yz | xy]; a unit of the first two planes
    // lies inside one image row (Z % 16 == 0), xy is kept linear
    const int nU = (D + 15) >> 4, nu = (nU + 255) >> 8;
    int dst[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int c = u * 256 + tid;
        const int e = 16 * (c < nU ? c : 0);
        const int row = e / Z;
        dst[u] = c >= nU ? -1 : (e < R1 * Z ? row * SZ + (e - row * Z) : R1 * SZ + 16 + (e - R1 * Z));
    }

    uint4 pa[PU];
    auto wanted = [&](int64_t bb) -> bool {         // is row bb this kernel's?
        if (!a.flags) return true;
        return (a.flags[bb] != 0) == CODES;
    };
    auto issue = [&](int64_t bb) {                  // CODES: the code row into registers
        const uint4* s16 = reinterpret_cast<const uint4*>(a.codes + bb * a.ldq);
#pragma unroll
        for (int u = 0; u < PU; ++u)
            if (u < nu) { const int c = u * 256 + tid; pa[u] = s16[c < nU ? c : nU - 1]; }
    };
    auto stage = [&](int64_t bb) {
        if constexpr (CODES) {
#pragma unroll
            for (int u = 0; u < PU; ++u)
                if (u < nu && dst[u] >= 0) {
                    const uint4 w4 = pa[u];
                    const uint32_t ws[4] = {w4.x ^ 0x80808080u, w4.y ^ 0x80808080u, w4.z ^ 0x80808080u, w4.w ^ 0x80808080u};
                    f16x8* d = reinterpret_cast<f16x8*>(inA + dst[u]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const uint32_t w0 = ws[2 * h], w1 = ws[2 * h + 1];
                        f16x8 v;
                        v[0] = (_Float16)(float)(w0 & 0xFFu); v[1] = (_Float16)(float)((w0 >> 8) & 0xFFu);
                        v[2] = (_Float16)(float)((w0 >> 16) & 0xFFu); v[3] = (_Float16)(float)(w0 >> 24);
                        v[4] = (_Float16)(float)(w1 & 0xFFu); v[5] = (_Float16)(float)((w1 >> 8) & 0xFFu);
                        v[6] = (_Float16)(float)((w1 >> 16) & 0xFFu); v[7] = (_Float16)(float)(w1 >> 24);
                        d[h] = v;
                    }
                }
        } else {
            // the float rows are the rare case (projections off the code grid): loaded here, no prefetch
            const float* src = a.rows + bb * a.ld;
#pragma unroll
            for (int u = 0; u < PU; ++u)
                if (u < nu && dst[u] >= 0) {
                    const int e = 16 * (u * 256 + tid);
                    float* d = inA + dst[u];
                    if (e + 16 <= D) {
                        const f32x4u* p = reinterpret_cast<const f32x4u*>(src + e);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4u v = p[q];
                            *reinterpret_cast<float4*>(d + 4 * q) = make_float4(v.x, v.y, v.z, v.w);
                        }
                    } else {
                        for (int t = 0; e + t < D; ++t) d[t] = src[e + t];
                    }
                }
        }
    };

    const int OW4 = OW >> 2;
    const int per = OH * OW4;                       // 8-byte output pieces per projection
    const int vdq = 256 / OW4, vdr = 256 - vdq * OW4;
    const int64_t opl = (int64_t)OH * OW;

    int64_t b = blockIdx.x;
    if (CODES && b < a.B) issue(b);
    __syncthreads();                                // LDS initialised (this one may wait for the loads: once)

    for (; b < a.B; b += gridDim.x) {
        const bool mine = wanted(b);                // uniform
        if (mine) stage(b);
        if (CODES && b + gridDim.x < a.B) issue(b + gridDim.x);
        if (!mine) continue;
        lds_barrier();

        if (hact) {
            // rows of [xz | yz] (length Z) and of xy (length Y) -> column hx of the intermediate, scaled to [-1, 1]
            float* tc = tmp + hx;
            int y = hg;
            for (; y + G < R1; y += 2 * G) {        // two rows at a time: independent chains
                const float r0 = window_dot<WZ, T>(inA + y * SZ + az, wz);
                const float r1 = window_dot<WZ, T>(inA + (y + G) * SZ + az, wz);
                tc[y * TW] = unit_range(r0);
                tc[(y + G) * TW] = unit_range(r1);
            }
            if (y < R1) tc[y * TW] = unit_range(window_dot<WZ, T>(inA + y * SZ + az, wz));
            tc += R1 * TW;
            for (y = hg; y < X; y += G) tc[y * TW] = unit_range(window_dot1<WY, T>(inB + y * Y + ay, wy));
        }
        lds_barrier();

#pragma unroll 1
        for (int p = 0; p < 3; ++p) {
            const float* tp = tmp + (p == 0 ? 0 : (p == 1 ? X : R1)) * TW;
            const float* wv = vw_s + (p == 1 ? OH * VT : 0);
            const int* fv = vf_s + (p == 1 ? OH : 0);
            uint2* o8 = reinterpret_cast<uint2*>(a.out[p] + b * opl);
            int yy = tid / OW4, c4 = tid - yy * OW4;
            for (int r = tid; r < per; r += 256) {
                const float4 w = *reinterpret_cast<const float4*>(wv + yy * VT);
                const float* s = tp + fv[yy] * TW + 4 * c4;
                const float4 v0 = *reinterpret_cast<const float4*>(s);
                const float4 v1 = *reinterpret_cast<const float4*>(s + TW);
                const float4 v2 = *reinterpret_cast<const float4*>(s + 2 * TW);
                const float4 v3 = *reinterpret_cast<const float4*>(s + 3 * TW);
                // two outputs per instruction (v_pk_mul_f32 / v_pk_fma_f32): the pairs (x, y) and (z, w) of the four outputs
                const f32x2 w0 = {w.x, w.x}, w1 = {w.y, w.y}, w2 = {w.z, w.z}, w3 = {w.w, w.w};
                f32x2 lo = f32x2{v0.x, v0.y} * w0, hi = f32x2{v0.z, v0.w} * w0;
                lo = __builtin_elementwise_fma(f32x2{v1.x, v1.y}, w1, lo); hi = __builtin_elementwise_fma(f32x2{v1.z, v1.w}, w1, hi);
                lo = __builtin_elementwise_fma(f32x2{v2.x, v2.y}, w2, lo); hi = __builtin_elementwise_fma(f32x2{v2.z, v2.w}, w2, hi);
                lo = __builtin_elementwise_fma(f32x2{v3.x, v3.y}, w3, lo); hi = __builtin_elementwise_fma(f32x2{v3.z, v3.w}, w3, hi);
                o8[r] = make_uint2(pk_bf16(lo.x, lo.y), pk_bf16(hi.x, hi.y));
                yy += vdq; c4 += vdr;
                if (c4 >= OW4) { c4 -= OW4; ++yy; }
            }
        }
        lds_barrier();                              // the images and the intermediate are free for the next sample
    }
}

// float32 window weights of one axis for the horizontal pass: zero padded to `win` taps and, with `align`, shifted so that the window
// starts on a multiple of four elements; false when a window does not fit
bool window_table(const AxisTable& t, int win, bool align, std::vector<int>& a0, std::vector<float>& w) {
    a0.assign(t.out, 0);
    w.assign((size_t)t.out * win, 0.0f);
    for (int xx = 0; xx < t.out; ++xx) {
        const int first = t.bounds[2 * xx], n = t.bounds[2 * xx + 1];
        const int al = align ? (first & ~3) : first, sh = first - al;
        if (sh + n > win) return false;
        a0[xx] = al;
        for (int k = 0; k < n; ++k) w[(size_t)xx * win + sh + k] = (float)t.kk[(size_t)xx * t.ksize + k];
    }
    return true;
}

AxisTable axis_or_identity(int in, int out) {
    AxisTable t;
    if (in != out) { precompute(in, out, t); return t; }
    // Pillow does not resample an axis whose size does not change
    t.in = in; t.out = out; t.ksize = 1;
    t.bounds.resize((size_t)out * 2);
    t.kk.assign(out, 1.0);
    for (int i = 0; i < out; ++i) { t.bounds[2 * i] = i; t.bounds[2 * i + 1] = 1; }
    return t;
}

int odd_slots(int n, int per_slot) {        // row stride in elements: whole 16-byte slots (per_slot elements each), an odd number of them
    int s = (n + per_slot - 1) / per_slot;
    if (!(s & 1)) ++s;
    return per_slot * s;
}

struct Plan {
    bool ok = false;
    int wz = 0, wy = 0, TW = 0;
    int SZ[2] = {0, 0};         // image row stride: float32 images, float16 images
    size_t lds[2] = {0, 0};
    std::vector<int> hz_a0, hy_a0, vfirst;
    std::vector<float> hz_w, hy_w, vw;
};

Plan build_plan(int X, int Y, int Z, int OH, int OW);

// the plan of a (grid, output size) pair is a pure function of the five numbers: built once per process, not once per call
// (four Pillow coefficient tables and their window tables: ~10^4 double operations at the Walabot grid)
Plan make_plan(int X, int Y, int Z, int OH, int OW) {
    static std::mutex mu;
    static std::map<std::array<int, 5>, Plan> cache;
    const std::array<int, 5> key{X, Y, Z, OH, OW};
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it == cache.end()) {
        if (cache.size() >= 64) cache.clear();          // a bound, not a policy: callers use a handful of shapes
        it = cache.emplace(key, build_plan(X, Y, Z, OH, OW)).first;
    }
    return it->second;                                  // a copy, made under the lock (a few KB of tables)
}

Plan build_plan(int X, int Y, int Z, int OH, int OW) {
    Plan p;
    if (X <= 0 || Y <= 0 || Z <= 0 || OH <= 0 || OW <= 0) return p;
    const int64_t D = (int64_t)(X + Y) * Z + (int64_t)X * Y;
    if (Z % 16 || OW % 4 || OW > 256 || D > 4096 * PU) return p;
    const AxisTable tz = axis_or_identity(Z, OW), ty = axis_or_identity(Y, OW);
    const AxisTable vx = axis_or_identity(X, OH), vy = axis_or_identity(Y, OH);
    for (int win : {8, 16})
        if (!p.wz && window_table(tz, win, true, p.hz_a0, p.hz_w)) p.wz = win;
    for (int win : {5, 12})
        if (!p.wy && window_table(ty, win, false, p.hy_a0, p.hy_w)) p.wy = win;
    if (!p.wz || !p.wy) return p;
    if (p.wz == 8 && p.wy == 12) {                  // instantiated: (16, 5), (8, 5), (16, 12)
        p.wz = 16;
        window_table(tz, 16, true, p.hz_a0, p.hz_w);
    }
    p.vw.assign((size_t)2 * OH * VT, 0.0f);
    p.vfirst.assign((size_t)2 * OH, 0);
    const AxisTable* vt[2] = {&vx, &vy};
    for (int k = 0; k < 2; ++k)
        for (int yy = 0; yy < OH; ++yy) {
            const int n = vt[k]->bounds[2 * yy + 1];
            if (n > VT) return p;
            p.vfirst[(size_t)k * OH + yy] = vt[k]->bounds[2 * yy];
            for (int t = 0; t < n; ++t) p.vw[((size_t)k * OH + yy) * VT + t] = (float)vt[k]->kk[(size_t)yy * vt[k]->ksize + t];
        }
    p.TW = odd_slots(OW, 4);
    const size_t head = ((size_t)(2 * X + Y + 3) * p.TW + (size_t)2 * OH * VT + (((size_t)2 * OH + 3) & ~(size_t)3)) * 4;
    for (int k = 0; k < 2; ++k) {
        const int esz = k ? 2 : 4;
        p.SZ[k] = odd_slots(Z, 16 / esz);
        const size_t nimg = (size_t)(X + Y) * p.SZ[k] + 16 + (((size_t)X * Y + 15) & ~(size_t)15) + 32;
        p.lds[k] = head + nimg * esz;
    }
    if (p.lds[0] > 158 * 1024) return p;
    p.ok = true;
    return p;
}

// device copy of a plan's tables, cached in the context per (X, Y, Z, OH, OW)
int pre_tables(rml_ctx* ctx, int X, int Y, int Z, int OH, int OW, const Plan& p, PreArgs& a) {
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    for (int pass = 0; pass < 2; ++pass) {
        for (const auto& e : ctx->pre_tabs)
            if (e.X == X && e.Y == Y && e.Z == Z && e.OH == OH && e.OW == OW) {
                const unsigned char* base = static_cast<const unsigned char*>(e.dev);
                a.hz_a0 = reinterpret_cast<const int*>(base + e.off[0]);
                a.hz_w = reinterpret_cast<const float*>(base + e.off[1]);
                a.hy_a0 = reinterpret_cast<const int*>(base + e.off[2]);
                a.hy_w = reinterpret_cast<const float*>(base + e.off[3]);
                a.vw = reinterpret_cast<const float*>(base + e.off[4]);
                a.vfirst = reinterpret_cast<const int*>(base + e.off[5]);
                return RML_OK;
            }
        rml_pre_tab e{};
        e.X = X; e.Y = Y; e.Z = Z; e.OH = OH; e.OW = OW;
        const void* src[6] = {p.hz_a0.data(), p.hz_w.data(), p.hy_a0.data(), p.hy_w.data(), p.vw.data(), p.vfirst.data()};
        const size_t len[6] = {p.hz_a0.size() * 4, p.hz_w.size() * 4, p.hy_a0.size() * 4, p.hy_w.size() * 4, p.vw.size() * 4, p.vfirst.size() * 4};
        size_t tot = 0;
        for (int i = 0; i < 6; ++i) { e.off[i] = tot; tot += (len[i] + 15) & ~(size_t)15; }
        void* d = nullptr;
        RML_HIP(hipMalloc(&d, tot));
        for (int i = 0; i < 6; ++i) {
            hipError_t err = hipMemcpy(static_cast<unsigned char*>(d) + e.off[i], src[i], len[i], hipMemcpyHostToDevice);
            if (err != hipSuccess) { (void)hipFree(d); RML_HIP(err); }
        }
        e.dev = d;
        ctx->pre_tabs.push_back(e);
    }
    return RML_ERR_INVALID;
}

template <bool CODES, int WZ, int WY>
void launch_pre3(const PreArgs& a, size_t lds, int num_cu, hipStream_t st) {
    RML_MAX_DYN_LDS(160 * 1024, &k_pre3<CODES, WZ, WY>);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pre3<CODES, WZ, WY>, 256, lds) != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        per_cu = 1;
    }
    const int64_t slots = (int64_t)per_cu * num_cu;
    hipLaunchKernelGGL((k_pre3<CODES, WZ, WY>), dim3((unsigned)(a.B < slots ? a.B : slots)), dim3(256), lds, st, a);
}

template <bool CODES>
void launch_pre3_w(const Plan& p, const PreArgs& a, int num_cu, hipStream_t st) {
    const size_t lds = p.lds[CODES ? 1 : 0];
    if (p.wz == 16 && p.wy == 5) launch_pre3<CODES, 16, 5>(a, lds, num_cu, st);
    else if (p.wz == 8 && p.wy == 5) launch_pre3<CODES, 8, 5>(a, lds, num_cu, st);
    else launch_pre3<CODES, 16, 12>(a, lds, num_cu, st);
}

}  // namespace

extern "C" int rml_dnn_preprocess_supported(int X, int Y, int Z, int out_h, int out_w) {
    const Plan p = make_plan(X, Y, Z, out_h, out_w);
    return p.ok ? 1 : 0;
}

// skip_rows: device flag "every row is on the code grid" (the float-row launch exits at once when it is set), or nullptr
static int preprocess_rows(rml_ctx* ctx, const float* rows, int64_t ld, const uint8_t* codes, int64_t ldq, const int32_t* flags,
                           const int32_t* skip_rows, int64_t B, int X, int Y, int Z, int out_h, int out_w, uint16_t* xz, uint16_t* yz,
                           uint16_t* xy, void* stream) {
    RML_REQUIRE(ctx && B >= 0, RML_ERR_INVALID, "rml_dnn_preprocess_rows: bad arguments");
    const Plan p = make_plan(X, Y, Z, out_h, out_w);
    RML_REQUIRE(p.ok, RML_ERR_UNSUPPORTED, "rml_dnn_preprocess_rows: %dx%dx%d -> %dx%d has no fused kernel (rml_resize_bicubic per projection does it)",
                X, Y, Z, out_h, out_w);
    if (B == 0) return RML_OK;
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_dnn_preprocess_rows: B too large");
    RML_REQUIRE((rows || codes) && xz && yz && xy, RML_ERR_INVALID, "rml_dnn_preprocess_rows: NULL argument");
    RML_REQUIRE(!(rows && codes) || flags, RML_ERR_INVALID, "rml_dnn_preprocess_rows: float rows AND code rows need the row flags");
    // (one kind of row WITH flags: only the rows whose flag selects that kind are written -- the two launches of
    // rml_dnn_preprocess_volumes on its two streams)
    const int64_t D = (int64_t)X * Z + (int64_t)Y * Z + (int64_t)X * Y;
    RML_REQUIRE(!rows || ld >= D, RML_ERR_INVALID, "rml_dnn_preprocess_rows: ld < D");
    RML_REQUIRE(!codes || (ldq >= D && ldq % 16 == 0 && (reinterpret_cast<uintptr_t>(codes) & 15) == 0), RML_ERR_INVALID,
                "rml_dnn_preprocess_rows: code rows need ldq >= D, ldq %% 16 == 0 and 16-byte alignment");
    RML_REQUIRE(!rows || (reinterpret_cast<uintptr_t>(rows) & 3) == 0, RML_ERR_INVALID, "rml_dnn_preprocess_rows: rows misaligned");
    RML_REQUIRE(((reinterpret_cast<uintptr_t>(xz) | reinterpret_cast<uintptr_t>(yz) | reinterpret_cast<uintptr_t>(xy)) & 7) == 0,
                RML_ERR_INVALID, "rml_dnn_preprocess_rows: outputs need 8-byte alignment");
    RML_HIP(hipSetDevice(ctx->device));
    PreArgs a{};
    a.rows = rows; a.ld = ld; a.codes = codes; a.ldq = ldq; a.B = B;
    a.X = X; a.Y = Y; a.Z = Z; a.OH = out_h; a.OW = out_w; a.TW = p.TW;
    a.out[0] = xz; a.out[1] = yz; a.out[2] = xy;
    int rc = pre_tables(ctx, X, Y, Z, out_h, out_w, p, a);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    a.flags = flags;                                // nullptr: every row
    if (codes) {
        a.SZ = p.SZ[1];
        launch_pre3_w<true>(p, a, ctx->num_cu, st);
    }
    if (rows) {
        a.skip_if_set = skip_rows;
        a.SZ = p.SZ[0];
        launch_pre3_w<false>(p, a, ctx->num_cu, st);
    }
    RML_HIP(hipGetLastError());
    return RML_OK;
}

extern "C" int rml_dnn_preprocess_rows(rml_ctx* ctx, const float* rows, int64_t ld, const uint8_t* codes, int64_t ldq,
                                       const int32_t* flags, int64_t B, int X, int Y, int Z, int out_h, int out_w,
                                       uint16_t* xz, uint16_t* yz, uint16_t* xy, void* stream) {
    return preprocess_rows(ctx, rows, ld, codes, ldq, flags, nullptr, B, X, Y, Z, out_h, out_w, xz, yz, xy, stream);
}

namespace {
// flags[B] = 1 when every row flag is set (the predicate of the float-row pass below), else 0
__global__ __launch_bounds__(256) void k_all_set(int32_t* flags, int64_t B) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int mine = 0;
    if ((reinterpret_cast<uintptr_t>(flags) & 15) == 0) {       // eight independent 16-byte loads per thread and trip
        const int4* f4 = reinterpret_cast<const int4*>(flags);
        const int64_t n4 = B >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += 256 * 8) {
            int4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int64_t idx = i + u * 256; v[u] = idx < n4 ? f4[idx] : make_int4(1, 1, 1, 1); }
#pragma unroll
            for (int u = 0; u < 8; ++u) mine |= (v[u].x == 0) | (v[u].y == 0) | (v[u].z == 0) | (v[u].w == 0);
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < B; i += 256) mine |= flags[i] == 0;
    } else {
        for (int64_t i = threadIdx.x; i < B; i += 256) mine |= flags[i] == 0;
    }
    if (mine) bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) flags[B] = bad ? 0 : 1;
}
}  // namespace

extern "C" int rml_dnn_preprocess_volumes(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode,
                                          const int32_t* ijk, uint8_t* codes, int64_t ldq, int32_t* flags, float* rows, int64_t ld,
                                          int out_h, int out_w, uint16_t* xz, uint16_t* yz, uint16_t* xy, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && X > 0 && Y > 0 && Z > 0, RML_ERR_INVALID, "rml_dnn_preprocess_volumes: bad arguments");
    RML_REQUIRE(rml_dnn_preprocess_supported(X, Y, Z, out_h, out_w), RML_ERR_UNSUPPORTED,
                "rml_dnn_preprocess_volumes: %dx%dx%d -> %dx%d has no fused kernel", X, Y, Z, out_h, out_w);
    if (B == 0) return RML_OK;
    RML_REQUIRE(V && codes && flags, RML_ERR_INVALID, "rml_dnn_preprocess_volumes: NULL argument");
    RML_REQUIRE(vdtype == RML_VOL_F32 || vdtype == RML_VOL_U8, RML_ERR_INVALID, "rml_dnn_preprocess_volumes: unknown volume dtype %d", vdtype);
    RML_REQUIRE(vdtype == RML_VOL_U8 || rows, RML_ERR_INVALID, "rml_dnn_preprocess_volumes: float32 volumes need the float-row scratch");
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_dnn_preprocess_volumes: B too large");
    const int64_t D = (int64_t)X * Z + (int64_t)Y * Z + (int64_t)X * Y;
    RML_REQUIRE(ldq >= D && ldq % 16 == 0 && (reinterpret_cast<uintptr_t>(codes) & 15) == 0, RML_ERR_INVALID,
                "rml_dnn_preprocess_volumes: code rows need ldq >= D, ldq %% 16 == 0 and 16-byte alignment");
    RML_REQUIRE(!rows || ld >= D, RML_ERR_INVALID, "rml_dnn_preprocess_volumes: ld < D");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // first pass: the projections as code rows + the per-row "every value is an integer in [0, 255]" flag -- no float rows
    ProjOut o{};
    int64_t off = 0;
    const int64_t plen[3] = {(int64_t)X * Z, (int64_t)Y * Z, (int64_t)X * Y};
    for (int pl = 0; pl < 3; ++pl) { o.q[pl] = codes + off; off += plen[pl]; }
    o.sel = RML_MASK_ALL;
    o.qstride = ldq; o.qrow = nullptr; o.qD = D;
    o.row_flags = flags;
    o.scale_div = 0.0f;
    int rc = rml_launch_project(ctx, V, vdtype, B, X, Y, Z, mode, ijk, o, st);
    if (rc) return rc;
    if (vdtype == RML_VOL_U8)           // bytes cannot leave the code grid
        return preprocess_rows(ctx, nullptr, 0, codes, ldq, nullptr, nullptr, B, X, Y, Z, out_h, out_w, xz, yz, xy, stream);
    // float32 volumes: rows that left the code grid need their float32 projections -- a second projection pass and the float-row
    // preprocessing launch, both predicated on the device (they exit at once when every row of the batch is on the grid: radar
    // returns are integers 0..255, common.py:30-31).  The three small launches of that fallback run on the context's second stream
    // BESIDE the code-row preprocessing of the batch (disjoint output rows), not in front of it.
    rml_ctx_guard guard(ctx, st);
    hipStream_t aux = ctx->aux_stream;
    RML_HIP(hipEventRecord(ctx->ev_fork, st));
    RML_HIP(hipStreamWaitEvent(aux, ctx->ev_fork, 0));
    hipLaunchKernelGGL(k_all_set, dim3(1), dim3(256), 0, aux, flags, B);
    ProjOut of{};
    off = 0;
    for (int pl = 0; pl < 3; ++pl) { of.p[pl] = rows + off; of.stride[pl] = ld; off += plen[pl]; }
    of.sel = RML_MASK_ALL;
    of.scale_div = 0.0f;
    of.skip_if_set = flags + B;
    of.no_pad = 1;
    rc = rml_launch_project(ctx, V, vdtype, B, X, Y, Z, mode, ijk, of, aux);
    if (rc) return rc;
    rc = preprocess_rows(ctx, rows, ld, nullptr, 0, flags, flags + B, B, X, Y, Z, out_h, out_w, xz, yz, xy, aux);
    if (rc) return rc;
    RML_HIP(hipEventRecord(ctx->ev_join, aux));
    rc = preprocess_rows(ctx, nullptr, 0, codes, ldq, flags, nullptr, B, X, Y, Z, out_h, out_w, xz, yz, xy, stream);
    if (rc) return rc;
    RML_HIP(hipStreamWaitEvent(st, ctx->ev_join, 0));
    return RML_OK;
}
