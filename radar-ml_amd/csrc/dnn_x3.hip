// k_dnn_trunk_x3: the conv trunk of dnn.py:45-52, 68-76 at float32-class accuracy on the bf16 matrix cores (round 6).
//
// The reference's model.predict is float32 Keras (dnn.py:373-381); the fast chain (k_dnn_trunk_rf) rounds planes, conv1
// activations and weights to bf16 and lands 1e-3 .. 7e-3 away from it in class probability, so rows whose two largest
// probabilities are closer than that are scored again (Classifier._guard).  Until round 5 that second opinion was PyTorch's
// float32 MIOpen layers: 7.9 us per row through ~40 launches -- 56 % on top of the whole call for 1.2 % of the rows.  This
// kernel is that second opinion in the schedule of the fast one: every float32 operand x is carried as TWO bf16 numbers,
// x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 significant bits), and every product as THREE matrix-core products
//      a * b  ~=  a_hi * b_hi + a_hi * b_lo + a_lo * b_hi          (the dropped a_lo * b_lo is <= 2^-16 |a b|)
// accumulated in float32: ~2^-16 per product instead of bf16's 2^-9, 1e-5-class probabilities at 3 x the matrix-core work.
// Arithmetic as k_dnn_trunk_rf (csrc/dnn.hip): float32 planes in LDS with a zero border (TensorFlow's bottom / right 'same'
// padding), conv1 evaluated on the matrix cores AT the 32 conv1 pixels a conv2 tap reads, its float32 accumulators -- relu'd
// and split in registers -- are conv2's B operand; the conv2 weight fragments (hi and lo, 72 KB per branch, in operand order)
// are shared through LDS.  float32 planes are twice the LDS of bf16 ones, so only THREE fit beside the weights (158 KB): a
// workgroup of eight waves takes a group of three samples and deals the group's 3 x 13 tiles of 32 conv2 pixels to its waves
// (two waves per SIMD: one's relu / split instructions under the other's matrix-core products; the first version -- a wave per
// sample, three waves per CU -- ran at 0.305 us per row, session r6b).  conv1's bias rides in three K slots of the hi fragment
// (24 bits, as in k_dnn_trunk_rf), conv2's bias starts the accumulator: both exact.
// k_dnn_trunk_xn<3, ...> ("x6", the guard's LAST word before float64): THREE bf16 parts per operand (24 significant bits = a whole
// float32 mantissa) and the SIX products whose part indices sum to <= 2 -- float32-class in the strict sense (what is dropped is
// <= 2^-24 |a b| per product, the size of float32's own rounding).  108 KB of weight fragments leave room for one plane: four waves
// share one sample's tiles.  Twice the matrix-core work of x3: for the handful of rows whose x3 gap is below LABEL_GUARD_X3.
// In: float32 planes (rml_resize_bicubic's Pillow-bit-identical output), float32 weights.  Out: float32 features, rows in
// Keras' Flatten order feat[b][(h * W/4 + w) * 96 + branch * 32 + n] -- the float32 dense tail runs on them.
#include "rml_internal.h"
#include <algorithm>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int C1 = 64, C2 = 32, KTAPS = 9, K2 = KTAPS * C1;

struct X3Args {
    const float* in[3];     // (B, H, W) float32 per branch
    int64_t B;
    int H, W;
    const float* w1;        // [3][64][9]
    const float* b1;        // [3][64]
    const float* w2;        // [3][32][576] float32, k = (ky*3+kx)*64 + cin
    const float* b2;        // [3][32]
    float* feat;            // [B][P*96] float32
};

// NS bf16 parts per operand; GROUP samples (planes in LDS) per workgroup pass
template <int NS, int GROUP>
struct XnLayout {
    int RSB;                                   // bytes per plane row in LDS: (W + 4) float32
    uint32_t plane, off_w, off_ones, off_zero, off_bias, total;
    __host__ __device__ XnLayout(int H, int W) {
        RSB = (W + 4) * 4;
        plane = ((uint32_t)(H + 4) * RSB + 15) & ~15u;
        off_w = GROUP * plane;                 // [part][tap][k-step][lane] 16 B: conv2 A operands, part p of every weight
        off_ones = off_w + NS * 36 * 1024;
        off_zero = off_ones + 16;
        off_bias = off_zero + 16;              // [h][16] float: conv2 bias in accumulator order
        total = off_bias + 128;
    }
};

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {     // v_cvt_pk_bf16_f32 (round to nearest even)
    bf16x2 b = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
    return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// (x0, x1) -> NS packed bf16 pairs: part 0 = bf16(x), part 1 = bf16 of what that rounding left, part 2 = bf16 of what THAT left
template <int NS>
__device__ __forceinline__ void split2(float x0, float x1, uint32_t (&part)[NS]) {
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        part[p] = pk_bf16(x0, x1);
        if (p + 1 < NS) {
            x0 -= __uint_as_float(part[p] << 16);
            x1 -= __uint_as_float(part[p] & 0xFFFF0000u);
        }
    }
}
template <int NS>
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, bf16x8 (&part)[NS]) {
    uint32_t q[4][NS];
    split2<NS>(a[0], a[1], q[0]); split2<NS>(a[2], a[3], q[1]);
    split2<NS>(b[0], b[1], q[2]); split2<NS>(b[2], b[3], q[3]);
#pragma unroll
    for (int p = 0; p < NS; ++p) {
        uint4 u = make_uint4(q[0][p], q[1][p], q[2][p], q[3][p]);
        part[p] = *reinterpret_cast<bf16x8*>(&u);
    }
}

// acc += sum over part pairs (i, j), i + j < NS, of A_i x B_j, the smallest terms first
template <int NS>
__device__ __forceinline__ f32x16 mfma_parts(const bf16x8 (&A)[NS], const bf16x8 (&B)[NS], f32x16 acc) {
#pragma unroll
    for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
        for (int i = sum; i >= 0; --i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B[sum - i], acc, 0, 0, 0);
    return acc;
}

template <int NS, int WAVES, int GROUP>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void k_dnn_trunk_xn(X3Args a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = a.H, W = a.W, OH2 = H / 4, OW2 = W / 4, P = OH2 * OW2;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XnLayout<NS, GROUP> L(H, W);
    const int RSB = L.RSB;
    const int off_ones = (int)L.off_ones, off_zero = (int)L.off_zero;
    for (uint32_t i = tid; i < (L.total >> 4); i += 64 * WAVES) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (tid < 4) reinterpret_cast<float*>(smem + off_ones)[tid] = 1.0f;

    const int WQ = W / 4, nquad = H * WQ;
    const int NT = (P + 31) >> 5;
    const int64_t ngroups = (a.B + GROUP - 1) / GROUP;

    // work items = (branch, group of GROUP samples), branch-major; a workgroup takes a contiguous run of them -- with many groups it
    // stays inside one branch (one weight load), with few (a handful of rows: the guard's later stages) the branches of a sample
    // run on different CUs
    const int64_t nitems = 3 * ngroups;
    const int64_t per = (nitems + gridDim.x - 1) / gridDim.x;
    const int64_t item0 = blockIdx.x * per, item1 = item0 + per < nitems ? item0 + per : nitems;
    for (int br = (int)(item0 / ngroups); br < 3 && (int64_t)br * ngroups < item1; ++br) {
        const int64_t g0 = item0 > (int64_t)br * ngroups ? item0 - (int64_t)br * ngroups : 0;
        const int64_t g1 = item1 < (int64_t)(br + 1) * ngroups ? item1 - (int64_t)br * ngroups : ngroups;
        __syncthreads();                        // every wave is done with the previous branch's weights
        // conv2 weights in operand order (as k_dnn_trunk_rf): block (tap t, k-step s), lane (cout m, k-group h) holds input channels
        // 16 s + 8 (i / 4) + 4 h + i % 4, i = 0..7 -- the order the conv1 accumulators come in; split into parts here
        for (int i = tid; i < 36 * 64; i += 64 * WAVES) {
            const int blk = i >> 6, l = i & 63, t = blk >> 2, s = blk & 3;
            const float* g = a.w2 + ((size_t)br * C2 + (l & 31)) * K2 + t * 64 + 16 * s + 4 * (l >> 5);
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(g), q1 = *reinterpret_cast<const f32x4*>(g + 8);
            bf16x8 wp[NS];
            split8<NS>(q0, q1, wp);
#pragma unroll
            for (int p = 0; p < NS; ++p) *reinterpret_cast<bf16x8*>(smem + L.off_w + p * 36 * 1024 + i * 16) = wp[p];
        }
        if (tid < 32) reinterpret_cast<float*>(smem + L.off_bias)[tid] = a.b2[br * C2 + 8 * ((tid & 15) >> 2) + 4 * (tid >> 4) + (tid & 3)];
        __syncthreads();
        // conv1 weights: K slots of k-group 0 [w00 w01 w02 0 w10 w11 w12 0], of k-group 1 [w20 w21 w22 0 bias bias' bias'' 0]
        // (the pixel side of the bias slots is 1.0: three bf16 pieces in part 0 = the float32 bias; the other parts hold zeros there)
        bf16x8 w1p[2][NS];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float* wr_ = a.w1 + ((size_t)br * C1 + ct * 32 + n) * KTAPS;
            const float bias = a.b1[br * C1 + ct * 32 + n];
            uint32_t q[4][NS];
            if (h == 0) {
                split2<NS>(wr_[0], wr_[1], q[0]); split2<NS>(wr_[2], 0.f, q[1]);
                split2<NS>(wr_[3], wr_[4], q[2]); split2<NS>(wr_[5], 0.f, q[3]);
            } else {
                split2<NS>(wr_[6], wr_[7], q[0]); split2<NS>(wr_[8], 0.f, q[1]);
                const float bh = __uint_as_float(pk_bf16(bias, 0.f) << 16);
                const float r1 = bias - bh;
                const float bm = __uint_as_float(pk_bf16(r1, 0.f) << 16);
#pragma unroll
                for (int p = 0; p < NS; ++p) { q[2][p] = 0; q[3][p] = 0; }
                q[2][0] = pk_bf16(bias, r1); q[3][0] = pk_bf16(r1 - bm, 0.f);
            }
#pragma unroll
            for (int p = 0; p < NS; ++p) {
                uint4 u = make_uint4(q[0][p], q[1][p], q[2][p], q[3][p]);
                w1p[ct][p] = *reinterpret_cast<bf16x8*>(&u);
            }
        }
        const unsigned char* wl = smem + L.off_w + lane * 16;
        const f32x4* biasl = reinterpret_cast<const f32x4*>(smem + L.off_bias + h * 64);

        for (int64_t grp = g0; grp < g1; ++grp) {
            __syncthreads();                    // every wave is done with the previous group's planes
            // ---- the group's planes: [0,H) x [0,W) of each LDS region (the borders stay zero)
            for (int sl = 0; sl < GROUP; ++sl) {
                const int64_t b = grp * GROUP + sl;
                if (b >= a.B) break;
                const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(a.in[br] + b * (int64_t)H * W);
                unsigned char* const plane = smem + sl * L.plane;
#pragma unroll 4
                for (int i = tid; i < nquad; i += 64 * WAVES) {
                    const int rr = i / WQ;
                    *reinterpret_cast<uint4*>(plane + rr * RSB + (i - rr * WQ) * 16) = src4[i];
                }
            }
            __syncthreads();
            for (int item = wave; item < GROUP * NT; item += WAVES) {
                const int sl = item / NT, tile = item - sl * NT;          // wave-uniform
                const int64_t b = grp * GROUP + sl;
                if (b >= a.B) break;
                const int q = tile * 32 + n;
                const bool live = q < P;
                const int pr = q / OW2, pc = q - pr * OW2;
                float* dst = a.feat + (b * (int64_t)P + q) * 96 + br * 32 + 4 * h;
                const int prc = live ? pr : OH2 - 1, pcc = live ? pc : OW2 - 1;
                const bool lastrow = prc == OH2 - 1, lastcol = pcc == OW2 - 1;
                const int base0 = (int)(sl * L.plane) + (4 * prc + 2 * h) * RSB + 16 * pcc;
                // a: window row r0 (r2 in k-group 1); b: r1 (k-group 1: the bias's 1.0, 0.0 on a padding pixel); index = ky.
                // The kx = 1 windows start 8 bytes off the 16-byte grid: two 8-byte reads through addresses the compiler cannot
                // fuse (a DS access off its natural alignment is replayed)
                int a0[3], a1[3], a2[3], b0[3], b1[3], b1b[3], b2[3];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    a0[ky] = base0 + 2 * ky * RSB;
                    a1[ky] = opaque(a0[ky] + 8);
                    a2[ky] = opaque(a0[ky] + 16);
                    const bool zr = ky == 2 && lastrow;
                    b0[ky] = h ? (zr ? off_zero : off_ones) : a0[ky] + RSB;
                    b1[ky] = opaque(h ? b0[ky] : b0[ky] + 8);
                    b1b[ky] = opaque(h ? b0[ky] + 8 : b0[ky] + 16);
                    b2[ky] = opaque(h ? ((zr || lastcol) ? off_zero : off_ones) : a0[ky] + RSB + 16);
                }
                auto gather = [&](int t, bf16x8 (&win)[NS]) {
                    const int ky = t / 3, kx = t - ky * 3;
                    f32x4 va, vb;
                    if (kx == 1) {
                        const f32x2 p0 = *reinterpret_cast<const f32x2*>(smem + a1[ky]), p1 = *reinterpret_cast<const f32x2*>(smem + a2[ky]);
                        const f32x2 q0 = *reinterpret_cast<const f32x2*>(smem + b1[ky]), q1 = *reinterpret_cast<const f32x2*>(smem + b1b[ky]);
                        va = f32x4{p0[0], p0[1], p1[0], p1[1]};
                        vb = f32x4{q0[0], q0[1], q1[0], q1[1]};
                    } else {
                        va = *reinterpret_cast<const f32x4*>(smem + (kx == 2 ? a2[ky] : a0[ky]));
                        vb = *reinterpret_cast<const f32x4*>(smem + (kx == 2 ? b2[ky] : b0[ky]));
                    }
                    split8<NS>(va, vb, win);
                };
                auto conv1 = [&](int ct, const bf16x8 (&x)[NS]) -> f32x16 {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    return mfma_parts<NS>(w1p[ct], x, z);
                };
                // relu in float32, then the split: registers 0..7 / 8..15 of a conv1 tile are the B operands of two conv2 k-steps
                auto cvt = [&](const f32x16& c, bf16x8 (&s0)[NS], bf16x8 (&s1)[NS]) {
                    f32x4 q4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) q4[j][r] = fmaxf(c[4 * j + r], 0.f);
                    split8<NS>(q4[0], q4[1], s0);
                    split8<NS>(q4[2], q4[3], s1);
                };
                auto wread = [&](int t, bf16x8 (&w)[4][NS]) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int p = 0; p < NS; ++p) w[s][p] = *reinterpret_cast<const bf16x8*>(wl + p * 36 * 1024 + (t * 4 + s) * 1024);
                };
                // ---- 9 taps, software-pipelined as in k_dnn_trunk_rf: round t issues the weight reads of tap t, the conv1 MFMAs of tap
                //      t+2 (the reads land under them), the window reads of tap t+3, the conv2 MFMAs of tap t and the relu / split of
                //      tap t+1.  The weights are single-buffered: 256 registers per wave at two waves per SIMD
                bf16x8 win[3][NS];
                gather(0, win[0]); gather(1, win[1]); gather(2, win[2]);
                bf16x8 wq[4][NS];
                f32x16 c1[2][2];
                c1[0][0] = conv1(0, win[0]); c1[0][1] = conv1(1, win[0]);
                c1[1][0] = conv1(0, win[1]); c1[1][1] = conv1(1, win[1]);
                f32x16 acc0, acc1;
                {
                    const f32x4 q0 = biasl[0], q1 = biasl[1], q2 = biasl[2], q3 = biasl[3];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { acc0[r] = q0[r]; acc0[4 + r] = q1[r]; acc0[8 + r] = q2[r]; acc0[12 + r] = q3[r]; acc1[r] = 0.f; acc1[4 + r] = 0.f; acc1[8 + r] = 0.f; acc1[12 + r] = 0.f; }
                }
                bf16x8 pp[2][4][NS];
                cvt(c1[0][0], pp[0][0], pp[0][1]);
                cvt(c1[0][1], pp[0][2], pp[0][3]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<9>([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int k = t & 1;
                    wread(t, wq);
                    if constexpr (t + 2 < 9) { c1[k][0] = conv1(0, win[(t + 2) % 3]); c1[k][1] = conv1(1, win[(t + 2) % 3]); }
                    if constexpr (t + 3 < 9) gather(t + 3, win[t % 3]);
                    // two accumulators: an MFMA never waits for the one issued just before it (the compiler interleaves the two chains)
                    acc0 = mfma_parts<NS>(wq[0], pp[k][0], acc0);
                    acc1 = mfma_parts<NS>(wq[1], pp[k][1], acc1);
                    acc0 = mfma_parts<NS>(wq[2], pp[k][2], acc0);
                    acc1 = mfma_parts<NS>(wq[3], pp[k][3], acc1);
                    if constexpr (t + 1 < 9) {
                        cvt(c1[k ^ 1][0], pp[k ^ 1][0], pp[k ^ 1][1]);
                        cvt(c1[k ^ 1][1], pp[k ^ 1][2], pp[k ^ 1][3]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // ---- relu; a lane holds channels 8 j + 4 h + (0..3) of its pixel: four 16-byte stores, the lane pair (n, n + 32)
                //      fills 32 contiguous bytes
                if (live) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = fmaxf(acc0[4 * j + r] + acc1[4 * j + r], 0.f);
                        *reinterpret_cast<f32x4*>(dst + 8 * j) = o;
                    }
                }
            }
        }
    }
}

// x3: two parts, eight waves on a group of three samples (158 KB of LDS at 80 x 80); x6: three parts, four waves on one sample (137 KB)
constexpr int X3_NS = 2, X3_WAVES = 8, X3_GROUP = 3;
constexpr int X6_NS = 3, X6_WAVES = 4, X6_GROUP = 1;

template <int NS, int WAVES, int GROUP>
int launch_xn(rml_ctx* ctx, const X3Args& a, hipStream_t st) {
    const XnLayout<NS, GROUP> L(a.H, a.W);
    RML_MAX_DYN_LDS(160 * 1024, &k_dnn_trunk_xn<NS, WAVES, GROUP>);
    const int64_t need = 3 * ((a.B + GROUP - 1) / GROUP);
    hipLaunchKernelGGL((k_dnn_trunk_xn<NS, WAVES, GROUP>), dim3((unsigned)(need < ctx->num_cu ? need : ctx->num_cu)), dim3(64 * WAVES), L.total, st, a);
    return RML_OK;
}

}  // namespace

extern "C" int rml_dnn_trunk_x3_supported(int H, int W) {
    if (H <= 0 || W <= 0 || H % 4 || W % 4) return 0;
    return XnLayout<X3_NS, X3_GROUP>(H, W).total <= 160 * 1024 && XnLayout<X6_NS, X6_GROUP>(H, W).total <= 160 * 1024;
}

extern "C" int rml_dnn_trunk_x3(rml_ctx* ctx, const float* xz, const float* yz, const float* xy, int64_t B, int H, int W,
                                const float* w1, const float* b1, const float* w2, const float* b2, int parts, float* feat, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && H > 0 && W > 0 && (parts == 2 || parts == 3), RML_ERR_INVALID, "rml_dnn_trunk_x3: bad arguments (parts: 2 or 3)");
    if (B == 0) return RML_OK;
    RML_REQUIRE(xz && yz && xy && w1 && b1 && w2 && b2 && feat, RML_ERR_INVALID, "rml_dnn_trunk_x3: NULL argument");
    RML_REQUIRE(rml_dnn_trunk_x3_supported(H, W), RML_ERR_UNSUPPORTED,
                "rml_dnn_trunk_x3: H and W must be multiples of 4 and three float32 planes + 72 KB of weights must fit the LDS (got %dx%d)", H, W);
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_dnn_trunk_x3: B too large");
    RML_REQUIRE(((reinterpret_cast<uintptr_t>(xz) | reinterpret_cast<uintptr_t>(yz) | reinterpret_cast<uintptr_t>(xy) |
                  reinterpret_cast<uintptr_t>(w2) | reinterpret_cast<uintptr_t>(feat)) & 15) == 0, RML_ERR_INVALID,
                "rml_dnn_trunk_x3: planes, w2 and feat must be 16-byte aligned");
    RML_HIP(hipSetDevice(ctx->device));
    X3Args a{};
    a.in[0] = xz; a.in[1] = yz; a.in[2] = xy; a.B = B; a.H = H; a.W = W;
    a.w1 = w1; a.b1 = b1; a.w2 = w2; a.b2 = b2; a.feat = feat;
    if (parts == 2) launch_xn<X3_NS, X3_WAVES, X3_GROUP>(ctx, a, static_cast<hipStream_t>(stream));
    else launch_xn<X6_NS, X6_WAVES, X6_GROUP>(ctx, a, static_cast<hipStream_t>(stream));
    RML_HIP(hipGetLastError());
    return RML_OK;
}

// ---- volumes -> float32-class features in one call: the margin guard's re-scoring front (radar-ml_amd/dnn.py rescore_exact) ----------
namespace {
template <typename T>
__global__ __launch_bounds__(256) void k_gather_frames(const T* __restrict__ V, const int64_t* __restrict__ rows, int64_t units, T* __restrict__ out) {
    const T* __restrict__ src = V + rows[blockIdx.x] * units;
    T* __restrict__ dst = out + (int64_t)blockIdx.x * units;
    for (int64_t u = (int64_t)blockIdx.y * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.y * 256) dst[u] = src[u];
}
inline int64_t up256(int64_t v) { return (v + 255) & ~(int64_t)255; }
}  // namespace

extern "C" int64_t rml_dnn_exact_features_scratch_bytes(int vdtype, int64_t n, int X, int Y, int Z, int out_h, int out_w, int gathered) {
    if (n <= 0 || X <= 0 || Y <= 0 || Z <= 0 || out_h <= 0 || out_w <= 0) return 0;
    const int64_t frame = (int64_t)X * Y * Z * (vdtype == RML_VOL_U8 ? 1 : 4);
    const int64_t D = (int64_t)X * Z + (int64_t)Y * Z + (int64_t)X * Y;
    return (gathered ? up256(n * frame) : 0) + up256(n * D * 4) + 3 * up256(n * (int64_t)out_h * out_w * 4);
}

extern "C" int rml_dnn_exact_features(rml_ctx* ctx, const void* V, int vdtype, const int64_t* rows, int64_t n, int X, int Y, int Z, int mode,
                                      int out_h, int out_w, const float* w1, const float* b1, const float* w2, const float* b2, int parts,
                                      void* scratch, int64_t scratch_bytes, float* feat, void* stream) {
    RML_REQUIRE(ctx && n >= 0 && X > 0 && Y > 0 && Z > 0, RML_ERR_INVALID, "rml_dnn_exact_features: bad arguments");
    if (n == 0) return RML_OK;
    RML_REQUIRE(V && scratch && feat, RML_ERR_INVALID, "rml_dnn_exact_features: NULL argument");
    RML_REQUIRE(vdtype == RML_VOL_F32 || vdtype == RML_VOL_U8, RML_ERR_INVALID, "rml_dnn_exact_features: unknown volume dtype %d", vdtype);
    RML_REQUIRE(mode != RML_MODE_SLICE, RML_ERR_UNSUPPORTED, "rml_dnn_exact_features: slice projections need (i, j, k) per frame: use rml_project + rml_resize_bicubic + rml_dnn_trunk_x3");
    RML_REQUIRE(n < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_dnn_exact_features: n too large");
    RML_REQUIRE(scratch_bytes >= rml_dnn_exact_features_scratch_bytes(vdtype, n, X, Y, Z, out_h, out_w, rows != nullptr) &&
                (reinterpret_cast<uintptr_t>(scratch) & 255) == 0, RML_ERR_INVALID,
                "rml_dnn_exact_features: scratch of rml_dnn_exact_features_scratch_bytes() bytes, 256-byte aligned, expected");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t frame = (int64_t)X * Y * Z * (vdtype == RML_VOL_U8 ? 1 : 4);
    const int64_t D = (int64_t)X * Z + (int64_t)Y * Z + (int64_t)X * Y;
    unsigned char* sp = static_cast<unsigned char*>(scratch);
    const void* vol = V;
    if (rows) {
        // the wanted frames, contiguous (the projection kernels stream whole batches)
        const bool q16 = frame % 16 == 0 && (reinterpret_cast<uintptr_t>(V) & 15) == 0, q4 = frame % 4 == 0 && (reinterpret_cast<uintptr_t>(V) & 3) == 0;
        const int64_t units = q16 ? frame / 16 : (q4 ? frame / 4 : frame);
        const unsigned ny = (unsigned)std::min<int64_t>(32, (units + 255) / 256);
        if (q16) hipLaunchKernelGGL(k_gather_frames<uint4>, dim3((unsigned)n, ny), dim3(256), 0, st, static_cast<const uint4*>(V), rows, units, reinterpret_cast<uint4*>(sp));
        else if (q4) hipLaunchKernelGGL(k_gather_frames<uint32_t>, dim3((unsigned)n, ny), dim3(256), 0, st, static_cast<const uint32_t*>(V), rows, units, reinterpret_cast<uint32_t*>(sp));
        else hipLaunchKernelGGL(k_gather_frames<uint8_t>, dim3((unsigned)n, ny), dim3(256), 0, st, static_cast<const uint8_t*>(V), rows, units, reinterpret_cast<uint8_t*>(sp));
        RML_HIP(hipGetLastError());
        vol = sp;
        sp += up256(n * frame);
    }
    float* rowsf = reinterpret_cast<float*>(sp);
    sp += up256(n * D * 4);
    int rc = rml_project(ctx, vol, vdtype, n, X, Y, Z, mode, nullptr, 0.0f, RML_MASK_ALL, rowsf, D, nullptr, 0, nullptr, nullptr, nullptr, stream);
    if (rc) return rc;
    const int ph[3] = {X, Y, X}, pw[3] = {Z, Z, Y};
    float* planes[3];
    int64_t off = 0;
    for (int pl = 0; pl < 3; ++pl) {
        planes[pl] = reinterpret_cast<float*>(sp);
        sp += up256(n * (int64_t)out_h * out_w * 4);
        rc = rml_resize_bicubic(ctx, rowsf + off, D, n, ph[pl], pw[pl], out_h, out_w, 127.5f, 127.5f, planes[pl], 0, stream);
        if (rc) return rc;
        off += (int64_t)ph[pl] * pw[pl];
    }
    return rml_dnn_trunk_x3(ctx, planes[0], planes[1], planes[2], n, out_h, out_w, w1, b1, w2, b2, parts, feat, stream);
}
