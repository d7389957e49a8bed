// Fused convolutional trunk of the reference's multi-view CNN (dnn.py:45-52, 68-76) for gfx950:
// per projection branch  Conv2D(1->64, 3x3, stride 2, 'same', relu) -> Conv2D(64->32, 3x3, stride 2, 'same', relu),
// the three branches concatenated on the channel axis and flattened in NHWC order (dnn.py:74-76) -- i.e. the
// 38 400-long feature vector that feeds the dense layers.
//
// PyTorch/MIOpen spends ~60 % of this forward in separate pad / bias / relu / cast passes over the 40x40x64
// intermediate (614 KB per sample in bf16).  Here the intermediate never leaves the CU.  Workgroups are persistent
// (two per CU, grid = resident slots): a workgroup owns one branch -- its weight fragments stay in registers -- and
// walks that branch's samples; while one sample is convolved the next plane is already in flight into registers.
// Per sample the conv2 output is produced in strips of SR rows:
//   1. the plane is written to LDS as bf16 (rows >= H and columns >= W are TensorFlow's bottom/right 'same' zeros),
//   2. conv1 + bias + relu on the matrix cores (v_mfma_f32_32x32x16_bf16), transposed: M = 64 channels (the weights = A), N = 32 strip pixels
//      per tile (B = the pixels' 3x3 windows, five LDS reads per lane: the K order is chosen so that row pairs of the
//      window are single aligned 32-bit reads, and K slots with zero weights may hold anything finite), K = 9 taps +
//      1 bias slot (pixel side forced to 1.0) padded to 16.  relu is one v_pk_max_i16 on the packed bf16 pair, and a
//      lane holds 4 consecutive channels of one pixel: one 8-byte LDS store into the conv1 image [rows][cols+1][64],
//   3. conv2 is an implicit GEMM (v_mfma_f32_16x16x32_bf16): M = 32 output channels (weights, in registers),
//      N = strip pixels, K = 9 taps x 64 channels; the pixel fragments are read straight from the conv1 image (16
//      contiguous bytes = 8 input channels of one tap); the K range is split over two wave pairs and reduced
//      through LDS, each half finishing part of the pixel tiles,
//   4. relu, bf16, and 8-byte stores straight from the accumulator layout to ((h*OW2 + w)*96 + branch*32 + n).
// The conv1 image uses a padded pixel stride (144 B) so that the ds_read_b128 fragment reads are bank-conflict
// free.  Barriers are
// LDS-only (s_waitcnt lgkmcnt(0) + s_barrier) so that they do not drain the plane prefetch.
// Numerics: bf16 operands, float32 accumulation (what torch.autocast(bf16) does), outputs bf16; the conv1 bias rides in the K
// slots too: one bf16 slot in k_dnn_trunk, three (= float32 precision) in k_dnn_trunk_rf since round 5.
// Measured (8192 samples of 3 x 80x80, MI355X): 0.68 ms = 12.1 M samples/s; the same layers through PyTorch/MIOpen
// in bf16 channels_last take 13.9 ms.  Planes whose LDS layout fits go to k_dnn_trunk_rf further down (0.55-0.61 ms).
#include "rml_internal.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

constexpr int C1 = 64, C2 = 32, KTAPS = 9;
constexpr int K2 = KTAPS * C1;            // 576
constexpr int PIX_STRIDE = C1 * 2 + 16;    // 144 B per conv1 pixel in LDS
constexpr int MT_MAX = 5;                  // conv2 pixel tiles per strip (template MT <= MT_MAX)
constexpr int TPW = 3;                     // conv1 pixel tiles (32 pixels) per wave and strip
constexpr int PLD = 7;                     // 16-byte loads of the input plane per thread that are prefetched

struct TrunkArgs {
    const void* in[3];      // (B, H, W) per branch: float32, or bf16 (template INBF)
    int64_t B;
    int H, W;
    const float* w1;        // [3][64][9]
    const float* b1;        // [3][64]
    const uint16_t* w2t;    // [3][32][576] bf16, k = (ky*3+kx)*64 + cin
    const float* b2;        // [3][32]
    uint16_t* feat;         // [B][OH2*OW2*96] bf16
    int kblock;             // k_dnn_trunk_rf: feat is [K/64][B][64] with K ordered (branch, pixel, channel) -- see rml_dnn_trunk_kblock
    int split;              // k_dnn_trunk_rf, few samples: a WORKGROUP owns a sample and its eight waves take the sample's tiles eight apart
};

__device__ __forceinline__ uint32_t pk_bf16(float lo, float hi) {     // v_cvt_pk_bf16_f32 (round to nearest even)
    bf16x2 b = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
    return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ uint32_t pk_relu(uint32_t v) {             // bf16 is sign-magnitude: max as int16 with 0
    s16x2 s = *reinterpret_cast<s16x2*>(&v);
    s = __builtin_elementwise_max(s, s16x2{0, 0});
    return *reinterpret_cast<uint32_t*>(&s);
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>)
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the prefetch of the next plane
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct TrunkLayout {        // LDS carve-up, shared by the kernel and the launcher
    int RS, nt1;
    size_t off_c1, off_red, total;
    __host__ __device__ TrunkLayout(int H, int W, int SR, int MT) {
        const int R1 = 2 * SR + 1, OW1 = W / 2;
        RS = W + 2;
        nt1 = (R1 * OW1 + 31) >> 5;          // 32-pixel tiles of a strip's conv1 image
        off_c1 = ((size_t)(H + 5) * RS * 2 + 15) & ~(size_t)15;
        const size_t image = ((size_t)R1 * (OW1 + 1) + 1) * PIX_STRIDE;
        off_red = off_c1 + ((image + 15) & ~(size_t)15);
        total = off_red + (size_t)2 * MT * 64 * 16;
    }
};

// MT = conv2 pixel tiles per strip, PF = prefetch the next plane into registers, WPC = workgroups per CU the register
// budget is compiled for (2: 256 VGPRs, 3: 168)
template <int SR, bool INBF, int MT, bool PF, int WPC>
__global__ __launch_bounds__(256, WPC) void k_dnn_trunk(TrunkArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = a.H, W = a.W;
    const int OH1 = H / 2, OW1 = W / 2, OH2 = H / 4, OW2 = W / 4;
    constexpr int R1 = 2 * SR + 1;        // conv1 rows per strip
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int br = blockIdx.y;
    const TrunkLayout L(H, W, SR, MT);
    const int RS = L.RS;

    uint16_t* in_s = reinterpret_cast<uint16_t*>(smem);          // [H+5][W+2] bf16: plane + zero pad
    unsigned char* c1_s = smem + L.off_c1;                       // [R1][OW1+1] pixels x 144 B (+ 1 dummy pixel)
    float* red_s = reinterpret_cast<float*>(smem + L.off_red);   // [2 nt][MT_MAX][64 lanes][4]: K-split partials
    const int dummy_off = R1 * (OW1 + 1) * PIX_STRIDE;

    const float* __restrict__ w1 = a.w1 + br * C1 * KTAPS;
    const float* __restrict__ b1 = a.b1 + br * C1;
    const float* __restrict__ b2 = a.b2 + br * C2;

    // everything the window reads may touch -> 0 once: the planes only ever overwrite [0,H) x [0,W); the conv1 image
    // too (its pad column is never written)
    {
        uint4 z = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < (int)(L.off_red >> 4); i += 256) *reinterpret_cast<uint4*>(smem + i * 16) = z;
    }
    lds_barrier();
    // the workgroup is persistent: it walks samples blockIdx.x, +gridDim.x, ... of its branch, and while one sample
    // is being convolved the next plane is already in flight into registers (PLD float4 per thread)
    // (a 16-byte quad = 4 float32 or 8 bf16 values of one row)
    constexpr int QE = INBF ? 8 : 4;        // elements per quad
    constexpr int NLD = INBF ? (PLD + 1) / 2 : PLD;
    const int WQ = W / QE, nquad = H * WQ;
    const bool prefetch = PF && nquad <= 256 * NLD;
    int lofs[NLD];                          // LDS destination (bf16 units) of the thread's quads, -1 past the plane
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int i = u * 256 + tid, rr = i / WQ;
        lofs[u] = i < nquad ? rr * RS + (i - rr * WQ) * QE : -1;
    }
    uint4 pv[NLD];
    auto issue = [&](int64_t bb) {
        const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(
            static_cast<const unsigned char*>(a.in[br]) + bb * (int64_t)H * W * (INBF ? 2 : 4));
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int i = u * 256 + tid;
            pv[u] = src4[i < nquad ? i : nquad - 1];
        }
    };
    auto put = [&](int lo, const uint4& v) {        // one quad into the bf16 plane (rows are only 4-byte aligned)
        uint32_t* d = reinterpret_cast<uint32_t*>(in_s + lo);
        if (INBF) {
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
            d[0] = pk_bf16(__uint_as_float(v.x), __uint_as_float(v.y));
            d[1] = pk_bf16(__uint_as_float(v.z), __uint_as_float(v.w));
        }
    };
    if (prefetch) issue(blockIdx.x);
    // conv2 weights: a wave only ever needs the fragments of its (K half, channel tile): 9 x 16 B per lane, in
    // registers for the whole workgroup lifetime (36 VGPRs instead of a 37 KB LDS image)
    const int nt = wave & 1, kh = wave >> 1, kg = lane >> 4;
    bf16x8 bfrag[9];
    {
        const unsigned char* g = reinterpret_cast<const unsigned char*>(a.w2t + (size_t)br * C2 * K2) +
                                 (size_t)(nt * 16 + (lane & 15)) * (K2 * 2) + kg * 16;
#pragma unroll
        for (int t = 0; t < 9; ++t) bfrag[t] = *reinterpret_cast<const bf16x8*>(g + (kh * 9 + t) * 64);
    }
    // conv1 runs as v_mfma_f32_32x32x16_bf16: M = 32 channels (two A fragments = the weights), N = 32 strip pixels per
    // tile, K = 16 = 9 taps + 1 bias slot + 6 zeros -- half the matrix-core time of a K = 32 instruction, and both
    // k-groups of lanes (lane / 32) carry window data.  K slots of k-group 0: the window's (r0c0 r0c1)(r1c0 r1c1)
    // (r2c0 r2c1)(r0c2 r1c2); k-group 1: r2c2, then the bias (its pixel-side slot is forced to 1.0), then zeros.
    const int l32 = lane & 31, kg1 = lane >> 5;
    bf16x8 w1frag[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const float* wr_ = w1 + (ct * 32 + l32) * KTAPS;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (kg1 == 0) u = make_uint4(pk_bf16(wr_[0], wr_[1]), pk_bf16(wr_[3], wr_[4]), pk_bf16(wr_[6], wr_[7]), pk_bf16(wr_[2], wr_[5]));
        else u.x = pk_bf16(wr_[8], b1[ct * 32 + l32]);
        w1frag[ct] = *reinterpret_cast<bf16x8*>(&u);
    }
    const uint32_t one_mask = kg1 == 1 ? 0xFFFF0000u : 0u;     // k-group 1 lanes: high half of dword 0 := bf16(1.0)
    f32x4 b2r;
#pragma unroll
    for (int r = 0; r < 4; ++r) b2r[r] = kh == 0 ? b2[nt * 16 + kg * 4 + r] : 0.0f;
    // conv1 tiles of this wave (tile = wave + 4j, the same in every strip): per lane the window base in the input
    // plane (bf16 units) and the byte offset in the conv1 image (+ the conv1 row in the top byte); pixels past the
    // strip recompute its last pixel into the dummy slot
    int tin[TPW], tout[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int q = (wave + 4 * j) * 32 + l32;
        const int qq = q < R1 * OW1 ? q : R1 * OW1 - 1;
        const int cr = qq / OW1, cc = qq - cr * OW1;
        tin[j] = (2 * cr) * RS + 2 * cc + (kg1 == 1 ? 2 * RS + 2 : 0);
        tout[j] = (((q < R1 * OW1) ? (cr * (OW1 + 1) + cc) * PIX_STRIDE : dummy_off) + kg1 * 8) | (cr << 24);
    }

    const int P = SR * OW2;                 // output pixels per strip
    constexpr int MS = (MT + 1) / 2;        // pixel tiles finished by the kh = 0 waves
    int poff[MT];                       // conv2: byte offset of the lane's pixel window in the conv1 image
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        int q = m * 16 + (lane & 15);
        q = q < P ? q : P - 1;
        const int pr = q / OW2, pc = q - pr * OW2;
        poff[m] = ((2 * pr) * (OW1 + 1) + 2 * pc) * PIX_STRIDE + kg * 16;
    }
    const int NT1 = L.nt1;

    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
    // 1. the plane as bf16 into LDS (every wave is past the conv1 of the previous sample: barriers in between)
    if (prefetch) {
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            if (lofs[u] >= 0) put(lofs[u], pv[u]);
        const int64_t nb = b + gridDim.x;
        issue(nb < a.B ? nb : b);           // (the last round reloads its own plane and drops it)
    } else {
        const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(
            static_cast<const unsigned char*>(a.in[br]) + b * (int64_t)H * W * (INBF ? 2 : 4));
        for (int i = tid; i < nquad; i += 256) {
            const int rr = i / WQ;
            put(rr * RS + (i - rr * WQ) * QE, src4[i]);
        }
    }
    for (int r0 = 0; r0 < OH2; r0 += SR) {
        lds_barrier();                      // previous strip stored, the plane is in place
        // 2. conv1 + bias + relu -> bf16 image.  conv1 rows past the bottom edge are 'same' zeros: their pixels go to
        //    the dummy slot and the rows are cleared here (only ever in the last strip).
        const int live = OH1 - 2 * r0;      // conv1 rows of this strip that exist
        if (live < R1) {
            uint4 z = make_uint4(0, 0, 0, 0);
            const int lo = (live < 0 ? 0 : live) * (OW1 + 1) * (PIX_STRIDE / 16);
            for (int i = lo + tid; i < R1 * (OW1 + 1) * (PIX_STRIDE / 16); i += 256) *reinterpret_cast<uint4*>(c1_s + i * 16) = z;
        }
        const uint16_t* plane0 = in_s + (4 * r0) * RS;
        {
            // the wave's (up to TPW) tiles: all window reads in flight before the first MFMA
            uint4 u[TPW];
            int to[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const uint16_t* xin = plane0 + tin[i];
                u[i].x = (*reinterpret_cast<const uint32_t*>(xin) & ~one_mask) | (0x3F800000u & one_mask);
                u[i].y = *reinterpret_cast<const uint32_t*>(xin + RS);
                u[i].z = *reinterpret_cast<const uint32_t*>(xin + 2 * RS);
                u[i].w = (uint32_t)xin[2] | ((uint32_t)xin[RS + 2] << 16);
                to[i] = (tout[i] >> 24) < live ? (tout[i] & 0xFFFFFF) : dummy_off + kg1 * 8;
            }
            __builtin_amdgcn_sched_barrier(0);
            // C/D map of the 32x32 tile: column (pixel) lane & 31, rows (channels) ct*32 + 8*(r/4) + 4*(lane/32) + r%4:
            // every four accumulator registers are 4 consecutive channels of the lane's pixel = one 8-byte store
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (wave + 4 * i < NT1) {
                    const bf16x8 xfrag = *reinterpret_cast<bf16x8*>(&u[i]);
                    f32x16 c[2];
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        c[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1frag[ct], xfrag, z, 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            *reinterpret_cast<uint2*>(c1_s + to[i] + ct * 64 + j * 16) =
                                make_uint2(pk_relu(pk_bf16(c[ct][4 * j], c[ct][4 * j + 1])), pk_relu(pk_bf16(c[ct][4 * j + 2], c[ct][4 * j + 3])));
                }
            }
        }
        lds_barrier();
        // 3. conv2 as implicit GEMM: wave = (K half kh, channel tile nt), all MT pixel tiles (tiles past the strip
        //    recompute its last pixel and are dropped in the epilogue: no divergent loads in the MFMA loop)
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = b2r;
        {
            // software pipeline over the wave's 9 K-steps: the fragment reads of step t+2 are issued before the MFMAs
            // of step t, so an LDS round trip is always covered by ten MFMAs
            bf16x8 afrag[3][MT];
            auto ldk = [&](int tt, bf16x8 (&dst)[MT]) {
                const int ks = kh * 9 + tt;
                const int tap = ks >> 1, ky = tap / 3, kx = tap - ky * 3;
                const int aoff = (ky * (OW1 + 1) + kx) * PIX_STRIDE + (ks & 1) * 64;
#pragma unroll
                for (int m = 0; m < MT; ++m) dst[m] = *reinterpret_cast<const bf16x8*>(c1_s + poff[m] + aoff);
            };
            ldk(0, afrag[0]);
            ldk(1, afrag[1]);
            __builtin_amdgcn_s_setprio(2);          // the MFMA phase goes first on its SIMD; the other workgroup's wave fills the gaps
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) {
                if (tt + 2 < 9) ldk(tt + 2, afrag[(tt + 2) % 3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag[tt], afrag[tt % 3][m], acc[m], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
        // K-split reduction, both halves busy: the kh=0 wave finishes pixel tiles 0..MS-1, the kh=1 wave the rest;
        // each hands the other its partials of the tiles it does not finish
        if (kh == 0) {
#pragma unroll
            for (int m = MS; m < MT; ++m) *reinterpret_cast<f32x4*>(red_s + ((nt * MT + m) * 64 + lane) * 4) = acc[m];
        } else {
#pragma unroll
            for (int m = 0; m < MS; ++m) *reinterpret_cast<f32x4*>(red_s + ((nt * MT + m) * 64 + lane) * 4) = acc[m];
        }
        lds_barrier();                    // partials visible
        // C/D map: rows (channels) nt*16 + (lane>>4)*4 + r, column (pixel) lane&15: a lane holds 4 consecutive channels
        // of one pixel -> one 8-byte store straight to ((h*OW2 + w)*96 + br*32 + channel); the four lanes of a pixel
        // cover 32 contiguous bytes and the partner wave (other nt) the other half of the branch's 64
        const int rows = (OH2 - r0) < SR ? (OH2 - r0) : SR;
        const int Pv = rows * OW2;
        uint16_t* dst0 = a.feat + b * (int64_t)OH2 * OW2 * 96 + (int64_t)r0 * OW2 * 96 + br * 32 + nt * 16 + kg * 4;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if ((m < MS) == (kh == 0)) {
                const int q = m * 16 + (lane & 15);
                const f32x4 o = *reinterpret_cast<const f32x4*>(red_s + ((nt * MT + m) * 64 + lane) * 4);
                const f32x4 s4 = acc[m] + o;
                if (q < Pv)
                    *reinterpret_cast<uint2*>(dst0 + q * 96) = make_uint2(pk_relu(pk_bf16(s4[0], s4[1])), pk_relu(pk_bf16(s4[2], s4[3])));
            }
        }
    }
    }
}

template <int SR, bool INBF, int MT, bool PF, int WPC>
int launch_trunk(const TrunkArgs& a, int num_cu, hipStream_t stream) {
    const TrunkLayout L(a.H, a.W, SR, MT);
    if (SR * (a.W / 4) > 16 * MT || L.nt1 > 4 * TPW || L.total > 150 * 1024) return RML_ERR_UNSUPPORTED;
    if (L.total * WPC > 160 * 1024) return RML_ERR_UNSUPPORTED;
    RML_MAX_DYN_LDS(160 * 1024, &k_dnn_trunk<SR, INBF, MT, PF, WPC>);
    const int64_t slots = (int64_t)WPC * num_cu;                                     // workgroups resident at once
    const int64_t gx = slots / 3 > 0 ? slots / 3 : 1;
    hipLaunchKernelGGL((k_dnn_trunk<SR, INBF, MT, PF, WPC>), dim3((unsigned)(a.B < gx ? a.B : gx), 3), dim3(256), L.total, stream, a);
    return RML_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// k_dnn_trunk_rf: the same two convolutions with the conv1 image kept in REGISTERS.
//
// k_dnn_trunk above is bound by the LDS, not by the matrix cores: every conv1 pixel is stored to LDS once and read back
// 2.25 times (3x3 taps at stride 2) by two waves each (one per 16-channel tile), 1 KB of ds_read_b128 per 16-cycle MFMA.
// Here the conv1 image never exists.  With v_mfma_f32_32x32x16_bf16 the conv1 accumulators of a 32-pixel tile already
// ARE the B operand of conv2 for that tile: a lane (pixel n = lane & 31, k-group h = lane / 32) holds, for register r,
// channel 8 (r / 4) + 4 h + r % 4 of pixel n, and conv2's B operand wants 8 k-values of column n per lane -- any K
// order will do as long as the weight fragments use the same one.  So for every (tile of 32 conv2 pixels, tap) conv1 is
// evaluated on the matrix cores AT the 32 conv1 pixels that tap reads (conv1 is recomputed 2.25 times: +25 % MFMA work
// in total), converted to bf16 and relu'd in registers (v_cvt_pk_bf16_f32 + v_pk_max_i16) and fed straight to the four
// conv2 MFMAs of the tap (K = 16 conv1 channels each, all 32 output channels).
//
// One 8-wave workgroup per CU, persistent.  A WAVE owns a sample: its bf16 plane sits in a wave-private LDS region (two
// 8-byte reads per tap and lane: window rows r0 r1, or r2 + the bias's 1.0; no VALU in the gather), it walks the sample's
// tiles with no barrier and no exchange with other waves, and stores finished 32-byte channel runs.  The conv2 weight
// fragments (36 KB per branch, already in MFMA operand order) are shared by the eight waves through LDS: one
// ds_read_b128 per conv2 MFMA, a quarter of the LDS read rate.  Branches are processed one after the other (three
// weight loads per workgroup and launch).
// Conv1 pixels that are conv2's 'same' padding (row OH1 / column OW1) come out as exact zeros: their window lies in the
// plane's zero border and their bias slot reads 0.0 instead of 1.0.  K slots of a conv1 window (8 per lane): k-group 0
// [r0c0 r0c1 r0c2 r0c3 r1c0 r1c1 r1c2 r1c3] with weights [w00 w01 w02 0 w10 w11 w12 0]; k-group 1 [r2c0 r2c1 r2c2 r2c3
// 1 1 1 1] with [w20 w21 w22 0 bias 0 0 0] (slots with zero weights hold plane values: finite inputs assumed, as above).
// A DS access wider than 4 bytes must be naturally aligned or it is replayed at 64 cycles (MI355X_MICROARCH.md, LDS):
// the windows of the kx = 1 taps start 4 bytes off the 8-byte grid and are read dword by dword, through addresses the
// compiler cannot see through (it would fuse them into one 8-byte read otherwise).
constexpr int RF_WAVES = 8;
struct RfLayout {
    int RSB;                                   // bytes per plane row in LDS: (W + 4) bf16
    uint32_t plane, off_w, off_ones, off_zero, off_bias, total;
    __host__ __device__ RfLayout(int H, int W) {
        RSB = (W + 4) * 2;
        plane = ((uint32_t)(H + 4) * RSB + 15) & ~15u;
        off_w = RF_WAVES * plane;              // [tap][k-step][lane] 16 B: conv2 A operands
        off_ones = off_w + 36 * 1024;
        off_zero = off_ones + 16;
        off_bias = off_zero + 16;              // [h][16] float: conv2 bias in accumulator order
        total = off_bias + 128;
    }
};

__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

template <bool INBF>
__global__ __launch_bounds__(64 * RF_WAVES, 2) void k_dnn_trunk_rf(TrunkArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int H = a.H, W = a.W, OH2 = H / 4, OW2 = W / 4, P = OH2 * OW2;
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const RfLayout L(H, W);
    const int RSB = L.RSB;
    const int off_ones = (int)L.off_ones, off_zero = (int)L.off_zero;
    unsigned char* const plane = smem + wave * L.plane;
    for (uint32_t i = tid; i < (L.total >> 4); i += 64 * RF_WAVES) *reinterpret_cast<uint4*>(smem + i * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (tid < 4) reinterpret_cast<uint32_t*>(smem + off_ones)[tid] = 0x3F803F80u;

    constexpr int QE = INBF ? 8 : 4;            // elements per 16-byte quad of a plane row
    const int WQ = W / QE, nquad = H * WQ;
    const int NT = (P + 31) >> 5;
    const int incr = 32 / OW2, incc = 32 - incr * OW2;
    const int pr0 = n / OW2, pc0 = n - pr0 * OW2;
    const int64_t sstride = (int64_t)gridDim.x * RF_WAVES;

    for (int br = 0; br < 3; ++br) {
        __syncthreads();                        // every wave is done with the previous branch's weights
        // conv2 weights in operand order: block (tap t, k-step s = 2 ct + half), lane (cout m, k-group h) holds input
        // channels 32 ct + 16 half + 8 (i / 4) + 4 h + i % 4, i = 0..7 -- the order the conv1 accumulators come in
        for (int i = tid; i < 36 * 64; i += 64 * RF_WAVES) {
            const int blk = i >> 6, l = i & 63, t = blk >> 2, s = blk & 3;
            const uint16_t* g = a.w2t + ((size_t)br * C2 + (l & 31)) * K2 + t * 64 + 16 * s + 4 * (l >> 5);
            const uint2 lo = *reinterpret_cast<const uint2*>(g), hi = *reinterpret_cast<const uint2*>(g + 8);
            *reinterpret_cast<uint4*>(smem + L.off_w + i * 16) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        if (tid < 32) reinterpret_cast<float*>(smem + L.off_bias)[tid] = a.b2[br * C2 + 8 * ((tid & 15) >> 2) + 4 * (tid >> 4) + (tid & 3)];
        __syncthreads();
        bf16x8 w1f[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float* wr_ = a.w1 + ((size_t)br * C1 + ct * 32 + n) * KTAPS;
            const float bias = a.b1[br * C1 + ct * 32 + n];
            uint4 u;
            if (h == 0) u = make_uint4(pk_bf16(wr_[0], wr_[1]), pk_bf16(wr_[2], 0.f), pk_bf16(wr_[3], wr_[4]), pk_bf16(wr_[5], 0.f));
            else {
                // the bias takes THREE of the four slots whose pixel side is 1.0 (0.0 on a padding pixel: all four at once): bf16(bias)
                // + bf16 of what that left + bf16 of what that left = the float32 bias to 24 bits.  One bf16 slot (rounds 1-4) left an
                // error of up to 2^-9 |bias| on EVERY conv1 pixel of a channel -- a coherent offset that conv2 and the 38 400-term
                // dense layer add up instead of averaging out: it was most of the chain's probability error on trained weights
                // (session r5l: 5.6e-3 with it against 2.0e-3 for a chain whose only roundings are the random ones)
                const float bh = __uint_as_float(pk_bf16(bias, 0.f) << 16);
                const float r1 = bias - bh;
                const float bm = __uint_as_float(pk_bf16(r1, 0.f) << 16);
                u = make_uint4(pk_bf16(wr_[6], wr_[7]), pk_bf16(wr_[8], 0.f), pk_bf16(bias, r1), pk_bf16(r1 - bm, 0.f));
            }
            w1f[ct] = *reinterpret_cast<bf16x8*>(&u);
        }
        const unsigned char* wl = smem + L.off_w + lane * 16;
        const f32x4* biasl = reinterpret_cast<const f32x4*>(smem + L.off_bias + h * 64);

        // few samples (a.split; dnn.py:373-381 predicts ONE target per call): a wave that owns a whole sample needs 116 us for it while
        // 2 047 others idle -- the workgroup owns the sample instead, every wave keeps its own copy of the plane (no exchange, no
        // barrier: the kernel's structure) and walks the tiles wave, wave + 8, ...: the same arithmetic per tile, an eighth of the time
        const int64_t b_first = a.split ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * RF_WAVES + wave;
        const int64_t b_step = a.split ? (int64_t)gridDim.x : sstride;
        const int t_first = a.split ? wave : 0, t_step = a.split ? RF_WAVES : 1;
        for (int64_t b = b_first; b < a.B; b += b_step) {
            // ---- the wave's plane: [0,H) x [0,W) of its LDS region (the border stays zero)
            {
                const uint4* __restrict__ src4 = reinterpret_cast<const uint4*>(
                    static_cast<const unsigned char*>(a.in[br]) + b * (int64_t)H * W * (INBF ? 2 : 4));
#pragma unroll 4
                for (int i = lane; i < nquad; i += 64) {
                    const int rr = i / WQ;
                    const uint4 v = src4[i];
                    uint2* d = reinterpret_cast<uint2*>(plane + rr * RSB + (i - rr * WQ) * QE * 2);      // rows are 8-byte aligned
                    if (INBF) { d[0] = make_uint2(v.x, v.y); d[1] = make_uint2(v.z, v.w); }
                    else d[0] = make_uint2(pk_bf16(__uint_as_float(v.x), __uint_as_float(v.y)), pk_bf16(__uint_as_float(v.z), __uint_as_float(v.w)));
                }
            }
            // the lane's 32 bytes of a pixel's feature row: channels 16 h .. 16 h + 15 of this branch
            // (K-block layout: element (branch, pixel, channel) of sample b at block (branch * P + pixel) / 2 -- two pixels of 32
            // channels -- , i.e. the lanes (n, n + 1) x (h = 0, 1) fill one 128-byte line and a tile advances 16 blocks)
            const int64_t dstep = a.kblock ? 16 * a.B * 64 : 32 * 96;
            uint16_t* dst = (a.kblock ? a.feat + ((int64_t)((br * P + n) >> 1) * a.B + b) * 64 + (n & 1) * 32 + 16 * h
                                      : a.feat + (b * (int64_t)P + n) * 96 + br * 32 + 16 * h) + t_first * dstep;
            int pr = pr0 + t_first * incr, pc = pc0 + t_first * incc;
            pr += pc / OW2; pc -= (pc / OW2) * OW2;
            for (int tile = t_first; tile < NT; tile += t_step) {
                // ---- addresses of this tile's windows
                const bool live = tile * 32 + n < P;
                const int prc = live ? pr : OH2 - 1, pcc = live ? pc : OW2 - 1;
                const bool lastrow = prc == OH2 - 1, lastcol = pcc == OW2 - 1;
                const int base0 = (int)(wave * L.plane) + (4 * prc + 2 * h) * RSB + 8 * pcc;
                // a: window rows r0 (r2 in k-group 1); b: r1 (the bias's 1.0 / 0.0); index = ky.  The kx = 1 windows start
                // 4 bytes off the 8-byte grid: dword reads at a1, a2 and b1, b1b
                int a0[3], a1[3], a2[3], b0[3], b1[3], b1b[3], b2[3];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    a0[ky] = base0 + 2 * ky * RSB;
                    a1[ky] = opaque(a0[ky] + 4);
                    a2[ky] = opaque(a0[ky] + 8);
                    const bool zr = ky == 2 && lastrow;
                    b0[ky] = h ? (zr ? off_zero : off_ones) : a0[ky] + RSB;
                    b1[ky] = opaque(b0[ky] + 4);
                    b1b[ky] = opaque(b0[ky] + 8);
                    b2[ky] = opaque(h ? ((zr || lastcol) ? off_zero : off_ones) : a0[ky] + RSB + 8);
                }
                auto gather = [&](int t) -> bf16x8 {
                    const int ky = t / 3, kx = t - ky * 3;
                    uint4 u;
                    if (kx == 1) {
                        u.x = *reinterpret_cast<const uint32_t*>(smem + a1[ky]);
                        u.y = *reinterpret_cast<const uint32_t*>(smem + a2[ky]);
                        u.z = *reinterpret_cast<const uint32_t*>(smem + b1[ky]);
                        u.w = *reinterpret_cast<const uint32_t*>(smem + b1b[ky]);
                    } else {
                        const uint2 a_ = *reinterpret_cast<const uint2*>(smem + (kx == 2 ? a2[ky] : a0[ky]));
                        const uint2 b_ = *reinterpret_cast<const uint2*>(smem + (kx == 2 ? b2[ky] : b0[ky]));
                        u = make_uint4(a_.x, a_.y, b_.x, b_.y);
                    }
                    return *reinterpret_cast<bf16x8*>(&u);
                };
                auto conv1 = [&](int ct, const bf16x8& w) -> f32x16 {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f[ct], w, z, 0, 0, 0);
                };
                auto cvt = [&](const f32x16& c, bf16x8& p0, bf16x8& p1) {
                    uint4 u0 = make_uint4(pk_relu(pk_bf16(c[0], c[1])), pk_relu(pk_bf16(c[2], c[3])), pk_relu(pk_bf16(c[4], c[5])), pk_relu(pk_bf16(c[6], c[7])));
                    uint4 u1 = make_uint4(pk_relu(pk_bf16(c[8], c[9])), pk_relu(pk_bf16(c[10], c[11])), pk_relu(pk_bf16(c[12], c[13])), pk_relu(pk_bf16(c[14], c[15])));
                    p0 = *reinterpret_cast<bf16x8*>(&u0);
                    p1 = *reinterpret_cast<bf16x8*>(&u1);
                };
                auto wread = [&](int t, bf16x8 (&w)[4]) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) w[s] = *reinterpret_cast<const bf16x8*>(wl + (t * 4 + s) * 1024);
                };
                // ---- 9 taps, software-pipelined: round t issues the conv1 MFMAs of tap t+2, the window reads of tap t+3, the
                //      weight reads of tap t+1, the four conv2 MFMAs of tap t and the convert/relu of tap t+1: nothing issued
                //      in a round depends on anything else issued in it
                bf16x8 win[3];
                win[0] = gather(0); win[1] = gather(1); win[2] = gather(2);
                bf16x8 wq[2][4];
                wread(0, wq[0]);
                f32x16 c1[2][2];
                c1[0][0] = conv1(0, win[0]); c1[0][1] = conv1(1, win[0]);
                c1[1][0] = conv1(0, win[1]); c1[1][1] = conv1(1, win[1]);
                f32x16 acc0, acc1;
                {
                    const f32x4 q0 = biasl[0], q1 = biasl[1], q2 = biasl[2], q3 = biasl[3];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { acc0[r] = q0[r]; acc0[4 + r] = q1[r]; acc0[8 + r] = q2[r]; acc0[12 + r] = q3[r]; }
                }
                bf16x8 p[2][4];
                cvt(c1[0][0], p[0][0], p[0][1]);
                cvt(c1[0][1], p[0][2], p[0][3]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    if (t + 2 < 9) { c1[t & 1][0] = conv1(0, win[(t + 2) % 3]); c1[t & 1][1] = conv1(1, win[(t + 2) % 3]); }
                    if (t + 3 < 9) win[t % 3] = gather(t + 3);
                    if (t + 1 < 9) wread(t + 1, wq[(t + 1) & 1]);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[t & 1][0], p[t & 1][0], acc0, 0, 0, 0);
                    if (t == 0) {
                        f32x16 z;
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[0][1], p[0][1], z, 0, 0, 0);
                    } else {
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[t & 1][1], p[t & 1][1], acc1, 0, 0, 0);
                    }
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[t & 1][2], p[t & 1][2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[t & 1][3], p[t & 1][3], acc1, 0, 0, 0);
                    if (t + 1 < 9) {
                        cvt(c1[(t + 1) & 1][0], p[(t + 1) & 1][0], p[(t + 1) & 1][1]);
                        cvt(c1[(t + 1) & 1][1], p[(t + 1) & 1][2], p[(t + 1) & 1][3]);
                    }
                    // issue order: one MFMA (32 cycles of the matrix pipe = 8 issue slots), one or two LDS reads, five or six of
                    // the round's 32 convert / relu instructions, ...
                    if (t + 2 < 9) {
#pragma unroll
                        for (int g = 0; g < 2; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (t + 2 < 9) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                        else __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- relu, bf16; the lane pair (n, n + 32) regroups its 4 x (4 + 4) channels into 16 contiguous ones per lane
                uint32_t d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = pk_relu(pk_bf16(acc0[2 * j] + acc1[2 * j], acc0[2 * j + 1] + acc1[2 * j + 1]));
                const auto s00 = __builtin_amdgcn_permlane32_swap(d[0], d[4], false, false);     // channels 0-3 | 16-19 <-> 4-7 | 20-23
                const auto s01 = __builtin_amdgcn_permlane32_swap(d[1], d[5], false, false);
                const auto s10 = __builtin_amdgcn_permlane32_swap(d[2], d[6], false, false);     // channels 8-11 | 24-27 <-> 12-15 | 28-31
                const auto s11 = __builtin_amdgcn_permlane32_swap(d[3], d[7], false, false);
                if (live) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(s00[0], s01[0], s00[1], s01[1]);
                    *reinterpret_cast<uint4*>(dst + 8) = make_uint4(s10[0], s11[0], s10[1], s11[1]);
                }
                dst += t_step * dstep;
                pr += t_step * incr; pc += t_step * incc;
                pr += pc / OW2; pc -= (pc / OW2) * OW2;
            }
        }
    }
}

template <bool INBF>
int launch_trunk_rf(const TrunkArgs& a, int num_cu, hipStream_t stream) {
    const RfLayout L(a.H, a.W);
    if (L.total > 160 * 1024) return RML_ERR_UNSUPPORTED;
    RML_MAX_DYN_LDS(160 * 1024, &k_dnn_trunk_rf<INBF>);
    TrunkArgs as = a;
    as.split = a.B <= 2 * (int64_t)num_cu ? 1 : 0;      // at most two samples per CU: a workgroup per sample (see the kernel)
    const int64_t need = as.split ? a.B : (a.B + RF_WAVES - 1) / RF_WAVES;
    hipLaunchKernelGGL((k_dnn_trunk_rf<INBF>), dim3((unsigned)(need < num_cu ? need : num_cu)), dim3(64 * RF_WAVES), L.total, stream, as);
    return RML_OK;
}

}  // namespace

namespace {
template <bool INBF>
int dispatch_trunk(const TrunkArgs& a, int num_cu, hipStream_t st) {
    int rc = RML_ERR_UNSUPPORTED;
    rc = launch_trunk_rf<INBF>(a, num_cu, st);
    if (rc != RML_ERR_UNSUPPORTED || a.kblock) return rc;          // the K-block layout exists in the register-resident kernel only
    // (measured and dropped: a wave-specialised variant -- one 8-wave workgroup per CU, 4 producer waves doing conv1 and 4
    // consumer waves doing conv2 on a double-buffered conv1 image, one producer and one consumer per SIMD: 0.72 ms against
    // 0.675, archived as tools/exp/dnn_trunk_wave_specialised_kernel.hip.txt; and 2-row strips at three workgroups per CU -- 148 VGPRs, 49 KB LDS -- 0.78 ms against 0.69:
    // the extra conv1 rows, tile padding and barriers cost more than the third wave per SIMD hides)
    // strips of 4 conv2 rows while 4 rows are at most 80 pixels and two workgroups fit a CU, else 2 rows, else 1
    if (rc == RML_ERR_UNSUPPORTED) rc = launch_trunk<4, INBF, 5, true, 2>(a, num_cu, st);
    if (rc == RML_ERR_UNSUPPORTED) rc = launch_trunk<2, INBF, 5, true, 1>(a, num_cu, st);
    if (rc == RML_ERR_UNSUPPORTED) rc = launch_trunk<1, INBF, 5, true, 1>(a, num_cu, st);
    return rc;
}
}  // namespace

static int trunk_entry(rml_ctx* ctx, const void* xz, const void* yz, const void* xy, int in_bf16, int64_t B, int H,
                       int W, const float* w1, const float* b1, const uint16_t* w2t, const float* b2,
                       uint16_t* feat, int kblock, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && H > 0 && W > 0, RML_ERR_INVALID, "rml_dnn_trunk: bad arguments");
    if (B == 0) return RML_OK;
    RML_REQUIRE(xz && yz && xy && w1 && b1 && w2t && b2 && feat, RML_ERR_INVALID, "rml_dnn_trunk: NULL argument");
    RML_REQUIRE(H % 4 == 0 && W % (in_bf16 ? 8 : 4) == 0, RML_ERR_UNSUPPORTED,
                "rml_dnn_trunk: H must be a multiple of 4 and W of %d (got %dx%d)", in_bf16 ? 8 : 4, H, W);
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_dnn_trunk: B too large");
    RML_REQUIRE((reinterpret_cast<uintptr_t>(w2t) & 15) == 0 && (reinterpret_cast<uintptr_t>(feat) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(xz) & 15) == 0 && (reinterpret_cast<uintptr_t>(yz) & 15) == 0 &&
                (reinterpret_cast<uintptr_t>(xy) & 15) == 0, RML_ERR_INVALID,
                "rml_dnn_trunk: planes, w2t and feat must be 16-byte aligned");
    RML_HIP(hipSetDevice(ctx->device));
    TrunkArgs a{};
    a.in[0] = xz; a.in[1] = yz; a.in[2] = xy; a.B = B; a.H = H; a.W = W;
    a.w1 = w1; a.b1 = b1; a.w2t = w2t; a.b2 = b2; a.feat = feat; a.kblock = kblock;
    RML_REQUIRE(!kblock || ((H / 4) * (W / 4)) % 2 == 0, RML_ERR_UNSUPPORTED, "rml_dnn_trunk_kblock: an even number of output pixels expected");
    const int rc = in_bf16 ? dispatch_trunk<true>(a, ctx->num_cu, static_cast<hipStream_t>(stream))
                           : dispatch_trunk<false>(a, ctx->num_cu, static_cast<hipStream_t>(stream));
    RML_REQUIRE(rc != RML_ERR_UNSUPPORTED, RML_ERR_UNSUPPORTED, "rml_dnn_trunk: plane %dx%d does not fit the LDS-resident trunk", H, W);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

extern "C" int rml_dnn_trunk(rml_ctx* ctx, const void* xz, const void* yz, const void* xy, int in_bf16, int64_t B, int H,
                             int W, const float* w1, const float* b1, const uint16_t* w2t, const float* b2,
                             uint16_t* feat, void* stream) {
    return trunk_entry(ctx, xz, yz, xy, in_bf16, B, H, W, w1, b1, w2t, b2, feat, 0, stream);
}

extern "C" int rml_dnn_trunk_kblock(rml_ctx* ctx, const void* xz, const void* yz, const void* xy, int in_bf16, int64_t B, int H,
                                    int W, const float* w1, const float* b1, const uint16_t* w2t, const float* b2,
                                    uint16_t* feat, void* stream) {
    return trunk_entry(ctx, xz, yz, xy, in_bf16, B, H, W, w1, b1, w2t, b2, feat, 1, stream);
}
