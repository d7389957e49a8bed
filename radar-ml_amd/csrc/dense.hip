// Dense tail of the multi-view CNN (dnn.py:78-88: Flatten -> Dense 64 relu -> Dense 64 relu -> Dense n softmax; Dropout is inactive at
// inference) on the 38 400-long bf16 feature rows the conv trunk (dnn.hip) writes.
//
// The first layer is the only one with bytes in it: 8 192 rows x 76.8 KB = 629 MB read once against 4.9 MFLOP per row -- HBM-bound
// (64 FLOP per byte at bf16 on the matrix cores is far below the machine's balance).  hipBLASLt's pick for this shape
// (MT64x64x256, one 135 KB workgroup per CU) streams it at 3.4 TB/s (186 us, profiles/r04_stats_dnn.txt); behind it PyTorch launches
// two GEMMs, two clamps, a cast and a softmax of ~5 us each.  Here:
//
//  * k_fc1_splitk: 128 rows x 64 outputs per workgroup and K-step of 64 elements (128 B per row), split-K -- in fixed pieces of
//    kFixSteps K-steps with one partial sum each, a workgroup taking as many whole pieces as make the grid about one round of three
//    workgroups per CU (round 6: the pieces, and with them every bit of the result, no longer depend on the batch size) --;
//    the tiles of a step are loaded lane-contiguously into registers three K-steps ahead and go
//    through two XOR-swizzled [rows][128 B] LDS stages, one LDS-only barrier per step; v_mfma_f32_32x32x16_bf16 with the weights
//    as the A (row) operand and the samples as B, so a lane ends up with 4 consecutive outputs of ONE sample per accumulator quad:
//    float4 stores of the float32 partial sums.
//  * k_dense_finish: one wave per sample sums the partials in piece order (deterministic), adds the bias, relu; the two small
//    layers run in float32 in the wave (lane = output unit, the previous layer's activations broadcast with v_readlane), softmax in
//    lane order.  Activations between the layers stay float32 (the autocast chain this replaces rounded them to bf16).
#include "rml_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kStepBytes = 128;                 // K-step: 64 bf16 per row
constexpr int kXRows = 128, kWRows = 64;        // rows of a tile (samples), hidden units
constexpr int kXBytes = kXRows * kStepBytes, kWBytes = kWRows * kStepBytes, kStageBytes = kXBytes + kWBytes;
constexpr int kHidden = 64;

struct Fc1Args {
    const uint8_t* x; int64_t ldx;      // bf16 rows, ldx BYTES apart
    int64_t kstride;                    // bytes from one K-step of a row to the next: 128 (rows), N * 128 (K-block layout, ldx = 128)
    int64_t N;
    const uint8_t* w; int64_t ldw;      // [64][K] bf16, ldw bytes apart ...
    int64_t wstride;                    // ... and 128 bytes per K-step; K-block layout: [K/64][64][64], ldw = 128, wstride = 8192
    int KT, S, steps;                   // K-steps in all, workgroup ranges, K-steps per range (a multiple of kFixSteps)
    float* partial;                     // [NF][N][64], NF = ceil(KT / kFixSteps)
};

// The K axis is cut into FIXED pieces of kFixSteps K-steps -- a function of K alone -- and every piece has its own partial sum, added
// up in order by k_dense_finish: a sample's probabilities do not depend on the batch it came in (round 6; rounds 4-5 cut K by the
// batch's tile count, so that another batch size gave other bits).  A workgroup takes a RANGE of whole pieces (as many as fill the
// machine at this batch size) in one run of its load pipeline and writes / clears its accumulators at every piece boundary.
constexpr int kFixSteps = 50;

// K-steps in flight per thread: the workgroup's sample tile and weight tile of a step wait in registers (six 16-byte chunks per thread)
constexpr int kDepth = 3;

// LDS-only barrier: __syncthreads() carries a fence that drains the vector-memory queue (the prefetched steps)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// How this kernel got here (each version measured on 8 192 x 38 400, the time to beat: hipBLASLt 186 us = 3.4 TB/s):
//  1. rows and weights by LDS-DMA (global_load_lds) into two stages, one in flight per workgroup: 174 us -- every step waits out
//     a full memory latency;
//  2. the same with a deeper DMA ring: hipcc treats a global_load_lds as a FLAT access that may return out of order and turns
//     EVERY later vmcnt wait into vmcnt(0) -- the queue drained once per loop trip;
//  3. no DMA, rows straight into the lanes that feed the matrix core (lane (n, h): 16 bytes of row n), three steps ahead: 190 us --
//     neighbouring lanes read rows 76.8 KB (K-block layout: 128 B) apart, so a wave instruction is 64 separate 16-byte requests;
//     the K-block layout alone: 159 us;
//  4. (this one) every global load instruction is lane-contiguous -- 8 lanes per 128-byte row piece, in the K-block layout the
//     whole instruction one 1 KiB run --, three steps ahead in registers, through two XOR-swizzled LDS stages into fragments.
__global__ __launch_bounds__(256, 3) void k_fc1_splitk(Fc1Args a) {
    __shared__ __align__(16) unsigned char smem[2 * kStageBytes];      // two stages of [128 sample rows | 64 weight rows] x 128 B
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x / a.S, split = blockIdx.x - tile * a.S;
    const int64_t n0 = (int64_t)tile * kXRows;
    const int k0 = split * a.steps;
    int cnt = a.KT - k0;
    cnt = cnt < a.steps ? cnt : a.steps;
    const int chalf = lane >> 5;

    // chunk s = q * 256 + tid of a tile: row s >> 3, 16-byte chunk s & 7 (a wave instruction: 8 whole 128-byte row pieces); LDS
    // image [row][128 B] with the chunk index XOR-swizzled by (row >> 1) & 7: conflict-free ds_read_b128 fragments
    const uint8_t* gx[4];
    const uint8_t* gw[2];
    int xdst[4], wdst[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int s = q * 256 + tid;
        const int r = s >> 3, c = s & 7;
        int64_t xr = n0 + r;
        xr = xr < a.N ? xr : a.N - 1;
        gx[q] = a.x + xr * a.ldx + c * 16 + (int64_t)k0 * a.kstride;
        xdst[q] = r * kStepBytes + ((c ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int s = q * 256 + tid;
        const int r = s >> 3, c = s & 7;
        gw[q] = a.w + (int64_t)r * a.ldw + c * 16 + (int64_t)k0 * a.wstride;
        wdst[q] = kXBytes + r * kStepBytes + ((c ^ ((r >> 1) & 7)) << 4);
    }
    const int last = cnt > 0 ? cnt - 1 : 0;
    v4i xs[kDepth][4], ws[kDepth][2];                   // step j lives in set j % kDepth
    auto issue = [&](int kt, v4i (&dx)[4], v4i (&dw)[2]) {      // step kt (clamped: the tail re-reads the last step, harmlessly)
        const int k = kt < last ? kt : last;
#pragma unroll
        for (int q = 0; q < 2; ++q) dw[q] = *reinterpret_cast<const v4i*>(gw[q] + (int64_t)k * a.wstride);
#pragma unroll
        for (int q = 0; q < 4; ++q) dx[q] = *reinterpret_cast<const v4i*>(gx[q] + (int64_t)k * a.kstride);
        // keep the steps in issue order: hipcc moves loads of one step behind the next one's, and the wait for them at the loop
        // head -- one static instruction for the prologue and the back edge -- then drains the queue every kDepth steps
        asm volatile("" ::: "memory");
    };
    auto put = [&](int stage, const v4i (&dx)[4], const v4i (&dw)[2]) {
        unsigned char* base = smem + stage * kStageBytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<v4i*>(base + xdst[q]) = dx[q];
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<v4i*>(base + wdst[q]) = dw[q];
    };

    // fragments: the wave's 32 samples are the B (column) operand, the 64 hidden units two A (row) tiles
    const int rb = wave * 32 + (lane & 31);
    const int boff = rb * kStepBytes, bsw = (rb >> 1) & 7;
    int aoff[2], asw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra = i * 32 + (lane & 31);
        aoff[i] = kXBytes + ra * kStepBytes; asw[i] = (ra >> 1) & 7;
    }
    v16f acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // acc[i][4 j + t] = hidden unit 32 i + 8 j + 4 (lane >> 5) + t of sample n0 + wave * 32 + (lane & 31)
    const int64_t n = n0 + rb;
    float* pdst = a.partial + ((int64_t)(k0 / kFixSteps) * a.N + (n < a.N ? n : 0)) * kHidden + 4 * chalf;
    int fill = 0;
    auto flush = [&]() __attribute__((always_inline)) {         // one fixed piece is complete: its partial sums leave, the accumulators restart
        if (n < a.N) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<float4*>(pdst + 32 * i + 8 * j) = make_float4(acc[i][4 * j], acc[i][4 * j + 1], acc[i][4 * j + 2], acc[i][4 * j + 3]);
        }
        pdst += a.N * kHidden;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
        fill = 0;
    };
    if (cnt > 0) {
#pragma unroll
        for (int d = 0; d < kDepth; ++d) issue(d, xs[d], ws[d]);
        put(0, xs[0], ws[0]);                           // step 0 goes to LDS; its set takes step kDepth
        issue(kDepth, xs[0], ws[0]);
        for (int t0 = 0; t0 < cnt; t0 += kDepth) {
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {
                const int kt = t0 + d;
                // step kt's stage (written during step kt - 1) is complete and visible; the other stage (read during step kt - 1)
                // is free for step kt + 1's tiles
                lds_barrier();
                put((kt + 1) & 1, xs[(d + 1) % kDepth], ws[(d + 1) % kDepth]);
                if (kt < cnt) {
                    const unsigned char* st = smem + (kt & 1) * kStageBytes;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        const int ch = 2 * kk + chalf;
                        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(st + boff + ((ch ^ bsw) << 4));
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const bf16x8 af = *reinterpret_cast<const bf16x8*>(st + aoff[i] + ((ch ^ asw[i]) << 4));
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[i], 0, 0, 0);
                        }
                    }
                    if (++fill == kFixSteps || kt == cnt - 1) flush();
                }
                issue(kt + kDepth + 1, xs[(d + 1) % kDepth], ws[(d + 1) % kDepth]);     // the set that has just gone to LDS
            }
        }
    }

}

struct FinishArgs {
    const float* partial; int S; int64_t N;
    const float* b1;            // [64]
    const float* w2t;           // [64 in][64 out] float32 (transposed: unit-contiguous)
    const float* b2;            // [64]
    const float* w3;            // [C][64]
    const float* b3;            // [C]
    int C;
    float* out;                 // [N][C] probabilities
};

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, BOUND));
}
// sum over the 64 lanes of a wave on DPP moves (six dependent VALU steps; the butterfly of __shfl_xor is six ds_bpermute round
// trips, and three of those chains per sample were most of k_dense_finish); the result is wave-uniform; a fixed order
__device__ __forceinline__ float wave_sum(float r) {
    r += dpp_mov<0xB1, 0xF, true>(0.0f, r);     // quad_perm [1,0,3,2]
    r += dpp_mov<0x4E, 0xF, true>(0.0f, r);     // quad_perm [2,3,0,1]
    r += dpp_mov<0x141, 0xF, true>(0.0f, r);    // row_half_mirror
    r += dpp_mov<0x140, 0xF, true>(0.0f, r);    // row_mirror: every lane of a 16-lane row holds the row's sum
    r += dpp_mov<0x142, 0xA, false>(0.0f, r);   // row_bcast15 into rows 1 and 3
    r += dpp_mov<0x143, 0xC, false>(0.0f, r);   // row_bcast31 into rows 2 and 3: row 3 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 63));
}

__global__ __launch_bounds__(256) void k_dense_finish(FinishArgs a) {
    __shared__ float w2s[kHidden * kHidden];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kHidden * kHidden; i += 256) w2s[i] = a.w2t[i];
    const float b1 = a.b1[lane], b2 = a.b2[lane];
    float w3l[16], b3l[16];                         // the last layer's kernel column of this lane, the biases
#pragma unroll
    for (int c = 0; c < 16; ++c) { w3l[c] = c < a.C ? a.w3[c * kHidden + lane] : 0.0f; b3l[c] = c < a.C ? a.b3[c] : 0.0f; }
    __syncthreads();
    // persistent: wave w of the grid takes samples w, w + #waves, ...; the partial sums of the next sample are in flight (16 splits
    // at most per batch of loads) while this one runs its 64-step layer
    const int64_t nw = (int64_t)gridDim.x * 4;
    constexpr int SB = 16;
    auto gather = [&](int64_t n) -> float {
        float h = 0.0f;
        for (int s0 = 0; s0 < a.S; s0 += SB) {
            float p[SB];
#pragma unroll
            for (int s = 0; s < SB; ++s) p[s] = (s0 + s < a.S && n < a.N) ? a.partial[((int64_t)(s0 + s) * a.N + n) * kHidden + lane] : 0.0f;
#pragma unroll
            for (int s = 0; s < SB; ++s) h += p[s];     // split order: deterministic
        }
        return h;
    };
    int64_t n = (int64_t)blockIdx.x * 4 + (tid >> 6);
    float hn = gather(n);
    for (; n < a.N; n += nw) {
        const float h = fmaxf(hn + b1, 0.0f);
        hn = gather(n + nw);
        // second layer: lane = output unit; four independent chains over the 64 inputs (each input broadcast with v_readlane)
        float g0 = b2, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;
#pragma unroll
        for (int k = 0; k < kHidden; k += 4) {
            g0 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k)), w2s[k * kHidden + lane], g0);
            g1 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k + 1)), w2s[(k + 1) * kHidden + lane], g1);
            g2 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k + 2)), w2s[(k + 2) * kHidden + lane], g2);
            g3 = fmaf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(h), k + 3)), w2s[(k + 3) * kHidden + lane], g3);
        }
        const float g = fmaxf((g0 + g1) + (g2 + g3), 0.0f);
        // logits in every lane (C <= 16), softmax in float32 like torch.softmax(lg.float())
        float lg[16];
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (c < a.C) {
                lg[c] = wave_sum(g * w3l[c]) + b3l[c];
                mx = fmaxf(mx, lg[c]);
            }
        }
        float den = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < a.C) { lg[c] = expf(lg[c] - mx); den += lg[c]; }
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c < a.C && lane == c) a.out[n * a.C + c] = lg[c] / den;
    }
}

// ---- the first dense layer on float32 feature rows (the margin guard's re-scoring tail: dnn.py Classifier._tail_float32) ------------
// What the guard needs from it is a result that is a function of the ROW alone: hipBLASLt's float32 GEMM for (a few hundred rows) x
// 38 400 x 64 splits K with atomics -- the same call twice gave probabilities 1e-7 apart, and a row scored alone or inside a larger
// candidate set 7e-7 apart (session r6b) -- so "the same call again: the same bits" did not hold with the guard on.  Here the K
// axis is cut into splits of kF32Split elements whatever the batch (a function of K only); a workgroup takes 128 rows x 64 units x
// one split, a wave 32 rows x 64 units on v_mfma_f32_32x32x2_f32 (float32 operands, float32 accumulation: the arithmetic class Keras
// runs these layers in), and an output element's sum runs over its split in ONE fixed order -- K-steps of 32 ascending; inside a
// step the instruction t = 0..15 adds the pair (k0 + t, k0 + 16 + t) -- that no other row takes part in; k_dense_finish adds the
// splits in order.  No LDS and no barrier: a lane's operands are the 64 contiguous bytes of ITS row (lane half h: k0 + 16 h ..) and
// of its two weight rows, loaded as four 16-byte pieces each, one K-step ahead in registers; per row and step a whole 128-byte
// line.  (The first version -- 4 x 4 register tiles on the vector ALU out of LDS tiles -- ran at 26 TFLOP/s: 3.1 ms per 16 384 rows
// against 3.4 for the float32-class trunk in front of it, session r6c.)
constexpr int kF32Split = 2560, kF32Step = 32;

struct Fc1F32Args {
    const float* x; int64_t ldx;        // float32 rows, ldx floats apart
    int64_t N, K;
    const float* w; int64_t ldw;        // [64][K] float32 (torch Linear layout)
    int S;                              // splits = ceil(K / kF32Split)
    float* partial;                     // [S][N][64]
};

// RB = row blocks of 32 per wave.  One block: 128 rows per workgroup, the shape for a few hundred candidates (more workgroups).  Two:
// 256 rows per workgroup and half the operand bytes per matrix instruction -- with one block a CU's twelve waves ask the vector
// cache for 70 B/clk (every wave re-reads the 8 KB weight slice of its step), more than it delivers: 52 TFLOP/s (session r6d); large
// candidate sets take two.  The sums of an output element are the same instruction sequence either way: the same bits.
template <int RB>
__global__ __launch_bounds__(256) void k_fc1_f32(Fc1F32Args a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x / a.S, s = blockIdx.x - tile * a.S;
    const int64_t r0 = (int64_t)tile * (128 * RB) + 32 * RB * wave;       // the wave's 32 RB rows
    const int64_t k0 = (int64_t)s * kF32Split;
    const int64_t k1 = a.K < k0 + kF32Split ? a.K : k0 + kF32Split;
    const int n = lane & 31, h = lane >> 5;
    const float* __restrict__ xp[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        int64_t row = r0 + 32 * b + n;
        row = row < a.N ? row : a.N - 1;                      // rows past the batch re-read the last one; nothing of them is stored
        xp[b] = a.x + row * a.ldx + 16 * h;
    }
    const float* __restrict__ wp0 = a.w + (int64_t)n * a.ldw + 16 * h;
    const float* __restrict__ wp1 = a.w + (int64_t)(n + 32) * a.ldw + 16 * h;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    v16f acc[RB][2];
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[b][0][r] = 0.0f; acc[b][1][r] = 0.0f; }
    float4 xa[RB][4], wa[4], wb[4];
    // K and the split are multiples of 4: a 16-byte piece is inside or outside [k0, k1) as a whole.  The loads are unconditional (a
    // piece outside re-reads the split's first); what they returned is replaced by zeros on both sides when the step USES it -- a
    // select behind the load would make the wave wait for its prefetch before the matrix instructions it should run under
    auto fetch = [&](int64_t kb) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t kk = kb + 4 * q;
            const int64_t kc = (kk + 16 * h < k1) ? kk : k0;
#pragma unroll
            for (int b = 0; b < RB; ++b) xa[b][q] = *reinterpret_cast<const float4*>(xp[b] + kc);
            wa[q] = *reinterpret_cast<const float4*>(wp0 + kc);
            wb[q] = *reinterpret_cast<const float4*>(wp1 + kc);
        }
    };
    fetch(k0);
    for (int64_t kb = k0; kb < k1; kb += kF32Step) {
        float4 xc[RB][4], wc[4], wd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool in = kb + 4 * q + 16 * h < k1;
#pragma unroll
            for (int b = 0; b < RB; ++b) xc[b][q] = in ? xa[b][q] : z4;
            wc[q] = in ? wa[q] : z4; wd[q] = in ? wb[q] : z4;
        }
        fetch(kb + kF32Step < k1 ? kb + kF32Step : k0);      // the next step's operands are in flight under this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);                    // ... and stay there: no wait for them is needed before the loop's next trip
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float wv[4] = {wc[q].x, wc[q].y, wc[q].z, wc[q].w};
            const float wu[4] = {wd[q].x, wd[q].y, wd[q].z, wd[q].w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int b = 0; b < RB; ++b) {
                    const float xv = c == 0 ? xc[b][q].x : (c == 1 ? xc[b][q].y : (c == 2 ? xc[b][q].z : xc[b][q].w));
                    acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, wv[c], acc[b][0], 0, 0, 0);
                    acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv, wu[c], acc[b][1], 0, 0, 0);
                }
            }
        }
    }
    // D[row of A = sample (r & 3) + 8 (r >> 2) + 4 h][column of B = unit n]: a lane half writes 32 consecutive units of one sample
#pragma unroll
    for (int b = 0; b < RB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t smp = r0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (smp < a.N) {
                float* dst = a.partial + ((int64_t)s * a.N + smp) * kHidden + n;
                dst[0] = acc[b][0][r];
                dst[32] = acc[b][1][r];
            }
        }
}

int fixed_pieces(int KT) { return (KT + kFixSteps - 1) / kFixSteps; }

// fixed pieces per workgroup range: about one round of three workgroups per CU where the K extent allows it
int pick_group(int64_t tiles, int KT, int num_cu) {
    const int nf = fixed_pieces(KT);
    int64_t want = ((int64_t)3 * num_cu + tiles - 1) / tiles;      // ranges wanted
    if (want < 1) want = 1;
    if (want > nf) want = nf;
    return (int)((nf + want - 1) / want);
}

}  // namespace

extern "C" int64_t rml_dnn_dense_workspace_bytes(rml_ctx* ctx, int64_t N, int64_t K) {
    if (!ctx || N <= 0 || K <= 0) return 0;
    return (int64_t)fixed_pieces((int)(K / 64)) * N * kHidden * (int64_t)sizeof(float);
}

extern "C" int rml_dnn_dense_tail(rml_ctx* ctx, const uint16_t* feat, int64_t ld_feat, int kblock, int64_t N, int64_t K, const uint16_t* w1, const float* b1,
                                  const float* w2t, const float* b2, const float* w3, const float* b3, int n_classes, float* workspace,
                                  int64_t workspace_bytes, float* proba, void* stream) {
    RML_REQUIRE(ctx && N >= 0 && K > 0, RML_ERR_INVALID, "rml_dnn_dense_tail: bad arguments");
    RML_REQUIRE(K % 64 == 0 && (kblock || (ld_feat >= K && ld_feat % 8 == 0)), RML_ERR_UNSUPPORTED,
                "rml_dnn_dense_tail: K = %lld must be a multiple of 64 (ld_feat of 8)", (long long)K);
    RML_REQUIRE(n_classes >= 1 && n_classes <= 16, RML_ERR_UNSUPPORTED, "rml_dnn_dense_tail: 1..16 classes");
    if (N == 0) return RML_OK;
    RML_REQUIRE(feat && w1 && b1 && w2t && b2 && w3 && b3 && workspace && proba, RML_ERR_INVALID, "rml_dnn_dense_tail: NULL argument");
    RML_REQUIRE(((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
                RML_ERR_INVALID, "rml_dnn_dense_tail: feat, w1 and the workspace need 16-byte alignment");
    RML_REQUIRE(N < (int64_t)1 << 31 && K < (int64_t)1 << 30, RML_ERR_UNSUPPORTED, "rml_dnn_dense_tail: too large");
    RML_REQUIRE(workspace_bytes >= rml_dnn_dense_workspace_bytes(ctx, N, K), RML_ERR_INVALID, "rml_dnn_dense_tail: workspace too small");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t tiles = (N + kXRows - 1) / kXRows;
    Fc1Args fa{};
    fa.x = reinterpret_cast<const uint8_t*>(feat); fa.N = N;
    fa.ldx = kblock ? kStepBytes : ld_feat * 2;
    fa.kstride = kblock ? N * kStepBytes : kStepBytes;
    fa.w = reinterpret_cast<const uint8_t*>(w1);
    fa.ldw = kblock ? kStepBytes : K * 2;
    fa.wstride = kblock ? kWBytes : kStepBytes;
    fa.KT = (int)(K / 64);
    const int nf = fixed_pieces(fa.KT), grp = pick_group(tiles, fa.KT, ctx->num_cu);
    fa.S = (nf + grp - 1) / grp;
    fa.steps = grp * kFixSteps;
    fa.partial = workspace;
    hipLaunchKernelGGL(k_fc1_splitk, dim3((unsigned)(tiles * fa.S)), dim3(256), 0, st, fa);
    FinishArgs fi{};
    fi.partial = workspace; fi.S = nf; fi.N = N; fi.b1 = b1; fi.w2t = w2t; fi.b2 = b2; fi.w3 = w3; fi.b3 = b3; fi.C = n_classes; fi.out = proba;
    const int64_t blocks = (N + 3) / 4;
    hipLaunchKernelGGL(k_dense_finish, dim3((unsigned)(blocks < 2 * ctx->num_cu ? blocks : 2 * ctx->num_cu)), dim3(256), 0, st, fi);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

extern "C" int64_t rml_dnn_dense_tail_f32_workspace_bytes(int64_t N, int64_t K) {
    if (N <= 0 || K <= 0) return 0;
    return ((K + kF32Split - 1) / kF32Split) * N * kHidden * (int64_t)sizeof(float);
}

extern "C" int rml_dnn_dense_tail_f32(rml_ctx* ctx, const float* feat, int64_t ld_feat, int64_t N, int64_t K, const float* w1, const float* b1,
                                      const float* w2t, const float* b2, const float* w3, const float* b3, int n_classes, float* workspace,
                                      int64_t workspace_bytes, float* proba, void* stream) {
    RML_REQUIRE(ctx && N >= 0 && K > 0, RML_ERR_INVALID, "rml_dnn_dense_tail_f32: bad arguments");
    RML_REQUIRE(K % 4 == 0 && ld_feat >= K && ld_feat % 4 == 0, RML_ERR_UNSUPPORTED,
                "rml_dnn_dense_tail_f32: K = %lld and ld_feat must be multiples of 4", (long long)K);
    RML_REQUIRE(n_classes >= 1 && n_classes <= 16, RML_ERR_UNSUPPORTED, "rml_dnn_dense_tail_f32: 1..16 classes");
    if (N == 0) return RML_OK;
    RML_REQUIRE(feat && w1 && b1 && w2t && b2 && w3 && b3 && workspace && proba, RML_ERR_INVALID, "rml_dnn_dense_tail_f32: NULL argument");
    RML_REQUIRE(((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
                RML_ERR_INVALID, "rml_dnn_dense_tail_f32: feat, w1 and the workspace need 16-byte alignment");
    RML_REQUIRE(N < (int64_t)1 << 31 && K < (int64_t)1 << 30, RML_ERR_UNSUPPORTED, "rml_dnn_dense_tail_f32: too large");
    RML_REQUIRE(workspace_bytes >= rml_dnn_dense_tail_f32_workspace_bytes(N, K), RML_ERR_INVALID, "rml_dnn_dense_tail_f32: workspace too small");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    Fc1F32Args fa{};
    fa.x = feat; fa.ldx = ld_feat; fa.N = N; fa.K = K; fa.w = w1; fa.ldw = K;
    fa.S = (int)((K + kF32Split - 1) / kF32Split);
    fa.partial = workspace;
    const bool two = N >= 4096;                             // (the choice changes no bit of the result: see k_fc1_f32)
    const int64_t tiles = two ? (N + 255) / 256 : (N + 127) / 128;
    RML_REQUIRE(tiles * fa.S < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_dnn_dense_tail_f32: too large");
    if (two) hipLaunchKernelGGL(k_fc1_f32<2>, dim3((unsigned)(tiles * fa.S)), dim3(256), 0, st, fa);
    else hipLaunchKernelGGL(k_fc1_f32<1>, dim3((unsigned)(tiles * fa.S)), dim3(256), 0, st, fa);
    FinishArgs fi{};
    fi.partial = workspace; fi.S = fa.S; fi.N = N; fi.b1 = b1; fi.w2t = w2t; fi.b2 = b2; fi.w3 = w3; fi.b3 = b3; fi.C = n_classes; fi.out = proba;
    const int64_t blocks = (N + 3) / 4;
    hipLaunchKernelGGL(k_dense_finish, dim3((unsigned)(blocks < 2 * ctx->num_cu ? blocks : 2 * ctx->num_cu)), dim3(256), 0, st, fi);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
