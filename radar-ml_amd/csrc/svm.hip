// RBF / linear C-SVC decision function on gfx950 matrix cores.
//
// Reference arithmetic replaced (sk: = scikit-learn, the reference's SVM dependency):
//   sk:svm/src/libsvm/svm.cpp:461-475,514   K[n,m] = exp(-gamma * ||x_n - sv_m||^2)      (RBF)
//   sk:svm/src/libsvm/svm.cpp:457            K[n,m] = x_n . sv_m                          (linear)
//   sk:svm/src/libsvm/svm.cpp:2847-2890      dec[n,p] = sum_m coef*K - rho[p]; OvO vote
//   sk:utils/multiclass.py:542-584           ovr = votes + s/(3(|s|+1))
//   sk:calibration.py:727-784,928-942        expit(-(a*T+b)), normalise, clip, argmax
// called from train.py:217,723-724 and predict.py:60.
//
// libsvm walks samples x SVs x D serially in float64.  Here the sample x SV inner products are
// one GEMM on the matrix cores and everything after it is a fused float64 epilogue; the N x M
// kernel matrix never exists in memory.
//
//   ||x - s||^2 = ||x||^2 + ||s||^2 - 2 x.s
//
// Two operand paths share one kernel skeleton (tiles staged by LDS-DMA, 128-byte rows):
//   I8   radar features are integer codes c in [0,255] (optionally scaled by 1/255), so
//        x.s is EXACT in int32 on v_mfma_i32_32x32x32_i8.  Codes are stored biased
//        (byte = c ^ 0x80 = int8 c-128):  sum (a-128)(b-128) = sum ab - 128 (sum a + sum b) + 128^2 D.
//        d^2 is then an exact integer, evaluated in float64.
//   F64  general rows (anything not on the code grid: augmented / zoomed data) on
//        v_mfma_f64_16x16x4_f64: the float32 operands are widened to float64 in registers, the
//        products and the accumulation are float64 -- the arithmetic class of libsvm itself, at
//        the f64 matrix rate (78.6 TF).  This is what RML_PATH_AUTO uses for non-grid rows.
//   F32  opt-in approximate path on v_mfma_f32_32x32x2_f32 (f32 accumulate, 2x the F64 rate;
//        measured error of the decision values ~1e-4..1e-3, i.e. outside the 1e-5 bar).
// Epilogue (float64): K = exp(-gamma d^2); per-pair weights W[p][m] (the libsvm pair loop
// unrolled into a P x M matrix at load) -> S[n][p] += W[p][m] K.  The MFMA is issued with the
// SV tile as the A (row) operand and the sample tile as the B (column) operand, so that a lane
// owns ONE sample column and 16 SV rows per accumulator: the sum over SVs is in-lane, only a
// 2-lane + 2-wave reduction per workgroup remains.  Per-SV-tile partial sums are written to
// HBM (ST x N x P float64, fixed order) and summed by the finishing kernel in tile order, so
// results are deterministic run to run.
//
// Workgroup tile 128 SVs x 128 samples, 4 waves as 2x2, each wave 2x2 MFMA tiles of 32x32.
// K-step = 128 bytes per row (128 codes or 32 floats).  LDS image of a tile: row-major
// [128 rows][128 B] with the 16-byte chunk index XOR-swizzled by (row>>1)&7, which makes the
// ds_read_b128 fragment reads (lane = row, 16 B each) bank-conflict free.  LDS-DMA writes
// lane-linear, so the swizzle is applied to the per-lane GLOBAL source address and again on
// the read (both-sides rule).  Double-buffered: the DMA of K-step t+1 is in flight while the
// MFMAs of K-step t run.  Block index -> (sample tile, SV tile) is XCD-aware: the 16 SV tiles
// that share a sample tile run on one XCD so the sample K-slices are L2 hits.
#include "rml_internal.h"
#include <math.h>
#include <vector>
#include <stdlib.h>
#include <type_traits>
#include <algorithm>
#include <new>

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kTile = 128;         // rows per operand tile
constexpr int kStepBytes = 128;    // K-step bytes per row
constexpr int kTileBytes = kTile * kStepBytes;   // 16 KiB
constexpr int PATH_I8 = 0, PATH_F32 = 1, PATH_F64 = 2;

// exp(x) for x <= 0 in float64, table-driven (Tang 1989): x = n L + r with L = ln2/64, n = 64 k + j, |r| <= L/2 = 0.0054,
//     exp(x) = 2^k * T[j] * (1 + p(r)),   p(r) = r + r^2 (1/2 + r (1/6 + r (1/24 + r (1/120 + r/720)))),   T[j] = 2^(j/64).
// Error: n L_hi is exact (L_hi carries 32 bits, |n| < 2^17), the reduction error is ~|n| ulp(L_lo) ~ 1e-22; the polynomial is
// truncated at r^7/5040 < 3e-20; T[j] is correctly rounded (0.5 ulp) and T + T p is one fma (0.5 ulp) on a p computed to
// ~1e-19 absolute: < 1.1 ulp in all, against ~1 ulp for the library routine it replaces (a degree-11 polynomial on |r| <= ln2/2
// plus range selects: ~35 instructions and a 20-deep dependent chain per kernel value, here 17 and 12).  Arguments below
// -1000 are clamped; v_ldexp_f64 then underflows to 0 like libm.  tab = the 64-entry table in LDS (exp_tab_init).
__constant__ double kExp2Tab[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0};
constexpr int kExpTabBytes = 64 * 8;

__device__ __forceinline__ void exp_tab_init(double* tab, int tid) {
    if (tid < 64) tab[tid] = kExp2Tab[tid];
}

__device__ __forceinline__ double rml_exp_neg(double x, const double* tab) {
    x = fmax(x, -1000.0);
    const double nf = rint(x * 0x1.71547652b82fep+6);              // x * 64/ln2
    double r = fma(nf, -0x1.62e42fee00000p-7, x);
    r = fma(nf, -0x1.a39ef35793c76p-39, r);
    const int n = (int)nf;
    const double T = tab[n & 63];
    double p = fma(r, 0x1.6c16c16c16c17p-10, 0x1.1111111111111p-7);   // 1/720, 1/120
    p = fma(r, p, 0x1.5555555555555p-5);                               // 1/24
    p = fma(r, p, 0x1.5555555555555p-3);                               // 1/6
    p = fma(r, p, 0.5);
    p = fma(r * r, p, r);
    return ldexp(fma(T, p, T), n >> 6);
}

struct GemmArgs {
    const uint8_t* sv; int64_t ld_sv;     // SV operand, bytes per row
    const uint8_t* x;  int64_t ld_x;      // sample operand, bytes per row
    int KT;                                // K-steps
    int64_t N;                             // valid sample rows
    int ST, FT;                            // SV tiles, sample tiles
    const int32_t* tile_exact; int want;   // process sample tile ft iff tile_exact[ft] == want (NULL: all)
    const int32_t* x_isum; const int64_t* x_isq;   // exact path row statistics
    const double* x_nsq;                            // f32 path row norms
    const double* sv_term;                 // Mpad per-SV term (path/kernel specific)
    const double* W; int64_t Mpad;         // PT x Mpad pair weights
    double gs;                             // gamma/scale^2 (rbf) ; 1/scale^2 (linear, exact path)
    int kernel;
    double* partial; int64_t Npart;        // ST x Npart x PT
    double* kmat; int64_t ld_k; int64_t M; // KM instantiations: kernel values K[n][m] for m < M (rml_svm_kernel_matrix)
    int64_t sv_rows;                       // SV rows that exist in memory (Mpad); the 256-row kernel clamps to it
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int PATH, int PT, bool KM = false>
__global__ __launch_bounds__(256, 2) void k_svm_gemm(GemmArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
        const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    // an XCD owns the sample tiles {xcd, xcd+8, ...} and walks them FASTEST, so the workgroups resident on an
    // XCD form an (all its sample tiles) x (few SV tiles) block sharing K-slices through that XCD's L2
    // (measured with TCC_HIT/MISS: L2 misses -30 % vs walking the SV tiles fastest)
    const int XPX = (a.FT + 7) >> 3;
    const int ftile = (slot % XPX) * 8 + xcd;
    const int stile = slot / XPX;
    if (ftile >= a.FT) return;
    if (a.tile_exact && a.tile_exact[ftile] != a.want) return;
    const int64_t f0 = (int64_t)ftile * kTile;
    const int64_t m0 = (int64_t)stile * kTile;

    // per-SV epilogue table in LDS: [128][1+PT] float64, then the 2^(j/64) table of rml_exp_neg
    double* svw = reinterpret_cast<double*>(smem + 4 * kTileBytes);
    const double* etab = svw + kTile * (1 + PT);
    exp_tab_init(svw + kTile * (1 + PT), tid);
    for (int idx = tid; idx < kTile * (1 + PT); idx += 256) {
        int m = idx / (1 + PT), c = idx - m * (1 + PT);
        svw[idx] = (c == 0) ? a.sv_term[m0 + m] : a.W[(int64_t)(c - 1) * a.Mpad + m0 + m];
    }

    // staging addresses: 16 wave-instructions of 1 KiB per operand tile, 4 per wave
    const uint8_t* gsv[4];
    const uint8_t* gx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int s = (wave * 4 + q) * 64 + lane;          // 16-byte slot in the LDS image
        int r = s >> 3;
        int c = (s & 7) ^ ((r >> 1) & 7);            // inverse swizzle on the source
        gsv[q] = a.sv + (m0 + r) * a.ld_sv + c * 16;
        int64_t xr = f0 + r; xr = xr < a.N ? xr : a.N - 1;
        gx[q] = a.x + xr * a.ld_x + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        unsigned char* base = smem + buf * 2 * kTileBytes;
        const int64_t ko = (int64_t)kt * kStepBytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(gsv[q] + ko, base + (wave * 4 + q) * 1024);
            glds16(gx[q] + ko, base + kTileBytes + (wave * 4 + q) * 1024);
        }
    };

    // fragment read offsets (bytes within a tile image); lane = row, swizzled chunk
    int aoff[2], asw[2], boff[2], bsw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int ra = wr * 64 + t * 32 + (lane & 31);
        int rb = wc * 64 + t * 32 + (lane & 31);
        aoff[t] = ra * kStepBytes; asw[t] = (ra >> 1) & 7;
        boff[t] = rb * kStepBytes; bsw[t] = (rb >> 1) & 7;
    }
    const int chalf = lane >> 5;

    // accumulators: I8/F32: 2x2 tiles of 32x32 (16 regs each); F64: 4x4 tiles of 16x16 (4 doubles each)
    using acc_t = typename std::conditional<PATH == PATH_I8, v16i, v16f>::type;
    acc_t acc[2][2];
    v4d accd[PATH == PATH_F64 ? 4 : 1][PATH == PATH_F64 ? 4 : 1];
    if constexpr (PATH == PATH_F64) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) accd[i][j] = v4d{0.0, 0.0, 0.0, 0.0};
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    }
    // F64 fragment addressing: lane = (row l&15, k-group l>>4) of a 16-row tile
    int doff_a[4], dsw_a[4], doff_b[4], dsw_b[4];
    if constexpr (PATH == PATH_F64) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int ra = wr * 64 + t * 16 + (lane & 15);
            int rb = wc * 64 + t * 16 + (lane & 15);
            doff_a[t] = ra * kStepBytes; dsw_a[t] = (ra >> 1) & 7;
            doff_b[t] = rb * kStepBytes; dsw_b[t] = (rb >> 1) & 7;
        }
    }
    const int kgrp = lane >> 4;

    stage(0, 0);
    for (int kt = 0; kt < a.KT; ++kt) {
        __syncthreads();                       // DMA of step kt landed (vmcnt(0)) and visible
        if (kt + 1 < a.KT) stage(kt + 1, (kt + 1) & 1);
        const unsigned char* sA = smem + (kt & 1) * 2 * kTileBytes;
        const unsigned char* sB = sA + kTileBytes;
        if constexpr (PATH == PATH_F64) {
            // 32 floats per row per K-step = 8 chunks of 4; pass h covers chunks 4h..4h+3, one per
            // k-group; MFMA c of a pass multiplies element c of every lane's chunk (k = 4*chunk + c).
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int ch = 4 * hh + kgrp;
                v4f af[4], bf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    af[t] = *reinterpret_cast<const v4f*>(sA + doff_a[t] + ((ch ^ dsw_a[t]) << 4));
                    bf[t] = *reinterpret_cast<const v4f*>(sB + doff_b[t] + ((ch ^ dsw_b[t]) << 4));
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double ad[4], bd[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) { ad[t] = (double)af[t][c]; bd[t] = (double)bf[t][c]; }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            accd[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad[i], bd[j], accd[i][j], 0, 0, 0);
                }
            }
        } else {
            // software-pipelined fragments: the ds_read_b128 of sub-step kk+1 are in flight while the MFMAs of kk
            // run (two register sets + sched_barrier; left alone hipcc reuses one set and waits lgkmcnt(0) every 4 MFMAs)
            v4i af[2][2], bf[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[0][t] = *reinterpret_cast<const v4i*>(sA + aoff[t] + ((chalf ^ asw[t]) << 4));
                bf[0][t] = *reinterpret_cast<const v4i*>(sB + boff[t] + ((chalf ^ bsw[t]) << 4));
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk < 3) {
                    const int ch = 2 * (kk + 1) + chalf;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        af[(kk + 1) & 1][t] = *reinterpret_cast<const v4i*>(sA + aoff[t] + ((ch ^ asw[t]) << 4));
                        bf[(kk + 1) & 1][t] = *reinterpret_cast<const v4i*>(sB + boff[t] + ((ch ^ bsw[t]) << 4));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ahead of this sub-step's MFMAs
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if constexpr (PATH == PATH_I8) {
                            acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[kk & 1][i], bf[kk & 1][j], acc[i][j], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(__int_as_float(af[kk & 1][i][c]),
                                                                                __int_as_float(bf[kk & 1][j][c]), acc[i][j], 0, 0, 0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    const bool rbf = (a.kernel == RML_KERNEL_RBF);
    // [128][PT] cross-thread exchange, over the tile images once they are consumed: the request stays at 4 tiles + the SV table
    // (69 632 B for three pairs; with the exchange behind the table it was 72 704 B.  Same-box A/B of the fused pipeline at
    // 64x64x128: +2 % with the smaller request, the byte-native rows unchanged)
    double* xch = reinterpret_cast<double*>(smem);
    if constexpr (PATH == PATH_F64) {
        // float64 accumulators: 128 x 128 x 8 B = 128 KiB, so the LDS round trip is done in two column
        // halves of 64 KiB (the waves with wc == pass own that half).  Thread t then owns sample column
        // n' = t & 63 of the half and the SV quarter t >> 6 (32 in-lane SV rows).
        double* gd = reinterpret_cast<double*>(smem);
        const int nq = tid & 63, qd = tid >> 6;
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            if (wc == pass) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int ml = wr * 64 + i * 16 + kgrp + 4 * r;     // f64 C/D map: row = (lane>>4) + 4 reg
                            const int nl = j * 16 + (lane & 15);                //              col = lane & 15
                            gd[ml * 64 + nl] = accd[i][j][r];
                        }
            }
            __syncthreads();
            const int64_t n = f0 + pass * 64 + nq;
            const int64_t nc = n < a.N ? n : a.N - 1;
            const double xt = rbf ? a.x_nsq[nc] : 0.0;
            double S[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) S[p] = 0.0;
#pragma unroll 2
            for (int mm = 0; mm < 32; ++mm) {
                const int ml = qd * 32 + mm;
                const double* e = svw + ml * (1 + PT);
                const double g = gd[ml * 64 + nq];
                double kv;
                if (rbf) {
                    double d2 = xt + e[0] - 2.0 * g;
                    d2 = d2 > 0.0 ? d2 : 0.0;
                    kv = rml_exp_neg(-a.gs * d2, etab);
                } else {
                    kv = g;
                }
                if constexpr (KM) {
                    const int64_t mg = (int64_t)stile * kTile + ml;
                    if (n < a.N && mg < a.M) a.kmat[n * a.ld_k + mg] = kv;
                }
#pragma unroll
                for (int p = 0; p < PT; ++p) S[p] = fma(e[1 + p], kv, S[p]);
            }
            __syncthreads();                   // G half consumed: reuse its LDS for the exchange
            double* x4 = gd;                   // [4 quarters][64][PT]
#pragma unroll
            for (int p = 0; p < PT; ++p) x4[(qd * 64 + nq) * PT + p] = S[p];
            __syncthreads();
            if (qd == 0 && n < a.N) {
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    double t = x4[(0 * 64 + nq) * PT + p] + x4[(1 * 64 + nq) * PT + p];
                    t += x4[(2 * 64 + nq) * PT + p] + x4[(3 * 64 + nq) * PT + p];
                    a.partial[((int64_t)stile * a.Npart + n) * PT + p] = t;
                }
            }
        }
        return;
    }

    // ---- fused float64 epilogue ----------------------------------------------------------
    // The accumulators go through LDS once (the four 16 KiB tile images are free now and are
    // exactly 128 x 128 x 4 B) so that the epilogue can use its own thread mapping: thread t
    // owns sample column n = t & 127 and the SV half h = t >> 7, i.e. 64 in-lane SV rows, reads
    // G[m][n] with consecutive lanes on consecutive banks and the per-SV table as broadcasts.
    __syncthreads();                           // everyone is done reading the tile images
    {
        int* gl = reinterpret_cast<int*>(smem);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
                    const int nl = wc * 64 + j * 32 + (lane & 31);
                    int bits;
                    if constexpr (PATH == PATH_I8) bits = acc[i][j][r]; else bits = __float_as_int(acc[i][j][r]);
                    gl[ml * kTile + nl] = bits;
                }
    }
    __syncthreads();
    const int nl = tid & 127, h = tid >> 7;
    int64_t n = f0 + nl;
    const int64_t nc = n < a.N ? n : a.N - 1;
    double xt;
    if constexpr (PATH == PATH_I8) {
        // d^2 = (isq_x - 256 isum_x) + (isq_s - 256 isum_s + 32768 D) - 2 G'
        xt = rbf ? (double)(a.x_isq[nc] - 256 * (int64_t)a.x_isum[nc]) : 128.0 * (double)a.x_isum[nc];
    } else {
        xt = rbf ? a.x_nsq[nc] : 0.0;
    }
    double S[PT];
#pragma unroll
    for (int p = 0; p < PT; ++p) S[p] = 0.0;
    const int* gcol = reinterpret_cast<const int*>(smem) + nl;
#pragma unroll 2
    for (int mm = 0; mm < 64; ++mm) {
        const int ml = h * 64 + mm;
        const double* e = svw + ml * (1 + PT);
        const int bits = gcol[ml * kTile];
        const double g = (PATH == PATH_I8) ? (double)bits : (double)__int_as_float(bits);
        double kv;
        if (rbf) {
            double d2 = xt + e[0] - 2.0 * g;
            d2 = d2 > 0.0 ? d2 : 0.0;
            kv = rml_exp_neg(-a.gs * d2, etab);
        } else {
            kv = (PATH == PATH_I8) ? (g + xt + e[0]) * a.gs : g;
        }
        if constexpr (KM) {
            const int64_t mg = (int64_t)stile * kTile + ml;
            if (n < a.N && mg < a.M) a.kmat[n * a.ld_k + mg] = kv;
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) S[p] = fma(e[1 + p], kv, S[p]);
    }
    __syncthreads();                           // G consumed: its LDS carries the exchange between the two SV halves
    if (h == 1) {
#pragma unroll
        for (int p = 0; p < PT; ++p) xch[nl * PT + p] = S[p];
    }
    __syncthreads();
    if (h == 0 && n < a.N) {
#pragma unroll
        for (int p = 0; p < PT; ++p) a.partial[((int64_t)stile * a.Npart + n) * PT + p] = S[p] + xch[nl * PT + p];
    }
}

// ------------------------------------------------------------------------------------------
// I8 hot path, large batches: 256 SVs x 256 samples per workgroup (512 threads, 8 waves as 2 x 4, wave tile 128 x 64 =
// 4 x 2 MFMA tiles of 32x32, 128 accumulator registers), K-step 128 B per row, two 64 KiB stages.
//
// Why: what a CU can pull from L2 into LDS is ~28 B/clk (measured: exp_l2bw, and the per-K-step cycle counters of the
// 128x128 kernel), and an i8 MFMA eats operand bytes twice as fast as bf16.  A 128x128 tile needs 64 B/clk per CU at full
// matrix rate (ceiling 44 %: measured 40-43 %); 256x256 needs 32 B/clk (ceiling ~87 %).  Round 1 measured this tile at
// 574 us vs 501 us on 8192 x 2560: that was the GRID, not the tile -- 32 x 10 = 320 workgroups on 256 CUs are two rounds
// with the second one a quarter full.  It is therefore only used when the launch has enough tiles for >= ~2.5 rounds
// (rml_project_svm sizes its chunks for it), and the 128x128 kernel keeps the small batches.
// Sample tiles are paired: tile_exact[] is evaluated per 256 samples (k_tile_flags group = 2); the partial slots 2*stile and
// 2*stile+1 carry the two 128-row halves, summed exactly like the 128x128 kernel sums them.
// ------------------------------------------------------------------------------------------
constexpr int kBig = 256;

// (The two-stage 64 KiB kernel that first ran this tile -- k_svm_gemm_i8_256, rounds 2-3: 0.46-0.50 of the int8 peak -- lost to the
// ring schedule below in every same-process A/B and left the tree in round 4; its numbers are in tools/exp/README.md.)

// ------------------------------------------------------------------------------------------
// k_svm_gemm_ring<PT, DIG>: the 256 x 256 tile with a 5-slot operand-stage ring and interleaved DMA issue (round 3).
//
// What limited the two-stage kernel that ran this tile in round 2 (A/B in one process, tools/gemm_ab.py, 16 384 x 2 562 x 20 480): all 64 DMA
// instructions of a stage leave in one burst after the barrier; the burst fills the CU's VMEM queue, every wave sits in its
// in-order issue stage until its eight instructions are accepted (~100 cycles each) and no MFMA is issued meanwhile:
// step = burst (~800-1000 cycles) + 2048 MFMA cycles.  Spreading the instructions between the MFMAs hides their issue under
// the SIMD partner's matrix work, but in a two-stage scheme a late issue is a late landing (measured slower in round 2).
// So the ring: the unit is one OPERAND stage (256 rows x 128 B = 32 KiB), all 160 KiB of LDS are ring, operand-stage n
// (n = 2t: SV rows of step t, 2t+1: sample rows of step t) lives in slot n % 5.  Step t sends the sample stage of step t+1 in
// its first half and the SV stage of step t+2 in its second half, one DMA instruction after every four MFMAs, into the
// two slots step t-1 has just released: every stage has 1 to 1.5 steps to land.  Waves wait with a counted
// s_waitcnt vmcnt(4) (everything but the newest stage) and meet at a raw s_barrier -- __syncthreads() would drain the queue.
// Measured: burst issue into the ring 0.99 ms (worse than two stages: 0.87), interleaved 0.80 ms = 2.15 PetaOP/s.
// The per-SV epilogue table is loaded after the K loop into slot 4 (the epilogue's G image takes slots 0-3).
//
// Tile order: block b runs on XCD b % 8 (observed, used for speed only).  The tiles are laid out as a sequence
// [group of 8 sample tiles][SV tile][sample tile of the group] and XCD x takes the x-th eighth of it (+-1 tile): the ~32
// tiles resident on an XCD are 8 sample tiles x 4 SV tiles sharing K-slices through that XCD's L2, and every XCD gets the
// same number of tiles whatever the batch (the previous map gave XCD x the sample tiles x, x+8, ...: 69 sample tiles ->
// nine on five XCDs, eight on three, i.e. a fourth round on five eighths of the chip).
//
// DIG = 1: general rows as four balanced int8 digits of a 32-bit fixed-point value (SURVEY 8 a-5 for data that is not on
// the code grid: train.py:496-517 augmentation, a non-unit proj_zoom of predict.py:109-116, the reference's generated_data
// pickles).  Every value v (sample feature or SV component) is read as u = (v - c0) / s in [-1, 1) (c0, s per model; s a
// power of two, so float32 inputs >= s 2^-8 are represented exactly and smaller ones to 2^-32 s), I = rint(u 2^31), split
// I = a0 2^24 + a1 2^16 + a2 2^8 + a3 with a_i in [-128, 127].  Then  u_x . u_s = 2^-14 sum_{i,j} 2^-8(i+j) (a_i^x . a_j^s)
// and every digit-plane product is an exact int32 GEMM (|.| <= 2^14 K < 2^29).  The ten pairs with i + j <= 3 are kept (the
// dropped ones weigh 2^-46 per digit product: typical 1e-8 on u.u), grouped by g = i + j and accumulated from
// the least significant group up IN THE SAME int32 accumulator: after g = 3 it is divided by 256 with rounding, (acc + 128) >> 8
// (2^-31 on u.u), after g = 2 it is split R2 = 256 q + r -- q stays, R2 is parked -- and the next group accumulates on top
// (4 x 2^28.3 < 2^31), which leaves R1 = G1 + floor(R2 / 2^8) and the remainder r for the epilogue.
// The top group G0 needs its own 32 bits (u.u = 2^-14 (G0 + R1/256) is a 38-bit quantity): it is computed FIRST and parked in a
// per-workgroup HBM scratch tile (256 KiB, written once, read once in the epilogue by the lane that wrote it: L2 traffic that
// is nothing next to ten K loops) -- a second accumulator set or packed remainders in registers pushed the kernel over 256
// VGPRs and the spills landed in the K loop.  u.u carries 2^-31 absolute precision on a quantity of magnitude <= D/4 -- the
// arithmetic class of the float64 path at ~3x its rate.  d^2 = s^2 (||u_x||^2 + ||u_s||^2 - 2 u_x.u_s) with the norms of the QUANTISED
// values in float64.  The K loop runs over (pair, K-step); the DMA cursors run one and two steps ahead across pair boundaries.
// ------------------------------------------------------------------------------------------
constexpr int kOpStageBytes = kBig * kStepBytes;          // 32 KiB
constexpr int kRingSlots = 5;
constexpr int kDigPairs = 10;
constexpr uint64_t kDigI = 0x1021032100ull;     // nibble p: sample digit of pair p   (0 = most significant)
constexpr uint64_t kDigJ = 0x0101201230ull;     // nibble p: SV digit of pair p; pair 0: g = 0, 1-4: g = 3, 5-7: g = 2, 8-9: g = 1
constexpr int kDigStashBytes = kBig * kBig * 4;  // the parked top-group accumulators of one workgroup

struct RingArgs {
    const uint8_t* sv; const uint8_t* x;       // operand bases (digit plane 0)
    int64_t ld_sv, ld_x;                       // bytes per row
    int64_t sv_plane, x_plane;                 // DIG: bytes between digit planes
    int KT;                                    // K-steps (per digit pair)
    int64_t N, Mpad, sv_rows; int ST, FT;      // ST / FT count 128-row tiles like GemmArgs
    const int32_t* tile_exact; int want;
    const int32_t* x_isum; const int64_t* x_isq;   // exact path row statistics
    const double* x_nsq;                            // DIG: ||u_x||^2
    const double* sv_term;                          // exact: per-SV term; DIG: ||u_s||^2
    const double* W;
    double gs; int kernel;
    double* partial; int64_t Npart;
    int32_t* stash;                            // DIG: gridDim.x * 2 * 256 KiB of scratch: the top digit group and the g = 2 level
};

// ring tile of block b: false = nothing to do
__device__ __forceinline__ bool ring_tile(int b, int FT2, int ST2, int& ftile, int& stile) {
    const int T = FT2 * ST2, q = T >> 3, r = T & 7;
    const int xcd = b & 7, k = b >> 3;
    if (k >= q + (xcd < r ? 1 : 0)) return false;
    const int pos = xcd * q + (xcd < r ? xcd : r) + k;
    const int G = 8 * ST2, fg = pos / G, rem = pos - fg * G;
    const int left = FT2 - 8 * fg, scnt = left < 8 ? left : 8;
    stile = rem / scnt;
    ftile = fg * 8 + (rem - stile * scnt);
    return true;
}
inline unsigned ring_grid(int FT2, int ST2) { const int T = FT2 * ST2; return (unsigned)(8 * ((T + 7) / 8)); }

template <int PT, int DIG>
__global__ __launch_bounds__(512, 2) void k_svm_gemm_ring(RingArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;           // 2 x 4 waves; wave tile = 128 SVs x 64 samples
    const int FT2 = (a.FT + 1) >> 1, ST2 = (int)((a.Mpad + kBig - 1) / kBig);
    int ftile, stile;
    if (!ring_tile(blockIdx.x, FT2, ST2, ftile, stile)) return;
    if (a.tile_exact && a.tile_exact[2 * ftile] != a.want) return;
    const int64_t f0 = (int64_t)ftile * kBig;
    const int64_t m0 = (int64_t)stile * kBig;

    const uint8_t* gsv[4];
    const uint8_t* gx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int s = (wave * 4 + q) * 64 + lane;
        int r = s >> 3;
        int c = (s & 7) ^ ((r >> 1) & 7);            // inverse swizzle on the source
        int64_t mr = m0 + r; mr = mr < a.sv_rows ? mr : a.sv_rows - 1;
        gsv[q] = a.sv + mr * a.ld_sv + c * 16;
        int64_t xr = f0 + r; xr = xr < a.N ? xr : a.N - 1;
        gx[q] = a.x + xr * a.ld_x + c * 16;
    }
    // DMA cursors: byte offset (digit plane + K-step) of the next sample stage / SV stage to send, and their ring slots
    const int64_t kbytes = (int64_t)a.KT * kStepBytes;
    int64_t ox = 0, os = 0;                            // offsets within the current pair
    int64_t px = 0, ps = 0;                            // plane offsets of the current pair
    int qx = 0, qs = 0;                                // pair indices (DIG)
    if constexpr (DIG) { px = (int64_t)(kDigI & 15) * a.x_plane; ps = (int64_t)(kDigJ & 15) * a.sv_plane; }
    auto adv_x = [&]() __attribute__((always_inline)) {
        ox += kStepBytes;
        if constexpr (DIG) { if (ox == kbytes) { ox = 0; ++qx; px = (int64_t)((kDigI >> (4 * qx)) & 15) * a.x_plane; } }
    };
    auto adv_s = [&]() __attribute__((always_inline)) {
        os += kStepBytes;
        if constexpr (DIG) { if (os == kbytes) { os = 0; ++qs; ps = (int64_t)((kDigJ >> (4 * qs)) & 15) * a.sv_plane; } }
    };
    auto burst_s = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(gsv[q] + ps + os, smem + slot * kOpStageBytes + wave * 4096 + q * 1024);
        adv_s();
    };
    auto burst_x = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(gx[q] + px + ox, smem + slot * kOpStageBytes + wave * 4096 + q * 1024);
        adv_x();
    };

    int aoff[4], asw[4], boff[2], bsw[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int ra = wr * 128 + t * 32 + (lane & 31);
        aoff[t] = ra * kStepBytes; asw[t] = (ra >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int rb = wc * 64 + t * 32 + (lane & 31);
        boff[t] = rb * kStepBytes; bsw[t] = (rb >> 1) & 7;
    }
    const int chalf = lane >> 5;

    v16i acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    const int TT = (DIG ? kDigPairs : 1) * a.KT;       // steps in all
    burst_s(0); burst_x(1);                            // SV_0, X_0
    if (TT > 1) burst_s(2);                            // SV_1
    int sa = 0;                                        // slot of the SV stage of the step being computed
    int sx = 3, ss = 4;                                // slots of the two stages step 0 sends
    int t = 0;                                         // global step
    // Fragment registers: two sets.  The loop is ROTATED by one MFMA group: the last sub-step (kk = 3) of a step is issued
    // after the next step's barrier, right behind that step's first fragment reads.  After a barrier all eight waves read
    // fragments at once (128 KiB per step through a 256 B/clk LDS = ~500 cycles in which, unrotated, no wave has an MFMA to
    // issue); the deferred group is matrix work that needs no LDS.
    v4i af[2][4], bf[2][2];
    auto mfma_half = [&](int set, int half) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 2 * half; i < 2 * half + 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[set][i], bf[set][j], acc[i][j], 0, 0, 0);
    };
    auto step = [&](auto first_) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_)::value;      // first step of a segment: no deferred group pending
        const bool hx = t + 1 < TT, hs = t + 2 < TT;
        // stages 2t and 2t+1 landed: everything this wave sent except the newest stage (the SV stage of step t+1); and this
        // wave's fragment reads of step t-1 are complete (their slots are released at the barrier)
        if (hx) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // ... for every wave's pieces; and step t-1 is consumed by everyone
        asm volatile("" ::: "memory");
        const int sbx = sa + 1 == kRingSlots ? 0 : sa + 1;
        const unsigned char* pa = smem + sa * kOpStageBytes;
        const unsigned char* pb = smem + sbx * kOpStageBytes;
        sa = sa + 2 >= kRingSlots ? sa + 2 - kRingSlots : sa + 2;
        unsigned char* dx = smem + sx * kOpStageBytes + wave * 4096;
        unsigned char* dsv = smem + ss * kOpStageBytes + wave * 4096;
        sx = sx + 2 >= kRingSlots ? sx + 2 - kRingSlots : sx + 2;
        ss = ss + 2 >= kRingSlots ? ss + 2 - kRingSlots : ss + 2;
        const int64_t offx = px + ox, offs = ps + os;
        auto reads = [&](int set, int kk) __attribute__((always_inline)) {
            const int ch = 2 * kk + chalf;
#pragma unroll
            for (int u = 0; u < 4; ++u) af[set][u] = *reinterpret_cast<const v4i*>(pa + aoff[u] + ((ch ^ asw[u]) << 4));
#pragma unroll
            for (int u = 0; u < 2; ++u) bf[set][u] = *reinterpret_cast<const v4i*>(pb + boff[u] + ((ch ^ bsw[u]) << 4));
        };
        auto dma = [&](int g) __attribute__((always_inline)) {                         // DMA instruction 0..7 of this step
            if (g < 4) { if (hx) glds16(gx[g] + offx, dx + g * 1024); }
            else       { if (hs) glds16(gsv[g - 4] + offs, dsv + (g - 4) * 1024); }
        };
        // group r of the step: r = 0 is the deferred kk = 3 of the previous step (set 1), r = 1..3 are kk = 0..2 of this one
        reads(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int set = (r + 1) & 1;                // r = 0 -> set 1 (deferred), r = 1 -> set 0 (kk = 0), ...
            if (r >= 1) {
                reads(r & 1, r);                        // fragments of kk = r into the set the previous group has just used
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (!(FIRST && r == 0)) mfma_half(set, half);
                __builtin_amdgcn_sched_barrier(0);
                dma(2 * r + half);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (hx) adv_x();
        if (hs) adv_s();
        ++t;
    };
    auto flush = [&]() __attribute__((always_inline)) {                               // the deferred kk = 3 of a segment's last step
        __builtin_amdgcn_sched_barrier(0);
        mfma_half(1, 0); mfma_half(1, 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto segment = [&](int nsteps) __attribute__((always_inline)) {
        step(std::true_type{});
        for (int n = nsteps - 1; n > 0; --n) step(std::false_type{});
        flush();
    };

    // DIG: the top group's accumulators are parked in this workgroup's scratch tile, 16 bytes per lane and store, coalesced
    v4i* stash = nullptr;
    if constexpr (DIG) {
        stash = reinterpret_cast<v4i*>(a.stash) + (int64_t)blockIdx.x * (2 * kDigStashBytes / 16) + tid;
        segment(a.KT);                                 // pair 0: g = 0
        {
            v4i* sp = stash;                           // a running pointer: 32 hoisted 64-bit addresses would be spilled
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        v4i v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                        *sp = v;
                        sp += 512;
                        asm volatile("" : "+v"(sp));
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][4 * r4 + r] = 0;
                    }
        }
        // stores and DMA loads share the VM counter and may complete out of order with respect to each other: drain once, so
        // that the counted waits of the next segment see DMA instructions only
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // groups g = 3 (pairs 1-4), g = 2 (5-7), g = 1 (8-9).  After g = 3 the accumulator is divided by 256 with rounding (2^-31
        // on u.u); after g = 2 it is split R2 = 256 q + r: q stays and g = 1 accumulates on top, R2 itself is parked in the second
        // scratch tile so that the epilogue puts the remainder r back -- no rounding at the 2^-23 level
#pragma nounroll                                       // one copy of the step bodies for the three groups (instruction cache)
        for (int g = 0; g < 3; ++g) {
            segment((4 - g) * a.KT);
            if (g == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] + 128) >> 8;
            } else if (g == 1) {
                v4i* sp = stash + kDigStashBytes / 16;   // second tile of this workgroup
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            v4i v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2], acc[i][j][4 * r4 + 3]};
                            *sp = v;
                            sp += 512;
                            asm volatile("" : "+v"(sp));
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[i][j][4 * r4 + r] >>= 8;      // floor: the remainder comes back in the epilogue
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores and DMA loads share the VM counter (see above)
            }
        }
    } else {
        segment(a.KT);
    }

    // ---- fused float64 epilogue; the per-SV table is loaded now, into slot 4 ----
    const bool rbf = (a.kernel == RML_KERNEL_RBF);
    double* svw = reinterpret_cast<double*>(smem + 4 * kOpStageBytes);     // [256][1+PT] + exp table
    const double* etab = svw + kBig * (1 + PT);
    __syncthreads();                                   // every wave is done with the ring
    exp_tab_init(svw + kBig * (1 + PT), tid);
    for (int idx = tid; idx < kBig * (1 + PT); idx += 512) {
        int m = idx / (1 + PT), c = idx - m * (1 + PT);
        const bool in = m0 + m < a.Mpad;
        svw[idx] = !in ? 0.0 : ((c == 0) ? a.sv_term[m0 + m] : a.W[(int64_t)(c - 1) * a.Mpad + m0 + m]);
    }
    if constexpr (DIG) {
        // one 64-sample quarter (= one wave column wc) at a time through LDS as float64:
        // u.u = 2^-22 (256 G0 + R1), G0 from the scratch tile;  d^2 = s^2 (||u_x||^2 + ||u_s||^2 - 2 u.u);  a.gs = gamma s^2
        double* gd = reinterpret_cast<double*>(smem);      // [256 SVs][64 samples]
        const int nq = tid & 63, qd = tid >> 6;            // sample column of the quarter, SV group (32 rows)
        for (int pass = 0; pass < 4; ++pass) {
            __syncthreads();
            if (wc == pass) {
                const v4i* sp = stash;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const v4i g0 = *sp;
                            const v4i r2 = *(sp + kDigStashBytes / 16);
                            sp += 512;
                            asm volatile("" : "+v"(sp));
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const int r = 4 * r4 + rr;
                                const int ml = wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
                                const int nn = j * 32 + (lane & 31);
                                // 2^22 u.u = 256 G0 + (G1 + floor(R2 / 256)) + (R2 mod 256) / 256
                                gd[ml * 64 + nn] = ((double)g0[rr] * 256.0 + (double)acc[i][j][r] + (double)(r2[rr] & 255) * 0x1p-8) * 0x1p-22;
                            }
                        }
            }
            __syncthreads();
            const int64_t n = f0 + pass * 64 + nq;
            const int64_t nc = n < a.N ? n : a.N - 1;
            const double xt = a.x_nsq[nc];
            double S[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) S[p] = 0.0;
#pragma unroll 2
            for (int mm = 0; mm < 32; ++mm) {
                const int ml = qd * 32 + mm;
                const double* e = svw + ml * (1 + PT);
                double d2 = xt + e[0] - 2.0 * gd[ml * 64 + nq];
                d2 = d2 > 0.0 ? d2 : 0.0;
                const double kv = rml_exp_neg(-a.gs * d2, etab);
#pragma unroll
                for (int p = 0; p < PT; ++p) S[p] = fma(e[1 + p], kv, S[p]);
            }
            __syncthreads();                               // G quarter consumed: reuse its LDS for the exchange
            double* x8 = gd;                               // [8 groups][64][PT]
#pragma unroll
            for (int p = 0; p < PT; ++p) x8[(qd * 64 + nq) * PT + p] = S[p];
            __syncthreads();
            if (qd == 0 && n < a.N) {
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    double tsum = 0.0;
#pragma unroll
                    for (int g = 0; g < 8; ++g) tsum += x8[(g * 64 + nq) * PT + p];
                    a.partial[((int64_t)(2 * stile) * a.Npart + n) * PT + p] = tsum;
                    if (2 * stile + 1 < a.ST) a.partial[((int64_t)(2 * stile + 1) * a.Npart + n) * PT + p] = 0.0;
                }
            }
        }
    } else {
        int* gl = reinterpret_cast<int*>(smem);
        const int nl = tid & 127, h = tid >> 7;
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            if ((wc >> 1) == pass) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int ml = wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
                            const int nn = (wc & 1) * 64 + j * 32 + (lane & 31);
                            gl[ml * kTile + nn] = acc[i][j][r];
                        }
            }
            __syncthreads();
            const int64_t n = f0 + pass * kTile + nl;
            const int64_t nc = n < a.N ? n : a.N - 1;
            const double xt = rbf ? (double)(a.x_isq[nc] - 256 * (int64_t)a.x_isum[nc]) : 128.0 * (double)a.x_isum[nc];
            double S[PT];
#pragma unroll
            for (int p = 0; p < PT; ++p) S[p] = 0.0;
            const int* gcol = gl + nl;
#pragma unroll 2
            for (int mm = 0; mm < 64; ++mm) {
                const int ml = h * 64 + mm;
                const double* e = svw + ml * (1 + PT);
                const double g = (double)gcol[ml * kTile];
                double kv;
                if (rbf) {
                    double d2 = xt + e[0] - 2.0 * g;
                    d2 = d2 > 0.0 ? d2 : 0.0;
                    kv = rml_exp_neg(-a.gs * d2, etab);
                } else {
                    kv = (g + xt + e[0]) * a.gs;
                }
#pragma unroll
                for (int p = 0; p < PT; ++p) S[p] = fma(e[1 + p], kv, S[p]);
            }
            __syncthreads();
            double* x4 = reinterpret_cast<double*>(smem);  // [4][128][PT]
#pragma unroll
            for (int p = 0; p < PT; ++p) x4[(h * kTile + nl) * PT + p] = S[p];
            __syncthreads();
            if (h == 0 && n < a.N) {
#pragma unroll
                for (int p = 0; p < PT; ++p) {
                    // one partial per 128 SV rows, each the sum of two 64-row in-lane chains: the very values, in the very order,
                    // the 128 x 128 kernel writes for these rows -- decision values do not depend on which kernel ran
                    a.partial[((int64_t)(2 * stile) * a.Npart + n) * PT + p] = x4[(0 * kTile + nl) * PT + p] + x4[(1 * kTile + nl) * PT + p];
                    if (2 * stile + 1 < a.ST)
                        a.partial[((int64_t)(2 * stile + 1) * a.Npart + n) * PT + p] = x4[(2 * kTile + nl) * PT + p] + x4[(3 * kTile + nl) * PT + p];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Single observations (round 6): the exact path for a handful of rows.
//
// The reference classifies ONE observation per call (predict.py:98-119).  A 128 x 128 tile kernel then launches one workgroup per
// 128 support vectors -- 21 workgroups, each streaming 2.6 MB of SV codes through one CU: 100-103 us per call
// (profiles/r06_stats_latency.txt), a twelfth of the machine.  For n <= RML_SMALL_FRAMES rows the work is a matrix-VECTOR product,
// bound by reading the SV codes once (52 MB at M = 2 562, D = 20 480): k_svm_dot_small gives every 8 SV rows a workgroup
// (Mpad / 8 = 336 of them), a thread 16 bytes of K per step, v_dot4_i32_i8 on the biased codes (the very int32 the MFMA path
// accumulates: exact, so the order does not matter), one wave reduction per (SV row, sample); k_svm_epi_small then evaluates the
// kernel values of a 128-SV tile in parallel and adds them up EXACTLY as the tile kernels do -- two chains of 64 support vectors in
// ascending order, fma(W, K, S), partial = chain 0 + chain 1 -- so decision values do not depend on which path ran (asserted:
// tests/test_svm_gpu.py::test_single_observations_take_the_small_path_with_the_same_bits).
// ------------------------------------------------------------------------------------------
struct SmallArgs {
    const uint8_t* sv; int64_t ld_sv;          // biased SV codes
    const uint8_t* x; int64_t ld_x;            // biased sample codes
    int64_t Kb;                                // bytes of K per row (a multiple of 128; pad bytes are 0 on both sides)
    int N; int64_t Mpad;
    const int32_t* tile_exact;                 // run iff NULL or tile_exact[0] == 1 (n <= 128: one sample tile)
    int32_t* G;                                // biased dot products of (SV m, sample n) at G[m * g_sm + n * g_sn]
    int64_t g_sm, g_sn;
    const int32_t* x_isum; const int64_t* x_isq;
    const double* sv_term; const double* W;
    double gs; int kernel;
    double* partial; int64_t Npart;
};

__device__ __forceinline__ int wave_sum_i32(int r) {
    auto mv = [](int v, auto ctrl, auto rowmask, auto bound) {
        return __builtin_amdgcn_update_dpp(0, v, decltype(ctrl)::value, decltype(rowmask)::value, 0xF, decltype(bound)::value);
    };
    r += mv(r, std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xF>{}, std::true_type{});     // quad_perm [1,0,3,2]
    r += mv(r, std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xF>{}, std::true_type{});     // quad_perm [2,3,0,1]
    r += mv(r, std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xF>{}, std::true_type{});    // row_half_mirror
    r += mv(r, std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xF>{}, std::true_type{});    // row_mirror
    r += mv(r, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{}, std::false_type{});   // row_bcast15 into rows 1, 3
    r += mv(r, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{}, std::false_type{});   // row_bcast31 into rows 2, 3
    return __builtin_amdgcn_readlane(r, 63);
}

constexpr int kSmallSv = 8;                    // SV rows per workgroup of k_svm_dot_small

template <int NS>
__global__ __launch_bounds__(256) void k_svm_dot_small(SmallArgs a) {
    if (a.tile_exact && a.tile_exact[0] != 1) return;
    __shared__ int red[4][kSmallSv * NS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * kSmallSv;
    int acc[kSmallSv][NS];
#pragma unroll
    for (int r = 0; r < kSmallSv; ++r)
#pragma unroll
        for (int n = 0; n < NS; ++n) acc[r][n] = 0;
    const uint8_t* __restrict__ xr[NS];
#pragma unroll
    for (int n = 0; n < NS; ++n) xr[n] = a.x + (int64_t)(n < a.N ? n : a.N - 1) * a.ld_x;
    const uint8_t* __restrict__ svr = a.sv + m0 * a.ld_sv;
#pragma unroll 2
    for (int64_t off = (int64_t)tid * 16; off < a.Kb; off += 256 * 16) {
        v4i xs[NS], ss[kSmallSv];
#pragma unroll
        for (int r = 0; r < kSmallSv; ++r) ss[r] = *reinterpret_cast<const v4i*>(svr + r * a.ld_sv + off);
#pragma unroll
        for (int n = 0; n < NS; ++n) xs[n] = *reinterpret_cast<const v4i*>(xr[n] + off);
#pragma unroll
        for (int r = 0; r < kSmallSv; ++r)
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                int t = acc[r][n];
                t = __builtin_amdgcn_sdot4(ss[r].x, xs[n].x, t, false);
                t = __builtin_amdgcn_sdot4(ss[r].y, xs[n].y, t, false);
                t = __builtin_amdgcn_sdot4(ss[r].z, xs[n].z, t, false);
                t = __builtin_amdgcn_sdot4(ss[r].w, xs[n].w, t, false);
                acc[r][n] = t;
            }
    }
#pragma unroll
    for (int r = 0; r < kSmallSv; ++r)
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            const int v = wave_sum_i32(acc[r][n]);
            if (lane == 0) red[wave][r * NS + n] = v;
        }
    __syncthreads();
    if (tid < kSmallSv * NS) {
        const int r = tid / NS, n = tid - r * NS;
        if (n < a.N) a.G[(m0 + r) * a.g_sm + (int64_t)n * a.g_sn] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

template <int PT>
__global__ __launch_bounds__(128) void k_svm_epi_small(SmallArgs a) {
    if (a.tile_exact && a.tile_exact[blockIdx.y / kTile] != 1) return;
    __shared__ double etab[64];
    __shared__ double kvs[kTile];
    __shared__ double wl[PT][kTile];
    __shared__ double xch[PT];
    const int tid = threadIdx.x, stile = blockIdx.x, n = blockIdx.y;
    exp_tab_init(etab, tid);
    const int64_t m = (int64_t)stile * kTile + tid;
#pragma unroll
    for (int p = 0; p < PT; ++p) wl[p][tid] = a.W[(int64_t)p * a.Mpad + m];
    __syncthreads();
    const bool rbf = (a.kernel == RML_KERNEL_RBF);
    // the arithmetic of the tile kernels' epilogue, value for value (k_svm_gemm<I8>)
    const double xt = rbf ? (double)(a.x_isq[n] - 256 * (int64_t)a.x_isum[n]) : 128.0 * (double)a.x_isum[n];
    const double g = (double)a.G[m * a.g_sm + (int64_t)n * a.g_sn];
    const double e0 = a.sv_term[m];
    double kv;
    if (rbf) {
        double d2 = xt + e0 - 2.0 * g;
        d2 = d2 > 0.0 ? d2 : 0.0;
        kv = rml_exp_neg(-a.gs * d2, etab);
    } else {
        kv = (g + xt + e0) * a.gs;
    }
    kvs[tid] = kv;
    __syncthreads();
    if (tid < 2) {                                  // the two 64-row chains of the tile, in the tile kernels' order
        double S[PT];
#pragma unroll
        for (int p = 0; p < PT; ++p) S[p] = 0.0;
        for (int mm = 0; mm < 64; ++mm) {
            const int ml = tid * 64 + mm;
            const double k = kvs[ml];
#pragma unroll
            for (int p = 0; p < PT; ++p) S[p] = fma(wl[p][ml], k, S[p]);
        }
        if (tid == 1) {
#pragma unroll
            for (int p = 0; p < PT; ++p) xch[p] = S[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // both chains live in one wave
        if (tid == 0) {
#pragma unroll
            for (int p = 0; p < PT; ++p) a.partial[((int64_t)stile * a.Npart + n) * PT + p] = S[p] + xch[p];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Batches that do not fill the machine with 128 x 128 tiles (9 .. ~1 500 rows: what train.py's `clf.predict(X_test)` and a
// few dozen observations look like): FT x ST tiles are 21 .. 250 workgroups on 256 CUs, each walking all 160 K-steps -- 104 us
// for 64 rows, whatever their number.  The int32 dot products are EXACT, so K may be cut anywhere and the pieces added in any
// order: k_svm_gemm_splitk gives every (tile, K range) a workgroup -- the tile kernel's staging and MFMA loop over its range -- and
// adds its accumulators into G[m][n] with int32 atomics (lanes run along n: whole 128-byte requests); k_svm_epi_small then forms the
// kernel values and the partial sums in the tile kernels' order.  Bit-identical decision values, asserted with the small path.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_zero16(v4i* p, int64_t n16) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) p[i] = v4i{0, 0, 0, 0};
}

struct SplitArgs {
    const uint8_t* sv; int64_t ld_sv;
    const uint8_t* x; int64_t ld_x;
    int KT, per;                               // K-steps in all, per K range
    int64_t N; int FT;
    const int32_t* tile_exact;
    int32_t* G; int64_t ldg;                   // [Mpad][ldg] (zeroed by the caller), ldg = FT * 128
};

__global__ __launch_bounds__(256, 2) void k_svm_gemm_splitk(SplitArgs a) {
    __shared__ __align__(16) unsigned char smem[4 * kTileBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int ftile = blockIdx.x % a.FT, stile = blockIdx.x / a.FT;
    if (a.tile_exact && a.tile_exact[ftile] != 1) return;
    const int kt0 = blockIdx.y * a.per;
    const int kt1 = a.KT < kt0 + a.per ? a.KT : kt0 + a.per;
    if (kt0 >= kt1) return;
    const int64_t f0 = (int64_t)ftile * kTile, m0 = (int64_t)stile * kTile;
    const uint8_t* gsv[4];
    const uint8_t* gx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int s = (wave * 4 + q) * 64 + lane;          // 16-byte slot in the LDS image
        int r = s >> 3;
        int c = (s & 7) ^ ((r >> 1) & 7);            // inverse swizzle on the source
        gsv[q] = a.sv + (m0 + r) * a.ld_sv + c * 16;
        int64_t xr = f0 + r; xr = xr < a.N ? xr : a.N - 1;
        gx[q] = a.x + xr * a.ld_x + c * 16;
    }
    auto stage = [&](int kt, int buf) {
        unsigned char* base = smem + buf * 2 * kTileBytes;
        const int64_t ko = (int64_t)kt * kStepBytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16(gsv[q] + ko, base + (wave * 4 + q) * 1024);
            glds16(gx[q] + ko, base + kTileBytes + (wave * 4 + q) * 1024);
        }
    };
    int aoff[2], asw[2], boff[2], bsw[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int ra = wr * 64 + t * 32 + (lane & 31);
        int rb = wc * 64 + t * 32 + (lane & 31);
        aoff[t] = ra * kStepBytes; asw[t] = (ra >> 1) & 7;
        boff[t] = rb * kStepBytes; bsw[t] = (rb >> 1) & 7;
    }
    const int chalf = lane >> 5;
    v16i acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    stage(kt0, 0);
    for (int kt = kt0; kt < kt1; ++kt) {
        const int b = (kt - kt0) & 1;
        __syncthreads();                       // DMA of step kt landed (vmcnt(0)) and visible
        if (kt + 1 < kt1) stage(kt + 1, b ^ 1);
        const unsigned char* sA = smem + b * 2 * kTileBytes;
        const unsigned char* sB = sA + kTileBytes;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int ch = 2 * kk + chalf;
            v4i af[2], bf[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = *reinterpret_cast<const v4i*>(sA + aoff[t] + ((ch ^ asw[t]) << 4));
                bf[t] = *reinterpret_cast<const v4i*>(sB + boff[t] + ((ch ^ bsw[t]) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    // D[row of A = SV (r & 3) + 8 (r >> 2) + 4 chalf][column of B = sample lane & 31]: the lanes of an atomic run along n
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * chalf;
                const int nl = wc * 64 + j * 32 + (lane & 31);
                __hip_atomic_fetch_add(a.G + (m0 + ml) * a.ldg + f0 + nl, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
}

// digit planes of float32 rows: one workgroup per row, a thread takes 4 consecutive features per step.
// ok[row] = every feature is finite and inside the model's fixed-point range.
__global__ __launch_bounds__(256) void k_digit_rows(const float* f32, int64_t ld, int64_t D, int64_t Dq, int64_t plane, int8_t* dig,
                                                    double* nsq, int32_t* ok, double c0, double k31, const int32_t* skip_if_set) {
    if (skip_if_set && *skip_if_set) return;
    __shared__ double redn[4];
    __shared__ int redo[4];
    const int64_t b = blockIdx.x;
    double nn = 0.0; int good = 1;
    for (int64_t i4 = (int64_t)threadIdx.x * 4; i4 < Dq; i4 += 1024) {
        uint32_t pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t idx = i4 + e;
            uint32_t packed = 0;
            if (idx < D) {
                const double t = rint(((double)f32[b * ld + idx] - c0) * k31);
                const bool in = t >= -2147483648.0 && t <= 2139062143.0;       // NaN fails both
                good &= in ? 1 : 0;
                const int I = in ? (int)t : 0;
                const double u = (double)I * 0x1p-31;
                nn = fma(u, u, nn);
                packed = ((uint32_t)I + 0x00808080u) ^ 0x00808080u;           // bytes = balanced digits a0 (top) .. a3
            }
            pk[e] = packed;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int sh = 8 * (3 - d);
            const uint32_t w = ((pk[0] >> sh) & 255u) | (((pk[1] >> sh) & 255u) << 8) | (((pk[2] >> sh) & 255u) << 16) | (((pk[3] >> sh) & 255u) << 24);
            *reinterpret_cast<uint32_t*>(dig + (int64_t)d * plane + b * Dq + i4) = w;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { nn += __shfl_xor(nn, off); good &= __shfl_xor(good, off); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { redn[wave] = nn; redo[wave] = good; }
    __syncthreads();
    if (threadIdx.x == 0) {
        nsq[b] = (redn[0] + redn[1]) + (redn[2] + redn[3]);
        ok[b] = redo[0] & redo[1] & redo[2] & redo[3];
    }
}

// ---- row preparation for callers that bring float32 feature rows -------------------------
// One workgroup per row: zero-padded float copy (ld = Df), float64 norm, codes + statistics.
__global__ __launch_bounds__(256) void k_prepare_rows(const float* feat, int64_t ld, int64_t D, float code_scale,
                                                      float* f32, int64_t Df, double* nsq,
                                                      uint8_t* q, int64_t Dq, int32_t* isum, int64_t* isq, int32_t* flags) {
    __shared__ int64_t red[16];
    const int64_t b = blockIdx.x;
    const bool scaled = code_scale > 1.0f;
    int32_t s = 0; int64_t sq = 0; int ok = 1; double nn = 0.0;
    // eight of the thread's values in flight at a time (the loop used to wait out a memory latency per value: its stores may alias
    // its loads as far as the compiler knows -- 44 us for ONE row of 20 480 values, most of a predict.py:60 call); the values are
    // consumed in the same ascending order, so every sum is the one it was
    const int64_t lim = Dq > Df ? Dq : Df;
    const float* __restrict__ src = feat + b * ld;
    for (int64_t base = threadIdx.x; base < lim; base += 256 * 8) {
        float vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t idx = base + (int64_t)u * 256;
            vv[u] = idx < D ? src[idx] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t idx = base + (int64_t)u * 256;
            if (idx >= lim) break;
            const float v = vv[u];
            if (idx < Df) f32[b * Df + idx] = v;
            nn += (double)v * (double)v;
            if (q && idx < Dq) {
                uint8_t code = 0;
                if (idx < D) {
                    float c = rintf(scaled ? v * code_scale : v);
                    float back = scaled ? __fdiv_rn(c, code_scale) : c;
                    bool good = (back == v) && c >= 0.0f && c <= 255.0f;
                    int ci = good ? (int)c : 0;
                    ok &= good ? 1 : 0;
                    s += ci; sq += (int64_t)(ci * ci);
                    code = (uint8_t)(ci ^ 0x80);
                }
                q[b * Dq + idx] = code;
            }
        }
    }
    int64_t s64 = s;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s64 += __shfl_xor(s64, off); sq += __shfl_xor(sq, off); ok &= __shfl_xor(ok, off); nn += __shfl_xor(nn, off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* redd = reinterpret_cast<double*>(red + 12);
    if (lane == 0) { red[wave * 3] = s64; red[wave * 3 + 1] = sq; red[wave * 3 + 2] = ok; redd[wave] = nn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t S = 0, Q = 0, G = 1; double NN = 0;
        for (int w = 0; w < 4; ++w) { S += red[w * 3]; Q += red[w * 3 + 1]; G &= red[w * 3 + 2]; NN += redd[w]; }
        if (isum) isum[b] = (int32_t)S;
        if (isq) isq[b] = Q;
        if (flags) flags[b] = q ? (int32_t)G : 0;
        nsq[b] = NN;
    }
}

// tile_exact[ft] = policy(flags of the 128 rows of tile ft); *all_exact = AND over tiles.
// One 128-thread block per tile, one row flag per thread; with all_exact, ONE MORE block that scans every row flag and writes the AND
// (round 3 pre-set the word with a one-thread kernel and cleared it with atomics: a launch and its gap per chunk on the stream whose
// chain is the period of the Walabot pipeline).
// group = sample tiles decided together (2 when the 256-sample GEMM kernel takes the exact tiles): blockDim = group * 128.
__global__ __launch_bounds__(256) void k_tile_flags(const int32_t* flags, int64_t N, int FT, int policy /*0 auto,1 force general,2 force i8*/,
                                                    int model_exact, int32_t* tile_exact, int32_t* all_exact, int group) {
    const int tile_blocks = (FT + group - 1) / group;
    if ((int)blockIdx.x >= tile_blocks) {
        int mine = 1;
        if (policy == 1) mine = 0;
        else if (policy != 2) {
            if (!(model_exact && flags != nullptr)) mine = 0;
            else if ((reinterpret_cast<uintptr_t>(flags) & 15) == 0) {
                // eight independent 16-byte loads per thread and trip (a plain `mine &= flags[r]` loop waits out a memory latency per
                // flag: 65-75 us for 8 192 rows in the kernel timeline of session r4aq -- on the stream whose chain is the period)
                const int4* f4 = reinterpret_cast<const int4*>(flags);
                const int64_t n4 = N >> 2;
                for (int64_t i = threadIdx.x; i < n4; i += (int64_t)blockDim.x * 8) {
                    int4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int64_t idx = i + (int64_t)u * blockDim.x;
                        v[u] = idx < n4 ? f4[idx] : make_int4(1, 1, 1, 1);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) mine &= (v[u].x != 0) & (v[u].y != 0) & (v[u].z != 0) & (v[u].w != 0);
                }
                for (int64_t r = (n4 << 2) + threadIdx.x; r < N; r += blockDim.x) mine &= flags[r] != 0;
            } else {
                for (int64_t r = threadIdx.x; r < N; r += blockDim.x) mine &= flags[r] != 0;
            }
        }
        const int e = __syncthreads_and(mine);
        if (threadIdx.x == 0) *all_exact = e;
        return;
    }
    const int ft0 = blockIdx.x * group;
    int e;
    if (policy == 1) e = 0;
    else if (policy == 2) e = 1;
    else {
        const int64_t r = (int64_t)ft0 * kTile + threadIdx.x;
        int mine = (model_exact && flags != nullptr) ? ((r < N) ? (flags[r] != 0) : 1) : 0;
        e = __syncthreads_and(mine);
    }
    if (threadIdx.x == 0)
        for (int g = 0; g < group; ++g) if (ft0 + g < FT) tile_exact[ft0 + g] = e;
}

// second decision, once the digit planes of a chunk exist: a tile group that is not on the code grid (tile_exact == 0) goes
// to the multi-digit int8 GEMM (tile_exact = 2) when every one of its rows fits the model's fixed-point range
__global__ __launch_bounds__(256) void k_tile_dig(const int32_t* dflags, int64_t N, int FT, int32_t* tile_exact, const int32_t* skip_if_set) {
    if (skip_if_set && *skip_if_set) return;
    const int ft0 = blockIdx.x * 2;
    const int64_t r = (int64_t)ft0 * kTile + threadIdx.x;
    const int mine = (r < N) ? (dflags[r] != 0) : 1;
    const int e = __syncthreads_and(mine);
    if (threadIdx.x == 0 && e && tile_exact[ft0] == 0) {
        tile_exact[ft0] = 2;
        if (ft0 + 1 < FT) tile_exact[ft0 + 1] = 2;
    }
}

// the multi-digit kernel for the general tiles of a chunk of n rows?  RML_DIGITS = 0 never, 1 always (when the model allows),
// default: when the launch has tiles for at least half a round of one workgroup per CU (below that the float64 MFMA kernel's
// 128 x 128 tiles fill the machine better)
inline bool use_dig_gemm(const rml_svm* m, int policy, int64_t n, int num_cu) {
    if (!m->dig_ok) return false;
    if (policy == RML_PATH_DIGITS) return true;
    if (policy != RML_PATH_AUTO) return false;
    const int64_t wgs = ((n + kBig - 1) / kBig) * ((m->Mpad + kBig - 1) / kBig);
    return wgs * 2 >= (int64_t)num_cu;
}

// does the 256x256 ring kernel take the exact tiles of a chunk of n rows against this model?  It runs at up to 0.6 of the int8
// peak when its tiles fill whole rounds of one workgroup per CU and proportionally less otherwise; the 128x128 kernel (two
// workgroups per CU, four times as many tiles) sits at 0.40-0.46 whatever the batch.  So: at least three quarters of a round,
// and the last round at least three quarters full.  knob = RML_OPT_GEMM_BIG: 0 turns it off, 1 forces it for every n >= 256.
inline bool use_big_gemm(const rml_svm* m, int64_t n, int num_cu, int knob) {
    if (knob == 0 || m->PT > 6) return false;
    if (knob == 1) return n >= 256;
    const int64_t wgs = ((n + kBig - 1) / kBig) * ((m->Mpad + kBig - 1) / kBig);
    const int64_t rounds = (wgs + num_cu - 1) / num_cu;
    return wgs * 4 >= (int64_t)num_cu * 3 && wgs * 4 >= rounds * num_cu * 3;
}


// ---- finishing kernel: fixed-order sum of the SV-tile partials + libsvm/sklearn tail ------
struct FinishArgs {
    const double* partial; int64_t Npart; int ST, PT;
    int64_t N; int C, P;
    const double* intercept; const double* calib; int has_calib;
    const int32_t* row_flags; const int32_t* tile_exact;   // forced-i8 validity (rows with flag 0 -> NaN)
    int forced_i8;
    double* dec_ovo; double* dec_ovr; double* proba; int32_t* label_vote; int32_t* label_calib;
};

__device__ __forceinline__ double expit_d(double x) {
    if (x >= 0.0) return 1.0 / (1.0 + exp(-x));
    double e = exp(x);
    return e / (1.0 + e);
}

// up to 6 classes (15 one-vs-one pairs): person / dog / cat plus the aliases of train.py:656-663 fit with room to spare
constexpr int kMaxC = 6, kMaxP = 15;

__global__ __launch_bounds__(256) void k_svm_finish(FinishArgs a) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const int C = a.C, P = a.P;
    double dec[kMaxP];
    for (int p = 0; p < P; ++p) {
        double s = 0.0;
        for (int st = 0; st < a.ST; ++st) s += a.partial[((int64_t)st * a.Npart + n) * a.PT + p];
        dec[p] = s + a.intercept[p];           // sum -= rho[p]  (rho = -intercept_)
    }
    bool valid = true;
    if (a.forced_i8 && a.row_flags) valid = a.row_flags[n] != 0;
    if (!valid) for (int p = 0; p < P; ++p) dec[p] = NAN;
    if (a.dec_ovo) for (int p = 0; p < P; ++p) a.dec_ovo[n * P + p] = dec[p];

    // libsvm vote: dec > 0 -> ++vote[i] else ++vote[j]; first maximum wins (svm.cpp:2884-2894)
    int vote[kMaxC];
    for (int c = 0; c < C; ++c) vote[c] = 0;
    {
        int p = 0;
        for (int i = 0; i < C; ++i)
            for (int j = i + 1; j < C; ++j, ++p) { if (dec[p] > 0) ++vote[i]; else ++vote[j]; }
    }
    int best = 0;
    for (int c = 1; c < C; ++c) if (vote[c] > vote[best]) best = c;
    if (a.label_vote) a.label_vote[n] = valid ? best : -1;

    double T[kMaxC];
    if (C == 2) {
        // sklearn flips the sign for binary problems (sk:svm/_base.py:546-547): T = -dec
        T[0] = -dec[0];
        if (a.dec_ovr) a.dec_ovr[n] = T[0];
    } else {
        // _ovr_decision_function(dec < 0, -dec, C)
        double soc[kMaxC]; double vt[kMaxC];
        for (int c = 0; c < C; ++c) { soc[c] = 0.0; vt[c] = 0.0; }
        int p = 0;
        for (int i = 0; i < C; ++i)
            for (int j = i + 1; j < C; ++j, ++p) {
                double conf = -dec[p];
                soc[i] -= conf; soc[j] += conf;
                if (dec[p] < 0) vt[j] += 1.0; else vt[i] += 1.0;
            }
        for (int c = 0; c < C; ++c) T[c] = vt[c] + soc[c] / (3.0 * (fabs(soc[c]) + 1.0));
        if (!valid) for (int c = 0; c < C; ++c) T[c] = NAN;
        if (a.dec_ovr) for (int c = 0; c < C; ++c) a.dec_ovr[n * C + c] = T[c];
    }
    if (a.has_calib && (a.proba || a.label_calib)) {
        double pr[kMaxC];
        if (C == 2) {
            pr[1] = expit_d(-(a.calib[0] * T[0] + a.calib[C + 0]));
            pr[0] = 1.0 - pr[1];
        } else {
            double den = 0.0;
            for (int c = 0; c < C; ++c) { pr[c] = expit_d(-(a.calib[c] * T[c] + a.calib[C + c])); den += pr[c]; }
            for (int c = 0; c < C; ++c) pr[c] = (den != 0.0) ? pr[c] / den : 1.0 / C;
        }
        for (int c = 0; c < C; ++c) if (pr[c] > 1.0 && pr[c] <= 1.0 + 1e-5) pr[c] = 1.0;
        if (a.proba) for (int c = 0; c < C; ++c) a.proba[n * C + c] = valid ? pr[c] : NAN;
        int bc = 0;
        for (int c = 1; c < C; ++c) if (pr[c] > pr[bc]) bc = c;
        if (a.label_calib) a.label_calib[n] = valid ? bc : -1;
    }
}

// ---- libsvm probability estimates (SVC(probability=True).predict_proba) ---------------------------------
// sigmoid_predict + multiclass_probability (method 2 of Wu, Lin & Weng) of sk:svm/src/libsvm/svm.cpp:2032-2104,
// 2918-2952, one thread per sample, float64, same iteration order as the C loops.
__global__ __launch_bounds__(256) void k_pairwise_proba(const double* dec, int64_t N, int k, const double* probA, const double* probB,
                                                        double* proba) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int P = k * (k - 1) / 2;
    double r[kMaxC][kMaxC], Q[kMaxC][kMaxC], p[kMaxC], Qp[kMaxC];
    int q = 0;
    for (int i = 0; i < k; ++i)
        for (int j = i + 1; j < k; ++j, ++q) {
            const double f = dec[n * P + q] * probA[q] + probB[q];
            double s = f >= 0 ? exp(-f) / (1.0 + exp(-f)) : 1.0 / (1.0 + exp(f));
            s = fmin(fmax(s, 1e-7), 1.0 - 1e-7);
            r[i][j] = s; r[j][i] = 1.0 - s;
        }
    for (int t = 0; t < k; ++t) {
        p[t] = 1.0 / k;
        Q[t][t] = 0.0;
        for (int j = 0; j < t; ++j) { Q[t][t] += r[j][t] * r[j][t]; Q[t][j] = Q[j][t]; }
        for (int j = t + 1; j < k; ++j) { Q[t][t] += r[j][t] * r[j][t]; Q[t][j] = -r[j][t] * r[t][j]; }
    }
    const double eps = 0.005 / k;
    const int max_iter = k > 100 ? k : 100;
    for (int iter = 0; iter < max_iter; ++iter) {
        double pQp = 0.0;
        for (int t = 0; t < k; ++t) {
            Qp[t] = 0.0;
            for (int j = 0; j < k; ++j) Qp[t] += Q[t][j] * p[j];
            pQp += p[t] * Qp[t];
        }
        double max_error = 0.0;
        for (int t = 0; t < k; ++t) max_error = fmax(max_error, fabs(Qp[t] - pQp));
        if (max_error < eps) break;
        for (int t = 0; t < k; ++t) {
            const double diff = (-Qp[t] + pQp) / Q[t][t];
            p[t] += diff;
            pQp = (pQp + diff * (diff * Q[t][t] + 2 * Qp[t])) / (1 + diff) / (1 + diff);
            for (int j = 0; j < k; ++j) { Qp[j] = (Qp[j] + diff * Q[t][j]) / (1 + diff); p[j] /= (1 + diff); }
        }
    }
    for (int t = 0; t < k; ++t) proba[n * k + t] = p[t];
}

// ---- linear classifier: one wave per row, float64 accumulation ----------------------------
__global__ __launch_bounds__(256) void k_linear(const float* feat, int64_t ld, int64_t N, int64_t D, int C,
                                                const double* coef, const double* intercept, const double* calib, int has_calib,
                                                double* dec, double* proba, int32_t* label, int32_t* label_calib) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    double s[kMaxC];
    for (int c = 0; c < C; ++c) s[c] = 0.0;
    for (int64_t d = lane; d < D; d += 64) {
        double x = (double)feat[n * ld + d];
        for (int c = 0; c < C; ++c) s[c] = fma(x, coef[c * D + d], s[c]);
    }
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[c] += __shfl_xor(s[c], off);
        s[c] += intercept[c];
    }
    if (lane != 0) return;
    if (C == 2) {
        // binary SGD: coef_ has one row (class 1 score)
        if (dec) dec[n] = s[0];
        if (label) label[n] = s[0] > 0 ? 1 : 0;
        if (has_calib) {
            double p1 = expit_d(-(calib[0] * s[0] + calib[C]));
            if (proba) { proba[n * 2] = 1.0 - p1; proba[n * 2 + 1] = p1; }
            if (label_calib) label_calib[n] = p1 > 1.0 - p1 ? 1 : 0;
        }
        return;
    }
    if (dec) for (int c = 0; c < C; ++c) dec[n * C + c] = s[c];
    int b = 0;
    for (int c = 1; c < C; ++c) if (s[c] > s[b]) b = c;
    if (label) label[n] = b;
    if (has_calib && (proba || label_calib)) {
        double pr[kMaxC]; double den = 0.0;
        for (int c = 0; c < C; ++c) { pr[c] = expit_d(-(calib[c] * s[c] + calib[C + c])); den += pr[c]; }
        for (int c = 0; c < C; ++c) pr[c] = (den != 0.0) ? pr[c] / den : 1.0 / C;
        for (int c = 0; c < C; ++c) if (pr[c] > 1.0 && pr[c] <= 1.0 + 1e-5) pr[c] = 1.0;
        if (proba) for (int c = 0; c < C; ++c) proba[n * C + c] = pr[c];
        int bc = 0;
        for (int c = 1; c < C; ++c) if (pr[c] > pr[bc]) bc = c;
        if (label_calib) label_calib[n] = bc;
    }
}

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

template <typename T> int dev_upload(T** dst, const std::vector<T>& h) {
    RML_HIP(hipMalloc(reinterpret_cast<void**>(dst), h.size() * sizeof(T)));
    RML_HIP(hipMemcpy(*dst, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return RML_OK;
}

template <int PATH, bool KM = false>
int launch_gemm(const rml_svm* m, const GemmArgs& ga, hipStream_t st) {
    const size_t lds = 4 * kTileBytes + (size_t)kTile * (1 + m->PT) * sizeof(double) + kExpTabBytes;
    const int FT8 = (int)round_up(ga.FT, 8);
    dim3 grid((unsigned)(FT8 * ga.ST)), block(256);
#define RML_GEMM_CASE(PTV)                                                                                         \
    case PTV: {                                                                                                    \
        RML_MAX_DYN_LDS(144 * 1024, &k_svm_gemm<PATH, PTV, KM>);                                                   \
        hipLaunchKernelGGL((k_svm_gemm<PATH, PTV, KM>), grid, block, lds, st, ga);                                 \
    } break;
    switch (m->PT) {
        RML_GEMM_CASE(1)
        RML_GEMM_CASE(3)
        RML_GEMM_CASE(6)
        RML_GEMM_CASE(10)
        RML_GEMM_CASE(15)
        default: RML_REQUIRE(false, RML_ERR_UNSUPPORTED, "svm: unsupported pair count");
    }
#undef RML_GEMM_CASE
    RML_HIP(hipGetLastError());
    return RML_OK;
}

template <int DIG>
int launch_gemm_ring(const rml_svm* m, const RingArgs& ra, hipStream_t st) {
    const int FT2 = (ra.FT + 1) / 2, ST2 = (int)((m->Mpad + kBig - 1) / kBig);
    dim3 grid(ring_grid(FT2, ST2)), block(512);
    const size_t lds = (size_t)kRingSlots * kOpStageBytes;
#define RML_RING_CASE(PTV)                                                                                         \
    case PTV: {                                                                                                    \
        RML_MAX_DYN_LDS(160 * 1024, &k_svm_gemm_ring<PTV, DIG>);                                                   \
        hipLaunchKernelGGL((k_svm_gemm_ring<PTV, DIG>), grid, block, lds, st, ra);                                 \
    } break;
    switch (m->PT) {
        RML_RING_CASE(1)
        RML_RING_CASE(3)
        RML_RING_CASE(6)
        default: RML_REQUIRE(false, RML_ERR_UNSUPPORTED, "svm: unsupported pair count for the 256x256 kernel");
    }
#undef RML_RING_CASE
    RML_HIP(hipGetLastError());
    return RML_OK;
}

int launch_gemm_big(const rml_svm* m, const GemmArgs& ga, hipStream_t st) {
    RingArgs ra{};
    ra.sv = ga.sv; ra.x = ga.x; ra.ld_sv = ga.ld_sv; ra.ld_x = ga.ld_x; ra.KT = ga.KT;
    ra.N = ga.N; ra.Mpad = ga.Mpad; ra.sv_rows = ga.sv_rows; ra.ST = ga.ST; ra.FT = ga.FT;
    ra.tile_exact = ga.tile_exact; ra.want = ga.want; ra.x_isum = ga.x_isum; ra.x_isq = ga.x_isq;
    ra.sv_term = ga.sv_term; ra.W = ga.W; ra.gs = ga.gs; ra.kernel = ga.kernel; ra.partial = ga.partial; ra.Npart = ga.Npart;
    return launch_gemm_ring<0>(m, ra, st);
}

// Rows per chunk of the chunked front doors.  Where the 256x256 ring kernel runs (an exact model, or the multi-digit path) a
// chunk costs ceil(tiles / CUs) rounds of one workgroup per CU, so the chunk size is chosen for the WHOLE batch: among the
// multiples of 256 rows in 4096..32768 the one with the fewest rounds in total (full chunks + the remainder) plus a small charge
// per chunk, the larger chunk on a tie.  Otherwise `fallback`.  RML_OPT_CHUNK overrides.
int64_t pick_chunk_opt(const rml_ctx* ctx, int64_t fallback) {
    const int64_t v = ctx->opt.chunk;
    return v >= 128 ? round_up(v, kTile) : fallback;
}

int64_t pick_chunk(const rml_ctx* ctx, const rml_svm* m, int64_t rows, int64_t fallback, int num_cu, bool dig = false) {
    const int64_t env = pick_chunk_opt(ctx, 0);
    int64_t ch = fallback;
    if (env) ch = env;
    else if (dig || (m->exact && use_big_gemm(m, 32768, num_cu, ctx->opt.gemm_big))) {
        const int64_t st2 = (m->Mpad + kBig - 1) / kBig;
        auto rounds = [&](int64_t n) { return n <= 0 ? (int64_t)0 : (((n + kBig - 1) / kBig) * st2 + num_cu - 1) / num_cu; };
        // cost in units of a twentieth of a round: a chunk also costs its launches and a fill / drain in which the CUs run in
        // lockstep (measured: 1.97 rounds per launch 0.47-0.51 of peak, 2.99 rounds 0.54-0.58) -- three twentieths per chunk
        const int64_t total = round_up(rows, kBig);
        int64_t best = -1;
        // digit path: every launched workgroup parks two 256 KiB scratch tiles (carve: ring_grid x 512 KiB per workspace, times the
        // workspaces in rotation), so the chunk is also capped by a scratch budget of 1 GiB per workspace: 2048 tiles.  M = 2 560
        // is not touched (32 768 rows = 1 280 tiles); a 20 480-SV model stops at 6 400 rows instead of asking for 5 GiB
        const int64_t cmax = dig ? std::max<int64_t>(4096 / 4, std::min<int64_t>(32768, (2048 / st2) * kBig)) : 32768;
        for (int64_t c = std::min<int64_t>(4096, cmax); c <= cmax; c += kBig) {
            const int64_t nfull = total / c, rem = total % c;
            const int64_t cost = 20 * (nfull * rounds(c) + rounds(rem)) + 3 * (nfull + (rem ? 1 : 0));
            if (best < 0 || cost <= best) { best = cost; ch = c; }
            if (c >= total) break;
        }
    }
    return std::min<int64_t>(round_up(rows, kTile), ch);
}

// Workspace carved from the ctx block for one chunk of CH rows.
struct ChunkWs {
    uint8_t* q; float* f32; int32_t* isum; int64_t* isq; double* nsq; int32_t* flags;
    int32_t* tile_exact; int32_t* all_exact; double* partial;
    int8_t* dig; int64_t dig_plane; double* dnsq; int32_t* dflags;      // multi-digit operand of the general rows (or NULL)
    int32_t* stash;                                                     // scratch tiles of k_svm_gemm_ring<.., 1>
    int32_t* ijk;                                                       // derived (i,j,k) of the chunk's frames (fused derive -> slice), or NULL
    int32_t* gsmall;                                                    // [RML_SMALL_FRAMES][Mpad] dot products of the single-observation path
    int32_t* gsplit;                                                    // [Mpad][CH] of the split-K path (chunks of at most kSplitRows rows), or NULL
    size_t bytes;
};

constexpr int64_t kSplitRows = 2048;        // the split-K path serves chunks of at most this many rows

ChunkWs carve(const rml_svm* m, int64_t CH, unsigned char* base, bool need_q, bool need_f32, bool need_dig = false, bool need_ijk = false) {
    ChunkWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return base ? base + o : (unsigned char*)nullptr; };
    w.q = (uint8_t*)take(need_q ? (size_t)CH * m->Dq : 0);
    w.f32 = (float*)take(need_f32 ? (size_t)CH * m->Df * 4 : 0);
    w.isum = (int32_t*)take((size_t)CH * 4);
    w.isq = (int64_t*)take((size_t)CH * 8);
    w.nsq = (double*)take((size_t)CH * 8);
    w.flags = (int32_t*)take((size_t)CH * 4);
    w.tile_exact = (int32_t*)take((size_t)(CH / kTile + 1) * 4);
    w.all_exact = (int32_t*)take(256);
    w.partial = (double*)take((size_t)(m->Mpad / kTile) * CH * m->PT * 8);
    w.dig_plane = CH * m->Dq;
    w.dig = (int8_t*)take(need_dig ? (size_t)4 * CH * m->Dq : 0);
    w.dnsq = (double*)take(need_dig ? (size_t)CH * 8 : 0);
    w.dflags = (int32_t*)take(need_dig ? (size_t)CH * 4 : 0);
    w.stash = (int32_t*)take(need_dig ? (size_t)ring_grid((int)((CH + kBig - 1) / kBig), (int)((m->Mpad + kBig - 1) / kBig)) * 2 * kDigStashBytes : 0);
    if (!need_dig) { w.dig = nullptr; w.dnsq = nullptr; w.dflags = nullptr; w.stash = nullptr; }
    w.ijk = (int32_t*)take(need_ijk ? (size_t)CH * 12 : 0);
    if (!need_ijk) w.ijk = nullptr;
    w.gsmall = (int32_t*)take(need_q ? (size_t)RML_SMALL_FRAMES * m->Mpad * 4 : 0);
    if (!need_q) w.gsmall = nullptr;
    const bool need_split = need_q && CH <= kSplitRows;
    w.gsplit = (int32_t*)take(need_split ? (size_t)m->Mpad * CH * 4 : 0);
    if (!need_split) w.gsplit = nullptr;
    w.bytes = off;
    return w;
}

struct DecisionOut {
    double* dec_ovo; double* dec_ovr; double* proba; int32_t* label_vote; int32_t* label_calib;
    DecisionOut at(int64_t r0, int C, int P) const {
        DecisionOut o = *this;
        if (o.dec_ovo) o.dec_ovo += r0 * P;
        if (o.dec_ovr) o.dec_ovr += r0 * (C == 2 ? 1 : C);
        if (o.proba) o.proba += r0 * C;
        if (o.label_vote) o.label_vote += r0;
        if (o.label_calib) o.label_calib += r0;
        return o;
    }
};

// GEMM(s) + finish for one chunk whose operands are already in place.
// the epilogue of a chunk: partial sums of every SV tile -> decision values, votes, calibrated probabilities, labels
int run_finish(const rml_svm* m, int64_t n, const int32_t* flags, const ChunkWs& w, const DecisionOut& out, hipStream_t st,
               bool all_exact_known, bool forced_i8) {
    FinishArgs fa{};
    fa.partial = w.partial; fa.Npart = n; fa.ST = (int)(m->Mpad / kTile); fa.PT = m->PT; fa.N = n; fa.C = m->C; fa.P = m->P;
    fa.intercept = m->intercept; fa.calib = m->calib; fa.has_calib = m->has_calib;
    fa.row_flags = all_exact_known ? nullptr : flags; fa.tile_exact = all_exact_known ? nullptr : w.tile_exact;
    fa.forced_i8 = forced_i8;
    fa.dec_ovo = out.dec_ovo; fa.dec_ovr = out.dec_ovr; fa.proba = out.proba;
    fa.label_vote = out.label_vote; fa.label_calib = out.label_calib;
    hipLaunchKernelGGL(k_svm_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, fa);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

int launch_epi(const rml_svm* m, const SmallArgs& sa, int ST, int64_t n, hipStream_t st) {
    const dim3 ge((unsigned)ST, (unsigned)n);
    switch (m->PT) {
        case 1: hipLaunchKernelGGL(k_svm_epi_small<1>, ge, dim3(128), 0, st, sa); break;
        case 3: hipLaunchKernelGGL(k_svm_epi_small<3>, ge, dim3(128), 0, st, sa); break;
        case 6: hipLaunchKernelGGL(k_svm_epi_small<6>, ge, dim3(128), 0, st, sa); break;
        case 10: hipLaunchKernelGGL(k_svm_epi_small<10>, ge, dim3(128), 0, st, sa); break;
        case 15: hipLaunchKernelGGL(k_svm_epi_small<15>, ge, dim3(128), 0, st, sa); break;
        default: RML_REQUIRE(false, RML_ERR_UNSUPPORTED, "svm: unsupported pair count");
    }
    RML_HIP(hipGetLastError());
    return RML_OK;
}

// the exact path for n <= RML_SMALL_FRAMES rows: k_svm_dot_small + k_svm_epi_small (bit-identical partial sums: see the kernels)
int launch_small(const rml_svm* m, const GemmArgs& ga, int32_t* G, hipStream_t st) {
    SmallArgs sa{};
    sa.sv = ga.sv; sa.ld_sv = ga.ld_sv; sa.x = ga.x; sa.ld_x = ga.ld_x; sa.Kb = (int64_t)ga.KT * kStepBytes;
    sa.N = (int)ga.N; sa.Mpad = ga.Mpad; sa.tile_exact = ga.tile_exact; sa.G = G; sa.g_sm = 1; sa.g_sn = ga.Mpad;
    sa.x_isum = ga.x_isum; sa.x_isq = ga.x_isq; sa.sv_term = ga.sv_term; sa.W = ga.W; sa.gs = ga.gs; sa.kernel = ga.kernel;
    sa.partial = ga.partial; sa.Npart = ga.Npart;
    const dim3 gd((unsigned)(ga.Mpad / kSmallSv));
    if (ga.N <= 1) hipLaunchKernelGGL(k_svm_dot_small<1>, gd, dim3(256), 0, st, sa);
    else if (ga.N <= 2) hipLaunchKernelGGL(k_svm_dot_small<2>, gd, dim3(256), 0, st, sa);
    else if (ga.N <= 4) hipLaunchKernelGGL(k_svm_dot_small<4>, gd, dim3(256), 0, st, sa);
    else hipLaunchKernelGGL(k_svm_dot_small<8>, gd, dim3(256), 0, st, sa);
    return launch_epi(m, sa, ga.ST, ga.N, st);
}

// the exact path for batches whose tiles do not fill the machine: split-K tile kernel + the chain epilogue (see k_svm_gemm_splitk)
int launch_split(const rml_svm* m, const GemmArgs& ga, int32_t* G, int num_cu, hipStream_t st) {
    const int64_t ldg = (int64_t)ga.FT * kTile;
    // (a kernel, not hipMemsetAsync: the memset of a captured stream was not replayed with the graph -- the second replay added
    // onto the first one's sums, tests/test_capi_gpu.py)
    {
        const int64_t n16 = ga.Mpad * ldg / 4;             // Mpad and ldg are multiples of 128
        hipLaunchKernelGGL(k_zero16, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, st, reinterpret_cast<v4i*>(G), n16);
    }
    SplitArgs sp{};
    sp.sv = ga.sv; sp.ld_sv = ga.ld_sv; sp.x = ga.x; sp.ld_x = ga.ld_x; sp.KT = ga.KT; sp.N = ga.N; sp.FT = ga.FT;
    sp.tile_exact = ga.tile_exact; sp.G = G; sp.ldg = ldg;
    const int tiles = ga.FT * ga.ST;
    int KS = (2 * num_cu + tiles - 1) / tiles;            // about two workgroups per CU
    if (KS > ga.KT / 4) KS = ga.KT / 4;                   // at least four K-steps per range
    if (KS < 1) KS = 1;
    sp.per = (ga.KT + KS - 1) / KS;
    KS = (ga.KT + sp.per - 1) / sp.per;
    hipLaunchKernelGGL(k_svm_gemm_splitk, dim3((unsigned)tiles, (unsigned)KS), dim3(256), 0, st, sp);
    SmallArgs sa{};
    sa.N = (int)ga.N; sa.Mpad = ga.Mpad; sa.tile_exact = ga.tile_exact; sa.G = G; sa.g_sm = ldg; sa.g_sn = 1;
    sa.x_isum = ga.x_isum; sa.x_isq = ga.x_isq; sa.sv_term = ga.sv_term; sa.W = ga.W; sa.gs = ga.gs; sa.kernel = ga.kernel;
    sa.partial = ga.partial; sa.Npart = ga.Npart;
    return launch_epi(m, sa, ga.ST, ga.N, st);
}

int run_chunk(const rml_ctx* ctx, const rml_svm* m, int policy, int64_t n, const uint8_t* q, int64_t ld_q, const int32_t* isum, const int64_t* isq,
              const int32_t* flags, const float* f32, const double* nsq, const ChunkWs& w, const DecisionOut& out, hipStream_t st,
              bool tiles_done = false, double* kmat = nullptr, int64_t ld_k = 0, bool all_exact_known = false, bool allow_big = true,
              bool dig_ready = false, bool defer_finish = false) {
    const int FT = (int)((n + kTile - 1) / kTile);
    const int ST = (int)(m->Mpad / kTile);
    // policy: RML_PATH_AUTO (i8 on exact tiles, f64 elsewhere) / _F32 / _I8 / _F64 (forced)
    const bool run_i8 = m->exact && q && (policy == RML_PATH_AUTO || policy == RML_PATH_I8);
    const bool run_gen = f32 && policy != RML_PATH_I8;
    const bool gen_f32 = (policy == RML_PATH_F32);
    RML_REQUIRE(run_i8 || run_gen, RML_ERR_STATE, "svm: no usable operand path (model exact=%d)", (int)m->exact);
    // large exact batches go to the 256x256 kernel; the tile predicate is then decided per pair of 128-sample tiles
    const int gemm_cus = ctx->num_cu;
    const bool big = allow_big && run_i8 && !kmat && use_big_gemm(m, n, gemm_cus, ctx->opt.gemm_big);
    // general tiles whose rows fit the model's fixed-point range go to the multi-digit int8 kernel (digit planes in w.dig)
    const bool run_dig = dig_ready && w.dig && run_gen && !gen_f32 && !kmat && m->dig_ok;
    if (!tiles_done) {
        const int group = (big || run_dig) ? 2 : 1;
        hipLaunchKernelGGL(k_tile_flags, dim3((FT + group - 1) / group), dim3(128 * group), 0, st, flags, n, FT,
                           run_i8 ? (run_gen ? 0 : 2) : 1, (int)m->exact, w.tile_exact, (int32_t*)nullptr, group);
        if (run_dig)
            hipLaunchKernelGGL(k_tile_dig, dim3((FT + 1) / 2), dim3(256), 0, st, w.dflags, n, FT, w.tile_exact, (const int32_t*)nullptr);
    }
    GemmArgs ga{};
    // all_exact_known: every row is on the code grid by construction (uint8 volumes): no tile predicate at all
    ga.N = n; ga.ST = ST; ga.FT = FT; ga.tile_exact = all_exact_known ? nullptr : w.tile_exact;
    ga.W = m->W; ga.Mpad = m->Mpad; ga.kernel = m->kernel; ga.partial = w.partial; ga.Npart = n;
    ga.kmat = kmat; ga.ld_k = ld_k; ga.M = m->M; ga.sv_rows = m->Mpad;
    if (run_i8) {
        ga.sv = m->sv_q; ga.ld_sv = m->Dq; ga.x = q; ga.ld_x = ld_q; ga.KT = (int)(m->Kq / kStepBytes);
        ga.want = 1; ga.x_isum = isum; ga.x_isq = isq; ga.sv_term = m->sv_term_q;
        const double sc2 = m->code_scale * m->code_scale;
        ga.gs = (m->kernel == RML_KERNEL_RBF ? m->gamma : 1.0) / sc2;
        // a handful of rows: the matrix-vector kernels (the SV codes read once by the whole chip instead of by Mpad / 128 workgroups)
        const bool small = !kmat && n <= RML_SMALL_FRAMES && w.gsmall && (m->Mpad % kSmallSv) == 0;
        // ... and batches whose 128 x 128 tiles are fewer than the CUs: the same tiles cut along K (exact: any order)
        const bool split = !kmat && !small && !big && w.gsplit && n <= kSplitRows && (int64_t)FT * ST < gemm_cus && ga.KT >= 8;
        int rc = kmat ? launch_gemm<PATH_I8, true>(m, ga, st)
                      : (small ? launch_small(m, ga, w.gsmall, st)
                               : (split ? launch_split(m, ga, w.gsplit, gemm_cus, st)
                                        : (big ? launch_gemm_big(m, ga, st) : launch_gemm<PATH_I8>(m, ga, st))));
        if (rc) return rc;
    }
    if (run_dig) {
        RingArgs ra{};
        ra.sv = reinterpret_cast<const uint8_t*>(m->sv_dig); ra.x = reinterpret_cast<const uint8_t*>(w.dig);
        ra.ld_sv = m->Dq; ra.ld_x = m->Dq; ra.sv_plane = m->Mpad * m->Dq; ra.x_plane = w.dig_plane;
        ra.KT = (int)(m->Kq / kStepBytes); ra.N = n; ra.Mpad = m->Mpad; ra.sv_rows = m->Mpad; ra.ST = ST; ra.FT = FT;
        ra.tile_exact = w.tile_exact; ra.want = 2; ra.x_nsq = w.dnsq; ra.sv_term = m->sv_dig_nsq; ra.W = m->W;
        ra.gs = m->gamma * m->dig_s * m->dig_s; ra.kernel = RML_KERNEL_RBF; ra.partial = w.partial; ra.Npart = n; ra.stash = w.stash;
        int rc = launch_gemm_ring<1>(m, ra, st);
        if (rc) return rc;
    }
    if (run_gen) {
        ga.sv = reinterpret_cast<const uint8_t*>(m->sv_f32); ga.ld_sv = m->Df * 4;
        ga.x = reinterpret_cast<const uint8_t*>(f32); ga.ld_x = m->Df * 4; ga.KT = (int)(m->Kf * 4 / kStepBytes);
        ga.want = 0; ga.x_nsq = nsq; ga.sv_term = m->sv_nsq; ga.gs = m->gamma;
        int rc = gen_f32 ? launch_gemm<PATH_F32>(m, ga, st)
                         : (kmat ? launch_gemm<PATH_F64, true>(m, ga, st) : launch_gemm<PATH_F64>(m, ga, st));
        if (rc) return rc;
    }
    if (kmat || defer_finish) { RML_HIP(hipGetLastError()); return RML_OK; }      // kernel values only, or the caller launches run_finish itself
    return run_finish(m, n, flags, w, out, st, all_exact_known, run_i8 && !run_gen);
}

}  // namespace

// ---- model load ---------------------------------------------------------------------------
// The host half of rml_svm_load: validation, geometry and every packed operand, no HIP call (the sanitizer build of
// tools/sanitize drives it on a box without a GPU).  Fills the geometry / flags of *m and the arrays of *pk.
int rml_svm_pack_host(const double* sv, int64_t M, int64_t D, const double* dual_coef, const int32_t* n_support,
                      int n_classes, int kernel, double gamma, double code_scale, bool has_calib, rml_svm* m, rml_svm_pack* pk) {
    RML_REQUIRE(sv && dual_coef && n_support && m && pk, RML_ERR_INVALID, "rml_svm_load: NULL argument");
    RML_REQUIRE(M > 0 && D > 0, RML_ERR_INVALID, "rml_svm_load: empty model");
    RML_REQUIRE(n_classes >= 2 && n_classes <= kMaxC, RML_ERR_UNSUPPORTED, "rml_svm_load: %d classes (supported: 2..%d)", n_classes, kMaxC);
    RML_REQUIRE(kernel == RML_KERNEL_RBF || kernel == RML_KERNEL_LINEAR, RML_ERR_UNSUPPORTED, "rml_svm_load: kernel %d", kernel);
    int64_t msum = 0;
    for (int c = 0; c < n_classes; ++c) { RML_REQUIRE(n_support[c] >= 0, RML_ERR_INVALID, "rml_svm_load: negative n_support"); msum += n_support[c]; }
    RML_REQUIRE(msum == M, RML_ERR_INVALID, "rml_svm_load: sum(n_support)=%lld != M=%lld", (long long)msum, (long long)M);
    m->M = M; m->D = D; m->C = n_classes; m->P = n_classes * (n_classes - 1) / 2;
    m->PT = m->P <= 1 ? 1 : (m->P <= 3 ? 3 : (m->P <= 6 ? 6 : (m->P <= 10 ? 10 : 15)));
    m->kernel = kernel; m->gamma = gamma; m->code_scale = code_scale > 1.0 ? code_scale : 1.0;
    m->Mpad = round_up(M, kTile);
    m->Kq = round_up(D, kStepBytes); m->Kf = round_up(D, 32);
    m->Dq = ((m->Kq / 128) & 1) ? m->Kq : m->Kq + 128;
    m->Df = ((m->Kf / 32) & 1) ? m->Kf : m->Kf + 32;
    m->has_calib = has_calib;

    // per-pair SV weights: the pair loop of svm_predict_values (svm.cpp:2864-2883)
    std::vector<double>& W = pk->W;
    W.assign((size_t)m->PT * m->Mpad, 0.0);
    {
        std::vector<int64_t> start(n_classes, 0);
        for (int c = 1; c < n_classes; ++c) start[c] = start[c - 1] + n_support[c - 1];
        int p = 0;
        for (int i = 0; i < n_classes; ++i)
            for (int j = i + 1; j < n_classes; ++j, ++p) {
                for (int64_t k = 0; k < n_support[i]; ++k) W[(size_t)p * m->Mpad + start[i] + k] = dual_coef[(size_t)(j - 1) * M + start[i] + k];
                for (int64_t k = 0; k < n_support[j]; ++k) W[(size_t)p * m->Mpad + start[j] + k] = dual_coef[(size_t)i * M + start[j] + k];
            }
    }
    // float operand + norms
    std::vector<float>& svf = pk->svf;
    std::vector<double>& nsq = pk->nsq;
    svf.assign((size_t)m->Mpad * m->Df, 0.0f);
    nsq.assign(m->Mpad, 0.0);
    // exact codes: every SV must be bit-identical to float32(c/scale) (or to c when unscaled)
    std::vector<uint8_t>& svq = pk->svq;
    std::vector<double>& term = pk->term;
    svq.assign((size_t)m->Mpad * m->Dq, 0);
    term.assign(m->Mpad, 0.0);
    bool exact = true;
    const float fscale = (float)m->code_scale;
    for (int64_t r = 0; r < M; ++r) {
        double nn = 0.0; int64_t isum = 0, isq = 0;
        for (int64_t d = 0; d < D; ++d) {
            const double v = sv[(size_t)r * D + d];
            const float vf = (float)v;
            svf[(size_t)r * m->Df + d] = vf;
            nn += (double)vf * (double)vf;
            if (exact) {
                double c = nearbyint(v * m->code_scale);
                bool good = c >= 0.0 && c <= 255.0;
                if (good) {
                    double back = m->code_scale > 1.0 ? (double)((float)c / fscale) : c;
                    good = (back == v);
                }
                if (!good) exact = false;
                else { int ci = (int)c; svq[(size_t)r * m->Dq + d] = (uint8_t)(ci ^ 0x80); isum += ci; isq += (int64_t)ci * ci; }
            }
        }
        nsq[r] = nn;
        // RBF:    d^2 = (isq_x - 256 isum_x) + [isq_s - 256 isum_s + 32768 D] - 2 G'
        // linear: x.s = G' + 128 isum_x + [128 isum_s - 16384 D]
        term[r] = (kernel == RML_KERNEL_RBF) ? (double)(isq - 256 * isum + 32768 * D) : (double)(128 * isum - 16384 * D);
    }
    // the int8 MFMA accumulates sum (a-128)(b-128) in int32: |.| <= 128^2 * K must stay below 2^31
    if (m->Kq >= 131072) exact = false;
    m->exact = exact;
    // multi-digit frame for general rows: u = (v - c0) / s with s the power of two >= the SV value range (SVs then sit in
    // [-1/2, 1/2] + rounding of c0; rows may leave the SV range by s/2 on either side before they fall back to float64) and
    // c0 the mid-range on a grid of s/256 (so that v - c0 is exact in float64 for float32 v).  Four digit groups share one
    // int32 accumulator: 4 * 128^2 * K < 2^31.
    std::vector<int8_t>& svd = pk->svd;
    std::vector<double>& dnsq = pk->dnsq;
    svd.clear(); dnsq.clear();
    m->dig_ok = false;
    if (kernel == RML_KERNEL_RBF && m->Kq < 32768 && m->PT <= 6) {
        double lo = sv[0], hi = sv[0];
        bool finite = true;
        for (size_t i = 0; i < (size_t)M * D; ++i) { const double v = sv[i]; finite &= std::isfinite(v); lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        if (finite && hi > lo) {
            int e = 0;
            (void)frexp(hi - lo, &e);                             // hi - lo = f 2^e, f in [0.5, 1)
            double sd = ldexp(1.0, e);
            if (ldexp(1.0, e - 1) == hi - lo) sd = hi - lo;       // an exact power of two is its own scale
            const double c0 = nearbyint(0.5 * (lo + hi) / sd * 256.0) / 256.0 * sd;
            const double k31 = 2147483648.0 / sd;
            svd.assign((size_t)4 * m->Mpad * m->Dq, 0);
            dnsq.assign(m->Mpad, 0.0);
            const size_t plane = (size_t)m->Mpad * m->Dq;
            bool ok = true;
            for (int64_t r = 0; r < M && ok; ++r) {
                double nn = 0.0;
                for (int64_t d = 0; d < D; ++d) {
                    const double t = nearbyint((sv[(size_t)r * D + d] - c0) * k31);
                    if (!(t >= -2147483648.0 && t <= 2139062143.0)) { ok = false; break; }
                    const int32_t I = (int32_t)t;
                    const double u = (double)I * 0x1p-31;
                    nn += u * u;
                    const uint32_t pk4 = ((uint32_t)I + 0x00808080u) ^ 0x00808080u;     // bytes = balanced digits, a0 on top
                    for (int dg = 0; dg < 4; ++dg) svd[(size_t)dg * plane + (size_t)r * m->Dq + d] = (int8_t)((pk4 >> (8 * (3 - dg))) & 255u);
                }
                dnsq[r] = nn;
            }
            if (ok) { m->dig_ok = true; m->dig_c0 = c0; m->dig_s = sd; }
        }
    }
    return RML_OK;
}

extern "C" int rml_svm_load(rml_ctx* ctx, const double* sv, int64_t M, int64_t D,
                            const double* dual_coef, const double* intercept, const int32_t* n_support,
                            int n_classes, int kernel, double gamma, double code_scale,
                            const double* calib_a, const double* calib_b, rml_svm** out) {
    RML_REQUIRE(ctx && sv && dual_coef && intercept && n_support && out, RML_ERR_INVALID, "rml_svm_load: NULL argument");
    RML_REQUIRE((calib_a == nullptr) == (calib_b == nullptr), RML_ERR_INVALID, "rml_svm_load: calib_a/calib_b must both be given");
    *out = nullptr;
    rml_svm* m = new (std::nothrow) rml_svm();
    RML_REQUIRE(m != nullptr, RML_ERR_NOMEM, "rml_svm_load: out of host memory");
    rml_svm_pack pk;
    int rc = rml_svm_pack_host(sv, M, D, dual_coef, n_support, n_classes, kernel, gamma, code_scale, calib_a != nullptr, m, &pk);
    if (rc) { delete m; return rc; }
    {
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) { delete m; RML_HIP(e); }
    }
    const bool exact = m->exact;
    do {
        if ((rc = dev_upload(&m->W, pk.W))) break;
        if ((rc = dev_upload(&m->sv_f32, pk.svf))) break;
        if ((rc = dev_upload(&m->sv_nsq, pk.nsq))) break;
        if (exact) {
            if ((rc = dev_upload(&m->sv_q, pk.svq))) break;
            if ((rc = dev_upload(&m->sv_term_q, pk.term))) break;
        }
        if (m->dig_ok) {
            if ((rc = dev_upload(&m->sv_dig, pk.svd))) break;
            if ((rc = dev_upload(&m->sv_dig_nsq, pk.dnsq))) break;
        }
        std::vector<double> ic(intercept, intercept + m->P);
        if ((rc = dev_upload(&m->intercept, ic))) break;
        if (m->has_calib) {
            std::vector<double> cal(2 * n_classes);
            const int ncal = n_classes == 2 ? 1 : n_classes;
            for (int c = 0; c < n_classes; ++c) { cal[c] = c < ncal ? calib_a[c] : 0.0; cal[n_classes + c] = c < ncal ? calib_b[c] : 0.0; }
            if ((rc = dev_upload(&m->calib, cal))) break;
        }
    } while (0);
    if (rc) { rml_svm_free(ctx, m); return rc; }
    *out = m;
    return RML_OK;
}

extern "C" int rml_svm_free(rml_ctx* ctx, rml_svm* m) {
    if (!m) return RML_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    void* bufs[] = {m->sv_f32, m->sv_nsq, m->sv_q, m->sv_term_q, m->W, m->intercept, m->calib, m->platt, m->sv_dig, m->sv_dig_nsq};
    for (void* b : bufs) if (b) (void)hipFree(b);
    delete m;
    return RML_OK;
}

extern "C" int rml_svm_is_exact(const rml_svm* m) { return m && m->exact ? 1 : 0; }
extern "C" int64_t rml_svm_num_sv(const rml_svm* m) { return m ? m->M : 0; }
extern "C" int64_t rml_svm_dim(const rml_svm* m) { return m ? m->D : 0; }

// ---- decision on caller-provided rows -----------------------------------------------------
extern "C" int rml_svm_decision(rml_ctx* ctx, const rml_svm* m, int path,
                                const float* feat, int64_t ld_feat,
                                const uint8_t* feat_q, int64_t ld_q, const int32_t* row_isum, const int64_t* row_isq,
                                const int32_t* row_flags, int64_t N,
                                double* dec_ovo, double* dec_ovr, double* proba,
                                int32_t* label_vote, int32_t* label_calib, void* stream) {
    RML_REQUIRE(ctx && m && N >= 0, RML_ERR_INVALID, "rml_svm_decision: bad arguments");
    if (N == 0) return RML_OK;
    RML_REQUIRE(feat || feat_q, RML_ERR_INVALID, "rml_svm_decision: need feat or feat_q");
    RML_REQUIRE(!feat || ld_feat >= m->D, RML_ERR_INVALID, "rml_svm_decision: ld_feat < D");
    RML_REQUIRE(path >= RML_PATH_AUTO && path <= RML_PATH_DIGITS, RML_ERR_INVALID, "rml_svm_decision: bad path %d", path);
    RML_REQUIRE(path != RML_PATH_DIGITS || m->dig_ok, RML_ERR_STATE, "rml_svm_decision: multi-digit path requested but the model has no digit frame "
                "(RBF kernel, D < 32768 and at most 6 class pairs)");
    RML_REQUIRE(!(proba || label_calib) || m->has_calib, RML_ERR_STATE, "rml_svm_decision: model has no calibrators");
    RML_REQUIRE(path != RML_PATH_I8 || m->exact, RML_ERR_STATE, "rml_svm_decision: exact path requested but the model is not on the code grid");
    if (!feat) {
        RML_REQUIRE(m->exact, RML_ERR_STATE, "rml_svm_decision: code rows given but the model is not on the code grid");
        RML_REQUIRE(path == RML_PATH_AUTO || path == RML_PATH_I8, RML_ERR_INVALID, "rml_svm_decision: the float paths need float rows");
        RML_REQUIRE(row_isum && row_isq, RML_ERR_INVALID, "rml_svm_decision: code rows need row_isum/row_isq");
        RML_REQUIRE(ld_q >= m->Kq && ld_q % 16 == 0 && (reinterpret_cast<uintptr_t>(feat_q) & 15) == 0, RML_ERR_INVALID,
                    "rml_svm_decision: code rows need ld_q >= %lld, ld_q %% 16 == 0 and 16-byte alignment", (long long)m->Kq);
    }
    RML_HIP(hipSetDevice(ctx->device));
    if (N == 0) return RML_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rml_ctx_guard guard(ctx, st);           // shared workspace
    // float rows of a model that is not on the code grid: the multi-digit kernel when the batch fills enough 256 x 256 tiles
    // (chunks sized for whole rounds of one workgroup per CU, like the exact 256 x 256 kernel's)
    const bool dig = feat != nullptr && !m->exact && use_dig_gemm(m, path, N, ctx->num_cu);
    const int64_t CH = feat ? ((dig || m->exact) ? pick_chunk(ctx, m, N, 8192, ctx->num_cu, dig) : std::min<int64_t>(round_up(N, kTile), 8192))
                            : pick_chunk(ctx, m, N, 8192, ctx->num_cu);
    const bool need_q = feat != nullptr && m->exact && (path == RML_PATH_AUTO || path == RML_PATH_I8);
    const bool need_f32 = feat != nullptr;
    // a model on the code grid can meet general rows as well (mixed batches): digit planes then ride along with the codes
    const bool need_dig = feat != nullptr && (dig || (m->exact && use_dig_gemm(m, path, N, ctx->num_cu)));
    ChunkWs probe = carve(m, CH, nullptr, need_q, need_f32, need_dig);
    void* ws = nullptr;
    int rc = rml_ws_reserve(ctx, probe.bytes, &ws, st);
    if (rc) return rc;
    ChunkWs w = carve(m, CH, static_cast<unsigned char*>(ws), need_q, need_f32, need_dig);
    DecisionOut out{dec_ovo, dec_ovr, proba, label_vote, label_calib};
    const int policy = path;
    for (int64_t r0 = 0; r0 < N; r0 += CH) {
        const int64_t n = std::min(CH, N - r0);
        if (feat) {
            hipLaunchKernelGGL(k_prepare_rows, dim3((unsigned)n), dim3(256), 0, st, feat + r0 * ld_feat, ld_feat, m->D,
                               (float)m->code_scale, w.f32, m->Df, w.nsq, need_q ? w.q : nullptr, m->Dq, w.isum, w.isq, w.flags);
            if (need_dig)
                hipLaunchKernelGGL(k_digit_rows, dim3((unsigned)n), dim3(256), 0, st, w.f32, m->Df, m->D, m->Dq, w.dig_plane, w.dig, w.dnsq,
                                   w.dflags, m->dig_c0, 2147483648.0 / m->dig_s, (const int32_t*)nullptr);
            RML_HIP(hipGetLastError());
            rc = run_chunk(ctx, m, policy, n, need_q ? w.q : nullptr, m->Dq, w.isum, w.isq, w.flags, w.f32, w.nsq, w, out.at(r0, m->C, m->P), st,
                           false, nullptr, 0, false, true, need_dig);
        } else {
            rc = run_chunk(ctx, m, RML_PATH_I8, n, feat_q + r0 * ld_q, ld_q, row_isum + r0, row_isq + r0, row_flags ? row_flags + r0 : nullptr,
                           nullptr, nullptr, w, out.at(r0, m->C, m->P), st);
        }
        if (rc) return rc;
    }
    return RML_OK;
}

// ---- kernel matrix K(X, SV): the Gram-matrix service for SVC training with kernel='precomputed' -------------
extern "C" int rml_svm_kernel_matrix(rml_ctx* ctx, const rml_svm* m, int path, const float* feat, int64_t ld_feat, int64_t N,
                                     double* kmat, int64_t ld_k, void* stream) {
    RML_REQUIRE(ctx && m && N >= 0, RML_ERR_INVALID, "rml_svm_kernel_matrix: bad arguments");
    if (N == 0) return RML_OK;
    RML_REQUIRE(feat && kmat, RML_ERR_INVALID, "rml_svm_kernel_matrix: NULL argument");
    RML_REQUIRE(ld_feat >= m->D && ld_k >= m->M, RML_ERR_INVALID, "rml_svm_kernel_matrix: ld_feat < D or ld_k < M");
    RML_REQUIRE(path == RML_PATH_AUTO || path == RML_PATH_I8 || path == RML_PATH_F64, RML_ERR_INVALID,
                "rml_svm_kernel_matrix: path must be AUTO, I8 or F64");
    RML_REQUIRE(path != RML_PATH_I8 || m->exact, RML_ERR_STATE, "rml_svm_kernel_matrix: exact path requested but the model is not on the code grid");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    rml_ctx_guard guard(ctx, st);           // shared workspace
    const int64_t CH = std::min<int64_t>(round_up(N, kTile), 8192);
    const bool need_q = m->exact && (path == RML_PATH_AUTO || path == RML_PATH_I8);
    ChunkWs probe = carve(m, CH, nullptr, need_q, true);
    void* ws = nullptr;
    int rc = rml_ws_reserve(ctx, probe.bytes, &ws, st);
    if (rc) return rc;
    ChunkWs w = carve(m, CH, static_cast<unsigned char*>(ws), need_q, true);
    DecisionOut none{nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int64_t r0 = 0; r0 < N; r0 += CH) {
        const int64_t n = std::min(CH, N - r0);
        hipLaunchKernelGGL(k_prepare_rows, dim3((unsigned)n), dim3(256), 0, st, feat + r0 * ld_feat, ld_feat, m->D,
                           (float)m->code_scale, w.f32, m->Df, w.nsq, need_q ? w.q : nullptr, m->Dq, w.isum, w.isq, w.flags);
        RML_HIP(hipGetLastError());
        rc = run_chunk(ctx, m, path, n, need_q ? w.q : nullptr, m->Dq, w.isum, w.isq, w.flags, w.f32, w.nsq, w, none, st,
                       /*tiles_done=*/false, kmat + r0 * ld_k, ld_k);
        if (rc) return rc;
    }
    return RML_OK;
}

// ---- fused front door: volumes -> projection -> SVM ---------------------------------------
namespace {
int project_svm_impl(rml_ctx* ctx, const rml_svm* m, const void* V, int vdtype, int64_t B, int X, int Y, int Z,
                     int mode, const int32_t* ijk, bool derive, int32_t* ijk_out, float scale_div, uint32_t mask,
                     double* dec_ovo, double* dec_ovr, double* proba,
                     int32_t* label_vote, int32_t* label_calib, void* stream);
}

extern "C" int rml_project_svm(rml_ctx* ctx, const rml_svm* m, const void* V, int vdtype, int64_t B, int X, int Y, int Z,
                               int mode, const int32_t* ijk, float scale_div, uint32_t mask,
                               double* dec_ovo, double* dec_ovr, double* proba,
                               int32_t* label_vote, int32_t* label_calib, void* stream) {
    return project_svm_impl(ctx, m, V, vdtype, B, X, Y, Z, mode, ijk, false, nullptr, scale_div, mask, dec_ovo, dec_ovr, proba, label_vote,
                            label_calib, stream);
}

extern "C" int rml_derive_project_svm(rml_ctx* ctx, const rml_svm* m, const void* V, int vdtype, int64_t B, int X, int Y, int Z,
                                      float scale_div, uint32_t mask, int32_t* ijk_out,
                                      double* dec_ovo, double* dec_ovr, double* proba,
                                      int32_t* label_vote, int32_t* label_calib, void* stream) {
    RML_REQUIRE(rml_derive_slice_supported(ctx, V, vdtype, X, Y, Z, 1) == 1, RML_ERR_UNSUPPORTED,
                "rml_derive_project_svm: no fused derive kernel for %dx%dx%d (rows of whole quads, Z <= 256, odd part of Z/4 <= 15): "
                "use rml_derive_targets + rml_project_svm(mode SLICE)", X, Y, Z);
    return project_svm_impl(ctx, m, V, vdtype, B, X, Y, Z, RML_MODE_SLICE, nullptr, true, ijk_out, scale_div, mask, dec_ovo, dec_ovr, proba,
                            label_vote, label_calib, stream);
}

namespace {
int project_svm_impl(rml_ctx* ctx, const rml_svm* m, const void* V, int vdtype, int64_t B, int X, int Y, int Z,
                     int mode, const int32_t* ijk, bool derive, int32_t* ijk_out, float scale_div, uint32_t mask,
                     double* dec_ovo, double* dec_ovr, double* proba,
                     int32_t* label_vote, int32_t* label_calib, void* stream) {
    RML_REQUIRE(ctx && m && B >= 0, RML_ERR_INVALID, "rml_project_svm: bad arguments");
    if (B == 0) return RML_OK;
    RML_REQUIRE(V != nullptr, RML_ERR_INVALID, "rml_project_svm: V is NULL");
    RML_REQUIRE(vdtype == RML_VOL_F32 || vdtype == RML_VOL_U8, RML_ERR_INVALID, "rml_project_svm: unknown volume dtype %d", vdtype);
    RML_REQUIRE(rml_feature_len(X, Y, Z, mask) == m->D, RML_ERR_INVALID, "rml_project_svm: grid/mask give D=%lld, model has D=%lld",
                (long long)rml_feature_len(X, Y, Z, mask), (long long)m->D);
    RML_REQUIRE(!(proba || label_calib) || m->has_calib, RML_ERR_STATE, "rml_project_svm: model has no calibrators");
    RML_REQUIRE(mode != RML_MODE_SLICE || ijk || derive, RML_ERR_INVALID, "rml_project_svm: mode SLICE needs ijk");
    if (mode == RML_MODE_MAX_NAN) mode = RML_MODE_MAX;         // round 6: mode MAX itself has NumPy's NaN policy
    // the code grid of the features must be the model's: codes are the unscaled values
    const bool scaled = scale_div > 1.0f;
    const bool grid_ok = m->exact && ((scaled && (double)scale_div == m->code_scale) || (!scaled && m->code_scale == 1.0));
    RML_HIP(hipSetDevice(ctx->device));
    if (B == 0) return RML_OK;
    hipStream_t caller = static_cast<hipStream_t>(stream);
    rml_ctx_guard guard(ctx, caller);       // shared workspaces, aux stream and chunk events (the caller's stream joins at the end)
    hipStream_t st = caller;                // the projections' stream
    // Single observations -- how the reference calls the surface (predict.py:98-119: one target per call) -- and other batches of
    // at most one sample tile (128 frames) on a code-grid model: everything on the caller's stream (no second stream, no events), the
    // frame split over the chip (rml_launch_project_split) and the matrix-vector SVM kernels (run_chunk picks them by the row
    // count): 64x64x128 float32 314 -> ~100 us per call on the host clock, GPU work 275 -> ~60 us.
    if (B <= kTile && grid_ok && !derive && (mode == RML_MODE_MAX || (mode == RML_MODE_SLICE && ijk))) {
        // (slices at given voxels -- the SDK target of predict.py:98-107 -- are one wave per row as they are: k_slice_rows)
        // (up to one sample tile of frames on this path; the frames are split while their pieces are fewer than ~4 per CU)
        const int S = (vdtype == RML_VOL_F32 && mode == RML_MODE_MAX && B <= 64) ? rml_project_split_pieces(X, Y, Z) : 0;      // byte volumes: k_project_u8_max takes 24 us as it is
        const int64_t CHs = kTile;
        ChunkWs probe = carve(m, CHs, nullptr, true, vdtype != RML_VOL_U8, false, false);
        const size_t sbytes = S ? ((rml_project_split_scratch_bytes(B, X, Y, Z, S) + 255) & ~(size_t)255) : 0;
        void* ws = nullptr;
        int rc = rml_ws_reserve(ctx, probe.bytes + sbytes, &ws, st);
        if (rc) return rc;
        const ChunkWs w = carve(m, CHs, static_cast<unsigned char*>(ws), true, vdtype != RML_VOL_U8, false, false);
        float* scratch = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + probe.bytes);
        const bool u8_exact = vdtype == RML_VOL_U8;
        ProjOut o{};
        int64_t off = 0;
        for (int pl = 0; pl < 3; ++pl)
            if (mask & (1u << pl)) {
                o.q[pl] = w.q + off;
                off += pl == 0 ? (int64_t)X * Z : (pl == 1 ? (int64_t)Y * Z : (int64_t)X * Y);
            }
        o.sel = mask & RML_MASK_ALL;
        o.qstride = m->Dq; o.qrow = w.q; o.qD = m->D;
        o.row_isum = w.isum; o.row_isq = w.isq; o.row_flags = u8_exact ? nullptr : w.flags; o.scale_div = scale_div;
        rc = S ? rml_launch_project_split(ctx, V, vdtype, B, X, Y, Z, o, scratch, S, st) : rml_launch_project(ctx, V, vdtype, B, X, Y, Z, mode, ijk, o, st);
        if (rc) return rc;
        DecisionOut out{dec_ovo, dec_ovr, proba, label_vote, label_calib};
        if (u8_exact)
            return run_chunk(ctx, m, RML_PATH_I8, B, w.q, m->Dq, w.isum, w.isq, nullptr, nullptr, nullptr, w, out, st, /*tiles_done=*/true, nullptr, 0,
                             /*all_exact_known=*/true, /*allow_big=*/false);
        hipLaunchKernelGGL(k_tile_flags, dim3(2), dim3(128), 0, st, w.flags, B, 1, 0, 1, w.tile_exact, w.all_exact, 1);
        // float rows + norms for frames that left the code grid (float64 path): a no-op when every frame is on it
        ProjOut of{};
        off = 0;
        for (int pl = 0; pl < 3; ++pl)
            if (mask & (1u << pl)) {
                of.p[pl] = w.f32 + off; of.stride[pl] = m->Df;
                off += pl == 0 ? (int64_t)X * Z : (pl == 1 ? (int64_t)Y * Z : (int64_t)X * Y);
            }
        of.sel = mask & RML_MASK_ALL;
        of.scale_div = scale_div; of.prow = w.f32; of.pD = m->D; of.pstride = m->Df; of.row_nsq = w.nsq;
        of.skip_if_set = w.all_exact;
        rc = rml_launch_project(ctx, V, vdtype, B, X, Y, Z, mode, ijk, of, st);
        if (rc) return rc;
        return run_chunk(ctx, m, RML_PATH_AUTO, B, w.q, m->Dq, w.isum, w.isq, w.flags, w.f32, w.nsq, w, out, st, /*tiles_done=*/true, nullptr, 0, false,
                         /*allow_big=*/false);
    }
    // chunk so that GEMM(c) overlaps projection(c+1): two workspaces, aux stream for the GEMMs
    // frames per chunk: see small_chunk below; RML_CHUNK overrides
    // Persistent wave-per-frame projection (Walabot-like grids): one projection workgroup per CU plus 128x128 GEMM workgroups
    // beside it overlap for real (GEMM hidden under the projection: 26.7 vs 28.0 ms per 262 144 frames), which the
    // 256x256 GEMM (205 VGPRs x 8 waves) cannot do -- it does not fit on a CU next to anything.
    // derive -> slice: the persistent k_derive_slice takes the same pairing (one 8-wave workgroup per CU beside 128x128 GEMM
    // workgroups; the ring GEMM in whole-round chunks, the two kernels taking turns, measured 1-2 % slower: DESIGN.md 3.3)
    const bool wave_proj = derive ? true
                                  : rml_project_uses_wave_kernel(ctx, vdtype, mode, X, Y, Z, /*share_cu=*/true, std::min<int64_t>(B, 8192));
    // Byte volumes (k_project_u8_max): the GEMM is a third of the step there, and since the projection's cross-lane steps left the
    // LDS pipe (round 3: 0.59 -> 0.71 of 8 TB/s alone) each kernel is worth more alone than beside the other: the 256x256 ring
    // kernel in whole-round chunks, the projection between its rounds (64x64x128 uint8, same box: 6.5-6.9 -> 7.4-7.5 M frames/s;
    // Walabot grid 17.9-18.0 -> 18.4).
    // (A CU partition -- the GEMM on g CUs of every XCD, the projection on the other 32 - g, six splits -- was measured in rounds
    // 2-3 and never won: DESIGN.md 3.3; the knob and its masked streams are gone since round 5.)
    const int gemm_cus = ctx->num_cu;
    const bool small_gemm = wave_proj;
    // 8192 frames per chunk; 16384 for small byte frames (same-box A/B: 22x31x176 float32 10.3 vs 9.6 M frames/s at 8192 vs 16384,
    // uint8 17.4 vs 17.7)
    // derive -> slice beside the 128x128 GEMM, frames of at most 1 MiB: 12 288 (six interleaved runs, session r4bk, Walabot grid:
    // 8.05 M frames/s at 8192, 8.21 at 12 288, 8.21 at 16 384; 64x64x128: 2.28 / 2.24 / 2.20 -- stays at 8192)
    const int64_t small_chunk = (vdtype == RML_VOL_U8 && (int64_t)X * Y * Z <= 200000) ? 16384
                                : (derive && vdtype != RML_VOL_U8 && (int64_t)X * Y * Z * 4 <= (1 << 20)) ? 12288 : 8192;
    // rows off the code grid: the multi-digit int8 kernel (256 x 256 tiles: chunks sized for whole rounds) where the model has a
    // digit frame and the batch is large enough, the float64 MFMA kernel otherwise
    // Only for models off the code grid: with a grid model the rows off the grid are the exception, and the digit kernel's
    // predicated launch -- a whole CU's LDS per workgroup even when it exits at once -- cannot start beside the resident projection
    // and GEMM workgroups: on the GEMM stream it waited for the end of the running projection launch, chunk after chunk
    // (profiles/r03_stats_walabot_f32.txt of session r3o: 224 launches of k_svm_gemm_ring<3,1>, up to 325 us each)
    const bool use_dig = !grid_ok && vdtype != RML_VOL_U8 && use_dig_gemm(m, RML_PATH_AUTO, B, ctx->num_cu);
    const int64_t CH = (grid_ok && !small_gemm) ? pick_chunk(ctx, m, B, small_chunk, gemm_cus)
                       : (!grid_ok && use_dig)  ? pick_chunk(ctx, m, B, 8192, ctx->num_cu, true)
                                                : std::min<int64_t>(round_up(B, kTile), grid_ok ? pick_chunk_opt(ctx, small_chunk) : 8192);
    const bool ws_ijk = derive && !ijk_out;             // the derived (i,j,k) stay in the chunk's workspace when the caller does not want them
    ChunkWs probe = carve(m, CH, nullptr, grid_ok, true, use_dig, ws_ijk);
    // workspaces in rotation: 2 (projection of chunk c+1 beside the GEMM of chunk c; a third one -- the projection two chunks
    // ahead -- changed nothing at 64x64x128 and cost 2 % at the Walabot grid: DESIGN.md 3.3)
    constexpr int NBUF = 2;
    void* ws = nullptr;
    int rc = rml_ws_reserve(ctx, (size_t)NBUF * probe.bytes, &ws, st);
    if (rc) return rc;
    ChunkWs w2[NBUF];
    for (int i = 0; i < NBUF; ++i) w2[i] = carve(m, CH, static_cast<unsigned char*>(ws) + (size_t)i * probe.bytes, grid_ok, true, use_dig, ws_ijk);
    DecisionOut out{dec_ovo, dec_ovr, proba, label_vote, label_calib};
    const int64_t frame_elems = (int64_t)X * Y * Z;
    hipStream_t aux = ctx->aux_stream;
    hipEvent_t* ev_proj = ctx->ev_proj;
    hipEvent_t* ev_done = ctx->ev_done;
    // (Three streams -- the small kernels either side of a chunk's GEMM on a third one, RML_PIPE_SPLIT in rounds 3-4 -- were
    // bimodal at the Walabot grid and -5 % at 64x64x128: DESIGN.md 3.3; removed in round 5.)
    // (Round 5, session r5j: the exact GEMM deciding its tiles from the row flags itself and starting the moment the projection is
    // done, with the tile decision and the predicated general path -- second pass, float64 GEMM -- on a third stream beside it and
    // the finish waiting for both: the chain WAS shorter, and the step slower -- 64x64x128 0.716-0.733 end to end against
    // 0.752-0.759 on boxes of the same class, Walabot 0.593-0.598 against 0.633; k_project_lin in situ 0.66-0.68 against 0.745.
    // A GEMM that starts WITH the next projection puts its workgroups on the CUs first, two per CU where the projection's
    // persistent workgroup should go, and the projection pays for the imbalance.  The ~30 us of small kernels in front of the GEMM
    // are what lets the projection settle first.)
    hipStream_t side = aux;
    // aux must start after everything already queued by the caller
    RML_HIP(hipEventRecord(ctx->ev_fork, caller));
    RML_HIP(hipStreamWaitEvent(aux, ctx->ev_fork, 0));
    // (Round 5, session r5i: tapering the last chunk -- 8 192 -> 4 096, 2 048, 2 048, so that the GEMM left exposed behind the last
    // projection is a quarter of a chunk's -- LOST: end to end / in-situ kernel 0.953 against 0.957-0.958 at 64x64x128 and
    // 0.607 against 0.633 of 8 TB/s at the Walabot grid; a persistent launch over 2 048 frames is two frames per wave.)
    int64_t c = 0;
    for (int64_t r0 = 0; r0 < B; r0 += CH, ++c) {
        const int64_t n = std::min(CH, B - r0);
        const ChunkWs& w = w2[c % NBUF];
        hipStream_t sp = st;                            // the stream of this chunk's first projection pass
        if (c >= NBUF) RML_HIP(hipStreamWaitEvent(sp, ev_done[c % NBUF], 0));    // workspace reuse
        const int FT = (int)((n + kTile - 1) / kTile);
        ProjOut o{};
        int64_t off = 0;
        for (int pl = 0; pl < 3; ++pl)
            if (mask & (1u << pl)) {
                o.q[pl] = grid_ok ? w.q + off : nullptr;
                off += pl == 0 ? (int64_t)X * Z : (pl == 1 ? (int64_t)Y * Z : (int64_t)X * Y);
            }
        o.sel = mask & RML_MASK_ALL;
        o.qstride = m->Dq; o.qrow = grid_ok ? w.q : nullptr; o.qD = m->D;
        o.row_isum = w.isum; o.row_isq = w.isq; o.row_flags = w.flags; o.scale_div = scale_div;
        o.share_cu = 1;
        o.q_rmw = ctx->opt.code_rmw >= 0 ? ctx->opt.code_rmw : rml_code_rmw(m->D, frame_elems * (vdtype == RML_VOL_U8 ? 1 : 4), derive, vdtype == RML_VOL_U8);
        const void* Vc = static_cast<const unsigned char*>(V) + r0 * frame_elems * (vdtype == RML_VOL_U8 ? 1 : 4);
        const int32_t* ijkc = ijk ? ijk + r0 * 3 : nullptr;
        // fused derive -> slice: the first pass derives (i,j,k) per frame and slices there in one launch; a second pass (float rows
        // for tiles that left the code grid) is a plain slice at the indices the first one wrote
        int32_t* ijkd = derive ? (ijk_out ? ijk_out + r0 * 3 : w.ijk) : nullptr;
        auto first_pass = [&](const ProjOut& po, hipStream_t ps) -> int {
            if (derive) return rml_launch_derive_slice(ctx, Vc, vdtype, n, X, Y, Z, 1, ijkd, nullptr, po, ps);
            return rml_launch_project(ctx, Vc, vdtype, n, X, Y, Z, mode, ijkc, po, ps);
        };
        // uint8 volumes are on the code grid by construction: one projection pass (codes + statistics), the exact GEMM on
        // every tile, no flag kernels, no predicated second pass and no predicated float64 GEMM launch
        const bool u8_exact = grid_ok && vdtype == RML_VOL_U8;
        if (u8_exact) {
            o.row_flags = nullptr;
            rml_prof_mark(ctx, st);
            rc = first_pass(o, st);
            rml_prof_mark(ctx, st);
            if (ctx->profiling) ctx->prof_frames += n;
            if (rc) return rc;
            RML_HIP(hipEventRecord(ev_proj[c % NBUF], st));
            RML_HIP(hipStreamWaitEvent(aux, ev_proj[c % NBUF], 0));
            rml_prof_mark_gemm(ctx, aux);
            rc = run_chunk(ctx, m, RML_PATH_I8, n, w.q, m->Dq, w.isum, w.isq, nullptr, nullptr, nullptr, w, out.at(r0, m->C, m->P), aux,
                           /*tiles_done=*/true, nullptr, 0, /*all_exact_known=*/true, /*allow_big=*/!small_gemm);
            rml_prof_mark_gemm(ctx, aux);
            if (ctx->profiling) ctx->prof_ops_g += 2.0 * (double)n * (double)m->M * (double)m->D;
            if (rc) return rc;
            RML_HIP(hipEventRecord(ev_done[c % NBUF], aux));
            continue;
        }
        // With a model on the code grid the caller's stream carries NOTHING but the first projection pass of every chunk (codes
        // + statistics): the tile decision, the predicated second pass (float rows for tiles that left the grid: a no-op
        // otherwise) and the digit planes follow on the second stream, in front of the chunk's GEMMs.  (Round 2 had them between
        // the projection launches: three launches and their gaps per chunk on the stream the step waits for.)
        hipStream_t s2 = grid_ok ? side : st;           // the stream of pass 2 and its followers
        if (grid_ok) {
            // pass 1: codes + statistics only (the exact path needs nothing else)
            rml_prof_mark(ctx, sp);
            rc = first_pass(o, sp);
            rml_prof_mark(ctx, sp);
            if (ctx->profiling) ctx->prof_frames += n;
            if (rc) return rc;
            RML_HIP(hipEventRecord(ev_proj[c % NBUF], sp));
            RML_HIP(hipStreamWaitEvent(side, ev_proj[c % NBUF], 0));
            const int group = (use_dig || (!small_gemm && use_big_gemm(m, n, gemm_cus, ctx->opt.gemm_big))) ? 2 : 1;      // the same decision run_chunk takes for this chunk
            hipLaunchKernelGGL(k_tile_flags, dim3((FT + group - 1) / group + 1), dim3(128 * group), 0, side, w.flags, n, FT, 0, 1, w.tile_exact,
                               w.all_exact, group);
        }
        // pass 2: float rows + norms for the f32 path; a no-op when every tile is exact
        ProjOut of{};
        off = 0;
        for (int pl = 0; pl < 3; ++pl)
            if (mask & (1u << pl)) {
                of.p[pl] = w.f32 + off; of.stride[pl] = m->Df;
                off += pl == 0 ? (int64_t)X * Z : (pl == 1 ? (int64_t)Y * Z : (int64_t)X * Y);
            }
        of.sel = mask & RML_MASK_ALL;
        of.scale_div = scale_div; of.prow = w.f32; of.pD = m->D; of.pstride = m->Df; of.row_nsq = w.nsq;
        of.share_cu = o.share_cu;
        of.no_pad = grid_ok ? 1 : 0;
        of.skip_if_set = grid_ok ? w.all_exact : nullptr;
        if (!grid_ok) of.row_flags = w.flags;
        if (!grid_ok) rml_prof_mark(ctx, st);
        // (a model off the code grid has no first pass: this one derives as well)
        rc = (derive && grid_ok) ? rml_launch_project(ctx, Vc, vdtype, n, X, Y, Z, RML_MODE_SLICE, ijkd, of, s2) : first_pass(of, s2);
        if (!grid_ok) { rml_prof_mark(ctx, st); if (ctx->profiling) ctx->prof_frames += n; }
        if (rc) return rc;
        if (use_dig) {
            // digit planes of the float rows (skipped with the float rows when every tile is exact); with an exact model the
            // general tiles are re-decided here, otherwise run_chunk decides
            hipLaunchKernelGGL(k_digit_rows, dim3((unsigned)n), dim3(256), 0, s2, w.f32, m->Df, m->D, m->Dq, w.dig_plane, w.dig, w.dnsq,
                               w.dflags, m->dig_c0, 2147483648.0 / m->dig_s, of.skip_if_set);
            if (grid_ok)
                hipLaunchKernelGGL(k_tile_dig, dim3((FT + 1) / 2), dim3(256), 0, s2, w.dflags, n, FT, w.tile_exact, of.skip_if_set);
            RML_HIP(hipGetLastError());
        }
        if (!grid_ok) {
            RML_HIP(hipEventRecord(ev_proj[c % NBUF], st));
            RML_HIP(hipStreamWaitEvent(aux, ev_proj[c % NBUF], 0));
        }
        rml_prof_mark_gemm(ctx, aux);
        rc = run_chunk(ctx, m, grid_ok ? RML_PATH_AUTO : RML_PATH_F64, n, grid_ok ? w.q : nullptr, m->Dq, w.isum, w.isq, w.flags, w.f32, w.nsq, w,
                       out.at(r0, m->C, m->P), aux, /*tiles_done=*/grid_ok, nullptr, 0, false, /*allow_big=*/!small_gemm, /*dig_ready=*/use_dig,
                       /*defer_finish=*/false);
        rml_prof_mark_gemm(ctx, aux);
        if (ctx->profiling) ctx->prof_ops_g += 2.0 * (double)n * (double)m->M * (double)m->D;
        if (rc) return rc;
        RML_HIP(hipEventRecord(ev_done[c % NBUF], aux));
    }
    RML_HIP(hipEventRecord(ctx->ev_join, aux));
    // join: the caller's stream continues after the last GEMMs and their finish
    RML_HIP(hipStreamWaitEvent(caller, ctx->ev_join, 0));
    return RML_OK;
}
}  // namespace

// libsvm's own Platt coefficients (SVC(probability=True): sk:svm/_base.py _probA / _probB, one pair per class pair):
// uploaded once, at load time, so that rml_svm_pairwise_proba is an ordinary asynchronous launch
extern "C" int rml_svm_set_platt(rml_ctx* ctx, rml_svm* m, const double* probA, const double* probB) {
    RML_REQUIRE(ctx && m && probA && probB, RML_ERR_INVALID, "rml_svm_set_platt: NULL argument");
    RML_HIP(hipSetDevice(ctx->device));
    std::vector<double> ab(2 * kMaxP, 0.0);
    for (int p = 0; p < m->P; ++p) { ab[p] = probA[p]; ab[kMaxP + p] = probB[p]; }
    if (!m->platt) RML_HIP(hipMalloc(reinterpret_cast<void**>(&m->platt), ab.size() * sizeof(double)));
    RML_HIP(hipMemcpy(m->platt, ab.data(), ab.size() * sizeof(double), hipMemcpyHostToDevice));
    return RML_OK;
}

extern "C" int rml_svm_pairwise_proba(rml_ctx* ctx, const rml_svm* m, const double* dec_ovo, int64_t N, double* proba, void* stream) {
    RML_REQUIRE(ctx && m && N >= 0, RML_ERR_INVALID, "rml_svm_pairwise_proba: bad arguments");
    if (N == 0) return RML_OK;
    RML_REQUIRE(dec_ovo && proba, RML_ERR_INVALID, "rml_svm_pairwise_proba: NULL array");
    RML_REQUIRE(m->platt != nullptr, RML_ERR_STATE, "rml_svm_pairwise_proba: the model has no Platt coefficients (rml_svm_set_platt)");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_pairwise_proba, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, dec_ovo, N, m->C, m->platt, m->platt + kMaxP, proba);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

// ---- linear classifier --------------------------------------------------------------------
extern "C" int rml_linear_load(rml_ctx* ctx, const double* coef, const double* intercept, int n_classes, int64_t D,
                               const double* calib_a, const double* calib_b, rml_linear** out) {
    RML_REQUIRE(ctx && coef && intercept && out && D > 0, RML_ERR_INVALID, "rml_linear_load: bad arguments");
    RML_REQUIRE(n_classes >= 2 && n_classes <= kMaxC, RML_ERR_UNSUPPORTED, "rml_linear_load: %d classes", n_classes);
    RML_REQUIRE((calib_a == nullptr) == (calib_b == nullptr), RML_ERR_INVALID, "rml_linear_load: calib_a/calib_b must both be given");
    *out = nullptr;
    RML_HIP(hipSetDevice(ctx->device));
    rml_linear* m = new (std::nothrow) rml_linear();
    RML_REQUIRE(m != nullptr, RML_ERR_NOMEM, "rml_linear_load: out of host memory");
    m->D = D; m->C = n_classes; m->has_calib = calib_a != nullptr;
    const int rows = n_classes == 2 ? 1 : n_classes;
    std::vector<double> cf((size_t)n_classes * D, 0.0), ic(n_classes, 0.0);
    std::copy(coef, coef + (size_t)rows * D, cf.begin());
    std::copy(intercept, intercept + rows, ic.begin());
    int rc = dev_upload(&m->coef, cf);
    if (!rc) rc = dev_upload(&m->intercept, ic);
    if (!rc && m->has_calib) {
        std::vector<double> cal(2 * n_classes, 0.0);
        for (int c = 0; c < rows; ++c) { cal[c] = calib_a[c]; cal[n_classes + c] = calib_b[c]; }
        rc = dev_upload(&m->calib, cal);
    }
    if (rc) { rml_linear_free(ctx, m); return rc; }
    *out = m;
    return RML_OK;
}

extern "C" int rml_linear_free(rml_ctx* ctx, rml_linear* m) {
    if (!m) return RML_OK;
    if (ctx) (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (m->coef) (void)hipFree(m->coef);
    if (m->intercept) (void)hipFree(m->intercept);
    if (m->calib) (void)hipFree(m->calib);
    delete m;
    return RML_OK;
}

extern "C" int rml_linear_decision(rml_ctx* ctx, const rml_linear* m, const float* feat, int64_t ld_feat, int64_t N,
                                   double* dec, double* proba, int32_t* label, int32_t* label_calib, void* stream) {
    RML_REQUIRE(ctx && m && N >= 0, RML_ERR_INVALID, "rml_linear_decision: bad arguments");
    if (N == 0) return RML_OK;
    RML_REQUIRE(feat && ld_feat >= m->D, RML_ERR_INVALID, "rml_linear_decision: bad arguments");
    RML_REQUIRE(!(proba || label_calib) || m->has_calib, RML_ERR_STATE, "rml_linear_decision: model has no calibrators");
    RML_HIP(hipSetDevice(ctx->device));
    if (N == 0) return RML_OK;
    hipLaunchKernelGGL(k_linear, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       feat, ld_feat, N, m->D, m->C, m->coef, m->intercept, m->calib, (int)m->has_calib, dec, proba, label, label_calib);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
