// 3-D radar volume -> (xz, yz, xy) projection kernels for gfx950 (MI355X).
//
// Reference sites replaced (paths in goruck/radar-ml):
//   slices            predict.py:102-107, ground_truth_samples.py:413-419   (RML_MODE_SLICE)
//   max-projection    BASELINE.json north_star / SURVEY.md §0.1 D1          (RML_MODE_MAX)
//   sum reductions    common.py:51-53 (DerivedTarget.find_max_indices)      (RML_MODE_SUM)
//   feature assembly  common.py:141-148 (process_samples at zoom 1): ravel + concatenate in
//                     (xz, yz, xy) order, optional "/ RADAR_MAX" in float32
//
// Design (HBM-bound streaming reduction, one workgroup per frame, one pass over V):
//   z is the contiguous axis.  Every lane loads float4 along z (16 B/lane, a wave covers
//   64/LPR complete (i,j) rows per load instruction, LPR = lanes per row).  A workgroup's
//   4 waves own NSLOT = 4*64/LPR row slots; slot s handles rows j = s + NSLOT*m, the same
//   for every i, so
//     yz[j,k] = op_i V   lives in registers of exactly one lane (NM float4 accumulators),
//     xz[i,k] = op_j V   is reduced in-lane over m, across the row slots of a wave with
//                        lane shuffles and across the 4 waves with LDS float atomics
//                        (ds_max_f32 / ds_add_f32) -- no barrier inside the streaming loop,
//     xy[i,j] = op_k V   is an in-lane float4 reduce + a butterfly over the LPR lanes of a row.
//   The next i-plane is prefetched into registers while the current one is reduced.
//   Epilogue: optional float32 division by RADAR_MAX (bit-identical to NumPy's), uint8 codes
//   (biased by 128 for the signed i8 MFMA of the exact SVM path), per-row sum / sum of squares
//   and the "all integers in [0,255]" flag, all fused -- the feature row never goes back to HBM
//   between projection and assembly.
//
// Algorithmic HBM bytes per frame (mode MAX): 4*X*Y*Z read + 4*D written (+ D code bytes).
#include "project_shared.h"

namespace {

using namespace rmlproj;

// ------------------------------------------------------------------------------------------
// Fast path: Z % 4 == 0, Z/4 <= 64, Y <= NM * NSLOT.  One workgroup (256 threads) per frame.
// ------------------------------------------------------------------------------------------
// PRED = launch predicated on a device flag (a separate instantiation so that profiles do not mix the
// no-op launches of the predicated second pass of rml_project_svm with the real ones).
// FULL = no padding anywhere (Y == NM*NSLOT and Z/4 == LPR): the streaming loop carries no masks.
// Loads are ALWAYS unconditional: padded rows / lanes re-read a valid neighbour (clamped offset) and
// are neutralised afterwards (a duplicate is harmless for max; sum selects 0) -- a conditional load
// makes hipcc branch around every load and drain vmcnt per element, which de-pipelines the stream.
template <typename VT, int MODE, int LPR, int NM, bool FULL, bool PRED>
__global__ __launch_bounds__(kThreads) void k_project_fast(ProjParams a) {
    extern __shared__ __align__(16) float lds[];
    if constexpr (PRED) { if (*a.o.skip_if_set) return; }
    const int X = a.X, Y = a.Y, Z = a.Z, ZQ = a.ZQ;
    float* xz_lds = lds;                 // X*Z
    float* xy_lds = lds + (size_t)X * Z; // X*Y
    int64_t* red = reinterpret_cast<int64_t*>(xy_lds + (((size_t)X * Y + 3) & ~(size_t)3));

    constexpr int RPW = 64 / LPR;       // rows per wave-load
    constexpr int NSLOT = 4 * RPW;      // row slots per workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = wave * RPW + lane / LPR;
    const int kq = lane % LPR;
    const bool act = FULL || (kq < ZQ);
    const int64_t b = blockIdx.x;
    typedef typename Quad<VT>::T QT;
    const QT* __restrict__ Vb = reinterpret_cast<const QT*>(static_cast<const VT*>(a.V) + b * (int64_t)X * Y * Z);

    const float id = Op<MODE>::ident();
    const float4 id4 = make_float4(id, id, id, id);
    for (int idx = tid; idx < X * Z; idx += kThreads) xz_lds[idx] = Op<MODE>::lds_ident();
    __syncthreads();

    float4 yz[NM];
    int roff[NM];
    bool rv[NM];
    const int kqc = FULL ? kq : min(kq, ZQ - 1);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int j = slot + NSLOT * m;
        rv[m] = FULL || (act && (j < Y));
        roff[m] = (FULL ? j : min(j, Y - 1)) * ZQ + kqc;
        yz[m] = id4;
    }
    const int plane = Y * ZQ;

    // (`#pragma unroll 2` below is not honoured -- hipcc reports "loop not unrolled", the ISA shows NM loads, NM counted waits and
    // nothing in flight at the loop's end --; a hand-written rolling window over the planes, 101 -> 131 registers at the same three
    // workgroups per CU, measured the same: tools/exp/README.md round 6)
#pragma unroll 2
    for (int i = 0; i < X; ++i) {
        const QT* __restrict__ Vi = Vb + (int64_t)i * plane;
        float4 cur[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) cur[m] = ld_stream(Vi + roff[m]);
        if constexpr (!FULL && MODE == RML_MODE_SUM) {
#pragma unroll
            for (int m = 0; m < NM; ++m) cur[m] = rv[m] ? cur[m] : id4;
        }
        float4 p = id4;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            yz[m] = op4<MODE>(yz[m], cur[m]);
            p = op4<MODE>(p, cur[m]);
            float r = Op<MODE>::f(Op<MODE>::f(cur[m].x, cur[m].y), Op<MODE>::f(cur[m].z, cur[m].w));
            r = row_reduce<MODE, LPR>(r);
            const int j = slot + NSLOT * m;
            if (kq == 0 && (FULL || j < Y)) xy_lds[i * Y + j] = r;
        }
        // combine the row slots that share this wave, then the 4 waves through LDS atomics
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            float4 q;
            q.x = __shfl_xor(p.x, off); q.y = __shfl_xor(p.y, off);
            q.z = __shfl_xor(p.z, off); q.w = __shfl_xor(p.w, off);
            p = op4<MODE>(p, q);
        }
        if (lane < LPR && act) {
            float* dst = xz_lds + i * Z + 4 * kq;
            Op<MODE>::lds_atomic(dst + 0, p.x);
            Op<MODE>::lds_atomic(dst + 1, p.y);
            Op<MODE>::lds_atomic(dst + 2, p.z);
            Op<MODE>::lds_atomic(dst + 3, p.w);
        }
    }

    Emitter em(a, b);
    em.rmw = false;                                     // plain stores here (see Emitter::rmw)
    // yz straight from registers: lane owns (j_m, 4kq..4kq+3)
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        int j = slot + NSLOT * m;
        if (rv[m]) em.put4(1, (int64_t)j * Z + 4 * kq, yz[m]);
    }
    __syncthreads();
    // xz and xy from LDS, coalesced
    const int nxz4 = (X * Z) >> 2;
    for (int idx = tid; idx < nxz4; idx += kThreads) {
        const float4 q = *reinterpret_cast<const float4*>(xz_lds + idx * 4);         // combined through Op::lds_atomic: its image
        em.put4(0, (int64_t)idx * 4, make_float4(Op<MODE>::lds_value(q.x), Op<MODE>::lds_value(q.y), Op<MODE>::lds_value(q.z), Op<MODE>::lds_value(q.w)));
    }
    const int nxy = X * Y;
    const int nxy4 = nxy >> 2;
    for (int idx = tid; idx < nxy4; idx += kThreads)
        em.put4(2, (int64_t)idx * 4, *reinterpret_cast<const float4*>(xy_lds + idx * 4));
    for (int idx = nxy4 * 4 + tid; idx < nxy; idx += kThreads) em.put1(2, idx, xy_lds[idx]);
    em.finish(red);
}

// ------------------------------------------------------------------------------------------
// Row-group path for rows that are not a power-of-two number of float4 (Walabot arena: Z = 176, 44 float4):
// the workgroup has T = R * ZQ threads with R rows chosen so that T is a multiple of 64 (R = 16, T = 704 =
// 11 waves at ZQ = 44), thread t owns column quad kq = t % ZQ of the rows j = t / ZQ + R * m -- every lane
// carries data (k_project_fast would idle 20 of 64 lanes there).  yz stays in registers (NM float4), xz is
// combined in-lane over m and across the R row slots with LDS float atomics, xy is a segmented reduction over
// the contiguous lanes of a row (rows straddle wave boundaries, so segment heads finish with an LDS atomic).
// ------------------------------------------------------------------------------------------
template <typename VT, int MODE, int NM, bool PRED>
__global__ __launch_bounds__(1024) void k_project_rowgroup(ProjParams a, int R) {
    extern __shared__ __align__(16) float lds[];
    if constexpr (PRED) { if (*a.o.skip_if_set) return; }
    const int X = a.X, Y = a.Y, Z = a.Z, ZQ = a.ZQ;
    float* xz_lds = lds;
    float* xy_lds = lds + (size_t)X * Z;
    int64_t* red = reinterpret_cast<int64_t*>(xy_lds + (((size_t)X * Y + 3) & ~(size_t)3));
    const int T = blockDim.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int slot = tid / ZQ, kq = tid - slot * ZQ;
    const int64_t b = blockIdx.x;
    typedef typename Quad<VT>::T QT;
    const QT* __restrict__ Vb = reinterpret_cast<const QT*>(static_cast<const VT*>(a.V) + b * (int64_t)X * Y * Z);
    const float id = Op<MODE>::ident();
    const float4 id4 = make_float4(id, id, id, id);
    for (int idx = tid; idx < X * Z; idx += T) xz_lds[idx] = Op<MODE>::lds_ident();
    for (int idx = tid; idx < X * Y; idx += T) xy_lds[idx] = Op<MODE>::lds_ident();
    __syncthreads();

    float4 yz[NM];
    int roff[NM];
    bool rv[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int j = slot + R * m;
        rv[m] = j < Y;
        roff[m] = min(j, Y - 1) * ZQ + kq;
        yz[m] = id4;
    }
    // segmented reduction over the lanes of a row: lane l+d belongs to my row iff it is in this wave and its
    // row slot equals mine; the masks depend on the lane only (the same for every m)
    unsigned same = 0;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        const int d = 1 << s;
        if (lane + d < 64 && (tid + d) / ZQ == slot) same |= 1u << s;
    }
    const bool head = (lane == 0) || ((tid - 1) / ZQ != slot);
    const int plane = Y * ZQ;

#pragma unroll 2
    for (int i = 0; i < X; ++i) {
        const QT* __restrict__ Vi = Vb + (int64_t)i * plane;
        float4 cur[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) cur[m] = ld_stream(Vi + roff[m]);
        if constexpr (MODE == RML_MODE_SUM) {
#pragma unroll
            for (int m = 0; m < NM; ++m) cur[m] = rv[m] ? cur[m] : id4;
        }
        float4 p = id4;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            yz[m] = op4<MODE>(yz[m], cur[m]);
            p = op4<MODE>(p, cur[m]);
            float r = Op<MODE>::f(Op<MODE>::f(cur[m].x, cur[m].y), Op<MODE>::f(cur[m].z, cur[m].w));
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const float o = __shfl_down(r, 1 << s);
                r = ((same >> s) & 1u) ? Op<MODE>::f(r, o) : r;
            }
            if (head && rv[m]) Op<MODE>::lds_atomic(xy_lds + i * Y + slot + R * m, r);
        }
        float* dst = xz_lds + i * Z + 4 * kq;
        Op<MODE>::lds_atomic(dst + 0, p.x);
        Op<MODE>::lds_atomic(dst + 1, p.y);
        Op<MODE>::lds_atomic(dst + 2, p.z);
        Op<MODE>::lds_atomic(dst + 3, p.w);
    }

    Emitter em(a, b);
    em.rmw = false;                                     // plain stores here (see Emitter::rmw)
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int j = slot + R * m;
        if (rv[m]) em.put4(1, (int64_t)j * Z + 4 * kq, yz[m]);
    }
    __syncthreads();
    auto val4 = [](const float4& q) { return make_float4(Op<MODE>::lds_value(q.x), Op<MODE>::lds_value(q.y), Op<MODE>::lds_value(q.z), Op<MODE>::lds_value(q.w)); };
    const int nxz4 = (X * Z) >> 2;
    for (int idx = tid; idx < nxz4; idx += T)
        em.put4(0, (int64_t)idx * 4, val4(*reinterpret_cast<const float4*>(xz_lds + idx * 4)));
    const int nxy = X * Y;
    const int nxy4 = nxy >> 2;
    for (int idx = tid; idx < nxy4; idx += T)
        em.put4(2, (int64_t)idx * 4, val4(*reinterpret_cast<const float4*>(xy_lds + idx * 4)));
    for (int idx = nxy4 * 4 + tid; idx < nxy; idx += T) em.put1(2, idx, Op<MODE>::lds_value(xy_lds[idx]));
    em.finish(red);
}

// ------------------------------------------------------------------------------------------
// Wave-per-frame path for small frames with long rows (Walabot arena 22 x 31 x 176: 32 < Z/4 <= 64, Y <= 32).
// ONE WAVE owns a frame and streams it row by row: lane = float4 column kq of the row (lanes >= Z/4 re-read the
// last column: a duplicate is harmless for max, sum zeroes it), so
//   * xz[i,k] = op_j V is an IN-LANE reduction over the rows of plane i and leaves as one coalesced store per plane,
//   * yz[j,k] = op_i V is an in-lane reduction over the planes (NY float4 accumulators per lane),
//   * xy[i,j] = op_k V: the lane's float4 is folded to one value and parked in a private LDS image [row][lane];
//     once per plane lane j folds row j with bank-conflict-free float4 reads (wave-private: no barrier).
// No atomics, no barrier, no LDS initialisation.  The kernel is persistent (a wave walks frames gw, gw + #waves, ...): the rows of
// the next group -- also across plane and frame boundaries -- are in flight in a second register buffer while the
// current group is reduced, so a frame's epilogue (yz stores, code statistics) overlaps the next frame's loads.
// G = rows per buffer: G == NY (a whole plane per buffer, ~400 VGPRs, one wave per SIMD, 22-44 KB in flight per
// wave) or NY/4 (<= 256 VGPRs, two waves per SIMD).
// ------------------------------------------------------------------------------------------
template <typename VT, int MODE, int NY, int G, bool PRED>
__global__ __launch_bounds__(256, (NY + 2 * G) * 4 + 40 > 256 ? 1 : 2) void k_project_wave(ProjParams a) {
    if constexpr (PRED) { if (*a.o.skip_if_set) return; }
    constexpr int NG = (NY + G - 1) / G;                // row groups per plane
    static_assert(NG == 1 || NG % 2 == 0, "the two row buffers must alternate statically");
    constexpr int U = NG == 1 ? 2 : NG;                 // steps per trip of the main loop (buffer = step parity)
    // Short rows (Z/4 = 32 or 16 quads): RPL = 2 or 4 consecutive rows form one VIRTUAL row of 64 quads -- the plane is
    // viewed as (Y/RPL) x (RPL*Z), which is the same memory, so every load instruction is still one contiguous <= 1 KB
    // piece and yz keeps its layout (row j' of the view = rows RPL*j' .. of yz).  Only the epilogues know about it: xz folds
    // the RPL lane groups of a virtual row, and xy lane j reads the lane group of its real row.
    const int X = a.X, Yr = a.Y, Zr = a.Z, ZQr = a.ZQ;
    const int RPL = a.rpl;                              // the launcher's decision: kernel and launcher cannot disagree
    const int Y = Yr / RPL, Z = Zr * RPL, ZQ = ZQr * RPL;           // the view the streaming loop works on
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t cf = (int64_t)blockIdx.x * 4 + wave;        // frame being reduced
    if (cf >= a.B) return;
    typedef typename Quad<VT>::T QT;
    const QT* __restrict__ Vall = reinterpret_cast<const QT*>(a.V);
    const int64_t fq = (int64_t)X * Y * ZQ;             // quads per frame
    const int pq = Y * ZQ;                              // quads per plane
    const bool act = lane < ZQ;
    const uint32_t kqc = (uint32_t)(act ? lane : ZQ - 1);      // unsigned: global_load saddr + 32-bit voffset form
    const float id = Op<MODE>::ident();
    const float4 id4 = make_float4(id, id, id, id);

    // load cursor: one group ahead of the reduction.  Frames are assigned statically (wave w: frames w, w + #waves, ...).
    // Dynamic assignment through an atomic ticket per frame was tried for the co-running case (a workgroup that starts late
    // behind a GEMM workgroup delays the whole persistent grid): reading the ticket back costs a full drain of the loads in
    // flight once per frame when the compiler schedules it (s_waitcnt vmcnt(0) behind the atomic), and an asynchronous
    // inline-asm ticket cannot be made safe against register copies -- dropped.
    int64_t lf = cf;
    int li = 0;
    const QT* __restrict__ lV = Vall + lf * fq;
    auto next_plane = [&]() {
        ++li;
        lV += pq;
        if (li == X) {                                  // next frame of this wave; past the end: re-read (never consumed)
            li = 0;
            const int64_t nf = lf + stride;
            lf = nf < a.B ? nf : lf;
            lV = Vall + lf * fq;
        }
    };
    float4 buf[2][G];
    // xy staging: [row j][column lane] per wave, row stride 68 floats: lane j's float4 reads hit 64 distinct banks
    constexpr int kXyStride = 68;
    __shared__ __align__(16) float xy_stage[4][NY * kXyStride];
    float* xyl = xy_stage[wave];
    // row j of a plane starts at byte offset min(j, Y-1) * rowb (rows past Y re-read the last one)
    const uint32_t rowb = (uint32_t)ZQ * sizeof(QT);
    auto fetch = [&](float4 (&dst)[G], auto gc) {
        constexpr int g = decltype(gc)::value;
        const char* __restrict__ base = reinterpret_cast<const char*>(lV);
        uint32_t rowb_t = rowb;                         // opaque per step: hipcc would otherwise keep NY hoisted row offsets
        asm volatile("" : "+s"(rowb_t));                // alive across the loop (and spill them)
        static_for<G>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            constexpr int j = g * G + r;
            if constexpr (j < NY) {
                const uint32_t ro = (uint32_t)(NY == 31 || j < Y ? j : Y - 1) * rowb_t;   // NY == 31: exact (no clamp)
                dst[r] = ld_stream(reinterpret_cast<const QT*>(base + ro) + kqc);
            }
        });
    };
    constexpr int PPT = NG == 1 ? 2 : 1;                // planes per trip; whole-plane buffers need an even X
    Emitter em(a, cf);
    em.rmw = false;                                     // plain stores here (see Emitter::rmw)
    extern __shared__ __align__(16) unsigned char wave_dyn[];      // codes-only launches: the wave's code stage (see Emitter::stage)
    if (a.stage_bytes) em.set_stage(wave_dyn + wave * a.stage_bytes, (X * Zr + 15) & ~15);
    fetch(buf[0], std::integral_constant<int, 0>{});    // group 0 of the first plane
    for (; cf < a.B; cf += stride) {
        em.reset(cf);
        float4 yz[NY];
        static_for<NY>([&](auto jc) { yz[decltype(jc)::value] = id4; });
        float4 xz = id4;
        for (int ci = 0; ci < X; ci += PPT) {
            static_for<U>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                constexpr int g = NG == 1 ? 0 : u;                  // group reduced in this step
                constexpr int gn = NG == 1 ? 0 : (u + 1) % NG;      // group fetched in this step
                if constexpr (gn == 0) next_plane();
                // unconditional prefetch (past the last frame it re-reads): a conditional one would make hipcc drain vmcnt
                fetch(buf[(u + 1) & 1], std::integral_constant<int, gn>{});
                // keep the software pipeline as written: left alone, hipcc hoists the loads of ALL later groups above this
                // group's reduction (and spills the buffers it then needs)
                __builtin_amdgcn_sched_barrier(0);
                static_for<G>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    constexpr int j = g * G + r;
                    if constexpr (j < NY) {
                        float4 v = buf[u & 1][r];
                        if constexpr (MODE == RML_MODE_SUM) v = (act && j < Y) ? v : id4;
                        yz[j] = op4_raw<MODE>(yz[j], v);
                        // pin the update here: yz is only "needed" at the loop back-edge, and instruction selection would
                        // otherwise place all NY updates there and keep every loaded row alive for the whole trip
                        asm volatile("" : "+v"(yz[j].x), "+v"(yz[j].y), "+v"(yz[j].z), "+v"(yz[j].w));
                        xz = op4_raw<MODE>(xz, v);
                        xyl[j * kXyStride + lane] = op_raw<MODE>(op_raw<MODE>(v.x, v.y), op_raw<MODE>(v.z, v.w));
                    }
                });
                // materialise xz here: its only use sits in a lane-conditional block and hipcc would sink the whole reduction
                // there, keeping every row of the plane alive until then
                asm volatile("" : "+v"(xz.x), "+v"(xz.y), "+v"(xz.z), "+v"(xz.w));
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (g == NG - 1) {                        // plane i of frame cf is complete
                    const int i = ci + (NG == 1 ? u : 0);
                    int lane_e = lane;                  // opaque: keeps the store addresses out of loop-invariant hoisting
                    asm volatile("" : "+v"(lane_e));    // (hipcc would hold NY pointer pairs across the streaming loop)
                    // xz[i, k]: fold the RPL lane groups of the virtual row (real rows of different parity)
                    for (int off = ZQr; off < ZQ; off <<= 1) {
                        float4 o;
                        o.x = __shfl_xor(xz.x, off); o.y = __shfl_xor(xz.y, off); o.z = __shfl_xor(xz.z, off); o.w = __shfl_xor(xz.w, off);
                        xz = op4<MODE>(xz, o);
                    }
                    if (lane_e < ZQr) em.put4(0, (int64_t)i * Zr + 4 * lane_e, xz);
                    // xy[i, j]: lane j folds the per-column partials of real row j = lane group j % RPL of virtual row j / RPL
                    // (transposed, conflict-free b128 reads)
                    for (int jr = lane_e; jr < Yr; jr += 64) {          // more than 64 real rows when 4 of them share a virtual row
                        const float* rowp = xyl + (jr / RPL) * kXyStride + (jr % RPL) * ZQr;
                        float m = id;
                        for (int q = 0; q < ZQr; q += 4) {
                            const float4 t = *reinterpret_cast<const float4*>(rowp + q);
                            m = Op<MODE>::f(m, Op<MODE>::f(Op<MODE>::f(t.x, t.y), Op<MODE>::f(t.z, t.w)));
                        }
                        em.put1(2, (int64_t)i * Yr + jr, m);
                    }
                    xz = id4;
                }
            });
        }
        int lane_f = lane, Y_f = Y;
        asm volatile("" : "+v"(lane_f), "+s"(Y_f));
        const bool act_f = lane_f < ZQ;
        static_for<NY>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if (j < Y_f && act_f) em.put4(1, (int64_t)j * Z + 4 * lane_f, yz[j]);
        });
        em.flush_wave(lane_f);
        em.finish_wave(lane_f);
    }
}

template <typename VT, int MODE, int NY, int G>
void launch_wave(const ProjParams& pp_in, int num_cu, hipStream_t st) {
    ProjParams pp = pp_in;
    constexpr int per_cu_max = (NY + 2 * G) * 4 + 40 > 256 ? 1 : 2;
    const size_t mine = (size_t)4 * NY * 68 * sizeof(float);
    pp.stage_bytes = code_stage_bytes(pp, sizeof(VT));
    // `mine` is the kernel's STATIC xy stage; the dynamic part (code stage / pad) may take what is left of the CU's 160 KB
    if (mine + (size_t)4 * pp.stage_bytes > 160 * 1024) pp.stage_bytes = 0;      // no room: direct stores
    const size_t stage = (size_t)4 * pp.stage_bytes;
    int per_cu = pp.o.share_cu ? 1 : per_cu_max;
    if (per_cu * (mine + stage) > 160 * 1024) per_cu = 1;      // (measured: one or two of these workgroups per CU stream equally fast)
    const int64_t want = (pp.B + 3) / 4;
    const int64_t cap = (int64_t)num_cu * per_cu;
    dim3 grid((unsigned)(want < cap ? want : cap)), block(kThreads);
    // beside a GEMM: the request is padded past half of the CU's LDS, so that the dispatcher cannot put two of these
    // persistent workgroups on one CU (and none on another) while GEMM workgroups (69.6 KB) still fit next to one
    size_t pad = stage;
    if (pp.o.share_cu && per_cu == 1 && !pp.o.no_pad && mine + stage < 81 * 1024) pad = 81 * 1024 - mine;
    if (pp.o.skip_if_set) {
        RML_MAX_DYN_LDS(160 * 1024 - (int)mine, &k_project_wave<VT, MODE, NY, G, true>);
        hipLaunchKernelGGL((k_project_wave<VT, MODE, NY, G, true>), grid, block, pad, st, pp);
    } else {
        RML_MAX_DYN_LDS(160 * 1024 - (int)mine, &k_project_wave<VT, MODE, NY, G, false>);
        hipLaunchKernelGGL((k_project_wave<VT, MODE, NY, G, false>), grid, block, pad, st, pp);
    }
}

// returns true when the wave-per-frame kernel took the launch
// rows per load instruction of the wave-per-frame kernel (0 = shape not handled): 1 for long rows (32 < Z/4 <= 64), 2 / 4
// for rows of 32 / 16 quads when Y divides; the VIEW (Y / RPL rows of RPL * Z/4 quads) must have at most 32 rows
int wave_kernel_rpl(int ZQ, int Y) {
    if (ZQ > 32 && ZQ <= 64) return Y <= 32 ? 1 : 0;
    if (ZQ == 32 || ZQ == 16) {
        const int rpl = 64 / ZQ;
        return (Y % rpl == 0 && Y / rpl <= 32) ? rpl : 0;
    }
    return 0;
}

// knob (RML_OPT_WAVEFRAME): 0 = off, 1 = on (default; long rows always, short rows -- where k_project_fast is as fast stand-alone --
// only beside a GEMM), 2 = quarter-plane buffers everywhere, 3 = also short rows stand-alone
// Small batches stay on the workgroup-per-frame kernels: a single wave streams a 480 KB frame in ~50 us, which is what a
// B = 1 call would wait for (single-observation latency 167 -> 210 us when this kernel took every batch size).
bool wave_kernel_wanted(int ZQ, int Y, bool share_cu, int64_t B, int num_cu, int knob) {
    if (B < 2 * (int64_t)num_cu) return false;
    const int rpl = wave_kernel_rpl(ZQ, Y);
    if (knob == 0 || rpl == 0) return false;
    return rpl == 1 || share_cu || knob == 3;
}

// returns true when the wave-per-frame kernel took the launch
template <typename VT, int MODE>
bool try_launch_wave(const ProjParams& pp_in, int num_cu, hipStream_t st) {
    const int knob = pp_in.k_waveframe;
    if (!wave_kernel_wanted(pp_in.ZQ, pp_in.Y, pp_in.o.share_cu != 0, pp_in.B, num_cu, knob)) return false;
    ProjParams pp = pp_in;
    pp.rpl = wave_kernel_rpl(pp.ZQ, pp.Y);
    const int Y = pp.Y / wave_kernel_rpl(pp.ZQ, pp.Y);  // rows of the view
    // whole-plane buffers need an even number of planes; beside a GEMM the quarter-plane variant (206 VGPRs) leaves it room
    // (half-plane buffers, 267 registers and twice the bytes in flight per wave, were measured beside the GEMM in round 3:
    // no gain on either grid -- the projection in the pipeline is not bound by bytes in flight; tools/exp/README.md)
    const bool quarter = knob == 2 || (pp.X & 1) || pp.o.share_cu;
#define RML_WAVE_CASE(NYV)                                                                     \
    { if (quarter) launch_wave<VT, MODE, NYV, (NYV + 3) / 4>(pp, num_cu, st);                 \
      else launch_wave<VT, MODE, NYV, NYV>(pp, num_cu, st); return true; }
    if (Y <= 8) RML_WAVE_CASE(8)
    if (Y <= 16) RML_WAVE_CASE(16)
    if (Y <= 24) RML_WAVE_CASE(24)
    if (Y == 31) RML_WAVE_CASE(31)
    RML_WAVE_CASE(32)
#undef RML_WAVE_CASE
}

// ------------------------------------------------------------------------------------------
// Generic fallback: any (X,Y,Z).  Three coalesced passes over the frame (L2 resident after
// the first), no shape restrictions.  One workgroup per frame.
// ------------------------------------------------------------------------------------------
template <typename VT, int MODE>
__global__ __launch_bounds__(kThreads) void k_project_generic(ProjParams a) {
    __shared__ int64_t red[64];
    if (a.o.skip_if_set && *a.o.skip_if_set) return;
    const int X = a.X, Y = a.Y, Z = a.Z;
    const int64_t b = blockIdx.x;
    const VT* __restrict__ Vb = static_cast<const VT*>(a.V) + b * (int64_t)X * Y * Z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Emitter em(a, b);
    const float id = Op<MODE>::ident();
    // xz[i,k] = op_j V[i,j,k]
    for (int idx = tid; idx < X * Z; idx += kThreads) {
        int i = idx / Z, k = idx - i * Z;
        float r = id;
        for (int j = 0; j < Y; ++j) r = Op<MODE>::f(r, (float)Vb[((int64_t)i * Y + j) * Z + k]);
        em.put1(0, idx, r);
    }
    // yz[j,k] = op_i V[i,j,k]
    for (int idx = tid; idx < Y * Z; idx += kThreads) {
        float r = id;
        for (int i = 0; i < X; ++i) r = Op<MODE>::f(r, (float)Vb[(int64_t)i * Y * Z + idx]);
        em.put1(1, idx, r);
    }
    // xy[i,j] = op_k V[i,j,k]: one wave per row
    for (int row = wave; row < X * Y; row += kThreads / 64) {
        float r = id;
        for (int k = lane; k < Z; k += 64) r = Op<MODE>::f(r, (float)Vb[(int64_t)row * Z + k]);
        r = row_reduce<MODE, 64>(r);
        if (lane == 0) em.put1(2, row, r);
    }
    em.finish(red);
}

// ------------------------------------------------------------------------------------------
// Slice mode: planes through (i,j,k) of each frame, Python negative-index wrap.
// ------------------------------------------------------------------------------------------
template <typename VT>
__global__ __launch_bounds__(kThreads) void k_project_slice(ProjParams a) {
    __shared__ int64_t red[64];
    if (a.o.skip_if_set && *a.o.skip_if_set) return;
    const int X = a.X, Y = a.Y, Z = a.Z;
    const int64_t b = blockIdx.x;       // output row; several rows (targets) may share one frame
    const VT* __restrict__ Vb = static_cast<const VT*>(a.V) + (b / a.tpf) * (int64_t)X * Y * Z;
    int i = a.ijk[b * 3 + 0], j = a.ijk[b * 3 + 1], k = a.ijk[b * 3 + 2];
    i = i < 0 ? i + X : i; j = j < 0 ? j + Y : j; k = k < 0 ? k + Z : k;
    // out-of-range after wrapping would raise IndexError in the reference; clamp defensively
    i = min(max(i, 0), X - 1); j = min(max(j, 0), Y - 1); k = min(max(k, 0), Z - 1);
    const int tid = threadIdx.x;
    Emitter em(a, b);
    for (int idx = tid; idx < X * Z; idx += kThreads) {          // xz = V[:, j, :]
        int ii = idx / Z, kk = idx - ii * Z;
        em.put1(0, idx, (float)Vb[((int64_t)ii * Y + j) * Z + kk]);
    }
    for (int idx = tid; idx < Y * Z; idx += kThreads)            // yz = V[i, :, :]
        em.put1(1, idx, (float)Vb[(int64_t)i * Y * Z + idx]);
    for (int idx = tid; idx < X * Y; idx += kThreads)            // xy = V[:, :, k]
        em.put1(2, idx, (float)Vb[(int64_t)idx * Z + k]);
    em.finish(red);
}

// planes -> rows (common.process_samples at zoom 1 on already separate projections)
__global__ __launch_bounds__(kThreads) void k_assemble(const float* xz, const float* yz, const float* xy, ProjParams a) {
    __shared__ int64_t red[64];
    const int X = a.X, Y = a.Y, Z = a.Z;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    Emitter em(a, b);
    if (xz) for (int idx = tid; idx < X * Z; idx += kThreads) em.put1(0, idx, xz[b * (int64_t)X * Z + idx]);
    if (yz) for (int idx = tid; idx < Y * Z; idx += kThreads) em.put1(1, idx, yz[b * (int64_t)Y * Z + idx]);
    if (xy) for (int idx = tid; idx < X * Y; idx += kThreads) em.put1(2, idx, xy[b * (int64_t)X * Y + idx]);
    em.finish(red);
}

// float rows -> codes + stats (rml_quantize_rows).  A value is on the code grid iff it is
// bit-identical to float32(c / scale_div) (the "p / 255." of train.py:667) or to c itself.
__global__ __launch_bounds__(kThreads) void k_quantize_rows(const float* feat, int64_t D, int64_t ld, float scale_div, ProjParams a) {
    __shared__ int64_t red[64];
    const int64_t b = blockIdx.x;
    Emitter em(a, b);
    const bool scaled = scale_div > 1.0f;
    for (int64_t idx = threadIdx.x; idx < D; idx += kThreads) {
        float v = feat[b * ld + idx];
        float c = rintf(scaled ? v * scale_div : v);
        float back = scaled ? __fdiv_rn(c, scale_div) : c;
        bool good = (back == v) && c >= 0.0f && c <= 255.0f;
        em.put_code1(0, idx, good ? (int)c : 0, good);
    }
    em.finish(red);
}

// energy profiles + top-n indices from the sum planes (common.py:51-55, 80)
__global__ __launch_bounds__(64) void k_profiles_topk(const float* xzs, const float* yzs, int X, int Y, int Z,
                                                      int ntgt, int32_t* ijk, float* profiles) {
    extern __shared__ float prof[];   // X + Y + Z
    const int64_t b = blockIdx.x;
    const float* xz = xzs + b * (int64_t)X * Z;
    const float* yz = yzs + b * (int64_t)Y * Z;
    const int lane = threadIdx.x;
    float* s_theta = prof; float* s_phi = prof + X; float* s_r = prof + X + Y;
    for (int i = lane; i < X; i += 64) { float s = 0; for (int k = 0; k < Z; ++k) s += xz[i * Z + k]; s_theta[i] = s; }
    for (int j = lane; j < Y; j += 64) { float s = 0; for (int k = 0; k < Z; ++k) s += yz[j * Z + k]; s_phi[j] = s; }
    for (int k = lane; k < Z; k += 64) { float s = 0; for (int j = 0; j < Y; ++j) s += yz[j * Z + k]; s_r[k] = s; }
    __syncthreads();
    if (profiles) for (int t = lane; t < X + Y + Z; t += 64) profiles[b * (int64_t)(X + Y + Z) + t] = prof[t];
    __syncthreads();
    if (lane < 3) {
        float* s = lane == 0 ? s_theta : (lane == 1 ? s_phi : s_r);
        int L = lane == 0 ? X : (lane == 1 ? Y : Z);
        // repeated arg-max; result list ascending by value (largest last), ties: higher index = larger
        for (int t = 0; t < ntgt; ++t) {
            int best = -1; float bv = -INFINITY;
            for (int q = 0; q < L; ++q) {
                float v = s[q];
                if (v != v) v = -INFINITY;
                if (best < 0 || v >= bv) { bv = v; best = q; }
            }
            ijk[(b * ntgt + (ntgt - 1 - t)) * 3 + lane] = best;
            s[best] = -INFINITY;
        }
    }
}

int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

template <typename VT, int MODE, int LPR, int NM, bool FULL>
void launch_fast_pred(const ProjParams& pp, size_t lds_bytes, hipStream_t st) {
    dim3 grid((unsigned)pp.B), block(kThreads);
    // > 64 KB of dynamic LDS needs the attribute (gfx950 has 160 KB per CU); harmless otherwise
    if (pp.o.skip_if_set) {
        RML_MAX_DYN_LDS(160 * 1024, &k_project_fast<VT, MODE, LPR, NM, FULL, true>);
        hipLaunchKernelGGL((k_project_fast<VT, MODE, LPR, NM, FULL, true>), grid, block, lds_bytes, st, pp);
    } else {
        RML_MAX_DYN_LDS(160 * 1024, &k_project_fast<VT, MODE, LPR, NM, FULL, false>);
        hipLaunchKernelGGL((k_project_fast<VT, MODE, LPR, NM, FULL, false>), grid, block, lds_bytes, st, pp);
    }
}

template <typename VT, int MODE, int LPR>
int launch_fast_nm(const ProjParams& pp, int nm, size_t lds_bytes, hipStream_t st) {
    constexpr int NSLOT = 4 * (64 / LPR);
    // up to 8 rows per lane and plane; 16 for the long rows (LPR = 64: planes of 33..64 rows of 129..256 voxels, e.g. 64x64x256,
    // which otherwise fell to the general kernel at 0.04 of 8 TB/s)
    if (nm > (LPR == 64 ? 16 : 8)) return 1;
    const int NMr = nm <= 4 ? 4 : (nm <= 8 ? 8 : 16);
    const bool full = (pp.ZQ == LPR) && (pp.Y == NMr * NSLOT);
    if (NMr == 4) { if (full) launch_fast_pred<VT, MODE, LPR, 4, true>(pp, lds_bytes, st); else launch_fast_pred<VT, MODE, LPR, 4, false>(pp, lds_bytes, st); }
    else if (NMr == 8) { if (full) launch_fast_pred<VT, MODE, LPR, 8, true>(pp, lds_bytes, st); else launch_fast_pred<VT, MODE, LPR, 8, false>(pp, lds_bytes, st); }
    else if constexpr (LPR == 64) { if (full) launch_fast_pred<VT, MODE, LPR, 16, true>(pp, lds_bytes, st); else launch_fast_pred<VT, MODE, LPR, 16, false>(pp, lds_bytes, st); }
    return 0;
}

template <typename VT, int MODE, int NM>
void launch_rowgroup(const ProjParams& pp, int R, size_t lds_bytes, hipStream_t st) {
    dim3 grid((unsigned)pp.B), block((unsigned)(R * pp.ZQ));
    if (pp.o.skip_if_set) {
        RML_MAX_DYN_LDS(160 * 1024, &k_project_rowgroup<VT, MODE, NM, true>);
        hipLaunchKernelGGL((k_project_rowgroup<VT, MODE, NM, true>), grid, block, lds_bytes, st, pp, R);
    } else {
        RML_MAX_DYN_LDS(160 * 1024, &k_project_rowgroup<VT, MODE, NM, false>);
        hipLaunchKernelGGL((k_project_rowgroup<VT, MODE, NM, false>), grid, block, lds_bytes, st, pp, R);
    }
}

int gcd_int(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

template <typename VT, int MODE>
int launch_mode(const ProjParams& pp, int num_cu, hipStream_t st, bool* used_fast) {
    const int X = pp.X, Y = pp.Y, Z = pp.Z;
    *used_fast = false;
    if constexpr (sizeof(VT) == 1 && MODE == RML_MODE_MAX) {
        if (try_launch_u8_max(pp, st)) { *used_fast = true; return 0; }
    }
    bool fast_ok = (Z % 4 == 0) && (Z / 4 <= 64) && ((reinterpret_cast<uintptr_t>(pp.V) & (4 * sizeof(VT) - 1)) == 0);
    size_t lds_bytes = ((size_t)X * Z + (((size_t)X * Y + 3) & ~(size_t)3)) * 4 + 64 * 8;
    if (lds_bytes > 150 * 1024) fast_ok = false;
    // rows of 44 quads (the Walabot arena grid): the linear-plane wave-per-frame kernel (project_lin.hip), one workgroup per CU
    // beside a GEMM like k_project_wave; the wave kernel keeps every other long-row shape (and this one with RML_LINPLANE=0)
    if constexpr (sizeof(VT) == 4) {
        if (fast_ok && wave_kernel_wanted(pp.ZQ, pp.Y, pp.o.share_cu != 0, pp.B, num_cu, pp.k_waveframe) && try_launch_lin(pp, MODE, num_cu, st)) {
            *used_fast = true;
            return 0;
        }
    }
    if (fast_ok && try_launch_wave<VT, MODE>(pp, num_cu, st)) { *used_fast = true; return 0; }
    if (fast_ok && (Z / 4) != next_pow2(Z / 4) && (Z / 4) >= 8) {
        // rows that are not a power-of-two number of float4: row groups with every lane busy
        const int zq = Z / 4;
        int R = 64 / gcd_int(zq, 64);
        while (R * zq < 512 && 2 * R * zq <= 1024 && 2 * R <= Y) R *= 2;
        if (R * zq <= 1024) {
            const int nm = (Y + R - 1) / R;
            if (nm <= 2) { launch_rowgroup<VT, MODE, 2>(pp, R, lds_bytes, st); *used_fast = true; return 0; }
            if (nm <= 4) { launch_rowgroup<VT, MODE, 4>(pp, R, lds_bytes, st); *used_fast = true; return 0; }
            if (nm <= 8) { launch_rowgroup<VT, MODE, 8>(pp, R, lds_bytes, st); *used_fast = true; return 0; }
        }
    }
    if (fast_ok) {
        int zq = Z / 4;
        int lpr = next_pow2(zq); if (lpr < 16) lpr = 16;
        int nslot = 4 * (64 / lpr);
        int nm = (Y + nslot - 1) / nslot;
        int rc = 1;
        if (lpr == 16) rc = launch_fast_nm<VT, MODE, 16>(pp, nm, lds_bytes, st);
        else if (lpr == 32) rc = launch_fast_nm<VT, MODE, 32>(pp, nm, lds_bytes, st);
        else rc = launch_fast_nm<VT, MODE, 64>(pp, nm, lds_bytes, st);
        if (rc == 0) { *used_fast = true; return 0; }
    }
    hipLaunchKernelGGL((k_project_generic<VT, MODE>), dim3((unsigned)pp.B), dim3(kThreads), 0, st, pp);
    return 0;
}

void fill_params(ProjParams& pp, const rml_ctx* ctx, const void* V, int64_t B, int X, int Y, int Z, const int32_t* ijk, const ProjOut& o) {
    pp.V = V; pp.B = B; pp.X = X; pp.Y = Y; pp.Z = Z; pp.ZQ = Z / 4; pp.ijk = ijk; pp.tpf = 1; pp.rpl = 1; pp.o = o;
    pp.ntgt = 1; pp.ijk_out = nullptr; pp.profiles = nullptr; pp.wave_lds = 0; pp.stage_bytes = 0;
    const rml_opts def;
    const rml_opts& op = ctx ? ctx->opt : def;
    pp.k_waveframe = op.waveframe; pp.k_linplane = op.linplane; pp.k_stage_codes = op.stage_codes; pp.k_slice_wave = op.slice_wave;
    pp.k_derive_fused = op.derive_fused;
    if (op.project_share_cu) pp.o.share_cu = 1;       // RML_OPT_PROJECT_SHARE_CU: the pipeline's kernel configuration in a stand-alone launch
    for (int pl = 0; pl < 3; ++pl)
        pp.vec_ok[pl] = o.p[pl] && ((reinterpret_cast<uintptr_t>(o.p[pl]) & 15) == 0) && (o.stride[pl] % 4 == 0);
}

}  // namespace

namespace {
template <typename VT>
int launch_project_t(const ProjParams& pp, int mode, int num_cu, hipStream_t st) {
    bool fast = false;
    if (mode == RML_MODE_MAX) launch_mode<VT, RML_MODE_MAX>(pp, num_cu, st, &fast);
    else if (mode == RML_MODE_SUM) launch_mode<VT, RML_MODE_SUM>(pp, num_cu, st, &fast);
    else if (mode == RML_MODE_SLICE) {
        RML_REQUIRE(pp.ijk != nullptr, RML_ERR_INVALID, "rml_project: mode SLICE needs ijk");
        if (!try_launch_slice(pp, (int)sizeof(VT), st))        // rows that are not whole quads: the general kernel
            hipLaunchKernelGGL(k_project_slice<VT>, dim3((unsigned)pp.B), dim3(kThreads), 0, st, pp);
    } else {
        RML_REQUIRE(false, RML_ERR_INVALID, "rml_project: unknown mode %d", mode);
    }
    return RML_OK;
}
}  // namespace

bool rml_project_uses_wave_kernel(const rml_ctx* ctx, int vdtype, int mode, int X, int Y, int Z, bool share_cu, int64_t B) {
    (void)X;
    const int num_cu = ctx ? ctx->num_cu : 256, knob = ctx ? ctx->opt.waveframe : 1;
    if (mode != RML_MODE_MAX && mode != RML_MODE_SUM) return false;
    if (vdtype == RML_VOL_U8 && mode == RML_MODE_MAX && Z % 16 == 0) return false;      // the byte-native kernel takes those
    return Z % 4 == 0 && wave_kernel_wanted(Z / 4, Y, share_cu, B, num_cu, knob);
}

int rml_launch_project(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode,
                       const int32_t* ijk, const ProjOut& o, hipStream_t st, int targets_per_frame) {
    if (B == 0) return RML_OK;
    RML_REQUIRE(vdtype == RML_VOL_F32 || vdtype == RML_VOL_U8, RML_ERR_INVALID, "rml_project: unknown volume dtype %d", vdtype);
    if (mode == RML_MODE_MAX_NAN) mode = RML_MODE_MAX;         // round 6: mode MAX itself has NumPy's NaN policy (the old name stays valid)
    RML_REQUIRE(targets_per_frame == 1 || mode == RML_MODE_SLICE, RML_ERR_INVALID, "rml_project: several targets per frame only in mode SLICE");
    ProjParams pp;
    fill_params(pp, ctx, V, B, X, Y, Z, ijk, o);       // B counts output rows
    pp.tpf = targets_per_frame;
    const int num_cu = !ctx ? 256 : ctx->num_cu;
    const int rc = vdtype == RML_VOL_U8 ? launch_project_t<uint8_t>(pp, mode, num_cu, st) : launch_project_t<float>(pp, mode, num_cu, st);
    if (rc) return rc;
    RML_HIP(hipGetLastError());
    return RML_OK;
}

// ---- single observations: the frame split over the chip ---------------------------------------------------------------------------
// The reference classifies ONE observation per call (predict.py:98-119: the loop over the radar's targets).  Every projection
// kernel above gives a frame to one workgroup or one wave -- the right grain for thousands of frames, and 118-141 us for ONE frame
// of 2 MiB (profiles/r06_stats_latency.txt: one workgroup streams at 15 GB/s).  A max-projection splits along x without any new
// arithmetic: the same memory viewed as B*S frames of X/S planes gives, per piece, its rows of xz and xy -- final, they belong to
// single planes -- and a partial yz; k_project_finalize takes the maximum of the S partial yz planes and sends the assembled row
// through the Emitter (float row, codes, statistics, pad: whatever the caller asked for).  Two launches, S workgroups per frame in
// the first: 141 -> ~15 us at 64x64x128.  Used for batches of at most RML_SMALL_FRAMES float32 frames.
namespace {
constexpr int kFinalThreads = 1024;      // one workgroup per frame: a few quads per thread, every load of a plane in flight at once
__global__ __launch_bounds__(kFinalThreads) void k_project_finalize(ProjParams a, const float* __restrict__ scr, int S, int Xs, int64_t ldv) {
    __shared__ int64_t red[64];
    const int X = a.X, Y = a.Y, Z = a.Z;
    const int64_t b = blockIdx.x;
    const float* __restrict__ fr = scr + b * S * ldv;              // the S pieces of this frame: [xz_s (Xs x Z) | yz_s (Y x Z) | xy_s (Xs x Y)]
    const int64_t oyz = (int64_t)Xs * Z, oxy = oyz + (int64_t)Y * Z;
    Emitter em(a, b);
    em.rmw = false;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < (X * Z) >> 2; idx += kFinalThreads) {      // xz[i, :] = piece i / Xs, row i % Xs
        const int i = (idx * 4) / Z, k = idx * 4 - i * Z, s = i / Xs;
        em.put4(0, (int64_t)idx * 4, *reinterpret_cast<const float4*>(fr + s * ldv + (int64_t)(i - s * Xs) * Z + k));
    }
    for (int idx = tid; idx < (Y * Z) >> 2; idx += kFinalThreads) {      // yz = max over the pieces (NumPy's NaN policy: Op<MAX>)
        // eight pieces' quads in flight at a time (one load behind the other's maximum is a memory latency per piece: 76 us per frame
        // with 256 threads, session r6m)
        const float* __restrict__ q0 = fr + oyz + (int64_t)idx * 4;
        float4 v = *reinterpret_cast<const float4*>(q0);
        for (int s0 = 1; s0 < S; s0 += 8) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(q0 + (int64_t)(s0 + u < S ? s0 + u : 0) * ldv);
#pragma unroll
            for (int u = 0; u < 8; ++u) v = op4<RML_MODE_MAX>(v, t[u]);       // (a piece past S re-reads piece 0: max(v, v0) = v)
        }
        em.put4(1, (int64_t)idx * 4, v);
    }
    const int nxy = X * Y;
    if ((Y & 3) == 0) {
        for (int idx = tid; idx < nxy >> 2; idx += kFinalThreads) {
            const int i = (idx * 4) / Y, j = idx * 4 - i * Y, s = i / Xs;
            em.put4(2, (int64_t)idx * 4, *reinterpret_cast<const float4*>(fr + s * ldv + oxy + (int64_t)(i - s * Xs) * Y + j));
        }
    } else {
        // rows of the xy plane that are not whole quads (Walabot grid: 31): quads of the ROW may straddle two planes' pieces
        for (int idx = tid; idx < nxy >> 2; idx += kFinalThreads) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = idx * 4 + e, i = t / Y, j = t - i * Y, s = i / Xs;
                v[e] = fr[s * ldv + oxy + (int64_t)(i - s * Xs) * Y + j];
            }
            em.put4(2, (int64_t)idx * 4, make_float4(v[0], v[1], v[2], v[3]));
        }
        for (int t = (nxy & ~3) + tid; t < nxy; t += kFinalThreads) {
            const int i = t / Y, j = t - i * Y, s = i / Xs;
            em.put1(2, t, fr[s * ldv + oxy + (int64_t)(i - s * Xs) * Y + j]);
        }
    }
    em.finish(red);
}
}  // namespace

// pieces per frame for the split projection of a single observation: 0 = shape not taken (rows that are not whole quads, one plane)
int rml_project_split_pieces(int X, int Y, int Z) {
    if (Z % 4 != 0 || X < 2 || Y < 1) return 0;
    int best = 0;
    for (int s = 2; s <= X && s <= 16; ++s)
        if (X % s == 0) best = s;                    // the largest divisor up to 16: 16 at X = 64, 11 at X = 22
    return best;
}
int64_t rml_project_split_ld(int X, int Y, int Z, int S) {
    const int Xs = X / S;
    return (((int64_t)Xs * Z + (int64_t)Y * Z + (int64_t)Xs * Y) + 3) & ~(int64_t)3;
}
size_t rml_project_split_scratch_bytes(int64_t B, int X, int Y, int Z, int S) {
    return (size_t)B * S * rml_project_split_ld(X, Y, Z, S) * sizeof(float);
}

// mode MAX of B float32 / uint8 frames through the split view; scratch: rml_project_split_scratch_bytes, 16-byte aligned
int rml_launch_project_split(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, const ProjOut& o, float* scratch, int S,
                             hipStream_t st) {
    RML_REQUIRE(S >= 2 && X % S == 0 && Z % 4 == 0 && scratch, RML_ERR_INVALID, "rml_launch_project_split: bad split");
    const int Xs = X / S;
    const int64_t ldv = rml_project_split_ld(X, Y, Z, S);
    ProjOut os{};
    os.p[0] = scratch; os.p[1] = scratch + (int64_t)Xs * Z; os.p[2] = os.p[1] + (int64_t)Y * Z;
    os.stride[0] = os.stride[1] = os.stride[2] = ldv;
    os.sel = RML_MASK_ALL;
    int rc = rml_launch_project(ctx, V, vdtype, B * S, Xs, Y, Z, RML_MODE_MAX, nullptr, os, st);
    if (rc) return rc;
    ProjParams pp;
    fill_params(pp, ctx, V, B, X, Y, Z, nullptr, o);
    hipLaunchKernelGGL(k_project_finalize, dim3((unsigned)B), dim3(kFinalThreads), 0, st, pp, (const float*)scratch, S, Xs, ldv);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

// derive (-> slice) in one pass where the shape has a fused kernel: RML_OK when launched, RML_ERR_UNSUPPORTED (no message: the
// callers fall back to the two-kernel path) otherwise.  o.sel == 0: derive only.
int rml_launch_derive_slice(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int num_targets,
                            int32_t* ijk_out, float* profiles, const ProjOut& o, hipStream_t st) {
    if (B == 0) return RML_OK;
    ProjParams pp;
    fill_params(pp, ctx, V, B, X, Y, Z, nullptr, o);
    pp.ntgt = num_targets; pp.ijk_out = ijk_out; pp.profiles = profiles;
    const int num_cu = !ctx ? 256 : ctx->num_cu;
    if (!try_launch_derive_slice(pp, vdtype == RML_VOL_U8 ? 1 : 4, num_cu, st)) return RML_ERR_UNSUPPORTED;
    RML_HIP(hipGetLastError());
    return RML_OK;
}

// ---- exported entry points ------------------------------------------------------------------
static int plane_len(int pl, int X, int Y, int Z) { return pl == 0 ? X * Z : (pl == 1 ? Y * Z : X * Y); }

extern "C" int64_t rml_feature_len(int X, int Y, int Z, uint32_t mask) {
    int64_t d = 0;
    for (int pl = 0; pl < 3; ++pl) if (mask & (1u << pl)) d += plane_len(pl, X, Y, Z);
    return d;
}

static int project_rows(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode, int tpf,
                        const int32_t* ijk, float scale_div, uint32_t mask,
                        float* feat, int64_t ld_feat, uint8_t* feat_q, int64_t ld_q,
                        int32_t* row_isum, int64_t* row_isq, int32_t* row_flags, void* stream);

extern "C" int rml_project(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode,
                           const int32_t* ijk, float scale_div, uint32_t mask,
                           float* feat, int64_t ld_feat, uint8_t* feat_q, int64_t ld_q,
                           int32_t* row_isum, int64_t* row_isq, int32_t* row_flags, void* stream) {
    return project_rows(ctx, V, vdtype, B, X, Y, Z, mode, 1, ijk, scale_div, mask, feat, ld_feat, feat_q, ld_q, row_isum, row_isq,
                        row_flags, stream);
}

extern "C" int rml_project_slices(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int T,
                                  const int32_t* ijk, float scale_div, uint32_t mask,
                                  float* feat, int64_t ld_feat, uint8_t* feat_q, int64_t ld_q,
                                  int32_t* row_isum, int64_t* row_isq, int32_t* row_flags, void* stream) {
    RML_REQUIRE(T >= 1 && B >= 0 && B * (int64_t)T < (int64_t)1 << 31, RML_ERR_INVALID, "rml_project_slices: bad target count");
    RML_REQUIRE(B == 0 || ijk != nullptr, RML_ERR_INVALID, "rml_project_slices: ijk is NULL");
    return project_rows(ctx, V, vdtype, B * T, X, Y, Z, RML_MODE_SLICE, T, ijk, scale_div, mask, feat, ld_feat, feat_q, ld_q, row_isum,
                        row_isq, row_flags, stream);
}

// the row outputs of rml_project / rml_project_slices / rml_derive_slice as a ProjOut
static void rows_out(ProjOut& o, int X, int Y, int Z, uint32_t mask, float scale_div, float* feat, int64_t ld_feat, uint8_t* feat_q,
                     int64_t ld_q, int32_t* row_isum, int64_t* row_isq, int32_t* row_flags) {
    int64_t off = 0;
    for (int pl = 0; pl < 3; ++pl) {
        if (mask & (1u << pl)) {
            o.p[pl] = feat ? feat + off : nullptr;
            o.stride[pl] = ld_feat;
            o.q[pl] = feat_q ? feat_q + off : nullptr;
            off += plane_len(pl, X, Y, Z);
        }
    }
    o.sel = mask & RML_MASK_ALL;
    o.qstride = ld_q;
    o.qrow = feat_q; o.qD = rml_feature_len(X, Y, Z, mask);
    o.row_isum = row_isum; o.row_isq = row_isq; o.row_flags = row_flags;
    o.scale_div = scale_div;
}

static int derive_two_kernels(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int num_targets, int32_t* ijk,
                              float* profiles, hipStream_t st);

extern "C" int rml_derive_slice_supported(const rml_ctx* ctx, const void* V, int vdtype, int X, int Y, int Z, int num_targets) {
    if (X <= 0 || Y <= 0 || Z <= 0 || (vdtype != RML_VOL_F32 && vdtype != RML_VOL_U8)) return 0;
    if (ctx && !ctx->opt.derive_fused) return 0;
    const size_t quad = vdtype == RML_VOL_U8 ? 4 : 16;
    if (V && (reinterpret_cast<uintptr_t>(V) & (quad - 1))) return 0;
    return derive_slice_shape_ok(X, Y, Z, num_targets) ? 1 : 0;
}

extern "C" int rml_derive_slice(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int num_targets,
                                int32_t* ijk, float* profiles, float scale_div, uint32_t mask,
                                float* feat, int64_t ld_feat, uint8_t* feat_q, int64_t ld_q,
                                int32_t* row_isum, int64_t* row_isq, int32_t* row_flags, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && X > 0 && Y > 0 && Z > 0, RML_ERR_INVALID, "rml_derive_slice: bad arguments");
    if (B == 0) return RML_OK;
    RML_REQUIRE(V != nullptr, RML_ERR_INVALID, "rml_derive_slice: V is NULL");
    RML_REQUIRE(vdtype == RML_VOL_F32 || vdtype == RML_VOL_U8, RML_ERR_INVALID, "rml_derive_slice: unknown volume dtype %d", vdtype);
    RML_REQUIRE((mask & RML_MASK_ALL) != 0, RML_ERR_INVALID, "rml_derive_slice: empty projection mask");
    RML_REQUIRE(num_targets >= 1 && num_targets <= X && num_targets <= Y && num_targets <= Z, RML_ERR_INVALID,
                "rml_derive_slice: num_targets out of range");
    RML_REQUIRE(B * (int64_t)num_targets < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_derive_slice: too many rows for one launch");
    const int64_t D = rml_feature_len(X, Y, Z, mask);
    RML_REQUIRE(!feat || ld_feat >= D, RML_ERR_INVALID, "rml_derive_slice: ld_feat < D");
    RML_REQUIRE(!feat_q || (ld_q >= D && ld_q % 4 == 0 && (reinterpret_cast<uintptr_t>(feat_q) & 3) == 0),
                RML_ERR_INVALID, "rml_derive_slice: feat_q needs ld_q >= D, ld_q %% 4 == 0 and 4-byte alignment");
    RML_HIP(hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    ProjOut o{};
    rows_out(o, X, Y, Z, mask, scale_div, feat, ld_feat, feat_q, ld_q, row_isum, row_isq, row_flags);
    int rc = rml_launch_derive_slice(ctx, V, vdtype, B, X, Y, Z, num_targets, ijk, profiles, o, st);
    if (rc != RML_ERR_UNSUPPORTED) return rc;
    // shapes without a fused kernel: sum planes -> profiles + top-n, then the slices (the (i,j,k) go through the caller's buffer or
    // the context's workspace)
    rml_ctx_guard guard(ctx, st);
    int32_t* ijk_w = ijk;
    if (!ijk_w) {
        // behind the sum planes of derive_two_kernels in the shared workspace
        const size_t planes = (size_t)B * ((size_t)X * Z + (size_t)Y * Z) * sizeof(float);
        void* ws = nullptr;
        rc = rml_ws_reserve(ctx, planes + (size_t)B * num_targets * 3 * sizeof(int32_t), &ws, st);
        if (rc) return rc;
        ijk_w = reinterpret_cast<int32_t*>(static_cast<unsigned char*>(ws) + planes);
    }
    rc = derive_two_kernels(ctx, V, vdtype, B, X, Y, Z, num_targets, ijk_w, profiles, st);
    if (rc) return rc;
    return rml_launch_project(ctx, V, vdtype, B * num_targets, X, Y, Z, RML_MODE_SLICE, ijk_w, o, st, num_targets);
}

static int project_rows(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode, int tpf,
                        const int32_t* ijk, float scale_div, uint32_t mask,
                        float* feat, int64_t ld_feat, uint8_t* feat_q, int64_t ld_q,
                        int32_t* row_isum, int64_t* row_isq, int32_t* row_flags, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && X > 0 && Y > 0 && Z > 0, RML_ERR_INVALID, "rml_project: bad arguments");
    if (B == 0) return RML_OK;
    RML_REQUIRE(V != nullptr, RML_ERR_INVALID, "rml_project: V is NULL");
    RML_REQUIRE((mask & RML_MASK_ALL) != 0, RML_ERR_INVALID, "rml_project: empty projection mask");
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_project: B too large for one launch");
    const int64_t D = rml_feature_len(X, Y, Z, mask);
    RML_REQUIRE(!feat || ld_feat >= D, RML_ERR_INVALID, "rml_project: ld_feat < D");
    RML_REQUIRE(!feat_q || (ld_q >= D && ld_q % 4 == 0 && (reinterpret_cast<uintptr_t>(feat_q) & 3) == 0),
                RML_ERR_INVALID, "rml_project: feat_q needs ld_q >= D, ld_q %% 4 == 0 and 4-byte alignment");
    RML_HIP(hipSetDevice(ctx->device));
    ProjOut o{};
    int64_t off = 0;
    for (int pl = 0; pl < 3; ++pl) {
        if (mask & (1u << pl)) {
            o.p[pl] = feat ? feat + off : nullptr;
            o.stride[pl] = ld_feat;
            o.q[pl] = feat_q ? feat_q + off : nullptr;
            off += plane_len(pl, X, Y, Z);
        }
    }
    o.sel = mask & RML_MASK_ALL;
    o.qstride = ld_q;
    o.qrow = feat_q; o.qD = D;
    o.row_isum = row_isum; o.row_isq = row_isq; o.row_flags = row_flags;
    o.scale_div = scale_div;
    return rml_launch_project(ctx, V, vdtype, B, X, Y, Z, mode, ijk, o, static_cast<hipStream_t>(stream), tpf);
}

extern "C" int rml_project_planes(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int mode,
                                  const int32_t* ijk, float* xz, float* yz, float* xy, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && X > 0 && Y > 0 && Z > 0, RML_ERR_INVALID, "rml_project_planes: bad arguments");
    if (B == 0) return RML_OK;
    RML_REQUIRE(V != nullptr, RML_ERR_INVALID, "rml_project_planes: V is NULL");
    RML_REQUIRE(B < (int64_t)1 << 31, RML_ERR_UNSUPPORTED, "rml_project_planes: B too large for one launch");
    RML_HIP(hipSetDevice(ctx->device));
    ProjOut o{};
    o.p[0] = xz; o.stride[0] = (int64_t)X * Z;
    o.p[1] = yz; o.stride[1] = (int64_t)Y * Z;
    o.p[2] = xy; o.stride[2] = (int64_t)X * Y;
    o.sel = (xz ? 1u : 0u) | (yz ? 2u : 0u) | (xy ? 4u : 0u);
    o.scale_div = 0.0f;
    return rml_launch_project(ctx, V, vdtype, B, X, Y, Z, mode, ijk, o, static_cast<hipStream_t>(stream));
}

extern "C" int rml_derive_targets(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z,
                                  int num_targets, int32_t* ijk, float* profiles, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && X > 0 && Y > 0 && Z > 0, RML_ERR_INVALID, "rml_derive_targets: bad arguments");
    if (B == 0) return RML_OK;
    RML_REQUIRE(V && ijk, RML_ERR_INVALID, "rml_derive_targets: NULL argument");
    RML_REQUIRE(num_targets >= 1 && num_targets <= X && num_targets <= Y && num_targets <= Z, RML_ERR_INVALID,
                "rml_derive_targets: num_targets out of range");
    RML_HIP(hipSetDevice(ctx->device));
    if (B == 0) return RML_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // one pass, no workspace, where the shape has the fused kernel (k_derive_slice without outputs)
    ProjOut none{};
    int rc1 = rml_launch_derive_slice(ctx, V, vdtype, B, X, Y, Z, num_targets, ijk, profiles, none, st);
    if (rc1 != RML_ERR_UNSUPPORTED) return rc1;
    rml_ctx_guard guard(ctx, st);           // shared workspace
    return derive_two_kernels(ctx, V, vdtype, B, X, Y, Z, num_targets, ijk, profiles, st);
}

// sum planes through the workspace, then profiles + top-n (any shape); the caller holds the context guard
static int derive_two_kernels(rml_ctx* ctx, const void* V, int vdtype, int64_t B, int X, int Y, int Z, int num_targets, int32_t* ijk,
                              float* profiles, hipStream_t st) {
    // workspace: sum planes xz (B,X,Z) and yz (B,Y,Z)
    size_t need = (size_t)B * ((size_t)X * Z + (size_t)Y * Z) * sizeof(float);
    void* ws = nullptr;
    int rc = rml_ws_reserve(ctx, need, &ws, st);
    if (rc) return rc;
    float* xzs = static_cast<float*>(ws);
    float* yzs = xzs + (size_t)B * X * Z;
    ProjOut o{};
    o.p[0] = xzs; o.stride[0] = (int64_t)X * Z;
    o.p[1] = yzs; o.stride[1] = (int64_t)Y * Z;
    o.sel = 3u;
    o.scale_div = 0.0f;
    rc = rml_launch_project(ctx, V, vdtype, B, X, Y, Z, RML_MODE_SUM, nullptr, o, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_profiles_topk, dim3((unsigned)B), dim3(64), (size_t)(X + Y + Z) * sizeof(float), st,
                       xzs, yzs, X, Y, Z, num_targets, ijk, profiles);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

extern "C" int rml_assemble_features(rml_ctx* ctx, const float* xz, const float* yz, const float* xy,
                                     int64_t B, int X, int Y, int Z, float scale_div, uint32_t mask,
                                     float* feat, int64_t ld_feat, void* stream) {
    RML_REQUIRE(ctx && B >= 0 && X > 0 && Y > 0 && Z > 0, RML_ERR_INVALID, "rml_assemble_features: bad arguments");
    if (B == 0) return RML_OK;
    RML_REQUIRE(feat != nullptr, RML_ERR_INVALID, "rml_assemble_features: feat is NULL");
    RML_REQUIRE((mask & RML_MASK_ALL) != 0, RML_ERR_INVALID, "rml_assemble_features: empty mask");
    const float* src[3] = {xz, yz, xy};
    for (int pl = 0; pl < 3; ++pl)
        RML_REQUIRE(!(mask & (1u << pl)) || src[pl], RML_ERR_INVALID, "rml_assemble_features: selected plane %d is NULL", pl);
    const int64_t D = rml_feature_len(X, Y, Z, mask);
    RML_REQUIRE(ld_feat >= D, RML_ERR_INVALID, "rml_assemble_features: ld_feat < D");
    RML_HIP(hipSetDevice(ctx->device));
    if (B == 0) return RML_OK;
    ProjOut o{};
    int64_t off = 0;
    for (int pl = 0; pl < 3; ++pl)
        if (mask & (1u << pl)) { o.p[pl] = feat + off; o.stride[pl] = ld_feat; off += plane_len(pl, X, Y, Z); }
    o.sel = mask & RML_MASK_ALL;
    o.scale_div = scale_div;
    ProjParams pp;
    fill_params(pp, ctx, nullptr, B, X, Y, Z, nullptr, o);
    hipLaunchKernelGGL(k_assemble, dim3((unsigned)B), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       (mask & 1u) ? xz : nullptr, (mask & 2u) ? yz : nullptr, (mask & 4u) ? xy : nullptr, pp);
    RML_HIP(hipGetLastError());
    return RML_OK;
}

extern "C" int rml_quantize_rows(rml_ctx* ctx, const float* feat, int64_t N, int64_t D, int64_t ld_feat,
                                 float scale_div, uint8_t* feat_q, int64_t ld_q,
                                 int32_t* row_isum, int64_t* row_isq, int32_t* row_flags, void* stream) {
    RML_REQUIRE(ctx && N >= 0 && D > 0 && ld_feat >= D && ld_q >= D, RML_ERR_INVALID, "rml_quantize_rows: bad arguments");
    if (N == 0) return RML_OK;
    RML_REQUIRE(feat && feat_q, RML_ERR_INVALID, "rml_quantize_rows: NULL argument");
    RML_HIP(hipSetDevice(ctx->device));
    if (N == 0) return RML_OK;
    ProjOut o{};
    o.q[0] = feat_q; o.qstride = ld_q; o.qrow = feat_q; o.qD = D; o.sel = 1u;
    o.row_isum = row_isum; o.row_isq = row_isq; o.row_flags = row_flags;
    o.scale_div = 0.0f;
    ProjParams pp;
    fill_params(pp, ctx, nullptr, N, 1, 1, 1, nullptr, o);
    hipLaunchKernelGGL(k_quantize_rows, dim3((unsigned)N), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                       feat, D, ld_feat, scale_div, pp);
    RML_HIP(hipGetLastError());
    return RML_OK;
}
