// k_project_lin: the wave-per-frame projection for rows that do NOT fill a load instruction -- the Walabot arena grid
// 22 x 31 x 176 (common.py:25-27, predict.py:74-77): a row is 44 float4, so k_project_wave's row-per-instruction loads carry
// 704 of 1024 bytes.  Measured with the row length as the only variable (22 x 31 x Z, round 2): 0.68 / 0.70 / 0.77 / 0.78 of
// 8 TB/s at Z = 176 / 192 / 224 / 256 -- a wave's requests reach the memory in bursts of 11 instead of 16 sixty-four-byte
// pieces.  Here a plane is loaded as the contiguous array of Y*Z/4 quads it is: 64 quads = 1 KB per instruction, 22
// instructions per 31 x 176 plane instead of 31.
//   * yz[j,k] = op_i V is ELEMENT-WISE in that linear layout (22 float4 accumulators per lane) and leaves linearly;
//   * xy[i,j] = op_k V: every quad belongs to one row; the per-quad partials go through a wave-private linear LDS strip and
//     lane j folds the Z/16 float4 of row j once per plane;
//   * xz[i,k] = op_j V is the only projection that needs the row structure: the quads of a GROUP of RG rows (RG * Z/4 = NI
//     whole instructions: 16 rows x 44 quads = 11 x 64) go through a wave-private LDS image and lane c folds column c.
// Same persistent, barrier-free, atomic-free structure as k_project_wave (static frame assignment, the next group's loads in
// flight in a second register buffer across plane and frame boundaries), same Emitter (bit-identical outputs).
#include "project_shared.h"

namespace {

using namespace rmlproj;

// ZQ4 = Z/16 (float4 of per-quad partials per row), NI = instructions per group of RG rows (RG * Z/4 = NI * 64), NGRP groups per
// plane (even: the two register buffers alternate statically), NMASK = trailing groups of a plane that may run past it (their
// loads clamp to the plane's last quad, their values are replaced by the identity): 1 when only the last group can be partial,
// 2 when the plane may end inside the second to last group (the last one is then empty)
template <int MODE, int ZQ4, int NI, int RG, int NGRP, int NMASK, bool PRED>
__global__ __launch_bounds__(256, 2) void k_project_lin(ProjParams a) {
    if constexpr (PRED) { if (*a.o.skip_if_set) return; }
    static_assert(NGRP % 2 == 0 && RG % 8 == 0 && NMASK >= 1 && NMASK <= NGRP,
                  "an even number of groups per plane (the row buffers alternate statically); xz folds 8 rows per wait");
    constexpr int NT = NI * NGRP;                       // load instructions per plane
    const int X = a.X, Y = a.Y, Z = a.Z, ZQ = a.ZQ;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    int64_t cf = (int64_t)blockIdx.x * 4 + wave;        // frame being reduced
    if (cf >= a.B) return;
    const float4* __restrict__ Vall = reinterpret_cast<const float4*>(a.V);
    const int pq = Y * ZQ;                              // quads per plane
    const int64_t fq = (int64_t)X * pq;
    const float id = Op<MODE>::ident();
    const float4 id4 = make_float4(id, id, id, id);
    // wave-private LDS: the current row group as a linear quad image, and the per-quad row partials of the current plane
    extern __shared__ __align__(16) unsigned char lin_smem[];
    float4* img = reinterpret_cast<float4*>(lin_smem) + wave * (NI * 64);
    float* xyl = reinterpret_cast<float*>(lin_smem + (size_t)4 * NI * 64 * sizeof(float4)) + wave * (NT * 64);

    // load cursor: one group ahead of the reduction (frames assigned statically: wave w takes w, w + #waves, ...)
    int64_t lf = cf;
    int li = 0;
    const float4* __restrict__ lV = Vall + lf * fq;
    auto next_plane = [&]() __attribute__((always_inline)) {
        ++li;
        lV += pq;
        if (li == X) {                                  // next frame of this wave; past the end: re-read (never consumed)
            li = 0;
            const int64_t nf = lf + stride;
            lf = nf < a.B ? nf : lf;
            lV = Vall + lf * fq;
        }
    };
    // the last instruction of a plane runs past it: those lanes re-read the plane's last quad (unconditional loads: a
    // conditional one makes hipcc branch around every load and drain vmcnt) and are replaced by the identity below
    const uint32_t qlast = (uint32_t)(pq - 1);
    float4 buf[2][NI];
    auto fetch = [&](float4 (&dst)[NI], auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value;
        uint32_t lane_t = (uint32_t)lane;               // opaque per step: no NT hoisted per-instruction offsets kept alive
        asm volatile("" : "+v"(lane_t));
        static_for<NI>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            const uint32_t q = (uint32_t)((g * NI + u) * 64) + lane_t;
            if constexpr (g >= NGRP - NMASK) dst[u] = ld_stream(lV + (q < qlast ? q : qlast));
            else dst[u] = ld_stream(lV + q);            // the leading groups are whole
        });
    };
    Emitter em(a, cf);
    em.rmw = false;                                     // plain stores here (see Emitter::rmw)
    if (a.stage_bytes) {                                // codes-only launches: the wave's code stage behind the images (Emitter::stage)
        em.set_stage(lin_smem + (size_t)4 * (NI * 64 * sizeof(float4) + NT * 64 * sizeof(float)) + wave * a.stage_bytes, (X * Z + 15) & ~15);
    }
    fetch(buf[0], std::integral_constant<int, 0>{});    // group 0 of the first plane
    for (; cf < a.B; cf += stride) {
        em.reset(cf);
        float4 yz[NT];
        static_for<NT>([&](auto tc) { yz[decltype(tc)::value] = id4; });
        float4 xz = id4;
        for (int ci = 0; ci < X; ++ci) {
            static_for<NGRP>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int gn = (g + 1) % NGRP;
                if constexpr (gn == 0) next_plane();
                fetch(buf[(g + 1) & 1], std::integral_constant<int, gn>{});
                // keep the software pipeline as written (see k_project_wave)
                __builtin_amdgcn_sched_barrier(0);
                static_for<NI>([&](auto uc) {
                    constexpr int u = decltype(uc)::value;
                    constexpr int t = g * NI + u;
                    float4 v = buf[g & 1][u];
                    // quads past the plane (only in its last group) hold the identity: xz folds whole rows of the image.
                    // Branch-free: a branch around the select makes hipcc drain vmcnt after every load of the group
                    if constexpr (g >= NGRP - NMASK) {
                        const bool in = t * 64 + lane < pq;
                        v.x = in ? v.x : id; v.y = in ? v.y : id; v.z = in ? v.z : id; v.w = in ? v.w : id;
                    }
                    yz[t] = op4_raw<MODE>(yz[t], v);
                    asm volatile("" : "+v"(yz[t].x), "+v"(yz[t].y), "+v"(yz[t].z), "+v"(yz[t].w));      // pin the update here
                    img[u * 64 + lane] = v;
                    xyl[t * 64 + lane] = op_raw<MODE>(op_raw<MODE>(v.x, v.y), op_raw<MODE>(v.z, v.w));
                });
                __builtin_amdgcn_sched_barrier(0);
                // xz: lane c folds column c of the group's RG rows (lanes >= ZQ fold a duplicate column: never stored)
                {
                    int lane_x = lane;
                    asm volatile("" : "+v"(lane_x));
                    const float4* col = img + (lane_x < ZQ ? lane_x : ZQ - 1);
                    // eight reads in flight per wait (left alone hipcc waits after every read or two: 16 LDS latencies per group)
                    static_for<RG / 8>([&](auto hc) {
                        constexpr int h = decltype(hc)::value;
                        float4 c8[8];
                        static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; c8[r] = col[(h * 8 + r) * ZQ]; });
                        __builtin_amdgcn_sched_barrier(0);
                        static_for<8>([&](auto rc) { constexpr int r = decltype(rc)::value; xz = op4_raw<MODE>(xz, c8[r]); });
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    asm volatile("" : "+v"(xz.x), "+v"(xz.y), "+v"(xz.z), "+v"(xz.w));
                }
                if constexpr (g == NGRP - 1) {                      // plane ci of frame cf is complete
                    int lane_e = lane;
                    asm volatile("" : "+v"(lane_e));
                    if (lane_e < ZQ) em.put4(0, (int64_t)ci * Z + 4 * lane_e, xz);
                    if (lane_e < Y) {
                        const float* rowp = xyl + lane_e * ZQ;      // row j = quads [j ZQ, (j+1) ZQ), one float per quad
                        float m = id;
                        float4 r4[ZQ4];                             // Z/16 float4 of per-quad partials per row (44 quads: 11)
                        static_for<ZQ4>([&](auto qc) { constexpr int q = decltype(qc)::value; r4[q] = *reinterpret_cast<const float4*>(rowp + 4 * q); });
                        static_for<ZQ4>([&](auto qc) {
                            constexpr int q = decltype(qc)::value;
                            m = Op<MODE>::f(m, Op<MODE>::f(Op<MODE>::f(r4[q].x, r4[q].y), Op<MODE>::f(r4[q].z, r4[q].w)));
                        });
                        em.put1(2, (int64_t)ci * Y + lane_e, m);
                    }
                    xz = id4;
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
        // yz leaves linearly: quad q of the plane = 4 consecutive values of the yz row image
        int lane_f = lane, pq_f = pq;
        asm volatile("" : "+v"(lane_f), "+s"(pq_f));
        static_for<NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            const int q = t * 64 + lane_f;
            if (q < pq_f) em.put4(1, (int64_t)q * 4, yz[t]);
        });
        em.flush_wave(lane_f);
        em.finish_wave(lane_f);
    }
}

template <int MODE, int ZQ4, int NI, int RG, int NGRP, int NMASK>
void launch_lin(const ProjParams& pp_in, int num_cu, hipStream_t st) {
    ProjParams pp = pp_in;
    pp.stage_bytes = code_stage_bytes(pp, 4);
    // wave-private images: 4 x (NI * 64 float4 + NT * 64 floats) (44 quads: 66 KB) + the code stage of a codes-only launch (4 x 4.6 KB
    // at the Walabot grid).  Beside a GEMM (share_cu = 1) the request is padded past half of the CU's LDS so that the dispatcher cannot
    // put two of these persistent workgroups on one CU (see launch_wave)
    constexpr size_t kMaxLds = 160 * 1024;              // the attribute set below: a CU's whole LDS (as k_derive_slice does)
    const size_t images = (size_t)4 * (NI * 64 * 16 + NI * NGRP * 64 * 4);
    if (images + (size_t)4 * pp.stage_bytes > kMaxLds) pp.stage_bytes = 0;      // no room for the code stage: direct stores
    const size_t mine = images + (size_t)4 * pp.stage_bytes;
    int per_cu = pp.o.share_cu ? 1 : 2;
    if (per_cu * mine > 160 * 1024) per_cu = 1;         // (measured: one or two of these workgroups per CU stream equally fast)
    const int64_t want = (pp.B + 3) / 4;
    const int64_t cap = (int64_t)num_cu * per_cu;
    dim3 grid((unsigned)(want < cap ? want : cap)), block(kThreads);
    const size_t lds = (pp.o.share_cu && per_cu == 1 && !pp.o.no_pad && mine < 81 * 1024) ? 81 * 1024 : mine;
    if (pp.o.skip_if_set) {
        RML_MAX_DYN_LDS(kMaxLds, &k_project_lin<MODE, ZQ4, NI, RG, NGRP, NMASK, true>);
        hipLaunchKernelGGL((k_project_lin<MODE, ZQ4, NI, RG, NGRP, NMASK, true>), grid, block, lds, st, pp);
    } else {
        RML_MAX_DYN_LDS(kMaxLds, &k_project_lin<MODE, ZQ4, NI, RG, NGRP, NMASK, false>);
        hipLaunchKernelGGL((k_project_lin<MODE, ZQ4, NI, RG, NGRP, NMASK, false>), grid, block, lds, st, pp);
    }
}

// rows of ZQ = 40 / 48 / 56 quads in groups of 8 rows (5 / 6 / 7 whole instructions), mode MAX
template <int ZQ>
bool launch_lin8(const ProjParams& pp, int num_cu, hipStream_t st) {
    if (pp.Y > 8 && pp.Y <= 16) { launch_lin<RML_MODE_MAX, ZQ / 4, ZQ / 8, 8, 2, 1>(pp, num_cu, st); return true; }
    if (pp.Y > 16 && pp.Y <= 32) { launch_lin<RML_MODE_MAX, ZQ / 4, ZQ / 8, 8, 4, 2>(pp, num_cu, st); return true; }
    return false;
}

}  // namespace

namespace rmlproj {

// Rows that do not fill a load instruction, loaded as the linear array of quads a plane is.  44 quads (the Walabot arena grid) in
// groups of 16 rows (11 whole instructions), two groups: 17..32 rows, modes MAX and SUM; 40 / 48 / 56 quads (Z = 160 / 192 / 224:
// other arenas the reference resizes to, predict.py:34-54) in groups of 8 rows, 9..32 rows, mode MAX.  float32 volumes.
// RML_OPT_LINPLANE = 0 turns it off (k_project_wave takes the shape then).
bool try_launch_lin(const ProjParams& pp, int mode, int num_cu, hipStream_t st) {
    if (pp.Z != 4 * pp.ZQ) return false;
    if (pp.B < 2 * (int64_t)num_cu) return false;       // small batches stay on the workgroup-per-frame kernels (latency)
    if (!pp.k_linplane) return false;
    if (pp.ZQ == 44) {
        if (pp.Y <= 16 || pp.Y > 32) return false;
        if (mode == RML_MODE_MAX) launch_lin<RML_MODE_MAX, 11, 11, 16, 2, 1>(pp, num_cu, st);
        else if (mode == RML_MODE_SUM) launch_lin<RML_MODE_SUM, 11, 11, 16, 2, 1>(pp, num_cu, st);
        else return false;
        return true;
    }
    if (mode != RML_MODE_MAX) return false;
    if (pp.ZQ == 40) return launch_lin8<40>(pp, num_cu, st);
    if (pp.ZQ == 48) return launch_lin8<48>(pp, num_cu, st);
    if (pp.ZQ == 56) return launch_lin8<56>(pp, num_cu, st);
    return false;
}

}  // namespace rmlproj
