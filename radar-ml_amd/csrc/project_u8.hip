// k_project_u8_max: the byte-native max-projection of uint8 volumes (data-set format datasets/README.md:8-20:
// the radar's magnitudes are integers 0..255).  Dispatched from launch_mode in project.hip.
#include "project_shared.h"

namespace {

using namespace rmlproj;

// ------------------------------------------------------------------------------------------
// uint8 volumes, mode MAX: the byte-native path.  Widening every voxel to float (the path above) is VALU-bound at a
// third of the HBM rate once the volume is 1 byte per voxel, so here the data stays packed: a lane loads 16
// voxels of a row (uint4), splits every dword once into its even and odd bytes as 16-bit pairs (v_perm_b32) and all
// maxima are v_pk_max_u16 on those pairs -- 2 voxels per instruction.
//   * wave w owns the planes i = w, w+4, ...; lane = (row slot s, 16-byte chunk c), rows j = s + S*m.
//   * yz (max over i) accumulates in registers per wave and the 4 wave partials meet in LDS once per frame;
//   * xz (max over j) is combined in-lane over m, then across the row slots with ds_bpermute -- a plane belongs to
//     one wave, so there are no LDS atomics at all;
//   * xy (max over z) is reduced in-lane to one 16-bit value per row, two rows share a dword through the
//     cross-lane steps over the chunks of a row.
// Projections are staged in LDS as bytes and leave through the same Emitter as every other path (scaled float rows,
// biased codes, statistics).  Invalid rows / idle lanes re-read a valid neighbour: a duplicate is harmless for max.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t pkmax_u16(uint32_t a, uint32_t b) {
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    u16x2 x = *reinterpret_cast<u16x2*>(&a), y = *reinterpret_cast<u16x2*>(&b);
    u16x2 r = __builtin_elementwise_max(x, y);
    return *reinterpret_cast<uint32_t*>(&r);
}

// FCPR = 8 / 16 (rows of 128 / 256 voxels: the chunks of a row and the row slots are lane-index bit fields): the cross-lane steps
// run on the VALU instead of LDS (ds_bpermute): xy through DPP moves, xz as a reduce-SCATTER -- v_permlane32_swap and
// v_permlane16_swap exchange half of the values a lane still holds, so 8 values per lane cost 4 + 2 (+ 2 DPP) exchanges instead of
// 3 x 8 -- and the lane that ends up with dword q of its chunk stores it.  FCPR = 0: any row length, the ds_bpermute path.
template <int CTRL> __device__ __forceinline__ uint32_t dpp_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
}

template <int NM, bool PRED, int FCPR, bool RMW = false>
// 192 registers: two of these workgroups and the 128-register waves of one k_svm_gemm workgroup share a SIMD's 512 in the fused pipeline
__global__ __launch_bounds__(kThreads) void k_project_u8_max(ProjParams a, int CPR_rt, int S) {
    extern __shared__ __align__(16) unsigned char lds8[];
    if constexpr (PRED) { if (*a.o.skip_if_set) return; }
    const int X = a.X, Y = a.Y, Z = a.Z;
    // CPR = chunks per row in the LANE geometry, CPRm = chunks per row in memory.  They differ when a row that is not a power of
    // two of chunks (the Walabot grid: 11) runs in the next power-of-two geometry (16 lanes per row, 5 of them idle: they re-read
    // the row's last chunk, a duplicate is harmless for a maximum, and store nothing) to get the VALU cross-lane steps
    const int CPR = FCPR ? FCPR : CPR_rt;
    const int CPRm = CPR_rt;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // plane indices and their addresses stay scalar
    const int s = lane / CPR, c = lane - s * CPR;
    const bool cval = c < CPRm;
    const int cm = cval ? c : CPRm - 1;
    const int64_t b = blockIdx.x;
    // LDS: yz [4 waves][Y][Z], xz [X][Z], xy [X*Y] (dense).  The reduction scratch of Emitter::finish lies over yz
    // (finish synchronises before it writes): two of these workgroups and one k_svm_gemm workgroup (69.6 KB) then
    // fit a CU together at 64x64x128, which is what lets the fused pipeline overlap them.
    unsigned char* yz_s = lds8;
    unsigned char* xz_s = yz_s + (size_t)4 * Y * Z;
    unsigned char* xy_s = xz_s + (size_t)X * Z;
    int64_t* red = reinterpret_cast<int64_t*>(lds8);
    const uint4* __restrict__ Vb = reinterpret_cast<const uint4*>(static_cast<const uint8_t*>(a.V) + b * (int64_t)X * Y * Z);
    const int plane = Y * CPRm;         // uint4 per x-plane

    int roff[NM];
    bool rv[NM];
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int j = s + S * m;
        rv[m] = (s < S) && (j < Y);
        roff[m] = min(j, Y - 1) * CPRm + cm;
    }
    // cross-lane sources (byte addresses for ds_bpermute); an invalid source is the lane itself (max(x,x) = x)
    int xsrc[6], ysrc[6];
    int xsteps = 0, ysteps = 0;
    if constexpr (FCPR == 0) {
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int ox = CPR << t, oy = 1 << t;
            xsrc[t] = ((lane + ox < 64) ? lane + ox : lane) << 2;
            ysrc[t] = ((c + oy < CPR) ? lane + oy : lane) << 2;
        }
        const int nslots = (64 + CPR - 1) / CPR;
        while ((1 << xsteps) < nslots) ++xsteps;
        while ((1 << ysteps) < CPR) ++ysteps;
    }

    uint32_t yz[NM][8];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int k = 0; k < 8; ++k) yz[m][k] = 0u;

    typedef uint32_t v4u_t __attribute__((ext_vector_type(4)));
    for (int i = wave; i < X; i += 4) {
        const uint4* __restrict__ Vi = Vb + (int64_t)i * plane;
        v4u_t cur[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) cur[m] = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(Vi + roff[m]));
        uint32_t xz[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xz[k] = 0u;
        uint32_t rowmax[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const uint32_t w[4] = {cur[m].x, cur[m].y, cur[m].z, cur[m].w};
            uint32_t h[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                h[2 * q] = __builtin_amdgcn_perm(w[q], w[q], 0x0c020c00u);          // bytes 0 and 2 as 16-bit values
                h[2 * q + 1] = __builtin_amdgcn_perm(w[q], w[q], 0x0c030c01u);      // bytes 1 and 3
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                yz[m][k] = pkmax_u16(yz[m][k], h[k]);
                xz[k] = pkmax_u16(xz[k], h[k]);
            }
            const uint32_t t0 = pkmax_u16(pkmax_u16(h[0], h[1]), pkmax_u16(h[2], h[3]));
            const uint32_t t1 = pkmax_u16(pkmax_u16(h[4], h[5]), pkmax_u16(h[6], h[7]));
            const uint32_t u = pkmax_u16(t0, t1);
            rowmax[m] = max(u & 0xFFFFu, u >> 16);
        }
        // xy: rows m and m+1 share a dword through the reduction over the chunks of a row
#pragma unroll
        for (int m = 0; m < NM; m += 2) {
            uint32_t rp = rowmax[m] | ((m + 1 < NM ? rowmax[m + 1] : 0u) << 16);
            if constexpr (FCPR != 0) {
                rp = pkmax_u16(rp, dpp_u32<0xB1>(rp));          // quad_perm [1,0,3,2]
                rp = pkmax_u16(rp, dpp_u32<0x4E>(rp));          // quad_perm [2,3,0,1]
                rp = pkmax_u16(rp, dpp_u32<0x141>(rp));         // row_half_mirror: the other quad of the 8 lanes
                if constexpr (FCPR == 16) rp = pkmax_u16(rp, dpp_u32<0x140>(rp));      // row_mirror: the other half of the 16 lanes
            } else {
                for (int t = 0; t < ysteps; ++t) rp = pkmax_u16(rp, (uint32_t)__builtin_amdgcn_ds_bpermute(ysrc[t], (int)rp));
            }
            if (c == 0) {
                if (rv[m]) xy_s[i * Y + s + S * m] = (unsigned char)(rp & 0xFFu);
                if (m + 1 < NM && rv[m + 1]) xy_s[i * Y + s + S * (m + 1)] = (unsigned char)(rp >> 16);
            }
        }
        // xz: across the row slots of the wave
        if constexpr (FCPR != 0) {
            // value k = 2*q + parity (dword q of the chunk; its even / odd bytes).  Lanes 32..63 keep q = 2,3, odd 16-lane rows keep
            // the odd q of their half: a lane ends with both parities of dword q = bit4 + 2*bit5 of its lane index
            uint32_t v4[4], u2[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const auto r = __builtin_amdgcn_permlane32_swap(xz[k], xz[k + 4], false, false);
                v4[k] = pkmax_u16(r[0], r[1]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const auto r = __builtin_amdgcn_permlane16_swap(v4[j], v4[j + 2], false, false);
                u2[j] = pkmax_u16(r[0], r[1]);
            }
            if constexpr (FCPR == 8) {
#pragma unroll
                for (int j = 0; j < 2; ++j) u2[j] = pkmax_u16(u2[j], dpp_u32<0x128>(u2[j]));       // row_ror:8 -- the other row slot of the 16 lanes
            }
            const int q = ((lane >> 4) & 1) + 2 * (lane >> 5);
            if ((FCPR == 16 || (lane & 8) == 0) && cval)
                *reinterpret_cast<uint32_t*>(xz_s + (size_t)i * Z + 16 * c + 4 * q) = u2[0] | (u2[1] << 8);
        } else {
            for (int t = 0; t < xsteps; ++t) {
#pragma unroll
                for (int k = 0; k < 8; ++k) xz[k] = pkmax_u16(xz[k], (uint32_t)__builtin_amdgcn_ds_bpermute(xsrc[t], (int)xz[k]));
            }
            if (s == 0)
                *reinterpret_cast<uint4*>(xz_s + (size_t)i * Z + 16 * c) =
                    make_uint4(xz[0] | (xz[1] << 8), xz[2] | (xz[3] << 8), xz[4] | (xz[5] << 8), xz[6] | (xz[7] << 8));
        }
    }
    // the wave's yz partial
#pragma unroll
    for (int m = 0; m < NM; ++m)
        if (rv[m] && cval)
            *reinterpret_cast<uint4*>(yz_s + ((size_t)wave * Y + s + S * m) * Z + 16 * c) =
                make_uint4(yz[m][0] | (yz[m][1] << 8), yz[m][2] | (yz[m][3] << 8), yz[m][4] | (yz[m][5] << 8), yz[m][6] | (yz[m][7] << 8));
    __syncthreads();

    Emitter em(a, b);
    em.rmw = RMW;                       // compile-time here: the plain launch is the kernel it was before read-compare-write came
    // plain stores: a word per thread and trip.  Read-compare-write of the code rows (ProjOut::q_rmw): eight words per thread and
    // trip, their old words in flight together.  (The batched loop for both cost the plain path 10 % at the Walabot grid, whose
    // planes are 1-6 words per thread -- sessions r4av / r4aw, 21.98 -> 19.95 M frames/s --, and a run-time choice between the two
    // loops still 5 %: session r4bf, 22.35 -> 21.28.)
    auto emit_plane = [&](int pl, int n4, auto word_of) __attribute__((always_inline)) {
        if constexpr (!RMW) {
            for (int idx = tid; idx < n4; idx += kThreads) em.put_bytes4(pl, (int64_t)idx * 4, word_of(idx));
        } else
        for (int idx0 = tid; idx0 < n4; idx0 += 8 * kThreads) {
            uint32_t ov[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int idx = idx0 + u * kThreads; ov[u] = em.old_word(pl, (int64_t)(idx < n4 ? idx : idx0) * 4); }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int idx = idx0 + u * kThreads;
                if (idx < n4) em.put_bytes4(pl, (int64_t)idx * 4, word_of(idx), true, ov[u]);
            }
        }
    };
    const int nxz4 = (X * Z) >> 2;
    emit_plane(0, nxz4, [&](int idx) { return *reinterpret_cast<const uint32_t*>(xz_s + idx * 4); });
    const int nyz4 = (Y * Z) >> 2;
    const int nw = X < 4 ? X : 4;       // waves that saw a plane
    emit_plane(1, nyz4, [&](int idx) {
        uint32_t lo = 0u, hi = 0u;
        for (int w = 0; w < nw; ++w) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(yz_s + (size_t)w * Y * Z + idx * 4);
            lo = pkmax_u16(lo, v & 0x00FF00FFu);
            hi = pkmax_u16(hi, (v >> 8) & 0x00FF00FFu);
        }
        return lo | (hi << 8);
    });
    const int nxy = X * Y, nxy4 = nxy >> 2;
    emit_plane(2, nxy4, [&](int idx) { return *reinterpret_cast<const uint32_t*>(xy_s + idx * 4); });
    for (int idx = nxy4 * 4 + tid; idx < nxy; idx += kThreads) em.put1(2, idx, (float)xy_s[idx]);
    em.finish(red);
}

template <int NM, int FCPR>
void launch_u8_max_f(const ProjParams& pp, int CPR, int S, size_t lds_bytes, hipStream_t st) {
    dim3 grid((unsigned)pp.B), block(kThreads);
    if (pp.o.skip_if_set) {
        RML_MAX_DYN_LDS(160 * 1024, (&k_project_u8_max<NM, true, FCPR>));
        hipLaunchKernelGGL((k_project_u8_max<NM, true, FCPR>), grid, block, lds_bytes, st, pp, CPR, S);
    } else if (pp.o.q_rmw) {
        RML_MAX_DYN_LDS(160 * 1024, (&k_project_u8_max<NM, false, FCPR, true>));
        hipLaunchKernelGGL((k_project_u8_max<NM, false, FCPR, true>), grid, block, lds_bytes, st, pp, CPR, S);
    } else {
        RML_MAX_DYN_LDS(160 * 1024, (&k_project_u8_max<NM, false, FCPR>));
        hipLaunchKernelGGL((k_project_u8_max<NM, false, FCPR>), grid, block, lds_bytes, st, pp, CPR, S);
    }
}
template <int NM>
void launch_u8_max(const ProjParams& pp, int CPR, int S, int G, size_t lds_bytes, hipStream_t st) {
    // rows of 128 / 256 voxels: the cross-lane steps on the VALU (the ds_bpermute path for the other geometries)
    // G: the power-of-two lane geometry try_launch_u8_max chose (0: the row's own chunk count -- S alone cannot tell, 64 / 13 is 4 too)
    if (G == 8) launch_u8_max_f<NM, 8>(pp, CPR, S, lds_bytes, st);
    else if (G == 16) launch_u8_max_f<NM, 16>(pp, CPR, S, lds_bytes, st);
    else launch_u8_max_f<NM, 0>(pp, CPR, S, lds_bytes, st);
}

}  // namespace

bool rmlproj::try_launch_u8_max(const ProjParams& pp, hipStream_t st) {
    const int X = pp.X, Y = pp.Y, Z = pp.Z;
    if (Z % 16 != 0 || Z / 16 > 64 || (reinterpret_cast<uintptr_t>(pp.V) & 15) != 0) return false;
    const int CPR = Z / 16;
    // lane geometry: the row's own chunk count, or -- rows of 5..7 / 9..15 chunks whose planes then still fit 8 rows per lane -- the
    // next power of two, which moves the cross-lane steps from ds_bpermute to the VALU
    int G = (CPR == 8 || CPR == 16) ? CPR : 0;
    if (G == 0) {
        const int g2 = CPR > 8 && CPR < 16 ? 16 : (CPR > 4 && CPR < 8 ? 8 : 0);
        if (g2 && (Y + 64 / g2 - 1) / (64 / g2) <= 8) G = g2;
    }
    const int S = 64 / (G ? G : CPR);
    const int nm = (Y + S - 1) / S;
    const ProjParams& pg = pp;
    size_t lds_bytes = (size_t)X * Z + (((size_t)X * Y + 15) & ~(size_t)15) + (size_t)4 * Y * Z;
    if (lds_bytes < 64 * 8 + 64) lds_bytes = 64 * 8 + 64;      // Emitter::finish scratch
    if (nm > 8 || lds_bytes > 150 * 1024) return false;
    switch (nm) {       // rows per lane and plane: exact, so that no lane re-reads rows it does not need
        case 1: launch_u8_max<1>(pg, CPR, S, G, lds_bytes, st); break;
        case 2: launch_u8_max<2>(pg, CPR, S, G, lds_bytes, st); break;
        case 3: launch_u8_max<3>(pg, CPR, S, G, lds_bytes, st); break;
        case 4: launch_u8_max<4>(pg, CPR, S, G, lds_bytes, st); break;
        case 5: launch_u8_max<5>(pg, CPR, S, G, lds_bytes, st); break;
        case 6: launch_u8_max<6>(pg, CPR, S, G, lds_bytes, st); break;
        case 7: launch_u8_max<7>(pg, CPR, S, G, lds_bytes, st); break;
        default: launch_u8_max<8>(pg, CPR, S, G, lds_bytes, st); break;
    }
    return true;
}

