// The reference-faithful projection path: plane SLICES through a target voxel, and the derived target itself.
//
// Reference sites replaced (paths in goruck/radar-ml):
//   slices            predict.py:102-107, ground_truth_samples.py:413-419:  yz = V[i,:,:], xz = V[:,j,:], xy = V[:,:,k]
//   derived targets   common.py:49-80 (DerivedTarget.get_derived_targets): s_theta[i] = sum_jk V, s_phi[j] = sum_ik V,
//                     s_r[k] = sum_ij V, arg-top-n of each (ascending by value), zipped to (i,j,k) triples
//   feature assembly  common.py:141-148 at zoom 1 (the shared Emitter: float rows, biased uint8 codes, row statistics)
//
// k_slice_rows<VT>: ONE WAVE per output row gathers the three planes.  yz is a contiguous plane (1 KB per load instruction),
// xz is X contiguous rows of Z values (whole 16-byte quads, lanes run over (row, quad)), xy is the only true gather: X*Y
// single values one row apart.  A lane takes FOUR consecutive (i,j) cells so that its float / code stores are whole 16- / 4-byte
// pieces.  Every load of a batch is issued before the first store (the old kernel's scalar loop waited for every element).
// Algorithmic bytes per row: 4*D read + the outputs.  The floor of what the memory system can do is higher: every xy value
// lives in its own 128-byte request (rows are >= 512 B apart), so the plane costs X*Y*128 B whatever the kernel does.
//
// k_derive_slice<VT, P, U>: persistent, ONE WAVE per frame.  The frame is streamed once as the linear array of quads it is
// (64 quads = 1 KB per instruction, non-temporal, the next group of U instructions in flight in a second register buffer across
// plane and frame boundaries -- the structure of k_project_lin), and all three energy profiles come out of that one pass:
//   s_r[k]      element-wise float4 accumulators: instruction t of a plane holds columns (64 t + lane) mod Z/4, which repeats
//               with period P = (Z/4) / gcd(Z/4, 64) instructions -> P float4 registers, folded once per frame through LDS;
//   s_phi[j]    every quad belongs to one row: its horizontal sum is added to a wave-private LDS strip [lane][instruction of the
//               plane] by a plain read-add-write of whole float4 (a lane's U slots of a group are contiguous: U/4 ds_read_b128 +
//               ds_write_b128 per group, conflict-free row stride) -- ds_add_f32 per load instruction, the first version, ran
//               the whole kernel at 0.40 of 8 TB/s instead of 0.86: LDS atomics are the slowest thing a CU has; lane j folds
//               row j once per frame;
//   s_theta[i]  the lane's running sum over the plane, one butterfly over the wave per plane.
// No sum planes through HBM (the old path wrote X*Z + Y*Z floats per frame and read them back), no second launch for the
// top-n: the wave does it on its LDS profiles (wave arg-max with shuffles; ties as radarml.h documents: the lower index ranks
// lower), then gathers the planes of every target with the code above while its next frame's first loads are already in
// flight, and leaves through the Emitter.  Sums of integer-valued data are exact in float32 in any order (< 2^24): the indices
// are bit-exact against NumPy; on other data the float32 rounding follows this kernel's (fixed) order.
#include "project_shared.h"

constexpr int RML_DERIVE_UB = 8;     // quads per gather batch of k_derive_slice (4 and 16 were measured: tools/exp/README.md)

namespace {

using namespace rmlproj;

template <typename VT> struct Cell;
template <> struct Cell<float> {
    typedef float4 Q;
    static __device__ __forceinline__ void emit4(Emitter& em, int pl, int64_t idx, const float4& v) { em.put4(pl, idx, v); }
    static __device__ __forceinline__ void emit4(Emitter& em, int pl, int64_t idx, const float4& v, uint32_t old) { em.put4(pl, idx, v, true, old); }
    static __device__ __forceinline__ float4 pack(float a, float b, float c, float d) { return make_float4(a, b, c, d); }
    static __device__ __forceinline__ float4 widen(const float4& v) { return v; }
};
template <> struct Cell<uint8_t> {
    typedef uint32_t Q;
    static __device__ __forceinline__ void emit4(Emitter& em, int pl, int64_t idx, uint32_t w) { em.put_bytes4(pl, idx, w); }
    static __device__ __forceinline__ void emit4(Emitter& em, int pl, int64_t idx, uint32_t w, uint32_t old) { em.put_bytes4(pl, idx, w, true, old); }
    static __device__ __forceinline__ uint32_t pack(uint8_t a, uint8_t b, uint8_t c, uint8_t d) {
        return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
    }
    static __device__ __forceinline__ float4 widen(uint32_t w) { return bytes_to_float4(w); }
};

// floor(q / d) for 0 <= q < 2^22 through the float reciprocal, corrected by one step either way
__device__ __forceinline__ int div_small(int q, int d, float rcp) {
    int r = (int)((float)q * rcp);
    r = (r * d > q) ? r - 1 : r;
    r = ((r + 1) * d <= q) ? r + 1 : r;
    return r;
}

// the three planes through (i, j, k) of the frame at Vf, by one wave; indices already wrapped into range.  Every address is the
// wave-uniform frame base plus a 32-bit element offset (a frame is far below 4 G elements): the loads take the scalar-base form
// and need one offset register each instead of a 64-bit pointer pair
// RMW (read-compare-write of the code rows, Emitter::rmw): the old words of a batch are loaded with the batch's volume quads --
// unconditionally, their latency under the gather's own -- and handed to the stores
template <typename VT, int UB, bool RMW = false>
__device__ __forceinline__ void slice_emit(const VT* __restrict__ Vf, int i, int j, int k, int X, int Y, int Z, int ZQ,
                                           Emitter& em, int lane) {
    typedef typename Cell<VT>::Q QT;
    const QT* __restrict__ Vq = reinterpret_cast<const QT*>(Vf);
    const uint32_t pq = (uint32_t)(Y * ZQ);
    const uint32_t sel = em.a.o.sel;
    const uint32_t ul = (uint32_t)lane;
    if (sel & 2u) {                                     // yz = V[i, :, :]: the plane as it lies in memory
        const uint32_t o0 = (uint32_t)i * pq;
        for (uint32_t base = 0; base < pq; base += 64 * UB) {
            QT v[UB];
            uint32_t ow[UB];
            static_for<UB>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const uint32_t q = base + u * 64 + ul;
                v[u] = Vq[o0 + (q < pq ? q : pq - 1)];
                if constexpr (RMW) ow[u] = em.old_word_nocheck(1, (int64_t)(q < pq ? q : pq - 1) * 4);
            });
            static_for<UB>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const uint32_t q = base + u * 64 + ul;
                if constexpr (RMW) { if (q < pq) Cell<VT>::emit4(em, 1, (int64_t)q * 4, v[u], ow[u]); }
                else if (q < pq) Cell<VT>::emit4(em, 1, (int64_t)q * 4, v[u]);
            });
        }
    }
    if (sel & 1u) {                                     // xz = V[:, j, :]: X rows of Z/4 quads, one plane apart
        const uint32_t n = (uint32_t)(X * ZQ);
        const float rcp = 1.0f / (float)ZQ;
        const uint32_t o0 = (uint32_t)(j * ZQ);
        for (uint32_t base = 0; base < n; base += 64 * UB) {
            QT v[UB];
            uint32_t ow[UB];
            static_for<UB>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                uint32_t q = base + u * 64 + ul;
                q = q < n ? q : n - 1;
                const uint32_t ii = (uint32_t)div_small((int)q, ZQ, rcp);
                v[u] = Vq[o0 + ii * pq + (q - ii * (uint32_t)ZQ)];
                if constexpr (RMW) ow[u] = em.old_word_nocheck(0, (int64_t)q * 4);
            });
            static_for<UB>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const uint32_t q = base + u * 64 + ul;
                if constexpr (RMW) { if (q < n) Cell<VT>::emit4(em, 0, (int64_t)q * 4, v[u], ow[u]); }
                else if (q < n) Cell<VT>::emit4(em, 0, (int64_t)q * 4, v[u]);
            });
        }
    }
    if (sel & 4u) {                                     // xy = V[:, :, k]: one value per row of the volume
        const uint32_t n = (uint32_t)(X * Y), n4 = n >> 2;
        const uint32_t uz = (uint32_t)Z, uk = (uint32_t)k;
        constexpr int UG = UB / 2 > 0 ? UB / 2 : 1;
        for (uint32_t base = 0; base < n4; base += 64 * UG) {
            VT v[UG][4];
            uint32_t ow[UG];
            static_for<UG>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                uint32_t q = base + u * 64 + ul;
                q = q < n4 ? q : n4 - 1;
                const uint32_t e0 = q * 4 * uz + uk;
                v[u][0] = Vf[e0]; v[u][1] = Vf[e0 + uz]; v[u][2] = Vf[e0 + 2 * uz]; v[u][3] = Vf[e0 + 3 * uz];
                if constexpr (RMW) ow[u] = em.old_word_nocheck(2, (int64_t)q * 4);
            });
            static_for<UG>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                const uint32_t q = base + u * 64 + ul;
                if constexpr (RMW) { if (q < n4) Cell<VT>::emit4(em, 2, (int64_t)q * 4, Cell<VT>::pack(v[u][0], v[u][1], v[u][2], v[u][3]), ow[u]); }
                else if (q < n4) Cell<VT>::emit4(em, 2, (int64_t)q * 4, Cell<VT>::pack(v[u][0], v[u][1], v[u][2], v[u][3]));
            });
        }
        const uint32_t idx = n4 * 4 + ul;
        if (ul < (n & 3)) em.put1(2, idx, (float)Vf[idx * uz + uk]);
    }
}

// Python's negative-index wrap (the host validated the range; clamp defensively)
__device__ __forceinline__ int wrap_index(int v, int n) {
    v = v < 0 ? v + n : v;
    return min(max(v, 0), n - 1);
}

// ------------------------------------------------------------------------------------------
// mode SLICE with (i,j,k) given: one wave per output row (row r reads frame r / tpf)
// ------------------------------------------------------------------------------------------
template <typename VT>
__global__ __launch_bounds__(kThreads) void k_slice_rows(ProjParams a) {
    if (a.o.skip_if_set && *a.o.skip_if_set) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= a.B) return;
    const int X = a.X, Y = a.Y, Z = a.Z;
    const VT* __restrict__ Vf = static_cast<const VT*>(a.V) + (r / a.tpf) * (int64_t)X * Y * Z;
    const int i = wrap_index(__builtin_amdgcn_readfirstlane(a.ijk[r * 3 + 0]), X);
    const int j = wrap_index(__builtin_amdgcn_readfirstlane(a.ijk[r * 3 + 1]), Y);
    const int k = wrap_index(__builtin_amdgcn_readfirstlane(a.ijk[r * 3 + 2]), Z);
    Emitter em(a, r);
    slice_emit<VT, 8>(Vf, i, j, k, X, Y, Z, a.ZQ, em, lane);
    em.finish_wave(lane);
}

// ------------------------------------------------------------------------------------------
// derive -> slice, fused: persistent, one wave per frame (see the header comment)
// ------------------------------------------------------------------------------------------
template <typename VT, int P, int U>
__global__ __launch_bounds__(512) void k_derive_slice(ProjParams a) {
    static_assert(U % P == 0, "the column of an instruction must be a compile-time function of its slot in the group");
    typedef typename Quad<VT>::T QT;
    const int X = a.X, Y = a.Y, Z = a.Z, ZQ = a.ZQ;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = (int)(blockDim.x >> 6);             // 4 waves per workgroup; 8 beside a GEMM (one workgroup per CU)
    const int64_t stride = (int64_t)gridDim.x * wpb;
    int64_t cf = (int64_t)blockIdx.x * wpb + wave;      // frame being reduced
    if (cf >= a.B) return;
    const QT* __restrict__ Vall = reinterpret_cast<const QT*>(a.V);
    const int pq = Y * ZQ;                              // quads per plane
    const int64_t fq = (int64_t)X * pq;
    const int NI = (pq + 63) >> 6;                      // load instructions per plane
    const int NG = (NI + U - 1) / U;                    // groups of U instructions per plane (the last one may run past it)
    const int GP = X * NG;                              // groups per frame
    const int T = a.ntgt;
    // wave-private LDS: [strip: the row sums per quad of a plane, later the staged s_r accumulators][s_theta | s_phi | s_r][targets]
    extern __shared__ __align__(16) unsigned char ds_smem[];
    unsigned char* mine = ds_smem + (size_t)wave * a.wave_lds;
    float* strip = reinterpret_cast<float*>(mine);
    constexpr int UP = (U + 3) & ~3;                    // a group's slots in a lane's strip row, padded to whole float4
    constexpr int PH = (P + 1) / 2;                     // the s_r accumulators are staged through the strip in two halves
    const int RS = NG * UP + 4;                         // row stride in floats: an odd number of float4 (NG * UP / 4 + 1 when that is
    const int RSo = ((RS / 4) & 1) ? RS : RS + 4;       // odd): the 16 lanes of a b128 access hit 16 different bank groups
    const int strip_n = 64 * RSo;
    const int strip_alloc = strip_n > PH * 256 ? strip_n : PH * 256;
    float* prof = strip + strip_alloc;
    int* tgt = reinterpret_cast<int*>(prof + ((X + Y + Z + 3) & ~3));
    for (int q = lane; q < strip_alloc; q += 64) strip[q] = 0.0f;
    float4* const myrow = reinterpret_cast<float4*>(strip + lane * RSo);

    // load cursor: the group whose loads are in flight; frames are assigned statically (wave w: w, w + #waves, ...).  ONE register
    // buffer of U quads: slot u is refilled with the next group's quad right behind its reduction, so U loads (U KB) are in
    // flight per wave at any time, across plane and frame boundaries (the first version alternated two buffers filled in bursts:
    // twice the registers for between U and 2 U loads in flight)
    int64_t lf = cf;
    int lgi = 0;                                        // group of the frame
    int lg = 0;                                         // group of the plane
    const QT* __restrict__ lV = Vall + lf * fq;
    const uint32_t qlast = (uint32_t)(pq - 1);
    auto advance = [&]() __attribute__((always_inline)) {
        ++lgi; ++lg;
        if (lgi == GP) {                                // next frame of this wave; past the end: re-read (never consumed)
            lgi = 0; lg = 0;
            const int64_t nf = lf + stride;
            lf = nf < a.B ? nf : lf;
            lV = Vall + lf * fq;
        } else if (lg == NG) {
            lg = 0;
            lV += pq;
        }
    };
    // unconditional: lanes past the plane re-read its last quad and are zeroed after the load
    auto load1 = [&](const QT* __restrict__ base, uint32_t q) __attribute__((always_inline)) -> QT {
        if constexpr (sizeof(VT) == 4) {
            typedef float v4f_t __attribute__((ext_vector_type(4)));
            v4f_t t = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(base + (q < qlast ? q : qlast)));
            return make_float4(t.x, t.y, t.z, t.w);
        } else {
            return __builtin_nontemporal_load(base + (q < qlast ? q : qlast));
        }
    };
    QT buf[U];
    static_for<U>([&](auto uc) { constexpr int u = decltype(uc)::value; buf[u] = load1(lV, (uint32_t)(u * 64 + lane)); });
    Emitter em(a, cf * T);
    for (; cf < a.B; cf += stride) {
        float4 accr[P];
        static_for<P>([&](auto pc) { accr[decltype(pc)::value] = make_float4(0.f, 0.f, 0.f, 0.f); });
        float th = 0.0f;
        int ci = 0, cg = 0;
        for (int gi = 0; gi < GP; ++gi) {
            advance();                                  // the group that refills the slots
            const QT* __restrict__ nV = lV;
            uint32_t nq0 = (uint32_t)(lg * U * 64 + lane);
            asm volatile("" : "+v"(nq0));               // opaque per step: no hoisted per-instruction offsets kept alive
            const int q0 = cg * U * 64 + lane;
            // the running row sums of this group's slots live in the lane's strip row: float4 by float4, read one ahead
            float4* const slot = myrow + cg * (UP / 4);
            float4 o = slot[0];
            static_for<UP / 4>([&](auto wc) {
                constexpr int w = decltype(wc)::value;
                float4 on = o;
                if constexpr (w + 1 < UP / 4) on = slot[w + 1];
                static_for<4>([&](auto kc) {
                    constexpr int u = 4 * w + decltype(kc)::value;
                    if constexpr (u < U) {
                        float4 v = Cell<VT>::widen(buf[u]);
                        const bool in = q0 + u * 64 < pq;
                        v.x = in ? v.x : 0.0f; v.y = in ? v.y : 0.0f; v.z = in ? v.z : 0.0f; v.w = in ? v.w : 0.0f;
                        const float h = (v.x + v.y) + (v.z + v.w);
                        th += h;
                        if constexpr (u % 4 == 0) o.x += h; else if constexpr (u % 4 == 1) o.y += h; else if constexpr (u % 4 == 2) o.z += h; else o.w += h;
                        accr[u % P].x += v.x; accr[u % P].y += v.y; accr[u % P].z += v.z; accr[u % P].w += v.w;
                        asm volatile("" : "+v"(accr[u % P].x), "+v"(accr[u % P].y), "+v"(accr[u % P].z), "+v"(accr[u % P].w));  // pin the update here
                        buf[u] = load1(nV, nq0 + (uint32_t)(u * 64));
                    }
                });
                slot[w] = o;
                o = on;
                asm volatile("" : "+v"(th));
                // keep the software pipeline as written: left alone hipcc renames the refills into fresh registers and hoists them
                __builtin_amdgcn_sched_barrier(0);
            });
            ++cg;
            if (cg == NG) {                             // plane ci of frame cf is complete
                float tot = th;                         // butterfly over the wave: a fixed order, every lane ends with the total
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) tot += __shfl_xor(tot, off);
                if (lane == 0) prof[ci] = tot;
                th = 0.0f;
                cg = 0; ++ci;
            }
        }
        // ---- the frame is in: profiles, top-n, planes.  The first group of this wave's next frame is in flight meanwhile. ----
        // (the LDS hand-offs below are between lanes of ONE wave: the DS unit runs a wave's instructions in order; the fences
        // only keep the compiler from moving an access across a hand-off)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // s_phi[j]: row j = quads [j Z/4, (j+1) Z/4) of the strip
        for (int j = lane; j < Y; j += 64) {
            float sum = 0.0f;
            for (int c = 0; c < ZQ; ++c) {
                const int q = j * ZQ + c;                // quad of the plane -> (instruction t, lane l) -> slot (group, u) of lane l's row
                const int t = q >> 6, l = q & 63;
                sum += strip[l * RSo + (t / U) * UP + (t % U)];
            }
            prof[X + j] = sum;
        }
        // s_r[k]: the P accumulators cover quads 0 .. 64 P - 1 of the linear plane modulo its period; column c = q mod Z/4.  Staged
        // through the strip in two halves (PH accumulators each), folded in ascending quad order
        float4* st4 = reinterpret_cast<float4*>(strip);
        float4 srs = make_float4(0.f, 0.f, 0.f, 0.f);
        const int csr = lane < ZQ ? lane : ZQ - 1;       // Z/4 <= 64: one column per lane
        static_for<2>([&](auto hc) {
            constexpr int hh = decltype(hc)::value;
            if constexpr (hh * PH < P) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                static_for<P>([&](auto pc) { constexpr int p = decltype(pc)::value; if constexpr (p / PH == hh) st4[(p - hh * PH) * 64 + lane] = accr[p]; });
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int lo = hh * PH * 64, hi = (hh + 1) * PH * 64 < P * 64 ? (hh + 1) * PH * 64 : P * 64;
                int m = csr + ((lo - csr + ZQ - 1) / ZQ) * ZQ;      // first quad >= lo of column csr (lo >= 0 > csr - ZQ)
                if (lo <= csr) m = csr;
                for (; m < hi; m += ZQ) {
                    const float4 t = st4[m - lo];
                    srs.x += t.x; srs.y += t.y; srs.z += t.z; srs.w += t.w;
                }
            }
        });
        if (lane < ZQ) {
            float* dst = prof + X + Y + 4 * lane;       // X + Y need not be a multiple of four: no 16-byte store
            dst[0] = srs.x; dst[1] = srs.y; dst[2] = srs.z; dst[3] = srs.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int q = lane; q < strip_alloc; q += 64) strip[q] = 0.0f;   // the next frame accumulates into a clean strip
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int NPROF = X + Y + Z;
        if (a.profiles)
            for (int t = lane; t < NPROF; t += 64) a.profiles[cf * NPROF + t] = prof[t];
        // arg-top-T of each profile, ascending by value (largest last); ties: the higher index is the larger (radarml.h)
        for (int ax = 0; ax < 3; ++ax) {
            float* sp = prof + (ax == 0 ? 0 : (ax == 1 ? X : X + Y));
            const int L = ax == 0 ? X : (ax == 1 ? Y : Z);
            for (int t = 0; t < T; ++t) {
                float bv = -INFINITY;
                int bi = -1;
                for (int q = lane; q < L; q += 64) {
                    float v = sp[q];
                    v = (v != v) ? -INFINITY : v;
                    if (bi < 0 || v >= bv) { bv = v; bi = q; }
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) {
                    const float ov = __shfl_xor(bv, off);
                    const int oi = __shfl_xor(bi, off);
                    const bool take = (ov > bv) || (ov == bv && oi > bi);
                    bv = take ? ov : bv;
                    bi = take ? oi : bi;
                }
                const int best = __builtin_amdgcn_readfirstlane(bi);
                if (lane == 0) {
                    sp[best] = -INFINITY;
                    tgt[(T - 1 - t) * 3 + ax] = best;
                    if (a.ijk_out) a.ijk_out[(cf * T + (T - 1 - t)) * 3 + ax] = best;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        if (a.o.sel) {
            const VT* __restrict__ Vf = static_cast<const VT*>(a.V) + cf * (int64_t)X * Y * Z;
            for (int t = 0; t < T; ++t) {
                const int i = __builtin_amdgcn_readfirstlane(tgt[t * 3 + 0]);
                const int j = __builtin_amdgcn_readfirstlane(tgt[t * 3 + 1]);
                const int k = __builtin_amdgcn_readfirstlane(tgt[t * 3 + 2]);
                em.reset(cf * T + t);
                // codes only (the pipelines' first pass) with read-compare-write: the batched form; float rows or plain stores: as before
                if (em.rmw && !a.o.p[0] && !a.o.p[1] && !a.o.p[2] && !a.o.row_nsq) slice_emit<VT, RML_DERIVE_UB, true>(Vf, i, j, k, X, Y, Z, ZQ, em, lane);
                else slice_emit<VT, RML_DERIVE_UB>(Vf, i, j, k, X, Y, Z, ZQ, em, lane);
                em.finish_wave(lane);
            }
        }
    }
}

int odd_part(int v) { while (v > 0 && !(v & 1)) v >>= 1; return v; }

struct DeriveGeom { int P, U, NG, strip_alloc; size_t wave_lds; };

bool derive_geom(int X, int Y, int Z, int ntgt, DeriveGeom* g) {
    if (Z % 4 != 0) return false;
    const int ZQ = Z / 4;
    if (ZQ > 64) return false;
    const int P = odd_part(ZQ);
    if (P > 15) return false;
    const int U = P == 1 ? 8 : (P == 3 ? 9 : (P == 5 ? 10 : P));     // (two periods in flight per wave -- U = 16 / 22 -- measured slower: tools/exp/README.md)
    const int pq = Y * ZQ;
    const int NI = (pq + 63) / 64, NG = (NI + U - 1) / U;
    const int UP = (U + 3) & ~3, PH = (P + 1) / 2;
    int RS = NG * UP + 4;
    if (!((RS / 4) & 1)) RS += 4;                       // as the kernel computes it
    const int strip_n = 64 * RS;
    g->P = P; g->U = U; g->NG = NG;
    g->strip_alloc = strip_n > PH * 256 ? strip_n : PH * 256;
    const size_t bytes = (size_t)g->strip_alloc * 4 + (size_t)((X + Y + Z + 3) & ~3) * 4 + (size_t)ntgt * 12;
    g->wave_lds = (bytes + 15) & ~(size_t)15;
    return 4 * g->wave_lds <= 150 * 1024;
}

template <typename VT, int P, int U>
void launch_derive_pu(const ProjParams& pp, size_t wave_lds, int num_cu, hipStream_t st) {
    // persistent grid: as many 4-wave workgroups per CU as LDS allows (at most five; the registers decide what is resident).
    // Beside a GEMM (share_cu, the fused pipeline): ONE workgroup of eight waves per CU, its LDS request padded past half of the
    // CU's LDS so that the dispatcher cannot put two on one CU and none on another, while a 128x128 GEMM workgroup (69.6 KB, 128
    // registers) still fits next to it (see launch_wave in project.hip)
    const bool share = pp.o.share_cu != 0;
    const int wpb = (share && 8 * wave_lds <= 150 * 1024) ? 8 : 4;
    size_t lds = (size_t)wpb * wave_lds;
    int per_cu = (int)((size_t)(160 * 1024) / (lds + 512));
    per_cu = per_cu < 1 ? 1 : (per_cu > 5 ? 5 : per_cu);
    if (share) {
        per_cu = 1;
        if (!pp.o.no_pad && lds < 81 * 1024) lds = 81 * 1024;
    }
    const int64_t want = (pp.B + wpb - 1) / wpb;
    const int64_t cap = (int64_t)num_cu * per_cu;
    dim3 grid((unsigned)(want < cap ? want : cap)), block((unsigned)(64 * wpb));
    RML_MAX_DYN_LDS(160 * 1024, &k_derive_slice<VT, P, U>);
    hipLaunchKernelGGL((k_derive_slice<VT, P, U>), grid, block, lds, st, pp);
}

template <typename VT>
bool launch_derive_t(const ProjParams& pp, const DeriveGeom& g, int num_cu, hipStream_t st) {
    const size_t lds = g.wave_lds;                      // per wave; the launcher multiplies by its waves per workgroup
    switch (g.P) {
        case 1: launch_derive_pu<VT, 1, 8>(pp, lds, num_cu, st); return true;
        case 3: launch_derive_pu<VT, 3, 9>(pp, lds, num_cu, st); return true;
        case 5: launch_derive_pu<VT, 5, 10>(pp, lds, num_cu, st); return true;
        case 7: launch_derive_pu<VT, 7, 7>(pp, lds, num_cu, st); return true;
        case 9: launch_derive_pu<VT, 9, 9>(pp, lds, num_cu, st); return true;
        case 11: launch_derive_pu<VT, 11, 11>(pp, lds, num_cu, st); return true;
        case 13: launch_derive_pu<VT, 13, 13>(pp, lds, num_cu, st); return true;
        case 15: launch_derive_pu<VT, 15, 15>(pp, lds, num_cu, st); return true;
        default: return false;
    }
}

bool quads_ok(const ProjParams& pp, int vbytes) {
    // whole quads: rows of a multiple of four voxels, frames that start on a quad
    return pp.Z % 4 == 0 && (reinterpret_cast<uintptr_t>(pp.V) & (size_t)(4 * vbytes - 1)) == 0;
}

}  // namespace

namespace rmlproj {

// mode SLICE, (i,j,k) given.  RML_OPT_SLICE_WAVE = 0 keeps the round-1 workgroup-per-row kernel (A/B knob).
bool try_launch_slice(const ProjParams& pp, int vbytes, hipStream_t st) {
    if (!quads_ok(pp, vbytes) || pp.B <= 0) return false;
    if (!pp.k_slice_wave) return false;
    dim3 grid((unsigned)((pp.B + 3) / 4)), block(kThreads);
    if (vbytes == 1) hipLaunchKernelGGL(k_slice_rows<uint8_t>, grid, block, 0, st, pp);
    else hipLaunchKernelGGL(k_slice_rows<float>, grid, block, 0, st, pp);
    return true;
}

bool derive_slice_shape_ok(int X, int Y, int Z, int ntgt) {
    DeriveGeom g;
    return ntgt >= 1 && derive_geom(X, Y, Z, ntgt, &g);
}

// derive (-> slice when pp.o.sel != 0) in one pass; pp.B frames, pp.ntgt targets each: outputs have B * ntgt rows.
// false: the shape has no fused kernel (rows that are not whole quads, Z > 256, odd part of Z/4 above 15)
bool try_launch_derive_slice(const ProjParams& pp_in, int vbytes, int num_cu, hipStream_t st) {
    DeriveGeom g;
    if (!quads_ok(pp_in, vbytes) || pp_in.B <= 0 || pp_in.ntgt < 1 || !pp_in.k_derive_fused) return false;
    if (!derive_geom(pp_in.X, pp_in.Y, pp_in.Z, pp_in.ntgt, &g)) return false;
    ProjParams pp = pp_in;
    pp.wave_lds = (int)g.wave_lds;
    return vbytes == 1 ? launch_derive_t<uint8_t>(pp, g, num_cu, st) : launch_derive_t<float>(pp, g, num_cu, st);
}

}  // namespace rmlproj
