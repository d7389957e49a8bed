// Declarations shared by the projection translation units (project.hip, project_lin.hip): launch parameters, the reduction
// operators, the fused feature / code / statistics emitter and the streaming-load helpers (all inline / templates).
#pragma once
#include "rml_internal.h"
#include <math.h>
#include <stdlib.h>
#include <type_traits>

namespace rmlproj {

constexpr int kThreads = 256;

struct ProjParams {
    const void* V;      // float32 or uint8 voxels (template VT of the kernels)
    int64_t B;
    int X, Y, Z, ZQ;
    const int32_t* ijk;
    int tpf;            // mode SLICE: targets (output rows) per frame; output row b reads frame b / tpf
    int rpl;            // k_project_wave: real rows per 64-quad virtual row, decided by the launcher (wave_kernel_rpl)
    ProjOut o;
    int vec_ok[3];   // float4 stores allowed for plane pl (16-B aligned base and stride)
    // k_derive_slice (project_slice.hip): targets per frame, optional (i,j,k) / profile outputs, LDS bytes per wave
    int ntgt;
    int32_t* ijk_out;
    float* profiles;
    int wave_lds;
    // wave-per-frame kernels, codes-only launches: bytes of wave-private LDS in which the codes of the per-plane outputs (xz, xy) wait
    // for the frame's end (Emitter::stage); 0 = every plane's codes go to memory as they are finished
    int stage_bytes;
    // host side only: the context's kernel-family options (rml_opts; fill_params copies them) for the launchers below
    int k_waveframe, k_linplane, k_stage_codes, k_slice_wave, k_derive_fused;
};

template <int MODE> struct Op;
// Mode MAX is np.max (SURVEY 8 a-1'): a line that holds a NaN gives NaN.  gfx950 has the IEEE-754-2019 `maximum` as ONE instruction
// (v_maximum3_f32: a NaN operand is the result) -- the same issue slot as v_max_f32 / v_max3_f32 (maxNum: drops the NaN), so NumPy's
// policy costs the streaming kernels nothing (rounds 1-5 ran maxNum by default and NumPy's rule only in the untuned RML_MODE_MAX_NAN).
// The cross-wave LDS combine of k_project_fast / k_project_rowgroup cannot use ds_max_f32 (maxNum again): it runs ds_max_u32 on an
// order-preserving key of the float -- every NaN maps to the largest key.
template <> struct Op<RML_MODE_MAX> {
    static __device__ __forceinline__ float ident() { return -INFINITY; }
    static __device__ __forceinline__ float f(float a, float b) { return __builtin_elementwise_maximum(a, b); }
    static __device__ __forceinline__ uint32_t key(float v) {
        const uint32_t b = __float_as_uint(v);
        const uint32_t k = b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);      // negative: ~b; non-negative: b | sign bit
        return v != v ? 0xFFFFFFFFu : k;
    }
    static __device__ __forceinline__ float lds_ident() { return __uint_as_float(key(-INFINITY)); }
    static __device__ __forceinline__ float lds_value(float stored) {            // what the LDS image holds -> the float it stands for
        const uint32_t k = __float_as_uint(stored);
        const uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
        return k == 0xFFFFFFFFu ? __uint_as_float(0x7FC00000u) : __uint_as_float(b);
    }
    static __device__ __forceinline__ void lds_atomic(float* p, float v) {
        __hip_atomic_fetch_max(reinterpret_cast<uint32_t*>(p), key(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};
template <> struct Op<RML_MODE_SUM> {
    static __device__ __forceinline__ float ident() { return 0.0f; }
    static __device__ __forceinline__ float f(float a, float b) { return a + b; }
    static __device__ __forceinline__ float lds_ident() { return 0.0f; }
    static __device__ __forceinline__ float lds_value(float stored) { return stored; }
    static __device__ __forceinline__ void lds_atomic(float* p, float v) {
        __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};

template <int MODE> __device__ __forceinline__ float4 op4(float4 a, float4 b) {
    return make_float4(Op<MODE>::f(a.x, b.x), Op<MODE>::f(a.y, b.y), Op<MODE>::f(a.z, b.z), Op<MODE>::f(a.w, b.w));
}

__device__ __forceinline__ float4 bytes_to_float4(uint32_t w) {
    return make_float4((float)(w & 0xFFu), (float)((w >> 8) & 0xFFu), (float)((w >> 16) & 0xFFu), (float)(w >> 24));
}

// Per-thread sink for finished projection values: scaled float store, uint8 code store and
// the row statistics of the exact-integer SVM path.
struct Emitter {
    const ProjParams& a;
    int64_t b;
    __device__ __forceinline__ int64_t qb() const { return b; }
    int32_t isum = 0;
    uint32_t isq = 0;       // per THREAD: < 66 000 codes of <= 255^2 each (the launchers keep a thread's share far below that)
    int ok = 1;
    double nsq = 0.0;
    bool want_stats;
    // read-compare-write of the code rows (ProjOut::q_rmw).  A member and not the launch argument: the float32 max kernels set it to
    // false after construction -- their stores then compile to what they were before (with the test and the prefetched old words in
    // place k_project_lin lost 9 % in situ at the Walabot grid, session r4be) --, and inside an `if (!em.rmw)` block the test folds away
    bool rmw;
    // Codes-only launches of the wave-per-frame kernels: a frame's xz and xy codes are finished plane by plane -- a 176-byte and a
    // 31-byte store per 21.8 KB plane at the Walabot grid, from each of 2 048 waves -- and those small writes, scattered in time
    // between the reads, cost the streaming kernels far more than their bytes (stand-alone, Walabot grid: 0.70 of 8 TB/s with them,
    // 0.73 without, 0.81 with no output at all; the LDS work, the VALU work and the number of resident waves change nothing:
    // tools/exp/README.md, round 4).  With `stage` set they wait in a wave-private LDS row [xz | xy] and leave as one burst of
    // 16-byte stores at the frame's end (flush_wave).
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u32x4 lds_u128;
    lds_u8* stage = nullptr;    // an LDS pointer, and a separate flag: a null test on a generic pointer into LDS does not compile here
    int staged = 0;
    int stage_xy = 0;           // byte offset of the xy codes in the stage
    __device__ __forceinline__ void set_stage(unsigned char* p, int xy_off) { stage = (lds_u8*)p; staged = 1; stage_xy = xy_off; }

    __device__ Emitter(const ProjParams& a_, int64_t b_) : a(a_), b(b_) {
        want_stats = a.o.row_isum || a.o.row_isq || a.o.row_flags || a.o.q[0] || a.o.q[1] || a.o.q[2];
        rmw = a.o.q_rmw != 0;
    }
    __device__ __forceinline__ float scaled(float v) const {
        // true IEEE division: bit-identical to NumPy's float32 "x / 255." (common.py:148)
        return (a.o.scale_div > 1.0f) ? __fdiv_rn(v, a.o.scale_div) : v;
    }
    __device__ __forceinline__ uint32_t code(float v) {
        int c = (int)v;
        bool good = ((float)c == v) && c >= 0 && c <= 255;
        ok &= good ? 1 : 0;
        c = good ? c : 0;
        isum += c;
        isq += (uint32_t)(c * c);
        return (uint32_t)(c ^ 0x80);
    }
    // already-decided code (rml_quantize_rows)
    __device__ __forceinline__ void put_code1(int pl, int64_t idx, int c, bool good) {
        ok &= good ? 1 : 0;
        c = good ? c : 0;
        isum += c;
        isq += (uint32_t)(c * c);
        if (a.o.q[pl]) a.o.q[pl][b * a.o.qstride + idx] = (uint8_t)(c ^ 0x80);
    }
    __device__ __forceinline__ void put1(int pl, int64_t idx, float v) {
        if (!((a.o.sel >> pl) & 1u)) return;
        if (a.o.p[pl]) a.o.p[pl][b * a.o.stride[pl] + idx] = scaled(v);
        if (a.o.row_nsq) { double t = (double)scaled(v); nsq += t * t; }
        if (want_stats) {
            uint32_t c = code(v);
            if (a.o.q[pl]) {
                if (staged && pl != 1) stage[(pl == 2 ? stage_xy : 0) + idx] = (uint8_t)c;
                else a.o.q[pl][b * a.o.qstride + idx] = (uint8_t)c;
            }
        }
    }
    // Read-compare-write of the code rows (ProjOut::q_rmw): the dword a later put4 / put_bytes4 of (pl, idx) would overwrite, so that
    // a kernel can have the old words of a whole region in flight before it compares the first one
    __device__ __forceinline__ uint32_t old_word(int pl, int64_t idx) const {
        if (!(rmw && a.o.q[pl] && ((a.o.sel >> pl) & 1u))) return 0u;
        return *reinterpret_cast<const uint32_t*>(a.o.q[pl] + qb() * a.o.qstride + idx);
    }
    // the same without the tests (a launch with rmw set has the code rows; the caller checked the plane): an unconditional load
    __device__ __forceinline__ uint32_t old_word_nocheck(int pl, int64_t idx) const {
        return *reinterpret_cast<const uint32_t*>(a.o.q[pl] + qb() * a.o.qstride + idx);
    }
    // four projection values that ARE bytes (uint8 volumes; idx a multiple of 4): when only codes and their statistics are
    // wanted the bytes are the codes -- biased with one xor, summed and squared with v_dot4_u32_u8 -- and nothing is widened
    __device__ __forceinline__ void put_bytes4(int pl, int64_t idx, uint32_t w, bool have_old = false, uint32_t old = 0u) {
        if (!((a.o.sel >> pl) & 1u)) return;
        if (a.o.p[pl] || a.o.row_nsq) { put4(pl, idx, bytes_to_float4(w), have_old, old); return; }
        if (want_stats) {
            isum += (int32_t)__builtin_amdgcn_udot4(w, 0x01010101u, 0u, false);
            isq = __builtin_amdgcn_udot4(w, w, isq, false);
            if (a.o.q[pl]) {
                uint32_t* dq = reinterpret_cast<uint32_t*>(a.o.q[pl] + qb() * a.o.qstride + idx);
                const uint32_t nv = w ^ 0x80808080u;
                if (!rmw || (have_old ? old : *dq) != nv) *dq = nv;
            }
        }
    }
    // idx is a multiple of 4
    __device__ __forceinline__ void put4(int pl, int64_t idx, float4 v, bool have_old = false, uint32_t old = 0u) {
        if (!((a.o.sel >> pl) & 1u)) return;
        if (a.o.p[pl]) {
            float* dst = a.o.p[pl] + b * a.o.stride[pl] + idx;
            float4 s = make_float4(scaled(v.x), scaled(v.y), scaled(v.z), scaled(v.w));
            if (a.o.row_nsq) {
                nsq += (double)s.x * (double)s.x + (double)s.y * (double)s.y;
                nsq += (double)s.z * (double)s.z + (double)s.w * (double)s.w;
            }
            if (a.vec_ok[pl]) {
                *reinterpret_cast<float4*>(dst) = s;
            } else {
                // rows that are only 4-byte aligned (D = 10 010 floats at the Walabot grid): still ONE 16-byte store -- a global
                // store needs dword alignment only, and four dword stores at a 16-byte lane stride cost 4x the instructions
                typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
                *reinterpret_cast<f32x4u*>(dst) = f32x4u{s.x, s.y, s.z, s.w};
            }
        }
        if (want_stats) {
            uint32_t c0 = code(v.x), c1 = code(v.y), c2 = code(v.z), c3 = code(v.w);
            if (a.o.q[pl]) {
                uint32_t packed = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
                if (staged && pl != 1) *(lds_u32*)(stage + (pl == 2 ? stage_xy : 0) + idx) = packed;
                else {
                    uint32_t* dq = reinterpret_cast<uint32_t*>(a.o.q[pl] + qb() * a.o.qstride + idx);
                    if (!rmw || (have_old ? old : *dq) != packed) *dq = packed;
                }
            }
        }
    }
    // next frame of a persistent kernel
    __device__ __forceinline__ void reset(int64_t b_) { b = b_; isum = 0; isq = 0; ok = 1; nsq = 0.0; }
    // the staged xz / xy codes of frame b -> the code row, 16 bytes per lane and store where the row allows it (one wave: the DS unit
    // runs its instructions in order; the fences only keep the compiler from moving an access across the hand-over)
    __device__ void flush_wave(int lane) {
        if (!staged) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int len[2] = {a.X * a.Z, a.X * a.Y};
        lds_u8* src[2] = {stage, stage + stage_xy};
        uint8_t* dst[2] = {a.o.q[0] + qb() * a.o.qstride, a.o.q[2] + qb() * a.o.qstride};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int done = 0;
            if ((reinterpret_cast<uintptr_t>(dst[r]) & 15) == 0) {
                for (int i = lane; i < (len[r] >> 4); i += 64) {
                    reinterpret_cast<u32x4*>(dst[r])[i] = ((lds_u128*)src[r])[i];
                }
                done = len[r] & ~15;
            }
            for (int i = done + 4 * lane; i + 4 <= len[r]; i += 256)       // rows are 4-byte aligned (rml_project's contract)
                *reinterpret_cast<uint32_t*>(dst[r] + i) = *(lds_u32*)(src[r] + i);
            for (int i = (len[r] & ~3) + lane; i < len[r]; i += 64) dst[r][i] = src[r][i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // the same as finish() for kernels in which ONE WAVE owns the frame: no LDS, no barrier
    __device__ void finish_wave(int lane) {
        if (a.o.qrow) {
            for (int64_t c = a.o.qD + lane; c < a.o.qstride; c += 64) a.o.qrow[b * a.o.qstride + c] = 0;
        }
        if (a.o.prow) {
            for (int64_t c = a.o.pD + lane; c < a.o.pstride; c += 64) a.o.prow[b * a.o.pstride + c] = 0.0f;
        }
        if (!(a.o.row_isum || a.o.row_isq || a.o.row_flags || a.o.row_nsq)) return;
        int64_t s = isum, q = isq;
        int g = ok;
        double nn = nsq;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s += __shfl_xor(s, off);
            q += __shfl_xor(q, off);
            g &= __shfl_xor(g, off);
            nn += __shfl_xor(nn, off);
        }
        if (lane == 0) {
            if (a.o.row_isum) a.o.row_isum[b] = (int32_t)s;
            if (a.o.row_isq) a.o.row_isq[b] = q;
            if (a.o.row_flags) a.o.row_flags[b] = (int32_t)g;
            if (a.o.row_nsq) a.o.row_nsq[b] = nn;
        }
    }
    // block reduction of the statistics (up to 16 waves); red must hold >= 64 int64 slots; all threads call it
    __device__ void finish(int64_t* red) {
        if (a.o.qrow) {   // zero the pad columns [qD, qstride) of the code row (i8 value 0)
            for (int64_t c = a.o.qD + threadIdx.x; c < a.o.qstride; c += blockDim.x) a.o.qrow[b * a.o.qstride + c] = 0;
        }
        if (a.o.prow) {   // zero the pad columns [pD, pstride) of the float row
            for (int64_t c = a.o.pD + threadIdx.x; c < a.o.pstride; c += blockDim.x) a.o.prow[b * a.o.pstride + c] = 0.0f;
        }
        if (!(a.o.row_isum || a.o.row_isq || a.o.row_flags || a.o.row_nsq)) return;
        int64_t s = isum, q = isq;
        int g = ok;
        double nn = nsq;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            s += __shfl_xor(s, off);
            q += __shfl_xor(q, off);
            g &= __shfl_xor(g, off);
            nn += __shfl_xor(nn, off);
        }
        int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double* redd = reinterpret_cast<double*>(red + 48);
        __syncthreads();
        if (lane == 0) { red[wave * 3 + 0] = s; red[wave * 3 + 1] = q; red[wave * 3 + 2] = g; redd[wave] = nn; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int64_t S = 0, Q = 0, G = 1;
            double NN = 0.0;
            for (int w = 0; w < (int)(blockDim.x + 63) / 64; ++w) { S += red[w * 3]; Q += red[w * 3 + 1]; G &= red[w * 3 + 2]; NN += redd[w]; }
            if (a.o.row_isum) a.o.row_isum[b] = (int32_t)S;
            if (a.o.row_isq) a.o.row_isq[b] = Q;
            if (a.o.row_flags) a.o.row_flags[b] = (int32_t)G;
            if (a.o.row_nsq) a.o.row_nsq[b] = NN;
        }
    }
};

// streaming load: every volume byte is read exactly once, so it is marked non-temporal (keeps the
// support-vector tiles of the concurrently running SVM GEMM resident in L2 / Infinity Cache)
__device__ __forceinline__ float4 ld_stream(const float4* p) {
    typedef float v4f_t __attribute__((ext_vector_type(4)));
    v4f_t v = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

// uint8 volumes (the radar's native 0..255 magnitudes, 4x fewer HBM bytes): a lane's quad is one dword,
// widened with v_cvt_f32_ubyte0..3; everything downstream is the float path, so the results are identical
__device__ __forceinline__ float4 ld_stream(const uint32_t* p) {
    const uint32_t w = __builtin_nontemporal_load(p);
    return make_float4((float)(w & 0xFFu), (float)((w >> 8) & 0xFFu), (float)((w >> 16) & 0xFFu), (float)(w >> 24));
}
template <typename VT> struct Quad;
template <> struct Quad<float> { typedef float4 T; };
template <> struct Quad<uint8_t> { typedef uint32_t T; };

// butterfly reduction over the LPR lanes of a row (all lanes end with the result)
template <int MODE, int LPR> __device__ __forceinline__ float row_reduce(float r) {
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) r = Op<MODE>::f(r, __shfl_xor(r, off));
    return r;
}

template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ float dpp_mov(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, BOUND));
}
// reduction over the 64 lanes of a wave; the result is wave-uniform (an SGPR)
template <int MODE> __device__ __forceinline__ float wave_reduce_uniform(float r) {
    const float id = Op<MODE>::ident();
    r = Op<MODE>::f(r, dpp_mov<0xB1, 0xF, true>(id, r));    // quad_perm [1,0,3,2]
    r = Op<MODE>::f(r, dpp_mov<0x4E, 0xF, true>(id, r));    // quad_perm [2,3,0,1]
    r = Op<MODE>::f(r, dpp_mov<0x141, 0xF, true>(id, r));   // row_half_mirror
    r = Op<MODE>::f(r, dpp_mov<0x140, 0xF, true>(id, r));   // row_mirror: every lane of a 16-lane row holds the row's result
    r = Op<MODE>::f(r, dpp_mov<0x142, 0xA, false>(id, r));  // row_bcast15 into rows 1 and 3
    r = Op<MODE>::f(r, dpp_mov<0x143, 0xC, false>(id, r));  // row_bcast31 into rows 2 and 3: row 3 holds the total
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), 63));
}

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) -- register arrays indexed by the
// constant stay in registers (a runtime-indexed array would go to scratch)
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
// acc[lane J] = uniform value (v_writelane_b32 with a constant lane select: one instruction, no compare mask)
template <int J> __device__ __forceinline__ float park_lane(float acc, float uniform_val) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(uniform_val), "n"(J));
    return acc;
}

// max as ONE instruction on a loaded value: hipcc puts a v_max_f32 x,x "canonicalise" copy in front of every fmaxf / maximum of a
// loaded value (IEEE mode: it would quiet a signalling NaN).  v_maximum3_f32 d, a, b, b is the IEEE-754-2019 maximum itself: a NaN
// operand -- quiet or signalling -- is the result (np.max).  4 VALU less per row of the streaming loop.
template <int MODE> __device__ __forceinline__ float op_raw(float a, float b) {
    if constexpr (MODE == RML_MODE_MAX) {
        float r;
        asm("v_maximum3_f32 %0, %1, %2, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    } else
    {
        return Op<MODE>::f(a, b);
    }
}
template <int MODE> __device__ __forceinline__ float4 op4_raw(float4 a, float4 b) {
    return make_float4(op_raw<MODE>(a.x, b.x), op_raw<MODE>(a.y, b.y), op_raw<MODE>(a.z, b.z), op_raw<MODE>(a.w, b.w));
}

// bytes of per-wave LDS code stage for a launch of a wave-per-frame kernel (0: not a codes-only launch, or RML_OPT_STAGE_CODES = 0)
inline int code_stage_bytes(const ProjParams& pp, size_t voxel_bytes) {
    const ProjOut& o = pp.o;
    if (voxel_bytes != 4 || o.p[0] || o.p[1] || o.p[2] || o.row_nsq || !o.q[0] || !o.q[2] || o.skip_if_set) return 0;
    if (!pp.k_stage_codes) return 0;
    const int bytes = ((pp.X * pp.Z + 15) & ~15) + ((pp.X * pp.Y + 15) & ~15);
    return bytes <= 16 * 1024 ? bytes : 0;
}

// project_lin.hip: the linear-plane wave-per-frame kernel for rows that do not fill a load instruction (32 < Z/4 < 64); returns
// true when it took the launch
bool try_launch_lin(const ProjParams& pp, int mode, int num_cu, hipStream_t st);

// project_slice.hip: mode SLICE with (i,j,k) given (wave per output row), and the fused derive -> slice pass (persistent, wave per
// frame; pp.ntgt / ijk_out / profiles set by the caller); true when they took the launch
bool try_launch_slice(const ProjParams& pp, int vbytes, hipStream_t st);
bool try_launch_derive_slice(const ProjParams& pp, int vbytes, int num_cu, hipStream_t st);
bool derive_slice_shape_ok(int X, int Y, int Z, int ntgt);        // the shape alone (RML_OPT_DERIVE_FUSED is the caller's)

// project_u8.hip: the byte-native max-projection of uint8 volumes (rows of whole 16-byte chunks); true when it took the launch
bool try_launch_u8_max(const ProjParams& pp, hipStream_t st);

}  // namespace rmlproj
