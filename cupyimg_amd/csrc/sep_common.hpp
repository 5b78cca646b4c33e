// sep_common.hpp -- helpers shared by the fused / streaming separable kernels.
#pragma once
#include "common.hpp"

namespace mi {

constexpr int kMaxTaps = 9;

// remembers (per host thread) which kernel a separable-filter call dispatched: mi_debug_last_kernel (bench.py names
// the kernel it timed from this, not from a literal)
void note_kernel(const char *fmt, ...);

// Dry run (per host thread): separable3d_impl and run_sep3d_long walk their whole decision tree and return MI_OK where
// they would launch, MI_ERR_UNSUPPORTED / MI_ERR_INVALID_ARG where they would refuse -- nothing is queued.  A request
// that names plane ranges is treated as a partial one even when the ranges happen to cover the array, so that the
// answer does not depend on which rank of a slab chain asks (mi_separable3d_f32_supports, mi_slab_separable3d_f32).
extern thread_local bool t_dry_run;
struct DryRun {
    bool saved;
    DryRun() : saved(t_dry_run) { t_dry_run = true; }
    ~DryRun() { t_dry_run = saved; }
};

struct Sep3dParams {
    int nx, ny, nz;
    int wy;                 // taps along y (run-time loop)
    int oy, oz;             // w/2 + origin for y and z (x offset is WX/2)
    int mx, my, mz;         // boundary modes (filter_mode()-normalised)
    float cval;
    int ty;                 // output rows per tile
    int zc;                 // output planes per chunk
    int nxt, nyt, nzc;      // tile counts
    int tw;                 // tile width in floats (<= 256, multiple of 4): the row is split into nxt EQUAL tiles
    // output planes to produce: up to two plane ranges [zb, zb + zn), the first
    // covered by chunks 0 .. nzc0-1, the second by the rest (whole volume:
    // zb0 = 0, zn0 = nz, nzc0 = nzc).  Boundary handling always refers to nz.
    int zb0, zn0, zb1, zn1, nzc0;
    float wx[kMaxTaps], wyv[kMaxTaps], wz[kMaxTaps];
    int dbg;                // tuning ablations (0 in production): 1 no x/z math, 2 no stores, 4 no loads, 8 no y math
    int zrev;               // lean kernel: odd z chunks stream DOWNWARDS (see sep3d_lean_kernel)
};

__device__ __forceinline__ void chunk_planes(const Sep3dParams &p, int zci, int *zs, int *ze)
{
    const bool second = zci >= p.nzc0;
    const int zb = second ? p.zb1 : p.zb0, zn = second ? p.zn1 : p.zn0;
    const int c = second ? zci - p.nzc0 : zci;
    *zs = zb + c * p.zc;
    *ze = min(*zs + p.zc, zb + zn);
}

struct __attribute__((packed, aligned(4))) float4u { float x, y, z, w; };

enum { EDGE_FWD = 0, EDGE_REV = 1, EDGE_SPLAT = 2, EDGE_CONST = 3 };

// where the 4 floats left of x0 (side 0) / right of xe (side 1) come from
__device__ __forceinline__ void edge_desc(int side, int x0, int xe, int nx, int mode, int *start, int *kind)
{
    if (side == 0) {
        if (x0 > 0) { *start = x0 - 4; *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = 0; *kind = EDGE_REV; break;          // x[-k] = x[k-1]
        case MI_MODE_MIRROR:    *start = 1; *kind = EDGE_REV; break;          // x[-k] = x[k]
        case MI_MODE_NEAREST:   *start = 0; *kind = EDGE_SPLAT; break;
        case MI_MODE_GRID_WRAP: *start = nx - 4; *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    } else {
        if (xe < nx) { *start = xe; *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = nx - 4; *kind = EDGE_REV; break;     // x[n-1+k] = x[n-k]
        case MI_MODE_MIRROR:    *start = nx - 5; *kind = EDGE_REV; break;     // x[n-1+k] = x[n-1-k]
        case MI_MODE_NEAREST:   *start = nx - 1; *kind = EDGE_SPLAT; break;
        case MI_MODE_GRID_WRAP: *start = 0; *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    }
}

__device__ __forceinline__ float comp(const float4 &v, int k)
{
    return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
}

__device__ __forceinline__ float dpp_from_left(float keep_for_lane0, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep_for_lane0), __float_as_int(v),
                                                      0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_right(float keep_for_lane63, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(keep_for_lane63), __float_as_int(v),
                                                      0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

// t[idx] for a per-lane idx, as selects on VALUES.  (Taking the array by
// reference lets instcombine turn select-of-loads into a load from a selected
// address; after inlining that is a dynamically indexed stack array, and the
// whole prefetch register set ends up in scratch memory.)
__device__ __forceinline__ float pick2(float t0, float t1, int idx) { return idx ? t1 : t0; }
__device__ __forceinline__ float pick4(float t0, float t1, float t2, float t3, int idx)
{
    const float a = idx & 1 ? t1 : t0, b = idx & 1 ? t3 : t2;
    return idx & 2 ? b : a;
}
template <int NE>
__device__ __forceinline__ float pick(const float (&t)[NE], int idx)
{
    if constexpr (NE == 2) return pick2(t[0], t[1], idx);
    else return pick4(t[0], t[1], t[2], t[3], idx);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kOOB = 0x80000000u;   // >= num_records of any descriptor we build

template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: two floats per lane per
// instruction, i.e. twice the FMA rate of the scalar forms on CDNA3/4).  A
// float4 of x-consecutive voxels is kept as two aligned pairs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct F4 { f32x2 lo, hi; };

__device__ __forceinline__ f32x2 splat2(float w) { return (f32x2){w, w}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ F4 f4_splat(float v) { F4 r; r.lo = splat2(v); r.hi = splat2(v); return r; }
__device__ __forceinline__ F4 f4_from(u32x4 u)
{
    F4 r;
    r.lo = (f32x2){__uint_as_float(u.x), __uint_as_float(u.y)};
    r.hi = (f32x2){__uint_as_float(u.z), __uint_as_float(u.w)};
    return r;
}
__device__ __forceinline__ F4 f4_from(float4 v) { F4 r; r.lo = (f32x2){v.x, v.y}; r.hi = (f32x2){v.z, v.w}; return r; }
__device__ __forceinline__ float4 f4_to_float4(F4 a) { return make_float4(a.lo.x, a.lo.y, a.hi.x, a.hi.y); }
__device__ __forceinline__ u32x4 f4_to_u32(F4 a)
{
    u32x4 u;
    u.x = __float_as_uint(a.lo.x); u.y = __float_as_uint(a.lo.y);
    u.z = __float_as_uint(a.hi.x); u.w = __float_as_uint(a.hi.y);
    return u;
}
__device__ __forceinline__ F4 f4_scale(float w, F4 a) { F4 r; r.lo = splat2(w) * a.lo; r.hi = splat2(w) * a.hi; return r; }
__device__ __forceinline__ F4 f4_fma(float w, F4 q, F4 a)
{
    F4 r;
    r.lo = fma2(splat2(w), q.lo, a.lo);
    r.hi = fma2(splat2(w), q.hi, a.hi);
    return r;
}

// the same with the weight given as an (w, w) pair, e.g. an aligned SGPR pair loaded from a table of doubled weights
__device__ __forceinline__ F4 f4_scale2(f32x2 w, F4 a) { F4 r; r.lo = w * a.lo; r.hi = w * a.hi; return r; }
__device__ __forceinline__ F4 f4_fma2(f32x2 w, F4 q, F4 a)
{
    F4 r;
    r.lo = fma2(w, q.lo, a.lo);
    r.hi = fma2(w, q.hi, a.hi);
    return r;
}

// x pass over a register window in "dot" form: the window is held as aligned
// pairs A[m] = (win[2m], win[2m+1]) and output c = sum_j wx[j] * win[BASE+c+j]
// is accumulated as a 2-vector sum_m (wx[2m-BASE-c], wx[2m+1-BASE-c]) * A[m]
// whose halves are added at the end.  Same v_pk_fma count as the shifted-copy
// form (xpass_packed) but no second copy of the window in registers.
template <int WX, int NP, int BASE>
__device__ __forceinline__ F4 xdot(const f32x2 (&A)[NP], const float *__restrict__ wx)
{
    float o[4];
    static_for<4>([&](auto CC) {
        constexpr int c = decltype(CC)::value;
        constexpr int t0 = BASE + c, t1 = BASE + c + WX - 1;
        constexpr int m0 = t0 / 2, m1 = t1 / 2;
        static_assert(m1 < NP, "window too short");
        f32x2 acc;
        static_for<m1 - m0 + 1>([&](auto MM) {
            constexpr int m = m0 + decltype(MM)::value;
            constexpr int j0 = 2 * m - t0, j1 = j0 + 1;
            constexpr bool in0 = j0 >= 0 && j0 < WX, in1 = j1 >= 0 && j1 < WX;
            const f32x2 wp = (f32x2){in0 ? wx[in0 ? j0 : 0] : 0.f, in1 ? wx[in1 ? j1 : 0] : 0.f};
            // a sample outside the taps is not multiplied at all (0 x inf would be a NaN): see xdot_tab
            if constexpr (m == m0 && !in0) acc = (f32x2){0.f, wp.y * A[m].y};
            else if constexpr (m == m0) acc = wp * A[m];
            else if constexpr (!in1) acc.x = __builtin_fmaf(wp.x, A[m].x, acc.x);
            else acc = fma2(wp, A[m], acc);
        });
        o[c] = acc.x + acc.y;
    });
    F4 r;
    r.lo = (f32x2){o[0], o[1]};
    r.hi = (f32x2){o[2], o[3]};
    return r;
}

// Weights read from the kernel-argument segment through a scalar pointer.
// `launder` makes the pointer opaque once per loop iteration so that the
// s_loads stay inside the loop: hoisted, 2 x 17 weights (+ their pairs) exceed
// the SGPR file and come back as v_readlane spill code, more VALU work than
// the filter itself.
typedef const __attribute__((address_space(4))) float *kfloats;
__device__ __forceinline__ kfloats kernarg_floats(int byte_offset)
{
    return (kfloats)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset);
}
__device__ __forceinline__ void launder(kfloats &p) { asm volatile("" : "+s"(p)); }

// xdot with the weight pairs precomputed by the host: tab[q] + 2 (m - m0) holds
// the pair for output parity q = c & 1 and window pair m (m0 = (BASE + c) / 2).
template <int WX, int NP, int BASE>
__device__ __forceinline__ F4 xdot_tab(const f32x2 (&A)[NP], kfloats tab0, kfloats tab1)
{
    float o[4];
    static_for<4>([&](auto CC) {
        constexpr int c = decltype(CC)::value;
        constexpr int t0 = BASE + c, t1 = BASE + c + WX - 1;
        constexpr int m0 = t0 / 2, m1 = t1 / 2;
        static_assert(m1 < NP, "window too short");
        kfloats tab = (c & 1) ? tab1 : tab0;
        f32x2 acc;
        // The first / last pair of the window holds a sample OUTSIDE the output's taps when the window starts at an odd / ends
        // at an even index: its table entry is a zero, and 0 x inf (or NaN) would leak a NaN one voxel beyond the taps, where
        // the reference (an explicit sum over the taps) stays finite.  Those two half-used pairs take a scalar multiply /
        // FMA on the half that counts -- the same number of instructions (r4b; found with non-finite samples in the volume).
        constexpr bool first_half = (t0 & 1) != 0, last_half = (t1 & 1) == 0;
        static_for<m1 - m0 + 1>([&](auto MM) {
            constexpr int u = decltype(MM)::value;
            const f32x2 wp = (f32x2){tab[2 * u], tab[2 * u + 1]};
            if constexpr (u == 0 && first_half) acc = (f32x2){0.f, wp.y * A[m0].y};
            else if constexpr (u == 0) acc = wp * A[m0];
            else if constexpr (u == m1 - m0 && last_half) acc.x = __builtin_fmaf(wp.x, A[m0 + u].x, acc.x);
            else acc = fma2(wp, A[m0 + u], acc);
        });
        o[c] = acc.x + acc.y;
    });
    F4 r;
    r.lo = (f32x2){o[0], o[1]};
    r.hi = (f32x2){o[2], o[3]};
    return r;
}

// buffer_store_dwordx4 with a REGISTER soffset, followed by two wait states.
// A VMEM store of more than 64 bits reads its data VGPRs late; a VALU write to one
// of them in the next instruction slots can overtake that read.  LLVM (ROCm 7.2)
// inserts the wait states for the soffset-less form only (GCNHazardRecognizer:
// "no hazard if the instruction uses a register in the soffset field", which is
// what the ISA manual says), but on gfx950 the soffset form is exposed as well as
// soon as two waves of the kernel share a SIMD: the register overwritten right
// after the store arrived in memory with the NEW value in lanes 12-15 of every
// 16-lane group (root cause of the round-1 streaming-pass failure; reproducer:
// scripts/diag/stream_diag.hip, one `s_nop 0` after the store is enough there).
// The fake "v"(d) input keeps every later writer of the data registers behind
// the s_nop.
__device__ __forceinline__ void buffer_store_b128_soff(u32x4 d, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 0);
    asm volatile("s_nop 1" ::"v"(d) : "memory");
}

__device__ __forceinline__ float4 as_f4(u32x4 u)
{
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}


}  // namespace mi
