// sep_common.hpp -- helpers shared by the fused / streaming separable kernels.
#pragma once
#include "common.hpp"

namespace mi {

constexpr int kMaxTaps = 9;

struct Sep3dParams {
    int nx, ny, nz;
    int wy;                 // taps along y (run-time loop)
    int oy, oz;             // w/2 + origin for y and z (x offset is WX/2)
    int mx, my, mz;         // boundary modes (filter_mode()-normalised)
    float cval;
    int ty;                 // output rows per tile
    int zc;                 // output planes per chunk
    int nxt, nyt, nzc;      // tile counts
    // output planes to produce: up to two plane ranges [zb, zb + zn), the first
    // covered by chunks 0 .. nzc0-1, the second by the rest (whole volume:
    // zb0 = 0, zn0 = nz, nzc0 = nzc).  Boundary handling always refers to nz.
    int zb0, zn0, zb1, zn1, nzc0;
    float wx[kMaxTaps], wyv[kMaxTaps], wz[kMaxTaps];
    int dbg;                // tuning ablations (0 in production): 1 no x/z math, 2 no stores, 4 no loads, 8 no y math
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

template <int NE>
__device__ __forceinline__ float pick(const float (&t)[NE], int idx)
{
    if constexpr (NE == 2) return idx ? t[1] : t[0];
    else return idx & 2 ? (idx & 1 ? t[3] : t[2]) : (idx & 1 ? t[1] : t[0]);
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

__device__ __forceinline__ float4 as_f4(u32x4 u)
{
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}


}  // namespace mi
