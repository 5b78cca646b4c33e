// interp_common.hpp -- geometry, coordinate folding and the float32 cubic tap machinery shared by interp.hip and
// cubic_fast.hip (moved out of interp.hip in round 5 so that new kernels compile on their own: interp.hip takes minutes).
#pragma once
#include "common.hpp"

namespace mi {

struct InterpGeom {
    int64_t shape[MI_MAX_NDIM];    // input, padded
    int64_t stride[MI_MAX_NDIM];   // input, elements
    int64_t oshape[MI_MAX_NDIM];   // output, padded (affine)
    double mat[MI_MAX_NDIM * (MI_MAX_NDIM + 1)];   // affine, padded (ND x (ND+1))
    int pad;                        // number of leading unit axes
};

__device__ __forceinline__ double wrap_coord(double c, int64_t n)
{
    if (n <= 1) return 0.0;
    const double s = (double)(n - 1);
    if (c < 0) c += s * ((double)(int64_t)(-c / s) + 1.0);
    else if (c > s) c -= s * (double)(int64_t)(c / s);
    return c;
}

// SciPy's map_coordinate(): fold a float coordinate into the array
__device__ __forceinline__ double fold_coord(double c, int64_t n, int mode)
{
    if (n <= 1) return 0.0;
    const double dn = (double)n;
    switch (mode) {
    case MI_MODE_MIRROR: {
        const double p = 2.0 * dn - 2.0;
        if (c < 0) { c = p * (double)(int64_t)(-c / p) + c; c = c <= 1.0 - dn ? c + p : -c; }
        else if (c > dn - 1.0) { c -= p * (double)(int64_t)(c / p); if (c >= dn) c = p - c; }
        return c;
    }
    case MI_MODE_REFLECT: {
        const double p = 2.0 * dn;
        if (c < 0) {
            if (c < -p) c = p * (double)(int64_t)(-c / p) + c;
            c = c < -dn ? c + p : (c > -1e-15 ? 1e-15 : -c) - 1.0;
        } else if (c > dn - 1.0) {
            c -= p * (double)(int64_t)(c / p);
            if (c >= dn) c = p - c - 1.0;
        }
        return c;
    }
    case MI_MODE_WRAP:
        return wrap_coord(c, n);
    case MI_MODE_GRID_WRAP:
        if (c < 0) c += dn * ((double)(int64_t)((-1.0 - c) / dn) + 1.0);
        else if (c > dn - 1.0) c -= dn * (double)(int64_t)((c + 1.0) / dn);
        return c;
    case MI_MODE_NEAREST:
        return c < 0 ? 0.0 : (c > dn - 1.0 ? dn - 1.0 : c);
    default:
        return c;
    }
}


// ---------------------------------------------------------------------------
// Cubic interpolation on float32 coefficients (float32 in / out, the reference's
// `allow_float32` route, interpolation.py:330-335): same tap selection as
// spline_point_t<3> (double coordinate arithmetic), float weights and float
// accumulation, and the four x taps of a (z, y) pair fetched with one 16-byte gather
// whenever they are consecutive in memory -- 16 gathers per voxel instead of 64.
// ---------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kOobOffset = 0x80000000u;   // beyond any buffer: the load returns 0 without touching memory

struct Cubic3 {
    float w[3][4];
    int off[3][4];      // element offset along the axis, -1: the tap reads cval
    int ntap[2];        // taps on z and y (1 on rank-padding axes)
    bool outside;       // constant mode, coordinate beyond the array: the voxel is cval
};

// 32-bit tap index outside [0, n): the symmetry the coefficients were computed with (spline_tap in int)
__device__ __forceinline__ int spline_tap32(int i, int n, int mode)
{
    if (i >= 0 && i < n) return i;
    if (mode == MI_MODE_GRID_CONSTANT) return -1;
    if (mode == MI_MODE_REFLECT) return bmap<int>(i, n, MI_MODE_REFLECT);
    if (mode == MI_MODE_NEAREST) return i < 0 ? 0 : n - 1;
    if (mode == MI_MODE_GRID_WRAP) return bmap<int>(i, n, MI_MODE_GRID_WRAP);
    return bmap<int>(i, n, MI_MODE_MIRROR);
}

// one axis of the tap selection: coordinate in double (as the double route), everything after the
// integer / fraction split in 32 bits.  Returns true when the coordinate is beyond the array in constant mode.
// the four weights of a fraction x in [0, 1)
__device__ __forceinline__ void cubic3_weights(float x, float (&w)[4])
{
    const float y = 1.f - x;
    w[1] = (x * x * (x - 2.f) * 3.f + 4.f) * (1.f / 6.f);
    w[2] = (y * y * (y - 2.f) * 3.f + 4.f) * (1.f / 6.f);
    w[0] = y * y * y * (1.f / 6.f);
    w[3] = 1.f - w[0] - w[1] - w[2];
}

// taps of an axis and the fraction their weights are made of (cubic3_weights)
__device__ __forceinline__ bool cubic3_axis_frac(int n, int stride, double cc, int mode, int npad, float &frac, int (&off)[4])
{
    bool outside = false;
    cc += (double)npad;
    if (mode == MI_MODE_CONSTANT) {
        if (cc < 0 || cc > (double)(n - 1)) { outside = true; cc = 0.0; }
    } else if (mode != MI_MODE_GRID_CONSTANT && mode != MI_MODE_NEAREST) {
        cc = fold_coord(cc, n, mode);
    } else {
        // taps are mapped one by one below; keep the integer part inside 32 bits for far-away coordinates
        cc = cc < -1.0e9 ? -1.0e9 : (cc > 1.0e9 ? 1.0e9 : cc);
    }
    const double fl = floor(cc);
    const int start = (int)fl - 1;
    frac = (float)(cc - fl);
    if (start >= 0 && start + 3 < n) {
#pragma unroll
        for (int k = 0; k < 4; k++) off[k] = (start + k) * stride;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = spline_tap32(start + k, n, mode);
            off[k] = j < 0 ? -1 : j * stride;
        }
    }
    return outside;
}

__device__ __forceinline__ bool cubic3_axis(int n, int stride, double cc, int mode, int npad, float (&w)[4], int (&off)[4])
{
    float x;
    const bool outside = cubic3_axis_frac(n, stride, cc, mode, npad, x, off);
    cubic3_weights(x, w);
    return outside;
}

__device__ __forceinline__ void cubic3_setup(const InterpGeom &g, const double (&c)[3], int mode, int npad, Cubic3 &t)
{
    t.outside = false;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        if (d < g.pad) {
            if (d < 2) t.ntap[d] = 1;
#pragma unroll
            for (int k = 0; k < 4; k++) { t.w[d][k] = 1.f; t.off[d][k] = 0; }
            continue;
        }
        if (d < 2) t.ntap[d] = 4;
        t.outside |= cubic3_axis((int)g.shape[d], (int)g.stride[d], c[d], mode, npad, t.w[d], t.off[d]);
    }
}


// CVTAPS: taps may read cval (grid-constant only); otherwise no tap offset is ever negative
template <int NTZ, int NTY, bool CVTAPS>
__device__ __forceinline__ float cubic3_gather_t(const __amdgpu_buffer_rsrc_t rin, const Cubic3 &t, float cval)
{
    const bool consec = t.off[2][0] >= 0 && t.off[2][3] == t.off[2][0] + 3;
    float v[NTZ][NTY][4];
    if (consec) {
        u32x4 q[NTZ][NTY];
#pragma unroll
        for (int kz = 0; kz < NTZ; kz++)
#pragma unroll
            for (int ky = 0; ky < NTY; ky++) {
                const bool oob_zy = CVTAPS && (t.off[0][kz] < 0 || t.off[1][ky] < 0);
                const int base = oob_zy ? 0 : t.off[0][kz] + t.off[1][ky];
                q[kz][ky] = __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)(base + t.off[2][0]) * 4u, 0, 0);
            }
#pragma unroll
        for (int kz = 0; kz < NTZ; kz++)
#pragma unroll
            for (int ky = 0; ky < NTY; ky++) {
                const bool oob_zy = CVTAPS && (t.off[0][kz] < 0 || t.off[1][ky] < 0);
                v[kz][ky][0] = oob_zy ? cval : __uint_as_float(q[kz][ky].x);
                v[kz][ky][1] = oob_zy ? cval : __uint_as_float(q[kz][ky].y);
                v[kz][ky][2] = oob_zy ? cval : __uint_as_float(q[kz][ky].z);
                v[kz][ky][3] = oob_zy ? cval : __uint_as_float(q[kz][ky].w);
            }
    } else {
#pragma unroll
        for (int kz = 0; kz < NTZ; kz++)
#pragma unroll
            for (int ky = 0; ky < NTY; ky++) {
                const bool oob_zy = CVTAPS && (t.off[0][kz] < 0 || t.off[1][ky] < 0);
                const int base = oob_zy ? 0 : t.off[0][kz] + t.off[1][ky];
#pragma unroll
                for (int kx = 0; kx < 4; kx++) {
                    const bool oob = CVTAPS && (oob_zy || t.off[2][kx] < 0);
                    const float q = __uint_as_float(
                        __builtin_amdgcn_raw_buffer_load_b32(rin, oob ? 0u : (unsigned)(base + t.off[2][kx]) * 4u, 0, 0));
                    v[kz][ky][kx] = oob ? cval : q;
                }
            }
    }
    float acc = 0.f;
#pragma unroll
    for (int kz = 0; kz < NTZ; kz++)
#pragma unroll
        for (int ky = 0; ky < NTY; ky++) {
            const float wzy = t.w[0][kz] * t.w[1][ky];
            float row = v[kz][ky][0] * t.w[2][0];
            row = fmaf(v[kz][ky][1], t.w[2][1], row);
            row = fmaf(v[kz][ky][2], t.w[2][2], row);
            row = fmaf(v[kz][ky][3], t.w[2][3], row);
            acc = fmaf(row, wzy, acc);
        }
    return t.outside ? cval : acc;
}

template <int NTZ, int NTY>
__device__ __forceinline__ float cubic3_gather(const __amdgpu_buffer_rsrc_t rin, const Cubic3 &t, float cval, int mode)
{
    if (mode == MI_MODE_GRID_CONSTANT) return cubic3_gather_t<NTZ, NTY, true>(rin, t, cval);
    return cubic3_gather_t<NTZ, NTY, false>(rin, t, cval);
}


// Diagonal transforms (zoom, shift): taps and weights of an axis depend on the output index along that axis only, so they are
// tabulated once per call (cubic3_axis_table_kernel, interp.hip).  In the separable passes `off` holds plain indices along the
// axis, -1 = the tap reads cval.
struct AxisTaps { float w[4]; int off[4]; int outside; int pad_[3]; };

void note_kernel(const char *fmt, ...);        // runtime.hip (sep_common.hpp declares it for the filter sources)
bool spline_pass_fast(const mi_array *shape, const void *src, int src_dtype, void *dst, int axis, int order, int spline_mode, hipStream_t s, int *rc);   // spline_fast.hip
constexpr int kCzP = 80, kCzRoundsMax = 8, kCzSlots = 5, kCzTY = 32, kCzNT = 256;
constexpr double kCzMinXStep = 0.09;    // |dx_in/dx_out| >= this x |dy_in/dx_out|: up to ~85 degrees (profiles/r4_cubic_zstream.txt: 3.4 ms against 4.1 ms there, 7.8 against 3.9 at 90)
constexpr int kCzSlot = 14336;          // the fixed slot size (44 rows): five of them + the tiles fit a CU twice, 4 x kCzSlot is an immediate offset

struct CubZParams {
    // names as for SAX = 0 (the stream axis is z, a plane's rows run along y); for SAX = 1 `z` is the array's axis 1 and `y`
    // its axis 0 -- lengths, strides (in elements) and matrix entries are filled in accordingly by the launch
    int nz, ny, nx;              // coefficient array (padded by npad on every axis): stream axis, row axis, x
    int oz, oy, ox;
    int ss, sr;                  // input strides of the stream axis and of the row axis
    int oss, osr;                // the same of the output
    int vol_bytes;
    double m00, m03, m11, m12, m13, m21, m22, m23;
    double cmin_y, cmin_x;       // minimum of cy / cx over a tile relative to its first voxel
    int ry, nchunks;             // rows of the staged rectangle, 16-byte chunks per plane (ry * 20)
    int slot_bytes;              // kCzSlot when the rectangle fits it, else exactly the rectangle
    int zc, nzc, ntx, nty;
    int mode, npad;
    float cval;
    int dbg;
    int sident;                  // r5 (cubic3_zfactor_kernel): the stream axis holds SAMPLES and maps onto itself with an integral shift: one plane per step
};

// lanes outside `mask` neither fetch nor write LDS (the last round of a plane: the slots are exactly as long as the rectangle)
__device__ __forceinline__ void cz_dma16(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_base, unsigned long long mask)
{
    unsigned keep;
    unsigned long long keep_exec;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b64 %1, exec\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_mov_b64 exec, %6\n\t"
        "buffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
        "s_mov_b64 exec, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(keep_exec)
        : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base), "s"(mask)
        : "memory");
}

struct CzPlane { float w[4]; int off[4]; int pl[4]; bool outside, cvtap; };      // cvtap: the step takes cubic3_gather (a cval tap along z, or a slot clash)

}  // namespace mi
