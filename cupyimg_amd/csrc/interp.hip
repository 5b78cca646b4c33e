// interp.hip -- spline order 0 / 1 interpolation: map_coordinates and
// affine_transform (K5).
//
// Reference: cupyimg/scipy/ndimage/interpolation.py:271-394 (launch :393) and
// :397-561 (launch :545,560); kernel body _interp_kernels.py:277-592
// (coordinate producers :17-47 and :198-242).
//
// Arithmetic follows SciPy, the reference's test oracle: coordinates and
// weights in double, accumulation in double, integer outputs rounded half
// away from zero and clipped (SciPy's rule; the reference uses rint()).  Order 1 uses 2^ndim taps and skips the upper tap on an axis whose
// coordinate is integral (_interp_kernels.py:409-471); 'constant' cuts off
// hard outside [0, n-1] (:340-353) while 'grid-constant' blends with cval;
// 'wrap' folds the float coordinate with period n-1.  Order 0 folds the float
// coordinate first and rounds half up afterwards like SciPy (the reference
// rounds first with lrint and excludes the tie case from its own tests,
// tests/test_interpolation.py:362-364).
//
// Geometry is padded with leading unit axes to a compile-time rank (3 or 8).
#include "interp_common.hpp"

namespace mi {

template <typename T, int ND>
__device__ __forceinline__ double interp_point(const T *__restrict__ in, const InterpGeom &g,
                                               const double (&c)[ND], int order, int mode, double cval)
{
    if (mode == MI_MODE_CONSTANT) {
        bool outside = false;
#pragma unroll
        for (int d = 0; d < ND; d++) outside |= (c[d] < 0 || c[d] > (double)(g.shape[d] - 1));
        if (outside) return cval;
    }
    if (order == 0) {
        int64_t pos = 0;
        bool oob = false;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            int64_t j;
            if (mode == MI_MODE_CONSTANT) j = (int64_t)floor(c[d] + 0.5);
            else if (mode == MI_MODE_GRID_CONSTANT) j = bmap<int64_t>((int64_t)floor(c[d] + 0.5), g.shape[d], mode);
            else j = bmap<int64_t>((int64_t)floor(fold_coord(c[d], g.shape[d], mode) + 0.5), g.shape[d], mode);
            oob |= j < 0;
            pos += j * g.stride[d];
        }
        return oob ? cval : (double)in[pos];
    }
    // order 1 with SciPy 1.15's arithmetic (ni_interpolation.c / ni_splines.c), so that integer outputs agree at
    // exact ties: fold the coordinate first, weights 1 - x and 1 - (1 - x) from the folded coordinate's fraction,
    // every sample multiplied by its weights one axis at a time before it is added
    int64_t lo[ND], hi[ND];
    double wlo[ND], whi[ND];
    bool two[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) {
        double cc = c[d];
        if (mode != MI_MODE_CONSTANT && mode != MI_MODE_GRID_CONSTANT && mode != MI_MODE_NEAREST)
            cc = fold_coord(cc, g.shape[d], mode);
        const double cf = floor(cc);
        two[d] = cc != cf;
        wlo[d] = 1.0 - (cc - cf);
        whi[d] = 1.0 - wlo[d];
        lo[d] = (int64_t)cf;
        hi[d] = lo[d] + 1;
        if (mode != MI_MODE_CONSTANT) {
            const int tm = mode == MI_MODE_WRAP ? MI_MODE_MIRROR : mode;
            lo[d] = bmap<int64_t>(lo[d], g.shape[d], tm);
            hi[d] = bmap<int64_t>(hi[d], g.shape[d], tm);
        }
    }
    double acc = 0.0;
    // enumerate corners in the oracle's order: axis 0 is the most significant bit
    for (int m = 0; m < (1 << ND); m++) {
        int64_t pos = 0;
        bool skip = false, oob = false;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            const bool up = (m >> (ND - 1 - d)) & 1;
            skip |= up && !two[d];
            const int64_t j = up ? hi[d] : lo[d];
            oob |= j < 0;
            pos += j * g.stride[d];
        }
        if (skip) continue;
        double coeff = oob ? cval : (double)in[oob ? 0 : pos];
#pragma unroll
        for (int d = 0; d < ND; d++) coeff *= ((m >> (ND - 1 - d)) & 1) ? whi[d] : wlo[d];
        acc += coeff;
    }
    return acc;
}

// ---------------------------------------------------------------------------
// B-spline orders 2..5 (reference: _spline_prefilter_core.py:14-139 poles and
// boundary initialisation, _spline_kernel_weights.py weights,
// _interp_kernels.py:473-549 tap loop; arithmetic as in SciPy 1.15.3, the
// parity target: see oracle/ndimage_oracle.c).  `in` holds the prefiltered
// float64 coefficients, padded by npad samples on every real axis for the
// modes SciPy pads (nearest, grid-constant).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void spline_weights(double x, int order, double *w)
{
    double y;
    switch (order) {
    case 2:
        w[1] = 0.75 - x * x; y = 0.5 - x; w[0] = 0.5 * y * y; w[2] = 1.0 - w[0] - w[1];
        break;
    case 3:
        y = 1.0 - x;
        w[1] = (x * x * (x - 2.0) * 3.0 + 4.0) / 6.0;
        w[2] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
        w[0] = y * y * y / 6.0;
        w[3] = 1.0 - w[0] - w[1] - w[2];
        break;
    case 4:
        y = x * x;
        w[2] = y * (y * 0.25 - 0.625) + 115.0 / 192.0;
        y = 1.0 + x;
        w[1] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        y = 1.0 - x;
        w[3] = y * (y * (y * (5.0 - y) / 6.0 - 1.25) + 5.0 / 24.0) + 55.0 / 96.0;
        y = 0.5 - x; y = y * y;
        w[0] = y * y / 24.0;
        w[4] = 1.0 - w[0] - w[1] - w[2] - w[3];
        break;
    default:
        y = x * x;
        w[2] = y * (y * (0.25 - x / 12.0) - 0.5) + 0.55;
        y = 1.0 - x; y = y * y;
        w[3] = y * (y * (0.25 - (1.0 - x) / 12.0) - 0.5) + 0.55;
        y = x + 1.0;
        w[1] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        y = 2.0 - x;
        w[4] = y * (y * (y * (y * (y / 24.0 - 0.375) + 1.25) - 1.75) + 0.625) + 0.425;
        y = 1.0 - x; y = y * y;
        w[0] = (1.0 - x) * y * y / 120.0;
        w[5] = 1.0 - w[0] - w[1] - w[2] - w[3] - w[4];
        break;
    }
}

// tap index outside [0, n): the symmetry the coefficients were computed with
__device__ __forceinline__ int64_t spline_tap(int64_t i, int64_t n, int mode)
{
    if (i >= 0 && i < n) return i;
    if (mode == MI_MODE_GRID_CONSTANT) return -1;
    if (mode == MI_MODE_REFLECT) return bmap<int64_t>(i, n, MI_MODE_REFLECT);
    if (mode == MI_MODE_NEAREST) return i < 0 ? 0 : n - 1;       // taps are clamped, the coordinate is not
    if (mode == MI_MODE_GRID_WRAP) return bmap<int64_t>(i, n, MI_MODE_GRID_WRAP);
    return bmap<int64_t>(i, n, MI_MODE_MIRROR);
}

template <typename T, int ND>
__device__ __forceinline__ double spline_point(const T *__restrict__ in, const InterpGeom &g, const double (&c)[ND],
                                               int order, int mode, double cval, int npad)
{
    double w[ND][6];
    int64_t idx[ND][6];
#pragma unroll
    for (int d = 0; d < ND; d++) {
        const int64_t n = g.shape[d];
        if (d < g.pad) {                         // unit axis added by the rank padding: a single tap
            for (int k = 0; k <= order; k++) { w[d][k] = k == 0 ? 1.0 : 0.0; idx[d][k] = 0; }
            continue;
        }
        double cc = c[d] + (double)npad;
        if (mode == MI_MODE_CONSTANT) {
            if (cc < 0 || cc > (double)(n - 1)) return cval;
        } else if (mode != MI_MODE_GRID_CONSTANT && mode != MI_MODE_NEAREST) {
            cc = fold_coord(cc, n, mode);
        }
        const double fl = (order & 1) ? floor(cc) : floor(cc + 0.5);
        const int64_t start = (int64_t)fl - order / 2;
        spline_weights(cc - fl, order, w[d]);
        for (int k = 0; k <= order; k++) idx[d][k] = spline_tap(start + k, n, mode);
    }
    // taps in the oracle's order (last axis fastest)
    int k[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) k[d] = 0;
    double acc = 0.0;
    for (;;) {
        int64_t pos = 0;
        bool oob = false;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            oob |= idx[d][k[d]] < 0;
            pos += idx[d][k[d]] * g.stride[d];
        }
        // SciPy multiplies the sample by its weights one axis at a time (matters at exact ties of integer outputs)
        double coeff = oob ? cval : (double)in[oob ? 0 : pos];
#pragma unroll
        for (int d = 0; d < ND; d++) coeff *= w[d][k[d]];
        acc += coeff;
        int d = ND - 1;
        while (d >= 0) {
            if (d >= g.pad && ++k[d] <= order) break;
            k[d] = 0;
            d--;
        }
        if (d < 0) break;
    }
    return acc;
}

template <typename T, typename C, int ND>
__global__ void __launch_bounds__(256)
map_coordinates_kernel(const T *__restrict__ in, const C *__restrict__ coords, void *__restrict__ out,
                       int out_dt, InterpGeom g, int64_t nout, int order, int mode, double cval,
                       int round_out, int npad)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout;
         i += (int64_t)gridDim.x * blockDim.x) {
        double c[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) c[d] = d < g.pad ? 0.0 : (double)coords[(int64_t)(d - g.pad) * nout + i];
        double v;
        if constexpr (ND == 3) v = order > 1 ? spline_point<T, ND>(in, g, c, order, mode, cval, npad)
                                             : interp_point<T, ND>(in, g, c, order, mode, cval);
        else v = interp_point<T, ND>(in, g, c, order, mode, cval);
        if (round_out) v = interp_round(v, out_dt);
        store_as(out, i, out_dt, v);
    }
}

template <typename T, int ND>
__global__ void __launch_bounds__(256)
affine_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, InterpGeom g, int64_t nout,
              int order, int mode, double cval, int round_out, int npad)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout;
         i += (int64_t)gridDim.x * blockDim.x) {
        double o[ND], c[ND];
        int64_t r = i;
#pragma unroll
        for (int d = ND - 1; d >= 0; d--) {
            const int64_t q = r / g.oshape[d];
            o[d] = (double)(r - q * g.oshape[d]);
            r = q;
        }
#pragma unroll
        for (int d = 0; d < ND; d++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < ND; k++) s += g.mat[d * (ND + 1) + k] * o[k];
            c[d] = s + g.mat[d * (ND + 1) + ND];
        }
        double v;
        if constexpr (ND == 3) v = order > 1 ? spline_point<T, ND>(in, g, c, order, mode, cval, npad)
                                             : interp_point<T, ND>(in, g, c, order, mode, cval);
        else v = interp_point<T, ND>(in, g, c, order, mode, cval);
        if (round_out) v = interp_round(v, out_dt);
        store_as(out, i, out_dt, v);
    }
}

// Orders 2..5 with the order as a template parameter: weights and tap indices
// stay in registers (every index is static after unrolling) -- the run-time
// order version above keeps them in scratch memory.  Same arithmetic, same tap
// order (z, y, x; weight product (wz * wy) * wx).
template <int ORDER>
__device__ __forceinline__ double spline_point_t(const double *__restrict__ in, const InterpGeom &g, const double (&c)[3],
                                                 int mode, double cval, int npad)
{
    constexpr int NT = ORDER + 1;
    double w[3][NT];
    int64_t off[3][NT];            // element offset of the tap along its axis, or -1: the tap reads cval
    int ntap[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const int64_t n = g.shape[d];
        if (d < g.pad) {
            ntap[d] = 1;
#pragma unroll
            for (int k = 0; k < NT; k++) { w[d][k] = 1.0; off[d][k] = 0; }
            continue;
        }
        ntap[d] = NT;
        double cc = c[d] + (double)npad;
        if (mode == MI_MODE_CONSTANT) {
            if (cc < 0 || cc > (double)(n - 1)) return cval;
        } else if (mode != MI_MODE_GRID_CONSTANT && mode != MI_MODE_NEAREST) {
            cc = fold_coord(cc, n, mode);
        }
        const double fl = (ORDER & 1) ? floor(cc) : floor(cc + 0.5);
        const int64_t start = (int64_t)fl - ORDER / 2;
        spline_weights(cc - fl, ORDER, w[d]);
        const bool interior = start >= 0 && start + ORDER < n;
#pragma unroll
        for (int k = 0; k < NT; k++) {
            const int64_t j = interior ? start + k : spline_tap(start + k, n, mode);
            off[d][k] = j < 0 ? -1 : j * g.stride[d];
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int kz = 0; kz < NT; kz++) {
        if (kz >= ntap[0]) break;
#pragma unroll
        for (int ky = 0; ky < NT; ky++) {
            if (ky >= ntap[1]) break;
            const bool oob_zy = off[0][kz] < 0 || off[1][ky] < 0;
            const int64_t base = off[0][kz] + off[1][ky];
#pragma unroll
            for (int kx = 0; kx < NT; kx++) {
                const bool oob = oob_zy || off[2][kx] < 0;
                const double v = oob ? cval : in[base + off[2][kx]];
                // the sample times its weights one axis at a time, as SciPy does (exact ties of integer outputs)
                acc += ((v * w[0][kz]) * w[1][ky]) * w[2][kx];
            }
        }
    }
    return acc;
}

template <typename C, int ORDER>
__global__ void __launch_bounds__(256)
spline_map_kernel(const double *__restrict__ in, const C *__restrict__ coords, void *__restrict__ out, int out_dt,
                  InterpGeom g, int64_t nout, int mode, double cval, int round_out, int npad)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (int64_t)gridDim.x * blockDim.x) {
        double c[3];
#pragma unroll
        for (int d = 0; d < 3; d++) c[d] = d < g.pad ? 0.0 : (double)coords[(int64_t)(d - g.pad) * nout + i];
        double v = spline_point_t<ORDER>(in, g, c, mode, cval, npad);
        if (round_out) v = interp_round(v, out_dt);
        store_as(out, i, out_dt, v);
    }
}

template <int ORDER>
__global__ void __launch_bounds__(256)
spline_affine_kernel(const double *__restrict__ in, void *__restrict__ out, int out_dt, InterpGeom g, int64_t nout,
                     int mode, double cval, int round_out, int npad)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (int64_t)gridDim.x * blockDim.x) {
        double o[3], c[3];
        int64_t r = i;
#pragma unroll
        for (int d = 2; d >= 0; d--) {
            const int64_t q = r / g.oshape[d];
            o[d] = (double)(r - q * g.oshape[d]);
            r = q;
        }
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += g.mat[d * 4 + k] * o[k];
            c[d] = s + g.mat[d * 4 + 3];
        }
        double v = spline_point_t<ORDER>(in, g, c, mode, cval, npad);
        if (round_out) v = interp_round(v, out_dt);
        store_as(out, i, out_dt, v);
    }
}


// Orders 2..5 on arrays of rank 4 .. 8 (r3; the reference's generator is rank-generic, _interp_kernels.py:473-549):
// the run-time-order tap loop of spline_point over eight (padded) axes, float64 coefficients, (order + 1)^rank taps
// per output sample in the oracle's order.  A correctness path: no tiling, weights and indices in scratch memory.
template <typename C>
__global__ void __launch_bounds__(256)
spline_map_nd_kernel(const double *__restrict__ in, const C *__restrict__ coords, void *__restrict__ out, int out_dt,
                     InterpGeom g, int64_t nout, int order, int mode, double cval, int round_out, int npad)
{
    constexpr int ND = MI_MAX_NDIM;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (int64_t)gridDim.x * blockDim.x) {
        double c[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) c[d] = d < g.pad ? 0.0 : (double)coords[(int64_t)(d - g.pad) * nout + i];
        double v = spline_point<double, ND>(in, g, c, order, mode, cval, npad);
        if (round_out) v = interp_round(v, out_dt);
        store_as(out, i, out_dt, v);
    }
}

__global__ void __launch_bounds__(256)
spline_affine_nd_kernel(const double *__restrict__ in, void *__restrict__ out, int out_dt, InterpGeom g, int64_t nout,
                        int order, int mode, double cval, int round_out, int npad)
{
    constexpr int ND = MI_MAX_NDIM;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (int64_t)gridDim.x * blockDim.x) {
        double o[ND], c[ND];
        int64_t r = i;
#pragma unroll
        for (int d = ND - 1; d >= 0; d--) {
            const int64_t q = r / g.oshape[d];
            o[d] = (double)(r - q * g.oshape[d]);
            r = q;
        }
#pragma unroll
        for (int d = 0; d < ND; d++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < ND; k++) s += g.mat[d * (ND + 1) + k] * o[k];
            c[d] = s + g.mat[d * (ND + 1) + ND];
        }
        double v = spline_point<double, ND>(in, g, c, order, mode, cval, npad);
        if (round_out) v = interp_round(v, out_dt);
        store_as(out, i, out_dt, v);
    }
}

// Diagonal transforms: taps and weights of an axis depend on the output index along that axis only, so
// they are tabulated once per call (oz + oy + ox entries) instead of once per voxel.
// (struct AxisTaps: interp_common.hpp)

__global__ void __launch_bounds__(256)
cubic3_axis_table_kernel(AxisTaps *__restrict__ tab, InterpGeom g, int mode, int npad, int unit_stride)
{
    const int d = blockIdx.y;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (int)g.oshape[d]) return;
    int base = 0;
    for (int k = 0; k < d; k++) base += (int)g.oshape[k];
    AxisTaps e;
    e.pad_[0] = e.pad_[1] = e.pad_[2] = 0;
    if (d < g.pad) {
        e.outside = 0;
        for (int k = 0; k < 4; k++) { e.w[k] = 1.f; e.off[k] = 0; }
    } else {
        const double c = g.mat[d * 4 + d] * (double)o + g.mat[d * 4 + 3];
        e.outside = cubic3_axis((int)g.shape[d], unit_stride ? 1 : (int)g.stride[d], c, mode, npad, e.w, e.off) ? 1 : 0;
    }
    tab[base + o] = e;
}

// Diagonal transforms as separable 1-D resampling passes (x, then y, then z): every pass reads four taps
// per output sample -- rows (y, z passes: wave-uniform taps, coalesced loads) or neighbours within a row
// (x pass) -- so the traffic is one read and one write of each intermediate instead of 16 rows per voxel.
// Tap indices in `tab` are plain indices along the axis (unit stride), -1 = the tap reads cval.
// The last pass writes cval where any axis' coordinate is beyond the array (constant mode).
//   AXIS 2: in (n0, n1, nin) -> out (n0, n1, nout);  AXIS 1: in (n0, nin, n2) -> out (n0, nout, n2);
//   AXIS 0: in (nin, n1, n2) -> out (nout, n1, n2).   Grid over the pass's output: (x tiles, y / 4, z).
template <int AXIS>
__global__ void __launch_bounds__(256)
cubic_resample_axis_kernel(const float *__restrict__ in, float *__restrict__ out, const AxisTaps *__restrict__ tab,
                           int d0, int d1, int d2, int nin, float cval, const AxisTaps *__restrict__ all_tabs, int oz, int oy,
                           int last)
{
    // (d0, d1, d2): output shape of this pass
    const int x = blockIdx.x * 64 + threadIdx.x, z = blockIdx.z;
    const int y = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    if (y >= d1 || x >= d2) return;
    float w[4];
    int off[4];
    size_t base, step;
    if constexpr (AXIS == 2) {
        const AxisTaps e = tab[x];
#pragma unroll
        for (int k = 0; k < 4; k++) { w[k] = e.w[k]; off[k] = e.off[k]; }
        base = ((size_t)z * d1 + y) * (size_t)nin;
        step = 1;
    } else if constexpr (AXIS == 1) {
        const AxisTaps e = tab[y];
#pragma unroll
        for (int k = 0; k < 4; k++) { w[k] = e.w[k]; off[k] = e.off[k]; }
        base = (size_t)z * nin * (size_t)d2 + x;
        step = (size_t)d2;
    } else {
        const AxisTaps e = tab[z];
#pragma unroll
        for (int k = 0; k < 4; k++) { w[k] = e.w[k]; off[k] = e.off[k]; }
        base = (size_t)y * d2 + x;
        step = (size_t)d1 * d2;
    }
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = in[base + (size_t)(off[k] < 0 ? 0 : off[k]) * step];
    float r = (off[0] < 0 ? cval : v[0]) * w[0];
#pragma unroll
    for (int k = 1; k < 4; k++) r = fmaf(off[k] < 0 ? cval : v[k], w[k], r);
    if (last) {
        const int outside = all_tabs[z].outside | all_tabs[oz + y].outside | all_tabs[oz + oy + x].outside;
        if (outside) r = cval;
    }
    out[((size_t)z * d1 + y) * (size_t)d2 + x] = r;
}

// y / z passes on four x-consecutive samples per thread (16-byte loads and stores; d2 % 4 == 0)
template <int AXIS>
__global__ void __launch_bounds__(256)
cubic_resample_rows4_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, const AxisTaps *__restrict__ tab,
                            int d0, int d1, int d2q, int nin, float cval, const AxisTaps *__restrict__ all_tabs, int oz, int oy,
                            int last)
{
    const int xq = blockIdx.x * 64 + threadIdx.x, z = blockIdx.z;
    const int y = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);
    if (y >= d1 || xq >= d2q) return;
    const AxisTaps e = tab[AXIS == 1 ? y : z];
    size_t base, step;
    if constexpr (AXIS == 1) { base = (size_t)z * nin * (size_t)d2q + xq; step = (size_t)d2q; }
    else { base = (size_t)y * d2q + xq; step = (size_t)d1 * d2q; }
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = in[base + (size_t)(e.off[k] < 0 ? 0 : e.off[k]) * step];
    float4 r;
    {
        const bool c0 = e.off[0] < 0;
        r.x = (c0 ? cval : v[0].x) * e.w[0]; r.y = (c0 ? cval : v[0].y) * e.w[0];
        r.z = (c0 ? cval : v[0].z) * e.w[0]; r.w = (c0 ? cval : v[0].w) * e.w[0];
    }
#pragma unroll
    for (int k = 1; k < 4; k++) {
        const bool c = e.off[k] < 0;
        r.x = fmaf(c ? cval : v[k].x, e.w[k], r.x); r.y = fmaf(c ? cval : v[k].y, e.w[k], r.y);
        r.z = fmaf(c ? cval : v[k].z, e.w[k], r.z); r.w = fmaf(c ? cval : v[k].w, e.w[k], r.w);
    }
    if (last) {
        const int ozy = all_tabs[z].outside | all_tabs[oz + y].outside;
        const AxisTaps *tx = all_tabs + oz + oy + 4 * xq;
        if (ozy | tx[0].outside) r.x = cval;
        if (ozy | tx[1].outside) r.y = cval;
        if (ozy | tx[2].outside) r.z = cval;
        if (ozy | tx[3].outside) r.w = cval;
    }
    out[((size_t)z * d1 + y) * (size_t)d2q + xq] = r;
}

// block (64, 4): 64 lanes along the output x axis, so that the gathers of a wave touch neighbouring
// coefficients; grid (x tiles, y tiles, z) -- no index divisions
template <typename C, bool AFFINE, int NTZ, int NTY>
__global__ void __launch_bounds__(256)
cubic3_f32_kernel(const float *__restrict__ in, const C *__restrict__ coords, float *__restrict__ out, InterpGeom g,
                  int64_t nout, int64_t nin, int mode, float cval, int npad)
{
    const int ox = (int)g.oshape[2], oy = (int)g.oshape[1];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y, z = blockIdx.z;
    if (x >= ox || y >= oy) return;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(nin * 4), 0x00020000);
    const int64_t i = ((int64_t)z * oy + y) * ox + x;
    double c[3];
    if constexpr (AFFINE) {
        const double o[3] = {(double)z, (double)y, (double)x};
#pragma unroll
        for (int d = 0; d < 3; d++) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; k++) s += g.mat[d * 4 + k] * o[k];
            c[d] = s + g.mat[d * 4 + 3];
        }
    } else {
#pragma unroll
        for (int d = 0; d < 3; d++)
            c[d] = d < g.pad ? 0.0 : (double)__builtin_nontemporal_load(coords + (int64_t)(d - g.pad) * nout + i);
    }
    Cubic3 t;
    cubic3_setup(g, c, mode, npad, t);
    __builtin_nontemporal_store((cubic3_gather<NTZ, NTY>(rin, t, cval, mode)), out + i);
}

// Diagonal transforms (zoom, shift; the reference's zoom/shift kernel, _interp_kernels.py:655-688): the 64
// voxels of a wave share their z and y taps and read one contiguous run of coefficients per (z, y) row.
// The 16 rows are blended first (coalesced loads, wave-uniform weights) into a strip of <= 128 values in
// LDS, then every lane takes its four x taps from the strip.  Waves whose x taps are not consecutive runs
// (boundary folds) or span more than 128 coefficients use the gather path.
__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
}
__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}

template <int NTZ, int NTY>
__global__ void __launch_bounds__(256)
cubic3_diag_f32_kernel(const float *__restrict__ in, float *__restrict__ out, const AxisTaps *__restrict__ tab, InterpGeom g,
                       int64_t nin, int mode, float cval)
{
    __shared__ float strips[4][128];
    const int ox = (int)g.oshape[2], oy = (int)g.oshape[1], oz = (int)g.oshape[0];
    const int x = blockIdx.x * 64 + threadIdx.x, z = blockIdx.z;
    const int y = __builtin_amdgcn_readfirstlane(blockIdx.y * 4 + threadIdx.y);     // one row per wave
    if (y >= oy) return;
    const bool live = x < ox;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(nin * 4), 0x00020000);
    const AxisTaps ez = tab[z], ey = tab[oz + y];                   // wave-uniform: scalar loads
    const AxisTaps ex = tab[oz + oy + (live ? x : ox - 1)];
    Cubic3 t;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        t.w[0][k] = ez.w[k]; t.off[0][k] = ez.off[k];
        t.w[1][k] = ey.w[k]; t.off[1][k] = ey.off[k];
        t.w[2][k] = ex.w[k]; t.off[2][k] = ex.off[k];
    }
    t.ntap[0] = NTZ; t.ntap[1] = NTY;
    t.outside = (ez.outside | ey.outside | ex.outside) != 0;
    // lanes beyond the array in constant mode produce cval whatever they read: they do not constrain the strip
    const bool consec = t.outside || (t.off[2][0] >= 0 && t.off[2][3] == t.off[2][0] + 3);
    const int x0 = wave_min((consec && !t.outside) ? t.off[2][0] : 0x7fffffff);
    const int x1 = wave_max((consec && !t.outside) ? t.off[2][0] : -1);
    const bool strip_ok = __all(consec) && x1 >= x0 && x1 - x0 + 4 <= 128;
    float res;
    if (strip_ok) {
        const int lane = threadIdx.x;
        const bool second = x1 - x0 + 4 > 64;
        float s0 = 0.f, s1 = 0.f;
        float a0[NTZ][NTY], a1[NTZ][NTY];
#pragma unroll
        for (int kz = 0; kz < NTZ; kz++)
#pragma unroll
            for (int ky = 0; ky < NTY; ky++) {
                const bool oob_zy = t.off[0][kz] < 0 || t.off[1][ky] < 0;
                const unsigned b = oob_zy ? 0u : (unsigned)(t.off[0][kz] + t.off[1][ky] + x0 + lane) * 4u;
                a0[kz][ky] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rin, b, 0, 0));
                a1[kz][ky] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rin, second ? b + 256u : kOobOffset, 0, 0));
            }
#pragma unroll
        for (int kz = 0; kz < NTZ; kz++)
#pragma unroll
            for (int ky = 0; ky < NTY; ky++) {
                const float wzy = t.w[0][kz] * t.w[1][ky];
                const bool oob_zy = t.off[0][kz] < 0 || t.off[1][ky] < 0;
                s0 = fmaf(oob_zy ? cval : a0[kz][ky], wzy, s0);
                s1 = fmaf(oob_zy ? cval : a1[kz][ky], wzy, s1);
            }
        float *strip = strips[threadIdx.y];
        strip[lane] = s0;
        strip[lane + 64] = s1;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): the strip is written by this wave only
        const int j = t.outside ? 0 : t.off[2][0] - x0;
        float r = strip[j] * t.w[2][0];
        r = fmaf(strip[j + 1], t.w[2][1], r);
        r = fmaf(strip[j + 2], t.w[2][2], r);
        r = fmaf(strip[j + 3], t.w[2][3], r);
        res = t.outside ? cval : r;
    } else {
        res = cubic3_gather<NTZ, NTY>(rin, t, cval, mode);
    }
    if (live) __builtin_nontemporal_store(res, out + ((int64_t)z * oy + y) * ox + x);
}

// ---------------------------------------------------------------------------
// r4b: cubic interpolation (float32 coefficients) for matrices that leave axis 0 to itself -- in-plane rotations, shears,
// scalings of every (y, x) slice with a step of at most one plane through the slices: `rotate(volume, a, axes=(1, 2))`,
// slice-wise registration -- STREAMING ALONG z like the order-1 kernels of interp_fast.hip.  cubic3_f32_kernel gathers
// 16 x 16 bytes per voxel through the L1 (3.5 ms on 512^3); here a workgroup (256 threads, 32 x 64 tile, eight voxels per
// thread) walks down a chunk of output planes, the in-plane taps of a voxel -- the LDS offset of its 4 x 4 block and its
// eight weights -- are computed ONCE, input planes enter LDS once per tile as the tile's bounding rectangle (LDS-DMA, a ring
// of five slots indexed by plane mod 5: four planes of the current output plane + the one the next needs), and a voxel
// reads its 64 taps as 32 ds_read2_b32.  Tap selection (cubic3_axis), weights, products and the order of the sums are
// those of cubic3_gather: bit-identical results.  A voxel whose 4 x 4 block is not a plain block inside the rectangle
// (boundary folds, cval taps, anything the rectangle does not hold) and a plane with a cval tap along z take
// cubic3_gather itself.
// ---------------------------------------------------------------------------
template <int SAX>
__global__ void __launch_bounds__(kCzNT, 2)
cubic3_zstream_kernel(const float *__restrict__ in, float *__restrict__ out, const CubZParams q)
{
    constexpr int P = kCzP, TY = kCzTY, NT = kCzNT, NS = kCzSlots;
    extern __shared__ __attribute__((aligned(16))) char smem_cz[];
    const unsigned slot_bytes = (unsigned)q.slot_bytes;                                 // the last DMA round of a plane is lane-masked: nothing beyond the rectangle is written
    float *tiles = reinterpret_cast<float *>(smem_cz + max((unsigned)NS * slot_bytes, 4u * 24u * 64u * 4u));      // [4 waves][8 rows][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = q.ntx * q.nty * q.nzc;
    int t = blockIdx.x;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);          // x-neighbouring tiles on one XCD
    const int tx_i = t % q.ntx, ty_i = (t / q.ntx) % q.nty, zc_i = t / (q.ntx * q.nty);
    const int x0 = tx_i * 64, y0 = ty_i * TY;
    const int zs = zc_i * q.zc, ze = min(zs + q.zc, q.oz);
    const int vol_bytes = q.vol_bytes;
    const unsigned plane_b = (unsigned)q.ss * 4u, row_b = (unsigned)q.sr * 4u;

    // ---- rectangle origin: the first tap row / column of the tile's smallest coordinates (a hair below, one tap to the
    // left, + the padding of the coefficient array), clamped into the array, x aligned down to 16 bytes
    int by0, bx0;
    {
        const double cy = ((q.m11 * (double)y0 + q.m12 * (double)x0) + q.m13) + q.cmin_y + (double)q.npad;
        const double cx = ((q.m21 * (double)y0 + q.m22 * (double)x0) + q.m23) + q.cmin_x + (double)q.npad;
        double fy = floor(cy - 1e-6 * (1.0 + fabs(cy))) - 1.0, fx = floor(cx - 1e-6 * (1.0 + fabs(cx))) - 1.0;
        fy = fy < 0.0 ? 0.0 : (fy > (double)(q.ny - 1) ? (double)(q.ny - 1) : fy);
        fx = fx < 0.0 ? 0.0 : (fx > (double)(q.nx - 1) ? (double)(q.nx - 1) : fx);
        by0 = __builtin_amdgcn_readfirstlane((int)fy);
        bx0 = __builtin_amdgcn_readfirstlane((int)fx & ~3);
    }
    const int rounds = (q.nchunks + NT - 1) / NT;
    unsigned rel[kCzRoundsMax];
    unsigned long long live[kCzRoundsMax];
#pragma unroll
    for (int j = 0; j < kCzRoundsMax; j++) {
        const unsigned ch = (unsigned)tid + (unsigned)(j * NT);
        const unsigned row = ch / 20u, c4 = ch - row * 20u;
        rel[j] = ch < (unsigned)q.nchunks ? row * row_b + c4 * 16u : 0x80000000u;
        live[j] = __builtin_amdgcn_ballot_w64(ch < (unsigned)q.nchunks);
    }
    const unsigned org_b = ((unsigned)by0 * (unsigned)q.sr + (unsigned)bx0) * 4u;

    // ---- per voxel (row y0 + 8 wave + k, column x0 + lane), once: the in-plane taps.  a_[k] = row and column of the
    // start the four taps per axis are counted from, relative to the rectangle, + 8 each, row in the high half.  A voxel is
    //   plain   when the 4 x 4 taps are a block inside the rectangle (or the voxel is `outside`: its value is cval);
    //   edge    when taps beyond the array come back by the cheap form of the boundary rule (czrule below: reflection about
    //           the end / clamping, what spline_tap32 gives within a few samples of the array for every mode but the two
    //           grid modes) and all sixteen are inside the rectangle -- CHECKED here against cubic3_axis's offsets;
    //   far     otherwise (cval taps of grid-constant, the other side of the array under grid-wrap, ...).
    // A wave with a far voxel takes cubic3_gather for its eight rows; every other wave reads the rectangle, rows and
    // columns through czrule.
    const int rule = q.mode == MI_MODE_NEAREST ? 2 : (q.mode == MI_MODE_REFLECT ? 1 : ((q.mode == MI_MODE_GRID_WRAP || q.mode == MI_MODE_GRID_CONSTANT) ? 3 : 0));
    auto czrule = [&](int i, int n) {               // rule 3: never used for taps beyond the array (those voxels are far)
        const int lo = rule == 2 ? 0 : -i - rule, hi = rule == 2 ? n - 1 : 2 * n - 2 + rule - i;
        return i < 0 ? lo : (i >= n ? hi : i);
    };
    int a_[8];
    float fy_[8], fx_[8];            // the fractions the eight in-plane weights of a voxel are made of
    unsigned farmask = 0, outmask = 0;
    auto inplane = [&](int k, float &fy, int (&offy)[4], float &fx, int (&offx)[4]) {
        const double o1 = (double)(y0 + 8 * wave + k), o2 = (double)(x0 + lane);
        // cubic3_f32_kernel's order: s = 0; s += m[d][0] z; s += m[d][1] y; s += m[d][2] x; c = s + m[d][3] (m[d][0] = 0 here)
        double s1 = 0.0; s1 += q.m11 * o1; s1 += q.m12 * o2;
        double s2 = 0.0; s2 += q.m21 * o1; s2 += q.m22 * o2;
        const bool oy_ = cubic3_axis_frac(q.ny, q.sr, s1 + q.m13, q.mode, q.npad, fy, offy);
        const bool ox_ = cubic3_axis_frac(q.nx, 1, s2 + q.m23, q.mode, q.npad, fx, offx);
        return oy_ | ox_;
    };
    // a ROLLED loop (its body is the boundary arithmetic of every mode, twice): the three values of a voxel wait in the
    // slot area, which no DMA has touched yet, and come back into registers below
    float *park = reinterpret_cast<float *>(smem_cz) + wave * (24 * 64) + lane;
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        int offy[4], offx[4];
        float fy, fx;
        const bool outside = inplane(k, fy, offy, fx, offx);
        bool block = offy[0] >= 0 && offx[0] >= 0;
#pragma unroll
        for (int j = 1; j < 4; j++) block = block && offy[j] == offy[0] + j * q.sr && offx[j] == offx[0] + j;
        int r0 = offy[0] / q.sr - by0, c0 = offx[0] - bx0;
        const bool held = block && r0 >= 0 && r0 + 3 < q.ry && c0 >= 0 && c0 + 3 < P;
        bool edge = false;
        if (!held && !outside && rule != 3) {
            // the start the taps were counted from: a tap inside the array is its own image and gives it away; the candidate
            // is taken when czrule reproduces all four of cubic3_axis's offsets from it
            int sy = 0, sx = 0;
            bool fy_ = false, fx_ = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!fy_ && offy[j] >= 0) {
                    const int s = offy[j] / q.sr - j;
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; i++) ok = ok && offy[i] == czrule(s + i, q.ny) * q.sr;
                    if (ok) { sy = s; fy_ = true; }
                }
                if (!fx_ && offx[j] >= 0) {
                    const int s = offx[j] - j;
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; i++) ok = ok && offx[i] == czrule(s + i, q.nx);
                    if (ok) { sx = s; fx_ = true; }
                }
            }
            edge = fy_ && fx_ && sy - by0 >= -8 && sx - bx0 >= -8 && sy - by0 < 4096 && sx - bx0 < 4096;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ty = czrule(sy + j, q.ny) - by0, tx = czrule(sx + j, q.nx) - bx0;
                edge = edge && ty >= 0 && ty < q.ry && tx >= 0 && tx < P;
            }
            r0 = sy - by0; c0 = sx - bx0;
        }
        park[(3 * k) * 64] = __int_as_float((held || edge) ? ((r0 + 8) << 16) | (c0 + 8) : (8 << 16) | 8);
        park[(3 * k + 1) * 64] = fy;
        park[(3 * k + 2) * 64] = fx;
        farmask |= (held || edge || outside) ? 0u : (1u << k);
        outmask |= outside ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        a_[k] = __float_as_int(park[(3 * k) * 64]);
        fy_[k] = park[(3 * k + 1) * 64];
        fx_[k] = park[(3 * k + 2) * 64];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // before anyone's DMA lands on the parked values
    float wy_[8][4], wx_[8][4];
#pragma unroll
    for (int k = 0; k < 8; k++) { cubic3_weights(fy_[k], wy_[k]); cubic3_weights(fx_[k], wx_[k]); }
    const bool any_far = __builtin_amdgcn_ballot_w64(farmask != 0u) != 0;
    float *tile = tiles + wave * 512;
    const bool wide = x0 + 64 <= q.ox && y0 + TY <= q.oy;

    // ---- the plane ring: slot (plane mod 5), resident planes a contiguous range [rlo, rhi] of at most five
    int rlo = 0, rhi = -1;
    auto slot_of = [&](int pl) { return (unsigned)(((pl % NS) + NS) % NS); };
    auto ensure = [&](int pl) {
        if (pl < 0 || pl >= q.nz) return;
        if (pl >= rlo && pl <= rhi) return;
        if (pl == rhi + 1 && rhi >= rlo) { rhi = pl; if (rhi - rlo > NS - 1) rlo = rhi - (NS - 1); }
        else if (pl == rlo - 1 && rhi >= rlo) { rlo = pl; if (rhi - rlo > NS - 1) rhi = rlo + (NS - 1); }
        else { rlo = rhi = pl; }
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
        const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)pl * plane_b + org_b);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane(slot_of(pl) * slot_bytes + (unsigned)(wave << 6) * 16u);
#pragma unroll
        for (int j = 0; j < kCzRoundsMax; j++)
            if (j < rounds && live[j] != 0) cz_dma16(rin, rel[j], base, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(j * NT) * 16u), live[j]);
    };
    auto zplane = [&](int z) {
        CzPlane p;
        double s0 = 0.0; s0 += q.m00 * (double)z;
        float w[4]; int off[4];
        p.outside = cubic3_axis(q.nz, q.ss, s0 + q.m03, q.mode, q.npad, w, off);
        p.cvtap = false;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            p.w[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(w[k])));
            p.off[k] = __builtin_amdgcn_readfirstlane(off[k]);
            p.pl[k] = p.off[k] < 0 ? -1 : p.off[k] / q.ss;
            p.cvtap = p.cvtap || p.off[k] < 0;
        }
        // two DIFFERENT planes of one step in the same ring slot (planes that wrap around the array, a multiple of five
        // apart): the step cannot be served from the ring
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = i + 1; j < 4; j++)
                p.cvtap = p.cvtap || (p.pl[i] >= 0 && p.pl[j] >= 0 && p.pl[i] != p.pl[j] && slot_of(p.pl[i]) == slot_of(p.pl[j]));
        p.outside = __builtin_amdgcn_readfirstlane((int)p.outside) != 0;
        return p;
    };
    CzPlane cur = zplane(zs);
    if (!cur.outside)
        for (int k = 0; k < 4; k++) ensure(cur.pl[k]);
    bool drain = false;

#pragma unroll 1
    for (int z = zs; z < ze; z++) {
        // the planes of this step have landed (the two stores of the previous step, issued after their DMAs, may still be in
        // flight on full tiles -- not in the first step, and not after a step that fetched late)
        if (wide && z > zs && !drain) asm volatile(MI_VMCNT(2) ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        drain = false;
        CzPlane nxt = cur;
        int late[4] = {-1, -1, -1, -1};
        bool any_late = false;
        if (z + 1 < ze) {
            nxt = zplane(z + 1);
            if (!nxt.outside) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int pl = nxt.pl[k];
                    if (pl < 0 || pl >= q.nz || (pl >= rlo && pl <= rhi)) continue;
                    bool busy = false;
                    if (!cur.outside)
                        for (int j = 0; j < 4; j++) busy = busy || (cur.pl[j] >= 0 && slot_of(cur.pl[j]) == slot_of(pl));
                    if (busy) { late[k] = pl; any_late = true; }
                    else ensure(pl);
                }
            }
        }
        // results go through the wave's LDS tile (full tiles: two 16-byte stores per lane afterwards).  ONE store target here:
        // a choice between the tile and `out` would be compiled to flat stores, which the compiler orders against every
        // LDS read with vmcnt(0) -- the end of the prefetch
        auto emit = [&](int k, float v) { tile[k * 64 + lane] = v; };
        if (cur.outside) {
#pragma unroll
            for (int k = 0; k < 8; k++) emit(k, q.cval);
        } else if (cur.cvtap || any_far || (q.dbg & 2)) {
            // a cval tap along z, or some voxel of this wave whose in-plane block the rectangle does not hold: cubic3_gather,
            // one voxel at a time in a rolled loop (this path must not set the kernel's register count)
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
#pragma unroll 1
            for (int k = 0; k < 8; k++) {
                Cubic3 tt;                                         // [0]: the array's axis 0 = the stream axis (SAX 0) / the row axis (SAX 1)
                float fy, fx;
                const bool outside = inplane(k, fy, tt.off[1 - SAX], fx, tt.off[2]);
                cubic3_weights(fy, tt.w[1 - SAX]);
                cubic3_weights(fx, tt.w[2]);
#pragma unroll
                for (int j = 0; j < 4; j++) { tt.w[SAX][j] = cur.w[j]; tt.off[SAX][j] = cur.off[j]; }
                tt.ntap[0] = 4; tt.ntap[1] = 4;
                tt.outside = outside;
                emit(k, cubic3_gather<4, 4>(rin, tt, q.cval, q.mode));
            }
        } else {
            // every tap addressed on its own, rows and columns by czrule (the identity inside the array): 64 ds_read_b32 per
            // voxel.  (Reading a plain voxel's rows as 2 x ds_read2_b32 at immediate offsets measured SLOWER, 1.54 ms against
            // 1.42 ms on 512^3 at 7 degrees.)  The sixteen (row, column) offsets of a voxel are added up once; the plane comes
            // in as the IMMEDIATE offset of the read when the slots have the fixed size kCzSlot and the four planes follow
            // each other (everywhere but at the ends of the array): slot (s + kz) mod 5 for the five values of s, five copies
            // of the code.  All 64 reads of a voxel are issued before its arithmetic.
            const bool consecutive = cur.pl[1] == cur.pl[0] + 1 && cur.pl[2] == cur.pl[0] + 2 && cur.pl[3] == cur.pl[0] + 3;
            const int s0 = (int)slot_of(cur.pl[0]);
            unsigned pb[4];
#pragma unroll
            for (int kz = 0; kz < 4; kz++) pb[kz] = slot_of(cur.pl[kz]) * slot_bytes;
            auto voxels = [&](auto svar) {
                constexpr int S = decltype(svar)::value;            // -1: slots by their run-time offsets
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    // (an opaque copy: the offsets derived from it are the same on every plane, and hoisted out of the loop
                    // over the planes they would sit in sixty-odd registers)
                    int ak = a_[k];
                    asm volatile("" : "+v"(ak));
                    const int sy = (ak >> 16) - 8 + by0, sx = (ak & 0xffff) - 8 + bx0;
                    unsigned ro[4], co[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        // `outside` voxels: anything inside the slot (their value is replaced below)
                        int ty = czrule(sy + j, q.ny) - by0, tx = czrule(sx + j, q.nx) - bx0;
                        ty = min(max(ty, 0), q.ry - 1); tx = min(max(tx, 0), P - 1);
                        ro[j] = (unsigned)ty * (unsigned)(P * 4); co[j] = (unsigned)tx * 4u;
                    }
                    float v[4][4][4];
#pragma unroll
                    for (int ky = 0; ky < 4; ky++)
#pragma unroll
                        for (int kx = 0; kx < 4; kx++) {
                            const unsigned rc = ro[ky] + co[kx];
#pragma unroll
                            for (int kz = 0; kz < 4; kz++) {
                                if constexpr (S >= 0) v[kz][ky][kx] = *reinterpret_cast<const float *>(smem_cz + rc + ((S + kz) % NS) * kCzSlot);
                                else v[kz][ky][kx] = *reinterpret_cast<const float *>(smem_cz + (rc + pb[kz]));
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);
                    // cubic3_gather's order: the taps of the array's axis 0 outermost (its weight the first factor)
                    float acc = 0.f;
#pragma unroll
                    for (int ka = 0; ka < 4; ka++) {
#pragma unroll
                        for (int kb = 0; kb < 4; kb++) {
                            const int kz = SAX == 0 ? ka : kb, ky = SAX == 0 ? kb : ka;
                            const float wzy = SAX == 0 ? cur.w[kz] * wy_[k][ky] : wy_[k][ky] * cur.w[kz];
                            float row = v[kz][ky][0] * wx_[k][0];
                            row = fmaf(v[kz][ky][1], wx_[k][1], row);
                            row = fmaf(v[kz][ky][2], wx_[k][2], row);
                            row = fmaf(v[kz][ky][3], wx_[k][3], row);
                            acc = fmaf(row, wzy, acc);
                        }
                    }
                    emit(k, ((outmask >> k) & 1u) ? q.cval : acc);
                    __builtin_amdgcn_sched_barrier(0);              // one voxel's taps at a time
                }
            };
            if (slot_bytes == (unsigned)kCzSlot && consecutive) {
                switch (s0) {
                case 0: voxels(std::integral_constant<int, 0>{}); break;
                case 1: voxels(std::integral_constant<int, 1>{}); break;
                case 2: voxels(std::integral_constant<int, 2>{}); break;
                case 3: voxels(std::integral_constant<int, 3>{}); break;
                default: voxels(std::integral_constant<int, 4>{}); break;
                }
            } else {
                voxels(std::integral_constant<int, -1>{});
            }
        }
        if (wide) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int i = lane >> 4, c = lane & 15;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float4 v = *reinterpret_cast<const float4 *>(tile + (4 * h + i) * 64 + 4 * c);
                typedef float f32x4c __attribute__((ext_vector_type(4)));
                const f32x4c vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4c *>(out + ((size_t)z * q.oss + (size_t)(y0 + 8 * wave + 4 * h + i) * q.osr + x0 + 4 * c)));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            // edge tiles (they wait for vmcnt(0) at the top of every step)
            const int x = x0 + lane;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int y = y0 + 8 * wave + k;
                const float v = tile[k * 64 + lane];
                if (x < q.ox && y < q.oy) __builtin_nontemporal_store(v, out + ((size_t)z * q.oss + (size_t)y * q.osr + x));
            }
        }
        if (any_late) {
            __builtin_amdgcn_s_barrier();                        // everyone has read this step's planes
#pragma unroll
            for (int k = 0; k < 4; k++) if (late[k] >= 0) ensure(late[k]);
            drain = true;
        }
        cur = nxt;
    }
}

Knob g_cubic_box{1};          // 0 = the gather kernel for matrices that couple all three axes; 1 = cubic3_box_kernel (taps out of an LDS-staged box) when the
                              // box fits; bits 2 / 4 / 8: timing ablations (no second phase / no taps / no box DMA)
extern "C" int mi_debug_set_cubic_box(int on) { g_cubic_box = on; return MI_OK; }
bool launch_cubic_box(const float *in, float *out, const int shape[3], const int oshape[3], const double *mat, int mode, double cval, int npad, hipStream_t s, int *rc, int dbg);   // cubic_fast.hip
bool launch_cubic_mapbox(const float *in, const void *coords, int coords_f64, float *out, const int shape[3], const int oshape[3], int mode, double cval, int npad,
                         hipStream_t s, int *rc, int dbg);   // cubic_fast.hip
Knob g_resample_fast{1};      // test hook: 0 = the r3 separable resampling passes for diagonal order-3 transforms
extern "C" int mi_debug_set_resample_fast(int on) { g_resample_fast = on; return MI_OK; }
int launch_resample_x_lds(const float *in, float *out, const AxisTaps *tabx, long long nrows, int ox, int nx, float cval, hipStream_t s);          // cubic_fast.hip
int launch_resample_zstream(const float *in, float *out, const AxisTaps *all_tabs, int oz, int oy, int oxq, int nz, float cval, hipStream_t s);   // cubic_fast.hip
Knob g_cubic_zfactor{1};      // test hook: 0 = cubic3_zstream_kernel (r4b: 64 taps per voxel, bit-identical to the gather kernel) instead of cubic3_zfactor_kernel
extern "C" int mi_debug_set_cubic_zfactor(int on) { g_cubic_zfactor = on; return MI_OK; }
int launch_cubic_zfactor(int sax, const float *in, float *out, const CubZParams &q, size_t lds, int blocks, hipStream_t s);      // cubic_fast.hip
Knob g_cubic_zstream{1};      // test hook: 0 = the gather kernel for every non-diagonal matrix; bit 2: every wave on cubic3_gather; bit 4: any x step; bit 8: the grid modes too
extern "C" int mi_debug_set_cubic_zstream(int on) { g_cubic_zstream = on; return MI_OK; }

// plan + launch; false = not taken (the caller runs cubic3_f32_kernel)
static bool launch_cubic_zstream(const mi_array *coef, const mi_array *out, const InterpGeom &g, int mode, double cval, int npad, int ident, hipStream_t s, int *rc)
{
    *rc = MI_OK;
    if (!g_cubic_zstream || g.pad != 0) return false;
    if (ident && !g_cubic_zfactor) return false;                 // the r4b kernel has no single-tap form of the stream axis
    // the two grid modes: a voxel with a tap beyond the array (cval / the far side) sends its whole wave to cubic3_gather, and
    // the tiles along the edges then hold the launch up -- 4.97 ms against the gather kernel's 3.99 ms on 512^3, 7 degrees,
    // grid-wrap (profiles/r4_cubic_zstream.txt).  Debug bit 8 takes them all the same (the tests of those paths).
    if ((mode == MI_MODE_GRID_WRAP || mode == MI_MODE_GRID_CONSTANT) && !(g_cubic_zstream & 8)) return false;
    const double *m = g.mat;
    for (int i = 0; i < 12; i++) if (!(fabs(m[i]) < 1e9)) return false;
    // the axis that streams: axis 0 when the matrix leaves it to itself (rotations in the (y, x) plane), else axis 1
    // (rotations in the (z, x) plane)
    int sax;
    if (m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[8] == 0.0) sax = 0;
    else if (m[1] == 0.0 && m[4] == 0.0 && m[6] == 0.0 && m[9] == 0.0) sax = 1;
    else return false;
    const int ra = 1 - sax;                                                                     // the axis a plane's rows run along
    if (ident && ident != (1 << sax)) return false;                                             // the unfiltered axis must be the one that streams
    if (ident && !(m[sax * 4 + sax] == 1.0 && m[sax * 4 + 3] == floor(m[sax * 4 + 3]))) return false;
    // up to one plane per step the five ring slots hold the four planes of a step + the one the next needs; beyond that (the
    // BASELINE matrix steps 1.02 planes) every 1 / (|m| - 1) steps need TWO new planes, the second of which lands on a slot the
    // current step still reads: it is fetched late (after the step's reads, one more barrier, the next wait drains) -- r5
    if (!(fabs(m[sax * 4 + sax]) <= 1.3)) return false;
    CubZParams q;
    q.nz = (int)g.shape[sax]; q.ny = (int)g.shape[ra]; q.nx = (int)g.shape[2];
    q.oz = (int)g.oshape[sax]; q.oy = (int)g.oshape[ra]; q.ox = (int)g.oshape[2];
    if ((int64_t)q.oz * q.oy * q.ox < (1 << 18) || q.ox < 64 || q.nx < 8 || ((uintptr_t)out->data & 15) || ((uintptr_t)coef->data & 15) || (q.nx & 3)) return false;
    if ((int64_t)q.nz * q.ny * q.nx * 4 >= ((int64_t)1 << 31) || (int64_t)q.oz * q.oy * q.ox >= ((int64_t)1 << 31)) return false;
    q.vol_bytes = q.nz * q.ny * q.nx * 4;
    q.ss = sax == 0 ? q.ny * q.nx : q.nx; q.sr = sax == 0 ? q.nx : q.nz * q.nx;
    q.oss = sax == 0 ? q.oy * q.ox : q.ox; q.osr = sax == 0 ? q.ox : q.oz * q.ox;
    q.m00 = m[sax * 4 + sax]; q.m03 = m[sax * 4 + 3];
    q.m11 = m[ra * 4 + ra]; q.m12 = m[ra * 4 + 2]; q.m13 = m[ra * 4 + 3];
    q.m21 = m[8 + ra]; q.m22 = m[10]; q.m23 = m[11];
    // the lanes of a wave are neighbours along the output's x: their taps must be spread along the rows of the rectangle
    // rather than down its columns (row pitch 80 floats: sixteen banks apart, a 16-way conflict at 90 degrees)
    if (!(g_cubic_zstream & 4) && !(fabs(q.m22) >= kCzMinXStep * fabs(q.m12))) return false;
    const int T[2] = {kCzTY - 1, 63};
    const double ey = fabs(q.m11) * T[0] + fabs(q.m12) * T[1], ex = fabs(q.m21) * T[0] + fabs(q.m22) * T[1];
    if (!(ey < 4096.0 && ex < 4096.0)) return false;
    // tap rows floor(min - hair) - 1 .. floor(max) + 2: floor(ext + hair) + 5 of them; x: + up to 3 for the alignment
    const int ry = (int)floor(ey * (1.0 + 1e-6) + 2e-3) + 5;
    const int rx = (int)floor(ex * (1.0 + 1e-6) + 2e-3) + 5 + 3;
    if (rx > kCzP) return false;
    q.ry = ry;
    q.nchunks = ry * (kCzP / 4);
    if ((q.nchunks + kCzNT - 1) / kCzNT > kCzRoundsMax) return false;
    const size_t slot = (size_t)q.nchunks * 16 <= (size_t)kCzSlot ? (size_t)kCzSlot : (size_t)q.nchunks * 16;
    q.slot_bytes = (int)slot;
    const size_t lds = std::max(kCzSlots * slot, (size_t)4 * 24 * 64 * 4) + 4 * 2048;     // the slots (which also park the per-voxel values during the set-up) + the output tiles
    if (lds + 1024 > 160 * 1024) return false;                                              // one workgroup per CU at the steepest angles (two up to ~10 degrees)
    const double e[4] = {q.m11 * T[0], q.m12 * T[1], q.m21 * T[0], q.m22 * T[1]};
    q.cmin_y = (e[0] < 0.0 ? e[0] : 0.0) + (e[1] < 0.0 ? e[1] : 0.0);
    q.cmin_x = (e[2] < 0.0 ? e[2] : 0.0) + (e[3] < 0.0 ? e[3] : 0.0);
    q.ntx = (q.ox + 63) / 64;
    q.nty = (q.oy + kCzTY - 1) / kCzTY;
    const int tiles = q.ntx * q.nty;
    int nzc = (2 * device_cus() + tiles - 1) / tiles;
    nzc = std::max(1, std::min(nzc, (q.oz + 15) / 16));
    q.zc = (q.oz + nzc - 1) / nzc;
    q.nzc = (q.oz + q.zc - 1) / q.zc;
    q.mode = mode; q.npad = npad; q.cval = (float)cval;
    q.dbg = g_cubic_zstream;
    q.sident = ident ? 1 : 0;
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        hipError_t e_ = hipFuncSetAttribute((const void *)cubic3_zstream_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (e_ == hipSuccess) e_ = hipFuncSetAttribute((const void *)cubic3_zstream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        if (e_ != hipSuccess) { *rc = hip_fail(e_, "hipFuncSetAttribute(cubic3_zstream_kernel)"); return true; }
        attr_done = true;
    }
    if (g_cubic_zfactor) {
        // r5: the same plan, evaluated plane by plane (csrc/cubic_fast.hip: in-plane values once per input plane, four-term blend
        // per voxel) -- float32 rounding away from the gather kernel, not bit-identical; knob 0 keeps the r4b kernel
        note_kernel("mi::cubic3_zfactor_kernel<%d> grid=%d (order-3 affine on float32 coefficients, axis %d decoupled: streams along it, in-plane values once per input plane, %d rows x %d staged per plane, %d chunks)",
                    sax, tiles * q.nzc, sax, q.ry, kCzP, q.nzc);
        const int frc = launch_cubic_zfactor(sax, (const float *)coef->data, (float *)out->data, q, lds, tiles * q.nzc, s);
        if (frc != MI_ERR_UNSUPPORTED) { *rc = frc; return true; }
        if (ident) return false;             // (more than 65535 output planes: the r4b kernel, which has no single-tap form)
    }
    note_kernel("mi::cubic3_zstream_kernel<%d> grid=%d (order-3 affine on float32 coefficients, axis %d decoupled: streams along it, %d rows x %d staged per plane, %d chunks)",
                sax, tiles * q.nzc, sax, q.ry, kCzP, q.nzc);
    if (sax == 0) hipLaunchKernelGGL(cubic3_zstream_kernel<0>, dim3((unsigned)(tiles * q.nzc)), dim3(kCzNT), lds, s, (const float *)coef->data, (float *)out->data, q);
    else hipLaunchKernelGGL(cubic3_zstream_kernel<1>, dim3((unsigned)(tiles * q.nzc)), dim3(kCzNT), lds, s, (const float *)coef->data, (float *)out->data, q);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) *rc = hip_fail(e2, "cubic3_zstream_kernel");
    return true;
}

// ---------------------------------------------------------------------------
// r4b: cubic interpolation (float32 coefficients) for matrices that leave the x axis to itself with unit step and an
// integral shift -- rotations / shears / scalings in the (z, y) plane: `rotate(volume, a)` with SciPy's DEFAULT axes and
// DEFAULT order.  The sixteen (z, y) taps of a voxel are the same for a whole output row and its x taps are its
// neighbours at the fraction 0: a wave takes (a 512-voxel piece of) one output row, computes the row's taps once, and
// every lane blends four consecutive voxels from 16 rows x two 16-byte loads at a UNIFORM row base (scalar offset) -- 8
// coalesced loads per voxel instead of cubic3_f32_kernel's 16 gathers at per-lane addresses.  Products, weights and the
// order of the sums are cubic3_gather's: bit-identical.  Voxels whose x taps touch the ends of the row (and anything
// else that is not four plain taps) are collected over the wave and take cubic3_gather itself, one voxel per lane.
// ---------------------------------------------------------------------------
struct CubRowParams {
    int nz, ny, nx, oz, oy, ox;
    double m00, m01, m03, m10, m11, m13;
    int xs;                      // the integral x shift
    int xsegs;                   // 512-voxel pieces per output row
    int mode, npad;
    float cval;
    int xident;                  // r5: the x axis holds SAMPLES (its prefilter pass was skipped): the x taps are the one sample at the voxel's column
};

__global__ void __launch_bounds__(256)
cubic3_rowblend_kernel(const float *__restrict__ in, float *__restrict__ out, const CubRowParams q)
{
    typedef float f32x4r __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    // (workgroups in launch order; renumbering them so that every XCD works through one contiguous eighth of the rows --
    // its neighbours' input rows in its own L2 -- measured slower: 907 / 918 against 881 / 833 us at 7 / 30 degrees)
    const int seg = blockIdx.x % q.xsegs, y = (blockIdx.x / q.xsegs) * 4 + (int)threadIdx.y, z = blockIdx.y;
    if (y >= q.oy) return;
    const int nxy = q.ny * q.nx;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, q.nz * nxy * 4, 0x00020000);
    // the row's taps along z and y (cubic3_f32_kernel's order of the coordinate sums; the x terms are 0 * x).  (Four rows
    // per wave with the taps of the four computed side by side in lanes 0-3 measured slower: 0.97 against 0.86 ms.)
    Cubic3 t;
    {
        double s0 = 0.0; s0 += q.m00 * (double)z; s0 += q.m01 * (double)y;
        double s1 = 0.0; s1 += q.m10 * (double)z; s1 += q.m11 * (double)y;
        const bool o0 = cubic3_axis(q.nz, nxy, s0 + q.m03, q.mode, q.npad, t.w[0], t.off[0]);
        const bool o1 = cubic3_axis(q.ny, q.nx, s1 + q.m13, q.mode, q.npad, t.w[1], t.off[1]);
        t.outside = o0 | o1;
        t.ntap[0] = 4; t.ntap[1] = 4;
#pragma unroll
        for (int d = 0; d < 2; d++)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                t.w[d][k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(t.w[d][k])));
                t.off[d][k] = __builtin_amdgcn_readfirstlane(t.off[d][k]);
            }
    }
    const bool row_outside = __builtin_amdgcn_readfirstlane((int)t.outside) != 0;
    bool cvrow = false;                       // a cval tap along z or y (grid-constant): the row takes cubic3_gather
#pragma unroll
    for (int k = 0; k < 4; k++) cvrow = cvrow || t.off[0][k] < 0 || t.off[1][k] < 0;
    float wx[4];
    cubic3_weights(0.f, wx);
    float *orow = out + ((size_t)z * q.oy + y) * q.ox;
    unsigned long long slow[2];
#pragma unroll
    for (int g = 0; g < 2; g++) {
        const int x4 = seg * 512 + (g * 64 + lane) * 4;
        const int start0 = x4 + q.xs + q.npad - 1;              // first tap of the first voxel, in the (padded) row
        const bool live = x4 < q.ox;
        const bool plain = live && x4 + 3 < q.ox && !cvrow &&
                           (row_outside || (q.xident ? (start0 + 1 >= 0 && start0 + 4 < q.nx) : (start0 >= 0 && start0 + 6 < q.nx)));
        slow[g] = __builtin_amdgcn_ballot_w64(live && !plain);
        if (!plain) continue;
        f32x4r r;
        if (row_outside) {
            r = f32x4r{q.cval, q.cval, q.cval, q.cval};
        } else if (q.xident) {
            // the x axis holds samples: one 16-byte load per input row, the row IS the x sum (sixteen loads and FMAs per four
            // voxels instead of 32 and 80)
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kz = 0; kz < 4; kz++) {
                u32x4 a[4];
#pragma unroll
                for (int ky = 0; ky < 4; ky++)
                    a[ky] = __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)(start0 + 1) * 4u, (unsigned)(t.off[0][kz] + t.off[1][ky]) * 4u, 0);
#pragma unroll
                for (int ky = 0; ky < 4; ky++) {
                    const float wzy = t.w[0][kz] * t.w[1][ky];
                    acc[0] = fmaf(__uint_as_float(a[ky].x), wzy, acc[0]);
                    acc[1] = fmaf(__uint_as_float(a[ky].y), wzy, acc[1]);
                    acc[2] = fmaf(__uint_as_float(a[ky].z), wzy, acc[2]);
                    acc[3] = fmaf(__uint_as_float(a[ky].w), wzy, acc[3]);
                }
            }
            r = f32x4r{acc[0], acc[1], acc[2], acc[3]};
        } else {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kz = 0; kz < 4; kz++) {
                u32x4 a[4], b[4];
#pragma unroll
                for (int ky = 0; ky < 4; ky++) {
                    const unsigned base = (unsigned)(t.off[0][kz] + t.off[1][ky]) * 4u;
                    a[ky] = __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)start0 * 4u, base, 0);
                    b[ky] = __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)start0 * 4u + 16u, base, 0);
                }
#pragma unroll
                for (int ky = 0; ky < 4; ky++) {
                    const float f[8] = {__uint_as_float(a[ky].x), __uint_as_float(a[ky].y), __uint_as_float(a[ky].z), __uint_as_float(a[ky].w),
                                        __uint_as_float(b[ky].x), __uint_as_float(b[ky].y), __uint_as_float(b[ky].z), __uint_as_float(b[ky].w)};
                    const float wzy = t.w[0][kz] * t.w[1][ky];
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        float row = f[i] * wx[0];
                        row = fmaf(f[i + 1], wx[1], row);
                        row = fmaf(f[i + 2], wx[2], row);
                        row = fmaf(f[i + 3], wx[3], row);
                        acc[i] = fmaf(row, wzy, acc[i]);
                    }
                }
            }
            r = f32x4r{acc[0], acc[1], acc[2], acc[3]};
        }
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4r *>(orow + x4));
    }
    // the other voxels: up to sixteen groups of four per round, one voxel per lane
    unsigned long long m0 = slow[0], m1 = slow[1];
    while ((m0 | m1) != 0) {
        int mine = -1;
#pragma unroll 1
        for (int k = 0; k < 16 && (m0 | m1) != 0; k++) {
            int gi;
            if (m0 != 0) { gi = __builtin_ctzll(m0); m0 &= m0 - 1; }
            else { gi = 64 + __builtin_ctzll(m1); m1 &= m1 - 1; }
            if ((lane >> 2) == k) mine = gi;
        }
        const int x = seg * 512 + mine * 4 + (lane & 3);
        if (mine >= 0 && x < q.ox) {
            double s2 = 0.0; s2 += (double)x;                       // 0 * z + 0 * y + 1 * x
            const bool o2 = cubic3_axis(q.nx, 1, s2 + (double)q.xs, q.mode, q.npad, t.w[2], t.off[2]);
            Cubic3 tt = t;
            tt.outside = t.outside | o2;
            if (q.xident) {
                // the tap AT the (integral) coordinate is the second of the four; the others do not exist for this axis
                const int at = tt.off[2][1];
#pragma unroll
                for (int k = 0; k < 4; k++) { tt.off[2][k] = at; tt.w[2][k] = k == 1 ? 1.f : 0.f; }
            }
            __builtin_nontemporal_store(cubic3_gather<4, 4>(rin, tt, q.cval, q.mode), orow + x);
        }
    }
}

Knob g_cubic_rowblend{1};      // test hook: 0 = the gather kernel
extern "C" int mi_debug_set_cubic_rowblend(int on) { g_cubic_rowblend = on; return MI_OK; }

// plan + launch; false = not taken (the caller runs cubic3_f32_kernel)
static bool launch_cubic_rowblend(const mi_array *coef, const mi_array *out, const InterpGeom &g, int mode, double cval, int npad, bool xident, hipStream_t s, int *rc)
{
    *rc = MI_OK;
    if (!g_cubic_rowblend || g.pad != 0) return false;
    const double *m = g.mat;
    for (int i = 0; i < 12; i++) if (!(fabs(m[i]) < 1e9)) return false;
    if (m[2] != 0.0 || m[6] != 0.0 || m[8] != 0.0 || m[9] != 0.0 || m[10] != 1.0) return false;       // x to itself, unit step
    if (m[11] != floor(m[11]) || fabs(m[11]) > 1048576.0) return false;                                // integral shift
    CubRowParams q;
    q.nz = (int)g.shape[0]; q.ny = (int)g.shape[1]; q.nx = (int)g.shape[2];
    q.oz = (int)g.oshape[0]; q.oy = (int)g.oshape[1]; q.ox = (int)g.oshape[2];
    // (rows of any length: the 16-byte loads and stores are element-aligned only, the last ox % 4 voxels of a row take the
    // per-voxel path)
    if ((int64_t)q.oz * q.oy * q.ox < (1 << 16) || q.ox < 64 || q.nx < 8 || ((uintptr_t)out->data & 3) || q.oz > 65535) return false;
    if ((int64_t)q.nz * q.ny * q.nx * 4 >= ((int64_t)1 << 31)) return false;
    q.m00 = m[0]; q.m01 = m[1]; q.m03 = m[3];
    q.m10 = m[4]; q.m11 = m[5]; q.m13 = m[7];
    q.xs = (int)m[11];
    q.xsegs = (q.ox + 511) / 512;
    q.mode = mode; q.npad = npad; q.cval = (float)cval;
    q.xident = xident ? 1 : 0;
    const long long blocks = (long long)q.xsegs * ((q.oy + 3) / 4);
    if (blocks > 0x7fffffffLL) return false;
    note_kernel("mi::cubic3_rowblend_kernel grid=%lldx%d (order-3 affine on float32 coefficients, x axis to itself: 16 rows x %s per four voxels)", blocks, q.oz,
                xident ? "ONE 16-byte load (x holds samples)" : "two 16-byte loads");
    hipLaunchKernelGGL(cubic3_rowblend_kernel, dim3((unsigned)blocks, (unsigned)q.oz), dim3(64, 4), 0, s, (const float *)coef->data, (float *)out->data, q);
    hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) *rc = hip_fail(e2, "cubic3_rowblend_kernel");
    return true;
}

// output geometry of the cubic kernel: the output's own shape, rank-padded with leading ones
static bool cubic3_grid(const mi_array *out, InterpGeom *g, dim3 *grid)
{
    const int pad = 3 - out->ndim;
    for (int d = 0; d < 3; d++) g->oshape[d] = d < pad ? 1 : out->shape[d - pad];
    if (g->oshape[0] > 65535 || (g->oshape[1] + 3) / 4 > 65535 || g->oshape[2] >= ((int64_t)1 << 31)) return false;
    *grid = dim3((unsigned)((g->oshape[2] + 63) / 64), (unsigned)((g->oshape[1] + 3) / 4), (unsigned)g->oshape[0]);
    return true;
}

// ---------------------------------------------------------------------------
// spline coefficients: padded float64 copy + in-place prefilter
// ---------------------------------------------------------------------------
// out (float64, shape = in.shape + 2 npad on the real axes) = in extended by edge
// replication (pad_mode 0) or by cval (pad_mode 1); rank padded to 3
template <typename T, typename CF, int ND = 3>
__global__ void __launch_bounds__(256)
spline_pad_kernel(const T *__restrict__ in, CF *__restrict__ out, InterpGeom g, int64_t nout, int npad, int pad_mode,
                  double cval)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i, pos = 0;
        bool outside = false;
#pragma unroll
        for (int d = ND - 1; d >= 0; d--) {
            const int64_t q = r / g.oshape[d];
            int64_t o = r - q * g.oshape[d];
            r = q;
            if (d >= g.pad) {
                o -= npad;
                if (o < 0 || o >= g.shape[d]) { outside = true; o = o < 0 ? 0 : g.shape[d] - 1; }
            }
            pos += o * g.stride[d];
        }
        out[i] = (CF)((outside && pad_mode == 1) ? cval : (double)in[pos]);
    }
}

// One thread per line; smode: 0 mirror, 1 reflect, 2 grid-wrap.  The recursions
// are sequential along the line, so the kernel is latency-bound: samples are
// fetched eight at a time (independent loads in flight) before the dependent
// chain runs over them in registers, the boundary sums stop once z^i is below
// 1e-20 (SciPy sums the whole line; the neglected tail is far below one ulp), and
// the gain is applied with the last store (the filter is linear).
// z^(n-1) and z^n of the (up to two) poles for a line length n, computed on the HOST with the C library's pow() --
// the function SciPy's ni_splines.c calls -- and handed to the kernels: the device pow() is accurate to an ulp, not
// bit-identical, and on very short lines (n = 2: the power IS the pole) that last bit decided exact .5 ties of
// integer outputs (the one mismatch class the round-2 fuzzer kept finding).
struct SplPow { double zn1[2], zn[2]; };

// `spline_mode | kSplExact` (mi_spline_filter1d / mi_spline_prefilter): only the kernels whose arithmetic is SciPy's
// operation for operation.  The blocked prefilter restarts the recursion 40 samples before a chunk: accurate to 1e-22
// of the data range, but with a different rounding history -- ~60 % of its float64 coefficients differ from SciPy's in
// the last bit (scripts/diag_spline_bits.py), which decides exact .5 ties of INTEGER outputs.  The Python layer sets
// the flag whenever the interpolated result is rounded to an integer dtype.
constexpr int kSplExact = 0x100;

constexpr int kSplBatch = 8;

// coefficient storage that reads / writes double whatever the element type (float32 coefficients
// for the float32 interpolation path: the recursion itself always runs in double)
template <typename CF>
struct CoefLine {
    CF *p;
    struct Ref {
        CF *q;
        __device__ __forceinline__ operator double() const { return (double)*q; }
        __device__ __forceinline__ Ref &operator=(double v) { *q = (CF)v; return *this; }
    };
    __device__ __forceinline__ Ref operator[](int64_t i) const { return Ref{p + i}; }
};

// one line of the prefilter: every pole's causal start, causal sweep, anti-causal start and sweep, on whatever memory the
// accessors point at (the array itself, or a copy of the line in LDS: spline_filter_rows_lds_kernel)
template <typename CF>
__device__ __forceinline__ void spline_line_body(const CoefLine<CF> c, const CoefLine<CF> first, int64_t n, int64_t st, int order,
                                                 int smode, int gain_first, const SplPow &pw);

template <typename CF>
__global__ void __launch_bounds__(64)
spline_filter1d_kernel(CF *__restrict__ data, const CF *__restrict__ src, int64_t n, int64_t inner, int64_t nlines, int order,
                       int smode, int gain_first, SplPow pw)
{
    const int64_t line = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (line >= nlines || n <= 1) return;
    const int64_t line_off = (line / inner) * n * inner + (line % inner);
    const CoefLine<CF> c{data + line_off};
    // out of place: the first pole's causal phase reads the source, everything else the coefficients
    const CoefLine<CF> first{src ? const_cast<CF *>(src) + line_off : data + line_off};
    spline_line_body<CF>(c, first, n, inner, order, smode, gain_first, pw);
}

// (defined below the kernel that was its only user until r4b; declared above it)
template <typename CF>
__device__ __forceinline__ void spline_line_body(const CoefLine<CF> c, const CoefLine<CF> first, int64_t n, int64_t st, int order,
                                                 int smode, int gain_first, const SplPow &pw)
{
    double zs[2];
    int np = 1;
    switch (order) {
    case 2: zs[0] = -0.171572875253809902396622551580603843; break;
    case 3: zs[0] = -0.267949192431122706472553658494127633; break;
    case 4: zs[0] = -0.361341225900220177092212841325675255; zs[1] = -0.013725429297339121360331226939128204; np = 2; break;
    default: zs[0] = -0.430575347099973791851434783493520110; zs[1] = -0.043096288203264653822712376822550182; np = 2; break;
    }
    double gain = 1.0;
    for (int k = 0; k < np; k++) gain *= (1.0 - zs[k]) * (1.0 - 1.0 / zs[k]);
    for (int k = 0; k < np; k++) {
        const double z = zs[k];
        const bool last_pole = k == np - 1;
        int64_t H = (int64_t)ceil(-46.0517 / log(fabs(z)));        // |z|^H < 1e-20
        // ---- causal initialisation
        // gain_first: the samples are scaled by the pole gain as they are read (SciPy scales the line before it filters;
        // the roundings then agree with SciPy's, which matters where an integer output sits on a tie)
        struct ScaledLine {
            CoefLine<CF> l;
            double g;
            __device__ __forceinline__ double operator[](int64_t i) const { return (double)l[i] * g; }
        };
        const ScaledLine rd{k == 0 ? first : c, (k == 0 && gain_first) ? gain : 1.0};
        double c0 = rd[0];
        {
            double z_i = z;
            if (smode == 0) {
                const double z_n_1 = pw.zn1[k];
                double acc = c0 + z_n_1 * rd[(n - 1) * st];
                const int64_t m = (n - 1 < H + 1) ? n - 1 : H + 1;
                for (int64_t i = 1; i < m; i++) { acc += z_i * (rd[i * st] + z_n_1 * rd[(n - 1 - i) * st]); z_i *= z; }
                c0 = acc / (1 - z_n_1 * z_n_1);
            } else if (smode == 2) {
                double acc = c0;
                const int64_t m = (n < H + 1) ? n : H + 1;
                for (int64_t i = 1; i < m; i++) { acc += z_i * rd[(n - i) * st]; z_i *= z; }
                // SciPy divides by 1 - z_i with the RUNNING product of its loop (z^n by repeated multiplication, not
                // pow): the same value here whenever the loop ran to the end; beyond the horizon 1 - z^n is 1 either way
                const double z_n = m == n ? z_i : pw.zn[k];
                c0 = acc / (1 - z_n);
            } else {
                const double z_n = pw.zn[k];
                double acc = c0 + z_n * rd[(n - 1) * st];
                const int64_t m = (n < H + 1) ? n : H + 1;
                for (int64_t i = 1; i < m; i++) {
                    // SciPy updates c[0] in place: the last term (i == n - 1) sees the partially summed value
                    const double mirror_term = (i == n - 1) ? acc : rd[(n - 1 - i) * st];
                    acc += z_i * (rd[i * st] + z_n * mirror_term);
                    z_i *= z;
                }
                acc *= z / (1 - z_n * z_n);
                c0 = acc + c0;
            }
        }
        c[0] = c0;
        // ---- causal sweep
        double prev = c0;
        int64_t i = 1;
        for (; i + kSplBatch <= n; i += kSplBatch) {
            double v[kSplBatch];
#pragma unroll
            for (int u = 0; u < kSplBatch; u++) v[u] = rd[(i + u) * st];
#pragma unroll
            for (int u = 0; u < kSplBatch; u++) { prev = v[u] + z * prev; v[u] = prev; }
#pragma unroll
            for (int u = 0; u < kSplBatch; u++) c[(i + u) * st] = v[u];
        }
        for (; i < n; i++) { prev = rd[i * st] + z * prev; c[i * st] = prev; }
        // ---- anti-causal initialisation (prev == c[n - 1] after the causal sweep)
        double last = prev;
        if (smode == 0) {
            last = (z * c[(n - 2) * st] + last) * z / (z * z - 1);
        } else if (smode == 2) {
            double z_i = z, acc = last;
            const int64_t m = (n - 1 < H) ? n - 1 : H;
            for (int64_t j = 0; j < m; j++) { acc += z_i * c[j * st]; z_i *= z; }
            const double z_n = m == n - 1 ? z_i : pw.zn[k];        // running product, as SciPy (see the causal start)
            last = acc * (z / (z_n - 1));       // SciPy: c[n-1] *= z / (z_i - 1) -- the quotient first
        } else {
            last *= z / (z - 1);
        }
        const double scale = (last_pole && !gain_first) ? gain : 1.0;
        c[(n - 1) * st] = last * scale;
        // ---- anti-causal sweep
        double nxt = last;
        int64_t j = n - 2;
        for (; j - (kSplBatch - 1) >= 0; j -= kSplBatch) {
            double v[kSplBatch];
#pragma unroll
            for (int u = 0; u < kSplBatch; u++) v[u] = c[(j - u) * st];
#pragma unroll
            for (int u = 0; u < kSplBatch; u++) { nxt = z * (nxt - v[u]); v[u] = nxt; }
#pragma unroll
            for (int u = 0; u < kSplBatch; u++) c[(j - u) * st] = v[u] * scale;
        }
        for (; j >= 0; j--) { nxt = z * (nxt - c[j * st]); c[j * st] = nxt * scale; }
    }
}

// The same filter for lines that are contiguous in memory (the last axis), where one thread per
// line would read 64 different cache lines per step: a wave owns 64 lines and moves 64 x 64 sample
// tiles through LDS -- coalesced row loads, each thread runs the recursion along its own row of
// the tile, coalesced row stores.  Arithmetic, rounding points and summation order are those of
// spline_filter1d_kernel (the results are identical).  The rows of a tile are written by other lanes
// than the one that filters them, hence the fences between the sweeps.
constexpr int kSplTile = 64;
constexpr int kSplRows = 16;      // tile rows moved per batch: that many loads in flight per lane

template <typename CF>
__device__ __forceinline__ void spline_tile_load(CF (&tile)[kSplTile][kSplTile + 1], const CF *__restrict__ base, int64_t n,
                                                 int64_t s0, int cnt, int nrows, int lane)
{
    for (int r0 = 0; r0 < nrows; r0 += kSplRows) {
        CF v[kSplRows];
#pragma unroll
        for (int u = 0; u < kSplRows; u++)
            v[u] = (r0 + u < nrows && lane < cnt) ? base[(int64_t)(r0 + u) * n + s0 + lane] : (CF)0;
#pragma unroll
        for (int u = 0; u < kSplRows; u++) tile[r0 + u][lane] = v[u];
    }
}

template <typename CF>
__device__ __forceinline__ void spline_tile_store(const CF (&tile)[kSplTile][kSplTile + 1], CF *__restrict__ base, int64_t n,
                                                  int64_t s0, int cnt, int nrows, int lane)
{
    for (int r0 = 0; r0 < nrows; r0 += kSplRows) {
#pragma unroll
        for (int u = 0; u < kSplRows; u++)
            if (r0 + u < nrows && lane < cnt) base[(int64_t)(r0 + u) * n + s0 + lane] = tile[r0 + u][lane];
    }
}

template <typename CF>
__global__ void __launch_bounds__(64)
spline_filter_rows_kernel(CF *__restrict__ data, int64_t n, int64_t nlines, int order, int smode, SplPow pw)
{
    __shared__ CF tile[kSplTile][kSplTile + 1];
    const int lane = threadIdx.x;
    const int64_t line0 = (int64_t)blockIdx.x * kSplTile;
    const int nrows = (int)((nlines - line0 < kSplTile) ? nlines - line0 : kSplTile);
    const bool valid = lane < nrows;
    CF *base = data + line0 * n;
    const CoefLine<CF> c{base + (int64_t)(valid ? lane : 0) * n};
    double zs[2];
    int np = 1;
    switch (order) {
    case 2: zs[0] = -0.171572875253809902396622551580603843; break;
    case 3: zs[0] = -0.267949192431122706472553658494127633; break;
    case 4: zs[0] = -0.361341225900220177092212841325675255; zs[1] = -0.013725429297339121360331226939128204; np = 2; break;
    default: zs[0] = -0.430575347099973791851434783493520110; zs[1] = -0.043096288203264653822712376822550182; np = 2; break;
    }
    double gain = 1.0;
    for (int k = 0; k < np; k++) gain *= (1.0 - zs[k]) * (1.0 - 1.0 / zs[k]);
    const int ntiles = (int)((n + kSplTile - 1) / kSplTile);
    for (int k = 0; k < np; k++) {
        const double z = zs[k];
        const bool last_pole = k == np - 1;
        const int64_t H = (int64_t)ceil(-46.0517 / log(fabs(z)));
        // ---- causal initialisation (reads the line as the previous sweep left it)
        double c0 = c[0];
        {
            double z_i = z;
            if (smode == 0) {
                const double z_n_1 = pw.zn1[k];
                double acc = c0 + z_n_1 * c[n - 1];
                const int64_t m = (n - 1 < H + 1) ? n - 1 : H + 1;
                for (int64_t i = 1; i < m; i++) { acc += z_i * (c[i] + z_n_1 * c[n - 1 - i]); z_i *= z; }
                c0 = acc / (1 - z_n_1 * z_n_1);
            } else if (smode == 2) {
                double acc = c0;
                const int64_t m = (n < H + 1) ? n : H + 1;
                for (int64_t i = 1; i < m; i++) { acc += z_i * c[n - i]; z_i *= z; }
                const double z_n = m == n ? z_i : pw.zn[k];
                c0 = acc / (1 - z_n);
            } else {
                const double z_n = pw.zn[k];
                double acc = c0 + z_n * c[n - 1];
                const int64_t m = (n < H + 1) ? n : H + 1;
                for (int64_t i = 1; i < m; i++) {
                    const double mirror_term = (i == n - 1) ? acc : (double)c[n - 1 - i];
                    acc += z_i * (c[i] + z_n * mirror_term);
                    z_i *= z;
                }
                acc *= z / (1 - z_n * z_n);
                c0 = acc + c0;
            }
        }
        // ---- causal sweep, tile by tile
        double prev = c0, prev2 = 0.0;          // c+[i - 1], c+[i - 2]
        double wrap_acc = 0.0, wrap_zi = z;     // grid-wrap: sum_j z^(j+1) c+[j], j < min(n - 1, H)
        const int64_t wrap_m = (n - 1 < H) ? n - 1 : H;
        for (int t = 0; t < ntiles; t++) {
            const int64_t s0 = (int64_t)t * kSplTile;
            const int cnt = (int)((n - s0 < kSplTile) ? n - s0 : kSplTile);
            spline_tile_load(tile, base, n, s0, cnt, nrows, lane);
            __syncthreads();
            if (valid) {
                for (int i0 = 0; i0 < cnt; i0 += kSplBatch) {
                    CF v[kSplBatch];
#pragma unroll
                    for (int u = 0; u < kSplBatch; u++) v[u] = tile[lane][i0 + u];
#pragma unroll
                    for (int u = 0; u < kSplBatch; u++) {
                        const int i = i0 + u;
                        if (i < cnt) {
                            double w;
                            if (t == 0 && i == 0) w = c0;
                            else { w = (double)v[u] + z * prev; prev2 = prev; }
                            prev = w;
                            v[u] = (CF)w;
                            if (smode == 2 && s0 + i < wrap_m) { wrap_acc += wrap_zi * (double)v[u]; wrap_zi *= z; }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kSplBatch; u++) tile[lane][i0 + u] = v[u];
                }
            }
            __syncthreads();
            spline_tile_store(tile, base, n, s0, cnt, nrows, lane);
            __syncthreads();
        }
        __threadfence();
        // c+[n - 2] is read back from the array (rounded to the coefficient type) by the one-thread-per-line kernel
        prev2 = (double)(CF)prev2;
        // ---- anti-causal initialisation
        double last = prev;
        if (smode == 0) {
            last = (z * prev2 + last) * z / (z * z - 1);
        } else if (smode == 2) {
            const double z_n = pw.zn[k];
            last = (last + wrap_acc) * (z / (z_n - 1));
        } else {
            last *= z / (z - 1);
        }
        const double scale = last_pole ? gain : 1.0;
        // ---- anti-causal sweep
        double nxt = last;
        for (int t = ntiles - 1; t >= 0; t--) {
            const int64_t s0 = (int64_t)t * kSplTile;
            const int cnt = (int)((n - s0 < kSplTile) ? n - s0 : kSplTile);
            spline_tile_load(tile, base, n, s0, cnt, nrows, lane);
            __syncthreads();
            if (valid) {
                for (int i0 = kSplTile - kSplBatch; i0 >= 0; i0 -= kSplBatch) {
                    if (i0 >= cnt) continue;
                    CF v[kSplBatch];
#pragma unroll
                    for (int u = 0; u < kSplBatch; u++) v[u] = tile[lane][i0 + u];
#pragma unroll
                    for (int u = kSplBatch - 1; u >= 0; u--) {
                        const int i = i0 + u;
                        if (i < cnt) {
                            if (t == ntiles - 1 && i == cnt - 1) nxt = last;
                            else nxt = z * (nxt - (double)v[u]);
                            v[u] = (CF)(nxt * scale);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kSplBatch; u++) tile[lane][i0 + u] = v[u];
                }
            }
            __syncthreads();
            spline_tile_store(tile, base, n, s0, cnt, nrows, lane);
            __syncthreads();
        }
        __threadfence();
    }
}

// r4b: contiguous lines that fit LDS WHOLE (twelve lines per wave, see the launch): the tiled kernel above moves every sample through memory twice per pole (causal sweep out, anti-causal sweep back
// in: 0.94 ms per 512^3 pass against 0.41 ms for the strided axes); here a wave loads its lines once (LDS-DMA,
// consecutive lines are consecutive in memory), lanes 0 .. lines-1 run spline_line_body -- the one-thread-per-line
// kernel's own code, hence its exact results -- on the copy (row pitch n + 1: the lanes' samples in different banks), and
// the wave stores the lines once.
// one dword per lane from memory straight into LDS (M0 = LDS byte address of lane 0's dword); lanes outside `mask` idle
__device__ __forceinline__ void spl_dma4(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_base, unsigned long long mask)
{
    unsigned keep;
    unsigned long long keep_exec;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b64 %1, exec\n\t"
        "s_mov_b32 m0, %5\n\t"
        "s_mov_b64 exec, %6\n\t"
        "buffer_load_dword %2, %3, %4 offen lds\n\t"
        "s_mov_b64 exec, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep), "=&s"(keep_exec)
        : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base), "s"(mask)
        : "memory");
}

template <typename CF>
__global__ void __launch_bounds__(64)
spline_filter_rows_lds_kernel(CF *__restrict__ data, int n, int64_t nlines, int lpb, int order, int smode, int gain_first, SplPow pw)
{
    extern __shared__ __attribute__((aligned(16))) char spl_smem[];
    CF *tile = reinterpret_cast<CF *>(spl_smem);
    constexpr int DW = sizeof(CF) / 4;                  // dwords per coefficient
    const int lane = threadIdx.x, pitch = n + 1;
    const int64_t line0 = (int64_t)blockIdx.x * lpb;
    const int nrows = __builtin_amdgcn_readfirstlane((int)((nlines - line0 < lpb) ? nlines - line0 : lpb));
    CF *base = data + line0 * n;
    // ---- in: LDS-DMA, one dword per lane and instruction, a row in pieces of 64 dwords -- every load of the block in flight
    // at once, no registers involved (eight scalar loads per lane at a time made the load phase a chain of 32 round trips)
    {
        const int rowdw = n * DW;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, nrows * rowdw * 4, 0x00020000);
        const int pieces = (rowdw + 63) / 64;
        for (int r = 0; r < nrows; r++)
            for (int j = 0; j < pieces; j++) {
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(64 * j + lane < rowdw);
                spl_dma4(rsrc, (unsigned)lane * 4u, (unsigned)(r * rowdw + 64 * j) * 4u, (unsigned)(r * pitch * DW + 64 * j) * 4u, mask);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (lane < nrows) {
        const CoefLine<CF> c{tile + lane * pitch};
        spline_line_body<CF>(c, c, n, 1, order, smode, gain_first, pw);
    }
    __syncthreads();
    // ---- out: coalesced rows
    for (int r = 0; r < nrows; r++)
        for (int x = lane; x < n; x += 64) base[(int64_t)r * n + x] = tile[r * pitch + x];
}

// ---------------------------------------------------------------------------
// r2: blocked prefilter for arrays with FEW, LONG lines (images: 4096 x 4096 has 4096 lines = 64 waves, each running
// two dependent sweeps of 4096 steps: 0.5 ms per axis, latency bound).  Orders 2 and 3 have one pole z, |z| < 0.27:
// the influence of a sample decays as |z|^k and is below 1e-22 of the data range after kSplHorizon = 40 steps, i.e.
// far below one ulp of a double.  A thread therefore filters one CHUNK of one line:
//   causal sweep from 40 samples before the chunk (started from the plain sample there; the first chunk uses the exact
//   boundary initialisation) to 40 samples past its end -- c+ of the chunk goes to the output, c+ of the 40 samples past
//   the end to an LDS column of the thread;
//   anti-causal sweep back from there (started with the steady-state value; the last chunk uses the exact boundary
//   initialisation), writing the chunk.
// Out of place (threads read source samples of neighbouring chunks); 1.3x the arithmetic, chunks x the parallelism.
// mirror / reflect initialisations only (grid-wrap needs c+ of the line start at the line end): smode 0 / 1.
// ---------------------------------------------------------------------------
constexpr int kSplHorizon = 40;

template <typename CF>
__global__ void __launch_bounds__(64)
spline_filter_chunked_kernel(const CF *__restrict__ src, CF *__restrict__ dst, int64_t n, int64_t inner, int64_t nlines, int order,
                             int smode, int64_t L, SplPow pw)
{
    __shared__ double ext[kSplHorizon][64];
    const int lane = threadIdx.x;
    int64_t line = (int64_t)blockIdx.x * 64 + lane;
    const bool valid = line < nlines;
    if (!valid) line = nlines - 1;                    // keeps the lane's loads in range; it stores nothing
    const int64_t line_off = (line / inner) * n * inner + (line % inner);
    const CF *__restrict__ rd = src + line_off;
    CF *__restrict__ wr = dst + line_off;
    const int64_t st = inner;
    const double z = order == 2 ? -0.171572875253809902396622551580603843 : -0.267949192431122706472553658494127633;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    const int64_t s0 = (int64_t)blockIdx.y * L, e0 = (s0 + L < n) ? s0 + L : n;
    const int64_t hi = (e0 + kSplHorizon < n) ? e0 + kSplHorizon : n;      // causal sweep runs to hi - 1

    // ---- causal start
    int64_t i0;
    double prev;
    if (s0 - kSplHorizon <= 0) {
        i0 = 0;
        const int64_t H = (int64_t)ceil(-46.0517 / log(fabs(z)));
        double c0 = (double)rd[0];
        double z_i = z;
        if (smode == 0) {
            const double z_n_1 = pw.zn1[0];
            double acc = c0 + z_n_1 * (double)rd[(n - 1) * st];
            const int64_t m = (n - 1 < H + 1) ? n - 1 : H + 1;
            for (int64_t i = 1; i < m; i++) { acc += z_i * ((double)rd[i * st] + z_n_1 * (double)rd[(n - 1 - i) * st]); z_i *= z; }
            c0 = acc / (1 - z_n_1 * z_n_1);
        } else {
            const double z_n = pw.zn[0];
            double acc = c0 + z_n * (double)rd[(n - 1) * st];
            const int64_t m = (n < H + 1) ? n : H + 1;
            for (int64_t i = 1; i < m; i++) {
                const double mirror_term = (i == n - 1) ? acc : (double)rd[(n - 1 - i) * st];
                acc += z_i * ((double)rd[i * st] + z_n * mirror_term);
                z_i *= z;
            }
            acc *= z / (1 - z_n * z_n);
            c0 = acc + c0;
        }
        prev = c0;
        if (valid && s0 == 0) wr[0] = (CF)prev;
    } else {
        i0 = s0 - kSplHorizon;
        prev = (double)rd[i0 * st];
    }
    double prev2 = 0.0;                               // c+[i - 1] before the last update: c+[n - 2] for the mirror end
    int64_t i = i0 + 1;
    // warm-up before the chunk
    for (; i + kSplBatch <= s0; i += kSplBatch) {
        double v[kSplBatch];
#pragma unroll
        for (int u = 0; u < kSplBatch; u++) v[u] = (double)rd[(i + u) * st];
#pragma unroll
        for (int u = 0; u < kSplBatch; u++) prev = v[u] + z * prev;
    }
    for (; i < s0; i++) prev = (double)rd[i * st] + z * prev;
    // the chunk
    for (; i + kSplBatch <= e0; i += kSplBatch) {
        double v[kSplBatch];
#pragma unroll
        for (int u = 0; u < kSplBatch; u++) v[u] = (double)rd[(i + u) * st];
#pragma unroll
        for (int u = 0; u < kSplBatch; u++) { prev2 = prev; prev = v[u] + z * prev; v[u] = prev; }
        if (valid) {
#pragma unroll
            for (int u = 0; u < kSplBatch; u++) wr[(i + u) * st] = (CF)v[u];
        }
    }
    for (; i < e0; i++) { prev2 = prev; prev = (double)rd[i * st] + z * prev; if (valid) wr[i * st] = (CF)prev; }
    // past the chunk: c+ into the LDS column
    for (; i < hi; i++) { prev2 = prev; prev = (double)rd[i * st] + z * prev; ext[i - e0][lane] = prev; }

    // ---- anti-causal start at hi - 1 (prev == c+[hi - 1])
    double nxt;
    if (hi == n) {
        // the sequential kernel reads c+[n - 2] back from the array, i.e. rounded to the coefficient type
        if (smode == 0) nxt = (z * (double)(CF)prev2 + prev) * z / (z * z - 1);
        else nxt = prev * z / (z - 1);
    } else {
        nxt = prev * z / (z - 1);
    }
    int64_t j = hi - 1;
    if (j >= e0) {
        // samples past the chunk (from the LDS column); nothing is stored
        for (j = hi - 2; j >= e0; j--) nxt = z * (nxt - ext[j - e0][lane]);
    } else {
        // the chunk ends the line: its last sample is the start value itself
        if (valid) wr[j * st] = (CF)(nxt * gain);
        j--;
    }
    for (; j - (kSplBatch - 1) >= s0; j -= kSplBatch) {
        double v[kSplBatch];
#pragma unroll
        for (int u = 0; u < kSplBatch; u++) v[u] = (double)wr[(j - u) * st];
#pragma unroll
        for (int u = 0; u < kSplBatch; u++) { nxt = z * (nxt - v[u]); v[u] = nxt; }
        if (valid) {
#pragma unroll
            for (int u = 0; u < kSplBatch; u++) wr[(j - u) * st] = (CF)(v[u] * gain);
        }
    }
    for (; j >= s0; j--) { nxt = z * (nxt - (double)wr[j * st]); if (valid) wr[j * st] = (CF)(nxt * gain); }
}

// The same blocked scheme for CONTIGUOUS lines (the last axis): with one lane per line every load of the kernel above
// touches 64 different cache lines (8192^2: 1010 us for this pass against 385 us for the other axis).  Here a wave owns
// 64 lines and one chunk of them and moves 64 x 64 tiles through LDS like spline_filter_rows_kernel: row-wise coalesced
// loads / stores, each lane filters its own line inside the tile.  Chunks are whole tiles, the warm-up before a chunk is
// the previous tile (64 samples), c+ of the kSplHorizon samples past the chunk goes to a second LDS array.
template <typename CF>
__global__ void __launch_bounds__(64)
spline_filter_rows_chunked_kernel(const CF *__restrict__ src, CF *__restrict__ dst, int64_t n, int64_t nlines, int order, int smode,
                                  int tiles_per_chunk, SplPow pw)
{
    __shared__ CF tile[kSplTile][kSplTile + 1];
    __shared__ double ext[kSplTile][kSplHorizon + 1];
    const int lane = threadIdx.x;
    const int64_t line0 = (int64_t)blockIdx.x * kSplTile;
    const int nrows = (int)((nlines - line0 < kSplTile) ? nlines - line0 : kSplTile);
    const bool valid = lane < nrows;
    const CF *sbase = src + line0 * n;
    CF *dbase = dst + line0 * n;
    const CF *__restrict__ rd = sbase + (int64_t)(valid ? lane : 0) * n;      // this lane's line (boundary sums only)
    const double z = order == 2 ? -0.171572875253809902396622551580603843 : -0.267949192431122706472553658494127633;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    const int ntiles = (int)((n + kSplTile - 1) / kSplTile);
    const int t0 = blockIdx.y * tiles_per_chunk;
    const int t1 = (t0 + tiles_per_chunk < ntiles) ? t0 + tiles_per_chunk : ntiles;
    auto tile_cnt = [&](int t) { const int64_t s0 = (int64_t)t * kSplTile; return (int)((n - s0 < kSplTile) ? n - s0 : kSplTile); };

    // ---- causal start
    double prev;
    if (t0 == 0) {
        const int64_t H = (int64_t)ceil(-46.0517 / log(fabs(z)));
        double c0 = (double)rd[0];
        double z_i = z;
        if (smode == 0) {
            const double z_n_1 = pw.zn1[0];
            double acc = c0 + z_n_1 * (double)rd[n - 1];
            const int64_t m = (n - 1 < H + 1) ? n - 1 : H + 1;
            for (int64_t i = 1; i < m; i++) { acc += z_i * ((double)rd[i] + z_n_1 * (double)rd[n - 1 - i]); z_i *= z; }
            c0 = acc / (1 - z_n_1 * z_n_1);
        } else {
            const double z_n = pw.zn[0];
            double acc = c0 + z_n * (double)rd[n - 1];
            const int64_t m = (n < H + 1) ? n : H + 1;
            for (int64_t i = 1; i < m; i++) {
                const double mirror_term = (i == n - 1) ? acc : (double)rd[n - 1 - i];
                acc += z_i * ((double)rd[i] + z_n * mirror_term);
                z_i *= z;
            }
            acc *= z / (1 - z_n * z_n);
            c0 = acc + c0;
        }
        prev = c0;
    } else {
        // warm-up over the tile before the chunk, started from the plain sample
        spline_tile_load(tile, sbase, n, (int64_t)(t0 - 1) * kSplTile, kSplTile, nrows, lane);
        __syncthreads();
        prev = (double)tile[lane][0];
        for (int k = 1; k < kSplTile; k++) prev = (double)tile[lane][k] + z * prev;
        __syncthreads();
    }
    double prev2 = 0.0;
    // ---- causal sweep over the chunk
    for (int t = t0; t < t1; t++) {
        const int64_t s0 = (int64_t)t * kSplTile;
        const int cnt = tile_cnt(t);
        spline_tile_load(tile, sbase, n, s0, cnt, nrows, lane);
        __syncthreads();
        if (valid) {
            for (int k = 0; k < cnt; k++) {
                double w;
                if (t == 0 && k == 0) w = prev;               // c+[0] is the boundary value itself
                else { w = (double)tile[lane][k] + z * prev; prev2 = prev; }
                prev = w;
                tile[lane][k] = (CF)w;
            }
        }
        __syncthreads();
        spline_tile_store(tile, dbase, n, s0, cnt, nrows, lane);
        __syncthreads();
    }
    // ---- c+ of the samples past the chunk (LDS only)
    int next = 0;
    if (t1 < ntiles) {
        const int cnt = tile_cnt(t1);
        next = cnt < kSplHorizon ? cnt : kSplHorizon;
        spline_tile_load(tile, sbase, n, (int64_t)t1 * kSplTile, cnt, nrows, lane);
        __syncthreads();
        for (int k = 0; k < next; k++) { prev2 = prev; prev = (double)tile[lane][k] + z * prev; ext[lane][k] = prev; }
        __syncthreads();
    }
    __threadfence();        // the c+ rows of a tile were stored by other lanes than the one that reads them back
    const int64_t hi = (int64_t)t1 * kSplTile + next < n ? (int64_t)t1 * kSplTile + next : n;
    // ---- anti-causal start at hi - 1
    double nxt;
    bool first = true;      // the start value is the value of sample hi - 1 itself when that sample ends the line
    if (hi == n) {
        if (smode == 0) nxt = (z * (double)(CF)prev2 + prev) * z / (z * z - 1);
        else nxt = prev * z / (z - 1);
    } else {
        nxt = prev * z / (z - 1);
    }
    // samples past the chunk
    for (int k = next - 1; k >= 0; k--) {
        if (first) { first = false; continue; }       // ext[next - 1] is sample hi - 1: its value is the start value
        nxt = z * (nxt - ext[lane][k]);
    }
    // ---- anti-causal sweep over the chunk
    for (int t = t1 - 1; t >= t0; t--) {
        const int64_t s0 = (int64_t)t * kSplTile;
        const int cnt = tile_cnt(t);
        spline_tile_load(tile, dbase, n, s0, cnt, nrows, lane);
        __syncthreads();
        if (valid) {
            for (int k = cnt - 1; k >= 0; k--) {
                if (first) first = false;             // sample hi - 1 == n - 1 inside the chunk: keeps the start value
                else nxt = z * (nxt - (double)tile[lane][k]);
                tile[lane][k] = (CF)(nxt * gain);
            }
        }
        __syncthreads();
        spline_tile_store(tile, dbase, n, s0, cnt, nrows, lane);
        __syncthreads();
    }
}

static int fill_geom(InterpGeom *g, const mi_array *in, int nd)
{
    const int pad = nd - in->ndim;
    g->pad = pad;
    for (int d = 0; d < pad; d++) { g->shape[d] = 1; g->stride[d] = 0; g->oshape[d] = 1; }
    int64_t st = 1;
    for (int d = in->ndim - 1; d >= 0; d--) {
        g->shape[pad + d] = in->shape[d];
        g->stride[pad + d] = st;
        st *= in->shape[d];
    }
    return MI_OK;
}

static int check_interp(const mi_array *in, const mi_array *out, int order, int mode)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1, MI_ERR_INVALID_ARG, "input must have at least one dimension");
    if (order < 0 || order > 5) { set_error("spline order is not supported"); return MI_ERR_INVALID_ARG; }
    MI_REQUIRE(mode >= MI_MODE_REFLECT && mode <= MI_MODE_GRID_CONSTANT, MI_ERR_INVALID_ARG,
               "boundary mode is not supported");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS,
               "interpolation needs C-contiguous arrays");
    for (int d = 0; d < in->ndim; d++)
        MI_REQUIRE(in->shape[d] > 0, MI_ERR_INVALID_ARG, "input has an empty axis");
    return MI_OK;
}

// float32 3-D throughput kernels (interp_fast.hip); MI_ERR_UNSUPPORTED = not covered
int map_coordinates_fast(const mi_array *in, const mi_array *coords, const mi_array *out, int order, int mode,
                         double cval, hipStream_t s);
int affine_transform_fast(const mi_array *in, const mi_array *out, const double *matrix, int order, int mode,
                          double cval, hipStream_t s);

}  // namespace mi

using namespace mi;

static mi::Knob g_interp_generic{0};   // test hook: 1 = always use the generic double kernels
static mi::Knob g_spline_gain_first{1};     // test hook: 0 = the one-thread-per-line kernel applies the gain with its last store
extern "C" int mi_debug_set_spline_gain_first(int k) { g_spline_gain_first = k; return MI_OK; }
static mi::Knob g_spline_threads{65536};   // threads the blocked prefilter aims for
extern "C" int mi_debug_set_spline_threads(int k) { g_spline_threads = k; return MI_OK; }
static mi::Knob g_spline_chunk{0};     // test hook: -1 = never the blocked prefilter, > 0 = force it with this minimum chunk length
extern "C" int mi_debug_set_spline_chunk(int k) { g_spline_chunk = k; return MI_OK; }
static mi::Knob g_spline_rows_off{0};  // test hook: 1 = one thread per line also for contiguous lines
static mi::Knob g_cubic_separable_off{0};   // test hook: 1 = diagonal transforms use the one-launch strip kernel
extern "C" int mi_debug_set_cubic_separable(int on) { g_cubic_separable_off = !on; return MI_OK; }
static mi::Knob g_cubic_diag_off{0};    // test hook: 1 = diagonal transforms use the gather kernel too
extern "C" int mi_debug_set_cubic_diag(int on) { g_cubic_diag_off = !on; return MI_OK; }
static mi::Knob g_spline_rows_force{0}; // test hook: 2 = tiled kernel whatever the line count
static mi::Knob g_spline_rows_lds{1};   // test hook: 0 = never the LDS-resident lines kernel (the tiled kernel instead)
extern "C" int mi_debug_set_spline_rows_lds(int on) { g_spline_rows_lds = on; return MI_OK; }
extern "C" int mi_debug_set_spline_rows(int on) { g_spline_rows_off = on == 0; g_spline_rows_force = on == 2; return MI_OK; }
extern "C" int mi_debug_set_interp_generic(int v) { g_interp_generic = v; return MI_OK; }

extern "C" {

int mi_map_coordinates(const mi_array *in, const mi_array *coords, const mi_array *out, int order,
                       int mode, double cval, mi_stream stream)
{
    int rc = check_interp(in, out, order, mode);
    if (rc) return rc;
    if (order > 1 && in->ndim > 3) { set_error("orders 2-5 on rank > 3 arrays: mi_spline_map_coordinates (on coefficients)"); return MI_ERR_UNSUPPORTED; }
    if ((rc = check_array(coords, "coordinates"))) return rc;
    MI_REQUIRE(coords->dtype == MI_F32 || coords->dtype == MI_F64, MI_ERR_INVALID_ARG,
               "coordinates should have floating point dtype");
    MI_REQUIRE(coords->ndim == out->ndim + 1 && coords->shape[0] == in->ndim, MI_ERR_INVALID_ARG,
               "invalid shape for coordinate array");
    for (int d = 0; d < out->ndim; d++)
        MI_REQUIRE(coords->shape[d + 1] == out->shape[d], MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(is_contiguous(coords), MI_ERR_NOT_CONTIGUOUS, "coordinates must be C-contiguous");
    const int64_t nout = numel(out);
    if (nout == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    if (!g_interp_generic) {
        rc = map_coordinates_fast(in, coords, out, order, mode, cval, s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    const int nd = in->ndim <= 3 ? 3 : MI_MAX_NDIM;
    InterpGeom g;
    fill_geom(&g, in, nd);
    const int round_out = out->dtype != MI_F32 && out->dtype != MI_F64 && out->dtype != MI_BOOL;
    dim3 grid;
    grid_for(nout, 256, &grid);
    if (nd != 3 && in->dtype != MI_F32 && in->dtype != MI_F64) {
        set_error("rank > 3 interpolation is built for float32/float64 input only");
        return MI_ERR_UNSUPPORTED;   // host converts the input to float64 (exact) and retries
    }
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        const T *ip = (const T *)in->data;
        constexpr bool kHighRank = std::is_floating_point<T>::value;
#define MI_LAUNCH(C, NDV)                                                                              \
    hipLaunchKernelGGL((map_coordinates_kernel<T, C, NDV>), grid, dim3(256), 0, s, ip,                 \
                       (const C *)coords->data, out->data, out->dtype, g, nout, order, mode, cval, round_out, 0)
        if (nd == 3) { if (coords->dtype == MI_F32) MI_LAUNCH(float, 3); else MI_LAUNCH(double, 3); }
        else if constexpr (kHighRank) {
            if (coords->dtype == MI_F32) MI_LAUNCH(float, MI_MAX_NDIM); else MI_LAUNCH(double, MI_MAX_NDIM);
        }
#undef MI_LAUNCH
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

int mi_affine_transform(const mi_array *in, const mi_array *out, const double *matrix, int order,
                        int mode, double cval, mi_stream stream)
{
    int rc = check_interp(in, out, order, mode);
    if (rc) return rc;
    if (order > 1 && in->ndim > 3) { set_error("orders 2-5 on rank > 3 arrays: mi_spline_affine_transform (on coefficients)"); return MI_ERR_UNSUPPORTED; }
    MI_REQUIRE(matrix, MI_ERR_INVALID_ARG, "matrix is NULL");
    MI_REQUIRE(out->ndim == in->ndim, MI_ERR_INVALID_ARG, "output rank must equal input rank");
    const int64_t nout = numel(out);
    if (nout == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    if (!g_interp_generic && (in->ndim == 3 || in->ndim == 2)) {
        rc = affine_transform_fast(in, out, matrix, order, mode, cval, s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    const int n = in->ndim;
    const int nd = n <= 3 ? 3 : MI_MAX_NDIM;
    InterpGeom g;
    fill_geom(&g, in, nd);
    for (int d = 0; d < n; d++) g.oshape[g.pad + d] = out->shape[d];
    // padded matrix: identity-free zero rows for the unit axes (c = 0 there)
    for (int i = 0; i < nd * (nd + 1); i++) g.mat[i] = 0.0;
    for (int d = 0; d < n; d++) {
        for (int k = 0; k < n; k++) g.mat[(g.pad + d) * (nd + 1) + g.pad + k] = matrix[d * (n + 1) + k];
        g.mat[(g.pad + d) * (nd + 1) + nd] = matrix[d * (n + 1) + n];
    }
    const int round_out = out->dtype != MI_F32 && out->dtype != MI_F64 && out->dtype != MI_BOOL;
    dim3 grid;
    grid_for(nout, 256, &grid);
    if (nd != 3 && in->dtype != MI_F32 && in->dtype != MI_F64) {
        set_error("rank > 3 interpolation is built for float32/float64 input only");
        return MI_ERR_UNSUPPORTED;
    }
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        const T *ip = (const T *)in->data;
        if (nd == 3)
            hipLaunchKernelGGL((affine_kernel<T, 3>), grid, dim3(256), 0, s, ip, out->data, out->dtype, g, nout,
                               order, mode, cval, round_out, 0);
        else if constexpr (std::is_floating_point<T>::value)
            hipLaunchKernelGGL((affine_kernel<T, MI_MAX_NDIM>), grid, dim3(256), 0, s, ip, out->data, out->dtype,
                               g, nout, order, mode, cval, round_out, 0);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

/* ---- B-spline orders 2..5 (declared in include/mi355img.h) ---- */
int mi_spline_pad(const mi_array *in, const mi_array *out, int npad, int pad_mode, double cval, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1 && in->ndim <= MI_MAX_NDIM && out->ndim == in->ndim, MI_ERR_INVALID_ARG, "rank 1..8");
    MI_REQUIRE(out->dtype == MI_F64 || out->dtype == MI_F32, MI_ERR_INVALID_ARG, "coefficients are float64 or float32");
    MI_REQUIRE(npad >= 0 && (pad_mode == 0 || pad_mode == 1), MI_ERR_INVALID_ARG, "bad padding");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS, "needs C-contiguous arrays");
    for (int d = 0; d < in->ndim; d++)
        MI_REQUIRE(out->shape[d] == in->shape[d] + 2 * npad && in->shape[d] > 0, MI_ERR_INVALID_ARG, "output shape is not correct");
    const int nd = in->ndim <= 3 ? 3 : MI_MAX_NDIM;
    InterpGeom g;
    fill_geom(&g, in, nd);
    for (int d = 0; d < nd; d++) g.oshape[d] = 1;
    for (int d = 0; d < in->ndim; d++) g.oshape[g.pad + d] = out->shape[d];
    const int64_t nout = numel(out);
    dim3 grid;
    grid_for(nout, 256, &grid);
    hipStream_t s = resolve_stream(stream);
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        if (nd != 3) {
            // rank 4 .. 8 (r3): same kernel over eight (padded) axes, float64 coefficients only
            if (out->dtype != MI_F64) { set_error("float32 spline coefficients are built for rank <= 3"); return MI_ERR_UNSUPPORTED; }
            hipLaunchKernelGGL((spline_pad_kernel<T, double, MI_MAX_NDIM>), grid, dim3(256), 0, s, (const T *)in->data,
                               (double *)out->data, g, nout, npad, pad_mode, cval);
        } else if (out->dtype == MI_F64)
            hipLaunchKernelGGL((spline_pad_kernel<T, double>), grid, dim3(256), 0, s, (const T *)in->data, (double *)out->data,
                               g, nout, npad, pad_mode, cval);
        else
            hipLaunchKernelGGL((spline_pad_kernel<T, float>), grid, dim3(256), 0, s, (const T *)in->data, (float *)out->data,
                               g, nout, npad, pad_mode, cval);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

// chunk length of the blocked prefilter for one pass, 0 = the pass does not qualify (few long lines; orders 2 / 3;
// mirror / reflect ends)
static int64_t spline_chunk_len(const mi_array *a, int axis, int order, int spline_mode)
{
    if (spline_mode & kSplExact) return 0;      // bit-exact request: never the blocked (restarted) recursion
    const int64_t total = numel(a);
    if (total == 0 || a->shape[axis] <= 1) return 0;
    const int64_t n = a->shape[axis], nlines = total / n;
    const int64_t min_chunk = g_spline_chunk > 0 ? g_spline_chunk : 128;
    if (!(g_spline_chunk >= 0 && order <= 3 && spline_mode != 2 && n >= 2 * min_chunk && n > 2 * kSplHorizon &&
          (nlines < 32768 || g_spline_chunk > 0)))
        return 0;
    // as many chunks as it takes to get ~64k threads, at least min_chunk samples each
    int64_t nch = (g_spline_threads + nlines - 1) / nlines;
    if (nch > n / min_chunk) nch = n / min_chunk;
    return nch >= 2 ? (n + nch - 1) / nch : 0;
}

// one prefilter pass along `axis`: `shape` describes the (contiguous) array, the samples are read from `src` and the
// coefficients written to `dst` (src == dst: in place).  The blocked kernel works out of place (an in-place request
// goes through a temporary and is copied back); the LDS-tiled kernel for contiguous lines works in place only.
static int spline_pass(const mi_array *shape, const void *src, void *dst, int axis, int order, int spline_mode, hipStream_t s)
{
    const int64_t total = numel(shape);
    if (total == 0) return MI_OK;
    // r5: volumes (many lines, single-pole orders, mirror / reflect ends, no bit-exactness request) at one memory sweep per
    // axis -- csrc/spline_fast.hip; the blocked kernels below keep the images (few long lines)
    if (spline_chunk_len(shape, axis, order, spline_mode) == 0) {
        int frc;
        if (spline_pass_fast(shape, src, shape->dtype, dst, axis, order, spline_mode, s, &frc)) return frc;
    }
    int64_t inner = 1;
    for (int d = axis + 1; d < shape->ndim; d++) inner *= shape->shape[d];
    const int64_t n = shape->shape[axis], nlines = total / n;
    const int64_t L = spline_chunk_len(shape, axis, order, spline_mode);
    // grid-wrap in the tiled rows kernel: the anti-causal start adds its boundary sum to c[n-1] in one piece, SciPy term
    // by term -- equal to an ulp, not to the bit
    const bool rows_ok = !((spline_mode & kSplExact) && (spline_mode & 0xff) == 2);
    spline_mode &= 0xff;
    SplPow pw;
    {
        static const double poles[4][2] = {{-0.171572875253809902396622551580603843, 0.0},
                                           {-0.267949192431122706472553658494127633, 0.0},
                                           {-0.361341225900220177092212841325675255, -0.013725429297339121360331226939128204},
                                           {-0.430575347099973791851434783493520110, -0.043096288203264653822712376822550182}};
        for (int k = 0; k < 2; k++) {
            const double z = poles[order - 2][k];
            pw.zn1[k] = z != 0.0 ? pow(z, (double)(n - 1)) : 0.0;
            pw.zn[k] = z != 0.0 ? pow(z, (double)n) : 0.0;
        }
    }
    if (L > 0) {
        const unsigned gy = (unsigned)((n + L - 1) / L);
        const size_t bytes = (size_t)total * dtype_size(shape->dtype);
        void *tmp = nullptr;
        if (src == dst) {
            int rc = pool_alloc(&tmp, bytes, s);
            if (rc) return rc;
        }
        void *to = tmp ? tmp : dst;
        if (inner == 1 && n >= 4 * kSplTile && !g_spline_rows_off) {
            // contiguous lines: whole 64-sample tiles per chunk through LDS (coalesced)
            const int tpc = (int)((L + kSplTile - 1) / kSplTile);
            const int ntiles = (int)((n + kSplTile - 1) / kSplTile);
            const dim3 grid((unsigned)((nlines + kSplTile - 1) / kSplTile), (unsigned)((ntiles + tpc - 1) / tpc));
            if (shape->dtype == MI_F64)
                hipLaunchKernelGGL(spline_filter_rows_chunked_kernel<double>, grid, dim3(64), 0, s, (const double *)src, (double *)to, n,
                                   nlines, order, spline_mode, tpc, pw);
            else
                hipLaunchKernelGGL(spline_filter_rows_chunked_kernel<float>, grid, dim3(64), 0, s, (const float *)src, (float *)to, n,
                                   nlines, order, spline_mode, tpc, pw);
            hipError_t err = hipGetLastError();
            if (tmp) {
                if (err == hipSuccess) err = hipMemcpyAsync(dst, tmp, bytes, hipMemcpyDeviceToDevice, s);
                pool_free(tmp);
            }
            MI_HIP(err);
            return MI_OK;
        }
        const dim3 grid((unsigned)((nlines + 63) / 64), gy);
        if (shape->dtype == MI_F64)
            hipLaunchKernelGGL(spline_filter_chunked_kernel<double>, grid, dim3(64), 0, s, (const double *)src, (double *)to, n, inner,
                               nlines, order, spline_mode, L, pw);
        else
            hipLaunchKernelGGL(spline_filter_chunked_kernel<float>, grid, dim3(64), 0, s, (const float *)src, (float *)to, n, inner,
                               nlines, order, spline_mode, L, pw);
        hipError_t err = hipGetLastError();
        if (tmp) {
            if (err == hipSuccess) err = hipMemcpyAsync(dst, tmp, bytes, hipMemcpyDeviceToDevice, s);
            pool_free(tmp);            // reuse is stream ordered
        }
        MI_HIP(err);
        return MI_OK;
    }
    // enough lines to fill the chip with one wave per 64 lines; images with few, long lines keep one thread per line
    if (src == dst && inner == 1 && n >= 2 * kSplTile && (nlines >= 16384 || g_spline_rows_force) && !g_spline_rows_off && g_spline_rows_lds) {
        const size_t esz = shape->dtype == MI_F64 ? 8 : 4;
        const size_t cap = 72 * 1024;                                   // two waves per CU
        // lines per wave: TWELVE (512 float32 coefficients: 24 KiB, six waves per CU) -- the phases of a wave (loads in
        // flight, a latency-bound recursion, stores) overlap with those of its neighbours on the CU, not with each other:
        // 2.26 / 2.22 / 2.31 / 2.36 / 2.29 / 2.42 ms for a default `rotate` of 512^3 at 8 / 12 / 16 / 20 / 24 / 32 lines,
        // 2.65 ms with the tiled kernel (profiles/r4_cubic_zstream.txt)
        int lpb = (int)std::min<size_t>(12, cap / ((size_t)(n + 1) * esz));
        if (lpb < 4) lpb = 0;
        if (g_spline_rows_lds >= 4 && g_spline_rows_lds <= 64 && (size_t)g_spline_rows_lds * (size_t)(n + 1) * esz <= cap) lpb = g_spline_rows_lds;
        if (lpb && n < (1 << 20)) {
            const size_t lds = (size_t)lpb * (size_t)(n + 1) * esz;
            static PerDeviceOnce attr_done;
            if (!attr_done) {
                MI_HIP(hipFuncSetAttribute((const void *)spline_filter_rows_lds_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap));
                MI_HIP(hipFuncSetAttribute((const void *)spline_filter_rows_lds_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap));
                attr_done = true;
            }
            const dim3 grid((unsigned)((nlines + lpb - 1) / lpb));
            if (shape->dtype == MI_F64)
                hipLaunchKernelGGL(spline_filter_rows_lds_kernel<double>, grid, dim3(64), lds, s, (double *)dst, (int)n, nlines, lpb, order, spline_mode, g_spline_gain_first, pw);
            else
                hipLaunchKernelGGL(spline_filter_rows_lds_kernel<float>, grid, dim3(64), lds, s, (float *)dst, (int)n, nlines, lpb, order, spline_mode, g_spline_gain_first, pw);
            MI_HIP(hipGetLastError());
            return MI_OK;
        }
    }
    if (src == dst && inner == 1 && n >= 2 * kSplTile && (nlines >= 16384 || g_spline_rows_force) && !g_spline_rows_off && rows_ok) {
        const dim3 grid((unsigned)((nlines + kSplTile - 1) / kSplTile));
        if (shape->dtype == MI_F64)
            hipLaunchKernelGGL(spline_filter_rows_kernel<double>, grid, dim3(64), 0, s, (double *)dst, n, nlines, order, spline_mode, pw);
        else
            hipLaunchKernelGGL(spline_filter_rows_kernel<float>, grid, dim3(64), 0, s, (float *)dst, n, nlines, order, spline_mode, pw);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    const void *from = src == dst ? nullptr : src;
    const dim3 grid((unsigned)((nlines + 63) / 64));
    if (shape->dtype == MI_F64)
        hipLaunchKernelGGL(spline_filter1d_kernel<double>, grid, dim3(64), 0, s, (double *)dst, (const double *)from, n, inner, nlines,
                           order, spline_mode, g_spline_gain_first, pw);
    else
        hipLaunchKernelGGL(spline_filter1d_kernel<float>, grid, dim3(64), 0, s, (float *)dst, (const float *)from, n, inner, nlines,
                           order, spline_mode, g_spline_gain_first, pw);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_spline_filter1d(const mi_array *data, int axis, int order, int spline_mode, mi_stream stream)
{
    int rc;
    if ((rc = check_array(data, "data"))) return rc;
    MI_REQUIRE(data->dtype == MI_F64 || data->dtype == MI_F32, MI_ERR_INVALID_ARG, "coefficients are float64 or float32");
    MI_REQUIRE(data->ndim >= 1 && axis >= 0 && axis < data->ndim, MI_ERR_INVALID_ARG, "invalid axis");
    MI_REQUIRE(order >= 2 && order <= 5, MI_ERR_INVALID_ARG, "spline order is not supported");
    MI_REQUIRE((spline_mode & 0xff) >= 0 && (spline_mode & 0xff) <= 2 && (spline_mode & ~(0xff | kSplExact)) == 0, MI_ERR_INVALID_ARG,
               "bad spline boundary mode");
    MI_REQUIRE(is_contiguous(data), MI_ERR_NOT_CONTIGUOUS, "needs a C-contiguous array");
    return spline_pass(data, data->data, data->data, axis, order, spline_mode, resolve_stream(stream));
}

int mi_spline_prefilter(const mi_array *in, const mi_array *out, int order, int spline_mode, int npad, int pad_mode,
                        double cval, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(order >= 2 && order <= 5, MI_ERR_INVALID_ARG, "spline order is not supported");
    MI_REQUIRE((spline_mode & 0xff) >= 0 && (spline_mode & 0xff) <= 2 && (spline_mode & ~(0xff | kSplExact | (0xff << 9))) == 0, MI_ERR_INVALID_ARG,
               "bad spline boundary mode");
    MI_REQUIRE(out->dtype == MI_F64 || out->dtype == MI_F32, MI_ERR_INVALID_ARG, "coefficients are float64 or float32");
    // r5: bit 9 + d = leave axis d unfiltered (MI_SPLINE_SKIP_AXIS(d)): an affine transform that maps an axis onto itself with
    // an integral shift evaluates the spline at the samples of that axis, where it returns them -- the pass and the taps
    // along it cancel (SciPy's own `rotate` never filters the axes outside the rotation plane: ndimage/_interpolation.py)
    const int skip = (spline_mode >> 9) & 0xff;
    spline_mode &= 0xff | kSplExact;
    auto filtered = [&](int d) { return out->shape[d] > 1 && !((skip >> d) & 1); };
    MI_REQUIRE(in->data != out->data, MI_ERR_INVALID_ARG, "mi_spline_prefilter works out of place");
    // the first filtered axis can read the source directly when no padding or conversion is asked for
    int first = -1;
    for (int d = 0; d < out->ndim && first < 0; d++)
        if (filtered(d)) first = d;
    int64_t inner_first = 1;
    for (int d = first + 1; first >= 0 && d < out->ndim; d++) inner_first *= out->shape[d];
    const bool direct = npad == 0 && in->dtype == out->dtype && same_shape(in, out) && is_contiguous(in) && is_contiguous(out)
                        && first >= 0 && inner_first > 1;
    hipStream_t s = resolve_stream(stream);
    // r5: float32 samples -> float64 coefficients (the public spline_filter's default output) without the conversion copy:
    // the streaming kernel of the first axis reads the float32 input itself; the other axes follow in place
    if (!direct && npad == 0 && in->dtype == MI_F32 && out->dtype == MI_F64 && same_shape(in, out) && is_contiguous(in) && is_contiguous(out)
        && first >= 0 && inner_first > 1 && spline_chunk_len(out, first, order, spline_mode) == 0) {
        int frc;
        if (spline_pass_fast(out, in->data, MI_F32, out->data, first, order, spline_mode, s, &frc)) {
            for (int d = first + 1; d < out->ndim && frc == MI_OK; d++)
                if (filtered(d)) frc = spline_pass(out, out->data, out->data, d, order, spline_mode, s);
            return frc;
        }
    }
    // Passes of the blocked kernel work out of place, the others in place: the data hops between `out` and one
    // temporary, laid out so that the last hop lands in `out` (K blocked passes: start in `out` when K is even).
    int nblocked = 0;
    for (int d = 0; d < out->ndim; d++)
        if (filtered(d) && spline_chunk_len(out, d, order, spline_mode) > 0) nblocked++;
    void *tmp = nullptr;
    if (nblocked > 0 && (rc = pool_alloc(&tmp, (size_t)numel(out) * dtype_size(out->dtype), s))) return rc;
    auto other = [&](const void *p) { return p == out->data ? tmp : out->data; };
    const void *cur;
    if (direct) {
        cur = in->data;
    } else {
        mi_array first_home = *out;
        first_home.data = (nblocked % 2 == 0) ? out->data : tmp;
        if ((rc = mi_spline_pad(in, &first_home, npad, pad_mode, cval, stream))) { if (tmp) pool_free(tmp); return rc; }
        cur = first_home.data;
    }
    int left = nblocked;
    for (int d = 0; d < out->ndim && rc == MI_OK; d++) {
        if (!filtered(d)) continue;
        const bool blocked = spline_chunk_len(out, d, order, spline_mode) > 0;
        void *dst;
        if (blocked) {
            left--;
            dst = cur == in->data ? ((left % 2 == 0) ? out->data : tmp) : other(cur);
        } else {
            dst = cur == in->data ? ((left % 2 == 0) ? out->data : tmp) : const_cast<void *>(cur);
        }
        rc = spline_pass(out, cur, dst, d, order, spline_mode, s);
        cur = dst;
    }
    if (rc == MI_OK && cur == in->data) {
        // nothing was filtered (every axis skipped or of length one): the coefficients are the samples
        hipError_t err = hipMemcpyAsync(out->data, in->data, (size_t)numel(out) * dtype_size(out->dtype), hipMemcpyDeviceToDevice, s);
        if (err != hipSuccess) rc = (int)err;
    }
    if (rc == MI_OK && cur != out->data && cur != in->data) {
        hipError_t err = hipMemcpyAsync(out->data, cur, (size_t)numel(out) * dtype_size(out->dtype), hipMemcpyDeviceToDevice, s);
        if (err != hipSuccess) rc = (int)err;
    }
    if (tmp) pool_free(tmp);           // reuse is stream ordered
    return rc;
}

static int check_spline(const mi_array *coef, const mi_array *out, int order, int mode, int npad)
{
    int rc = check_interp(coef, out, order, mode);
    if (rc) return rc;
    MI_REQUIRE(order >= 2, MI_ERR_INVALID_ARG, "orders 0 and 1 use mi_map_coordinates / mi_affine_transform");
    MI_REQUIRE(coef->dtype == MI_F64 || (coef->dtype == MI_F32 && order == 3 && out->dtype == MI_F32), MI_ERR_INVALID_ARG,
               "coefficients are float64 (float32 only for order 3 with a float32 output)");
    MI_REQUIRE(npad >= 0, MI_ERR_INVALID_ARG, "negative padding");
    if (coef->ndim > 3 && coef->dtype != MI_F64) { set_error("float32 spline coefficients are built for rank <= 3"); return MI_ERR_UNSUPPORTED; }
    return MI_OK;
}

int mi_spline_map_coordinates(const mi_array *coef, const mi_array *coords, const mi_array *out, int order, int mode,
                              double cval, int npad, mi_stream stream)
{
    int rc = check_spline(coef, out, order, mode, npad);
    if (rc) return rc;
    if ((rc = check_array(coords, "coordinates"))) return rc;
    MI_REQUIRE(coords->dtype == MI_F32 || coords->dtype == MI_F64, MI_ERR_INVALID_ARG,
               "coordinates should have floating point dtype");
    MI_REQUIRE(coords->ndim == out->ndim + 1 && coords->shape[0] == coef->ndim, MI_ERR_INVALID_ARG,
               "invalid shape for coordinate array");
    for (int d = 0; d < out->ndim; d++)
        MI_REQUIRE(coords->shape[d + 1] == out->shape[d], MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(is_contiguous(coords), MI_ERR_NOT_CONTIGUOUS, "coordinates must be C-contiguous");
    const int64_t nout = numel(out);
    if (nout == 0) return MI_OK;
    InterpGeom g;
    fill_geom(&g, coef, coef->ndim <= 3 ? 3 : MI_MAX_NDIM);
    const int round_out = out->dtype != MI_F32 && out->dtype != MI_F64 && out->dtype != MI_BOOL;
    dim3 grid;
    grid_for(nout, 256, &grid);
    hipStream_t s = resolve_stream(stream);
    if (coef->ndim > 3) {
        if (coords->dtype == MI_F32)
            hipLaunchKernelGGL((spline_map_nd_kernel<float>), grid, dim3(256), 0, s, (const double *)coef->data, (const float *)coords->data,
                               out->data, out->dtype, g, nout, order, mode, cval, round_out, npad);
        else
            hipLaunchKernelGGL((spline_map_nd_kernel<double>), grid, dim3(256), 0, s, (const double *)coef->data, (const double *)coords->data,
                               out->data, out->dtype, g, nout, order, mode, cval, round_out, npad);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    if (coef->dtype == MI_F32) {
        // float32 coefficients: the cubic gather kernel (32-bit element offsets)
        MI_REQUIRE(numel(coef) < ((int64_t)1 << 29), MI_ERR_UNSUPPORTED, "float32 coefficient volume too large");
        MI_REQUIRE(is_contiguous(out), MI_ERR_NOT_CONTIGUOUS, "needs a C-contiguous output");
        MI_REQUIRE(out->ndim >= 1 && out->ndim <= 3, MI_ERR_UNSUPPORTED, "float32 cubic route: output rank 1..3");
        dim3 cgrid;
        MI_REQUIRE(cubic3_grid(out, &g, &cgrid), MI_ERR_UNSUPPORTED, "float32 cubic route: output too large");
#define MI_CUBIC_MAP(C, NTZ, NTY)                                                                                     \
    hipLaunchKernelGGL((cubic3_f32_kernel<C, false, NTZ, NTY>), cgrid, dim3(64, 4), 0, s, (const float *)coef->data,   \
                       (const C *)coords->data, (float *)out->data, g, nout, numel(coef), mode, (float)cval, npad)
#define MI_CUBIC_MAP_RANK(C)                                                   \
    if (g.pad == 0) MI_CUBIC_MAP(C, 4, 4);                                     \
    else if (g.pad == 1) MI_CUBIC_MAP(C, 1, 4);                                \
    else MI_CUBIC_MAP(C, 1, 1)
        // r5: volumes take their taps out of a box the workgroup sizes from its own coordinates (csrc/cubic_fast.hip)
        if (g_cubic_box && g.pad == 0 && out->ndim == 3) {
            const int shp[3] = {(int)g.shape[0], (int)g.shape[1], (int)g.shape[2]}, osh[3] = {(int)out->shape[0], (int)out->shape[1], (int)out->shape[2]};
            int mrc = MI_OK;
            if (launch_cubic_mapbox((const float *)coef->data, coords->data, coords->dtype == MI_F64, (float *)out->data, shp, osh, mode, cval, npad, s, &mrc, g_cubic_box))
                return mrc;
        }
        note_kernel("mi::cubic3_f32_kernel (order-3 map_coordinates on float32 coefficients: 16 x 16-byte gathers per voxel)");
        if (coords->dtype == MI_F32) { MI_CUBIC_MAP_RANK(float); } else { MI_CUBIC_MAP_RANK(double); }
#undef MI_CUBIC_MAP_RANK
#undef MI_CUBIC_MAP
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
#define MI_SPL_MAP(C, ORD)                                                                                         \
    hipLaunchKernelGGL((spline_map_kernel<C, ORD>), grid, dim3(256), 0, s, (const double *)coef->data,            \
                       (const C *)coords->data, out->data, out->dtype, g, nout, mode, cval, round_out, npad)
#define MI_SPL_MAP_ORD(C)                                                              \
    switch (order) {                                                                   \
    case 2: MI_SPL_MAP(C, 2); break;                                                   \
    case 3: MI_SPL_MAP(C, 3); break;                                                   \
    case 4: MI_SPL_MAP(C, 4); break;                                                   \
    default: MI_SPL_MAP(C, 5); break;                                                  \
    }
    if (coords->dtype == MI_F32) { MI_SPL_MAP_ORD(float) } else { MI_SPL_MAP_ORD(double) }
#undef MI_SPL_MAP_ORD
#undef MI_SPL_MAP
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_spline_affine_transform(const mi_array *coef, const mi_array *out, const double *matrix, int order, int mode,
                               double cval, int npad, mi_stream stream)
{
    // r5: order | MI_SPLINE_SAMPLES_AXIS(d) (bit 8 + d): axis d of `coef` holds SAMPLES (mi_spline_prefilter skipped it) and the
    // matrix maps it onto itself with an integral shift.  Only the kernels that evaluate such an axis as the single tap it is
    // take the call (order 3, float32 coefficients, one such axis); anything else answers MI_ERR_UNSUPPORTED and the caller
    // filters the axis after all (mi_spline_filter1d: the passes commute) and calls again without the flag.
    const int ident = (order >> 8) & 0xff;
    order &= 0xff;
    int rc = check_spline(coef, out, order, mode, npad);
    if (rc) return rc;
    MI_REQUIRE(matrix, MI_ERR_INVALID_ARG, "matrix is NULL");
    MI_REQUIRE(out->ndim == coef->ndim, MI_ERR_INVALID_ARG, "output rank must equal input rank");
    const int64_t nout = numel(out);
    if (nout == 0) return MI_OK;
    const int n = coef->ndim, nd = n <= 3 ? 3 : MI_MAX_NDIM;
    InterpGeom g;
    fill_geom(&g, coef, nd);
    for (int d = 0; d < nd; d++) g.oshape[d] = 1;
    for (int d = 0; d < n; d++) g.oshape[g.pad + d] = out->shape[d];
    for (int i = 0; i < nd * (nd + 1); i++) g.mat[i] = 0.0;
    for (int d = 0; d < n; d++) {
        for (int k = 0; k < n; k++) g.mat[(g.pad + d) * (nd + 1) + g.pad + k] = matrix[d * (n + 1) + k];
        g.mat[(g.pad + d) * (nd + 1) + nd] = matrix[d * (n + 1) + n];
    }
    const int round_out = out->dtype != MI_F32 && out->dtype != MI_F64 && out->dtype != MI_BOOL;
    dim3 grid;
    grid_for(nout, 256, &grid);
    hipStream_t s = resolve_stream(stream);
    if (ident && !(n == 3 && coef->dtype == MI_F32 && order == 3 && (ident == 1 || ident == 2 || ident == 4))) {
        set_error("no kernel for this transform with unfiltered axes (mask %d)", ident);
        return MI_ERR_UNSUPPORTED;
    }
    if (n > 3) {
        hipLaunchKernelGGL(spline_affine_nd_kernel, grid, dim3(256), 0, s, (const double *)coef->data, out->data, out->dtype, g,
                           nout, order, mode, cval, round_out, npad);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    if (coef->dtype == MI_F32) {
        MI_REQUIRE(numel(coef) < ((int64_t)1 << 29), MI_ERR_UNSUPPORTED, "float32 coefficient volume too large");
        MI_REQUIRE(is_contiguous(out), MI_ERR_NOT_CONTIGUOUS, "needs a C-contiguous output");
        dim3 cgrid;
        MI_REQUIRE(cubic3_grid(out, &g, &cgrid), MI_ERR_UNSUPPORTED, "float32 cubic route: output too large");
        bool diagonal = !g_cubic_diag_off && !ident;
        for (int d = 0; d < n; d++)
            for (int k = 0; k < n; k++)
                if (d != k && matrix[d * (n + 1) + k] != 0.0) diagonal = false;
        void *tab = nullptr;
        if (diagonal) {
            const int64_t entries = g.oshape[0] + g.oshape[1] + g.oshape[2];
            int64_t longest = g.oshape[0] > g.oshape[1] ? g.oshape[0] : g.oshape[1];
            if (g.oshape[2] > longest) longest = g.oshape[2];
            if ((rc = pool_alloc(&tab, (size_t)entries * sizeof(AxisTaps), s))) return rc;
            const bool separable = !g_cubic_separable_off && g.oshape[0] <= 65535 && (g.oshape[1] + 3) / 4 <= 65535 &&
                                   g.shape[0] <= 65535 && (g.shape[1] + 3) / 4 <= 65535;
            hipLaunchKernelGGL(cubic3_axis_table_kernel, dim3((unsigned)((longest + 255) / 256), 3), dim3(256), 0, s,
                               (AxisTaps *)tab, g, mode, npad, separable ? 1 : 0);
            if (separable) {
                // x, then y, then z; rank-padding axes have nothing to resample
                const AxisTaps *T = (const AxisTaps *)tab;
                const int nz = (int)g.shape[0], ny = (int)g.shape[1], nx = (int)g.shape[2];
                const int oz = (int)g.oshape[0], oy = (int)g.oshape[1], ox = (int)g.oshape[2];
                const bool do_y = g.pad < 2, do_z = g.pad < 1;
                void *bufA = nullptr, *bufB = nullptr;
                float *dst_x = (float *)out->data, *dst_y = (float *)out->data;
                if (do_y) {
                    if ((rc = pool_alloc(&bufA, (size_t)nz * ny * ox * sizeof(float), s))) { pool_free(tab); return rc; }
                    dst_x = (float *)bufA;
                }
                if (do_z) {
                    if ((rc = pool_alloc(&bufB, (size_t)nz * oy * ox * sizeof(float), s))) { pool_free(bufA); pool_free(tab); return rc; }
                    dst_y = (float *)bufB;
                }
                const dim3 blk(64, 4);
                // r5 (csrc/cubic_fast.hip): x from an LDS-staged row span, z with the window of planes in registers -- volumes only
                // (x is then never the last pass); mi_debug_set_resample_fast(0) keeps the r3 passes (bit-identical)
                const bool fast3 = g_resample_fast && do_y && do_z;
                if (!(fast3 && launch_resample_x_lds((const float *)coef->data, dst_x, T + oz + oy, (long long)nz * ny, ox, nx, (float)cval, s) == MI_OK))
                hipLaunchKernelGGL(cubic_resample_axis_kernel<2>, dim3((ox + 63) / 64, (ny + 3) / 4, nz), blk, 0, s,
                                   (const float *)coef->data, dst_x, T + oz + oy, nz, ny, ox, nx, (float)cval, T, oz, oy,
                                   do_y ? 0 : 1);
                const bool quad = ox % 4 == 0 && ((uintptr_t)out->data & 15) == 0;     // pool blocks are 256-byte aligned
                const int oxq = ox / 4;
                if (do_y) {
                    if (quad)
                        hipLaunchKernelGGL(cubic_resample_rows4_kernel<1>, dim3((oxq + 63) / 64, (oy + 3) / 4, nz), blk, 0, s,
                                           (const float4 *)bufA, (float4 *)dst_y, T + oz, nz, oy, oxq, ny, (float)cval, T, oz, oy,
                                           do_z ? 0 : 1);
                    else
                        hipLaunchKernelGGL(cubic_resample_axis_kernel<1>, dim3((ox + 63) / 64, (oy + 3) / 4, nz), blk, 0, s,
                                           (const float *)bufA, dst_y, T + oz, nz, oy, ox, ny, (float)cval, T, oz, oy,
                                           do_z ? 0 : 1);
                }
                if (do_z) {
                    if (quad && fast3 && launch_resample_zstream((const float *)bufB, (float *)out->data, T, oz, oy, oxq, nz, (float)cval, s) == MI_OK) {
                        note_kernel("mi::cubic_resample_zstream_kernel (order-3 diagonal transform: x from LDS-staged row spans, y, z with the plane window in registers)");
                    } else if (quad)
                        hipLaunchKernelGGL(cubic_resample_rows4_kernel<0>, dim3((oxq + 63) / 64, (oy + 3) / 4, oz), blk, 0, s,
                                           (const float4 *)bufB, (float4 *)out->data, T, oz, oy, oxq, nz, (float)cval, T, oz, oy, 1);
                    else
                        hipLaunchKernelGGL(cubic_resample_axis_kernel<0>, dim3((ox + 63) / 64, (oy + 3) / 4, oz), blk, 0, s,
                                           (const float *)bufB, (float *)out->data, T, oz, oy, ox, nz, (float)cval, T, oz, oy, 1);
                }
                pool_free(bufA);
                pool_free(bufB);
                pool_free(tab);
                MI_HIP(hipGetLastError());
                return MI_OK;
            }
        }
#define MI_CUBIC_AFF(NTZ, NTY)                                                                                          \
    if (diagonal)                                                                                                       \
        hipLaunchKernelGGL((cubic3_diag_f32_kernel<NTZ, NTY>), cgrid, dim3(64, 4), 0, s, (const float *)coef->data,     \
                           (float *)out->data, (const AxisTaps *)tab, g, numel(coef), mode, (float)cval);               \
    else                                                                                                                \
        hipLaunchKernelGGL((cubic3_f32_kernel<float, true, NTZ, NTY>), cgrid, dim3(64, 4), 0, s,                        \
                           (const float *)coef->data, (const float *)nullptr, (float *)out->data, g, nout, numel(coef), \
                           mode, (float)cval, npad)
        if (!diagonal) {
            int zrc = MI_OK;
            if (ident != 4 && launch_cubic_zstream(coef, out, g, mode, cval, npad, ident, s, &zrc)) return zrc;
            if ((ident == 0 || ident == 4) && launch_cubic_rowblend(coef, out, g, mode, cval, npad, ident == 4, s, &zrc)) return zrc;
            if (ident) { set_error("no kernel for this transform with unfiltered axes (mask %d)", ident); return MI_ERR_UNSUPPORTED; }
            // r5: all three axes coupled (small rotations about a general axis): taps out of an LDS-staged box (csrc/cubic_fast.hip)
            if (g_cubic_box && g.pad == 0) {
                const int shp[3] = {(int)g.shape[0], (int)g.shape[1], (int)g.shape[2]}, osh[3] = {(int)g.oshape[0], (int)g.oshape[1], (int)g.oshape[2]};
                if (launch_cubic_box((const float *)coef->data, (float *)out->data, shp, osh, g.mat, mode, cval, npad, s, &zrc, g_cubic_box)) return zrc;
            }
        }
        note_kernel(diagonal ? "mi::cubic3_diag_f32_kernel (order-3 affine on float32 coefficients, diagonal matrix: tabulated taps)"
                             : "mi::cubic3_f32_kernel (order-3 affine on float32 coefficients: 16 x 16-byte gathers per voxel)");
        if (g.pad == 0) { MI_CUBIC_AFF(4, 4); } else if (g.pad == 1) { MI_CUBIC_AFF(1, 4); } else { MI_CUBIC_AFF(1, 1); }
#undef MI_CUBIC_AFF
        if (tab) pool_free(tab);       // stream-ordered pool: the block is reused only by later work on the stream
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
#define MI_SPL_AFF(ORD)                                                                                           \
    hipLaunchKernelGGL((spline_affine_kernel<ORD>), grid, dim3(256), 0, s, (const double *)coef->data, out->data, \
                       out->dtype, g, nout, mode, cval, round_out, npad)
    switch (order) {
    case 2: MI_SPL_AFF(2); break;
    case 3: MI_SPL_AFF(3); break;
    case 4: MI_SPL_AFF(4); break;
    default: MI_SPL_AFF(5); break;
    }
#undef MI_SPL_AFF
    MI_HIP(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
