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
    int64_t lo[ND], hi[ND];
    double wlo[ND], whi[ND];
    bool two[ND];
#pragma unroll
    for (int d = 0; d < ND; d++) {
        const double cf = floor(c[d]);
        two[d] = c[d] != cf;
        wlo[d] = (cf + 1.0) - c[d];
        whi[d] = c[d] - cf;
        if (mode == MI_MODE_WRAP) {
            const double f = wrap_coord(c[d], g.shape[d]);
            lo[d] = (int64_t)floor(f);
            hi[d] = (int64_t)floor(f + 1.0);
        } else {
            lo[d] = (int64_t)cf;
            hi[d] = lo[d] + 1;
            if (mode != MI_MODE_CONSTANT) {
                lo[d] = bmap<int64_t>(lo[d], g.shape[d], mode);
                hi[d] = bmap<int64_t>(hi[d], g.shape[d], mode);
            }
        }
    }
    double acc = 0.0;
    // enumerate corners in the oracle's order: axis 0 is the most significant bit
    for (int m = 0; m < (1 << ND); m++) {
        double wt = 1.0;
        int64_t pos = 0;
        bool skip = false, oob = false;
#pragma unroll
        for (int d = 0; d < ND; d++) {
            const bool up = (m >> (ND - 1 - d)) & 1;
            skip |= up && !two[d];
            const int64_t j = up ? hi[d] : lo[d];
            wt *= up ? whi[d] : wlo[d];
            oob |= j < 0;
            pos += j * g.stride[d];
        }
        if (skip) continue;
        acc += (oob ? cval : (double)in[pos]) * wt;
    }
    return acc;
}

template <typename T, typename C, int ND>
__global__ void __launch_bounds__(256)
map_coordinates_kernel(const T *__restrict__ in, const C *__restrict__ coords, void *__restrict__ out,
                       int out_dt, InterpGeom g, int64_t nout, int order, int mode, double cval,
                       int round_out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nout;
         i += (int64_t)gridDim.x * blockDim.x) {
        double c[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) c[d] = d < g.pad ? 0.0 : (double)coords[(int64_t)(d - g.pad) * nout + i];
        double v = interp_point<T, ND>(in, g, c, order, mode, cval);
        if (round_out) v = interp_round(v, out_dt);
        store_as(out, i, out_dt, v);
    }
}

template <typename T, int ND>
__global__ void __launch_bounds__(256)
affine_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, InterpGeom g, int64_t nout,
              int order, int mode, double cval, int round_out)
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
        double v = interp_point<T, ND>(in, g, c, order, mode, cval);
        if (round_out) v = interp_round(v, out_dt);
        store_as(out, i, out_dt, v);
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
    if (order > 1) { set_error("spline order %d has no kernel yet (orders 0 and 1 are built)", order); return MI_ERR_UNSUPPORTED; }
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

static int g_interp_generic = 0;   // test hook: 1 = always use the generic double kernels
extern "C" int mi_debug_set_interp_generic(int v) { g_interp_generic = v; return MI_OK; }

extern "C" {

int mi_map_coordinates(const mi_array *in, const mi_array *coords, const mi_array *out, int order,
                       int mode, double cval, mi_stream stream)
{
    int rc = check_interp(in, out, order, mode);
    if (rc) return rc;
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
                       (const C *)coords->data, out->data, out->dtype, g, nout, order, mode, cval, round_out)
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
                               order, mode, cval, round_out);
        else if constexpr (std::is_floating_point<T>::value)
            hipLaunchKernelGGL((affine_kernel<T, MI_MAX_NDIM>), grid, dim3(256), 0, s, ip, out->data, out->dtype,
                               g, nout, order, mode, cval, round_out);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

}  // extern "C"
