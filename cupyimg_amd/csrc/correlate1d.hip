// correlate1d.hip -- generic one-axis correlate and box-mean kernels.
//
// These are the "any dtype, any axis, any mode" kernels behind correlate1d /
// convolve1d / gaussian_filter1d / uniform_filter1d (reference launch site:
// cupyimg/scipy/ndimage/filters.py:213-283 -> :441-495 ->
// _filters_core.py:112-156, generated body _filters_core.py:298-324).
// The roofline kernels for the benchmarked shapes live in separable3d.hip.
//
// Arithmetic: taps are read as double and accumulated in double, in the order
// SciPy's NI_Correlate1D uses (centre tap first and outside-in pairs for
// symmetric / antisymmetric odd kernels, otherwise last tap then left to
// right), so float64 results are bit-identical to SciPy.  This file is built
// with -ffp-contract=off for that reason.  acc_f32 selects the reference's
// dtype_mode="float" (float32 accumulate, plain left-to-right order,
// _util.py:28-40).
//
// Layout: the array is viewed as (outer, n, inner) around the filtered axis;
// one thread per output element, consecutive lanes walk `inner` (or the axis
// itself when it is the last one), so every tap is a coalesced wave load and
// the taps of neighbouring outputs are served by L1/L2.
#include "common.hpp"

namespace mi {

constexpr int kInlineTaps = 64;

struct Taps {
    double w[kInlineTaps];    // used when wlen <= kInlineTaps (kernarg, scalar loads)
    const double *dev;        // otherwise
};

template <typename T, typename I>
__device__ __forceinline__ double tap_value(const T *__restrict__ in, I base, I inner, I l, I n,
                                            int mode, double cval)
{
    const I j = bmap<I>(l, n, mode);
    return j < 0 ? cval : (double)in[base + j * inner];
}

// The kernels below read the weights with a wave-uniform index straight from
// the kernel arguments (scalar loads) when they are inline, and issue the data
// loads of a group of taps before consuming any of them: with one load and one
// weight fetch per loop trip, each waited for, a wave keeps a single 256-byte
// request in flight and the pass runs at ~1 TB/s.

// sym: +1 symmetric, -1 antisymmetric, 0 general
template <typename T, typename I, typename W>
__device__ __forceinline__ void corr1d_f64_body(const T *__restrict__ in, void *__restrict__ out, int out_dt, I n, I inner,
                                                I total, W w, int wlen, int off, int mode, double cval, int sym, int geom)
{
    const int size1 = wlen / 2, size2 = wlen - size1 - 1;
    for_each_line_output<I>(geom, n, inner, total, [&](I i, I l, I base) {
        const I c = l - (I)off + (I)size1;   // input index under the centre tap
        const bool inside = c - (I)size1 >= 0 && c + (I)size1 < n;   // every tap inside the array: no boundary map
        double acc;
        if (inside) {
            const T *__restrict__ pc = in + base + c * inner;
            if (sym != 0) {
                acc = (double)pc[0] * w[size1];
                int j = -size1;
                for (; j + 2 <= 0; j += 2) {           // two tap pairs per trip, loads first
                    const T a0 = pc[(I)j * inner], b0 = pc[-(I)j * inner];
                    const T a1 = pc[(I)(j + 1) * inner], b1 = pc[-(I)(j + 1) * inner];
                    acc += (sym > 0 ? (double)a0 + (double)b0 : (double)a0 - (double)b0) * w[size1 + j];
                    acc += (sym > 0 ? (double)a1 + (double)b1 : (double)a1 - (double)b1) * w[size1 + j + 1];
                }
                for (; j < 0; j++) {
                    const double a = (double)pc[(I)j * inner], b = (double)pc[-(I)j * inner];
                    acc += (sym > 0 ? a + b : a - b) * w[size1 + j];
                }
            } else {
                acc = (double)pc[(I)size2 * inner] * w[wlen - 1];
                int j = -size1;
                for (; j + 4 <= size2; j += 4) {
                    T x[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) x[u] = pc[(I)(j + u) * inner];
#pragma unroll
                    for (int u = 0; u < 4; u++) acc += (double)x[u] * w[size1 + j + u];
                }
                for (; j < size2; j++) acc += (double)pc[(I)j * inner] * w[size1 + j];
            }
        } else if (sym != 0) {
            acc = tap_value<T, I>(in, base, inner, c, n, mode, cval) * w[size1];
            for (int j = -size1; j < 0; j++) {
                const double a = tap_value<T, I>(in, base, inner, c + j, n, mode, cval);
                const double b = tap_value<T, I>(in, base, inner, c - j, n, mode, cval);
                acc += (sym > 0 ? a + b : a - b) * w[size1 + j];
            }
        } else {
            acc = tap_value<T, I>(in, base, inner, c + size2, n, mode, cval) * w[wlen - 1];
            for (int j = -size1; j < size2; j++)
                acc += tap_value<T, I>(in, base, inner, c + j, n, mode, cval) * w[size1 + j];
        }
        store_as(out, (int64_t)i, out_dt, acc);
    });
}

template <typename T, typename I>
__global__ void __launch_bounds__(256)
corr1d_f64(const T *__restrict__ in, void *__restrict__ out, int out_dt, I n, I inner, I total,
           const Taps taps, int wlen, int off, int mode, double cval, int sym, int geom)
{
    if (wlen <= kInlineTaps) corr1d_f64_body<T, I>(in, out, out_dt, n, inner, total, taps.w, wlen, off, mode, cval, sym, geom);
    else corr1d_f64_body<T, I>(in, out, out_dt, n, inner, total, taps.dev, wlen, off, mode, cval, sym, geom);
}

template <typename T, typename I, typename W>
__device__ __forceinline__ void corr1d_f32_body(const T *__restrict__ in, void *__restrict__ out, int out_dt, I n, I inner,
                                                I total, W w, int wlen, int off, int mode, double cval, int geom)
{
    for_each_line_output<I>(geom, n, inner, total, [&](I i, I l, I base) {
        const I first = l - (I)off;
        const bool inside = first >= 0 && first + (I)wlen <= n;
        float acc = 0.f;
        if (inside) {
            const T *__restrict__ pf = in + base + first * inner;
            int k = 0;
            for (; k + 4 <= wlen; k += 4) {
                T x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) x[u] = pf[(I)(k + u) * inner];
#pragma unroll
                for (int u = 0; u < 4; u++) acc += (float)x[u] * (float)w[k + u];
            }
            for (; k < wlen; k++) acc += (float)pf[(I)k * inner] * (float)w[k];
        } else {
            for (int k = 0; k < wlen; k++) {
                const I j = bmap<I>(first + (I)k, n, mode);
                const float v = j < 0 ? (float)cval : (float)in[base + j * inner];
                acc += v * (float)w[k];
            }
        }
        store_as(out, (int64_t)i, out_dt, (double)acc);
    });
}

template <typename T, typename I>
__global__ void __launch_bounds__(256)
corr1d_f32(const T *__restrict__ in, void *__restrict__ out, int out_dt, I n, I inner, I total,
           const Taps taps, int wlen, int off, int mode, double cval, int geom)
{
    if (wlen <= kInlineTaps) corr1d_f32_body<T, I>(in, out, out_dt, n, inner, total, taps.w, wlen, off, mode, cval, geom);
    else corr1d_f32_body<T, I>(in, out, out_dt, n, inner, total, taps.dev, wlen, off, mode, cval, geom);
}

// box mean: exact double sum of the window, then one division (SciPy's
// NI_UniformFilter1D keeps a running sum and divides every sample)
template <typename T, typename I>
__global__ void __launch_bounds__(256)
box1d_f64(const T *__restrict__ in, void *__restrict__ out, int out_dt, I n, I inner, I total,
          int size, int off, int mode, double cval, int geom)
{
    for_each_line_output<I>(geom, n, inner, total, [&](I i, I l, I base) {
        const I first = l - (I)off;
        const bool inside = first >= 0 && first + (I)size <= n;
        double acc = 0.0;
        if (inside) {
            const T *__restrict__ pf = in + base + first * inner;
            int k = 0;
            for (; k + 4 <= size; k += 4) {
                T x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) x[u] = pf[(I)(k + u) * inner];
#pragma unroll
                for (int u = 0; u < 4; u++) acc += (double)x[u];
            }
            for (; k < size; k++) acc += (double)pf[(I)k * inner];
        } else {
            for (int k = 0; k < size; k++) acc += tap_value<T, I>(in, base, inner, first + (I)k, n, mode, cval);
        }
        store_as(out, (int64_t)i, out_dt, acc / (double)size);
    });
}

static int axis_view(const mi_array *in, int axis, int64_t *n, int64_t *inner)
{
    *n = in->shape[axis];
    *inner = 1;
    for (int d = axis + 1; d < in->ndim; d++) *inner *= in->shape[d];
    return MI_OK;
}

static int check_1d_args(const mi_array *in, const mi_array *out, int axis, int wlen, int origin)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1, MI_ERR_INVALID_ARG, "input must have at least one dimension");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(axis >= 0 && axis < in->ndim, MI_ERR_INVALID_ARG, "invalid axis");
    MI_REQUIRE(wlen >= 1, MI_ERR_INVALID_ARG, "incorrect filter size");
    MI_REQUIRE(wlen / 2 + origin >= 0 && wlen / 2 + origin < wlen, MI_ERR_INVALID_ARG, "invalid origin");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS,
               "correlate1d needs C-contiguous arrays");
    MI_REQUIRE(in->data != out->data, MI_ERR_INVALID_ARG, "in-place filtering is not supported by the kernel");
    return MI_OK;
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_correlate1d(const mi_array *in, const mi_array *out, int axis, const double *weights,
                   int wlen, int origin, int mode, double cval, int acc_f32, mi_stream stream)
{
    int rc = check_1d_args(in, out, axis, wlen, origin);
    if (rc) return rc;
    MI_REQUIRE(weights, MI_ERR_INVALID_ARG, "weights is NULL");
    const int64_t total = numel(in);
    if (total == 0) return MI_OK;
    int64_t n, inner;
    axis_view(in, axis, &n, &inner);
    hipStream_t s = resolve_stream(stream);
    mode = filter_mode(mode);

    Taps taps;
    taps.dev = nullptr;
    Scratch scratch;
    if (wlen <= kInlineTaps) {
        memcpy(taps.w, weights, sizeof(double) * wlen);
    } else {
        if ((rc = scratch.upload(weights, sizeof(double) * wlen, s))) return rc;
        taps.dev = (const double *)scratch.ptr;
    }
    const int size1 = wlen / 2;
    int sym = 0;
    if (wlen & 1) {
        sym = 1;
        for (int i = 1; i <= size1; i++)
            if (fabs(weights[size1 + i] - weights[size1 - i]) > 2.220446049250313e-16) { sym = 0; break; }
        if (!sym) {
            sym = -1;
            for (int i = 1; i <= size1; i++)
                if (fabs(weights[size1 + i] + weights[size1 - i]) > 2.220446049250313e-16) { sym = 0; break; }
        }
    }
    const int off = size1 + origin;
    // float32 accumulate only where promote(in, float32) == float32
    const bool f32ok = in->dtype == MI_F32 || in->dtype == MI_BOOL || dtype_size(in->dtype) <= 2;
    const bool use_f32 = acc_f32 && f32ok;
    dim3 grid, block(256);
    const int geom = line_grid(total, n, inner, &grid, &block);
    if (geom == 0) grid_for(total, 256, &grid);
    const bool big = total >= ((int64_t)1 << 31) - 256 * 8192;
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        const T *ip = (const T *)in->data;
        if (big) {
            if (use_f32)
                hipLaunchKernelGGL((corr1d_f32<T, int64_t>), grid, block, 0, s, ip, out->data, out->dtype,
                                   n, inner, total, taps, wlen, off, mode, cval, geom);
            else
                hipLaunchKernelGGL((corr1d_f64<T, int64_t>), grid, block, 0, s, ip, out->data, out->dtype,
                                   n, inner, total, taps, wlen, off, mode, cval, sym, geom);
        } else {
            if (use_f32)
                hipLaunchKernelGGL((corr1d_f32<T, int32_t>), grid, block, 0, s, ip, out->data, out->dtype,
                                   (int32_t)n, (int32_t)inner, (int32_t)total, taps, wlen, off, mode, cval, geom);
            else
                hipLaunchKernelGGL((corr1d_f64<T, int32_t>), grid, block, 0, s, ip, out->data, out->dtype,
                                   (int32_t)n, (int32_t)inner, (int32_t)total, taps, wlen, off, mode, cval, sym, geom);
        }
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

int mi_uniform_filter1d(const mi_array *in, const mi_array *out, int axis, int size, int origin,
                        int mode, double cval, mi_stream stream)
{
    int rc = check_1d_args(in, out, axis, size, origin);
    if (rc) return rc;
    const int64_t total = numel(in);
    if (total == 0) return MI_OK;
    int64_t n, inner;
    axis_view(in, axis, &n, &inner);
    hipStream_t s = resolve_stream(stream);
    mode = filter_mode(mode);
    const int off = size / 2 + origin;
    dim3 grid, block(256);
    const int geom = line_grid(total, n, inner, &grid, &block);
    if (geom == 0) grid_for(total, 256, &grid);
    const bool big = total >= ((int64_t)1 << 31) - 256 * 8192;
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        const T *ip = (const T *)in->data;
        if (big)
            hipLaunchKernelGGL((box1d_f64<T, int64_t>), grid, block, 0, s, ip, out->data, out->dtype, n,
                               inner, total, size, off, mode, cval, geom);
        else
            hipLaunchKernelGGL((box1d_f64<T, int32_t>), grid, block, 0, s, ip, out->data, out->dtype,
                               (int32_t)n, (int32_t)inner, (int32_t)total, size, off, mode, cval, geom);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

}  // extern "C"
