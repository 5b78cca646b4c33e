// copy.hip -- strided copy / dtype cast / fill / compare.
//
// The reference gets these from CuPy (`output[...] = input`,
// `temp[...] = output[...]`, `.astype`, `(a == b).all()`:
// cupyimg/scipy/ndimage/_filters_core.py:94,107,154, morphology.py:313,321).
// Cast rules follow _filters_core.py:166-187 / SciPy: truncate toward zero,
// negative -> unsigned wraps.
#include <vector>

#include "common.hpp"

namespace mi {

struct CopyParams {
    int ndim;
    int64_t shape[MI_MAX_NDIM];
    int64_t sstride[MI_MAX_NDIM];   // bytes
    int64_t dstride[MI_MAX_NDIM];   // bytes
};

template <typename S, typename D>
__device__ __forceinline__ D convert(S v, int rhe)
{
    if constexpr (std::is_same<S, D>::value) {
        return v;
    } else if constexpr (std::is_same<D, bool>::value) {
        return v != S(0);
    } else if constexpr (std::is_floating_point<D>::value) {
        return (D)v;
    } else if constexpr (std::is_floating_point<S>::value) {
        double a = (double)v;
        if (rhe) a = rint(a);
        return cast_from_f64<D>(a);
    } else {
        return (D)v;   // integer -> integer: modular, like C
    }
}

template <typename S, typename D>
__global__ void __launch_bounds__(256) copy_strided(const char *__restrict__ src, char *__restrict__ dst,
                                                    CopyParams p, int64_t total, int rhe)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i, so = 0, d_o = 0;
        for (int d = p.ndim - 1; d >= 0; d--) {
            const int64_t q = r / p.shape[d];
            const int64_t k = r - q * p.shape[d];
            so += k * p.sstride[d];
            d_o += k * p.dstride[d];
            r = q;
        }
        *(D *)(dst + d_o) = convert<S, D>(*(const S *)(src + so), rhe);
    }
}

template <typename S, typename D>
__global__ void __launch_bounds__(256) copy_linear(const S *__restrict__ src, D *__restrict__ dst,
                                                   int64_t total, int rhe)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = convert<S, D>(src[i], rhe);
}

template <typename D>
__global__ void __launch_bounds__(256) fill_strided(char *__restrict__ dst, CopyParams p, int64_t total, double v)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i, d_o = 0;
        for (int d = p.ndim - 1; d >= 0; d--) {
            const int64_t q = r / p.shape[d];
            d_o += (r - q * p.shape[d]) * p.dstride[d];
            r = q;
        }
        *(D *)(dst + d_o) = cast_from_f64<D>(v);
    }
}

// float16 (MI_F16) is a storage dtype: converted through float32 on the way in and out (r3)
template <typename S, typename D>
__device__ __forceinline__ D convert16(S v, int rhe)
{
    if constexpr (std::is_same<S, _Float16>::value && std::is_same<D, _Float16>::value) return v;
    else if constexpr (std::is_same<S, _Float16>::value) return convert<float, D>((float)v, rhe);
    else return (_Float16)convert<S, float>(v, rhe);
}

template <typename S, typename D>
__global__ void __launch_bounds__(256) copy16_strided(const char *__restrict__ src, char *__restrict__ dst, CopyParams p,
                                                      int64_t total, int rhe)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i, so = 0, d_o = 0;
        for (int d = p.ndim - 1; d >= 0; d--) {
            const int64_t q = r / p.shape[d];
            const int64_t k = r - q * p.shape[d];
            so += k * p.sstride[d];
            d_o += k * p.dstride[d];
            r = q;
        }
        *(D *)(dst + d_o) = convert16<S, D>(*(const S *)(src + so), rhe);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) any_diff_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                       int64_t total, int32_t *flag)
{
    bool diff = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x)
        diff |= (a[i] != b[i]);
    if (__any(diff) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// out = a (op) b elementwise on contiguous arrays of one dtype, in that dtype's
// own arithmetic (integers wrap like NumPy's same-dtype ufuncs); the building
// block of the composite filters (laplace, gradient magnitude, top-hats ...).
//   0 add, 1 subtract, 2 multiply, 3 sqrt(a) (integers: computed in double,
//   truncated), 4 a + b - 2 c is not needed: composites chain the binary forms.
template <typename T>
__global__ void __launch_bounds__(256) elementwise_kernel(const T *__restrict__ a, const T *__restrict__ b, T *__restrict__ out,
                                                          int64_t n, int op)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T x = a[i];
        const T y = b ? b[i] : T(0);
        T r;
        if constexpr (std::is_same<T, bool>::value) {
            r = op == 0 ? (x || y) : (op == 1 ? (x != y) : (op == 2 ? (x && y) : x));   // numpy: bool - bool is an error; xor is what the top-hats use
        } else if constexpr (std::is_floating_point<T>::value) {
            r = op == 0 ? x + y : (op == 1 ? x - y : (op == 2 ? x * y : (T)sqrt((double)x)));
        } else {
            typedef typename std::make_unsigned<T>::type U;
            if (op == 0) r = (T)((U)x + (U)y);
            else if (op == 1) r = (T)((U)x - (U)y);
            else if (op == 2) r = (T)((U)x * (U)y);
            else r = cast_from_f64<T>(sqrt((double)x));
        }
        out[i] = r;
    }
}

// out = a * s + t (op 0) or clip(a, lo, hi) keeping entries equal to `keep`
// when keep_flag is set (op 1); floating-point arrays (skimage facade: dtype
// range scaling, warp's output clipping, _warps.py:745-787)
template <typename T, typename TO = T>
__global__ void __launch_bounds__(256) scalar_op_kernel(const T *__restrict__ a, TO *__restrict__ out, int64_t n, int op,
                                                        double p0, double p1, double keep, int keep_flag)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = (double)a[i];
        double r;
        if (op == 0) r = x * p0 + p1;
        else r = (keep_flag && x == keep) ? x : fmin(fmax(x, p0), p1);
        out[i] = (TO)r;
    }
}

// per-block minimum / maximum of a contiguous array (NaNs are skipped, like fmin / fmax)
template <typename T>
__global__ void __launch_bounds__(256) minmax_reduce_kernel(const T *__restrict__ a, int64_t n, double *__restrict__ part)
{
    __shared__ double slo[256], shi[256];
    double lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = (double)a[i];
        lo = fmin(lo, x);
        hi = fmax(hi, x);
    }
    slo[threadIdx.x] = lo; shi[threadIdx.x] = hi;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) {
            slo[threadIdx.x] = fmin(slo[threadIdx.x], slo[threadIdx.x + sft]);
            shi[threadIdx.x] = fmax(shi[threadIdx.x], shi[threadIdx.x + sft]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[2 * blockIdx.x] = slo[0]; part[2 * blockIdx.x + 1] = shi[0]; }
}


// ---------------------------------------------------------------------------
// r4b: rows that are not a multiple of 16 bytes.  The fused kernels load rows 16 bytes at a time and need them 16-byte
// aligned; volumes such as 181 x 217 x 181 are first EXTENDED along x -- `left` elements in front (a multiple of 16 bytes,
// so the kept columns stay aligned), the boundary continuation behind, to a multiple of 16 bytes -- and the result's
// columns are copied back.  Both are row copies, 16 bytes per thread: one side aligned, the other an unaligned 16-byte
// access (legal on this target) when the whole vector lies inside the row, element by element at the row ends.
// ---------------------------------------------------------------------------
struct __attribute__((packed, aligned(1))) Unaligned16 { unsigned w[4]; };
struct __attribute__((aligned(16))) Aligned16 { unsigned w[4]; };

template <typename T>
__global__ void __launch_bounds__(256) extend_rows_kernel(const T *__restrict__ in, T *__restrict__ out, int64_t rows, int nx, int total, int left,
                                                          int mode, T cval)
{
    constexpr int V = 16 / (int)sizeof(T);
    const int q = total / V;                                     // 16-byte vectors per output row
    const int64_t n = rows * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / q;
        const int c = (int)(i - row * q) * V;
        const T *src = in + row * nx;
        const int x0 = c - left;
        Aligned16 r;
        if (x0 >= 0 && x0 + V <= nx) {
            const Unaligned16 u = *reinterpret_cast<const Unaligned16 *>(src + x0);
            r.w[0] = u.w[0]; r.w[1] = u.w[1]; r.w[2] = u.w[2]; r.w[3] = u.w[3];
        } else {
            T *v = reinterpret_cast<T *>(&r);
#pragma unroll
            for (int e = 0; e < V; e++) {
                const int x = x0 + e;
                const int xs = (unsigned)x < (unsigned)nx ? x : bmap<int>(x, nx, mode);      // -1: constant fill
                v[e] = xs >= 0 ? src[xs] : cval;
            }
        }
        *reinterpret_cast<Aligned16 *>(out + row * total + c) = r;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) crop_rows_kernel(const T *__restrict__ in, T *__restrict__ out, int64_t rows, int nx, int total, int left)
{
    constexpr int V = 16 / (int)sizeof(T);
    const int q = (nx + V - 1) / V;                              // vectors per kept row (the last one partial)
    const int64_t n = rows * q;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / q;
        const int c = (int)(i - row * q) * V;
        const Aligned16 r = *reinterpret_cast<const Aligned16 *>(in + row * total + left + c);      // left % V == 0, total % V == 0
        T *dst = out + row * nx + c;
        if (c + V <= nx) {
            Unaligned16 u;
            u.w[0] = r.w[0]; u.w[1] = r.w[1]; u.w[2] = r.w[2]; u.w[3] = r.w[3];
            *reinterpret_cast<Unaligned16 *>(dst) = u;
        } else {
            const T *v = reinterpret_cast<const T *>(&r);
#pragma unroll
            for (int e = 0; e < V; e++) if (c + e < nx) dst[e] = v[e];
        }
    }
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_copy(const mi_array *src, const mi_array *dst, int round_half_even, mi_stream stream)
{
    int rc;
    if ((rc = check_array(src, "src")) || (rc = check_array(dst, "dst"))) return rc;
    MI_REQUIRE(same_shape(src, dst), MI_ERR_INVALID_ARG, "mi_copy: shapes differ");
    const int64_t total = numel(src);
    if (total == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    const bool lin = is_contiguous(src) && is_contiguous(dst);
    if (lin && src->dtype == dst->dtype) {
        if (src->data != dst->data)
            MI_HIP(hipMemcpyAsync(dst->data, src->data, (size_t)total * dtype_size(src->dtype),
                                  hipMemcpyDeviceToDevice, s));
        return MI_OK;
    }
    CopyParams p;
    p.ndim = src->ndim;
    for (int d = 0; d < src->ndim; d++) {
        p.shape[d] = src->shape[d];
        p.sstride[d] = src->strides[d];
        p.dstride[d] = dst->strides[d];
    }
    dim3 grid;
    grid_for(total, 256, &grid);
    if (src->dtype == MI_F16 || dst->dtype == MI_F16) {
        if (src->dtype == MI_F16 && dst->dtype == MI_F16)
            hipLaunchKernelGGL((copy16_strided<_Float16, _Float16>), grid, dim3(256), 0, s, (const char *)src->data, (char *)dst->data, p,
                               total, round_half_even);
        else if (src->dtype == MI_F16)
            return dispatch_dtype(dst->dtype, [&]<typename D>() -> int {
                hipLaunchKernelGGL((copy16_strided<_Float16, D>), grid, dim3(256), 0, s, (const char *)src->data, (char *)dst->data, p,
                                   total, round_half_even);
                MI_HIP(hipGetLastError());
                return MI_OK;
            });
        else
            return dispatch_dtype(src->dtype, [&]<typename S>() -> int {
                hipLaunchKernelGGL((copy16_strided<S, _Float16>), grid, dim3(256), 0, s, (const char *)src->data, (char *)dst->data, p,
                                   total, round_half_even);
                MI_HIP(hipGetLastError());
                return MI_OK;
            });
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
    return dispatch_dtype(src->dtype, [&]<typename S>() -> int {
        return dispatch_dtype(dst->dtype, [&]<typename D>() -> int {
            if (lin)
                hipLaunchKernelGGL((copy_linear<S, D>), grid, dim3(256), 0, s, (const S *)src->data,
                                   (D *)dst->data, total, round_half_even);
            else
                hipLaunchKernelGGL((copy_strided<S, D>), grid, dim3(256), 0, s,
                                   (const char *)src->data, (char *)dst->data, p, total,
                                   round_half_even);
            MI_HIP(hipGetLastError());
            return MI_OK;
        });
    });
}

int mi_fill(const mi_array *dst, double value, mi_stream stream)
{
    int rc;
    if ((rc = check_array(dst, "dst"))) return rc;
    const int64_t total = numel(dst);
    if (total == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    if (value == 0.0 && is_contiguous(dst)) {
        MI_HIP(hipMemsetAsync(dst->data, 0, (size_t)total * dtype_size(dst->dtype), s));
        return MI_OK;
    }
    CopyParams p;
    p.ndim = dst->ndim;
    for (int d = 0; d < dst->ndim; d++) {
        p.shape[d] = dst->shape[d];
        p.sstride[d] = 0;
        p.dstride[d] = dst->strides[d];
    }
    dim3 grid;
    grid_for(total, 256, &grid);
    return dispatch_dtype(dst->dtype, [&]<typename D>() -> int {
        hipLaunchKernelGGL((fill_strided<D>), grid, dim3(256), 0, s, (char *)dst->data, p, total, value);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

int mi_any_diff(const mi_array *a, const mi_array *b, int32_t *flag_dev, mi_stream stream)
{
    int rc;
    if ((rc = check_array(a, "a")) || (rc = check_array(b, "b"))) return rc;
    MI_REQUIRE(flag_dev, MI_ERR_INVALID_ARG, "flag_dev is NULL");
    MI_REQUIRE(same_shape(a, b) && a->dtype == b->dtype, MI_ERR_INVALID_ARG,
               "mi_any_diff: shape/dtype mismatch");
    MI_REQUIRE(is_contiguous(a) && is_contiguous(b), MI_ERR_NOT_CONTIGUOUS,
               "mi_any_diff needs contiguous arrays");
    const int64_t total = numel(a);
    if (total == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    dim3 grid;
    grid_for(total, 256, &grid);
    // compare raw bytes by size class
    const size_t es = dtype_size(a->dtype);
    if (es == 1)
        hipLaunchKernelGGL((any_diff_kernel<uint8_t>), grid, dim3(256), 0, s, (const uint8_t *)a->data,
                           (const uint8_t *)b->data, total, flag_dev);
    else if (es == 2)
        hipLaunchKernelGGL((any_diff_kernel<uint16_t>), grid, dim3(256), 0, s, (const uint16_t *)a->data,
                           (const uint16_t *)b->data, total, flag_dev);
    else if (es == 4)
        hipLaunchKernelGGL((any_diff_kernel<uint32_t>), grid, dim3(256), 0, s, (const uint32_t *)a->data,
                           (const uint32_t *)b->data, total, flag_dev);
    else
        hipLaunchKernelGGL((any_diff_kernel<uint64_t>), grid, dim3(256), 0, s, (const uint64_t *)a->data,
                           (const uint64_t *)b->data, total, flag_dev);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_elementwise(int op, const mi_array *a, const mi_array *b, const mi_array *out, mi_stream stream)
{
    int rc;
    if ((rc = check_array(a, "a")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(op >= 0 && op <= 3, MI_ERR_INVALID_ARG, "unknown elementwise operation");
    MI_REQUIRE(op == 3 || b, MI_ERR_INVALID_ARG, "binary operation needs two operands");
    if (b && (rc = check_array(b, "b"))) return rc;
    MI_REQUIRE(same_shape(a, out) && a->dtype == out->dtype, MI_ERR_INVALID_ARG, "operands must agree in shape and dtype");
    MI_REQUIRE(!b || (same_shape(a, b) && a->dtype == b->dtype), MI_ERR_INVALID_ARG, "operands must agree in shape and dtype");
    MI_REQUIRE(is_contiguous(a) && is_contiguous(out) && (!b || is_contiguous(b)), MI_ERR_NOT_CONTIGUOUS,
               "elementwise operations need C-contiguous arrays");
    const int64_t n = numel(a);
    if (n == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    dim3 grid;
    grid_for(n, 256, &grid);
    return dispatch_dtype(a->dtype, [&]<typename T>() -> int {
        hipLaunchKernelGGL((elementwise_kernel<T>), grid, dim3(256), 0, s, (const T *)a->data, b ? (const T *)b->data : nullptr,
                           (T *)out->data, n, op);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

int mi_scalar_op(int op, const mi_array *a, const mi_array *out, double p0, double p1, double keep, int keep_flag,
                 mi_stream stream)
{
    int rc;
    if ((rc = check_array(a, "a")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(op == 0 || op == 1, MI_ERR_INVALID_ARG, "unknown scalar operation");
    MI_REQUIRE(is_contiguous(a) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS, "needs C-contiguous arrays");
    const int64_t n = numel(a);
    dim3 grid;
    grid_for(n > 0 ? n : 1, 256, &grid);
    hipStream_t s = resolve_stream(stream);
    if (op == 0 && same_shape(a, out) && a->dtype != out->dtype && (out->dtype == MI_F32 || out->dtype == MI_F64) &&
        (a->dtype == MI_U8 || a->dtype == MI_I8 || a->dtype == MI_U16 || a->dtype == MI_I16 || a->dtype == MI_BOOL)) {
        // integer image -> float image with the range scaling in the same pass (img_as_float: convert, then scale)
        if (n == 0) return MI_OK;
        return dispatch_dtype(a->dtype, [&]<typename T>() -> int {
            if constexpr (sizeof(T) <= 2) {
                if (out->dtype == MI_F32)
                    hipLaunchKernelGGL((scalar_op_kernel<T, float>), grid, dim3(256), 0, s, (const T *)a->data, (float *)out->data, n, 0,
                                       p0, p1, 0.0, 0);
                else
                    hipLaunchKernelGGL((scalar_op_kernel<T, double>), grid, dim3(256), 0, s, (const T *)a->data, (double *)out->data, n,
                                       0, p0, p1, 0.0, 0);
                MI_HIP(hipGetLastError());
            }
            return MI_OK;
        });
    }
    MI_REQUIRE(same_shape(a, out) && a->dtype == out->dtype && (a->dtype == MI_F32 || a->dtype == MI_F64), MI_ERR_INVALID_ARG,
               "float32 / float64 arrays of one shape (or an 8- / 16-bit integer input with a float output for op 0)");
    if (n == 0) return MI_OK;
    if (a->dtype == MI_F32)
        hipLaunchKernelGGL((scalar_op_kernel<float>), grid, dim3(256), 0, s, (const float *)a->data, (float *)out->data, n, op, p0,
                           p1, keep, keep_flag);
    else
        hipLaunchKernelGGL((scalar_op_kernel<double>), grid, dim3(256), 0, s, (const double *)a->data, (double *)out->data, n, op,
                           p0, p1, keep, keep_flag);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_min_max(const mi_array *a, double *lo, double *hi, mi_stream stream)
{
    int rc;
    if ((rc = check_array(a, "a"))) return rc;
    MI_REQUIRE(lo && hi, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(is_contiguous(a), MI_ERR_NOT_CONTIGUOUS, "needs a C-contiguous array");
    const int64_t n = numel(a);
    MI_REQUIRE(n > 0, MI_ERR_INVALID_ARG, "empty array");
    const int blocks = (int)std::min<int64_t>(1024, (n + 255) / 256);
    void *part = nullptr;
    if ((rc = pool_alloc(&part, (size_t)blocks * 2 * sizeof(double), resolve_stream(stream)))) return rc;
    hipStream_t s = resolve_stream(stream);
    rc = dispatch_dtype(a->dtype, [&]<typename T>() -> int {
        hipLaunchKernelGGL((minmax_reduce_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T *)a->data, n, (double *)part);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
    if (rc == MI_OK) {
        std::vector<double> host((size_t)blocks * 2);
        hipError_t e = hipMemcpyAsync(host.data(), part, host.size() * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("HIP error: %s", hipGetErrorString(e)); rc = MI_ERR_INTERNAL; }
        else {
            double l = INFINITY, h = -INFINITY;
            for (int b = 0; b < blocks; b++) { l = fmin(l, host[2 * b]); h = fmax(h, host[2 * b + 1]); }
            *lo = l; *hi = h;
        }
    }
    pool_free(part);
    return rc;
}

/* out[..., x] = in[..., map(x - left)] for x in [0, out.shape[-1]): rows extended along the last axis by the boundary mode
 * (filter semantics, include/mi355img.h mi_mode; MI_MODE_CONSTANT fills with cval).  1-, 2-, 4- and 8-byte dtypes, in and out
 * of one dtype, C-contiguous, same leading shape; the output rows and `left` multiples of 16 bytes, out 16-byte aligned. */
static int rows_common(const char *who, const mi_array *wide, const mi_array *narrow, int left, int64_t *rows, int *esize)
{
#define ROWS_REQUIRE(cond, code, msg) do { if (!(cond)) { set_error("%s: %s", who, msg); return (code); } } while (0)
    ROWS_REQUIRE(wide->dtype == narrow->dtype && wide->ndim == narrow->ndim && wide->ndim >= 1, MI_ERR_INVALID_ARG, "arrays of one dtype and rank");
    const int es = (int)dtype_size(wide->dtype);
    ROWS_REQUIRE(es == 1 || es == 2 || es == 4 || es == 8, MI_ERR_UNSUPPORTED, "1-, 2-, 4- and 8-byte dtypes");
    ROWS_REQUIRE(is_contiguous(wide) && is_contiguous(narrow), MI_ERR_NOT_CONTIGUOUS, "C-contiguous arrays");
    const int nd = wide->ndim, v = 16 / es;
    *rows = 1;
    for (int d = 0; d < nd - 1; d++) {
        ROWS_REQUIRE(wide->shape[d] == narrow->shape[d], MI_ERR_INVALID_ARG, "leading shapes differ");
        *rows *= wide->shape[d];
    }
    const int64_t nx = narrow->shape[nd - 1], total = wide->shape[nd - 1];
    ROWS_REQUIRE(left >= 0 && left % v == 0 && total % v == 0 && total >= left + (nx + v - 1) / v * v && nx >= 1 && total < ((int64_t)1 << 30),
                 MI_ERR_INVALID_ARG, "`left` and the extended rows must be multiples of 16 bytes, the kept columns (rounded up to 16 bytes) inside a row");
    ROWS_REQUIRE(((uintptr_t)wide->data & 15) == 0, MI_ERR_INVALID_ARG, "the extended array must be 16-byte aligned");
#undef ROWS_REQUIRE
    *esize = es;
    return MI_OK;
}

int mi_extend_rows(const mi_array *in, const mi_array *out, int left, int mode, double cval, mi_stream stream)
{
    int rc, es;
    int64_t rows;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    if ((rc = rows_common("mi_extend_rows", out, in, left, &rows, &es))) return rc;
    if (rows == 0) return MI_OK;
    const int nd = in->ndim;
    const int nx = (int)in->shape[nd - 1], total = (int)out->shape[nd - 1];
    dim3 grid;
    grid_for(rows * (total / (16 / es)), 256, &grid);
    hipStream_t s = resolve_stream(stream);
    const int m = filter_mode(mode);
    if (es == 8) {
        // r5: float64 / 64-bit integers (rows of an odd number of doubles: 181 x 217 x 181 as nibabel's get_fdata() hands it out)
        unsigned long long bits;
        if (in->dtype == MI_F64) memcpy(&bits, &cval, 8);
        else bits = (unsigned long long)(int64_t)cval;
        hipLaunchKernelGGL(extend_rows_kernel<unsigned long long>, grid, dim3(256), 0, s, (const unsigned long long *)in->data, (unsigned long long *)out->data, rows, nx,
                           total, left, m, bits);
    } else if (es == 4) {
        // the fill value in the array's own 4-byte dtype
        unsigned bits;
        if (in->dtype == MI_F32) { const float f = (float)cval; memcpy(&bits, &f, 4); }
        else bits = (unsigned)(int64_t)cval;
        hipLaunchKernelGGL(extend_rows_kernel<unsigned>, grid, dim3(256), 0, s, (const unsigned *)in->data, (unsigned *)out->data, rows, nx, total, left, m, bits);
    } else if (es == 2) {
        hipLaunchKernelGGL(extend_rows_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short *)in->data, (unsigned short *)out->data, rows, nx, total,
                           left, m, (unsigned short)(int64_t)cval);
    } else {
        hipLaunchKernelGGL(extend_rows_kernel<unsigned char>, grid, dim3(256), 0, s, (const unsigned char *)in->data, (unsigned char *)out->data, rows, nx, total,
                           left, m, (unsigned char)(int64_t)cval);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

/* out[..., x] = in[..., left + x]: the inverse of mi_extend_rows (same requirements, roles of in / out swapped). */
int mi_crop_rows(const mi_array *in, const mi_array *out, int left, mi_stream stream)
{
    int rc, es;
    int64_t rows;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    if ((rc = rows_common("mi_crop_rows", in, out, left, &rows, &es))) return rc;
    if (rows == 0) return MI_OK;
    const int nd = in->ndim;
    const int total = (int)in->shape[nd - 1], nx = (int)out->shape[nd - 1];
    const int v = 16 / es;
    dim3 grid;
    grid_for(rows * ((nx + v - 1) / v), 256, &grid);
    hipStream_t s = resolve_stream(stream);
    if (es == 8) hipLaunchKernelGGL(crop_rows_kernel<unsigned long long>, grid, dim3(256), 0, s, (const unsigned long long *)in->data, (unsigned long long *)out->data, rows, nx, total, left);
    else if (es == 4) hipLaunchKernelGGL(crop_rows_kernel<unsigned>, grid, dim3(256), 0, s, (const unsigned *)in->data, (unsigned *)out->data, rows, nx, total, left);
    else if (es == 2) hipLaunchKernelGGL(crop_rows_kernel<unsigned short>, grid, dim3(256), 0, s, (const unsigned short *)in->data, (unsigned short *)out->data, rows, nx, total, left);
    else hipLaunchKernelGGL(crop_rows_kernel<unsigned char>, grid, dim3(256), 0, s, (const unsigned char *)in->data, (unsigned char *)out->data, rows, nx, total, left);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

}  // extern "C"
