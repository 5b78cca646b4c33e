// Consumers of the filtering path: the SSIM map from its five filtered moments in one pass
// (skimage/metrics/_structural_similarity.py:189-251 does the same arithmetic as ~20 array
// expressions) and a strided sum for the cropped mean / the simple metrics.
#include <cmath>
#include <vector>
#include "common.hpp"

namespace mi {

template <typename T>
__global__ void __launch_bounds__(256)
ssim_combine_kernel(const T *__restrict__ ux, const T *__restrict__ uy, const T *__restrict__ uxx, const T *__restrict__ uyy,
                    const T *__restrict__ uxy, T *__restrict__ S, T *__restrict__ gA, T *__restrict__ gB, T *__restrict__ gC,
                    int64_t n, T cov, T C1, T C2)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T mx = ux[i], my = uy[i];
        const T vx = cov * (uxx[i] - mx * mx);
        const T vy = cov * (uyy[i] - my * my);
        const T vxy = cov * (uxy[i] - mx * my);
        const T A1 = (T)2 * mx * my + C1;
        const T A2 = (T)2 * vxy + C2;
        const T B1 = mx * mx + my * my + C1;
        const T B2 = vx + vy + C2;
        const T D = B1 * B2;
        const T s = (A1 * A2) / D;
        S[i] = s;
        if (gA) {
            // the three fields whose filtered versions make the gradient (Avanaki 2009, eqs. 7-8)
            gA[i] = A1 / D;
            gB[i] = -s / B2;
            gC[i] = (mx * (A2 - A1) - my * (B2 - B1) * s) / D;
        }
    }
}

// the same with the mean over the map cropped by `pad` samples on every side accumulated on the way (double partial
// sums per workgroup): a workgroup walks whole rows, rows and columns outside the crop box do not count.  S may be null
// (mean only).  Arrays are C-contiguous of shape (n0, n1, n2) (leading unit axes for lower ranks).
template <typename T>
__global__ void __launch_bounds__(256)
ssim_combine_mean_kernel(const T *__restrict__ ux, const T *__restrict__ uy, const T *__restrict__ uxx, const T *__restrict__ uyy,
                         const T *__restrict__ uxy, T *__restrict__ S, int64_t n0, int64_t n1, int64_t n2, int p0, int p1, int p2,
                         T cov, T C1, T C2, double *__restrict__ part)
{
    __shared__ double sh[256];
    double acc = 0.0;
    const int64_t nrows = n0 * n1;
    for (int64_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int64_t i0 = row / n1, i1 = row - i0 * n1;
        const bool row_in = i0 >= p0 && i0 < n0 - p0 && i1 >= p1 && i1 < n1 - p1;
        const int64_t base = row * n2;
        for (int64_t c = threadIdx.x; c < n2; c += blockDim.x) {
            const int64_t i = base + c;
            const T mx = ux[i], my = uy[i];
            const T vx = cov * (uxx[i] - mx * mx);
            const T vy = cov * (uyy[i] - my * my);
            const T vxy = cov * (uxy[i] - mx * my);
            const T A1 = (T)2 * mx * my + C1;
            const T A2 = (T)2 * vxy + C2;
            const T B1 = mx * mx + my * my + C1;
            const T B2 = vx + vy + C2;
            const T s = (A1 * A2) / (B1 * B2);
            if (S) S[i] = s;
            if (row_in && c >= p2 && c < n2 - p2) acc += (double)s;
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) sh[threadIdx.x] += sh[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

struct SumParams {
    int ndim;
    int64_t shape[MI_MAX_NDIM];
    int64_t stride[MI_MAX_NDIM];   // bytes
};

// op 0: sum a; op 1: sum (a - b)^2; op 2: sum a^2 -- in double.  A workgroup walks whole rows (the last axis): the index
// of a row is decomposed once per row (64-bit divisions), the threads stride along it -- the per-element decomposition
// of the first version made the cropped mean of a 512^3 SSIM map the slowest kernel of the metric (711 us).
template <typename T>
__global__ void __launch_bounds__(256)
sum_kernel(const char *__restrict__ a, const char *__restrict__ b, int64_t nrows, SumParams pa, SumParams pb, int op,
           double *__restrict__ part)
{
    __shared__ double sh[256];
    double acc = 0.0;
    const int last = pa.ndim - 1;
    const int64_t len = last >= 0 ? pa.shape[last] : 1;
    const int64_t sa = last >= 0 ? pa.stride[last] : 0, sb = last >= 0 ? pb.stride[last] : 0;
    for (int64_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        int64_t rem = row, oa = 0, ob = 0;
        for (int d = last - 1; d >= 0; d--) {
            const int64_t q = rem / pa.shape[d];
            const int64_t c = rem - q * pa.shape[d];
            rem = q;
            oa += c * pa.stride[d];
            ob += c * pb.stride[d];
        }
        for (int64_t c = threadIdx.x; c < len; c += blockDim.x) {
            const double x = (double)*reinterpret_cast<const T *>(a + oa + c * sa);
            if (op == 0) acc += x;
            else if (op == 2) acc += x * x;
            else {
                const double d = x - (double)*reinterpret_cast<const T *>(b + ob + c * sb);
                acc += d * d;
            }
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) sh[threadIdx.x] += sh[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// x*x, y*y, x*y of two images in one pass (the second-moment inputs of SSIM): one read of the pair, three writes
template <typename T>
__global__ void __launch_bounds__(256) ssim_products_kernel(const T *__restrict__ x, const T *__restrict__ y, T *__restrict__ xx,
                                                            T *__restrict__ yy, T *__restrict__ xy, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T a = x[i], b = y[i];
        xx[i] = a * a;
        yy[i] = b * b;
        xy[i] = a * b;
    }
}

}  // namespace mi

using namespace mi;

extern "C" {

int mi_ssim_products(const mi_array *x, const mi_array *y, const mi_array *xx, const mi_array *yy, const mi_array *xy,
                     mi_stream stream)
{
    int rc;
    for (const mi_array *a : {x, y, xx, yy, xy}) {
        if ((rc = check_array(a, "image"))) return rc;
        MI_REQUIRE(same_shape(a, x) && a->dtype == x->dtype, MI_ERR_INVALID_ARG, "arrays must agree in shape and dtype");
        MI_REQUIRE(is_contiguous(a), MI_ERR_NOT_CONTIGUOUS, "mi_ssim_products needs C-contiguous arrays");
    }
    MI_REQUIRE(x->dtype == MI_F32 || x->dtype == MI_F64, MI_ERR_INVALID_ARG, "float32 / float64 images");
    const int64_t n = numel(x);
    if (n == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    const int blocks = (int)std::min<int64_t>(256 * 16, (n + 255) / 256);
    if (x->dtype == MI_F32)
        hipLaunchKernelGGL((ssim_products_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float *)x->data, (const float *)y->data,
                           (float *)xx->data, (float *)yy->data, (float *)xy->data, n);
    else
        hipLaunchKernelGGL((ssim_products_kernel<double>), dim3(blocks), dim3(256), 0, s, (const double *)x->data,
                           (const double *)y->data, (double *)xx->data, (double *)yy->data, (double *)xy->data, n);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_ssim_combine(const mi_array *ux, const mi_array *uy, const mi_array *uxx, const mi_array *uyy, const mi_array *uxy,
                    const mi_array *S, const mi_array *gA, const mi_array *gB, const mi_array *gC, double cov_norm, double C1,
                    double C2, mi_stream stream)
{
    int rc;
    const mi_array *req[6] = {ux, uy, uxx, uyy, uxy, S};
    for (const mi_array *a : req) {
        if ((rc = check_array(a, "moment"))) return rc;
        MI_REQUIRE(same_shape(a, ux) && a->dtype == ux->dtype, MI_ERR_INVALID_ARG, "moments must agree in shape and dtype");
        MI_REQUIRE(is_contiguous(a), MI_ERR_NOT_CONTIGUOUS, "mi_ssim_combine needs C-contiguous arrays");
    }
    MI_REQUIRE(ux->dtype == MI_F32 || ux->dtype == MI_F64, MI_ERR_INVALID_ARG, "float32 / float64 moments");
    MI_REQUIRE((!gA && !gB && !gC) || (gA && gB && gC), MI_ERR_INVALID_ARG, "gradient fields come as a set of three");
    if (gA)
        for (const mi_array *a : {gA, gB, gC}) {
            if ((rc = check_array(a, "gradient field"))) return rc;
            MI_REQUIRE(same_shape(a, ux) && a->dtype == ux->dtype && is_contiguous(a), MI_ERR_INVALID_ARG,
                       "gradient fields must match the moments");
        }
    const int64_t n = numel(ux);
    if (n == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    const int blocks = (int)std::min<int64_t>(256 * 16, (n + 255) / 256);
    if (ux->dtype == MI_F32)
        hipLaunchKernelGGL((ssim_combine_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float *)ux->data,
                           (const float *)uy->data, (const float *)uxx->data, (const float *)uyy->data, (const float *)uxy->data,
                           (float *)S->data, gA ? (float *)gA->data : nullptr, gA ? (float *)gB->data : nullptr,
                           gA ? (float *)gC->data : nullptr, n, (float)cov_norm, (float)C1, (float)C2);
    else
        hipLaunchKernelGGL((ssim_combine_kernel<double>), dim3(blocks), dim3(256), 0, s, (const double *)ux->data,
                           (const double *)uy->data, (const double *)uxx->data, (const double *)uyy->data,
                           (const double *)uxy->data, (double *)S->data, gA ? (double *)gA->data : nullptr,
                           gA ? (double *)gB->data : nullptr, gA ? (double *)gC->data : nullptr, n, cov_norm, C1, C2);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int mi_ssim_combine_mean(const mi_array *ux, const mi_array *uy, const mi_array *uxx, const mi_array *uyy, const mi_array *uxy,
                         const mi_array *S, int pad, double cov_norm, double C1, double C2, double *sum_out, mi_stream stream)
{
    int rc;
    const mi_array *req[5] = {ux, uy, uxx, uyy, uxy};
    for (const mi_array *a : req) {
        if ((rc = check_array(a, "moment"))) return rc;
        MI_REQUIRE(same_shape(a, ux) && a->dtype == ux->dtype, MI_ERR_INVALID_ARG, "moments must agree in shape and dtype");
        MI_REQUIRE(is_contiguous(a), MI_ERR_NOT_CONTIGUOUS, "mi_ssim_combine_mean needs C-contiguous arrays");
    }
    if (S) {
        if ((rc = check_array(S, "S"))) return rc;
        MI_REQUIRE(same_shape(S, ux) && S->dtype == ux->dtype && is_contiguous(S), MI_ERR_INVALID_ARG, "S must match the moments");
    }
    MI_REQUIRE(ux->dtype == MI_F32 || ux->dtype == MI_F64, MI_ERR_INVALID_ARG, "float32 / float64 moments");
    MI_REQUIRE(sum_out && pad >= 0, MI_ERR_INVALID_ARG, "bad argument");
    if (ux->ndim < 1 || ux->ndim > 3) { set_error("ssim_combine_mean: rank 1..3"); return MI_ERR_UNSUPPORTED; }
    *sum_out = 0.0;
    if (numel(ux) == 0) return MI_OK;
    int64_t n[3] = {1, 1, 1};
    int pd[3] = {0, 0, 0};
    for (int d = 0; d < ux->ndim; d++) { n[3 - ux->ndim + d] = ux->shape[d]; pd[3 - ux->ndim + d] = pad; }
    hipStream_t s = resolve_stream(stream);
    const int blocks = (int)std::min<int64_t>(4096, n[0] * n[1]);
    void *part = nullptr;
    if ((rc = pool_alloc(&part, (size_t)blocks * sizeof(double), s))) return rc;
    if (ux->dtype == MI_F32)
        hipLaunchKernelGGL((ssim_combine_mean_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float *)ux->data,
                           (const float *)uy->data, (const float *)uxx->data, (const float *)uyy->data, (const float *)uxy->data,
                           S ? (float *)S->data : nullptr, n[0], n[1], n[2], pd[0], pd[1], pd[2], (float)cov_norm, (float)C1, (float)C2,
                           (double *)part);
    else
        hipLaunchKernelGGL((ssim_combine_mean_kernel<double>), dim3(blocks), dim3(256), 0, s, (const double *)ux->data,
                           (const double *)uy->data, (const double *)uxx->data, (const double *)uyy->data, (const double *)uxy->data,
                           S ? (double *)S->data : nullptr, n[0], n[1], n[2], pd[0], pd[1], pd[2], cov_norm, C1, C2, (double *)part);
    hipError_t e = hipGetLastError();
    std::vector<double> host((size_t)blocks);
    if (e == hipSuccess) e = hipMemcpyAsync(host.data(), part, host.size() * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    pool_free(part);
    if (e != hipSuccess) { set_error("HIP error: %s", hipGetErrorString(e)); return MI_ERR_INTERNAL; }
    double t = 0.0;
    for (int k = 0; k < blocks; k++) t += host[k];
    *sum_out = t;
    return MI_OK;
}

int mi_sum(int op, const mi_array *a, const mi_array *b, double *result, mi_stream stream)
{
    int rc;
    if ((rc = check_array(a, "a"))) return rc;
    MI_REQUIRE(op >= 0 && op <= 2 && result, MI_ERR_INVALID_ARG, "unknown reduction");
    MI_REQUIRE(op != 1 || b, MI_ERR_INVALID_ARG, "squared difference needs two operands");
    if (b) {
        if ((rc = check_array(b, "b"))) return rc;
        MI_REQUIRE(same_shape(a, b) && a->dtype == b->dtype, MI_ERR_INVALID_ARG, "operands must agree in shape and dtype");
    }
    const int64_t n = numel(a);
    *result = 0.0;
    if (n == 0) return MI_OK;
    SumParams pa, pb;
    pa.ndim = pb.ndim = a->ndim;
    for (int d = 0; d < a->ndim; d++) {
        pa.shape[d] = pb.shape[d] = a->shape[d];
        pa.stride[d] = a->strides[d];
        pb.stride[d] = b ? b->strides[d] : 0;
    }
    // contiguous arrays are summed as rows of 4096 samples (a short last axis would leave most threads of a row idle)
    if (is_contiguous(a) && (!b || is_contiguous(b))) {
        const int64_t isz = (int64_t)dtype_size(a->dtype);
        const int64_t len = std::min<int64_t>(n, 4096);
        if (n % len == 0) {
            pa.ndim = pb.ndim = 2;
            pa.shape[0] = pb.shape[0] = n / len; pa.shape[1] = pb.shape[1] = len;
            pa.stride[0] = len * isz; pa.stride[1] = isz;
            pb.stride[0] = b ? len * isz : 0; pb.stride[1] = b ? isz : 0;
        }
    }
    const int64_t row_len = pa.ndim > 0 ? pa.shape[pa.ndim - 1] : 1;
    const int64_t nrows = n / row_len;
    const int blocks = (int)std::min<int64_t>(4096, nrows);
    void *part = nullptr;
    if ((rc = pool_alloc(&part, (size_t)blocks * sizeof(double), resolve_stream(stream)))) return rc;
    hipStream_t s = resolve_stream(stream);
    rc = dispatch_dtype(a->dtype, [&]<typename T>() -> int {
        hipLaunchKernelGGL((sum_kernel<T>), dim3(blocks), dim3(256), 0, s, (const char *)a->data,
                           b ? (const char *)b->data : nullptr, nrows, pa, pb, op, (double *)part);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
    if (rc == MI_OK) {
        std::vector<double> host((size_t)blocks);
        hipError_t e = hipMemcpyAsync(host.data(), part, host.size() * sizeof(double), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("HIP error: %s", hipGetErrorString(e)); rc = MI_ERR_INTERNAL; }
        else {
            double t = 0.0;
            for (int k = 0; k < blocks; k++) t += host[k];
            *result = t;
        }
    }
    pool_free(part);
    return rc;
}

}  // extern "C"
