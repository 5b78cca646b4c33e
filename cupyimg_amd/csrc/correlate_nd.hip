// correlate_nd.hip -- dense n-D correlate (K1 for correlate/convolve).
//
// Reference launch site: cupyimg/scipy/ndimage/filters.py:65-210 -> :441-495
// -> _filters_core.py:112-156; generated body _filters_core.py:298-324.
// out[o] = sum over non-zero taps t (C order) of w[t] * ext(in)[o - off + t],
// accumulated in double from 0 in that order (SciPy's NI_Correlate does the
// same, so float64 results are bit-identical; built with -ffp-contract=off).
// In constant mode a tap is cval as soon as one axis is outside (:276-293).
#include "nd_common.hpp"

namespace mi {

template <typename T, typename Acc, int ND>
__global__ void __launch_bounds__(256)
corr_nd_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, NdGeom g, TapTable tt,
               int64_t total, int mode, double cval)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const Voxel<ND> v = locate<ND>(g, i);
        Acc acc = 0;
        if (v.interior) {
            for (int t = 0; t < tt.ntaps; t++)
                acc += (Acc)in[i + tt.lin[t]] * (Acc)tt.val[t];
        } else {
            for (int t = 0; t < tt.ntaps; t++) {
                const int64_t pos = tap_pos<ND>(g, v, tt.idx, t, mode);
                const Acc x = pos < 0 ? (Acc)cval : (Acc)in[pos];
                acc += x * (Acc)tt.val[t];
            }
        }
        store_as(out, i, out_dt, (double)acc);
    }
}

// rank <= 3 fast geometry (nd_common.hpp): same arithmetic, same tap order
template <typename T, typename Acc>
__global__ void __launch_bounds__(256)
corr3_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, Geom3 g, Taps3 tt, int mode, double cval)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsTaps lt = stage_taps(tt, smem);
    const Vox3 v = locate3(g);
    if (!v.valid) return;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)((unsigned)g.nz * g.ny * g.nx * sizeof(T)), 0x00020000);
    Acc acc = 0;
    if (v.interior) {
        // taps in groups of 8: the group's loads are issued back to back (one memory
        // latency per group instead of per tap), then accumulated in tap order
        const unsigned base = (unsigned)v.lin * (unsigned)sizeof(T);
        int t0 = 0;
        for (; t0 + 8 <= tt.ntaps; t0 += 8) {
            T x[8];
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = buf_load<T>(rin, base + (unsigned)(lt.lin[t0 + k] * (int)sizeof(T)));
#pragma unroll
            for (int k = 0; k < 8; k++) acc += (Acc)x[k] * (Acc)lt.val[t0 + k];
        }
        for (; t0 < tt.ntaps; t0++)
            acc += (Acc)buf_load<T>(rin, base + (unsigned)(lt.lin[t0] * (int)sizeof(T))) * (Acc)lt.val[t0];
    } else {
        for (int t = 0; t < tt.ntaps; t++) {
            const int pos = tap_pos3(g, v, lt, t, mode);
            const Acc x = pos < 0 ? (Acc)cval : (Acc)buf_load<T>(rin, (unsigned)pos * (unsigned)sizeof(T));
            acc += x * (Acc)lt.val[t];
        }
    }
    store_as(out, v.lin, out_dt, (double)acc);
}

}  // namespace mi

namespace mi {
int stencil3_tiled(const mi_array *in, const mi_array *out, const double *weights, const int64_t *wshape,
                 const int *origins, int mode, double cval, bool acc_f32, hipStream_t s);   // stencil3d.hip
int stencil3_scatter(const mi_array *in, const mi_array *out, const double *weights, const int64_t *wshape,
                     const int *origins, int mode, double cval, bool acc_f32, hipStream_t s);   // stencil3s.hip
}

using namespace mi;

// test hook (not part of the C-ABI): 0 = never use the LDS-tiled stencil kernel
static mi::Knob g_stencil_enabled{1};
extern "C" int mi_debug_set_stencil(int enabled) { g_stencil_enabled = enabled; return MI_OK; }

// The dense 3 / 5 / 7-cubed window on a float32 volume through stencil3s_kernel ONLY -- the one stencil kernel that takes rows of any
// length -- or nothing at all: what the Python layer asks first for rows that are not a multiple of 16 bytes, before it
// extends them for the tiled kernel (mi_correlate_nd itself never refuses: it ends at the generic gather kernel).
extern "C" int mi_correlate3_dense(const mi_array *in, const mi_array *out, const double *weights,
                                   const int64_t *wshape, const int *origins, int mode, double cval,
                                   int acc_f32, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(weights && wshape && origins, MI_ERR_INVALID_ARG, "NULL argument");
    if (in->ndim != 3 || !is_contiguous(in) || !is_contiguous(out) || in->data == out->data || in->dtype != out->dtype || !g_stencil_enabled) {
        set_error("correlate3_dense: contiguous, distinct 3-D arrays of one dtype only");
        return MI_ERR_UNSUPPORTED;
    }
    if (numel(in) == 0) return MI_OK;
    return stencil3_scatter(in, out, weights, wshape, origins, filter_mode(mode), cval, acc_f32 && in->dtype == MI_F32, resolve_stream(stream));
}

extern "C" int mi_correlate_nd(const mi_array *in, const mi_array *out, const double *weights,
                               const int64_t *wshape, const int *origins, int mode, double cval,
                               int acc_f32, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1, MI_ERR_INVALID_ARG, "input must have at least one dimension");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(weights && wshape && origins, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS,
               "correlate needs C-contiguous arrays");
    MI_REQUIRE(in->data != out->data, MI_ERR_INVALID_ARG, "in-place filtering is not supported by the kernel");
    const int64_t total = numel(in);
    if (total == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    mode = filter_mode(mode);

    const bool f32ok = in->dtype == MI_F32 || in->dtype == MI_BOOL || dtype_size(in->dtype) <= 2;
    const bool use_f32 = acc_f32 && f32ok;
    if (g_stencil_enabled && in->dtype == out->dtype) {
        rc = stencil3_scatter(in, out, weights, wshape, origins, mode, cval, use_f32, s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
        rc = stencil3_tiled(in, out, weights, wshape, origins, mode, cval, use_f32, s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    if (Taps3Builder::eligible(in, wshape)) {
        Taps3Builder t3;
        if ((rc = t3.build(in, wshape, origins, [&](int64_t k) { return weights[k] != 0.0; },
                           [&](int64_t k) { return weights[k]; }, true))) return rc;
        Taps3 tt3;
        if ((rc = t3.finish(&tt3, s))) return rc;
        return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
            const T *ip = (const T *)in->data;
            if (use_f32)
                hipLaunchKernelGGL((corr3_kernel<T, float>), grid3(t3.g), dim3(64, 4, 1), taps3_lds_bytes(tt3), s, ip, out->data, out->dtype,
                                   t3.g, tt3, mode, cval);
            else
                hipLaunchKernelGGL((corr3_kernel<T, double>), grid3(t3.g), dim3(64, 4, 1), taps3_lds_bytes(tt3), s, ip, out->data, out->dtype,
                                   t3.g, tt3, mode, cval);
            MI_HIP(hipGetLastError());
            return MI_OK;
        });
    }
    TapBuilder tb;
    if ((rc = tb.init(in, wshape, origins, "weights"))) return rc;
    tb.fill([&](int64_t k) { return weights[k] != 0.0; }, [&](int64_t k) { return weights[k]; }, true);
    TapTable tt;
    if ((rc = tb.upload(&tt, s))) return rc;
    dim3 grid;
    grid_for(total, 256, &grid);
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        const T *ip = (const T *)in->data;
#define MI_LAUNCH(ACC, NDV)                                                                         \
    hipLaunchKernelGGL((corr_nd_kernel<T, ACC, NDV>), grid, dim3(256), 0, s, ip, out->data, out->dtype, \
                       tb.g, tt, total, mode, cval)
        if (tb.g.ndim == 3) { if (use_f32) MI_LAUNCH(float, 3); else MI_LAUNCH(double, 3); }
        else                { if (use_f32) MI_LAUNCH(float, MI_MAX_NDIM); else MI_LAUNCH(double, MI_MAX_NDIM); }
#undef MI_LAUNCH
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}
