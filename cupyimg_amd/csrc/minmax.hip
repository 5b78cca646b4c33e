// minmax.hip -- generic min / max filters (K2): one axis, and n-D footprint
// with optional non-flat structure (grey erosion / dilation).
//
// Reference: cupyimg/scipy/ndimage/filters.py:1373-1557 (launch sites :1505
// and :1417, kernel :1510-1557), morphology.py:769-884.
//
// Integer exactness follows SciPy (the reference's test oracle):
//   * 1-D passes compare doubles (SciPy line buffers; the reference also
//     promotes to double there, filters.py:1522-1528);
//   * the n-D footprint kernel converts cval to the *input* dtype, evaluates
//     the first set tap in double and adds the structure value to every later
//     tap in the input dtype (unsigned types wrap), then compares as double.
#include "nd_common.hpp"

namespace mi { void note_kernel(const char *fmt, ...); }      // runtime.hip
namespace mi {

template <typename T, typename I>
__global__ void __launch_bounds__(256)
minmax1d_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, I n, I inner, I total,
                int size, int off, int mode, double cval, int is_max, int geom)
{
    for_each_line_output<I>(geom, n, inner, total, [&](I i, I l, I base) {
        const I first = l - (I)off;
        const bool inside = first >= 0 && first + (I)size <= n;      // whole window inside: no boundary map
        double best = 0.0;
        if (inside) {
            const T *__restrict__ pf = in + base + first * inner;
            int k = 0;
            for (; k + 4 <= size; k += 4) {          // loads of a group first, then the comparisons in tap order
                T x[4];
#pragma unroll
                for (int u = 0; u < 4; u++) x[u] = pf[(I)(k + u) * inner];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const double v = (double)x[u];
                    if (k + u == 0 || (is_max ? v > best : v < best)) best = v;
                }
            }
            for (; k < size; k++) {
                const double v = (double)pf[(I)k * inner];
                if (k == 0 || (is_max ? v > best : v < best)) best = v;
            }
        } else {
            for (int k = 0; k < size; k++) {
                const I j = bmap<I>(first + (I)k, n, mode);
                const double v = j < 0 ? cval : (double)in[base + j * inner];
                if (k == 0 || (is_max ? v > best : v < best)) best = v;
            }
        }
        store_as(out, (int64_t)i, out_dt, best);
    });
}

// value + structure in the arithmetic of T (what `_tmp += (_type)ss` does)
template <typename T>
__device__ __forceinline__ double add_in_type(double v, double s)
{
    if constexpr (std::is_same<T, double>::value) return v + s;
    else if constexpr (std::is_same<T, float>::value) return (double)((float)v + (float)s);
    else if constexpr (std::is_same<T, bool>::value) return (double)(((int64_t)v + (int64_t)s) != 0);
    else return (double)(T)((int64_t)v + (int64_t)cast_from_f64<T>(s));
}

template <typename T, int ND>
__global__ void __launch_bounds__(256)
minmax_nd_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, NdGeom g, TapTable tt,
                 int64_t total, int mode, double cval, int is_max)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const Voxel<ND> v = locate<ND>(g, i);
        double best = 0.0;
        for (int t = 0; t < tt.ntaps; t++) {
            double x;
            if (v.interior) {
                x = (double)in[i + tt.lin[t]];
            } else {
                const int64_t pos = tap_pos<ND>(g, v, tt.idx, t, mode);
                x = pos < 0 ? cval : (double)in[pos];
            }
            if (tt.val) x = (t == 0) ? x + tt.val[0] : add_in_type<T>(x, tt.val[t]);
            if (t == 0 || (is_max ? x > best : x < best)) best = x;
        }
        store_as(out, i, out_dt, best);
    }
}

// rank <= 3 fast geometry (nd_common.hpp): same comparisons, same tap order
template <typename T>
__global__ void __launch_bounds__(256)
minmax3_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, Geom3 g, Taps3 tt, int mode, double cval,
               int is_max)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsTaps lt = stage_taps(tt, smem);
    const Vox3 v = locate3(g);
    if (!v.valid) return;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)((unsigned)g.nz * g.ny * g.nx * sizeof(T)), 0x00020000);
    double best = 0.0;
    auto fold = [&](int t, double x) {
        if (tt.has_val) x = (t == 0) ? x + lt.val[0] : add_in_type<T>(x, lt.val[t]);
        if (t == 0 || (is_max ? x > best : x < best)) best = x;
    };
    if (v.interior) {
        const unsigned base = (unsigned)v.lin * (unsigned)sizeof(T);
        int t0 = 0;
        for (; t0 + 8 <= tt.ntaps; t0 += 8) {
            T raw[8];
#pragma unroll
            for (int k = 0; k < 8; k++) raw[k] = buf_load<T>(rin, base + (unsigned)(lt.lin[t0 + k] * (int)sizeof(T)));
#pragma unroll
            for (int k = 0; k < 8; k++) fold(t0 + k, (double)raw[k]);
        }
        for (; t0 < tt.ntaps; t0++) fold(t0, (double)buf_load<T>(rin, base + (unsigned)(lt.lin[t0] * (int)sizeof(T))));
    } else {
        for (int t = 0; t < tt.ntaps; t++) {
            const int pos = tap_pos3(g, v, lt, t, mode);
            fold(t, pos < 0 ? cval : (double)buf_load<T>(rin, (unsigned)pos * (unsigned)sizeof(T)));
        }
    }
    store_as(out, v.lin, out_dt, best);
}

// rank / median / percentile filters (reference: filters.py:1560-1848, kernel
// _get_rank_kernel :1716-1800): the footprint's samples are gathered like in
// minmax3_kernel and the `rank`-th smallest is found by partial selection on a
// per-thread array (footprints up to kMaxRankTaps samples, rank <= 3 arrays).
constexpr int kMaxRankTaps = 128;

template <typename T>
__global__ void __launch_bounds__(256)
rank3_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, Geom3 g, Taps3 tt, int mode, double cval, int rank)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsTaps lt = stage_taps(tt, smem);
    const Vox3 v = locate3(g);
    if (!v.valid) return;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)((unsigned)g.nz * g.ny * g.nx * sizeof(T)), 0x00020000);
    double vals[kMaxRankTaps];
    const int n = tt.ntaps;
    if (v.interior) {
        const unsigned base = (unsigned)v.lin * (unsigned)sizeof(T);
        for (int t = 0; t < n; t++) vals[t] = (double)buf_load<T>(rin, base + (unsigned)(lt.lin[t] * (int)sizeof(T)));
    } else {
        for (int t = 0; t < n; t++) {
            const int pos = tap_pos3(g, v, lt, t, mode);
            vals[t] = pos < 0 ? cval : (double)buf_load<T>(rin, (unsigned)pos * (unsigned)sizeof(T));
        }
    }
    // partial selection from whichever end is closer
    double res;
    if (rank <= n / 2) {
        for (int k = 0; k <= rank; k++) {
            int m = k;
            for (int j = k + 1; j < n; j++) if (vals[j] < vals[m]) m = j;
            const double tmp = vals[k]; vals[k] = vals[m]; vals[m] = tmp;
        }
        res = vals[rank];
    } else {
        const int r = n - 1 - rank;
        for (int k = 0; k <= r; k++) {
            int m = k;
            for (int j = k + 1; j < n; j++) if (vals[j] > vals[m]) m = j;
            const double tmp = vals[k]; vals[k] = vals[m]; vals[m] = tmp;
        }
        res = vals[r];
    }
    store_as(out, v.lin, out_dt, res);
}

// Rank filter for everything the register kernels above do not take (arrays of rank 4 .. 8, footprints of more
// than kMaxRankTaps samples, volumes beyond the 32-bit geometry): the window of a voxel is gathered into a scratch
// column in device memory (element t of thread q at scratch[t * nthreads + q]: neighbouring threads touch
// neighbouring addresses) and shell-sorted there, the counterpart of the reference's per-thread shell sort
// (cupyimg/scipy/ndimage/filters.py:1753-1768, 1829-1835), which has no size limit either.  The window is sorted in
// the input dtype and the selected element is stored THROUGH double, on purpose: SciPy's NI_RankFilter (the parity
// oracle, ni_filters.c) gathers the window into a double buffer and casts the selected double back, so 64-bit
// integers beyond 2^53 come back rounded there, and rounding is monotonic -- the rank-th smallest of the rounded
// samples is the rounded rank-th smallest -- so this kernel and rank3_kernel (double samples) agree with SciPy bit for
// bit; the reference's native-dtype sort would differ from SciPy in exactly those elements (r3 advisor finding:
// answered by keeping SciPy's behaviour; tests/test_gpu_vs_oracle.py::test_rank_filter_int64_beyond_2p53_follows_scipy).
// A correctness path: ~n log^2 n scratch accesses per voxel.
template <typename T, int ND>
__global__ void __launch_bounds__(256)
rank_nd_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, NdGeom g, TapTable tt, int64_t total, int mode,
               T cval, int rank, T *__restrict__ scratch, int64_t nthreads)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nthreads) return;
    T *v = scratch + q;
    const int n = tt.ntaps;
    for (int64_t i = q; i < total; i += nthreads) {
        const Voxel<ND> vx = locate<ND>(g, i);
        for (int t = 0; t < n; t++) {
            T x;
            if (vx.interior) {
                x = in[i + tt.lin[t]];
            } else {
                const int64_t pos = tap_pos<ND>(g, vx, tt.idx, t, mode);
                x = pos < 0 ? cval : in[pos];
            }
            v[(int64_t)t * nthreads] = x;
        }
        for (int gap = n / 2; gap > 0; gap = gap == 2 ? 1 : (int)(gap / 2.2)) {
            for (int j = gap; j < n; j++) {
                const T tmp = v[(int64_t)j * nthreads];
                int k = j;
                while (k >= gap) {
                    const T lo = v[(int64_t)(k - gap) * nthreads];
                    if (!(lo > tmp)) break;
                    v[(int64_t)k * nthreads] = lo;
                    k -= gap;
                }
                v[(int64_t)k * nthreads] = tmp;
            }
        }
        store_as(out, i, out_dt, (double)v[(int64_t)rank * nthreads]);
    }
}

// sorting-network rank kernel: rank_sorted.hpp, instantiated in rank_sorted_*.hip
template <typename T, typename V, int P>
int run_rank_sorted(const T *in, T *out, const Geom3 &g, const Taps3 &tt, int mode, V cval, int rank, hipStream_t s);
template <typename T, typename V, int N>
int run_median_sorted(const T *in, T *out, const Geom3 &g, const Taps3 &tt, int mode, V cval, hipStream_t s);   // rank_sorted_med.hip
template <typename T>
int run_rank27(const T *in, T *out, int64_t nz, int64_t ny, int64_t nx, int mode, double cval, int rank, hipStream_t s);      // median3d*.hip

}  // namespace mi

namespace mi {
int minmax3_tiled(const mi_array *in, const mi_array *out, const uint8_t *footprint, const int64_t *fshape,
                      const int *origins, int mode, double cval, bool is_max, hipStream_t s);   // stencil3d.hip
}

using namespace mi;

// test hook (not part of the C-ABI): 0 = never use the LDS-tiled kernel
static mi::Knob g_rank_sorted{1};     // test hook: 0 = rank filters always use the selection kernel
static mi::Knob g_rank_median{1};     // test hook: 0 = medians of 25 / 27 samples take the full 32-sample network
extern "C" int mi_debug_set_rank_median(int enabled) { g_rank_median = enabled; return MI_OK; }
extern "C" int mi_debug_set_rank_sorted(int enabled) { g_rank_sorted = enabled; return MI_OK; }
static mi::Knob g_minmax_tiled{1};
extern "C" int mi_debug_set_minmax_tiled(int enabled) { g_minmax_tiled = enabled; return MI_OK; }

extern "C" {

int mi_minmax1d(const mi_array *in, const mi_array *out, int axis, int size, int origin, int mode,
                double cval, int is_max, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1, MI_ERR_INVALID_ARG, "input must have at least one dimension");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(axis >= 0 && axis < in->ndim, MI_ERR_INVALID_ARG, "invalid axis");
    MI_REQUIRE(size >= 1, MI_ERR_INVALID_ARG, "incorrect filter size");
    MI_REQUIRE(size / 2 + origin >= 0 && size / 2 + origin < size, MI_ERR_INVALID_ARG, "invalid origin");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS,
               "min/max filter needs C-contiguous arrays");
    MI_REQUIRE(in->data != out->data, MI_ERR_INVALID_ARG, "in-place filtering is not supported by the kernel");
    const int64_t total = numel(in);
    if (total == 0) return MI_OK;
    int64_t n = in->shape[axis], inner = 1;
    for (int d = axis + 1; d < in->ndim; d++) inner *= in->shape[d];
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
            hipLaunchKernelGGL((minmax1d_kernel<T, int64_t>), grid, block, 0, s, ip, out->data, out->dtype,
                               n, inner, total, size, off, mode, cval, is_max, geom);
        else
            hipLaunchKernelGGL((minmax1d_kernel<T, int32_t>), grid, block, 0, s, ip, out->data, out->dtype,
                               (int32_t)n, (int32_t)inner, (int32_t)total, size, off, mode, cval, is_max, geom);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

int mi_minmax_nd(const mi_array *in, const mi_array *out, const uint8_t *footprint,
                 const double *structure, const int64_t *fshape, const int *origins, int mode,
                 double cval, int is_max, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1, MI_ERR_INVALID_ARG, "input must have at least one dimension");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(footprint && fshape && origins, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS,
               "min/max filter needs C-contiguous arrays");
    MI_REQUIRE(in->data != out->data, MI_ERR_INVALID_ARG, "in-place filtering is not supported by the kernel");
    const int64_t total = numel(in);
    if (total == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    mode = filter_mode(mode);
    if (g_minmax_tiled && !structure && in->dtype == out->dtype &&
        (in->dtype == MI_F32 || in->dtype == MI_U8 || in->dtype == MI_U16 || in->dtype == MI_I16)) {
        // cval is converted to the input dtype first (SciPy: `_cv = (_type)_cval`)
        double cv = cval;
        if (in->dtype == MI_F32) cv = (double)(float)cval;
        else if (in->dtype == MI_U8) cv = (double)(uint8_t)(int64_t)cval;
        else if (in->dtype == MI_U16) cv = (double)(uint16_t)(int64_t)cval;
        else cv = (double)(int16_t)(int64_t)cval;
        rc = minmax3_tiled(in, out, footprint, fshape, origins, mode, cv, is_max != 0, s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }

    Taps3Builder t3;
    Taps3 tt3;
    const bool fast3 = Taps3Builder::eligible(in, fshape);
    TapBuilder tb;
    TapTable tt;
    if (fast3) {
        if ((rc = t3.build(in, fshape, origins, [&](int64_t k) { return footprint[k] != 0; },
                           [&](int64_t k) { return is_max ? structure[k] : -structure[k]; }, structure != nullptr)))
            return rc;
        MI_REQUIRE(!t3.lin.empty(), MI_ERR_INVALID_ARG, "all-zero footprint is not supported");
        if ((rc = t3.finish(&tt3, s))) return rc;
    } else {
        if ((rc = tb.init(in, fshape, origins, "footprint"))) return rc;
        tb.fill([&](int64_t k) { return footprint[k] != 0; },
                [&](int64_t k) { return is_max ? structure[k] : -structure[k]; }, structure != nullptr);
        MI_REQUIRE(!tb.lin.empty(), MI_ERR_INVALID_ARG, "all-zero footprint is not supported");
        if ((rc = tb.upload(&tt, s))) return rc;
    }

    dim3 grid;
    grid_for(total, 256, &grid);
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        // cval is converted to the input dtype first (SciPy: `_cv = (_type)_cval`)
        double cv;
        if constexpr (std::is_same<T, double>::value) cv = cval;
        else if constexpr (std::is_same<T, float>::value) cv = (double)(float)cval;
        else if constexpr (std::is_same<T, bool>::value) cv = cval != 0.0;
        else if constexpr (std::is_same<T, uint64_t>::value)
            cv = (double)(cval >= 0 ? (uint64_t)cval : (uint64_t)(-(int64_t)(uint64_t)(-cval)));
        else cv = (double)(T)(int64_t)cval;
        const T *ip = (const T *)in->data;
        if (fast3) {
            hipLaunchKernelGGL((minmax3_kernel<T>), grid3(t3.g), dim3(64, 4, 1), taps3_lds_bytes(tt3), s, ip, out->data, out->dtype, t3.g,
                               tt3, mode, cv, is_max);
            MI_HIP(hipGetLastError());
            return MI_OK;
        }
        if (tb.g.ndim == 3)
            hipLaunchKernelGGL((minmax_nd_kernel<T, 3>), grid, dim3(256), 0, s, ip, out->data, out->dtype, tb.g,
                               tt, total, mode, cv, is_max);
        else
            hipLaunchKernelGGL((minmax_nd_kernel<T, MI_MAX_NDIM>), grid, dim3(256), 0, s, ip, out->data,
                               out->dtype, tb.g, tt, total, mode, cv, is_max);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

int mi_rank_filter(const mi_array *in, const mi_array *out, const uint8_t *footprint, const int64_t *fshape,
                   const int *origins, int rank, int mode, double cval, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1, MI_ERR_INVALID_ARG, "input must have at least one dimension");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(footprint && fshape && origins, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS, "rank filter needs C-contiguous arrays");
    MI_REQUIRE(in->data != out->data, MI_ERR_INVALID_ARG, "in-place filtering is not supported by the kernel");
    if (numel(in) == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    mode = filter_mode(mode);
    int64_t nset = 0, fsize = 1;
    for (int d = 0; d < in->ndim; d++) fsize *= fshape[d];
    for (int64_t k = 0; k < fsize; k++) nset += footprint[k] != 0;
    MI_REQUIRE(nset > 0, MI_ERR_INVALID_ARG, "all-zero footprint is not supported");
    MI_REQUIRE(rank >= 0 && rank < nset, MI_ERR_INVALID_ARG, "rank not within filter footprint size");
    if (!Taps3Builder::eligible(in, fshape) || nset > kMaxRankTaps) {
        // r3: any rank up to 8, any footprint size: window in a scratch column, shell sort (rank_nd_kernel)
        TapBuilder tb;
        TapTable tt;
        if ((rc = tb.init(in, fshape, origins, "footprint"))) return rc;
        tb.fill([&](int64_t k) { return footprint[k] != 0; }, [](int64_t) { return 0.0; }, false);
        if ((rc = tb.upload(&tt, s))) return rc;
        const int64_t total = numel(in);
        const size_t esz = dtype_size(in->dtype);
        // threads: as many as a 256 MiB scratch holds columns for, at most 64 Ki, at least one workgroup
        int64_t nthreads = (int64_t)(((size_t)256 << 20) / ((size_t)nset * esz));
        nthreads = std::min<int64_t>(nthreads, 65536);
        nthreads = std::min<int64_t>(nthreads, (total + 255) / 256 * 256);
        nthreads = std::max<int64_t>(nthreads / 256 * 256, 256);
        void *scratch = nullptr;
        if ((rc = pool_alloc(&scratch, (size_t)nset * (size_t)nthreads * esz, s))) return rc;
        rc = dispatch_dtype(in->dtype, [&]<typename T>() -> int {
            T cv;
            if constexpr (std::is_same<T, double>::value) cv = cval;
            else if constexpr (std::is_same<T, float>::value) cv = (float)cval;
            else if constexpr (std::is_same<T, bool>::value) cv = cval != 0.0;
            else if constexpr (std::is_same<T, uint64_t>::value) cv = cval >= 0 ? (uint64_t)cval : (uint64_t)(-(int64_t)(uint64_t)(-cval));
            else cv = (T)(int64_t)cval;
            const dim3 grid((unsigned)(nthreads / 256));
            note_kernel("mi::rank_nd_kernel ntaps=%d rank=%d threads=%d (window in a scratch column)", (int)nset, rank, (int)nthreads);
            if (tb.g.ndim == 3)
                hipLaunchKernelGGL((rank_nd_kernel<T, 3>), grid, dim3(256), 0, s, (const T *)in->data, out->data, out->dtype, tb.g, tt,
                                   total, mode, cv, rank, (T *)scratch, nthreads);
            else
                hipLaunchKernelGGL((rank_nd_kernel<T, MI_MAX_NDIM>), grid, dim3(256), 0, s, (const T *)in->data, out->data, out->dtype,
                                   tb.g, tt, total, mode, cv, rank, (T *)scratch, nthreads);
            MI_HIP(hipGetLastError());
            return MI_OK;
        });
        pool_free(scratch);             // reuse is stream ordered
        return rc;
    }
    Taps3Builder t3;
    Taps3 tt3;
    if ((rc = t3.build(in, fshape, origins, [&](int64_t k) { return footprint[k] != 0; }, [](int64_t) { return 0.0; }, false)))
        return rc;
    if ((rc = t3.finish(&tt3, s))) return rc;
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        double cv;
        if constexpr (std::is_same<T, double>::value) cv = cval;
        else if constexpr (std::is_same<T, float>::value) cv = (double)(float)cval;
        else if constexpr (std::is_same<T, bool>::value) cv = cval != 0.0;
        else if constexpr (std::is_same<T, uint64_t>::value)
            cv = (double)(cval >= 0 ? (uint64_t)cval : (uint64_t)(-(int64_t)(uint64_t)(-cval)));
        else cv = (double)(T)(int64_t)cval;
        constexpr bool as_float = std::is_same<T, float>::value || std::is_same<T, uint8_t>::value ||
                                  std::is_same<T, int8_t>::value || std::is_same<T, uint16_t>::value ||
                                  std::is_same<T, int16_t>::value;
        constexpr bool as_double = std::is_same<T, double>::value || std::is_same<T, int32_t>::value ||
                                   std::is_same<T, uint32_t>::value;
        if constexpr (as_float || as_double) {
            using V = std::conditional_t<as_float, float, double>;
            // 65..128 samples (r5: 5 x 5 x 5, 9 x 9, 11 x 11): 128 key registers -- every type with 32-bit keys (float64 would spill)
            constexpr bool key32 = !std::is_same<T, double>::value;
            if (tt3.ntaps <= (key32 ? 128 : 64) && out->dtype == in->dtype && g_rank_sorted) {
                const T *ip = (const T *)in->data;
                T *op = (T *)out->data;
                {
                    // r5: the ranks of the full 3 x 3 x 3 window of a volume -- the kernel that shares its sorting between windows
                    if (tt3.ntaps == 27 && t3.g.wz == 3 && t3.g.wy == 3 && t3.g.wx == 3 && t3.g.oz == 1 && t3.g.oy == 1 && t3.g.ox == 1) {
                        const int r27 = run_rank27<T>(ip, op, t3.g.nz, t3.g.ny, t3.g.nx, mode, cv, rank, s);
                        if (r27 != MI_ERR_UNSUPPORTED) return r27;
                    }
                }
                if constexpr (std::is_same<T, float>::value || std::is_same<T, uint8_t>::value || std::is_same<T, uint16_t>::value ||
                              std::is_same<T, int16_t>::value) {
                    // the medians of 5 x 5 and 3 x 3 x 3 windows: network pruned for the one output that is needed
                    if (g_rank_median && (tt3.ntaps == 25 || tt3.ntaps == 27) && rank == tt3.ntaps / 2)
                        note_kernel("mi::rank3_sorted_kernel<32,%d,%d> (register sorting network pruned for the median)", tt3.ntaps, rank);
                    if (g_rank_median && tt3.ntaps == 25 && rank == 12) return run_median_sorted<T, V, 25>(ip, op, t3.g, tt3, mode, (V)cv, s);
                    if (g_rank_median && tt3.ntaps == 27 && rank == 13) return run_median_sorted<T, V, 27>(ip, op, t3.g, tt3, mode, (V)cv, s);
                }
                note_kernel("mi::rank3_sorted_kernel<%s,%d,> ntaps=%d rank=%d (register sorting network)", sizeof(V) == 4 ? "float" : "double",
                            tt3.ntaps <= 16 ? 16 : (tt3.ntaps <= 32 ? 32 : (tt3.ntaps <= 64 ? 64 : 128)), tt3.ntaps, rank);
                if (tt3.ntaps <= 16) return run_rank_sorted<T, V, 16>(ip, op, t3.g, tt3, mode, (V)cv, rank, s);
                if (tt3.ntaps <= 32) return run_rank_sorted<T, V, 32>(ip, op, t3.g, tt3, mode, (V)cv, rank, s);
                if (tt3.ntaps <= 64) return run_rank_sorted<T, V, 64>(ip, op, t3.g, tt3, mode, (V)cv, rank, s);
                if constexpr (key32) return run_rank_sorted<T, V, 128>(ip, op, t3.g, tt3, mode, (V)cv, rank, s);
            }
        }
        note_kernel("mi::rank3_kernel ntaps=%d rank=%d (selection in a per-thread array)", tt3.ntaps, rank);
        hipLaunchKernelGGL((rank3_kernel<T>), grid3(t3.g), dim3(64, 4, 1), taps3_lds_bytes(tt3), s, (const T *)in->data, out->data,
                           out->dtype, t3.g, tt3, mode, cv, rank);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}

}  // extern "C"
