/*
 * mi355img.h -- C-ABI of libmi355img.so, the MI355X (gfx950) n-D image
 * filtering engine behind cupyimg_amd.
 *
 * The reference (mritools/cupyimg) has no FFI: its seam is Python-level.  Every
 * hot-path API function funnels into one of three launch sites, and each entry
 * point below replaces exactly one of them (reference file:line in the
 * comment above each declaration):
 *
 *   _filters_core._call_kernel(kernel, input, weights, output, structure)
 *       cupyimg/scipy/ndimage/_filters_core.py:112-156   (K1, K2 kernels)
 *   erode_kernel(input, structure[, mask], output)
 *       cupyimg/scipy/ndimage/morphology.py:292-322      (K4)
 *   kern(filtered, coordinates | matrix, output)
 *       cupyimg/scipy/ndimage/interpolation.py:393,545,560 (K5)
 *
 * plus the part of CuPy the reference leans on for memory/streams (L0 in
 * SURVEY.md), which this library owns itself: no CuPy, no PyTorch.
 *
 * Conventions
 *   - plain C linkage, POD arguments only; every function returns int:
 *       0 = ok, < 0 = MI_ERR_*, > 0 = hipError_t of the failing HIP call.
 *     mi_last_error() returns a thread-local human-readable message.
 *   - device arrays are described by mi_array (pointer + dtype + shape +
 *     byte strides).  Compute entry points require C-contiguous arrays
 *     (MI_ERR_NOT_CONTIGUOUS otherwise); mi_copy handles arbitrary strides
 *     and dtype conversion.
 *   - the caller owns all buffers; the library owns only its scratch pool,
 *     streams/events it created and RCCL communicators.
 *   - small host-side parameter arrays (weights, footprints, matrices) are
 *     copied during the call; they may be freed as soon as it returns.
 *   - all work is enqueued on `stream` (NULL = the library's per-device
 *     default stream) and is asynchronous with respect to the host.
 *   - thread-safe: no unsynchronised global state (the reference's tests call
 *     filters from 4 threads, tests/test_filters.py:354-412).
 */
#ifndef MI355IMG_H
#define MI355IMG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_MAX_NDIM 8
#define MI_VERSION 100 /* 0.1.0 */

typedef enum mi_dtype {
    MI_BOOL = 0, MI_I8 = 1, MI_U8 = 2, MI_I16 = 3, MI_U16 = 4, MI_I32 = 5,
    MI_U32 = 6, MI_I64 = 7, MI_U64 = 8, MI_F32 = 9, MI_F64 = 10,
    MI_F16 = 11   /* storage only: mi_copy converts to / from it (the reference keeps float16 results in float16,
                     _filters_core.py:169-171, computing in float32 / float64); every compute entry point answers
                     MI_ERR_INVALID_ARG for it and the Python layer converts around the call */
} mi_dtype;

/* scipy.ndimage boundary modes (cupyimg/scipy/ndimage/_util.py:105-119).
 * For filters WRAP == GRID_WRAP and GRID_CONSTANT == CONSTANT
 * (_filters_core.py:224-225); interpolation distinguishes them. */
typedef enum mi_mode {
    MI_MODE_REFLECT = 0,      /* also 'grid-mirror' */
    MI_MODE_CONSTANT = 1,
    MI_MODE_NEAREST = 2,
    MI_MODE_MIRROR = 3,
    MI_MODE_WRAP = 4,
    MI_MODE_GRID_WRAP = 5,
    MI_MODE_GRID_CONSTANT = 6
} mi_mode;

enum {
    MI_OK = 0,
    MI_ERR_INVALID_ARG = -1,
    MI_ERR_UNSUPPORTED = -2,   /* valid request, no kernel for it (host falls back) */
    MI_ERR_NOMEM = -3,
    MI_ERR_NOT_CONTIGUOUS = -4,
    MI_ERR_RCCL = -5,
    MI_ERR_INTERNAL = -6
};

typedef struct mi_array {
    void *data;                    /* device pointer */
    int32_t dtype;                 /* mi_dtype */
    int32_t ndim;                  /* 0..MI_MAX_NDIM */
    int64_t shape[MI_MAX_NDIM];
    int64_t strides[MI_MAX_NDIM];  /* bytes */
} mi_array;

typedef void *mi_stream; /* hipStream_t */
typedef void *mi_event;  /* hipEvent_t  */
typedef void *mi_comm;   /* ncclComm_t  */

/* ------------------------------------------------------------------ */
/* runtime: replaces the CuPy runtime the reference imports             */
/* (cupyimg/__init__.py:23-28; cupy.zeros at _util.py:80)               */
/* ------------------------------------------------------------------ */
int mi_version(void);
const char *mi_last_error(void);
int mi_device_count(int *count);
int mi_set_device(int device);
int mi_get_device(int *device);
int mi_device_name(int device, char *buf, size_t buflen);
int mi_device_attr(int device, int *cu_count, int *clock_khz, size_t *total_mem);
int mi_mem_info(size_t *free_bytes, size_t *total_bytes);

/* pooled device memory (size-bucketed free lists, mutex guarded) */
int mi_malloc(void **ptr, size_t nbytes);
int mi_free(void *ptr);
int mi_pool_trim(void);
int mi_pool_stats(size_t *bytes_in_use, size_t *bytes_cached);

int mi_memcpy_h2d(void *dst, const void *src, size_t nbytes, mi_stream stream);
int mi_memcpy_d2h(void *dst, const void *src, size_t nbytes, mi_stream stream); /* syncs */
int mi_memcpy_d2d(void *dst, const void *src, size_t nbytes, mi_stream stream);
int mi_memcpy_peer(void *dst, int dst_dev, const void *src, int src_dev, size_t nbytes,
                   mi_stream stream);
int mi_memset(void *dst, int value, size_t nbytes, mi_stream stream);

int mi_stream_create(mi_stream *stream);
int mi_stream_destroy(mi_stream stream);
int mi_stream_sync(mi_stream stream);
int mi_default_stream(mi_stream *stream);
int mi_device_sync(void);
int mi_event_create(mi_event *event);
int mi_event_destroy(mi_event event);
int mi_event_record(mi_event event, mi_stream stream);
int mi_stream_wait_event(mi_stream stream, mi_event event); /* later work on stream waits for event */
/* Later work on `waiter` waits for everything queued on `producer` so far.  Either
 * argument: a hipStream_t (also one owned by another runtime: torch, CuPy), NULL = the
 * library's default stream, or the __cuda_array_interface__ stream codes 1 (legacy
 * default stream) / 2 (per-thread default stream).  This is how zero-copy imports and
 * exports are ordered against the other runtime's stream: the library's default stream
 * is non-blocking, it never synchronises with the null stream by itself. */
int mi_stream_wait_stream(mi_stream waiter, mi_stream producer);
int mi_event_sync(mi_event event);
int mi_event_elapsed_ms(mi_event start, mi_event stop, float *ms);

/* ------------------------------------------------------------------ */
/* array plumbing: `output[...] = input` and dtype casts               */
/* (_filters_core.py:94,107,154; cast<> at :166-187)                    */
/* ------------------------------------------------------------------ */
/* dst[...] = (dst dtype) src, arbitrary strides, same shape.  Float ->
 * integer conversion truncates toward zero; negative -> unsigned wraps.
 * round_half_even != 0 applies rint() first (interpolation integer outputs,
 * _interp_kernels.py:580-583). */
int mi_copy(const mi_array *src, const mi_array *dst, int round_half_even, mi_stream stream);
int mi_fill(const mi_array *dst, double value, mi_stream stream);
/* r4b: rows that are not a multiple of 16 bytes (181 x 217 x 181 ...): out[..., x] = in[..., map(x - left)] for
 * x in [0, out.shape[-1]) -- every row extended along the last axis by a filter boundary mode (MI_MODE_CONSTANT: cval) --
 * and its inverse out[..., x] = in[..., left + x].  1-, 2-, 4- and 8-byte dtypes (r5: 8), C-contiguous; the extended rows and `left`
 * are multiples of 16 bytes and the extended array is 16-byte aligned (what the fused kernels need: the Python layer runs
 * them on the extended volume and copies the columns back; the reference has no counterpart -- its kernels index
 * element by element, _filters_core.py:190-348). */
int mi_extend_rows(const mi_array *in, const mi_array *out, int left, int mode, double cval, mi_stream stream);
int mi_crop_rows(const mi_array *in, const mi_array *out, int left, mi_stream stream);
/* *flag_dev (device int32) |= any(a != b); a, b contiguous, same dtype/shape */
int mi_any_diff(const mi_array *a, const mi_array *b, int32_t *flag_dev, mi_stream stream);
/* out = a (op) b, elementwise, one dtype, C-contiguous; op: 0 add, 1 subtract,
 * 2 multiply, 3 sqrt(a) (b may be NULL).  Integers wrap like NumPy's same-dtype
 * ufuncs.  Used by the composite filters built on the path (generic_laplace
 * `output += tmp` filters.py:1005-1011, generic_gradient_magnitude :1134-1147,
 * top-hats / morphological gradient morphology.py:887-1226). */
int mi_elementwise(int op, const mi_array *a, const mi_array *b, const mi_array *out, mi_stream stream);
/* float32 / float64: op 0: out = a * p0 + p1; op 1: out = clip(a, p0, p1), entries equal
 * to `keep` left alone when keep_flag != 0 (skimage warp's output clipping,
 * skimage/transform/_warps.py:745-787; dtype range scaling of img_as_float).  op 0 also
 * takes an 8- / 16-bit integer or bool input with a float32 / float64 output: conversion
 * and scaling in one pass. */
int mi_scalar_op(int op, const mi_array *a, const mi_array *out, double p0, double p1, double keep,
                 int keep_flag, mi_stream stream);
/* minimum and maximum of a contiguous array (synchronises the stream) */
int mi_min_max(const mi_array *a, double *lo, double *hi, mi_stream stream);
/* Reductions in double over an arbitrarily strided array (synchronises the stream): op 0 sum(a),
 * op 1 sum((a - b)^2), op 2 sum(a^2).  Serves the cropped mean of the SSIM map
 * (skimage/metrics/_structural_similarity.py:229-233) and skimage/metrics/simple_metrics.py. */
int mi_sum(int op, const mi_array *a, const mi_array *b, double *result, mi_stream stream);
/* x*x, y*y, x*y of two float images in one pass: the second-moment inputs of SSIM
 * (skimage/metrics/_structural_similarity.py:189-214 forms them with three multiplies). */
int mi_ssim_products(const mi_array *x, const mi_array *y, const mi_array *xx, const mi_array *yy,
                     const mi_array *xy, mi_stream stream);
/* SSIM map from the five filtered moments in one pass, float32 / float64, contiguous:
 * S = ((2 ux uy + C1)(2 vxy + C2)) / ((ux^2 + uy^2 + C1)(vx + vy + C2)), v* = cov_norm (u** - u* u*)
 * (skimage/metrics/_structural_similarity.py:206-227).  gA/gB/gC (all or none) receive the three
 * fields the gradient filters (A1/D, -S/B2, (ux(A2-A1) - uy(B2-B1)S)/D; :235-243). */
int mi_ssim_combine(const mi_array *ux, const mi_array *uy, const mi_array *uxx, const mi_array *uyy,
                    const mi_array *uxy, const mi_array *S, const mi_array *gA, const mi_array *gB,
                    const mi_array *gC, double cov_norm, double C1, double C2, mi_stream stream);
/* The SSIM map (optional: S may be NULL) and the SUM of the map cropped by `pad` samples on
 * every side (the mean of _structural_similarity.py:240-243) in the same pass; rank 1..3,
 * no gradient fields.  *sum_out is a host double (the call synchronises the stream). */
int mi_ssim_combine_mean(const mi_array *ux, const mi_array *uy, const mi_array *uxx,
                         const mi_array *uyy, const mi_array *uxy, const mi_array *S, int pad,
                         double cov_norm, double C1, double C2, double *sum_out,
                         mi_stream stream);

/* ------------------------------------------------------------------ */
/* K1: correlate family                                                 */
/* ------------------------------------------------------------------ */
/* One separable pass: out[.., o, ..] = sum_k w[k] * ext(in)[.., o - (wlen/2 + origin) + k, ..]
 * Replaces the K1 launch reached from correlate1d/convolve1d/gaussian_filter1d
 * (filters.py:213-283 -> :441-495 -> _filters_core.py:112-156).
 * acc_f32 != 0 selects float32 accumulation (dtype_mode="float",
 * _util.py:28-40), only honoured when promote(in, f32) == f32. */
int mi_correlate1d(const mi_array *in, const mi_array *out, int axis, const double *weights,
                   int wlen, int origin, int mode, double cval, int acc_f32, mi_stream stream);

/* Box mean along one axis with SciPy's sum-then-divide arithmetic (exact for
 * integer data); replaces the K1 launch behind uniform_filter1d
 * (filters.py:549-599). */
int mi_uniform_filter1d(const mi_array *in, const mi_array *out, int axis, int size, int origin,
                        int mode, double cval, mi_stream stream);

/* Fused separable 3-D filter, float32 -> float32, all three 1-D passes in one
 * launch (8 B/voxel of HBM traffic).  Replaces the three K1 launches plus the
 * zero-fills and copy-backs of uniform_filter / gaussian_filter
 * (filters.py:602-665, :725-792; _filters_core.py:148-155).
 * weights[a] (host, wlen[a] doubles) may be NULL for an axis that is not
 * filtered.  is_box != 0: all given weights are 1/wlen (sum-then-scale path).
 * Returns MI_ERR_UNSUPPORTED when no fused kernel covers the request. */
int mi_separable3d_f32(const mi_array *in, const mi_array *out, const double *const weights[3],
                       const int wlen[3], const int origin[3], const int mode[3], double cval,
                       int is_box, mi_stream stream);

/* Same filter restricted to one or two ranges of OUTPUT planes (axis 0):
 * planes = {begin0, end0[, begin1, end1]}, ascending and disjoint.  Boundary
 * handling still refers to the whole array; planes outside the ranges are
 * neither read for their own sake nor written.  New (the reference is
 * single-GPU): lets a slab rank filter its interior while the halo planes are
 * still in flight and finish the planes next to the halos afterwards. */
int mi_separable3d_f32_planes(const mi_array *in, const mi_array *out, const double *const weights[3],
                              const int wlen[3], const int origin[3], const int mode[3], double cval,
                              const int64_t *planes, int nranges, mi_stream stream);

/* Would mi_separable3d_f32 (plane_ranges = 0) / mi_separable3d_f32_planes (plane_ranges != 0) take this request?
 * MI_OK / MI_ERR_UNSUPPORTED / an argument error, decided by a dry run of the same dispatch code; nothing is
 * queued.  A plane-range request is judged as a partial one whatever ranges a caller would pass, so every rank of a
 * slab chain gets the same answer (r4; lets callers pick a schedule identically on all ranks). */
int mi_separable3d_f32_supports(const mi_array *in, const mi_array *out, const double *const weights[3],
                                const int wlen[3], const int origin[3], const int mode[3], double cval,
                                int plane_ranges);

/* Dense n-D stencil (filters.py:65-210 -> :441-495): weights is a host array
 * of prod(wshape) doubles in C order, already flipped for convolution by the
 * caller; zero weights are skipped (_filters_core.py:242-246). */
int mi_correlate_nd(const mi_array *in, const mi_array *out, const double *weights,
                    const int64_t *wshape, const int *origins, int mode, double cval,
                    int acc_f32, mi_stream stream);
/* The dense 3 x 3 x 3 / 5 x 5 x 5 (with acc_f32: 7 x 7 x 7) window without zero weights on a float32 volume through stencil3s_kernel only
 * -- rows of ANY length, x origin 0 -- or MI_ERR_UNSUPPORTED with nothing queued (mi_correlate_nd never refuses: it ends at the
 * generic gather kernel).  For callers that would otherwise extend ragged rows for the tiled kernel (filters.py:65-210). */
int mi_correlate3_dense(const mi_array *in, const mi_array *out, const double *weights,
                        const int64_t *wshape, const int *origins, int mode, double cval,
                        int acc_f32, mi_stream stream);

/* ------------------------------------------------------------------ */
/* K2: min / max family (grey morphology)                               */
/* ------------------------------------------------------------------ */
/* 1-D running min/max (filters.py:1478-1507), compared in double like SciPy's
 * line buffers. */
int mi_minmax1d(const mi_array *in, const mi_array *out, int axis, int size, int origin,
                int mode, double cval, int is_max, mi_stream stream);

/* Fused separable 3-D min/max for uint8 volumes (grey_erosion/grey_dilation
 * with `size`, morphology.py:769-884 -> filters.py:1385-1396): one launch,
 * 2 B/voxel.  MI_ERR_UNSUPPORTED when not applicable. */
int mi_minmax3d_u8(const mi_array *in, const mi_array *out, const int size[3],
                   const int origin[3], const int mode[3], int cval, int is_max,
                   mi_stream stream);

/* Same for float32 volumes (flat odd sizes <= 9 per axis, x origin 0): streaming
 * passes with the comparisons of the generic kernel, x window fused into the z
 * (or, for one-plane volumes = images, the y) pass.  cval is converted to
 * float32 first, as SciPy converts it to the input dtype. */
int mi_minmax3d_f32(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                    const int mode[3], double cval, int is_max, mi_stream stream);
/* mi_minmax3d_u8 / mi_minmax3d_f32 restricted to one or two ranges of output planes [b0, e0), [b1, e1) along axis 0
 * (ascending, inside the volume) -- the multi-GPU slab schedule filters a rank's interior planes while the halo exchange
 * is in flight and the planes next to a neighbour afterwards (r3).  Only what the single-launch kernels take: uint8
 * cubic sizes 3 / 5 / 7 with origin 0; float32 cubic odd sizes 3 .. 9, index-mapping boundary modes;
 * MI_ERR_UNSUPPORTED otherwise (callers then use the plain schedule on the whole extended slab). */
int mi_minmax3d_u8_planes(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                          const int mode[3], int cval, int is_max, const int64_t *planes, int nranges,
                          mi_stream stream);
int mi_minmax3d_f32_planes(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                           const int mode[3], double cval, int is_max, const int64_t *planes, int nranges,
                           mi_stream stream);


/* uint16 / int16 images and volumes (2-D or 3-D arrays; size / origin / mode always have
 * three entries, the first one for the absent axis of an image): flat min / max with odd
 * sizes <= 9 and origin 0 as streaming passes on packed 16-bit pairs -- images and
 * slice-wise filters one launch (4 B/pixel), volumes two.  MI_ERR_UNSUPPORTED otherwise. */
int mi_minmax3d_16(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                   const int mode[3], int cval, int is_max, mi_stream stream);

/* uniform_filter on a uint8 image (a uint8 volume: slice by slice) with a uint8 result
 * (filters.py:602-665 -> :549-599, every intermediate stored in the output dtype): one
 * streaming launch in integer arithmetic -- q = trunc(row sum / size[0]) as uint8, then
 * trunc(column sum of q / size[1]).  Odd sizes <= 9 (1 = axis not filtered), origin along y
 * only, mode[2] = y, x.  MI_ERR_UNSUPPORTED otherwise (-> mi_uniform_filter1d per axis). */
int mi_uniform2d_u8(const mi_array *in, const mi_array *out, const int size[2], int origin_y,
                    const int mode[2], int cval, mi_stream stream);
/* the same for uint16 / int16 images (32-bit sums) */
int mi_uniform2d_16(const mi_array *in, const mi_array *out, const int size[2], int origin_y,
                    const int mode[2], int cval, mi_stream stream);
/* The z pass of uniform_filter on a uint8 / 16-bit VOLUME, intermediate in the same dtype
 * (SciPy filters axis 0 first): trunc(sum of size_z planes / size_z), odd size 3 .. 9;
 * mi_uniform2d_* on the result completes the filter. */
int mi_uniform_z_u8(const mi_array *in, const mi_array *out, int size_z, int origin_z, int mode_z,
                    int cval, mi_stream stream);
int mi_uniform_z_16(const mi_array *in, const mi_array *out, int size_z, int origin_z, int mode_z,
                    int cval, mi_stream stream);

/* Flat footprint min / max on uint8 images (volumes: slice by slice) for footprints whose
 * rows are centred runs -- disk, diamond / cross, octagon, square, i.e. what skimage's
 * morphology passes (skimage/morphology/grey.py -> morphology.py:769-884 ->
 * filters.py:1398-1419): footprint row r (0 .. nrows-1, nrows odd <= 9) covers the columns
 * -half_width[r] .. +half_width[r] (<= 4; -1 = empty row), origin 0.  One streaming
 * launch, bit-exact.  mode[2]: y and x.  MI_ERR_UNSUPPORTED otherwise (-> mi_minmax_nd). */
int mi_minmax_runs_u8(const mi_array *in, const mi_array *out, int nrows, const int *half_width,
                      const int mode[2], int cval, int is_max, mi_stream stream);
/* 3 x 3 x 3 footprints of centred x runs on uint8 / bool volumes -- the 6- / 18- / 26-connected
 * structures of generate_binary_structure(3, k), morphology.py:174-201 -- as one streaming
 * launch: half_width[3 * (dz + 1) + (dy + 1)] in {-1 (no sample), 0 (centre voxel), 1 (three
 * voxels)}, mode[3] = z, y, x.  Grey erosion / dilation with such a footprint
 * (morphology.py:769-884 -> filters.py:1398-1419).  MI_ERR_UNSUPPORTED otherwise. */
int mi_minmax_runs3d_u8(const mi_array *in, const mi_array *out, const int half_width[9],
                        const int mode[3], int cval, int is_max, mi_stream stream);
/* the same for float32 images (compare-select like the generic kernel; cval converted to
 * float32 first) */
int mi_minmax_runs_f32(const mi_array *in, const mi_array *out, int nrows, const int *half_width,
                       const int mode[2], double cval, int is_max, mi_stream stream);
/* the same for uint16 / int16 images */
int mi_minmax_runs_16(const mi_array *in, const mi_array *out, int nrows, const int *half_width,
                      const int mode[2], int cval, int is_max, mi_stream stream);

/* float64 images and volumes (skimage's working dtype): the separable filter and the
 * flat min / max as streaming passes, x fused into the streamed pass when the tap
 * counts agree -- an image or a slice-wise filter is one launch (16 B/pixel), a volume
 * two.  weights[a] NULL = axis not filtered; odd tap counts <= 33 (min / max: odd
 * sizes <= 9), x origin 0, even x extent.  MI_ERR_UNSUPPORTED otherwise (the caller
 * runs mi_correlate1d / mi_minmax1d per axis). */
int mi_separable3d_f64(const mi_array *in, const mi_array *out, const double *const weights[3],
                       const int wlen[3], const int origin[3], const int mode[3], double cval,
                       mi_stream stream);
int mi_minmax3d_f64(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                    const int mode[3], double cval, int is_max, mi_stream stream);

/* n-D footprint (+ optional non-flat structure) min/max
 * (filters.py:1398-1419, kernel :1510-1557).  footprint: host uint8
 * prod(fshape); structure: host doubles or NULL. */
int mi_minmax_nd(const mi_array *in, const mi_array *out, const uint8_t *footprint,
                 const double *structure, const int64_t *fshape, const int *origins, int mode,
                 double cval, int is_max, mi_stream stream);

/* rank-th smallest sample under the footprint (rank_filter / median_filter /
 * percentile_filter, filters.py:1560-1848).  Arrays of rank <= 3 with footprints of at
 * most 128 set elements take register kernels (sorting networks / partial selection);
 * anything else -- rank 4..8, larger footprints -- a scratch-column shell sort (r3; the
 * reference's shell-sort path, filters.py:1753-1768, has no limit either; scratch from the
 * pool, at most 256 MiB).  cval is converted to the input dtype like SciPy does.
 * NaN contract (r5 advisor finding): float32 windows are sorted on order-preserving integer keys -- NaNs with a clear
 * sign bit sort above +inf, with a set sign bit below -inf, and -0 below +0 (what numpy.sort does for the positive
 * NaNs NumPy produces); float64 windows use compare-select / v_min_f64 / v_max_f64, which pass over NaNs: a float64
 * volume that holds NaNs has NO guaranteed rank-filter result (SciPy's own depends on its partial sort), finite data
 * and infinities are exact for both dtypes. */
int mi_rank_filter(const mi_array *in, const mi_array *out, const uint8_t *footprint,
                   const int64_t *fshape, const int *origins, int rank, int mode, double cval,
                   mi_stream stream);

/* 3 x 3 median over the last two axes of a 2-D / 3-D float32, float64, uint8, uint16 or int16 array
 * (median_filter(size=3) / rank_filter(rank=4) with a full 3 x 3 footprint,
 * filters.py:1560-1701,1751-1792; skimage.filters.median's default on images):
 * one streaming launch at 8 (float32) / 2 (uint8) / 4 (16-bit) B/pixel instead of the
 * generic gather-and-sort kernel.  mode[2]: boundary modes along y and x.
 * MI_ERR_UNSUPPORTED when not applicable (the caller runs mi_rank_filter). */
int mi_median3x3(const mi_array *in, const mi_array *out, const int mode[2], double cval,
                 mi_stream stream);

/* ------------------------------------------------------------------ */
/* K4: binary erosion / dilation                                        */
/* ------------------------------------------------------------------ */
/* One iteration of erode_kernel (morphology.py:41-128, launched at :292-322).
 * `in` any real dtype (nonzero = true); `out` any real dtype (0/1 written).
 * structure: host uint8 prod(sshape).  mask may be NULL.  invert = 1 turns
 * it into dilation (morphology.py:443-461).  If changed_dev is not NULL,
 * *changed_dev (device int32) is OR-ed with 1 when any output voxel differs
 * from the input voxel -- the on-device replacement for the host-side
 * `(tmp_in == tmp_out).all()` sync at morphology.py:313,321. */
int mi_binary_erosion(const mi_array *in, const mi_array *out, const uint8_t *structure,
                      const int64_t *sshape, const int *origins, const mi_array *mask,
                      int border_value, int invert, int32_t *changed_dev, mi_stream stream);

/* `iterations` (1 .. MI_BINARY_MAX_FUSED) iterations of the same erosion / dilation in ONE launch -- the host loop of
 * morphology.py:301-327 (one launch and one host synchronisation per iteration) folded into the tile residency: the
 * volume is staged as 1 bit per voxel and the intermediate results never reach HBM (csrc/bitmorph3d.hip).  The result
 * is that of `iterations` calls of mi_binary_erosion (mask, border_value, origins applied in every iteration).
 * changed_dev: NULL or a device int32[iterations]; element j is OR-ed with 1 when iteration j + 1 changed a voxel
 * (so a run "until stable" stops at the first zero).  3-D volumes of 1-byte voxels with rows that are a multiple of
 * 16 bytes; anything else returns MI_ERR_UNSUPPORTED with nothing queued and the caller iterates mi_binary_erosion. */
#define MI_BINARY_MAX_FUSED 8
/* binary_opening (closing = 0) / binary_closing (closing = 1) with `iterations` iterations of each half in ONE launch
 * (morphology.py:464-613 run the two halves as separate calls with a temporary volume): stages 1 .. iterations are the
 * first operation, the rest the second with the mirrored structure; 2 * iterations <= MI_BINARY_MAX_FUSED, odd structure
 * extents, origin 0; otherwise MI_ERR_UNSUPPORTED with nothing queued (the caller runs the two halves). */
int mi_binary_open_close_fused(const mi_array *in, const mi_array *out, const uint8_t *structure,
                               const int64_t *sshape, const mi_array *mask, int border_value, int closing,
                               int iterations, mi_stream stream);
/* One launch of a masked dilation's block-wise fill towards its fixed point -- binary_propagation / binary_fill_holes
 * (morphology.py:684-766: dilation with iterations = -1 inside a mask).  Every workgroup sweeps a block of the volume IN
 * PLACE (bits in LDS; whole runs of mask bits along x filled per sweep by the carry of an addition) until nothing inside
 * the block changes, its halo being what the neighbours held when the launch began; the operator is monotone, so repeating
 * launches (in -> out, ping-pong) until *changed_dev stays 0 reaches exactly the fixed point the reference's
 * one-iteration-per-launch loop reaches.  structure / origins in the form mi_binary_erosion takes with invert = 1
 * (already mirrored).  MI_ERR_UNSUPPORTED outside the envelope (3-D byte volumes with a byte mask, rows of 64 .. 2048
 * voxels, a structure that holds its centre): the caller iterates mi_binary_erosion_fused. */
int mi_binary_propagation_step(const mi_array *in, const mi_array *out, const uint8_t *structure,
                               const int64_t *sshape, const int *origins, const mi_array *mask,
                               int border_value, int32_t *changed_dev, mi_stream stream);
int mi_binary_erosion_fused(const mi_array *in, const mi_array *out, const uint8_t *structure,
                            const int64_t *sshape, const int *origins, const mi_array *mask,
                            int border_value, int invert, int iterations, int32_t *changed_dev, mi_stream stream);

/* ------------------------------------------------------------------ */
/* K5: interpolation, spline order 0 and 1                              */
/* ------------------------------------------------------------------ */
/* coords: (ndim, *out.shape) C-contiguous f32 or f64
 * (interpolation.py:271-394, kernel _interp_kernels.py:595-621). */
int mi_map_coordinates(const mi_array *in, const mi_array *coords, const mi_array *out,
                       int order, int mode, double cval, mi_stream stream);

/* matrix: host doubles (ndim, ndim+1) row-major, c = M[:, :n] @ o + M[:, n]
 * (interpolation.py:397-561, kernel _interp_kernels.py:723-751; the diagonal
 * zoom+shift form :655-688 is the same entry point with a diagonal M). */
int mi_affine_transform(const mi_array *in, const mi_array *out, const double *matrix,
                        int order, int mode, double cval, mi_stream stream);

/* B-spline interpolation of order 2..5 (interpolation.py:105-268 spline_filter,
 * :271-561 map_coordinates / affine_transform with order > 1; kernels
 * _spline_prefilter_core.py, _spline_kernel_weights.py, _interp_kernels.py:473-549).
 * The caller builds the float64 coefficient array -- mi_spline_pad (copy of any
 * real dtype, extended by npad samples per side on every axis: pad_mode 0 edge
 * replication, 1 cval; SciPy pads by 12 for `nearest` / `grid-constant`, else 0)
 * followed by mi_spline_filter1d along every axis (spline_mode 0 mirror,
 * 1 reflect, 2 grid-wrap: reflect for reflect / nearest, grid-wrap for grid-wrap,
 * mirror otherwise) -- and interpolates it; coordinates refer to the unpadded
 * array.  Rank <= 8 (r3; float32 coefficients: rank <= 3).
 * spline_mode | 0x100 (mi_spline_filter1d, mi_spline_prefilter): only the kernels whose
 * arithmetic is SciPy's operation for operation (no blocked recursion) -- pass it when the
 * interpolated result will be rounded to an integer dtype, where the last bit of a
 * coefficient decides exact .5 ties.
 * r5 -- axes that hold SAMPLES.  An affine transform that maps an axis onto itself with an
 * integral shift evaluates the spline at the samples of that axis, where it returns them: the
 * prefilter pass along the axis and the taps along it cancel (SciPy's `rotate` filters the two
 * axes of the rotation plane only; the reference's `rotate`, interpolation.py:576-709, filters
 * all of them -- the results agree to rounding).  mi_spline_prefilter with
 * spline_mode | MI_SPLINE_SKIP_AXIS(d) leaves axis d unfiltered;
 * mi_spline_affine_transform with order | MI_SPLINE_SAMPLES_AXIS(d) interpolates such an array
 * (order 3, float32 coefficients, rank 3, ONE such axis: x -- `rotate` with the default axes --
 * or the axis the streaming kernel walks along) and answers MI_ERR_UNSUPPORTED when no kernel
 * evaluates the axis as a single tap: the caller then filters it after all
 * (mi_spline_filter1d; the passes commute) and calls again without the flag. */
#define MI_SPLINE_SKIP_AXIS(d) (0x200 << (d))
#define MI_SPLINE_SAMPLES_AXIS(d) (0x100 << (d))
int mi_spline_pad(const mi_array *in, const mi_array *out, int npad, int pad_mode, double cval,
                  mi_stream stream);
int mi_spline_filter1d(const mi_array *data, int axis, int order, int spline_mode, mi_stream stream);
/* mi_spline_pad followed by mi_spline_filter1d along every axis longer than one sample, in one call
 * (the loop of spline_filter, interpolation.py:185-268); without padding and conversion the first
 * pass reads `in` directly instead of copying it first.
 * Precision of the streaming passes (orders 2 / 3 on volumes of >= 16384 lines: the default route; r5 advisor
 * finding): the causal sweep's running values stay in the COEFFICIENT type between the two sweeps -- rounded to
 * float32 on the float32 route -- and a line starts from a 20-term truncated sum; the one-thread-per-line kernels
 * keep them in double.  Bound: 1 ulp of a float32 coefficient (the whole-volume order-3 tests hold 2e-5 against
 * SciPy's double); strided and contiguous axes round differently.  spline_mode | 0x100 (exact) selects the
 * double-state kernels. */
int mi_spline_prefilter(const mi_array *in, const mi_array *out, int order, int spline_mode, int npad,
                        int pad_mode, double cval, mi_stream stream);
int mi_spline_map_coordinates(const mi_array *coef, const mi_array *coords, const mi_array *out,
                              int order, int mode, double cval, int npad, mi_stream stream);
int mi_spline_affine_transform(const mi_array *coef, const mi_array *out, const double *matrix,
                               int order, int mode, double cval, int npad, mi_stream stream);

/* ------------------------------------------------------------------ */
/* multi-GPU: slab partition on axis 0, halo exchange over RCCL/xGMI    */
/* (new design; the reference is single-GPU)      */
/* ------------------------------------------------------------------ */
#define MI_UNIQUE_ID_BYTES 128
int mi_comm_unique_id(char id[MI_UNIQUE_ID_BYTES]);
int mi_comm_init_rank(mi_comm *comm, int nranks, int rank, const char id[MI_UNIQUE_ID_BYTES]);
int mi_comm_destroy(mi_comm comm);
/* Slab buffer layout on every rank: [lo halo planes][n local planes][hi halo planes],
 * plane_bytes each.  Sends the first `hi` / last `lo` local planes to the
 * previous / next rank and receives into the halo regions, all inside one
 * ncclGroupStart/End.  prev/next < 0 means no neighbour (global edge). */
int mi_halo_exchange(mi_comm comm, void *slab, size_t plane_bytes, int64_t n_local, int lo,
                     int hi, int prev_rank, int next_rank, mi_stream stream);

/* n point-to-point transfers in ONE RCCL group: ptrs[i] / nbytes[i] sent to (is_send[i] != 0) or received from rank
 * peers[i].  Between one pair of ranks the sends of one side must come in the order of the receives of the other
 * (RCCL matches them by order).  Used once per call by the output-sharded interpolation to fetch the input planes a
 * rank's output planes read (the pre-image slab) -- not a collective, nothing in a timed step. */
int mi_comm_sendrecv(mi_comm comm, int n, void *const ptrs[], const size_t nbytes[], const int peers[],
                     const int is_send[], mi_stream stream);

/* One filtering step of a slab rank: fused separable filter of the extended
 * slab [lo halo | local planes | hi halo] (a side without neighbour has no
 * halo planes) with the exchange overlapped:
 *   stream      : record input_free | interior planes ....... | wait halos_ready | edge planes
 *   comm_stream : wait input_free   | RCCL send/recv of halos | record halos_ready
 * overlap = 1 selects that schedule, 0 the plain one (exchange, then one
 * launch over all local planes, both on `stream`), -1 decides by halo size:
 * the two cross-stream waits and the extra launch cost ~20 us, so overlapping
 * pays from ~8 MiB of halo per direction.  Only the local planes of ext_out
 * are meaningful (kernels longer than 9 taps also overwrite its halo planes).  With overlap = 1, MI_ERR_UNSUPPORTED is returned before anything is
 * queued when the fused kernel does not take plane ranges for the request. */
int mi_slab_separable3d_f32(mi_comm comm, const mi_array *ext_in, const mi_array *ext_out,
                            const double *const weights[3], const int wlen[3], const int origin[3],
                            const int mode[3], double cval, int lo, int hi, int prev_rank, int next_rank,
                            int overlap, mi_stream comm_stream, mi_event input_free,
                            mi_event halos_ready, mi_stream stream);

/* r4 -- the pipelined slab schedule: `nbuf` (2 or 3; 1 = serial) resident input slabs per rank, one output slab.
 * The halo exchange of the NEXT input runs on an internal high-priority comm stream underneath the single launch that
 * filters the current one (steady state: step time = max(kernel, exchange); no split launch, no wait on the critical
 * path).  All slabs are extended buffers [lo halo | local planes | hi halo] of one shape.
 *   mi_slab_pipe_step(pipe, submit, compute): "input `submit` is final -- as of the work queued on `stream` so far --
 *       exchange its halos" and / or "filter input `compute` into the output slab" (-1 = nothing); a buffer must be
 *       submitted before it is computed;
 *   mi_slab_pipe_run(pipe, nsteps, use_graph): nsteps steps of the rotation 0, 1, .., nbuf-1, 0, .. with the
 *       submits nbuf-1 steps ahead (resident inputs: benchmarks, repeated filtering); use_graph > 0 replays a captured
 *       hipGraph of one rotation (or of `use_graph` steps, rounded down to whole rotations) where possible and queues
 *       directly otherwise;
 *   mi_slab_pipe_info: graph_state 0 not tried / 1 in use / -1 capture refused; planes_ok 0 = the kernel takes no plane
 *       ranges and filters the halo planes of the output too (scratch).
 * Every refusal (MI_ERR_UNSUPPORTED, argument errors) comes from mi_slab_pipe_create, before anything is queued. */
typedef void *mi_slab_pipe;
int mi_slab_pipe_create(mi_slab_pipe *pipe, mi_comm comm, int nbuf, const mi_array *const ext_in[],
                        const mi_array *ext_out, const double *const weights[3], const int wlen[3],
                        const int origin[3], const int mode[3], double cval, int lo, int hi, int prev_rank,
                        int next_rank, mi_stream stream);
int mi_slab_pipe_destroy(mi_slab_pipe pipe);
int mi_slab_pipe_step(mi_slab_pipe pipe, int submit, int compute);
int mi_slab_pipe_run(mi_slab_pipe pipe, int nsteps, int use_graph);
int mi_slab_pipe_info(mi_slab_pipe pipe, int *graph_state, int *graph_steps, int *planes_ok);

#ifdef __cplusplus
}
#endif
#endif /* MI355IMG_H */
