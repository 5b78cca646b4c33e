// common.hpp -- shared device/host helpers for libmi355img (gfx950 only).
//
// Arithmetic spec followed (reference, read-only):
//   boundary maps      cupyimg/scipy/ndimage/_util.py:170-228
//   offset rule        cupyimg/scipy/ndimage/_filters_core.py:10-11
//   cast<> semantics   cupyimg/scipy/ndimage/_filters_core.py:166-187
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <atomic>
#include <type_traits>

#include "../../include/mi355img.h"

namespace mi {

// ------------------------------------------------------------------ errors
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);

#define MI_HIP(call)                                   \
    do {                                               \
        hipError_t e__ = (call);                       \
        if (e__ != hipSuccess) return mi::hip_fail(e__, #call); \
    } while (0)

#define MI_REQUIRE(cond, code, msg)      \
    do {                                 \
        if (!(cond)) {                   \
            mi::set_error("%s", msg);    \
            return (code);               \
        }                                \
    } while (0)

// ------------------------------------------------------------------ hand-counted waits
// Every `s_waitcnt vmcnt(n)` with n > 0 in this tree is written MI_VMCNT(n): the kernel leaves n younger vector-memory
// operations (LDS-DMA of the plane after next, the stores of the previous step) in flight across the wait.  A count that
// is too generous by one is a timing race no single launch on an idle GPU shows (round 4: affine3d_zstream_kernel).  The
// STRICT build (`python -m cupyimg_amd._build --strict` -> libmi355img_strict.so, -DMI_STRICT_WAITS on the files that
// count) turns every one of them into vmcnt(0); tests/test_gpu_burst.py runs both libraries under back-to-back load and
// requires bit-identical outputs -- one test for the whole class of kernels.
#ifdef MI_STRICT_WAITS
#define MI_VMCNT(n) "s_waitcnt vmcnt(0)"
#else
#define MI_VMCNT(n) "s_waitcnt vmcnt(" #n ")"
#endif

// ------------------------------------------------------------------ test / tuning knobs
// Process-wide switches behind the mi_debug_set_* entry points (include/mi355img_debug.h): relaxed atomics, so a
// thread flipping one while other threads dispatch is a benign race (each call reads a knob once and sees either
// value), never a data race.  They exist for tests and tuning sweeps; production code never writes them.
// Every write bumps g_knob_generation (mi_debug_generation()): host-side caches of the library's refusals (the Python
// layer's _EXT_REFUSED) carry it in their keys, so a refusal recorded while a test had a knob flipped does not outlive
// the flip (r5 advisor finding).
extern std::atomic<int> g_knob_generation;
struct Knob {
    std::atomic<int> v;
    constexpr explicit Knob(int x) : v(x) {}
    operator int() const { return v.load(std::memory_order_relaxed); }
    Knob &operator=(int x)
    {
        v.store(x, std::memory_order_relaxed);
        g_knob_generation.fetch_add(1, std::memory_order_relaxed);
        return *this;
    }
};

// ------------------------------------------------------------------ runtime hooks
hipStream_t resolve_stream(mi_stream s);   // NULL -> per-device default stream
int device_cus();                          // compute units of the CURRENT device (cached per device, thread safe)
int current_device_slot();                 // hipGetDevice() folded into 0 .. 63 (0 when it cannot be asked)

// "done once" flag of something that is PER DEVICE -- hipFuncSetAttribute(MaxDynamicSharedMemorySize) above all: a process
// that drives a second device must set it there too (r4 advisor finding: one process-wide `static bool` left the second
// device launching 96-159 KiB of dynamic LDS without the attribute), and two host threads may ask at once (relaxed
// atomics: every writer stores the same value; the attribute call itself is idempotent).
struct PerDeviceOnce {
    std::atomic<bool> done[64] = {};
    bool get() const { return done[current_device_slot()].load(std::memory_order_relaxed); }
    void set() { done[current_device_slot()].store(true, std::memory_order_relaxed); }
    explicit operator bool() const { return get(); }
    PerDeviceOnce &operator=(bool v) { done[current_device_slot()].store(v, std::memory_order_relaxed); return *this; }
};
// block for work on `stream` (NULL = the default stream); pool_free returns it to that stream's arena
int pool_alloc(void **p, size_t n, hipStream_t stream);
int pool_free(void *p);

// Small host -> device parameter upload (weights, offset tables) that is
// ordered on `stream` and released back to the pool when the object dies.
struct Scratch {
    void *ptr = nullptr;
    ~Scratch() { if (ptr) pool_free(ptr); }
    int upload(const void *host, size_t nbytes, hipStream_t stream);
};

// ------------------------------------------------------------------ dtype helpers
static inline size_t dtype_size(int dt)
{
    switch (dt) {
    case MI_BOOL: case MI_I8: case MI_U8: return 1;
    case MI_I16: case MI_U16: case MI_F16: return 2;
    case MI_I32: case MI_U32: case MI_F32: return 4;
    default: return 8;
    }
}

static inline int64_t numel(const mi_array *a)
{
    int64_t n = 1;
    for (int d = 0; d < a->ndim; d++) n *= a->shape[d];
    return n;
}

static inline bool is_contiguous(const mi_array *a)
{
    int64_t expect = (int64_t)dtype_size(a->dtype);
    for (int d = a->ndim - 1; d >= 0; d--) {
        if (a->shape[d] == 0) return true;
        if (a->shape[d] != 1 && a->strides[d] != expect) return false;
        expect *= a->shape[d];
    }
    return true;
}

static inline bool same_shape(const mi_array *a, const mi_array *b)
{
    if (a->ndim != b->ndim) return false;
    for (int d = 0; d < a->ndim; d++)
        if (a->shape[d] != b->shape[d]) return false;
    return true;
}

static inline int check_array(const mi_array *a, const char *name)
{
    if (!a || a->ndim < 0 || a->ndim > MI_MAX_NDIM || a->dtype < MI_BOOL || a->dtype > MI_F16) {
        set_error("invalid array descriptor for %s", name);
        return MI_ERR_INVALID_ARG;
    }
    return MI_OK;
}

// filters treat 'wrap' as 'grid-wrap' and 'grid-constant' as 'constant'
static inline int filter_mode(int mode)
{
    if (mode == MI_MODE_WRAP) return MI_MODE_GRID_WRAP;
    if (mode == MI_MODE_GRID_CONSTANT) return MI_MODE_CONSTANT;
    return mode;
}

// Call F.template operator()<T>() for the C type behind a dtype code.
template <typename F>
static inline int dispatch_dtype(int dt, F &&f)
{
    switch (dt) {
    case MI_BOOL: return f.template operator()<bool>();
    case MI_I8:   return f.template operator()<int8_t>();
    case MI_U8:   return f.template operator()<uint8_t>();
    case MI_I16:  return f.template operator()<int16_t>();
    case MI_U16:  return f.template operator()<uint16_t>();
    case MI_I32:  return f.template operator()<int32_t>();
    case MI_U32:  return f.template operator()<uint32_t>();
    case MI_I64:  return f.template operator()<int64_t>();
    case MI_U64:  return f.template operator()<uint64_t>();
    case MI_F32:  return f.template operator()<float>();
    case MI_F64:  return f.template operator()<double>();
    }
    if (dt == MI_F16) { set_error("float16 arrays are storage only: convert with mi_copy (to float32) around this call"); return MI_ERR_INVALID_ARG; }
    set_error("unsupported dtype code %d", dt);
    return MI_ERR_INVALID_ARG;
}

// ------------------------------------------------------------------ device side

// Boundary index map for filters; -1 = use cval.  C truncated '%' as in the spec.
template <typename I>
__device__ __forceinline__ I bmap(I i, I n, int mode)
{
    switch (mode) {
    case MI_MODE_REFLECT:
        if (i < 0) i = -1 - i;
        i %= 2 * n;
        return min(i, 2 * n - 1 - i);
    case MI_MODE_MIRROR:
        if (n == 1) return 0;
        if (i < 0) i = -i;
        i = 1 + (i - 1) % (2 * n - 2);
        return min(i, 2 * n - 2 - i);
    case MI_MODE_NEAREST:
        return min(max(i, (I)0), n - 1);
    case MI_MODE_GRID_WRAP:
        i %= n;
        return i < 0 ? i + n : i;
    case MI_MODE_WRAP:
        if (n == 1) return 0;
        if (i < 0) i += (n - 1) * (-i / (n - 1) + 1);
        else if (i > n - 1) i -= (n - 1) * (i / (n - 1));
        return i;
    default:
        return (i < 0 || i >= n) ? (I)-1 : i;
    }
}

// Same map, cheap when the index is at most one array length outside (the case
// for every window that is not longer than the array): no integer division.
template <typename I>
__device__ __forceinline__ I bmap_near(I i, I n, int mode)
{
    if (i >= 0 && i < n) return i;
    if (i < -n || i >= 2 * n || n == 1) return bmap<I>(i, n, mode);
    switch (mode) {
    case MI_MODE_REFLECT:   return i < 0 ? -1 - i : 2 * n - 1 - i;
    case MI_MODE_MIRROR:    return (i <= -n || i >= 2 * n - 1) ? bmap<I>(i, n, mode) : (i < 0 ? -i : 2 * n - 2 - i);
    case MI_MODE_NEAREST:   return i < 0 ? (I)0 : n - 1;
    case MI_MODE_GRID_WRAP: return i < 0 ? i + n : i - n;
    case MI_MODE_WRAP:      return bmap<I>(i, n, mode);
    default:                return (I)-1;
    }
}

// Iteration space of the one-axis kernels: the array viewed as (outer, n, inner)
// around the filtered axis.  f(i, l, base): i = flat index of the output, l =
// its position on the axis, base = flat index of (outer, 0, inner-offset).
//   geom 0: flat grid-stride loop (two integer divisions per output);
//   geom 1: inner > 1 -- threads along `inner`, blockIdx.y walks chunks of the
//           axis, blockIdx.z the outer index: no division, and a whole block
//           shares l, so the interior / boundary decision is uniform;
//   geom 2: inner == 1 (last axis) -- threads along the axis itself, a block
//           handles a chunk of rows.
// A thread produces kLineChunk outputs: one block per output element per thread
// is bound by the workgroup dispatch rate, not by memory.
constexpr int kLineChunk = 16;

template <typename I, typename F>
__device__ __forceinline__ void for_each_line_output(int geom, I n, I inner, I total, F f)
{
    if (geom == 0) {
        for (I i = (I)blockIdx.x * (I)blockDim.x + (I)threadIdx.x; i < total; i += (I)gridDim.x * (I)blockDim.x) {
            const I l = (i / inner) % n;
            f(i, l, i - l * inner);
        }
    } else if (geom == 1) {
        const I k = (I)blockIdx.x * (I)blockDim.x + (I)threadIdx.x;
        if (k >= inner) return;
        const I outer = total / (n * inner);
        const I l0 = (I)blockIdx.y * (I)kLineChunk;
        const I l1 = l0 + (I)kLineChunk < n ? l0 + (I)kLineChunk : n;
        for (I o = blockIdx.z; o < outer; o += gridDim.z) {
            const I base = o * n * inner + k;
            for (I l = l0; l < l1; l++) f(base + l * inner, l, base);
        }
    } else {
        const I l = (I)blockIdx.x * (I)blockDim.x + (I)threadIdx.x;
        if (l >= n) return;
        const I outer = total / n;
        const I o0 = ((I)blockIdx.y + (I)gridDim.y * (I)blockIdx.z) * (I)kLineChunk;
        const I o1 = o0 + (I)kLineChunk < outer ? o0 + (I)kLineChunk : outer;
        for (I o = o0; o < o1; o++) {
            const I base = o * n;
            f(base + l, l, base);
        }
    }
}

// double -> T with the C-cast semantics SciPy/x86 shows: truncate toward zero
// through a wide signed integer, low bits kept (so negative -> unsigned wraps).
template <typename T>
__device__ __forceinline__ T cast_from_f64(double a)
{
    if constexpr (std::is_same<T, uint64_t>::value) {
        return a >= 0 ? (T)a : (T)(-(int64_t)(uint64_t)(-a));
    } else if constexpr (std::is_floating_point<T>::value) {
        return (T)a;
    } else {
        return (T)(int64_t)a;
    }
}
template <>
__device__ __forceinline__ bool cast_from_f64<bool>(double a) { return a != 0.0; }

__device__ __forceinline__ void store_as(void *p, int64_t i, int dt, double v)
{
    switch (dt) {
    case MI_BOOL: ((uint8_t *)p)[i] = (uint8_t)(v != 0.0); break;
    case MI_I8:   ((int8_t *)p)[i] = cast_from_f64<int8_t>(v); break;
    case MI_U8:   ((uint8_t *)p)[i] = cast_from_f64<uint8_t>(v); break;
    case MI_I16:  ((int16_t *)p)[i] = cast_from_f64<int16_t>(v); break;
    case MI_U16:  ((uint16_t *)p)[i] = cast_from_f64<uint16_t>(v); break;
    case MI_I32:  ((int32_t *)p)[i] = cast_from_f64<int32_t>(v); break;
    case MI_U32:  ((uint32_t *)p)[i] = cast_from_f64<uint32_t>(v); break;
    case MI_I64:  ((int64_t *)p)[i] = cast_from_f64<int64_t>(v); break;
    case MI_U64:  ((uint64_t *)p)[i] = cast_from_f64<uint64_t>(v); break;
    case MI_F32:  ((float *)p)[i] = (float)v; break;
    default:      ((double *)p)[i] = v; break;
    }
}

// SciPy's rounding of interpolation results into integer outputs: half away
// from zero, clipped to the output range (the reference uses rint(),
// _interp_kernels.py:580-583; they differ only at exact .5 values).
__device__ __forceinline__ double interp_round(double t, int dt)
{
    double lo, hi;
    switch (dt) {
    case MI_I8:  lo = -128.0; hi = 127.0; break;
    case MI_U8:  lo = 0.0; hi = 255.0; break;
    case MI_I16: lo = -32768.0; hi = 32767.0; break;
    case MI_U16: lo = 0.0; hi = 65535.0; break;
    case MI_I32: lo = -2147483648.0; hi = 2147483647.0; break;
    case MI_U32: lo = 0.0; hi = 4294967295.0; break;
    case MI_I64: lo = -9223372036854775808.0; hi = 9223372036854775807.0; break;
    case MI_U64: lo = 0.0; hi = 18446744073709551615.0; break;
    default: return t;
    }
    if (lo == 0.0) t = t > 0 ? t + 0.5 : 0.0;
    else t = t > 0 ? t + 0.5 : t - 0.5;
    return fmin(fmax(t, lo), hi);
}

__device__ __forceinline__ double load_as_f64(const void *p, int64_t i, int dt)
{
    switch (dt) {
    case MI_BOOL: return (double)(((const uint8_t *)p)[i] != 0);
    case MI_I8:   return (double)((const int8_t *)p)[i];
    case MI_U8:   return (double)((const uint8_t *)p)[i];
    case MI_I16:  return (double)((const int16_t *)p)[i];
    case MI_U16:  return (double)((const uint16_t *)p)[i];
    case MI_I32:  return (double)((const int32_t *)p)[i];
    case MI_U32:  return (double)((const uint32_t *)p)[i];
    case MI_I64:  return (double)((const int64_t *)p)[i];
    case MI_U64:  return (double)((const uint64_t *)p)[i];
    case MI_F32:  return (double)((const float *)p)[i];
    default:      return ((const double *)p)[i];
    }
}

// launch geometry for for_each_line_output(): returns geom and fills grid / block
static inline int line_grid(int64_t total, int64_t n, int64_t inner, dim3 *grid, dim3 *block)
{
    const int64_t outer = (n * inner) ? total / (n * inner) : 0;
    if (inner >= 64) {
        const int bx = inner >= 256 ? 256 : (inner >= 128 ? 128 : 64);
        const int64_t gy = (n + kLineChunk - 1) / kLineChunk;
        if (gy > 65535) return 0;
        *block = dim3(bx);
        *grid = dim3((unsigned)((inner + bx - 1) / bx), (unsigned)gy, (unsigned)(outer < 65535 ? outer : 65535));
        return 1;
    }
    if (inner == 1 && n >= 64) {
        const int bx = n >= 256 ? 256 : (n >= 128 ? 128 : 64);
        const int64_t groups = (outer + kLineChunk - 1) / kLineChunk;
        const int64_t gy = groups < 65535 ? groups : 65535;
        const int64_t gz = (groups + gy - 1) / gy;
        if (gz > 65535) return 0;
        *block = dim3(bx);
        *grid = dim3((unsigned)((n + bx - 1) / bx), (unsigned)gy, (unsigned)gz);
        return 2;
    }
    return 0;
}

// launch geometry for 1-thread-per-element kernels: cap the grid and stride
static inline void grid_for(int64_t total, int block, dim3 *grid)
{
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 256 * 32;   // 256 CUs x 32 blocks: plenty of waves, bounded launch
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    *grid = dim3((unsigned)g);
}

}  // namespace mi
