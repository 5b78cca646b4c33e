// nd_common.hpp -- shared machinery of the n-D neighbourhood kernels
// (dense correlate, footprint min/max, binary erosion).
//
// The reference generates one nested tap loop per (mode, weights shape,
// offsets) and JIT-compiles it (_filters_core.py:190-348).  Here the taps that
// matter (non-zero weights / set footprint entries, the same skip the
// reference does at run time, _filters_core.py:242-246) are flattened on the
// host into a tap table that is uploaded per call; kernels are pre-compiled
// and take shapes, offsets and mode as run-time arguments.
//
// Every output voxel first tests whether its whole neighbourhood lies inside
// the array; if so taps are plain `base + linear_offset` loads, otherwise each
// tap goes through the boundary map per axis.
#pragma once
#include <vector>

#include "common.hpp"

namespace mi {

// Geometry is always padded with leading unit axes to a compile-time rank ND
// (3 for the common 1-3-D case, 8 otherwise) so the per-axis loops unroll and
// coordinates stay in registers.
struct NdGeom {
    int ndim;                      // padded rank (== ND of the kernel)
    int64_t shape[MI_MAX_NDIM];
    int64_t stride[MI_MAX_NDIM];   // elements
    int32_t wshape[MI_MAX_NDIM];
    int32_t off[MI_MAX_NDIM];      // wshape/2 + origin
};

struct TapTable {
    int ntaps = 0;
    const int64_t *lin = nullptr;   // interior linear offsets         [ntaps]
    const int32_t *idx = nullptr;   // tap coordinates                 [ntaps * ndim]
    const double *val = nullptr;    // weights / structure values      [ntaps] (may be null)
};

// Host side: build geometry + tap table.  `keep(t)` says whether tap t (C
// order over wshape) participates; `value(t)` is its payload.
struct TapBuilder {
    NdGeom g;
    std::vector<int64_t> lin;
    std::vector<int32_t> idx;
    std::vector<double> val;
    Scratch s_lin, s_idx, s_val;

    static int rank_for(int ndim) { return ndim <= 3 ? 3 : MI_MAX_NDIM; }

    int init(const mi_array *in, const int64_t *wshape, const int *origins, const char *what)
    {
        const int nd = rank_for(in->ndim), pad = nd - in->ndim;
        g.ndim = nd;
        for (int d = 0; d < pad; d++) {
            g.shape[d] = 1;
            g.stride[d] = 0;
            g.wshape[d] = 1;
            g.off[d] = 0;
        }
        int64_t st = 1;
        for (int d = in->ndim - 1; d >= 0; d--) {
            g.shape[pad + d] = in->shape[d];
            g.stride[pad + d] = st;
            st *= in->shape[d];
            if (wshape[d] < 1 || wshape[d] > 32767) {
                set_error("%s: unsupported extent %lld on axis %d", what, (long long)wshape[d], d);
                return MI_ERR_INVALID_ARG;
            }
            g.wshape[pad + d] = (int32_t)wshape[d];
            g.off[pad + d] = (int32_t)(wshape[d] / 2 + origins[d]);
            if (g.off[pad + d] < 0 || g.off[pad + d] >= wshape[d]) {
                set_error("invalid origin");
                return MI_ERR_INVALID_ARG;
            }
        }
        return MI_OK;
    }

    template <typename Keep, typename Value>
    void fill(Keep keep, Value value, bool with_values)
    {
        int64_t ntot = 1;
        for (int d = 0; d < g.ndim; d++) ntot *= g.wshape[d];
        int32_t t[MI_MAX_NDIM];
        for (int64_t k = 0; k < ntot; k++) {
            if (!keep(k)) continue;
            int64_t r = k, lo = 0;
            for (int d = g.ndim - 1; d >= 0; d--) {
                t[d] = (int32_t)(r % g.wshape[d]);
                r /= g.wshape[d];
                lo += (int64_t)(t[d] - g.off[d]) * g.stride[d];
            }
            lin.push_back(lo);
            for (int d = 0; d < g.ndim; d++) idx.push_back(t[d]);
            if (with_values) val.push_back(value(k));
        }
    }

    int upload(TapTable *tt, hipStream_t s)
    {
        tt->ntaps = (int)lin.size();
        if (tt->ntaps == 0) return MI_OK;
        int rc;
        if ((rc = s_lin.upload(lin.data(), lin.size() * sizeof(int64_t), s))) return rc;
        if ((rc = s_idx.upload(idx.data(), idx.size() * sizeof(int32_t), s))) return rc;
        tt->lin = (const int64_t *)s_lin.ptr;
        tt->idx = (const int32_t *)s_idx.ptr;
        if (!val.empty()) {
            if ((rc = s_val.upload(val.data(), val.size() * sizeof(double), s))) return rc;
            tt->val = (const double *)s_val.ptr;
        }
        return MI_OK;
    }
};

// Device side: position of one output voxel.
template <int ND>
struct Voxel {
    int64_t base;                   // linear index of the voxel itself
    int64_t c[ND];                  // coordinates minus offsets (first tap position)
    bool interior;
};

template <int ND>
__device__ __forceinline__ Voxel<ND> locate(const NdGeom &g, int64_t i)
{
    Voxel<ND> v;
    v.base = i;
    v.interior = true;
    int64_t r = i;
#pragma unroll
    for (int d = ND - 1; d >= 0; d--) {
        const int64_t q = r / g.shape[d];
        const int64_t k = r - q * g.shape[d];
        r = q;
        v.c[d] = k - g.off[d];
        v.interior = v.interior && v.c[d] >= 0 && v.c[d] + g.wshape[d] <= g.shape[d];
    }
    return v;
}

// Linear index of tap t for a boundary voxel, or -1 when the constant applies.
template <int ND>
__device__ __forceinline__ int64_t tap_pos(const NdGeom &g, const Voxel<ND> &v,
                                           const int32_t *__restrict__ idx, int t, int mode)
{
    int64_t pos = 0;
#pragma unroll
    for (int d = 0; d < ND; d++) {
        const int64_t j = bmap<int64_t>(v.c[d] + idx[t * ND + d], g.shape[d], mode);
        if (j < 0) return -1;
        pos += j * g.stride[d];
    }
    return pos;
}

}  // namespace mi

// ---------------------------------------------------------------------------
// Rank <= 3 fast geometry: 3-D launch grid (lanes along x, no index division),
// 32-bit offsets through an SRSRC buffer descriptor, tap table read with
// wave-uniform (scalar) loads.  Used by the n-D correlate / min-max / binary
// kernels whenever the (padded) rank is 3 and the array is < 4 GiB.
// ---------------------------------------------------------------------------
namespace mi {

struct Geom3 {
    int nz, ny, nx;
    int wz, wy, wx;
    int oz, oy, ox;
};

constexpr int kInlineTaps3 = 128;

// Tap table of the fast kernels.  Up to kInlineTaps3 taps travel inside the
// kernel arguments (no allocation, no host-to-device copy per call: the
// uploads were the dominant cost of small calls); larger tables use one
// device buffer.
struct Taps3 {
    int ntaps;
    int inl;                             // 1: the inline arrays below are valid
    const int *lin;                      // element offset of the tap relative to the voxel (interior)  [ntaps]
    const int *zyx;                      // tap coordinates packed z << 20 | y << 10 | x               [ntaps]
    const double *val;                   // payload (weight / structure value), may be null             [ntaps]
    int has_val;
    int inl_lin[kInlineTaps3];
    int inl_zyx[kInlineTaps3];
    double inl_val[kInlineTaps3];
};

constexpr int kMaxTaps3 = 2048;          // taps staged in LDS by the fast kernels (32 KiB)

// LDS copy of the tap table, filled cooperatively at kernel start; reads with a
// wave-uniform index are single broadcast ds_reads
struct LdsTaps {
    double *val;
    int *lin;
    int *zyx;
};

__device__ __forceinline__ LdsTaps stage_taps(const Taps3 &tt, char *smem)
{
    LdsTaps l;
    l.val = reinterpret_cast<double *>(smem);
    l.lin = reinterpret_cast<int *>(smem + (size_t)tt.ntaps * 8);
    l.zyx = l.lin + tt.ntaps;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x, nth = blockDim.x * blockDim.y;
    for (int t = tid; t < tt.ntaps; t += nth) {
        l.lin[t] = tt.inl ? tt.inl_lin[t] : tt.lin[t];
        l.zyx[t] = tt.inl ? tt.inl_zyx[t] : tt.zyx[t];
        l.val[t] = tt.has_val ? (tt.inl ? tt.inl_val[t] : tt.val[t]) : 0.0;
    }
    __syncthreads();
    return l;
}

// Host side: geometry + taps for the rank <= 3 fast kernels, built directly
// from the window description (no intermediate wide table).
struct Taps3Builder {
    Geom3 g;
    std::vector<int> lin, zyx;
    std::vector<double> val;
    Scratch s_all;

    static bool eligible(const mi_array *in, const int64_t *wshape)
    {
        if (in->ndim < 1 || in->ndim > 3) return false;
        const int pad = 3 - in->ndim;
        int64_t shape[3] = {1, 1, 1};
        for (int d = 0; d < in->ndim; d++) shape[pad + d] = in->shape[d];
        const int64_t total = shape[0] * shape[1] * shape[2];
        if (total * (int64_t)dtype_size(in->dtype) >= ((int64_t)1 << 32) || total >= ((int64_t)1 << 31)) return false;
        if (shape[0] > 65535 || (shape[1] + 3) / 4 > 65535) return false;
        int64_t nw = 1;
        for (int d = 0; d < in->ndim; d++) {
            if (wshape[d] < 1 || wshape[d] >= 1024) return false;
            nw *= wshape[d];
        }
        return nw <= kMaxTaps3;
    }

    // keep(k) / value(k): k = C-order index into the window
    template <typename Keep, typename Value>
    int build(const mi_array *in, const int64_t *wshape, const int *origins, Keep keep, Value value, bool with_values)
    {
        const int pad = 3 - in->ndim;
        int shape[3] = {1, 1, 1}, w[3] = {1, 1, 1}, off[3] = {0, 0, 0};
        for (int d = 0; d < in->ndim; d++) {
            shape[pad + d] = (int)in->shape[d];
            w[pad + d] = (int)wshape[d];
            off[pad + d] = (int)(wshape[d] / 2 + origins[d]);
            if (off[pad + d] < 0 || off[pad + d] >= wshape[d]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
        }
        g.nz = shape[0]; g.ny = shape[1]; g.nx = shape[2];
        g.wz = w[0]; g.wy = w[1]; g.wx = w[2];
        g.oz = off[0]; g.oy = off[1]; g.ox = off[2];
        int64_t k = 0;
        for (int z = 0; z < w[0]; z++)
            for (int y = 0; y < w[1]; y++)
                for (int x = 0; x < w[2]; x++, k++) {
                    if (!keep(k)) continue;
                    lin.push_back(((z - off[0]) * g.ny + (y - off[1])) * g.nx + (x - off[2]));
                    zyx.push_back(z << 20 | y << 10 | x);
                    if (with_values) val.push_back(value(k));
                }
        return MI_OK;
    }

    int finish(Taps3 *tt, hipStream_t s)
    {
        memset(tt, 0, sizeof(*tt));
        const size_t nt = lin.size();
        tt->ntaps = (int)nt;
        tt->has_val = !val.empty();
        if (nt <= (size_t)kInlineTaps3) {
            tt->inl = 1;
            memcpy(tt->inl_lin, lin.data(), nt * sizeof(int));
            memcpy(tt->inl_zyx, zyx.data(), nt * sizeof(int));
            if (tt->has_val) memcpy(tt->inl_val, val.data(), nt * sizeof(double));
            return MI_OK;
        }
        // one buffer: [val (8 B each) | lin | zyx]
        std::vector<char> blob(nt * 16);
        if (tt->has_val) memcpy(blob.data(), val.data(), nt * 8);
        memcpy(blob.data() + nt * 8, lin.data(), nt * 4);
        memcpy(blob.data() + nt * 12, zyx.data(), nt * 4);
        int rc = s_all.upload(blob.data(), blob.size(), s);
        if (rc) return rc;
        // the upload is asynchronous on `s`, but `blob` dies here: wait for the staging copy
        MI_HIP(hipStreamSynchronize(s));
        tt->val = (const double *)s_all.ptr;
        tt->lin = (const int *)((const char *)s_all.ptr + nt * 8);
        tt->zyx = (const int *)((const char *)s_all.ptr + nt * 12);
        return MI_OK;
    }
};

static inline size_t taps3_lds_bytes(const Taps3 &tt) { return (size_t)tt.ntaps * 16; }
static inline dim3 grid3(const Geom3 &g) { return dim3((unsigned)((g.nx + 63) / 64), (unsigned)((g.ny + 3) / 4), (unsigned)g.nz); }

template <typename T>
__device__ __forceinline__ T buf_load(const __amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
    if constexpr (sizeof(T) == 1) {
        const unsigned char u = __builtin_amdgcn_raw_buffer_load_b8(r, byte_off, 0, 0);
        return __builtin_bit_cast(T, u);
    } else if constexpr (sizeof(T) == 2) {
        const unsigned short u = __builtin_amdgcn_raw_buffer_load_b16(r, byte_off, 0, 0);
        return __builtin_bit_cast(T, u);
    } else if constexpr (sizeof(T) == 4) {
        const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0);
        return __builtin_bit_cast(T, u);
    } else {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const u2 u = __builtin_amdgcn_raw_buffer_load_b64(r, byte_off, 0, 0);
        return __builtin_bit_cast(T, u);
    }
}

// position of a thread's voxel in the 3-D grid launch (block = 64 x 4)
struct Vox3 {
    int z, y, x;
    int lin;          // element index
    bool valid, interior;
};

__device__ __forceinline__ Vox3 locate3(const Geom3 &g)
{
    Vox3 v;
    v.x = blockIdx.x * 64 + threadIdx.x;
    v.y = blockIdx.y * 4 + threadIdx.y;
    v.z = blockIdx.z;
    v.valid = v.x < g.nx && v.y < g.ny;
    v.lin = (v.z * g.ny + v.y) * g.nx + v.x;
    const int cz = v.z - g.oz, cy = v.y - g.oy, cx = v.x - g.ox;
    v.interior = cz >= 0 && cz + g.wz <= g.nz && cy >= 0 && cy + g.wy <= g.ny && cx >= 0 && cx + g.wx <= g.nx;
    return v;
}

// element index of tap t for a boundary voxel, or -1 when the constant applies
__device__ __forceinline__ int tap_pos3(const Geom3 &g, const Vox3 &v, const LdsTaps &tt, int t, int mode)
{
    const int c = tt.zyx[t];
    const int jz = bmap_near<int>(v.z - g.oz + (c >> 20), g.nz, mode);
    const int jy = bmap_near<int>(v.y - g.oy + ((c >> 10) & 1023), g.ny, mode);
    const int jx = bmap_near<int>(v.x - g.ox + (c & 1023), g.nx, mode);
    if ((jz | jy | jx) < 0) return -1;
    return (jz * g.ny + jy) * g.nx + jx;
}

}  // namespace mi
