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
