// binary.hip -- binary erosion / dilation, one iteration per launch (K4).
//
// Reference: cupyimg/scipy/ndimage/morphology.py:41-128 (kernel), launch at
// :292-322.  Output is true unless a set structure tap sees a false voxel
// (outside the array the tap sees border_value); `invert` swaps true/false and
// the border, which is how dilation is expressed (:443-461).  With a mask,
// voxels whose mask is 0 keep the input value (:60-72).
//
// The reference decides convergence of iterated morphology on the host with
// `(tmp_in == tmp_out).all()` every iteration (:313,321).  Here the kernel
// ORs a device flag when any voxel changed, so the host reads one int32.
#include "nd_common.hpp"

namespace mi {

template <typename T, int ND>
__global__ void __launch_bounds__(256)
binary_erosion_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt,
                      const uint8_t *__restrict__ mask, NdGeom g, TapTable tt, int64_t total,
                      int border_value, int invert, int32_t *changed)
{
    const bool tv = !invert, fv = invert;
    const bool bv = invert ? !border_value : (border_value != 0);
    bool any_change = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const bool cur = in[i] != T(0);
        bool res;
        if (mask && !mask[i]) {
            res = cur;
        } else {
            const Voxel<ND> v = locate<ND>(g, i);
            res = tv;
            if (v.interior) {
                for (int t = 0; t < tt.ntaps; t++)
                    if ((in[i + tt.lin[t]] != T(0)) == fv) { res = fv; break; }
            } else {
                for (int t = 0; t < tt.ntaps; t++) {
                    // binary morphology always uses the constant boundary
                    const int64_t pos = tap_pos<ND>(g, v, tt.idx, t, MI_MODE_CONSTANT);
                    const bool nn = pos < 0 ? bv : ((in[pos] != T(0)) == tv);
                    if (!nn) { res = fv; break; }
                }
            }
        }
        any_change |= (res != cur);
        store_as(out, i, out_dt, res ? 1.0 : 0.0);
    }
    if (changed && __any(any_change) && (threadIdx.x & 63) == 0) atomicOr(changed, 1);
}

// rank <= 3 fast geometry (nd_common.hpp)
template <typename T>
__global__ void __launch_bounds__(256)
binary3_kernel(const T *__restrict__ in, void *__restrict__ out, int out_dt, const uint8_t *__restrict__ mask, Geom3 g,
               Taps3 tt, int border_value, int invert, int32_t *changed)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsTaps lt = stage_taps(tt, smem);
    const Vox3 v = locate3(g);
    bool any_change = false;
    if (v.valid) {
        const __amdgpu_buffer_rsrc_t rin =
            __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)((unsigned)g.nz * g.ny * g.nx * sizeof(T)), 0x00020000);
        const bool tv = !invert, fv = invert;
        const bool bv = invert ? !border_value : (border_value != 0);
        const bool cur = in[v.lin] != T(0);
        bool res;
        if (mask && !mask[v.lin]) {
            res = cur;
        } else {
            res = tv;
            if (v.interior) {
                for (int t = 0; t < tt.ntaps; t++)
                    if ((buf_load<T>(rin, (unsigned)(v.lin + lt.lin[t]) * (unsigned)sizeof(T)) != T(0)) == fv) { res = fv; break; }
            } else {
                for (int t = 0; t < tt.ntaps; t++) {
                    const int pos = tap_pos3(g, v, lt, t, MI_MODE_CONSTANT);
                    const bool nn = pos < 0 ? bv : ((buf_load<T>(rin, (unsigned)pos * (unsigned)sizeof(T)) != T(0)) == tv);
                    if (!nn) { res = fv; break; }
                }
            }
        }
        any_change = res != cur;
        store_as(out, v.lin, out_dt, res ? 1.0 : 0.0);
    }
    if (changed && __any(any_change) && (threadIdx.x & 63) == 0) atomicOr(changed, 1);
}

}  // namespace mi

namespace mi {
int binary3_tiled(const mi_array *in, const mi_array *out, const uint8_t *structure, const int64_t *sshape,
                  const int *origins, const mi_array *mask, int border_value, int invert, int32_t *changed,
                  hipStream_t s);   // binary3d.hip
int bitmorph3(const mi_array *in, const mi_array *out, const uint8_t *structure, const int64_t *sshape, const int *origins,
              const mi_array *mask, int border_value, int invert, int k, int32_t *flags, hipStream_t s, int open_close = 0);   // bitmorph3d.hip
int bitfill3(const mi_array *in, const mi_array *out, const uint8_t *structure, const int64_t *sshape, const int *origins,
             const mi_array *mask, int border_value, int32_t *flag, hipStream_t s);                                            // bitmorph3d.hip
}

using namespace mi;

// test hook (not part of the C-ABI): 0 = never use the LDS-tiled kernel
static mi::Knob g_binary_tiled{1};
extern "C" int mi_debug_set_binary_tiled(int enabled) { g_binary_tiled = enabled; return MI_OK; }

static int check_binary_args(const mi_array *in, const mi_array *out, const uint8_t *structure, const int64_t *sshape,
                             const int *origins, const mi_array *mask)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1, MI_ERR_INVALID_ARG, "input must have at least one dimension");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(structure && sshape && origins, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS,
               "binary morphology needs C-contiguous arrays");
    MI_REQUIRE(in->data != out->data, MI_ERR_INVALID_ARG, "output and input may not overlap in memory");
    if (mask) {
        if ((rc = check_array(mask, "mask"))) return rc;
        MI_REQUIRE(same_shape(in, mask), MI_ERR_INVALID_ARG, "mask and input must have equal sizes");
        MI_REQUIRE(is_contiguous(mask) && dtype_size(mask->dtype) == 1, MI_ERR_NOT_CONTIGUOUS,
                   "mask must be a C-contiguous 1-byte array");
    }
    return MI_OK;
}

// `iterations` erosions (dilations with invert) in ONE launch: the k intermediate volumes never exist in HBM
// (bitmorph3d.hip).  The reference runs one launch and one host synchronisation per iteration (morphology.py:292-322).
extern "C" int mi_binary_erosion_fused(const mi_array *in, const mi_array *out, const uint8_t *structure,
                                       const int64_t *sshape, const int *origins, const mi_array *mask,
                                       int border_value, int invert, int iterations, int32_t *changed_dev, mi_stream stream)
{
    int rc;
    if ((rc = check_binary_args(in, out, structure, sshape, origins, mask))) return rc;
    MI_REQUIRE(iterations >= 1, MI_ERR_INVALID_ARG, "iterations must be >= 1");
    if (numel(in) == 0) return MI_OK;
    return bitmorph3(in, out, structure, sshape, origins, mask, border_value, invert, iterations, changed_dev,
                     resolve_stream(stream));
}

// One launch of a masked DILATION's block-wise fill towards its fixed point (binary_propagation / binary_fill_holes): see
// bitfill3_kernel.  The caller ping-pongs two buffers until *changed_dev stays 0.
extern "C" int mi_binary_propagation_step(const mi_array *in, const mi_array *out, const uint8_t *structure,
                                          const int64_t *sshape, const int *origins, const mi_array *mask,
                                          int border_value, int32_t *changed_dev, mi_stream stream)
{
    int rc;
    if ((rc = check_binary_args(in, out, structure, sshape, origins, mask))) return rc;
    MI_REQUIRE(mask, MI_ERR_INVALID_ARG, "mask is NULL");
    if (numel(in) == 0) return MI_OK;
    return bitfill3(in, out, structure, sshape, origins, mask, border_value, changed_dev, resolve_stream(stream));
}

// Opening (erosions, then dilations) or closing (dilations, then erosions), `iterations` of each, in ONE launch: the
// intermediate volume of morphology.py:464-613 (`tmp = binary_erosion(...)`) never exists.
extern "C" int mi_binary_open_close_fused(const mi_array *in, const mi_array *out, const uint8_t *structure,
                                          const int64_t *sshape, const mi_array *mask, int border_value, int closing,
                                          int iterations, mi_stream stream)
{
    int rc;
    const int zero[MI_MAX_NDIM] = {0};
    if ((rc = check_binary_args(in, out, structure, sshape, zero, mask))) return rc;
    MI_REQUIRE(iterations >= 1, MI_ERR_INVALID_ARG, "iterations must be >= 1");
    if (numel(in) == 0) return MI_OK;
    return bitmorph3(in, out, structure, sshape, zero, mask, border_value, 0, iterations, nullptr, resolve_stream(stream),
                     closing ? 2 : 1);
}

extern "C" int mi_binary_erosion(const mi_array *in, const mi_array *out, const uint8_t *structure,
                                 const int64_t *sshape, const int *origins, const mi_array *mask,
                                 int border_value, int invert, int32_t *changed_dev, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(in->ndim >= 1, MI_ERR_INVALID_ARG, "input must have at least one dimension");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    MI_REQUIRE(structure && sshape && origins, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(is_contiguous(in) && is_contiguous(out), MI_ERR_NOT_CONTIGUOUS,
               "binary morphology needs C-contiguous arrays");
    MI_REQUIRE(in->data != out->data, MI_ERR_INVALID_ARG, "output and input may not overlap in memory");
    if (mask) {
        if ((rc = check_array(mask, "mask"))) return rc;
        MI_REQUIRE(same_shape(in, mask), MI_ERR_INVALID_ARG, "mask and input must have equal sizes");
        MI_REQUIRE(is_contiguous(mask) && dtype_size(mask->dtype) == 1, MI_ERR_NOT_CONTIGUOUS,
                   "mask must be a C-contiguous 1-byte array");
    }
    const int64_t total = numel(in);
    if (total == 0) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    if (g_binary_tiled) {
        rc = bitmorph3(in, out, structure, sshape, origins, mask, border_value, invert, 1, changed_dev, s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
        rc = binary3_tiled(in, out, structure, sshape, origins, mask, border_value, invert, changed_dev, s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }

    Taps3Builder t3;
    Taps3 tt3;
    const bool fast3 = Taps3Builder::eligible(in, sshape);
    TapBuilder tb;
    TapTable tt;
    if (fast3) {
        if ((rc = t3.build(in, sshape, origins, [&](int64_t k) { return structure[k] != 0; }, [](int64_t) { return 0.0; },
                           false))) return rc;
        if ((rc = t3.finish(&tt3, s))) return rc;
    } else {
        if ((rc = tb.init(in, sshape, origins, "structure"))) return rc;
        tb.fill([&](int64_t k) { return structure[k] != 0; }, [](int64_t) { return 0.0; }, false);
        if ((rc = tb.upload(&tt, s))) return rc;
    }

    dim3 grid;
    grid_for(total, 256, &grid);
    const uint8_t *mp = mask ? (const uint8_t *)mask->data : nullptr;
    return dispatch_dtype(in->dtype, [&]<typename T>() -> int {
        const T *ip = (const T *)in->data;
        if (fast3) {
            hipLaunchKernelGGL((binary3_kernel<T>), grid3(t3.g), dim3(64, 4, 1), taps3_lds_bytes(tt3), s, ip, out->data, out->dtype, mp, t3.g,
                               tt3, border_value, invert, changed_dev);
            MI_HIP(hipGetLastError());
            return MI_OK;
        }
        if (tb.g.ndim == 3)
            hipLaunchKernelGGL((binary_erosion_kernel<T, 3>), grid, dim3(256), 0, s, ip, out->data, out->dtype,
                               mp, tb.g, tt, total, border_value, invert, changed_dev);
        else
            hipLaunchKernelGGL((binary_erosion_kernel<T, MI_MAX_NDIM>), grid, dim3(256), 0, s, ip, out->data,
                               out->dtype, mp, tb.g, tt, total, border_value, invert, changed_dev);
        MI_HIP(hipGetLastError());
        return MI_OK;
    });
}
