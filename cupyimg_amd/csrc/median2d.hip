// median2d.hip -- 3 x 3 median filter of float32 / uint8 images in ONE streaming launch.
//
// Replaces, for median_filter(size=3) / rank_filter(rank=4, size=3) / skimage.filters.median with a 3 x 3 square on
// images (and on volumes with a (1, 3, 3) footprint), the generic rank kernel: the reference gathers the nine
// footprint samples of every pixel into a local array and sorts them (cupyimg/scipy/ndimage/filters.py:1560-1701,
// selection networks of _filters_optimal_medians.py for small footprints), i.e. nine scattered loads and a 19-exchange
// network per pixel.  Here a wave owns a 256-float (1024-byte) row segment and streams down the image (the
// barrier-free layout of stream3d.hip): the two previous raw rows stay in registers, every row is loaded once, and
// the median of nine is taken as
//     columns sorted:  lo = min3, mid = med3, hi = max3 of the three rows        (shared by three output pixels)
//     median = med3( max3(lo[x-1], lo[x], lo[x+1]),  med3(mid[x-1 .. x+1]),  min3(hi[x-1 .. x+1]) )
// -- v_min3_f32 / v_med3_f32 / v_max3_f32 are single instructions on gfx950, so a float4 of outputs costs ~35
// VALU instructions including the lane shifts; the kernel is bound by the 8 B/pixel it moves.  The result is one of
// the nine samples: bit-exact against SciPy for data without NaNs (only the sign of a zero may differ).
// The uint8 kernel (same structure on 16 pixels per lane, even/odd split u16 pairs) lives in minmax3d_u8.hip.
#include "sep_common.hpp"
#include "stream3d.hpp"

namespace mi {

struct Med2dParams {
    int nx, ny, nz;
    int mx, my;          // boundary modes along x / y (filter_mode()-normalised)
    float cval;
    int chunk, nchunks, nxt;
    int swz;             // XCD-aware workgroup order (xcd_block())
};

__device__ __forceinline__ float fmin3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
__device__ __forceinline__ float fmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float fmed3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

struct Col4 { float4 lo, mid, hi; };

__device__ __forceinline__ Col4 sort_cols(const float4 a, const float4 b, const float4 c)
{
    Col4 r;
    r.lo = make_float4(fmin3(a.x, b.x, c.x), fmin3(a.y, b.y, c.y), fmin3(a.z, b.z, c.z), fmin3(a.w, b.w, c.w));
    r.mid = make_float4(fmed3(a.x, b.x, c.x), fmed3(a.y, b.y, c.y), fmed3(a.z, b.z, c.z), fmed3(a.w, b.w, c.w));
    r.hi = make_float4(fmax3(a.x, b.x, c.x), fmax3(a.y, b.y, c.y), fmax3(a.z, b.z, c.z), fmax3(a.w, b.w, c.w));
    return r;
}

__global__ void __launch_bounds__(256)
median3x3_f32_kernel(const float *__restrict__ in, float *__restrict__ out, const Med2dParams p)
{
    constexpr int DEPTH = 4;
    constexpr int U = 4;             // lcm(ring of 2 rows, DEPTH)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nlines = nz * p.nxt;
    const int wid = xcd_block((int)blockIdx.x, (int)gridDim.x, p.swz) * 4 + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int z = line / p.nxt, xt = line - z * p.nxt;
    const int x0 = xt * 256;
    const int nlanes = min(64, (nx - x0) >> 2);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned rowbase = (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 4 * lane)) * 4u : kOOB;

    // the 4-float block outside the tile (lane 0: left, lane `last`: right); only its nearest column is used
    const int side = lane == 0 ? 0 : 1;
    int ekind, est;
    edge_block(side, 1, x0, x0 + 4 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? (rowbase + (unsigned)est) * 4u : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, ny);
    const int nsteps = a1 - a0 + 2;
    const int ai0 = a0 - 1;

    struct Slot { float4 v; float4 e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)ny) ai = bmap<int>(ai, ny, p.my);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * (unsigned)nx * 4u;
        s.v = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0));
        s.e = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : evoff, soff, 0));
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    const float4 cv4 = make_float4(p.cval, p.cval, p.cval, p.cval);
    float4 rv[2];                   // the two previous raw rows
    float re[2];                    // ... and their edge column (left of lane 0 / right of lane `last`)
    rv[0] = rv[1] = cv4;
    re[0] = re[1] = p.cval;
    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J % DEPTH];
                const float4 v = s.cst ? cv4 : s.v;
                const float4 eb = s.cst ? cv4 : apply_kind(s.e, ekind, side, p.cval);
                const float ec = side == 0 ? eb.w : eb.x;
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                if (i >= 2) {
                    const Col4 q = sort_cols(rv[J % 2], rv[(J + 1) % 2], v);
                    const float elo = fmin3(re[0], re[1], ec), emid = fmed3(re[0], re[1], ec), ehi = fmax3(re[0], re[1], ec);
                    // neighbours of the outer components come from the next lane (edge lanes: the edge column)
                    const float lo_l = dpp_from_left(elo, q.lo.w), mid_l = dpp_from_left(emid, q.mid.w), hi_l = dpp_from_left(ehi, q.hi.w);
                    float lo_r = dpp_from_right(elo, q.lo.x), mid_r = dpp_from_right(emid, q.mid.x), hi_r = dpp_from_right(ehi, q.hi.x);
                    if (lane == last) { lo_r = elo; mid_r = emid; hi_r = ehi; }
                    float4 o;
                    o.x = fmed3(fmax3(lo_l, q.lo.x, q.lo.y), fmed3(mid_l, q.mid.x, q.mid.y), fmin3(hi_l, q.hi.x, q.hi.y));
                    o.y = fmed3(fmax3(q.lo.x, q.lo.y, q.lo.z), fmed3(q.mid.x, q.mid.y, q.mid.z), fmin3(q.hi.x, q.hi.y, q.hi.z));
                    o.z = fmed3(fmax3(q.lo.y, q.lo.z, q.lo.w), fmed3(q.mid.y, q.mid.z, q.mid.w), fmin3(q.hi.y, q.hi.z, q.hi.w));
                    o.w = fmed3(fmax3(q.lo.z, q.lo.w, lo_r), fmed3(q.mid.z, q.mid.w, mid_r), fmin3(q.hi.z, q.hi.w, hi_r));
                    u32x4 u;
                    u.x = __float_as_uint(o.x); u.y = __float_as_uint(o.y); u.z = __float_as_uint(o.z); u.w = __float_as_uint(o.w);
                    const unsigned so = (unsigned)(a0 + i - 2) * (unsigned)nx * 4u;
                    buffer_store_b128_soff(u, rout, voff, so);
                }
                rv[J % 2] = v;
                re[J % 2] = ec;
            }
        });
    }
}

int run_median3x3_u8(const uint8_t *in, uint8_t *out, int nz, int ny, int nx, int mx, int my, int cval, hipStream_t s);  // minmax3d_u8.hip
int run_median3x3_16(const mi_array *in, const mi_array *out, int mx, int my, double cval, hipStream_t s);                   // minmax_16.hip
int run_median3x3_f64(const mi_array *in, const mi_array *out, int mx, int my, double cval, hipStream_t s);                  // stream_f64.hip

// chunks along y: enough waves to fill the chip (256 CUs x 16 resident waves), chunks of at least 16 rows
static void median_chunks(int nlines, int ny, int *chunk, int *nchunks)
{
    int nch = (4096 + nlines - 1) / nlines;
    if (nch > ny / 16) nch = ny / 16;
    if (nch < 1) nch = 1;
    *chunk = (ny + nch - 1) / nch;
    *nchunks = (ny + *chunk - 1) / *chunk;
}

}  // namespace mi

using namespace mi;

/* 3 x 3 median over the last two axes of a 2-D / 3-D float32 or uint8 array (declared in include/mi355img.h). */
extern "C" int mi_median3x3(const mi_array *in, const mi_array *out, const int mode[2], double cval, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("median3x3: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->dtype == MI_F64 && out->dtype == MI_F64)
        return run_median3x3_f64(in, out, filter_mode(mode[1]), filter_mode(mode[0]), cval, resolve_stream(stream));
    if (in->dtype == out->dtype && (in->dtype == MI_U16 || in->dtype == MI_I16))
        return run_median3x3_16(in, out, filter_mode(mode[1]), filter_mode(mode[0]), cval, resolve_stream(stream));
    if ((in->ndim != 2 && in->ndim != 3) || in->dtype != out->dtype || (in->dtype != MI_F32 && in->dtype != MI_U8))
        UNSUP("needs 2-D / 3-D float32, float64, uint8, uint16 or int16 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int nd = in->ndim;
    const int64_t nz = nd == 3 ? in->shape[0] : 1, ny = in->shape[nd - 2], nx = in->shape[nd - 1];
    if (nz < 1 || ny < 1 || nx < 1) return MI_OK;
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    const int my = filter_mode(mode[0]), mx = filter_mode(mode[1]);
    hipStream_t s = resolve_stream(stream);
    if (in->dtype == MI_U8) {
        if (nx < 32 || (nx & 15) || (nx & 1023) == 16) UNSUP("x extent must be a multiple of 16, >= 32");
        if (nz * ny * nx >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
        if ((my == MI_MODE_CONSTANT || mx == MI_MODE_CONSTANT) && !(cval >= 0 && cval <= 255 && cval == (double)(int)cval))
            UNSUP("cval is not a uint8 value");
        return run_median3x3_u8((const uint8_t *)in->data, (uint8_t *)out->data, (int)nz, (int)ny, (int)nx, mx, my, (int)cval, s);
    }
    if (nx < 8 || (nx & 3)) UNSUP("x extent must be a multiple of 4, >= 8");
    { const int64_t tail = nx & 255; if (tail != 0 && tail < 8) UNSUP("x extent unsuitable for the streaming x window"); }
    if (nz * ny * nx * 4 >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
    Med2dParams p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.mx = mx; p.my = my;
    p.cval = (float)cval;
    p.nxt = (int)((nx + 255) / 256);
    median_chunks(p.nz * p.nxt, p.ny, &p.chunk, &p.nchunks);
    const int waves = p.nz * p.nxt * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz * 4);
    hipLaunchKernelGGL(median3x3_f32_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, (const float *)in->data, (float *)out->data, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
#undef UNSUP
}
