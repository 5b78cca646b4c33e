// stencil3d.hip -- LDS-tiled dense 3-D correlate for float32 volumes.
//
// Reference path replaced: correlate / convolve with a small dense kernel,
// cupyimg/scipy/ndimage/filters.py:65-210 -> :441-495 (generated nested tap
// loop _filters_core.py:298-324: one global load per tap per voxel).
//
// Design (2.5-D blocking, no separability assumed):
//   * a workgroup (8 waves) owns a 256 x TY column of the volume and streams
//     along z over a chunk of planes;
//   * the raw input planes the window spans live in an LDS ring of wz + 1
//     slots, each (TY + wy - 1) rows of 4 + 256 + 4 floats (x halo included,
//     already boundary-mapped: reflect / mirror / nearest / wrap / constant
//     are resolved when a plane is staged, never in the tap loop);
//   * the next plane is fetched into registers before the current plane is
//     computed and written to the free slot afterwards: global latency hides
//     behind the tap loop, one barrier per plane;
//   * every lane produces 4 x-consecutive outputs of RW rows; an input row is
//     read once from LDS (three aligned 16-byte reads: left block, own block,
//     right block), converted once, and feeds all (row, tap-row) pairs it
//     belongs to.  Taps are accumulated in C order of the window (z, y, x), in
//     double by default, zero weights skipped -- the same arithmetic as the
//     generic kernel in correlate_nd.hip (and SciPy's NI_Correlate), so the
//     two paths agree bit for bit.
// HBM traffic: 8 B/voxel plus the tile halos; the tap loop is LDS/VALU work.
#include "nd_common.hpp"
#include "sep_common.hpp"

namespace mi {

constexpr int kStNW = 8;            // waves per workgroup
constexpr int kStPitch = 264;       // floats per LDS row: 4 halo + 256 + 4 halo
constexpr int kStMaxWeights = 384;  // doubles carried in the kernel arguments
constexpr int kStMaxRows = 49;      // (tz, ty) pairs

struct Stencil3Params {
    int nx, ny, nz;
    int wz, wy;                 // window extent along z, y (x extent = WX of the kernel, zero-padded)
    int oz, oy;                 // w/2 + origin along z, y
    int mode;
    float cval;
    int zc, nzc, nxt, nyt;      // planes per chunk, tile counts
    unsigned mask[kStMaxRows];  // per (tz, ty): bit tx set = tap (tz, ty, tx) participates (non-zero weight)
    double w[kStMaxWeights];    // [wz][wy][WX]
};

typedef const __attribute__((address_space(4))) double *kdoubles;

// what a tap does: weighted sum (correlate) or running minimum / maximum over
// the set footprint elements (flat grey erosion / dilation, min / max filters).
// Min / max compare exactly like the generic kernel (`x < best` / `x > best`,
// first tap taken as is), so NaNs propagate the same way.
enum { ST_CORR = 0, ST_MIN = 1, ST_MAX = 2 };

// Element types: float32, or 8 / 16-bit integers (converted to float when a
// plane is staged -- exact -- and back on store with the C-cast semantics of
// the generic kernels).  A lane always owns 4 elements: 16 / 8 / 4 bytes.
template <typename T> struct Raw4 { unsigned d[sizeof(T) == 4 ? 4 : (sizeof(T) == 2 ? 2 : 1)]; };

template <typename T>
__device__ __forceinline__ Raw4<T> load4(const __amdgpu_buffer_rsrc_t r, unsigned voff)
{
    Raw4<T> q;
    if constexpr (sizeof(T) == 4) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
        q.d[0] = v.x; q.d[1] = v.y; q.d[2] = v.z; q.d[3] = v.w;
    } else if constexpr (sizeof(T) == 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
        q.d[0] = v.x; q.d[1] = v.y;
    } else {
        q.d[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
    }
    return q;
}

template <typename T>
__device__ __forceinline__ void to_floats(const Raw4<T> &q, float (&f)[4])
{
    if constexpr (std::is_same<T, float>::value) {
#pragma unroll
        for (int c = 0; c < 4; c++) f[c] = __uint_as_float(q.d[c]);
    } else if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int c = 0; c < 4; c++) f[c] = (float)(T)(q.d[c >> 1] >> (16 * (c & 1)));
    } else {
#pragma unroll
        for (int c = 0; c < 4; c++) f[c] = (float)(T)(q.d[0] >> (8 * c));
    }
}

template <typename T, typename Acc>
__device__ __forceinline__ void store4(const __amdgpu_buffer_rsrc_t r, unsigned voff, const Acc (&a)[4])
{
    if constexpr (std::is_same<T, float>::value) {
        u32x4 u;
        u.x = __float_as_uint((float)a[0]); u.y = __float_as_uint((float)a[1]);
        u.z = __float_as_uint((float)a[2]); u.w = __float_as_uint((float)a[3]);
        __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0, 0);
    } else if constexpr (sizeof(T) == 2) {
        unsigned short h[4];
#pragma unroll
        for (int c = 0; c < 4; c++) h[c] = (unsigned short)cast_from_f64<T>((double)a[c]);
        __builtin_amdgcn_raw_buffer_store_b64((u32x2){(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)},
                                              r, voff, 0, 0);
    } else {
        unsigned w = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) w |= (unsigned)(unsigned char)cast_from_f64<T>((double)a[c]) << (8 * c);
        __builtin_amdgcn_raw_buffer_store_b32(w, r, voff, 0, 0);
    }
}

template <int WX, int TY, typename Acc, bool DENSE, int OP = ST_CORR, typename T = float>
__global__ void __launch_bounds__(kStNW * 64)
stencil3_kernel(const T *__restrict__ in, T *__restrict__ out, const Stencil3Params p)
{
    constexpr int RX = WX / 2;
    constexpr int RW = TY / kStNW;                       // output rows per wave
    constexpr int RPW = (TY + 6 + kStNW - 1) / kStNW;    // staged rows per wave (wy <= 7)
    static_assert(WX >= 1 && WX <= 9 && (WX & 1), "odd x extent up to 9");
    static_assert(TY % kStNW == 0, "rows per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *ring = reinterpret_cast<float *>(smem);       // [wz + 1][rows_l][kStPitch]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    int b = blockIdx.x;
    const int total = p.nxt * p.nyt * p.nzc;
    if ((total & 7) == 0) b = (b & 7) * (total >> 3) + (b >> 3);     // one contiguous tile range per XCD
    const int per_chunk = p.nxt * p.nyt;
    const int zci = b / per_chunk;
    const int rem = b - zci * per_chunk;
    const int yt = rem / p.nxt, xt = rem - yt * p.nxt;

    const int nx = p.nx, ny = p.ny, nz = p.nz, wz = p.wz, wy = p.wy, mode = p.mode;
    const int x0 = xt * 256, y0 = yt * TY;
    const int zs = zci * p.zc, ze = min(zs + p.zc, nz);
    const int nout = ze - zs;
    const int ty_act = min(TY, ny - y0);
    const int nlanes = min(64, (nx - x0) >> 2);
    const int rows_l = TY + wy - 1;
    const int slot_floats = rows_l * kStPitch;
    const int nslots = wz + 1;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * (unsigned)sizeof(T);
    const size_t plane_elems = (size_t)ny * (size_t)nx;

    // ---------------------------------------------------------------- staging recipe (loop invariant)
    // wave w stages rows w, w + 8, w + 16 of the tile: one 16-byte load per lane
    // for the 256 floats of the row, plus one 4-byte load in lanes 0..7 for the
    // eight halo floats (each at its boundary-mapped column).
    unsigned voff_main[RPW], voff_halo[RPW];
    bool row_const[RPW];
    int lds_row[RPW];
    const bool halo_lane = lane < 8;
    const int xh = lane < 4 ? x0 - 4 + lane : x0 + 4 * nlanes + (lane - 4);
    const int xsrc = halo_lane ? bmap_near<int>(xh, nx, mode) : -1;
    const int halo_pos = lane < 4 ? lane : 4 + 4 * nlanes + (lane - 4);
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int j = wave + kStNW * k;
        const int ysrc = j < rows_l ? bmap_near<int>(y0 - p.oy + j, ny, mode) : -2;
        row_const[k] = ysrc == -1;
        lds_row[k] = j < rows_l ? j * kStPitch : -1;
        voff_main[k] = (ysrc >= 0 && lane < nlanes) ? (unsigned)(ysrc * nx + x0 + 4 * lane) * (unsigned)sizeof(T) : kOOB;
        voff_halo[k] = (ysrc >= 0 && xsrc >= 0) ? (unsigned)(ysrc * nx + xsrc) * (unsigned)sizeof(T) : kOOB;
    }
    const bool halo_const = halo_lane && xsrc < 0;       // only in constant mode

    Raw4<T> pm[RPW];
    T ph[RPW];
    bool pconst = false;
    auto fetch = [&](int q) {                             // input plane q of the chunk (0 = zs - oz)
        int zsrc = zs - p.oz + q;
        if ((unsigned)zsrc >= (unsigned)nz) zsrc = bmap<int>(zsrc, nz, mode);
        pconst = zsrc < 0;
        zsrc = __builtin_amdgcn_readfirstlane(max(zsrc, 0));
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(in + (size_t)zsrc * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            pm[k] = load4<T>(rin, pconst ? kOOB : voff_main[k]);
            ph[k] = buf_load<T>(rin, pconst ? kOOB : voff_halo[k]);
        }
    };
    auto stage = [&](int q) {                             // registers -> ring slot of plane q
        float *slot = ring + (q % nslots) * slot_floats;
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            if (lds_row[k] < 0) continue;
            const bool c = pconst || row_const[k];
            float f[4];
            to_floats<T>(pm[k], f);
            if (c) f[0] = f[1] = f[2] = f[3] = p.cval;
            if (lane < nlanes) *reinterpret_cast<float4 *>(slot + lds_row[k] + 4 + 4 * lane) = make_float4(f[0], f[1], f[2], f[3]);
            if (halo_lane) slot[lds_row[k] + halo_pos] = (c || halo_const) ? p.cval : (float)ph[k];
        }
    };

    // ---------------------------------------------------------------- prologue: planes 0 .. wz - 1
    for (int q = 0; q < wz; q++) {
        fetch(q);
        stage(q);
    }
    __syncthreads();

    kdoubles kw = (kdoubles)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() +
                             2 * sizeof(void *) + offsetof(Stencil3Params, w));
    const int r0 = wave * RW;
    unsigned ovoff[RW];
#pragma unroll
    for (int rr = 0; rr < RW; rr++)
        ovoff[rr] = (r0 + rr < ty_act && lane < nlanes) ? (unsigned)((y0 + r0 + rr) * nx + x0 + 4 * lane) * (unsigned)sizeof(T) : kOOB;

    for (int s = 0; s < nout; s++) {
        const bool more = s + 1 < nout;
        if (more) fetch(s + wz);                          // in flight during the tap loop

        Acc acc[RW][4];
        bool started[RW];
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            started[rr] = false;
#pragma unroll
            for (int c = 0; c < 4; c++) acc[rr][c] = (Acc)0;
        }

        for (int tz = 0; tz < wz; tz++) {
            const float *slot = ring + ((s + tz) % nslots) * slot_floats + 4 * lane;
            for (int i = 0; i < RW + wy - 1; i++) {
                // weight rows this input row meets: (tz, i - rr).  Fetched first (clamped, so without a
                // branch) to overlap the scalar loads with the LDS reads below.
                Acc wv[RW][WX];
                unsigned m[RW];
                bool live[RW];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {
                    const int ty = i - rr;
                    live[rr] = ty >= 0 && ty < wy;
                    const int row = tz * wy + min(max(ty, 0), wy - 1);
                    m[rr] = DENSE ? ~0u : p.mask[row];
                    if constexpr (OP == ST_CORR) {
#pragma unroll
                        for (int tx = 0; tx < WX; tx++) wv[rr][tx] = (Acc)kw[row * WX + tx];
                    }
                }
                const float4 *rowp = reinterpret_cast<const float4 *>(slot + (r0 + i) * kStPitch);
                const float4 L = rowp[0], C = rowp[1], R = rowp[2];
                const float f[12] = {L.x, L.y, L.z, L.w, C.x, C.y, C.z, C.w, R.x, R.y, R.z, R.w};
                Acc d[4 + WX - 1];
#pragma unroll
                for (int n = 0; n < 4 + WX - 1; n++) d[n] = (Acc)f[4 - RX + n];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {
                    if (!live[rr]) continue;              // wave-uniform
#pragma unroll
                    for (int tx = 0; tx < WX; tx++) {
                        if (!DENSE && !(m[rr] >> tx & 1u)) continue;   // zero weight: skipped like the reference does
                        if constexpr (OP == ST_CORR) {
#pragma unroll
                            for (int c = 0; c < 4; c++) acc[rr][c] += d[c + tx] * wv[rr][tx];
                        } else {
                            if (!started[rr]) {           // wave-uniform: the first set tap is taken as is
#pragma unroll
                                for (int c = 0; c < 4; c++) acc[rr][c] = d[c + tx];
                                started[rr] = true;
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; c++) {
                                    const Acc x = d[c + tx];
                                    acc[rr][c] = (OP == ST_MAX ? x > acc[rr][c] : x < acc[rr][c]) ? x : acc[rr][c];
                                }
                            }
                        }
                    }
                }
            }
        }

        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(out + (size_t)(zs + s) * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
        for (int rr = 0; rr < RW; rr++) store4<T, Acc>(rout, ovoff[rr], acc[rr]);
        if (more) stage(s + wz);                          // slot of plane s - 1: nobody reads it in this step
        __syncthreads();
    }
}

static int stencil_cus() { return device_cus(); }

template <int WX, int TY, typename Acc, bool DENSE, int OP = ST_CORR, typename T = float>
static int launch_stencil3(const T *in, T *out, Stencil3Params &p, hipStream_t s)
{
    const size_t lds = (size_t)(p.wz + 1) * (TY + p.wy - 1) * kStPitch * sizeof(float);
    static size_t attr = 0;
    if (lds > attr) {
        MI_HIP(hipFuncSetAttribute((const void *)stencil3_kernel<WX, TY, Acc, DENSE, OP, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(160 * 1024)));
        attr = 160 * 1024;
    }
    p.nxt = (p.nx + 255) / 256;
    p.nyt = (p.ny + TY - 1) / TY;
    // z chunks: fill the CUs (workgroups resident per CU limited by LDS) while keeping the wz - 1 plane ramp small
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds));
    const int64_t slots = (int64_t)stencil_cus() * per_cu;
    const int64_t tiles = (int64_t)p.nxt * p.nyt;
    double best = 1e300;
    int best_nzc = 1;
    for (int nzc = 1; nzc <= std::min(p.nz, 64); nzc++) {
        const int chunk = (p.nz + nzc - 1) / nzc;
        const int real = (p.nz + chunk - 1) / chunk;
        const double rounds = (double)((tiles * real + slots - 1) / slots);
        const double cost = rounds * (chunk + p.wz - 1 + 2.0);
        if (cost < best) { best = cost; best_nzc = real; }
    }
    p.zc = (p.nz + best_nzc - 1) / best_nzc;
    p.nzc = (p.nz + p.zc - 1) / p.zc;
    const int64_t total = tiles * p.nzc;
    if (total > 0x7fffffff) { set_error("stencil: too many tiles"); return MI_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL((stencil3_kernel<WX, TY, Acc, DENSE, OP, T>), dim3((unsigned)total), dim3(kStNW * 64), lds, s, in, out, p);
    MI_HIP(hipGetLastError());
    note_kernel("mi::stencil3_kernel<%d,%d,%s,%s,%s> grid=%lld (window planes in an LDS ring)", WX, TY, sizeof(Acc) == 8 ? "double" : "float",
                DENSE ? "dense" : "mask", OP == ST_CORR ? "correlate" : OP == ST_MIN ? "min" : "max", (long long)total);
    return MI_OK;
}

template <int WX, typename Acc, int OP = ST_CORR>
static int launch_stencil3_ty(const float *in, float *out, Stencil3Params &p, bool dense, hipStream_t s)
{
    const size_t lds16 = (size_t)(p.wz + 1) * (16 + p.wy - 1) * kStPitch * sizeof(float);
    if (lds16 <= 150 * 1024)
        return dense ? launch_stencil3<WX, 16, Acc, true, OP>(in, out, p, s) : launch_stencil3<WX, 16, Acc, false, OP>(in, out, p, s);
    return dense ? launch_stencil3<WX, 8, Acc, true, OP>(in, out, p, s) : launch_stencil3<WX, 8, Acc, false, OP>(in, out, p, s);
}

// Geometry checks + parameter block shared by the tiled entry points.  keep(k) /
// value(k): k = C-order index into the window.  Returns MI_ERR_UNSUPPORTED when
// the request is outside the envelope of the tiled kernel.
template <typename Keep, typename Value>
static int stencil3_setup(const mi_array *in, const mi_array *out, const int64_t *wshape, const int *origins, int mode,
                          double cval, Keep keep, Value value, Stencil3Params *pp, int *wxk, bool *dense)
{
#define NOPE(msg) do { set_error("stencil3: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->dtype != out->dtype) NOPE("input and output dtypes differ");
    if (in->dtype != MI_F32 && in->dtype != MI_U8 && in->dtype != MI_U16 && in->dtype != MI_I16)
        NOPE("float32, uint8, uint16 or int16 only");
    if (in->ndim < 2 || in->ndim > 3) NOPE("2-D / 3-D only");
    const int pad = 3 - in->ndim;
    int64_t shape[3] = {1, 1, 1};
    int w[3] = {1, 1, 1}, off[3] = {0, 0, 0};
    for (int d = 0; d < in->ndim; d++) {
        shape[pad + d] = in->shape[d];
        if (wshape[d] < 1 || wshape[d] > 9) NOPE("window extent > 9");
        w[pad + d] = (int)wshape[d];
        off[pad + d] = (int)(wshape[d] / 2 + origins[d]);
        if (off[pad + d] < 0 || off[pad + d] >= wshape[d]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    const int64_t nz = shape[0], ny = shape[1], nx = shape[2];
    if (nx < 8 || (nx & 3)) NOPE("x extent must be a multiple of 4, >= 8");
    if (ny * nx * 4 >= ((int64_t)1 << 31) || nz > (1 << 24) || ny > (1 << 24)) NOPE("plane too large");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) NOPE("needs 16-byte aligned data");
    if (w[0] > 7 || w[1] > 7) NOPE("z / y extent > 7");
    // boundary maps are resolved with the cheap near map: the window must not be longer than the array
    if (w[0] > nz || w[1] > ny) NOPE("window longer than the array");
    if (mode == MI_MODE_CONSTANT && (double)(float)cval != cval && !std::isnan(cval)) NOPE("cval is not a float32 value");
    // x extent of the kernel: odd, centred, covering taps -off .. w-1-off (zero padded)
    const int reach = std::max(off[2], w[2] - 1 - off[2]);
    const int WXk = 2 * reach + 1;
    if (WXk > 9) NOPE("x reach > 4");
    if ((int64_t)w[0] * w[1] * WXk > kStMaxWeights || w[0] * w[1] > kStMaxRows) NOPE("window too large");
    const size_t lds8 = (size_t)(w[0] + 1) * (8 + w[1] - 1) * kStPitch * sizeof(float);
    if (lds8 > 150 * 1024) NOPE("window does not fit LDS");
#undef NOPE
    Stencil3Params &p = *pp;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.wz = w[0]; p.wy = w[1];
    p.oz = off[0]; p.oy = off[1];
    p.mode = mode;
    p.cval = (float)cval;
    int ntaps = 0;
    for (int tz = 0; tz < w[0]; tz++)
        for (int ty = 0; ty < w[1]; ty++)
            for (int tx = 0; tx < w[2]; tx++) {
                const int64_t k = ((int64_t)tz * w[1] + ty) * w[2] + tx;
                if (!keep(k)) continue;
                const int kx = tx - off[2] + reach;              // position in the padded row
                p.w[(tz * w[1] + ty) * WXk + kx] = value(k);
                p.mask[tz * w[1] + ty] |= 1u << kx;
                ntaps++;
            }
    *wxk = WXk;
    *dense = ntaps == w[0] * w[1] * WXk;          // no skipped (or padding) taps: no mask tests in the loop
    return ntaps > 0 ? MI_OK : MI_ERR_UNSUPPORTED;
}

// integer element types: one shape per (WX, OP) -- 16-row tiles, mask tests kept, double accumulation
template <int WX, int OP, typename T>
static int launch_stencil3_int(const mi_array *in, const mi_array *out, Stencil3Params &p, hipStream_t s)
{
    const size_t lds16 = (size_t)(p.wz + 1) * (16 + p.wy - 1) * kStPitch * sizeof(float);
    if (lds16 > 150 * 1024) { set_error("stencil3: window does not fit LDS for integer volumes"); return MI_ERR_UNSUPPORTED; }
    if constexpr (OP == ST_CORR)
        return launch_stencil3<WX, 16, double, false, OP, T>((const T *)in->data, (T *)out->data, p, s);
    else
        return launch_stencil3<WX, 16, float, false, OP, T>((const T *)in->data, (T *)out->data, p, s);
}

template <int OP>
static int launch_stencil3_any(const mi_array *in, const mi_array *out, Stencil3Params &p, int WXk, bool dense, bool acc_f32,
                               hipStream_t s)
{
#define BY_WX(CALL)                 \
    switch (WXk) {                  \
    case 1: return CALL(1);         \
    case 3: return CALL(3);         \
    case 5: return CALL(5);         \
    case 7: return CALL(7);         \
    default: return CALL(9);        \
    }
    if (in->dtype == MI_F32) {
        const float *ip = (const float *)in->data;
        float *op = (float *)out->data;
        if constexpr (OP == ST_CORR) {
#define CALL(WXV) (acc_f32 ? launch_stencil3_ty<WXV, float>(ip, op, p, dense, s) : launch_stencil3_ty<WXV, double>(ip, op, p, dense, s))
            BY_WX(CALL)
#undef CALL
        } else {
#define CALL(WXV) launch_stencil3_ty<WXV, float, OP>(ip, op, p, dense, s)
            BY_WX(CALL)
#undef CALL
        }
    } else if (in->dtype == MI_U8) {
#define CALL(WXV) launch_stencil3_int<WXV, OP, uint8_t>(in, out, p, s)
        BY_WX(CALL)
#undef CALL
    } else if (in->dtype == MI_U16) {
#define CALL(WXV) launch_stencil3_int<WXV, OP, uint16_t>(in, out, p, s)
        BY_WX(CALL)
#undef CALL
    } else {
#define CALL(WXV) launch_stencil3_int<WXV, OP, int16_t>(in, out, p, s)
        BY_WX(CALL)
#undef CALL
    }
#undef BY_WX
}

// Dense correlate.  MI_ERR_UNSUPPORTED (and no launch) when the request is
// outside the envelope -- the caller then uses the generic kernels.
int stencil3_tiled(const mi_array *in, const mi_array *out, const double *weights, const int64_t *wshape,
                   const int *origins, int mode, double cval, bool acc_f32, hipStream_t s)
{
    Stencil3Params p;
    int WXk;
    bool dense;
    if (in->dtype != MI_F32 && acc_f32) { set_error("stencil3: float accumulation only for float32 volumes"); return MI_ERR_UNSUPPORTED; }
    int rc = stencil3_setup(in, out, wshape, origins, mode, cval, [&](int64_t k) { return weights[k] != 0.0; },
                            [&](int64_t k) { return weights[k]; }, &p, &WXk, &dense);
    if (rc != MI_OK) return rc;
    return launch_stencil3_any<ST_CORR>(in, out, p, WXk, dense, acc_f32, s);
}

// Flat footprint minimum / maximum (grey erosion / dilation without structure
// values).  cval must already be converted to the input dtype.
int minmax3_tiled(const mi_array *in, const mi_array *out, const uint8_t *footprint, const int64_t *fshape,
                  const int *origins, int mode, double cval, bool is_max, hipStream_t s)
{
    Stencil3Params p;
    int WXk;
    bool dense;
    int rc = stencil3_setup(in, out, fshape, origins, mode, cval, [&](int64_t k) { return footprint[k] != 0; },
                            [](int64_t) { return 1.0; }, &p, &WXk, &dense);
    if (rc != MI_OK) return rc;
    return is_max ? launch_stencil3_any<ST_MAX>(in, out, p, WXk, dense, false, s)
                  : launch_stencil3_any<ST_MIN>(in, out, p, WXk, dense, false, s);
}

}  // namespace mi
