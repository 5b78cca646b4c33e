// minmax_16.hip -- flat min / max and 3 x 3 median for 16-bit images and volumes (uint16 / int16: raw microscopy,
// CT and MRI data), as barrier-free streaming passes.
//
// Replaces, for grey_erosion / grey_dilation / minimum_filter / maximum_filter with a flat `size` and for
// median_filter(size=3) on 16-bit arrays (cupyimg/scipy/ndimage/morphology.py:769-884 -> filters.py:1385-1396,
// :1560-1701), the generic one-thread-per-output kernels (2-byte loads, compares in double).  Same layout as the uint8
// kernels (minmax3d_u8.hip): lane l of a wave holds 8 consecutive pixels (one uint4, two pixels per dword), a wave a
// 1 KiB row segment, every access a coalesced 16-byte buffer load / store; comparisons are v_pk_min/max_u16 (i16), two
// pixels per lane and instruction, a one-pixel shift along x is one v_alignbit.  The wave streams along y (images,
// slice-wise filters: x window fused, ONE launch at 4 B/pixel) or z then y (volumes: two launches).  Results are
// bit-exact (comparisons only).
#include "sep_common.hpp"
#include "stream3d.hpp"

namespace mi {

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
typedef short i16x2_t __attribute__((ext_vector_type(2)));

template <bool IS_MAX, bool SIGNED>
__device__ __forceinline__ unsigned op16(unsigned a, unsigned b)
{
    if constexpr (SIGNED) {
        const i16x2_t x = __builtin_bit_cast(i16x2_t, a), y = __builtin_bit_cast(i16x2_t, b);
        return __builtin_bit_cast(unsigned, IS_MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y));
    } else {
        const u16x2_t x = __builtin_bit_cast(u16x2_t, a), y = __builtin_bit_cast(u16x2_t, b);
        return __builtin_bit_cast(unsigned, IS_MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y));
    }
}
template <bool IS_MAX, bool SIGNED>
__device__ __forceinline__ u32x4 op16v(const u32x4 a, const u32x4 b)
{
    u32x4 r;
    r.x = op16<IS_MAX, SIGNED>(a.x, b.x); r.y = op16<IS_MAX, SIGNED>(a.y, b.y);
    r.z = op16<IS_MAX, SIGNED>(a.z, b.z); r.w = op16<IS_MAX, SIGNED>(a.w, b.w);
    return r;
}
__device__ __forceinline__ unsigned al16(unsigned hi, unsigned lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }   // (lo >> 16) | (hi << 16)

struct P16Params {
    int nx, ny, nz;
    int axis;            // streamed axis: 0 = z, 1 = y
    int oa, ma;          // offset (w/2 + origin) / mode along the streamed axis
    int mx;              // x boundary mode
    unsigned cval2;      // cval replicated into both halves of a dword
    int chunk, nchunks, nxt;
    int swz;             // XCD-aware workgroup order (xcd_block())
};

// the 4-pixel block outside the tile after its boundary fix-up (two dwords, pixel order preserved)
__device__ __forceinline__ u32x2 fix_edge16(u32x2 e, int kind, int side, unsigned cval2)
{
    if (kind == EDGE_REV) return (u32x2){al16(e.y, e.y), al16(e.x, e.x)};            // p3 p2 | p1 p0
    if (kind == EDGE_SPLAT) { const unsigned s = side == 0 ? (e.x & 0xFFFFu) * 0x10001u : (e.y >> 16) * 0x10001u; return (u32x2){s, s}; }
    if (kind == EDGE_CONST) return (u32x2){cval2, cval2};
    return e;
}

// sliding min / max of width WX along x for the lane's 8 pixels; D = { L0 L1 | V0 V1 V2 V3 | R0 R1 } (2 pixels a dword)
template <int WX, bool IS_MAX, bool SIGNED>
__device__ __forceinline__ u32x4 xwin16(const unsigned (&D)[8])
{
    if constexpr (WX == 1) {
        return (u32x4){D[2], D[3], D[4], D[5]};
    } else {
        constexpr int RX = WX / 2;
        unsigned A[7];                                   // A[m] = pixels (2m + 1, 2m + 2)
#pragma unroll
        for (int m = 0; m < 7; m++) A[m] = al16(D[m + 1], D[m]);
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            // output dword k = pixels (4 + 2k, 5 + 2k); the view shifted by s pixels starts at pixel 4 + 2k + s
            unsigned r = 0;
#pragma unroll
            for (int s = -RX; s <= RX; s++) {
                const int p = 4 + 2 * k + s;
                const unsigned v = (p & 1) ? A[(p - 1) / 2] : D[p / 2];
                r = s == -RX ? v : op16<IS_MAX, SIGNED>(r, v);
            }
            o[k] = r;
        }
        return (u32x4){o[0], o[1], o[2], o[3]};
    }
}

template <int WX, int WA, bool IS_MAX, bool SIGNED>
__global__ void __launch_bounds__(256)
stream_minmax16_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, const P16Params p)
{
    constexpr int RINGN = WA - 1;
    constexpr int DEPTH = 4;
    constexpr int U = RINGN > 0 ? lcm_(RINGN, DEPTH) : DEPTH;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nother = p.axis == 0 ? ny : nz;
    const int nA = p.axis == 0 ? nz : ny;
    const int nlines = nother * p.nxt;
    const int wid = xcd_block((int)blockIdx.x, (int)gridDim.x, p.swz) * 4 + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int oth = line / p.nxt, xt = line - oth * p.nxt;
    const int x0 = xt * 512;
    const int nlanes = min(64, (nx - x0) >> 3);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;               // elements
    const unsigned strideA = p.axis == 0 ? plane : (unsigned)nx;
    const unsigned rowbase = p.axis == 0 ? (unsigned)oth * nx : (unsigned)oth * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 2u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 8 * lane)) * 2u : kOOB;

    unsigned evoff = kOOB;
    int ekind = EDGE_FWD;
    const int side = lane == 0 ? 0 : 1;
    if constexpr (WX > 1) {
        int st;
        edge_block(side, 1, x0, x0 + 8 * nlanes, nx, p.mx, &st, &ekind);
        if ((lane == 0 || lane == last) && ekind != EDGE_CONST) evoff = (rowbase + (unsigned)st) * 2u;
    }

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, nA);
    const int nsteps = a1 - a0 + WA - 1;
    const int ai0 = a0 - p.oa;

    struct Slot { u32x4 v; u32x2 e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)nA) ai = bmap<int>(ai, nA, p.ma);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * strideA * 2u;
        s.v = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0);
        if constexpr (WX > 1) s.e = __builtin_amdgcn_raw_buffer_load_b64(rin, s.cst ? kOOB : evoff, soff, 0);
        else s.e = (u32x2){0u, 0u};
    };

    u32x4 ring[RINGN > 0 ? RINGN : 1];
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J % DEPTH];
                u32x4 v = s.v;
                u32x2 ed = s.e;
                if (s.cst) { v = (u32x4){p.cval2, p.cval2, p.cval2, p.cval2}; ed = (u32x2){p.cval2, p.cval2}; }
                else ed = fix_edge16(ed, ekind, side, p.cval2);
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                u32x4 xf;
                if constexpr (WX > 1) {
                    unsigned D[8];
                    // neighbours: the left lane's last two dwords, the right lane's first two (edge lanes: the edge block)
                    D[0] = (unsigned)__builtin_amdgcn_update_dpp((int)ed.x, (int)v.z, 0x138, 0xf, 0xf, false);
                    D[1] = (unsigned)__builtin_amdgcn_update_dpp((int)ed.y, (int)v.w, 0x138, 0xf, 0xf, false);
                    unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp((int)ed.x, (int)v.x, 0x130, 0xf, 0xf, false);
                    unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp((int)ed.y, (int)v.y, 0x130, 0xf, 0xf, false);
                    if (lane == last) { r0 = ed.x; r1 = ed.y; }
                    D[2] = v.x; D[3] = v.y; D[4] = v.z; D[5] = v.w; D[6] = r0; D[7] = r1;
                    xf = xwin16<WX, IS_MAX, SIGNED>(D);
                } else {
                    xf = v;
                }
                if (i >= WA - 1) {
                    u32x4 a = xf;
                    if constexpr (RINGN > 0) {
#pragma unroll
                        for (int k = 0; k < RINGN; k++) a = op16v<IS_MAX, SIGNED>(a, ring[k]);
                    }
                    const unsigned so = (unsigned)(a0 + i - (WA - 1)) * strideA * 2u;
                    buffer_store_b128_soff(a, rout, voff, so);
                }
                if constexpr (RINGN > 0) ring[J % RINGN] = xf;
            }
        });
    }
}

static void plan_chunks16(int nlines, int nA, int ramp, int *chunk, int *nchunks);

// ---------------------------------------------------------------------------
// Flat footprints whose rows are centred runs (disk, diamond, cross, square; see runs_minmax_u8_kernel in
// minmax3d_u8.hip): the previous WA - 1 raw rows stay in registers as eight-dword windows, every footprint row
// contributes one x window of its own width.
// ---------------------------------------------------------------------------
struct P16RunParams {
    int nx, ny, nz;
    int mx, my;
    unsigned cval2;
    int chunk, nchunks, nxt;
    int swz;
    int hw[9];           // half width of the run of footprint row r, -1 = empty row
};

struct Win16 { unsigned d[8]; };

template <bool IS_MAX, bool SIGNED>
__device__ __forceinline__ u32x4 xrun16(const Win16 &w, int hw)
{
    switch (hw) {           // wave-uniform
    case 0: return xwin16<1, IS_MAX, SIGNED>(w.d);
    case 1: return xwin16<3, IS_MAX, SIGNED>(w.d);
    case 2: return xwin16<5, IS_MAX, SIGNED>(w.d);
    case 3: return xwin16<7, IS_MAX, SIGNED>(w.d);
    default: return xwin16<9, IS_MAX, SIGNED>(w.d);
    }
}

template <int WA, bool IS_MAX, bool SIGNED>
__global__ void __launch_bounds__(256)
runs_minmax16_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, const P16RunParams p)
{
    constexpr int DEPTH = 2;
    constexpr int RINGN = WA - 1;
    constexpr int U = RINGN > 0 ? lcm_(RINGN, DEPTH) : DEPTH;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nlines = nz * p.nxt;
    const int wid = xcd_block((int)blockIdx.x, (int)gridDim.x, p.swz) * 4 + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int z = line / p.nxt, xt = line - z * p.nxt;
    const int x0 = xt * 512;
    const int nlanes = min(64, (nx - x0) >> 3);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned rowbase = (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 2u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 8 * lane)) * 2u : kOOB;
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_block(side, 1, x0, x0 + 8 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? (rowbase + (unsigned)est) * 2u : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, ny);
    const int nsteps = a1 - a0 + WA - 1;
    const int ai0 = a0 - WA / 2;

    struct Slot { u32x4 v; u32x2 e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)ny) ai = bmap<int>(ai, ny, p.my);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * (unsigned)nx * 2u;
        s.v = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0);
        s.e = __builtin_amdgcn_raw_buffer_load_b64(rin, s.cst ? kOOB : evoff, soff, 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    Win16 ring[RINGN > 0 ? RINGN : 1];
    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J % DEPTH];
                u32x4 v = s.v;
                u32x2 ed = s.e;
                if (s.cst) { v = (u32x4){p.cval2, p.cval2, p.cval2, p.cval2}; ed = (u32x2){p.cval2, p.cval2}; }
                else ed = fix_edge16(ed, ekind, side, p.cval2);
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                Win16 w;
                w.d[0] = (unsigned)__builtin_amdgcn_update_dpp((int)ed.x, (int)v.z, 0x138, 0xf, 0xf, false);
                w.d[1] = (unsigned)__builtin_amdgcn_update_dpp((int)ed.y, (int)v.w, 0x138, 0xf, 0xf, false);
                unsigned r0 = (unsigned)__builtin_amdgcn_update_dpp((int)ed.x, (int)v.x, 0x130, 0xf, 0xf, false);
                unsigned r1 = (unsigned)__builtin_amdgcn_update_dpp((int)ed.y, (int)v.y, 0x130, 0xf, 0xf, false);
                if (lane == last) { r0 = ed.x; r1 = ed.y; }
                w.d[2] = v.x; w.d[3] = v.y; w.d[4] = v.z; w.d[5] = v.w; w.d[6] = r0; w.d[7] = r1;
                if (i >= WA - 1) {
                    // rows that share a half width are combined first, then ONE x window per distinct half width
                    u32x4 a = (u32x4){0u, 0u, 0u, 0u};
                    bool have = false;
                    static_for<5>([&](auto HH) {
                        constexpr int h = decltype(HH)::value;
                        Win16 g;
                        bool any = false;
                        static_for<WA>([&](auto KK) {
                            constexpr int k = decltype(KK)::value;
                            if (p.hw[k] == h) {
                                const Win16 &r = k == WA - 1 ? w : ring[(J + k) % (RINGN > 0 ? RINGN : 1)];
                                if (any) {
#pragma unroll
                                    for (int d = 0; d < 8; d++) g.d[d] = op16<IS_MAX, SIGNED>(g.d[d], r.d[d]);
                                } else {
                                    g = r;
                                }
                                any = true;
                            }
                        });
                        if (any) {
                            const u32x4 t = xwin16<2 * h + 1, IS_MAX, SIGNED>(g.d);
                            a = have ? op16v<IS_MAX, SIGNED>(a, t) : t;
                            have = true;
                        }
                    });
                    const unsigned so = (unsigned)(a0 + i - (WA - 1)) * (unsigned)nx * 2u;
                    buffer_store_b128_soff(a, rout, voff, so);
                }
                if constexpr (RINGN > 0) ring[J % RINGN] = w;
            }
        });
    }
}

template <int WA>
static int launch_runs16(const uint16_t *in, uint16_t *out, P16RunParams &p, bool is_max, bool is_signed, hipStream_t s)
{
    plan_chunks16(p.nz * p.nxt, p.ny, WA - 1, &p.chunk, &p.nchunks);
    const int waves = p.nz * p.nxt * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz * 2);
    const dim3 grid((waves + 3) / 4), block(256);
    if (is_signed) {
        if (is_max) hipLaunchKernelGGL((runs_minmax16_kernel<WA, true, true>), grid, block, 0, s, in, out, p);
        else hipLaunchKernelGGL((runs_minmax16_kernel<WA, false, true>), grid, block, 0, s, in, out, p);
    } else {
        if (is_max) hipLaunchKernelGGL((runs_minmax16_kernel<WA, true, false>), grid, block, 0, s, in, out, p);
        else hipLaunchKernelGGL((runs_minmax16_kernel<WA, false, false>), grid, block, 0, s, in, out, p);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------
// uniform_filter on uint16 / int16 images with a result of the same dtype, in integer arithmetic (method and the
// argument for the division: box2d_u8_kernel, minmax3d_u8.hip).  Sums need 32 bits here (9 x 65535), so the pixels of a
// lane are unpacked to one int each; |S| < 2^24 is exact in float32, the quotient is at most 65535 (product error below
// 0.008), the offset is 0.02 with the sign of S (truncation toward zero, as the C cast of SciPy's double).
// ---------------------------------------------------------------------------
struct Box16Params {
    int nx, ny, nz;
    int axis;            // streamed axis: 1 = y (x window fused), 0 = z (WX == 1)
    int oy;
    int mx, my;
    unsigned cval2;
    int chunk, nchunks, nxt;
    int swz;
    float ry, rx;
};

template <bool SIGNED> __device__ __forceinline__ int px_lo(unsigned d) { return SIGNED ? ((int)(d << 16)) >> 16 : (int)(d & 0xFFFFu); }
template <bool SIGNED> __device__ __forceinline__ int px_hi(unsigned d) { return SIGNED ? ((int)d) >> 16 : (int)(d >> 16); }
__device__ __forceinline__ int div_trunc(int s, float r) { return (int)fmaf((float)s, r, s >= 0 ? 0.02f : -0.02f); }

template <int WX, int WY, bool SIGNED>
__global__ void __launch_bounds__(256)
box2d_16_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, const Box16Params p)
{
    constexpr int DEPTH = 2;
    constexpr int U = WY % DEPTH == 0 ? WY : WY * DEPTH;
    constexpr int RX = WX / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nother = p.axis == 0 ? ny : nz;
    const int nA = p.axis == 0 ? nz : ny;
    const int nlines = nother * p.nxt;
    const int wid = xcd_block((int)blockIdx.x, (int)gridDim.x, p.swz) * 4 + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int z = line / p.nxt, xt = line - z * p.nxt;    // z: index along the other axis
    const int x0 = xt * 512;
    const int nlanes = min(64, (nx - x0) >> 3);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned strideA = p.axis == 0 ? plane : (unsigned)nx;
    const unsigned rowbase = p.axis == 0 ? (unsigned)z * (unsigned)nx : (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 2u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 8 * lane)) * 2u : kOOB;
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_block(side, 1, x0, x0 + 8 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? (rowbase + (unsigned)est) * 2u : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, nA);
    const int nsteps = a1 - a0 + WY - 1;
    const int ai0 = a0 - p.oy;

    struct Slot { u32x4 v; u32x2 e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)nA) ai = bmap<int>(ai, nA, p.my);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * strideA * 2u;
        s.v = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0);
        s.e = __builtin_amdgcn_raw_buffer_load_b64(rin, s.cst ? kOOB : evoff, soff, 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    // a row: the lane's 8 pixels and the 4 pixels of its edge block, one int each
    struct Row { int v[12]; };
    auto unpack = [&](const u32x4 v, const u32x2 e) {
        Row r;
        r.v[0] = px_lo<SIGNED>(v.x); r.v[1] = px_hi<SIGNED>(v.x); r.v[2] = px_lo<SIGNED>(v.y); r.v[3] = px_hi<SIGNED>(v.y);
        r.v[4] = px_lo<SIGNED>(v.z); r.v[5] = px_hi<SIGNED>(v.z); r.v[6] = px_lo<SIGNED>(v.w); r.v[7] = px_hi<SIGNED>(v.w);
        r.v[8] = px_lo<SIGNED>(e.x); r.v[9] = px_hi<SIGNED>(e.x); r.v[10] = px_lo<SIGNED>(e.y); r.v[11] = px_hi<SIGNED>(e.y);
        return r;
    };
    struct Packed { u32x4 v; u32x2 e; };
    Packed ring[WY];                                  // the last WY raw rows, packed (unpacked again when they leave the window)
    Row sum;
#pragma unroll
    for (int k = 0; k < 12; k++) sum.v[k] = 0;
#pragma unroll
    for (int r = 0; r < WY; r++) { ring[r].v = (u32x4){0u, 0u, 0u, 0u}; ring[r].e = (u32x2){0u, 0u}; }

    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J % DEPTH];
                u32x4 v = s.v;
                u32x2 ed = s.e;
                if (s.cst) { v = (u32x4){p.cval2, p.cval2, p.cval2, p.cval2}; ed = (u32x2){p.cval2, p.cval2}; }
                else ed = fix_edge16(ed, ekind, side, p.cval2);
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                const Row cur = unpack(v, ed);
                const Row old = unpack(ring[J % WY].v, ring[J % WY].e);
#pragma unroll
                for (int k = 0; k < 12; k++) sum.v[k] += cur.v[k] - old.v[k];
                ring[J % WY].v = v; ring[J % WY].e = ed;
                if (i >= WY - 1) {
                    int q[12];
#pragma unroll
                    for (int k = 0; k < 12; k++) q[k] = WY == 1 ? sum.v[k] : div_trunc(sum.v[k], p.ry);
                    int o[8];
                    if constexpr (WX == 1) {
#pragma unroll
                        for (int k = 0; k < 8; k++) o[k] = q[k];
                    } else {
                        // Q = [4 pixels left | own 8 | 4 pixels right] of the quotient row
                        int Q[16];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            Q[k] = __builtin_amdgcn_update_dpp(q[8 + k], q[4 + k], 0x138, 0xf, 0xf, false);
                            const int r = __builtin_amdgcn_update_dpp(q[8 + k], q[k], 0x130, 0xf, 0xf, false);
                            Q[12 + k] = lane == last ? q[8 + k] : r;
                        }
#pragma unroll
                        for (int k = 0; k < 8; k++) Q[4 + k] = q[k];
                        int acc = 0;
#pragma unroll
                        for (int t = -RX; t <= RX; t++) acc += Q[4 + t];
                        o[0] = div_trunc(acc, p.rx);
#pragma unroll
                        for (int j = 1; j < 8; j++) {
                            acc += Q[4 + j + RX] - Q[4 + j - 1 - RX];
                            o[j] = div_trunc(acc, p.rx);
                        }
                    }
                    u32x4 u;
                    u.x = ((unsigned)o[0] & 0xFFFFu) | ((unsigned)o[1] << 16);
                    u.y = ((unsigned)o[2] & 0xFFFFu) | ((unsigned)o[3] << 16);
                    u.z = ((unsigned)o[4] & 0xFFFFu) | ((unsigned)o[5] << 16);
                    u.w = ((unsigned)o[6] & 0xFFFFu) | ((unsigned)o[7] << 16);
                    const unsigned so = (unsigned)(a0 + i - (WY - 1)) * strideA * 2u;
                    buffer_store_b128_soff(u, rout, voff, so);
                }
            }
        });
    }
}

template <int WX, int WY>
static int launch_box16(const uint16_t *in, uint16_t *out, Box16Params &p, bool is_signed, hipStream_t s)
{
    const int nlines = (p.axis == 0 ? p.ny : p.nz) * p.nxt;
    plan_chunks16(nlines, p.axis == 0 ? p.nz : p.ny, WY - 1, &p.chunk, &p.nchunks);
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz * 2);
    if (is_signed) hipLaunchKernelGGL((box2d_16_kernel<WX, WY, true>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    else hipLaunchKernelGGL((box2d_16_kernel<WX, WY, false>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <int WX>
static int launch_box16_wy(int wy, const uint16_t *in, uint16_t *out, Box16Params &p, bool is_signed, hipStream_t s)
{
    switch (wy) {
    case 1: return launch_box16<WX, 1>(in, out, p, is_signed, s);
    case 3: return launch_box16<WX, 3>(in, out, p, is_signed, s);
    case 5: return launch_box16<WX, 5>(in, out, p, is_signed, s);
    case 7: return launch_box16<WX, 7>(in, out, p, is_signed, s);
    default: return launch_box16<WX, 9>(in, out, p, is_signed, s);
    }
}

// ---------------------------------------------------------------------------
// 3 x 3 median (method: median2d.hip), two pixels per dword
// ---------------------------------------------------------------------------
template <bool SIGNED>
__device__ __forceinline__ unsigned med3_16(unsigned a, unsigned b, unsigned c)
{
    const unsigned mn = op16<false, SIGNED>(a, b), mx = op16<true, SIGNED>(a, b);
    return op16<true, SIGNED>(mn, op16<false, SIGNED>(mx, c));
}
template <bool SIGNED>
__device__ __forceinline__ void sort3_16(unsigned a, unsigned b, unsigned c, unsigned &lo, unsigned &mid, unsigned &hi)
{
    const unsigned mn = op16<false, SIGNED>(a, b), mx = op16<true, SIGNED>(a, b);
    lo = op16<false, SIGNED>(mn, c);
    hi = op16<true, SIGNED>(mx, c);
    mid = op16<true, SIGNED>(mn, op16<false, SIGNED>(mx, c));
}

template <bool SIGNED>
__global__ void __launch_bounds__(256)
median3x3_16_kernel(const uint16_t *__restrict__ in, uint16_t *__restrict__ out, const P16Params p)
{
    constexpr int DEPTH = 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nlines = nz * p.nxt;
    const int wid = xcd_block((int)blockIdx.x, (int)gridDim.x, p.swz) * 4 + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int z = line / p.nxt, xt = line - z * p.nxt;
    const int x0 = xt * 512;
    const int nlanes = min(64, (nx - x0) >> 3);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned rowbase = (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 2u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 8 * lane)) * 2u : kOOB;
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_block(side, 1, x0, x0 + 8 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? (rowbase + (unsigned)est) * 2u : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, ny);
    const int nsteps = a1 - a0 + 2;
    const int ai0 = a0 - 1;

    struct Slot { u32x4 v; u32x2 e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        const int ai = bmap<int>(ai0 + i, ny, p.ma);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * (unsigned)nx * 2u;
        s.v = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0);
        s.e = __builtin_amdgcn_raw_buffer_load_b64(rin, s.cst ? kOOB : evoff, soff, 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    u32x4 rv[2] = {(u32x4){0u, 0u, 0u, 0u}, (u32x4){0u, 0u, 0u, 0u}};
    unsigned re[2] = {0u, 0u};          // the neighbouring pixel of the two previous rows, in the half the shifts read
    for (int i0 = 0; i0 < nsteps; i0 += DEPTH) {
        static_for<DEPTH>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J];
                u32x4 v = s.v;
                u32x2 ed = s.e;
                if (s.cst) { v = (u32x4){p.cval2, p.cval2, p.cval2, p.cval2}; ed = (u32x2){p.cval2, p.cval2}; }
                else ed = fix_edge16(ed, ekind, side, p.cval2);
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                // lane 0 needs the LAST pixel of the left block (high half of its second dword), lane `last` the FIRST
                // pixel of the right block (low half of its first dword)
                const unsigned ce = side == 0 ? ed.y : ed.x;
                if (i >= 2) {
                    unsigned lo[4], mid[4], hi[4];
                    const unsigned a[4] = {rv[0].x, rv[0].y, rv[0].z, rv[0].w}, b[4] = {rv[1].x, rv[1].y, rv[1].z, rv[1].w};
                    const unsigned cc[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 4; k++) sort3_16<SIGNED>(a[k], b[k], cc[k], lo[k], mid[k], hi[k]);
                    unsigned elo, emid, ehi;
                    sort3_16<SIGNED>(re[0], re[1], ce, elo, emid, ehi);
                    auto from_left = [&](unsigned keep, unsigned x) {
                        return (unsigned)__builtin_amdgcn_update_dpp((int)keep, (int)x, 0x138, 0xf, 0xf, false);
                    };
                    auto from_right = [&](unsigned keep, unsigned x) {
                        const unsigned r = (unsigned)__builtin_amdgcn_update_dpp((int)keep, (int)x, 0x130, 0xf, 0xf, false);
                        return lane == last ? keep : r;
                    };
                    const unsigned lo_p = from_left(elo, lo[3]), mid_p = from_left(emid, mid[3]), hi_p = from_left(ehi, hi[3]);
                    const unsigned lo_n = from_right(elo, lo[0]), mid_n = from_right(emid, mid[0]), hi_n = from_right(ehi, hi[0]);
                    unsigned o[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        // views shifted by one pixel: left = (prev.hi, cur.lo), right = (cur.hi, next.lo)
                        const unsigned lo_l = al16(lo[k], k ? lo[k - 1] : lo_p), lo_r = al16(k < 3 ? lo[k + 1] : lo_n, lo[k]);
                        const unsigned mid_l = al16(mid[k], k ? mid[k - 1] : mid_p), mid_r = al16(k < 3 ? mid[k + 1] : mid_n, mid[k]);
                        const unsigned hi_l = al16(hi[k], k ? hi[k - 1] : hi_p), hi_r = al16(k < 3 ? hi[k + 1] : hi_n, hi[k]);
                        const unsigned mxlo = op16<true, SIGNED>(op16<true, SIGNED>(lo_l, lo[k]), lo_r);
                        const unsigned mnhi = op16<false, SIGNED>(op16<false, SIGNED>(hi_l, hi[k]), hi_r);
                        o[k] = med3_16<SIGNED>(mxlo, med3_16<SIGNED>(mid_l, mid[k], mid_r), mnhi);
                    }
                    const unsigned so = (unsigned)(a0 + i - 2) * (unsigned)nx * 2u;
                    buffer_store_b128_soff((u32x4){o[0], o[1], o[2], o[3]}, rout, voff, so);
                }
                rv[J % 2] = v;
                re[J % 2] = ce;
            }
        });
    }
}

static void plan_chunks16(int nlines, int nA, int ramp, int *chunk, int *nchunks)
{
    int nch = 1;
    double best = 1e300;
    for (int c = 1; c <= nA && c <= 2048; c++) {
        const int ck = (nA + c - 1) / c;
        if (c > 1 && ck < 8) break;
        const int real = (nA + ck - 1) / ck;
        const double rounds = std::max(1.0, (double)nlines * real / 4096.0);
        const double cost = rounds * (ck + ramp + 4.0);
        if (cost < best * 0.999) { best = cost; nch = real; }
    }
    *chunk = (nA + nch - 1) / nch;
    *nchunks = (nA + *chunk - 1) / *chunk;
}

template <int WX, int WA, bool IS_MAX, bool SIGNED>
static int launch16(const uint16_t *in, uint16_t *out, P16Params &p, hipStream_t s)
{
    const int nA = p.axis == 0 ? p.nz : p.ny;
    const int nlines = (p.axis == 0 ? p.ny : p.nz) * p.nxt;
    plan_chunks16(nlines, nA, WA - 1, &p.chunk, &p.nchunks);
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz * 2);
    hipLaunchKernelGGL((stream_minmax16_kernel<WX, WA, IS_MAX, SIGNED>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <bool IS_MAX, bool SIGNED>
static int pass16(const uint16_t *in, uint16_t *out, P16Params &p, int wx, int wa, hipStream_t s)
{
#define MM(WXV, WAV) return launch16<WXV, WAV, IS_MAX, SIGNED>(in, out, p, s)
    if (wx == 1) {
        switch (wa) { case 3: MM(1, 3); case 5: MM(1, 5); case 7: MM(1, 7); case 9: MM(1, 9); }
    } else if (wa == 1) {
        switch (wx) { case 3: MM(3, 1); case 5: MM(5, 1); case 7: MM(7, 1); case 9: MM(9, 1); }
    } else if (wa == wx) {
        switch (wx) { case 3: MM(3, 3); case 5: MM(5, 5); case 7: MM(7, 7); case 9: MM(9, 9); }
    }
#undef MM
    set_error("16-bit min/max pass: unsupported sizes %d/%d", wx, wa);
    return MI_ERR_UNSUPPORTED;
}

static int pass16_any(bool is_max, bool is_signed, const uint16_t *in, uint16_t *out, P16Params &p, int wx, int wa, hipStream_t s)
{
    if (is_signed) return is_max ? pass16<true, true>(in, out, p, wx, wa, s) : pass16<false, true>(in, out, p, wx, wa, s);
    return is_max ? pass16<true, false>(in, out, p, wx, wa, s) : pass16<false, false>(in, out, p, wx, wa, s);
}

// geometry shared by the two entry points below; 2-D arrays are one-plane volumes
static int geometry16(const mi_array *in, const mi_array *out, const char *who, int64_t *nz, int64_t *ny, int64_t *nx)
{
#define UNSUP(msg) do { set_error("%s: %s", who, msg); return MI_ERR_UNSUPPORTED; } while (0)
    if ((in->ndim != 2 && in->ndim != 3) || in->dtype != out->dtype || (in->dtype != MI_U16 && in->dtype != MI_I16))
        UNSUP("needs 2-D / 3-D uint16 or int16 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int nd = in->ndim;
    *nz = nd == 3 ? in->shape[0] : 1; *ny = in->shape[nd - 2]; *nx = in->shape[nd - 1];
    if (*nz < 1 || *ny < 1 || *nx < 16 || (*nx & 7)) UNSUP("x extent must be a multiple of 8, >= 16");
    // the last tile needs two lanes: lane 0 takes the block left of the tile, lane `last` the block right of it
    { const int64_t tail = *nx & 511; if (tail != 0 && tail < 16) UNSUP("x extent unsuitable for the streaming x window"); }
    if (*nz * *ny * *nx * 2 >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
#undef UNSUP
    return MI_OK;
}

int run_median3x3_16(const mi_array *in, const mi_array *out, int mx, int my, double cval, hipStream_t s)
{
    int64_t nz, ny, nx;
    int rc = geometry16(in, out, "median3x3", &nz, &ny, &nx);
    if (rc) return rc;
    const bool is_signed = in->dtype == MI_I16;
    if (my == MI_MODE_CONSTANT || mx == MI_MODE_CONSTANT) {
        const double lo = is_signed ? -32768.0 : 0.0, hi = is_signed ? 32767.0 : 65535.0;
        if (!(cval >= lo && cval <= hi && cval == (double)(int)cval)) { set_error("median3x3: cval is not a value of the dtype"); return MI_ERR_UNSUPPORTED; }
    }
    P16Params p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.axis = 1; p.oa = 1; p.ma = my; p.mx = mx;
    p.cval2 = ((unsigned)(int)cval & 0xFFFFu) * 0x10001u;
    p.nxt = (int)((nx + 511) / 512);
    plan_chunks16(p.nz * p.nxt, p.ny, 2, &p.chunk, &p.nchunks);
    const int waves = p.nz * p.nxt * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz * 2);
    if (is_signed)
        hipLaunchKernelGGL(median3x3_16_kernel<true>, dim3((waves + 3) / 4), dim3(256), 0, s, (const uint16_t *)in->data, (uint16_t *)out->data, p);
    else
        hipLaunchKernelGGL(median3x3_16_kernel<false>, dim3((waves + 3) / 4), dim3(256), 0, s, (const uint16_t *)in->data, (uint16_t *)out->data, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

}  // namespace mi

using namespace mi;

/* Separable flat min / max on a uint16 / int16 image or volume (declared in include/mi355img.h). */
namespace mi {
int minmax3d_16_ragged(const mi_array *in, const mi_array *out, const int size[3], const int mode[3], int cval, int is_max,
                       hipStream_t s);   // minmax3d_16r.hip
}

extern "C" int mi_minmax3d_16(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                              const int mode[3], int cval, int is_max, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(size && origin && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    // r6: rows that are not a multiple of 16 bytes as they lie (cubic 3 / 5 / 7, volumes the caches hold)
    if (in->ndim == 3 && in->dtype == out->dtype && (in->dtype == MI_U16 || in->dtype == MI_I16) && is_contiguous(in) && is_contiguous(out) &&
        in->data != out->data && (in->shape[2] & 7) && !origin[0] && !origin[1] && !origin[2] &&
        cval >= (in->dtype == MI_I16 ? -32768 : 0) && cval <= (in->dtype == MI_I16 ? 32767 : 65535)) {
        rc = minmax3d_16_ragged(in, out, size, mode, cval, is_max, resolve_stream(stream));
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    int64_t nz, ny, nx;
    if ((rc = geometry16(in, out, "minmax3d_16", &nz, &ny, &nx))) return rc;
#define UNSUP(msg) do { set_error("minmax3d_16: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    for (int a = 0; a < 3; a++)
        if (size[a] < 1 || size[a] > 9 || !(size[a] & 1) || origin[a] != 0) UNSUP("sizes must be odd, <= 9, with origin 0");
    const bool is_signed = in->dtype == MI_I16;
    if (cval < (is_signed ? -32768 : 0) || cval > (is_signed ? 32767 : 65535)) UNSUP("cval outside the dtype");
    const int mz = filter_mode(mode[0]), my = filter_mode(mode[1]), mx = filter_mode(mode[2]);
    hipStream_t s = resolve_stream(stream);
    struct Pass { int axis, wa, oa, ma, wx; };
    Pass passes[3];
    int np = 0;
    const int *w = size;
    const bool fuse_xz = w[2] > 1 && w[2] == w[0];
    const bool fuse_xy = !fuse_xz && w[2] > 1 && w[2] == w[1];
    if (w[2] > 1 && !fuse_xz && !fuse_xy) passes[np++] = {1, 1, 0, my, w[2]};
    if (w[0] > 1) passes[np++] = {0, w[0], w[0] / 2, mz, fuse_xz ? w[2] : 1};
    if (w[1] > 1) passes[np++] = {1, w[1], w[1] / 2, my, fuse_xy ? w[2] : 1};
    if (np == 0) UNSUP("nothing to filter");
    const size_t bytes = (size_t)(nz * ny * nx) * 2;
    void *tmp[2] = {nullptr, nullptr};
    for (int t = 0; t < np - 1 && t < 2; t++)
        if ((rc = pool_alloc(&tmp[t], bytes, s))) { if (tmp[0]) pool_free(tmp[0]); return rc; }
    const uint16_t *src = (const uint16_t *)in->data;
    for (int i = 0; i < np && rc == MI_OK; i++) {
        uint16_t *dst = i == np - 1 ? (uint16_t *)out->data : (uint16_t *)tmp[i & 1];
        const Pass &q = passes[i];
        P16Params p;
        memset(&p, 0, sizeof(p));
        p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
        p.axis = q.axis; p.oa = q.oa; p.ma = q.ma; p.mx = mx;
        p.cval2 = ((unsigned)cval & 0xFFFFu) * 0x10001u;
        p.nxt = (int)((nx + 511) / 512);
        rc = pass16_any(is_max != 0, is_signed, src, dst, p, q.wx, q.wa, s);
        src = dst;
    }
    for (int t = 0; t < 2; t++) if (tmp[t]) pool_free(tmp[t]);
    return rc;
#undef UNSUP
}

/* Flat footprint given as centred runs per row, uint16 / int16 (declared in include/mi355img.h). */
extern "C" int mi_minmax_runs_16(const mi_array *in, const mi_array *out, int nrows, const int *half_width, const int mode[2],
                                 int cval, int is_max, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(half_width && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    int64_t nz, ny, nx;
    if ((rc = geometry16(in, out, "minmax_runs_16", &nz, &ny, &nx))) return rc;
#define UNSUP(msg) do { set_error("minmax_runs_16: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (nrows < 1 || nrows > 9 || !(nrows & 1)) UNSUP("1, 3, 5, 7 or 9 footprint rows");
    const bool is_signed = in->dtype == MI_I16;
    if (cval < (is_signed ? -32768 : 0) || cval > (is_signed ? 32767 : 65535)) UNSUP("cval outside the dtype");
    P16RunParams p;
    memset(&p, 0, sizeof(p));
    bool any = false;
    for (int r = 0; r < 9; r++) p.hw[r] = -1;
    for (int r = 0; r < nrows; r++) {
        if (half_width[r] < -1 || half_width[r] > 4) UNSUP("runs of at most 9 pixels");
        p.hw[r] = half_width[r];
        any = any || half_width[r] >= 0;
    }
    if (!any) UNSUP("empty footprint");
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.my = filter_mode(mode[0]); p.mx = filter_mode(mode[1]);
    p.cval2 = ((unsigned)cval & 0xFFFFu) * 0x10001u;
    p.nxt = (int)((nx + 511) / 512);
    hipStream_t s = resolve_stream(stream);
    const uint16_t *ip = (const uint16_t *)in->data;
    uint16_t *op = (uint16_t *)out->data;
    switch (nrows) {
    case 1: return launch_runs16<1>(ip, op, p, is_max != 0, is_signed, s);
    case 3: return launch_runs16<3>(ip, op, p, is_max != 0, is_signed, s);
    case 5: return launch_runs16<5>(ip, op, p, is_max != 0, is_signed, s);
    case 7: return launch_runs16<7>(ip, op, p, is_max != 0, is_signed, s);
    default: return launch_runs16<9>(ip, op, p, is_max != 0, is_signed, s);
    }
#undef UNSUP
}

/* uniform_filter on a uint16 / int16 image (volume: slice by slice), result in the same dtype (declared in
 * include/mi355img.h). */
extern "C" int mi_uniform2d_16(const mi_array *in, const mi_array *out, const int size[2], int origin_y, const int mode[2],
                               int cval, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(size && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    int64_t nz, ny, nx;
    if ((rc = geometry16(in, out, "uniform2d_16", &nz, &ny, &nx))) return rc;
#define UNSUP(msg) do { set_error("uniform2d_16: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    const int wy = size[0], wx = size[1];
    if (wy < 1 || wy > 9 || !(wy & 1) || wx < 1 || wx > 9 || !(wx & 1)) UNSUP("sizes must be odd and <= 9");
    if (wy == 1 && wx == 1) UNSUP("nothing to filter");
    const int oy = wy / 2 + origin_y;
    if (oy < 0 || oy >= wy) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    const bool is_signed = in->dtype == MI_I16;
    if (cval < (is_signed ? -32768 : 0) || cval > (is_signed ? 32767 : 65535)) UNSUP("cval outside the dtype");
    Box16Params p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.axis = 1;
    p.oy = oy;
    p.my = filter_mode(mode[0]); p.mx = filter_mode(mode[1]);
    p.cval2 = ((unsigned)cval & 0xFFFFu) * 0x10001u;
    p.nxt = (int)((nx + 511) / 512);
    p.ry = (float)(1.0 / wy); p.rx = (float)(1.0 / wx);
    hipStream_t s = resolve_stream(stream);
    const uint16_t *ip = (const uint16_t *)in->data;
    uint16_t *op = (uint16_t *)out->data;
    switch (wx) {
    case 1: return launch_box16_wy<1>(wy, ip, op, p, is_signed, s);
    case 3: return launch_box16_wy<3>(wy, ip, op, p, is_signed, s);
    case 5: return launch_box16_wy<5>(wy, ip, op, p, is_signed, s);
    case 7: return launch_box16_wy<7>(wy, ip, op, p, is_signed, s);
    default: return launch_box16_wy<9>(wy, ip, op, p, is_signed, s);
    }
#undef UNSUP
}

/* The z pass of uniform_filter on a uint16 / int16 volume (declared in include/mi355img.h). */
extern "C" int mi_uniform_z_16(const mi_array *in, const mi_array *out, int size_z, int origin_z, int mode_z, int cval,
                               mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
    int64_t nz, ny, nx;
    if ((rc = geometry16(in, out, "uniform_z_16", &nz, &ny, &nx))) return rc;
#define UNSUP(msg) do { set_error("uniform_z_16: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3) UNSUP("needs a volume");
    if (size_z < 3 || size_z > 9 || !(size_z & 1)) UNSUP("odd size 3 .. 9");
    const int oz = size_z / 2 + origin_z;
    if (oz < 0 || oz >= size_z) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    const bool is_signed = in->dtype == MI_I16;
    if (cval < (is_signed ? -32768 : 0) || cval > (is_signed ? 32767 : 65535)) UNSUP("cval outside the dtype");
    Box16Params p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.axis = 0;
    p.oy = oz;
    p.my = filter_mode(mode_z); p.mx = MI_MODE_REFLECT;
    p.cval2 = ((unsigned)cval & 0xFFFFu) * 0x10001u;
    p.nxt = (int)((nx + 511) / 512);
    p.ry = (float)(1.0 / size_z); p.rx = 1.0f;
    return launch_box16_wy<1>(size_z, (const uint16_t *)in->data, (uint16_t *)out->data, p, is_signed, resolve_stream(stream));
#undef UNSUP
}
