// minmax3d_u8.hip -- separable 3-D min / max for uint8 volumes in two
// barrier-free streaming launches (x pass fused into the z pass, then y pass).
//
// Replaces, for grey_erosion / grey_dilation / minimum_filter / maximum_filter
// with a flat full `size` structuring element on uint8 volumes
// (cupyimg/scipy/ndimage/morphology.py:769-884 -> filters.py:1385-1396), the
// reference's three K2 launches + two zero-filled ping-pong volumes, whose
// generated kernel reads one byte per lane per tap and compares in double
// (filters.py:1522-1528).  Results are integer-exact (no arithmetic, only
// comparisons), so parity with SciPy is bit-exact.
//
// Data layout: lane l of a wave holds 16 consecutive voxels (one uint4), a wave
// a 1 KiB row segment, so every access is a coalesced buffer_load/store_dwordx4.
// gfx950 has no packed 8-bit min/max, so bytes are kept "even/odd split":
// every dword is carried as E = bytes {0,2} and O = bytes {1,3}, each widened to
// two u16 lanes, and all comparisons are v_pk_min_u16 / v_pk_max_u16 (two
// voxels per lane per instruction).  A one-voxel shift along x swaps the roles
// of E and O plus one v_alignbit; sliding windows along x are built by
// doubling (2, 4, then W), along the streamed axis from a register ring of the
// previous W-1 samples that is rotated by unrolling.
#include "long_common.hpp"       // stream_nt_for (+ sep_common.hpp, stream3d.hpp)

namespace mi {


typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <bool IS_MAX>
__device__ __forceinline__ unsigned op2(unsigned a, unsigned b)
{
    // elementwise on two packed u16 (v_pk_min_u16 / v_pk_max_u16)
    const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    const u16x2 r = IS_MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y);
    return __builtin_bit_cast(unsigned, r);
}

// 3-input form: byte values 0..255 in u16 lanes are ordered identically as
// float16 bit patterns (zero and denormals), so v_pk_maximum3_f16 /
// v_pk_minimum3_f16 do two byte comparisons per lane pair in one instruction
template <bool IS_MAX>
__device__ __forceinline__ unsigned op3(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    if constexpr (IS_MAX) asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    else asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ unsigned align16(unsigned hi, unsigned lo)   // (lo >> 16) | (hi << 16)
{
    return __builtin_amdgcn_alignbit(hi, lo, 16);
}

// 16 voxels of one lane, even/odd split: e[k] = bytes {4k, 4k+2}, o[k] = bytes {4k+1, 4k+3}
struct Vec16 { unsigned e[4], o[4]; };

__device__ __forceinline__ void split(unsigned d, unsigned &e, unsigned &o)
{
    e = d & 0x00FF00FFu;
    o = (d >> 8) & 0x00FF00FFu;
}
__device__ __forceinline__ unsigned join(unsigned e, unsigned o) { return e | (o << 8); }

template <bool IS_MAX>
__device__ __forceinline__ Vec16 op16(const Vec16 &a, const Vec16 &b)
{
    Vec16 r;
#pragma unroll
    for (int k = 0; k < 4; k++) { r.e[k] = op2<IS_MAX>(a.e[k], b.e[k]); r.o[k] = op2<IS_MAX>(a.o[k], b.o[k]); }
    return r;
}

// Byte stream of 6 dwords in split form (dword 0 = left neighbour's last 4
// voxels, 1..4 = own 16 voxels, 5 = right neighbour's first 4 voxels).
struct Win { unsigned e[7], o[7]; };   // index 6 = padding (never reaches a used byte)

// T[i] = S[i + s] for s = 1, 2, 3, 4 (voxels)
template <int SH>
__device__ __forceinline__ Win shiftw(const Win &s)
{
    Win t;
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if constexpr (SH == 1) { t.e[k] = s.o[k]; t.o[k] = align16(s.e[k + 1], s.e[k]); }
        else if constexpr (SH == 2) { t.e[k] = align16(s.e[k + 1], s.e[k]); t.o[k] = align16(s.o[k + 1], s.o[k]); }
        else if constexpr (SH == 3) { t.e[k] = align16(s.o[k + 1], s.o[k]); t.o[k] = s.e[k + 1]; }
        else { t.e[k] = s.e[k + 1]; t.o[k] = s.o[k + 1]; }
    }
    t.e[6] = s.e[6]; t.o[6] = s.o[6];
    return t;
}

template <bool IS_MAX>
__device__ __forceinline__ Win opw(const Win &a, const Win &b)
{
    Win r;
#pragma unroll
    for (int k = 0; k < 7; k++) { r.e[k] = op2<IS_MAX>(a.e[k], b.e[k]); r.o[k] = op2<IS_MAX>(a.o[k], b.o[k]); }
    return r;
}

template <bool IS_MAX>
__device__ __forceinline__ Win op3w(const Win &a, const Win &b, const Win &c)
{
    Win r;
#pragma unroll
    for (int k = 0; k < 7; k++) { r.e[k] = op3<IS_MAX>(a.e[k], b.e[k], c.e[k]); r.o[k] = op3<IS_MAX>(a.o[k], b.o[k], c.o[k]); }
    return r;
}

// sliding min/max of width WX along x, centred: out[x] = op(b[x-r .. x+r]), r = WX/2 <= 4
template <int WX, bool IS_MAX>
__device__ __forceinline__ Vec16 xpass_u8(const Win &b)
{
    constexpr int RX = WX / 2;
    Win m;
    if constexpr (WX == 1) {
        m = b;
    } else if constexpr (WX == 3) {
        m = op3w<IS_MAX>(b, shiftw<1>(b), shiftw<2>(b));
    } else if constexpr (WX == 5 || WX == 7) {
        const Win m3 = op3w<IS_MAX>(b, shiftw<1>(b), shiftw<2>(b));            // covers i .. i+2
        if constexpr (WX == 5) m = opw<IS_MAX>(m3, shiftw<2>(m3));              // i .. i+4
        else m = op3w<IS_MAX>(m3, shiftw<2>(m3), shiftw<4>(m3));               // i .. i+6
    } else {
        const Win m2 = opw<IS_MAX>(b, shiftw<1>(b));
        const Win m4 = opw<IS_MAX>(m2, shiftw<2>(m2));
        if constexpr (WX == 5) m = opw<IS_MAX>(m4, shiftw<4>(b));
        else if constexpr (WX == 7) m = opw<IS_MAX>(m4, shiftw<3>(m4));
        else {   // 9
            const Win m8 = opw<IS_MAX>(m4, shiftw<4>(m4));
            Win b8;   // b shifted by 8 voxels = two dwords
#pragma unroll
            for (int k = 0; k < 7; k++) { b8.e[k] = b.e[k + 2 < 7 ? k + 2 : 6]; b8.o[k] = b.o[k + 2 < 7 ? k + 2 : 6]; }
            m = opw<IS_MAX>(m8, b8);
        }
    }
    // m[i] covers b[i .. i+WX-1]; output voxel j (window byte 4 + j) = m[4 - RX + j]
    Win s;
    if constexpr (RX == 4 || WX == 1) {
        // WX == 1: plain copy, own voxels start at byte 4 -> shift 4; RX == 4: shift 0
        if constexpr (WX == 1) s = shiftw<4>(m); else s = m;
    } else if constexpr (RX == 3) s = shiftw<1>(m);
    else if constexpr (RX == 2) s = shiftw<2>(m);
    else s = shiftw<3>(m);
    Vec16 r;
#pragma unroll
    for (int k = 0; k < 4; k++) { r.e[k] = s.e[k]; r.o[k] = s.o[k]; }
    return r;
}

struct U8StreamParams {
    int nx, ny, nz;
    int axis;            // streamed axis: 0 = z, 1 = y
    int oa, ma;          // offset (w/2 + origin) / mode along the streamed axis
    int mx;              // x boundary mode
    unsigned cval4;      // cval byte replicated to 4 bytes
    int chunk, nchunks, nxt;
    int swz;             // XCD-aware workgroup order (xcd_block())
};

__device__ __forceinline__ unsigned bswap32(unsigned x) { return __builtin_bswap32(x); }

// the 4 voxels outside the tile on `side`: byte offset inside the row to load a
// dword from, and the fix-up
__device__ __forceinline__ void edge_u8(int side, int x0, int xe, int nx, int mode, int *start, int *kind)
{
    if (side == 0) {
        if (x0 > 0) { *start = x0 - 4; *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = 0; *kind = EDGE_REV; break;
        case MI_MODE_MIRROR:    *start = 1; *kind = EDGE_REV; break;
        case MI_MODE_NEAREST:   *start = 0; *kind = EDGE_SPLAT; break;
        case MI_MODE_GRID_WRAP: *start = nx - 4; *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    } else {
        if (xe < nx) { *start = xe; *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = nx - 4; *kind = EDGE_REV; break;
        case MI_MODE_MIRROR:    *start = nx - 5; *kind = EDGE_REV; break;
        case MI_MODE_NEAREST:   *start = nx - 4; *kind = EDGE_SPLAT; break;
        case MI_MODE_GRID_WRAP: *start = 0; *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    }
}

template <int WX, int WA, bool IS_MAX>
__global__ void __launch_bounds__(256)
stream_minmax_u8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const U8StreamParams p)
{
    constexpr int RINGN = WA - 1;
    constexpr int DEPTH = WA == 11 ? 2 : 4;          // loads in flight per wave (11 samples: the unrolled ring would be 20 steps)
    constexpr int U = RINGN > 0 ? (RINGN % DEPTH == 0 ? RINGN : (RINGN % 2 == 0 && DEPTH == 4 ? 2 * RINGN : DEPTH * RINGN)) : DEPTH;
    static_assert(U % DEPTH == 0 && (RINGN == 0 || U % RINGN == 0), "unroll must cover ring and slots");

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
    const int x0 = xt * 1024;
    const int nlanes = min(64, (nx - x0) >> 4);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;               // bytes
    const unsigned strideA = p.axis == 0 ? plane : (unsigned)nx;
    const unsigned rowbase = p.axis == 0 ? (unsigned)oth * nx : (unsigned)oth * plane;
    const unsigned total_bytes = plane * (unsigned)nz;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? rowbase + (unsigned)(x0 + 16 * lane) : kOOB;

    // tile-edge dword for the x pass (lane 0: the 4 voxels left of the tile, lane `last`: right of it)
    unsigned evoff = kOOB;
    int ekind = EDGE_FWD;
    const int side = lane == 0 ? 0 : 1;
    if constexpr (WX > 1) {
        int st;
        edge_u8(side, x0, x0 + 16 * nlanes, nx, p.mx, &st, &ekind);
        if ((lane == 0 || lane == last) && ekind != EDGE_CONST) evoff = rowbase + (unsigned)st;
    }

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, nA);
    const int nsteps = a1 - a0 + WA - 1;
    const int ai0 = a0 - p.oa;

    struct Slot { u32x4 v; unsigned e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        const int ai = bmap<int>(ai0 + i, nA, p.ma);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * strideA;
        s.v = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0);
        if constexpr (WX > 1) s.e = __builtin_amdgcn_raw_buffer_load_b32(rin, s.cst ? kOOB : evoff, soff, 0);
        else s.e = 0;
    };

    Vec16 ring[RINGN > 0 ? RINGN : 1];

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
                unsigned ed = s.e;
                if (s.cst) { v.x = v.y = v.z = v.w = p.cval4; ed = p.cval4; }
                Vec16 xf;
                if constexpr (WX > 1) {
                    // edge dword fix-up for this lane's side
                    if (!s.cst) {
                        if (ekind == EDGE_REV) ed = bswap32(ed);
                        else if (ekind == EDGE_SPLAT) ed = (side == 0 ? (ed & 0xFFu) : (ed >> 24)) * 0x01010101u;
                        else if (ekind == EDGE_CONST) ed = p.cval4;
                    }
                    // neighbour dwords: left lane's last dword, right lane's first dword
                    unsigned l = (unsigned)__builtin_amdgcn_update_dpp((int)ed, (int)v.w, 0x138, 0xf, 0xf, false);
                    unsigned r = (unsigned)__builtin_amdgcn_update_dpp((int)ed, (int)v.x, 0x130, 0xf, 0xf, false);
                    if (lane == last) r = ed;
                    Win w;
                    split(l, w.e[0], w.o[0]);
                    split(v.x, w.e[1], w.o[1]);
                    split(v.y, w.e[2], w.o[2]);
                    split(v.z, w.e[3], w.o[3]);
                    split(v.w, w.e[4], w.o[4]);
                    split(r, w.e[5], w.o[5]);
                    w.e[6] = w.e[5]; w.o[6] = w.o[5];
                    xf = xpass_u8<WX, IS_MAX>(w);
                } else {
                    split(v.x, xf.e[0], xf.o[0]);
                    split(v.y, xf.e[1], xf.o[1]);
                    split(v.z, xf.e[2], xf.o[2]);
                    split(v.w, xf.e[3], xf.o[3]);
                }
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                if (i >= WA - 1) {
                    Vec16 a = xf;
                    if constexpr (RINGN > 0) {
#pragma unroll
                        for (int k = 0; k < RINGN; k++) a = op16<IS_MAX>(a, ring[k]);
                    }
                    u32x4 u;
                    u.x = join(a.e[0], a.o[0]); u.y = join(a.e[1], a.o[1]);
                    u.z = join(a.e[2], a.o[2]); u.w = join(a.e[3], a.o[3]);
                    const unsigned so = (unsigned)(a0 + i - (WA - 1)) * strideA;
                    buffer_store_b128_soff(u, rout, voff, so);
                }
                if constexpr (RINGN > 0) ring[J % RINGN] = xf;
            }
        });
    }
}

// ---------------------------------------------------------------------------
// Flat footprints whose rows are centred runs (disk, diamond / cross, octagon, square: what skimage's morphology
// passes) on uint8 images and slice-wise on volumes, ONE streaming launch:
//     out(y, x) = op over rows dy of [ sliding op of width 2 hw[dy] + 1 along x of row y + dy ]
// The wave streams down the image with the previous WA - 1 raw rows in registers as split windows (the six-dword
// window xpass_u8 works on); per output row every footprint row contributes one x window of its own width.  Replaces
// the LDS-tiled footprint kernel for these shapes (a 2.5-D tile kernel has no pipeline on a one-plane volume:
// 8192^2 disk(1) ran 139 us, disk(3) 263 us).  Bit-exact (comparisons only).
// ---------------------------------------------------------------------------
struct U8RunParams {
    int nx, ny, nz;
    int mx, my;
    unsigned cval4;
    int chunk, nchunks, nxt;
    int swz;
    int hw[9];           // half width of the run of footprint row r (0 .. WA-1), -1 = empty row
};

template <bool IS_MAX>
__device__ __forceinline__ Vec16 xrun_u8(const Win &w, int hw)
{
    switch (hw) {           // wave-uniform
    case 0: return xpass_u8<1, IS_MAX>(w);
    case 1: return xpass_u8<3, IS_MAX>(w);
    case 2: return xpass_u8<5, IS_MAX>(w);
    case 3: return xpass_u8<7, IS_MAX>(w);
    default: return xpass_u8<9, IS_MAX>(w);
    }
}

template <int WA, bool IS_MAX>
__global__ void __launch_bounds__(256)
runs_minmax_u8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const U8RunParams p)
{
    constexpr int DEPTH = 2;             // 4 loads in flight measured slower here (longer unrolled ring, more registers)
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
    const int x0 = xt * 1024;
    const int nlanes = min(64, (nx - x0) >> 4);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned rowbase = (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? rowbase + (unsigned)(x0 + 16 * lane) : kOOB;
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_u8(side, x0, x0 + 16 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? rowbase + (unsigned)est : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, ny);
    const int nsteps = a1 - a0 + WA - 1;
    const int ai0 = a0 - WA / 2;

    struct Slot { u32x4 v; unsigned e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)ny) ai = bmap<int>(ai, ny, p.my);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * (unsigned)nx;
        s.v = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0);
        s.e = __builtin_amdgcn_raw_buffer_load_b32(rin, s.cst ? kOOB : evoff, soff, 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    Win ring[RINGN > 0 ? RINGN : 1];        // raw rows as split windows, ring[(J + k) % RINGN] = footprint row k at unrolled step J
    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J % DEPTH];
                u32x4 v = s.v;
                unsigned ed = s.e;
                if (s.cst) { v.x = v.y = v.z = v.w = p.cval4; ed = p.cval4; }
                else {
                    if (ekind == EDGE_REV) ed = bswap32(ed);
                    else if (ekind == EDGE_SPLAT) ed = (side == 0 ? (ed & 0xFFu) : (ed >> 24)) * 0x01010101u;
                    else if (ekind == EDGE_CONST) ed = p.cval4;
                }
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                const unsigned l = (unsigned)__builtin_amdgcn_update_dpp((int)ed, (int)v.w, 0x138, 0xf, 0xf, false);
                unsigned r = (unsigned)__builtin_amdgcn_update_dpp((int)ed, (int)v.x, 0x130, 0xf, 0xf, false);
                if (lane == last) r = ed;
                Win w;
                split(l, w.e[0], w.o[0]);
                split(v.x, w.e[1], w.o[1]);
                split(v.y, w.e[2], w.o[2]);
                split(v.z, w.e[3], w.o[3]);
                split(v.w, w.e[4], w.o[4]);
                split(r, w.e[5], w.o[5]);
                w.e[6] = w.e[5]; w.o[6] = w.o[5];
                if (i >= WA - 1) {
                    // footprint row WA - 1 is the row just loaded, row k < WA - 1 sits in ring[(J + k) % RINGN].
                    // min / max commute: rows that share a half width are combined first (14 operations per row) and
                    // ONE x window per distinct half width follows -- disk(3) needs three windows, not seven
                    Vec16 a;
                    bool have = false;
                    static_for<5>([&](auto HH) {
                        constexpr int h = decltype(HH)::value;
                        Win g;
                        bool any = false;
                        static_for<WA>([&](auto KK) {
                            constexpr int k = decltype(KK)::value;
                            if (p.hw[k] == h) {
                                if constexpr (k == WA - 1) g = any ? opw<IS_MAX>(g, w) : w;
                                else g = any ? opw<IS_MAX>(g, ring[(J + k) % (RINGN > 0 ? RINGN : 1)]) : ring[(J + k) % (RINGN > 0 ? RINGN : 1)];
                                any = true;
                            }
                        });
                        if (any) {
                            const Vec16 t = xpass_u8<2 * h + 1, IS_MAX>(g);
                            a = have ? op16<IS_MAX>(a, t) : t;
                            have = true;
                        }
                    });
                    u32x4 u;
                    u.x = join(a.e[0], a.o[0]); u.y = join(a.e[1], a.o[1]);
                    u.z = join(a.e[2], a.o[2]); u.w = join(a.e[3], a.o[3]);
                    const unsigned so = (unsigned)(a0 + i - (WA - 1)) * (unsigned)nx;
                    buffer_store_b128_soff(u, rout, voff, so);
                }
                if constexpr (RINGN > 0) ring[J % RINGN] = w;
            }
        });
    }
}

template <int WA, bool IS_MAX>
static int launch_runs_u8(const uint8_t *in, uint8_t *out, U8RunParams &p, hipStream_t s)
{
    const int nlines = p.nz * p.nxt;
    int nch = 1;
    {
        double best = 1e300;
        for (int c = 1; c <= p.ny && c <= 2048; c++) {
            const int chunk = (p.ny + c - 1) / c;
            if (c > 1 && chunk < 8) break;
            const int real = (p.ny + chunk - 1) / chunk;
            const double rounds = std::max(1.0, (double)nlines * real / 4096.0);
            const double cost = rounds * (chunk + (WA - 1) + 4.0);
            if (cost < best * 0.999) { best = cost; nch = real; }
        }
    }
    p.chunk = (p.ny + nch - 1) / nch;
    p.nchunks = (p.ny + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz);
    hipLaunchKernelGGL((runs_minmax_u8_kernel<WA, IS_MAX>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------
// 3 x 3 x 3 footprints whose x rows are centred runs -- generate_binary_structure(3, 1 / 2 / 3), i.e. the 6- / 18- /
// 26-connected structures of grey and binary morphology on volumes -- on uint8 / bool volumes, ONE streaming launch:
// a wave owns one (y, 1024-voxel x segment) line and streams along z; per plane it loads the rows y-1, y, y+1 (the
// neighbouring lines load them too: L1 / L2 hits, HBM sees every row about once), keeps the rows of the two previous
// planes in registers as split windows, and takes  op over (dz, dy) of [x window of half width hw(dz, dy)].
// hw: -1 = no sample in that row, 0 = the centre voxel, 1 = three voxels.
// Replaces the LDS-tiled footprint kernel for these structures on uint8 volumes (512^3, 6-connected: 328 -> 186 us;
// 512-wide rows fill only half of the 1024-voxel wave segment).  Binary erosion / dilation of bool volumes was tried on
// top of it and is NOT routed here: 199 us against 143 us for the byte-parallel tiled binary kernel.  Bit-exact.
// ---------------------------------------------------------------------------
struct U8Run3Params {
    int nx, ny, nz;
    int mx, my, mz;
    unsigned cval4;
    int chunk, nchunks, nxt;
    int swz;
    int hw[3][3];        // [dz + 1][dy + 1]
};

template <bool IS_MAX>
__device__ __forceinline__ Vec16 xrun3_u8(const Win &w, int hw)
{
    return hw == 0 ? xpass_u8<1, IS_MAX>(w) : xpass_u8<3, IS_MAX>(w);       // wave-uniform
}

template <bool IS_MAX>
__global__ void __launch_bounds__(256)
runs3d_minmax_u8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const U8Run3Params p)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nlines = ny * p.nxt;
    const int wid = xcd_block((int)blockIdx.x, (int)gridDim.x, p.swz) * 4 + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int y = line / p.nxt, xt = line - y * p.nxt;
    const int x0 = xt * 1024;
    const int nlanes = min(64, (nx - x0) >> 4);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned total_bytes = plane * (unsigned)nz;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_u8(side, x0, x0 + 16 * nlanes, nx, p.mx, &est, &ekind);
    const bool edge_lane = (lane == 0 || lane == last) && ekind != EDGE_CONST;
    // the three rows of a plane this wave reads: y - 1, y, y + 1 after the y boundary map (-1: a row of cval)
    int ys[3];
    unsigned voff[3], evoff[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        ys[r] = bmap<int>(y - 1 + r, ny, p.my);
        const unsigned rb = (unsigned)max(ys[r], 0) * (unsigned)nx;
        voff[r] = (lane < nlanes && ys[r] >= 0) ? rb + (unsigned)(x0 + 16 * lane) : kOOB;
        evoff[r] = (edge_lane && ys[r] >= 0) ? rb + (unsigned)est : kOOB;
    }
    const unsigned ovoff = lane < nlanes ? (unsigned)y * (unsigned)nx + (unsigned)(x0 + 16 * lane) : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, nz);
    const int nsteps = a1 - a0 + 2;
    const int ai0 = a0 - 1;

    struct Slot { u32x4 v[3]; unsigned e[3]; bool cst; };
    Slot S[2];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)nz) ai = bmap<int>(ai, nz, p.mz);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * plane;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            s.v[r] = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff[r], soff, 0);
            s.e[r] = __builtin_amdgcn_raw_buffer_load_b32(rin, s.cst ? kOOB : evoff[r], soff, 0);
        }
    };
    issue(0, S[0]);
    if (1 < nsteps) issue(1, S[1]);

    Win ring[2][3];                  // the rows of the two previous planes; ring[(J + k) % 2] = plane dz = k - 1 at step J
    for (int i0 = 0; i0 < nsteps; i0 += 2) {
        static_for<2>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J];
                Win w[3];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    u32x4 v = s.v[r];
                    unsigned ed = s.e[r];
                    if (s.cst || ys[r] < 0) { v.x = v.y = v.z = v.w = p.cval4; ed = p.cval4; }
                    else {
                        if (ekind == EDGE_REV) ed = bswap32(ed);
                        else if (ekind == EDGE_SPLAT) ed = (side == 0 ? (ed & 0xFFu) : (ed >> 24)) * 0x01010101u;
                        else if (ekind == EDGE_CONST) ed = p.cval4;
                    }
                    const unsigned l = (unsigned)__builtin_amdgcn_update_dpp((int)ed, (int)v.w, 0x138, 0xf, 0xf, false);
                    unsigned rr = (unsigned)__builtin_amdgcn_update_dpp((int)ed, (int)v.x, 0x130, 0xf, 0xf, false);
                    if (lane == last) rr = ed;
                    split(l, w[r].e[0], w[r].o[0]);
                    split(v.x, w[r].e[1], w[r].o[1]);
                    split(v.y, w[r].e[2], w[r].o[2]);
                    split(v.z, w[r].e[3], w[r].o[3]);
                    split(v.w, w[r].e[4], w[r].o[4]);
                    split(rr, w[r].e[5], w[r].o[5]);
                    w[r].e[6] = w[r].e[5]; w[r].o[6] = w[r].o[5];
                }
                if (i + 2 < nsteps) issue(i + 2, s);
                if (i >= 2) {
                    Vec16 a;
                    bool have = false;
                    static_for<9>([&](auto KK) {
                        constexpr int k = decltype(KK)::value;
                        constexpr int dz = k / 3, dy = k % 3;
                        const int hw = p.hw[dz][dy];
                        if (hw >= 0) {
                            Vec16 t;
                            if constexpr (dz == 2) t = xrun3_u8<IS_MAX>(w[dy], hw);
                            else t = xrun3_u8<IS_MAX>(ring[(J + dz) % 2][dy], hw);
                            a = have ? op16<IS_MAX>(a, t) : t;
                            have = true;
                        }
                    });
                    u32x4 u;
                    u.x = join(a.e[0], a.o[0]); u.y = join(a.e[1], a.o[1]);
                    u.z = join(a.e[2], a.o[2]); u.w = join(a.e[3], a.o[3]);
                    const unsigned so = (unsigned)(a0 + i - 2) * plane;
                    buffer_store_b128_soff(u, rout, ovoff, so);
                }
#pragma unroll
                for (int r = 0; r < 3; r++) ring[J][r] = w[r];
            }
        });
    }
}

// ---------------------------------------------------------------------------
// uniform_filter on uint8 images (volumes: slice by slice) with a uint8 result, ONE streaming launch in integer
// arithmetic.  SciPy filters axis by axis and stores every intermediate in the OUTPUT dtype
// (cupyimg/scipy/ndimage/filters.py:602-665: `uniform_filter1d(input, ..., output); input = output`), i.e. for uint8 in
// and out:  q1 = trunc(sum of the wy rows / wy)  as uint8, then  out = trunc(sum of wx columns of q1 / wx).  The sums are
// small integers (<= 9 x 255), exact in u16 lanes; trunc(S / w) is taken as  (unsigned)(S * fl(1 / w) + 0.001): the
// product is within 1e-4 of S / w and a non-integral quotient is at least 1 / 9 away from the next integer.  The wave
// streams down the image with a running row sum (ring of the last wy raw rows, split byte form), divides, and runs the
// x window on the quotient row.  Replaces two generic launches with a double accumulate per sample (8192^2: 370 us).
// ---------------------------------------------------------------------------
struct U8BoxParams {
    int nx, ny, nz;
    int axis;            // streamed axis: 1 = y (x window fused), 0 = z (WX == 1: the z pass of a volume)
    int oy;              // w / 2 + origin along the streamed axis
    int mx, my;
    unsigned cval4;
    int chunk, nchunks, nxt;
    int swz;
    float ry, rx;        // fl(1 / wy), fl(1 / wx)
};

__device__ __forceinline__ unsigned div_pk(unsigned s, float r)
{
    const unsigned ql = (unsigned)fmaf((float)(s & 0xFFFFu), r, 0.001f), qh = (unsigned)fmaf((float)(s >> 16), r, 0.001f);
    return ql | (qh << 16);
}

__device__ __forceinline__ Win addw(const Win &a, const Win &b)
{
    Win r;
#pragma unroll
    for (int k = 0; k < 7; k++) { r.e[k] = a.e[k] + b.e[k]; r.o[k] = a.o[k] + b.o[k]; }      // u16 lanes, no carry (sums < 65536)
    return r;
}

// sum of the WX voxels centred on each of the lane's 16 voxels; b = [left dword | own 4 dwords | right dword], split
template <int WX>
__device__ __forceinline__ Vec16 xsum_u8(const Win &b)
{
    constexpr int RX = WX / 2;
    Win m;                                   // m[i] = b[i] + ... + b[i + WX - 1]
    if constexpr (WX == 1) {
        m = b;
    } else {
        const Win m2 = addw(b, shiftw<1>(b));
        if constexpr (WX == 3) {
            m = addw(m2, shiftw<2>(b));
        } else {
            const Win m4 = addw(m2, shiftw<2>(m2));
            if constexpr (WX == 5) m = addw(m4, shiftw<4>(b));
            else if constexpr (WX == 7) m = addw(addw(m4, shiftw<4>(m2)), shiftw<2>(shiftw<4>(b)));
            else {
                Win b8;
#pragma unroll
                for (int k = 0; k < 7; k++) { b8.e[k] = b.e[k + 2 < 7 ? k + 2 : 6]; b8.o[k] = b.o[k + 2 < 7 ? k + 2 : 6]; }
                m = addw(addw(m4, shiftw<4>(m4)), b8);
            }
        }
    }
    Win s;
    if constexpr (RX == 4) s = m;
    else if constexpr (RX == 3) s = shiftw<1>(m);
    else if constexpr (RX == 2) s = shiftw<2>(m);
    else if constexpr (RX == 1) s = shiftw<3>(m);
    else s = shiftw<4>(m);
    Vec16 r;
#pragma unroll
    for (int k = 0; k < 4; k++) { r.e[k] = s.e[k]; r.o[k] = s.o[k]; }
    return r;
}

template <int WX, int WY>
__global__ void __launch_bounds__(256)
box2d_u8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const U8BoxParams p)
{
    constexpr int DEPTH = 2;
    constexpr int U = WY % DEPTH == 0 ? WY : WY * DEPTH;       // ring of WY rows, DEPTH slots
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nother = p.axis == 0 ? ny : nz;             // lines: (other axis, x segment)
    const int nA = p.axis == 0 ? nz : ny;                 // extent of the streamed axis
    const int nlines = nother * p.nxt;
    const int wid = xcd_block((int)blockIdx.x, (int)gridDim.x, p.swz) * 4 + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int z = line / p.nxt, xt = line - z * p.nxt;    // z: index along the other axis
    const int x0 = xt * 1024;
    const int nlanes = min(64, (nx - x0) >> 4);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned strideA = p.axis == 0 ? plane : (unsigned)nx;
    const unsigned rowbase = p.axis == 0 ? (unsigned)z * (unsigned)nx : (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? rowbase + (unsigned)(x0 + 16 * lane) : kOOB;
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_u8(side, x0, x0 + 16 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? rowbase + (unsigned)est : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, nA);
    const int nsteps = a1 - a0 + WY - 1;
    const int ai0 = a0 - p.oy;

    struct Slot { u32x4 v; unsigned e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)nA) ai = bmap<int>(ai, nA, p.my);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * strideA;
        s.v = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0);
        s.e = __builtin_amdgcn_raw_buffer_load_b32(rin, s.cst ? kOOB : evoff, soff, 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    // rows in split form: e[0..3] / o[0..3] own voxels, e[4] / o[4] the edge dword
    struct Row { unsigned e[5], o[5]; };
    Row ring[WY];
    Row sum;
#pragma unroll
    for (int k = 0; k < 5; k++) { sum.e[k] = 0u; sum.o[k] = 0u; }
#pragma unroll
    for (int r = 0; r < WY; r++)
#pragma unroll
        for (int k = 0; k < 5; k++) { ring[r].e[k] = 0u; ring[r].o[k] = 0u; }

    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J % DEPTH];
                u32x4 v = s.v;
                unsigned ed = s.e;
                if (s.cst) { v.x = v.y = v.z = v.w = p.cval4; ed = p.cval4; }
                else {
                    if (ekind == EDGE_REV) ed = bswap32(ed);
                    else if (ekind == EDGE_SPLAT) ed = (side == 0 ? (ed & 0xFFu) : (ed >> 24)) * 0x01010101u;
                    else if (ekind == EDGE_CONST) ed = p.cval4;
                }
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                Row cur;
                split(v.x, cur.e[0], cur.o[0]); split(v.y, cur.e[1], cur.o[1]);
                split(v.z, cur.e[2], cur.o[2]); split(v.w, cur.e[3], cur.o[3]);
                split(ed, cur.e[4], cur.o[4]);
                // running sum of the last WY rows: ring[J % WY] holds the row that leaves the window
                Row &old = ring[J % WY];
#pragma unroll
                for (int k = 0; k < 5; k++) {
                    sum.e[k] += cur.e[k] - old.e[k];
                    sum.o[k] += cur.o[k] - old.o[k];
                }
                old = cur;
                if (i >= WY - 1) {
                    Row q;
#pragma unroll
                    for (int k = 0; k < 5; k++) {
                        if constexpr (WY == 1) { q.e[k] = sum.e[k]; q.o[k] = sum.o[k]; }
                        else { q.e[k] = div_pk(sum.e[k], p.ry); q.o[k] = div_pk(sum.o[k], p.ry); }
                    }
                    Vec16 a;
                    if constexpr (WX == 1) {
#pragma unroll
                        for (int k = 0; k < 4; k++) { a.e[k] = q.e[k]; a.o[k] = q.o[k]; }
                    } else {
                        Win w;
                        // neighbour dwords of the quotient row (split): the left lane's last dword, the right lane's first
                        w.e[0] = (unsigned)__builtin_amdgcn_update_dpp((int)q.e[4], (int)q.e[3], 0x138, 0xf, 0xf, false);
                        w.o[0] = (unsigned)__builtin_amdgcn_update_dpp((int)q.o[4], (int)q.o[3], 0x138, 0xf, 0xf, false);
                        unsigned re = (unsigned)__builtin_amdgcn_update_dpp((int)q.e[4], (int)q.e[0], 0x130, 0xf, 0xf, false);
                        unsigned ro = (unsigned)__builtin_amdgcn_update_dpp((int)q.o[4], (int)q.o[0], 0x130, 0xf, 0xf, false);
                        if (lane == last) { re = q.e[4]; ro = q.o[4]; }
#pragma unroll
                        for (int k = 0; k < 4; k++) { w.e[k + 1] = q.e[k]; w.o[k + 1] = q.o[k]; }
                        w.e[5] = re; w.o[5] = ro;
                        w.e[6] = re; w.o[6] = ro;
                        const Vec16 sx = xsum_u8<WX>(w);
#pragma unroll
                        for (int k = 0; k < 4; k++) { a.e[k] = div_pk(sx.e[k], p.rx); a.o[k] = div_pk(sx.o[k], p.rx); }
                    }
                    u32x4 u;
                    u.x = join(a.e[0], a.o[0]); u.y = join(a.e[1], a.o[1]);
                    u.z = join(a.e[2], a.o[2]); u.w = join(a.e[3], a.o[3]);
                    const unsigned so = (unsigned)(a0 + i - (WY - 1)) * strideA;
                    buffer_store_b128_soff(u, rout, voff, so);
                }
            }
        });
    }
}

template <int WX, int WY>
static int launch_box2d_u8(const uint8_t *in, uint8_t *out, U8BoxParams &p, hipStream_t s)
{
    const int nlines = (p.axis == 0 ? p.ny : p.nz) * p.nxt;
    const int nA = p.axis == 0 ? p.nz : p.ny;
    int nch = 1;
    {
        double best = 1e300;
        for (int c = 1; c <= nA && c <= 2048; c++) {
            const int chunk = (nA + c - 1) / c;
            if (c > 1 && chunk < 8) break;
            const int real = (nA + chunk - 1) / chunk;
            const double rounds = std::max(1.0, (double)nlines * real / 4096.0);
            const double cost = rounds * (chunk + (WY - 1) + 4.0);
            if (cost < best * 0.999) { best = cost; nch = real; }
        }
    }
    p.chunk = (nA + nch - 1) / nch;
    p.nchunks = (nA + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz);
    hipLaunchKernelGGL((box2d_u8_kernel<WX, WY>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <int WX>
static int launch_box2d_u8_wy(int wy, const uint8_t *in, uint8_t *out, U8BoxParams &p, hipStream_t s)
{
    switch (wy) {
    case 1: return launch_box2d_u8<WX, 1>(in, out, p, s);
    case 3: return launch_box2d_u8<WX, 3>(in, out, p, s);
    case 5: return launch_box2d_u8<WX, 5>(in, out, p, s);
    case 7: return launch_box2d_u8<WX, 7>(in, out, p, s);
    default: return launch_box2d_u8<WX, 9>(in, out, p, s);
    }
}

// ---------------------------------------------------------------------------
// 3 x 3 median of uint8 images, one streaming launch (entry point: mi_median3x3, median2d.hip; the float32 kernel and
// the method are described there).  16 pixels per lane in even/odd split form; med3 = max(min(a,b), min(max(a,b),c)).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned med3u(unsigned a, unsigned b, unsigned c)
{
    const unsigned mn = op2<false>(a, b), mx = op2<true>(a, b);
    return op2<true>(mn, op2<false>(mx, c));
}
__device__ __forceinline__ void sort3u(unsigned a, unsigned b, unsigned c, unsigned &lo, unsigned &mid, unsigned &hi)
{
    lo = op3<false>(a, b, c);
    hi = op3<true>(a, b, c);
    mid = med3u(a, b, c);
}
// T[i] = S[i - 1] (prev_o3 = the O register that ends just left of the lane) / T[i] = S[i + 1] (next_e0 likewise)
__device__ __forceinline__ Vec16 shl1(const Vec16 &s, unsigned prev_o3)
{
    Vec16 t;
#pragma unroll
    for (int k = 0; k < 4; k++) { t.e[k] = align16(s.o[k], k ? s.o[k - 1] : prev_o3); t.o[k] = s.e[k]; }
    return t;
}
__device__ __forceinline__ Vec16 shr1(const Vec16 &s, unsigned next_e0)
{
    Vec16 t;
#pragma unroll
    for (int k = 0; k < 4; k++) { t.e[k] = s.o[k]; t.o[k] = align16(k < 3 ? s.e[k + 1] : next_e0, s.e[k]); }
    return t;
}

__global__ void __launch_bounds__(256)
median3x3_u8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const U8StreamParams p)
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
    const int x0 = xt * 1024;
    const int nlanes = min(64, (nx - x0) >> 4);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned rowbase = (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? rowbase + (unsigned)(x0 + 16 * lane) : kOOB;
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_u8(side, x0, x0 + 16 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? rowbase + (unsigned)est : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, ny);
    const int nsteps = a1 - a0 + 2;
    const int ai0 = a0 - 1;

    struct Slot { u32x4 v; unsigned e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        const int ai = bmap<int>(ai0 + i, ny, p.ma);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * (unsigned)nx;
        s.v = __builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0);
        s.e = __builtin_amdgcn_raw_buffer_load_b32(rin, s.cst ? kOOB : evoff, soff, 0);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    Vec16 rv[2];
    unsigned ree[2], reo[2];            // edge dword of the two previous rows, split
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int k = 0; k < 4; k++) rv[h].e[k] = rv[h].o[k] = 0;
        ree[h] = reo[h] = 0;
    }
    for (int i0 = 0; i0 < nsteps; i0 += DEPTH) {
        static_for<DEPTH>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J];
                u32x4 v = s.v;
                unsigned ed = s.e;
                if (s.cst) { v.x = v.y = v.z = v.w = p.cval4; ed = p.cval4; }
                else {
                    if (ekind == EDGE_REV) ed = bswap32(ed);
                    else if (ekind == EDGE_SPLAT) ed = (side == 0 ? (ed & 0xFFu) : (ed >> 24)) * 0x01010101u;
                    else if (ekind == EDGE_CONST) ed = p.cval4;
                }
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                Vec16 cur;
                split(v.x, cur.e[0], cur.o[0]); split(v.y, cur.e[1], cur.o[1]);
                split(v.z, cur.e[2], cur.o[2]); split(v.w, cur.e[3], cur.o[3]);
                unsigned ce, co;
                split(ed, ce, co);
                if (i >= 2) {
                    Vec16 lo, mid, hi;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        sort3u(rv[0].e[k], rv[1].e[k], cur.e[k], lo.e[k], mid.e[k], hi.e[k]);
                        sort3u(rv[0].o[k], rv[1].o[k], cur.o[k], lo.o[k], mid.o[k], hi.o[k]);
                    }
                    // edge column, sorted: the left neighbour is byte 3 of the edge dword (high half of O), the right
                    // neighbour byte 0 (low half of E)
                    unsigned elo_e, emid_e, ehi_e, elo_o, emid_o, ehi_o;
                    sort3u(ree[0], ree[1], ce, elo_e, emid_e, ehi_e);
                    sort3u(reo[0], reo[1], co, elo_o, emid_o, ehi_o);
                    auto from_left = [&](unsigned keep, unsigned x) {
                        return (unsigned)__builtin_amdgcn_update_dpp((int)keep, (int)x, 0x138, 0xf, 0xf, false);
                    };
                    auto from_right = [&](unsigned keep, unsigned x) {
                        const unsigned r = (unsigned)__builtin_amdgcn_update_dpp((int)keep, (int)x, 0x130, 0xf, 0xf, false);
                        return lane == last ? keep : r;
                    };
                    const Vec16 lo_l = shl1(lo, from_left(elo_o, lo.o[3])), lo_r = shr1(lo, from_right(elo_e, lo.e[0]));
                    const Vec16 mid_l = shl1(mid, from_left(emid_o, mid.o[3])), mid_r = shr1(mid, from_right(emid_e, mid.e[0]));
                    const Vec16 hi_l = shl1(hi, from_left(ehi_o, hi.o[3])), hi_r = shr1(hi, from_right(ehi_e, hi.e[0]));
                    u32x4 u;
                    unsigned oe[4], oo[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        oe[k] = med3u(op3<true>(lo_l.e[k], lo.e[k], lo_r.e[k]), med3u(mid_l.e[k], mid.e[k], mid_r.e[k]),
                                      op3<false>(hi_l.e[k], hi.e[k], hi_r.e[k]));
                        oo[k] = med3u(op3<true>(lo_l.o[k], lo.o[k], lo_r.o[k]), med3u(mid_l.o[k], mid.o[k], mid_r.o[k]),
                                      op3<false>(hi_l.o[k], hi.o[k], hi_r.o[k]));
                    }
                    u.x = join(oe[0], oo[0]); u.y = join(oe[1], oo[1]); u.z = join(oe[2], oo[2]); u.w = join(oe[3], oo[3]);
                    const unsigned so = (unsigned)(a0 + i - 2) * (unsigned)nx;
                    buffer_store_b128_soff(u, rout, voff, so);
                }
                rv[J % 2] = cur;        // overwrites the older of the two rows (the sorts are symmetric in their inputs)
                ree[J % 2] = ce;
                reo[J % 2] = co;
            }
        });
    }
}

int run_median3x3_u8(const uint8_t *in, uint8_t *out, int nz, int ny, int nx, int mx, int my, int cval, hipStream_t s)
{
    U8StreamParams p;
    memset(&p, 0, sizeof(p));
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.axis = 1; p.oa = 1; p.ma = my; p.mx = mx;
    p.cval4 = (unsigned)cval * 0x01010101u;
    p.nxt = (nx + 1023) / 1024;
    const int nlines = nz * p.nxt;
    int nch = (4096 + nlines - 1) / nlines;
    if (nch > ny / 8) nch = ny / 8;
    if (nch < 1) nch = 1;
    p.chunk = (ny + nch - 1) / nch;
    p.nchunks = (ny + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz);
    hipLaunchKernelGGL(median3x3_u8_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <int WX, int WA, bool IS_MAX>
static int launch_u8(const uint8_t *in, uint8_t *out, U8StreamParams &p, hipStream_t s)
{
    const int nA = p.axis == 0 ? p.nz : p.ny;
    const int nother = p.axis == 0 ? p.ny : p.nz;
    const int nlines = nother * p.nxt;
    // chunks along the streamed axis: time ~ rounds x (chunk + ramp) with 256 CUs x 16 resident waves; volumes have
    // many lines and get long chunks, images (a handful of 1024-pixel columns) many short ones
    int nch = 1;
    {
        double best = 1e300;
        for (int c = 1; c <= nA && c <= 2048; c++) {
            const int chunk = (nA + c - 1) / c;
            if (c > 1 && chunk < 8) break;
            const int real = (nA + chunk - 1) / chunk;
            const double rounds = std::max(1.0, (double)nlines * real / 4096.0);
            const double cost = rounds * (chunk + (WA - 1) + 4.0);
            if (cost < best * 0.999) { best = cost; nch = real; }
        }
    }
    p.chunk = (nA + nch - 1) / nch;
    p.nchunks = (nA + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz);
    hipLaunchKernelGGL((stream_minmax_u8_kernel<WX, WA, IS_MAX>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <int WX, bool IS_MAX>
static int launch_u8_wa(int wa, const uint8_t *in, uint8_t *out, U8StreamParams &p, hipStream_t s)
{
    switch (wa) {
    case 1: return launch_u8<WX, 1, IS_MAX>(in, out, p, s);
    case 3: return launch_u8<WX, 3, IS_MAX>(in, out, p, s);
    case 5: return launch_u8<WX, 5, IS_MAX>(in, out, p, s);
    case 7: return launch_u8<WX, 7, IS_MAX>(in, out, p, s);
    case 9: return launch_u8<WX, 9, IS_MAX>(in, out, p, s);
    case 11: return launch_u8<WX, 11, IS_MAX>(in, out, p, s);
    case 13: return launch_u8<WX, 13, IS_MAX>(in, out, p, s);
    }
    return MI_ERR_UNSUPPORTED;
}

template <bool IS_MAX>
static int launch_u8_wx(int wx, int wa, const uint8_t *in, uint8_t *out, U8StreamParams &p, hipStream_t s)
{
    switch (wx) {
    case 1: return launch_u8_wa<1, IS_MAX>(wa, in, out, p, s);
    case 3: return launch_u8_wa<3, IS_MAX>(wa, in, out, p, s);
    case 5: return launch_u8_wa<5, IS_MAX>(wa, in, out, p, s);
    case 7: return launch_u8_wa<7, IS_MAX>(wa, in, out, p, s);
    case 9: return launch_u8_wa<9, IS_MAX>(wa, in, out, p, s);
    }
    return MI_ERR_UNSUPPORTED;
}


// ---------------------------------------------------------------------------
// Single-launch version for cubic sizes 3 / 5 / 7 (grey_erosion(size=7), ...): the 2.5-D producer / consumer structure
// of sep3d_lean_kernel on bytes.  Windows are built from 3-input steps, v_pk_maximum3_f16 / v_pk_minimum3_f16: byte
// values 0..255 in u16 lanes are ordered identically as float16 bit patterns (zero and denormals), so one instruction
// does two comparisons per lane pair.  HBM traffic: 2 B/voxel (the two-launch version above moves 4).
// ---------------------------------------------------------------------------
struct U8FusedParams {
    int nx, ny, nz;
    int mx, my, mz;
    unsigned cval4;
    int zc, nzc, nxt, nyt;
    int zb, zn;             // output planes to produce: [zb, zb + zn) (0 / 0 = the whole volume); boundary handling refers to nz
    int nt;                 // 1 = rows no other workgroup reads are loaded non-temporally (long_common.hpp, stream_nt_for)
};

constexpr int kU8MaxChunk = 2048;

// ---------------------------------------------------------------------------
// Single-launch min / max for cubic sizes 3 / 5 / 7: producer / consumer tile with the voxels kept "even/odd split"
// (two u16 per register) from the load to the store.  The round-1 kernel (mm3u8_fused_kernel, in the history) kept its
// z history and its LDS tile PACKED (bytes) to save registers, and paid for it: every step re-split the history and
// the tile and re-joined the results, its x window worked on seven-dword windows, its y window reduced seven rows per
// output -- 0.29 VALU instructions per voxel where the three 7-wide windows themselves need about 0.08.  Here a lane holds 8 voxels (512-voxel tiles), so the split z history of a
// row is 24 registers and three rows per producer wave fit; the windows are built in two 3-input stages that share
// their first stage between neighbouring outputs:
//     t[i] = op3(v[i], v[i+1], v[i+2]);   out[p] = op3(t[p-3], t[p-1], t[p+1])   (W = 7)
// along x (shifts by v_alignbit on the split registers), along z (register history) and along y (consumers, rows in
// LDS in split form: 16 bytes per lane and row = (E0, O0, E1, O1)); bytes are joined once, at the store.
// ---------------------------------------------------------------------------
struct Split8 { unsigned e0, o0, e1, o1; };      // 8 voxels: dword 0 = (e0, o0), dword 1 = (e1, o1)

template <bool IS_MAX>
__device__ __forceinline__ Split8 op3s(const Split8 &a, const Split8 &b, const Split8 &c)
{
    return {op3<IS_MAX>(a.e0, b.e0, c.e0), op3<IS_MAX>(a.o0, b.o0, c.o0), op3<IS_MAX>(a.e1, b.e1, c.e1), op3<IS_MAX>(a.o1, b.o1, c.o1)};
}
template <bool IS_MAX>
__device__ __forceinline__ Split8 op2s(const Split8 &a, const Split8 &b)
{
    return {op2<IS_MAX>(a.e0, b.e0), op2<IS_MAX>(a.o0, b.o0), op2<IS_MAX>(a.e1, b.e1), op2<IS_MAX>(a.o1, b.o1)};
}

// x window of width W over [left dword | own 2 dwords | right dword] (raw bytes), result for the own 8 voxels
template <int W, bool IS_MAX>
__device__ __forceinline__ Split8 xwin_split(unsigned l, unsigned v0, unsigned v1, unsigned rg)
{
    unsigned E[4], O[4];
    split(l, E[0], O[0]); split(v0, E[1], O[1]); split(v1, E[2], O[2]); split(rg, E[3], O[3]);
    if constexpr (W == 1) return {E[1], O[1], E[2], O[2]};
    // stage 1: t[i] = op3(x[i], x[i+1], x[i+2]);  tE[k] = (t[4k], t[4k+2]),  tO[k] = (t[4k+1], t[4k+3])
    unsigned AE[4], AO[3], tE[4], tO[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { AE[k] = align16(E[k + 1], E[k]); AO[k] = align16(O[k + 1], O[k]); }
    AE[3] = E[3] >> 16;
#pragma unroll
    for (int k = 0; k < 4; k++) tE[k] = op3<IS_MAX>(E[k], O[k], AE[k]);
#pragma unroll
    for (int k = 0; k < 3; k++) tO[k] = op3<IS_MAX>(O[k], AE[k], AO[k]);
    Split8 r;
    unsigned *re[2] = {&r.e0, &r.e1}, *ro[2] = {&r.o0, &r.o1};
#pragma unroll
    for (int k = 1; k <= 2; k++) {
        if constexpr (W == 3) {            // out[p] = t[p-1]
            *re[k - 1] = align16(tO[k], tO[k - 1]);
            *ro[k - 1] = tE[k];
        } else if constexpr (W == 5) {     // out[p] = op2(t[p-2], t[p])
            *re[k - 1] = op2<IS_MAX>(align16(tE[k], tE[k - 1]), tE[k]);
            *ro[k - 1] = op2<IS_MAX>(align16(tO[k], tO[k - 1]), tO[k]);
        } else {                           // W == 7: out[p] = op3(t[p-3], t[p-1], t[p+1])
            *re[k - 1] = op3<IS_MAX>(tO[k - 1], align16(tO[k], tO[k - 1]), tO[k]);
            *ro[k - 1] = op3<IS_MAX>(align16(tE[k], tE[k - 1]), tE[k], align16(tE[k + 1], tE[k]));
        }
    }
    return r;
}

template <int W, bool IS_MAX, int NWP, int NWC, int R, int TY, bool HAS_CONST>
__global__ void __launch_bounds__((NWP + NWC) * 64)
mm3u8_split_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, const U8FusedParams p)
{
    constexpr int ROWS = TY + W - 1;                                     // rows staged per plane
    constexpr int G = (TY + NWC - 1) / NWC;                              // output rows per consumer wave
    constexpr int LROWS = (NWC * G + W - 1) > NWP * R ? (NWC * G + W - 1) : NWP * R;
    constexpr int RX = W / 2;
    static_assert((W == 3 || W == 5 || W == 7) && ROWS <= NWP * R && R <= 16, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4 *lds = reinterpret_cast<u32x4 *>(smem);                         // [2][LROWS][64] x (E0, O0, E1, O1)
    int *ztab = reinterpret_cast<int *>(smem + (size_t)2 * LROWS * 1024);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b = blockIdx.x;
    const int total = p.nxt * p.nyt * p.nzc;
    if ((total & 7) == 0) b = (b & 7) * (total >> 3) + (b >> 3);
    const int per_chunk = p.nxt * p.nyt;
    const int zci = b / per_chunk;
    const int rem = b - zci * per_chunk;
    const int yt = rem / p.nxt, xt = rem - yt * p.nxt;

    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int x0 = xt * 512, y0 = yt * TY, zs = p.zb + zci * p.zc;
    const int ze = min(zs + p.zc, p.zb + p.zn);
    const int ty_act = min(TY, ny - y0);
    const int rows_needed = ty_act + W - 1;
    const int nlanes = min(64, (nx - x0) >> 3);
    const int last = nlanes - 1;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx;
    const int zi0 = zs - RX;
    const int nsteps = ze - zs + W - 1;

    for (int i = threadIdx.x; i < nsteps; i += (NWP + NWC) * 64) ztab[i] = bmap<int>(zi0 + i, nz, p.mz);
    __syncthreads();

    if (wave < NWP) {
        // ------------------------------------------------------------ producer
        int es0, ek0, es1, ek1;
        edge_u8(0, x0, x0 + 8 * nlanes, nx, p.mx, &es0, &ek0);
        edge_u8(1, x0, x0 + 8 * nlanes, nx, p.mx, &es1, &ek1);
        const bool left_side = lane < 32;
        const int erow = left_side ? lane : lane - 32;       // row this lane fetches the edge dword of
        const int ekind = left_side ? ek0 : ek1;
        const int estart = left_side ? es0 : es1;
        unsigned voff[R];
        unsigned eoffv = kOOB;
        bool yconst[R];
        bool e_is_cval = ekind == EDGE_CONST;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int rr = wave * R + r;
            const int ys = rr < rows_needed ? bmap<int>(y0 - RX + rr, ny, p.my) : -2;
            yconst[r] = ys == -1;
            voff[r] = (ys >= 0 && lane < nlanes) ? (unsigned)(ys * nx + x0 + 8 * lane) : kOOB;
            if (erow == r) {
                if (ys >= 0 && ekind != EDGE_CONST) eoffv = (unsigned)(ys * nx + estart);
                if (ys == -1) e_is_cval = true;
            }
        }

        struct Regs { u32x2 v[R]; unsigned e; bool zconst; };
        Regs S[2];
        // (sizes 3 and 5 gain 3 % on 1024^3; size 7 is bound by its VALU work and lost 1 %: not there)
        const bool nt_wave = W <= 5 && p.nt != 0 && wave * R >= W - 1 && wave * R + R <= TY;
        auto issue = [&](int i, Regs &s) {
            int zsrc = zi0 + i;
            if ((unsigned)zsrc >= (unsigned)nz) zsrc = ztab[i];
            s.zconst = zsrc < 0;
            zsrc = __builtin_amdgcn_readfirstlane(max(zsrc, 0));
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(in + (unsigned long long)(unsigned)zsrc * (unsigned long long)plane_bytes), 0, (int)plane_bytes, 0x00020000);
            const bool skip = HAS_CONST && s.zconst;
            // r4: rows W - 1 .. TY - 1 of the staged window are read by this workgroup only (the y-neighbours' windows
            // end at row W - 2 / start at row TY): a wave whose R rows all lie there loads them non-temporally
            if (nt_wave) {
#pragma unroll
                for (int r = 0; r < R; r++) s.v[r] = __builtin_amdgcn_raw_buffer_load_b64(rin, skip ? kOOB : voff[r], 0, 2);
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) s.v[r] = __builtin_amdgcn_raw_buffer_load_b64(rin, skip ? kOOB : voff[r], 0, 0);
            }
            s.e = __builtin_amdgcn_raw_buffer_load_b32(rin, skip ? kOOB : eoffv, 0, 0);
        };

        // z window: t3[t] = op3(x[t], x[t-1], x[t-2]); W = 7: out = op3(t3[t], t3[t-2], t3[t-4]); W = 5: op2(t3[t], t3[t-2])
        constexpr int NT = W == 7 ? 4 : (W == 5 ? 2 : 0);
        constexpr int U = NT > 2 ? NT : 2;                    // steps per unrolled round (history slots are compile-time)
        Split8 hx[2][R], ht[NT > 0 ? NT : 1][R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            hx[0][r] = hx[1][r] = Split8{0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < (NT > 0 ? NT : 1); k++) ht[k][r] = Split8{0u, 0u, 0u, 0u};
        }

        issue(0, S[0]);
        if (nsteps > 1) issue(1, S[1]);
        for (int i0 = 0; i0 < nsteps; i0 += U) {
            static_for<U>([&](auto JJ) {
                constexpr int J = decltype(JJ)::value;
                const int i = i0 + J;
                if (i < nsteps) {
                    Regs &s = S[J & 1];
                    const bool emit = i >= W - 1;
                    u32x4 *wbuf = lds + (J & 1) * (LROWS * 64) + (wave * R) * 64 + lane;
                    unsigned ed = s.e;
                    if (ekind == EDGE_REV) ed = bswap32(ed);
                    else if (ekind == EDGE_SPLAT) ed = (left_side ? (ed & 0xFFu) : (ed >> 24)) * 0x01010101u;
                    if constexpr (HAS_CONST) ed = (e_is_cval || s.zconst) ? p.cval4 : ed;
                    Split8 xf[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        u32x2 v = s.v[r];
                        if constexpr (HAS_CONST)
                            if (yconst[r] || s.zconst) v = (u32x2){p.cval4, p.cval4};
                        const unsigned sL = (unsigned)__builtin_amdgcn_readlane((int)ed, r);
                        const unsigned sR = (unsigned)__builtin_amdgcn_readlane((int)ed, 32 + r);
                        const unsigned l = (unsigned)__builtin_amdgcn_update_dpp((int)sL, (int)v.y, 0x138, 0xf, 0xf, false);
                        unsigned rg = (unsigned)__builtin_amdgcn_update_dpp((int)sR, (int)v.x, 0x130, 0xf, 0xf, false);
                        if (lane == last) rg = sR;
                        xf[r] = xwin_split<W, IS_MAX>(l, v.x, v.y, rg);
                    }
                    if (i + 2 < nsteps) issue(i + 2, s);                  // all rows' registers are consumed
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        const Split8 t3 = op3s<IS_MAX>(xf[r], hx[(J + 1) % 2][r], hx[J % 2][r]);     // planes t, t-1, t-2
                        Split8 o = t3;
                        if constexpr (W == 5) o = op2s<IS_MAX>(t3, ht[J % NT][r]);                           // t3[t-2]
                        else if constexpr (W == 7) o = op3s<IS_MAX>(t3, ht[(J + 2) % NT][r], ht[J % NT][r]);   // t3[t-2], t3[t-4]
                        if (emit) {
                            if constexpr (HAS_CONST)
                                if (yconst[r]) {
                                    unsigned ce, co;
                                    split(p.cval4, ce, co);
                                    o = Split8{ce, co, ce, co};
                                }
                            wbuf[r * 64] = (u32x4){o.e0, o.o0, o.e1, o.o1};
                        }
                        hx[J % 2][r] = xf[r];
                        if constexpr (NT > 0) ht[J % NT][r] = t3;
                    }
                    __syncthreads();
                }
            });
        }
    } else {
        // ------------------------------------------------------------ consumer
        const int cw = wave - NWP;
        const int j0 = cw * G;
        unsigned ovoff[G];
#pragma unroll
        for (int g = 0; g < G; g++)
            ovoff[g] = (j0 + g < ty_act && lane < nlanes) ? (unsigned)((y0 + j0 + g) * nx + x0 + 8 * lane) : kOOB;
        for (int i = 0; i < nsteps; i++) {
            __syncthreads();
            if (i < W - 1) continue;
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(out + (unsigned long long)(unsigned)(zs + i - (W - 1)) * (unsigned long long)plane_bytes), 0,
                (int)plane_bytes, 0x00020000);
            const u32x4 *rbuf = lds + (i & 1) * (LROWS * 64) + j0 * 64 + lane;
            Split8 row[G + W - 1];
#pragma unroll
            for (int k = 0; k < G + W - 1; k++) {
                const u32x4 q = rbuf[k * 64];
                row[k] = Split8{q.x, q.y, q.z, q.w};
            }
            Split8 s1[G + W - 3];
#pragma unroll
            for (int k = 0; k < G + W - 3; k++) s1[k] = op3s<IS_MAX>(row[k], row[k + 1], row[k + 2]);
#pragma unroll
            for (int g = 0; g < G; g++) {
                Split8 o = s1[g];
                if constexpr (W == 5) o = op2s<IS_MAX>(s1[g], s1[g + 2]);
                else if constexpr (W == 7) o = op3s<IS_MAX>(s1[g], s1[g + 2], s1[g + 4]);
                // written once, never read back by this launch: non-temporal
                __builtin_amdgcn_raw_buffer_store_b64((u32x2){join(o.e0, o.o0), join(o.e1, o.o1)}, rout, ovoff[g], 0, 2);
            }
        }
    }
}

template <int W, bool IS_MAX, int NWP, int NWC, int R, int TY>
static int launch_u8_split(const uint8_t *in, uint8_t *out, U8FusedParams &p, bool has_const, hipStream_t s)
{
    constexpr int G = (TY + NWC - 1) / NWC;
    constexpr int LROWS = (NWC * G + W - 1) > NWP * R ? (NWC * G + W - 1) : NWP * R;
    const size_t lds = (size_t)2 * LROWS * 1024 + (size_t)(kU8MaxChunk + 8) * sizeof(int);
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)mm3u8_split_kernel<W, IS_MAX, NWP, NWC, R, TY, false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MI_HIP(hipFuncSetAttribute((const void *)mm3u8_split_kernel<W, IS_MAX, NWP, NWC, R, TY, true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    p.nxt = (p.nx + 511) / 512;
    p.nyt = (p.ny + TY - 1) / TY;
    p.nt = stream_nt_for(2ll * p.nx * p.ny * p.nz, p.nxt);
    const int cus = device_cus();
    const int64_t tiles = (int64_t)p.nxt * p.nyt;
    double best = 1e300;
    int best_nzc = 1;
    if (p.zn <= 0) { p.zb = 0; p.zn = p.nz; }
    const int zn = p.zn;
    for (int nzc = 1; nzc <= std::min(zn, 64); nzc++) {
        const int chunk = (zn + nzc - 1) / nzc;
        if (chunk > kU8MaxChunk) continue;
        const int real = (zn + chunk - 1) / chunk;
        const double rounds = (double)((tiles * real + cus - 1) / cus);
        const double cost = rounds * (chunk + W - 1 + 2.0);
        if (cost < best) { best = cost; best_nzc = real; }
    }
    p.zc = (zn + best_nzc - 1) / best_nzc;
    if (p.zc > kU8MaxChunk) p.zc = kU8MaxChunk;
    p.nzc = (zn + p.zc - 1) / p.zc;
    const int64_t total = tiles * p.nzc;
    note_kernel("mi::mm3u8_split_kernel<%d,%s,%d,%d,%d,%d,%s> grid=%d (fused flat min / max of a uint8 volume: producer / consumer waves)", W, IS_MAX ? "max" : "min", NWP, NWC, R, TY,
                has_const ? "true" : "false", (int)total);
    if (has_const)
        hipLaunchKernelGGL((mm3u8_split_kernel<W, IS_MAX, NWP, NWC, R, TY, true>), dim3((unsigned)total), dim3((NWP + NWC) * 64), lds, s, in, out, p);
    else
        hipLaunchKernelGGL((mm3u8_split_kernel<W, IS_MAX, NWP, NWC, R, TY, false>), dim3((unsigned)total), dim3((NWP + NWC) * 64), lds, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <bool IS_MAX>
static int launch_u8_fused_w(int w, const uint8_t *in, uint8_t *out, U8FusedParams &p, bool has_const, hipStream_t s)
{
    switch (w) {
    case 3: return launch_u8_split<3, IS_MAX, 12, 4, 3, 32>(in, out, p, has_const, s);
    case 5: return launch_u8_split<5, IS_MAX, 12, 4, 3, 32>(in, out, p, has_const, s);
    default: return launch_u8_split<7, IS_MAX, 13, 3, 3, 32>(in, out, p, has_const, s);
    }
}

}  // namespace mi

using namespace mi;

// test / tuning hook (not part of the C-ABI): 0 = always take the two-launch path
static mi::Knob g_u8_fused{1};
extern "C" int mi_debug_set_u8_fused(int enabled) { g_u8_fused = enabled; return MI_OK; }

/* The same restricted to one or two ranges of output planes (uint8 volumes, cubic sizes 3 / 5 / 7: the fused split kernel;
 * MI_ERR_UNSUPPORTED otherwise) -- what the multi-GPU slab schedule needs to overlap the halo exchange (config C's kernel). */
extern "C" int mi_minmax3d_u8_planes(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                                     const int mode[3], int cval, int is_max, const int64_t *planes, int nranges,
                                     mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(size && origin && mode && planes, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(nranges >= 1 && nranges <= 2, MI_ERR_INVALID_ARG, "one or two plane ranges");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("minmax3d_u8_planes: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3 || in->dtype != MI_U8 || out->dtype != MI_U8) UNSUP("needs 3-D uint8 in/out");
    if (!is_contiguous(in) || !is_contiguous(out) || in->data == out->data) UNSUP("needs distinct C-contiguous arrays");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nz < 1 || ny < 1 || nx < 64 || (nx & 15) || (nx & 1023) == 16) UNSUP("x extent must be a multiple of 16, >= 64");
    if (nz * ny * nx >= ((int64_t)1 << 31)) UNSUP("needs a volume < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    const int w = size[0];
    if (size[1] != w || size[2] != w || (w != 3 && w != 5 && w != 7)) UNSUP("cubic sizes 3 / 5 / 7 only");
    if (origin[0] || origin[1] || origin[2]) UNSUP("origin must be 0");
    if (cval < 0 || cval > 255) UNSUP("cval outside uint8");
    int64_t prev_end = 0;
    for (int r = 0; r < nranges; r++) {
        const int64_t b = planes[2 * r], e = planes[2 * r + 1];
        MI_REQUIRE(b >= prev_end && e >= b && e <= nz, MI_ERR_INVALID_ARG, "plane ranges must be ascending and inside the volume");
        prev_end = e;
    }
    hipStream_t s = resolve_stream(stream);
    for (int r = 0; r < nranges; r++) {
        const int64_t b = planes[2 * r], e = planes[2 * r + 1];
        if (e == b) continue;
        U8FusedParams f;
        memset(&f, 0, sizeof(f));
        f.nx = (int)nx; f.ny = (int)ny; f.nz = (int)nz;
        f.mz = filter_mode(mode[0]); f.my = filter_mode(mode[1]); f.mx = filter_mode(mode[2]);
        f.cval4 = (unsigned)cval * 0x01010101u;
        f.zb = (int)b; f.zn = (int)(e - b);
        const bool has_const = f.mz == MI_MODE_CONSTANT || f.my == MI_MODE_CONSTANT || f.mx == MI_MODE_CONSTANT;
        rc = is_max ? launch_u8_fused_w<true>(w, (const uint8_t *)in->data, (uint8_t *)out->data, f, has_const, s)
                    : launch_u8_fused_w<false>(w, (const uint8_t *)in->data, (uint8_t *)out->data, f, has_const, s);
        if (rc != MI_OK) return rc;
    }
    return MI_OK;
#undef UNSUP
}

namespace mi {
int minmax3d_u8_ragged(const mi_array *in, const mi_array *out, const int size[3], const int mode[3], int cval, int is_max,
                       hipStream_t s);   // minmax3d_u8r.hip
}

extern "C" int mi_minmax3d_u8(const mi_array *in, const mi_array *out, const int size[3],
                              const int origin[3], const int mode[3], int cval, int is_max,
                              mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(size && origin && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("minmax3d_u8: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3 || in->dtype != MI_U8 || out->dtype != MI_U8) UNSUP("needs 3-D uint8 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nz * ny * nx >= ((int64_t)1 << 31)) UNSUP("needs a volume < 2 GiB");
    if (nz >= 1 && ny >= 1 && (nx & 15) && cval >= 0 && cval <= 255 && !origin[0] && !origin[1] && !origin[2]) {
        // r6: rows that are not a multiple of 16 bytes as they lie (cubic 3 / 5 / 7, volumes the caches hold)
        rc = minmax3d_u8_ragged(in, out, size, mode, cval, is_max, resolve_stream(stream));
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    if (nz < 1 || ny < 1 || nx < 32 || (nx & 15) || (nx & 1023) == 16) UNSUP("x extent must be a multiple of 16, >= 32");
    if (ny * nx >= ((int64_t)1 << 31)) UNSUP("plane too large");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    for (int a = 0; a < 3; a++) {
        if (size[a] < 1 || !(size[a] & 1) || origin[a] != 0) UNSUP("sizes must be odd with origin 0");
    }
    if (size[2] > 9 || size[0] > 13 || size[1] > 13) UNSUP("size too large for the register kernels");
    if (cval < 0 || cval > 255) UNSUP("cval outside uint8");
    hipStream_t s = resolve_stream(stream);

    U8StreamParams p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.nxt = (int)((nx + 1023) / 1024);
    p.mx = filter_mode(mode[2]);
    p.cval4 = (unsigned)cval * 0x01010101u;
    const uint8_t *ip = (const uint8_t *)in->data;
    uint8_t *op = (uint8_t *)out->data;

    // cubic 3 / 5 / 7: one fused launch
    if (g_u8_fused && size[0] == size[1] && size[1] == size[2] && size[0] >= 3 && size[0] <= 7 && nx >= 64) {
        U8FusedParams f;
        memset(&f, 0, sizeof(f));
        f.nx = (int)nx; f.ny = (int)ny; f.nz = (int)nz;
        f.mz = filter_mode(mode[0]); f.my = filter_mode(mode[1]); f.mx = filter_mode(mode[2]);
        f.cval4 = p.cval4;
        const bool has_const = f.mz == MI_MODE_CONSTANT || f.my == MI_MODE_CONSTANT || f.mx == MI_MODE_CONSTANT;
        return is_max ? launch_u8_fused_w<true>(size[0], ip, op, f, has_const, s)
                      : launch_u8_fused_w<false>(size[0], ip, op, f, has_const, s);
    }

    // images (one plane) and volumes without a z window: x fused into the y pass, one launch
    if (size[0] == 1 && size[1] > 1 && size[2] > 1) {
        p.axis = 1; p.oa = size[1] / 2; p.ma = filter_mode(mode[1]);
        rc = is_max ? launch_u8_wx<true>(size[2], size[1], ip, op, p, s) : launch_u8_wx<false>(size[2], size[1], ip, op, p, s);
        if (rc == MI_ERR_UNSUPPORTED) set_error("minmax3d_u8: no kernel for this size");
        return rc;
    }

    // pass A: x fused with z (into tmp if a y pass follows), pass B: y
    const bool need_y = size[1] > 1;
    const bool need_a = size[0] > 1 || size[2] > 1 || !need_y;
    void *tmp = nullptr;
    if (need_y && need_a) {
        if ((rc = pool_alloc(&tmp, (size_t)(nz * ny * nx), s))) return rc;
    }
    if (need_a) {
        p.axis = 0; p.oa = size[0] / 2; p.ma = filter_mode(mode[0]);
        uint8_t *dst = need_y ? (uint8_t *)tmp : op;
        rc = is_max ? launch_u8_wx<true>(size[2], size[0], ip, dst, p, s) : launch_u8_wx<false>(size[2], size[0], ip, dst, p, s);
        ip = dst;
    }
    if (rc == MI_OK && need_y) {
        p.axis = 1; p.oa = size[1] / 2; p.ma = filter_mode(mode[1]);
        rc = is_max ? launch_u8_wx<true>(1, size[1], ip, op, p, s) : launch_u8_wx<false>(1, size[1], ip, op, p, s);
    }
    if (tmp) pool_free(tmp);
    if (rc == MI_ERR_UNSUPPORTED) set_error("minmax3d_u8: no kernel for this size");
    return rc;
#undef UNSUP
}

/* Flat footprint given as centred runs per row (declared in include/mi355img.h). */
extern "C" int mi_minmax_runs_u8(const mi_array *in, const mi_array *out, int nrows, const int *half_width, const int mode[2],
                                 int cval, int is_max, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(half_width && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("minmax_runs_u8: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if ((in->ndim != 2 && in->ndim != 3) || in->dtype != MI_U8 || out->dtype != MI_U8) UNSUP("needs 2-D / 3-D uint8 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int nd = in->ndim;
    const int64_t nz = nd == 3 ? in->shape[0] : 1, ny = in->shape[nd - 2], nx = in->shape[nd - 1];
    if (nz < 1 || ny < 1 || nx < 32 || (nx & 15) || (nx & 1023) == 16) UNSUP("x extent must be a multiple of 16, >= 32");
    if (nz * ny * nx >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    if (nrows < 1 || nrows > 9 || !(nrows & 1)) UNSUP("1, 3, 5, 7 or 9 footprint rows");
    if (cval < 0 || cval > 255) UNSUP("cval outside uint8");
    U8RunParams p;
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
    p.cval4 = (unsigned)cval * 0x01010101u;
    p.nxt = (int)((nx + 1023) / 1024);
    hipStream_t s = resolve_stream(stream);
    const uint8_t *ip = (const uint8_t *)in->data;
    uint8_t *op = (uint8_t *)out->data;
#define RUNS(N) return is_max ? launch_runs_u8<N, true>(ip, op, p, s) : launch_runs_u8<N, false>(ip, op, p, s)
    switch (nrows) {
    case 1: RUNS(1);
    case 3: RUNS(3);
    case 5: RUNS(5);
    case 7: RUNS(7);
    default: RUNS(9);
    }
#undef RUNS
#undef UNSUP
}

/* 3 x 3 x 3 footprint of centred x runs on a uint8 / bool volume (declared in include/mi355img.h). */
extern "C" int mi_minmax_runs3d_u8(const mi_array *in, const mi_array *out, const int half_width[9], const int mode[3],
                                   int cval, int is_max, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(half_width && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("minmax_runs3d_u8: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3 || (in->dtype != MI_U8 && in->dtype != MI_BOOL) || out->dtype != in->dtype) UNSUP("needs 3-D uint8 / bool in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nz * ny * nx >= ((int64_t)1 << 31)) UNSUP("needs a volume < 2 GiB");
    if (nz < 1 || ny < 1 || nx < 32 || (nx & 15) || (nx & 1023) == 16) UNSUP("x extent must be a multiple of 16, >= 32");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    if (cval < 0 || cval > 255) UNSUP("cval outside uint8");
    U8Run3Params p;
    memset(&p, 0, sizeof(p));
    bool any = false;
    for (int k = 0; k < 9; k++) {
        if (half_width[k] < -1 || half_width[k] > 1) UNSUP("runs of at most 3 voxels");
        p.hw[k / 3][k % 3] = half_width[k];
        any = any || half_width[k] >= 0;
    }
    if (!any) UNSUP("empty footprint");
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.mz = filter_mode(mode[0]); p.my = filter_mode(mode[1]); p.mx = filter_mode(mode[2]);
    p.cval4 = (unsigned)cval * 0x01010101u;
    p.nxt = (int)((nx + 1023) / 1024);
    const int nlines = p.ny * p.nxt;
    int nch = 1;
    {
        double best = 1e300;
        for (int c = 1; c <= p.nz && c <= 1024; c++) {
            const int chunk = (p.nz + c - 1) / c;
            if (c > 1 && chunk < 8) break;
            const int real = (p.nz + chunk - 1) / chunk;
            const double rounds = std::max(1.0, (double)nlines * real / 4096.0);
            const double cost = rounds * (chunk + 2 + 4.0);
            if (cost < best * 0.999) { best = cost; nch = real; }
        }
    }
    p.chunk = (p.nz + nch - 1) / nch;
    p.nchunks = (p.nz + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)nx * ny * nz);
    hipStream_t s = resolve_stream(stream);
    if (is_max)
        hipLaunchKernelGGL(runs3d_minmax_u8_kernel<true>, dim3((waves + 3) / 4), dim3(256), 0, s, (const uint8_t *)in->data, (uint8_t *)out->data, p);
    else
        hipLaunchKernelGGL(runs3d_minmax_u8_kernel<false>, dim3((waves + 3) / 4), dim3(256), 0, s, (const uint8_t *)in->data, (uint8_t *)out->data, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
#undef UNSUP
}

/* uniform_filter on a uint8 image / slice-wise on a uint8 volume, uint8 result (declared in include/mi355img.h). */
extern "C" int mi_uniform2d_u8(const mi_array *in, const mi_array *out, const int size[2], int origin_y, const int mode[2],
                               int cval, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(size && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("uniform2d_u8: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if ((in->ndim != 2 && in->ndim != 3) || in->dtype != MI_U8 || out->dtype != MI_U8) UNSUP("needs 2-D / 3-D uint8 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int nd = in->ndim;
    const int64_t nz = nd == 3 ? in->shape[0] : 1, ny = in->shape[nd - 2], nx = in->shape[nd - 1];
    if (nz < 1 || ny < 1 || nx < 32 || (nx & 15) || (nx & 1023) == 16) UNSUP("x extent must be a multiple of 16, >= 32");
    if (nz * ny * nx >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    const int wy = size[0], wx = size[1];
    if (wy < 1 || wy > 9 || !(wy & 1) || wx < 1 || wx > 9 || !(wx & 1)) UNSUP("sizes must be odd and <= 9");
    if (wy == 1 && wx == 1) UNSUP("nothing to filter");
    const int oy = wy / 2 + origin_y;
    if (oy < 0 || oy >= wy) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    if (cval < 0 || cval > 255) UNSUP("cval outside uint8");
    U8BoxParams p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.axis = 1;
    p.oy = oy;
    p.my = filter_mode(mode[0]); p.mx = filter_mode(mode[1]);
    p.cval4 = (unsigned)cval * 0x01010101u;
    p.nxt = (int)((nx + 1023) / 1024);
    p.ry = (float)(1.0 / wy); p.rx = (float)(1.0 / wx);
    hipStream_t s = resolve_stream(stream);
    const uint8_t *ip = (const uint8_t *)in->data;
    uint8_t *op = (uint8_t *)out->data;
    switch (wx) {
    case 1: return launch_box2d_u8_wy<1>(wy, ip, op, p, s);
    case 3: return launch_box2d_u8_wy<3>(wy, ip, op, p, s);
    case 5: return launch_box2d_u8_wy<5>(wy, ip, op, p, s);
    case 7: return launch_box2d_u8_wy<7>(wy, ip, op, p, s);
    default: return launch_box2d_u8_wy<9>(wy, ip, op, p, s);
    }
#undef UNSUP
}

/* The z pass of uniform_filter on a uint8 volume (uint8 intermediate, as SciPy stores it): trunc(sum of size_z planes /
 * size_z); mi_uniform2d_u8 on the result completes the filter (declared in include/mi355img.h). */
extern "C" int mi_uniform_z_u8(const mi_array *in, const mi_array *out, int size_z, int origin_z, int mode_z, int cval,
                               mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("uniform_z_u8: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3 || in->dtype != MI_U8 || out->dtype != MI_U8) UNSUP("needs 3-D uint8 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nz < 1 || ny < 1 || nx < 32 || (nx & 15)) UNSUP("x extent must be a multiple of 16, >= 32");
    if (nz * ny * nx >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    if (size_z < 3 || size_z > 9 || !(size_z & 1)) UNSUP("odd size 3 .. 9");
    const int oz = size_z / 2 + origin_z;
    if (oz < 0 || oz >= size_z) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    if (cval < 0 || cval > 255) UNSUP("cval outside uint8");
    U8BoxParams p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.axis = 0;
    p.oy = oz;
    p.my = filter_mode(mode_z); p.mx = MI_MODE_REFLECT;
    p.cval4 = (unsigned)cval * 0x01010101u;
    p.nxt = (int)((nx + 1023) / 1024);
    p.ry = (float)(1.0 / size_z); p.rx = 1.0f;
    return launch_box2d_u8_wy<1>(size_z, (const uint8_t *)in->data, (uint8_t *)out->data, p, resolve_stream(stream));
#undef UNSUP
}
