// stream3d.hpp -- barrier-free streaming passes for long separable kernels
// (float32, 3-D).  Used by mi_separable3d_f32 when a tap count exceeds what
// the fused single-launch kernels hold in registers (e.g. gaussian sigma=2 ->
// 17 taps per axis, BASELINE config B): the filter runs as TWO launches,
//     A:  x pass (in registers, DPP lane shifts) fused with the z pass
//     B:  y pass
// i.e. 16 B/voxel instead of the reference's three launches + fills + copies.
//
// Each wave is independent (no LDS, no barrier): it owns one 256-float row
// segment position (lane l holds the float4 at x0 + 4l) and streams along the
// pass axis over a chunk, keeping the last W-1 samples of every lane in a
// register ring that is rotated by unrolling.  Loads are buffer_load_dwordx4
// with the chunk position in the scalar offset; DEPTH loads are kept in flight.
#pragma once
#include "sep_common.hpp"

namespace mi {

constexpr int kStreamMaxTaps = 33;
constexpr int kStreamFusedMax = 17;    // longest kernel whose x pass is fused into the streamed pass (same tap count on both)

struct StreamParams {
    int nx, ny, nz;
    int axis;            // streamed axis: 0 = z, 1 = y
    int wa, oa, ma;      // taps / offset (w/2+origin) / mode along the streamed axis
    int mx;              // x boundary mode (x pass fused when WX > 1)
    int mo;              // boundary mode of the other (non-streamed, non-x) axis: unused (no taps there)
    float cval;
    int chunk, nchunks;  // outputs per chunk along the streamed axis
    int nxt;             // x tiles of 256 floats
    float wav[kStreamMaxTaps];
    float wxv[kStreamMaxTaps];
    // x weights as the pairs the packed dot product multiplies with aligned
    // window pairs: xpair[q][2u], xpair[q][2u+1] = wx[2u - (B+q)%2], wx[2u + 1 - (B+q)%2]
    // for output parity q (B = window offset of tap 0, see xpass_hops); 0 outside the kernel
    float xpair[2][2 * (kStreamMaxTaps / 2 + 2)];
    int wid_base;        // first wave index of this launch (a pass is issued in slices of waves)
    int wpb;             // waves per workgroup of this launch
    int swz;             // XCD-aware workgroup order (xcd_block())
};

// Workgroups are handed to the 8 XCDs round robin (workgroup i -> XCD i % 8), and every XCD has its own L2: with the
// plain order, neighbouring chunks of an image -- which share their ramp rows -- land on different XCDs and each L2
// fetches those rows from HBM again (measured 1.3-1.4x the algorithmic reads for 7-17-row windows).  This order gives
// every XCD a contiguous run of workgroups.  g_xcd_swizzle: test hook (0 = plain order).
extern Knob g_xcd_swizzle;        // 0 = plain order, 1 = always, 2 = auto
// auto: arrays up to 128 MiB (4096^2 float32 +7 %, 8192^2 float32 +4 %; a 1 GiB image LOSES 7 %: eight distant streams
// instead of one front moving through the DRAM pages)
static inline int xcd_swizzle_for(size_t array_bytes)
{
    return g_xcd_swizzle == 1 || (g_xcd_swizzle == 2 && array_bytes <= ((size_t)128 << 20));
}
__device__ __forceinline__ int xcd_block(int b, int nblocks, int enabled)
{
    return (enabled && (nblocks & 7) == 0) ? (b & 7) * (nblocks >> 3) + (b >> 3) : b;
}

// what a pass computes: weighted sum, or running minimum / maximum
enum { SP_CORR = 0, SP_MIN = 1, SP_MAX = 2 };

constexpr int gcd_(int a, int b) { return b == 0 ? a : gcd_(b, a % b); }
constexpr int lcm_(int a, int b) { return a / gcd_(a, b) * b; }

// block j (1 = nearest, 2 = next) of 4 floats outside the tile on `side`
// (0 left, 1 right): element offset inside the row to load 4 floats from, and
// what to do with them
__device__ __forceinline__ void edge_block(int side, int j, int x0, int xe, int nx, int mode, int *start, int *kind)
{
    if (side == 0) {
        if (x0 - 4 * j >= 0) { *start = x0 - 4 * j; *kind = EDGE_FWD; return; }
        const int k0 = 4 * j - x0;   // how far the block's far end reaches beyond the array (x0 is a multiple of 256: 0 here)
        (void)k0;
        switch (mode) {
        case MI_MODE_REFLECT:   *start = 4 * (j - 1); *kind = EDGE_REV; break;        // ext -k = x[k-1]
        case MI_MODE_MIRROR:    *start = 4 * (j - 1) + 1; *kind = EDGE_REV; break;    // ext -k = x[k]
        case MI_MODE_NEAREST:   *start = 0; *kind = EDGE_SPLAT; break;
        case MI_MODE_GRID_WRAP: *start = nx - 4 * j; *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    } else {
        if (xe + 4 * j <= nx) { *start = xe + 4 * (j - 1); *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = nx - 4 * j; *kind = EDGE_REV; break;         // ext n-1+k = x[n-k]
        case MI_MODE_MIRROR:    *start = nx - 1 - 4 * j; *kind = EDGE_REV; break;     // ext n-1+k = x[n-1-k]
        case MI_MODE_NEAREST:   *start = nx - 4; *kind = EDGE_SPLAT; break;           // splat component 3
        case MI_MODE_GRID_WRAP: *start = 4 * (j - 1); *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    }
}

__device__ __forceinline__ float4 apply_kind(float4 t, int kind, int side, float cval)
{
    if (kind == EDGE_REV) return make_float4(t.w, t.z, t.y, t.x);
    if (kind == EDGE_SPLAT) { const float s = side == 0 ? t.x : t.w; return make_float4(s, s, s, s); }
    if (kind == EDGE_CONST) return make_float4(cval, cval, cval, cval);
    return t;
}

}  // namespace mi
