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
#include "common.hpp"

namespace mi {

constexpr int kStreamMaxTaps = 33;

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
};

}  // namespace mi
