// cubic_fast.hip -- r5: order-3 (cubic B-spline) affine transforms on float32 coefficients whose matrix leaves the stream axis
// to itself, evaluated PLANE BY PLANE.
//
// Reference: the 4 x 4 x 4 tap loop of cupyimg/scipy/ndimage/_interp_kernels.py:473-549 behind affine_transform / rotate
// (interpolation.py:397-709, order 3 = the default).
//
// cubic3_zstream_kernel (interp.hip, r4b) stages the input planes of a tile in LDS once, but still evaluates every output voxel
// as 64 taps: 64 ds_read_b32 + 80 FMA + ~170 other VALU instructions per voxel -- issue bound at 0.10 of the HBM roofline.
// When the matrix couples only the two in-plane axes, the in-plane part of a voxel (its 4 x 4 taps, the eight weights) does
// not depend on the output plane, and the interpolated value factors:
//
//     out(z, y, x) = sum_kz wz[z][kz] * Q_p(kz)(y, x),      Q_p(y, x) = sum_ky wy[ky] sum_kx wx[kx] c[p][ty + ky][tx + kx]
//
// Q_p -- the in-plane interpolation of INPUT plane p at the voxel's in-plane position -- is the same for every output plane
// that reads plane p: four of them at a step of one plane.  A thread keeps Q of the four planes of the current step for its
// eight voxels in registers (a ring indexed by plane mod 4), evaluates Q once per input plane and voxel (16 LDS reads at
// IMMEDIATE offsets from one base address per voxel, 20 FMA) and blends four values per output voxel: 16 reads + 24 FMA per
// voxel instead of 64 + 80, no per-plane address arithmetic.  Staging, ring, chunking and the hand-counted waits are those
// of cubic3_zstream_kernel (same CubZParams, same plan: launch_cubic_zstream in interp.hip).
//
// CONTRACT.  The sums are taken in another order than cubic3_gather's (in-plane first, then along the stream axis): results
// agree with the gather kernel to float32 rounding (<= 2e-6 of the coefficient range in the tests), not bit for bit; the
// bound that matters is SciPy's: 2e-5 max(1, max|ref|) for this float32 route (tests/test_gpu_baseline_full.py, every plane
// of 512^3).  Voxels / steps the factored form does not cover -- taps that fold at the array ends in ways the rectangle
// does not hold, cval taps along the stream axis, two planes of a step in one ring slot -- take cubic3_gather itself, as in
// the r4b kernel.
#include <algorithm>

#include "interp_common.hpp"

namespace mi {

constexpr int kBoxT = 16;          // edge of the output cube a workgroup of the box kernels owns (cubic3_box_kernel, cubic3_mapbox_kernel)

// cubic3_gather's taps, products and order of sums (bit-identical to it), with the four rows of ONE stream-axis tap in flight
// at a time: the fallback of a kernel that keeps ~130 registers of per-voxel state (cubic3_gather itself holds all 64 taps:
// the kernel spilled, which a kernel that counts its vector-memory operations must not)
__device__ __forceinline__ float cubic3_gather_lean(const __amdgpu_buffer_rsrc_t rin, const Cubic3 &t, float cval)
{
    const bool consec = t.off[2][0] >= 0 && t.off[2][3] == t.off[2][0] + 3;
    float acc = 0.f;
#pragma unroll
    for (int kz = 0; kz < 4; kz++) {
        __builtin_amdgcn_sched_barrier(0);          // (unrolled: a rolled loop would index the tap arrays at run time -- scratch)
        float v[4][4];
#pragma unroll
        for (int ky = 0; ky < 4; ky++) {
            const bool oob_zy = t.off[0][kz] < 0 || t.off[1][ky] < 0;
            const int base = oob_zy ? 0 : t.off[0][kz] + t.off[1][ky];
            if (consec) {
                const u32x4 qv = __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)(base + t.off[2][0]) * 4u, 0, 0);
                v[ky][0] = oob_zy ? cval : __uint_as_float(qv.x); v[ky][1] = oob_zy ? cval : __uint_as_float(qv.y);
                v[ky][2] = oob_zy ? cval : __uint_as_float(qv.z); v[ky][3] = oob_zy ? cval : __uint_as_float(qv.w);
            } else {
#pragma unroll
                for (int kx = 0; kx < 4; kx++) {
                    const bool oob = oob_zy || t.off[2][kx] < 0;
                    const float qv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rin, oob ? 0u : (unsigned)(base + t.off[2][kx]) * 4u, 0, 0));
                    v[ky][kx] = oob ? cval : qv;
                }
            }
        }
#pragma unroll
        for (int ky = 0; ky < 4; ky++) {
            const float wzy = t.w[0][kz] * t.w[1][ky];
            float row = v[ky][0] * t.w[2][0];
            row = fmaf(v[ky][1], t.w[2][1], row);
            row = fmaf(v[ky][2], t.w[2][2], row);
            row = fmaf(v[ky][3], t.w[2][3], row);
            acc = fmaf(row, wzy, acc);
        }
    }
    return t.outside ? cval : acc;
}

// the taps of output plane z along the stream axis (wave-uniform).  Away from the array ends (all four planes inside, whatever the
// mode) they follow from floor() alone; the boundary arithmetic of cubic3_axis (coordinate folding with divisions, per-tap
// maps) runs only in the steps that need it.  Shared by the streaming kernel and the fix-up kernel: the two must agree on which
// steps the streaming kernel leaves out.
struct ZTaps { float w[4]; int pl[4]; bool outside, cvtap, plain; };      // plain: four consecutive planes inside the array (pl[k] = pl[0] + k)
// the table entry of an output plane (cubic3_ztaps_kernel): what the taps of a step are does not depend on the tile -- one
// thread per output plane computes them once per launch, the streaming kernel and the fix-up kernel read a record per step with
// scalar loads (r5b: computed per step by every wave -- double arithmetic on the vector unit, a dozen readfirstlanes -- they were
// a quarter of the streaming kernel's per-step instructions)
struct ZRec { float w[4]; int pl[4]; int flags; int pad_[3]; };           // flags: 1 outside, 2 cvtap, 4 plain
__device__ __forceinline__ ZRec cz_ztaps_lane(const CubZParams &q, int z)
{
    ZRec p;
    double s0 = 0.0; s0 += q.m00 * (double)z;
    const double cz = s0 + q.m03;
    const double cc = cz + (double)q.npad;
    const double fl = floor(cc);
    const bool plain = fl >= 1.0 && fl + 2.0 <= (double)(q.nz - 1);
    float w[4]; int pl[4];
    bool outside = false, cv = false;
    if (plain) {
        cubic3_weights((float)(cc - fl), w);
        const int st = (int)fl - 1;
#pragma unroll
        for (int k = 0; k < 4; k++) pl[k] = st + k;
    } else {
        int off[4];
        outside = cubic3_axis(q.nz, 1, cz, q.mode, q.npad, w, off);
#pragma unroll
        for (int k = 0; k < 4; k++) { pl[k] = off[k]; cv = cv || off[k] < 0; }
    }
    if (q.sident) {
        // the stream axis holds samples (its prefilter pass was skipped) at an integral coordinate: the tap AT the coordinate
        // is the second of the four and the only one -- one plane per step, weight one
        cv = pl[1] < 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { pl[k] = pl[1]; w[k] = k == 1 ? 1.f : 0.f; }
    }
    // two DIFFERENT planes of one step in the same slot of the register ring (plane mod 4): planes that wrap around the array
    // (four consecutive planes never do) -- the step is left to the fix-up kernel
    if (!plain) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = i + 1; j < 4; j++)
                cv = cv || (pl[i] >= 0 && pl[j] >= 0 && pl[i] != pl[j] && (pl[i] & 3) == (pl[j] & 3));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) { p.w[k] = w[k]; p.pl[k] = pl[k]; }
    p.flags = (outside ? 1 : 0) | (cv ? 2 : 0) | ((plain && !q.sident) ? 4 : 0);
    p.pad_[0] = p.pad_[1] = p.pad_[2] = 0;
    return p;
}

__global__ void __launch_bounds__(64)
cubic3_ztaps_kernel(ZRec *__restrict__ tab, const CubZParams q)
{
    const int z = blockIdx.x * 64 + threadIdx.x;
    if (z < q.oz) tab[z] = cz_ztaps_lane(q, z);
}

// a record as wave-uniform values (z is uniform: the loads are scalar)
__device__ __forceinline__ ZTaps cz_ztaps(const ZRec *__restrict__ tab, int z)
{
    const ZRec r = tab[z];
    ZTaps p;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        p.w[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(r.w[k])));
        p.pl[k] = __builtin_amdgcn_readfirstlane(r.pl[k]);
    }
    const int f = __builtin_amdgcn_readfirstlane(r.flags);
    p.outside = (f & 1) != 0; p.cvtap = (f & 2) != 0; p.plain = (f & 4) != 0;
    return p;
}

// FIX-UP: the voxels the streaming kernel leaves out -- waves with a voxel whose in-plane taps the staged rectangle does not
// hold (flag per tile and wave, written by the streaming kernel), and whole steps with a cval tap along the stream axis or
// two planes in one ring slot -- by the gather routine, one wave per (tile, wave, output plane); almost all of them return
// after reading one flag.  Out of the streaming kernel since r5: with the gather code inline its register allocation was
// that of the fallback (256 registers and spills in the hot loop).
constexpr int kCzFixPlanes = 16;        // output planes per fix-up workgroup (one per plane: 262 144 workgroups on 512^3, 75 us of dispatch)

template <int SAX>
__global__ void __launch_bounds__(64)
cubic3_zfix_kernel(const float *__restrict__ in, float *__restrict__ out, const CubZParams q, const int *__restrict__ far_flags, const ZRec *__restrict__ ztab)
{
    const int tw = blockIdx.x;
    const int wave = tw & 3, tile = tw >> 2;
    const int lane = threadIdx.x;
    const bool far = far_flags[tw] != 0 || (q.dbg & 2);
    const int tx_i = tile % q.ntx, ty_i = tile / q.ntx;
    const int x = tx_i * 64 + lane;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, q.vol_bytes, 0x00020000);
    const int z1 = min((int)(blockIdx.y + 1) * kCzFixPlanes, q.oz);
#pragma unroll 1
    for (int z = blockIdx.y * kCzFixPlanes; z < z1; z++) {
        double s0 = 0.0; s0 += q.m00 * (double)z;
        const ZTaps zt = cz_ztaps(ztab, z);
        if (!far && zt.plain) continue;                                    // four plain planes inside the array: nothing for this wave to do
        if (zt.outside) continue;                                          // the streaming kernel wrote cval
        if (!(far || zt.cvtap)) continue;
        if (x >= q.ox) continue;
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            const int y = ty_i * kCzTY + 8 * wave + k;
            if (y >= q.oy) break;
            Cubic3 tt;
            const double o1 = (double)y, o2 = (double)x;
            double s1 = 0.0; s1 += q.m11 * o1; s1 += q.m12 * o2;
            double s2 = 0.0; s2 += q.m21 * o1; s2 += q.m22 * o2;
            bool outside = cubic3_axis(q.ny, q.sr, s1 + q.m13, q.mode, q.npad, tt.w[1 - SAX], tt.off[1 - SAX]);
            outside |= cubic3_axis(q.nx, 1, s2 + q.m23, q.mode, q.npad, tt.w[2], tt.off[2]);
            (void)cubic3_axis(q.nz, q.ss, s0 + q.m03, q.mode, q.npad, tt.w[SAX], tt.off[SAX]);
            if (q.sident) {
#pragma unroll
                for (int j = 0; j < 4; j++) { tt.off[SAX][j] = tt.off[SAX][1]; tt.w[SAX][j] = j == 1 ? 1.f : 0.f; }
            }
            tt.ntap[0] = 4; tt.ntap[1] = 4;
            tt.outside = outside;
            __builtin_nontemporal_store(cubic3_gather_lean(rin, tt, q.cval), out + ((size_t)z * q.oss + (size_t)y * q.osr + x));
        }
    }
}

template <int SAX>
__global__ void __launch_bounds__(kCzNT, 2)
cubic3_zfactor_kernel(const float *__restrict__ in, float *__restrict__ out, const CubZParams q, int *__restrict__ far_flags, const ZRec *__restrict__ ztab)
{
    constexpr int P = kCzP, TY = kCzTY, NT = kCzNT, NS = kCzSlots;
    extern __shared__ __attribute__((aligned(16))) char smem_cz[];
    const unsigned slot_bytes = (unsigned)q.slot_bytes;
    float *tiles = reinterpret_cast<float *>(smem_cz + max((unsigned)NS * slot_bytes, 4u * 24u * 64u * 4u));      // [4 waves][8 rows][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = q.ntx * q.nty * q.nzc;
    int t = blockIdx.x;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);          // x-neighbouring tiles on one XCD
    const int tx_i = t % q.ntx, ty_i = (t / q.ntx) % q.nty, zc_i = t / (q.ntx * q.nty);
    const int x0 = tx_i * 64, y0 = ty_i * TY;
    const int zs = zc_i * q.zc, ze = min(zs + q.zc, q.oz);
    const int vol_bytes = q.vol_bytes;
    const unsigned plane_b = (unsigned)q.ss * 4u, row_b = (unsigned)q.sr * 4u;

    // ---- rectangle origin (as cubic3_zstream_kernel)
    int by0, bx0;
    {
        const double cy = ((q.m11 * (double)y0 + q.m12 * (double)x0) + q.m13) + q.cmin_y + (double)q.npad;
        const double cx = ((q.m21 * (double)y0 + q.m22 * (double)x0) + q.m23) + q.cmin_x + (double)q.npad;
        double fy = floor(cy - 1e-6 * (1.0 + fabs(cy))) - 1.0, fx = floor(cx - 1e-6 * (1.0 + fabs(cx))) - 1.0;
        fy = fy < 0.0 ? 0.0 : (fy > (double)(q.ny - 1) ? (double)(q.ny - 1) : fy);
        fx = fx < 0.0 ? 0.0 : (fx > (double)(q.nx - 1) ? (double)(q.nx - 1) : fx);
        by0 = __builtin_amdgcn_readfirstlane((int)fy);
        bx0 = __builtin_amdgcn_readfirstlane((int)fx & ~3);
    }
    const int rounds = (q.nchunks + NT - 1) / NT;
    unsigned rel[kCzRoundsMax];
#pragma unroll
    for (int j = 0; j < kCzRoundsMax; j++) {
        const unsigned ch = (unsigned)tid + (unsigned)(j * NT);
        const unsigned row = ch / 20u, c4 = ch - row * 20u;
        rel[j] = ch < (unsigned)q.nchunks ? row * row_b + c4 * 16u : 0x80000000u;
    }
    // lanes of round j that hold a chunk (the r4b kernel kept the eight masks in sixteen scalar registers; here they are
    // recomputed where a plane is staged: one compare per DMA instruction)
    auto live_of = [&](int j) { return __builtin_amdgcn_ballot_w64(rel[j] != 0x80000000u); };
    // rounds in which THIS wave has any chunk: the wave's first chunk of round j is chunk 64 wave + 256 j
    int ndma = 0;                                            // DMA instructions of this wave per plane
#pragma unroll
    for (int j = 0; j < kCzRoundsMax; j++) ndma += (j < rounds && (unsigned)(64 * wave + NT * j) < (unsigned)q.nchunks) ? 1 : 0;
    const unsigned org_b = ((unsigned)by0 * (unsigned)q.sr + (unsigned)bx0) * 4u;

    // ---- per voxel (row y0 + 8 wave + k, column x0 + lane), once: the in-plane taps -- plain / edge / far exactly as in
    // cubic3_zstream_kernel (see there); an edge voxel additionally sets bit 31 of its packed start
    const int rule = q.mode == MI_MODE_NEAREST ? 2 : (q.mode == MI_MODE_REFLECT ? 1 : ((q.mode == MI_MODE_GRID_WRAP || q.mode == MI_MODE_GRID_CONSTANT) ? 3 : 0));
    auto czrule = [&](int i, int n) {
        const int lo = rule == 2 ? 0 : -i - rule, hi = rule == 2 ? n - 1 : 2 * n - 2 + rule - i;
        return i < 0 ? lo : (i >= n ? hi : i);
    };
    int a_[8];
    float fy_[8], fx_[8];
    unsigned farmask = 0, outmask = 0, edgemask = 0;
    auto inplane = [&](int k, float &fy, int (&offy)[4], float &fx, int (&offx)[4]) {
        const double o1 = (double)(y0 + 8 * wave + k), o2 = (double)(x0 + lane);
        double s1 = 0.0; s1 += q.m11 * o1; s1 += q.m12 * o2;
        double s2 = 0.0; s2 += q.m21 * o1; s2 += q.m22 * o2;
        const bool oy_ = cubic3_axis_frac(q.ny, q.sr, s1 + q.m13, q.mode, q.npad, fy, offy);
        const bool ox_ = cubic3_axis_frac(q.nx, 1, s2 + q.m23, q.mode, q.npad, fx, offx);
        return oy_ | ox_;
    };
    float *park = reinterpret_cast<float *>(smem_cz) + wave * (24 * 64) + lane;
#pragma unroll 1
    for (int k = 0; k < 8; k++) {
        int offy[4], offx[4];
        float fy, fx;
        const bool outside = inplane(k, fy, offy, fx, offx);
        bool block = offy[0] >= 0 && offx[0] >= 0;
#pragma unroll
        for (int j = 1; j < 4; j++) block = block && offy[j] == offy[0] + j * q.sr && offx[j] == offx[0] + j;
        int r0 = offy[0] / q.sr - by0, c0 = offx[0] - bx0;
        const bool held = block && r0 >= 0 && r0 + 3 < q.ry && c0 >= 0 && c0 + 3 < P;
        bool edge = false;
        if (!held && !outside && rule != 3) {
            int sy = 0, sx = 0;
            bool fy_ok = false, fx_ok = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!fy_ok && offy[j] >= 0) {
                    const int s = offy[j] / q.sr - j;
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; i++) ok = ok && offy[i] == czrule(s + i, q.ny) * q.sr;
                    if (ok) { sy = s; fy_ok = true; }
                }
                if (!fx_ok && offx[j] >= 0) {
                    const int s = offx[j] - j;
                    bool ok = true;
#pragma unroll
                    for (int i = 0; i < 4; i++) ok = ok && offx[i] == czrule(s + i, q.nx);
                    if (ok) { sx = s; fx_ok = true; }
                }
            }
            edge = fy_ok && fx_ok && sy - by0 >= -8 && sx - bx0 >= -8 && sy - by0 < 4096 && sx - bx0 < 4096;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int ty = czrule(sy + j, q.ny) - by0, tx = czrule(sx + j, q.nx) - bx0;
                edge = edge && ty >= 0 && ty < q.ry && tx >= 0 && tx < P;
            }
            r0 = sy - by0; c0 = sx - bx0;
        }
        park[(3 * k) * 64] = __int_as_float((held || edge) ? ((r0 + 8) << 16) | (c0 + 8) : (8 << 16) | 8);
        park[(3 * k + 1) * 64] = fy;
        park[(3 * k + 2) * 64] = fx;
        farmask |= (held || edge || outside) ? 0u : (1u << k);
        outmask |= outside ? (1u << k) : 0u;
        edgemask |= (edge && !held) ? (1u << k) : 0u;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        a_[k] = __float_as_int(park[(3 * k) * 64]);
        fy_[k] = park[(3 * k + 1) * 64];
        fx_[k] = park[(3 * k + 2) * 64];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // before anyone's DMA lands on the parked values
    float wy_[8][4], wx_[8][4];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        cubic3_weights(fy_[k], wy_[k]);
        cubic3_weights(fx_[k], wx_[k]);
    }
    const bool any_far = __builtin_amdgcn_ballot_w64(farmask != 0u) != 0;
    // (every z chunk of the tile writes the same value: the flag depends on the in-plane geometry only)
    if (lane == 0) far_flags[(ty_i * q.ntx + tx_i) * 4 + wave] = any_far ? 1 : 0;
    const bool any_edge = __builtin_amdgcn_ballot_w64(edgemask != 0u) != 0;
    const bool imm_candidate = slot_bytes == (unsigned)kCzSlot && !any_edge;      // every voxel of the wave a plain block, slots of the fixed size: immediate offsets
    float *tile = tiles + wave * 512;
    const bool wide = x0 + 64 <= q.ox && y0 + TY <= q.oy;

    // ---- the plane ring in LDS.  Unlike the r4b kernel, a plane is READ from LDS in one step only (the step that evaluates its
    // in-plane values into the register ring); afterwards its slot is free.  So the five slots are a prefetch QUEUE, filled
    // round robin (`head`; a plane's slot is found by content, not by plane mod 5): planes of the next kLook steps are
    // requested as soon as the slot at the head holds a plane that has been evaluated -- at the r4b kernel's depth of one
    // step a step now ends (16 reads + 24 FMA per voxel) long before its successor's plane has crossed the memory system.
    //
    // Bookkeeping (wave-uniform scalars, IDENTICAL in every wave of the workgroup -- each wave stages its share of every plane,
    // so all must take the same decisions; nothing below depends on a wave's own voxels):
    //   res[s]   plane in slot s (or none);  mark[s]  value of `issued` right after this wave's DMAs for it;
    //   issued   vector-memory instructions this wave has CERTAINLY issued: its DMAs and the two stores of a full-tile step.
    // The wait before a step needs the DMAs of the planes it evaluates: every vector-memory operation up to max(mark) must be
    // complete, i.e. at most N = issued - max(mark) younger ones may be in flight (they retire in order): s_waitcnt vmcnt(N')
    // with N' the largest encodable choice <= N.  Instructions the count does not know (stores of partial tiles) only make N
    // an UNDER-estimate of what may legally be in flight: the wait is then longer, never shorter.
    constexpr int kLook = 3;
    constexpr int kNone = -0x7fffffff;
    // the tables live in the LANES of one vector register (wave-uniform content, the same in every wave): lanes 0-4 res[], 8-12
    // mark[], 16-19 the tags of the register ring -- an element is one v_readlane (v_cndmask to write) with the index in a scalar, a
    // search one compare + ballot.  (As fourteen scalars selected by ?: chains the compiler built branch trees: 460 scalar
    // instructions per step.)
    int bk = lane < 5 ? kNone : (lane >= 16 && lane < 20 ? kNone : 0);
    int issued = 0;
    int head = 0;                                            // the slot the next plane goes to
    auto find = [&](int pl) {
        const unsigned m = (unsigned)__builtin_amdgcn_ballot_w64(bk == pl) & 0x1fu;
        return m ? (int)__builtin_ctz(m) : -1;
    };
    auto res_at = [&](int sl) { return __builtin_amdgcn_readlane(bk, sl); };
    auto mark_at = [&](int sl) { return __builtin_amdgcn_readlane(bk, 8 + sl); };
    auto fetch = [&](int pl, int sl) {                       // stage plane pl (0 <= pl < nz) into slot sl
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
        const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)pl * plane_b + org_b);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)sl * slot_bytes + (unsigned)(wave << 6) * 16u);
#pragma unroll
        for (int j = 0; j < kCzRoundsMax; j++)
            if (j < ndma) cz_dma16(rin, rel[j], base, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(j * NT) * 16u), live_of(j));
        issued += ndma;
        bk = lane == sl ? pl : (lane == 8 + sl ? issued : bk);
        head = sl == 4 ? 0 : sl + 1;
    };
    auto zplane = [&](int z) { return cz_ztaps(ztab, z); };

    // ---- Q ring: in-plane values of the planes of the current step, slot (plane & 3); the tags are wave-uniform
    float Q[4][8];
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int k = 0; k < 8; k++) Q[s][k] = 0.f;

    // in-plane interpolation of the plane in LDS slot `sl` for the eight voxels
    auto eval_imm = [&](auto svar, float (&T)[8]) {
        constexpr int S = decltype(svar)::value;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            // plain voxels: byte offset of the block's first tap in a slot (from the packed start: three instructions, no register held)
            const char *b = smem_cz + (unsigned)((((a_[k] >> 16) - 8) * P + ((a_[k] & 0xffff) - 8)) * 4);
            float v[4][4];
#pragma unroll
            for (int ky = 0; ky < 4; ky++)
#pragma unroll
                for (int kx = 0; kx < 4; kx++) v[ky][kx] = *reinterpret_cast<const float *>(b + (S * kCzSlot + ky * (P * 4) + kx * 4));
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ky++) {
                float row = v[ky][0] * wx_[k][0];
                row = fmaf(v[ky][1], wx_[k][1], row);
                row = fmaf(v[ky][2], wx_[k][2], row);
                row = fmaf(v[ky][3], wx_[k][3], row);
                acc = ky == 0 ? row * wy_[k][0] : fmaf(row, wy_[k][ky], acc);
            }
            T[k] = acc;
            __builtin_amdgcn_sched_barrier(0);              // one voxel's sixteen reads at a time (all eight at once: 128 registers)
        }
    };
    // waves with an edge voxel (tiles along the array's borders: a third of the tiles of 512^3) and rectangles that do not fit
    // the fixed slot: every tap at its own row / column offset in the rectangle.  The eight offsets of a voxel are computed
    // ONCE (here) and kept as four packed registers -- recomputing them per plane through czrule, as the r4b kernel does, cost
    // 1 340 instructions per plane against the immediate form's 335 and set the time of the whole launch.
    unsigned rop_[8][2], cop_[8][2];
    if (!imm_candidate) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int sy = (a_[k] >> 16) - 8 + by0, sx = (a_[k] & 0xffff) - 8 + bx0;
            unsigned ro[4], co[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // `outside` voxels: anything inside the slot (their value is replaced)
                int ty = czrule(sy + j, q.ny) - by0, tx = czrule(sx + j, q.nx) - bx0;
                ty = min(max(ty, 0), q.ry - 1); tx = min(max(tx, 0), P - 1);
                ro[j] = (unsigned)ty * (unsigned)(P * 4); co[j] = (unsigned)tx * 4u;
            }
            rop_[k][0] = ro[0] | (ro[1] << 16); rop_[k][1] = ro[2] | (ro[3] << 16);
            cop_[k][0] = co[0] | (co[1] << 16); cop_[k][1] = co[2] | (co[3] << 16);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) { rop_[k][0] = rop_[k][1] = cop_[k][0] = cop_[k][1] = 0u; }
    }
    auto eval_general = [&](unsigned pbase, float (&T)[8]) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const unsigned rb[4] = {pbase + (rop_[k][0] & 0xffffu), pbase + (rop_[k][0] >> 16), pbase + (rop_[k][1] & 0xffffu), pbase + (rop_[k][1] >> 16)};
            const unsigned co[4] = {cop_[k][0] & 0xffffu, cop_[k][0] >> 16, cop_[k][1] & 0xffffu, cop_[k][1] >> 16};
            float acc = 0.f;
#pragma unroll
            for (int ky = 0; ky < 4; ky++) {
                float v[4];
#pragma unroll
                for (int kx = 0; kx < 4; kx++) v[kx] = *reinterpret_cast<const float *>(smem_cz + (rb[ky] + co[kx]));
                float row = v[0] * wx_[k][0];
                row = fmaf(v[1], wx_[k][1], row);
                row = fmaf(v[2], wx_[k][2], row);
                row = fmaf(v[3], wx_[k][3], row);
                acc = ky == 0 ? row * wy_[k][0] : fmaf(row, wy_[k][ky], acc);
            }
            T[k] = acc;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const bool imm_ok = imm_candidate;
    auto eval_plane = [&](int pl, float (&T)[8]) {
        const unsigned sl = (unsigned)find(pl);
        if (imm_ok) {
            switch (sl) {
            case 0: eval_imm(std::integral_constant<int, 0>{}, T); break;
            case 1: eval_imm(std::integral_constant<int, 1>{}, T); break;
            case 2: eval_imm(std::integral_constant<int, 2>{}, T); break;
            case 3: eval_imm(std::integral_constant<int, 3>{}, T); break;
            default: eval_imm(std::integral_constant<int, 4>{}, T); break;
            }
        } else {
            eval_general(sl * slot_bytes, T);
        }
    };

    // at most N vector-memory operations of this wave may stay in flight
    auto wait_vm_le = [&](int n) {
        if (n >= 24) asm volatile(MI_VMCNT(24) ::: "memory");
        else if (n >= 18) asm volatile(MI_VMCNT(18) ::: "memory");
        else if (n >= 14) asm volatile(MI_VMCNT(14) ::: "memory");
        else if (n >= 12) asm volatile(MI_VMCNT(12) ::: "memory");
        else if (n >= 10) asm volatile(MI_VMCNT(10) ::: "memory");
        else if (n >= 8) asm volatile(MI_VMCNT(8) ::: "memory");
        else if (n >= 6) asm volatile(MI_VMCNT(6) ::: "memory");
        else if (n >= 4) asm volatile(MI_VMCNT(4) ::: "memory");
        else if (n >= 2) asm volatile(MI_VMCNT(2) ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto in_q = [&](int pl) { return __builtin_amdgcn_readlane(bk, 16 + (pl & 3)) == pl; };
    auto set_tag = [&](int pl) { bk = lane == 16 + (pl & 3) ? pl : bk; };
    // first plane of the four taps of output plane z when they are four plain planes inside the array (else kNone): the cheap
    // form of cubic3_axis for looking ahead -- a wrong guess costs time (the plane is then fetched at its own step), never data
    auto plain_start = [&](int z) {
        const int f = __builtin_amdgcn_readfirstlane(ztab[z].flags);
        const int p1 = __builtin_amdgcn_readfirstlane(ztab[z].pl[1]);
        // (a step whose taps are four plain planes, or -- stream axis of samples -- one plane inside the array)
        const bool ok = (f & 4) != 0 || (q.sident && !(f & 3) && p1 >= 0);
        return ok ? p1 - 1 : kNone;
    };
    const int kfirst = q.sident ? 1 : 0, klast = q.sident ? 1 : 3;      // the taps of a step that exist (one, when the stream axis holds samples)

    ZTaps cur = zplane(zs);
    int pf = zs + 1;                                             // look-ahead: the first step whose planes have not been requested

#pragma unroll 1
    for (int z = zs; z < ze; z++) {
        const bool lds_step = !cur.outside && !cur.cvtap && !(q.dbg & 2);      // the step reads planes from LDS (workgroup-uniform)
        // ---- (1) the planes this step evaluates: fetch what the look-ahead did not bring, then wait for them
        int need_mark = 0;
        bool missing = false;
        unsigned need_slots = 0, newmask = 0;                   // newmask: taps whose plane the register ring does not hold yet
        if (lds_step) {
#pragma unroll
            for (int kz = 0; kz < 4; kz++) {
                const int pl = cur.pl[kz];
                if (in_q(pl)) continue;
                if (!cur.plain) {                                // (four consecutive planes have no twins)
                    bool dup = false;
#pragma unroll
                    for (int j = 0; j < kz; j++) dup = dup || cur.pl[j] == pl;
                    if (dup) continue;
                }
                newmask |= 1u << kz;
                const int sl = find(pl);
                if (sl < 0) { missing = true; continue; }
                need_mark = max(need_mark, mark_at(sl));
                need_slots |= 1u << sl;
            }
        }
        const bool any_need = newmask != 0;
        if (missing) {
            // a plane the look-ahead did not bring (the first step of a chunk, steps at the array ends): nobody may still be
            // reading the slot it goes to -- barrier, stage (into slots that hold no plane of this step), wait for everything,
            // barrier
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int kz = 0; kz < 4; kz++) {
                const int pl = cur.pl[kz];
                if (in_q(pl) || find(pl) >= 0) continue;          // (a duplicate tap finds the plane its twin has just staged)
                int sl = head;
#pragma unroll
                for (int t_ = 0; t_ < 4; t_++) sl = ((need_slots >> sl) & 1u) ? (sl == 4 ? 0 : sl + 1) : sl;
                fetch(pl, sl);
                need_slots |= 1u << sl;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (!missing) {
            if (any_need) wait_vm_le(issued - need_mark);
            __builtin_amdgcn_s_barrier();        // the planes of this step are in LDS for every wave; every wave has left the previous step's reads
        }
        // ---- (2) look ahead: request the planes of the next steps into slots whose plane has been evaluated.  `pf` is the
        // first step whose planes have not been requested: usually one step and one plane per iteration of the z loop.
        ZTaps nxt = cur;
        if (z + 1 < ze) nxt = zplane(z + 1);
        if (pf <= z) pf = z + 1;
        {
            int lo = 0x7fffffff, hi = -0x7fffffff;           // the planes between this step and the step looked at
            if (lds_step) {
                lo = min(min(cur.pl[0], cur.pl[1]), min(cur.pl[2], cur.pl[3]));
                hi = max(max(cur.pl[0], cur.pl[1]), max(cur.pl[2], cur.pl[3]));
            }
#pragma unroll 1
            while (pf < ze && pf <= z + kLook) {
                const int st = plain_start(pf);
                if (st == kNone) { pf++; continue; }                                       // a step at the array ends: fetched at its own top
                {
                    // planes are requested in the order the steps need them: when the tap at the far end of this step is there
                    // (or evaluated), so are the others
                    const int far_pl = st + (q.m00 >= 0.0 ? klast : kfirst);
                    if (in_q(far_pl) || find(far_pl) >= 0) { pf++; continue; }
                }
                lo = min(lo, st + kfirst); hi = max(hi, st + klast);
                bool blocked = false;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (k < kfirst || k > klast || blocked) continue;
                    const int pl = st + k;
                    if (in_q(pl) || find(pl) >= 0) continue;                              // evaluated already, or there
                    const int occ = res_at(head);
                    if (((need_slots >> head) & 1u) ||                                    // the slot is being read in this step
                        (occ != kNone && occ >= lo && occ <= hi && !in_q(occ))) {         // its plane is still to be evaluated
                        blocked = true;
                        continue;
                    }
                    fetch(pl, head);
                }
                if (blocked) break;
                pf++;
            }
        }
        // ---- (3) this step's output
        auto emit = [&](int k, float v) { tile[k * 64 + lane] = v; };
        bool skip_store = false;
        if (cur.outside) {
#pragma unroll
            for (int k = 0; k < 8; k++) emit(k, q.cval);
        } else if (!lds_step || any_far) {
            if (lds_step) {
                // a wave with a far voxel: the same bookkeeping as its neighbours (the tags decide what everybody fetches), no values
#pragma unroll
                for (int kz = 0; kz < 4; kz++) {
                    const int pl = cur.pl[kz];
                    set_tag(pl);
                }
            }
            skip_store = true;                                   // cubic3_zfix_kernel writes these voxels
        } else {
            // (A) the in-plane values of this step's planes that the register ring does not hold yet: one plane per step at a
            // step of up to one plane, four at the start of a chunk
#pragma unroll
            for (int kz = 0; kz < 4; kz++) {
                if (!((newmask >> kz) & 1u)) continue;
                const int pl = cur.pl[kz];
                const int s = pl & 3;
                float T[8];
                eval_plane(pl, T);
                set_tag(pl);
                switch (s) {
                case 0:
#pragma unroll
                    for (int k = 0; k < 8; k++) Q[0][k] = T[k];
                    break;
                case 1:
#pragma unroll
                    for (int k = 0; k < 8; k++) Q[1][k] = T[k];
                    break;
                case 2:
#pragma unroll
                    for (int k = 0; k < 8; k++) Q[2][k] = T[k];
                    break;
                default:
#pragma unroll
                    for (int k = 0; k < 8; k++) Q[3][k] = T[k];
                    break;
                }
            }
            // (B) four values per voxel
            if (cur.plain) {
                // four consecutive planes: tap kz sits in ring slot (pl[0] + kz) & 3 -- one of four rotations, no weights to sort
                auto blend = [&](auto rvar) {
                    constexpr int R = decltype(rvar)::value;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        float acc = Q[R][k] * cur.w[0];
                        acc = fmaf(Q[(R + 1) & 3][k], cur.w[1], acc);
                        acc = fmaf(Q[(R + 2) & 3][k], cur.w[2], acc);
                        acc = fmaf(Q[(R + 3) & 3][k], cur.w[3], acc);
                        emit(k, ((outmask >> k) & 1u) ? q.cval : acc);
                    }
                };
                switch (cur.pl[0] & 3) {
                case 0: blend(std::integral_constant<int, 0>{}); break;
                case 1: blend(std::integral_constant<int, 1>{}); break;
                case 2: blend(std::integral_constant<int, 2>{}); break;
                default: blend(std::integral_constant<int, 3>{}); break;
                }
            } else {
                // the array's ends (a plane read twice at a reflecting end), a stream axis that holds samples: weights per ring
                // slot.  A slot without a plane of this step holds an OLDER plane: it must not enter the sum even with weight
                // 0 (0 x inf = NaN for non-finite coefficients)
                float W0 = 0.f, W1 = 0.f, W2 = 0.f, W3 = 0.f;
                unsigned used = 0;
#pragma unroll
                for (int kz = 0; kz < 4; kz++) {
                    const int s = cur.pl[kz] & 3;
                    const float wk = cur.w[kz];
                    W0 += s == 0 ? wk : 0.f; W1 += s == 1 ? wk : 0.f; W2 += s == 2 ? wk : 0.f; W3 += s == 3 ? wk : 0.f;
                    used |= 1u << s;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    float acc = 0.f;
                    if (used & 1u) acc = fmaf(Q[0][k], W0, acc);
                    if (used & 2u) acc = fmaf(Q[1][k], W1, acc);
                    if (used & 4u) acc = fmaf(Q[2][k], W2, acc);
                    if (used & 8u) acc = fmaf(Q[3][k], W3, acc);
                    emit(k, ((outmask >> k) & 1u) ? q.cval : acc);
                }
            }
        }
        if (skip_store) {
        } else if (wide) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int i = lane >> 4, c = lane & 15;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float4 v = *reinterpret_cast<const float4 *>(tile + (4 * h + i) * 64 + 4 * c);
                typedef float f32x4c __attribute__((ext_vector_type(4)));
                const f32x4c vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<f32x4c *>(out + ((size_t)z * q.oss + (size_t)(y0 + 8 * wave + 4 * h + i) * q.osr + x0 + 4 * c)));
            }
            issued += 2;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            // partial tiles: stores the count does not know (it only under-estimates what may be in flight: see above)
            const int x = x0 + lane;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const int y = y0 + 8 * wave + k;
                const float v = tile[k * 64 + lane];
                if (x < q.ox && y < q.oy) __builtin_nontemporal_store(v, out + ((size_t)z * q.oss + (size_t)y * q.osr + x));
            }
        }
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// r5: map_coordinates, ORDER 3 (its default: interpolation.py:275) on float32 coefficients -- the taps out of a box the
// workgroup sizes from ITS OWN coordinates.  cubic3_f32_kernel gathers through the L1: 3.4 ms on 512^3 against 0.49 ms for order 1
// (the largest cliff left on the path).  As cubic3_box_kernel: a 16^3 cube of output voxels per workgroup; here the bounding
// box of the tap blocks is reduced from the workgroup's coordinates (per thread over its eight voxels, DPP / shuffle per wave, six
// LDS atomics per wave -- only voxels whose three tap blocks are plain blocks inside the array count), staged by LDS-DMA when it
// fits 64 KiB (smooth warps: it does), and the taps come out of LDS.  Everything else -- voxels at the array's faces, non-finite
// coordinates, tiles whose box does not fit (noise, folds) -- goes through the workgroup-wide queue to cubic3_gather.  Tap
// selection, weights, products, order of the sums: cubic3_gather's -- bit-identical to the gather kernel.
// ---------------------------------------------------------------------------------------------------------------------
struct MapBoxParams {
    int nz, ny, nx, oz, oy, ox;
    int mode, npad;
    float cval;
    int dbg;
};

constexpr int kMapBoxBytes = 64 * 1024;

template <typename C>
__global__ void __launch_bounds__(512, 4)
cubic3_mapbox_kernel(const float *__restrict__ in, const C *__restrict__ coords, float *__restrict__ out, const MapBoxParams q)
{
    constexpr int T = kBoxT, RW = 4, WY = 4;
    extern __shared__ __attribute__((aligned(16))) char smem_mb[];
    float *box = reinterpret_cast<float *>(smem_mb);
    int *red = reinterpret_cast<int *>(smem_mb + kMapBoxBytes);              // [0..2] min start, [3..5] max end, [6] queue count
    float *tiles = reinterpret_cast<float *>(smem_mb + kMapBoxBytes + 64);   // [8 waves][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lx = lane & (T - 1), yy = lane / T;
    const int wy = wave % WY, wz = wave / WY;
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T, z0 = blockIdx.z * T;
    const int yrow = RW * wy + yy;
    const int x = x0 + lx, y = y0 + yrow;
    const bool col_ok = x < q.ox && y < q.oy;
    const size_t nvox = (size_t)q.oz * q.oy * q.ox;
    const int nxy = q.ny * q.nx;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, q.nz * nxy * 4, 0x00020000);
    if (tid < 3) red[tid] = 0x7fffffff;
    else if (tid < 6) red[tid] = -0x7fffffff;
    else if (tid == 6) red[6] = 0;
    // ---- this thread's eight coordinates (planes z0 + 8 wz + k) and their tap blocks
    C cz[8], cy[8], cx[8];
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
    unsigned plainmask = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int z = z0 + 8 * wz + k;
        const bool ok = col_ok && z < q.oz;
        const size_t i = ((size_t)(ok ? z : 0) * q.oy + (ok ? y : 0)) * q.ox + (ok ? x : 0);
        cz[k] = __builtin_nontemporal_load(coords + i);
        cy[k] = __builtin_nontemporal_load(coords + nvox + i);
        cx[k] = __builtin_nontemporal_load(coords + 2 * nvox + i);
        const double p0 = (double)cz[k] + (double)q.npad, p1 = (double)cy[k] + (double)q.npad, p2 = (double)cx[k] + (double)q.npad;
        const double f0 = floor(p0), f1 = floor(p1), f2 = floor(p2);
        const bool plain = ok && f0 >= 1.0 && f0 + 2.0 <= (double)(q.nz - 1) && f1 >= 1.0 && f1 + 2.0 <= (double)(q.ny - 1) && f2 >= 1.0 && f2 + 2.0 <= (double)(q.nx - 1);
        if (plain) {
            plainmask |= 1u << k;
            const int s0 = (int)f0 - 1, s1 = (int)f1 - 1, s2 = (int)f2 - 1;
            lo[0] = min(lo[0], s0); hi[0] = max(hi[0], s0 + 3);
            lo[1] = min(lo[1], s1); hi[1] = max(hi[1], s1 + 3);
            lo[2] = min(lo[2], s2); hi[2] = max(hi[2], s2 + 3);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { lo[a] = min(lo[a], __shfl_xor(lo[a], m, 64)); hi[a] = max(hi[a], __shfl_xor(hi[a], m, 64)); }
    __syncthreads();                                         // red[] initialised
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) { atomicMin(&red[a], lo[a]); atomicMax(&red[3 + a], hi[a]); }
    }
    __syncthreads();
    const int b0z = red[0], b0y = red[1], b0x = red[2] & ~3;
    const int bz = red[3] - b0z + 1, by = red[4] - b0y + 1, bx = (red[5] - b0x + 1 + 3) & ~3;
    const bool any_plain = red[3] >= red[0];
    const long long floats = any_plain ? (long long)bz * by * bx : 0;
    const bool fits = any_plain && floats * 4 <= (long long)kMapBoxBytes && !(q.dbg & 2);
    // ---- stage the box
    if (fits) {
        const unsigned cpr = (unsigned)bx >> 2;
        const unsigned nchunks = (unsigned)(floats >> 2);
        const unsigned cpr_magic = (unsigned)((((unsigned long long)1 << 32) + cpr - 1) / cpr);
        const unsigned by_magic = (unsigned)((((unsigned long long)1 << 32) + (unsigned)by - 1) / (unsigned)by);
        const unsigned row_b = (unsigned)q.nx * 4u, plane_b = (unsigned)q.ny * row_b;
        const unsigned base = (unsigned)((b0z * q.ny + b0y) * q.nx + b0x) * 4u;
        const int rounds = (int)((nchunks + 511u) >> 9);
#pragma unroll 1
        for (int j = 0; j < rounds; j++) {
            const unsigned ch = (unsigned)tid + ((unsigned)j << 9);
            // (a box ONE 16-byte chunk wide -- a tile whose only plain voxels touch the array's last column -- has the divisor 1,
            // whose magic number does not fit 32 bits)
            const unsigned row = cpr == 1u ? ch : __umulhi(ch, cpr_magic), c4 = ch - row * cpr;
            const unsigned rz = __umulhi(row, by_magic), ry = row - rz * (unsigned)by;
            const bool ok = ch < nchunks && b0x + 4 * (int)c4 < q.nx;      // (rows and planes of a box of plain blocks lie inside the array)
            const unsigned voff = ok ? rz * plane_b + ry * row_b + c4 * 16u : 0x80000000u;
            cz_dma16(rin, voff, base, __builtin_amdgcn_readfirstlane((unsigned)((wave << 6) + (j << 9)) * 16u), ~0ull);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int plane_f = by * bx;
    float *tile = tiles + wave * 256;
    const bool wide = x0 + T <= q.ox && y0 + T <= q.oy && z0 + T <= q.oz;
    unsigned todo = 0;
#pragma unroll
    for (int bt = 0; bt < 2; bt++) {              // (unrolled: a rolled loop would index the coordinate registers at run time -- scratch)
        float r[4];
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int k = 4 * bt + kk;
            const double p0 = (double)cz[k] + (double)q.npad, p1 = (double)cy[k] + (double)q.npad, p2 = (double)cx[k] + (double)q.npad;
            const double f0 = floor(p0), f1 = floor(p1), f2 = floor(p2);
            // constant mode: a coordinate beyond the array gives cval (cubic3_axis_frac's test)
            const bool out_c = q.mode == MI_MODE_CONSTANT && (p0 < 0.0 || p0 > (double)(q.nz - 1) || p1 < 0.0 || p1 > (double)(q.ny - 1) ||
                                                             p2 < 0.0 || p2 > (double)(q.nx - 1));
            float val = 0.f;
            if (out_c) {
                val = q.cval;
            } else if (fits && ((plainmask >> k) & 1u)) {
                float wz_[4], wy_[4], wx_[4];
                cubic3_weights((float)(p0 - f0), wz_);
                cubic3_weights((float)(p1 - f1), wy_);
                cubic3_weights((float)(p2 - f2), wx_);
                const float *b = box + (((int)f0 - 1 - b0z) * by + ((int)f1 - 1 - b0y)) * bx + ((int)f2 - 1 - b0x);
                float acc = 0.f;
#pragma unroll
                for (int kz = 0; kz < 4; kz++) {
                    float v[4][4];
#pragma unroll
                    for (int ky = 0; ky < 4; ky++) {
                        const float *t = b + kz * plane_f + ky * bx;
#pragma unroll
                        for (int kx = 0; kx < 4; kx++) v[ky][kx] = t[kx];
                    }
#pragma unroll
                    for (int ky = 0; ky < 4; ky++) {
                        const float wzy = wz_[kz] * wy_[ky];
                        float row = v[ky][0] * wx_[0];
                        row = fmaf(v[ky][1], wx_[1], row);
                        row = fmaf(v[ky][2], wx_[2], row);
                        row = fmaf(v[ky][3], wx_[3], row);
                        acc = fmaf(row, wzy, acc);
                    }
                }
                val = acc;
            } else {
                todo |= 1u << k;
            }
            r[kk] = val;
        }
        if (wide) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) tile[kk * 64 + lane] = r[kk];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int i = lane >> 4, c = lane & 15;
            typedef float f32x4m __attribute__((ext_vector_type(4)));
            const f32x4m v = *reinterpret_cast<const f32x4m *>(tile + i * 64 + 4 * c);
            const int orow = y0 + RW * wy + (4 * c) / T, ox4 = x0 + ((4 * c) & (T - 1));
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4m *>(out + ((size_t)(z0 + 8 * wz + 4 * bt + i) * q.oy + orow) * q.ox + ox4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int z = z0 + 8 * wz + 4 * bt + kk;
                if (col_ok && z < q.oz) __builtin_nontemporal_store(r[kk], out + ((size_t)z * q.oy + y) * q.ox + x);
            }
        }
    }
    // ---- second phase (see cubic3_box_kernel): the flagged voxels of the workgroup through a queue, 512 at a time
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned short *queue = reinterpret_cast<unsigned short *>(smem_mb);
    if (todo != 0u) {
#pragma unroll 1
        for (int kl = 0; kl < 8; kl++) {
            if (!((todo >> kl) & 1u)) continue;
            const int slot = atomicAdd(&red[6], 1);
            queue[slot] = (unsigned short)(((8 * wz + kl) << 8) | (yrow << 4) | lx);
        }
    }
    __syncthreads();
    const int nq = red[6];
#pragma unroll 1
    for (int e = tid; e < nq; e += 512) {
        const int id = queue[e];
        const int zz = z0 + (id >> 8), yq = y0 + ((id >> 4) & 15), xq = x0 + (id & 15);
        if (xq >= q.ox || yq >= q.oy || zz >= q.oz) continue;
        const size_t i = ((size_t)zz * q.oy + yq) * q.ox + xq;
        const double c0 = (double)coords[i], c1 = (double)coords[nvox + i], c2 = (double)coords[2 * nvox + i];
        Cubic3 tt;
        bool outside = cubic3_axis(q.nz, nxy, c0, q.mode, q.npad, tt.w[0], tt.off[0]);
        outside |= cubic3_axis(q.ny, q.nx, c1, q.mode, q.npad, tt.w[1], tt.off[1]);
        outside |= cubic3_axis(q.nx, 1, c2, q.mode, q.npad, tt.w[2], tt.off[2]);
        tt.ntap[0] = 4; tt.ntap[1] = 4;
        tt.outside = outside;
        out[i] = cubic3_gather_lean(rin, tt, q.cval);
    }
}

// false = not taken (small outputs, rows that are not whole 16-byte vectors, volumes beyond 32-bit offsets)
bool launch_cubic_mapbox(const float *in, const void *coords, int coords_f64, float *out, const int shape[3], const int oshape[3], int mode, double cval, int npad,
                         hipStream_t s, int *rc, int dbg)
{
    *rc = MI_OK;
    MapBoxParams q;
    q.nz = shape[0]; q.ny = shape[1]; q.nx = shape[2];
    q.oz = oshape[0]; q.oy = oshape[1]; q.ox = oshape[2];
    if ((long long)q.oz * q.oy * q.ox < (1 << 18) || (q.ox & 3) || ((uintptr_t)out & 15) || ((uintptr_t)in & 15) || (q.nx & 3)) return false;
    if ((long long)q.nz * q.ny * q.nx * 4 >= (1LL << 31)) return false;
    q.mode = mode; q.npad = npad; q.cval = (float)cval; q.dbg = dbg & 14;
    const dim3 grid((unsigned)((q.ox + kBoxT - 1) / kBoxT), (unsigned)((q.oy + kBoxT - 1) / kBoxT), (unsigned)((q.oz + kBoxT - 1) / kBoxT));
    if (grid.y > 65535 || grid.z > 65535) return false;
    const size_t lds = kMapBoxBytes + 64 + 8 * 256 * sizeof(float);
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void *)cubic3_mapbox_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void *)cubic3_mapbox_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) { *rc = (int)e; return true; }
        attr_done = true;
    }
    note_kernel("mi::cubic3_mapbox_kernel<%s> grid=%ux%ux%u (order-3 map_coordinates on float32 coefficients: per 16^3 tile the box of its own coordinates staged in LDS)",
                coords_f64 ? "double" : "float", grid.x, grid.y, grid.z);
    if (coords_f64) hipLaunchKernelGGL(cubic3_mapbox_kernel<double>, grid, dim3(512), lds, s, in, (const double *)coords, out, q);
    else hipLaunchKernelGGL(cubic3_mapbox_kernel<float>, grid, dim3(512), lds, s, in, (const float *)coords, out, q);
    const hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) *rc = (int)e2;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// r5: the separable resampling passes of DIAGONAL order-3 transforms (zoom, shift -- the commonest order-3 calls; the
// reference's zoom / shift kernel, _interp_kernels.py:655-688).  The r3 passes (interp.hip) read every tap from memory:
// along x four scalar gathers per output (540 us for 512^3: the L1 serves the overlapping windows of a wave's lanes one
// request at a time), along z four whole-plane reads per output plane (430 us).  Here
//   * x: a wave stages the span of the input row its 256 outputs read in LDS (coalesced loads, four rows in flight) and takes
//     EVERY tap from there, the folded ones at the ends of a row included (540 -> 250 us);
//   * z: a thread walks along z with the four input planes of its window in registers (and the next one on its way): every
//     input plane is read once.
// Products and order of the sums are those of the r3 kernels: bit-identical results (tests/test_gpu_spline_fast.py).  Along z,
// outputs whose taps are not four consecutive planes (array ends: reflected / cval taps) read memory directly, as before.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kRxOut = 256, kRxSpan = 640, kRxRows = 32;      // outputs per wave and row; staged input samples per row; rows per wave

__global__ void __launch_bounds__(256)
cubic_resample_x_lds_kernel(const float *__restrict__ in, float *__restrict__ out, const AxisTaps *__restrict__ tabx, long long nrows, int ox, int nx,
                            float cval)
{
    __shared__ float rowbuf[4][4][kRxSpan];          // [wave][row of the group][sample]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x0 = blockIdx.x * kRxOut;
    const int nout = min(kRxOut, ox - x0);
    // this lane's four outputs: weights and first tap (kept for every row of the wave)
    float w[4][4];
    int off[4][4];
    bool mine[4];
    {
        // the segment's 256 table entries (48 bytes each) through LDS: coalesced dwords in, one entry per output out -- read per
        // lane from memory they were 48 strided loads per lane, more L1 traffic than the sixteen rows of data behind them
        // (the first version of this kernel ran at the r3 pass's 530 us for that reason alone)
        int *tl = reinterpret_cast<int *>(&rowbuf[0][0][0]);
        const int *tg = reinterpret_cast<const int *>(tabx + x0);
        for (int i = threadIdx.x; i < nout * 12; i += 256) tl[i] = tg[i];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int xo = 4 * lane + k;
            mine[k] = xo < nout;
            const int *e = tl + 12 * (mine[k] ? xo : 0);
#pragma unroll
            for (int j = 0; j < 4; j++) { w[k][j] = __int_as_float(e[j]); off[k][j] = e[4 + j]; }
        }
        __syncthreads();                                          // the buffer becomes the waves' row buffers
    }
    // the span of the row the wave's taps lie in: [lo, hi] over every tap that reads the array (cval taps are -1) -- at the ends
    // of a row the taps fold back by a few samples, under the wrapping modes to the other end (the span is then the whole
    // row, which is staged when it fits).  (First version: only the outputs with four consecutive taps read LDS, the others
    // memory -- one dependent round trip per row for the wave that holds the end of a row: 1.7 us per row.)
    int lo = 0x7fffffff, hi = -1;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (mine[k] && off[k][j] >= 0) { lo = min(lo, off[k][j]); hi = max(hi, off[k][j]); }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { lo = min(lo, __shfl_xor(lo, m, 64)); hi = max(hi, __shfl_xor(hi, m, 64)); }
    const int s0 = lo & ~3;                                       // (aligned down: whole dwords anyway, friendlier to the loads)
    const bool staged = hi >= 0 && hi < nx && hi - s0 + 1 <= kRxSpan;
    const int span = staged ? hi - s0 + 1 : 0;
    constexpr int NL = kRxSpan / 64;                              // loads per lane and row at most
    constexpr int G = 4;                                          // rows in flight per wave: with one (1 KiB) the pass ran at the latency of
                                                                  // memory, 2 TB/s -- no faster than the gathers it replaces
    const long long row0 = ((long long)blockIdx.y * 4 + wave) * kRxRows;
    float *buf = rowbuf[wave][0];
#pragma unroll 1
    for (int i0 = 0; i0 < kRxRows; i0 += G) {
        if (row0 + i0 >= nrows) break;
        if (staged) {
            float raw[G][NL];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const long long row = min(row0 + i0 + g, nrows - 1);
                const float *src = in + row * nx + s0;
#pragma unroll
                for (int j = 0; j < NL; j++) raw[g][j] = (64 * j + lane < span) ? src[64 * j + lane] : 0.f;
            }
#pragma unroll
            for (int g = 0; g < G; g++)
#pragma unroll
                for (int j = 0; j < NL; j++) if (64 * j < span) buf[g * kRxSpan + 64 * j + lane] = raw[g][j];
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // the wave's own LDS writes (no other wave touches this buffer)
#pragma unroll
            for (int g = 0; g < G; g++) {
                const long long row = row0 + i0 + g;
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = buf[g * kRxSpan + (off[k][j] < 0 ? 0 : off[k][j] - s0)];
                    float a = (off[k][0] < 0 ? cval : v[0]) * w[k][0];
#pragma unroll
                    for (int j = 1; j < 4; j++) a = fmaf(off[k][j] < 0 ? cval : v[j], w[k][j], a);
                    r[k] = a;
                }
                if (row < nrows) {
                    float *dst = out + row * ox + x0 + 4 * lane;
                    if (mine[3] && (ox & 3) == 0) {
                        typedef float f32x4x __attribute__((ext_vector_type(4)));
                        *reinterpret_cast<f32x4x *>(dst) = f32x4x{r[0], r[1], r[2], r[3]};
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) if (mine[k]) dst[k] = r[k];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // reads done before the next group overwrites the buffer
        } else {
#pragma unroll 1
            for (int g = 0; g < G; g++) {
                const long long row = row0 + i0 + g;
                if (row >= nrows) break;
                const float *src = in + row * nx;
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = src[off[k][j] < 0 ? 0 : off[k][j]];
                    float a = (off[k][0] < 0 ? cval : v[0]) * w[k][0];
#pragma unroll
                    for (int j = 1; j < 4; j++) a = fmaf(off[k][j] < 0 ? cval : v[j], w[k][j], a);
                    r[k] = a;
                }
                float *dst = out + row * ox + x0 + 4 * lane;
                if (mine[3] && (ox & 3) == 0) {
                    typedef float f32x4x __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<f32x4x *>(dst) = f32x4x{r[0], r[1], r[2], r[3]};
                } else {
#pragma unroll
                    for (int k = 0; k < 4; k++) if (mine[k]) dst[k] = r[k];
                }
            }
        }
    }
}

// z pass, four x-consecutive samples per thread: in (nz, d1, 4 d2q) -> out (oz, d1, 4 d2q); the LAST pass of a 3-D transform:
// voxels whose coordinate lies beyond the array along any axis (constant mode) become cval
__global__ void __launch_bounds__(256)
cubic_resample_zstream_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, const AxisTaps *__restrict__ all_tabs, int oz, int oy, int d2q,
                              int nz, float cval, int zchunk)
{
    const int xq = blockIdx.x * 64 + threadIdx.x;
    const int y = __builtin_amdgcn_readfirstlane((int)(blockIdx.y * 4 + threadIdx.y));
    if (y >= oy || xq >= d2q) return;
    const size_t plane = (size_t)oy * d2q;
    const float4 *col = in + (size_t)y * d2q + xq;
    const int out_y = all_tabs[oz + y].outside;
    bool out_x[4];
    {
        const AxisTaps *tx = all_tabs + oz + oy + 4 * xq;
#pragma unroll
        for (int k = 0; k < 4; k++) out_x[k] = tx[k].outside != 0;
    }
    constexpr int kNoBase = -0x40000000;
    float4 R[4] = {};
    float4 ahead = {};                       // plane base + 4, requested one step early
    int base = kNoBase;
    bool have_ahead = false;
    const int z1 = min((int)(blockIdx.z + 1) * zchunk, oz);
#pragma unroll 1
    for (int z = blockIdx.z * zchunk; z < z1; z++) {
        const AxisTaps e = all_tabs[z];          // wave-uniform: scalar loads
        const int o0 = __builtin_amdgcn_readfirstlane(e.off[0]);
        const bool regular = o0 >= 0 && e.off[1] == o0 + 1 && e.off[2] == o0 + 2 && e.off[3] == o0 + 3 && o0 + 3 < nz;
        float4 v[4];
        if (__builtin_amdgcn_readfirstlane((int)regular)) {
            const int d = o0 - base;
            if (d == 0) {
            } else if (d == 1 && have_ahead) {
                R[0] = R[1]; R[1] = R[2]; R[2] = R[3]; R[3] = ahead;
            } else if (d == 2 && have_ahead) {
                R[0] = R[2]; R[1] = R[3]; R[2] = ahead; R[3] = col[(size_t)(o0 + 3) * plane];
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++) R[k] = col[(size_t)(o0 + k) * plane];
            }
            if (d != 0) {
                base = o0;
                have_ahead = o0 + 4 < nz;
                if (have_ahead) ahead = col[(size_t)(o0 + 4) * plane];     // used when the window moves on
            }
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = R[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = col[(size_t)(e.off[k] < 0 ? 0 : e.off[k]) * plane];
        }
        float4 r;
        {
            const bool c0 = e.off[0] < 0;
            r.x = (c0 ? cval : v[0].x) * e.w[0]; r.y = (c0 ? cval : v[0].y) * e.w[0];
            r.z = (c0 ? cval : v[0].z) * e.w[0]; r.w = (c0 ? cval : v[0].w) * e.w[0];
        }
#pragma unroll
        for (int k = 1; k < 4; k++) {
            const bool c = e.off[k] < 0;
            r.x = fmaf(c ? cval : v[k].x, e.w[k], r.x); r.y = fmaf(c ? cval : v[k].y, e.w[k], r.y);
            r.z = fmaf(c ? cval : v[k].z, e.w[k], r.z); r.w = fmaf(c ? cval : v[k].w, e.w[k], r.w);
        }
        const bool ozy = (e.outside | out_y) != 0;
        if (ozy || out_x[0]) r.x = cval;
        if (ozy || out_x[1]) r.y = cval;
        if (ozy || out_x[2]) r.z = cval;
        if (ozy || out_x[3]) r.w = cval;
        out[((size_t)z * oy + y) * (size_t)d2q + xq] = r;
    }
}

int launch_resample_x_lds(const float *in, float *out, const AxisTaps *tabx, long long nrows, int ox, int nx, float cval, hipStream_t s)
{
    const long long gy = (nrows + 4 * kRxRows - 1) / (4 * kRxRows);
    if (gy > 65535) return MI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(cubic_resample_x_lds_kernel, dim3((unsigned)((ox + kRxOut - 1) / kRxOut), (unsigned)gy), dim3(256), 0, s, in, out, tabx, nrows, ox, nx, cval);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int launch_resample_zstream(const float *in, float *out, const AxisTaps *all_tabs, int oz, int oy, int oxq, int nz, float cval, hipStream_t s)
{
    if ((oy + 3) / 4 > 65535) return MI_ERR_UNSUPPORTED;
    // z chunks: enough workgroups for a few waves per SIMD; every chunk starts with four plane loads of its own
    const long long cols = (long long)((oxq + 63) / 64) * ((oy + 3) / 4);
    int nch = (int)std::max<long long>(1, std::min<long long>((4LL * 4 * device_cus() + cols - 1) / cols, (oz + 31) / 32));
    const int zchunk = (oz + nch - 1) / nch;
    nch = (oz + zchunk - 1) / zchunk;
    hipLaunchKernelGGL(cubic_resample_zstream_kernel, dim3((unsigned)((oxq + 63) / 64), (unsigned)((oy + 3) / 4), (unsigned)nch), dim3(64, 4), 0, s,
                       (const float4 *)in, (float4 *)out, all_tabs, oz, oy, oxq, nz, cval, zchunk);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// r5: order-3 affine transforms whose matrix couples all three axes (registration resampling: a few degrees about a general
// axis) -- the taps out of an LDS-staged BOX.  cubic3_f32_kernel gathers 16 x 16 bytes per voxel through the L1 (4.5 ms on
// 512^3, 0.03 of the roofline).  Here a workgroup (512 threads) owns a 16 x 16 x 16 cube of output voxels -- the tile shape whose
// pre-image has the smallest bounding box under a general rotation -- stages that box (+ the three extra samples a cubic tap
// block needs per axis) with LDS-DMA exactly as affine3d_lds_kernel (interp_fast.hip) does for order 1, and reads a voxel's
// 64 taps as 32 ds_read2_b32 from sixteen row addresses.  Tap selection, weights, products and the order of the sums are
// cubic3_gather's (cubic3_f32_kernel's): bit-identical results.  Voxels whose taps are not a plain 4 x 4 x 4 block inside
// the array (coordinates within a sample or two of its faces, folded or cval taps) take cubic3_gather itself in a second
// phase over a workgroup-wide queue; voxels beyond the array (constant mode) are cval at once; matrices whose box exceeds
// 128 KiB (down-scaling by more than ~1.5) stay with the gather kernel.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kBoxBytesMax = 128 * 1024, kBoxRoundsMax = 16; // box budget; staging rounds of 512 chunks.  Up to 64 KiB two workgroups share a CU (1.1-1.6 ms on
                                                            // 512^3, rotations up to ~12 degrees); beyond, one (1.9-2.1 ms up to 45 degrees) -- the gathers take 4.2-5.7 ms there

struct CubBoxParams {
    int nz, ny, nx, oz, oy, ox;
    double m[12];
    int bz, by, bx, nchunks;
    unsigned cpr_magic, by_magic;
    int drow, dc4, drz, dry;
    double cmin[3];
    int mode, npad;
    float cval;
    int dbg;                     // timing ablations (0 in production): 2 = no second phase, 4 = no taps, 8 = no box DMA
};

__global__ void __launch_bounds__(512, 4)          // four waves per SIMD = two workgroups per CU (one computes while the other's box is in flight)
cubic3_box_kernel(const float *__restrict__ in, float *__restrict__ out, const CubBoxParams q)
{
    constexpr int T = kBoxT, RW = 4, WY = 4;                 // lanes: 16 x by 4 rows; waves: 4 along y, 2 along z (8 planes each)
    extern __shared__ __attribute__((aligned(16))) char smem_box[];
    float *box = reinterpret_cast<float *>(smem_box);
    const unsigned box_bytes = ((unsigned)q.nchunks * 16u + 8191u) & ~8191u;
    double (*ptab)[3] = reinterpret_cast<double (*)[3]>(smem_box + box_bytes);        // [T * T][3]: the z and y terms of a row's coordinates
    float *tiles = reinterpret_cast<float *>(smem_box + box_bytes + T * T * 3 * sizeof(double));   // [8 waves][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lx = lane & (T - 1), yy = lane / T;
    const int wy = wave % WY, wz = wave / WY;
    const int x0 = blockIdx.x * T, y0 = blockIdx.y * T, z0 = blockIdx.z * T;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, q.nz * q.ny * q.nx * 4, 0x00020000);
    // ---- box origin: first tap of the smallest coordinate over the tile (a hair below it; + the padding of the coefficient array)
    int b0[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double lo = ((q.m[4 * a] * (double)z0 + q.m[4 * a + 1] * (double)y0) + q.m[4 * a + 2] * (double)x0) + (q.m[4 * a + 3] + q.cmin[a]) + (double)q.npad;
        const int n = a == 0 ? q.nz : (a == 1 ? q.ny : q.nx);
        double f = floor(lo - 1e-6 * (1.0 + fabs(lo))) - 1.0;
        f = f < 0.0 ? 0.0 : (f > (double)(n - 1) ? (double)(n - 1) : f);
        b0[a] = __builtin_amdgcn_readfirstlane(a == 2 ? ((int)f & ~3) : (int)f);
    }
    // ---- stage the box (chunk -> plane, row, 16-byte piece by multiply-high; chunks beyond the volume read zeros)
    {
        const int cpr = q.bx >> 2;
        const int rounds = (q.nchunks + 511) >> 9;
        const unsigned row_b = (unsigned)q.nx * 4u, plane_b = (unsigned)q.ny * row_b;
        const bool inside = b0[0] + q.bz <= q.nz && b0[1] + q.by <= q.ny && b0[2] + q.bx <= q.nx;
        const unsigned base = (unsigned)((b0[0] * q.ny + b0[1]) * q.nx + b0[2]) * 4u;
#pragma unroll
        for (int j = 0; j < kBoxRoundsMax; j++) {
            if (j >= rounds) break;
            const unsigned ch = (unsigned)tid + ((unsigned)j << 9);
            const unsigned row = __umulhi(ch, q.cpr_magic), c4 = ch - row * (unsigned)cpr;
            const unsigned rz = __umulhi(row, q.by_magic), ry = row - rz * (unsigned)q.by;
            bool ok = ch < (unsigned)q.nchunks;
            if (!inside) ok = ok && b0[0] + (int)rz < q.nz && b0[1] + (int)ry < q.ny && b0[2] + 4 * (int)c4 < q.nx;
            const unsigned voff = ok ? rz * plane_b + ry * row_b + c4 * 16u : 0x80000000u;
            if (!(q.dbg & 8)) cz_dma16(rin, voff, base, (unsigned)((wave << 6) + (j << 9)) * 16u, ~0ull);
        }
    }
    for (int e = tid; e < T * T * 3; e += 512) {
        const int rr = e / 3, a = e - 3 * rr;                      // rr = T k + row  <->  plane z0 + k, row y0 + row
        ptab[rr][a] = q.m[4 * a] * (double)(z0 + rr / T) + q.m[4 * a + 1] * (double)(y0 + rr % T);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int yrow = RW * wy + yy;
    const double dx = (double)(x0 + lx);
    const double xz_ = q.m[2] * dx, xy_ = q.m[6] * dx, xx_ = q.m[10] * dx;
    const int plane_f = q.by * q.bx;
    float *tile = tiles + wave * 256;
    const bool wide = x0 + T <= q.ox && y0 + T <= q.oy && z0 + T <= q.oz;
    const int nxy = q.ny * q.nx;
    unsigned todo = 0;                                       // this thread's voxels (bit 4 bt + kk) that are not plain blocks inside the box
    auto k_local = [](int bt, int kk) { return 4 * bt + kk; };
#pragma unroll 1
    for (int bt = 0; bt < 2; bt++) {
        float r[4];
#pragma unroll 1
        for (int kk = 0; kk < 4; kk++) {
            const int k = 8 * wz + 4 * bt + kk;
            const int rr = T * k + yrow;
            // cubic3_f32_kernel's coordinate sums: s = m0 z; s += m1 y; s += m2 x; c = s + m3
            const double c0 = (ptab[rr][0] + xz_) + q.m[3], c1 = (ptab[rr][1] + xy_) + q.m[7], c2 = (ptab[rr][2] + xx_) + q.m[11];
            // the plain case: all three tap blocks inside the array (whatever the mode: no folding there)
            const double p0 = c0 + (double)q.npad, p1 = c1 + (double)q.npad, p2 = c2 + (double)q.npad;
            const double f0 = floor(p0), f1 = floor(p1), f2 = floor(p2);
            const bool plain = f0 >= 1.0 && f0 + 2.0 <= (double)(q.nz - 1) && f1 >= 1.0 && f1 + 2.0 <= (double)(q.ny - 1) && f2 >= 1.0 && f2 + 2.0 <= (double)(q.nx - 1);
            const int sz = (int)f0 - 1 - b0[0], sy = (int)f1 - 1 - b0[1], sx = (int)f2 - 1 - b0[2];
            const bool held = plain && sz >= 0 && sz + 3 < q.bz && sy >= 0 && sy + 3 < q.by && sx >= 0 && sx + 3 < q.bx;
            // constant mode: a coordinate beyond the array gives cval (cubic3_axis_frac's test) -- decided here: whole tiles of a
            // rotated volume's corners lie outside, and through the second phase they cost 3.5 of the launch's 4.5 ms
            const bool out_c = q.mode == MI_MODE_CONSTANT && (p0 < 0.0 || p0 > (double)(q.nz - 1) || p1 < 0.0 || p1 > (double)(q.ny - 1) ||
                                                             p2 < 0.0 || p2 > (double)(q.nx - 1));
            float val;
            if (out_c) {
                val = q.cval;
            } else if (held && (q.dbg & 4)) {
                val = (float)(p0 + p1 + p2);
            } else if (held) {
                float wz_[4], wy_[4], wx_[4];
                cubic3_weights((float)(p0 - f0), wz_);
                cubic3_weights((float)(p1 - f1), wy_);
                cubic3_weights((float)(p2 - f2), wx_);
                const float *b = box + (sz * q.by + sy) * q.bx + sx;
                float acc = 0.f;
#pragma unroll
                for (int kz = 0; kz < 4; kz++) {
                    float v[4][4];
#pragma unroll
                    for (int ky = 0; ky < 4; ky++) {
                        const float *t = b + kz * plane_f + ky * q.bx;
#pragma unroll
                        for (int kx = 0; kx < 4; kx++) v[ky][kx] = t[kx];
                    }
#pragma unroll
                    for (int ky = 0; ky < 4; ky++) {
                        const float wzy = wz_[kz] * wy_[ky];
                        float row = v[ky][0] * wx_[0];
                        row = fmaf(v[ky][1], wx_[1], row);
                        row = fmaf(v[ky][2], wx_[2], row);
                        row = fmaf(v[ky][3], wx_[3], row);
                        acc = fmaf(row, wzy, acc);
                    }
                }
                val = acc;
            } else {
                val = 0.f;
                todo |= 1u << k_local(bt, kk);                       // second phase below
            }
            r[kk] = val;
        }
        if (wide) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) tile[kk * 64 + lane] = r[kk];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int i = lane >> 4, c = lane & 15;                 // plane of the batch, 16-byte chunk of the wave's 64 voxels (4 rows x 16)
            typedef float f32x4b __attribute__((ext_vector_type(4)));
            const f32x4b v = *reinterpret_cast<const f32x4b *>(tile + i * 64 + 4 * c);
            const int orow = y0 + RW * wy + (4 * c) / T, ox4 = x0 + ((4 * c) & (T - 1));
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4b *>(out + ((size_t)(z0 + 8 * wz + 4 * bt + i) * q.oy + orow) * q.ox + ox4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            const int x = x0 + lx, y = y0 + yrow;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int z = z0 + 8 * wz + 4 * bt + kk;
                if (x < q.ox && y < q.oy && z < q.oz) __builtin_nontemporal_store(r[kk], out + ((size_t)z * q.oy + y) * q.ox + x);
            }
        }
    }
    // ---- second phase: the voxels whose taps fold at the array's faces or read cval (a shell of two samples along the six faces:
    // ~2 % of the voxels, spread over a fifth of the tiles) -- cubic3_gather itself.  The flagged voxels of the WORKGROUP are
    // collected in a queue (LDS, where the box was) and worked off 512 at a time: run by the wave that owns them, every wave with
    // one such lane paid the gather routine's full latency up to eight times (1.5 ms of a 2.6 ms launch).
    // Their place in the output was written (as zeros) by OTHER lanes of the wave in the transposed 16-byte stores above: those
    // stores are complete before the values follow.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                         // everybody has left the box
    unsigned short *queue = reinterpret_cast<unsigned short *>(smem_box);
    int *qcount = reinterpret_cast<int *>(tiles);
    if (tid == 0) *qcount = 0;
    __syncthreads();
    if (todo != 0u && !(q.dbg & 2)) {
#pragma unroll 1
        for (int kl = 0; kl < 8; kl++) {
            if (!((todo >> kl) & 1u)) continue;
            const int slot = atomicAdd(qcount, 1);
            queue[slot] = (unsigned short)(((8 * wz + kl) << 8) | (yrow << 4) | lx);
        }
    }
    __syncthreads();
    const int nq = *qcount;
#pragma unroll 1
    for (int e = tid; e < nq; e += 512) {
        const int id = queue[e];
        const int kz_ = id >> 8, ky_ = (id >> 4) & 15, kx_ = id & 15;
        const int z = z0 + kz_, y = y0 + ky_, x = x0 + kx_;
        if (x >= q.ox || y >= q.oy || z >= q.oz) continue;
        const int rr = T * kz_ + ky_;
        const double dxe = (double)x;
        const double c0 = (ptab[rr][0] + q.m[2] * dxe) + q.m[3], c1 = (ptab[rr][1] + q.m[6] * dxe) + q.m[7], c2 = (ptab[rr][2] + q.m[10] * dxe) + q.m[11];
        Cubic3 tt;
        bool outside = cubic3_axis(q.nz, nxy, c0, q.mode, q.npad, tt.w[0], tt.off[0]);
        outside |= cubic3_axis(q.ny, q.nx, c1, q.mode, q.npad, tt.w[1], tt.off[1]);
        outside |= cubic3_axis(q.nx, 1, c2, q.mode, q.npad, tt.w[2], tt.off[2]);
        tt.ntap[0] = 4; tt.ntap[1] = 4;
        tt.outside = outside;
        out[((size_t)z * q.oy + y) * q.ox + x] = cubic3_gather_lean(rin, tt, q.cval);
    }
}

// plan + launch; false = not taken (diagonal / decoupled matrices have better kernels; box too large; small outputs)
bool launch_cubic_box(const float *in, float *out, const int shape[3], const int oshape[3], const double *mat, int mode, double cval, int npad,
                      hipStream_t s, int *rc, int dbg)
{
    *rc = MI_OK;
    CubBoxParams q;
    q.nz = shape[0]; q.ny = shape[1]; q.nx = shape[2];
    q.oz = oshape[0]; q.oy = oshape[1]; q.ox = oshape[2];
    if ((long long)q.oz * q.oy * q.ox < (1 << 18) || (q.ox & 3) || ((uintptr_t)out & 15) || ((uintptr_t)in & 15) || (q.nx & 3)) return false;
    if ((long long)q.nz * q.ny * q.nx * 4 >= (1LL << 31)) return false;
    for (int i = 0; i < 12; i++) { if (!(fabs(mat[i]) < 1e9)) return false; q.m[i] = mat[i]; }
    int dim[3];
    for (int a = 0; a < 3; a++) {
        double ext = 0.0;
        for (int j = 0; j < 3; j++) ext += fabs(q.m[4 * a + j]) * (kBoxT - 1);
        if (!(ext < 4096.0)) return false;
        // taps floor(min - hair) - 1 .. floor(max) + 2: at most floor(ext + hair) + 6 of them
        dim[a] = (int)floor(ext * (1.0 + 1e-6) + 2e-3) + 6;
    }
    dim[2] = (dim[2] + 3 + 3) & ~3;                 // origin aligned down by up to 3, length a multiple of 4
    const int n[3] = {q.nz, q.ny, q.nx};
    for (int a = 0; a < 2; a++) if (dim[a] > n[a]) dim[a] = n[a];
    if (dim[2] > ((n[2] + 3) & ~3) + 4) dim[2] = ((n[2] + 3) & ~3) + 4;
    const long long floats = (long long)dim[0] * dim[1] * dim[2];
    const long long budget = (dbg >> 4) > 0 ? (long long)(dbg >> 4) * 1024 : (long long)kBoxBytesMax;       // test hook: knob >> 4 = budget in KiB
    if (floats * 4 > budget || (floats / 4 + 511) / 512 > kBoxRoundsMax) return false;
    q.bz = dim[0]; q.by = dim[1]; q.bx = dim[2];
    q.nchunks = (int)(floats / 4);
    {
        const unsigned cpr = (unsigned)dim[2] / 4u, by = (unsigned)dim[1];
        q.cpr_magic = (unsigned)((((unsigned long long)1 << 32) + cpr - 1) / cpr);
        q.by_magic = (unsigned)((((unsigned long long)1 << 32) + by - 1) / by);
        q.drow = q.dc4 = q.drz = q.dry = 0;
    }
    for (int a = 0; a < 3; a++) {
        double c = 0.0;
        for (int j = 0; j < 3; j++) { const double e = q.m[4 * a + j] * (kBoxT - 1); if (e < 0.0) c += e; }
        q.cmin[a] = c;
    }
    q.mode = mode; q.npad = npad; q.cval = (float)cval;
    q.dbg = dbg & 14;
    const dim3 grid((unsigned)((q.ox + kBoxT - 1) / kBoxT), (unsigned)((q.oy + kBoxT - 1) / kBoxT), (unsigned)((q.oz + kBoxT - 1) / kBoxT));
    if (grid.y > 65535 || grid.z > 65535) return false;
    const size_t lds = (((size_t)q.nchunks * 16 + 8191) & ~(size_t)8191) + kBoxT * kBoxT * 3 * sizeof(double) + 8 * 256 * sizeof(float);
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        const hipError_t e = hipFuncSetAttribute((const void *)cubic3_box_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) { *rc = (int)e; return true; }
        attr_done = true;
    }
    note_kernel("mi::cubic3_box_kernel grid=%ux%ux%u (order-3 affine on float32 coefficients, all axes coupled: %d x %d x %d box staged per 16^3 tile)", grid.x, grid.y, grid.z,
                q.bz, q.by, q.bx);
    hipLaunchKernelGGL(cubic3_box_kernel, grid, dim3(512), lds, s, in, out, q);
    const hipError_t e2 = hipGetLastError();
    if (e2 != hipSuccess) *rc = (int)e2;
    return true;
}

// launch with the plan of launch_cubic_zstream (interp.hip): same parameters, same dynamic LDS; MI_ERR_UNSUPPORTED = not taken
int launch_cubic_zfactor(int sax, const float *in, float *out, const CubZParams &q, size_t lds, int blocks, hipStream_t s)
{
    const long long tw = (long long)q.ntx * q.nty * 4;
    if (tw > 0x7fffffffLL || (q.oz + kCzFixPlanes - 1) / kCzFixPlanes > 65535) return MI_ERR_UNSUPPORTED;
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)cubic3_zfactor_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        MI_HIP(hipFuncSetAttribute((const void *)cubic3_zfactor_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
        attr_done = true;
    }
    void *flags = nullptr;
    // one block: the far flag of every (tile, wave) + the table of z taps (one record per output plane)
    const size_t flag_bytes = ((size_t)tw * sizeof(int) + 255) & ~(size_t)255;
    int rc = pool_alloc(&flags, flag_bytes + (size_t)q.oz * sizeof(ZRec), s);
    if (rc) return rc;
    ZRec *ztab = reinterpret_cast<ZRec *>((char *)flags + flag_bytes);
    hipLaunchKernelGGL(cubic3_ztaps_kernel, dim3((unsigned)((q.oz + 63) / 64)), dim3(64), 0, s, ztab, q);
    const dim3 fgrid((unsigned)tw, (unsigned)((q.oz + kCzFixPlanes - 1) / kCzFixPlanes));
    if (sax == 0) {
        hipLaunchKernelGGL(cubic3_zfactor_kernel<0>, dim3((unsigned)blocks), dim3(kCzNT), lds, s, in, out, q, (int *)flags, (const ZRec *)ztab);
        hipLaunchKernelGGL(cubic3_zfix_kernel<0>, fgrid, dim3(64), 0, s, in, out, q, (const int *)flags, (const ZRec *)ztab);
    } else {
        hipLaunchKernelGGL(cubic3_zfactor_kernel<1>, dim3((unsigned)blocks), dim3(kCzNT), lds, s, in, out, q, (int *)flags, (const ZRec *)ztab);
        hipLaunchKernelGGL(cubic3_zfix_kernel<1>, fgrid, dim3(64), 0, s, in, out, q, (const int *)flags, (const ZRec *)ztab);
    }
    const hipError_t e = hipGetLastError();
    pool_free(flags);                    // stream-ordered pool: reused only by later work on the stream
    MI_HIP(e);
    return MI_OK;
}

}  // namespace mi
