// sep3d_long.hip -- fused separable 3-D filter for LONG cubic kernels (9..17
// taps per axis, e.g. gaussian sigma=2 -> 17 taps, BASELINE config B; 9 taps =
// config E), float32, ONE launch at the algorithmic 8 B/voxel.
//
// Reference path replaced: gaussian_filter / uniform_filter as three K1 launches
// with fp64 taps from global memory, zero-fill + copy-back per in-place pass
// (cupyimg/scipy/ndimage/filters.py:602-665,725-792, _filters_core.py:148-155).
// Round 1 ran kernels beyond 9 taps as two streaming launches (stream3d.hip,
// 16 B/voxel, latency bound: two 1 KiB loads in flight per wave).
//
// Design (one workgroup of 16 waves per CU, tile = 256 x by TY = 16 y, streaming
// along z over a chunk of planes):
//   * LDS-DMA staging: every raw input row of the tile's (16 + W - 1)-row window
//     goes global -> LDS with `buffer_load_dwordx4 ... lds` (1 KiB per wave
//     instruction, no VGPRs) into a ring of four planes, so two planes (~68 KiB
//     per CU) are in flight while the registers hold the z state.  The 8-float x
//     halo of a row is a second, 16-lane `buffer_load_dword ... lds` whose per-lane
//     source address is already boundary mapped (reflect / mirror / nearest / wrap
//     resolved here).
//   * y pass: wave w owns output row w and reads its W raw rows straight from the
//     staged plane (W lane-contiguous ds_read_b128).
//   * x pass: on that ONE y-filtered row, in registers (DPP lane shifts, packed fp32
//     dot product against host-made weight pairs); the y-filtered halo blocks come
//     from a small LDS table that one wave per plane fills one plane ahead.
//   * z pass: in registers as a scatter -- the x/y-filtered sample is added into
//     W pending output accumulators (68 VGPRs), the oldest one is complete and is
//     stored (non-temporal buffer_store_dwordx4).  The rotation is by unrolling W
//     steps, like the rings of the other kernels.
//   * one s_barrier per plane; DMA completion is counted by hand (s_waitcnt
//     vmcnt(4): the youngest plane stays in flight across the barrier).
// Boundary modes: every index-mapping mode on every axis; `constant` as zero fill
// (what the DMA writes for out-of-range lanes, rows and planes) plus a separable
// coverage correction at the store (HAS_CONST).
#include "long_common.hpp"

namespace mi {

struct LongParams {
    int nx, ny, nz;
    int oy, oz;             // w/2 + origin along y and z (x: W/2)
    int mx, my, mz;         // boundary modes (filter_mode()-normalised, never constant)
    int zc;                 // output planes per chunk
    int nxt, nyt, nzc;      // tile counts
    int tw;                 // tile width in floats (<= 256, multiple of 4)
    // output planes to produce: up to two plane ranges [zb, zb + zn), the first covered by chunks 0 .. nzc0-1, the
    // second by the rest (whole volume: zb0 = 0, zn0 = nz, nzc0 = nzc).  Boundary handling always refers to nz.
    int zb0, zn0, zb1, zn1, nzc0;
    // y / z weights, each one TWICE (an aligned SGPR pair is what v_pk_fma_f32 takes: no s_mov per odd tap)
    float wyv[2 * kStreamMaxTaps], wzv[2 * kStreamMaxTaps];
    float xpair[2][2 * (kStreamMaxTaps / 2 + 2)];   // see StreamParams::xpair
    // constant mode (HAS_CONST kernels): plain x weights, cval, cval * (product of the three weight sums)
    float wxs[kStreamMaxTaps];
    float cval, cval_sum;
    // sep3d_long3_kernel: plain weights padded with a zero (pairs starting at an even tap); wxo[2k] = wx[2k-1],
    // wxo[2k+1] = wx[2k] (pairs starting at an odd tap; wxo[0] = 0)
    float wyp[kStreamMaxTaps + 1], wzp[kStreamMaxTaps + 1], wxe[kStreamMaxTaps + 1], wxo[kStreamMaxTaps + 1];
    int nt;                 // 1 = rows no other workgroup reads are staged non-temporally (long_common.hpp, stream_nt_for)
    int dbg;                // tuning ablations (0 in production): 1 y pass reads one row, 2 no x pass, 4 no z scatter, 8 no DMA, 16 no stores, 32 no halo table,
                            // 64 halo table at the end of the step (r3 kernel), 128 nothing (selects the ablation build)
};

// SAME: the three axes share one weight vector (uniform_filter(size=W), isotropic gaussian_filter): x pair tables
// plus ONE set of 17 splat weights fit the SGPR file for the whole loop.  Otherwise the tables are re-loaded from
// the kernel-argument segment every step (launder()): hoisted, 2 x 17 weights + the pair tables exceed the SGPR
// file and come back as v_readlane spill code.

// ---------------------------------------------------------------------------
// Pass order y, x, z.  The first version of this kernel filtered along x first (window reads out of LDS, result
// written back in place, then the y pass): ablating its phases on 512^3 / 17 taps showed the cost was the LDS
// traffic of the x pass (five window reads + one write-back per raw row, on twice as many rows as are output:
// 162 us of 388) and of the y pass (105 us), not the FMAs (the whole z pass: 22 us); DMA + stores alone ran 230 us.
// Here a wave reads the W raw rows of ITS output row once (y pass, straight from the DMA-staged rows), filters that
// one row along x in registers (lane shifts by DPP, as the streaming passes do) and scatters it along z: 19 LDS
// reads per wave and plane instead of 27 reads and 2 writes, and no x pass on the 16 halo rows (388 -> 302 us
// before the clocks drop).  The y-filtered x HALO of a row (two 4-float blocks per side) is not
// something the wave's own lanes hold: one wave per plane (rotating) filters the halo blocks of all 16 rows in one
// extra pass, one plane ahead, and leaves them in a small LDS table for the edge lanes.
// ---------------------------------------------------------------------------
// x pass over the row a wave holds one float4 per lane of: eL[j] / eR[j] = the (j+1)-th 4-float block left / right of
// the tile (valid in lane 0 / lane `last`)
template <int W>
__device__ __forceinline__ F4 xhops(const float4 v, const float4 (&eL)[2], const float4 (&eR)[2], int lane, int last,
                                    kfloats tab0, kfloats tab1)
{
    constexpr int RX = W / 2;
    constexpr int NBK = (RX + 3) / 4;
    float4 blk[2 * NBK + 1];
    blk[NBK] = v;
    float4 l = v, r = v;
#pragma unroll
    for (int j = 1; j <= NBK; j++) {
        l = dpp4_shr(eL[j - 1], l);
        const float4 rr = dpp4_shl(eR[j - 1], r);
        r = lane == last ? eR[j - 1] : rr;
        blk[NBK - j] = l;
        blk[NBK + j] = r;
    }
    constexpr int NP = 2 * (2 * NBK + 1);
    f32x2 A[NP];
#pragma unroll
    for (int b = 0; b < 2 * NBK + 1; b++) {
        A[2 * b] = (f32x2){blk[b].x, blk[b].y};
        A[2 * b + 1] = (f32x2){blk[b].z, blk[b].w};
    }
    return xdot_tab<W, NP, 4 * NBK - RX>(A, tab0, tab1);
}


template <int W, bool SAME, bool HAS_CONST>
__global__ void __launch_bounds__(kLongTY * 64)
sep3d_long_kernel(const float *__restrict__ in, float *__restrict__ out, const LongParams p)
{
    constexpr int ROWS = kLongTY + W - 1;
    static_assert(W >= 3 && (W & 1) && ROWS <= kLongRowsMax && W / 2 <= 8, "long kernel: odd W, 3..17");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: planes[kLongNB][32] records | hy[2][16][4] float4 | ztab
    constexpr unsigned HY0 = kLongRawBytes;
    int *ztab = reinterpret_cast<int *>(smem + kLongRawBytes + kLongHyBytes);
    float *cztab = reinterpret_cast<float *>(ztab + kLongMaxChunk + kStreamMaxTaps);     // HAS_CONST: z coverage per output plane

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
    const int x0 = xt * p.tw, y0 = yt * kLongTY;
    int zs, ze;
    {
        const bool second = zci >= p.nzc0;
        const int zb = second ? p.zb1 : p.zb0, zn = second ? p.zn1 : p.zn0;
        zs = zb + (second ? zci - p.nzc0 : zci) * p.zc;
        ze = min(zs + p.zc, zb + zn);
    }
    const int ty_act = min(kLongTY, ny - y0);
    const int rows_needed = ty_act + W - 1;
    const int nlanes = min(p.tw >> 2, (nx - x0) >> 2);
    const int last = nlanes - 1;
    const int xe = x0 + 4 * nlanes;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * 4u;
    const int zi0 = zs - p.oz;
    const int nsteps = ze - zs + W - 1;

    for (int i = threadIdx.x; i < nsteps; i += kLongTY * 64) ztab[i] = bmap<int>(zi0 + i, nz, p.mz);
    // Constant mode: the DMA cannot substitute cval, but it zero-fills what lies outside (out-of-range lanes, rows and
    // planes), and  filter(x extended by cval) = filter(x extended by 0) + cval * (S - cz(z) cy(y) cx(x)),  S = product
    // of the three weight sums, c_a(i) = sum of the axis-a weights whose tap lands inside the volume (all of them on
    // an axis that is not in constant mode).  cz per output plane goes to an LDS table, cy is one number per wave,
    // cx four per lane: the correction is two packed FMAs per output float4.
    [[maybe_unused]] float cyv = 0.f;
    [[maybe_unused]] F4 cxv = f4_splat(0.f);
    if constexpr (HAS_CONST) {
        for (int t = threadIdx.x; t < ze - zs; t += kLongTY * 64) {
            float c = 0.f;
            for (int k = 0; k < W; k++) {
                const int q = zs + t - p.oz + k;
                c += (p.mz != MI_MODE_CONSTANT || (q >= 0 && q < nz)) ? p.wzv[2 * k] : 0.f;
            }
            cztab[t] = c;
        }
        float cx4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < W; k++) {
            const int qy = y0 + wave - p.oy + k;
            cyv += (p.my != MI_MODE_CONSTANT || (qy >= 0 && qy < ny)) ? p.wyv[2 * k] : 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int qx = x0 + 4 * lane + c - W / 2 + k;
                cx4[c] += (p.mx != MI_MODE_CONSTANT || (qx >= 0 && qx < nx)) ? p.wxs[k] : 0.f;
            }
        }
        cxv.lo = (f32x2){cx4[0], cx4[1]};
        cxv.hi = (f32x2){cx4[2], cx4[3]};
    }
    __syncthreads();

    unsigned vmain[2], vhalo[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = wave + 16 * h;
        const int ys = bmap<int>(y0 - p.oy + r, ny, p.my);
        const bool valid = r < rows_needed && ys >= 0;          // ys < 0: a row of cval (constant mode) = zeros here
        vmain[h] = (valid && lane < nlanes) ? (unsigned)(ys * nx + x0 + 4 * lane) * 4u : kOOB;
        const int j = lane & 15;
        const int xsrc = bmap<int>(j < 8 ? x0 - 8 + j : xe + j - 8, nx, p.mx);
        vhalo[h] = (valid && xsrc >= 0) ? (unsigned)(ys * nx + xsrc) * 4u : kOOB;
    }
    const unsigned own = (unsigned)wave * kLongRec + (unsigned)lane * 16u;      // this lane's block of record `wave`
    // halo pass (one wave per plane): lane -> (row = lane / 4, block = lane % 4) of the record's 64 halo bytes
    const unsigned hsrc = (unsigned)(lane >> 2) * kLongRec + 1024u + (unsigned)(lane & 3) * 16u;
    // edge lanes' view of the table: lane 0 takes blocks 1 (nearest), 0; lane `last` (and everyone else) blocks 2, 3
    const unsigned hy_near = HY0 + (unsigned)wave * 64u + (lane == 0 ? 16u : 32u);
    const unsigned hy_far = HY0 + (unsigned)wave * 64u + (lane == 0 ? 0u : 48u);
    const unsigned ovoff = (wave < ty_act && lane < nlanes) ? (unsigned)((y0 + wave) * nx + x0 + 4 * lane) * 4u : kOOB;
    constexpr unsigned kPlane = kLongRowsMax * kLongRec;

    auto issue = [&](int i, unsigned bufoff) {
        // plane of step i; beyond the last step the descriptor has zero records (no fetch), so that every step
        // issues the same four DMAs and the vmcnt arithmetic stays uniform.  Scalar work is kept short here: the
        // sixteen waves of the workgroup share ONE scalar unit (the empty loop -- barrier, address arithmetic and
        // DMA issue only -- cost 0.47 us per plane in the first version).
        bool live = i < nsteps;
        int zsrc = zi0 + i;
        if ((unsigned)zsrc >= (unsigned)nz) zsrc = __builtin_amdgcn_readfirstlane(ztab[live ? i : 0]);   // boundary planes only
        if constexpr (HAS_CONST) {
            live = live && zsrc >= 0;           // a plane of cval = a plane of zeros here
            zsrc = max(zsrc, 0);
        }
        const unsigned long long a = (unsigned long long)in + (unsigned long long)(unsigned)zsrc * (unsigned long long)plane_bytes;
        u32x4_t rin;
        rin.x = (unsigned)a;
        rin.y = (unsigned)(a >> 32);       // stride 0: the upper 16 bits of a device address are zero
        rin.z = live ? plane_bytes : 0u;
        rin.w = 0x00020000u;
        if (!(p.dbg & 8)) dma_two_rows(rin, vmain[0], vhalo[0], vmain[1], vhalo[1], bufoff + (unsigned)wave * kLongRec);
    };
    constexpr int kArgBase = 2 * sizeof(void *);
    kfloats wyk = kernarg_floats(kArgBase + offsetof(LongParams, wyv));
    kfloats wzk = SAME ? wyk : kernarg_floats(kArgBase + offsetof(LongParams, wzv));
    kfloats xt0 = kernarg_floats(kArgBase + offsetof(LongParams, xpair));
    kfloats xt1 = xt0 + 2 * (kStreamMaxTaps / 2 + 2);

    // y pass of W consecutive records starting at LDS byte address `at`
    auto ypass = [&](unsigned at) {
        const float4 t0 = *reinterpret_cast<const float4 *>(smem + at);
        F4 yv = f4_scale2((f32x2){wyk[0], wyk[1]}, f4_from(t0));
        if (p.dbg & 1) return yv;
#pragma unroll
        for (int k = 1; k < W; k++) {
            const float4 t = *reinterpret_cast<const float4 *>(smem + at + k * kLongRec);
            yv = f4_fma2((f32x2){wyk[2 * k], wyk[2 * k + 1]}, f4_from(t), yv);
        }
        return yv;
    };

    F4 acc[W];
#pragma unroll
    for (int k = 0; k < W; k++) acc[k] = f4_splat(0.f);

    // prologue: planes 0..2 in flight; plane 0 complete -> its halo table
    issue(0, 0);
    issue(1, kPlane);
    issue(2, 2 * kPlane);
    asm volatile(MI_VMCNT(8) "\n\ts_barrier" ::: "memory");
    if (wave == 15) {
        const F4 hv = ypass(hsrc);
        *reinterpret_cast<float4 *>(smem + HY0 + (unsigned)lane * 16u) = f4_to_float4(hv);
    }

    // Interval i (after barrier i): planes i and i + 1 have landed (each wave waited for its own DMAs of plane i + 1;
    // those of plane i + 2 may be in flight: vmcnt(4)), the halo table of plane i is complete, nobody reads plane i - 1
    // any more: its slot takes plane i + 3.  Then y / x / z of plane i, and one wave makes the halo table of plane i + 1.
    unsigned bi = 0;                    // LDS offset of plane i
    for (int i0 = 0; i0 < nsteps; i0 += W) {
        static_for<W>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                if constexpr (!SAME) { launder(wyk); launder(wzk); launder(xt0); launder(xt1); }
                const unsigned b1 = bi == (kLongNB - 1) * kPlane ? 0u : bi + kPlane;     // plane i + 1
                const unsigned b3 = bi == 0u ? (kLongNB - 1) * kPlane : bi - kPlane;     // slot of plane i - 1
                // Oldest first, this wave has in flight: the 4 DMAs of plane i + 1 (issued two steps ago), the store of
                // step i - 2, the 4 DMAs of plane i + 2 and the store of step i - 1 (vector memory operations of a wave
                // retire in issue order on gfx9).  Plane i + 1 must have landed; plane i + 2 AND the store behind it stay
                // in flight: vmcnt(5) once stores have begun.  (r2 waited vmcnt(4) throughout, i.e. for the first DMA of
                // the plane issued one step earlier: a prefetch distance of one plane, not two -- removing the DMAs
                // altogether saved 86 us of 358 on config B, the waves were stalling on them.)
                if (i >= W && !(p.dbg & 16)) asm volatile(MI_VMCNT(5) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile(MI_VMCNT(4) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue(i + 3, b3);
                const unsigned hyoff = (unsigned)(i & 1) * (kLongHyBytes / 2);
                // ---- y pass of output row `wave`, then its x pass in registers
                const F4 yv = ypass(own + bi);
                float4 eL[2], eR[2];
                {
                    const float4 n = *reinterpret_cast<const float4 *>(smem + hy_near + hyoff);
                    const float4 f = *reinterpret_cast<const float4 *>(smem + hy_far + hyoff);
                    eL[0] = n; eL[1] = f; eR[0] = n; eR[1] = f;
                }
                const F4 xy = (p.dbg & 2) ? yv : xhops<W>(f4_to_float4(yv), eL, eR, lane, last, xt0, xt1);
                // ---- z pass: scatter into the pending outputs; output i - k takes tap k
                acc[J] = f4_scale2((f32x2){wzk[0], wzk[1]}, xy);
                if (!(p.dbg & 4)) {
#pragma unroll
                    for (int k = 1; k < W; k++)
                        acc[(J - k + W) % W] = f4_fma2((f32x2){wzk[2 * k], wzk[2 * k + 1]}, xy, acc[(J - k + W) % W]);
                }
                if (i >= W - 1) {
                    const unsigned long long oa = (unsigned long long)out +
                                                  (unsigned long long)(unsigned)(zs + i - (W - 1)) * (unsigned long long)plane_bytes;
                    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)oa, 0, (int)plane_bytes, 0x00020000);
                    F4 o = acc[(J + 1) % W];
                    if constexpr (HAS_CONST) {
                        const float g = -p.cval * cztab[i - (W - 1)] * cyv;
                        o.lo = fma2(splat2(g), cxv.lo, o.lo + splat2(p.cval_sum));
                        o.hi = fma2(splat2(g), cxv.hi, o.hi + splat2(p.cval_sum));
                    }
                    if (!(p.dbg & 16)) __builtin_amdgcn_raw_buffer_store_b128(f4_to_u32(o), rout, ovoff, 0, 2);
                }
                // ---- halo table of plane i + 1 (the wave changes every plane)
                if (i + 1 < nsteps && wave == (i & 15) && !(p.dbg & 32)) {
                    const F4 hv = ypass(hsrc + b1);
                    *reinterpret_cast<float4 *>(smem + HY0 + (kLongHyBytes / 2 - hyoff) + (unsigned)lane * 16u) = f4_to_float4(hv);
                }
                bi = b1;
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


#ifdef MI_LONG_TUNE
// ---------------------------------------------------------------------------
// r3 (tuning builds, MI_LONG_TUNE): the r2 stream with TWO output rows per wave (8 waves, 256 x 16 tile, same LDS ring and DMA scheme).
// The y pass of rows j and j + 1 reads W + 1 raw rows instead of 2 W (the 17-tap kernel is bound by VALU issue and by
// the LDS reads the y pass waits for: rocprofv3 showed the VALU 61 % busy and ~600 k cycles per CU for ~365 k cycles of
// VALU work; ablations in DESIGN.md): 9.5 instead of 17 ds_read_b128 per output row, twice the independent FMA work
// behind every LDS wait, two waves per SIMD (up to 256 VGPRs: the 2 x W z accumulators take 136).
// ---------------------------------------------------------------------------
template <int W, bool SAME, bool HAS_CONST>
__global__ void __launch_bounds__(512)
sep3d_long2_kernel(const float *__restrict__ in, float *__restrict__ out, const LongParams p)
{
    constexpr int NW = 8;                       // waves; wave w owns output rows 2 w, 2 w + 1
    constexpr int ROWS = kLongTY + W - 1;
    static_assert(W >= 3 && (W & 1) && ROWS <= kLongRowsMax && W / 2 <= 8, "long kernel: odd W, 3..17");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr unsigned HY0 = kLongRawBytes;
    int *ztab = reinterpret_cast<int *>(smem + kLongRawBytes + kLongHyBytes);
    float *cztab = reinterpret_cast<float *>(ztab + kLongMaxChunk + kStreamMaxTaps);

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
    const int x0 = xt * p.tw, y0 = yt * kLongTY;
    int zs, ze;
    {
        const bool second = zci >= p.nzc0;
        const int zb = second ? p.zb1 : p.zb0, zn = second ? p.zn1 : p.zn0;
        zs = zb + (second ? zci - p.nzc0 : zci) * p.zc;
        ze = min(zs + p.zc, zb + zn);
    }
    const int ty_act = min(kLongTY, ny - y0);
    const int rows_needed = ty_act + W - 1;
    const int nlanes = min(p.tw >> 2, (nx - x0) >> 2);
    const int last = nlanes - 1;
    const int xe = x0 + 4 * nlanes;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * 4u;
    const int zi0 = zs - p.oz;
    const int nsteps = ze - zs + W - 1;

    for (int i = threadIdx.x; i < nsteps; i += NW * 64) ztab[i] = bmap<int>(zi0 + i, nz, p.mz);
    [[maybe_unused]] float cyv[2] = {0.f, 0.f};
    [[maybe_unused]] F4 cxv = f4_splat(0.f);
    if constexpr (HAS_CONST) {
        for (int t = threadIdx.x; t < ze - zs; t += NW * 64) {
            float c = 0.f;
            for (int k = 0; k < W; k++) {
                const int q = zs + t - p.oz + k;
                c += (p.mz != MI_MODE_CONSTANT || (q >= 0 && q < nz)) ? p.wzv[2 * k] : 0.f;
            }
            cztab[t] = c;
        }
        float cx4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < W; k++) {
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int qy = y0 + 2 * wave + r - p.oy + k;
                cyv[r] += (p.my != MI_MODE_CONSTANT || (qy >= 0 && qy < ny)) ? p.wyv[2 * k] : 0.f;
            }
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int qx = x0 + 4 * lane + c - W / 2 + k;
                cx4[c] += (p.mx != MI_MODE_CONSTANT || (qx >= 0 && qx < nx)) ? p.wxs[k] : 0.f;
            }
        }
        cxv.lo = (f32x2){cx4[0], cx4[1]};
        cxv.hi = (f32x2){cx4[2], cx4[3]};
    }
    __syncthreads();

    // raw rows this wave stages per plane: wave, wave + 8, wave + 16, wave + 24
    unsigned vmain[4], vhalo[4];
#pragma unroll
    for (int h = 0; h < 4; h++) {
        const int r = wave + NW * h;
        const int ys = bmap<int>(y0 - p.oy + r, ny, p.my);
        const bool valid = r < rows_needed && ys >= 0;
        vmain[h] = (valid && lane < nlanes) ? (unsigned)(ys * nx + x0 + 4 * lane) * 4u : kOOB;
        const int j = lane & 15;
        const int xsrc = bmap<int>(j < 8 ? x0 - 8 + j : xe + j - 8, nx, p.mx);
        vhalo[h] = (valid && xsrc >= 0) ? (unsigned)(ys * nx + xsrc) * 4u : kOOB;
    }
    const unsigned own = (unsigned)(2 * wave) * kLongRec + (unsigned)lane * 16u;    // first raw record of output row 2 w
    const unsigned hsrc = (unsigned)(lane >> 2) * kLongRec + 1024u + (unsigned)(lane & 3) * 16u;
    unsigned hy_near[2], hy_far[2], ovoff[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int j = 2 * wave + r;
        hy_near[r] = HY0 + (unsigned)j * 64u + (lane == 0 ? 16u : 32u);
        hy_far[r] = HY0 + (unsigned)j * 64u + (lane == 0 ? 0u : 48u);
        ovoff[r] = (j < ty_act && lane < nlanes) ? (unsigned)((y0 + j) * nx + x0 + 4 * lane) * 4u : kOOB;
    }
    constexpr unsigned kPlane = kLongRowsMax * kLongRec;

    auto issue = [&](int i, unsigned bufoff) {
        bool live = i < nsteps;
        int zsrc = zi0 + i;
        if ((unsigned)zsrc >= (unsigned)nz) zsrc = __builtin_amdgcn_readfirstlane(ztab[live ? i : 0]);
        if constexpr (HAS_CONST) {
            live = live && zsrc >= 0;
            zsrc = max(zsrc, 0);
        }
        const unsigned long long a = (unsigned long long)in + (unsigned long long)(unsigned)zsrc * (unsigned long long)plane_bytes;
        u32x4_t rin;
        rin.x = (unsigned)a;
        rin.y = (unsigned)(a >> 32);
        rin.z = live ? plane_bytes : 0u;
        rin.w = 0x00020000u;
        if (!(p.dbg & 8)) {
            dma_two_rows(rin, vmain[0], vhalo[0], vmain[2], vhalo[2], bufoff + (unsigned)wave * kLongRec);            // rows w, w + 16
            dma_two_rows(rin, vmain[1], vhalo[1], vmain[3], vhalo[3], bufoff + (unsigned)(wave + NW) * kLongRec);     // rows w + 8, w + 24
        }
    };
    constexpr int kArgBase = 2 * sizeof(void *);
    kfloats wyk = kernarg_floats(kArgBase + offsetof(LongParams, wyv));
    kfloats wzk = SAME ? wyk : kernarg_floats(kArgBase + offsetof(LongParams, wzv));
    kfloats xt0 = kernarg_floats(kArgBase + offsetof(LongParams, xpair));
    kfloats xt1 = xt0 + 2 * (kStreamMaxTaps / 2 + 2);

    // y pass of ONE row (the halo table): W consecutive records starting at LDS byte address `at`
    auto ypass = [&](unsigned at) {
        const float4 t0 = *reinterpret_cast<const float4 *>(smem + at);
        F4 yv = f4_scale2((f32x2){wyk[0], wyk[1]}, f4_from(t0));
#pragma unroll
        for (int k = 1; k < W; k++) {
            const float4 t = *reinterpret_cast<const float4 *>(smem + at + k * kLongRec);
            yv = f4_fma2((f32x2){wyk[2 * k], wyk[2 * k + 1]}, f4_from(t), yv);
        }
        return yv;
    };
    // y pass of rows j and j + 1 from the W + 1 records starting at `at`: raw row k is tap k of row j, tap k - 1 of row j + 1
    auto ypass2 = [&](unsigned at, F4 &ya, F4 &yb) {
        const float4 t0 = *reinterpret_cast<const float4 *>(smem + at);
        ya = f4_scale2((f32x2){wyk[0], wyk[1]}, f4_from(t0));
        if (p.dbg & 1) { yb = ya; return; }
#pragma unroll
        for (int k = 1; k < W; k++) {
            const F4 t = f4_from(*reinterpret_cast<const float4 *>(smem + at + k * kLongRec));
            ya = f4_fma2((f32x2){wyk[2 * k], wyk[2 * k + 1]}, t, ya);
            if (k == 1) yb = f4_scale2((f32x2){wyk[0], wyk[1]}, t);
            else yb = f4_fma2((f32x2){wyk[2 * k - 2], wyk[2 * k - 1]}, t, yb);
        }
        const F4 t = f4_from(*reinterpret_cast<const float4 *>(smem + at + W * kLongRec));
        yb = f4_fma2((f32x2){wyk[2 * W - 2], wyk[2 * W - 1]}, t, yb);
    };

    F4 acc[2][W];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int k = 0; k < W; k++) acc[r][k] = f4_splat(0.f);

    issue(0, 0);
    issue(1, kPlane);
    issue(2, 2 * kPlane);
    asm volatile(MI_VMCNT(16) "\n\ts_barrier" ::: "memory");
    if (wave == NW - 1) {
        const F4 hv = ypass(hsrc);
        *reinterpret_cast<float4 *>(smem + HY0 + (unsigned)lane * 16u) = f4_to_float4(hv);
    }

    unsigned bi = 0;
    for (int i0 = 0; i0 < nsteps; i0 += W) {
        static_for<W>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                if constexpr (!SAME) { launder(wyk); launder(wzk); launder(xt0); launder(xt1); }
                const unsigned b1 = bi == (kLongNB - 1) * kPlane ? 0u : bi + kPlane;
                const unsigned b3 = bi == 0u ? (kLongNB - 1) * kPlane : bi - kPlane;
                // in flight, oldest first: 8 DMAs of plane i + 1, 2 stores, 8 DMAs of plane i + 2, 2 stores (see the
                // one-row kernel): plane i + 1 must have landed, the rest stays in flight
                if (i >= W && !(p.dbg & 16)) asm volatile(MI_VMCNT(10) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile(MI_VMCNT(8) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue(i + 3, b3);
                const unsigned hyoff = (unsigned)(i & 1) * (kLongHyBytes / 2);
                F4 yv[2];
                ypass2(own + bi, yv[0], yv[1]);
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    float4 eL[2], eR[2];
                    {
                        const float4 n = *reinterpret_cast<const float4 *>(smem + hy_near[r] + hyoff);
                        const float4 f = *reinterpret_cast<const float4 *>(smem + hy_far[r] + hyoff);
                        eL[0] = n; eL[1] = f; eR[0] = n; eR[1] = f;
                    }
                    const F4 xy = (p.dbg & 2) ? yv[r] : xhops<W>(f4_to_float4(yv[r]), eL, eR, lane, last, xt0, xt1);
                    acc[r][J] = f4_scale2((f32x2){wzk[0], wzk[1]}, xy);
                    if (!(p.dbg & 4)) {
#pragma unroll
                        for (int k = 1; k < W; k++)
                            acc[r][(J - k + W) % W] = f4_fma2((f32x2){wzk[2 * k], wzk[2 * k + 1]}, xy, acc[r][(J - k + W) % W]);
                    }
                }
                if (i >= W - 1) {
                    const unsigned long long oa = (unsigned long long)out +
                                                  (unsigned long long)(unsigned)(zs + i - (W - 1)) * (unsigned long long)plane_bytes;
                    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)oa, 0, (int)plane_bytes, 0x00020000);
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        F4 o = acc[r][(J + 1) % W];
                        if constexpr (HAS_CONST) {
                            const float g = -p.cval * cztab[i - (W - 1)] * cyv[r];
                            o.lo = fma2(splat2(g), cxv.lo, o.lo + splat2(p.cval_sum));
                            o.hi = fma2(splat2(g), cxv.hi, o.hi + splat2(p.cval_sum));
                        }
                        if (!(p.dbg & 16)) __builtin_amdgcn_raw_buffer_store_b128(f4_to_u32(o), rout, ovoff[r], 0, 2);
                    }
                }
                if (i + 1 < nsteps && wave == (i & (NW - 1)) && !(p.dbg & 32)) {
                    const F4 hv = ypass(hsrc + b1);
                    *reinterpret_cast<float4 *>(smem + HY0 + (kLongHyBytes / 2 - hyoff) + (unsigned)lane * 16u) = f4_to_float4(hv);
                }
                bi = b1;
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif

// ---------------------------------------------------------------------------
// r3 kernel (`sep3d_long3_kernel`): same tile, ring, DMA and barrier scheme; what changed is the instruction stream,
// after the ablations showed the 17-tap kernel bound by VALU issue (158 VALU instructions per wave and plane for 102
// FMAs) with the VALU idle while the four waves of a SIMD all wait for the y-pass LDS reads right after the barrier.
//   * Software pipelining: in step i a wave runs the x and z passes of plane i on the y-filtered row it made one
//     step EARLIER, and in the same basic block the y pass of plane i + 1 (which barrier i already certifies as
//     landed): the 17 LDS reads travel under 70 FMAs of independent work instead of in front of them.
//   * x pass as 2 x (W + 1) packed FMAs with operand selection (op_sel) instead of a 2-vector dot product:
//     (out0, out1) += (win[q], win[q]) * (w[t], w[t-1]),  (out2, out3) += (win[q], win[q]) * (w[t-2], w[t-3]),  t = q - BASE.
//     The broadcast of win[q] and the swap of an aligned weight pair are free (op_sel / op_sel_hi), so there is no
//     horizontal add and no register shuffling, and the weight pairs are the plain weights (pairs starting at an
//     even tap) plus ONE shifted copy (pairs starting at an odd tap): 36 SGPRs for all three passes when the axes
//     share a kernel, instead of 34 doubled weights + 40 pair-table entries that had to be re-loaded per phase.
//   * The DPP lane shifts take the left halo blocks straight from their own LDS reads as the `old` operand (no
//     v_mov per shifted register) and the right-hand shifts leave lane 63 undefined (the select for lane `last`
//     follows anyway).
//   * The ablation flags are a template parameter: the production instance is straight-line code.
// Index-mapping boundary modes only: constant mode (zero fill + correction terms, five more live registers) stays on
// sep3d_long_kernel, whose register budget has room for them.
// ---------------------------------------------------------------------------
template <int I> __device__ __forceinline__ float f4_comp(const float4 &v)
{
    if constexpr (I == 0) return v.x;
    else if constexpr (I == 1) return v.y;
    else if constexpr (I == 2) return v.z;
    else return v.w;
}

__device__ __forceinline__ float dpp_from_right_any(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /* wave_shl:1 */, 0xf, 0xf, true));
}

// (w[t], w[t-1]) out of te = plain weights (te[W] = 0) and to[2k] = w[2k-1], to[2k+1] = w[2k] (to[0] = 0)
template <int T> __device__ __forceinline__ f32x2 xpair_rev(kfloats te, kfloats to)
{
    if constexpr (T & 1) return (f32x2){te[T], te[T - 1]};
    else return (f32x2){to[T + 1], to[T]};
}

// x pass in two stages, so that the caller can place LDS reads between them and the register peak stays low:
// `left` consumes the blocks left of the lane's own float4 (oL[j] = the (j+1)-th block left of the tile, what lane 0
// takes instead of a neighbour's), `right` the own block and the blocks to the right (eR[j], for lane `last`).
// Window float q = BASE + t feeds (out0, out1) with (w[t], w[t-1]) and (out2, out3) with (w[t-2], w[t-3]).
template <int W>
struct XPass3 {
    static constexpr int RX = W / 2;
    static constexpr int NBK = (RX + 3) / 4;
    static constexpr int BASE = 4 * NBK - RX;
    f32x2 P0, P1;

    template <int Q>
    __device__ __forceinline__ void feed(float wq, kfloats te, kfloats to)
    {
        constexpr int t = Q - BASE;
        if constexpr (t >= 0 && t <= W + 2) {
            const f32x2 ws = splat2(wq);
            // The first and the last sample of a pair of outputs' windows belong to ONE of the two outputs: a packed FMA
            // against (w[0], 0) / (0, w[W-1]) would multiply the other's accumulator with 0 x sample -- a NaN when the
            // sample is not finite, one voxel beyond the taps, where an explicit sum over the taps stays finite.  Those
            // four updates are scalar (same instruction count; r4b, found with non-finite samples in the volume).
            if constexpr (t <= W) {
                if constexpr (t == 0) P0 = (f32x2){wq * te[0], 0.f};
                else if constexpr (t == W) P0.y = __builtin_fmaf(wq, te[W - 1], P0.y);
                else P0 = fma2(ws, xpair_rev<t>(te, to), P0);
            }
            if constexpr (t >= 2) {
                if constexpr (t == 2) P1 = (f32x2){wq * te[0], 0.f};
                else if constexpr (t == W + 2) P1.y = __builtin_fmaf(wq, te[W - 1], P1.y);
                else P1 = fma2(ws, xpair_rev<t - 2>(te, to), P1);
            }
        }
    }
    template <int B>
    __device__ __forceinline__ void feed_block(const float4 &v, kfloats te, kfloats to)
    {
        feed<4 * B>(v.x, te, to);
        feed<4 * B + 1>(v.y, te, to);
        feed<4 * B + 2>(v.z, te, to);
        feed<4 * B + 3>(v.w, te, to);
    }
    // the whole window at once: blk[b] = the b-th 4-float block of the 2 NBK + 1 around (and including) the lane's own
    __device__ __forceinline__ F4 window(const float4 (&blk)[2 * NBK + 1], kfloats te, kfloats to)
    {
        static_for<2 * NBK + 1>([&](auto BB) {
            constexpr int b = decltype(BB)::value;
            feed_block<b>(blk[b], te, to);
        });
        F4 o;
        o.lo = P0;
        o.hi = P1;
        return o;
    }
    __device__ __forceinline__ void left(const float4 v, const float4 (&oL)[2], kfloats te, kfloats to)
    {
        float4 l[NBK];
        float4 c = v;
#pragma unroll
        for (int j = 0; j < NBK; j++) {
            c = dpp4_shr(oL[j], c);
            l[j] = c;
        }
        // taps in ascending order: farthest block first
        static_for<NBK>([&](auto BB) {
            constexpr int bq = decltype(BB)::value;
            feed_block<bq>(l[NBK - 1 - bq], te, to);
        });
    }
    __device__ __forceinline__ F4 right(const float4 v, const float4 (&eR)[2], int lane, int last, kfloats te, kfloats to)
    {
        feed_block<NBK>(v, te, to);
        float4 c = v;
        static_for<NBK>([&](auto BB) {
            constexpr int j = decltype(BB)::value;
            const float4 rr = make_float4(dpp_from_right_any(c.x), dpp_from_right_any(c.y), dpp_from_right_any(c.z), dpp_from_right_any(c.w));
            c = lane == last ? eR[j] : rr;
            feed_block<NBK + 1 + j>(c, te, to);
        });
        F4 o;
        o.lo = P0;
        o.hi = P1;
        return o;
    }
};

// CFG (tuning, see launch_long): bits 0-2 first y read group, bits 3-6 end of the second group, bit 7 halo table at
// the end of the step instead of the start, bit 8 no priority raise for the wave that makes the halo table, bit 9 x pass through LDS; 0 = the
// defaults below.
// WZ: taps along z (the streamed axis); W: taps along y and x.  WZ != W serves volumes with anisotropic voxels
// (gaussian sigma given in millimetres: fewer taps through the slices), where the in-plane kernels agree.
// RG (r6): rows of any length >= 16 floats.  The LDS-DMA takes 16-byte records from addresses that are only 4-byte aligned
// (probed: scratch/unal_dma.hip), so a row is staged from wherever it starts; the LAST lane of a row's last x tile then holds
// `tail` = 1 .. 3 floats of its row followed by the head of the next row (zeros beyond the plane): before the x pass those
// floats are replaced by the first y-filtered halo floats right of the row -- the halo DMA starts at the row's true end -- and
// the two right-hand halo blocks move up by 4 - tail floats; the lane stores `tail` floats (one 8-byte and one 4-byte store
// that every lane issues with an out-of-range offset, so that the vmcnt arithmetic stays uniform: three stores per step).
template <int W, bool SAME, bool DBG, int CFG = 0, int WZ = W, bool RG = false>
__global__ void __launch_bounds__(kLongTY * 64)
sep3d_long3_kernel(const float *__restrict__ in, float *__restrict__ out, const LongParams p)
{
    static_assert(!RG || (CFG == 0 && !DBG), "the ragged build is the production variant only");
    constexpr int ROWS = kLongTY + W - 1;
    static_assert(W >= 3 && (W & 1) && ROWS <= kLongRowsMax && W / 2 <= 8, "long kernel: odd W, 3..17");
    static_assert(WZ >= 3 && (WZ & 1) && WZ <= kLongRowsMax / 2 + 1 && (WZ == W || !SAME), "long kernel: odd WZ, 3..17");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr unsigned HY0 = kLongRawBytes;
    int *ztab = reinterpret_cast<int *>(smem + kLongRawBytes + kLongHyBytes);
    const int dbg = DBG ? p.dbg : 0;
    // CFG bit 9: the x pass through LDS.  The pipelined y pass never reads plane i in step i, so three ring slots do; the
    // fourth holds, twice (step parity), one record per wave: [8 halo floats][the wave's y-filtered row][8 halo floats].
    // A wave stores its row, the wave that makes the halo table stores the halo blocks there, and the x window is five
    // aligned 16-byte reads -- no lane shifts, no edge selects (24 + copies of the 136 VALU instructions per step).
    constexpr bool kXlds = ((CFG >> 9) & 1) != 0;
    constexpr int kSlots = kXlds ? 3 : kLongNB;
    constexpr unsigned kXrow0 = 3u * kLongRowsMax * kLongRec;                   // byte offset of the x records (kXlds)
    constexpr unsigned kXpar = (unsigned)kLongTY * kLongRec;                    // bytes per parity

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
    const int x0 = xt * p.tw, y0 = yt * kLongTY;
    int zs, ze;
    {
        const bool second = zci >= p.nzc0;
        const int zb = second ? p.zb1 : p.zb0, zn = second ? p.zn1 : p.zn0;
        zs = zb + (second ? zci - p.nzc0 : zci) * p.zc;
        ze = min(zs + p.zc, zb + zn);
    }
    const int ty_act = min(kLongTY, ny - y0);
    const int rows_needed = ty_act + W - 1;
    const int width = min(p.tw, nx - x0);                                   // RG: floats of the row in this tile
    const int nlanes = RG ? (width + 3) >> 2 : min(p.tw >> 2, (nx - x0) >> 2);
    const int last = nlanes - 1;
    const int tail = RG ? width - 4 * last : 4;                             // floats of its row the last lane holds
    const int xe = RG ? x0 + width : x0 + 4 * nlanes;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * 4u;
    const int zi0 = zs - p.oz;
    const int nsteps = ze - zs + WZ - 1;

    for (int i = threadIdx.x; i < nsteps; i += kLongTY * 64) ztab[i] = bmap<int>(zi0 + i, nz, p.mz);
    __syncthreads();

    unsigned vmain[2], vhalo[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = wave + 16 * h;
        const int ys = bmap<int>(y0 - p.oy + r, ny, p.my);
        const bool valid = r < rows_needed;
        // (r5: `constant` with a zero fill value -- a row, a halo column or (below) a plane beyond the array is simply not
        // fetched: what the DMA leaves in LDS for an out-of-range lane is the zero the mode prescribes)
        vmain[h] = (valid && ys >= 0 && lane < nlanes) ? (unsigned)(ys * nx + x0 + 4 * lane) * 4u : kOOB;
        const int j = lane & 15;
        const int xsrc = bmap<int>(j < 8 ? x0 - 8 + j : xe + j - 8, nx, p.mx);
        vhalo[h] = (valid && ys >= 0 && xsrc >= 0) ? (unsigned)(ys * nx + xsrc) * 4u : kOOB;
    }
    const unsigned own = (unsigned)wave * kLongRec + (unsigned)lane * 16u;      // this lane's block of record `wave`
    // halo pass (one wave per plane): lane -> (row = lane / 4, block = lane % 4) of the record's 64 halo bytes
    const unsigned hsrc = (unsigned)(lane >> 2) * kLongRec + 1024u + (unsigned)(lane & 3) * 16u;
    // this row's four y-filtered halo blocks in the table: far left, near left, near right, far right
    const unsigned hy_row = HY0 + (unsigned)wave * 64u;
    // kXlds: where lane (row = lane / 4, block = lane % 4) of the halo pass stores: blocks 0, 1 in front of the row, blocks
    // 2, 3 right behind its last valid float4
    const unsigned hx_dst = kXrow0 + (unsigned)(lane >> 2) * kLongRec +
                            ((lane & 3) < 2 ? (unsigned)(lane & 3) * 16u : 32u + 16u * (unsigned)nlanes + (unsigned)((lane & 3) - 2) * 16u);
    const unsigned xr_own = kXrow0 + (unsigned)wave * kLongRec + (unsigned)lane * 16u;     // block -2 of this lane's window
    const bool rg_last = RG && lane == last && tail < 4;                     // stores `tail` floats in pieces
    const bool rg_p1 = rg_last && tail < 2, rg_p2 = rg_last && tail < 3, rg_p3 = rg_last;     // floats 1 / 2 / 3 of its block are the row's continuation
    const unsigned ovoff0 = (unsigned)((y0 + wave) * nx + x0 + 4 * lane) * 4u;
    const unsigned ovoff = (wave < ty_act && lane < nlanes && !rg_last) ? ovoff0 : kOOB;
    const unsigned ovoff2 = (wave < ty_act && rg_last && (tail & 2)) ? ovoff0 : kOOB;                           // floats 0, 1
    const unsigned ovoff1 = (wave < ty_act && rg_last && (tail & 1)) ? ovoff0 + ((tail & 2) ? 8u : 0u) : kOOB;  // float 2 or 0
    constexpr unsigned kPlane = kLongRowsMax * kLongRec;

    auto issue = [&](int i, unsigned bufoff) {
        // plane of step i; beyond the last step the descriptor has zero records (no fetch), so that every step issues
        // the same four DMAs and the vmcnt arithmetic stays uniform
        const bool live = i < nsteps;
        int zsrc = zi0 + i;
        if ((unsigned)zsrc >= (unsigned)nz) zsrc = __builtin_amdgcn_readfirstlane(ztab[live ? i : 0]);   // boundary planes only
        const unsigned long long a = (unsigned long long)in + (unsigned long long)(unsigned)max(zsrc, 0) * (unsigned long long)plane_bytes;
        u32x4_t rin;
        rin.x = (unsigned)a;
        rin.y = (unsigned)(a >> 32);       // stride 0: the upper 16 bits of a device address are zero
        rin.z = (live && zsrc >= 0) ? plane_bytes : 0u;
        rin.w = 0x00020000u;
        // staged rows W - 1 .. 15 of the tile are read by no other workgroup (the neighbours' windows end at row W - 2 /
        // start at row 16): non-temporal (long_common.hpp); chunk-ramp planes are re-read by the next z chunk, rarely
        // (up to 9 taps: the kernels that wait for memory; the longer ones are bound by what a wave issues, and the second
        // copy of the DMA statement cost the 17-tap kernel 6 %)
        if constexpr (W <= 9) {
            if (!(dbg & 8)) dma_two_rows(rin, vmain[0], vhalo[0], vmain[1], vhalo[1], bufoff + (unsigned)wave * kLongRec, p.nt && wave >= W - 1);
        } else {
            if (!(dbg & 8)) dma_two_rows(rin, vmain[0], vhalo[0], vmain[1], vhalo[1], bufoff + (unsigned)wave * kLongRec);
        }
    };
    constexpr int kArgBase = 2 * sizeof(void *);
    kfloats wyk = kernarg_floats(kArgBase + offsetof(LongParams, wyp));
    kfloats wzk = SAME ? wyk : kernarg_floats(kArgBase + offsetof(LongParams, wzp));
    kfloats xte = SAME ? wyk : kernarg_floats(kArgBase + offsetof(LongParams, wxe));
    kfloats xto = kernarg_floats(kArgBase + offsetof(LongParams, wxo));

    // y pass of W consecutive records starting at LDS byte address `at`
    auto ypass = [&](unsigned at) {
        const float4 t0 = *reinterpret_cast<const float4 *>(smem + at);
        F4 yv = f4_scale(wyk[0], f4_from(t0));
        if (dbg & 1) return yv;
#pragma unroll
        for (int k = 1; k < W; k++) {
            const float4 t = *reinterpret_cast<const float4 *>(smem + at + k * kLongRec);
            yv = f4_fma(wyk[k], f4_from(t), yv);
        }
        return yv;
    };

    // the same for the wave that makes the halo table on top of its own step: all reads in flight as early as the
    // registers allow (9, then 4 more as each 4 are consumed) -- what this pass costs the workgroup is its latency
    auto ypass_batched = [&](unsigned at) {
        constexpr int H0 = DBG ? 5 : 9, H = W < H0 ? W : H0;
        float4 t[W];
        const int rows = (dbg & 1) ? 1 : W;
        const char *src = smem + at;
#pragma unroll
        for (int k = 0; k < H; k++) if (k < rows) t[k] = *reinterpret_cast<const float4 *>(src + k * kLongRec);
        __builtin_amdgcn_sched_barrier(0);
        F4 hv = f4_scale(wyk[0], f4_from(t[0]));
        static_for<(W + 3) / 4>([&](auto GG) {
            constexpr int g = decltype(GG)::value;
#pragma unroll
            for (int k = 4 * g; k < 4 * g + 4 && k < W; k++) if (k >= 1 && k < rows) hv = f4_fma(wyk[k], f4_from(t[k]), hv);
#pragma unroll
            for (int k = H + 4 * g; k < H + 4 * g + 4 && k < W; k++) if (k < rows) t[k] = *reinterpret_cast<const float4 *>(src + k * kLongRec);
            __builtin_amdgcn_sched_barrier(0);
        });
        return hv;
    };

    F4 acc[WZ];
#pragma unroll
    for (int k = 0; k < WZ; k++) acc[k] = f4_splat(0.f);

    // prologue: planes 0..2 in flight; plane 0 complete -> its y pass (every wave) and its halo table (wave 15)
    issue(0, 0);
    issue(1, kPlane);
    issue(2, 2 * kPlane);
    asm volatile(MI_VMCNT(8) "\n\ts_barrier" ::: "memory");
    F4 yv = ypass(own);
    if (wave == 15) {
        const F4 hv = ypass(hsrc);
        if constexpr (kXlds) *reinterpret_cast<float4 *>(smem + hx_dst) = f4_to_float4(hv);
        else *reinterpret_cast<float4 *>(smem + HY0 + (unsigned)lane * 16u) = f4_to_float4(hv);
    }

    // Interval i (after barrier i): plane i + 1 has landed (each wave waited for its own DMAs of plane i + 1; those of
    // plane i + 2 and the store behind them may be in flight), the halo table of plane i is complete, the y-filtered row
    // of plane i is in `yv`; nobody reads plane i - 1 any more (nor plane i): its slot takes plane i + 3.
    unsigned bi = 0;                    // LDS offset of plane i
    for (int i0 = 0; i0 < nsteps; i0 += WZ) {
        static_for<WZ>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                if constexpr (!SAME) { launder(wyk); launder(wzk); launder(xte); launder(xto); }
                const unsigned b1 = bi == (kSlots - 1) * kPlane ? 0u : bi + kPlane;     // plane i + 1
                // where plane i + 3 goes: the slot of plane i - 1 in the ring of four, of plane i in the ring of three
                const unsigned b3 = kXlds ? bi : (bi == 0u ? (kLongNB - 1) * kPlane : bi - kPlane);
                // In flight from this wave, oldest first: the 4 DMAs of plane i + 1, the store of step i - 2, the 4 DMAs
                // of plane i + 2, the store of step i - 1 (a store is issued in EVERY step: before the first complete
                // output it goes to a descriptor of zero records).  Plane i + 1 must have landed: vmcnt(5); step 0 has
                // only the 8 DMAs of the prologue behind it: vmcnt(4).
                if ((J == 0 && i0 == 0) || (dbg & 16)) asm volatile(MI_VMCNT(4) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else if constexpr (RG) asm volatile(MI_VMCNT(7) " lgkmcnt(0)\n\ts_barrier" ::: "memory");    // three stores per step
                else asm volatile(MI_VMCNT(5) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const unsigned hyoff = (unsigned)(i & 1) * (kLongHyBytes / 2);
                // ---- halo table of plane i + 1 (the wave changes every plane).  First thing in the step: its reads travel
                // while the other waves of the SIMD have their whole step to issue.
                auto halo_job = [&]() {
                    if (i + 1 < nsteps && wave == (i & 15) && !(dbg & 32)) {
                        // this wave has one row more to filter than the other three of its SIMD: it goes first for the rest
                        // of the step, so that the extra work is shared out instead of left over at the barrier
                        if constexpr (((CFG >> 8) & 1) == 0) __builtin_amdgcn_s_setprio(3);
                        const F4 hv = ypass_batched(hsrc + b1);
                        if constexpr (kXlds) *reinterpret_cast<float4 *>(smem + hx_dst + (unsigned)((i + 1) & 1) * kXpar) = f4_to_float4(hv);
                        else *reinterpret_cast<float4 *>(smem + HY0 + (kLongHyBytes / 2 - hyoff) + (unsigned)lane * 16u) = f4_to_float4(hv);
                    }
                };
                constexpr bool kHaloEnd = (CFG >> 7) & 1;
                if (kHaloEnd ? (dbg & 64) != 0 : !(dbg & 64)) halo_job();
                // ---- x and z passes of plane i, with the y pass of plane i + 1 (for the next step; past the last plane it
                // reads a slot nobody needs) threaded through them by hand: the LDS reads of a group are issued one block
                // of FMAs before they are consumed (sched_barrier keeps the compiler from regrouping them; the group
                // sizes are what the register file allows: 128 VGPRs, 68 of them z accumulators).  
#ifndef MI_LONG_GA
#define MI_LONG_GA 4
#endif
                constexpr int GA0 = CFG ? (CFG & 7) : MI_LONG_GA;
                constexpr int GB0 = CFG ? ((CFG >> 3) & 15) : DBG ? 8 : SAME ? 12 : 10;     // the re-loading variants keep more scalars alive, the ablation build its flags
                constexpr int GA = W < GA0 ? W : GA0, GB = W < GB0 ? W : GB0, ZH = WZ / 2;
                const int wy_rows = (dbg & 1) ? 1 : W, wz_taps = (dbg & 4) ? 1 : WZ;
                float4 R[W];
                const char *ysrc = smem + own + b1;
                issue(i + 3, b3);
                XPass3<W> xp;
                float4 yv4 = f4_to_float4(yv);
                F4 xy;
                if constexpr (kXlds) {
                    constexpr int NBK = XPass3<W>::NBK;
                    float4 blk[2 * NBK + 1];
                    char *xr = smem + xr_own + (unsigned)(i & 1) * kXpar;
                    *reinterpret_cast<float4 *>(xr + 32) = yv4;                    // own block: behind the 8 halo floats
                    // LDS operations of a wave are carried out in order: the reads below see the row
#pragma unroll
                    for (int b = 0; b < 2 * NBK + 1; b++)
                        if (b != NBK) blk[b] = *reinterpret_cast<const float4 *>(xr + 32 + 16 * (b - NBK));
                    blk[NBK] = yv4;
#pragma unroll
                    for (int k = 0; k < GA; k++) if (k < wy_rows) R[k] = *reinterpret_cast<const float4 *>(ysrc + k * kLongRec);
                    __builtin_amdgcn_sched_barrier(0);
                    xy = (dbg & 2) ? yv : xp.window(blk, xte, xto);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    float4 oL[2], eR[2];
                    oL[1] = *reinterpret_cast<const float4 *>(smem + hy_row + hyoff);
                    oL[0] = *reinterpret_cast<const float4 *>(smem + hy_row + hyoff + 16u);
                    if constexpr (RG) {
                        // the row's continuation h0 .. h7 (the y-filtered halo floats from the row's true end: blocks 2, 3 of the
                        // table row) moves into the last lane's block behind its `tail` floats, and the two right-hand blocks
                        // start 4 - tail floats later: twelve consecutive floats of the table row from float 8 - tail on (dword
                        // reads: the address is only 4-byte aligned; past h7 they are another row's, wanted by no stored output)
                        const float *hp = reinterpret_cast<const float *>(smem + hy_row + hyoff + 32u - 4u * (unsigned)tail);
                        const float u1 = hp[1], u2 = hp[2], u3 = hp[3];
                        eR[0] = make_float4(hp[4], hp[5], hp[6], hp[7]);
                        eR[1] = make_float4(hp[8], hp[9], hp[10], hp[11]);
                        yv4.y = rg_p1 ? u1 : yv4.y;
                        yv4.z = rg_p2 ? u2 : yv4.z;
                        yv4.w = rg_p3 ? u3 : yv4.w;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(dbg & 2)) xp.left(yv4, oL, xte, xto);
                    if constexpr (!RG) {
                        eR[0] = *reinterpret_cast<const float4 *>(smem + hy_row + hyoff + 32u);
                        eR[1] = *reinterpret_cast<const float4 *>(smem + hy_row + hyoff + 48u);
                    }
#pragma unroll
                    for (int k = 0; k < GA; k++) if (k < wy_rows) R[k] = *reinterpret_cast<const float4 *>(ysrc + k * kLongRec);
                    __builtin_amdgcn_sched_barrier(0);
                    xy = (dbg & 2) ? yv : xp.right(yv4, eR, lane, last, xte, xto);
                    __builtin_amdgcn_sched_barrier(0);
                }
                yv = f4_scale(wyk[0], f4_from(R[0]));
#pragma unroll
                for (int k = 1; k < GA; k++) if (k < wy_rows) yv = f4_fma(wyk[k], f4_from(R[k]), yv);
#pragma unroll
                for (int k = GA; k < GB; k++) if (k < wy_rows) R[k] = *reinterpret_cast<const float4 *>(ysrc + k * kLongRec);
                __builtin_amdgcn_sched_barrier(0);
                // z pass: scatter into the pending outputs; output i - k takes tap k
                acc[J] = f4_scale(wzk[0], xy);
#pragma unroll
                for (int k = 1; k < ZH; k++) if (k < wz_taps) acc[(J - k + WZ) % WZ] = f4_fma(wzk[k], xy, acc[(J - k + WZ) % WZ]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = GA; k < GB; k++) if (k < wy_rows) yv = f4_fma(wyk[k], f4_from(R[k]), yv);
#pragma unroll
                for (int k = GB; k < W; k++) if (k < wy_rows) R[k] = *reinterpret_cast<const float4 *>(ysrc + k * kLongRec);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = ZH < 1 ? 1 : ZH; k < WZ; k++) if (k < wz_taps) acc[(J - k + WZ) % WZ] = f4_fma(wzk[k], xy, acc[(J - k + WZ) % WZ]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = GB; k < W; k++) if (k < wy_rows) yv = f4_fma(wyk[k], f4_from(R[k]), yv);
                {
                    const unsigned long long oa = (unsigned long long)out +
                                                  (unsigned long long)(unsigned)(zs + i - (WZ - 1)) * (unsigned long long)plane_bytes;
                    const __amdgpu_buffer_rsrc_t rout =
                        __builtin_amdgcn_make_buffer_rsrc((void *)oa, 0, i >= WZ - 1 ? (int)plane_bytes : 0, 0x00020000);
                    const F4 o = acc[(J + 1) % WZ];
                    if (!(dbg & 16)) __builtin_amdgcn_raw_buffer_store_b128(f4_to_u32(o), rout, ovoff, 0, 2);
                    if constexpr (RG) {
                        const u32x4 ou = f4_to_u32(o);
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){ou.x, ou.y}, rout, ovoff2, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b32((tail & 2) ? ou.z : ou.x, rout, ovoff1, 0, 2);
                    }
                }
                if (kHaloEnd ? !(dbg & 64) : (dbg & 64) != 0) halo_job();
                if constexpr (((CFG >> 8) & 1) == 0) __builtin_amdgcn_s_setprio(0);
                bi = b1;
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---------------------------------------------------------------------------
// r4b kernel (`sep3d_long4_kernel`): the y pass on the MATRIX cores.  The 13- / 17-tap kernels are bound by what a wave's
// VALU issues (138 instructions per wave and plane for 102 packed FMAs, 68 % busy at the clock the chip holds), not by
// memory; the matrix pipe sits idle next to it.  Along y the tile is a product with a banded Toeplitz matrix:
//     Y[16 rows x 16 columns] = T[16 x 4 KS] . RAW[4 KS staged rows x 16 columns],   T[i][k] = wy[k - i] (0 outside the band)
// i.e. KS = ROWS / 4 `v_mfma_f32_16x16x4_f32` per 16-column block (full fp32 products and sums).  Wave w owns column
// block w of the tile (16 x 16 outputs): KS ds_read_b32 (lane -> staged row 4 kk + lane / 16, column 16 w + lane % 16:
// conflict-free, the record stride is 272 words), KS MFMAs, four ds_write_b32 into a y-filtered plane `Y` in LDS (two
// copies, by step parity; the raw ring needs three slots only because the y pass runs a plane ahead); the rotating wave
// of the halo table does a 17th block (the 16 halo floats of the records).  One step later every wave reads ITS output
// row of Y -- five aligned 16-byte reads, halo blocks in place: no DPP shifts, no edge selects -- for the x and z passes,
// which stay on the VALU: 34 + 34 packed FMAs + ~15 other VALU instructions per wave and plane instead of 138, and
// 17 x 16 = 272 bytes per lane of LDS reads instead of 17 x 16 + halos with 8.5 x less data.  The band's zeros multiply
// samples OUTSIDE an output's window: a non-finite sample there would leak a NaN (0 x inf) into rows SciPy leaves
// alone, so a block whose MFMA result holds a NaN is recomputed tap by tap (rare; exact semantics).
// Same tile, DMA statement, barrier and vmcnt scheme as the r3 kernel; index-mapping boundary modes; W with ROWS % 4 == 0.
// ---------------------------------------------------------------------------
typedef float f32x4m __attribute__((ext_vector_type(4)));
template <int W, bool SAME, bool DBG = false, int WZ = W>
__global__ void __launch_bounds__(kLongTY * 64)
sep3d_long4_kernel(const float *__restrict__ in, float *__restrict__ out, const LongParams p)
{
    constexpr int ROWS = kLongTY + W - 1, KS = ROWS / 4;
    const int dbg = DBG ? p.dbg : 0;         // ablations (timing aids): 1 no y pass, 2 no x pass, 4 no z scatter, 8 no DMA, 16 no stores
    static_assert(W >= 5 && (W & 1) && ROWS <= kLongRowsMax && ROWS % 4 == 0, "long4 kernel: W = 5, 9, 13, 17");
    static_assert(WZ >= 3 && (WZ & 1) && WZ <= 17 && (WZ == W || !SAME), "long4 kernel: odd WZ");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr unsigned kPlane = kLongRowsMax * kLongRec;
    constexpr unsigned kY0 = 3u * kPlane;                                    // two y-filtered planes of 16 records
    constexpr unsigned kYpar = (unsigned)kLongTY * kLongRec;
    int *ztab = reinterpret_cast<int *>(smem + kY0 + 2u * kYpar);

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
    const int x0 = xt * p.tw, y0 = yt * kLongTY;
    int zs, ze;
    {
        const bool second = zci >= p.nzc0;
        const int zb = second ? p.zb1 : p.zb0, zn = second ? p.zn1 : p.zn0;
        zs = zb + (second ? zci - p.nzc0 : zci) * p.zc;
        ze = min(zs + p.zc, zb + zn);
    }
    const int ty_act = min(kLongTY, ny - y0);
    const int rows_needed = ty_act + W - 1;
    const int nlanes = min(p.tw >> 2, (nx - x0) >> 2);
    const int xe = x0 + 4 * nlanes;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * 4u;
    const int zi0 = zs - p.oz;
    const int nsteps = ze - zs + WZ - 1;

    for (int i = threadIdx.x; i < nsteps; i += kLongTY * 64) ztab[i] = bmap<int>(zi0 + i, nz, p.mz);
    __syncthreads();

    unsigned vmain[2], vhalo[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = wave + 16 * h;
        const int ys = bmap<int>(y0 - p.oy + r, ny, p.my);
        const bool valid = r < rows_needed;
        vmain[h] = (valid && lane < nlanes) ? (unsigned)(ys * nx + x0 + 4 * lane) * 4u : kOOB;
        const int j = lane & 15;
        const int xsrc = bmap<int>(j < 8 ? x0 - 8 + j : xe + j - 8, nx, p.mx);
        vhalo[h] = (valid && xsrc >= 0) ? (unsigned)(ys * nx + xsrc) * 4u : kOOB;
    }
    const unsigned ovoff = (wave < ty_act && lane < nlanes) ? (unsigned)((y0 + wave) * nx + x0 + 4 * lane) * 4u : kOOB;

    auto issue = [&](int i, unsigned bufoff) {
        const bool live = i < nsteps;
        int zsrc = zi0 + i;
        if ((unsigned)zsrc >= (unsigned)nz) zsrc = __builtin_amdgcn_readfirstlane(ztab[live ? i : 0]);   // boundary planes only
        const unsigned long long a = (unsigned long long)in + (unsigned long long)(unsigned)zsrc * (unsigned long long)plane_bytes;
        u32x4_t rin;
        rin.x = (unsigned)a;
        rin.y = (unsigned)(a >> 32);
        rin.z = live ? plane_bytes : 0u;
        rin.w = 0x00020000u;
        if (!(dbg & 8)) dma_two_rows(rin, vmain[0], vhalo[0], vmain[1], vhalo[1], bufoff + (unsigned)wave * kLongRec);
    };
    constexpr int kArgBase = 2 * sizeof(void *);
    kfloats wyk = kernarg_floats(kArgBase + offsetof(LongParams, wyp));
    kfloats wzk = SAME ? wyk : kernarg_floats(kArgBase + offsetof(LongParams, wzp));
    kfloats xte = SAME ? wyk : kernarg_floats(kArgBase + offsetof(LongParams, wxe));
    kfloats xto = kernarg_floats(kArgBase + offsetof(LongParams, wxo));

    // ---- the band: A operand of step kk = T[i = lane % 16][k = 4 kk + lane / 16] = wy[k - i]
    float Ta[KS];
    {
        const int ti = lane & 15, tg = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            const int tap = 4 * kk + tg - ti;
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < W; k++) a = tap == k ? wyk[k] : a;
            Ta[kk] = a;
        }
    }
    // B operand / result geometry of a block: lane -> (staged row 4 kk + lane / 16, column 16 blk + lane % 16); the result
    // registers r = 0 .. 3 are rows 4 (lane / 16) + r of column 16 blk + lane % 16
    const unsigned rd_lane = (unsigned)(lane >> 4) * kLongRec + (unsigned)(lane & 15) * 4u;
    const unsigned wr_row = (unsigned)(lane >> 4) * 4u * kLongRec;
    // where the result goes in a Y record: [8 left halo floats][the row][8 right halo floats behind its last valid float4]
    const int mycol = 16 * wave + (lane & 15);
    const bool main_ok = mycol < 4 * nlanes;
    const unsigned wr_main = wr_row + 32u + (unsigned)mycol * 4u;
    const unsigned wr_halo = wr_row + ((lane & 15) < 8 ? (unsigned)(lane & 15) * 4u : 32u + 16u * (unsigned)nlanes + (unsigned)((lane & 15) - 8) * 4u);

    // y pass of one 16-column block of the plane staged at `sb` -> Y plane at `yb`
    auto yblock = [&](unsigned sb, unsigned yb, int blk, unsigned wr, bool ok) {
        const char *src = smem + sb + rd_lane + (unsigned)blk * 64u;
        float bv[KS];
#pragma unroll
        for (int kk = 0; kk < KS; kk++) bv[kk] = *reinterpret_cast<const float *>(src + (unsigned)(4 * kk) * kLongRec);
        f32x4m d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < KS; kk++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(Ta[kk], bv[kk], d, 0, 0, 0);
        const float chk = (d.x + d.y) + (d.z + d.w);
        char *dst = smem + yb + wr;
        if (__builtin_amdgcn_ballot_w64(chk != chk) != 0) {
            // a NaN: either the window of one of these outputs holds a non-finite sample (then the tap-by-tap sum says what
            // SciPy says) or a zero of the band met one outside it (then it must not show).  Row by row: few live registers.
            const char *col = smem + sb + (unsigned)blk * 64u + (unsigned)(lane & 15) * 4u + wr_row;
#pragma unroll 1
            for (int r = 0; r < 4; r++) {                                    // rolled loops: this path is rare, its code should be small
                float acc = 0.f;
#pragma unroll 1
                for (int k = 0; k < W; k++) acc = fmaf(wyk[k], *reinterpret_cast<const float *>(col + (unsigned)(r + k) * kLongRec), acc);
                if (ok) *reinterpret_cast<float *>(dst + (unsigned)r * kLongRec) = acc;
            }
        } else if (ok) {
            *reinterpret_cast<float *>(dst) = d.x;
            *reinterpret_cast<float *>(dst + kLongRec) = d.y;
            *reinterpret_cast<float *>(dst + 2 * kLongRec) = d.z;
            *reinterpret_cast<float *>(dst + 3 * kLongRec) = d.w;
        }
    };
    auto yplane = [&](int i, unsigned sb) {                                  // y pass of the plane of step i
        const unsigned yb = kY0 + (unsigned)(i & 1) * kYpar;
        yblock(sb, yb, wave, wr_main, main_ok);
        if (wave == (i & 15)) yblock(sb, yb, 16, wr_halo, true);             // the 16 halo floats of the records: block 16
    };

    F4 acc[WZ];
#pragma unroll
    for (int k = 0; k < WZ; k++) acc[k] = f4_splat(0.f);

    // prologue: planes 0..2 in flight; plane 0 complete -> its y pass
    issue(0, 0);
    issue(1, kPlane);
    issue(2, 2 * kPlane);
    asm volatile(MI_VMCNT(8) "\n\ts_barrier" ::: "memory");
    yplane(0, 0);

    // Interval i (after barrier i): plane i + 1 has landed, Y(i) is complete (every wave waited for its LDS writes before
    // the barrier), nobody reads Y(i - 1) or raw plane i any more: the slot of plane i takes plane i + 3, Y(i + 1) goes
    // where Y(i - 1) was.
    const unsigned xr_own = kY0 + (unsigned)wave * kLongRec + (unsigned)lane * 16u;       // block -2 of this lane's window in Y(i)
    const bool vector_first = ((wave >> 2) & 1) != 0 && !(dbg & 64);
    unsigned bi = 0;                    // LDS offset of raw plane i
    for (int i0 = 0; i0 < nsteps; i0 += WZ) {
        static_for<WZ>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                if constexpr (!SAME) { launder(wyk); launder(wzk); launder(xte); launder(xto); }
                const unsigned b1 = bi == 2u * kPlane ? 0u : bi + kPlane;     // plane i + 1
                // in flight from this wave, oldest first: the 4 DMAs of plane i + 1, the store of step i - 2, the 4 DMAs of
                // plane i + 2, the store of step i - 1 (see sep3d_long3_kernel): plane i + 1 must have landed
                if (J == 0 && i0 == 0) asm volatile(MI_VMCNT(4) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile(MI_VMCNT(5) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue(i + 3, bi);
                // ---- x and z passes of plane i (VALU) and y pass of plane i + 1 (matrix cores): independent of each other.  The
                // waves of a SIMD are w, w + 4, w + 8, w + 12: half of them take the matrix part first, half the vector part, so
                // that the two pipes of a SIMD have work at the same time instead of one after the other.
                auto xz = [&]() {
                    // x pass of plane i out of Y(i): the window's blocks around (and including) the lane's own float4
                    constexpr int NBK = XPass3<W>::NBK;
                    float4 blk[2 * NBK + 1];
                    const char *xr = smem + xr_own + (unsigned)(i & 1) * kYpar;
#pragma unroll
                    for (int q = 0; q < 2 * NBK + 1; q++) blk[q] = *reinterpret_cast<const float4 *>(xr + 32 + 16 * (q - NBK));
                    XPass3<W> xp;
                    const F4 xy = (dbg & 2) ? f4_from(blk[NBK]) : xp.window(blk, xte, xto);
                    // z pass: scatter into the pending outputs; output i - k takes tap k
                    acc[J] = f4_scale(wzk[0], xy);
                    const int wz_taps = (dbg & 4) ? 1 : WZ;
#pragma unroll
                    for (int k = 1; k < WZ; k++) if (k < wz_taps) acc[(J - k + WZ) % WZ] = f4_fma(wzk[k], xy, acc[(J - k + WZ) % WZ]);
                    const unsigned long long oa = (unsigned long long)out +
                                                  (unsigned long long)(unsigned)(zs + i - (WZ - 1)) * (unsigned long long)plane_bytes;
                    const __amdgpu_buffer_rsrc_t rout =
                        __builtin_amdgcn_make_buffer_rsrc((void *)oa, 0, i >= WZ - 1 ? (int)plane_bytes : 0, 0x00020000);
                    const F4 o = acc[(J + 1) % WZ];
                    if (!(dbg & 16)) __builtin_amdgcn_raw_buffer_store_b128(f4_to_u32(o), rout, ovoff, 0, 2);
                };
                if (vector_first) xz();
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < nsteps && !(dbg & 1)) yplane(i + 1, b1);
                __builtin_amdgcn_sched_barrier(0);
                if (!vector_first) xz();
                bi = b1;
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static mi::Knob g_long_rows{0};        // kernel generation: 0 / 3 = the r3 pipelined kernel, 1 = the r2 kernel (kept for 9 / 13 / 17 taps as the comparator), 4 = the r4 kernel with the y pass on the matrix cores (9 / 13 / 17 taps; an experiment that did not pay)
static mi::Knob g_long_dbg{0};         // tuning ablations, see LongParams::dbg
static mi::Knob g_long_const0{1};     // r5: 1 = constant mode with cval == 0 takes the r3 kernel (zero fill by the staging), 0 = the r2 kernel with its correction
static mi::Knob g_long_cfg{0};         // MI_LONG_TUNE builds: which tuning variant of the 17-tap kernel runs

#ifdef MI_LONG_DEV
#define MI_LONG_OLD(W) ((W) == 17)
#else
#define MI_LONG_OLD(W) ((W) == 9 || (W) == 13 || (W) == 17)
#endif

template <typename K>
static int long_launch_one(K kernel, PerDeviceOnce &attr_done, size_t lds, int total, const float *in, float *out, const LongParams &p, hipStream_t s)
{
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL(kernel, dim3(total), dim3(kLongTY * 64), lds, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <int W, bool SAME, bool HAS_CONST>
static int launch_long(const float *in, float *out, LongParams &p, hipStream_t s)
{
    const size_t lds = (size_t)kLongRawBytes + kLongHyBytes + (size_t)(kLongMaxChunk + kStreamMaxTaps) * sizeof(int) +
                       (HAS_CONST ? (size_t)kLongMaxChunk * sizeof(float) : 0);
    const int total = p.nxt * p.nyt * p.nzc;
    if constexpr (HAS_CONST) {
        // r5: a zero fill value needs no correction at all -- zero fill is what the staging leaves for whatever lies beyond
        // the array -- and takes the r3 kernel like every other mode
        if (p.cval == 0.0f && g_long_const0) {
            static PerDeviceOnce attr_z;
            note_kernel("mi::sep3d_long3_kernel<%d,%s> grid=%d (fused y/x/z separable pass, LDS-DMA staged, y pass one plane ahead; constant mode, zero fill)",
                        W, SAME ? "true" : "false", total);
            return long_launch_one(sep3d_long3_kernel<W, SAME, false>, attr_z, lds, total, in, out, p, s);
        }
        // any other fill value keeps the r2 kernel: its correction terms (five more live registers) do not fit beside the r3
        // kernel's read groups without spilling, and a spill is a vector-memory operation the vmcnt arithmetic does not count
        static PerDeviceOnce attr_c;
        note_kernel("mi::sep3d_long_kernel<%d,%s,true> grid=%d (fused y/x/z separable pass, LDS-DMA staged, constant mode)", W,
                    SAME ? "true" : "false", total);
        return long_launch_one(sep3d_long_kernel<W, SAME, true>, attr_c, lds, total, in, out, p, s);
    } else {
        if constexpr (MI_LONG_OLD(W)) {
            if (g_long_rows == 1) {
                static PerDeviceOnce attr_old;
                note_kernel("mi::sep3d_long_kernel<%d,%s,false> grid=%d (fused y/x/z separable pass, LDS-DMA staged, r2 instruction stream)", W,
                            SAME ? "true" : "false", total);
                return long_launch_one(sep3d_long_kernel<W, SAME, false>, attr_old, lds, total, in, out, p, s);
            }
            // the ablation flags exist in these instances only
            if (p.dbg != 0 && g_long_rows != 4) {
                static PerDeviceOnce attr_dbg;
                note_kernel("mi::sep3d_long3_kernel<%d,%s,true> grid=%d (ablation build, dbg=%d)", W, SAME ? "true" : "false", total, p.dbg);
                return long_launch_one(sep3d_long3_kernel<W, SAME, true>, attr_dbg, lds, total, in, out, p, s);
            }
        }
#ifdef MI_LONG_TUNE
        if constexpr (W == 17 && SAME) {
            if (g_long_rows == 2) {
                static PerDeviceOnce attr2;
                if (!attr2) {
                    MI_HIP(hipFuncSetAttribute((const void *)sep3d_long2_kernel<17, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    attr2 = true;
                }
                note_kernel("mi::sep3d_long2_kernel<17,true,false> grid=%d (r2 stream, two rows per wave; tuning build)", total);
                hipLaunchKernelGGL((sep3d_long2_kernel<17, true, false>), dim3(total), dim3(512), lds, s, in, out, p);
                MI_HIP(hipGetLastError());
                return MI_OK;
            }
            const int cfg = g_long_cfg;
#define MI_LONG_CFG(C)                                                                                           \
            if (cfg == (C)) {                                                                                    \
                static PerDeviceOnce attr_c;                                                                      \
                note_kernel("mi::sep3d_long3_kernel<17,true,false,%d> grid=%d (tuning variant)", (C), total);    \
                return long_launch_one(sep3d_long3_kernel<17, true, false, (C)>, attr_c, lds, total, in, out, p, s); \
            }
            MI_LONG_CFG(4 | (12 << 3) | 512) MI_LONG_CFG(4 | (10 << 3) | 512) MI_LONG_CFG(4 | (8 << 3) | 512)
#undef MI_LONG_CFG
        }
#endif
        if constexpr (W == 9 || W == 13 || W == 17) {
            if (g_long_rows == 4) {                 // not the default: 268 against 261 us on config B (fp32 MFMAs and the packed FMAs share one datapath)
                static PerDeviceOnce attr4;
                const size_t lds4 = 3 * (size_t)kLongRowsMax * kLongRec + 2 * (size_t)kLongTY * kLongRec + (size_t)(kLongMaxChunk + kStreamMaxTaps) * sizeof(int);
                note_kernel("mi::sep3d_long4_kernel<%d,%s> grid=%d (fused y/x/z separable pass, LDS-DMA staged, y pass on the matrix cores)", W,
                            SAME ? "true" : "false", total);
                if constexpr (W == 17 && SAME) {
                    if (p.dbg != 0) {
                        static PerDeviceOnce attr4d;
                        return long_launch_one(sep3d_long4_kernel<W, SAME, true>, attr4d, lds4, total, in, out, p, s);
                    }
                }
                return long_launch_one(sep3d_long4_kernel<W, SAME>, attr4, lds4, total, in, out, p, s);
            }
        }
        static PerDeviceOnce attr_done;
        note_kernel("mi::sep3d_long3_kernel<%d,%s> grid=%d (fused y/x/z separable pass, LDS-DMA staged, y pass one plane ahead)", W,
                    SAME ? "true" : "false", total);
        return long_launch_one(sep3d_long3_kernel<W, SAME, false>, attr_done, lds, total, in, out, p, s);
    }
}

static int long_cus() { return device_cus(); }

// (in-plane taps, z taps) pairs the r3 kernel is instantiated for besides the cubic ones (r4: with more z taps too): volumes with anisotropic
// voxels, where a gaussian given in millimetres has fewer taps through the slices (each pair is one more kernel to
// compile: the list is what sigma = 0.5 ... 2 voxels in the plane, in steps of a quarter, needs with 2-4 x thicker slices)
#ifdef MI_LONG_DEV
#define MI_LONG_ANISO_PAIRS(X) X(17, 9)
#else
#define MI_LONG_ANISO_PAIRS(X)                                                                             \
    X(5, 3) X(7, 3) X(7, 5) X(9, 3) X(9, 5) X(9, 7) X(11, 3) X(11, 5) X(11, 7) X(13, 3) X(13, 5) X(13, 7) X(13, 9)     \
    X(15, 5) X(15, 7) X(15, 9) X(17, 3) X(17, 5) X(17, 7) X(17, 9) X(17, 13)                                           \
    /* r4: MORE taps along z than in the plane (sigma larger through the slices: gaussian (2, 1, 1) = 17 x 9 x 9) */     \
    X(5, 9) X(5, 13) X(5, 17) X(7, 13) X(9, 13) X(9, 17) X(13, 17)
#endif
bool long_aniso_pair(int w, int wzn)
{
#define MI_LONG_ANISO_TEST(N, NZ) if (w == (N) && wzn == (NZ)) return true;
    MI_LONG_ANISO_PAIRS(MI_LONG_ANISO_TEST)
#undef MI_LONG_ANISO_TEST
    return false;
}

Knob g_stream_nt{-1};                  // long_common.hpp
static mi::Knob g_long_zchunks{0};     // test hook: number of z chunks (0 = cost model)
static mi::Knob g_long_same{1};        // test hook: 0 = always the reloading variant

// Fused long-kernel path: cubic odd W in 11..17 (9 behind the test hook), origins on y / z allowed, no constant mode.
// Returns MI_ERR_UNSUPPORTED when the request is outside that (the caller runs the streaming passes).
int run_sep3d_long(const float *in, float *out, int nz, int ny, int nx, int w, int wzn, const float *wx, const float *wy,
                   const float *wz, int oy, int oz, int mx, int my, int mz, float cval, const int64_t zb[2],
                   const int64_t zn[2], hipStream_t s, bool ragged)
{
    // w: taps along y and x, wzn: taps along z (== w: the cubic kernels; a few (w, wzn) pairs with wzn < w besides)
    if (w < 3 || w > 17 || !(w & 1) || wzn < 3 || wzn > 17 || !(wzn & 1)) return MI_ERR_UNSUPPORTED;
    // rows that are not a multiple of 4 floats (r6): the cubic kernels of 9 .. 17 taps, index-mapping modes or a zero fill value
    if (ragged && (wzn != w || w < 9 || nx < 16)) return MI_ERR_UNSUPPORTED;
    const bool has_const = mx == MI_MODE_CONSTANT || my == MI_MODE_CONSTANT || mz == MI_MODE_CONSTANT;
    if (wzn != w && ((has_const && !(cval == 0.0f && g_long_const0)) || !long_aniso_pair(w, wzn))) return MI_ERR_UNSUPPORTED;
    if (has_const && w <= 7 && !(cval == 0.0f && g_long_const0)) return MI_ERR_UNSUPPORTED;      // below 9 taps only the zero-fill form runs here (r5); the lean kernel has the fill values
    if ((int64_t)ny * nx * 4 >= ((int64_t)1 << 31)) return MI_ERR_UNSUPPORTED;
    if (t_dry_run) return MI_OK;          // every odd (w, w) in 3 .. 17 and every pair of long_aniso_pair() has an instance
    LongParams p;
    memset(&p, 0, sizeof(p));
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.oy = oy; p.oz = oz;
    p.mx = mx; p.my = my; p.mz = mz;
    p.nxt = (nx + 255) / 256;
    p.tw = (((nx + p.nxt - 1) / p.nxt) + 3) & ~3;          // equal tiles (see separable3d.hip)
    p.nyt = (ny + kLongTY - 1) / kLongTY;
    double sx = 0, sy = 0, sz = 0;
    for (int k = 0; k < w; k++) {
        p.wyv[2 * k] = p.wyv[2 * k + 1] = wy[k];
        p.wxs[k] = wx[k];
        p.wyp[k] = wy[k]; p.wxe[k] = wx[k];
        p.wxo[k + 1] = wx[k];                               // wxo[2m] = wx[2m-1], wxo[2m+1] = wx[2m]
        sx += wx[k]; sy += wy[k];
    }
    for (int k = 0; k < wzn; k++) {
        p.wzv[2 * k] = p.wzv[2 * k + 1] = wz[k];
        p.wzp[k] = wz[k];
        sz += wz[k];
    }
    p.dbg = g_long_dbg;
    p.nt = stream_nt_for((long long)nz * ny * nx * 8, p.nxt);
    p.cval = cval;
    p.cval_sum = (float)((double)cval * sx * sy * sz);
    {
        const int rx = w / 2, nb = (rx + 3) / 4, base = 4 * nb - rx;
        for (int q = 0; q < 2; q++) {
            const int t0 = base + q, m0 = t0 / 2;
            for (int u = 0; u < kStreamMaxTaps / 2 + 2; u++)
                for (int h = 0; h < 2; h++) {
                    const int j = 2 * (m0 + u) + h - t0;
                    p.xpair[q][2 * u + h] = (j >= 0 && j < w) ? wx[j] : 0.0f;
                }
        }
    }
    // z chunks: rounds of (columns x chunks) workgroups over the CUs, each costing chunk + ramp plane steps
    const int ncu = long_cus();
    const int cols = p.nxt * p.nyt;
    const int nzr = (int)(zn[0] + zn[1]);                 // planes to produce
    int best_nzc = 1;
    double best = 1e300;
    for (int nzc = 1; nzc <= nzr && nzc <= 256; nzc++) {
        const int chunk = (nzr + nzc - 1) / nzc;
        if (chunk > kLongMaxChunk) continue;
        const int real = (nzr + chunk - 1) / chunk;
        const double rounds = (double)(((int64_t)cols * real + ncu - 1) / ncu);
        const double cost = rounds * (chunk + wzn - 1 + 3);
        if (cost < best) { best = cost; best_nzc = real; }
    }
    if (g_long_zchunks > 0) best_nzc = std::min((int)g_long_zchunks, nzr);
    p.zc = (nzr + best_nzc - 1) / best_nzc;
    if (p.zc > kLongMaxChunk) p.zc = kLongMaxChunk;
    p.zb0 = (int)zb[0]; p.zn0 = (int)zn[0]; p.zb1 = (int)zb[1]; p.zn1 = (int)zn[1];
    p.nzc0 = (int)((zn[0] + p.zc - 1) / p.zc);
    p.nzc = p.nzc0 + (int)((zn[1] + p.zc - 1) / p.zc);
    if (wzn != w) {
#define MI_LONG_ANISO(N, NZ)                                                                                         \
        if (w == (N) && wzn == (NZ)) {                                                                                \
            const size_t lds = (size_t)kLongRawBytes + kLongHyBytes + (size_t)(kLongMaxChunk + kStreamMaxTaps) * sizeof(int); \
            static PerDeviceOnce attr_a;                                                                              \
            note_kernel("mi::sep3d_long3_kernel<%d,false,false,0,%d> grid=%d (fused y/x/z separable pass, %d taps in the plane, %d along z)", \
                        (N), (NZ), p.nxt * p.nyt * p.nzc, (N), (NZ));                                                 \
            return long_launch_one(sep3d_long3_kernel<(N), false, false, 0, (NZ)>, attr_a, lds, p.nxt * p.nyt * p.nzc, in, out, p, s); \
        }
        MI_LONG_ANISO_PAIRS(MI_LONG_ANISO)
#undef MI_LONG_ANISO
        return MI_ERR_UNSUPPORTED;
    }
    if (ragged) {
        if (has_const && !(cval == 0.0f && g_long_const0)) return MI_ERR_UNSUPPORTED;
        // (the three axes share one weight vector -- isotropic gaussian_filter, uniform_filter: the variant whose scalars stay
        // resident; the re-loading variant has no registers left for the ragged edge and would spill)
        for (int k = 0; k < w; k++)
            if (!(wx[k] == wy[k] && wy[k] == wz[k])) return MI_ERR_UNSUPPORTED;
        const size_t lds = (size_t)kLongRawBytes + kLongHyBytes + (size_t)(kLongMaxChunk + kStreamMaxTaps) * sizeof(int);
        const int total = p.nxt * p.nyt * p.nzc;
#define MI_LONG_RAGGED(N)                                                                                                 \
        case N: {                                                                                                         \
            static PerDeviceOnce attr_r;                                                                                  \
            note_kernel("mi::sep3d_long3_kernel<%d,true,ragged> grid=%d (fused y/x/z separable pass, LDS-DMA staged, rows of any length)", (N), total); \
            return long_launch_one(sep3d_long3_kernel<(N), true, false, 0, (N), true>, attr_r, lds, total, in, out, p, s); \
        }
        switch (w) {
            MI_LONG_RAGGED(9) MI_LONG_RAGGED(11) MI_LONG_RAGGED(13) MI_LONG_RAGGED(15) MI_LONG_RAGGED(17)
        }
#undef MI_LONG_RAGGED
        return MI_ERR_UNSUPPORTED;
    }
    bool same = g_long_same != 0;
    for (int k = 0; k < w; k++) same = same && wx[k] == wy[k] && wy[k] == wz[k];
#define MI_LONG_CASE(N)                                                                                   \
    case N:                                                                                               \
        if (has_const) return same ? launch_long<N, true, true>(in, out, p, s) : launch_long<N, false, true>(in, out, p, s); \
        return same ? launch_long<N, true, false>(in, out, p, s) : launch_long<N, false, false>(in, out, p, s);
    switch (w) {
#ifdef MI_LONG_DEV
        MI_LONG_CASE(17)
#else
        MI_LONG_CASE(3) MI_LONG_CASE(5) MI_LONG_CASE(7) MI_LONG_CASE(9) MI_LONG_CASE(11) MI_LONG_CASE(13) MI_LONG_CASE(15) MI_LONG_CASE(17)
#endif
    }
#undef MI_LONG_CASE
    return MI_ERR_UNSUPPORTED;
}

}  // namespace mi

extern "C" int mi_debug_set_stream_nt(int k) { mi::g_stream_nt = k; return MI_OK; }
extern "C" int mi_debug_set_long_zchunks(int n) { mi::g_long_zchunks = n; return MI_OK; }
extern "C" int mi_debug_set_long_same(int n) { mi::g_long_same = n; return MI_OK; }
extern "C" int mi_debug_set_long_dbg(int f) { mi::g_long_dbg = f; return MI_OK; }
extern "C" int mi_debug_set_long_rows(int k) { mi::g_long_rows = k; return MI_OK; }
extern "C" int mi_debug_set_long_cfg(int k) { mi::g_long_cfg = k; return MI_OK; }
extern "C" int mi_debug_set_long_const0(int k) { mi::g_long_const0 = k; return MI_OK; }
