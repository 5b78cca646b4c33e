// separable3d.hip -- fused separable 3-D filter, float32, one launch.
//
// What it replaces: uniform_filter / gaussian_filter in the reference are
// three K1 launches plus two zero-fills and two full-volume copy-backs
// (cupyimg/scipy/ndimage/filters.py:602-665, :725-792; the in-place temp +
// copy at _filters_core.py:148-155) -- about 52 B/voxel of HBM traffic.  This
// kernel reads every input voxel once and writes every output voxel once:
// 8 B/voxel, the algorithmic minimum, so the bound is HBM bandwidth.
//
// Data layout / work decomposition (C-contiguous volume, axes z, y, x):
//   * a workgroup owns a column of TX=256 x TY voxels and streams along z over
//     a chunk of planes (2.5-D blocking);
//   * a wave owns whole 256-float row segments: lane l holds the float4 at
//     x0 + 4l, so every global access is a fully coalesced 1 KiB wave
//     transaction (global_load_dwordx4 / global_store_dwordx4);
//   * x pass: in registers.  A lane needs `reach` floats from each neighbour
//     lane: wave-wide lane shifts (__shfl_up/__shfl_down by one lane); only
//     lanes 0 / last take their halo from a separately loaded 16-byte edge
//     vector (boundary-mapped at the array edge);
//   * z pass: a register ring of the last WZ x-filtered planes per owned row;
//   * y pass: the x/z-filtered rows (TY + wy - 1 of them) are exchanged
//     through LDS (ds_write_b128 / ds_read_b128, lane-contiguous, conflict
//     free), double buffered so one barrier per plane suffices;
//   * the next plane's loads are issued before the current plane is
//     processed (1 plane = ROWS KiB in flight per workgroup).
//   * blockIdx -> tile mapping gives every XCD a contiguous z-chunk so that
//     y-neighbouring columns, which re-read each other's 2*reach halo rows,
//     share one L2.
//
// Pass order is x, z, y instead of the reference's 0, 1, 2.  Index-mapping
// boundary modes commute exactly with filtering along other axes; the
// constant mode only does when every kernel sums to one, which the host
// checks (otherwise MI_ERR_UNSUPPORTED and the caller runs 1-D passes).
// Arithmetic is float32 FMA; against SciPy's double-accumulate-round-per-pass
// the difference is ~1e-7 relative (tolerance 1e-6, tests/test_gpu_filters.py).
#include <stdarg.h>

#include "sep_common.hpp"
#include "stream3d.hpp"

namespace mi {

thread_local bool t_dry_run = false;
static thread_local char t_last_kernel[160] = "";
void note_kernel(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_last_kernel, sizeof(t_last_kernel), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------
// v2: wave-specialised variant.  NWP producer waves own the rows (load, x pass,
// z ring, LDS write); NWC consumer waves do the y pass and the stores.  One
// barrier per plane hands a finished LDS buffer from producers to consumers;
// producers fill the other buffer meanwhile.  Loads (producers) and stores
// (consumers) therefore sit on different waves' vmcnt counters: nobody ever
// waits for a store, the next plane's loads are issued as soon as the x pass
// has consumed the registers, and they stay in flight across the barrier.
// Lane shifts for the x pass use DPP wave_shr:1 / wave_shl:1 (one VALU op, no
// LDS crossbar); lane 0 keeps `old` = its left halo value.
// ---------------------------------------------------------------------------
// edge halo: NE floats per side held by lanes 0 and `last` (NE = 2 for reach <= 2, else 4)
template <int WX>
__device__ __forceinline__ float4 xpass_dpp(const float4 v, const float (&edge)[(WX / 2 <= 2) ? 2 : 4], int lane,
                                            int last, const float *__restrict__ wx)
{
    if constexpr (WX == 1) {
        return make_float4(v.x * wx[0], v.y * wx[0], v.z * wx[0], v.w * wx[0]);
    } else {
        constexpr int RX = WX / 2;
        constexpr int NE = RX <= 2 ? 2 : 4;
        float e[4 + 2 * RX];
#pragma unroll
        for (int j = 0; j < RX; j++) {
            // left halo: the NE floats just left of the tile, last RX of them are used
            float l = dpp_from_left(edge[NE - RX + j], comp(v, 4 - RX + j));
            float r = dpp_from_right(edge[j], comp(v, j));
            if (lane == last) r = edge[j];
            e[j] = l;
            e[RX + 4 + j] = r;
        }
        e[RX + 0] = v.x; e[RX + 1] = v.y; e[RX + 2] = v.z; e[RX + 3] = v.w;
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float a = wx[0] * e[c];
#pragma unroll
            for (int k = 1; k < WX; k++) a = fmaf(wx[k], e[c + k], a);
            o[c] = a;
        }
        return make_float4(o[0], o[1], o[2], o[3]);
    }
}

// one plane's worth of a producer wave's rows, in flight or ready
template <int R, int NE>
struct RowRegs {
    float4 v[R];
    float t[R][NE];   // raw edge floats as loaded (lanes 0 / last only)
};

constexpr int kMaxChunk = 2048;   // planes per z chunk (plane-index table lives in LDS)

template <int WX, int WZ, int NWP, int NWC, int R>
__global__ void __launch_bounds__((NWP + NWC) * 64)
sep3d_ws_kernel(const float *__restrict__ in, float *__restrict__ out, const Sep3dParams p)
{
    constexpr int ROWS = NWP * R;
    constexpr int RX = WX / 2;
    constexpr int NE = RX <= 2 ? 2 : 4;
    constexpr int RING = WZ > 1 ? WZ - 1 : 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *lds = reinterpret_cast<float4 *>(smem);                    // [2][ROWS][64]
    int *ztab = reinterpret_cast<int *>(smem + (size_t)2 * ROWS * 1024); // [zc + WZ - 1] source plane or -1

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
    const int x0 = xt * p.tw, y0 = yt * p.ty;
    int zs, ze;
    chunk_planes(p, zci, &zs, &ze);
    const int ty_act = min(p.ty, ny - y0);
    const int rows_needed = ty_act + p.wy - 1;
    const int nlanes = min(p.tw >> 2, (nx - x0) >> 2);
    const int last = nlanes - 1;
    const int64_t plane = (int64_t)ny * nx;
    const int zi0 = zs - p.oz;
    const int nsteps = ze - zs + WZ - 1;          // input planes zi0 .. zi0 + nsteps - 1

    // source plane of every step, boundary mapped once (branchy code kept out of the loop)
    for (int i = threadIdx.x; i < nsteps; i += (NWP + NWC) * 64) ztab[i] = bmap<int>(zi0 + i, nz, p.mz);
    __syncthreads();

    if (wave < NWP) {
        // ------------------------------------------------------------ producer
        // Per-lane edge recipe (lane 0: the NE floats left of the tile; lane
        // `last`: the NE floats right of it): ONE load of NE consecutive floats
        // at row + eoff, then component picks idx[k] (forward / reversed / splat).
        int es0, ek0, es1, ek1;
        edge_desc(0, x0, x0 + 4 * nlanes, nx, p.mx, &es0, &ek0);
        edge_desc(1, x0, x0 + 4 * nlanes, nx, p.mx, &es1, &ek1);
        const bool is_edge_lane = (lane == 0) || (lane == last);
        const int ekind = lane == 0 ? ek0 : ek1;
        int eoff = lane == 0 ? es0 : es1;
        int eidx[NE];
        if (lane == 0) {
            if (ekind == EDGE_FWD) eoff += 4 - NE;             // x0-NE .. x0-1
        } else {
            if (ekind == EDGE_REV) eoff += 4 - NE;             // last NE floats of the block
            if (ekind == EDGE_SPLAT) eoff -= NE - 1;           // keep the load inside the row
        }
#pragma unroll
        for (int k = 0; k < NE; k++)
            eidx[k] = ekind == EDGE_FWD ? k : (ekind == EDGE_REV ? NE - 1 - k : (lane == 0 ? 0 : NE - 1));
        const bool edge_load = is_edge_lane && ekind != EDGE_CONST;
        // lanes beyond the row end re-read the last valid float4 (never used)
        const int xoff = x0 + 4 * min(lane, last);

        // rows of this wave: element offset of the (boundary mapped) source row
        // inside a plane; invalid rows (constant boundary / unused) read row 0
        // and are overridden with cval afterwards
        int64_t yoff[R];
        bool yconst[R], yused[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int rr = wave * R + r;
            const int ys = rr < rows_needed ? bmap<int>(y0 - p.oy + rr, ny, p.my) : -2;
            yused[r] = ys != -2;
            yconst[r] = ys < 0;
            yoff[r] = (int64_t)max(ys, 0) * nx;
        }

        auto load_row = [&](int zsrc, int r, RowRegs<R, NE> &S) {
            const float *row = in + (int64_t)max(zsrc, 0) * plane + yoff[r];
            S.v[r] = *reinterpret_cast<const float4 *>(row + xoff);
            if (edge_load) {
                if constexpr (NE == 2) {
                    struct __attribute__((packed, aligned(4))) f2u { float a, b; };
                    const f2u q = *reinterpret_cast<const f2u *>(row + eoff);
                    S.t[r][0] = q.a; S.t[r][1] = q.b;
                } else {
                    const float4u q = *reinterpret_cast<const float4u *>(row + eoff);
                    S.t[r][0] = q.x; S.t[r][1] = q.y; S.t[r][2] = q.z; S.t[r][3] = q.w;
                }
            }
        };

        float4 ring[RING][R];
#pragma unroll
        for (int k = 0; k < RING; k++)
#pragma unroll
            for (int r = 0; r < R; r++) ring[k][r] = make_float4(0.f, 0.f, 0.f, 0.f);

        int buf = 0;
        // One plane, row by row: x pass on the arrived row, refill the same
        // registers with the row two planes ahead, z pass, hand over via LDS.
        // zcur / znext: source planes (or -1 = constant plane).
        auto step = [&](int i, RowRegs<R, NE> &S) {
            const int zcur = ztab[i];
            const bool more = i + 2 < nsteps;
            const int znext = more ? ztab[i + 2] : 0;
            const bool emit = i >= WZ - 1;
            float4 *wbuf = lds + buf * (ROWS * 64);
#pragma unroll
            for (int r = 0; r < R; r++) {
                const bool cst = yconst[r] || zcur < 0;
                float4 v = S.v[r];
                float eg[NE];
#pragma unroll
                for (int k = 0; k < NE; k++) eg[k] = (edge_load && !cst) ? pick<NE>(S.t[r], eidx[k]) : p.cval;
                if (cst) v = make_float4(p.cval, p.cval, p.cval, p.cval);
                const float4 xf = (p.dbg & 1) ? v : xpass_dpp<WX>(v, eg, lane, last, p.wx);
                if (more && !(p.dbg & 4)) load_row(znext, r, S);
                if (emit) {
                    float4 a;
                    if constexpr (WZ == 1) {
                        a = make_float4(p.wz[0] * xf.x, p.wz[0] * xf.y, p.wz[0] * xf.z, p.wz[0] * xf.w);
                    } else {
                        a = make_float4(p.wz[0] * ring[0][r].x, p.wz[0] * ring[0][r].y, p.wz[0] * ring[0][r].z,
                                        p.wz[0] * ring[0][r].w);
#pragma unroll
                        for (int k = 1; k < WZ - 1; k++) {
                            a.x = fmaf(p.wz[k], ring[k][r].x, a.x);
                            a.y = fmaf(p.wz[k], ring[k][r].y, a.y);
                            a.z = fmaf(p.wz[k], ring[k][r].z, a.z);
                            a.w = fmaf(p.wz[k], ring[k][r].w, a.w);
                        }
                        a.x = fmaf(p.wz[WZ - 1], xf.x, a.x);
                        a.y = fmaf(p.wz[WZ - 1], xf.y, a.y);
                        a.z = fmaf(p.wz[WZ - 1], xf.z, a.z);
                        a.w = fmaf(p.wz[WZ - 1], xf.w, a.w);
                    }
                    if (p.dbg & 1) a = xf;
                    // a constant-mode row is exactly cval at the y stage
                    if (yconst[r]) a = make_float4(p.cval, p.cval, p.cval, p.cval);
                    if (yused[r]) wbuf[(wave * R + r) * 64 + lane] = a;
                }
                if constexpr (WZ > 1) {
#pragma unroll
                    for (int k = 0; k < RING - 1; k++) ring[k][r] = ring[k + 1][r];
                    ring[RING - 1][r] = xf;
                }
            }
            if (emit) buf ^= 1;
            __syncthreads();
        };

        RowRegs<R, NE> A, B;
#pragma unroll
        for (int r = 0; r < R; r++) {
            load_row(ztab[0], r, A);
            load_row(ztab[nsteps > 1 ? 1 : 0], r, B);
        }
        for (int i = 0; i < nsteps; i += 2) {
            step(i, A);
            if (i + 1 < nsteps) step(i + 1, B);
        }
    } else {
        // ------------------------------------------------------------ consumer
        const int cw = wave - NWP;
        const bool active = lane < nlanes;
        int buf = 0;
        for (int i = 0; i < nsteps; i++) {
            __syncthreads();
            if (i < WZ - 1) continue;
            const int zo = zs + i - (WZ - 1);
            const float4 *rbuf = lds + buf * (ROWS * 64);
            buf ^= 1;
            float *oplane = out + (int64_t)zo * plane;
            for (int j = cw; j < ty_act; j += NWC) {
                const float4 *src = rbuf + j * 64 + lane;
                float4 a = src[0];
                a.x *= p.wyv[0]; a.y *= p.wyv[0]; a.z *= p.wyv[0]; a.w *= p.wyv[0];
                for (int k = 1; k < ((p.dbg & 8) ? 1 : p.wy); k++) {
                    const float4 t = src[k * 64];
                    const float w = p.wyv[k];
                    a.x = fmaf(w, t.x, a.x);
                    a.y = fmaf(w, t.y, a.y);
                    a.z = fmaf(w, t.z, a.z);
                    a.w = fmaf(w, t.w, a.w);
                }
                if (active && !(p.dbg & 2)) *reinterpret_cast<float4 *>(oplane + (int64_t)(y0 + j) * nx + x0 + 4 * lane) = a;
            }
        }
    }
}

template <int WX, int WZ, int NWP, int NWC, int R>
static int launch_sep3d_ws(const float *in, float *out, const Sep3dParams &p, hipStream_t s)
{
    const size_t lds = (size_t)2 * NWP * R * 64 * sizeof(float4) + (size_t)(kMaxChunk + kMaxTaps) * sizeof(int);
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)sep3d_ws_kernel<WX, WZ, NWP, NWC, R>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int total = p.nxt * p.nyt * p.nzc;
    note_kernel("mi::sep3d_ws_kernel<%d,%d,%d,%d,%d> grid=%d", WX, WZ, NWP, NWC, R, total);
    hipLaunchKernelGGL((sep3d_ws_kernel<WX, WZ, NWP, NWC, R>), dim3(total), dim3((NWP + NWC) * 64), lds, s, in,
                       out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------
// v4 "lean": the shipped kernel for cubic kernels (same odd tap count W on all
// three axes: uniform_filter(size=W), isotropic gaussian_filter).  Same
// producer / consumer structure as sep3d_ws_kernel, with the per-voxel
// instruction overhead stripped:
//   * buffer_load / buffer_store with one SRSRC descriptor per plane (scalar
//     base + range), the row + lane offset a loop-invariant VGPR, so the
//     loop contains no address arithmetic; lanes / rows that must not touch
//     memory get an out-of-range voffset (hardware range check: loads return 0
//     without a fetch, stores are dropped) -- no exec-mask branches;
//   * the z ring is rotated by unrolling W-1 steps with compile-time slot
//     numbers instead of moving registers;
//   * consumers read their G + W - 1 LDS rows once and slide over them.
//   * packed fp32 math (v_pk_fma_f32) in all three passes.
// Precondition (host): one plane < 2 GiB (32-bit offsets inside a plane).
// ---------------------------------------------------------------------------
// x pass with the tile-edge halo given as wave-uniform scalars (sL: the NE
// floats left of the tile, sR: the NE floats right of it), packed math.
// Window positions p = -RXE .. 3 + RXE (RXE = halo rounded up to an even
// count) live in e[p + RXE]; aligned pairs
// A[m] = (e[2m], e[2m+1]) feed the taps at even distance, the shifted pairs
// S[m] = (e[2m+1], e[2m+2]) (one v_pk_mov each) the taps at odd distance, so an
// output pair costs WX v_pk_fma instead of 2 WX v_fma.
template <int WX, int NE>
__device__ __forceinline__ F4 xpass_packed(const F4 v, const float (&sL)[NE], const float (&sR)[NE], int lane, int last,
                                           const float *__restrict__ wx)
{
    constexpr int RX = WX / 2;
    constexpr int RXE = (RX + 1) & ~1;
    constexpr int NP = (4 + 2 * RXE) / 2;
    const float c[4] = {v.lo.x, v.lo.y, v.hi.x, v.hi.y};
    float e[4 + 2 * RXE];
#pragma unroll
    for (int j = 0; j < 4 + 2 * RXE; j++) e[j] = 0.f;
#pragma unroll
    for (int j = 0; j < RX; j++) {
        const float l = dpp_from_left(0.f, c[4 - RX + j]);
        const float r = dpp_from_right(0.f, c[j]);
        e[RXE - RX + j] = lane == 0 ? sL[NE - RX + j] : l;
        e[RXE + 4 + j] = lane == last ? sR[j] : r;
    }
    f32x2 A[NP], S[NP - 1];
#pragma unroll
    for (int m = 0; m < NP; m++) A[m] = (f32x2){e[2 * m], e[2 * m + 1]};
    A[RXE / 2] = v.lo;
    A[RXE / 2 + 1] = v.hi;
#pragma unroll
    for (int m = 0; m < NP - 1; m++) S[m] = (f32x2){A[m].y, A[m + 1].x};
    F4 o;
    {
        constexpr int d0 = RXE - RX;
        o.lo = splat2(wx[0]) * ((d0 & 1) ? S[d0 / 2] : A[d0 / 2]);
        o.hi = splat2(wx[0]) * ((d0 & 1) ? S[d0 / 2 + 1] : A[d0 / 2 + 1]);
    }
    static_for<WX - 1>([&](auto KK) {
        constexpr int k = decltype(KK)::value + 1;
        constexpr int d = k - RX + RXE;
        o.lo = fma2(splat2(wx[k]), (d & 1) ? S[d / 2] : A[d / 2], o.lo);
        o.hi = fma2(splat2(wx[k]), (d & 1) ? S[d / 2 + 1] : A[d / 2 + 1], o.hi);
    });
    return o;
}

// r6: OP 1 / 2 = flat minimum / maximum instead of the weighted sum (the ragged build only: grey erosion / dilation and
// min / max filters with a cubic `size` on rows that are not a multiple of 16 bytes used to run on explicitly extended
// rows -- mi_extend_rows + the LDS-DMA kernel + mi_crop_rows, 55 us on 181 x 217 x 181 where uniform_filter(3) takes 16).
// A pass is the compare-select form of the generic kernels (`x < best ? x : best` in ascending tap order, first tap
// taken as is) evaluated as a v_min3 / v_max3 chain plus the first-tap NaN fix-up of minmax3d_f32.hip.
template <bool IS_MAX, int N>
__device__ __forceinline__ float lean_win(const float (&t)[N])
{
    float r = t[0];
    static_for<(N - 1) / 2>([&](auto KK) {
        constexpr int k = decltype(KK)::value;
        r = IS_MAX ? __builtin_fmaxf(__builtin_fmaxf(r, t[1 + 2 * k]), t[2 + 2 * k]) : __builtin_fminf(__builtin_fminf(r, t[1 + 2 * k]), t[2 + 2 * k]);
    });
    if constexpr ((N - 1) % 2 == 1) r = IS_MAX ? __builtin_fmaxf(r, t[N - 1]) : __builtin_fminf(r, t[N - 1]);
    return t[0] != t[0] ? t[0] : r;
}
template <bool IS_MAX, int N>
__device__ __forceinline__ F4 lean_win4(const F4 (&t)[N])
{
    float a[N], b[N], c[N], d[N];
#pragma unroll
    for (int k = 0; k < N; k++) { a[k] = t[k].lo.x; b[k] = t[k].lo.y; c[k] = t[k].hi.x; d[k] = t[k].hi.y; }
    F4 o;
    o.lo = (f32x2){lean_win<IS_MAX, N>(a), lean_win<IS_MAX, N>(b)};
    o.hi = (f32x2){lean_win<IS_MAX, N>(c), lean_win<IS_MAX, N>(d)};
    return o;
}
template <int WX, int NE, bool IS_MAX>
__device__ __forceinline__ F4 xpass_minmax(const F4 v, const float (&sL)[NE], const float (&sR)[NE], int lane, int last)
{
    constexpr int RX = WX / 2;
    const float c[4] = {v.lo.x, v.lo.y, v.hi.x, v.hi.y};
    float e[4 + 2 * RX];
#pragma unroll
    for (int j = 0; j < 4; j++) e[RX + j] = c[j];
#pragma unroll
    for (int j = 0; j < RX; j++) {
        const float l = dpp_from_left(0.f, c[4 - RX + j]);
        const float r = dpp_from_right(0.f, c[j]);
        e[j] = lane == 0 ? sL[NE - RX + j] : l;
        e[RX + 4 + j] = lane == last ? sR[j] : r;
    }
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        float t[WX];
#pragma unroll
        for (int k = 0; k < WX; k++) t[k] = e[q + k];
        o[q] = lean_win<IS_MAX, WX>(t);
    }
    F4 r;
    r.lo = (f32x2){o[0], o[1]};
    r.hi = (f32x2){o[2], o[3]};
    return r;
}

// RAGGED (r5): rows of any length >= 16 (181 x 217 x 181: 724-byte rows).  Rows then start on 4-byte boundaries only -- the
// 16-byte buffer loads and stores do not mind -- and the LAST lane of the last x tile holds `tail` (1..3) floats of its row
// followed by the head of the next one: those are replaced by the row's boundary continuation E[0], E[1], ... before the
// x pass (E[0..7] come with the edge load: lanes 32 + r and 48 + r, four floats each), the floats right of the tile are
// E[4 - tail ..], and the last lane stores `tail` floats.  No mi_extend_rows / mi_crop_rows copies around the launch.
template <int W, int NWP, int NWC, int R, int DEPTH, bool HAS_CONST, bool RAGGED = false, int OP = 0>
__global__ void __launch_bounds__((NWP + NWC) * 64)
sep3d_lean_kernel(const float *__restrict__ in, float *__restrict__ out, const Sep3dParams p)
{
    constexpr int ROWS = NWP * R;
    constexpr int TY = ROWS - (W - 1);
    constexpr int G = (TY + NWC - 1) / NWC;              // output rows per consumer wave
    constexpr int LROWS = (NWC * G + W - 1) > ROWS ? (NWC * G + W - 1) : ROWS;
    constexpr int RX = W / 2;
    constexpr int NE = RAGGED ? 4 : (RX <= 2 ? 2 : 4);
    constexpr int RINGN = W - 1;                          // even (W odd), >= 2
    static_assert(W >= 3 && W <= 9 && (W & 1) && (W <= 7 || RAGGED), "lean kernel: odd W, 3 .. 7 (9 .. 17: sep3d_long.hip; r5: 9 for ragged rows)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *lds = reinterpret_cast<float4 *>(smem);                     // [2][LROWS][64]
    int *ztab = reinterpret_cast<int *>(smem + (size_t)2 * LROWS * 1024); // source plane per step

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
    const int x0 = xt * p.tw, y0 = yt * TY;
    int zs, ze;
    chunk_planes(p, zci, &zs, &ze);
    const int ty_act = min(TY, ny - y0);
    const int rows_needed = ty_act + W - 1;
    const int width = min(p.tw, nx - x0);
    const int nlanes = RAGGED ? (width + 3) >> 2 : min(p.tw >> 2, (nx - x0) >> 2);
    const int last = nlanes - 1;
    const int tail = RAGGED ? width - 4 * last : 4;       // floats of its row the last lane holds
    // one buffer descriptor per plane (base = plane start, range = one plane):
    // offsets stay 32-bit inside a plane, the volume itself may exceed 4 GiB
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * 4u;
    const size_t plane_elems = (size_t)ny * (size_t)nx;
    const int nsteps = ze - zs + W - 1;
    // Every other z chunk streams DOWNWARDS.  Neighbouring chunks re-read each other's W - 1 ramp planes; with
    // alternating directions the two chunks that share a boundary are both there at the same time (both start or
    // both end at it), so the second read is served by the L2 / the memory-side cache instead of HBM again.  The
    // z taps are still accumulated in ascending tap order (see `rev` below): results do not depend on the direction.
    const bool rev = p.zrev && (((zci >= p.nzc0 ? zci - p.nzc0 : zci) & 1) != 0);
    const int zi0 = rev ? ze - 1 - p.oz + (W - 1) : zs - p.oz;      // input plane of step 0
    const int zdir = rev ? -1 : 1;

    for (int i = threadIdx.x; i < nsteps; i += (NWP + NWC) * 64) ztab[i] = bmap<int>(zi0 + zdir * i, nz, p.mz);
    __syncthreads();

    if (wave < NWP) {
        // ------------------------------------------------------------ producer
        static_assert(R <= 16, "edge lanes");
        // Tile-edge halo: ONE buffer_load per plane fetches the edges of all R
        // rows -- lane r (< R) loads the NE floats left of the tile for row r,
        // lane 32 + r the NE floats right of it; every other lane is out of
        // range (no fetch).  Each row's x pass then reads its two edge vectors
        // with v_readlane into scalars.
        int es0, ek0, es1, ek1;
        edge_desc(0, x0, x0 + 4 * nlanes, nx, p.mx, &es0, &ek0);
        edge_desc(1, x0, x0 + width, nx, p.mx, &es1, &ek1);
        const bool left_side = lane < 32;
        const int erow = left_side ? lane : (RAGGED ? (lane & 15) : lane - 32);     // row this lane fetches the edge of
        const int ekind = left_side ? ek0 : ek1;
        int eoff = left_side ? es0 : es1;
        // RAGGED: lanes 32 + r fetch what replaces floats 1..3 of the last lane (slot j = E[j - tail]: the standard four floats
        // E[0..3], re-indexed below), lanes 48 + r the floats right of the row as the x pass wants them (slot j = E[4 - tail + j]:
        // the load moved by 4 - tail floats) -- everything that depends on `tail` is settled here, once per workgroup
        if constexpr (RAGGED)
            if (lane >= 48) eoff += ekind == EDGE_FWD ? 4 - tail : (ekind == EDGE_REV ? tail - 4 : 0);
        if (left_side) {
            if (ekind == EDGE_FWD) eoff += 4 - NE;
        } else {
            if (ekind == EDGE_REV) eoff += 4 - NE;
            if (ekind == EDGE_SPLAT) eoff -= NE - 1;
        }
        int eidx[NE];
#pragma unroll
        for (int k = 0; k < NE; k++) {
            eidx[k] = ekind == EDGE_FWD ? k : (ekind == EDGE_REV ? NE - 1 - k : (left_side ? 0 : NE - 1));
            if constexpr (RAGGED)
                if (lane >= 32 && lane < 48 && ekind != EDGE_SPLAT)
                    eidx[k] = ekind == EDGE_FWD ? max(k - tail, 0) : min(NE - 1 - k + tail, NE - 1);
        }
        const bool patch1 = RAGGED && lane == last && tail < 2, patch2 = RAGGED && lane == last && tail < 3,
                   patch3 = RAGGED && lane == last && tail < 4;

        unsigned voff[R];
        unsigned eoffv = kOOB;
        bool yconst[R];
        bool e_is_cval = (ekind == EDGE_CONST);   // this lane's edge value is cval
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int rr = wave * R + r;
            const int ys = rr < rows_needed ? bmap<int>(y0 - p.oy + rr, ny, p.my) : -2;
            yconst[r] = ys == -1;
            voff[r] = (ys >= 0 && lane < nlanes) ? (unsigned)(ys * nx + x0 + 4 * lane) * 4u : kOOB;
            if (erow == r) {
                if (ys >= 0 && ekind != EDGE_CONST) eoffv = (unsigned)(ys * nx + eoff) * 4u;
                if (ys == -1) e_is_cval = true;
            }
        }

        struct Regs { F4 v[R]; float t[NE]; bool zconst; };
        Regs S[DEPTH];
        auto issue = [&](int i, Regs &s) {
            int zsrc = zi0 + zdir * i;
            if ((unsigned)zsrc >= (unsigned)nz) zsrc = ztab[i];
            s.zconst = zsrc < 0;
            zsrc = __builtin_amdgcn_readfirstlane(max(zsrc, 0));
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(in + (size_t)zsrc * plane_elems), 0, (int)plane_bytes, 0x00020000);
            const bool skip = HAS_CONST && s.zconst;
            // tuning knob (mi_debug_set_sep3d_dbg(256 * k)): cache policy of the row loads -- 0 default, 1 nt, 2 sc1, 3 sc0 sc1
            const int ldpol = (p.dbg >> 8) & 3;
            if (ldpol == 0) {
#pragma unroll
                for (int r = 0; r < R; r++) s.v[r] = f4_from(__builtin_amdgcn_raw_buffer_load_b128(rin, skip ? kOOB : voff[r], 0, 0));
            } else if (ldpol == 1) {
#pragma unroll
                for (int r = 0; r < R; r++) s.v[r] = f4_from(__builtin_amdgcn_raw_buffer_load_b128(rin, skip ? kOOB : voff[r], 0, 2));
            } else if (ldpol == 2) {
#pragma unroll
                for (int r = 0; r < R; r++) s.v[r] = f4_from(__builtin_amdgcn_raw_buffer_load_b128(rin, skip ? kOOB : voff[r], 0, 16));
            } else {
#pragma unroll
                for (int r = 0; r < R; r++) s.v[r] = f4_from(__builtin_amdgcn_raw_buffer_load_b128(rin, skip ? kOOB : voff[r], 0, 17));
            }
            if constexpr (NE == 2) {
                const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(rin, skip ? kOOB : eoffv, 0, 0);
                s.t[0] = __uint_as_float(q.x); s.t[1] = __uint_as_float(q.y);
            } else {
                const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rin, skip ? kOOB : eoffv, 0, 0);
                s.t[0] = __uint_as_float(q.x); s.t[1] = __uint_as_float(q.y);
                s.t[2] = __uint_as_float(q.z); s.t[3] = __uint_as_float(q.w);
            }
        };

        F4 ring[RINGN][R];
#pragma unroll
        for (int k = 0; k < RINGN; k++)
#pragma unroll
            for (int r = 0; r < R; r++) ring[k][r] = f4_splat(0.f);

        issue(0, S[0]);
        if constexpr (DEPTH == 2) if (nsteps > 1) issue(1, S[1]);

        for (int i0 = 0; i0 < nsteps; i0 += RINGN) {
            static_for<RINGN>([&](auto JJ) {
                constexpr int J = decltype(JJ)::value;
                const int i = i0 + J;
                if (i < nsteps) {
                    Regs &s = S[DEPTH == 2 ? (J & 1) : 0];
                    const bool emit = i >= W - 1;
                    float4 *wbuf = lds + (J & 1) * (LROWS * 64) + (wave * R) * 64 + lane;
                    // this lane's edge floats, ordered / replaced by cval as the boundary mode wants
                    float eg[NE];
#pragma unroll
                    for (int k = 0; k < NE; k++) {
                        eg[k] = pick<NE>(s.t, eidx[k]);
                        if constexpr (HAS_CONST) eg[k] = (e_is_cval || s.zconst) ? p.cval : eg[k];
                    }
                    F4 xf[R];
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        F4 v = s.v[r];
                        if constexpr (HAS_CONST)
                            if (yconst[r] || s.zconst) v = f4_splat(p.cval);
                        float sL[NE], sR[NE];
#pragma unroll
                        for (int k = 0; k < NE; k++) {
                            sL[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(eg[k]), r));
                            sR[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(eg[k]), 32 + r));
                        }
                        if constexpr (RAGGED) {
                            v.lo.y = patch1 ? sR[1] : v.lo.y;
                            v.hi.x = patch2 ? sR[2] : v.hi.x;
                            v.hi.y = patch3 ? sR[3] : v.hi.y;
#pragma unroll
                            for (int k = 0; k < RX; k++) sR[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(eg[k]), 48 + r));
                        }
                        if constexpr (OP == 0) xf[r] = xpass_packed<W, NE>(v, sL, sR, lane, last, p.wx);
                        else xf[r] = xpass_minmax<W, NE, OP == 2>(v, sL, sR, lane, last);
                    }
                    if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                    if (emit) {
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            F4 a;
                            if constexpr (OP != 0) {
                                F4 t[W];          // the W planes of the window in ascending z
                                if (rev) {
                                    t[0] = xf[r];
#pragma unroll
                                    for (int k = 1; k < W; k++) t[k] = ring[(J + RINGN - k) % RINGN][r];
                                } else {
#pragma unroll
                                    for (int k = 0; k < RINGN; k++) t[k] = ring[(J + k) % RINGN][r];
                                    t[W - 1] = xf[r];
                                }
                                a = lean_win4<OP == 2, W>(t);
                            } else if (rev) {          // the newest plane is the lowest: tap 0 first, the ring newest to oldest
                                a = f4_scale(p.wz[0], xf[r]);
#pragma unroll
                                for (int k = 1; k < W; k++) a = f4_fma(p.wz[k], ring[(J + RINGN - k) % RINGN][r], a);
                            } else {
                                a = f4_scale(p.wz[0], ring[J % RINGN][r]);
#pragma unroll
                                for (int k = 1; k < RINGN; k++) a = f4_fma(p.wz[k], ring[(J + k) % RINGN][r], a);
                                a = f4_fma(p.wz[W - 1], xf[r], a);
                            }
                            if constexpr (HAS_CONST)
                                if (yconst[r]) a = f4_splat(p.cval);
                            wbuf[r * 64] = f4_to_float4(a);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < R; r++) ring[J % RINGN][r] = xf[r];
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
            ovoff[g] = (j0 + g < ty_act && lane < nlanes) ? (unsigned)((y0 + j0 + g) * nx + x0 + 4 * lane) * 4u : kOOB;
        const bool part = RAGGED && lane == last && tail < 4;      // this lane stores `tail` floats
        for (int i = 0; i < nsteps; i++) {
            __syncthreads();
            if (i < W - 1) continue;
            const int zo = rev ? ze - 1 - (i - (W - 1)) : zs + i - (W - 1);
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(out + (size_t)zo * plane_elems), 0, (int)plane_bytes, 0x00020000);
            const float4 *rbuf = lds + (i & 1) * (LROWS * 64) + j0 * 64 + lane;
            F4 win[G + W - 1];
#pragma unroll
            for (int k = 0; k < G + W - 1; k++) win[k] = f4_from(rbuf[k * 64]);
#pragma unroll
            for (int g = 0; g < G; g++) {
                F4 a;
                if constexpr (OP != 0) {
                    F4 t[W];
#pragma unroll
                    for (int k = 0; k < W; k++) t[k] = win[g + k];
                    a = lean_win4<OP == 2, W>(t);
                } else {
                    a = f4_scale(p.wyv[0], win[g]);
#pragma unroll
                    for (int k = 1; k < W; k++) a = f4_fma(p.wyv[k], win[g + k], a);
                }
                // written once, never read back by this launch: non-temporal (measured 1.2 % on config H)
                if constexpr (RAGGED) {
                    const u32x4 q = f4_to_u32(a);
                    if (!part) {
                        __builtin_amdgcn_raw_buffer_store_b128(q, rout, ovoff[g], 0, 2);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(q.x, rout, ovoff[g], 0, 2);
                        if (tail > 1) __builtin_amdgcn_raw_buffer_store_b32(q.y, rout, ovoff[g] + 4u, 0, 2);
                        if (tail > 2) __builtin_amdgcn_raw_buffer_store_b32(q.z, rout, ovoff[g] + 8u, 0, 2);
                    }
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(f4_to_u32(a), rout, ovoff[g], 0, 2);
                }
            }
        }
    }
}

static Knob g_sep3d_ragged{1};    // r5: 1 = the lean kernel takes rows that are not a multiple of 4 floats itself, 0 = refuse (callers extend the rows), 2 = the ragged build for every row length (measurement)
// RG: the tile shape is also built for rows that are not a multiple of 4 floats (the shapes choose_plan picks by itself)
template <int W, int NWP, int NWC, int R, int DEPTH = 2, bool RG = false>
static int launch_sep3d_lean(const float *in, float *out, Sep3dParams &p, bool has_const, hipStream_t s, int op = 0)
{
    constexpr int ROWS = NWP * R;
    constexpr int TY = ROWS - (W - 1);
    constexpr int G = (TY + NWC - 1) / NWC;
    constexpr int LROWS = (NWC * G + W - 1) > ROWS ? (NWC * G + W - 1) : ROWS;
    const size_t lds = (size_t)2 * LROWS * 1024 + (size_t)(kMaxChunk + kMaxTaps) * sizeof(int);
    constexpr bool kAligned = W <= 7;            // 9 taps: the ragged build only (aligned rows take sep3d_long.hip)
    static PerDeviceOnce attr_done;
    if constexpr (kAligned) {
        if (!attr_done) {
            MI_HIP(hipFuncSetAttribute((const void *)sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            MI_HIP(hipFuncSetAttribute((const void *)sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_done = true;
        }
    }
    if (p.ty != TY) { set_error("internal: lean tile mismatch"); return MI_ERR_INTERNAL; }
    const int total = p.nxt * p.nyt * p.nzc;
    if (op != 0) {
        // flat min / max (r6): the ragged build with every boundary test compiled in (HAS_CONST), rows of any length
        if constexpr (RG) {
            static PerDeviceOnce mm_done;
            if (!mm_done) {
                MI_HIP(hipFuncSetAttribute((const void *)sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, true, true, 1>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                MI_HIP(hipFuncSetAttribute((const void *)sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, true, true, 2>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                mm_done = true;
            }
            note_kernel("mi::sep3d_lean_kernel<%d,%d,%d,%d,%d,true,ragged,%s> grid=%d (fused x/z/y flat %s of %d^3 samples, rows of any length)",
                        W, NWP, NWC, R, DEPTH, op == 2 ? "max" : "min", total, op == 2 ? "maximum" : "minimum", W);
            if (op == 2)
                hipLaunchKernelGGL((sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, true, true, 2>), dim3(total), dim3((NWP + NWC) * 64), lds, s, in, out, p);
            else
                hipLaunchKernelGGL((sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, true, true, 1>), dim3(total), dim3((NWP + NWC) * 64), lds, s, in, out, p);
            MI_HIP(hipGetLastError());
            return MI_OK;
        } else {
            set_error("separable3d: min / max runs on the ragged tile shapes only");
            return MI_ERR_UNSUPPORTED;
        }
    }
    if ((p.nx & 3) || (RG && g_sep3d_ragged == 2)) {
        if constexpr (RG) {
            static PerDeviceOnce rg_done;
            if (!rg_done) {
                MI_HIP(hipFuncSetAttribute((const void *)sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, false, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                MI_HIP(hipFuncSetAttribute((const void *)sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, true, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                rg_done = true;
            }
            note_kernel("mi::sep3d_lean_kernel<%d,%d,%d,%d,%d,%s,ragged> grid=%d (fused x/z/y separable pass, rows of any length)", W, NWP, NWC,
                        R, DEPTH, has_const ? "true" : "false", total);
            if (has_const)
                hipLaunchKernelGGL((sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, true, true>), dim3(total), dim3((NWP + NWC) * 64), lds, s, in, out, p);
            else
                hipLaunchKernelGGL((sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, false, true>), dim3(total), dim3((NWP + NWC) * 64), lds, s, in, out, p);
            MI_HIP(hipGetLastError());
            return MI_OK;
        } else {
            set_error("separable3d: this tile shape needs rows that are a multiple of 4 floats");
            return MI_ERR_UNSUPPORTED;
        }
    }
    if constexpr (kAligned) {
        note_kernel("mi::sep3d_lean_kernel<%d,%d,%d,%d,%d,%s> grid=%d (fused x/z/y separable pass)", W, NWP, NWC, R, DEPTH,
                    has_const ? "true" : "false", total);
        if (has_const)
            hipLaunchKernelGGL((sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, true>), dim3(total), dim3((NWP + NWC) * 64), lds, s, in, out, p);
        else
            hipLaunchKernelGGL((sep3d_lean_kernel<W, NWP, NWC, R, DEPTH, false>), dim3(total), dim3((NWP + NWC) * 64), lds, s, in, out, p);
        MI_HIP(hipGetLastError());
        return MI_OK;
    } else {
        set_error("separable3d: this tap count runs here on ragged rows only");
        return MI_ERR_UNSUPPORTED;
    }
}

// ---------------------------------------------------------------------------
// host side: kernel / tile selection
// ---------------------------------------------------------------------------
// float4 copies: the practical HBM ceiling on this chip for the same byte count.  copy_f4_kernel is the plain grid-stride
// copy of rounds 1-3 (5.8 TB/s); copy_f4_nt_kernel (r4, scripts/diag/copy_bw.hip) keeps four 16-byte loads in flight per
// thread and marks loads AND stores non-temporal: 6.4-6.5 TB/s, the best streaming copy found on this chip (read only:
// 6.5 TB/s, write only: 4.7-5.6 TB/s -- a 1 : 1 read / write stream cannot do better than ~6.5).
__global__ void __launch_bounds__(256) copy_f4_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = in[i];
}
typedef unsigned int u32x4cp __attribute__((ext_vector_type(4)));
// (buffer instructions: the same copy with global_load / global_store and the nt hint moves 6.0-6.25 TB/s, copy_bw.hip)
__global__ void __launch_bounds__(256) copy_f4_nt_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t n4)
{
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(n4 * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)(n4 * 16), 0x00020000);
    const int64_t span = (int64_t)blockDim.x * 4;
    for (int64_t b = (int64_t)blockIdx.x * span; b < n4; b += (int64_t)gridDim.x * span) {
        u32x4cp v[4];
#pragma unroll
        for (int u = 0; u < 4; u++)       // beyond the end: the descriptor's range check (reads zero, writes nothing)
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, (unsigned)(b + (int64_t)u * blockDim.x + threadIdx.x) * 16u, 0, 2);
#pragma unroll
        for (int u = 0; u < 4; u++)
            __builtin_amdgcn_raw_buffer_store_b128(v[u], ro, (unsigned)(b + (int64_t)u * blockDim.x + threadIdx.x) * 16u, 0, 2);
    }
}

// general kernel (any odd taps <= 9 per axis, per-axis origins on y/z, >= 2 GiB
// volumes): rows staged per plane for a given z tap count
static int ws_rows(int wz) { return wz <= 5 ? 24 : (wz <= 7 ? 24 : 16); }

template <int WX, int WZ>
static int launch_ws(const float *in, float *out, const Sep3dParams &p, hipStream_t s)
{
    if constexpr (WZ <= 5) return launch_sep3d_ws<WX, WZ, 8, 4, 3>(in, out, p, s);
    else if constexpr (WZ <= 7) return launch_sep3d_ws<WX, WZ, 8, 4, 3>(in, out, p, s);
    else return launch_sep3d_ws<WX, WZ, 8, 4, 2>(in, out, p, s);
}

// lean kernel tile shapes per tap count; cfg is a tuning knob
static int lean_rows(int w, int cfg)
{
    switch (w) {
    case 3: return cfg == 1 ? 32 : (cfg == 5 ? 20 : 36);
    case 5: return cfg == 2 ? 24 : (cfg == 3 ? 30 : (cfg >= 4 && cfg <= 6 ? 20 : 36));
    case 7: return 24;
    case 9: return 18;
    default: return 24;
    }
}

static int launch_lean(int w, int cfg, const float *in, float *out, Sep3dParams &p, bool hc, hipStream_t s, int op = 0)
{
    if (op != 0) {            // flat min / max: the tile shapes that have a ragged build
        switch (w) {
        case 3: return cfg == 5 ? launch_sep3d_lean<3, 10, 4, 2, 2, true>(in, out, p, hc, s, op) : launch_sep3d_lean<3, 12, 4, 3, 2, true>(in, out, p, hc, s, op);
        case 5: return cfg == 5 ? launch_sep3d_lean<5, 10, 4, 2, 2, true>(in, out, p, hc, s, op) : launch_sep3d_lean<5, 12, 4, 3, 1, true>(in, out, p, hc, s, op);
        default: return launch_sep3d_lean<7, 8, 4, 3, 2, true>(in, out, p, hc, s, op);
        }
    }
    switch (w) {
    case 3:
        if (cfg == 1) return launch_sep3d_lean<3, 8, 4, 4>(in, out, p, hc, s);
        if (cfg == 5) return launch_sep3d_lean<3, 10, 4, 2, 2, true>(in, out, p, hc, s);
        return launch_sep3d_lean<3, 12, 4, 3, 2, true>(in, out, p, hc, s);
    case 5:
        if (cfg == 1) return launch_sep3d_lean<5, 9, 3, 4>(in, out, p, hc, s);
        if (cfg == 2) return launch_sep3d_lean<5, 8, 4, 3>(in, out, p, hc, s);
        if (cfg == 3) return launch_sep3d_lean<5, 10, 2, 3>(in, out, p, hc, s);
        if (cfg == 4) return launch_sep3d_lean<5, 10, 2, 2>(in, out, p, hc, s);
        if (cfg == 5) return launch_sep3d_lean<5, 10, 4, 2, 2, true>(in, out, p, hc, s);
        if (cfg == 6) return launch_sep3d_lean<5, 10, 6, 2>(in, out, p, hc, s);
        if (cfg == 8) return launch_sep3d_lean<5, 12, 4, 3, 2>(in, out, p, hc, s);
        return launch_sep3d_lean<5, 12, 4, 3, 1, true>(in, out, p, hc, s);   // measured best: 1 WG/CU, 16 waves
    case 9:
        // r5: rows that are not a multiple of four floats only (gaussian sigma 1 on 181 x 217 x 181: aligned rows take sep3d_long.hip)
        if (!(p.nx & 3)) { set_error("separable3d: 9 taps on aligned rows belong to the long kernel"); return MI_ERR_UNSUPPORTED; }
        return launch_sep3d_lean<9, 9, 3, 2, 2, true>(in, out, p, hc, s);        // 12 waves: 168 registers a lane (16 waves: 21 spilled)
    default:
        return launch_sep3d_lean<7, 8, 4, 3, 2, true>(in, out, p, hc, s);
    }
}

int run_stream_pass(const float *in, float *out, int nz, int ny, int nx, int axis, const float *wav, int wa, int oa,
                    int ma, const float *wxv, int wx, int mx, float cval, hipStream_t s);   // stream3d.hip
int run_sep3d_long(const float *in, float *out, int nz, int ny, int nx, int w, int wzn, const float *wx, const float *wy,
                   const float *wz, int oy, int oz, int mx, int my, int mz, float cval, const int64_t zb[2],
                   const int64_t zn[2], hipStream_t s, bool ragged = false);   // sep3d_long.hip
bool long_aniso_pair(int w, int wzn);                     // sep3d_long.hip: (in-plane, z) tap pairs it is built for

// Tile / z-chunk choice by a small cost model.  One workgroup per CU is
// resident, so the launch runs in ceil(workgroups / CUs) rounds; a workgroup
// costs (chunk + w - 1) plane steps of (rows + fixed) row-units each.  Big
// tiles (fewer halo rows) win on large volumes, small tiles keep every CU
// busy on thin slabs (multi-GPU) and odd shapes.
static void choose_plan(int w, const int *cand_rows, const int *cand_cfg, int ncand, int wy, int64_t nz, int64_t ny,
                        int64_t nx, int *best_cfg, int *best_rows, int *best_nzc)
{
    const int ncu = device_cus();
    const int nxt = (int)((nx + 255) / 256);
    double best = 1e300;
    for (int c = 0; c < ncand; c++) {
        const int rows = cand_rows[c];
        const int ty = rows - (wy - 1);
        if (ty < 1) continue;
        const int cols = nxt * (int)((ny + ty - 1) / ty);
        const int max_nzc = (int)(nz < 64 ? nz : 64);
        for (int nzc = 1; nzc <= max_nzc; nzc++) {
            const int chunk = (int)((nz + nzc - 1) / nzc);
            if (chunk > kMaxChunk) continue;
            const int real_nzc = (int)((nz + chunk - 1) / chunk);
            const int64_t wgs = (int64_t)cols * real_nzc;
            const double rounds = (double)((wgs + ncu - 1) / ncu);
            const double cost = rounds * (chunk + w - 1) * (rows + 6.0);
            if (cost < best) { best = cost; *best_cfg = cand_cfg[c]; *best_rows = rows; *best_nzc = real_nzc; }
        }
    }
}

}  // namespace mi

using namespace mi;

// test / tuning hooks (not part of the C-ABI in include/mi355img.h)
static mi::Knob g_sep3d_cfg{0};       // tile shape variant
static mi::Knob g_sep3d_zchunks{0};   // 0 = heuristic
static mi::Knob g_sep3d_dbg{0};       // ablation flags
static mi::Knob g_sep3d_kernel{0};    // 0 = auto, 1 = force general (ws) kernel
static mi::Knob g_sep3d_zrev{0};      // 1 = odd z chunks of the lean kernel stream downwards (ramp planes shared in time); r3: off by default (measured 1-2 % slower on three boxes)
extern "C" int mi_debug_set_sep3d_zrev(int k) { g_sep3d_zrev = k; return MI_OK; }
static mi::Knob g_stream_fused_max{kStreamFusedMax};    // test hook: longest kernel with the x pass fused into the streamed pass
extern "C" int mi_debug_set_stream_fused_max(int k) { g_stream_fused_max = k; return MI_OK; }
static mi::Knob g_sep3d_image2d{1};   // test hook: 0 = images with <= 9 taps take the tiled volume kernel (round-1 behaviour)
extern "C" int mi_debug_set_sep3d_image2d(int k) { g_sep3d_image2d = k; return MI_OK; }
static mi::Knob g_sep3d_long{0};      // 0 = auto (cubic 9..17 taps; 3..7 taps where it measured faster, see separable3d_impl), 1 = off (lean kernel / streaming passes), 2 = always for 3..17 taps
extern "C" int mi_debug_set_sep3d_long(int k) { g_sep3d_long = k; return MI_OK; }
extern "C" int mi_debug_set_sep3d_cfg(int cfg) { g_sep3d_cfg = cfg; return MI_OK; }
extern "C" int mi_debug_set_sep3d_zchunks(int n) { g_sep3d_zchunks = n; return MI_OK; }
extern "C" int mi_debug_set_sep3d_dbg(int f) { g_sep3d_dbg = f; return MI_OK; }
extern "C" int mi_debug_set_sep3d_kernel(int k) { g_sep3d_kernel = k; return MI_OK; }
static mi::Knob g_sep3d_box{0};       // 0 = auto (running-sum box kernel where it applies), 1 = off
extern "C" int mi_debug_set_sep3d_ragged(int k) { g_sep3d_ragged = k; return MI_OK; }
extern "C" int mi_debug_set_sep3d_box(int k) { g_sep3d_box = k; return MI_OK; }
extern "C" int mi_debug_last_kernel(char *buf, size_t n)
{
    MI_REQUIRE(buf && n > 0, MI_ERR_INVALID_ARG, "NULL buffer");
    snprintf(buf, n, "%s", mi::t_last_kernel);
    return MI_OK;
}
extern "C" int mi_debug_copy_f32(const float *in, float *out, int64_t n, int blocks, mi_stream stream)
{
    hipLaunchKernelGGL(copy_f4_kernel, dim3(blocks), dim3(256), 0, resolve_stream(stream), (const float4 *)in,
                       (float4 *)out, n / 4);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
extern "C" int mi_debug_copy_f32_nt(const float *in, float *out, int64_t n, int blocks, mi_stream stream)
{
    MI_REQUIRE(n >= 0 && n * 4 < ((int64_t)1 << 32), MI_ERR_UNSUPPORTED, "copy comparator: arrays below 4 GiB");
    hipLaunchKernelGGL(copy_f4_nt_kernel, dim3(blocks), dim3(256), 0, resolve_stream(stream), in, out, n / 4);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// planes: nranges (<= 2) pairs [begin, end) of output planes to produce, or
// nullptr for the whole volume
static int separable3d_impl(const mi_array *in, const mi_array *out, const double *const weights[3],
                            const int wlen[3], const int origin[3], const int mode[3], double cval,
                            const int64_t *planes, int nranges, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(weights && wlen && origin && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("separable3d: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3 || in->dtype != MI_F32 || out->dtype != MI_F32) UNSUP("needs 3-D float32 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    // r5: rows that are not a multiple of 4 floats are taken by the lean kernel's ragged build (3 / 5 / 7 cubic taps, below)
    const bool ragged = (nx & 3) != 0;
    if (nz < 1 || ny < 1 || nx < 8 || (ragged && (nx < 16 || !g_sep3d_ragged))) UNSUP("x extent must be a multiple of 4, >= 8");
    if (nz * ny * nx >= ((int64_t)1 << 40) || nx > (1 << 24) || ny > (1 << 24) || nz > (1 << 24)) UNSUP("too large");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");

    Sep3dParams p;
    memset(&p, 0, sizeof(p));
    int w[3];
    bool normalised = true;
    float wbuf[3][kStreamMaxTaps];
    for (int a = 0; a < 3; a++) {
        w[a] = weights[a] ? wlen[a] : 1;
        if (w[a] < 1 || w[a] > kStreamMaxTaps || !(w[a] & 1)) UNSUP("taps must be odd and <= 33");
        const int off = w[a] / 2 + (weights[a] ? origin[a] : 0);
        if (off < 0 || off >= w[a]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
        double sum = 0.0;
        for (int k = 0; k < w[a]; k++) {
            const double v = weights[a] ? weights[a][k] : 1.0;
            wbuf[a][k] = (float)v;
            sum += v;
        }
        if (fabs(sum - 1.0) > 1e-6) normalised = false;
    }
    if (origin[2] != 0 && weights[2]) UNSUP("x origin must be 0");
    p.mz = filter_mode(mode[0]); p.my = filter_mode(mode[1]); p.mx = filter_mode(mode[2]);
    const bool any_const = p.mz == MI_MODE_CONSTANT || p.my == MI_MODE_CONSTANT || p.mx == MI_MODE_CONSTANT;
    if (any_const && !normalised) UNSUP("constant mode needs kernels that sum to one");

    // plane ranges: ascending, disjoint, inside the volume; empty ranges are dropped
    int64_t zb[2] = {0, 0}, zn[2] = {nz, 0};
    if (planes) {
        MI_REQUIRE(nranges >= 1 && nranges <= 2, MI_ERR_INVALID_ARG, "one or two plane ranges");
        zn[0] = 0;
        int k = 0;
        int64_t prev_end = 0;
        for (int r = 0; r < nranges; r++) {
            const int64_t b = planes[2 * r], e = planes[2 * r + 1];
            MI_REQUIRE(b >= prev_end && e >= b && e <= nz, MI_ERR_INVALID_ARG, "plane ranges must be ascending and inside the volume");
            prev_end = e;
            if (e > b) { zb[k] = b; zn[k] = e - b; k++; }
        }
        if (k == 0) return MI_OK;
    }
    const bool whole = !(t_dry_run && planes) && zb[0] == 0 && zn[0] == nz;
    const int64_t nzr = zn[0] + zn[1];

    const bool cubic_w = w[0] == w[1] && w[1] == w[2];
    // r6: 9 .. 17 cubic taps on such rows through the LDS-DMA kernel's ragged build (origins along y / z allowed there)
    const bool ragged_long = ragged && cubic_w && w[0] >= 9 && w[0] <= 17 && g_sep3d_ragged && g_sep3d_long != 1 &&
                             (!any_const || (float)cval == 0.0f) && ny * nx * 4 < ((int64_t)1 << 31);
    if (ragged && !ragged_long &&
        !(cubic_w && w[0] >= 3 && w[0] <= 9 && g_sep3d_kernel != 1 && g_sep3d_cfg == 0 && ny * nx * 4 < ((int64_t)1 << 31) &&
          (weights[0] ? origin[0] : 0) == 0 && (weights[1] ? origin[1] : 0) == 0))
        UNSUP("rows that are not a multiple of 4 floats: cubic kernels of 3 .. 17 taps only (3 .. 9: without origins)");
    if (ragged_long) {
        const int oz = w[0] / 2 + (weights[0] ? origin[0] : 0), oy = w[1] / 2 + (weights[1] ? origin[1] : 0);
        rc = run_sep3d_long((const float *)in->data, (float *)out->data, (int)nz, (int)ny, (int)nx, w[0], w[0], wbuf[2], wbuf[1],
                            wbuf[0], oy, oz, p.mx, p.my, p.mz, (float)cval, zb, zn, resolve_stream(stream), true);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
        // (per-axis weight vectors: 9 taps go on to the lean kernel's ragged build below, longer ones to the extended-rows route)
        if (!(w[0] == 9 && g_sep3d_kernel != 1 && g_sep3d_cfg == 0 && (weights[0] ? origin[0] : 0) == 0 && (weights[1] ? origin[1] : 0) == 0))
            UNSUP("rows that are not a multiple of 4 floats: the long kernel's ragged build refused");
    }
    // r3: with its re-scheduled instruction stream (sep3d_long3_kernel) the LDS-DMA kernel also beats the lean kernel
    // below 9 taps on volumes that fill the chip (profiles/r3_long3_small_taps.txt, sustained, lean -> long: 7 taps
    // 15-31 % faster on every shape of 4 Mvoxels and more; 5 and 3 taps 3-5 % faster on 512^3, 256^3, 64 x 1024^2 and
    // 1024 x 128^2, but 4-6 % SLOWER on 300^3 and 200 x 500 x 760, whose rows do not fill its 256-voxel wave tiles, and
    // equal or 3 % slower where the launch is latency bound).  Constant mode stays on the lean kernel below 9 taps.
    const int64_t nvox_out = nz * ny * nx;      // of the whole array: plane-range launches of one filter call take the same kernel
    // 3 / 5 taps (r4b, scripts/bench_long_rule.py -> profiles/r4_long_rule.txt): the long kernel also wins on rows that are not
    // multiples of 256 as long as its 256-float wave tiles are reasonably full (512 x 512 x 500: 171 against 189 us, 600^3:
    // 322 / 350, 512 x 500 x 512: 160 / 180; 300^3 with 150-float tiles: 46 against 42 -- not taken) and whatever ny is
    const int64_t nxt_l = (nx + 255) / 256;
    const bool tiles_full = nx * 10 >= nxt_l * 256 * 7;
    const bool long_small = (!any_const || (float)cval == 0.0f) && w[0] >= 3 && w[0] <= 7 && nx >= 128 && ny >= 16 &&      // r5: `constant` with a zero fill value is a mode like the others for that kernel
                            (w[0] == 7 ? nvox_out >= ((int64_t)1 << 22)
                                       : nvox_out >= ((int64_t)1 << 23) && (tiles_full || nx == 128));
    if (cubic_w && !ragged && g_sep3d_long != 1 && nx >= 16 &&
        ((w[0] >= 9 && w[0] <= 17) || (w[0] >= 3 && w[0] <= 7 && (g_sep3d_long == 2 || long_small)))) {
        // long cubic kernels: ONE launch with LDS-DMA staging and the z state in registers (sep3d_long.hip)
        const int oz = w[0] / 2 + (weights[0] ? origin[0] : 0), oy = w[1] / 2 + (weights[1] ? origin[1] : 0);
        rc = run_sep3d_long((const float *)in->data, (float *)out->data, (int)nz, (int)ny, (int)nx, w[0], w[0], wbuf[2], wbuf[1],
                            wbuf[0], oy, oz, p.mx, p.my, p.mz, (float)cval, zb, zn, resolve_stream(stream));
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    // r3: anisotropic voxels -- the same kernel with fewer taps along z than in the plane (one launch at 8 B/voxel
    // where the streaming passes below take two at 16), for the tap pairs it is instantiated for, on volumes that fill
    // the chip; index-mapping boundary modes only
    if (g_sep3d_long != 1 && !ragged && (!any_const || (float)cval == 0.0f) && w[1] == w[2] && w[0] != w[1] && nx >= 128 && ny >= 16 &&
        nvox_out >= ((int64_t)1 << 22) && mi::long_aniso_pair(w[1], w[0])) {
        const int oz = w[0] / 2 + (weights[0] ? origin[0] : 0), oy = w[1] / 2 + (weights[1] ? origin[1] : 0);
        rc = run_sep3d_long((const float *)in->data, (float *)out->data, (int)nz, (int)ny, (int)nx, w[1], w[0], wbuf[2], wbuf[1],
                            wbuf[0], oy, oz, p.mx, p.my, p.mz, (float)cval, zb, zn, resolve_stream(stream));
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    // 2-D images (one-plane volumes) and slice-wise filtering of volumes (no z taps), same tap count on y and x: the
    // streaming pass along y with the x pass fused is ONE launch at 8 B/pixel for every odd tap count up to 17 (r2:
    // 3..9 taps used to take the tiled volume kernel below, whose one-plane "chunks" have no pipeline: 8192^2
    // uniform 5 ran at 37 % of the roofline, now 60 %; 160 x 384 x 384 with (1, 9, 9): 88 -> 42 us)
    bool image2d = false;
    if (g_sep3d_image2d && w[0] == 1 && w[1] == w[2] && w[1] >= 3 && w[1] <= kMaxTaps && whole &&
        ny * nx * 4 < ((int64_t)1 << 31)) {
        const int nb = (w[2] / 2 + 3) / 4;
        const int64_t tail = nx & 255;
        image2d = !(nx < 4 * nb + 4 || (tail != 0 && tail < 4 * nb + 4));
    }
    if (image2d || w[0] > kMaxTaps || w[1] > kMaxTaps || w[2] > kMaxTaps) {
        // long kernels: streaming passes (stream3d.hip), x fused into the z pass when the tap counts agree
        if (!whole) UNSUP("plane ranges are not available for kernels longer than 9 taps");
        if (nz * ny * nx * 4 >= ((int64_t)1 << 31)) UNSUP("streaming passes need a volume < 2 GiB");
        {
            // the x pass takes whole 4-float blocks from beyond the tile edge: the last (partial) tile must be
            // followed by either the array edge or enough columns, and the array must hold the mirrored blocks
            const int nb = (w[2] / 2 + 3) / 4;
            const int64_t tail = nx & 255;
            if (w[2] > 1 && (nx < 4 * nb + 4 || (tail != 0 && tail < 4 * nb + 4))) UNSUP("x extent unsuitable for the streaming x pass");
        }
        if (t_dry_run) return MI_OK;
        hipStream_t s = resolve_stream(stream);
        const int oz = w[0] / 2 + (weights[0] ? origin[0] : 0), oy = w[1] / 2 + (weights[1] ? origin[1] : 0);
        struct Pass { int axis, wa, oa, ma, wx; };
        Pass passes[3];
        int np = 0;
        const bool fuse_xz = w[2] > 1 && w[2] == w[0] && w[2] <= g_stream_fused_max;   // longer x kernels: separate x pass (registers)
        // x fused into the y pass: images / slice-wise filters (one launch), and volumes whose in-plane kernels agree
        // while the z kernel differs (anisotropic voxels: z pass + fused y/x pass, two launches instead of three)
        const bool fuse_xy = !fuse_xz && w[2] > 1 && w[2] == w[1] && w[2] <= g_stream_fused_max;
        if (w[2] > 1 && !fuse_xz && !fuse_xy) passes[np++] = {1, 1, 0, p.my, w[2]};            // x only (streams over y)
        if (w[0] > 1) passes[np++] = {0, w[0], oz, p.mz, fuse_xz ? w[2] : 1};
        if (w[1] > 1) passes[np++] = {1, w[1], oy, p.my, fuse_xy ? w[2] : 1};
        const size_t bytes = (size_t)(nz * ny * nx) * sizeof(float);
        void *tmp[2] = {nullptr, nullptr};
        for (int t = 0; t < np - 1 && t < 2; t++)
            if ((rc = pool_alloc(&tmp[t], bytes, s))) { if (tmp[0]) pool_free(tmp[0]); return rc; }
        const float *src = (const float *)in->data;
        const float one = 1.0f;
        for (int i = 0; i < np && rc == MI_OK; i++) {
            float *dst = i == np - 1 ? (float *)out->data : (float *)tmp[i & 1];
            const Pass &q = passes[i];
            const float *wav = q.wa > 1 ? wbuf[q.axis == 0 ? 0 : 1] : &one;
            rc = run_stream_pass(src, dst, (int)nz, (int)ny, (int)nx, q.axis, wav, q.wa, q.oa, q.ma,
                                 q.wx > 1 ? wbuf[2] : nullptr, q.wx, p.mx, (float)cval, s);
            src = dst;
        }
        for (int t = 0; t < 2; t++) if (tmp[t]) pool_free(tmp[t]);   // reuse is stream ordered
        return rc;
    }
    memcpy(p.wz, wbuf[0], sizeof(float) * w[0]);
    memcpy(p.wyv, wbuf[1], sizeof(float) * w[1]);
    memcpy(p.wx, wbuf[2], sizeof(float) * w[2]);
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.wy = w[1];
    p.oz = w[0] / 2 + (weights[0] ? origin[0] : 0);
    p.oy = w[1] / 2 + (weights[1] ? origin[1] : 0);
    p.cval = (float)cval;
    p.dbg = g_sep3d_dbg;
    p.zrev = g_sep3d_zrev;

    const int cfg = g_sep3d_cfg;
    const bool cubic = w[0] == w[1] && w[1] == w[2] && w[0] >= 3 && p.oz == w[0] / 2 && p.oy == w[1] / 2;
    const bool lean = cubic && (w[0] <= 7 || (ragged && w[0] == 9)) && g_sep3d_kernel != 1 && ny * nx * 4 < ((int64_t)1 << 31);    // 9 taps: sep3d_long.hip (ragged rows: here)
    int cfg_use = cfg, rows = 0, nzc = 1;
    if (lean && cfg == 0) {
        // candidates: the big tile and (3 / 5 taps) a small one
        const int big = lean_rows(w[0], 0), small = lean_rows(w[0], 5);
        const int cand_rows[2] = {big, small}, cand_cfg[2] = {0, 5};
        choose_plan(w[0], cand_rows, cand_cfg, (w[0] <= 5 && small != big) ? 2 : 1, w[1], nzr, ny, nx, &cfg_use, &rows, &nzc);
    } else {
        rows = lean ? lean_rows(w[0], cfg) : ws_rows(w[0]);
        const int cand_rows[1] = {rows}, cand_cfg[1] = {cfg};
        choose_plan(w[0], cand_rows, cand_cfg, 1, w[1], nzr, ny, nx, &cfg_use, &rows, &nzc);
    }
    p.ty = rows - (w[1] - 1);
    if (p.ty < 1) UNSUP("y kernel too long for the tile");
    p.nxt = (int)((nx + 255) / 256);
    // equal tiles: 300 columns are 152 + 148, not 256 + 44 (a narrow last tile keeps its CU busy for as many plane
    // steps as a full one: 300^3 ran at 4.5 TB/s where 512 x 512 x 256 reaches 6.5)
    p.tw = (int)((((nx + p.nxt - 1) / p.nxt) + 3) & ~(int64_t)3);
    p.nyt = (int)((ny + p.ty - 1) / p.ty);
    if (g_sep3d_zchunks > 0) nzc = g_sep3d_zchunks;
    if ((nzr + nzc - 1) / nzc > kMaxChunk) nzc = (int)((nzr + kMaxChunk - 1) / kMaxChunk);
    p.zc = (int)((nzr + nzc - 1) / nzc);
    p.zb0 = (int)zb[0]; p.zn0 = (int)zn[0]; p.zb1 = (int)zb[1]; p.zn1 = (int)zn[1];
    p.nzc0 = (int)((zn[0] + p.zc - 1) / p.zc);
    p.nzc = p.nzc0 + (int)((zn[1] + p.zc - 1) / p.zc);

    if (t_dry_run) return MI_OK;
    hipStream_t s = resolve_stream(stream);
    const float *ip = (const float *)in->data;
    float *op = (float *)out->data;
    if (lean) return launch_lean(w[0], cfg_use, ip, op, p, any_const, s);
#define CASE_Z(WXV)                                            \
    switch (w[0]) {                                            \
    case 1: return launch_ws<WXV, 1>(ip, op, p, s);            \
    case 3: return launch_ws<WXV, 3>(ip, op, p, s);            \
    case 5: return launch_ws<WXV, 5>(ip, op, p, s);            \
    case 7: return launch_ws<WXV, 7>(ip, op, p, s);            \
    default: return launch_ws<WXV, 9>(ip, op, p, s);           \
    }
    switch (w[2]) {
    case 1: CASE_Z(1)
    case 3: CASE_Z(3)
    case 5: CASE_Z(5)
    case 7: CASE_Z(7)
    default: CASE_Z(9)
    }
#undef CASE_Z
#undef UNSUP
}

// Flat minimum / maximum of a cubic window (size w = 3 / 5 / 7 / 9, origins 0) through the lean kernel's ragged build:
// what mi_minmax3d_f32 (stream3d.hip) calls for rows that are not a multiple of four floats.  MI_ERR_UNSUPPORTED with
// nothing queued outside that envelope.
namespace mi {
int run_sep3d_lean_minmax(const mi_array *in, const mi_array *out, int w, const int mode[3], double cval, bool is_max, hipStream_t s)
{
#define UNSUP(msg) do { set_error("separable3d min/max: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (!g_sep3d_ragged || g_sep3d_kernel == 1 || g_sep3d_cfg != 0) UNSUP("ragged build switched off");
    // 9 samples per axis: 163 us here against 142 us through extended rows + the LDS-DMA kernel on 300 x 300 x 301
    // (profiles/r6_ragged_minmax.txt); 3 / 5 / 7 win everywhere (181 x 217 x 181: 58 -> 21 / 28 / 33 us)
    if (w < 3 || w > 7 || !(w & 1)) UNSUP("cubic sizes 3 / 5 / 7");
    if (nx < 16 || ny * nx * 4 >= ((int64_t)1 << 31) || nz * ny * nx >= ((int64_t)1 << 40)) UNSUP("row / plane extent");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    Sep3dParams p;
    memset(&p, 0, sizeof(p));
    p.mz = filter_mode(mode[0]); p.my = filter_mode(mode[1]); p.mx = filter_mode(mode[2]);
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.wy = w;
    p.oz = w / 2; p.oy = w / 2;
    p.cval = (float)cval;
    p.zrev = g_sep3d_zrev;
    int cfg_use = 0, rows = 0, nzc = 1;
    const int big = lean_rows(w, 0), small = lean_rows(w, 5);
    const int cand_rows[2] = {big, small}, cand_cfg[2] = {0, 5};
    choose_plan(w, cand_rows, cand_cfg, (w <= 5 && small != big) ? 2 : 1, w, nz, ny, nx, &cfg_use, &rows, &nzc);
    p.ty = rows - (w - 1);
    if (p.ty < 1) UNSUP("window too long for the tile");
    p.nxt = (int)((nx + 255) / 256);
    p.tw = (int)((((nx + p.nxt - 1) / p.nxt) + 3) & ~(int64_t)3);
    p.nyt = (int)((ny + p.ty - 1) / p.ty);
    if ((nz + nzc - 1) / nzc > kMaxChunk) nzc = (int)((nz + kMaxChunk - 1) / kMaxChunk);
    p.zc = (int)((nz + nzc - 1) / nzc);
    p.zb0 = 0; p.zn0 = (int)nz; p.zb1 = 0; p.zn1 = 0;
    p.nzc0 = (int)((nz + p.zc - 1) / p.zc);
    p.nzc = p.nzc0;
    return launch_lean(w, cfg_use, (const float *)in->data, (float *)out->data, p, true, s, is_max ? 2 : 1);
#undef UNSUP
}
}  // namespace mi

extern "C" int mi_separable3d_f32(const mi_array *in, const mi_array *out, const double *const weights[3],
                                  const int wlen[3], const int origin[3], const int mode[3], double cval,
                                  int is_box, mi_stream stream)
{
    (void)is_box;
    return separable3d_impl(in, out, weights, wlen, origin, mode, cval, nullptr, 0, stream);
}

extern "C" int mi_separable3d_f32_supports(const mi_array *in, const mi_array *out, const double *const weights[3],
                                           const int wlen[3], const int origin[3], const int mode[3], double cval,
                                           int plane_ranges)
{
    mi::DryRun dry;
    const int64_t some[2] = {0, in && in->ndim == 3 ? in->shape[0] : 0};
    return separable3d_impl(in, out, weights, wlen, origin, mode, cval, plane_ranges ? some : nullptr, plane_ranges ? 1 : 0,
                            nullptr);
}

extern "C" int mi_separable3d_f32_planes(const mi_array *in, const mi_array *out, const double *const weights[3],
                                         const int wlen[3], const int origin[3], const int mode[3], double cval,
                                         const int64_t *planes, int nranges, mi_stream stream)
{
    MI_REQUIRE(planes, MI_ERR_INVALID_ARG, "planes is NULL");
    return separable3d_impl(in, out, weights, wlen, origin, mode, cval, planes, nranges, stream);
}
