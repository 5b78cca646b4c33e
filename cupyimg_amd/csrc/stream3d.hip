// stream3d.hip -- barrier-free streaming separable passes, float32 3-D
// (design notes in stream3d.hpp).  Entry point for the host: run_stream_passes()
// called from mi_separable3d_f32 (separable3d.hip).
//
// Reference path replaced: gaussian_filter / uniform_filter with long kernels,
// cupyimg/scipy/ndimage/filters.py:602-665,725-792 (three K1 launches, fp64
// taps from global memory, zero-fill + copy-back per in-place pass).
#include "sep_common.hpp"
#include "stream3d.hpp"

namespace mi {

__device__ __forceinline__ float4 dpp4_from_left(const float4 keep, const float4 v)
{
    return make_float4(dpp_from_left(keep.x, v.x), dpp_from_left(keep.y, v.y), dpp_from_left(keep.z, v.z),
                       dpp_from_left(keep.w, v.w));
}
__device__ __forceinline__ float4 dpp4_from_right(const float4 keep, const float4 v)
{
    return make_float4(dpp_from_right(keep.x, v.x), dpp_from_right(keep.y, v.y), dpp_from_right(keep.z, v.z),
                       dpp_from_right(keep.w, v.w));
}
__device__ __forceinline__ float4 sel4(bool c, const float4 a, const float4 b) { return c ? a : b; }

// x pass, odd WX <= 33 (reach <= 16 = up to four lane hops).  eL[j] / eR[j]: the j-th
// 4-float block outside the tile, valid in lane 0 / lane `last` respectively.
// OP: what a pass computes -- weighted sum, or running minimum / maximum (the
// comparisons of the generic min/max kernel: first sample taken as is, then
// `x < best` / `x > best` in ascending tap order).

template <int OP>
__device__ __forceinline__ float pick_mm(float x, float best) { return (OP == SP_MAX ? x > best : x < best) ? x : best; }
template <int OP>
__device__ __forceinline__ F4 f4_mm(const F4 x, const F4 best)
{
    F4 r;
    r.lo = (f32x2){pick_mm<OP>(x.lo.x, best.lo.x), pick_mm<OP>(x.lo.y, best.lo.y)};
    r.hi = (f32x2){pick_mm<OP>(x.hi.x, best.hi.x), pick_mm<OP>(x.hi.y, best.hi.y)};
    return r;
}

template <int WX, int OP = SP_CORR>
__device__ __forceinline__ F4 xpass_hops(const float4 v, const float4 (&eL)[4], const float4 (&eR)[4], int lane,
                                         int last, kfloats tab0, kfloats tab1)
{
    if constexpr (WX == 1) {
        return f4_from(v);
    } else {
        constexpr int RX = WX / 2;
        constexpr int NB = (RX + 3) / 4;            // blocks per side
        // window of 4 * (2 NB + 1) floats: [L_NB .. L_1 | v | R_1 .. R_NB]
        float4 blk[2 * NB + 1];
        blk[NB] = v;
        float4 l = v, r = v;
#pragma unroll
        for (int j = 1; j <= NB; j++) {
            l = dpp4_from_left(eL[j - 1], l);       // lane 0 keeps the edge block, lane 1 then inherits it
            r = sel4(lane == last, eR[j - 1], dpp4_from_right(eR[j - 1], r));
            blk[NB - j] = l;
            blk[NB + j] = r;
        }
        constexpr int NP = 2 * (2 * NB + 1);
        f32x2 A[NP];
#pragma unroll
        for (int b = 0; b < 2 * NB + 1; b++) {
            A[2 * b] = (f32x2){blk[b].x, blk[b].y};
            A[2 * b + 1] = (f32x2){blk[b].z, blk[b].w};
        }
        constexpr int BASE = 4 * NB - RX;            // window[BASE + c + k] = in[x + c - RX + k]
        if constexpr (OP == SP_CORR) {
            return xdot_tab<WX, NP, BASE>(A, tab0, tab1);
        } else {
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                auto win = [&](int t) { return (t & 1) ? A[t >> 1].y : A[t >> 1].x; };
                float best = win(BASE + c);
#pragma unroll
                for (int k = 1; k < WX; k++) best = pick_mm<OP>(win(BASE + c + k), best);
                o[c] = best;
            }
            F4 r;
            r.lo = (f32x2){o[0], o[1]};
            r.hi = (f32x2){o[2], o[3]};
            return r;
        }
    }
}


template <int WX, int WA, int DEPTH, int OP = SP_CORR>
__global__ void __launch_bounds__(256)
stream_pass_kernel(const float *__restrict__ in, float *__restrict__ out, const StreamParams p)
{
    constexpr int RX = WX / 2;
    constexpr int NB = WX > 1 ? (RX + 3) / 4 : 0;
    constexpr int RINGN = WA - 1;
    constexpr int U = RINGN > 0 ? lcm_(RINGN, DEPTH) : DEPTH;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int nother = p.axis == 0 ? ny : nz;
    const int nA = p.axis == 0 ? nz : ny;
    const int nlines = nother * p.nxt;
    const int wid = p.wid_base + xcd_block((int)blockIdx.x, (int)gridDim.x, p.swz) * p.wpb + wave;
    if (wid >= nlines * p.nchunks) return;
    const int c = wid / nlines;
    const int line = wid - c * nlines;
    const int oth = line / p.nxt, xt = line - oth * p.nxt;
    const int x0 = xt * 256;
    const int nlanes = min(64, (nx - x0) >> 2);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;               // elements
    const unsigned strideA = p.axis == 0 ? plane : (unsigned)nx;      // elements between streamed samples
    const unsigned rowbase = p.axis == 0 ? (unsigned)oth * nx : (unsigned)oth * plane;   // elements
    const unsigned total_bytes = plane * (unsigned)nz * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 4 * lane)) * 4u : kOOB;

    // tile-edge blocks for the x pass (lane 0: left, lane `last`: right)
    unsigned evoff[4] = {kOOB, kOOB, kOOB, kOOB};
    int ekind[4] = {EDGE_FWD, EDGE_FWD, EDGE_FWD, EDGE_FWD};
    const int side = lane == 0 ? 0 : 1;
    if constexpr (WX > 1) {
        const bool is_edge_lane = lane == 0 || lane == last;
#pragma unroll
        for (int j = 1; j <= NB; j++) {
            int st, kd;
            edge_block(side, j, x0, x0 + 4 * nlanes, nx, p.mx, &st, &kd);
            ekind[j - 1] = kd;
            if (is_edge_lane && kd != EDGE_CONST) evoff[j - 1] = (rowbase + (unsigned)st) * 4u;
        }
    }

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, nA);
    const int nsteps = a1 - a0 + WA - 1;
    const int ai0 = a0 - p.oa;

    struct Slot { float4 v; float4 e[NB > 0 ? NB : 1]; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)nA) ai = bmap<int>(ai, nA, p.ma);    // only near the ends of the axis
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * strideA * 4u;
        s.v = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0));
#pragma unroll
        for (int j = 0; j < NB; j++)
            s.e[j] = as_f4(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : evoff[j], soff, 0));
    };

    F4 ring[RINGN > 0 ? RINGN : 1];
#pragma unroll
    for (int k = 0; k < (RINGN > 0 ? RINGN : 1); k++) ring[k] = f4_splat(0.f);

#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    const float4 cv4 = make_float4(p.cval, p.cval, p.cval, p.cval);
    // weights stay in the kernel-argument segment and are s_loaded per step (see launder())
    constexpr int kArgBase = 2 * sizeof(void *);
    kfloats wav = kernarg_floats(kArgBase + offsetof(StreamParams, wav));
    kfloats xt0 = kernarg_floats(kArgBase + offsetof(StreamParams, xpair));
    kfloats xt1 = xt0 + 2 * (kStreamMaxTaps / 2 + 2);
    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                // few weights (no fused x pass, <= 17 taps) stay in SGPRs for the whole loop; more are re-loaded
                // per step (see launder())
                if constexpr (OP == SP_CORR && (WX > 1 || WA > 17)) {
                    launder(wav);
                    if constexpr (WX > 1) { launder(xt0); launder(xt1); }
                }
                Slot &s = S[J % DEPTH];
                float4 v = s.cst ? cv4 : s.v;
                float4 eL[4] = {cv4, cv4, cv4, cv4}, eR[4] = {cv4, cv4, cv4, cv4};
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const float4 t = s.cst ? cv4 : apply_kind(s.e[j], ekind[j], side, p.cval);
                    eL[j] = t;
                    eR[j] = t;
                }
                const F4 xf = xpass_hops<WX, OP>(v, eL, eR, lane, last, xt0, xt1);
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                if (i >= WA - 1) {
                    F4 a;
                    if constexpr (OP != SP_CORR) {
                        if constexpr (WA == 1) {
                            a = xf;
                        } else {
                            a = ring[J % RINGN];                   // oldest sample = tap 0
#pragma unroll
                            for (int k = 1; k < RINGN; k++) a = f4_mm<OP>(ring[(J + k) % RINGN], a);
                            a = f4_mm<OP>(xf, a);
                        }
                    } else if constexpr (WA == 1) {
                        a = f4_scale(wav[0], xf);
                    } else {
                        a = f4_scale(wav[0], ring[J % RINGN]);
#pragma unroll
                        for (int k = 1; k < RINGN; k++) a = f4_fma(wav[k], ring[(J + k) % RINGN], a);
                        a = f4_fma(wav[WA - 1], xf, a);
                    }
                    const unsigned so = (unsigned)(a0 + i - (WA - 1)) * strideA * 4u;
                    buffer_store_b128_soff(f4_to_u32(a), rout, voff, so);
                }
                if constexpr (RINGN > 0) ring[J % RINGN] = xf;
            }
        });
    }
}

// ---------------------------------------------------------------------------
// Flat footprints whose rows are centred runs (disk, diamond, cross, square; see runs_minmax_u8_kernel in
// minmax3d_u8.hip) on float32 images / slice-wise on volumes, one streaming launch: the previous WA - 1 raw rows (a
// float4 and the edge block per lane) stay in registers; rows that share a half width are combined first (min / max
// commute), then one x window per distinct half width (xpass_hops with compare-select).
// ---------------------------------------------------------------------------
struct RunsF32Params {
    int nx, ny, nz;
    int mx, my;
    float cval;
    int chunk, nchunks, nxt;
    int swz;
    int hw[9];           // half width of the run of footprint row r, -1 = empty row
};

template <int WA, int OP>
__global__ void __launch_bounds__(256)
runs_minmax_f32_kernel(const float *__restrict__ in, float *__restrict__ out, const RunsF32Params p)
{
    constexpr int DEPTH = WA <= 5 ? 4 : 2;       // 8192^2 disk(1): 142 -> 114 us with four loads in flight; 7 rows: slower
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
    const int x0 = xt * 256;
    const int nlanes = min(64, (nx - x0) >> 2);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned rowbase = (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 4u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 4 * lane)) * 4u : kOOB;
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_block(side, 1, x0, x0 + 4 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? (rowbase + (unsigned)est) * 4u : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, ny);
    const int nsteps = a1 - a0 + WA - 1;
    const int ai0 = a0 - WA / 2;

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

    struct Row { float4 v, e; };
    Row ring[RINGN > 0 ? RINGN : 1];
    const float4 cv4 = make_float4(p.cval, p.cval, p.cval, p.cval);
    auto mm4 = [](const float4 a, const float4 b) {
        return make_float4(pick_mm<OP>(b.x, a.x), pick_mm<OP>(b.y, a.y), pick_mm<OP>(b.z, a.z), pick_mm<OP>(b.w, a.w));
    };
    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J % DEPTH];
                Row cur;
                cur.v = s.cst ? cv4 : s.v;
                cur.e = s.cst ? cv4 : apply_kind(s.e, ekind, side, p.cval);
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                if (i >= WA - 1) {
                    float4 a = cv4;
                    bool have = false;
                    static_for<5>([&](auto HH) {
                        constexpr int h = decltype(HH)::value;
                        Row g;
                        bool any = false;
                        static_for<WA>([&](auto KK) {
                            constexpr int k = decltype(KK)::value;
                            if (p.hw[k] == h) {
                                const Row &r = k == WA - 1 ? cur : ring[(J + k) % (RINGN > 0 ? RINGN : 1)];
                                if (any) { g.v = mm4(g.v, r.v); g.e = mm4(g.e, r.e); }
                                else g = r;
                                any = true;
                            }
                        });
                        if (any) {
                            float4 eL[4] = {g.e, g.e, g.e, g.e}, eR[4] = {g.e, g.e, g.e, g.e};
                            const F4 t = xpass_hops<2 * h + 1, OP>(g.v, eL, eR, lane, last, nullptr, nullptr);
                            const float4 tf = f4_to_float4(t);
                            a = have ? mm4(a, tf) : tf;
                            have = true;
                        }
                    });
                    const unsigned so = (unsigned)(a0 + i - (WA - 1)) * (unsigned)nx * 4u;
                    u32x4 u;
                    u.x = __float_as_uint(a.x); u.y = __float_as_uint(a.y); u.z = __float_as_uint(a.z); u.w = __float_as_uint(a.w);
                    buffer_store_b128_soff(u, rout, voff, so);
                }
                if constexpr (RINGN > 0) ring[J % RINGN] = cur;
            }
        });
    }
}

template <int WA>
static int launch_runs_f32(const float *in, float *out, RunsF32Params &p, bool is_max, hipStream_t s)
{
    const int nlines = p.nz * p.nxt;
    int nch = 1;
    {
        double best = 1e300;
        for (int c = 1; c <= p.ny && c <= 1024; c++) {
            const int chunk = (p.ny + c - 1) / c;
            if (c > 1 && chunk < 16) break;
            const int real = (p.ny + chunk - 1) / chunk;
            const double rounds = std::max(1.0, (double)nlines * real / 4096.0);
            const double cost = rounds * (chunk + (WA - 1) + 4.0);
            if (cost < best * 0.999) { best = cost; nch = real; }
        }
    }
    p.chunk = (p.ny + nch - 1) / nch;
    p.nchunks = (p.ny + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz * 4);
    if (is_max) hipLaunchKernelGGL((runs_minmax_f32_kernel<WA, SP_MAX>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    else hipLaunchKernelGGL((runs_minmax_f32_kernel<WA, SP_MIN>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// Round 1 issued a pass in launches of at most 1000 waves because larger launches produced wrong samples; the
// cause was the store-data hazard described at buffer_store_b128_soff() (sep_common.hpp), not the launch size.
// The slice hook stays for tests (0 = one launch, the default).
mi::Knob g_xcd_swizzle{2};               // test hook: 0 = plain workgroup order, 1 = always XCD-contiguous, 2 = auto
extern "C" int mi_debug_set_xcd_swizzle(int k) { g_xcd_swizzle = k; return MI_OK; }
static mi::Knob g_stream_wpb{4};             // test hook: waves per workgroup (1, 2 or 4)
extern "C" int mi_debug_set_stream_wpb(int n) { g_stream_wpb = n; return MI_OK; }
static mi::Knob g_stream_slice{0};           // test hook: waves per launch of a pass (0 = one launch)
extern "C" int mi_debug_set_stream_slice(int n) { g_stream_slice = n; return MI_OK; }
static mi::Knob g_stream_min_chunk{16};      // test hook: shortest chunk the planner may choose
extern "C" int mi_debug_set_stream_min_chunk(int n) { g_stream_min_chunk = n; return MI_OK; }

template <int WX, int WA, int OP = SP_CORR>
static int launch_stream(const float *in, float *out, StreamParams &p, hipStream_t s)
{
    // in-flight loads per wave: 4 where registers allow (the fused long x pass needs them for its window)
    constexpr int DEPTH = ((WA - 1) % 4 == 0 && WA > 1 && WX <= 9) ? 4 : 2;
    const int nA = p.axis == 0 ? p.nz : p.ny;
    const int nother = p.axis == 0 ? p.ny : p.nz;
    const int nlines = nother * p.nxt;
    // chunks along the streamed axis: time ~ rounds x (chunk + ramp), with 256 CUs x 16 resident waves;
    // many lines (3-D volumes) -> few long chunks, few lines (2-D images, thin slabs) -> many short ones
    int nch = 1;
    {
        double best = 1e300;
        for (int c = 1; c <= nA && c <= 1024; c++) {
            const int chunk = (nA + c - 1) / c;
            if (c > 1 && chunk < g_stream_min_chunk) break;
            const int real = (nA + chunk - 1) / chunk;
            const double rounds = std::max(1.0, (double)nlines * real / 4096.0);
            const double cost = rounds * (chunk + (WA - 1) + 4.0);
            if (cost < best * 0.999) { best = cost; nch = real; }
        }
    }
    p.chunk = (nA + nch - 1) / nch;
    p.nchunks = (nA + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    const int slice = g_stream_slice > 0 ? (int)g_stream_slice : waves;
    const int wpb = g_stream_wpb;
    p.wpb = wpb;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz * 4);
    for (int base = 0; base < waves; base += slice) {
        p.wid_base = base;
        const int n = std::min(slice, waves - base);
        note_kernel("mi::stream_pass_kernel<%d,%d,%d,%d> grid=%d", WX, WA, DEPTH, (int)OP, (n + wpb - 1) / wpb);
        hipLaunchKernelGGL((stream_pass_kernel<WX, WA, DEPTH, OP>), dim3((n + wpb - 1) / wpb), dim3(64 * wpb), 0, s, in, out, p);
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// switch over odd tap counts 1..33 for one template slot
#define MI_ODD_CASES(X) X(1) X(3) X(5) X(7) X(9) X(11) X(13) X(15) X(17) X(19) X(21) X(23) X(25) X(27) X(29) X(31) X(33)

template <int WX>
static int launch_stream_wa(int wa, const float *in, float *out, StreamParams &p, hipStream_t s)
{
    switch (wa) {
#define X(N) case N: return launch_stream<WX, N>(in, out, p, s);
        MI_ODD_CASES(X)
#undef X
    }
    set_error("stream pass: unsupported tap count %d", wa);
    return MI_ERR_UNSUPPORTED;
}

// one streaming pass along `axis` (0 = z, 1 = y) with `wa` taps; optional
// fused x pass with `wx` taps (1 = none; only together with the same tap count
// or none, to bound the number of instantiations)
int run_stream_pass(const float *in, float *out, int nz, int ny, int nx, int axis, const float *wav, int wa, int oa,
                    int ma, const float *wxv, int wx, int mx, float cval, hipStream_t s)
{
    StreamParams p;
    memset(&p, 0, sizeof(p));
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.axis = axis;
    p.wa = wa; p.oa = oa; p.ma = ma; p.mx = mx;
    p.cval = cval;
    p.nxt = (nx + 255) / 256;
    for (int k = 0; k < wa; k++) p.wav[k] = wav[k];
    for (int k = 0; k < wx; k++) p.wxv[k] = wxv ? wxv[k] : 1.0f;
    if (wx > 1) {
        const int rx = wx / 2, nb = (rx + 3) / 4, base = 4 * nb - rx;
        for (int q = 0; q < 2; q++) {
            const int t0 = base + q, m0 = t0 / 2;
            for (int u = 0; u < kStreamMaxTaps / 2 + 2; u++)
                for (int h = 0; h < 2; h++) {
                    const int j = 2 * (m0 + u) + h - t0;
                    p.xpair[q][2 * u + h] = (j >= 0 && j < wx) ? p.wxv[j] : 0.0f;
                }
        }
    }
    if (wx == 1) return launch_stream_wa<1>(wa, in, out, p, s);
    if (wa == 1) {
        switch (wx) {
#define X(N) case N: if constexpr (N > 1) return launch_stream<N, 1>(in, out, p, s); break;
            MI_ODD_CASES(X)
#undef X
        }
    } else if (wx == wa) {
        switch (wx) {
#define X(N) case N: if constexpr (N <= kStreamFusedMax && N > 1) return launch_stream<N, N>(in, out, p, s); break;
            MI_ODD_CASES(X)
#undef X
        }
    }
    set_error("stream pass: unsupported x/axis tap combination %d/%d", wx, wa);
    return MI_ERR_UNSUPPORTED;
}

// one streaming min / max pass (odd sizes <= 9): along `axis` with `wa` samples, x window `wx` fused when
// wx == wa or one of them is 1
template <int OP>
static int minmax_pass_op(const float *in, float *out, StreamParams &p, int wa, int wx, hipStream_t s)
{
#define MM(WXV, WAV) return launch_stream<WXV, WAV, OP>(in, out, p, s)
    if (wx == 1) {
        switch (wa) { case 3: MM(1, 3); case 5: MM(1, 5); case 7: MM(1, 7); case 9: MM(1, 9); }
    } else if (wa == 1) {
        switch (wx) { case 3: MM(3, 1); case 5: MM(5, 1); case 7: MM(7, 1); case 9: MM(9, 1); }
    } else if (wa == wx) {
        switch (wx) { case 3: MM(3, 3); case 5: MM(5, 5); case 7: MM(7, 7); case 9: MM(9, 9); }
    }
#undef MM
    set_error("stream min/max pass: unsupported sizes %d/%d", wx, wa);
    return MI_ERR_UNSUPPORTED;
}

int run_stream_minmax_pass(const float *in, float *out, int nz, int ny, int nx, int axis, int wa, int oa, int ma, int wx,
                           int mx, float cval, bool is_max, hipStream_t s)
{
    StreamParams p;
    memset(&p, 0, sizeof(p));
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.axis = axis;
    p.wa = wa; p.oa = oa; p.ma = ma; p.mx = mx;
    p.cval = cval;
    p.nxt = (nx + 255) / 256;
    return is_max ? minmax_pass_op<SP_MAX>(in, out, p, wa, wx, s) : minmax_pass_op<SP_MIN>(in, out, p, wa, wx, s);
}

}  // namespace mi

namespace mi {
int run_minmax3d_f32_fused(const float *in, float *out, int nz, int ny, int nx, int w, int oy, int oz, int mx, int my, int mz,
                           bool is_max, hipStream_t s);     // minmax3d_f32.hip
int run_sep3d_lean_minmax(const mi_array *in, const mi_array *out, int w, const int mode[3], double cval, bool is_max,
                          hipStream_t s);                   // separable3d.hip (r6: rows of any length)
}

using namespace mi;

static mi::Knob g_minmax_f32_fused{1};      // test hook: 0 = always the two streaming launches
extern "C" int mi_debug_set_minmax_f32_fused(int k) { g_minmax_f32_fused = k; return MI_OK; }

// test hook: one streaming pass with arbitrary float weights (not part of the C-ABI)
extern "C" int mi_debug_stream_pass(const float *in, float *out, int nz, int ny, int nx, int axis, const float *wav,
                                    int wa, int oa, int ma, const float *wxv, int wx, int mx, float cval, mi_stream stream)
{
    return run_stream_pass(in, out, nz, ny, nx, axis, wav, wa, oa, ma, wxv, wx, mx, cval, resolve_stream(stream));
}

/* Separable flat min / max filter on a float32 volume as streaming passes
 * (declared in include/mi355img.h). */
extern "C" int mi_minmax3d_f32(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                               const int mode[3], double cval, int is_max, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(size && origin && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("minmax3d_f32: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3 || in->dtype != MI_F32 || out->dtype != MI_F32) UNSUP("needs 3-D float32 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if ((nx & 3) && nz >= 1 && ny >= 1) {
        // r6: rows that are not a multiple of four floats -- cubic sizes without origins in ONE launch on the rows as they
        // are (the lean kernel's ragged build with min / max for its three passes); anything else: the caller extends the rows
        if (size[0] == size[1] && size[1] == size[2] && !origin[0] && !origin[1] && !origin[2]) {
            rc = run_sep3d_lean_minmax(in, out, size[0], mode, cval, is_max != 0, resolve_stream(stream));
            if (rc != MI_ERR_UNSUPPORTED || size[0] != 9 || !g_minmax_f32_fused || nx < 16 || nz * ny * nx * 4 >= ((int64_t)1 << 31)) return rc;
            // size 9: the LDS-DMA kernel's ragged build (index-mapping modes; it refuses the rest)
            return run_minmax3d_f32_fused((const float *)in->data, (float *)out->data, (int)nz, (int)ny, (int)nx, 9, 4, 4,
                                          filter_mode(mode[2]), filter_mode(mode[1]), filter_mode(mode[0]), is_max != 0, resolve_stream(stream));
        }
        UNSUP("rows that are not a multiple of 4 floats: cubic sizes 3 / 5 / 7 / 9 without origins only");
    }
    if (nz < 1 || ny < 1 || nx < 8 || (nx & 3)) UNSUP("x extent must be a multiple of 4, >= 8");
    if (nz * ny * nx * 4 >= ((int64_t)1 << 31)) UNSUP("needs a volume < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    int w[3], off[3];
    for (int a = 0; a < 3; a++) {
        w[a] = size[a];
        if (w[a] < 1 || w[a] > 9 || !(w[a] & 1)) UNSUP("sizes must be odd and <= 9");
        off[a] = w[a] / 2 + origin[a];
        if (off[a] < 0 || off[a] >= w[a]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    if (origin[2] != 0) UNSUP("x origin must be 0");
    {
        const int nb = (w[2] / 2 + 3) / 4;
        const int64_t tail = nx & 255;
        if (w[2] > 1 && (nx < 4 * nb + 4 || (tail != 0 && tail < 4 * nb + 4))) UNSUP("x extent unsuitable for the streaming x pass");
    }
    const int mz = filter_mode(mode[0]), my = filter_mode(mode[1]), mx = filter_mode(mode[2]);
    hipStream_t s = resolve_stream(stream);
    if (g_minmax_f32_fused && w[0] == w[1] && w[1] == w[2] && w[0] >= 3) {
        // cubic sizes: ONE launch (minmax3d_f32.hip); it refuses what it does not cover (constant mode, tiny rows)
        rc = run_minmax3d_f32_fused((const float *)in->data, (float *)out->data, (int)nz, (int)ny, (int)nx, w[0], off[1], off[0],
                                    mx, my, mz, is_max != 0, s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    struct Pass { int axis, wa, oa, ma, wx; };
    Pass passes[3];
    int np = 0;
    const bool fuse_xz = w[2] > 1 && w[2] == w[0];
    const bool fuse_xy = !fuse_xz && w[0] == 1 && w[2] > 1 && w[2] == w[1];
    if (w[2] > 1 && !fuse_xz && !fuse_xy) passes[np++] = {1, 1, 0, my, w[2]};      // x only (streams over y)
    if (w[0] > 1) passes[np++] = {0, w[0], off[0], mz, fuse_xz ? w[2] : 1};
    if (w[1] > 1) passes[np++] = {1, w[1], off[1], my, fuse_xy ? w[2] : 1};
    if (np == 0) UNSUP("nothing to filter");
    const size_t bytes = (size_t)(nz * ny * nx) * sizeof(float);
    void *tmp[2] = {nullptr, nullptr};
    for (int t = 0; t < np - 1 && t < 2; t++)
        if ((rc = pool_alloc(&tmp[t], bytes, s))) { if (tmp[0]) pool_free(tmp[0]); return rc; }
    const float *src = (const float *)in->data;
    for (int i = 0; i < np && rc == MI_OK; i++) {
        float *dst = i == np - 1 ? (float *)out->data : (float *)tmp[i & 1];
        const Pass &q = passes[i];
        rc = run_stream_minmax_pass(src, dst, (int)nz, (int)ny, (int)nx, q.axis, q.wa, q.oa, q.ma, q.wx, mx, (float)cval,
                                    is_max != 0, s);
        src = dst;
    }
    for (int t = 0; t < 2; t++) if (tmp[t]) pool_free(tmp[t]);   // reuse is stream ordered
    return rc;
#undef UNSUP
}

/* Flat footprint given as centred runs per row, float32 images (declared in include/mi355img.h). */
extern "C" int mi_minmax_runs_f32(const mi_array *in, const mi_array *out, int nrows, const int *half_width, const int mode[2],
                                  double cval, int is_max, mi_stream stream)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(half_width && mode, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("minmax_runs_f32: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if ((in->ndim != 2 && in->ndim != 3) || in->dtype != MI_F32 || out->dtype != MI_F32) UNSUP("needs 2-D / 3-D float32 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int nd = in->ndim;
    const int64_t nz = nd == 3 ? in->shape[0] : 1, ny = in->shape[nd - 2], nx = in->shape[nd - 1];
    if (nz < 1 || ny < 1 || nx < 8 || (nx & 3)) UNSUP("x extent must be a multiple of 4, >= 8");
    { const int64_t tail = nx & 255; if (tail != 0 && tail < 8) UNSUP("x extent unsuitable for the streaming x window"); }
    if (nz * ny * nx * 4 >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    if (nrows < 1 || nrows > 9 || !(nrows & 1)) UNSUP("1, 3, 5, 7 or 9 footprint rows");
    RunsF32Params p;
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
    p.cval = (float)cval;
    p.nxt = (int)((nx + 255) / 256);
    hipStream_t s = resolve_stream(stream);
    const float *ip = (const float *)in->data;
    float *op = (float *)out->data;
    switch (nrows) {
    case 1: return launch_runs_f32<1>(ip, op, p, is_max != 0, s);
    case 3: return launch_runs_f32<3>(ip, op, p, is_max != 0, s);
    case 5: return launch_runs_f32<5>(ip, op, p, is_max != 0, s);
    case 7: return launch_runs_f32<7>(ip, op, p, is_max != 0, s);
    default: return launch_runs_f32<9>(ip, op, p, is_max != 0, s);
    }
#undef UNSUP
}
