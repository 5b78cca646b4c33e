// stream_f64.hip -- barrier-free streaming separable passes for float64 images and volumes.
//
// skimage pipelines run in float64 (img_as_float of uint8 data), and the reference serves every dtype from the same
// generated K1 kernel (cupyimg/scipy/ndimage/filters.py:213-283,602-665,725-792 -> _filters_core.py:112-156): one
// launch per axis, one thread per output, W strided loads each.  This is the float64 counterpart of stream3d.hip:
// a wave owns a 128-double (1 KiB) row segment, lane l holds the double2 at x0 + 2l, and streams along y (or z) over a
// chunk with the previous W - 1 samples of every lane in a register ring; the x pass runs in registers on every
// loaded row (lane shifts by DPP) and is fused into the streamed pass when the tap counts agree.  An image or a
// slice-wise filter is ONE launch at 16 B/pixel, a volume two (x fused into z, then y) -- against 2 / 3 launches of
// the generic kernel at 15-20 % of the roofline each.
//
// Arithmetic: float64 multiply-adds in ascending tap order (x, then the streamed axis).  SciPy sums the taps of a
// symmetric kernel in centre-out pairs and filters the axes in order 0, 1, 2; the results agree to a few ulp of
// float64 (tests: 1e-12 relative), like the float32 kernels agree to a few ulp of float32.
#include "sep_common.hpp"
#include "stream3d.hpp"

namespace mi {

struct StreamParamsD {
    int nx, ny, nz;
    int axis;            // streamed axis: 0 = z, 1 = y
    int wa, oa, ma;      // taps / offset (w/2 + origin) / mode along the streamed axis
    int mx;              // x boundary mode
    double cval;
    int chunk, nchunks, nxt;
    int swz;             // XCD-aware workgroup order (xcd_block())
    double wav[kStreamMaxTaps];
    double wxv[kStreamMaxTaps];
};

// the j-th block of 2 doubles outside the tile on `side` (0 left, 1 right): element offset to load it from, fix-up
__device__ __forceinline__ void edge_block2(int side, int j, int x0, int xe, int nx, int mode, int *start, int *kind)
{
    if (side == 0) {
        if (x0 - 2 * j >= 0) { *start = x0 - 2 * j; *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = 2 * (j - 1); *kind = EDGE_REV; break;        // ext -k = x[k-1]
        case MI_MODE_MIRROR:    *start = 2 * (j - 1) + 1; *kind = EDGE_REV; break;    // ext -k = x[k]
        case MI_MODE_NEAREST:   *start = 0; *kind = EDGE_SPLAT; break;
        case MI_MODE_GRID_WRAP: *start = nx - 2 * j; *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    } else {
        if (xe + 2 * j <= nx) { *start = xe + 2 * (j - 1); *kind = EDGE_FWD; return; }
        switch (mode) {
        case MI_MODE_REFLECT:   *start = nx - 2 * j; *kind = EDGE_REV; break;         // ext n-1+k = x[n-k]
        case MI_MODE_MIRROR:    *start = nx - 1 - 2 * j; *kind = EDGE_REV; break;     // ext n-1+k = x[n-1-k]
        case MI_MODE_NEAREST:   *start = nx - 2; *kind = EDGE_SPLAT; break;           // splat component 1
        case MI_MODE_GRID_WRAP: *start = 2 * (j - 1); *kind = EDGE_FWD; break;
        default:                *start = 0; *kind = EDGE_CONST; break;
        }
    }
}

__device__ __forceinline__ double2 apply_kind2(double2 t, int kind, int side, double cval)
{
    if (kind == EDGE_REV) return make_double2(t.y, t.x);
    if (kind == EDGE_SPLAT) { const double s = side == 0 ? t.x : t.y; return make_double2(s, s); }
    if (kind == EDGE_CONST) return make_double2(cval, cval);
    return t;
}

__device__ __forceinline__ double dppd_from_left(double keep, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), 0x138, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), 0x138, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dppd_from_right(double keep, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), 0x130, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), 0x130, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double2 as_d2(u32x4 u)
{
    return make_double2(__hiloint2double((int)u.y, (int)u.x), __hiloint2double((int)u.w, (int)u.z));
}
__device__ __forceinline__ u32x4 d2_to_u32(double2 d)
{
    u32x4 u;
    u.x = (unsigned)__double2loint(d.x); u.y = (unsigned)__double2hiint(d.x);
    u.z = (unsigned)__double2loint(d.y); u.w = (unsigned)__double2hiint(d.y);
    return u;
}

typedef const __attribute__((address_space(4))) double *kdoubles;
__device__ __forceinline__ kdoubles kernarg_doubles(int byte_offset)
{
    return (kdoubles)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + byte_offset);
}
__device__ __forceinline__ void launder(kdoubles &p) { asm volatile("" : "+s"(p)); }

template <int OP> __device__ __forceinline__ double pick_mmd(double x, double best) { return (OP == SP_MAX ? x > best : x < best) ? x : best; }

// x pass, odd WX <= 33: eL[j] / eR[j] = the (j+1)-th 2-double block outside the tile, valid in lane 0 / lane `last`
template <int WX, int OP>
__device__ __forceinline__ double2 xpass_d(const double2 v, const double2 (&eL)[8], const double2 (&eR)[8], int lane, int last,
                                           kdoubles wx)
{
    if constexpr (WX == 1) {
        return v;
    } else {
        constexpr int RX = WX / 2;
        constexpr int NB = (RX + 1) / 2;
        double win[2 * (2 * NB + 1)];                // [L_NB .. L_1 | v | R_1 .. R_NB]
        win[2 * NB] = v.x; win[2 * NB + 1] = v.y;
        double2 l = v, r = v;
#pragma unroll
        for (int j = 1; j <= NB; j++) {
            l = make_double2(dppd_from_left(eL[j - 1].x, l.x), dppd_from_left(eL[j - 1].y, l.y));
            const double2 rr = make_double2(dppd_from_right(eR[j - 1].x, r.x), dppd_from_right(eR[j - 1].y, r.y));
            r = lane == last ? eR[j - 1] : rr;
            win[2 * (NB - j)] = l.x; win[2 * (NB - j) + 1] = l.y;
            win[2 * (NB + j)] = r.x; win[2 * (NB + j) + 1] = r.y;
        }
        constexpr int BASE = 2 * NB - RX;            // win[BASE + c + k] = in[x + c - RX + k]
        double o[2];
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if constexpr (OP == SP_CORR) {
                double acc = wx[0] * win[BASE + c];
#pragma unroll
                for (int k = 1; k < WX; k++) acc = fma(wx[k], win[BASE + c + k], acc);
                o[c] = acc;
            } else {
                double best = win[BASE + c];
#pragma unroll
                for (int k = 1; k < WX; k++) best = pick_mmd<OP>(win[BASE + c + k], best);
                o[c] = best;
            }
        }
        return make_double2(o[0], o[1]);
    }
}

template <int WX, int WA, int DEPTH, int OP>
__global__ void __launch_bounds__(256)
stream_pass_f64_kernel(const double *__restrict__ in, double *__restrict__ out, const StreamParamsD p)
{
    constexpr int RX = WX / 2;
    constexpr int NB = WX > 1 ? (RX + 1) / 2 : 0;
    constexpr int RINGN = WA - 1;
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
    const int x0 = xt * 128;
    const int nlanes = min(64, (nx - x0) >> 1);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;               // elements
    const unsigned strideA = p.axis == 0 ? plane : (unsigned)nx;
    const unsigned rowbase = p.axis == 0 ? (unsigned)oth * nx : (unsigned)oth * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 2 * lane)) * 8u : kOOB;

    unsigned evoff[NB > 0 ? NB : 1];
    int ekind[NB > 0 ? NB : 1];
    const int side = lane == 0 ? 0 : 1;
    if constexpr (WX > 1) {
        const bool is_edge_lane = lane == 0 || lane == last;
#pragma unroll
        for (int j = 1; j <= NB; j++) {
            int st, kd;
            edge_block2(side, j, x0, x0 + 2 * nlanes, nx, p.mx, &st, &kd);
            ekind[j - 1] = kd;
            evoff[j - 1] = (is_edge_lane && kd != EDGE_CONST) ? (rowbase + (unsigned)st) * 8u : kOOB;
        }
    }

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, nA);
    const int nsteps = a1 - a0 + WA - 1;
    const int ai0 = a0 - p.oa;

    struct Slot { double2 v; double2 e[NB > 0 ? NB : 1]; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)nA) ai = bmap<int>(ai, nA, p.ma);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * strideA * 8u;
        s.v = as_d2(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0));
#pragma unroll
        for (int j = 0; j < NB; j++)
            s.e[j] = as_d2(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : evoff[j], soff, 0));
    };

    double2 ring[RINGN > 0 ? RINGN : 1];
#pragma unroll
    for (int k = 0; k < (RINGN > 0 ? RINGN : 1); k++) ring[k] = make_double2(0.0, 0.0);
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    const double2 cv2 = make_double2(p.cval, p.cval);
    constexpr int kArgBase = 2 * sizeof(void *);
    kdoubles wav = kernarg_doubles(kArgBase + offsetof(StreamParamsD, wav));
    kdoubles wxv = kernarg_doubles(kArgBase + offsetof(StreamParamsD, wxv));
    for (int i0 = 0; i0 < nsteps; i0 += U) {
        static_for<U>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                // weights are s_loaded per step (2 x 17 doubles = 68 SGPRs would not stay resident; see launder())
                if constexpr (OP == SP_CORR && (WX > 1 || WA > 9)) { launder(wav); launder(wxv); }
                Slot &s = S[J % DEPTH];
                const double2 v = s.cst ? cv2 : s.v;
                double2 eL[8], eR[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { eL[j] = cv2; eR[j] = cv2; }
#pragma unroll
                for (int j = 0; j < NB; j++) {
                    const double2 t = s.cst ? cv2 : apply_kind2(s.e[j], ekind[j], side, p.cval);
                    eL[j] = t;
                    eR[j] = t;
                }
                const double2 xf = xpass_d<WX, OP>(v, eL, eR, lane, last, wxv);
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                if (i >= WA - 1) {
                    double2 a;
                    if constexpr (OP != SP_CORR) {
                        if constexpr (WA == 1) {
                            a = xf;
                        } else {
                            a = ring[J % RINGN];
#pragma unroll
                            for (int k = 1; k < RINGN; k++)
                                a = make_double2(pick_mmd<OP>(ring[(J + k) % RINGN].x, a.x), pick_mmd<OP>(ring[(J + k) % RINGN].y, a.y));
                            a = make_double2(pick_mmd<OP>(xf.x, a.x), pick_mmd<OP>(xf.y, a.y));
                        }
                    } else if constexpr (WA == 1) {
                        a = make_double2(wav[0] * xf.x, wav[0] * xf.y);
                    } else {
                        a = make_double2(wav[0] * ring[J % RINGN].x, wav[0] * ring[J % RINGN].y);
#pragma unroll
                        for (int k = 1; k < RINGN; k++)
                            a = make_double2(fma(wav[k], ring[(J + k) % RINGN].x, a.x), fma(wav[k], ring[(J + k) % RINGN].y, a.y));
                        a = make_double2(fma(wav[WA - 1], xf.x, a.x), fma(wav[WA - 1], xf.y, a.y));
                    }
                    const unsigned so = (unsigned)(a0 + i - (WA - 1)) * strideA * 8u;
                    buffer_store_b128_soff(d2_to_u32(a), rout, voff, so);
                }
                if constexpr (RINGN > 0) ring[J % RINGN] = xf;
            }
        });
    }
}

template <int WX, int WA, int OP>
static int launch_stream_d(const double *in, double *out, StreamParamsD &p, hipStream_t s)
{
    constexpr int DEPTH = ((WA - 1) % 4 == 0 && WA > 1 && WX <= 9) ? 4 : 2;
    const int nA = p.axis == 0 ? p.nz : p.ny;
    const int nother = p.axis == 0 ? p.ny : p.nz;
    const int nlines = nother * p.nxt;
    int nch = 1;
    {
        double best = 1e300;
        for (int c = 1; c <= nA && c <= 1024; c++) {
            const int chunk = (nA + c - 1) / c;
            if (c > 1 && chunk < 16) break;
            const int real = (nA + chunk - 1) / chunk;
            const double rounds = std::max(1.0, (double)nlines * real / 4096.0);
            const double cost = rounds * (chunk + (WA - 1) + 4.0);
            if (cost < best * 0.999) { best = cost; nch = real; }
        }
    }
    p.chunk = (nA + nch - 1) / nch;
    p.nchunks = (nA + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)p.nx * p.ny * p.nz * 8);
    hipLaunchKernelGGL((stream_pass_f64_kernel<WX, WA, DEPTH, OP>), dim3((waves + 3) / 4), dim3(256), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

#define MI_ODD_CASES_D(X) X(1) X(3) X(5) X(7) X(9) X(11) X(13) X(15) X(17) X(19) X(21) X(23) X(25) X(27) X(29) X(31) X(33)

// one streaming pass along `axis` with `wa` taps, x pass with `wx` taps fused (wx == wa <= 17, or one of them 1)
static int run_stream_pass_f64(const double *in, double *out, int nz, int ny, int nx, int axis, const double *wav, int wa,
                               int oa, int ma, const double *wxv, int wx, int mx, double cval, hipStream_t s)
{
    StreamParamsD p;
    memset(&p, 0, sizeof(p));
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.axis = axis;
    p.wa = wa; p.oa = oa; p.ma = ma; p.mx = mx;
    p.cval = cval;
    p.nxt = (nx + 127) / 128;
    for (int k = 0; k < wa; k++) p.wav[k] = wav ? wav[k] : 1.0;
    for (int k = 0; k < wx; k++) p.wxv[k] = wxv ? wxv[k] : 1.0;
    if (wx == 1) {
        switch (wa) {
#define X(N) case N: return launch_stream_d<1, N, SP_CORR>(in, out, p, s);
            MI_ODD_CASES_D(X)
#undef X
        }
    } else if (wa == 1) {
        switch (wx) {
#define X(N) case N: if constexpr (N > 1) return launch_stream_d<N, 1, SP_CORR>(in, out, p, s); break;
            MI_ODD_CASES_D(X)
#undef X
        }
    } else if (wx == wa) {
        switch (wx) {
#define X(N) case N: if constexpr (N <= 17 && N > 1) return launch_stream_d<N, N, SP_CORR>(in, out, p, s); break;
            MI_ODD_CASES_D(X)
#undef X
        }
    }
    set_error("float64 stream pass: unsupported x/axis tap combination %d/%d", wx, wa);
    return MI_ERR_UNSUPPORTED;
}

template <int OP>
static int minmax_pass_d(const double *in, double *out, StreamParamsD &p, int wa, int wx, hipStream_t s)
{
#define MM(WXV, WAV) return launch_stream_d<WXV, WAV, OP>(in, out, p, s)
    if (wx == 1) {
        switch (wa) { case 3: MM(1, 3); case 5: MM(1, 5); case 7: MM(1, 7); case 9: MM(1, 9); }
    } else if (wa == 1) {
        switch (wx) { case 3: MM(3, 1); case 5: MM(5, 1); case 7: MM(7, 1); case 9: MM(9, 1); }
    } else if (wa == wx) {
        switch (wx) { case 3: MM(3, 3); case 5: MM(5, 5); case 7: MM(7, 7); case 9: MM(9, 9); }
    }
#undef MM
    set_error("float64 stream min/max pass: unsupported sizes %d/%d", wx, wa);
    return MI_ERR_UNSUPPORTED;
}

// common argument checks; returns MI_OK and fills the extents, or an error / MI_ERR_UNSUPPORTED
static int f64_geometry(const mi_array *in, const mi_array *out, const char *who, int64_t *nz, int64_t *ny, int64_t *nx)
{
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("%s: %s", who, msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3 || in->dtype != MI_F64 || out->dtype != MI_F64) UNSUP("needs 3-D float64 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    *nz = in->shape[0]; *ny = in->shape[1]; *nx = in->shape[2];
    if (*nz < 1 || *ny < 1 || *nx < 4 || (*nx & 1)) UNSUP("x extent must be even, >= 4");
    if (*nz * *ny * *nx * 8 >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
#undef UNSUP
    return MI_OK;
}

static bool x_extent_ok(int64_t nx, int wx)
{
    if (wx <= 1) return true;
    const int nb = (wx / 2 + 1) / 2;
    const int64_t tail = nx & 127;
    return !(nx < 2 * nb + 2 || (tail != 0 && tail < 2 * nb + 2));
}

// ---------------------------------------------------------------------------
// 3 x 3 median of float64 images (method: median2d.hip), two pixels per lane
// ---------------------------------------------------------------------------
struct Med2dParamsD {
    int nx, ny, nz;
    int mx, my;
    double cval;
    int chunk, nchunks, nxt;
    int swz;
};

__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double dmed3(double a, double b, double c) { return dmax(dmin(a, b), dmin(dmax(a, b), c)); }
__device__ __forceinline__ void dsort3(double a, double b, double c, double &lo, double &mid, double &hi)
{
    const double mn = dmin(a, b), mx = dmax(a, b);
    lo = dmin(mn, c);
    hi = dmax(mx, c);
    mid = dmax(mn, dmin(mx, c));
}

__global__ void __launch_bounds__(256)
median3x3_f64_kernel(const double *__restrict__ in, double *__restrict__ out, const Med2dParamsD p)
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
    const int x0 = xt * 128;
    const int nlanes = min(64, (nx - x0) >> 1);
    const int last = nlanes - 1;

    const unsigned plane = (unsigned)ny * (unsigned)nx;
    const unsigned rowbase = (unsigned)z * plane;
    const unsigned total_bytes = plane * (unsigned)nz * 8u;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)total_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)total_bytes, 0x00020000);
    const unsigned voff = lane < nlanes ? (rowbase + (unsigned)(x0 + 2 * lane)) * 8u : kOOB;
    const int side = lane == 0 ? 0 : 1;
    int est, ekind;
    edge_block2(side, 1, x0, x0 + 2 * nlanes, nx, p.mx, &est, &ekind);
    const unsigned evoff = ((lane == 0 || lane == last) && ekind != EDGE_CONST) ? (rowbase + (unsigned)est) * 8u : kOOB;

    const int a0 = c * p.chunk;
    const int a1 = min(a0 + p.chunk, ny);
    const int nsteps = a1 - a0 + 2;
    const int ai0 = a0 - 1;

    struct Slot { double2 v; double2 e; bool cst; };
    Slot S[DEPTH];
    auto issue = [&](int i, Slot &s) {
        int ai = ai0 + i;
        if ((unsigned)ai >= (unsigned)ny) ai = bmap<int>(ai, ny, p.my);
        s.cst = ai < 0;
        const unsigned soff = (unsigned)max(ai, 0) * (unsigned)nx * 8u;
        s.v = as_d2(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : voff, soff, 0));
        s.e = as_d2(__builtin_amdgcn_raw_buffer_load_b128(rin, s.cst ? kOOB : evoff, soff, 0));
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++)
        if (d < nsteps) issue(d, S[d]);

    const double2 cv2 = make_double2(p.cval, p.cval);
    double2 rv[2] = {cv2, cv2};
    double re[2] = {p.cval, p.cval};
    for (int i0 = 0; i0 < nsteps; i0 += DEPTH) {
        static_for<DEPTH>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                Slot &s = S[J];
                const double2 v = s.cst ? cv2 : s.v;
                const double2 eb = s.cst ? cv2 : apply_kind2(s.e, ekind, side, p.cval);
                const double ec = side == 0 ? eb.y : eb.x;
                if (i + DEPTH < nsteps) issue(i + DEPTH, s);
                if (i >= 2) {
                    double lo0, mid0, hi0, lo1, mid1, hi1, elo, emid, ehi;
                    dsort3(rv[0].x, rv[1].x, v.x, lo0, mid0, hi0);
                    dsort3(rv[0].y, rv[1].y, v.y, lo1, mid1, hi1);
                    dsort3(re[0], re[1], ec, elo, emid, ehi);
                    const double lo_l = dppd_from_left(elo, lo1), mid_l = dppd_from_left(emid, mid1), hi_l = dppd_from_left(ehi, hi1);
                    double lo_r = dppd_from_right(elo, lo0), mid_r = dppd_from_right(emid, mid0), hi_r = dppd_from_right(ehi, hi0);
                    if (lane == last) { lo_r = elo; mid_r = emid; hi_r = ehi; }
                    double2 o;
                    o.x = dmed3(dmax(dmax(lo_l, lo0), lo1), dmed3(mid_l, mid0, mid1), dmin(dmin(hi_l, hi0), hi1));
                    o.y = dmed3(dmax(dmax(lo0, lo1), lo_r), dmed3(mid0, mid1, mid_r), dmin(dmin(hi0, hi1), hi_r));
                    const unsigned so = (unsigned)(a0 + i - 2) * (unsigned)nx * 8u;
                    buffer_store_b128_soff(d2_to_u32(o), rout, voff, so);
                }
                rv[J % 2] = v;
                re[J % 2] = ec;
            }
        });
    }
}

int run_median3x3_f64(const mi_array *in, const mi_array *out, int mx, int my, double cval, hipStream_t s)
{
#define UNSUP(msg) do { set_error("median3x3: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if ((in->ndim != 2 && in->ndim != 3) || in->dtype != MI_F64 || out->dtype != MI_F64) UNSUP("needs 2-D / 3-D float64 in/out");
    if (!is_contiguous(in) || !is_contiguous(out)) UNSUP("needs C-contiguous arrays");
    if (in->data == out->data) UNSUP("in-place");
    const int nd = in->ndim;
    const int64_t nz = nd == 3 ? in->shape[0] : 1, ny = in->shape[nd - 2], nx = in->shape[nd - 1];
    if (nz < 1 || ny < 1 || nx < 4 || (nx & 1)) UNSUP("x extent must be even, >= 4");
    { const int64_t tail = nx & 127; if (tail != 0 && tail < 4) UNSUP("x extent unsuitable for the streaming x window"); }
    if (nz * ny * nx * 8 >= ((int64_t)1 << 31)) UNSUP("needs an array < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
#undef UNSUP
    Med2dParamsD p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.mx = mx; p.my = my;
    p.cval = cval;
    p.nxt = (int)((nx + 127) / 128);
    const int nlines = p.nz * p.nxt;
    int nch = (4096 + nlines - 1) / nlines;
    if (nch > p.ny / 16) nch = p.ny / 16;
    if (nch < 1) nch = 1;
    p.chunk = (p.ny + nch - 1) / nch;
    p.nchunks = (p.ny + p.chunk - 1) / p.chunk;
    const int waves = nlines * p.nchunks;
    p.swz = xcd_swizzle_for((size_t)nx * ny * nz * 8);
    hipLaunchKernelGGL(median3x3_f64_kernel, dim3((waves + 3) / 4), dim3(256), 0, s, (const double *)in->data, (double *)out->data, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

}  // namespace mi

using namespace mi;

/* Separable filter on a float64 volume / image (one-plane volume) as streaming passes (declared in
 * include/mi355img.h). */
extern "C" int mi_separable3d_f64(const mi_array *in, const mi_array *out, const double *const weights[3], const int wlen[3],
                                  const int origin[3], const int mode[3], double cval, mi_stream stream)
{
    int64_t nz, ny, nx;
    int rc = f64_geometry(in, out, "separable3d_f64", &nz, &ny, &nx);
    if (rc) return rc;
    MI_REQUIRE(weights && wlen && origin && mode, MI_ERR_INVALID_ARG, "NULL argument");
#define UNSUP(msg) do { set_error("separable3d_f64: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    int w[3], off[3];
    for (int a = 0; a < 3; a++) {
        w[a] = weights[a] ? wlen[a] : 1;
        if (w[a] < 1 || w[a] > kStreamMaxTaps || !(w[a] & 1)) UNSUP("taps must be odd and <= 33");
        off[a] = w[a] / 2 + (weights[a] ? origin[a] : 0);
        if (off[a] < 0 || off[a] >= w[a]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    if (weights[2] && origin[2] != 0) UNSUP("x origin must be 0");
    if (!x_extent_ok(nx, w[2])) UNSUP("x extent unsuitable for the streaming x pass");
    const int mz = filter_mode(mode[0]), my = filter_mode(mode[1]), mx = filter_mode(mode[2]);
    if (mz == MI_MODE_CONSTANT || my == MI_MODE_CONSTANT || mx == MI_MODE_CONSTANT) {
        // SciPy extends the INTERMEDIATE array of every pass by cval; the passes here run in another order (x first),
        // which is the same thing only for kernels that map a constant onto itself
        for (int a = 0; a < 3; a++) {
            if (!weights[a]) continue;
            double sum = 0.0;
            for (int k = 0; k < w[a]; k++) sum += weights[a][k];
            if (fabs(sum - 1.0) > 4e-16 * w[a]) UNSUP("constant mode needs kernels that sum to one");
        }
    }
    hipStream_t s = resolve_stream(stream);
    struct Pass { int axis, wa, oa, ma, wx; };
    Pass passes[3];
    int np = 0;
    const bool fuse_xz = w[2] > 1 && w[2] == w[0] && w[2] <= 17;
    const bool fuse_xy = !fuse_xz && w[2] > 1 && w[2] == w[1] && w[2] <= 17;
    if (w[2] > 1 && !fuse_xz && !fuse_xy) passes[np++] = {1, 1, 0, my, w[2]};      // x only (streams over y)
    if (w[0] > 1) passes[np++] = {0, w[0], off[0], mz, fuse_xz ? w[2] : 1};
    if (w[1] > 1) passes[np++] = {1, w[1], off[1], my, fuse_xy ? w[2] : 1};
    if (np == 0) UNSUP("nothing to filter");
    const size_t bytes = (size_t)(nz * ny * nx) * sizeof(double);
    void *tmp[2] = {nullptr, nullptr};
    for (int t = 0; t < np - 1 && t < 2; t++)
        if ((rc = pool_alloc(&tmp[t], bytes, s))) { if (tmp[0]) pool_free(tmp[0]); return rc; }
    const double *src = (const double *)in->data;
    for (int i = 0; i < np && rc == MI_OK; i++) {
        double *dst = i == np - 1 ? (double *)out->data : (double *)tmp[i & 1];
        const Pass &q = passes[i];
        rc = run_stream_pass_f64(src, dst, (int)nz, (int)ny, (int)nx, q.axis, q.wa > 1 ? weights[q.axis == 0 ? 0 : 1] : nullptr,
                                 q.wa, q.oa, q.ma, q.wx > 1 ? weights[2] : nullptr, q.wx, mx, cval, s);
        src = dst;
    }
    for (int t = 0; t < 2; t++) if (tmp[t]) pool_free(tmp[t]);
    return rc;
#undef UNSUP
}

/* Separable flat min / max on a float64 volume / image, odd sizes <= 9 (declared in include/mi355img.h). */
extern "C" int mi_minmax3d_f64(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                               const int mode[3], double cval, int is_max, mi_stream stream)
{
    int64_t nz, ny, nx;
    int rc = f64_geometry(in, out, "minmax3d_f64", &nz, &ny, &nx);
    if (rc) return rc;
    MI_REQUIRE(size && origin && mode, MI_ERR_INVALID_ARG, "NULL argument");
#define UNSUP(msg) do { set_error("minmax3d_f64: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    int w[3], off[3];
    for (int a = 0; a < 3; a++) {
        w[a] = size[a];
        if (w[a] < 1 || w[a] > 9 || !(w[a] & 1)) UNSUP("sizes must be odd and <= 9");
        off[a] = w[a] / 2 + origin[a];
        if (off[a] < 0 || off[a] >= w[a]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    if (origin[2] != 0) UNSUP("x origin must be 0");
    if (!x_extent_ok(nx, w[2])) UNSUP("x extent unsuitable for the streaming x pass");
    const int mz = filter_mode(mode[0]), my = filter_mode(mode[1]), mx = filter_mode(mode[2]);
    hipStream_t s = resolve_stream(stream);
    struct Pass { int axis, wa, oa, ma, wx; };
    Pass passes[3];
    int np = 0;
    const bool fuse_xz = w[2] > 1 && w[2] == w[0];
    const bool fuse_xy = !fuse_xz && w[2] > 1 && w[2] == w[1];
    if (w[2] > 1 && !fuse_xz && !fuse_xy) passes[np++] = {1, 1, 0, my, w[2]};
    if (w[0] > 1) passes[np++] = {0, w[0], off[0], mz, fuse_xz ? w[2] : 1};
    if (w[1] > 1) passes[np++] = {1, w[1], off[1], my, fuse_xy ? w[2] : 1};
    if (np == 0) UNSUP("nothing to filter");
    const size_t bytes = (size_t)(nz * ny * nx) * sizeof(double);
    void *tmp[2] = {nullptr, nullptr};
    for (int t = 0; t < np - 1 && t < 2; t++)
        if ((rc = pool_alloc(&tmp[t], bytes, s))) { if (tmp[0]) pool_free(tmp[0]); return rc; }
    const double *src = (const double *)in->data;
    for (int i = 0; i < np && rc == MI_OK; i++) {
        double *dst = i == np - 1 ? (double *)out->data : (double *)tmp[i & 1];
        const Pass &q = passes[i];
        StreamParamsD p;
        memset(&p, 0, sizeof(p));
        p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
        p.axis = q.axis; p.wa = q.wa; p.oa = q.oa; p.ma = q.ma; p.mx = mx;
        p.cval = cval;
        p.nxt = (int)((nx + 127) / 128);
        rc = is_max ? minmax_pass_d<SP_MAX>(src, dst, p, q.wa, q.wx, s) : minmax_pass_d<SP_MIN>(src, dst, p, q.wa, q.wx, s);
        src = dst;
    }
    for (int t = 0; t < 2; t++) if (tmp[t]) pool_free(tmp[t]);
    return rc;
#undef UNSUP
}
