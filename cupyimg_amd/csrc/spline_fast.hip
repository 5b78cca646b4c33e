// spline_fast.hip -- r5: the B-spline prefilter at ONE memory sweep per axis.
//
// Reference: cupyimg/scipy/ndimage/interpolation.py:105-268 (spline_filter1d / spline_filter), the recursion itself
// _spline_prefilter_core.py:216-287 (causal / anti-causal IIR sweeps of one thread per line).  Order 3 is the DEFAULT
// order of rotate / affine_transform / zoom / shift / map_coordinates, so these passes run before every default call.
//
// What was there (interp.hip): one thread per line sweeps forward over the whole line and back -- every coefficient goes
// through HBM twice per pass (2.06 x the algorithmic bytes, 0.41 ms per strided pass on 512^3 float32), and the pass along
// the contiguous axis keeps twelve lines per wave in LDS with twelve of the 64 lanes computing (0.5 ms).
//
// Here (single-pole orders 2 and 3; mirror / reflect ends; lines of >= 64 samples; many lines):
//
//  * STRIDED axes -- `spline_stream_kernel`: a thread owns one line and walks along it in chunks of C samples that it holds
//    IN REGISTERS.  The causal sweep is exact (its state is one register carried from chunk to chunk).  The anti-causal
//    sweep of a chunk starts H samples BEYOND the chunk from the steady-state value c+ * z / (z - 1): the influence of a
//    start value decays as |z|^k (|z| <= 0.268), i.e. below 4e-12 (float32 coefficients, H = 20) / 5e-19 (float64,
//    H = 32) of the data range before the first sample that is stored -- far below one ulp of the coefficient type.  Those
//    H causal values are not thrown away: they are the first H of the next chunk.  Every sample is read once and written
//    once (1.0 x the algorithmic bytes), C independent loads are in flight per thread, lanes run along the contiguous
//    axis (coalesced rows).  The last chunk of a line uses the exact end condition.
//
//  * the CONTIGUOUS axis -- `spline_rows_scan_kernel`: a wave owns a line, every lane four consecutive samples (one 16-byte
//    load), and the recursion y[i] = x[i] + z y[i-1] is evaluated as a weighted prefix scan over the lanes with DPP row
//    shifts: four samples locally, then shifts by 1 / 2 / 4 / 8 lanes with weights z^4, z^8, z^16, z^32 inside a row of 16
//    lanes, the carry between the rows (z^64 < 3e-37: one term suffices) by row_bcast:15 / one bpermute.  No LDS, no
//    restart, all 64 lanes busy; a line of up to 2048 samples stays in registers between the two sweeps (no rounding to
//    the coefficient type in between).
//
// Both compute in double (like every prefilter kernel of this library) and scale the samples by the pole gain as they
// are read (SciPy's order).  Results agree with the sequential kernels to the truncation bound above -- i.e. to rounding
// of the coefficient type -- not bit for bit: a request for SciPy's exact arithmetic (kSplExact: integer outputs, where
// the last bit decides .5 ties) never comes here.
#include "common.hpp"

namespace mi {
void note_kernel(const char *fmt, ...);        // runtime.hip

static Knob g_spline_fast{1};      // test hook: 0 = never these kernels, 1 = by the rules below, 2 = also for few lines
}  // namespace mi
extern "C" int mi_debug_set_spline_fast(int k) { mi::g_spline_fast = k; return MI_OK; }

namespace mi {

constexpr double kPole2 = -0.171572875253809902396622551580603843;
constexpr double kPole3 = -0.267949192431122706472553658494127633;

struct SplStream {
    long long n;             // samples per line
    long long stride;        // elements between consecutive samples of a line (= lines per outer block)
    long long outer;         // outer blocks (each n * stride elements)
    int cols_blocks;         // workgroups per outer block along the lines
    int smode;               // 0 mirror, 1 reflect
    double z, gain;
};

template <typename T> struct SplCfg;
template <> struct SplCfg<float> { static constexpr int C = 28, H = 20; };      // 128 registers: four waves per SIMD (C = 32 spills)
template <> struct SplCfg<double> { static constexpr int C = 12, H = 32; };     // 146 registers: three waves per SIMD

// element access through a buffer descriptor: the per-sample part of the address (sample index x line stride) is UNIFORM and
// travels in the scalar offset, the lane's column in one vector register -- with plain pointers every load in flight held
// its own 64-bit address (64 registers for the 32 loads of a chunk: 162 VGPRs, three waves per SIMD)
template <typename T> __device__ __forceinline__ T buf_load(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ float buf_load<float>(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <> __device__ __forceinline__ double buf_load<double>(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
    const u32x2s q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return __longlong_as_double(((long long)q.y << 32) | (unsigned long long)q.x);
}
__device__ __forceinline__ void buf_store(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double v)
{
    typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const u32x2s q = {(unsigned)(b & 0xffffffffull), (unsigned)(b >> 32)};
    __builtin_amdgcn_raw_buffer_store_b64(q, r, voff, soff, 0);
}

// one thread per line; blockIdx.x = outer * cols_blocks + column block.  The descriptors start at the workgroup's first
// line: every offset inside is below 2^32 (checked by the launch).
template <typename CF, typename SRC>
__global__ void __launch_bounds__(256, sizeof(CF) == 4 ? 4 : 3)          // waves per SIMD: 128 / 168 registers
spline_stream_kernel(const SRC *src, CF *dst, const SplStream p)
{
    constexpr int C = SplCfg<CF>::C, H = SplCfg<CF>::H;
    const int cb = blockIdx.x % p.cols_blocks;
    const long long o = blockIdx.x / p.cols_blocks;
    const long long col0 = (long long)cb * 256;
    const int lines_here = (int)min((long long)256, p.stride - col0);
    const bool live = (int)threadIdx.x < lines_here;
    const unsigned lcol = live ? threadIdx.x : lines_here - 1;                     // dead lanes shadow the last line, store nothing
    const long long base = o * p.n * p.stride + col0;
    const unsigned n = (unsigned)p.n;
    const unsigned span = (unsigned)((p.n - 1) * p.stride + lines_here);          // elements the workgroup touches
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(src + base), 0, (int)(span * (unsigned)sizeof(SRC)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rdst = __builtin_amdgcn_make_buffer_rsrc((void *)(dst + base), 0, (int)(span * (unsigned)sizeof(CF)), 0x00020000);
    const unsigned vs = lcol * (unsigned)sizeof(SRC), vd = lcol * (unsigned)sizeof(CF);
    const unsigned ss = (unsigned)p.stride * (unsigned)sizeof(SRC), sd = (unsigned)p.stride * (unsigned)sizeof(CF);       // bytes between samples
    const double z = p.z, g = p.gain;

    CF cp[C + H];                         // causal values of samples a .. a + C + H - 1 (a = start of the current chunk)
    CF cm1 = (CF)0;                       // c+[a - 1]
    double prev;
    // ---- start of the line: c+[0] from the boundary sum over the first H samples (the terms of the far end are
    // z^(n-1) <= z^63 times smaller: dropped), then the causal values of samples 1 .. H - 1
    {
        SRC xs[H];
#pragma unroll
        for (int i = 0; i < H; i++) xs[i] = buf_load<SRC>(rs, vs, (unsigned)i * ss);
        double acc = 0.0, zi = 1.0;
#pragma unroll
        for (int i = 0; i < H; i++) { acc = fma(zi, (double)xs[i] * g, acc); zi *= z; }        // sum z^i x[i]
        prev = p.smode == 0 ? acc : fma(z, acc, (double)xs[0] * g);                            // mirror: the sum; reflect: x0 + z * sum
        cp[0] = (CF)prev;
#pragma unroll
        for (int i = 1; i < H; i++) { prev = fma(z, prev, (double)xs[i] * g); cp[i] = (CF)prev; }
    }
    const double zr = z / (z - 1.0);                  // steady-state start of an anti-causal sweep: c = c+ * z / (z - 1)
    const double ze = z / (z * z - 1.0);              // mirror end: c[n-1] = (z c+[n-2] + c+[n-1]) * z / (z^2 - 1)
    for (unsigned a = 0; a < n; a += C) {
        const bool full = a + C + H <= n;             // every sample of the chunk and its look-ahead exists
        if (full) {
            SRC xs[C];
#pragma unroll
            for (int u = 0; u < C; u++) xs[u] = buf_load<SRC>(rs, vs, (a + H + u) * ss);
#pragma unroll
            for (int u = 0; u < C; u++) { prev = fma(z, prev, (double)xs[u] * g); cp[H + u] = (CF)prev; }
            cm1 = cp[C - 1];                          // c+ just before the next chunk (its mirror end condition may need it)
            // anti-causal from sample a + C + H - 1: the exact end condition when that is the line's last sample
            double nxt;
            if (a + C + H == n) nxt = p.smode == 0 ? fma(z, (double)cp[C + H - 2], prev) * ze : prev * zr;
            else nxt = prev * zr;
#pragma unroll
            for (int j = C + H - 2; j >= C; j--) nxt = fma(z, nxt, -z * (double)cp[j]);
            // (the results replace cp[0 .. C): dead after this sweep, overwritten by the shift below; one exec mask for all stores)
#pragma unroll
            for (int j = C - 1; j >= 0; j--) { nxt = fma(z, nxt, -z * (double)cp[j]); cp[j] = (CF)nxt; }
            if (live) {
#pragma unroll
                for (int j = 0; j < C; j++) buf_store(rdst, vd, (a + j) * sd, cp[j]);
            }
        } else {
            // the tail of the line: samples a .. n - 1 (at most C + H - 1 of them), the end condition at n - 1
            const int e = (int)(n - 1 - a);           // index of the last sample in cp[]
            const CF before = cm1;                    // c+[a - 1]
#pragma unroll
            for (int u = 0; u < C; u++) {
                if (H + u <= e) { prev = fma(z, prev, (double)buf_load<SRC>(rs, vs, (a + H + u) * ss) * g); cp[H + u] = (CF)prev; }
            }
            cm1 = cp[C - 1];
            double nxt = 0.0;
#pragma unroll
            for (int j = C + H - 1; j >= 0; j--) {
                if (j == e) nxt = p.smode == 0 ? fma(z, (double)(j > 0 ? cp[j > 0 ? j - 1 : 0] : before), (double)cp[j]) * ze : (double)cp[j] * zr;
                else if (j < e) nxt = fma(z, nxt, -z * (double)cp[j]);
                if (j < C && j <= e && live) buf_store(rdst, vd, (a + j) * sd, (CF)nxt);
            }
        }
#pragma unroll
        for (int i = 0; i < H; i++) cp[i] = cp[C + i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// contiguous lines: weighted prefix scans over the lanes
// ---------------------------------------------------------------------------------------------------------------------
// row_shr:N inside rows of 16 lanes, lanes without a source get `fill`
template <int CTRL>
__device__ __forceinline__ double dpp_move(double fill, double v)
{
    const long long vi = __double_as_longlong(v), fi = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)(fi & 0xffffffffll), (int)(vi & 0xffffffffll), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(fi >> 32), (int)(vi >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// lane 15 of every row of 16 to all lanes of the NEXT row; row 0 keeps `fill`
__device__ __forceinline__ double dpp_bcast15(double fill, double v)
{
    const long long vi = __double_as_longlong(v), fi = __double_as_longlong(fill);
    const int lo = __builtin_amdgcn_update_dpp((int)(fi & 0xffffffffll), (int)(vi & 0xffffffffll), 0x142, 0xe, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(fi >> 32), (int)(vi >> 32), 0x142, 0xe, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ double lane_read(double v, int lane)
{
    const long long vi = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(vi & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(vi >> 32), lane);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

struct SplScanW { double z1, z2, z3, z4, z8, z16, z32; };

// forward scan of one segment of 256 samples (4 per lane): y[k] = x[k] + z y[k - 1] with `cin` = y just before the segment;
// returns y of the segment's last sample (the next segment's cin)
template <bool WIDE>
__device__ __forceinline__ double scan_fwd(double (&v)[4], double cin, const SplScanW &w, double zl /* z^(4 (lane % 16) + 4) */)
{
    v[1] = fma(w.z1, v[0], v[1]);
    v[2] = fma(w.z1, v[1], v[2]);
    v[3] = fma(w.z1, v[2], v[3]);
    double P = v[3];
    P = fma(w.z4, dpp_move<0x111>(0.0, P), P);
    P = fma(w.z8, dpp_move<0x112>(0.0, P), P);
    P = fma(w.z16, dpp_move<0x114>(0.0, P), P);
    if (WIDE) P = fma(w.z32, dpp_move<0x118>(0.0, P), P);
    const double rin = dpp_bcast15(cin, P);                 // what enters the row: the previous row's end (row 0: cin)
    const double E = fma(zl, rin, P);                       // y at the end of this lane
    const double in = dpp_move<0x111>(rin, E);              // y just before this lane (first lane of a row: rin)
    v[0] = fma(w.z1, in, v[0]);
    v[1] = fma(w.z2, in, v[1]);
    v[2] = fma(w.z3, in, v[2]);
    v[3] = fma(w.z4, in, v[3]);
    return lane_read(E, 63);
}

// backward scan: y[k] = u[k] + z y[k + 1], `cin` = y just after the segment; returns y of the segment's first sample
template <bool WIDE>
__device__ __forceinline__ double scan_bwd(double (&v)[4], double cin, const SplScanW &w, double zl /* z^(4 (15 - lane % 16) + 4) */, int lane)
{
    v[2] = fma(w.z1, v[3], v[2]);
    v[1] = fma(w.z1, v[2], v[1]);
    v[0] = fma(w.z1, v[1], v[0]);
    double P = v[0];
    P = fma(w.z4, dpp_move<0x101>(0.0, P), P);              // row_shl:1 .. 8
    P = fma(w.z8, dpp_move<0x102>(0.0, P), P);
    P = fma(w.z16, dpp_move<0x104>(0.0, P), P);
    if (WIDE) P = fma(w.z32, dpp_move<0x108>(0.0, P), P);
    // what enters the row from above: the first lane of the NEXT row (row 3: cin)
    const double up = __shfl(P, ((lane | 15) + 1) & 63, 64);
    const double rin = lane >= 48 ? cin : up;
    const double E = fma(zl, rin, P);                       // y at the START of this lane
    const double in = dpp_move<0x101>(rin, E);              // y just after this lane (last lane of a row: rin)
    v[3] = fma(w.z1, in, v[3]);
    v[2] = fma(w.z2, in, v[2]);
    v[1] = fma(w.z3, in, v[1]);
    v[0] = fma(w.z4, in, v[0]);
    return lane_read(E, 0);
}

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec4<double> { typedef double type __attribute__((ext_vector_type(4))); };

struct SplRows {
    int n;                   // samples per line (>= 64)
    long long nlines;
    int smode;
    double z, gain;
};

// one wave per line, four waves per workgroup; NSEG = segments of 256 samples per line
template <typename CF, typename SRC, int NSEG>
__global__ void __launch_bounds__(256)
spline_rows_scan_kernel(const SRC *src, CF *dst, const SplRows p)
{
    constexpr bool WIDE = sizeof(CF) == 8;          // float32 coefficients: z^32 < 5e-19 of the data range is not representable
    const int lane = threadIdx.x & 63;
    const long long line = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= p.nlines) return;
    const int n = p.n;
    const double z = p.z;
    SplScanW w;
    w.z1 = z; w.z2 = z * z; w.z3 = w.z2 * z; w.z4 = w.z2 * w.z2; w.z8 = w.z4 * w.z4; w.z16 = w.z8 * w.z8; w.z32 = w.z16 * w.z16;
    // z^(4 m + 4), m = lane % 16 (forward) / 15 - lane % 16 (backward): by squaring
    auto zpow4 = [&](int m) {
        double r = w.z4;                           // m = 0
        if (m & 1) r *= w.z4;
        if (m & 2) r *= w.z8;
        if (m & 4) r *= w.z16;
        if (m & 8) r *= w.z32;
        return r;
    };
    const double zf = zpow4(lane & 15), zb = zpow4(15 - (lane & 15));
    const SRC *rd = src + line * n;
    CF *wr = dst + line * n;
    typedef typename Vec4<SRC>::type SV;
    typedef typename Vec4<CF>::type CV;
    double v[NSEG][4];
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
        const int i0 = s * 256 + lane * 4;
        if (i0 + 3 < n) {
            // (lines of any length, r6: a line then starts on an element boundary only -- vector accesses need no more)
            const SV q = *reinterpret_cast<const SV *>(rd + i0);
            v[s][0] = (double)q.x * p.gain; v[s][1] = (double)q.y * p.gain; v[s][2] = (double)q.z * p.gain; v[s][3] = (double)q.w * p.gain;
        } else {
            // the lane that holds the end of the line (n % 4 samples of it), and the lanes beyond
#pragma unroll
            for (int j = 0; j < 4; j++) v[s][j] = i0 + j < n ? (double)rd[i0 + j] * p.gain : 0.0;
        }
    }
    // ---- c+[0]: the boundary sum over the first 64 samples (lanes 0 .. 15 of segment 0; z^64 ends the series)
    {
        double q = fma(w.z1, fma(w.z1, fma(w.z1, v[0][3], v[0][2]), v[0][1]), v[0][0]);      // sum_k z^k x[4 lane + k]
        q = fma(w.z4, dpp_move<0x101>(0.0, q), q);
        q = fma(w.z8, dpp_move<0x102>(0.0, q), q);
        q = fma(w.z16, dpp_move<0x104>(0.0, q), q);
        q = fma(w.z32, dpp_move<0x108>(0.0, q), q);
        const double sum = lane_read(q, 0);
        const double x0 = lane_read(v[0][0], 0);
        const double c0 = p.smode == 0 ? sum : fma(z, sum, x0);
        if (lane == 0) v[0][0] = c0;               // with nothing entering the line, y[0] = c0 and the recursion goes on from it
    }
    // ---- causal
    double carry = 0.0;
#pragma unroll
    for (int s = 0; s < NSEG; s++) carry = scan_fwd<WIDE>(v[s], carry, w, zf);
    // ---- the anti-causal input: u[i] = -z c+[i] below the last sample, the end condition at it, nothing beyond
    const int last = n - 1;
    // c+[last] and c+[last - 1] wherever they sit (segment, lane, element: the same in every line)
    double cl = 0.0, cp = 0.0;
    const int prv = last - 1;
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
        if ((last >> 8) == s) {
            const int e = last & 3;
            cl = lane_read(e == 0 ? v[s][0] : (e == 1 ? v[s][1] : (e == 2 ? v[s][2] : v[s][3])), (last & 255) >> 2);
        }
        if ((prv >> 8) == s) {
            const int e = prv & 3;
            cp = lane_read(e == 0 ? v[s][0] : (e == 1 ? v[s][1] : (e == 2 ? v[s][2] : v[s][3])), (prv & 255) >> 2);
        }
    }
    const double end = p.smode == 0 ? fma(z, cp, cl) * (z / (z * z - 1.0)) : cl * (z / (z - 1.0));
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
        const int i0 = s * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < 4; j++) v[s][j] = i0 + j > last ? 0.0 : (i0 + j == last ? end : -z * v[s][j]);
    }
    carry = 0.0;
#pragma unroll
    for (int s = NSEG - 1; s >= 0; s--) carry = scan_bwd<WIDE>(v[s], carry, w, zb, lane);
#pragma unroll
    for (int s = 0; s < NSEG; s++) {
        const int i0 = s * 256 + lane * 4;
        if (i0 + 3 < n) {
            CV q;
            q.x = (CF)v[s][0]; q.y = (CF)v[s][1]; q.z = (CF)v[s][2]; q.w = (CF)v[s][3];
            *reinterpret_cast<CV *>(wr + i0) = q;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (i0 + j < n) wr[i0 + j] = (CF)v[s][j];
        }
    }
}

template <typename CF, typename SRC>
static int launch_rows(const void *src, void *dst, const SplRows &p, hipStream_t s)
{
    const int nseg = (p.n + 255) / 256;
    const dim3 grid((unsigned)((p.nlines + 3) / 4)), block(256);
    switch (nseg) {
    case 1: hipLaunchKernelGGL((spline_rows_scan_kernel<CF, SRC, 1>), grid, block, 0, s, (const SRC *)src, (CF *)dst, p); break;
    case 2: hipLaunchKernelGGL((spline_rows_scan_kernel<CF, SRC, 2>), grid, block, 0, s, (const SRC *)src, (CF *)dst, p); break;
    case 3: hipLaunchKernelGGL((spline_rows_scan_kernel<CF, SRC, 3>), grid, block, 0, s, (const SRC *)src, (CF *)dst, p); break;
    case 4: hipLaunchKernelGGL((spline_rows_scan_kernel<CF, SRC, 4>), grid, block, 0, s, (const SRC *)src, (CF *)dst, p); break;
    case 5: case 6: hipLaunchKernelGGL((spline_rows_scan_kernel<CF, SRC, 6>), grid, block, 0, s, (const SRC *)src, (CF *)dst, p); break;
    default: hipLaunchKernelGGL((spline_rows_scan_kernel<CF, SRC, 8>), grid, block, 0, s, (const SRC *)src, (CF *)dst, p); break;
    }
    MI_HIP(hipGetLastError());
    return MI_OK;
}

template <typename CF, typename SRC>
static int launch_stream(const void *src, void *dst, const SplStream &p, hipStream_t s)
{
    const long long blocks = p.outer * p.cols_blocks;
    hipLaunchKernelGGL((spline_stream_kernel<CF, SRC>), dim3((unsigned)blocks), dim3(256), 0, s, (const SRC *)src, (CF *)dst, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// One prefilter pass along `axis` of the contiguous array described by `shape` (its dtype = the coefficient type), samples
// read from `src` (of dtype src_dtype: the array itself, or the caller's input on the first pass), coefficients written to
// `dst`.  Returns false when these kernels do not take the pass (the caller runs the sequential ones); *rc is the result
// of the launch otherwise.  spline_mode as mi_spline_filter1d (0 mirror, 1 reflect, 2 grid-wrap | kSplExact 0x100).
bool spline_pass_fast(const mi_array *shape, const void *src, int src_dtype, void *dst, int axis, int order, int spline_mode, hipStream_t s, int *rc)
{
    *rc = MI_OK;
    const int knob = g_spline_fast;
    if (!knob || (spline_mode & 0x100) || (spline_mode & 0xff) > 1 || (order != 2 && order != 3)) return false;
    if (shape->dtype != MI_F32 && shape->dtype != MI_F64) return false;
    if (src_dtype != MI_F32 && src_dtype != MI_F64) return false;
    if (src_dtype == MI_F64 && shape->dtype == MI_F32) return false;              // never narrows on the way in
    const long long total = numel(shape);
    const long long n = shape->shape[axis];
    if (total == 0 || n < 64) return false;
    long long inner = 1;
    for (int d = axis + 1; d < shape->ndim; d++) inner *= shape->shape[d];
    const long long nlines = total / n, outer = nlines / inner;
    // few long lines are the blocked kernels' ground (interp.hip): a thread per line wants >= 16384 lines (one wave per CU), a
    // wave per line >= 2048
    if (nlines < (inner == 1 ? 2048 : 16384) && knob < 2) return false;
    const double z = order == 2 ? kPole2 : kPole3;
    const double gain = (1.0 - z) * (1.0 - 1.0 / z);
    const bool f64 = shape->dtype == MI_F64, sf64 = src_dtype == MI_F64;
    if (inner == 1) {
        // four coefficients per lane and access; lines of any length (r6): element alignment is all the vector accesses need
        if (n > 2048 || ((uintptr_t)src & (sf64 ? 7 : 3)) || ((uintptr_t)dst & (f64 ? 7 : 3))) return false;
        if ((nlines + 3) / 4 > 0x7fffffffLL) return false;
        SplRows p;
        p.n = (int)n; p.nlines = nlines; p.smode = spline_mode & 0xff; p.z = z; p.gain = gain;
        note_kernel("mi::spline_rows_scan_kernel<%s,%s,%d> grid=%lld (B-spline prefilter along the contiguous axis: prefix scans over the lanes, one sweep through memory)",
                    f64 ? "double" : "float", sf64 ? "double" : "float", (int)((n + 255) / 256), (nlines + 3) / 4);
        if (f64) *rc = sf64 ? launch_rows<double, double>(src, dst, p, s) : launch_rows<double, float>(src, dst, p, s);
        else *rc = launch_rows<float, float>(src, dst, p, s);
        return true;
    }
    SplStream p;
    p.n = n; p.stride = inner; p.outer = outer;
    p.cols_blocks = (int)((inner + 255) / 256);
    if ((long long)p.cols_blocks * outer > 0x7fffffffLL) return false;
    if ((unsigned long long)n * (unsigned long long)inner * (f64 ? 8ull : 4ull) >= (1ull << 32)) return false;     // 32-bit offsets inside a workgroup's descriptor
    p.smode = spline_mode & 0xff; p.z = z; p.gain = gain;
    note_kernel("mi::spline_stream_kernel<%s,%s> grid=%lld (B-spline prefilter along a strided axis: register chunks of %d samples + %d look-ahead, one sweep through memory)",
                f64 ? "double" : "float", sf64 ? "double" : "float", (long long)p.cols_blocks * outer, f64 ? SplCfg<double>::C : SplCfg<float>::C,
                f64 ? SplCfg<double>::H : SplCfg<float>::H);
    if (f64) *rc = sf64 ? launch_stream<double, double>(src, dst, p, s) : launch_stream<double, float>(src, dst, p, s);
    else *rc = launch_stream<float, float>(src, dst, p, s);
    return true;
}

}  // namespace mi
