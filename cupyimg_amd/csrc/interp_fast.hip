// interp_fast.hip -- order-1 / order-0 interpolation, float32 3-D volumes:
// the throughput kernels behind map_coordinates / affine_transform
// (reference: cupyimg/scipy/ndimage/interpolation.py:271-394, :397-561; kernel
// body _interp_kernels.py:277-592, launched at interpolation.py:393,545,560).
//
// Same tap / weight / boundary logic as the generic kernel in interp.hip, with
// the per-voxel overhead removed: 3-D launch grid (no index division), 32-bit
// indexing, one voxel per lane with lanes along x (a wave's gather touches
// neighbouring input voxels), coordinates in the precision they are given in
// (float32 coordinates of map_coordinates: floor / fraction are exact in
// float32; affine: double, like the reference's float64 matrix path,
// interpolation.py:476,501), weights and the 8-tap accumulation in float32.
// Against SciPy's all-double arithmetic that is ~1e-7 relative; the stated
// tolerance for float32 interpolation is 2e-6 * max|ref| (tests/test_gpu_*).
// Anything else (other dtypes, ranks, float64 output) runs interp.hip.
#include <algorithm>
#include <vector>
#include <type_traits>

#include "common.hpp"

namespace mi {

void note_kernel(const char *fmt, ...);      // separable3d.hip: which kernel a call dispatched (mi_debug_last_kernel)

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct FastInterpParams {
    int nz, ny, nx;          // input
    int oz, oy, ox;          // output
    int order, mode;
    double cval;
    int two_d;               // image (ny, nx) handled as a one-plane volume: the z coordinate is 0
    double m[12];            // affine 3 x 4 (row major)
};

__device__ __forceinline__ double wrapc(double c, int n)
{
    if (n <= 1) return 0.0;
    const double s = (double)(n - 1);
    if (c < 0) c += s * ((double)(long long)(-c / s) + 1.0);
    else if (c > s) c -= s * (double)(long long)(c / s);
    return c;
}

__device__ __forceinline__ double foldc(double c, int n, int mode)
{
    if (n <= 1) return 0.0;
    const double dn = (double)n;
    switch (mode) {
    case MI_MODE_MIRROR: {
        const double p = 2.0 * dn - 2.0;
        if (c < 0) { c = p * (double)(long long)(-c / p) + c; c = c <= 1.0 - dn ? c + p : -c; }
        else if (c > dn - 1.0) { c -= p * (double)(long long)(c / p); if (c >= dn) c = p - c; }
        return c;
    }
    case MI_MODE_REFLECT: {
        const double p = 2.0 * dn;
        if (c < 0) {
            if (c < -p) c = p * (double)(long long)(-c / p) + c;
            c = c < -dn ? c + p : (c > -1e-15 ? 1e-15 : -c) - 1.0;
        } else if (c > dn - 1.0) {
            c -= p * (double)(long long)(c / p);
            if (c >= dn) c = p - c - 1.0;
        }
        return c;
    }
    case MI_MODE_WRAP: return wrapc(c, n);
    case MI_MODE_GRID_WRAP:
        if (c < 0) c += dn * ((double)(long long)((-1.0 - c) / dn) + 1.0);
        else if (c > dn - 1.0) c -= dn * (double)(long long)((c + 1.0) / dn);
        return c;
    case MI_MODE_NEAREST: return c < 0 ? 0.0 : (c > dn - 1.0 ? dn - 1.0 : c);
    default: return c;
    }
}

// taps along one axis for coordinate c: indices (-1 = use cval), float weights.
// `outside` is set when mode == constant and c lies outside [0, n-1].
template <typename CT, bool FASTC, int ORDER, typename WT>
__device__ __forceinline__ void axis_taps(CT c, int n, int mode, int order, int &i0, int &i1, WT &w0, WT &w1,
                                          bool &outside)
{
    if constexpr (FASTC && ORDER == 1) {
        // mode == constant, order 1: the common case, no boundary map needed
        // (inside [0, n-1] both taps are valid; at c == n-1 the upper tap is skipped)
        outside = outside || c < (CT)0 || c > (CT)(n - 1);
        const CT cf = floor(c);
        w1 = (WT)(c - cf);
        w0 = (WT)1 - w1;
        i0 = (int)cf;
        i1 = w1 == (WT)0 ? i0 : i0 + 1;
        return;
    }
    if (mode == MI_MODE_CONSTANT && (c < (CT)0 || c > (CT)(n - 1))) outside = true;
    if (order == 0) {
        int j;
        if (mode == MI_MODE_CONSTANT) j = (int)floor((double)c + 0.5);
        else if (mode == MI_MODE_GRID_CONSTANT) j = bmap<int>((int)floor((double)c + 0.5), n, mode);
        else j = bmap<int>((int)floor(foldc((double)c, n, mode) + 0.5), n, mode);
        i0 = i1 = j;
        w0 = (WT)1;
        w1 = (WT)0;
        return;
    }
    const CT cf = floor(c);
    const CT fr = c - cf;               // exact in CT
    w1 = (WT)fr;
    w0 = (WT)(((CT)1 + cf) - c);     // (cf + 1) - c as the reference / SciPy form it
    if (mode == MI_MODE_WRAP) {
        const double f = wrapc((double)c, n);
        i0 = (int)floor(f);
        i1 = n <= 1 ? 0 : (int)floor(f + 1.0);      // a single sample: SciPy maps every coordinate to it
    } else {
        i0 = (int)cf;
        i1 = i0 + 1;
        if (mode != MI_MODE_CONSTANT) {
            i0 = bmap<int>(i0, n, mode);
            i1 = bmap<int>(i1, n, mode);
        }
    }
    if (fr == (CT)0) { i1 = i0; w1 = (WT)0; }   // integral coordinate: the upper tap is skipped
}

// One output voxel in two phases so that a thread can keep several voxels'
// gathers in flight: taps() computes indices / weights and issues the eight
// loads, finish() blends them.  Blending is three nested linear interpolations
// (x, then y, then z); a zero upper weight selects the lower sample outright,
// which is the reference's "second tap skipped at integral coordinates"
// (_interp_kernels.py:416) and keeps inf / nan of a skipped tap out.
template <typename T> struct Taps { T v[8]; T wz1, wy1, wx1; unsigned oobmask; bool outside; };

// two neighbouring samples with one gather (8 bytes of float32, 16 bytes of float64) / one sample
__device__ __forceinline__ void load_pair(const __amdgpu_buffer_rsrc_t in, unsigned off, float &a, float &b)
{
    const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(in, off, 0, 0);
    a = __uint_as_float(q.x); b = __uint_as_float(q.y);
}
__device__ __forceinline__ void load_pair(const __amdgpu_buffer_rsrc_t in, unsigned off, double &a, double &b)
{
    typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));
    const u32x4_ q = __builtin_amdgcn_raw_buffer_load_b128(in, off, 0, 0);
    a = __hiloint2double((int)q.y, (int)q.x); b = __hiloint2double((int)q.w, (int)q.z);
}
__device__ __forceinline__ void load_one(const __amdgpu_buffer_rsrc_t in, unsigned off, float &a)
{
    a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(in, off, 0, 0));
}
__device__ __forceinline__ void load_one(const __amdgpu_buffer_rsrc_t in, unsigned off, double &a)
{
    const u32x2 q = __builtin_amdgcn_raw_buffer_load_b64(in, off, 0, 0);
    a = __hiloint2double((int)q.y, (int)q.x);
}

template <typename T, typename CT, bool FASTC, int ORDER>
__device__ __forceinline__ void taps(const __amdgpu_buffer_rsrc_t in, const FastInterpParams &p, CT cz, CT cy, CT cx, Taps<T> &t)
{
    constexpr unsigned ES = sizeof(T);
    if constexpr (FASTC && ORDER == 1) {
        // constant mode, order 1: integer/fraction split once per axis, the range test on the integers, one base
        // index plus three strides.  r2: written branch-free (bitwise logic instead of && / ||: the short-circuit
        // form compiled into twenty exec-masked branches per four voxels) -- the kernel is VALU bound, not gather bound
        // (rocprofv3: 93 VALU instructions per voxel, VALU 78 % busy), so every instruction counts.
        const CT fz = floor(cz), fy = floor(cy), fx = floor(cx);
        const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
        t.wz1 = (T)(cz - fz); t.wy1 = (T)(cy - fy); t.wx1 = (T)(cx - fx);
        const bool zz = t.wz1 == (T)0, yz = t.wy1 == (T)0, xz = t.wx1 == (T)0;
        // 0 <= i < n - 1, or i == n - 1 with a zero fraction (unsigned compare folds the sign test in)
        const bool in_z = ((unsigned)z0 < (unsigned)(p.nz - 1)) | ((z0 == p.nz - 1) & zz);
        const bool in_y = ((unsigned)y0 < (unsigned)(p.ny - 1)) | ((y0 == p.ny - 1) & yz);
        const bool in_x = ((unsigned)x0 < (unsigned)(p.nx - 1)) | ((x0 == p.nx - 1) & xz);
        t.outside = !(in_z & in_y & in_x);
        t.oobmask = 0;
        // x0 / x1 sit next to each other in memory: one 8-byte gather per (z, y) pair.
        // At the last column (only reachable with wx1 == 0) the pair is shifted left by one.
        const bool lastcol = x0 >= p.nx - 1;
        const int xb = x0 - (lastcol ? 1 : 0);
        const unsigned base = t.outside ? 0u : (unsigned)((z0 * p.ny + y0) * p.nx + xb) * ES;
        const unsigned sz = (t.outside | zz) ? 0u : (unsigned)(p.ny * p.nx) * ES;
        const unsigned sy = (t.outside | yz) ? 0u : (unsigned)p.nx * ES;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            T a, b;
            load_pair(in, base + m * sy, a, b);
            t.v[2 * m] = lastcol ? b : a;
            t.v[2 * m + 1] = b;
        }
        if (sizeof(T) == 4 && p.two_d) {
            // an image has no z taps (wz1 == 0 selects the lower plane): two gathers per pixel instead of four
            // (float32 8192^2: rotate 275 -> 176 us; float64 measured 15 % SLOWER with the branch: left as it was)
#pragma unroll
            for (int m = 4; m < 8; m++) t.v[m] = t.v[m - 4];
        } else {
#pragma unroll
            for (int m = 2; m < 4; m++) {
                T a, b;
                load_pair(in, base + sz + (m & 1) * sy, a, b);
                t.v[2 * m] = lastcol ? b : a;
                t.v[2 * m + 1] = b;
            }
        }
        return;
    } else {
        int zi[2], yi[2], xi[2];
        T wz[2], wy[2], wx[2];
        bool outside = false;
        axis_taps<CT, FASTC, ORDER, T>(cz, p.nz, p.mode, p.order, zi[0], zi[1], wz[0], wz[1], outside);
        axis_taps<CT, FASTC, ORDER, T>(cy, p.ny, p.mode, p.order, yi[0], yi[1], wy[0], wy[1], outside);
        axis_taps<CT, FASTC, ORDER, T>(cx, p.nx, p.mode, p.order, xi[0], xi[1], wx[0], wx[1], outside);
        t.outside = outside;
        t.wz1 = wz[1]; t.wy1 = wy[1]; t.wx1 = wx[1];
        t.oobmask = 0;
        // taps that must not be read (index -1 = cval, or everything when the point is
        // outside in constant mode) load voxel 0 instead and are replaced afterwards
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int a = m >> 2, b = (m >> 1) & 1, c = m & 1;
            const bool oob = (zi[a] | yi[b] | xi[c]) < 0;
            if (oob) t.oobmask |= 1u << m;
            const unsigned off = (oob || outside) ? 0u : (unsigned)((zi[a] * p.ny + yi[b]) * p.nx + xi[c]) * ES;
            load_one(in, off, t.v[m]);
        }
    }
}

// (1 - w) lo + w hi, the upper sample skipped when its weight is zero.  Not lo + w (hi - lo): with an infinite `lo`
// that form gives inf - inf = NaN where SciPy's weighted sum gives the infinity.
// float32: the skip is the multiply itself -- v_mul_legacy_f32 returns 0 for 0 x anything (infinities and NaNs
// included), so a zero weight removes the upper sample without a compare and a select per blend (r3: 10 of the 72
// VALU instructions per voxel of the order-1 kernels were those), and (1 - 0) lo + 0 = lo.
__device__ __forceinline__ float mul_zero_wins(float a, float b)
{
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float lerp_skip(float lo, float hi, float w)
{
    return fmaf(1.f - w, lo, mul_zero_wins(w, hi));
}
__device__ __forceinline__ double lerp_skip(double lo, double hi, double w)
{
    return w == 0.0 ? lo : fma(w, hi, (1.0 - w) * lo);
}

template <typename T>
__device__ __forceinline__ T finish(const Taps<T> &t, T cval)
{
    T v[8];
#pragma unroll
    for (int m = 0; m < 8; m++) v[m] = ((t.oobmask >> m) & 1u) ? cval : t.v[m];
    const T x00 = lerp_skip(v[0], v[1], t.wx1), x01 = lerp_skip(v[2], v[3], t.wx1);
    const T x10 = lerp_skip(v[4], v[5], t.wx1), x11 = lerp_skip(v[6], v[7], t.wx1);
    const T y0 = lerp_skip(x00, x01, t.wy1), y1 = lerp_skip(x10, x11, t.wy1);
    const T r = lerp_skip(y0, y1, t.wz1);
    return t.outside ? cval : r;
}


// ---------------------------------------------------------------------------
// r3: constant mode, order 1, 3-D float32 -- the BASELINE D / D' kernels.
//
// rocprofv3 (profiles/r3_affine_counters.txt) shows both kernels bound by the texture addresser, i.e. by the NUMBER of
// vector-memory instructions a wave issues (a wave instruction occupies the TA for >= 16 cycles whatever its width),
// with the VALU second (affine: 28 float64 operations per voxel).  What these kernels change against
// map_coords3d_fast / affine3d_fast above, results unchanged:
//   * stores (and the coordinate loads of map_coordinates) are 16 bytes per lane: a wave owns four output rows of 64
//     voxels; the interpolated values cross the wave through a 1 KiB LDS tile (written [row][lane], read back as
//     float4 by lane -> (row = lane / 16, 4 x (lane % 16))), one buffer_store_dwordx4 instead of four dword stores,
//     three dwordx4 coordinate loads instead of twelve dword loads: 20 instead of 32 vector-memory instructions per
//     four voxels for map_coordinates, 17 instead of 20 for affine_transform;
//   * affine: the row prefix (m0 z + m1 y) of the coordinate -- the same for the 64 lanes of a wave -- is computed once
//     per workgroup (48 lanes, LDS table) in the oracle's summation order, so a voxel adds two terms per axis; the
//     integer / fraction split is v_cvt_i32_f64 + v_fract_f64 (exact for the non-negative coordinates that can be
//     inside; negative ones are outside by their sign bit): 15 instead of 28 float64 operations per voxel.
// ---------------------------------------------------------------------------
typedef float f32x4n __attribute__((ext_vector_type(4)));      // what the non-temporal builtins take
struct C1Split { int i0; float w1; bool in; };

// c -> (trunc(c), fraction as float, 0 <= c <= n - 1) for an axis of n samples.  For c >= 0: trunc = floor and
// v_fract_f64 = c - floor(c) exactly.  `in` is the closed interval on the coordinate itself, which is what the tap
// logic of the reference amounts to in `constant` mode (lower tap floor(c) inside, and the upper tap either inside or
// unused because the fraction is zero): c in [0, n - 1) has both taps inside, c == n - 1 has fraction 0, anything else
// is cval -- two float64 compares instead of six mixed ones per axis.  -0.0 is inside, a NaN coordinate stays "inside"
// and blends to NaN as before.
__device__ __forceinline__ C1Split c1_split(double c, int n)
{
    C1Split r;
    r.in = !(c < 0.0) & !(c > (double)(n - 1));
    r.i0 = __double2int_rz(c);
    r.w1 = (float)__builtin_amdgcn_fract(c);
    return r;
}
__device__ __forceinline__ C1Split c1_split(float c, int n)
{
    C1Split r;
    const float f = floorf(c);
    r.in = !(c < 0.f) & !(c > (float)(n - 1));
    r.i0 = (int)f;
    r.w1 = c - f;                      // exact in float
    return r;
}

// the eight taps of one voxel as four 8-byte gathers (same addressing as taps<..., FASTC = true, 1>)
__device__ __forceinline__ void c1_gather(const __amdgpu_buffer_rsrc_t in, const FastInterpParams &p, const C1Split &sz,
                                          const C1Split &sy, const C1Split &sx, Taps<float> &t)
{
    t.wz1 = sz.w1; t.wy1 = sy.w1; t.wx1 = sx.w1;
    const bool in_z = sz.in;
    const bool in_y = sy.in;
    const bool in_x = sx.in;
    t.outside = !(in_z & in_y & in_x);
    t.oobmask = 0;
    // An upper tap is skipped when the FLOAT weight is zero (a double fraction below the float range blends to the
    // same value either way) -- by the blend itself (lerp_skip), so the upper row and plane are read unconditionally:
    // a row or plane past the end of the volume fails the descriptor's range check and reads as zero.
    // The same holds along x: at the last column the pair (x, x + 1) takes the first sample of the next row (or zero
    // past the end of the volume) as its upper half, with weight zero.
    const unsigned base = t.outside ? 0u : (unsigned)((sz.i0 * p.ny + sy.i0) * p.nx + sx.i0) * 4u;
    const unsigned stz = (unsigned)(p.ny * p.nx) * 4u;
    const unsigned sty = (unsigned)p.nx * 4u;
#pragma unroll
    for (int m = 0; m < 4; m++) load_pair(in, base + (m >> 1) * stz + (m & 1) * sty, t.v[2 * m], t.v[2 * m + 1]);
}

// Which four voxels a lane owns (k = 0..3) and which four "rows" a wave's tile holds:
//   ZMAJ = false: one plane, rows ybase + 4 k + ty                 (workgroup = 64 x by 16 y by 1 z)
//   ZMAJ = true : planes zbase + k, row ybase + ty                  (workgroup = 64 x by 4 y by 4 z)
// The second form reads fewer distinct cache lines per workgroup when the warp is close to the identity along z
// (output planes z and z + 1 share an input plane, rows y and y + 1 share an input row: 5 x 5 row segments per
// workgroup instead of ~17 x 2), which is what the per-CU L1 (32 KiB for eight resident workgroups) needs.
template <bool ZMAJ>
struct C1Map {
    int zb, yb;
    __device__ __forceinline__ C1Map(const dim3 &b) : zb(ZMAJ ? (int)b.z * 4 : (int)b.z), yb(ZMAJ ? (int)b.y * 4 : (int)b.y * 16) {}
    __device__ __forceinline__ int z(int k) const { return ZMAJ ? zb + k : zb; }
    __device__ __forceinline__ int y(int k, int ty) const { return ZMAJ ? yb + ty : yb + 4 * k + ty; }
    __device__ __forceinline__ bool full(const FastInterpParams &p, int x0w) const
    {
        return x0w + 64 <= p.ox && (ZMAJ ? (yb + 4 <= p.oy && zb + 4 <= p.oz) : (yb + 16 <= p.oy));
    }
};

// values r[k] (voxel k of this lane, x = x0w + lane) -> one 16-byte store per lane through the wave's LDS tile
template <bool ZMAJ>
__device__ __forceinline__ void c1_store_rows(float *__restrict__ out, const FastInterpParams &p, float *tile, int lane,
                                              const C1Map<ZMAJ> &mp, int ty, int x0w, const float (&r)[4], bool wide)
{
    if (wide) {
#pragma unroll
        for (int k = 0; k < 4; k++) tile[k * 64 + lane] = r[k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int i = lane >> 4, q = lane & 15;
        const f32x4n v = *reinterpret_cast<const f32x4n *>(tile + i * 64 + 4 * q);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4n *>(out + ((size_t)mp.z(i) * p.oy + mp.y(i, ty)) * p.ox + x0w + 4 * q));
    } else {
        const int x = x0w + lane;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = mp.y(k, ty), z = mp.z(k);
            if (x < p.ox && y < p.oy && z < p.oz) __builtin_nontemporal_store(r[k], out + ((size_t)z * p.oy + y) * p.ox + x);
        }
    }
}

template <bool WIDE, bool ZMAJ>
__global__ void __launch_bounds__(256)
affine3d_c1_kernel(const float *__restrict__ in, float *__restrict__ out, const FastInterpParams p)
{
    __shared__ double ptab[16][3];
    __shared__ __attribute__((aligned(16))) float tiles[4][4 * 64];
    const int lane = threadIdx.x, ty = threadIdx.y;
    const C1Map<ZMAJ> mp(blockIdx);
    const int x0w = blockIdx.x * 64;
    const int tid = ty * 64 + lane;
    if (tid < 48) {
        // row prefix of (k, ty) in the oracle's order: (m0 * z) + m1 * y
        const int rr = tid / 3, a = tid - 3 * rr;
        ptab[rr][a] = p.m[4 * a] * (double)mp.z(rr >> 2) + p.m[4 * a + 1] * (double)mp.y(rr >> 2, rr & 3);
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, p.nz * p.ny * p.nx * 4, 0x00020000);
    const double dx = (double)(x0w + lane);
    const double xz_ = p.m[2] * dx, xy_ = p.m[6] * dx, xx_ = p.m[10] * dx;
    Taps<float> t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int rr = 4 * k + ty;
        const double cz = (ptab[rr][0] + xz_) + p.m[3];
        const double cy = (ptab[rr][1] + xy_) + p.m[7];
        const double cx = (ptab[rr][2] + xx_) + p.m[11];
        c1_gather(rin, p, c1_split(cz, p.nz), c1_split(cy, p.ny), c1_split(cx, p.nx), t[k]);
    }
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = finish<float>(t[k], (float)p.cval);
    c1_store_rows<ZMAJ>(out, p, tiles[ty], lane, mp, ty, x0w, r, WIDE && mp.full(p, x0w));      // block-uniform
}

template <bool WIDE, bool ZMAJ>
__global__ void __launch_bounds__(256)
map_coords3d_c1_kernel(const float *__restrict__ in, const float *__restrict__ coords, float *__restrict__ out,
                       const FastInterpParams p)
{
    __shared__ __attribute__((aligned(16))) float tiles[4][3 * 4 * 64];
    const int lane = threadIdx.x, ty = threadIdx.y;
    const C1Map<ZMAJ> mp(blockIdx);
    const int x0w = blockIdx.x * 64;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, p.nz * p.ny * p.nx * 4, 0x00020000);
    const size_t nout = (size_t)p.oz * p.oy * p.ox;
    const bool wide = WIDE && mp.full(p, x0w);                      // block-uniform
    float *tile = tiles[ty];
    float c[4][3];
    if (wide) {
        // coordinates: one 16-byte load per lane, "row" and axis (lane -> row lane / 16, x = 4 (lane % 16)), handed to
        // the lane that owns the voxel through the wave's LDS tile
        const int i = lane >> 4, q = lane & 15;
        const size_t o = ((size_t)mp.z(i) * p.oy + mp.y(i, ty)) * p.ox + x0w + 4 * q;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n *>(coords + a * nout + o));
            *reinterpret_cast<f32x4n *>(tile + (a * 4 + i) * 64 + 4 * q) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int a = 0; a < 3; a++) c[k][a] = tile[(a * 4 + k) * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the tile is reused for the results below
    } else {
        const int x = min(x0w + lane, p.ox - 1);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const size_t o = ((size_t)min(mp.z(k), p.oz - 1) * p.oy + min(mp.y(k, ty), p.oy - 1)) * p.ox + x;
#pragma unroll
            for (int a = 0; a < 3; a++) c[k][a] = __builtin_nontemporal_load(coords + a * nout + o);
        }
    }
    Taps<float> t[4];
#pragma unroll
    for (int k = 0; k < 4; k++) c1_gather(rin, p, c1_split(c[k][0], p.nz), c1_split(c[k][1], p.ny), c1_split(c[k][2], p.nx), t[k]);
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) r[k] = finish<float>(t[k], (float)p.cval);
    c1_store_rows<ZMAJ>(out, p, tile, lane, mp, ty, x0w, r, wide);
}

// ---------------------------------------------------------------------------
// r3: affine_transform, order 1, constant mode, float32 volumes -- gathers out of LDS.
//
// Why: the gather kernels above are bound by the L1's tag pipeline, not by HBM and not by the VALU: rocprofv3 counts
// 99 cache accesses per 64 voxels (TCP_TOTAL_CACHE_ACCESSES; ~25 per 8-byte gather instruction, the L1 serves a wave
// four lanes at a time) = 386 us of the 414 us config D' takes, while a ds_read2_b32 serves the same pair in 4 cycles
// per wave.  An affine map sends an output tile to a parallelepiped whose bounding box is known from |M| alone, so:
//   * a workgroup (8 waves) owns 64 x 8 x 8 output voxels; the bounding box of their taps -- BZ x BY x BX input
//     samples, dimensions computed by the host from the matrix, origin from the tile's eight corners -- is staged
//     global -> LDS with `buffer_load_dwordx4 ... lds` (16 bytes per lane, no VGPRs, LDS layout = the box row-major, so
//     a wave's 64 chunks are consecutive LDS addresses); rows / planes beyond the volume are zero-filled by the
//     descriptor's range check and never read;
//   * coordinates, in-range tests and blending are those of affine3d_c1_kernel (same c1_split, same finish): the
//     results are bit-identical to it;
//   * the eight taps of a voxel are four ds_read2_b32 (x and x + 1 in one instruction);
//   * two workgroups per CU (<= 64 KiB of box each): one computes while the other's box is in flight.
// Transforms whose box exceeds that budget (large rotations about z / y with this tile, down-scaling by > 2) keep the
// L1 gather kernel -- the choice is made on the host from the matrix.
// ---------------------------------------------------------------------------
constexpr int kLdsVox = 4096;                               // output voxels per tile: TX x TY x (4096 / (TX TY)): 64 x 8 x 8, 32 x 16 x 8, 16 x 32 x 8,
                                                            // and (r5) the cube 16 x 16 x 16, whose bounding box under a rotation about a
                                                            // general axis is the smallest of all (37 KiB at 10 degrees about (1, 1, 1)
                                                            // against 46 - 77 KiB: profiles/r5_affine_general.txt)
constexpr int kLdsBoxBytesMax = 64 * 1024;                  // box budget per workgroup: two workgroups per CU.  (r3 / r4: 36 KiB -- with the
                                                            // 64-wide tile a bigger box lost to the L1 gathers on config D'; with the cube
                                                            // it wins up to ~25 degrees, where the gathers have fallen to 0.2 of the roofline.)
Knob g_affine_box_kib{0};                                   // test hook: box budget in KiB (0 = kLdsBoxBytesMax)

struct LdsAffineParams {
    FastInterpParams f;
    int bz, by, bx;          // box dimensions (bx a multiple of 4)
    int nchunks;             // bz * by * bx / 4 sixteen-byte chunks
    // staging loop without divisions: chunk -> (row, 16-byte chunk of the row), row -> (plane, row of the plane) by
    // multiply-high with ceil(2^32 / d) (exact for the few thousand chunks of a box); the per-round increments
    unsigned cpr_magic, by_magic;
    int drow, dc4, drz, dry;
    // box origin in closed form: the minimum of an affine coordinate over the tile is its value at the tile's first
    // voxel plus cmin[a] = sum_j min(0, m[a][j] (T_j - 1))
    double cmin[3];
    int dbg;                 // tuning ablations (0 in production): 1 no box DMA, 2 no interpolation (stores only), 4 no stores
};

constexpr int kLdsRoundsMax = 16;                           // staging rounds of 512 chunks (8 KiB) a box can take

// the same with a scalar byte offset added to every lane's
__device__ __forceinline__ void dma_16s(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_base)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %3 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_base)
        : "memory");
}

__device__ __forceinline__ void dma_16(const __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_base)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rsrc), "s"(lds_base)
        : "memory");
}

template <int TX, int TY>
__global__ void __launch_bounds__(512)
affine3d_lds_kernel(const float *__restrict__ in, float *__restrict__ out, const LdsAffineParams q)
{
    constexpr int RW = 64 / TX;              // output rows per wave
    constexpr int WY = TY / RW;              // waves along y; the other 8 / WY along z, eight planes each
    constexpr int TZ = kLdsVox / (TX * TY);
    static_assert(TY % RW == 0 && 8 % WY == 0 && TZ == 8 * (8 / WY), "tile = 8 waves x 64 lanes x 8 planes");
    extern __shared__ __attribute__((aligned(16))) char smem_lds[];
    const FastInterpParams &p = q.f;
    float *box = reinterpret_cast<float *>(smem_lds);
    const unsigned box_bytes = ((unsigned)q.nchunks * 16u + 8191u) & ~8191u;          // whole rounds of 512 chunks
    double (*ptab)[3] = reinterpret_cast<double (*)[3]>(smem_lds + box_bytes);        // [TZ * TY][3]
    float *tiles = reinterpret_cast<float *>(smem_lds + box_bytes + TZ * TY * 3 * sizeof(double));   // [8 waves][256]

    int (*org)[4] = reinterpret_cast<int (*)[4]>(tiles + 8 * 256);                   // [2][4]: box origin of this / the next tile

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lx = lane & (TX - 1), yy = lane / TX;
    const int x0w = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, p.nz * p.ny * p.nx * 4, 0x00020000);
    const int cpr = q.bx >> 2;
    const int rounds = (q.nchunks + 511) >> 9;
    // chunk tid of the box -> (plane, row, chunk of the row): the same for every tile, and so is the byte offset of the
    // chunks tid + 512 j from the box origin (rel[j]; chunks past the box fail the range check and write zeros)
    const int row0 = (int)__umulhi((unsigned)tid, q.cpr_magic), c40 = tid - row0 * cpr;
    const int rz0 = (int)__umulhi((unsigned)row0, q.by_magic), ry0 = row0 - rz0 * q.by;
    unsigned rel[kLdsRoundsMax];
    {
        const unsigned row_b = (unsigned)p.nx * 4u, plane_b = (unsigned)p.ny * row_b;
#pragma unroll
        for (int j = 0; j < kLdsRoundsMax; j++) {
            const unsigned ch = (unsigned)tid + ((unsigned)j << 9);
            const unsigned row = __umulhi(ch, q.cpr_magic), c4 = ch - row * (unsigned)cpr;
            const unsigned rz = __umulhi(row, q.by_magic), ry = row - rz * (unsigned)q.by;
            rel[j] = ch < (unsigned)q.nchunks ? rz * plane_b + ry * row_b + c4 * 16u : 0x80000000u;
        }
    }
    const int wy = wave % WY, wz = wave / WY;
    const int yrow = RW * wy + yy;
    const double dx = (double)(x0w + lx);
    const double xz_ = p.m[2] * dx, xy_ = p.m[6] * dx, xx_ = p.m[10] * dx;
    const int plane_f = q.by * q.bx;
    float *tile = tiles + wave * 256;

    // Box origin of the tile at z tile index tz_, axis a: floor of the smallest coordinate over the tile (the map is
    // affine: its minimum is the value at the first voxel plus a constant of the matrix), a hair below it because this
    // sum is not the per-voxel sum to the last bit; clamped into the volume; x aligned down to a multiple of four samples
    // (16-byte chunks).
    auto origin = [&](int tz_, int a) {
        const double lo = ((p.m[4 * a] * (double)(tz_ * TZ) + p.m[4 * a + 1] * (double)y0) + p.m[4 * a + 2] * (double)x0w) +
                          (p.m[4 * a + 3] + q.cmin[a]);
        const int n = a == 0 ? p.nz : (a == 1 ? p.ny : p.nx);
        double f = floor(lo - 1e-6 * (1.0 + fabs(lo)));
        f = f < 0.0 ? 0.0 : (f > (double)(n - 1) ? (double)(n - 1) : f);
        return a == 2 ? ((int)f & ~3) : (int)f;
    };
    // the first tile's: every lane computes the same three numbers (no table, no barrier before the first DMA)
    int b0[3] = {__builtin_amdgcn_readfirstlane(origin((int)blockIdx.z, 0)), __builtin_amdgcn_readfirstlane(origin((int)blockIdx.z, 1)),
                 __builtin_amdgcn_readfirstlane(origin((int)blockIdx.z, 2))};

    // A workgroup owns the tiles (blockIdx.x, blockIdx.y, blockIdx.z + k gridDim.z) -- one tile with the default grid.
    // What a tile costs besides its voxels was 126 us of config D's 392 (profiles/r3_affine_ablation.txt): two integer
    // divisions and the address arithmetic of five staging rounds per thread, the origin by three lanes and a barrier.
    // Here the divisions are multiplications, the staging rounds are one DMA each with a precomputed offset, and a
    // workgroup that walks several tiles has three lanes compute the next origin while the current box is in flight.
    int slot = 0;
#pragma unroll 1
    for (int tz = blockIdx.z; tz * TZ < p.oz; tz += gridDim.z, slot ^= 1) {
    const int z0 = tz * TZ;
    const bool more = (tz + (int)gridDim.z) * TZ < p.oz;

    // ---- stage the box.  Away from the upper faces of the volume every chunk of the box exists: the rounds are one DMA
    // each (offset from the box origin in the VGPR, the origin itself in the scalar offset).  A box that sticks out of
    // the volume checks its chunks one by one (rows and planes past the end read as zeros).
    if (b0[0] + q.bz <= p.nz && b0[1] + q.by <= p.ny && b0[2] + q.bx <= p.nx) {
        const unsigned base = (unsigned)((b0[0] * p.ny + b0[1]) * p.nx + b0[2]) * 4u;
#pragma unroll
        for (int j = 0; j < kLdsRoundsMax; j++)
            if (j < rounds && !(q.dbg & 1)) dma_16s(rin, rel[j], base, (unsigned)((wave << 6) + (j << 9)) * 16u);
    } else {
        int c4 = c40, ry = ry0, rz = rz0;
        for (int j = 0; j < rounds; j++) {
            const int sz_ = b0[0] + rz, sy_ = b0[1] + ry, sx_ = b0[2] + 4 * c4;
            const bool ok = tid + (j << 9) < q.nchunks && sz_ < p.nz && sy_ < p.ny && sx_ < p.nx;
            const unsigned voff = ok ? (unsigned)((sz_ * p.ny + sy_) * p.nx + sx_) * 4u : 0x80000000u;
            if (!(q.dbg & 1)) dma_16(rin, voff, (unsigned)((wave << 6) + (j << 9)) * 16u);
            c4 += q.dc4; ry += q.dry; rz += q.drz;
            if (c4 >= cpr) { c4 -= cpr; ry++; }
            if (ry >= q.by) { ry -= q.by; rz++; }
            if (ry >= q.by) { ry -= q.by; rz++; }
        }
    }
    for (int e = tid; e < TZ * TY * 3; e += 512) {
        const int rr = e / 3, a = e - 3 * rr;                      // rr = TY k + row  <->  plane z0 + k, row y0 + row
        ptab[rr][a] = p.m[4 * a] * (double)(z0 + rr / TY) + p.m[4 * a + 1] * (double)(y0 + rr % TY);
    }
    if (more && tid < 3) org[slot][tid] = origin(tz + (int)gridDim.z, tid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- interpolate: lane = (x, row of the wave), k = plane; two batches of four planes
    const int lorg = (b0[0] * q.by + b0[1]) * q.bx + b0[2];      // box index of sample (z, y, x) = (z by + y) bx + x - lorg
    const bool wide = x0w + TX <= p.ox && y0 + TY <= p.oy && z0 + TZ <= p.oz;      // block-uniform
#pragma unroll 1
    for (int bt = 0; bt < 2; bt++) {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (!(q.dbg & 2))
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
            const int k = 8 * wz + 4 * bt + kk;
            const int rr = TY * k + yrow;
            const C1Split sz = c1_split((ptab[rr][0] + xz_) + p.m[3], p.nz);
            const C1Split sy = c1_split((ptab[rr][1] + xy_) + p.m[7], p.ny);
            const C1Split sx = c1_split((ptab[rr][2] + xx_) + p.m[11], p.nx);
            Taps<float> t;
            t.wz1 = sz.w1; t.wy1 = sy.w1; t.wx1 = sx.w1;
            const bool in_z = sz.in;
            const bool in_y = sy.in;
            const bool in_x = sx.in;
            t.outside = !(in_z & in_y & in_x);
            t.oobmask = 0;
            // The upper taps are read unconditionally: a zero weight removes them in the blend (lerp_skip), whatever they
            // read then -- one sample past the row / the plane (zero-filled by the DMA beyond the volume), the tables behind
            // the box, or nothing at all (an LDS read beyond the workgroup's allocation returns zero).
            const int li = t.outside ? 0 : (sz.i0 * q.by + sy.i0) * q.bx + sx.i0 - lorg;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float *src = box + li + (m >> 1) * plane_f + (m & 1) * q.bx;
                t.v[2 * m] = src[0];
                t.v[2 * m + 1] = src[1];
            }
            r[kk] = finish<float>(t, (float)p.cval);
        }
        if (q.dbg & 4) continue;
        if (wide) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) tile[kk * 64 + lane] = r[kk];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int i = lane >> 4, c = lane & 15;                 // plane of the batch, 16-byte chunk of the wave's 64 voxels
            const f32x4n v = *reinterpret_cast<const f32x4n *>(tile + i * 64 + 4 * c);
            const int orow = y0 + RW * wy + (4 * c) / TX, ox4 = x0w + ((4 * c) & (TX - 1));
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4n *>(out + ((size_t)(z0 + 8 * wz + 4 * bt + i) * p.oy + orow) * p.ox + ox4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            const int x = x0w + lx, y = y0 + yrow;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int z = z0 + 8 * wz + 4 * bt + kk;
                if (x < p.ox && y < p.oy && z < p.oz) __builtin_nontemporal_store(r[kk], out + ((size_t)z * p.oy + y) * p.ox + x);
            }
        }
    }
    if (!more) break;
    __syncthreads();            // the box and the prefix table are rewritten for the next tile
    b0[0] = __builtin_amdgcn_readfirstlane(org[slot][0]);
    b0[1] = __builtin_amdgcn_readfirstlane(org[slot][1]);
    b0[2] = __builtin_amdgcn_readfirstlane(org[slot][2]);
    }
}

extern "C" int mi_debug_set_affine_box_kib(int k) { g_affine_box_kib = k; return MI_OK; }
Knob g_affine_dbg{0};     // tuning ablations of affine3d_lds_kernel, see LdsAffineParams::dbg
Knob g_affine_gz{0};      // test hook: workgroups along z of affine3d_lds_kernel (0 = auto; the number of z tiles = one tile per workgroup)

// box dimensions of a TZ x TY x TX output tile under the matrix (upper bound from |M|); returns the number of floats,
// or 0 when the box does not fit the LDS budget
static long long lds_affine_plan(const FastInterpParams &p, int tx, int ty, LdsAffineParams *q)
{
    const int T[3] = {kLdsVox / (tx * ty) - 1, ty - 1, tx - 1};
    int dim[3];
    for (int a = 0; a < 3; a++) {
        double ext = 0.0;
        for (int j = 0; j < 3; j++) ext += fabs(p.m[4 * a + j]) * T[j];
        if (!(ext < 4096.0)) return 0;
        // samples floor(min - hair) .. floor(max) + 1 with max - min <= ext: at most floor(ext + hair) + 3 of them (the
        // hair: what the origin is moved below the corner minimum, 1e-6 (1 + |c|) <= 2e-3)
        dim[a] = (int)floor(ext * (1.0 + 1e-6) + 2e-3) + 3;
    }
    dim[2] = (dim[2] + 3 + 3) & ~3;                 // origin aligned down by up to 3, length a multiple of 4
    const int n[3] = {p.nz, p.ny, p.nx};
    for (int a = 0; a < 2; a++) if (dim[a] > n[a]) dim[a] = n[a];
    if (dim[2] > ((n[2] + 3) & ~3) + 4) dim[2] = ((n[2] + 3) & ~3) + 4;
    const long long floats = (long long)dim[0] * dim[1] * dim[2];
    const long long budget = (g_affine_box_kib % 1000) > 0 ? (long long)(g_affine_box_kib % 1000) * 1024 : (long long)kLdsBoxBytesMax;
    if (floats * 4 > budget || (floats / 4 + 511) / 512 > kLdsRoundsMax) return 0;
    q->f = p;
    q->bz = dim[0]; q->by = dim[1]; q->bx = dim[2];
    q->nchunks = (int)(floats / 4);
    {
        const unsigned cpr = (unsigned)dim[2] / 4u, by = (unsigned)dim[1];
        q->cpr_magic = (unsigned)((((unsigned long long)1 << 32) + cpr - 1) / cpr);
        q->by_magic = (unsigned)((((unsigned long long)1 << 32) + by - 1) / by);
        q->drow = (int)(512u / cpr); q->dc4 = (int)(512u - (unsigned)q->drow * cpr);
        q->drz = q->drow / (int)by; q->dry = q->drow - q->drz * (int)by;
    }
    for (int a = 0; a < 3; a++) {
        double c = 0.0;
        for (int j = 0; j < 3; j++) { const double e = p.m[4 * a + j] * T[j]; if (e < 0.0) c += e; }
        q->cmin[a] = c;
    }
    q->dbg = g_affine_dbg;
    return floats;
}

template <int TX, int TY>
static int launch_affine_lds(const float *in, float *out, const LdsAffineParams &q, hipStream_t s)
{
    constexpr int kLdsTZ = kLdsVox / (TX * TY);
    const FastInterpParams &p = q.f;
    // Workgroups along z: one per tile by default.  The kernel can walk several tiles of a column (gz < ntz), which
    // measured slower on config D' (346 us with one tile each; 360-405 us with 3 ... 12 tiles each: the tiles of a
    // workgroup run strictly one after the other, a fresh workgroup overlaps with its neighbours on the CU).
    const unsigned ntz = (unsigned)((p.oz + kLdsTZ - 1) / kLdsTZ);
    unsigned gz = g_affine_gz > 0 ? (unsigned)g_affine_gz : ntz;
    if (gz > ntz) gz = ntz;
    if (gz < 1) gz = 1;
    const dim3 gl((unsigned)((p.ox + TX - 1) / TX), (unsigned)((p.oy + TY - 1) / TY), gz);
    if (gl.y > 65535 || gl.z > 65535) return MI_ERR_UNSUPPORTED;
    const size_t lds = (((size_t)q.nchunks * 16 + 8191) & ~(size_t)8191) + kLdsTZ * TY * 3 * sizeof(double) + 8 * 256 * sizeof(float) + 2 * 4 * sizeof(int);
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)affine3d_lds_kernel<TX, TY>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_done = true;
    }
    note_kernel("mi::affine3d_lds_kernel<%d,%d> grid=%ux%ux%u (order-1 affine, tile %d x %d x %d, %d x %d x %d box staged per tile)", TX, TY, gl.x, gl.y, gl.z,
                TX, TY, kLdsTZ, q.bz, q.by, q.bx);
    hipLaunchKernelGGL((affine3d_lds_kernel<TX, TY>), gl, dim3(512), lds, s, in, out, q);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------
// r3: map_coordinates, order 1, constant mode, float32 -- the same LDS gathers for ARBITRARY coordinates (config D).
// The box of a 32 x 16 x 8 output tile is not known in advance: every workgroup reduces the integer parts of its own
// coordinates (per lane, DPP / shuffle per wave, six LDS atomics per wave) to the bounding box of the taps it will read,
// stages that box if it fits the LDS budget (smooth warps: it does) and otherwise gathers through the L1 as
// map_coords3d_c1_kernel does -- decided per workgroup, so a warp that is smooth in most places keeps the fast path
// there.  Coordinates are loaded 16 bytes per lane and handed to their lanes through LDS (the region the box occupies
// afterwards), results leave 16 bytes per lane as well.  Same splits, same in-range tests, same blend: bit-identical
// to the other order-1 kernels.
// ---------------------------------------------------------------------------
constexpr int kMapBoxFloats = 9216;          // 36 KiB

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(v, m, 64); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
    return v;
}

__global__ void __launch_bounds__(512)
map_coords3d_lds_kernel(const float *__restrict__ in, const float *__restrict__ coords, float *__restrict__ out,
                        const FastInterpParams p)
{
    constexpr int TX = 32, RW = 2, TY = 16, TZ = 8;
    __shared__ __attribute__((aligned(16))) float box[kMapBoxFloats + 2048];      // + slack: whole rounds of 512 chunks
    __shared__ int ctl[8];                                                         // lo[3], hi[3]
    __shared__ __attribute__((aligned(16))) float tiles[8][256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lx = lane & (TX - 1), yy = lane / TX;
    const int x0w = blockIdx.x * TX, y0 = blockIdx.y * TY, z0 = blockIdx.z * TZ;
    const int yrow = RW * wave + yy;
    const size_t nout = (size_t)p.oz * p.oy * p.ox;
    const bool wide = x0w + TX <= p.ox && y0 + TY <= p.oy && z0 + TZ <= p.oz;      // block-uniform
    if (tid < 3) ctl[tid] = 0x7fffffff;
    else if (tid < 6) ctl[tid] = -1;

    // ---- phase 1: the coordinates of this lane's eight voxels (planes z0 .. z0 + 7 at (y0 + yrow, x0w + lx))
    float c[8][3];
    if (wide) {
        float *stage = box + wave * (3 * 4 * 64);             // 3 KiB per wave, inside the future box
        const int i = lane >> 4, cc = lane & 15;
        const int srow = y0 + RW * wave + (4 * cc) / TX, sx = x0w + ((4 * cc) & (TX - 1));
#pragma unroll
        for (int bt = 0; bt < 2; bt++) {
            const size_t o = ((size_t)(z0 + 4 * bt + i) * p.oy + srow) * p.ox + sx;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const f32x4n v = __builtin_nontemporal_load(reinterpret_cast<const f32x4n *>(coords + a * nout + o));
                *reinterpret_cast<f32x4n *>(stage + (a * 4 + i) * 64 + 4 * cc) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 4; kk++)
#pragma unroll
                for (int a = 0; a < 3; a++) c[4 * bt + kk][a] = stage[(a * 4 + kk) * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
        const int x = min(x0w + lx, p.ox - 1), y = min(y0 + yrow, p.oy - 1);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const size_t o = ((size_t)min(z0 + k, p.oz - 1) * p.oy + y) * p.ox + x;
#pragma unroll
            for (int a = 0; a < 3; a++) c[k][a] = __builtin_nontemporal_load(coords + a * nout + o);
        }
    }
    // ---- bounding box of the taps of the voxels that are inside the volume
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const C1Split sz = c1_split(c[k][0], p.nz), sy = c1_split(c[k][1], p.ny), sx = c1_split(c[k][2], p.nx);
        const bool in_z = sz.in;
        const bool in_y = sy.in;
        const bool in_x = sx.in;
        if (in_z & in_y & in_x) {
            lo[0] = min(lo[0], sz.i0); hi[0] = max(hi[0], sz.i0);
            lo[1] = min(lo[1], sy.i0); hi[1] = max(hi[1], sy.i0);
            lo[2] = min(lo[2], sx.i0); hi[2] = max(hi[2], sx.i0);
        }
    }
    __syncthreads();                      // ctl initialised; (coordinates staging is wave-private)
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const int l = wave_min_i32(lo[a]), h = wave_max_i32(hi[a]);
        if (lane == 0) { atomicMin(&ctl[a], l); atomicMax(&ctl[3 + a], h); }
    }
    __syncthreads();                      // every wave has read its coordinates back: the box may overwrite the staging area
    int b0[3], bd[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        b0[a] = __builtin_amdgcn_readfirstlane(ctl[a]);
        bd[a] = __builtin_amdgcn_readfirstlane(ctl[3 + a]);
    }
    const bool any_inside = bd[0] >= 0;
    b0[2] &= ~3;
#pragma unroll
    for (int a = 0; a < 3; a++) bd[a] = any_inside ? bd[a] + 2 - b0[a] : 0;            // samples lo .. hi + 1
    bd[2] = (bd[2] + 3) & ~3;
    const long long box_floats = (long long)bd[0] * bd[1] * bd[2];
    const bool use_box = any_inside && box_floats <= kMapBoxFloats;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, p.nz * p.ny * p.nx * 4, 0x00020000);
    if (use_box) {
        const int nchunks = (int)(box_floats >> 2);
        const int cpr = bd[2] >> 2;
        const int rounds = (nchunks + 511) >> 9;
        int row = tid / cpr, c4 = tid - row * cpr;
        int rz = row / bd[1], ry = row - rz * bd[1];
        const int drow = 512 / cpr, dc4 = 512 - drow * cpr;
        const int drz = drow / bd[1], dry = drow - drz * bd[1];
        for (int j = 0; j < rounds; j++) {
            const int sz_ = b0[0] + rz, sy_ = b0[1] + ry, sx_ = b0[2] + 4 * c4;
            const bool ok = tid + (j << 9) < nchunks && sz_ < p.nz && sy_ < p.ny && sx_ < p.nx;
            const unsigned voff = ok ? (unsigned)((sz_ * p.ny + sy_) * p.nx + sx_) * 4u : 0x80000000u;
            dma_16(rin, voff, (unsigned)(size_t)box + (unsigned)((wave << 6) + (j << 9)) * 16u);
            c4 += dc4; ry += dry; rz += drz;
            if (c4 >= cpr) { c4 -= cpr; ry++; }
            if (ry >= bd[1]) { ry -= bd[1]; rz++; }
            if (ry >= bd[1]) { ry -= bd[1]; rz++; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // ---- phase 2: interpolate, two batches of four planes
    const int plane_f = bd[1] * bd[2];
    float *tile = tiles[wave];
#pragma unroll 1
    for (int bt = 0; bt < 2; bt++) {
        float r[4];
        // this batch's coordinates, selected by value (a run-time index into c[] would put the array into scratch memory)
        float cb[4][3];
#pragma unroll
        for (int kk = 0; kk < 4; kk++)
#pragma unroll
            for (int a = 0; a < 3; a++) cb[kk][a] = bt ? c[4 + kk][a] : c[kk][a];
        if (use_box) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const C1Split sz = c1_split(cb[kk][0], p.nz), sy = c1_split(cb[kk][1], p.ny), sx = c1_split(cb[kk][2], p.nx);
                Taps<float> t;
                t.wz1 = sz.w1; t.wy1 = sy.w1; t.wx1 = sx.w1;
                const bool in_z = sz.in;
                const bool in_y = sy.in;
                const bool in_x = sx.in;
                t.outside = !(in_z & in_y & in_x);
                t.oobmask = 0;
                const bool zz = t.wz1 == 0.f, yz = t.wy1 == 0.f;
                const int li = t.outside ? 0 : ((sz.i0 - b0[0]) * bd[1] + (sy.i0 - b0[1])) * bd[2] + (sx.i0 - b0[2]);
                const int stz = (t.outside | zz) ? 0 : plane_f;
                const int sty = (t.outside | yz) ? 0 : bd[2];
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const float *src = box + li + (m >> 1) * stz + (m & 1) * sty;
                    t.v[2 * m] = src[0];
                    t.v[2 * m + 1] = src[1];
                }
                r[kk] = finish<float>(t, (float)p.cval);
            }
        } else {
            Taps<float> t[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) c1_gather(rin, p, c1_split(cb[kk][0], p.nz), c1_split(cb[kk][1], p.ny), c1_split(cb[kk][2], p.nx), t[kk]);
#pragma unroll
            for (int kk = 0; kk < 4; kk++) r[kk] = finish<float>(t[kk], (float)p.cval);
        }
        if (wide) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) tile[kk * 64 + lane] = r[kk];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int i = lane >> 4, cc = lane & 15;
            const f32x4n v = *reinterpret_cast<const f32x4n *>(tile + i * 64 + 4 * cc);
            const int orow = y0 + RW * wave + (4 * cc) / TX, ox4 = x0w + ((4 * cc) & (TX - 1));
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4n *>(out + ((size_t)(z0 + 4 * bt + i) * p.oy + orow) * p.ox + ox4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            const int x = x0w + lx, y = y0 + yrow;
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                const int z = z0 + 4 * bt + kk;
                if (x < p.ox && y < p.oy && z < p.oz) __builtin_nontemporal_store(r[kk], out + ((size_t)z * p.oy + y) * p.ox + x);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// r3: map_coordinates, order 1, constant mode, float32 -- two x-neighbouring voxels per lane sharing their gathers.
// The L1 serves a gather four lanes at a time and pays per cache line touched (profiles/r3_affine_counters.txt): what
// costs is the NUMBER of lane-quads that gather, not the bytes.  A lane owns voxels x and x + 1 of a row; when the second
// voxel's taps lie in the same two rows / planes and start 0, 1 or 2 samples after the first one's (every smooth warp:
// ~85 % of the pairs of config D), ONE 16-byte gather per (plane, row) serves both voxels: four gathers per pair
// instead of eight.  Pairs that do not qualify (row change between the two voxels, one of them outside, zoom-out by
// more than 2) issue the second voxel's own 8-byte gathers under the lane mask -- any coordinates are handled, the
// result is bit-identical to the other order-1 kernels (same splits, same finish()).
// ---------------------------------------------------------------------------
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));

struct C1Addr { unsigned base, stz, sty; bool lastcol; int xb; };

__device__ __forceinline__ void c1_address(const FastInterpParams &p, const C1Split &sz, const C1Split &sy, const C1Split &sx, Taps<float> &t,
                                           C1Addr &ad)
{
    t.wz1 = sz.w1; t.wy1 = sy.w1; t.wx1 = sx.w1;
    const bool in_z = sz.in;
    const bool in_y = sy.in;
    const bool in_x = sx.in;
    t.outside = !(in_z & in_y & in_x);
    t.oobmask = 0;
    const bool zz = t.wz1 == 0.f, yz = t.wy1 == 0.f;
    ad.lastcol = sx.i0 >= p.nx - 1;
    ad.xb = sx.i0 - (ad.lastcol ? 1 : 0);
    ad.base = t.outside ? 0u : (unsigned)((sz.i0 * p.ny + sy.i0) * p.nx + ad.xb) * 4u;
    ad.stz = (t.outside | zz) ? 0u : (unsigned)(p.ny * p.nx) * 4u;
    ad.sty = (t.outside | yz) ? 0u : (unsigned)p.nx * 4u;
}

__device__ __forceinline__ float pick3(const u32x4g &q, int d)      // component d of q, d in 0..3
{
    const unsigned lo = d & 1 ? q.y : q.x, hi = d & 1 ? q.w : q.z;
    return __uint_as_float(d & 2 ? hi : lo);
}

__global__ void __launch_bounds__(256)
map_coords3d_pair_kernel(const float *__restrict__ in, const float *__restrict__ coords, float *__restrict__ out,
                         const FastInterpParams p)
{
    // block (64, 4): lane -> voxels x0w + 2 lane, + 1; ty -> row; four planes per thread (z-major ownership)
    const int lane = threadIdx.x, ty = threadIdx.y;
    const int x = blockIdx.x * 128 + 2 * lane, y = blockIdx.y * 4 + ty, zb = blockIdx.z * 4;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, p.nz * p.ny * p.nx * 4, 0x00020000);
    const size_t nout = (size_t)p.oz * p.oy * p.ox;
    const bool live = x < p.ox && y < p.oy;                       // ox is even: x + 1 < ox as well
    const int xc = min(x, p.ox - 2), yc = min(y, p.oy - 1);
    typedef float f32x2n __attribute__((ext_vector_type(2)));
    f32x2n c[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t o = ((size_t)min(zb + k, p.oz - 1) * p.oy + yc) * p.ox + xc;
#pragma unroll
        for (int a = 0; a < 3; a++) c[k][a] = __builtin_nontemporal_load(reinterpret_cast<const f32x2n *>(coords + a * nout + o));
    }
#pragma unroll
    for (int h = 0; h < 2; h++) {                                 // two planes at a time: eight 16-byte gathers in flight
        Taps<float> tA[2], tB[2];
        C1Addr aA[2], aB[2];
        u32x4g q[2][4];
        bool shared[2];
        int d[2];
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int k = 2 * h + kk;
            c1_address(p, c1_split(c[k][0].x, p.nz), c1_split(c[k][1].x, p.ny), c1_split(c[k][2].x, p.nx), tA[kk], aA[kk]);
            c1_address(p, c1_split(c[k][0].y, p.nz), c1_split(c[k][1].y, p.ny), c1_split(c[k][2].y, p.nx), tB[kk], aB[kk]);
#pragma unroll
            for (int m = 0; m < 4; m++)
                q[kk][m] = __builtin_amdgcn_raw_buffer_load_b128(rin, aA[kk].base + (m >> 1) * aA[kk].stz + (m & 1) * aA[kk].sty, 0, 0);
            // B's taps inside A's four 16-byte rows?  same (z0, y0), strides equal or not needed, x start 0..2 after A's
            const unsigned rowA = aA[kk].base - (unsigned)aA[kk].xb * 4u, rowB = aB[kk].base - (unsigned)aB[kk].xb * 4u;
            d[kk] = aB[kk].xb + (aB[kk].lastcol ? 1 : 0) - aA[kk].xb;               // B's x0 relative to A's first sample
            shared[kk] = !tA[kk].outside & !tB[kk].outside & (rowA == rowB) & ((aB[kk].stz == aA[kk].stz) | (aB[kk].stz == 0u)) &
                         ((aB[kk].sty == aA[kk].sty) | (aB[kk].sty == 0u)) & ((unsigned)d[kk] <= 2u);
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            // A: components 0, 1 (at the last column the pair was shifted left: the sample is component 1)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float a0 = __uint_as_float(q[kk][m].x), a1 = __uint_as_float(q[kk][m].y);
                tA[kk].v[2 * m] = aA[kk].lastcol ? a1 : a0;
                tA[kk].v[2 * m + 1] = a1;
            }
            if (shared[kk]) {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    // a row / plane B skips (its stride 0) has weight 0 in finish(): whatever A loaded there is never used
                    tB[kk].v[2 * m] = pick3(q[kk][m], d[kk]);
                    tB[kk].v[2 * m + 1] = pick3(q[kk][m], d[kk] + 1);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    float a, b;
                    load_pair(rin, aB[kk].base + (m >> 1) * aB[kk].stz + (m & 1) * aB[kk].sty, a, b);
                    tB[kk].v[2 * m] = aB[kk].lastcol ? b : a;
                    tB[kk].v[2 * m + 1] = b;
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            const int z = zb + 2 * h + kk;
            f32x2n r;
            r.x = finish<float>(tA[kk], (float)p.cval);
            r.y = finish<float>(tB[kk], (float)p.cval);
            if (live && z < p.oz) __builtin_nontemporal_store(r, reinterpret_cast<f32x2n *>(out + ((size_t)z * p.oy + y) * p.ox + x));
        }
    }
}

constexpr int kNV = 4;   // voxels per thread (rows 4 apart), all gathers issued before any is used

// block = (64, 4): 64 lanes along x (one voxel each, so every gather instruction of a
// wave touches neighbouring input voxels); a thread handles kNV rows; grid = (x tiles, y tiles, z)
template <typename T, typename CT, bool FASTC, int ORDER>
__global__ void __launch_bounds__(256)
map_coords3d_fast(const T *__restrict__ in, const CT *__restrict__ coords, T *__restrict__ out,
                  const FastInterpParams p)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int z = blockIdx.z;
    if (x >= p.ox) return;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, p.nz * p.ny * p.nx * (int)sizeof(T), 0x00020000);
    const size_t nout = (size_t)p.oz * p.oy * p.ox;
    CT c[kNV][3];
    size_t o[kNV];
    bool ok[kNV];
#pragma unroll
    for (int k = 0; k < kNV; k++) {
        const int y = (blockIdx.y * kNV + k) * 4 + threadIdx.y;
        ok[k] = y < p.oy;
        o[k] = ((size_t)z * p.oy + (ok[k] ? y : 0)) * p.ox + x;
        // coordinates are read once: keep them out of the caches the gathered volume lives in
        if (p.two_d) {
            c[k][0] = (CT)0;
            c[k][1] = __builtin_nontemporal_load(coords + o[k]);
            c[k][2] = __builtin_nontemporal_load(coords + nout + o[k]);
        } else {
            c[k][0] = __builtin_nontemporal_load(coords + o[k]);
            c[k][1] = __builtin_nontemporal_load(coords + nout + o[k]);
            c[k][2] = __builtin_nontemporal_load(coords + 2 * nout + o[k]);
        }
    }
    Taps<T> t[kNV];
#pragma unroll
    for (int k = 0; k < kNV; k++) taps<T, CT, FASTC, ORDER>(rin, p, c[k][0], c[k][1], c[k][2], t[k]);
#pragma unroll
    for (int k = 0; k < kNV; k++)
        if (ok[k]) __builtin_nontemporal_store(finish<T>(t[k], (T)p.cval), out + o[k]);
}

template <typename T, bool FASTC, int ORDER>
__global__ void __launch_bounds__(256)
affine3d_fast(const T *__restrict__ in, T *__restrict__ out, const FastInterpParams p)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int z = blockIdx.z;
    if (x >= p.ox) return;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, p.nz * p.ny * p.nx * (int)sizeof(T), 0x00020000);
    Taps<T> t[kNV];
    size_t o[kNV];
    bool ok[kNV];
    const double dz = (double)z, dx = (double)x;
#pragma unroll
    for (int k = 0; k < kNV; k++) {
        const int y = (blockIdx.y * kNV + k) * 4 + threadIdx.y;
        ok[k] = y < p.oy;
        o[k] = ((size_t)z * p.oy + (ok[k] ? y : 0)) * p.ox + x;
        // same summation order as the oracle: ((m0*z + m1*y) + m2*x) + offset
        const double dy = (double)y;
        const double cz = ((0.0 + p.m[0] * dz) + p.m[1] * dy + p.m[2] * dx) + p.m[3];
        const double cy = ((0.0 + p.m[4] * dz) + p.m[5] * dy + p.m[6] * dx) + p.m[7];
        const double cx = ((0.0 + p.m[8] * dz) + p.m[9] * dy + p.m[10] * dx) + p.m[11];
        taps<T, double, FASTC, ORDER>(rin, p, cz, cy, cx, t[k]);
    }
#pragma unroll
    for (int k = 0; k < kNV; k++)
        if (ok[k]) __builtin_nontemporal_store(finish<T>(t[k], (T)p.cval), out + o[k]);
}

// ---------------------------------------------------------------------------
// r4: affine_transform, order 1, constant mode, float32 volumes whose matrix leaves axis 0 to itself
// (m01 = m02 = m10 = m20 = 0: an in-plane rotation / shear / scaling of every slice plus a scaling / shift through the
// slices -- BASELINE config D', `rotate(volume, angle, axes=(1, 2))`, slice-wise registration).  STREAMS ALONG z.
//
// Why a third kernel: affine3d_lds_kernel stages a 3-D bounding box per 8-plane tile (2.05 x the tile's samples from
// L2, 2.03 x the input from HBM) and recomputes three float64 coordinates per voxel (49 VALU instructions per voxel,
// profiles/r3_kernel_counters.txt) -- 337-348 us on config D', 0.39 of the roofline.  With axis 0 decoupled
//   * the in-plane coordinates (cy, cx) of an output voxel do not depend on z: a thread computes the LDS address and
//     the two weights of each of its eight (y, x) positions ONCE per workgroup and keeps them in registers while the
//     workgroup walks down a chunk of output planes -- per voxel and plane what is left is two address adds, four
//     ds_read2_b32 and the blend (~20 VALU instructions);
//   * the z coordinate is the same for a whole output plane: its two input planes and their weights are wave-uniform;
//   * input planes enter LDS once per (y, x) tile as the tile's in-plane bounding rectangle (RY rows of P = 80 samples,
//     `buffer_load_dwordx4 ... lds`), in a ring of four slots indexed by (input plane & 3): the planes the NEXT output
//     plane needs are fetched while the current one is interpolated (|m00| <= 2 keeps the four live planes in distinct
//     slots).  No amplification along z (1.27 x in the box kernel), in-plane 1.3-1.6 x from L2, most of which the L2 /
//     MALL serve: x-neighbouring tiles -- whose rectangles overlap -- are given to the same XCD.
// Coordinates, splits, in-range tests and the blend (finish) are those of affine3d_c1_kernel in the oracle's summation
// order (the zero terms of the decoupled matrix add exact zeros): results are bit-identical to the other order-1 kernels.
// ---------------------------------------------------------------------------
constexpr int kZsP = 80;                    // LDS row pitch (floats) of a staged plane rectangle = 20 sixteen-byte chunks
constexpr int kZsRoundsMax = 8;             // staging rounds (NT chunks each) a plane rectangle can take

struct ZStreamParams {
    // The STREAM axis S (0 or 1) is the one the matrix leaves to itself; the plane it is streamed through has the ROW
    // axis R (the other of 0 / 1) and x.  S = 0: planes are (y, x) slices, consecutive planes nx ny samples apart;
    // S = 1: planes are (z, x) slices -- rows nx ny samples apart, consecutive planes one row (nx) apart.
    int stream_axis;
    int nS, nR, nx;          // input extents along S, R, x
    int oS, oR, ox;          // output extents
    unsigned in_sS, in_sR;   // input strides in samples (x stride 1)
    size_t out_sS, out_sR;   // output strides in samples
    int vol_bytes;           // input bytes (buffer range)
    double mS, offS;         // cS = mS s + offS
    double mRR, mRx, offR;   // cR = (mRR r + mRx x) + offR      (the oracle's summation order: the decoupled axis adds exact zeros)
    double mxR, mxx, offx;   // cx = (mxR r + mxx x) + offx
    double cval;
    int ry;                  // rows of the staged window
    int P;                   // LDS row pitch in floats (a multiple of 4, <= kZsP): the widest span a staged row needs
    int shear;               // 1 = every staged row starts at its own first needed column, 0 = bounding rectangle (pitch kZsP)
    int nslots;              // ring slots: 4, or 3 when |mS| <= 1.3 (up to 1 an output plane and the next read at most three input planes;
                             // beyond it a plane whose slot is still being read is fetched late)
    int nchunks;             // ry * P / 4
    double cmin_y, cmin_x;   // minimum of cR / cx over a tile relative to its first voxel (see LdsAffineParams::cmin)
    int zc, nzc;             // output planes per chunk, chunks
    int ntx, nty;            // tiles along x / R
    int dbg;
};

// r4b: the staged window of a plane is SHEARED.  A tile of TY x 64 output voxels reads, in one input plane, a parallelogram
// (the in-plane part of the matrix applied to the tile); its bounding rectangle is 2.3 x the tile at 30 degrees and its
// rows fill LDS (one workgroup per CU, 322 us against 202 us at 7 degrees).  Each staged input row now starts at its OWN
// first needed column (aligned down to 16 bytes) and is as long as the widest row needs: zs_row_span() gives, for input
// row iy, the columns read by the voxels whose lower or upper tap row it is -- the tile rectangle cut by the band
// iy - 1 <= cR < iy + 1, extremes of cx over the cut polygon's vertices.  Chunks of a row beyond its own span are not
// fetched at all (their DMA offset fails the descriptor's range check), so the traffic follows the parallelogram.
struct ZsSpan { int lo, hi; bool any; };
__host__ __device__ inline ZsSpan zs_row_span(double mRR, double mRx, double mxR, double mxx, double cR, double cx, int T0, int T1, int iy)
{
    const double h = 1e-6 * (2.0 + fabs((double)iy) + fabs(cR));
    const double vlo = (double)iy - 1.0 - h, vhi = (double)iy + 1.0 + h;
    double umin = 1e300, umax = -1e300;
    auto take = [&](double y, double x) {
        const double u = (mxR * y + mxx * x) + cx;
        umin = u < umin ? u : umin;
        umax = u > umax ? u : umax;
    };
    const double ys[2] = {0.0, (double)T0}, xs[2] = {0.0, (double)T1};
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 2; b++) {
            const double v = (mRR * ys[a] + mRx * xs[b]) + cR;
            if (v >= vlo && v <= vhi) take(ys[a], xs[b]);
        }
    const double vb[2] = {vlo, vhi};
    for (int e = 0; e < 2; e++) {
        for (int b = 0; b < 2; b++) {
            if (mRR != 0.0) {                                    // edges x = 0 / x = T1: where they cross the band's boundaries
                const double y = (vb[e] - cR - mRx * xs[b]) / mRR;
                if (y >= -1e-9 && y <= (double)T0 + 1e-9) take(y, xs[b]);
            }
            if (mRx != 0.0) {                                    // edges y = 0 / y = T0
                const double x = (vb[e] - cR - mRR * ys[b]) / mRx;
                if (x >= -1e-9 && x <= (double)T1 + 1e-9) take(ys[b], x);
            }
        }
    }
    ZsSpan s;
    s.any = umax >= umin;
    const double hu = 1e-6 * (2.0 + fabs(umin) + fabs(umax));
    s.lo = s.any ? (int)floor(fmax(umin - hu, -1e9)) : 0;
    s.hi = s.any ? (int)floor(fmin(umax + hu, 1e9)) + 1 : -1;
    return s;
}

struct ZSplit { int i0; float w1; bool in; };
__device__ __forceinline__ ZSplit zs_split(double m0, double m3, int z, int nz)
{
    const C1Split s = c1_split(m0 * (double)z + m3, nz);
    ZSplit r;
    r.i0 = __builtin_amdgcn_readfirstlane(s.i0);
    r.w1 = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s.w1)));
    r.in = __builtin_amdgcn_readfirstlane((int)s.in) != 0;
    return r;
}

// ---------------------------------------------------------------------------
// The RECTANGLE form of the z-streaming affine kernel (r4, first half; config D'): every staged row starts at the same
// column, four ring slots, the upper tap row an immediate offset away.  Kept as a kernel of its own: folded into the
// sheared kernel below as a template variant it ran 7 % slower on config D' (214-219 against 202 us: the per-voxel
// check, the gathering copy of the loop and the row table cost registers and code the loop does not need here).
// Taken when the bounding rectangle fits LDS twice per CU with four slots; everything else goes to the sheared kernel.
// ---------------------------------------------------------------------------
// Input plane `pl` into ring slot (pl & 3) unless it is resident or outside the volume; ROUNDS DMAs per thread.  The
// resident planes are a contiguous range [rlo, rhi] of at most four (two scalars: a tag per slot would be indexed by a
// run-time slot number, i.e. live in scratch memory -- whose accesses count in vmcnt and would break the kernel's
// hand-counted waits); a plane next to the range extends it (dropping the far end beyond four), any other plane
// restarts it.
template <int NT>
__device__ __forceinline__ void zs_ensure_rect(const float *in, int vol_bytes, int pl, int nz, int &rlo, int &rhi,
                                          const unsigned (&rel)[kZsRoundsMax], int rounds, unsigned plane_b, unsigned org_b,
                                          unsigned slot_bytes, int wave, bool off)
{
    if (pl < 0 || pl >= nz) return;
    if (pl >= rlo && pl <= rhi) return;
    if (pl == rhi + 1 && rhi >= rlo) { rhi = pl; if (rhi - rlo > 3) rlo = rhi - 3; }
    else if (pl == rlo - 1 && rhi >= rlo) { rlo = pl; if (rhi - rlo > 3) rhi = rlo + 3; }
    else { rlo = rhi = pl; }
    const int sl = pl & 3;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)pl * plane_b + org_b);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)sl * slot_bytes + (unsigned)(wave << 6) * 16u);
#pragma unroll
    for (int j = 0; j < kZsRoundsMax; j++)
        if (j < rounds && !off) dma_16s(rin, rel[j], base, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(j * NT) * 16u));
}

template <int TY, int SAX>
__global__ void __launch_bounds__(TY * 8)
affine3d_zrect_kernel(const float *__restrict__ in, float *__restrict__ out, const ZStreamParams q)
{
    constexpr int NT = TY * 8;                         // threads = TY / 8 waves; a wave owns 8 output rows of 64 voxels
    constexpr int P = kZsP;
    extern __shared__ __attribute__((aligned(16))) char smem_zs[];
    const unsigned slot_bytes = (((unsigned)q.nchunks + NT - 1) / NT) * NT * 16u;       // whole rounds of NT chunks
    float *tiles = reinterpret_cast<float *>(smem_zs + 4u * slot_bytes);                // [NW][8 rows][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (tile x, tile y, z chunk): consecutive block indices go round the XCDs, so block b is given the
    // tile whose index in (chunk, ty, tx) order is (b % 8) * (total / 8) + b / 8 when the grid divides by 8 -- every XCD
    // then holds runs of x-neighbouring tiles, whose rectangles overlap, of the same chunk
    const int total = q.ntx * q.nty * q.nzc;
    int t = blockIdx.x;
    if ((total & 7) == 0 && !(q.dbg & 64)) t = (t & 7) * (total >> 3) + (t >> 3);
    const int tx_i = t % q.ntx, ty_i = (t / q.ntx) % q.nty, zc_i = t / (q.ntx * q.nty);
    const int x0 = tx_i * 64, y0 = ty_i * TY;
    const int zs = zc_i * q.zc, ze = min(zs + q.zc, q.oS);

    // ---- the rectangle: origin from the tile's first voxel (closed form, a hair below the true minimum, clamped
    // into the volume, x aligned down to 16 bytes), chunk -> byte offset from the origin once per thread
    int borg[2];
#pragma unroll
    for (int a = 1; a <= 2; a++) {
        const double lo = a == 1 ? (q.mRR * (double)y0 + q.mRx * (double)x0) + (q.offR + q.cmin_y)
                                 : (q.mxR * (double)y0 + q.mxx * (double)x0) + (q.offx + q.cmin_x);
        const int n = a == 1 ? q.nR : q.nx;
        double f = floor(lo - 1e-6 * (1.0 + fabs(lo)));
        f = f < 0.0 ? 0.0 : (f > (double)(n - 1) ? (double)(n - 1) : f);
        borg[a - 1] = __builtin_amdgcn_readfirstlane(a == 2 ? ((int)f & ~3) : (int)f);
    }
    const int by0 = borg[0], bx0 = borg[1];
    const int rounds = (q.nchunks + NT - 1) / NT;
    unsigned rel[kZsRoundsMax];
#pragma unroll
    for (int j = 0; j < kZsRoundsMax; j++) {
        const unsigned ch = (unsigned)tid + (unsigned)(j * NT);
        const unsigned row = ch / 20u, c4 = ch - row * 20u;
        // rows past the rectangle are not fetched (0x80000000 fails the descriptor's range check: zeros)
        rel[j] = ch < (unsigned)q.nchunks ? row * q.in_sR * 4u + c4 * 16u : 0x80000000u;
    }
    const unsigned plane_b = q.in_sS * 4u;
    const unsigned org_b = ((unsigned)by0 * q.in_sR + (unsigned)bx0) * 4u;

    // ---- per-thread, per-(y, x): LDS byte offset of the lower-left tap, weights, in-plane range test.  Voxel k of a
    // lane: row y0 + 8 wave + k, column x0 + lane.
    int a_[8];
    float wy_[8], wx_[8];
    unsigned inmask = 0;
    {
        const double dx = (double)(x0 + lane);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double dy = (double)(y0 + 8 * wave + k);
            // the oracle's order ((m0 z + m1 y) + m2 x) + offset with m0 = 0 for these two rows
            const C1Split sy = c1_split((q.mRR * dy + q.mRx * dx) + q.offR, q.nR);
            const C1Split sx = c1_split((q.mxR * dy + q.mxx * dx) + q.offx, q.nx);
            const bool in = sy.in & sx.in;
            inmask |= in ? (1u << k) : 0u;
            a_[k] = in ? ((sy.i0 - by0) * P + (sx.i0 - bx0)) * 4 : 0;
            wy_[k] = sy.w1; wx_[k] = sx.w1;
        }
    }
    float *tile = tiles + wave * 512;
    const bool wide = x0 + 64 <= q.ox && y0 + TY <= q.oR;      // block-uniform: 16-byte stores through the wave's LDS tile

    // ---- the plane ring: slot (plane & 3) holds input plane `plane` for the planes of [rlo, rhi]
    int rlo = 0, rhi = -1;                    // resident input planes (empty)
    const int nz_ = q.nS, vol_bytes = q.vol_bytes;
    const double m0_ = q.mS, m3_ = q.offS;
    constexpr bool s0 = SAX == 0;
    const bool no_dma = (q.dbg & 1) != 0;
#define ZS_ENSURE(PL) zs_ensure_rect<NT>(in, vol_bytes, (PL), nz_, rlo, rhi, rel, rounds, plane_b, org_b, slot_bytes, wave, no_dma)
    ZSplit cur = zs_split(m0_, m3_, zs, nz_);
    if (cur.in) { ZS_ENSURE(cur.i0); ZS_ENSURE(cur.i0 + 1); }
    const float cval = (float)q.cval;

#pragma unroll 1
    for (int z = zs; z < ze; z++) {
        // the planes of this step have landed (the stores of the previous step, issued after their DMAs, may still be
        // in flight: two per thread on full tiles), and everyone has finished reading the previous step's planes
        // (r4b: the FIRST step has no stores behind the prologue's DMAs -- vmcnt(2) there let the last two chunks of a
        // chunk's first plane be read before they had landed: a few wrong voxels in the first plane of a z chunk under
        // back-to-back launches, found by the whole-volume check of scripts/bench_configs.py)
        if (wide && z > zs) asm volatile(MI_VMCNT(2) ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ZSplit nxt = cur;
        if (z + 1 < ze) {
            nxt = zs_split(m0_, m3_, z + 1, nz_);
            if (nxt.in) { ZS_ENSURE(nxt.i0); ZS_ENSURE(nxt.i0 + 1); }
        }
        float r[8];
        if (cur.in && !(q.dbg & 2)) {
            const char *lo_p = smem_zs + (unsigned)(cur.i0 & 3) * slot_bytes;
            const char *hi_p = smem_zs + (unsigned)((cur.i0 + 1) & 3) * slot_bytes;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float *A = reinterpret_cast<const float *>(lo_p + a_[k]);
                const float *B = reinterpret_cast<const float *>(hi_p + a_[k]);
                Taps<float> t;
                // v[(z << 2) | (y << 1) | x]: the stream axis selects the slot (A / B), the row axis the LDS row
                const float a00 = A[0], a01 = A[1], a10 = A[P], a11 = A[P + 1];
                const float b00 = B[0], b01 = B[1], b10 = B[P], b11 = B[P + 1];
                t.v[0] = a00; t.v[1] = a01;
                t.v[2] = s0 ? a10 : b00; t.v[3] = s0 ? a11 : b01;
                t.v[4] = s0 ? b00 : a10; t.v[5] = s0 ? b01 : a11;
                t.v[6] = b10; t.v[7] = b11;
                t.wz1 = s0 ? cur.w1 : wy_[k]; t.wy1 = s0 ? wy_[k] : cur.w1; t.wx1 = wx_[k];
                t.oobmask = 0;
                t.outside = !((inmask >> k) & 1u);
                r[k] = finish<float>(t, cval);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) r[k] = cval;
        }
        if (!(q.dbg & 4)) {
            if (wide) {
#pragma unroll
                for (int k = 0; k < 8; k++) tile[k * 64 + lane] = r[k];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int i = lane >> 4, c = lane & 15;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const f32x4n v = *reinterpret_cast<const f32x4n *>(tile + (4 * h + i) * 64 + 4 * c);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4n *>(out + (size_t)z * q.out_sS + (size_t)(y0 + 8 * wave + 4 * h + i) * q.out_sR + x0 + 4 * c));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the tile is rewritten next step
            } else {
                const int x = x0 + lane;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int y = y0 + 8 * wave + k;
                    if (x < q.ox && y < q.oR) __builtin_nontemporal_store(r[k], out + (size_t)z * q.out_sS + (size_t)y * q.out_sR + x);
                }
            }
        }
        cur = nxt;
    }
#undef ZS_ENSURE
}

// Input plane `pl` into ring slot (pl & 3) unless it is resident or outside the volume; ROUNDS DMAs per thread.  The
// resident planes are a contiguous range [rlo, rhi] of at most four (two scalars: a tag per slot would be indexed by a
// run-time slot number, i.e. live in scratch memory -- whose accesses count in vmcnt and would break the kernel's
// hand-counted waits); a plane next to the range extends it (dropping the far end beyond four), any other plane
// restarts it.
template <int NT>
__device__ __forceinline__ void zs_ensure(const float *in, int vol_bytes, int pl, int nz, int &rlo, int &rhi,
                                          const unsigned (&rel)[kZsRoundsMax], int rounds, unsigned plane_b, unsigned org_b,
                                          unsigned slot_bytes, int wave, bool off, int nslots)
{
    if (pl < 0 || pl >= nz) return;
    if (pl >= rlo && pl <= rhi) return;
    if (pl == rhi + 1 && rhi >= rlo) { rhi = pl; if (rhi - rlo > nslots - 1) rlo = rhi - (nslots - 1); }
    else if (pl == rlo - 1 && rhi >= rlo) { rlo = pl; if (rhi - rlo > nslots - 1) rhi = rlo + (nslots - 1); }
    else { rlo = rhi = pl; }
    const int sl = nslots == 4 ? (pl & 3) : pl % 3;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
    const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)pl * plane_b + org_b);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)sl * slot_bytes + (unsigned)(wave << 6) * 16u);
#pragma unroll
    for (int j = 0; j < kZsRoundsMax; j++)
        if (j < rounds && !off) dma_16s(rin, rel[j], base, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(j * NT) * 16u));
}

// SHEAR = true is what is instantiated (row starts of their own); the rectangle form lives in affine3d_zrect_kernel above.
template <int TY, int SAX, bool SHEAR>
__global__ void __launch_bounds__(TY * 8)
affine3d_zstream_kernel(const float *__restrict__ in, float *__restrict__ out, const ZStreamParams q)
{
    constexpr int NT = TY * 8;                         // threads = TY / 8 waves; a wave owns 8 output rows of 64 voxels
    const int P = SHEAR ? q.P : kZsP, PC = P >> 2;     // LDS row pitch of the staged window: floats / 16-byte chunks
    extern __shared__ __attribute__((aligned(16))) char smem_zs[];
    const unsigned slot_bytes = (((unsigned)q.nchunks + NT - 1) / NT) * NT * 16u;       // whole rounds of NT chunks
    const int NS = q.nslots;                                                            // ring slots (3 or 4), slot = plane mod NS
    float *tiles = reinterpret_cast<float *>(smem_zs + (unsigned)NS * slot_bytes);     // [NW][8 rows][64]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup -> (tile x, tile y, z chunk): consecutive block indices go round the XCDs, so block b is given the
    // tile whose index in (chunk, ty, tx) order is (b % 8) * (total / 8) + b / 8 when the grid divides by 8 -- every XCD
    // then holds runs of x-neighbouring tiles, whose windows overlap, of the same chunk
    const int total = q.ntx * q.nty * q.nzc;
    int t = blockIdx.x;
    if ((total & 7) == 0 && !(q.dbg & 64)) t = (t & 7) * (total >> 3) + (t >> 3);
    const int tx_i = t % q.ntx, ty_i = (t / q.ntx) % q.nty, zc_i = t / (q.ntx * q.nty);
    const int x0 = tx_i * 64, y0 = ty_i * TY;
    const int zs = zc_i * q.zc, ze = min(zs + q.zc, q.oS);

    // ---- the window: first staged row from the tile's first voxel (closed form, a hair below the true minimum, clamped
    // into the volume); per staged row its first column (aligned down to 16 bytes) and its last one, once per workgroup
    const double cR0 = (q.mRR * (double)y0 + q.mRx * (double)x0) + q.offR, cx0 = (q.mxR * (double)y0 + q.mxx * (double)x0) + q.offx;
    int by0;
    {
        const double lo = cR0 + q.cmin_y;
        double f = floor(lo - 1e-6 * (1.0 + fabs(lo)));
        f = f < 0.0 ? 0.0 : (f > (double)(q.nR - 1) ? (double)(q.nR - 1) : f);
        by0 = __builtin_amdgcn_readfirstlane((int)f);
    }
    int *rowtab = reinterpret_cast<int *>(tiles);               // [2][ry] until the loop starts: first column, last column
    if constexpr (SHEAR) {
        for (int r = tid; r < q.ry; r += NT) {
            const ZsSpan sp = zs_row_span(q.mRR, q.mRx, q.mxR, q.mxx, cR0, cx0, TY - 1, 63, by0 + r);
            const int st = sp.lo < 0 ? 0 : (sp.lo > q.nx - 1 ? q.nx - 1 : sp.lo);
            rowtab[r] = st & ~3;
            rowtab[q.ry + r] = sp.any ? sp.hi : -1;
        }
    } else {
        // one origin for all rows: the closed-form minimum of cx over the tile, a hair below, clamped, aligned down
        const double lo = cx0 + q.cmin_x;
        double f = floor(lo - 1e-6 * (1.0 + fabs(lo)));
        f = f < 0.0 ? 0.0 : (f > (double)(q.nx - 1) ? (double)(q.nx - 1) : f);
        const int bx0 = (int)f & ~3;
        for (int r = tid; r < q.ry; r += NT) {
            rowtab[r] = bx0;
            rowtab[q.ry + r] = bx0 + P - 1;
        }
    }
    __syncthreads();
    const int rounds = (q.nchunks + NT - 1) / NT;
    unsigned rel[kZsRoundsMax];
#pragma unroll
    for (int j = 0; j < kZsRoundsMax; j++) {
        const unsigned ch = (unsigned)tid + (unsigned)(j * NT);
        const unsigned row = ch / (unsigned)PC, c4 = ch - row * (unsigned)PC;
        bool need = ch < (unsigned)q.nchunks;
        int st = 0;
        if (need) {
            st = rowtab[row];
            need = st + 4 * (int)c4 <= rowtab[q.ry + (int)row];
        }
        // chunks past a row's own span and rows past the window are not fetched (0x80000000 fails the descriptor's range check)
        rel[j] = need ? (row * q.in_sR + (unsigned)st + 4u * c4) * 4u : 0x80000000u;
    }
    const unsigned plane_b = q.in_sS * 4u, row_b = q.in_sR * 4u;
    const unsigned org_b = (unsigned)by0 * q.in_sR * 4u;

    // ---- per-thread, per-(y, x): LDS byte offsets of the lower-left tap in its row and in the row above (the rows start
    // at different columns), weights, in-plane range test.  Voxel k of a lane: row y0 + 8 wave + k, column x0 + lane.
    // A voxel whose taps the window does not hold (the plan sizes P from sampled tile offsets: this is the net under it)
    // keeps its in-plane offset into the VOLUME instead, flagged by the sign bit, and gathers for itself in every plane.
    int a_[8], a2_[8];
    float wy_[8], wx_[8];
    unsigned inmask = 0, bad = 0;
    {
        const double dx = (double)(x0 + lane);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double dy = (double)(y0 + 8 * wave + k);
            // the oracle's order ((m0 z + m1 y) + m2 x) + offset with m0 = 0 for these two rows
            const C1Split sy = c1_split((q.mRR * dy + q.mRx * dx) + q.offR, q.nR);
            const C1Split sx = c1_split((q.mxR * dy + q.mxx * dx) + q.offx, q.nx);
            const bool in = sy.in & sx.in;
            inmask |= in ? (1u << k) : 0u;
            const int r = sy.i0 - by0;
            const bool rows_ok = in && r >= 0 && r + 1 < q.ry;
            const int c0 = sx.i0 - rowtab[rows_ok ? r : 0], c1 = sx.i0 - rowtab[rows_ok ? r + 1 : 0];
            const bool held = rows_ok && c0 >= 0 && c0 + 1 < P && c1 >= 0 && c1 + 1 < P &&
                              sx.i0 + 1 <= rowtab[q.ry + r] && sx.i0 + 1 <= rowtab[q.ry + r + 1];
            a_[k] = !in ? 0 : held ? (r * P + c0) * 4 : (int)(0x80000000u | ((unsigned)sy.i0 * q.in_sR + (unsigned)sx.i0) * 4u);
            a2_[k] = (in && held) ? ((r + 1) * P + c1) * 4 : 0;
            bad |= (in && !held) ? (1u << k) : 0u;
            wy_[k] = sy.w1; wx_[k] = sx.w1;
        }
    }
    const bool any_bad = __builtin_amdgcn_ballot_w64(bad != 0) != 0;
    __syncthreads();                                            // the row table lives where the output tiles are staged
    float *tile = tiles + wave * 512;
    const bool wide = x0 + 64 <= q.ox && y0 + TY <= q.oR;      // block-uniform: 16-byte stores through the wave's LDS tile

    // ---- the plane ring: slot (plane & 3) holds input plane `plane` for the planes of [rlo, rhi]
    int rlo = 0, rhi = -1;                    // resident input planes (empty)
    const int nz_ = q.nS, vol_bytes = q.vol_bytes;
    const double m0_ = q.mS, m3_ = q.offS;
    constexpr bool s0 = SAX == 0;
    const bool no_dma = (q.dbg & 1) != 0;
#define ZS_ENSURE(PL) zs_ensure<NT>(in, vol_bytes, (PL), nz_, rlo, rhi, rel, rounds, plane_b, org_b, slot_bytes, wave, no_dma, NS)
    auto slot_of = [&](int pl) { return (unsigned)(NS == 4 ? (pl & 3) : (pl % 3 + 3) % 3); };
    ZSplit cur = zs_split(m0_, m3_, zs, nz_);
    if (cur.in) { ZS_ENSURE(cur.i0); ZS_ENSURE(cur.i0 + 1); }
    bool drain = false;                       // the previous step issued DMAs after its stores
    const float cval = (float)q.cval;

#pragma unroll 1
    for (int z = zs; z < ze; z++) {
        // the planes of this step have landed (the stores of the previous step, issued after their DMAs, may still be
        // in flight: two per thread on full tiles), and everyone has finished reading the previous step's planes
        // (r4b: the FIRST step has no stores behind the prologue's DMAs -- vmcnt(2) there let the last two chunks of a
        // chunk's first plane be read before they had landed: a few wrong voxels in the first plane of a z chunk under
        // back-to-back launches, found by the whole-volume check of scripts/bench_configs.py)
        // (a step that fetched planes LATE -- after its stores, see below -- is followed by a full drain as well)
        if (wide && z > zs && !drain) asm volatile(MI_VMCNT(2) ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        drain = false;
        ZSplit nxt = cur;
        int late0 = -1, late1 = -1;
        if (z + 1 < ze) {
            nxt = zs_split(m0_, m3_, z + 1, nz_);
            if (nxt.in) {
                // With THREE ring slots (1 < |mS| <= 1.3: the next output plane usually shares a plane with this one, now and
                // then it does not) a plane whose slot one of this step's two planes occupies is fetched after they have been
                // read: one extra barrier and a drained pipeline every 1 / (|mS| - 1) steps, against a third fewer LDS.
                auto busy = [&](int pl) {
                    if (NS != 3 || !cur.in || (pl >= rlo && pl <= rhi) || pl < 0 || pl >= nz_) return false;
                    const int d0 = pl - cur.i0, d1 = pl - (cur.i0 + 1);
                    return d0 % 3 == 0 || d1 % 3 == 0;
                };
                if (busy(nxt.i0)) late0 = nxt.i0; else ZS_ENSURE(nxt.i0);
                if (busy(nxt.i0 + 1)) late1 = nxt.i0 + 1; else ZS_ENSURE(nxt.i0 + 1);
            }
        }
        float r[8];
        if (cur.in && !(q.dbg & 2) && any_bad) {
            // some voxel of this wave is not held by the window: the same blend with a per-voxel choice of the source
            const char *lo_p = smem_zs + slot_of(cur.i0) * slot_bytes;
            const char *hi_p = smem_zs + slot_of(cur.i0 + 1) * slot_bytes;
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
            const unsigned pbase = (unsigned)cur.i0 * plane_b;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float a00, a01, a10, a11, b00, b01, b10, b11;
                if (a_[k] < 0) {
                    const unsigned base = pbase + ((unsigned)a_[k] & 0x7fffffffu);
                    load_pair(rin, base, a00, a01);
                    load_pair(rin, base + row_b, a10, a11);
                    load_pair(rin, base + plane_b, b00, b01);
                    load_pair(rin, base + plane_b + row_b, b10, b11);
                } else {
                    const float *A = reinterpret_cast<const float *>(lo_p + a_[k]), *A1 = reinterpret_cast<const float *>(lo_p + a2_[k]);
                    const float *B = reinterpret_cast<const float *>(hi_p + a_[k]), *B1 = reinterpret_cast<const float *>(hi_p + a2_[k]);
                    a00 = A[0]; a01 = A[1]; a10 = A1[0]; a11 = A1[1];
                    b00 = B[0]; b01 = B[1]; b10 = B1[0]; b11 = B1[1];
                }
                Taps<float> t;
                t.v[0] = a00; t.v[1] = a01;
                t.v[2] = s0 ? a10 : b00; t.v[3] = s0 ? a11 : b01;
                t.v[4] = s0 ? b00 : a10; t.v[5] = s0 ? b01 : a11;
                t.v[6] = b10; t.v[7] = b11;
                t.wz1 = s0 ? cur.w1 : wy_[k]; t.wy1 = s0 ? wy_[k] : cur.w1; t.wx1 = wx_[k];
                t.oobmask = 0;
                t.outside = !((inmask >> k) & 1u);
                r[k] = finish<float>(t, cval);
            }
        } else if (cur.in && !(q.dbg & 2)) {
            const char *lo_p = smem_zs + slot_of(cur.i0) * slot_bytes;
            const char *hi_p = smem_zs + slot_of(cur.i0 + 1) * slot_bytes;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float *A = reinterpret_cast<const float *>(lo_p + a_[k]), *B = reinterpret_cast<const float *>(hi_p + a_[k]);
                const float *A1 = SHEAR ? reinterpret_cast<const float *>(lo_p + a2_[k]) : A + kZsP;
                const float *B1 = SHEAR ? reinterpret_cast<const float *>(hi_p + a2_[k]) : B + kZsP;
                Taps<float> t;
                // v[(z << 2) | (y << 1) | x]: the stream axis selects the slot (A / B), the row axis the LDS row
                const float a00 = A[0], a01 = A[1], a10 = A1[0], a11 = A1[1];
                const float b00 = B[0], b01 = B[1], b10 = B1[0], b11 = B1[1];
                t.v[0] = a00; t.v[1] = a01;
                t.v[2] = s0 ? a10 : b00; t.v[3] = s0 ? a11 : b01;
                t.v[4] = s0 ? b00 : a10; t.v[5] = s0 ? b01 : a11;
                t.v[6] = b10; t.v[7] = b11;
                t.wz1 = s0 ? cur.w1 : wy_[k]; t.wy1 = s0 ? wy_[k] : cur.w1; t.wx1 = wx_[k];
                t.oobmask = 0;
                t.outside = !((inmask >> k) & 1u);
                r[k] = finish<float>(t, cval);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) r[k] = cval;
        }
        if (!(q.dbg & 4)) {
            if (wide) {
#pragma unroll
                for (int k = 0; k < 8; k++) tile[k * 64 + lane] = r[k];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int i = lane >> 4, c = lane & 15;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const f32x4n v = *reinterpret_cast<const f32x4n *>(tile + (4 * h + i) * 64 + 4 * c);
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4n *>(out + (size_t)z * q.out_sS + (size_t)(y0 + 8 * wave + 4 * h + i) * q.out_sR + x0 + 4 * c));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the tile is rewritten next step
            } else {
                const int x = x0 + lane;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int y = y0 + 8 * wave + k;
                    if (x < q.ox && y < q.oR) __builtin_nontemporal_store(r[k], out + (size_t)z * q.out_sS + (size_t)y * q.out_sR + x);
                }
            }
        }
        if (late0 >= 0 || late1 >= 0) {
            __builtin_amdgcn_s_barrier();                        // everyone has read this step's planes
            if (late0 >= 0) ZS_ENSURE(late0);
            if (late1 >= 0) ZS_ENSURE(late1);
            drain = true;
        }
        cur = nxt;
    }
#undef ZS_ENSURE
}

Knob g_affine_zstream{1};     // test hook: 0 = off (box / gather kernels), 1 = auto (TY by LDS budget), 32 / 64 = that tile height
Knob g_affine_zchunks{0};     // test hook: z chunks of the streaming kernel (0 = planner)

// plan for the z-streaming kernel; false when the matrix / sizes are outside what it takes.  S = the decoupled axis
// (0 or 1): row and column S of the matrix are zero except the diagonal entry.
template <int TY>
static bool zstream_plan(const FastInterpParams &p, int S, ZStreamParams *q)
{
    const double *m = p.m;
    const int R = 1 - S;
    for (int i = 0; i < 12; i++) if (!(fabs(m[i]) < 1e9)) return false;
    for (int j = 0; j < 3; j++)
        if (j != S && (m[4 * S + j] != 0.0 || m[4 * j + S] != 0.0)) return false;          // axis S decoupled
    if (!(fabs(m[4 * S + S]) <= 2.0)) return false;                                         // four ring slots suffice
    const int nin[3] = {p.nz, p.ny, p.nx}, nout[3] = {p.oz, p.oy, p.ox};
    const double mRR = m[4 * R + R], mRx = m[4 * R + 2], mxR = m[8 + R], mxx = m[10];
    const int T[2] = {TY - 1, 63};
    const double ey = fabs(mRR) * T[0] + fabs(mRx) * T[1], ex = fabs(mxR) * T[0] + fabs(mxx) * T[1];
    if (!(ey < 4096.0 && ex < 4096.0)) return false;
    // samples floor(min - hair) .. floor(max) + 1: at most floor(ext + hair) + 3 (LdsAffineParams); x: the origin is
    // aligned down by up to 3 samples
    const int ry = (int)floor(ey * (1.0 + 1e-6) + 2e-3) + 3;
    // r4b: the row pitch = the widest span a staged row needs (zs_row_span), sampled over the fractional positions a tile's
    // first voxel can have against the input grid, + the up-to-3 samples the row start is aligned down by, + 1 of slack
    // (a voxel the window then still misses gathers for itself: the kernel checks every voxel)
    int span = 0;
    {
        const double e[4] = {mRR * T[0], mRx * T[1], mxR * T[0], mxx * T[1]};
        const double cminy = (e[0] < 0.0 ? e[0] : 0.0) + (e[1] < 0.0 ? e[1] : 0.0);
        for (int fy = 0; fy < 4; fy++)
            for (int fx = 0; fx < 4; fx++) {
                const double cR = 1000.0 + 0.25 * fy, cx = 1000.0 + 0.25 * fx;
                const int b0 = (int)floor(cR + cminy - 1e-3);
                for (int r = 0; r < ry; r++) {
                    const ZsSpan sp = zs_row_span(mRR, mRx, mxR, mxx, cR, cx, T[0], T[1], b0 + r);
                    if (sp.any && sp.hi - sp.lo + 1 > span) span = sp.hi - sp.lo + 1;
                }
            }
    }
    const int Pmin = (span + 3 + 1 + 3) & ~3;
    // the bounding rectangle (one origin for all rows, pitch kZsP) when it fits LDS twice per CU: cheaper taps (SHEAR = false)
    constexpr int NT_ = TY * 8;
    const int rx = (int)floor(ex * (1.0 + 1e-6) + 2e-3) + 3 + 3;
    size_t NSl = fabs(m[4 * S + S]) <= 1.3 ? 3 : 4;                                          // ring slots of the sheared kernel (beyond 1: with late fetches)
    const size_t rect_lds = 4 * (size_t)(((ry * (kZsP / 4) + NT_ - 1) / NT_) * NT_ * 16) + (size_t)(TY / 8) * 2048;   // (the rectangle kernel: four)
    const bool rect_fits = rx <= kZsP && (ry * (kZsP / 4) + NT_ - 1) / NT_ <= kZsRoundsMax && rect_lds <= 150 * 1024;
    bool shear = !(rect_fits && 2 * (rect_lds + 1024) <= 160 * 1024);
    if (shear && (Pmin > kZsP || Pmin < 8)) {
        if (!rect_fits) return false;
        shear = false;
    }
    // The pitch also decides the LDS bank conflicts of the tap reads: the 64 lanes of a wave read along a slanted line
    // of the window -- every 1 / |sin| lanes the row changes and the address jumps by (first-column difference - P) --
    // and a pitch that is congruent to that run length modulo the 32 banks puts whole runs on the same banks (7 degrees:
    // runs of 8 lanes, P = 72 -> eight-way conflicts, 288 instead of 202 us; P = 80 none).  Simulated here for one wave
    // with the row starts the kernel will compute: the candidate with the fewest serialised bank accesses wins.
    int P = shear ? Pmin : kZsP;
    if (!shear) NSl = 4;
    if (shear) {
        double best = 1e300;
        const double cR = 1000.3, cx = 1000.3;
        const double e[4] = {mRR * T[0], mRx * T[1], mxR * T[0], mxx * T[1]};
        const double cminy = (e[0] < 0.0 ? e[0] : 0.0) + (e[1] < 0.0 ? e[1] : 0.0);
        const int b0 = (int)floor(cR + cminy - 1e-3);
        std::vector<int> start(ry + 1, 0);
        for (int r = 0; r < ry; r++) {
            const ZsSpan sp = zs_row_span(mRR, mRx, mxR, mxx, cR, cx, T[0], T[1], b0 + r);
            start[r] = (sp.lo < 0 ? 0 : sp.lo) & ~3;
        }
        for (int cand = Pmin; cand <= kZsP; cand += 4) {
            if ((size_t)(((ry * (cand / 4) + TY * 8 - 1) / (TY * 8)) * (TY * 8) * 16) * NSl + (size_t)(TY / 8) * 2048 > 150 * 1024) break;
            double cost = 0;
            for (int row = 0; row < TY; row += 5)                      // a few output rows of the tile
                for (int half = 0; half < 2; half++) {
                    int count[32] = {0};
                    int seen[32][4];
                    for (int l = 32 * half; l < 32 * half + 32; l++) {
                        const double v = (mRR * row + mRx * l) + cR, u = (mxR * row + mxx * l) + cx;
                        const int r = (int)floor(v) - b0, c = (int)floor(u) - start[r < 0 ? 0 : (r >= ry ? ry - 1 : r)];
                        const int addr = r * cand + c, bank = ((addr % 32) + 32) % 32;
                        bool dup = false;
                        for (int j = 0; j < count[bank] && j < 4; j++) dup = dup || seen[bank][j] == addr;
                        if (!dup) { if (count[bank] < 4) seen[bank][count[bank]] = addr; count[bank]++; }
                    }
                    int worst = 0;
                    for (int bk = 0; bk < 32; bk++) worst = count[bk] > worst ? count[bk] : worst;
                    cost += worst;
                }
            cost *= 1.0 + 0.004 * (cand - Pmin);                       // a wider pitch stages a little more: ties go to the narrow one
            if (cost < best) { best = cost; P = cand; }
        }
    }
    constexpr int NT = TY * 8;
    int nchunks = ry * (P / 4);
    size_t slot = (size_t)((nchunks + NT - 1) / NT) * NT * 16;
    if (shear && ((nchunks + NT - 1) / NT > kZsRoundsMax || NSl * slot + (size_t)(TY / 8) * 2048 > 150 * 1024 ||
                  (rect_fits && 2 * (NSl * slot + (size_t)(TY / 8) * 2048 + 1024) > 160 * 1024))) {
        // the sheared window does not fit (or not twice per CU either): the rectangle, if it fits at all
        if (!rect_fits) return false;
        shear = false;
        P = kZsP;
        NSl = 4;
        nchunks = ry * (P / 4);
        slot = (size_t)((nchunks + NT - 1) / NT) * NT * 16;
    }
    if ((nchunks + NT - 1) / NT > kZsRoundsMax) return false;
    if (NSl * slot + (size_t)(TY / 8) * 2048 > 150 * 1024) return false;
    q->stream_axis = S;
    q->nS = nin[S]; q->nR = nin[R]; q->nx = p.nx;
    q->oS = nout[S]; q->oR = nout[R]; q->ox = p.ox;
    q->in_sS = S == 0 ? (unsigned)p.ny * (unsigned)p.nx : (unsigned)p.nx;
    q->in_sR = R == 0 ? (unsigned)p.ny * (unsigned)p.nx : (unsigned)p.nx;
    q->out_sS = S == 0 ? (size_t)p.oy * p.ox : (size_t)p.ox;
    q->out_sR = R == 0 ? (size_t)p.oy * p.ox : (size_t)p.ox;
    q->vol_bytes = p.nz * p.ny * p.nx * 4;
    q->mS = m[4 * S + S]; q->offS = m[4 * S + 3];
    q->mRR = mRR; q->mRx = mRx; q->offR = m[4 * R + 3];
    q->mxR = mxR; q->mxx = mxx; q->offx = m[11];
    q->cval = p.cval;
    q->ry = ry;
    q->P = P;
    q->shear = shear ? 1 : 0;
    q->nslots = (int)NSl;
    q->nchunks = nchunks;
    const double e[4] = {mRR * T[0], mRx * T[1], mxR * T[0], mxx * T[1]};
    q->cmin_y = (e[0] < 0.0 ? e[0] : 0.0) + (e[1] < 0.0 ? e[1] : 0.0);
    q->cmin_x = (e[2] < 0.0 ? e[2] : 0.0) + (e[3] < 0.0 ? e[3] : 0.0);
    q->ntx = (p.ox + 63) / 64;
    q->nty = (q->oR + TY - 1) / TY;
    q->dbg = g_affine_dbg;
    return true;
}

template <int TY>
static int launch_affine_zstream(const float *in, float *out, ZStreamParams &q, hipStream_t s)
{
    constexpr int NT = TY * 8;
    const size_t slot = (size_t)((q.nchunks + NT - 1) / NT) * NT * 16;
    const size_t lds = (size_t)q.nslots * slot + (size_t)(TY / 8) * 2048;
    // chunks along the stream axis: fill the chip (workgroups per CU by LDS) with chunks of >= 16 output planes
    const int ncu = device_cus();
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / (lds + 1024)));
    const int tiles = q.ntx * q.nty;
    int nzc = g_affine_zchunks > 0 ? (int)g_affine_zchunks : (ncu * per_cu + tiles - 1) / tiles;
    nzc = std::max(1, std::min(nzc, (q.oS + 15) / 16));
    q.zc = (q.oS + nzc - 1) / nzc;
    q.nzc = (q.oS + q.zc - 1) / q.zc;
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)affine3d_zrect_kernel<TY, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        MI_HIP(hipFuncSetAttribute((const void *)affine3d_zrect_kernel<TY, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        MI_HIP(hipFuncSetAttribute((const void *)affine3d_zstream_kernel<TY, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        MI_HIP(hipFuncSetAttribute((const void *)affine3d_zstream_kernel<TY, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        attr_done = true;
    }
    if (q.shear)
        note_kernel("mi::affine3d_zstream_kernel<%d,%d,true> grid=%d (order-1 affine, axis %d decoupled: streams along it, %d rows x %d staged per plane, sheared, %d slots, %d chunks)",
                    TY, q.stream_axis, tiles * q.nzc, q.stream_axis, q.ry, q.P, q.nslots, q.nzc);
    else
        note_kernel("mi::affine3d_zrect_kernel<%d,%d> grid=%d (order-1 affine, axis %d decoupled: streams along it, %d rows x %d staged per plane, %d chunks)",
                    TY, q.stream_axis, tiles * q.nzc, q.stream_axis, q.ry, q.P, q.nzc);
    const dim3 grid((unsigned)(tiles * q.nzc)), block(NT);
    if (q.stream_axis == 0 && !q.shear) hipLaunchKernelGGL((affine3d_zrect_kernel<TY, 0>), grid, block, lds, s, in, out, q);
    else if (q.stream_axis == 0) hipLaunchKernelGGL((affine3d_zstream_kernel<TY, 0, true>), grid, block, lds, s, in, out, q);
    else if (!q.shear) hipLaunchKernelGGL((affine3d_zrect_kernel<TY, 1>), grid, block, lds, s, in, out, q);
    else hipLaunchKernelGGL((affine3d_zstream_kernel<TY, 1, true>), grid, block, lds, s, in, out, q);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// ---------------------------------------------------------------------------
// r4: map_coordinates, order 1, constant mode, float32 volumes -- STREAMS ALONG z with the taps out of LDS (config D).
//
// map_coords3d_c1_kernel is bound by the L1's tag pipeline (four 8-byte gathers per voxel, >= 16 cycles of the texture
// addresser per wave instruction whatever its width); the r3 attempt to stage a box per tile in LDS lost to it because
// a tile's two long-latency phases -- read the coordinates, reduce them to a box, THEN fetch the box -- ran one after the
// other.  Here a workgroup owns a (y, x) tile and walks down a chunk of output planes, and the phases of consecutive
// planes overlap:
//     step z:  coordinates of plane z + 1 (loaded during step z - 1) -> integer parts, weights, bounding box (wave
//              reductions + six LDS atomics per wave) -> the input planes / rectangle plane z + 1 needs are fetched by
//              LDS-DMA into a ring of four plane slots;  the coordinates of plane z + 2 are requested;  plane z is
//              interpolated from LDS (four ds_read2_b32 per voxel) with the splits made one step earlier.
// All resident planes share one rectangle origin (RY = 48 rows of 80 samples: a 32 x 64 tile under rotations up to ~10
// degrees with slack for the drift from plane to plane); a plane whose taps leave the rectangle re-centres it (its planes
// are fetched after the current plane's taps have been read: "late"), a plane whose taps do not fit a rectangle or four
// z slots at all -- coordinates with no structure -- takes the L1 gathers for that step.  Same splits, same in-range
// tests, same blend as the other order-1 kernels: bit-identical results.
// ---------------------------------------------------------------------------
constexpr int kMzTY = 32, kMzRY = 48;
constexpr int kMzSlotBytes = 16384;                         // RY x 20 chunks of 16 bytes, rounded up to whole rounds of 256 / 512 threads
static_assert(kMzRY * 20 * 16 <= kMzSlotBytes, "slot holds the rectangle");
// V = voxels per thread: 8 -> 256 threads (4 waves, 8 rows each), 4 -> 512 threads (8 waves, 4 rows each) per 32 x 64 tile.
// Two workgroups per CU either way (LDS), i.e. 2 or 4 waves per SIMD: with V = 4 a wave's dependent chain per plane is
// half as long and twice as many waves hide it (one workgroup per CU instead of two cost 740 against 522 us on config D:
// the kernel is bound by that chain, not by bandwidth).
template <int V> struct MzGeo {
    static constexpr int NT = 2048 / V, NW = NT / 64, ROUNDS = kMzSlotBytes / (NT * 16), BT = V / 4;
    static constexpr int STG = V == 8 ? 3072 : 2048;        // bytes of stage tile per wave: three axes of a batch at once / two, then one
};

struct MapZParams {
    FastInterpParams f;
    int zc, nzc, ntx, nty;
    int dbg;                 // 1 = no DMA, 2 = every step takes the L1 gathers, 4 = no stores, 8 = no interpolation (timing / test aids)
};

// wave-wide minimum / maximum of a float, result as a wave-uniform scalar: four DPP row shifts inside each row of 16 lanes
// (lanes without a source keep their own value), then the four row results by v_readlane
template <bool IS_MAX>
__device__ __forceinline__ float wave_reduce_f32(float v)
{
#define MI_DPP_STEP(CTRL)                                                                                              \
    {                                                                                                                  \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), (CTRL), 0xf, 0xf, false)); \
        v = IS_MAX ? fmaxf(v, o) : fminf(v, o);                                                                        \
    }
    MI_DPP_STEP(0x111) MI_DPP_STEP(0x112) MI_DPP_STEP(0x114) MI_DPP_STEP(0x118)          // row_shr:1, 2, 4, 8: lane 15 of a row holds the row's result
#undef MI_DPP_STEP
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return IS_MAX ? fmaxf(fmaxf(r0, r1), fmaxf(r2, r3)) : fminf(fminf(r0, r1), fminf(r2, r3));
}

// r4b: the coordinates of up to TWO planes are in flight per wave, in ACCUMULATION registers: with 8 voxels per thread set 0
// = a[0:23] + a24 (the corner sample), set 1 = a[26:49] + a50; with 4 voxels per thread a[0:11] + a12 and a[14:25] + a26.  A compiler-visible load let only one plane travel (the compiler's counter logic
// drained vmcnt to zero before the first use of a plane), loads issued by inline asm into ordinary registers were copied
// by the register allocator while still in flight (loop-carried values), and LDS has no room for two planes of
// coordinates per wave beside the ring.  AGPRs are outside the allocator's reach: global_load writes them, ds_write_b128
// reads them (gfx90a+ take AGPRs as the data operand of memory instructions), nothing is copied, and the hand-counted
// s_waitcnt at the top of a step is the only thing between the two.  The build refuses this file if the compiler itself
// touches an AGPR in these kernels (_build.py NO_AGPR_MOVES: a VGPR spill into a[..] could land under an in-flight load).
template <int V, int SET, int IDX> __device__ __forceinline__ void mz_coord_load(const float *addr);
template <int V, int SET, int IDX> __device__ __forceinline__ void mz_coord_stage(unsigned lds_addr);      // 16 bytes per lane to lds_addr + the offset of (V, IDX)
template <int V, int SET> __device__ __forceinline__ void mz_corner_load(const float *addr);
template <int V, int SET> __device__ __forceinline__ float mz_corner_get();
#define MI_MZ_COORD(V, SET, IDX, R0, R1, R2, R3, OFF)                                                                  \
    template <> __device__ __forceinline__ void mz_coord_load<V, SET, IDX>(const float *addr)                          \
    {                                                                                                                  \
        asm volatile("global_load_dwordx4 a[" #R0 ":" #R3 "], %0, off nt" ::"v"(addr) : "memory", "a" #R0, "a" #R1, "a" #R2, "a" #R3); \
    }                                                                                                                  \
    template <> __device__ __forceinline__ void mz_coord_stage<V, SET, IDX>(unsigned lds_addr)                         \
    {                                                                                                                  \
        asm volatile("ds_write_b128 %0, a[" #R0 ":" #R3 "] offset:" #OFF ::"v"(lds_addr) : "memory");                   \
    }
#define MI_MZ_CORNER(V, SET, R)                                                                                        \
    template <> __device__ __forceinline__ void mz_corner_load<V, SET>(const float *addr)                              \
    {                                                                                                                  \
        asm volatile("global_load_dword a" #R ", %0, off" ::"v"(addr) : "memory", "a" #R);                             \
    }                                                                                                                  \
    template <> __device__ __forceinline__ float mz_corner_get<V, SET>()                                               \
    {                                                                                                                  \
        float v;                                                                                                       \
        asm volatile("v_accvgpr_read_b32 %0, a" #R : "=v"(v)::"memory");                                               \
        return v;                                                                                                      \
    }
// V = 8: IDX = 3 batch + axis, the three axes of a batch staged side by side
MI_MZ_COORD(8, 0, 0, 0, 1, 2, 3, 0)        MI_MZ_COORD(8, 0, 1, 4, 5, 6, 7, 1024)     MI_MZ_COORD(8, 0, 2, 8, 9, 10, 11, 2048)
MI_MZ_COORD(8, 0, 3, 12, 13, 14, 15, 0)    MI_MZ_COORD(8, 0, 4, 16, 17, 18, 19, 1024) MI_MZ_COORD(8, 0, 5, 20, 21, 22, 23, 2048)
MI_MZ_COORD(8, 1, 0, 26, 27, 28, 29, 0)    MI_MZ_COORD(8, 1, 1, 30, 31, 32, 33, 1024) MI_MZ_COORD(8, 1, 2, 34, 35, 36, 37, 2048)
MI_MZ_COORD(8, 1, 3, 38, 39, 40, 41, 0)    MI_MZ_COORD(8, 1, 4, 42, 43, 44, 45, 1024) MI_MZ_COORD(8, 1, 5, 46, 47, 48, 49, 2048)
MI_MZ_CORNER(8, 0, 24)
MI_MZ_CORNER(8, 1, 50)
// V = 4: IDX = axis; z and y staged side by side, x afterwards in the place of z (2 KiB of stage tile per wave)
MI_MZ_COORD(4, 0, 0, 0, 1, 2, 3, 0)        MI_MZ_COORD(4, 0, 1, 4, 5, 6, 7, 1024)     MI_MZ_COORD(4, 0, 2, 8, 9, 10, 11, 0)
MI_MZ_COORD(4, 1, 0, 14, 15, 16, 17, 0)    MI_MZ_COORD(4, 1, 1, 18, 19, 20, 21, 1024) MI_MZ_COORD(4, 1, 2, 22, 23, 24, 25, 0)
MI_MZ_CORNER(4, 0, 12)
MI_MZ_CORNER(4, 1, 26)
template <int N> __device__ __forceinline__ void mz_wait_vm()
{
    static_assert(N >= 0 && N <= 9, "s_waitcnt strings below");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 1) asm volatile(MI_VMCNT(1) " lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile(MI_VMCNT(2) " lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 3) asm volatile(MI_VMCNT(3) " lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile(MI_VMCNT(4) " lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 5) asm volatile(MI_VMCNT(5) " lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile(MI_VMCNT(6) " lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 7) asm volatile(MI_VMCNT(7) " lgkmcnt(0)" ::: "memory");
    else if constexpr (N == 8) asm volatile(MI_VMCNT(8) " lgkmcnt(0)" ::: "memory");
    else asm volatile(MI_VMCNT(9) " lgkmcnt(0)" ::: "memory");
}
#undef MI_MZ_COORD
#undef MI_MZ_CORNER
template <int N, typename F> __device__ __forceinline__ void mz_static_for(F &&f)
{
    if constexpr (N > 0) {
        mz_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// CORNER (r4b): the rectangle of a plane is placed from the coordinates of the tile's four corner voxels (every wave loads
// the same twelve floats and reduces them with v_readlane + scalar min / max: no wave reductions over all voxels, no LDS
// atomics, no second barrier per plane), and every voxel VERIFIES that its taps lie inside what was staged; a voxel that
// does not (warps that bulge inside the tile, noise) takes four L1 gathers for itself.  For affine-like coordinates the
// corner box is the exact box.  false = the exact reduction over all voxels (kept as the comparator, knob 3).
// V: voxels per thread (MzGeo); DEEP: planes of coordinates in flight per wave (1 or 2).
template <bool CORNER, int V, int DEEP>
__global__ void __launch_bounds__(2048 / V)
map_coords3d_zstream_kernel(const float *__restrict__ in, const float *__restrict__ coords, float *__restrict__ out, const MapZParams q)
{
    using G = MzGeo<V>;
    constexpr int P = kZsP, RY = kMzRY, NT = G::NT, TY = kMzTY, kRounds = G::ROUNDS, BT = G::BT;
    static_assert(kMzSlotBytes == 16384, "the ring is addressed by bit arithmetic: slot = address bits 14 .. 15");
    static_assert((V == 4 || V == 8) && (DEEP == 1 || DEEP == 2) && (CORNER || V == 8), "instances");
    extern __shared__ __attribute__((aligned(16))) char smem_mz[];
    float *stage = reinterpret_cast<float *>(smem_mz + 4 * kMzSlotBytes);            // [waves][STG bytes]: coordinates in, results out
    int *ctl = reinterpret_cast<int *>(smem_mz + 4 * kMzSlotBytes + G::NW * G::STG); // !CORNER: [2][8]: min z, y, x, max z, y, x (order-preserving ints)

    const int nz = q.f.nz, ny = q.f.ny, nx = q.f.nx, oz = q.f.oz, oy = q.f.oy, ox = q.f.ox;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = q.ntx * q.nty * q.nzc;
    int t = blockIdx.x;
    if ((total & 7) == 0 && !(q.dbg & 64)) t = (t & 7) * (total >> 3) + (t >> 3);          // x-neighbouring tiles on one XCD
    const int tx_i = t % q.ntx, ty_i = (t / q.ntx) % q.nty, zc_i = t / (q.ntx * q.nty);
    const int x0 = tx_i * 64, y0 = ty_i * TY;
    const int zs = zc_i * q.zc, ze = min(zs + q.zc, oz);
    const size_t nout = (size_t)oz * oy * ox;
    const bool wide = x0 + 64 <= ox && y0 + TY <= oy;
    const int vol_bytes = nz * ny * nx * 4;
    const unsigned plane_b = (unsigned)ny * (unsigned)nx * 4u, row_b = (unsigned)nx * 4u;
    const float cval = (float)q.f.cval;
    float *my_stage = stage + wave * (G::STG / 4);
    // a partial tile's last lanes read their own column of the clamped 16-byte coordinate load
    const int col = wide ? lane : min(x0 + lane, ox - 1) - min(x0 + 4 * (lane >> 2), ox - 4) + 4 * (lane >> 2);

    unsigned rel[kRounds];
#pragma unroll
    for (int j = 0; j < kRounds; j++) {
        const unsigned ch = (unsigned)tid + (unsigned)(j * NT);
        const unsigned row = ch / 20u, c4 = ch - row * 20u;
        rel[j] = ch < (unsigned)(RY * 20) ? row * row_b + c4 * 16u : 0x80000000u;
    }
    if constexpr (!CORNER) { if (tid < 16) ctl[tid] = (tid & 7) < 3 ? 0x7fffffff : 0; }     // both sets: minima, then maxima (all values are >= 0)
    // CORNER: lane -> (axis = (lane / 4) % 3, corner = lane % 4) of the tile's corner voxels (clamped into the output)
    const int c_axis = (lane >> 2) % 3;
    const size_t c_off = (size_t)c_axis * ((size_t)oz * oy * ox) +
                         (size_t)((lane & 2) ? min(y0 + TY - 1, oy - 1) : y0) * ox + (size_t)((lane & 1) ? min(x0 + 63, ox - 1) : x0);
    const float c_max = (float)((c_axis == 0 ? nz : c_axis == 1 ? ny : nx) - 1);

    // ---- coordinates of one output plane's tile: 16-byte loads (lane -> row lane / 16 of a batch of four, x = 4 (lane % 16))
    // into AGPR set SET (see mz_coord_load); the wait counts are derived at the top of `step`
    auto load_coords = [&](int z, auto SS) {
        constexpr int SET = decltype(SS)::value;
        const int i = lane >> 4, c = lane & 15;
        if constexpr (CORNER) mz_corner_load<V, SET>(coords + c_off + (size_t)z * oy * ox);
        mz_static_for<BT>([&](auto BB) {
            constexpr int bt = decltype(BB)::value;
            const int y = min(y0 + V * wave + 4 * bt + i, oy - 1);
            const int xq = min(x0 + 4 * c, ox - 4);
            const size_t o = ((size_t)z * oy + y) * ox + xq;
            mz_static_for<3>([&](auto AA) {
                constexpr int a = decltype(AA)::value;
                mz_coord_load<V, SET, 3 * bt + a>(coords + a * nout + o);
            });
        });
    };
    // what a plane's voxels need: integer parts until the rectangle is known, then ONE byte offset (LDS: slot in bits 14-15;
    // L1 path: offset into the volume), weights, in-range bits
    // CORNER: bit k of `bad` = voxel k's taps are not (all) in LDS; a[k] is then 0x80000000 | its byte offset in the volume
    struct Step { int a[V]; float wz[V], wy[V], wx[V]; unsigned inmask; int lds; unsigned bad; };

    // resident planes [rlo, rhi] (<= 4, slot = plane & 3), all staged with the rectangle origin (org_y, org_x)
    int rlo = 0, rhi = -1, org_y = 0, org_x = 0;
    auto ensure = [&](int pl) {
        if (pl < 0 || pl >= nz) return;
        if (pl >= rlo && pl <= rhi) return;
        if (pl == rhi + 1 && rhi >= rlo) { rhi = pl; if (rhi - rlo > 3) rlo = rhi - 3; }
        else if (pl == rlo - 1 && rhi >= rlo) { rlo = pl; if (rhi - rlo > 3) rhi = rlo + 3; }
        else { rlo = rhi = pl; }
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
        const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)pl * plane_b + (unsigned)org_y * row_b + (unsigned)org_x * 4u);
        const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(pl & 3) * (unsigned)kMzSlotBytes + (unsigned)(wave << 6) * 16u);
#pragma unroll
        for (int j = 0; j < kRounds; j++)
            if (!(q.dbg & 1)) dma_16s(rin, rel[j], base, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(j * NT) * 16u));
    };

    // ---- phase A of a plane: raw coordinates -> integer parts / weights (into `st`) + the workgroup's bounding box of the
    // lower tap corners.  The box is reduced on the COORDINATES (v_min3 / v_max3, DPP, one pair of LDS atomics per axis and
    // wave) and clamped into the volume afterwards: a coordinate outside only stretches the box outwards, which the clamp
    // removes again; NaNs are ignored by min / max (their voxels blend to NaN whatever they read).
    int need_lo[3], need_hi[3];
    int iz_[V], iy_[V], ix_[V];
    // byte address of this lane's 16 bytes in the wave's stage tile (dynamic shared memory starts at LDS address 0: the
    // kernel has no static __shared__ data, as the M0 arithmetic of the DMAs assumes already)
    const unsigned stage_lds = (unsigned)(4 * kMzSlotBytes) + (unsigned)wave * (unsigned)G::STG + (unsigned)lane * 16u;
    auto analyse = [&](auto SS, Step &st, int set) {
        constexpr int SET = decltype(SS)::value;
        st.inmask = 0;
        float cz[V], cy[V], cx[V];
        // (axis a, row i = lane / 16, x = 4 (lane % 16)) -> float (a * 4 + i) * 64 + 4 (lane % 16) = a KiB + 16 lane bytes
        if constexpr (V == 8) {
            mz_static_for<2>([&](auto BB) {
                constexpr int bt = decltype(BB)::value;
                mz_coord_stage<V, SET, 3 * bt>(stage_lds);
                mz_coord_stage<V, SET, 3 * bt + 1>(stage_lds);
                mz_coord_stage<V, SET, 3 * bt + 2>(stage_lds);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    cz[4 * bt + kk] = my_stage[(0 * 4 + kk) * 64 + col];
                    cy[4 * bt + kk] = my_stage[(1 * 4 + kk) * 64 + col];
                    cx[4 * bt + kk] = my_stage[(2 * 4 + kk) * 64 + col];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            });
        } else {
            mz_coord_stage<V, SET, 0>(stage_lds);
            mz_coord_stage<V, SET, 1>(stage_lds);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                cz[kk] = my_stage[kk * 64 + col];
                cy[kk] = my_stage[(4 + kk) * 64 + col];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            mz_coord_stage<V, SET, 2>(stage_lds);                    // x where z was (a wave's LDS operations are carried out in order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < 4; kk++) cx[kk] = my_stage[kk * 64 + col];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < V; k++) {
            const C1Split sz = c1_split(cz[k], nz), sy = c1_split(cy[k], ny), sx = c1_split(cx[k], nx);
            iz_[k] = sz.i0; iy_[k] = sy.i0; ix_[k] = sx.i0;
            st.wz[k] = sz.w1; st.wy[k] = sy.w1; st.wx[k] = sx.w1;
            st.inmask |= (sz.in & sy.in & sx.in) ? (1u << k) : 0u;
        }
        if constexpr (CORNER) {
            // lower tap corner of the four corner voxels, clamped into the volume (a NaN clamps to 0): lanes 4 a .. 4 a + 3
            const int ci = (int)fminf(fmaxf(mz_corner_get<V, SET>(), 0.f), c_max);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const int v0 = __builtin_amdgcn_readlane(ci, 4 * a), v1 = __builtin_amdgcn_readlane(ci, 4 * a + 1);
                const int v2 = __builtin_amdgcn_readlane(ci, 4 * a + 2), v3 = __builtin_amdgcn_readlane(ci, 4 * a + 3);
                need_lo[a] = min(min(v0, v1), min(v2, v3));
                need_hi[a] = max(max(v0, v1), max(v2, v3));
            }
        } else if constexpr (V == 8) {
            auto lo8 = [](const float (&v)[8]) { return fminf(__builtin_fminf(__builtin_fminf(v[0], v[1]), v[2]) < __builtin_fminf(__builtin_fminf(v[3], v[4]), v[5]) ? __builtin_fminf(__builtin_fminf(v[0], v[1]), v[2]) : __builtin_fminf(__builtin_fminf(v[3], v[4]), v[5]), fminf(v[6], v[7])); };
            auto hi8 = [](const float (&v)[8]) { return fmaxf(__builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]) > __builtin_fmaxf(__builtin_fmaxf(v[3], v[4]), v[5]) ? __builtin_fmaxf(__builtin_fmaxf(v[0], v[1]), v[2]) : __builtin_fmaxf(__builtin_fmaxf(v[3], v[4]), v[5]), fmaxf(v[6], v[7])); };
            float lo[3] = {lo8(cz), lo8(cy), lo8(cx)}, hi[3] = {hi8(cz), hi8(cy), hi8(cx)};
            int *cs = ctl + 8 * set;
            const int nn[3] = {nz, ny, nx};
#pragma unroll
            for (int a = 0; a < 3; a++) {
                // clamped to [0, n - 1] and floored: the lower tap corner of the extreme coordinates (ints >= 0)
                float l = wave_reduce_f32<false>(lo[a]), h = wave_reduce_f32<true>(hi[a]);
                l = fminf(fmaxf(l, 0.f), (float)(nn[a] - 1));
                h = fminf(fmaxf(h, 0.f), (float)(nn[a] - 1));
                if (lane == 0) { atomicMin(&cs[a], (int)l); atomicMax(&cs[3 + a], (int)h); }
            }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 3; a++) {
                need_lo[a] = __builtin_amdgcn_readfirstlane(cs[a]);
                need_hi[a] = __builtin_amdgcn_readfirstlane(cs[3 + a]);
            }
            // re-arm the OTHER set: its readers finished a step ago, its next atomics come after the next top barrier
            if (tid < 6) ctl[8 * (set ^ 1) + tid] = tid < 3 ? 0x7fffffff : 0;
        }
    };

    // ---- phase B: where do this plane's taps come from?  Returns 0 = LDS, planes fetched now; 1 = LDS, planes fetched after
    // the current plane has been read (re-centred rectangle, or its planes would land on slots in use); 2 = L1 gathers
    auto place = [&](int cur_zlo, int cur_zhi, bool cur_lds) {
        if (q.dbg & 2) return 2;
        const int nyy = need_hi[1] + 2 - need_lo[1], nxx = need_hi[2] + 2 - need_lo[2], nzz = need_hi[0] + 2 - need_lo[0];
        if (nyy > RY || nxx > P - 3 || nzz > 4) return 2;
        const bool covered = need_lo[1] >= org_y && need_hi[1] + 1 < org_y + RY && need_lo[2] >= org_x && need_hi[2] + 1 < org_x + P;
        bool late = !covered || rhi < rlo;
        if (late) {
            int oyy = need_lo[1] - (RY - nyy) / 2, oxx = (need_lo[2] - (P - nxx) / 2) & ~3;
            org_y = oyy < 0 ? 0 : oyy;
            org_x = oxx < 0 ? 0 : oxx;
            rlo = 0; rhi = -1;                                                    // the resident planes have the old origin
        }
        if (cur_lds && cur_zhi >= cur_zlo) {
            const int ulo = min(cur_zlo, need_lo[0]), uhi = max(cur_zhi, need_hi[0] + 1);
            if (uhi - ulo > 3) late = true;                                       // would overwrite a slot the current plane reads
        }
        return late ? 1 : 0;
    };
    auto fetch = [&]() {
        for (int pl = need_lo[0]; pl <= need_hi[0] + 1; pl++) ensure(pl);
    };
    auto addresses = [&](int how, Step &st) {
        st.lds = how != 2 ? 1 : 0;
        st.bad = 0;
        if (how != 2) {
            const int org = org_y * P + org_x;
            if constexpr (CORNER) {
                // a voxel's taps are in LDS when its lower tap corner lies in [org, org + (RY - 2, P - 2)] and in the planes
                // staged for this step except the last; one whose do not takes the L1 gathers for itself (offset into the
                // volume, flagged by the sign bit).  Anything not in range (cval whatever it reads) reads offset 0.
                const int zl = need_lo[0];
                const unsigned zn = (unsigned)(need_hi[0] + 1 - need_lo[0]);
#pragma unroll
                for (int k = 0; k < V; k++) {
                    const bool in = (st.inmask >> k) & 1u;
                    const bool here = ((unsigned)(iy_[k] - org_y) <= (unsigned)(RY - 2)) & ((unsigned)(ix_[k] - org_x) <= (unsigned)(P - 2)) &
                                      ((unsigned)(iz_[k] - zl) < zn);
                    const int off = (iy_[k] * P + ix_[k] - org) * 4;
                    st.a[k] = (in & here) ? (((iz_[k] & 3) << 14) | off) : 0;
                    st.bad |= (in & !here) ? (1u << k) : 0u;
                }
                if (__builtin_amdgcn_ballot_w64(st.bad != 0) != 0) {                      // rare: the offsets into the volume only then
#pragma unroll
                    for (int k = 0; k < V; k++)
                        if ((st.bad >> k) & 1u) st.a[k] = (int)(0x80000000u | (unsigned)(((iz_[k] * ny + iy_[k]) * nx + ix_[k]) * 4));
                }
            } else {
                // in-range taps lie inside the rectangle by construction; anything else (a NaN coordinate: "inside", integer
                // part 0) reads offset 0 -- its value does not matter (NaN weights) but its address must exist
#pragma unroll
                for (int k = 0; k < V; k++) {
                    const int off = (iy_[k] * P + ix_[k] - org) * 4;
                    const bool ok = ((st.inmask >> k) & 1u) && (unsigned)off < (unsigned)(RY * P * 4 - 4 * (P + 1));
                    st.a[k] = ok ? (((iz_[k] & 3) << 14) | off) : 0;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; k++) st.a[k] = ((st.inmask >> k) & 1u) ? ((iz_[k] * ny + iy_[k]) * nx + ix_[k]) * 4 : 0;
        }
    };

    // ---- prologue: plane zs analysed and fetched, coordinates of the next DEEP planes requested
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, DEEP - 1> S1{};                         // DEEP == 1: one set
    constexpr int kLoads = 3 * BT + (CORNER ? 1 : 0), kStores = BT;               // vector-memory operations per plane and wave
    Step stA, stB;
    load_coords(zs, S0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                              // ctl initialised
    analyse(S0, stA, 0);
    int how = place(0, -1, false);
    if (how != 2) fetch();
    addresses(how, stA);
    int cur_zlo = need_lo[0], cur_zhi = need_hi[0] + 1;
    if (zs + 1 < ze) load_coords(zs + 1, S1);
    if constexpr (DEEP == 2) { if (zs + 2 < ze) load_coords(zs + 2, S0); }

    // one step; `cur` / `nxt` alternate between stA and stB and (DEEP == 2) the coordinates of plane z + 1 -- requested two
    // steps ago; the set is re-loaded with plane z + 3 -- between AGPR sets 1 and 0: two copies of the body (handing `nxt`
    // over by assignment cost 34 register moves per thread and plane)
    auto step = [&](const int z, Step &cur, Step &nxt, auto SS) {
        // The planes of this step (DMAs issued in the previous step BEFORE its coordinate loads) have landed and the
        // coordinates of plane z + 1 are here.  DEEP == 2: the coordinate loads of plane z + 2 (kLoads) and the stores of the
        // previous step (kStores), issued after them, may still be in flight on full tiles; DEEP == 1: the stores.  The
        // first step has no stores behind it (and a two-plane chunk no younger loads).  Everyone has finished reading the
        // previous planes.
        if constexpr (DEEP == 2) {
            if (wide && z > zs) {
                if (z + 2 < ze) mz_wait_vm<kLoads + kStores>();
                else mz_wait_vm<kStores>();
            } else if (wide && z + 2 < ze) {
                mz_wait_vm<3 * BT>();
            } else {
                mz_wait_vm<0>();
            }
        } else {
            if (wide && z > zs) mz_wait_vm<kStores>();
            else mz_wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        int nhow = 0, nzlo = 0, nzhi = -1;
        const bool more = z + 1 < ze;
        if (more) {
            analyse(SS, nxt, (z + 1 - zs) & 1);
            nhow = place(cur_zlo, cur_zhi, cur.lds != 0);
            if (nhow == 0) fetch();
            addresses(nhow, nxt);
            nzlo = need_lo[0]; nzhi = need_hi[0] + 1;
            if (z + 1 + DEEP < ze) load_coords(z + 1 + DEEP, SS);
        }
        // ---- interpolate plane z
        float r[V];
        if (q.dbg & 8) {
#pragma unroll
            for (int k = 0; k < V; k++) r[k] = cur.wz[k] + (float)cur.a[k];
        } else if (CORNER && cur.lds && __builtin_amdgcn_ballot_w64(cur.bad != 0) != 0) {
            // some voxel of this wave verifies to "not staged": the same loop with a per-voxel choice of the source
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
#pragma unroll
            for (int k = 0; k < V; k++) {                                                 // unrolled: a run-time index would put `cur` into scratch memory
                Taps<float> t;
                const int a_lo = cur.a[k];
                if (a_lo < 0) {
                    const unsigned base = (unsigned)a_lo & 0x7fffffffu;
#pragma unroll
                    for (int m = 0; m < 4; m++) load_pair(rin, base + (m >> 1) * plane_b + (m & 1) * row_b, t.v[2 * m], t.v[2 * m + 1]);
                } else {
                    const int a_hi = (a_lo + kMzSlotBytes) & 0xFFFF;
                    const float *A = reinterpret_cast<const float *>(smem_mz + a_lo);
                    const float *B = reinterpret_cast<const float *>(smem_mz + a_hi);
                    t.v[0] = A[0]; t.v[1] = A[1]; t.v[2] = A[P]; t.v[3] = A[P + 1];
                    t.v[4] = B[0]; t.v[5] = B[1]; t.v[6] = B[P]; t.v[7] = B[P + 1];
                }
                t.wz1 = cur.wz[k]; t.wy1 = cur.wy[k]; t.wx1 = cur.wx[k];
                t.oobmask = 0;
                t.outside = !((cur.inmask >> k) & 1u);
                r[k] = finish<float>(t, cval);
            }
        } else if (cur.lds) {
#pragma unroll
            for (int k = 0; k < V; k++) {
                const int a_lo = cur.a[k];
                const int a_hi = (a_lo + kMzSlotBytes) & 0xFFFF;                         // plane iz + 1 lives in slot (iz + 1) & 3
                const float *A = reinterpret_cast<const float *>(smem_mz + a_lo);
                const float *B = reinterpret_cast<const float *>(smem_mz + a_hi);
                Taps<float> t;
                t.v[0] = A[0]; t.v[1] = A[1]; t.v[2] = A[P]; t.v[3] = A[P + 1];
                t.v[4] = B[0]; t.v[5] = B[1]; t.v[6] = B[P]; t.v[7] = B[P + 1];
                t.wz1 = cur.wz[k]; t.wy1 = cur.wy[k]; t.wx1 = cur.wx[k];
                t.oobmask = 0;
                t.outside = !((cur.inmask >> k) & 1u);
                r[k] = finish<float>(t, cval);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, vol_bytes, 0x00020000);
#pragma unroll
            for (int h = 0; h < BT; h++) {
                Taps<float> t[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const int k = 4 * h + kk;
                    const unsigned base = (unsigned)cur.a[k];
#pragma unroll
                    for (int m = 0; m < 4; m++) load_pair(rin, base + (m >> 1) * plane_b + (m & 1) * row_b, t[kk].v[2 * m], t[kk].v[2 * m + 1]);
                    t[kk].wz1 = cur.wz[k]; t[kk].wy1 = cur.wy[k]; t[kk].wx1 = cur.wx[k];
                    t[kk].oobmask = 0;
                    t[kk].outside = !((cur.inmask >> k) & 1u);
                }
#pragma unroll
                for (int kk = 0; kk < 4; kk++) r[4 * h + kk] = finish<float>(t[kk], cval);
            }
        }
        if (q.dbg & 4) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < V; k++) sum += r[k];
            if (sum == 1.2345e-30f) out[0] = r[0];                                      // keeps the work alive
        } else if (wide) {
#pragma unroll
            for (int k = 0; k < V; k++) my_stage[k * 64 + lane] = r[k];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int i = lane >> 4, c = lane & 15;
#pragma unroll
            for (int h = 0; h < BT; h++) {
                const f32x4n v = *reinterpret_cast<const f32x4n *>(my_stage + (4 * h + i) * 64 + 4 * c);
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4n *>(out + ((size_t)z * oy + (y0 + V * wave + 4 * h + i)) * ox + x0 + 4 * c));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            const int x = x0 + lane;
#pragma unroll
            for (int k = 0; k < V; k++) {
                const int y = y0 + V * wave + k;
                if (x < ox && y < oy) __builtin_nontemporal_store(r[k], out + ((size_t)z * oy + y) * ox + x);
            }
        }
        if (more && nhow == 1) {
            // late fetch: nobody reads the old planes any more
            __builtin_amdgcn_s_barrier();
            fetch();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        cur_zlo = nzlo; cur_zhi = nzhi;
    };
#pragma unroll 1
    for (int z = zs; z < ze; z += 2) {
        step(z, stA, stB, S1);
        if (z + 1 < ze) step(z + 1, stB, stA, S0);                                // DEEP == 1: S1 is S0
    }
}

// ---------------------------------------------------------------------------
// r4: affine_transform, order 1, constant mode, float32 volumes whose matrix leaves the x axis to itself with unit step and
// an integral shift -- a rotation / shear / scaling in the (z, y) plane: `rotate(volume, angle)` with SciPy's default axes
// (1, 0).  Along x every coordinate is an integer, so the x interpolation degenerates (weight 0) and an output ROW is the
// blend of four input rows with weights that are the same for the whole row:
//     out[z, y, x] = lerp_z( lerp_y(in[iz, iy, x + s], in[iz, iy + 1, x + s]), lerp_y(in[iz + 1, iy, ..], in[iz + 1, iy + 1, ..]) )
// No gathers, no LDS: four coalesced 16-byte loads and one 16-byte store per four voxels; the rows a workgroup's 16 output
// rows need overlap and come out of the L1 / L2.  Same coordinate arithmetic (the oracle's summation order; the x terms
// add exact zeros), same in-range tests, same blend as the other order-1 kernels (finish() with the upper x tap equal to
// the lower one and weight 0): bit-identical results.  Was: the LDS box kernel, 266 us on 512^3 at 7 degrees.
// ---------------------------------------------------------------------------
struct RowBlendParams {
    FastInterpParams f;
    int xshift;              // cx = x + xshift
    int ntx, nty, ntz;       // tiles of 256 x, 8 y, 2 z
};

__global__ void __launch_bounds__(256)
affine3d_rowblend_kernel(const float *__restrict__ in, float *__restrict__ out, const RowBlendParams q)
{
    const FastInterpParams &p = q.f;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int total = q.ntx * q.nty * q.ntz;
    int t = blockIdx.x;
    if ((total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);          // y- / z-neighbouring blocks (shared input rows) on one XCD
    const int tx = t % q.ntx, ty = (t / q.ntx) % q.nty, tz = t / (q.ntx * q.nty);
    const int x = tx * 256 + 4 * lane;
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, p.nz * p.ny * p.nx * 4, 0x00020000);
    const unsigned plane_b = (unsigned)p.ny * (unsigned)p.nx * 4u, row_b = (unsigned)p.nx * 4u;
    const float cval = (float)p.cval;
    // the four voxels of a lane along x: source column x + e + xshift, inside iff 0 <= it <= nx - 1 (coordinate = integer)
    const int sx0 = x + q.xshift;
    bool inx[4];
#pragma unroll
    for (int e = 0; e < 4; e++) inx[e] = sx0 + e >= 0 && sx0 + e <= p.nx - 1;
    const bool any_x = (inx[0] | inx[1] | inx[2] | inx[3]) && x < p.ox;
    // a 16-byte load at column sx0 is in range of the row when all four columns are; otherwise four dword loads
    const bool whole = inx[0] & inx[3];

    struct Row { int z, y; bool live, in; unsigned base; float wz, wy; };
    Row rw[4];
    typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));
    u32x4r v[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int wr = 4 * wave + r;                                  // 16 rows per workgroup: 2 planes x 8 rows
        Row &R = rw[r];
        R.z = tz * 2 + (wr >> 3);
        R.y = ty * 8 + (wr & 7);
        R.live = R.z < p.oz && R.y < p.oy;
        const double dz = (double)R.z, dy = (double)R.y;
        // ((m0 z + m1 y) + m2 x) + offset with m2 = 0
        const C1Split sz = c1_split((p.m[0] * dz + p.m[1] * dy) + p.m[3], p.nz);
        const C1Split sy = c1_split((p.m[4] * dz + p.m[5] * dy) + p.m[7], p.ny);
        R.in = sz.in & sy.in;
        R.wz = sz.w1; R.wy = sy.w1;
        R.base = (R.in && R.live) ? (unsigned)((sz.i0 * p.ny + sy.i0) * p.nx) * 4u : 0x80000000u;
        const unsigned col = (unsigned)(sx0 * 4);
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const unsigned rb = R.base + (m >> 1) * plane_b + (m & 1) * row_b;
            if (whole) {
                v[r][m] = __builtin_amdgcn_raw_buffer_load_b128(rin, (R.base & 0x80000000u) ? 0x80000000u : rb + col, 0, 0);
            } else {
                unsigned e4[4];
#pragma unroll
                for (int e = 0; e < 4; e++)
                    e4[e] = __builtin_amdgcn_raw_buffer_load_b32(rin, (inx[e] && !(R.base & 0x80000000u)) ? rb + col + 4u * e : 0x80000000u, 0, 0);
                v[r][m] = (u32x4r){e4[0], e4[1], e4[2], e4[3]};
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const Row &R = rw[r];
        if (!R.live || x >= p.ox) continue;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            Taps<float> tp;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const unsigned w = e == 0 ? v[r][m].x : (e == 1 ? v[r][m].y : (e == 2 ? v[r][m].z : v[r][m].w));
                tp.v[2 * m] = tp.v[2 * m + 1] = __uint_as_float(w);       // upper x tap: weight 0
            }
            tp.wz1 = R.wz; tp.wy1 = R.wy; tp.wx1 = 0.f;
            tp.oobmask = 0;
            tp.outside = !(R.in && inx[e]);
            o[e] = finish<float>(tp, cval);
        }
        float *dst = out + ((size_t)R.z * p.oy + R.y) * p.ox + x;
        if (x + 4 <= p.ox) __builtin_nontemporal_store((f32x4n){o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4n *>(dst));
        else
            for (int e = 0; e < 4 && x + e < p.ox; e++) dst[e] = o[e];
    }
    (void)any_x;
}

Knob g_affine_rowblend{1};    // test hook: 0 = off

static bool rowblend_plan(const FastInterpParams &p, RowBlendParams *q)
{
    const double *m = p.m;
    if (m[2] != 0.0 || m[6] != 0.0 || m[8] != 0.0 || m[9] != 0.0 || m[10] != 1.0) return false;      // x left to itself, unit step
    if (!(fabs(m[11]) < 16777216.0) || m[11] != floor(m[11])) return false;                          // integral shift
    for (int i = 0; i < 8; i++) if (!(fabs(m[i]) < 1e9)) return false;
    q->f = p;
    q->xshift = (int)m[11];
    q->ntx = (p.ox + 255) / 256;
    q->nty = (p.oy + 7) / 8;
    q->ntz = (p.oz + 1) / 2;
    return (long long)q->ntx * q->nty * q->ntz < (1ll << 31);
}

static int launch_affine_rowblend(const float *in, float *out, const RowBlendParams &q, hipStream_t s)
{
    const int total = q.ntx * q.nty * q.ntz;
    note_kernel("mi::affine3d_rowblend_kernel grid=%d (order-1 affine, x axis untouched: an output row = the blend of four input rows)", total);
    hipLaunchKernelGGL(affine3d_rowblend_kernel, dim3((unsigned)total), dim3(256), 0, s, in, out, q);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

Knob g_map_zstream{1};        // test hook: 0 = off (L1-gather kernel), 1 = on, 2 = on with every step on the L1 gathers, 3 = on with the exact box reduction (the first r4 kernel)
Knob g_map_zchunks{0};

Knob g_map_zvariant{0};       // test hook: 10 V + DEEP of the kernel instance (81, 82, 41); 0 = default.  <true, 4, 2> needs 139 registers: one workgroup per CU
template <bool CORNER, int V, int DEEP>
static int launch_map_zstream_as(const float *in, const float *coords, float *out, MapZParams &q, int tiles, hipStream_t s)
{
    using G = MzGeo<V>;
    size_t lds = 4 * (size_t)kMzSlotBytes + (size_t)G::NW * G::STG + (CORNER ? 0 : 64);
    if (g_affine_dbg & 32) lds = 96 * 1024;                 // occupancy experiment: one workgroup per CU
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)map_coords3d_zstream_kernel<CORNER, V, DEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        attr_done = true;
    }
    note_kernel("mi::map_coords3d_zstream_kernel<%s,%d,%d> grid=%d (order-1 map_coordinates: streams along z, taps out of LDS, rectangle from %s, %d z chunks)",
                CORNER ? "true" : "false", V, DEEP, tiles * q.nzc, CORNER ? "the tile corners + per-voxel check" : "the exact box", q.nzc);
    hipLaunchKernelGGL((map_coords3d_zstream_kernel<CORNER, V, DEEP>), dim3((unsigned)(tiles * q.nzc)), dim3(G::NT), lds, s, in, coords, out, q);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

static int launch_map_zstream(const float *in, const float *coords, float *out, const FastInterpParams &p, hipStream_t s)
{
    MapZParams q;
    q.f = p;
    q.ntx = (p.ox + 63) / 64;
    q.nty = (p.oy + kMzTY - 1) / kMzTY;
    const int ncu = device_cus();
    const int tiles = q.ntx * q.nty;
    int nzc = g_map_zchunks > 0 ? (int)g_map_zchunks : (8 * ncu + tiles - 1) / tiles;      // two workgroups per CU, four rounds (measured on config D: 501 against 511 us with two)
    nzc = std::max(1, std::min(nzc, (p.oz + 15) / 16));
    q.zc = (p.oz + nzc - 1) / nzc;
    q.nzc = (p.oz + q.zc - 1) / q.zc;
    q.dbg = (g_affine_dbg & (13 | 64)) | (g_map_zstream == 2 ? 2 : 0);
    if (g_map_zstream == 3) return launch_map_zstream_as<false, 8, 1>(in, coords, out, q, tiles, s);
    switch ((int)g_map_zvariant) {
    case 81: return launch_map_zstream_as<true, 8, 1>(in, coords, out, q, tiles, s);
    case 82: return launch_map_zstream_as<true, 8, 2>(in, coords, out, q, tiles, s);
    case 41: return launch_map_zstream_as<true, 4, 1>(in, coords, out, q, tiles, s);
    default: return launch_map_zstream_as<true, 8, 1>(in, coords, out, q, tiles, s);       // config D: 499 us; <true, 4, 1> 519 (but 779 against 917 us on a warp where a third of the voxels gather for themselves)
    }
}

Knob g_interp_c1{1};     // test hook: 0 = round-2 kernels for constant-mode order-1 float32 volumes, 1 = r3 kernels, 2 = r3 without the wide stores / loads, 3 = r3 with z-major voxel ownership, 5 = r3 (L1 gathers) without the LDS-staged affine kernel; 1 (default) and 4 use the LDS-staged affine kernel when the box fits, 6 = row-major ownership for map_coordinates, 7 = pair-sharing map_coordinates kernel (4 = LDS-staged map_coordinates)

static bool fast_ok(const mi_array *in, const mi_array *out, int order)
{
    if (in->ndim != out->ndim || (in->ndim != 3 && in->ndim != 2) || (in->dtype != MI_F32 && in->dtype != MI_F64) ||
        out->dtype != in->dtype)
        return false;
    if (order < 0 || order > 1) return false;
    if (numel(in) * (int64_t)dtype_size(in->dtype) >= ((int64_t)1 << 31) || numel(out) >= ((int64_t)1 << 31)) return false;   // 32-bit byte offsets
    const int nd = in->ndim;
    if (in->shape[nd - 1] < 2) return false;
    if ((nd == 3 && out->shape[0] > 65535) || (out->shape[nd - 2] + 3) / 4 > 65535) return false;
    if ((uintptr_t)out->data & 15) return false;
    return true;
}

static void fill_params(FastInterpParams *p, const mi_array *in, const mi_array *out, int order, int mode, double cval)
{
    const int pad = 3 - in->ndim;
    p->two_d = pad;
    p->nz = pad ? 1 : (int)in->shape[0]; p->ny = (int)in->shape[1 - pad]; p->nx = (int)in->shape[2 - pad];
    p->oz = pad ? 1 : (int)out->shape[0]; p->oy = (int)out->shape[1 - pad]; p->ox = (int)out->shape[2 - pad];
    p->order = order; p->mode = mode; p->cval = in->dtype == MI_F32 ? (double)(float)cval : cval;
}

// returns MI_ERR_UNSUPPORTED when the request is not covered (caller runs the generic kernel)
int map_coordinates_fast(const mi_array *in, const mi_array *coords, const mi_array *out, int order, int mode,
                         double cval, hipStream_t s)
{
    if (!fast_ok(in, out, order)) return MI_ERR_UNSUPPORTED;
    FastInterpParams p;
    fill_params(&p, in, out, order, mode, cval);
    const dim3 block(64, 4, 1);
    const dim3 grid((unsigned)((p.ox + 63) / 64), (unsigned)((p.oy + 4 * kNV - 1) / (4 * kNV)), (unsigned)p.oz);
    const bool fastc = mode == MI_MODE_CONSTANT && order == 1;
    const int var = g_interp_c1;
    if (fastc && var && !p.two_d && in->dtype == MI_F32 && coords->dtype == MI_F32 && (p.ox & 3) == 0 &&
        ((uintptr_t)coords->data & 15) == 0) {
        const float *ip = (const float *)in->data, *cp = (const float *)coords->data;
        float *op = (float *)out->data;
        const dim3 gridz((unsigned)((p.ox + 63) / 64), (unsigned)((p.oy + 3) / 4), (unsigned)((p.oz + 3) / 4));
        // knob 4 only: gathers out of an LDS-staged box found per workgroup (map_coords3d_lds_kernel).  Measured on config D:
        // 773 us against 608 us for the L1 gathers below -- two dependent long-latency phases per tile (coordinates, then
        // the box) with two workgroups per CU (118 VGPRs) leave the CU idle; kept for the record, not the default.
        // r4: the z-streaming kernel (taps out of LDS, phases of consecutive planes overlapped)
        if (g_map_zstream != 0 && (var == 1) && p.ox >= 64 && (int64_t)p.oz * p.oy * p.ox >= (1 << 18))
            return launch_map_zstream(ip, cp, op, p, s);
        const dim3 gridl((unsigned)((p.ox + 31) / 32), (unsigned)((p.oy + 15) / 16), (unsigned)((p.oz + 7) / 8));
        if (var == 4 && (int64_t)p.oz * p.oy * p.ox >= (1 << 18) && gridl.y <= 65535 && gridl.z <= 65535) {
            hipLaunchKernelGGL(map_coords3d_lds_kernel, gridl, dim3(512), 0, s, ip, cp, op, p);
            MI_HIP(hipGetLastError());
            return MI_OK;
        }
        // knob 7 only: two x-neighbours per lane sharing their gathers (map_coords3d_pair_kernel).  Measured on config D:
        // 721 us against 590-611 us for the kernel below (half the gather instructions, but 24 selects per shared voxel,
        // 89 VGPRs and 8-byte coordinate loads / stores) -- correct (tests, fuzz), slower, kept for the record.
        const dim3 gridp((unsigned)((p.ox + 127) / 128), (unsigned)((p.oy + 3) / 4), (unsigned)((p.oz + 3) / 4));
        if (var == 7 && (p.ox & 1) == 0 && p.ox >= 2 && gridp.y <= 65535 && gridp.z <= 65535) {
            hipLaunchKernelGGL(map_coords3d_pair_kernel, gridp, block, 0, s, ip, cp, op, p);
            MI_HIP(hipGetLastError());
            return MI_OK;
        }
        // otherwise z-major voxel ownership (config D: 605 us against 623 us row-major, profiles/r3_interp_variants.txt)
        if (var == 2) hipLaunchKernelGGL((map_coords3d_c1_kernel<false, false>), grid, block, 0, s, ip, cp, op, p);
        else if (var != 6 && gridz.y <= 65535 && gridz.z <= 65535) hipLaunchKernelGGL((map_coords3d_c1_kernel<true, true>), gridz, block, 0, s, ip, cp, op, p);
        else hipLaunchKernelGGL((map_coords3d_c1_kernel<true, false>), grid, block, 0, s, ip, cp, op, p);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
#define MI_MAP(CT, FC, ORD)                                                                                        \
    do {                                                                                                           \
        if (in->dtype == MI_F32)                                                                                   \
            hipLaunchKernelGGL((map_coords3d_fast<float, CT, FC, ORD>), grid, block, 0, s, (const float *)in->data, \
                               (const CT *)coords->data, (float *)out->data, p);                                   \
        else                                                                                                       \
            hipLaunchKernelGGL((map_coords3d_fast<double, CT, FC, ORD>), grid, block, 0, s, (const double *)in->data, \
                               (const CT *)coords->data, (double *)out->data, p);                                  \
    } while (0)
    if (coords->dtype == MI_F32) {
        if (fastc) MI_MAP(float, true, 1); else if (order == 1) MI_MAP(float, false, 1); else MI_MAP(float, false, 0);
    } else {
        if (fastc) MI_MAP(double, true, 1); else if (order == 1) MI_MAP(double, false, 1); else MI_MAP(double, false, 0);
    }
#undef MI_MAP
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int affine_transform_fast(const mi_array *in, const mi_array *out, const double *matrix, int order, int mode,
                          double cval, hipStream_t s)
{
    if (!fast_ok(in, out, order)) return MI_ERR_UNSUPPORTED;
    FastInterpParams p;
    fill_params(&p, in, out, order, mode, cval);
    if (p.two_d) {
        // rows of a 2 x 3 matrix embedded in the 3 x 4 one; the z coordinate is exactly 0
        for (int i = 0; i < 12; i++) p.m[i] = 0.0;
        for (int r = 0; r < 2; r++) {
            p.m[4 * (r + 1) + 1] = matrix[3 * r + 0];
            p.m[4 * (r + 1) + 2] = matrix[3 * r + 1];
            p.m[4 * (r + 1) + 3] = matrix[3 * r + 2];
        }
    } else {
        for (int i = 0; i < 12; i++) p.m[i] = matrix[i];
    }
    const dim3 block(64, 4, 1);
    const dim3 grid((unsigned)((p.ox + 63) / 64), (unsigned)((p.oy + 4 * kNV - 1) / (4 * kNV)), (unsigned)p.oz);
    const int var = g_interp_c1;
    if (mode == MI_MODE_CONSTANT && order == 1 && (var == 1 || var == 4) && !p.two_d && in->dtype == MI_F32 && (p.ox & 3) == 0 &&
        (int64_t)p.oz * p.oy * p.ox >= (1 << 18)) {
        // gathers out of LDS when a tile's bounding box is small enough (decided from the matrix alone): the tile shape
        // with the smaller box of 64 x 8 x 8 and 32 x 16 x 8
        // r4: matrices that leave axis 0 (or axis 1) to itself stream along it (affine3d_zstream_kernel)
        if (g_affine_zstream != 0) {
            ZStreamParams zq;
            const int want = g_affine_zstream;
            for (int S = 0; S < 2; S++) {
                if (want != 64 && zstream_plan<32>(p, S, &zq)) return launch_affine_zstream<32>((const float *)in->data, (float *)out->data, zq, s);
                if (want != 32 && zstream_plan<64>(p, S, &zq)) return launch_affine_zstream<64>((const float *)in->data, (float *)out->data, zq, s);
            }
        }
        if (g_affine_rowblend != 0) {
            RowBlendParams rq;
            if (rowblend_plan(p, &rq)) return launch_affine_rowblend((const float *)in->data, (float *)out->data, rq, s);
        }
        // the tile shape with the smallest box: 64 x 8 x 8, 32 x 16 x 8, and for matrices that couple all three axes the
        // cube 16 x 16 x 16 / 16 x 32 x 8 (r5)
        LdsAffineParams qs[4];
        long long fl[4];
        fl[0] = lds_affine_plan(p, 64, 8, &qs[0]);
        fl[1] = (p.ox & 31) == 0 || p.ox > 256 ? lds_affine_plan(p, 32, 16, &qs[1]) : 0;
        fl[2] = (p.ox & 15) == 0 || p.ox > 256 ? lds_affine_plan(p, 16, 16, &qs[2]) : 0;
        fl[3] = (p.ox & 15) == 0 || p.ox > 256 ? lds_affine_plan(p, 16, 32, &qs[3]) : 0;
        int best = -1;
        if (g_affine_box_kib >= 1000) {              // test hook: + 1000 x (1 + shape) forces a tile shape
            const int force = g_affine_box_kib / 1000 - 1;
            for (int i = 0; i < 4; i++) if (i != force) fl[i] = 0;
        }
        // the smaller box of the two wide tiles when one fits; the cube (then 16 x 32 x 8) only when neither does: at equal
        // LDS occupancy the 16-wide tiles lose 10-15 % to their 64-byte store rows (profiles/r5_affine_general.txt: 10
        // degrees about (1, 1, 1): 416 us with 32 x 16 x 8 / 46 KiB, 480 us with the cube / 37 KiB), and every shape collapses
        // once its box leaves room for one workgroup per CU only (780 - 930 us: the budget stays at 64 KiB)
        if (fl[0] && (!fl[1] || fl[0] <= fl[1])) best = 0;
        else if (fl[1]) best = 1;
        else if (fl[2]) best = 2;
        else if (fl[3]) best = 3;
        int rc = MI_ERR_UNSUPPORTED;
        const float *ip_ = (const float *)in->data;
        float *op_ = (float *)out->data;
        if (best == 0) rc = launch_affine_lds<64, 8>(ip_, op_, qs[0], s);
        else if (best == 1) rc = launch_affine_lds<32, 16>(ip_, op_, qs[1], s);
        else if (best == 2) rc = launch_affine_lds<16, 16>(ip_, op_, qs[2], s);
        else if (best == 3) rc = launch_affine_lds<16, 32>(ip_, op_, qs[3], s);
        if (rc != MI_ERR_UNSUPPORTED) return rc;
    }
    if (mode == MI_MODE_CONSTANT && order == 1 && var && !p.two_d && in->dtype == MI_F32 && (p.ox & 3) == 0) {
        const float *ip = (const float *)in->data;
        float *op = (float *)out->data;
        const dim3 gridz((unsigned)((p.ox + 63) / 64), (unsigned)((p.oy + 3) / 4), (unsigned)((p.oz + 3) / 4));
        note_kernel("mi::affine3d_c1_kernel (order-1 affine, L1 gathers)");
        if (var == 3 && gridz.y <= 65535 && gridz.z <= 65535) hipLaunchKernelGGL((affine3d_c1_kernel<true, true>), gridz, block, 0, s, ip, op, p);
        else if (var == 2) hipLaunchKernelGGL((affine3d_c1_kernel<false, false>), grid, block, 0, s, ip, op, p);
        else hipLaunchKernelGGL((affine3d_c1_kernel<true, false>), grid, block, 0, s, ip, op, p);
        MI_HIP(hipGetLastError());
        return MI_OK;
    }
#define MI_AFF(FC, ORD)                                                                                                    \
    do {                                                                                                                   \
        if (in->dtype == MI_F32)                                                                                           \
            hipLaunchKernelGGL((affine3d_fast<float, FC, ORD>), grid, block, 0, s, (const float *)in->data, (float *)out->data, p); \
        else                                                                                                               \
            hipLaunchKernelGGL((affine3d_fast<double, FC, ORD>), grid, block, 0, s, (const double *)in->data, (double *)out->data, p); \
    } while (0)
    note_kernel("mi::affine3d_fast (order %d)", order);
    if (mode == MI_MODE_CONSTANT && order == 1) MI_AFF(true, 1);
    else if (order == 1) MI_AFF(false, 1);
    else MI_AFF(false, 0);
#undef MI_AFF
    MI_HIP(hipGetLastError());
    return MI_OK;
}

}  // namespace mi

extern "C" int mi_debug_set_interp_c1(int k) { mi::g_interp_c1 = k; return MI_OK; }
extern "C" int mi_debug_set_affine_dbg(int k) { mi::g_affine_dbg = k; return MI_OK; }
extern "C" int mi_debug_set_affine_gz(int k) { mi::g_affine_gz = k; return MI_OK; }
extern "C" int mi_debug_set_affine_rowblend(int k) { mi::g_affine_rowblend = k; return MI_OK; }
extern "C" int mi_debug_set_map_zstream(int k) { mi::g_map_zstream = k; return MI_OK; }
extern "C" int mi_debug_set_map_zchunks(int k) { mi::g_map_zchunks = k; return MI_OK; }
extern "C" int mi_debug_set_map_zvariant(int k) { mi::g_map_zvariant = k; return MI_OK; }
extern "C" int mi_debug_set_affine_zstream(int k) { mi::g_affine_zstream = k; return MI_OK; }
extern "C" int mi_debug_set_affine_zchunks(int k) { mi::g_affine_zchunks = k; return MI_OK; }
