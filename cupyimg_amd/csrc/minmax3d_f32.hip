// minmax3d_f32.hip -- fused separable 3-D min / max for float32 volumes, ONE launch.
//
// Replaces, for minimum_filter / maximum_filter / grey_erosion / grey_dilation with a flat cubic `size` (3 .. 9) on
// float32 volumes (cupyimg/scipy/ndimage/filters.py:1373-1419 -> three K2 launches with a compare in double per tap;
// morphology.py:769-884), the two streaming launches of round 1 (stream3d.hip, 16 B/voxel, each at copy speed:
// 0.40-0.45 ms on 512^3).  Same skeleton as the fused long separable kernel (sep3d_long.hip, long_common.hpp): tile
// 256 x 16, sixteen waves, raw rows staged by LDS-DMA into a ring of four planes (x halo boundary mapped at the
// DMA), pass order y (rows out of LDS), x (registers, DPP lane shifts), z (register ring of the W - 1 previous planes).
//
// Arithmetic: v_min3_f32 / v_max3_f32 chains, then one fix-up per pass that reproduces the compare-select form of the
// generic kernels (`x < best ? x : best` in ascending tap order, first tap taken as is): the hardware min / max ignore
// NaNs, the compare-select form returns NaN exactly when the FIRST tap is one -- so result = isnan(first) ? first :
// chain.  (Only the sign of a zero can differ: -0.0 < +0.0 is false for the compare, true for v_min.)
// `constant` mode is left to the streaming passes (the DMA zero-fills, and min / max are not linear).
#include "long_common.hpp"

namespace mi {

struct MmLongParams {
    int nt;                 // 1 = exclusive rows staged non-temporally (long_common.hpp)
    int nx, ny, nz;
    int oy, oz;             // w/2 + origin along y and z (x: W/2)
    int mx, my, mz;         // boundary modes (never constant)
    int zc, nxt, nyt, nzc;
    int tw;                 // tile width in floats (<= 256, multiple of 4)
    int zb, zn;             // output planes to produce: [zb, zb + zn) (whole volume: 0, nz); boundary handling refers to nz
};

template <bool IS_MAX> __device__ __forceinline__ float mm2(float a, float b) { return IS_MAX ? fmaxf(a, b) : fminf(a, b); }
template <bool IS_MAX> __device__ __forceinline__ float mm3(float a, float b, float c)
{
    return IS_MAX ? __builtin_fmaxf(__builtin_fmaxf(a, b), c) : __builtin_fminf(__builtin_fminf(a, b), c);   // -> v_max3 / v_min3
}
// window of N taps t[0..N-1] (ascending): compare-select semantics via min3 / max3 chain + first-tap NaN fix-up
template <bool IS_MAX, int N>
__device__ __forceinline__ float win(const float (&t)[N])
{
    float r = t[0];
    static_for<(N - 1) / 2>([&](auto KK) {
        constexpr int k = decltype(KK)::value;
        r = mm3<IS_MAX>(r, t[1 + 2 * k], t[2 + 2 * k]);
    });
    if constexpr ((N - 1) % 2 == 1) r = mm2<IS_MAX>(r, t[N - 1]);
    return t[0] != t[0] ? t[0] : r;
}

// RG (r6): rows of any length, as sep3d_long3_kernel's ragged build (sep3d_long.hip): 4-byte-aligned LDS-DMA records, the halo
// from the row's true end, the last lane's floats beyond its `tail` replaced by the y-filtered continuation, `tail` floats
// stored in pieces (three stores per step from the first complete output on: vmcnt(7) where the aligned build has vmcnt(5)).
template <int W, bool IS_MAX, bool RG = false>
__global__ void __launch_bounds__(kLongTY * 64)
mm3f32_long_kernel(const float *__restrict__ in, float *__restrict__ out, const MmLongParams p)
{
    constexpr int RX = W / 2;
    constexpr int NBK = (RX + 3) / 4;                 // 4-float blocks per side in the x window (1 for W <= 9)
    constexpr int RINGN = W - 1;                      // even
    static_assert(W >= 3 && W <= 9 && (W & 1), "cubic sizes 3 .. 9");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: planes[kLongNB][32] records | hy[2][16][4] float4 | ztab
    constexpr unsigned HY0 = kLongRawBytes;
    int *ztab = reinterpret_cast<int *>(smem + kLongRawBytes + kLongHyBytes);

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
    const int zs = p.zb + zci * p.zc, ze = min(zs + p.zc, p.zb + p.zn);
    const int ty_act = min(kLongTY, ny - y0);
    const int rows_needed = ty_act + W - 1;
    const int width = min(p.tw, nx - x0);
    const int nlanes = RG ? (width + 3) >> 2 : min(p.tw >> 2, (nx - x0) >> 2);
    const int last = nlanes - 1;
    const int tail = RG ? width - 4 * last : 4;             // floats of its row the last lane holds
    const int xe = RG ? x0 + width : x0 + 4 * nlanes;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * 4u;
    const int zi0 = zs - p.oz;
    const int nsteps = ze - zs + W - 1;

    for (int i = threadIdx.x; i < nsteps; i += kLongTY * 64) ztab[i] = bmap<int>(zi0 + i, nz, p.mz);
    __syncthreads();

    unsigned vmain[2], vhalo[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = wave + 16 * h;
        const bool valid = r < rows_needed;
        const int ys = bmap<int>(y0 - p.oy + r, ny, p.my);
        vmain[h] = (valid && lane < nlanes) ? (unsigned)(ys * nx + x0 + 4 * lane) * 4u : kOOB;
        const int j = lane & 15;
        const int xsrc = bmap<int>(j < 8 ? x0 - 8 + j : xe + j - 8, nx, p.mx);
        vhalo[h] = valid ? (unsigned)(ys * nx + xsrc) * 4u : kOOB;
    }
    const unsigned own = (unsigned)wave * kLongRec + (unsigned)lane * 16u;
    const unsigned hsrc = (unsigned)(lane >> 2) * kLongRec + 1024u + (unsigned)(lane & 3) * 16u;
    const unsigned hy_near = HY0 + (unsigned)wave * 64u + (lane == 0 ? 16u : 32u);     // lane 0: block x0-4..x0-1; others: xe..xe+3
    const bool rg_last = RG && lane == last && tail < 4;
    const bool rg_p1 = rg_last && tail < 2, rg_p2 = rg_last && tail < 3, rg_p3 = rg_last;
    const unsigned ovoff0 = (unsigned)((y0 + wave) * nx + x0 + 4 * lane) * 4u;
    const unsigned ovoff = (wave < ty_act && lane < nlanes && !rg_last) ? ovoff0 : kOOB;
    const unsigned ovoff2 = (wave < ty_act && rg_last && (tail & 2)) ? ovoff0 : kOOB;
    const unsigned ovoff1 = (wave < ty_act && rg_last && (tail & 1)) ? ovoff0 + ((tail & 2) ? 8u : 0u) : kOOB;
    constexpr unsigned kPlane = kLongRowsMax * kLongRec;

    auto issue = [&](int i, unsigned bufoff) {
        const bool live = i < nsteps;
        int zsrc = zi0 + i;
        if ((unsigned)zsrc >= (unsigned)nz) zsrc = __builtin_amdgcn_readfirstlane(ztab[live ? i : 0]);
        const unsigned long long a = (unsigned long long)in + (unsigned long long)(unsigned)zsrc * (unsigned long long)plane_bytes;
        u32x4_t rin;
        rin.x = (unsigned)a;
        rin.y = (unsigned)(a >> 32);
        rin.z = live ? plane_bytes : 0u;
        rin.w = 0x00020000u;
        // rows W - 1 .. 15 of the tile are read by this workgroup only: non-temporal (long_common.hpp)
        dma_two_rows(rin, vmain[0], vhalo[0], vmain[1], vhalo[1], bufoff + (unsigned)wave * kLongRec, p.nt && wave >= W - 1);
    };

    // y window of W consecutive records starting at LDS byte address `at` (one float4 per lane)
    auto ypass = [&](unsigned at) {
        float4 rows[W];
#pragma unroll
        for (int k = 0; k < W; k++) rows[k] = *reinterpret_cast<const float4 *>(smem + at + k * kLongRec);
        float tx[W], ty[W], tz[W], tw[W];
#pragma unroll
        for (int k = 0; k < W; k++) { tx[k] = rows[k].x; ty[k] = rows[k].y; tz[k] = rows[k].z; tw[k] = rows[k].w; }
        return make_float4(win<IS_MAX, W>(tx), win<IS_MAX, W>(ty), win<IS_MAX, W>(tz), win<IS_MAX, W>(tw));
    };

    float4 ring[RINGN];
#pragma unroll
    for (int k = 0; k < RINGN; k++) ring[k] = make_float4(0.f, 0.f, 0.f, 0.f);

    issue(0, 0);
    issue(1, kPlane);
    issue(2, 2 * kPlane);
    asm volatile(MI_VMCNT(8) "\n\ts_barrier" ::: "memory");
    if (wave == 15) {
        const float4 hv = ypass(hsrc);
        *reinterpret_cast<float4 *>(smem + HY0 + (unsigned)lane * 16u) = hv;
    }

    // interval i: see sep3d_long_kernel (planes i and i + 1 landed, halo table of plane i complete, slot of plane i - 1 free)
    unsigned bi = 0;
    for (int i0 = 0; i0 < nsteps; i0 += RINGN) {
        static_for<RINGN>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                const unsigned b1 = bi == (kLongNB - 1) * kPlane ? 0u : bi + kPlane;
                const unsigned b3 = bi == 0u ? (kLongNB - 1) * kPlane : bi - kPlane;
                // Oldest first, this wave has in flight: the 4 DMAs of plane i + 1 (issued two steps ago), the store of
                // step i - 2, the 4 DMAs of plane i + 2 and the store of step i - 1 (vector memory operations of a wave
                // retire in issue order on gfx9).  Plane i + 1 must have landed; plane i + 2 AND the store behind it stay
                // in flight: vmcnt(5) once stores have begun.  (r2 waited vmcnt(4) throughout, i.e. for the first DMA of
                // the plane issued one step earlier: a prefetch distance of one plane, not two -- removing the DMAs
                // altogether saved 86 us of 358 on config B, the waves were stalling on them.)
                if (i >= W) {
                    if constexpr (RG) asm volatile(MI_VMCNT(7) " lgkmcnt(0)\n\ts_barrier" ::: "memory");     // three stores per step
                    else asm volatile(MI_VMCNT(5) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                } else asm volatile(MI_VMCNT(4) " lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue(i + 3, b3);
                const unsigned hyoff = (unsigned)(i & 1) * (kLongHyBytes / 2);
                // ---- y window of output row `wave`, then its x window in registers
                float4 yv = ypass(own + bi);
                const float4 eg = *reinterpret_cast<const float4 *>(smem + hy_near + hyoff);
                float4 eright = eg;
                if constexpr (RG) {
                    // the row's continuation (near-right block h0 .. h3 of the table row, from the row's true end) behind the last
                    // lane's `tail` floats; what follows it: eight dword reads from float 8 - tail of the table row on
                    const float *hp = reinterpret_cast<const float *>(smem + HY0 + (unsigned)wave * 64u + hyoff + 32u - 4u * (unsigned)tail);
                    yv.y = rg_p1 ? hp[1] : yv.y;
                    yv.z = rg_p2 ? hp[2] : yv.z;
                    yv.w = rg_p3 ? hp[3] : yv.w;
                    eright = make_float4(hp[4], hp[5], hp[6], hp[7]);
                }
                const float4 l = dpp4_shr(eg, yv);
                const float4 rr = dpp4_shl(eg, yv);
                const float4 r = lane == last ? eright : rr;
                const float e[12] = {l.x, l.y, l.z, l.w, yv.x, yv.y, yv.z, yv.w, r.x, r.y, r.z, r.w};
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float t[W];
#pragma unroll
                    for (int k = 0; k < W; k++) t[k] = e[4 - RX + c + k];
                    o[c] = win<IS_MAX, W>(t);
                }
                const float4 xy = make_float4(o[0], o[1], o[2], o[3]);
                // ---- z window: the W - 1 previous planes (oldest first) and this one
                if (i >= W - 1) {
                    float t0[W], t1[W], t2[W], t3[W];
#pragma unroll
                    for (int k = 0; k < RINGN; k++) {
                        const float4 q = ring[(J + k) % RINGN];
                        t0[k] = q.x; t1[k] = q.y; t2[k] = q.z; t3[k] = q.w;
                    }
                    t0[W - 1] = xy.x; t1[W - 1] = xy.y; t2[W - 1] = xy.z; t3[W - 1] = xy.w;
                    const float4 res = make_float4(win<IS_MAX, W>(t0), win<IS_MAX, W>(t1), win<IS_MAX, W>(t2), win<IS_MAX, W>(t3));
                    const unsigned long long oa = (unsigned long long)out +
                                                  (unsigned long long)(unsigned)(zs + i - (W - 1)) * (unsigned long long)plane_bytes;
                    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)oa, 0, (int)plane_bytes, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(res.x), __float_as_uint(res.y), __float_as_uint(res.z),
                                                                   __float_as_uint(res.w)}, rout, ovoff, 0, 2);
                    if constexpr (RG) {
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(res.x), __float_as_uint(res.y)}, rout, ovoff2, 0, 2);
                        __builtin_amdgcn_raw_buffer_store_b32((tail & 2) ? __float_as_uint(res.z) : __float_as_uint(res.x), rout, ovoff1, 0, 2);
                    }
                }
                ring[J % RINGN] = xy;
                // ---- halo table of plane i + 1 (the wave changes every plane)
                if (i + 1 < nsteps && wave == (i & 15)) {
                    const float4 hv = ypass(hsrc + b1);
                    *reinterpret_cast<float4 *>(smem + HY0 + (kLongHyBytes / 2 - hyoff) + (unsigned)lane * 16u) = hv;
                }
                bi = b1;
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int mm_long_cus() { return device_cus(); }

template <int W, bool IS_MAX, bool RG = false>
static int launch_mm_long(const float *in, float *out, MmLongParams &p, hipStream_t s)
{
    const size_t lds = (size_t)kLongRawBytes + kLongHyBytes + (size_t)(kLongMaxChunk + kStreamMaxTaps) * sizeof(int);
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)mm3f32_long_kernel<W, IS_MAX, RG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    note_kernel("mi::mm3f32_long_kernel<%d,%s%s> grid=%d (fused y/x/z flat min / max, LDS-DMA staged%s)", W, IS_MAX ? "max" : "min", RG ? ",ragged" : "",
                p.nxt * p.nyt * p.nzc, RG ? ", rows of any length" : "");
    hipLaunchKernelGGL((mm3f32_long_kernel<W, IS_MAX, RG>), dim3(p.nxt * p.nyt * p.nzc), dim3(kLongTY * 64), lds, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// Fused min / max: cubic odd w in 3..9, origin 0 along x, no constant mode; MI_ERR_UNSUPPORTED otherwise (the caller runs
// the streaming passes).
int run_minmax3d_f32_fused_planes(const float *in, float *out, int nz, int ny, int nx, int w, int oy, int oz, int mx, int my, int mz,
                                  bool is_max, int zb, int zn, hipStream_t s);

int run_minmax3d_f32_fused(const float *in, float *out, int nz, int ny, int nx, int w, int oy, int oz, int mx, int my, int mz,
                           bool is_max, hipStream_t s)
{
    return run_minmax3d_f32_fused_planes(in, out, nz, ny, nx, w, oy, oz, mx, my, mz, is_max, 0, nz, s);
}

// output planes [zb, zb + zn) only (multi-GPU slabs: the planes next to a neighbour wait for the halo exchange)
int run_minmax3d_f32_fused_planes(const float *in, float *out, int nz, int ny, int nx, int w, int oy, int oz, int mx, int my, int mz,
                                  bool is_max, int zb, int zn, hipStream_t s)
{
    if (zn <= 0) return MI_OK;
    if (w < 3 || w > 9 || !(w & 1) || nx < 16) return MI_ERR_UNSUPPORTED;
    if (mx == MI_MODE_CONSTANT || my == MI_MODE_CONSTANT || mz == MI_MODE_CONSTANT) return MI_ERR_UNSUPPORTED;
    if ((int64_t)ny * nx * 4 >= ((int64_t)1 << 31)) return MI_ERR_UNSUPPORTED;
    MmLongParams p;
    memset(&p, 0, sizeof(p));
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.oy = oy; p.oz = oz;
    p.mx = mx; p.my = my; p.mz = mz;
    p.nxt = (nx + 255) / 256;
    p.nt = stream_nt_for((long long)nz * ny * nx * 8, p.nxt);
    p.tw = (((nx + p.nxt - 1) / p.nxt) + 3) & ~3;          // equal tiles (see separable3d.hip)
    p.nyt = (ny + kLongTY - 1) / kLongTY;
    const int ncu = mm_long_cus();
    const int cols = p.nxt * p.nyt;
    int best_nzc = 1;
    double best = 1e300;
    p.zb = zb; p.zn = zn;
    for (int nzc = 1; nzc <= zn && nzc <= 256; nzc++) {
        const int chunk = (zn + nzc - 1) / nzc;
        if (chunk > kLongMaxChunk) continue;
        const int real = (zn + chunk - 1) / chunk;
        const double rounds = (double)(((int64_t)cols * real + ncu - 1) / ncu);
        const double cost = rounds * (chunk + w - 1 + 3);
        if (cost < best) { best = cost; best_nzc = real; }
    }
    p.zc = (zn + best_nzc - 1) / best_nzc;
    if (p.zc > kLongMaxChunk) p.zc = kLongMaxChunk;
    p.nzc = (zn + p.zc - 1) / p.zc;
    if (nx & 3) {                 // r6: rows of any length -- size 9 (3 / 5 / 7 on such rows take the lean kernel's ragged build, which is faster there)
        if (w != 9) return MI_ERR_UNSUPPORTED;
        return is_max ? launch_mm_long<9, true, true>(in, out, p, s) : launch_mm_long<9, false, true>(in, out, p, s);
    }
#define MI_MM_CASE(N) case N: return is_max ? launch_mm_long<N, true>(in, out, p, s) : launch_mm_long<N, false>(in, out, p, s);
    switch (w) { MI_MM_CASE(3) MI_MM_CASE(5) MI_MM_CASE(7) MI_MM_CASE(9) }
#undef MI_MM_CASE
    return MI_ERR_UNSUPPORTED;
}

}  // namespace mi

/* Separable flat min / max filter restricted to one or two ranges of output planes (float32 volumes, cubic odd sizes
 * 3 .. 9, index-mapping boundary modes: what the fused kernel takes; MI_ERR_UNSUPPORTED otherwise).  The multi-GPU slab
 * schedule filters the planes whose taps stay inside a rank's own planes while the halo exchange is in flight and the
 * planes next to a neighbour afterwards (include/mi355img.h). */
extern "C" int mi_minmax3d_f32_planes(const mi_array *in, const mi_array *out, const int size[3], const int origin[3],
                                      const int mode[3], double cval, int is_max, const int64_t *planes, int nranges,
                                      mi_stream stream)
{
    using namespace mi;
    (void)cval;
    int rc;
    if ((rc = check_array(in, "in")) || (rc = check_array(out, "out"))) return rc;
    MI_REQUIRE(size && origin && mode && planes, MI_ERR_INVALID_ARG, "NULL argument");
    MI_REQUIRE(nranges >= 1 && nranges <= 2, MI_ERR_INVALID_ARG, "one or two plane ranges");
    MI_REQUIRE(same_shape(in, out), MI_ERR_INVALID_ARG, "output shape is not correct");
#define UNSUP(msg) do { set_error("minmax3d_f32_planes: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (in->ndim != 3 || in->dtype != MI_F32 || out->dtype != MI_F32) UNSUP("needs 3-D float32 in/out");
    if (!is_contiguous(in) || !is_contiguous(out) || in->data == out->data) UNSUP("needs distinct C-contiguous arrays");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nz < 1 || ny < 1 || nx < 16 || (nx & 3)) UNSUP("x extent must be a multiple of 4, >= 16");
    if (nz * ny * nx * 4 >= ((int64_t)1 << 31)) UNSUP("needs a volume < 2 GiB");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15)) UNSUP("needs 16-byte aligned data");
    const int w = size[0];
    if (size[1] != w || size[2] != w || w < 3 || w > 9 || !(w & 1)) UNSUP("cubic odd sizes 3 .. 9 only");
    if (origin[2] != 0) UNSUP("x origin must be 0");
    int off[2];
    for (int a = 0; a < 2; a++) {
        off[a] = w / 2 + origin[a];
        if (off[a] < 0 || off[a] >= w) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
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
        rc = run_minmax3d_f32_fused_planes((const float *)in->data, (float *)out->data, (int)nz, (int)ny, (int)nx, w, off[1], off[0],
                                           filter_mode(mode[2]), filter_mode(mode[1]), filter_mode(mode[0]), is_max != 0, (int)b,
                                           (int)(e - b), s);
        if (rc != MI_OK) return rc;
    }
    return MI_OK;
#undef UNSUP
}
