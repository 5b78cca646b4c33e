// sep3d_long.hip -- fused separable 3-D filter for LONG cubic kernels (11..17
// taps per axis, e.g. gaussian sigma=2 -> 17 taps, BASELINE config B), float32,
// ONE launch at the algorithmic 8 B/voxel.
//
// Reference path replaced: gaussian_filter / uniform_filter as three K1 launches
// with fp64 taps from global memory, zero-fill + copy-back per in-place pass
// (cupyimg/scipy/ndimage/filters.py:602-665,725-792, _filters_core.py:148-155).
// Round 1 ran long kernels as two streaming launches (stream3d.hip, 16 B/voxel,
// latency bound: two 1 KiB loads in flight per wave).
//
// Design (one workgroup of 16 waves per CU, tile = 256 x by TY = 16 y, streaming
// along z over a chunk of planes):
//   * LDS-DMA staging: every raw input row of the tile's (16 + W - 1)-row window
//     goes global -> LDS with `buffer_load_dwordx4 ... lds` (1 KiB per wave
//     instruction, no VGPRs), three planes deep, so ~100 KiB per CU are in flight
//     while the registers hold the z state.  The 8-float x halo of a row is a
//     second, 16-lane `buffer_load_dword ... lds` whose per-lane source address is
//     already boundary mapped (reflect / mirror / nearest / wrap resolved here).
//   * x pass: wave w filters raw rows w and w + 16 straight out of LDS (five
//     lane-contiguous ds_read_b128 give the 20-float window, no lane shuffles),
//     packed fp32 dot product against host-made weight pairs, result to an LDS
//     row buffer.
//   * y pass: wave w owns output row w: W lane-contiguous ds_read_b128.
//   * z pass: in registers as a scatter -- the x/y-filtered sample is added into
//     W pending output accumulators (68 VGPRs), the oldest one is complete and is
//     stored (non-temporal buffer_store_dwordx4).  The rotation is by unrolling W
//     steps, like the rings of the other kernels.
//   * two s_barriers per plane; DMA completion is counted by hand
//     (s_waitcnt vmcnt(8): the two younger planes stay in flight across barriers).
// Boundary modes: every index-mapping mode on every axis; `constant` is left to
// the streaming passes (DMA cannot substitute cval).
#include "sep_common.hpp"
#include "stream3d.hpp"

namespace mi {

constexpr int kLongTY = 16;           // output rows per tile = waves per workgroup
constexpr int kLongRowsMax = 32;      // raw rows per plane (TY + 17 - 1)
constexpr int kLongRec = 1024 + 64;   // LDS bytes per raw row: 256 floats + 16 halo floats
constexpr int kLongNB = 3;            // raw planes in LDS
constexpr int kLongRawBytes = kLongNB * kLongRowsMax * kLongRec;
constexpr int kLongXfBytes = kLongRowsMax * 1024;
constexpr int kLongMaxChunk = 1024;   // planes per z chunk (ztab in LDS)

struct LongParams {
    int nx, ny, nz;
    int oy, oz;             // w/2 + origin along y and z (x: W/2)
    int mx, my, mz;         // boundary modes (filter_mode()-normalised, never constant)
    int zc;                 // output planes per chunk
    int nxt, nyt, nzc;      // tile counts
    float wyv[kStreamMaxTaps], wzv[kStreamMaxTaps];
    float xpair[2][2 * (kStreamMaxTaps / 2 + 2)];   // see StreamParams::xpair
};

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// global -> LDS, 16 bytes per lane, LDS destination = lds_dst + 16 * lane (wave uniform base in M0)
__device__ __forceinline__ void dma_row16(u32x4_t rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
// global -> LDS, 4 bytes per lane, lanes 0..15 only (64 bytes at lds_dst)
__device__ __forceinline__ void dma_halo16(u32x4_t rsrc, unsigned voff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_mov_b64 exec, 0xffff\n\t"
                 "buffer_load_dword %1, %2, 0 offen lds\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ u32x4_t plane_rsrc(const float *base, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)base;
    u32x4_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = bytes;
    r.w = 0x00020000u;
    return r;
}

template <int W>
__global__ void __launch_bounds__(kLongTY * 64)
sep3d_long_kernel(const float *__restrict__ in, float *__restrict__ out, const LongParams p)
{
    constexpr int RX = W / 2;
    constexpr int NBK = (RX + 3) / 4;                 // 4-float blocks per side in the x window
    constexpr int NP = 2 * (2 * NBK + 1);             // window pairs
    constexpr int BASE = 4 * NBK - RX;                // window[BASE + c + k] = in[x + c - RX + k]
    constexpr int ROWS = kLongTY + W - 1;
    static_assert(W >= 3 && (W & 1) && ROWS <= kLongRowsMax && NBK <= 2, "long kernel: odd W, 3..17");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: raw[kLongNB][32] records | xf[32][1024] | ztab
    constexpr unsigned XF0 = kLongRawBytes;
    int *ztab = reinterpret_cast<int *>(smem + kLongRawBytes + kLongXfBytes);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    int b = blockIdx.x;
    const int total = p.nxt * p.nyt * p.nzc;
    if ((total & 7) == 0) b = (b & 7) * (total >> 3) + (b >> 3);      // an XCD gets one contiguous range of tiles
    const int per_chunk = p.nxt * p.nyt;
    const int zci = b / per_chunk;
    const int rem = b - zci * per_chunk;
    const int yt = rem / p.nxt, xt = rem - yt * p.nxt;

    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int x0 = xt * 256, y0 = yt * kLongTY;
    const int zs = zci * p.zc, ze = min(zs + p.zc, nz);
    const int ty_act = min(kLongTY, ny - y0);
    const int rows_needed = ty_act + W - 1;
    const int nlanes = min(64, (nx - x0) >> 2);
    const int last = nlanes - 1;
    const int xe = x0 + 4 * nlanes;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * 4u;
    const size_t plane_elems = (size_t)ny * (size_t)nx;
    const int zi0 = zs - p.oz;
    const int nsteps = ze - zs + W - 1;

    for (int i = threadIdx.x; i < nsteps; i += kLongTY * 64) ztab[i] = bmap<int>(zi0 + i, nz, p.mz);
    __syncthreads();

    // ---- per-lane DMA source offsets (bytes inside a plane) of this wave's two raw rows
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
    // ---- x window: LDS byte offsets of the 2 NBK + 1 blocks inside a record
    unsigned cb[2 * NBK + 1];
#pragma unroll
    for (int k = 0; k < 2 * NBK + 1; k++) {
        const int idx = lane + k - NBK;
        unsigned off;
        if (idx < 0) off = 1024u + (unsigned)(2 + idx) * 16u;                 // left halo: floats x0-8 .. x0-1
        else if (idx > last) off = 1024u + 32u + (unsigned)min(idx - last - 1, 1) * 16u;   // right halo
        else off = (unsigned)idx * 16u;
        cb[k] = (unsigned)wave * kLongRec + off;
    }
    const unsigned xfw = XF0 + (unsigned)wave * 1024u + (unsigned)lane * 16u;   // xf[wave][lane]
    const unsigned ovoff = (wave < ty_act && lane < nlanes) ? (unsigned)((y0 + wave) * nx + x0 + 4 * lane) * 4u : kOOB;

    auto issue = [&](int i, int buf) {
        // plane of step i into raw[buf]; beyond the last step: four no-fetch DMAs keep the vmcnt arithmetic uniform
        const bool live = i < nsteps;
        int zsrc = live ? ztab[i] : 0;
        zsrc = __builtin_amdgcn_readfirstlane(zsrc);
        const u32x4_t rin = plane_rsrc(in + (size_t)zsrc * plane_elems, plane_bytes);
        const unsigned rec0 = (unsigned)(buf * kLongRowsMax + wave) * kLongRec;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const unsigned rec = rec0 + (unsigned)h * 16u * kLongRec;
            dma_row16(rin, live ? vmain[h] : kOOB, rec);
            dma_halo16(rin, live ? vhalo[h] : kOOB, rec + 1024u);
        }
    };

    F4 acc[W];
#pragma unroll
    for (int k = 0; k < W; k++) acc[k] = f4_splat(0.f);

    issue(0, 0);
    issue(1, 1);
    issue(2, 2);

    constexpr int kArgBase = 2 * sizeof(void *);
    kfloats wyk = kernarg_floats(kArgBase + offsetof(LongParams, wyv));
    kfloats wzk = kernarg_floats(kArgBase + offsetof(LongParams, wzv));
    kfloats xt0 = kernarg_floats(kArgBase + offsetof(LongParams, xpair));
    kfloats xt1 = xt0 + 2 * (kStreamMaxTaps / 2 + 2);

    int buf = 0;
    for (int i0 = 0; i0 < nsteps; i0 += W) {
        static_for<W>([&](auto JJ) {
            constexpr int J = decltype(JJ)::value;
            const int i = i0 + J;
            if (i < nsteps) {
                launder(wyk); launder(wzk); launder(xt0); launder(xt1);
                // [A] this wave's part of plane i has landed (the two younger planes = 8 DMAs may stay in flight);
                // after the barrier everybody's has, and nobody reads xf any more
                asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                const unsigned bufoff = (unsigned)buf * (kLongRowsMax * kLongRec);
                // ---- x pass: raw rows wave and wave + 16 -> xf
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    f32x2 A[NP];
#pragma unroll
                    for (int k = 0; k < 2 * NBK + 1; k++) {
                        const float4 t = *reinterpret_cast<const float4 *>(smem + cb[k] + bufoff + h * 16 * kLongRec);
                        A[2 * k] = (f32x2){t.x, t.y};
                        A[2 * k + 1] = (f32x2){t.z, t.w};
                    }
                    const F4 xr = xdot_tab<W, NP, BASE>(A, xt0, xt1);
                    *reinterpret_cast<float4 *>(smem + xfw + h * 16 * 1024) = f4_to_float4(xr);
                }
                // [B] xf complete, raw[buf] free
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                issue(i + kLongNB, buf);
                buf = buf == kLongNB - 1 ? 0 : buf + 1;
                // ---- y pass: output row `wave`
                F4 yv;
                {
                    const float4 t0 = *reinterpret_cast<const float4 *>(smem + xfw);
                    yv = f4_scale(wyk[0], f4_from(t0));
#pragma unroll
                    for (int k = 1; k < W; k++) {
                        const float4 t = *reinterpret_cast<const float4 *>(smem + xfw + k * 1024);
                        yv = f4_fma(wyk[k], f4_from(t), yv);
                    }
                }
                // ---- z pass: scatter into the pending outputs; output i - k takes tap k
                acc[J] = f4_scale(wzk[0], yv);
#pragma unroll
                for (int k = 1; k < W; k++) acc[(J - k + W) % W] = f4_fma(wzk[k], yv, acc[(J - k + W) % W]);
                if (i >= W - 1) {
                    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)(out + (size_t)(zs + i - (W - 1)) * plane_elems), 0, (int)plane_bytes, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(f4_to_u32(acc[(J + 1) % W]), rout, ovoff, 0, 2);
                }
            }
        });
    }
    // the no-fetch DMAs of the last steps are still counted: drain before the LDS allocation goes away
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static int long_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int W>
static int launch_long(const float *in, float *out, LongParams &p, hipStream_t s)
{
    const size_t lds = (size_t)kLongRawBytes + kLongXfBytes + (size_t)(kLongMaxChunk + kStreamMaxTaps) * sizeof(int);
    static bool attr_done = false;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)sep3d_long_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int total = p.nxt * p.nyt * p.nzc;
    hipLaunchKernelGGL((sep3d_long_kernel<W>), dim3(total), dim3(kLongTY * 64), lds, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

static int g_long_zchunks = 0;     // test hook: number of z chunks (0 = cost model)

// Fused long-kernel path: cubic odd W in 11..17 (9 behind the test hook), origins on y / z allowed, no constant mode.
// Returns MI_ERR_UNSUPPORTED when the request is outside that (the caller runs the streaming passes).
int run_sep3d_long(const float *in, float *out, int nz, int ny, int nx, int w, const float *wx, const float *wy,
                   const float *wz, int oy, int oz, int mx, int my, int mz, hipStream_t s)
{
    if (w < 3 || w > 17 || !(w & 1)) return MI_ERR_UNSUPPORTED;
    if (mx == MI_MODE_CONSTANT || my == MI_MODE_CONSTANT || mz == MI_MODE_CONSTANT) return MI_ERR_UNSUPPORTED;
    if ((int64_t)ny * nx * 4 >= ((int64_t)1 << 31)) return MI_ERR_UNSUPPORTED;
    LongParams p;
    memset(&p, 0, sizeof(p));
    p.nx = nx; p.ny = ny; p.nz = nz;
    p.oy = oy; p.oz = oz;
    p.mx = mx; p.my = my; p.mz = mz;
    p.nxt = (nx + 255) / 256;
    p.nyt = (ny + kLongTY - 1) / kLongTY;
    for (int k = 0; k < w; k++) { p.wyv[k] = wy[k]; p.wzv[k] = wz[k]; }
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
    int best_nzc = 1;
    double best = 1e300;
    for (int nzc = 1; nzc <= nz && nzc <= 256; nzc++) {
        const int chunk = (nz + nzc - 1) / nzc;
        if (chunk > kLongMaxChunk) continue;
        const int real = (nz + chunk - 1) / chunk;
        const double rounds = (double)(((int64_t)cols * real + ncu - 1) / ncu);
        const double cost = rounds * (chunk + w - 1 + 3);
        if (cost < best) { best = cost; best_nzc = real; }
    }
    if (g_long_zchunks > 0) best_nzc = std::min(g_long_zchunks, nz);
    p.zc = (nz + best_nzc - 1) / best_nzc;
    if (p.zc > kLongMaxChunk) p.zc = kLongMaxChunk;
    p.nzc = (nz + p.zc - 1) / p.zc;
    switch (w) {
    case 9: return launch_long<9>(in, out, p, s);
    case 11: return launch_long<11>(in, out, p, s);
    case 13: return launch_long<13>(in, out, p, s);
    case 15: return launch_long<15>(in, out, p, s);
    case 17: return launch_long<17>(in, out, p, s);
    }
    return MI_ERR_UNSUPPORTED;
}

}  // namespace mi

extern "C" int mi_debug_set_long_zchunks(int n) { mi::g_long_zchunks = n; return MI_OK; }
