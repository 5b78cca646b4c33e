// binary3d.hip -- LDS-tiled binary erosion / dilation for 1-byte volumes.
//
// Reference path replaced: cupyimg/scipy/ndimage/morphology.py:41-128 (kernel:
// one global load per structure tap per voxel), launch at :292-322.  Same
// result as binary3_kernel in binary.hip (output true unless a set structure
// tap sees a false voxel; outside the array a tap sees border_value; `invert`
// expresses dilation; masked-out voxels keep the input; `changed` flag).
//
// Design: the 2.5-D blocking of stencil3d.hip on bytes.  A workgroup (8 waves)
// owns a 1024 x 16 column and streams along z; the planes the structure spans
// sit in an LDS ring of sz + 1 slots, staged as "good" masks -- 0xFF where the
// voxel lets the output stay true, 0x00 where it forces false, border bytes
// already resolved.  A lane owns 16 x-consecutive voxels (four dwords); a tap
// is one v_alignbyte per dword to shift the row by dx bytes plus one v_and:
// 0.5 instructions per voxel and tap, no branches, no global traffic in the tap
// loop.  HBM traffic: 2 B/voxel (+1 with a mask).
#include "nd_common.hpp"
#include "sep_common.hpp"

namespace mi {

constexpr int kBnNW = 8;
constexpr int kBnTY = 16;
// bytes per LDS row: 16 halo + tile width (64 lanes x 4 ND bytes) + 16 halo
constexpr int bn_pitch(int nd) { return 32 + 256 * nd; }
constexpr int kBnMaxRows = 81;      // (tz, ty) pairs

struct Binary3Params {
    int nx, ny, nz;
    int wz, wy;                 // structure extent along z, y (x extent = WX of the kernel, zero-padded)
    int oz, oy;
    int border_good;            // what a tap outside the array contributes (after `invert`): 1 = keeps true
    int invert;
    int zc, nzc, nxt, nyt;
    unsigned mask[kBnMaxRows];  // per (tz, ty): bit tx set = structure element set
};

// bytes != 0 -> 0xFF, == 0 -> 0x00, four at a time
__device__ __forceinline__ unsigned nonzero_bytes(unsigned v)
{
    const unsigned t = (v | ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
    return (t >> 7) * 0xffu;
}

// ND = dwords (4 voxels each) per lane: tile width 256 ND voxels, chosen by the x extent
template <int ND>
__device__ __forceinline__ void load_dwords(const __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned (&d)[ND])
{
    if constexpr (ND == 1) {
        d[0] = __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0);
    } else if constexpr (ND == 2) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
        d[0] = v.x; d[1] = v.y;
    } else {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}
template <int ND>
__device__ __forceinline__ void store_dwords(const __amdgpu_buffer_rsrc_t r, unsigned voff, const unsigned (&d)[ND])
{
    if constexpr (ND == 1) __builtin_amdgcn_raw_buffer_store_b32(d[0], r, voff, 0, 0);
    else if constexpr (ND == 2) __builtin_amdgcn_raw_buffer_store_b64((u32x2){d[0], d[1]}, r, voff, 0, 0);
    else __builtin_amdgcn_raw_buffer_store_b128((u32x4){d[0], d[1], d[2], d[3]}, r, voff, 0, 0);
}
template <int ND>
__device__ __forceinline__ void load_lds_dwords(const unsigned char *p, unsigned *d)
{
    if constexpr (ND == 1) {
        d[0] = *reinterpret_cast<const unsigned *>(p);
    } else if constexpr (ND == 2) {
        const u32x2 v = *reinterpret_cast<const u32x2 *>(p);
        d[0] = v.x; d[1] = v.y;
    } else {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(p);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}
template <int ND>
__device__ __forceinline__ void store_lds_dwords(unsigned char *p, const unsigned (&d)[ND])
{
    if constexpr (ND == 1) *reinterpret_cast<unsigned *>(p) = d[0];
    else if constexpr (ND == 2) *reinterpret_cast<u32x2 *>(p) = (u32x2){d[0], d[1]};
    else *reinterpret_cast<u32x4 *>(p) = (u32x4){d[0], d[1], d[2], d[3]};
}

template <int WX, bool HAS_MASK, int ND>
__global__ void __launch_bounds__(kBnNW * 64)
binary3_tiled_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                     const unsigned char *__restrict__ msk, const Binary3Params p, int32_t *changed)
{
    constexpr int RX = WX / 2;
    constexpr int TY = kBnTY;
    constexpr int kBnPitch = bn_pitch(ND);
    constexpr int LB = 4 * ND;                             // bytes per lane
    typedef unsigned int lanev __attribute__((ext_vector_type(ND)));
    constexpr int RW = TY / kBnNW;                         // output rows per wave
    constexpr int RPW = (TY + 8 + kBnNW - 1) / kBnNW;      // staged rows per wave (wy <= 9)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned char *ring = reinterpret_cast<unsigned char *>(smem);     // [wz + 1][rows_l][kBnPitch]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b = blockIdx.x;
    const int total = p.nxt * p.nyt * p.nzc;
    if ((total & 7) == 0) b = (b & 7) * (total >> 3) + (b >> 3);
    const int per_chunk = p.nxt * p.nyt;
    const int zci = b / per_chunk;
    const int rem = b - zci * per_chunk;
    const int yt = rem / p.nxt, xt = rem - yt * p.nxt;

    const int nx = p.nx, ny = p.ny, nz = p.nz, wz = p.wz, wy = p.wy;
    const int x0 = xt * (64 * LB), y0 = yt * TY;
    const int zs = zci * p.zc, ze = min(zs + p.zc, nz);
    const int nout = ze - zs;
    const int ty_act = min(TY, ny - y0);
    const int nlanes = min(64, (nx - x0) / LB);
    const int rows_l = TY + wy - 1;
    const int slot_bytes = rows_l * kBnPitch;
    const int nslots = wz + 1;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx;
    const size_t plane_elems = (size_t)ny * (size_t)nx;
    const unsigned inv = p.invert ? 0xffffffffu : 0u;      // good = nonzero ^ inv
    const unsigned border = p.border_good ? 0xffffffffu : 0u;

    // staging recipe: wave w stages rows w, w + 8, ...; lanes 0..7 also fetch one halo byte each
    unsigned voff_main[RPW], voff_halo[RPW];
    bool row_out[RPW];
    int lds_row[RPW];
    const bool halo_lane = lane < 8;
    const int xh = lane < 4 ? x0 - 4 + lane : x0 + LB * nlanes + (lane - 4);
    const bool x_in = halo_lane && xh >= 0 && xh < nx;
    const int halo_pos = lane < 4 ? 12 + lane : 16 + LB * nlanes + (lane - 4);
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int j = wave + kBnNW * k;
        const int ysrc = y0 - p.oy + j;
        const bool y_in = j < rows_l && ysrc >= 0 && ysrc < ny;
        row_out[k] = !y_in;
        lds_row[k] = j < rows_l ? j * kBnPitch : -1;
        voff_main[k] = (y_in && lane < nlanes) ? (unsigned)(ysrc * nx + x0 + LB * lane) : kOOB;
        voff_halo[k] = (y_in && x_in) ? (unsigned)(ysrc * nx + xh) : kOOB;
    }

    unsigned pm[RPW][ND];
    unsigned ph[RPW];
    bool pout = false;
    auto fetch = [&](int q) {
        int zsrc = zs - p.oz + q;
        pout = (unsigned)zsrc >= (unsigned)nz;
        zsrc = __builtin_amdgcn_readfirstlane(pout ? 0 : zsrc);
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(in + (size_t)zsrc * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            load_dwords<ND>(rin, pout ? kOOB : voff_main[k], pm[k]);
            ph[k] = __builtin_amdgcn_raw_buffer_load_b8(rin, pout ? kOOB : voff_halo[k], 0, 0);
        }
    };
    auto stage = [&](int q) {
        unsigned char *slot = ring + (q % nslots) * slot_bytes;
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            if (lds_row[k] < 0) continue;
            const bool o = pout || row_out[k];
            unsigned g[ND];
#pragma unroll
            for (int c = 0; c < ND; c++) g[c] = o ? border : nonzero_bytes(pm[k][c]) ^ inv;
            if (lane < nlanes) store_lds_dwords<ND>(slot + lds_row[k] + 16 + LB * lane, g);
            if (halo_lane) {
                const unsigned hb = (o || !x_in) ? border : ((ph[k] != 0 ? 0xffu : 0u) ^ inv);
                slot[lds_row[k] + halo_pos] = (unsigned char)hb;
            }
        }
    };

    for (int q = 0; q < wz; q++) {
        fetch(q);
        stage(q);
    }
    __syncthreads();

    const int r0 = wave * RW;
    unsigned ovoff[RW];
#pragma unroll
    for (int rr = 0; rr < RW; rr++)
        ovoff[rr] = (r0 + rr < ty_act && lane < nlanes) ? (unsigned)((y0 + r0 + rr) * nx + x0 + LB * lane) : kOOB;
    bool any_change = false;

    for (int s = 0; s < nout; s++) {
        const bool more = s + 1 < nout;
        if (more) fetch(s + wz);

        unsigned acc[RW][ND];
#pragma unroll
        for (int rr = 0; rr < RW; rr++)
#pragma unroll
            for (int c = 0; c < ND; c++) acc[rr][c] = 0xffffffffu;

        for (int tz = 0; tz < wz; tz++) {
            const unsigned char *slot = ring + ((s + tz) % nslots) * slot_bytes + 16 + LB * lane;
            for (int i = 0; i < RW + wy - 1; i++) {
                unsigned m[RW];
                bool live[RW];
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {
                    const int ty = i - rr;
                    live[rr] = ty >= 0 && ty < wy;
                    m[rr] = p.mask[tz * wy + min(max(ty, 0), wy - 1)];
                }
                const unsigned char *rowp = slot + (r0 + i) * kBnPitch;
                unsigned w6[ND + 2];                       // bytes x - 4 .. x + LB + 3
                w6[0] = *reinterpret_cast<const unsigned *>(rowp - 4);
                load_lds_dwords<ND>(rowp, &w6[1]);
                w6[ND + 1] = *reinterpret_cast<const unsigned *>(rowp + LB);
#pragma unroll
                for (int rr = 0; rr < RW; rr++) {
                    if (!live[rr]) continue;
#pragma unroll
                    for (int tx = 0; tx < WX; tx++) {
                        if (!(m[rr] >> tx & 1u)) continue;
                        const int dx = tx - RX;               // compile-time after unrolling
                        const int sh = (dx + 4) & 3, j = (4 + dx) >> 2;   // window byte 4 + dx + 4k = 4 (j + k) + sh
#pragma unroll
                        for (int c = 0; c < ND; c++)
                            acc[rr][c] &= sh == 0 ? w6[j + c] : __builtin_amdgcn_alignbyte(w6[j + c + 1], w6[j + c], sh);
                    }
                }
            }
        }

        // current voxels (good space) of the output rows: centre plane = input plane z = chunk plane s + oz
        const unsigned char *cslot = ring + ((s + p.oz) % nslots) * slot_bytes + 16 + LB * lane;
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(out + (size_t)(zs + s) * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
        for (int rr = 0; rr < RW; rr++) {
            unsigned cur[ND], res[ND];
            load_lds_dwords<ND>(cslot + (r0 + rr + p.oy) * kBnPitch, cur);
#pragma unroll
            for (int c = 0; c < ND; c++) {
                cur[c] = (cur[c] ^ inv) & 0x01010101u;
                res[c] = (acc[rr][c] ^ inv) & 0x01010101u;
            }
            if constexpr (HAS_MASK) {
                const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(msk + (size_t)(zs + s) * plane_elems), 0, (int)plane_bytes, 0x00020000);
                unsigned mk[ND];
                load_dwords<ND>(rm, ovoff[rr], mk);
#pragma unroll
                for (int c = 0; c < ND; c++) {
                    const unsigned mm = nonzero_bytes(mk[c]);
                    res[c] = (res[c] & mm) | (cur[c] & ~mm);
                }
            }
            unsigned diff = 0;
#pragma unroll
            for (int c = 0; c < ND; c++) diff |= res[c] ^ cur[c];
            if (ovoff[rr] != kOOB) any_change |= diff != 0;
            store_dwords<ND>(rout, ovoff[rr], res);
        }
        if (more) stage(s + wz);
        __syncthreads();
    }
    if (changed && __any(any_change) && lane == 0) atomicOr(changed, 1);
}

template <int WX, bool HAS_MASK, int ND>
static int launch_binary3(const unsigned char *in, unsigned char *out, const unsigned char *msk, Binary3Params &p,
                          int32_t *changed, hipStream_t s)
{
    const size_t lds = (size_t)(p.wz + 1) * (kBnTY + p.wy - 1) * bn_pitch(ND);
    static PerDeviceOnce attr;
    if (!attr) {
        MI_HIP(hipFuncSetAttribute((const void *)binary3_tiled_kernel<WX, HAS_MASK, ND>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        attr = true;
    }
    p.nxt = (p.nx + 256 * ND - 1) / (256 * ND);
    p.nyt = (p.ny + kBnTY - 1) / kBnTY;
    const int cus = device_cus();
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds));
    const int64_t slots = (int64_t)cus * per_cu, tiles = (int64_t)p.nxt * p.nyt;
    double best = 1e300;
    int best_nzc = 1;
    for (int nzc = 1; nzc <= std::min(p.nz, 128); nzc++) {
        const int chunk = (p.nz + nzc - 1) / nzc;
        const int real = (p.nz + chunk - 1) / chunk;
        const double rounds = (double)((tiles * real + slots - 1) / slots);
        const double cost = rounds * (chunk + p.wz - 1 + 2.0);
        if (cost < best) { best = cost; best_nzc = real; }
    }
    p.zc = (p.nz + best_nzc - 1) / best_nzc;
    p.nzc = (p.nz + p.zc - 1) / p.zc;
    const int64_t total = tiles * p.nzc;
    if (total > 0x7fffffff) { set_error("binary3: too many tiles"); return MI_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL((binary3_tiled_kernel<WX, HAS_MASK, ND>), dim3((unsigned)total), dim3(kBnNW * 64), lds, s, in, out, msk,
                       p, changed);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// Tries the tiled kernel; MI_ERR_UNSUPPORTED (nothing launched) outside its envelope.
int binary3_tiled(const mi_array *in, const mi_array *out, const uint8_t *structure, const int64_t *sshape,
                  const int *origins, const mi_array *mask, int border_value, int invert, int32_t *changed,
                  hipStream_t s)
{
#define NOPE(msg) do { set_error("binary3 tiled: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (dtype_size(in->dtype) != 1 || dtype_size(out->dtype) != 1) NOPE("1-byte volumes only");
    if (in->ndim < 2 || in->ndim > 3) NOPE("2-D / 3-D only");
    const int pad = 3 - in->ndim;
    int64_t shape[3] = {1, 1, 1};
    int w[3] = {1, 1, 1}, off[3] = {0, 0, 0};
    for (int d = 0; d < in->ndim; d++) {
        shape[pad + d] = in->shape[d];
        if (sshape[d] < 1 || sshape[d] > 9) NOPE("structure extent > 9");
        w[pad + d] = (int)sshape[d];
        off[pad + d] = (int)(sshape[d] / 2 + origins[d]);
        if (off[pad + d] < 0 || off[pad + d] >= sshape[d]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    const int64_t nz = shape[0], ny = shape[1], nx = shape[2];
    if (nx < 16 || (nx & 3)) NOPE("x extent must be a multiple of 4, >= 16");
    const int nd = (nx >= 1024 && !(nx & 15)) ? 4 : ((nx >= 512 && !(nx & 7)) ? 2 : 1);
    if (ny * nx >= ((int64_t)1 << 31) || nz > (1 << 24) || ny > (1 << 24)) NOPE("plane too large");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15) || (mask && ((uintptr_t)mask->data & 15)))
        NOPE("needs 16-byte aligned data");
    const int reach = std::max(off[2], w[2] - 1 - off[2]);
    const int WXk = 2 * reach + 1;
    if (WXk > 9) NOPE("x reach > 4");
    if (w[0] * w[1] > kBnMaxRows) NOPE("structure too large");
    const size_t lds = (size_t)(w[0] + 1) * (kBnTY + w[1] - 1) * bn_pitch(nd);
    if (lds > 150 * 1024) NOPE("structure does not fit LDS");

    Binary3Params p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.wz = w[0]; p.wy = w[1];
    p.oz = off[0]; p.oy = off[1];
    p.invert = invert != 0;
    p.border_good = invert ? !border_value : (border_value != 0);
    for (int tz = 0; tz < w[0]; tz++)
        for (int ty = 0; ty < w[1]; ty++)
            for (int tx = 0; tx < w[2]; tx++)
                if (structure[((int64_t)tz * w[1] + ty) * w[2] + tx])
                    p.mask[tz * w[1] + ty] |= 1u << (tx - off[2] + reach);
    const unsigned char *ip = (const unsigned char *)in->data;
    unsigned char *op = (unsigned char *)out->data;
    const unsigned char *mp = mask ? (const unsigned char *)mask->data : nullptr;
#define GO2(WXV, NDV) return mp ? launch_binary3<WXV, true, NDV>(ip, op, mp, p, changed, s) : launch_binary3<WXV, false, NDV>(ip, op, mp, p, changed, s)
#define GO(WXV) do { if (nd == 4) { GO2(WXV, 4); } else if (nd == 2) { GO2(WXV, 2); } else { GO2(WXV, 1); } } while (0)
    switch (WXk) {
    case 1: GO(1);
    case 3: GO(3);
    case 5: GO(5);
    case 7: GO(7);
    default: GO(9);
    }
    return MI_ERR_INTERNAL;
#undef GO
#undef GO2
#undef NOPE
}

}  // namespace mi
