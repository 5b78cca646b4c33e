// minmax3d_u8r.hip -- flat cubic min / max (3 / 5 / 7) of uint8 volumes whose ROWS ARE NOT A MULTIPLE OF 16 BYTES
// (181 x 217 x 181, 91 x 109 x 91, ...), one launch, rows taken as they lie.
//
// Reference path replaced: minimum_filter / maximum_filter / grey_erosion / grey_dilation with a flat cubic footprint,
// cupyimg/scipy/ndimage/filters.py:1373-1419 (three 1-D launches, :1478-1507) and morphology.py:769-884.
// Until round 6 such volumes went through mi_extend_rows -> the LDS-DMA kernel (mm3u8_split_kernel, which needs rows
// of whole 16-byte granules) -> mi_crop_rows: three launches, 55-65 us on an MNI-grid volume whatever the window
// (profiles/r6_ragged_minmax.txt holds the float32 twin of this file, sep3d_lean_kernel<..., ragged, min>).
//
// Design.  These volumes are SMALL (7 MB for 181 x 217 x 181: they live in the L2 / Infinity Cache), so the kernel trades
// re-reads that hit the cache for having no staging at all:
//   * 16-byte buffer loads and stores take any BYTE alignment on this chip (probed for the binary kernels, bitmorph3d.hip),
//     so lane c of a row group holds bytes 16 c .. 16 c + 15 of its row wherever the row starts; a wave holds 64 / L row
//     groups (L = granules per row), so short rows still fill the lanes.
//   * A lane produces YB = 4 consecutive output rows (same z, same granule): it loads the W input rows (z - r .. z + r) of
//     each of the YB + 2 r rows y - r .. y + YB - 1 + r -- mapped by the boundary mode, or the fill value -- and reduces them
//     along z as it goes, then along y: (YB + 2 r) W / YB loads per output granule (4.5 / 10 / 17.5) instead of W^2.
//   * Bytes are compared as 16-bit lanes: a dword is split into its even bytes (x & 0x00ff00ff) and its odd bytes
//     ((x >> 8) & 0x00ff00ff) once when it is loaded, and v_pk_min_u16 / v_pk_max_u16 work on those.
//   * x pass: the y / z-reduced granule goes to a row buffer in LDS (wave-local: a row never spans waves, LDS operations
//     of a wave complete in order, so there is no barrier), the first lane of the row writes the 2 r boundary bytes the
//     mode prescribes left of byte 0 and right of byte nx - 1 (which overwrites what the last granule holds beyond its row:
//     the head of the next row), and every lane reads back its granule with one dword either side.  In the split form
//     a byte's neighbours are 16-bit lane shifts of the other parity (v_alignbit).
//   * The last granule of a row stores its nx % 16 bytes in 8 / 4 / 2 / 1-byte pieces.
// The loads of the last granule of the last row reach up to 15 bytes past the volume: the descriptor is 16 bytes longer
// than the volume and the host takes this route only when hipMemGetAddressRange shows those bytes inside the allocation.
#include "nd_common.hpp"
#include "sep_common.hpp"

namespace mi {

constexpr int kU8rYB = 4;            // output rows per lane
constexpr int kU8rNW = 4;            // waves per workgroup

struct U8RagParams {
    int nx, ny, nz;
    int mz, my, mx;                  // boundary modes (filter_mode()-normalised)
    unsigned cval4;                  // fill byte x 0x01010101
    int L;                           // granules (lanes) per row
    int rpw;                         // row groups per wave = 64 / L
    int nyb;                         // blocks of kU8rYB rows along y
    int nitems;                      // nz * nyb
    unsigned vol_bytes;
};

template <bool IS_MAX> __device__ __forceinline__ unsigned pk16(unsigned a, unsigned b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    const u16x2 r = IS_MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y);
    return __builtin_bit_cast(unsigned, r);
}

// sixteen-bit lane shifts of a sequence held two elements per register: the register whose first element is one
// element later / earlier than `cur`'s
__device__ __forceinline__ unsigned seq_next(unsigned cur, unsigned nxt) { return __builtin_amdgcn_alignbit(nxt, cur, 16); }
__device__ __forceinline__ unsigned seq_prev(unsigned prv, unsigned cur) { return __builtin_amdgcn_alignbit(cur, prv, 16); }

template <int W, bool IS_MAX>
__global__ void __launch_bounds__(kU8rNW * 64)
mm3u8_ragged_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, const U8RagParams p)
{
    constexpr int R = W / 2, YB = kU8rYB, NR = YB + 2 * R;
    // row buffers: per wave and row group [16 bytes in front of the row][16 L bytes][16 bytes behind it]
    __shared__ __attribute__((aligned(16))) unsigned char rowbuf[kU8rNW * 192 * 16];      // rpw (L + 2) <= 64 + 2 rpw <= 192 granules per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = p.L, nx = p.nx, ny = p.ny, nz = p.nz;
    const int g = lane / L, c = lane - g * L;
    const int item = ((int)blockIdx.x * kU8rNW + wave) * p.rpw + g;
    const bool live = g < p.rpw && item < p.nitems;
    const int z = live ? item / p.nyb : 0;
    const int y0 = live ? (item - z * p.nyb) * YB : 0;
    const int nv = min(16, nx - 16 * c);                                  // bytes of this granule that belong to its row
    unsigned char *buf = rowbuf + (size_t)(wave * 192 + g * (L + 2)) * 16;

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(p.vol_bytes + 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)p.vol_bytes, 0x00020000);
    const unsigned M = 0x00ff00ffu;

    // ---- z pass while loading: E[j][d] / O[j][d] = even / odd bytes of dword d of staged row j (y0 - R + j), reduced over z
    unsigned E[NR][4], O[NR][4];
    int zsrc[W];
#pragma unroll
    for (int k = 0; k < W; k++) zsrc[k] = bmap_near<int>(z - R + k, nz, p.mz);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int ysrc = bmap_near<int>(y0 - R + j, ny, p.my);
#pragma unroll
        for (int k = 0; k < W; k++) {
            u32x4 v;
            if (ysrc < 0 || zsrc[k] < 0 || !live) v = (u32x4){p.cval4, p.cval4, p.cval4, p.cval4};
            else v = __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)((zsrc[k] * ny + ysrc) * nx + 16 * c), 0, 0);
            const unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const unsigned e = d[q] & M, o = (d[q] >> 8) & M;
                E[j][q] = k == 0 ? e : pk16<IS_MAX>(E[j][q], e);
                O[j][q] = k == 0 ? o : pk16<IS_MAX>(O[j][q], o);
            }
        }
    }

    // boundary bytes of a row along x: where byte -k and byte nx - 1 + k come from (the same for every row)
    int xl[R], xr[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        xl[k] = bmap<int>(-1 - k, nx, p.mx);
        xr[k] = bmap<int>(nx + k, nx, p.mx);
    }

#pragma unroll
    for (int t = 0; t < YB; t++) {
        // ---- y pass: rows t .. t + 2 R
        unsigned e[4], o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            e[q] = E[t][q]; o[q] = O[t][q];
#pragma unroll
            for (int k = 1; k < W; k++) { e[q] = pk16<IS_MAX>(e[q], E[t + k][q]); o[q] = pk16<IS_MAX>(o[q], O[t + k][q]); }
        }
        // ---- x pass through the wave's row buffer
        if (live) {
            u32x4 v;
            v.x = e[0] | (o[0] << 8); v.y = e[1] | (o[1] << 8); v.z = e[2] | (o[2] << 8); v.w = e[3] | (o[3] << 8);
            *reinterpret_cast<u32x4 *>(buf + 16 + 16 * c) = v;
        }
        // (compiler barriers: the bytes another lane writes into this lane's granule must be re-read, not forwarded from
        // the store above; the hardware carries out a wave's LDS operations in order)
        asm volatile("" ::: "memory");
        if (live && c == 0) {
            unsigned char lb[R], rb[R];
#pragma unroll
            for (int k = 0; k < R; k++) {
                lb[k] = xl[k] < 0 ? (unsigned char)p.cval4 : buf[16 + xl[k]];
                rb[k] = xr[k] < 0 ? (unsigned char)p.cval4 : buf[16 + xr[k]];
            }
#pragma unroll
            for (int k = 0; k < R; k++) {
                buf[15 - k] = lb[k];
                buf[16 + nx + k] = rb[k];
            }
        }
        asm volatile("" ::: "memory");
        unsigned dq[6];
        {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(buf + 16 + 16 * c);
            dq[0] = *reinterpret_cast<const unsigned *>(buf + 12 + 16 * c);
            dq[1] = v.x; dq[2] = v.y; dq[3] = v.z; dq[4] = v.w;
            dq[5] = *reinterpret_cast<const unsigned *>(buf + 32 + 16 * c);
        }
        asm volatile("" ::: "memory");
        unsigned se[6], so[6];
#pragma unroll
        for (int q = 0; q < 6; q++) { se[q] = dq[q] & M; so[q] = (dq[q] >> 8) & M; }
        unsigned res[4];
#pragma unroll
        for (int q = 1; q <= 4; q++) {
            // even bytes: e_i with o_{i-1}, o_i (r = 1); + e_{i-1}, e_{i+1} (r = 2); + o_{i-2}, o_{i+1} (r = 3)
            unsigned a = pk16<IS_MAX>(se[q], pk16<IS_MAX>(so[q], seq_prev(so[q - 1], so[q])));
            // odd bytes: o_i with e_i, e_{i+1} (r = 1); + o_{i-1}, o_{i+1} (r = 2); + e_{i-1}, e_{i+2} (r = 3)
            unsigned b = pk16<IS_MAX>(so[q], pk16<IS_MAX>(se[q], seq_next(se[q], se[q + 1])));
            if constexpr (R >= 2) {
                a = pk16<IS_MAX>(a, pk16<IS_MAX>(seq_prev(se[q - 1], se[q]), seq_next(se[q], se[q + 1])));
                b = pk16<IS_MAX>(b, pk16<IS_MAX>(seq_prev(so[q - 1], so[q]), seq_next(so[q], so[q + 1])));
            }
            if constexpr (R >= 3) {
                a = pk16<IS_MAX>(a, pk16<IS_MAX>(so[q - 1], seq_next(so[q], so[q + 1])));
                b = pk16<IS_MAX>(b, pk16<IS_MAX>(seq_prev(se[q - 1], se[q]), se[q + 1]));
            }
            res[q - 1] = a | (b << 8);
        }
        // ---- store
        const int y = y0 + t;
        if (live && y < ny) {
            const unsigned off = (unsigned)((z * ny + y) * nx + 16 * c);
            if (nv == 16) {
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){res[0], res[1], res[2], res[3]}, rout, off, 0, 0);
            } else {
                const unsigned o4 = (nv & 8) ? 8u : 0u, o2 = o4 + ((nv & 4) ? 4u : 0u), o1 = o2 + ((nv & 2) ? 2u : 0u);
                const unsigned d4 = (nv & 8) ? res[2] : res[0];
                const unsigned d2 = o2 >= 8 ? (o2 >= 12 ? res[3] : res[2]) : (o2 >= 4 ? res[1] : res[0]);
                const unsigned d1 = o1 >= 8 ? (o1 >= 12 ? res[3] : res[2]) : (o1 >= 4 ? res[1] : res[0]);
                if (nv & 8) __builtin_amdgcn_raw_buffer_store_b64((u32x2){res[0], res[1]}, rout, off, 0, 0);
                if (nv & 4) __builtin_amdgcn_raw_buffer_store_b32(d4, rout, off + o4, 0, 0);
                if (nv & 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)d2, rout, off + o2, 0, 0);
                if (nv & 1) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(d1 >> (8 * (o1 & 3u))), rout, off + o1, 0, 0);
            }
        }
    }
}

static Knob g_u8_ragged{1};

template <int W, bool IS_MAX>
static int launch_u8_ragged(const unsigned char *in, unsigned char *out, const U8RagParams &p, hipStream_t s)
{
    const int waves = (p.nitems + p.rpw - 1) / p.rpw;
    const int blocks = (waves + kU8rNW - 1) / kU8rNW;
    hipLaunchKernelGGL((mm3u8_ragged_kernel<W, IS_MAX>), dim3((unsigned)blocks), dim3(kU8rNW * 64), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    note_kernel("mi::mm3u8_ragged_kernel<%d,%s> grid=%d (flat %d^3 uint8 %s on rows of %d bytes as they lie: %d granules per row, %d rows per wave)",
                W, IS_MAX ? "max" : "min", blocks, W, IS_MAX ? "max" : "min", p.nx, p.L, p.rpw);
    return MI_OK;
}

// MI_ERR_UNSUPPORTED (nothing launched) outside the envelope: the caller (mi_minmax3d_u8) answers the same and the Python
// layer goes on to the extended-rows route.
int minmax3d_u8_ragged(const mi_array *in, const mi_array *out, const int size[3], const int mode[3], int cval, int is_max,
                       hipStream_t s)
{
#define NOPE(msg) do { set_error("minmax3d_u8 (ragged rows): %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (!g_u8_ragged) NOPE("switched off (mi_debug_set_u8_ragged)");
    const int W = size[0];
    if (size[1] != W || size[2] != W || (W != 3 && W != 5 && W != 7)) NOPE("cubic sizes 3 / 5 / 7 only");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nx < 8 || nx > 1024) NOPE("rows of 8 .. 1024 bytes");
    const int64_t total = nz * ny * nx;
    if (total < (1 << 15)) NOPE("small volume");
    // the (YB + 2 r) W / YB loads per granule (4.5 / 10 / 17.5) hit the caches on small volumes; measured against the
    // extended-rows route (profiles/r6_ragged_minmax.txt): sizes 3 / 5 win up to 400^3 at least, size 7 up to 256^3
    if (total > ((int64_t)1 << (W == 7 ? 24 : 26))) NOPE("large volume: the extended-rows route is faster");
    void *base = nullptr;
    size_t sz = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &sz, (hipDeviceptr_t)in->data) != hipSuccess) {
        (void)hipGetLastError();
        NOPE("the extent of the allocation is unknown");
    }
    if ((uintptr_t)base + sz < (uintptr_t)in->data + (size_t)total + 16) NOPE("no 16 readable bytes after the array");
#undef NOPE
    U8RagParams p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.mz = filter_mode(mode[0]); p.my = filter_mode(mode[1]); p.mx = filter_mode(mode[2]);
    p.cval4 = (unsigned)cval * 0x01010101u;
    p.L = (int)((nx + 15) / 16);
    p.rpw = 64 / p.L;
    p.nyb = (int)((ny + kU8rYB - 1) / kU8rYB);
    p.nitems = (int)(nz * p.nyb);
    p.vol_bytes = (unsigned)total;
    const unsigned char *ip = (const unsigned char *)in->data;
    unsigned char *op = (unsigned char *)out->data;
    switch (W * 2 + (is_max ? 1 : 0)) {
    case 6: return launch_u8_ragged<3, false>(ip, op, p, s);
    case 7: return launch_u8_ragged<3, true>(ip, op, p, s);
    case 10: return launch_u8_ragged<5, false>(ip, op, p, s);
    case 11: return launch_u8_ragged<5, true>(ip, op, p, s);
    case 14: return launch_u8_ragged<7, false>(ip, op, p, s);
    default: return launch_u8_ragged<7, true>(ip, op, p, s);
    }
}

}  // namespace mi

extern "C" int mi_debug_set_u8_ragged(int on) { mi::g_u8_ragged = on; return MI_OK; }
