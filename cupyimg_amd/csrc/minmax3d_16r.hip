// minmax3d_16r.hip -- flat cubic min / max (3 / 5 / 7) of uint16 / int16 volumes whose ROWS ARE NOT A MULTIPLE OF 16 BYTES
// (what a scanner hands out on the MNI grid: 181 x 217 x 181 int16), one launch, rows taken as they lie.
//
// The 16-bit twin of minmax3d_u8r.hip (same reference path: filters.py:1373-1419, morphology.py:769-884; same design: no staging --
// these volumes live in the caches --, byte-unaligned 16-byte buffer accesses, a wave holds 64 / L rows of L granules, four output
// rows per lane from (4 + 2 r) W loads reduced along z as they arrive and then along y, the x pass through a wave-local LDS row
// buffer whose boundary samples the first lane of a row writes, the last granule stored in pieces).  Differences: a granule is
// eight samples; a dword holds two samples as they are, so v_pk_min / v_pk_max (u16 or i16) work on the loaded dwords without
// the even / odd split, and a sample's neighbours are 16-bit shifts of the SAME register sequence (v_alignbit), two dwords
// either side for the seven-wide window.
#include "nd_common.hpp"
#include "sep_common.hpp"

namespace mi {

constexpr int kS16rYB = 4;            // output rows per lane
constexpr int kS16rNW = 4;            // waves per workgroup

struct S16RagParams {
    int nx, ny, nz;
    int mz, my, mx;                  // boundary modes (filter_mode()-normalised)
    unsigned cval2;                  // fill sample x 0x00010001
    int L;                           // granules (lanes) per row
    int rpw;                         // row groups per wave = 64 / L
    int nyb;                         // blocks of kS16rYB rows along y
    int nitems;                      // nz * nyb
    unsigned vol_bytes;
};

template <bool IS_MAX, bool SIGNED> __device__ __forceinline__ unsigned pk16(unsigned a, unsigned b)
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    if constexpr (SIGNED) {
        const i16x2 x = __builtin_bit_cast(i16x2, a), y = __builtin_bit_cast(i16x2, b);
        const i16x2 r = IS_MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y);
        return __builtin_bit_cast(unsigned, r);
    } else {
        const u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
        const u16x2 r = IS_MAX ? __builtin_elementwise_max(x, y) : __builtin_elementwise_min(x, y);
        return __builtin_bit_cast(unsigned, r);
    }
}

// sixteen-bit lane shifts of a sequence held two elements per register: the register whose first element is one
// element later / earlier than `cur`'s
__device__ __forceinline__ unsigned seq_next(unsigned cur, unsigned nxt) { return __builtin_amdgcn_alignbit(nxt, cur, 16); }
__device__ __forceinline__ unsigned seq_prev(unsigned prv, unsigned cur) { return __builtin_amdgcn_alignbit(cur, prv, 16); }

template <int W, bool IS_MAX, bool SIGNED>
__global__ void __launch_bounds__(kS16rNW * 64)
mm3s16_ragged_kernel(const unsigned short *__restrict__ in, unsigned short *__restrict__ out, const S16RagParams p)
{
    constexpr int R = W / 2, YB = kS16rYB, NR = YB + 2 * R;
    // row buffers: per wave and row group [16 bytes in front of the row][16 L bytes][16 bytes behind it]
    __shared__ __attribute__((aligned(16))) unsigned char rowbuf[kS16rNW * 192 * 16];      // rpw (L + 2) <= 64 + 2 rpw <= 192 granules per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int L = p.L, nx = p.nx, ny = p.ny, nz = p.nz;
    const int g = lane / L, c = lane - g * L;
    const int item = ((int)blockIdx.x * kS16rNW + wave) * p.rpw + g;
    const bool live = g < p.rpw && item < p.nitems;
    const int z = live ? item / p.nyb : 0;
    const int y0 = live ? (item - z * p.nyb) * YB : 0;
    const int nv = min(8, nx - 8 * c);                                   // samples of this granule that belong to its row
    unsigned char *buf = rowbuf + (size_t)(wave * 192 + g * (L + 2)) * 16;
    unsigned short *bufs = reinterpret_cast<unsigned short *>(buf);

    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(p.vol_bytes + 16u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)p.vol_bytes, 0x00020000);

    // ---- z pass while loading: A[j][d] = dword d (two samples) of staged row j (y0 - R + j), reduced over z
    unsigned A[NR][4];
    int zsrc[W];
#pragma unroll
    for (int k = 0; k < W; k++) zsrc[k] = bmap_near<int>(z - R + k, nz, p.mz);
#pragma unroll
    for (int j = 0; j < NR; j++) {
        const int ysrc = bmap_near<int>(y0 - R + j, ny, p.my);
#pragma unroll
        for (int k = 0; k < W; k++) {
            u32x4 v;
            if (ysrc < 0 || zsrc[k] < 0 || !live) v = (u32x4){p.cval2, p.cval2, p.cval2, p.cval2};
            else v = __builtin_amdgcn_raw_buffer_load_b128(rin, (unsigned)((zsrc[k] * ny + ysrc) * nx + 8 * c) * 2u, 0, 0);
            const unsigned d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; q++) A[j][q] = k == 0 ? d[q] : pk16<IS_MAX, SIGNED>(A[j][q], d[q]);
        }
    }

    // boundary samples of a row along x: where sample -1 - k and sample nx + k come from (the same for every row)
    int xl[R], xr[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
        xl[k] = bmap<int>(-1 - k, nx, p.mx);
        xr[k] = bmap<int>(nx + k, nx, p.mx);
    }

#pragma unroll
    for (int t = 0; t < YB; t++) {
        // ---- y pass: rows t .. t + 2 R
        unsigned a[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            a[q] = A[t][q];
#pragma unroll
            for (int k = 1; k < W; k++) a[q] = pk16<IS_MAX, SIGNED>(a[q], A[t + k][q]);
        }
        // ---- x pass through the wave's row buffer (see minmax3d_u8r.hip for the ordering argument)
        if (live) *reinterpret_cast<u32x4 *>(buf + 16 + 16 * c) = (u32x4){a[0], a[1], a[2], a[3]};
        asm volatile("" ::: "memory");
        if (live && c == 0) {
            unsigned short lb[R], rb[R];
#pragma unroll
            for (int k = 0; k < R; k++) {
                lb[k] = xl[k] < 0 ? (unsigned short)p.cval2 : bufs[8 + xl[k]];
                rb[k] = xr[k] < 0 ? (unsigned short)p.cval2 : bufs[8 + xr[k]];
            }
#pragma unroll
            for (int k = 0; k < R; k++) {
                bufs[7 - k] = lb[k];
                bufs[8 + nx + k] = rb[k];
            }
        }
        asm volatile("" ::: "memory");
        unsigned dq[8];                                    // two dwords left of the granule, its four, two right of it
        {
            const u32x2 l = *reinterpret_cast<const u32x2 *>(buf + 8 + 16 * c);
            const u32x4 v = *reinterpret_cast<const u32x4 *>(buf + 16 + 16 * c);
            const u32x2 r = *reinterpret_cast<const u32x2 *>(buf + 32 + 16 * c);
            dq[0] = l.x; dq[1] = l.y; dq[2] = v.x; dq[3] = v.y; dq[4] = v.z; dq[5] = v.w; dq[6] = r.x; dq[7] = r.y;
        }
        asm volatile("" ::: "memory");
        unsigned res[4];
#pragma unroll
        for (int q = 2; q <= 5; q++) {
            // x_{i-1}, x_i, x_{i+1} (r = 1); + x_{i-2}, x_{i+2}: the registers either side (r = 2); + x_{i-3}, x_{i+3} (r = 3)
            unsigned m = pk16<IS_MAX, SIGNED>(dq[q], pk16<IS_MAX, SIGNED>(seq_prev(dq[q - 1], dq[q]), seq_next(dq[q], dq[q + 1])));
            if constexpr (R >= 2) m = pk16<IS_MAX, SIGNED>(m, pk16<IS_MAX, SIGNED>(dq[q - 1], dq[q + 1]));
            if constexpr (R >= 3) m = pk16<IS_MAX, SIGNED>(m, pk16<IS_MAX, SIGNED>(seq_prev(dq[q - 2], dq[q - 1]), seq_next(dq[q + 1], dq[q + 2])));
            res[q - 2] = m;
        }
        // ---- store
        const int y = y0 + t;
        if (live && y < ny) {
            const unsigned off = (unsigned)((z * ny + y) * nx + 8 * c) * 2u;
            if (nv == 8) {
                __builtin_amdgcn_raw_buffer_store_b128((u32x4){res[0], res[1], res[2], res[3]}, rout, off, 0, 0);
            } else {
                // nv = 4 a + 2 b + e samples: an 8-byte, a 4-byte and a 2-byte piece
                const unsigned o4 = (nv & 4) ? 8u : 0u, o2 = o4 + ((nv & 2) ? 4u : 0u);
                if (nv & 4) __builtin_amdgcn_raw_buffer_store_b64((u32x2){res[0], res[1]}, rout, off, 0, 0);
                if (nv & 2) __builtin_amdgcn_raw_buffer_store_b32((nv & 4) ? res[2] : res[0], rout, off + o4, 0, 0);
                if (nv & 1) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(o2 >= 8 ? (o2 >= 12 ? res[3] : res[2]) : (o2 >= 4 ? res[1] : res[0])), rout, off + o2, 0, 0);
            }
        }
    }
}

static Knob g_s16_ragged{1};

template <int W, bool IS_MAX, bool SIGNED>
static int launch_s16_ragged(const unsigned short *in, unsigned short *out, const S16RagParams &p, hipStream_t s)
{
    const int waves = (p.nitems + p.rpw - 1) / p.rpw;
    const int blocks = (waves + kS16rNW - 1) / kS16rNW;
    hipLaunchKernelGGL((mm3s16_ragged_kernel<W, IS_MAX, SIGNED>), dim3((unsigned)blocks), dim3(kS16rNW * 64), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    note_kernel("mi::mm3s16_ragged_kernel<%d,%s,%s> grid=%d (flat %d^3 %s %s on rows of %d samples as they lie: %d granules per row, %d rows per wave)",
                W, IS_MAX ? "max" : "min", SIGNED ? "int16" : "uint16", blocks, W, SIGNED ? "int16" : "uint16", IS_MAX ? "max" : "min", p.nx, p.L, p.rpw);
    return MI_OK;
}

// MI_ERR_UNSUPPORTED (nothing launched) outside the envelope: the caller (mi_minmax3d_16) goes on with its own checks.
int minmax3d_16_ragged(const mi_array *in, const mi_array *out, const int size[3], const int mode[3], int cval, int is_max,
                       hipStream_t s)
{
#define NOPE(msg) do { set_error("minmax3d_16 (ragged rows): %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (!g_s16_ragged) NOPE("switched off (mi_debug_set_s16_ragged)");
    if (in->ndim != 3) NOPE("volumes only");
    const int W = size[0];
    if (size[1] != W || size[2] != W || (W != 3 && W != 5 && W != 7)) NOPE("cubic sizes 3 / 5 / 7 only");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nx < 8 || nx > 512) NOPE("rows of 8 .. 512 samples");
    const int64_t total = nz * ny * nx;
    if (total < (1 << 15)) NOPE("small volume");
    // as for uint8 (minmax3d_u8r.hip): the re-reads hit the caches on volumes of this size only
    if (total * 2 > ((int64_t)1 << (W == 7 ? 24 : 26))) NOPE("large volume: the extended-rows route is faster");
    void *base = nullptr;
    size_t sz = 0;
    if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &sz, (hipDeviceptr_t)in->data) != hipSuccess) {
        (void)hipGetLastError();
        NOPE("the extent of the allocation is unknown");
    }
    if ((uintptr_t)base + sz < (uintptr_t)in->data + (size_t)total * 2 + 16) NOPE("no 16 readable bytes after the array");
#undef NOPE
    S16RagParams p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.mz = filter_mode(mode[0]); p.my = filter_mode(mode[1]); p.mx = filter_mode(mode[2]);
    p.cval2 = ((unsigned)cval & 0xffffu) * 0x10001u;
    p.L = (int)((nx + 7) / 8);
    p.rpw = 64 / p.L;
    p.nyb = (int)((ny + kS16rYB - 1) / kS16rYB);
    p.nitems = (int)(nz * p.nyb);
    p.vol_bytes = (unsigned)(total * 2);
    const unsigned short *ip = (const unsigned short *)in->data;
    unsigned short *op = (unsigned short *)out->data;
    const bool sg = in->dtype == MI_I16;
#define MI_S16_CASE(N)                                                                                              \
    case N:                                                                                                         \
        if (sg) return is_max ? launch_s16_ragged<N, true, true>(ip, op, p, s) : launch_s16_ragged<N, false, true>(ip, op, p, s);   \
        return is_max ? launch_s16_ragged<N, true, false>(ip, op, p, s) : launch_s16_ragged<N, false, false>(ip, op, p, s);
    switch (W) {
        MI_S16_CASE(3) MI_S16_CASE(5) MI_S16_CASE(7)
    }
#undef MI_S16_CASE
    return MI_ERR_UNSUPPORTED;
}

}  // namespace mi

extern "C" int mi_debug_set_s16_ragged(int on) { mi::g_s16_ragged = on; return MI_OK; }
