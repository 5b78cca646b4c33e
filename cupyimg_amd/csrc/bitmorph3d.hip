// bitmorph3d.hip -- binary erosion / dilation of 1-byte volumes on ONE BIT per voxel, k iterations per tile residency.
//
// Reference path replaced: cupyimg/scipy/ndimage/morphology.py:41-128 (kernel: one global load per structure tap and
// voxel) and its host loop :292-322 (one launch + one host synchronisation per iteration).  Same results as
// binary3_tiled_kernel (binary3d.hip) / binary3_kernel (binary.hip): an output voxel stays true unless a set structure
// tap sees a false voxel; outside the array a tap sees border_value; `invert` expresses dilation (erosion of the
// complement, complemented); voxels where the mask is false keep their value; `changed` per iteration.
//
// Design (round 6).  The byte kernel spends 0.5 VALU instructions per voxel AND TAP and keeps whole byte planes in LDS;
// here a plane of a tile is 1 bit per voxel:
//
//   * a workgroup (4 waves) owns a tile of TY rows x (whole rows | 1024 voxels) and streams along z;
//   * stage 0: every lane loads 16 bytes (one coalesced 16-byte load), turns them into 16 bits -- "good" space: bit = the
//     voxel lets the output stay true, i.e. (byte != 0) ^ invert -- with one carry trick per dword and v_dot4_u32_u8 as
//     the bit gather, and writes them as ONE ds_write_b16: a staged row of 1024 voxels is 128 bytes of LDS;
//   * stage j = 1 .. k (one per fused iteration): a lane owns a 32-voxel word; a structure row (dz, dy) costs three LDS
//     dwords (left, centre, right word), a tap one v_alignbit + v_and on 32 voxels; the result goes to stage j's own ring;
//   * output: every lane reads its 16 bits of stage k, spreads them to 16 bytes (v_mul_u32_u24 by 0x204081 per nibble)
//     and stores 16 bytes.
//
// Stage j works on the plane stage j - 1 finished one step EARLIER, so a step (= one input plane) needs ONE barrier
// whatever k is; a chunk of ZC output planes takes ZC + k * wz + 1 steps and stages TY + k (wy - 1) rows: the halo of k
// iterations is paid once in LDS-resident bits, not k times in HBM bytes.  Out-of-array rows / planes / x positions are
// re-set to the border bit at EVERY stage (a tap outside the array sees border_value in every iteration, not the
// eroded border).  HBM traffic: 2 B/voxel for any k (+1 with a mask).
#include "nd_common.hpp"
#include "sep_common.hpp"

namespace mi {

constexpr int kBmNT = 256;            // threads per workgroup
constexpr int kBmMaxRows = 96;        // structure rows (dz, dy) with at least one tap
constexpr int kBmMaxK = 8;            // fused iterations per launch
constexpr int kBmMaxLds = 64 * 1024;  // per workgroup (two or more workgroups per CU)

struct BitMorphParams {
    int nx, ny, nz;
    int wz, oz, hz;         // structure extent along z, lo reach (w/2 + origin), hi reach
    int oy, hy, ox;
    int nrows, k;
    int border, invert;
    int ty, gy;             // output rows per tile, staged rows = ty + k (oy + hy)
    int txw, gxw, hlw;      // output words (32 voxels) per tile row, staged words per row, halo words on the left
    int pitch;              // LDS words per staged row: 1 pad + gxw + 1 pad (+ skew)
    int zc, nzc, nxt, nyt;
    int ns, nms;            // ring slots per stage (wz + 1), slots of the mask ring
    unsigned short rowpos[kBmMaxRows];   // tz | ty << 8
    unsigned rowmask[kBmMaxRows];        // bit tx set = structure[tz][ty][tx]
};

// 16 bytes -> 16 bits, bit i = (byte i != 0)
__device__ __forceinline__ unsigned pack16(const u32x4 v)
{
    auto top = [](unsigned x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; };   // 0x80 per nonzero byte
    unsigned lo = __builtin_amdgcn_udot4(top(v.x), 0x08040201u, 0u, false);
    lo = __builtin_amdgcn_udot4(top(v.y), 0x80402010u, lo, false);
    unsigned hi = __builtin_amdgcn_udot4(top(v.z), 0x08040201u, 0u, false);
    hi = __builtin_amdgcn_udot4(top(v.w), 0x80402010u, hi, false);
    return ((hi << 8) | lo) >> 7;
}

// 16 bits -> 16 bytes of 0 / 1
__device__ __forceinline__ u32x4 unpack16(const unsigned w)
{
    auto spread = [](unsigned n) { return __umul24(n, 0x204081u) & 0x01010101u; };
    u32x4 r;
    r.x = spread(w & 15u);
    r.y = spread((w >> 4) & 15u);
    r.z = spread((w >> 8) & 15u);
    r.w = spread((w >> 12) & 15u);
    return r;
}

// NL = 16-byte granules a thread stages per plane (at most); a thread owns at most NW = NL / 2 words per stage
template <bool HAS_MASK, int NL>
__global__ void __launch_bounds__(kBmNT)
bitmorph3_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, const unsigned char *__restrict__ msk,
                 const BitMorphParams p, int32_t *flags)
{
    constexpr int NW = NL > 1 ? NL / 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];

    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int total = p.nxt * p.nyt * p.nzc;
    if ((total & 7) == 0) b = (b & 7) * (total >> 3) + (b >> 3);          // neighbouring tiles on one XCD (shared L2)
    const int per_chunk = p.nxt * p.nyt;
    const int zci = b / per_chunk;
    const int rem = b - zci * per_chunk;
    const int yt = rem / p.nxt, xt = rem - yt * p.nxt;

    const int nx = p.nx, ny = p.ny, nz = p.nz, k = p.k, wz = p.wz, gy = p.gy, gxw = p.gxw, pitch = p.pitch, ns = p.ns;
    const int y0 = yt * p.ty, xw0 = xt * p.txw;
    const int zs = zci * p.zc, ze = min(zs + p.zc, nz), nout = ze - zs;
    const int row_first = y0 - k * p.oy;                   // array row of staged row 0
    const int xw_first = xw0 - p.hlw;                      // array word of staged word 0
    const int ngx = 2 * gxw;                               // granules per staged row
    const int slot_words = gy * pitch;
    const int stage_words = ns * slot_words;
    unsigned *mring = lds + (k + 1) * stage_words;         // HAS_MASK: nms slots
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx;
    const size_t plane_elems = (size_t)ny * (size_t)nx;
    const unsigned inv16 = p.invert ? 0xffffu : 0u;
    const unsigned border32 = p.border ? 0xffffffffu : 0u;

    // ---- staging recipe: granule g = tid + 256 i of the gy x ngx staged granules
    unsigned voff[NL];
    int ldsb[NL];                                          // byte offset inside a slot, -1 = nothing to stage
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int g = tid + kBmNT * i;
        const int row = g / ngx, col = g - row * ngx;
        const int y = row_first + row, xg = 2 * xw_first + col;
        const bool staged = g < gy * ngx;
        const bool inside = staged && y >= 0 && y < ny && xg >= 0 && 16 * xg < nx;
        voff[i] = inside ? (unsigned)(y * nx + 16 * xg) : kOOB;
        ldsb[i] = staged ? (row * pitch + 1) * 4 + 2 * col : -1;
    }
    // ---- stage recipe: word q = tid + 256 i of the gy x gxw staged words
    int woff[NW], wrow[NW];
    unsigned wvalid[NW];                                   // bits of the word that lie inside the array (0: row / word outside)
    bool wown[NW];                                         // the word belongs to this tile's OUTPUT region (changed flags)
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int q = tid + kBmNT * i;
        const int row = q / gxw, wc = q - row * gxw;
        const int y = row_first + row;
        const int xbit = 32 * (xw_first + wc);
        const bool staged = q < gy * gxw;
        woff[i] = staged ? row * pitch + 1 + wc : -1;
        wrow[i] = row;
        unsigned vm = 0;
        if (staged && y >= 0 && y < ny && xbit >= 0 && xbit < nx)
            vm = nx - xbit >= 32 ? 0xffffffffu : ((1u << (nx - xbit)) - 1u);
        wvalid[i] = vm;
        wown[i] = staged && row >= k * p.oy && row < k * p.oy + p.ty && wc >= p.hlw && wc < p.hlw + p.txw;
    }
    // ---- output recipe: granule g = tid + 256 i of the ty x (2 txw) output granules
    const int ogx = 2 * p.txw;
    unsigned ovoff[NL];
    int olds[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int g = tid + kBmNT * i;
        const int row = g / ogx, col = g - row * ogx;
        const int y = y0 + row, xg = 2 * xw0 + col;
        const bool live = g < p.ty * ogx && y < ny && 16 * xg < nx;
        ovoff[i] = live ? (unsigned)(y * nx + 16 * xg) : kOOB;
        olds[i] = ((k * p.oy + row) * pitch + 1 + p.hlw) * 4 + 2 * col;
        if (!(g < p.ty * ogx)) olds[i] = 0;
    }

    // the pad words either side of every staged row hold the border bit for good (a tile edge that is not an array
    // edge may read anything there: its outermost k * reach columns are recomputed by the neighbour)
    {
        const int nrows_all = ((k + 1) * ns + (HAS_MASK ? p.nms : 0)) * gy;
        for (int r = tid; r < nrows_all; r += kBmNT) {
            lds[r * pitch] = border32;
            lds[r * pitch + gxw + 1] = border32;
        }
    }

    u32x4 pin[NL], pmk[NL];
    bool pin_out = false;
    auto fetch = [&](int s) {
        // input plane A0(s) = zs - k oz + s; mask plane A0(s) - hz (what stage 1 works on in the NEXT step)
        // (the last k + 1 steps of a chunk only drain the stages: nothing any output depends on is left to fetch)
        int zsrc = zs - k * p.oz + s;
        pin_out = (unsigned)zsrc >= (unsigned)nz || s > nout - 1 + k * (wz - 1);
        zsrc = pin_out ? 0 : zsrc;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(in + (size_t)zsrc * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < NL; i++)
            if (!pin_out) pin[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, voff[i], 0, 0);
        if constexpr (HAS_MASK) {
            int zm = zs - k * p.oz + s - p.hz;
            const bool mout = (unsigned)zm >= (unsigned)nz || zm > ze - 1 + (k - 1) * p.hz;
            zm = mout ? 0 : zm;
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(msk + (size_t)zm * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
            for (int i = 0; i < NL; i++)
                pmk[i] = __builtin_amdgcn_raw_buffer_load_b128(rm, mout ? kOOB : voff[i], 0, 0);
        }
    };

    unsigned chg = 0;                                       // bit j - 1: iteration j changed a voxel of this tile
    const int nsteps = nout + k * wz + 1;
    fetch(0);
    __syncthreads();
    for (int s = 0; s < nsteps; s++) {
        const int wslot = s % ns;
        // ---- stage 0: the plane fetched during the previous step, as bits
        {
            unsigned char *slot = reinterpret_cast<unsigned char *>(lds + wslot * slot_words);
#pragma unroll
            for (int i = 0; i < NL; i++) {
                if (ldsb[i] < 0) continue;
                const bool o = pin_out || voff[i] == kOOB;
                const unsigned g16 = o ? border32 : (pack16(pin[i]) ^ inv16);
                *reinterpret_cast<unsigned short *>(slot + ldsb[i]) = (unsigned short)g16;
            }
            if constexpr (HAS_MASK) {
                unsigned char *mslot = reinterpret_cast<unsigned char *>(mring + (s % p.nms) * slot_words);
#pragma unroll
                for (int i = 0; i < NL; i++) {
                    if (ldsb[i] < 0) continue;
                    *reinterpret_cast<unsigned short *>(mslot + ldsb[i]) = (unsigned short)pack16(pmk[i]);
                }
            }
        }
        if (s + 1 < nsteps) fetch(s + 1);

        // ---- stages 1 .. k: plane A0(s) - j (1 + hz) from the ring of stage j - 1 (written in earlier steps)
        for (int j = 1; j <= k; j++) {
            const int zj = zs - k * p.oz + s - j * (1 + p.hz);
            const unsigned *src = lds + (j - 1) * stage_words;
            unsigned *dst = lds + j * stage_words + wslot * slot_words;
            const bool plane_in = (unsigned)zj < (unsigned)nz;          // uniform
            unsigned acc[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) acc[i] = 0xffffffffu;
            if (plane_in) {
                for (int r = 0; r < p.nrows; r++) {
                    const int tz = p.rowpos[r] & 255, ty = p.rowpos[r] >> 8;
                    const unsigned m = p.rowmask[r];
                    const unsigned *sl = src + ((s + 1 + tz) % ns) * slot_words;
                    const bool sides = (m & ~(1u << p.ox)) != 0;
#pragma unroll
                    for (int i = 0; i < NW; i++) {
                        if (woff[i] < 0) continue;
                        const int rr = min(max(wrow[i] + ty - p.oy, 0), gy - 1);
                        const unsigned *a = sl + woff[i] + (rr - wrow[i]) * pitch;
                        const unsigned C = a[0];
                        unsigned L = 0, R = 0;
                        if (sides) { L = a[-1]; R = a[1]; }
                        unsigned mm = m;
                        while (mm) {
                            const int tx = __builtin_ctz(mm);
                            mm &= mm - 1;
                            const int dx = tx - p.ox;
                            unsigned v;
                            if (dx == 0) v = C;
                            else if (dx > 0) v = __builtin_amdgcn_alignbit(R, C, (unsigned)dx);
                            else v = __builtin_amdgcn_alignbit(C, L, (unsigned)(32 + dx));
                            acc[i] &= v;
                        }
                    }
                }
            }
            // centre word of the previous stage (mask blend, changed flag): plane zj is tz = oz of the window
            const unsigned *cs = src + ((s + 1 + p.oz) % ns) * slot_words;
            const bool count = plane_in && zj >= zs && zj < ze;
#pragma unroll
            for (int i = 0; i < NW; i++) {
                if (woff[i] < 0) continue;
                unsigned res = acc[i];
                if (plane_in) {
                    const unsigned cur = cs[woff[i]];
                    if constexpr (HAS_MASK) {
                        // mask plane zj was staged at step s' with A0(s') - hz = zj
                        const unsigned mk = mring[((s - j * (1 + p.hz) + p.hz + 2 * p.nms * 64) % p.nms) * slot_words + woff[i]];
                        res = (res & mk) | (cur & ~mk);
                    }
                    if (count && wown[i]) chg |= ((res ^ cur) & wvalid[i]) ? (1u << (j - 1)) : 0u;
                    res = (res & wvalid[i]) | (border32 & ~wvalid[i]);
                } else {
                    res = border32;
                }
                dst[woff[i]] = res;
            }
        }

        // ---- output: plane A0(s) - 1 - k (1 + hz), finished by stage k in the previous step
        {
            const int zo = zs + s - 1 - k * wz;
            if (zo >= zs && zo < ze) {
                const unsigned char *slot = reinterpret_cast<const unsigned char *>(lds + k * stage_words + ((s - 1) % ns) * slot_words);
                const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(out + (size_t)zo * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
                for (int i = 0; i < NL; i++) {
                    if (ovoff[i] == kOOB) continue;
                    const unsigned w = *reinterpret_cast<const unsigned short *>(slot + olds[i]);
                    __builtin_amdgcn_raw_buffer_store_b128(unpack16(w ^ inv16), rout, ovoff[i], 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (flags) {
        for (int j = 0; j < k; j++)
            if (__any((chg >> j) & 1u) && (tid & 63) == 0) atomicOr(flags + j, 1);
    }
}

// test / tuning hook: on = 0 never, 1 the production rule, 2 also on small volumes; (ty, nzc) of the next launches, 0 = the planner's
static Knob g_bm_ty{0}, g_bm_nzc{0}, g_bm_on{1};

template <bool HAS_MASK, int NL>
static int launch_bitmorph(const unsigned char *in, unsigned char *out, const unsigned char *msk, const BitMorphParams &p,
                           size_t lds, int32_t *flags, hipStream_t s)
{
    static PerDeviceOnce attr;
    if (!attr) {
        MI_HIP(hipFuncSetAttribute((const void *)bitmorph3_kernel<HAS_MASK, NL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kBmMaxLds));
        attr = true;
    }
    const int64_t total = (int64_t)p.nxt * p.nyt * p.nzc;
    hipLaunchKernelGGL((bitmorph3_kernel<HAS_MASK, NL>), dim3((unsigned)total), dim3(kBmNT), lds, s, in, out, msk, p, flags);
    MI_HIP(hipGetLastError());
    note_kernel("mi::bitmorph3_kernel<%s,%d> grid=%lld k=%d tile=%dx%d rows, %d planes (1 bit per voxel, %d fused iteration%s per launch)",
                HAS_MASK ? "mask" : "nomask", NL, (long long)total, p.k, p.ty, p.txw * 32, p.zc, p.k, p.k == 1 ? "" : "s");
    return MI_OK;
}

// k fused iterations on a 3-D volume of 1-byte voxels; MI_ERR_UNSUPPORTED (nothing launched) outside the envelope.
int bitmorph3(const mi_array *in, const mi_array *out, const uint8_t *structure, const int64_t *sshape, const int *origins,
              const mi_array *mask, int border_value, int invert, int k, int32_t *flags, hipStream_t s)
{
#define NOPE(msg) do { set_error("bitmorph3: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (!g_bm_on) NOPE("switched off (mi_debug_set_bitmorph)");
    if (dtype_size(in->dtype) != 1 || dtype_size(out->dtype) != 1) NOPE("1-byte volumes only");
    if (in->ndim != 3) NOPE("3-D only");
    if (k < 1 || k > kBmMaxK) NOPE("1 .. 8 fused iterations");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    int w[3], off[3];
    for (int d = 0; d < 3; d++) {
        if (sshape[d] < 1 || sshape[d] > 31) NOPE("structure extent > 31");
        w[d] = (int)sshape[d];
        off[d] = (int)(sshape[d] / 2 + origins[d]);
        if (off[d] < 0 || off[d] >= sshape[d]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    if (nx < 64 || (nx & 15)) NOPE("rows must be a multiple of 16 bytes, >= 64");
    if (ny * nx >= ((int64_t)1 << 31) || nz > (1 << 24) || ny > (1 << 24)) NOPE("plane too large");
    if (g_bm_on != 2 && nz * ny * nx < (1 << 18)) NOPE("small volume: the byte kernel's launch is as fast");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15) || (mask && ((uintptr_t)mask->data & 15)))
        NOPE("needs 16-byte aligned data");

    BitMorphParams p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.wz = w[0]; p.oz = off[0]; p.hz = w[0] - 1 - off[0];
    p.oy = off[1]; p.hy = w[1] - 1 - off[1];
    p.ox = off[2];
    p.k = k;
    p.invert = invert != 0;
    p.border = invert ? !border_value : (border_value != 0);
    // rows of the structure that hold a tap; the reach along each axis is that of the SET taps' bounding box only as far
    // as the halo goes, the extents themselves stay (an all-false outer row costs a staged row, nothing else)
    int nrows = 0;
    for (int tz = 0; tz < w[0]; tz++)
        for (int ty = 0; ty < w[1]; ty++) {
            unsigned m = 0;
            for (int tx = 0; tx < w[2]; tx++)
                if (structure[((int64_t)tz * w[1] + ty) * w[2] + tx]) m |= 1u << tx;
            if (!m) continue;
            if (nrows == kBmMaxRows) NOPE("structure has too many rows");
            p.rowpos[nrows] = (unsigned short)(tz | ty << 8);
            p.rowmask[nrows++] = m;
        }
    p.nrows = nrows;                                        // 0: an empty structure erodes nothing (output = true)
    p.ns = p.wz + 1;
    p.nms = mask ? k * (1 + p.hz) - p.hz + 1 : 0;

    // ---- tile geometry
    const int words = (int)((nx + 31) / 32);
    const int hx = w[2] - 1 - off[2];
    if (words <= 32) { p.nxt = 1; p.txw = words; p.hlw = 0; p.gxw = words; }
    else {
        p.txw = 32;
        p.nxt = (words + 31) / 32;
        p.hlw = (k * off[2] + 31) / 32;
        p.gxw = p.txw + p.hlw + (k * hx + 31) / 32;
    }
    p.pitch = p.gxw + 2;
    if (!(p.pitch & 1)) p.pitch++;                          // odd pitch: the rows of a column of words fall into different banks
    const int halo_y = k * (p.oy + p.hy);
    const int ngx = 2 * p.gxw;
    const int max_gy = std::min(8 * kBmNT / ngx, (int)(kBmMaxLds / 4 / ((size_t)((k + 1) * p.ns + p.nms) * p.pitch)));
    if (max_gy - halo_y < 2) NOPE("structure / iteration count too large for one tile");
    const int cus = device_cus();
    const int64_t slots = 2 * (int64_t)cus;                 // two resident workgroups per CU saturate the memory system
    double best = 1e300;
    int best_ty = 0, best_nzc = 1;
    const int ty_hi = (int)std::min<int64_t>(max_gy - halo_y, ny);
    for (int ty = std::min(ty_hi, 2); ty <= ty_hi; ty++) {
        const int gy = ty + halo_y;
        const int64_t nyt = (ny + ty - 1) / ty;
        const double step = (double)gy * ngx * (mask ? 2 : 1) + (double)ty * 2 * p.txw + 192.0 + 96.0 * k;
        for (int nzc = 1; nzc <= std::min<int64_t>(nz, 256); nzc++) {
            const int chunk = (int)((nz + nzc - 1) / nzc);
            const int real = (int)((nz + chunk - 1) / chunk);
            const int64_t wgs = nyt * p.nxt * real;
            const double rounds = (double)((wgs + slots - 1) / slots);
            const double cost = rounds * (chunk + k * p.wz + 1) * step;
            if (cost < best) { best = cost; best_ty = ty; best_nzc = real; }
        }
    }
    if (g_bm_ty > 0 && g_bm_ty <= ty_hi) best_ty = g_bm_ty;
    if (g_bm_nzc > 0) best_nzc = (int)std::min<int64_t>(g_bm_nzc, nz);
    p.ty = best_ty;
    p.gy = best_ty + halo_y;
    p.nyt = (int)((ny + p.ty - 1) / p.ty);
    p.zc = (int)((nz + best_nzc - 1) / best_nzc);
    p.nzc = (int)((nz + p.zc - 1) / p.zc);
    if ((int64_t)p.nxt * p.nyt * p.nzc > 0x7fffffff) NOPE("too many tiles");
    const size_t lds = (size_t)((k + 1) * p.ns + p.nms) * p.gy * p.pitch * 4;
    if (lds > (size_t)kBmMaxLds) NOPE("does not fit LDS");
    const int nl = (p.gy * ngx + kBmNT - 1) / kBmNT;

    const unsigned char *ip = (const unsigned char *)in->data;
    unsigned char *op = (unsigned char *)out->data;
    const unsigned char *mp = mask ? (const unsigned char *)mask->data : nullptr;
#define GO(NLV) return mp ? launch_bitmorph<true, NLV>(ip, op, mp, p, lds, flags, s) : launch_bitmorph<false, NLV>(ip, op, mp, p, lds, flags, s)
    if (nl <= 2) { GO(2); }
    if (nl <= 4) { GO(4); }
    if (nl <= 6) { GO(6); }
    GO(8);
#undef GO
#undef NOPE
}

}  // namespace mi

extern "C" int mi_debug_set_bitmorph(int on, int ty, int nzc)
{
    mi::g_bm_on = on;
    mi::g_bm_ty = ty;
    mi::g_bm_nzc = nzc;
    return MI_OK;
}
