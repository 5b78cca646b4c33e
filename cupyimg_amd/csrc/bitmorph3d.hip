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
#include <algorithm>
#include <cstddef>
#include <vector>

namespace mi {

constexpr int kBmNT = 256;            // threads per workgroup
constexpr int kBmMaxRows = 96;        // structure rows (dz, dy) with at least one tap
constexpr int kBmMaxK = 8;            // fused iterations per launch
constexpr int kBmMaxLds = 64 * 1024;  // per workgroup (two or more workgroups per CU)

struct BitMorphParams {
    int nx, ny, nz;
    int wz, oz, hz;         // structure extent along z, lo reach (w/2 + origin), hi reach
    int oy, hy, ox;
    int nrows, k;           // nrows: table entries (even; rows of one dx-mask are consecutive, padded by repetition)
    int border, invert;
    int ty, gy;             // output rows per tile, staged rows = ty + k (oy + hy)
    int txw, gxw, hlw;      // output words (32 voxels) per tile row, staged words per row, halo words on the left
    int pitch;              // LDS words per staged row: 1 pad + gxw + 1 pad (+ skew)
    int zc, nzc, nxt, nyt;
    int ns, nms;            // ring slots per stage (wz + 1), slots of the mask ring
    // opening / closing in one launch: stages 1 .. kflip are the first operation, stage kflip writes the COMPLEMENT of its
    // result (the erosion's result seen as the dilation-of-the-complement's input, or the other way round), stages
    // kflip + 1 .. k the second one with the mirrored structure (table entries nrows .. nrows + nrows2 - 1) and the
    // complemented border; the output is complemented once more than the input.  0 = k iterations of one operation.
    int kflip, nrows2;
    int pad_[1];
    // per structure row (dz, dy) with a tap: { tz | last-of-its-group << 8, dx mask (bit tx), (ty - oy) * pitch * 4, 0 }
    alignas(16) int rows[kBmMaxRows][4];
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

// Structures known at compile time (origin 0): the stage is straight-line code -- every LDS read of a word is issued
// before the first is consumed, no scalar loop, no table.  KIND 0 = the run-time table (any structure, any origin).
struct BmCross {      // generate_binary_structure(3, 1): the default structure
    static constexpr int n = 5;
    static constexpr int tz[5] = {0, 1, 1, 1, 2}, ty[5] = {1, 0, 1, 2, 1};
    static constexpr unsigned m[5] = {2, 2, 7, 2, 2};
};
struct BmConn18 {     // generate_binary_structure(3, 2)
    static constexpr int n = 9;
    static constexpr int tz[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2}, ty[9] = {0, 1, 2, 0, 1, 2, 0, 1, 2};
    static constexpr unsigned m[9] = {2, 7, 2, 7, 7, 7, 2, 7, 2};
};
struct BmCube3 {      // generate_binary_structure(3, 3) = ones((3, 3, 3))
    static constexpr int n = 9;
    static constexpr int tz[9] = {0, 0, 0, 1, 1, 1, 2, 2, 2}, ty[9] = {0, 1, 2, 0, 1, 2, 0, 1, 2};
    static constexpr unsigned m[9] = {7, 7, 7, 7, 7, 7, 7, 7, 7};
};

// one word of a fixed 3 x 3 x 3 structure: `c` = byte address of the word in the CENTRE row of slot 0 of the source
// ring; so[tz] = byte offset of the slot that holds plane z - 1 + tz; pitch4 = bytes per staged row
template <typename S>
__device__ __forceinline__ unsigned bm_fixed_word(const char *c, const int (&so)[3], int pitch4)
{
    unsigned C[S::n], L[S::n], R[S::n];
#pragma unroll
    for (int r = 0; r < S::n; r++) {
        const unsigned *a = reinterpret_cast<const unsigned *>(c + so[S::tz[r]] + (S::ty[r] - 1) * pitch4);
        C[r] = a[0];
        if (S::m[r] & 1u) L[r] = a[-1];
        if (S::m[r] & 4u) R[r] = a[1];
    }
    unsigned aC = 0xffffffffu, aL = 0xffffffffu, aR = 0xffffffffu, plain = 0xffffffffu;
#pragma unroll
    for (int r = 0; r < S::n; r++) {
        if (S::m[r] == 2u) plain &= C[r];                   // centre tap only
        else { aC &= C[r]; aL &= L[r]; aR &= R[r]; }        // all three taps (the fixed structures have no other rows)
    }
    return plain & aC & __builtin_amdgcn_alignbit(aR, aC, 1u) & __builtin_amdgcn_alignbit(aC, aL, 31u);
}

// LDS (words): [oy rows of slack][ring 0 .. ring k: ns slots x gy rows x pitch][mask ring: nms slots][hy rows of slack]
// [NT dump words].  A tap row above / below the staged rows reads the neighbouring slot or the slack: anything may be
// there -- such outputs lie in the halo this tile recomputes for nobody.  Threads without a granule / word of their own
// write to their dump word, so that no stage needs a divergent branch.
//
// NL = 16-byte granules a thread stages per plane (at most); a thread owns at most NW = ceil(NL / 2) words per stage
// RG (r6): rows of ANY length >= 64 bytes (181 x 217 x 181 masks ...).  16-byte buffer loads and stores take any byte
// alignment on this chip, so a row is staged from wherever it starts; what differs is the LAST granule of a row, which holds
// nx % 16 voxels of its row followed by the head of the next one: stage 0 replaces those bits by the border bit (taps beyond
// the row end see border_value, like every other out-of-array position), the output stores that granule in 8 / 4 / 2 / 1-byte
// pieces, and the load descriptors reach 16 bytes past a plane (the host has checked that the allocation does: a 16-byte
// access that straddles num_records loses its straddling DWORD, in-range bytes included).
template <bool HAS_MASK, int NL, int NT, int KIND, bool RG = false>
__global__ void __launch_bounds__(NT)
bitmorph3_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, const unsigned char *__restrict__ msk,
                 const BitMorphParams p, int32_t *flags)
{
    constexpr int NW = (NL + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned lds_raw[];

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
    unsigned *lds = lds_raw + p.oy * pitch;
    unsigned *mring = lds + (k + 1) * stage_words;         // HAS_MASK: nms slots
    const int dump = ((k + 1) * ns + (HAS_MASK ? p.nms : 0)) * slot_words + p.hy * pitch + tid;   // word index
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx;
    const size_t plane_elems = (size_t)ny * (size_t)nx;
    const unsigned inv16 = p.invert ? 0xffffu : 0u;
    const unsigned inv16out = (p.invert != 0) != (p.kflip != 0) ? 0xffffu : 0u;
    const unsigned border32 = p.border ? 0xffffffffu : 0u;

    // ---- staging recipe: granule g = tid + NT i of the gy x ngx staged granules
    unsigned voff[NL];
    int ldsb[NL];                                          // byte offset inside a slot (the dump word: nothing to stage)
    unsigned gvalid[RG ? NL : 1];                          // RG: the bits of the granule that belong to its row
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int g = tid + NT * i;
        const int row = g / ngx, col = g - row * ngx;
        const int y = row_first + row, xg = 2 * xw_first + col;
        const bool staged = g < gy * ngx;
        const bool inside = staged && y >= 0 && y < ny && xg >= 0 && 16 * xg < nx;
        voff[i] = inside ? (unsigned)(y * nx + 16 * xg) : kOOB;
        ldsb[i] = staged ? (row * pitch + 1) * 4 + 2 * col : -1;
        if constexpr (RG) gvalid[i] = !inside ? 0u : (nx - 16 * xg >= 16 ? 0xffffu : ((1u << (nx - 16 * xg)) - 1u));
    }
    // ---- stage recipe: word q = tid + NT i of the gy x gxw staged words
    int woff[NW];                                          // word offset inside a slot, -1 = none
    int wrd[NW];                                           // the same for reads (a thread without a word reads word 0 of the slot)
    unsigned wvalid[NW];                                   // bits of the word that lie inside the array (0: row / word outside)
    unsigned wown[NW];                                     // all ones: the word belongs to this tile's OUTPUT region (changed flags)
#pragma unroll
    for (int i = 0; i < NW; i++) {
        const int q = tid + NT * i;
        const int row = q / gxw, wc = q - row * gxw;
        const int y = row_first + row;
        const int xbit = 32 * (xw_first + wc);
        const bool staged = q < gy * gxw;
        woff[i] = staged ? row * pitch + 1 + wc : -1;
        wrd[i] = staged ? woff[i] : 1;
        unsigned vm = 0;
        if (staged && y >= 0 && y < ny && xbit >= 0 && xbit < nx)
            vm = nx - xbit >= 32 ? 0xffffffffu : ((1u << (nx - xbit)) - 1u);
        wvalid[i] = vm;
        wown[i] = (staged && row >= k * p.oy && row < k * p.oy + p.ty && wc >= p.hlw && wc < p.hlw + p.txw) ? 0xffffffffu : 0u;
    }
    // ---- output recipe: granule g = tid + NT i of the ty x (2 txw) output granules
    const int ogx = 2 * p.txw;
    unsigned ovoff[NL];
    int olds[NL];
    int otail[RG ? NL : 1];                                // RG: voxels of the granule that belong to its row (16: a whole one)
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int g = tid + NT * i;
        const int row = g / ogx, col = g - row * ogx;
        const int y = y0 + row, xg = 2 * xw0 + col;
        const bool live = g < p.ty * ogx && y < ny && 16 * xg < nx;
        ovoff[i] = live ? (unsigned)(y * nx + 16 * xg) : kOOB;                 // a store at kOOB is dropped by the descriptor
        olds[i] = g < p.ty * ogx ? ((k * p.oy + row) * pitch + 1 + p.hlw) * 4 + 2 * col : 0;
        if constexpr (RG) otail[i] = live ? min(nx - 16 * xg, 16) : 16;
    }

    // the pad words either side of every staged row hold the border bit for good (a tile edge that is not an array
    // edge may read anything there: its outermost k * reach columns are recomputed by the neighbour)
    {
        const int nrows_all = ((k + 1) * ns + (HAS_MASK ? p.nms : 0)) * gy;
        const int flip_row = p.kflip ? p.kflip * ns * gy : nrows_all;        // rings kflip .. k hold the second operation's bits
        for (int r = tid; r < nrows_all; r += NT) {
            const unsigned bpad = r >= flip_row ? ~border32 : border32;
            lds[r * pitch] = bpad;
            lds[r * pitch + gxw + 1] = bpad;
        }
    }

    const int last_fetch = nout - 1 + k * (wz - 1);         // the last k + 1 steps of a chunk only drain the stages
    // Every path through a step issues the SAME number of vector-memory operations (planes that are not there are
    // fetched at kOOB offsets, which the descriptor answers with zeros without touching memory; stores at kOOB are
    // dropped): the compiler's s_waitcnt insertion can then count how many younger operations may stay in flight when a
    // plane is consumed -- with conditional fetches it has to assume none and every step would wait for the write
    // acknowledgements of the step before it.
    auto fetch = [&](int s, u32x4 (&pin)[NL], u32x4 (&pmk)[NL], bool &pout) {
        // input plane A0(s) = zs - k oz + s; mask plane A0(s) - hz (what stage 1 works on in the NEXT step)
        int zsrc = zs - k * p.oz + s;
        pout = (unsigned)zsrc >= (unsigned)nz || s > last_fetch;
        zsrc = pout ? 0 : zsrc;
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(in + (size_t)zsrc * plane_elems), 0, (int)plane_bytes + (RG ? 16 : 0), 0x00020000);
#pragma unroll
        for (int i = 0; i < NL; i++) pin[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, pout ? kOOB : voff[i], 0, 0);
        if constexpr (HAS_MASK) {
            int zm = zs - k * p.oz + s - p.hz;
            const bool mout = (unsigned)zm >= (unsigned)nz || zm > ze - 1 + (k - 1) * p.hz;
            zm = mout ? 0 : zm;
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(msk + (size_t)zm * plane_elems), 0, (int)plane_bytes + (RG ? 16 : 0), 0x00020000);
#pragma unroll
            for (int i = 0; i < NL; i++) pmk[i] = __builtin_amdgcn_raw_buffer_load_b128(rm, mout ? kOOB : voff[i], 0, 0);
        }
    };

    unsigned chg = 0;                                       // bit j - 1: iteration j changed a voxel of this tile
    const int nsteps = (nout + k * wz + 1 + 1) & ~1;       // even: the loop body is two steps; a step too many stores nothing
    int wslot = 0, pslot = ns - 1, mslot = 0;               // s % ns, (s - 1) % ns, s % nms

    auto step = [&](int s, u32x4 (&pin)[NL], u32x4 (&pmk)[NL], bool &pout) {
        // ---- stage 0: the plane fetched two steps ago, as bits
        {
            unsigned char *slot = reinterpret_cast<unsigned char *>(lds + wslot * slot_words);
            unsigned char *dumpb = reinterpret_cast<unsigned char *>(lds + dump);
#pragma unroll
            for (int i = 0; i < NL; i++) {
                // no branch: a granule outside the array (or a plane that is not there) packs the zeros its kOOB load
                // returned and is replaced by the border bits with one select
                unsigned om;
                if constexpr (RG) om = pout ? 0xffffffffu : ~gvalid[i];
                else om = (pout || voff[i] == kOOB) ? 0xffffffffu : 0u;
                const unsigned g16 = ((pack16(pin[i]) ^ inv16) & ~om) | (border32 & om);
                *reinterpret_cast<unsigned short *>(ldsb[i] < 0 ? dumpb : slot + ldsb[i]) = (unsigned short)g16;
            }
            if constexpr (HAS_MASK) {
                unsigned char *mslotp = reinterpret_cast<unsigned char *>(mring + mslot * slot_words);
#pragma unroll
                for (int i = 0; i < NL; i++)
                    *reinterpret_cast<unsigned short *>(ldsb[i] < 0 ? dumpb : mslotp + ldsb[i]) = (unsigned short)pack16(pmk[i]);
            }
        }
        fetch(s + 2, pin, pmk, pout);

        // ---- stages 1 .. k: plane A0(s) - j (1 + hz) from the ring of stage j - 1 (written in earlier steps)
        for (int j = 1; j <= k; j++) {
            const int zj = zs - k * p.oz + s - j * (1 + p.hz);
            const unsigned *src = lds + (j - 1) * stage_words;
            unsigned *dst = lds + j * stage_words + wslot * slot_words;
            const bool plane_in = (unsigned)zj < (unsigned)nz;          // uniform
            const bool second = p.kflip != 0 && j > p.kflip;            // the second operation of an opening / closing
            const unsigned bst = second ? ~border32 : border32;         // what a tap outside the array sees in this stage
            const unsigned wflip = j == p.kflip ? 0xffffffffu : 0u;     // the first operation's last stage hands over the complement
            const int rfirst = second ? p.nrows : 0, rlast = second ? p.nrows + p.nrows2 : p.nrows;
            unsigned res[NW];
#pragma unroll
            for (int i = 0; i < NW; i++) res[i] = 0xffffffffu;
            if (plane_in && KIND != 0) {
                int so[3];
#pragma unroll
                for (int t = 0; t < 3; t++) {
                    int sl = wslot + 1 + t;
                    sl -= sl >= ns ? ns : 0;
                    so[t] = sl * slot_words * 4;
                }
#pragma unroll
                for (int i = 0; i < NW; i++) {
                    const char *c = reinterpret_cast<const char *>(src) + wrd[i] * 4;
                    if constexpr (KIND == 1) res[i] = bm_fixed_word<BmCross>(c, so, pitch * 4);
                    else if constexpr (KIND == 2) res[i] = bm_fixed_word<BmConn18>(c, so, pitch * 4);
                    else res[i] = bm_fixed_word<BmCube3>(c, so, pitch * 4);
                }
            }
            if (plane_in && KIND == 0) {
                unsigned aL[NW], aC[NW], aR[NW];
#pragma unroll
                for (int i = 0; i < NW; i++) aL[i] = aC[i] = aR[i] = 0xffffffffu;
                for (int r = rfirst; r < rlast; r += 2) {
                    const int e0 = p.rows[r][0], m = p.rows[r][1], ro0 = p.rows[r][2];
                    const int e1 = p.rows[r + 1][0], ro1 = p.rows[r + 1][2];
                    int s0 = wslot + 1 + (e0 & 255), s1 = wslot + 1 + (e1 & 255);
                    s0 -= s0 >= ns ? ns : 0;
                    s1 -= s1 >= ns ? ns : 0;
                    const char *b0 = reinterpret_cast<const char *>(src + s0 * slot_words) + ro0;
                    const char *b1 = reinterpret_cast<const char *>(src + s1 * slot_words) + ro1;
                    const bool sides = ((unsigned)m & ~(1u << p.ox)) != 0;
                    unsigned c0[NW], c1[NW], l0[NW], l1[NW], r0[NW], r1[NW];
#pragma unroll
                    for (int i = 0; i < NW; i++) {
                        const int wo = wrd[i] * 4;
                        const unsigned *a0 = reinterpret_cast<const unsigned *>(b0 + wo);
                        const unsigned *a1 = reinterpret_cast<const unsigned *>(b1 + wo);
                        c0[i] = a0[0];
                        c1[i] = a1[0];
                        if (sides) { l0[i] = a0[-1]; r0[i] = a0[1]; l1[i] = a1[-1]; r1[i] = a1[1]; }
                    }
#pragma unroll
                    for (int i = 0; i < NW; i++) {
                        aC[i] &= c0[i] & c1[i];
                        if (sides) { aL[i] &= l0[i] & l1[i]; aR[i] &= r0[i] & r1[i]; }
                    }
                    if (e1 & 256) {                         // last pair of the rows that share this dx mask
                        unsigned mm = (unsigned)m;
                        while (mm) {
                            const int tx = __builtin_ctz(mm);
                            mm &= mm - 1;
                            const int dx = tx - p.ox;
#pragma unroll
                            for (int i = 0; i < NW; i++) {
                                unsigned v;
                                if (dx == 0) v = aC[i];
                                else if (dx > 0) v = __builtin_amdgcn_alignbit(aR[i], aC[i], (unsigned)dx);
                                else v = __builtin_amdgcn_alignbit(aC[i], aL[i], (unsigned)(32 + dx));
                                res[i] &= v;
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NW; i++) aL[i] = aC[i] = aR[i] = 0xffffffffu;
                    }
                }
            }
            // centre word of the previous stage (mask blend, changed flag): plane zj is tz = oz of the window
            int cslot = wslot + 1 + p.oz;
            cslot -= cslot >= ns ? ns : 0;
            const unsigned *cs = src + cslot * slot_words;
            const unsigned count = (plane_in && zj >= zs && zj < ze) ? 0xffffffffu : 0u;
            int mj = 0;
            if constexpr (HAS_MASK) {
                mj = mslot + p.hz - j * (1 + p.hz);         // mask plane zj was staged at step s' with A0(s') - hz = zj
                mj += mj < 0 ? p.nms : 0;
            }
#pragma unroll
            for (int i = 0; i < NW; i++) {
                const int wo = wrd[i];
                unsigned r = bst;
                if (plane_in) {
                    const unsigned cur = cs[wo];
                    r = res[i];
                    if constexpr (HAS_MASK) {
                        const unsigned mk = mring[mj * slot_words + wo];
                        r = (r & mk) | (cur & ~mk);
                    }
                    chg |= ((r ^ cur) & wvalid[i] & wown[i] & count) ? (1u << (j - 1)) : 0u;
                    r = (r & wvalid[i]) | (bst & ~wvalid[i]);
                }
                dst[woff[i] < 0 ? dump - wslot * slot_words - j * stage_words : woff[i]] = r ^ wflip;
            }
        }

        // ---- output: plane A0(s) - 1 - k (1 + hz), finished by stage k in the previous step
        {
            const int zo = zs + s - 1 - k * wz;
            const bool live = zo >= zs && zo < ze;
            const unsigned char *slot = reinterpret_cast<const unsigned char *>(lds + k * stage_words + pslot * slot_words);
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(out + (size_t)(live ? zo : 0) * plane_elems), 0, (int)plane_bytes, 0x00020000);
            unsigned w[NL];
#pragma unroll
            for (int i = 0; i < NL; i++) w[i] = *reinterpret_cast<const unsigned short *>(slot + olds[i]);
#pragma unroll
            for (int i = 0; i < NL; i++) {
                const u32x4 v = unpack16(w[i] ^ inv16out);
                if constexpr (!RG) {
                    __builtin_amdgcn_raw_buffer_store_b128(v, rout, live ? ovoff[i] : kOOB, 0, 0);
                } else {
                    // a whole granule, or the 1 .. 15 voxels of a row's last one as 8 + 4 + 2 + 1 bytes (every store is issued,
                    // at kOOB when it has nothing to write: the operation count of a step stays the same on every path)
                    const int nv = otail[i];
                    const unsigned o = live ? ovoff[i] : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b128(v, rout, nv == 16 ? o : kOOB, 0, 0);
                    const bool part = nv < 16 && o != kOOB;
                    const unsigned o4 = (nv & 8) ? 8u : 0u, o2 = o4 + ((nv & 4) ? 4u : 0u), o1 = o2 + ((nv & 2) ? 2u : 0u);
                    const unsigned d4 = (nv & 8) ? v.z : v.x;
                    const unsigned d2s = o2 >= 8 ? (o2 >= 12 ? v.w : v.z) : (o2 >= 4 ? v.y : v.x);
                    const unsigned d1w = o1 >= 8 ? (o1 >= 12 ? v.w : v.z) : (o1 >= 4 ? v.y : v.x);
                    const unsigned d1 = d1w >> (8 * (o1 & 3u));
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){v.x, v.y}, rout, (part && (nv & 8)) ? o : kOOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(d4, rout, (part && (nv & 4)) ? o + o4 : kOOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)d2s, rout, (part && (nv & 2)) ? o + o2 : kOOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)d1, rout, (part && (nv & 1)) ? o + o1 : kOOB, 0, 0);
                }
            }
        }
        pslot = wslot;
        wslot = wslot + 1 == ns ? 0 : wslot + 1;
        if constexpr (HAS_MASK) mslot = mslot + 1 == p.nms ? 0 : mslot + 1;
        __syncthreads();
    };

    // two planes in flight per thread: buffer A holds the planes of the even steps, B those of the odd ones
    u32x4 pinA[NL], pinB[NL], pmkA[HAS_MASK ? NL : 1], pmkB[HAS_MASK ? NL : 1];
    bool outA = true, outB = true;
    if constexpr (HAS_MASK) {
        fetch(0, pinA, pmkA, outA);
        fetch(1, pinB, pmkB, outB);
    } else {
        fetch(0, pinA, pinA, outA);
        fetch(1, pinB, pinB, outB);
    }
    __syncthreads();
    for (int s = 0; s < nsteps; s += 2) {
        if constexpr (HAS_MASK) {
            step(s, pinA, pmkA, outA);
            step(s + 1, pinB, pmkB, outB);
        } else {
            step(s, pinA, pinA, outA);
            step(s + 1, pinB, pinB, outB);
        }
    }
    if (flags) {
        for (int j = 0; j < k; j++)
            if (__any((chg >> j) & 1u) && (tid & 63) == 0) atomicOr(flags + j, 1);
    }
}

// test / tuning hook: on = 0 never, 1 the production rule, 2 also on small volumes; (ty, nzc) of the next launches, 0 = the planner's
static Knob g_bm_ty{0}, g_bm_nzc{0}, g_bm_on{1}, g_bm_kind0{0}, g_bm_2d{1}, g_bm_ragged{1};

template <bool HAS_MASK, int NL, int NT, int KIND, bool RG>
static int launch_bitmorph(const unsigned char *in, unsigned char *out, const unsigned char *msk, const BitMorphParams &p,
                           size_t lds, int32_t *flags, hipStream_t s)
{
    static PerDeviceOnce attr;
    if (!attr) {
        MI_HIP(hipFuncSetAttribute((const void *)bitmorph3_kernel<HAS_MASK, NL, NT, KIND, RG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kBmMaxLds));
        attr = true;
    }
    const int64_t total = (int64_t)p.nxt * p.nyt * p.nzc;
    hipLaunchKernelGGL((bitmorph3_kernel<HAS_MASK, NL, NT, KIND, RG>), dim3((unsigned)total), dim3(NT), lds, s, in, out, msk, p, flags);
    MI_HIP(hipGetLastError());
    note_kernel("mi::bitmorph3_kernel<%s,%d,%d,%s%s> grid=%lld k=%d%s tile=%dx%d rows, %d planes (1 bit per voxel, %d fused iteration%s per launch)",
                HAS_MASK ? "mask" : "nomask", NL, NT, KIND == 0 ? "table" : KIND == 1 ? "cross" : KIND == 2 ? "conn18" : "cube3", RG ? ",ragged" : "", (long long)total, p.k, p.kflip ? (p.invert ? " (closing)" : " (opening)") : "", p.ty, p.txw * 32, p.zc, p.k, p.k == 1 ? "" : "s");
    return MI_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// bitfill3_kernel (r6): masked dilation UNTIL STABLE -- binary_propagation, binary_fill_holes (morphology.py:684-766; the
// reference: one launch, one full-volume comparison and one host synchronisation per ITERATION).  The operator
// s' = s | (mask & OR of the structure's taps of s) is monotone when the structure holds its centre, so ANY fair order of
// local updates reaches the same least fixed point.  A workgroup takes a block of BZ planes x BY rows x whole rows (+ halo of
// the structure's reach) as bits in LDS; every thread owns ONE ROW and per sweep (1) ORs the taps of the neighbouring rows
// into it, (2) FILLS along x: inside a run of mask bits everything above / below a set bit is set in one pass, by the carry
// of an addition (u = m + (s & m) + carry: the bits a carry ran through are the filled ones; down: the same on bit-reversed
// words) -- what x-adjacent taps would need one iteration per voxel for.  Sweeps repeat until the block is stable (a row
// step per sweep along y / z), its halo being what the neighbours held when the launch began; the host repeats launches
// (ping-pong) until no block changed.  A plain block relaxation without the fill was measured first and was no faster than
// the fused iterations (profiles/r6_relaxation_experiment.txt): a sweep then moves information one voxel, like an iteration.
// True space (bit = voxel set); outside the array a tap sees border_value.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFillNT = 256;
constexpr int kFillSweeps = 64;

struct FillParams {
    int nx, ny, nz;
    int oz, oy, ox;         // lo reach (w/2 + origin)
    int hz, hy;             // hi reach
    int nrows;
    int border;             // true-space bit outside the array
    int bz, by;             // block: planes x rows (bz * by <= 256: one row per thread)
    int gxw, pitch;         // words per row (whole rows), LDS words per row (1 pad + gxw + 1 pad)
    int nyt, nzt;
    int fill_up, fill_down; // the centre row holds the tap at dx = -1 / dx = +1
    alignas(16) int rows[kBmMaxRows][4];      // { tz, dx mask, ty, 0 }
};

template <bool RG>
__global__ void __launch_bounds__(kFillNT)
bitfill3_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out, const unsigned char *__restrict__ msk,
                const FillParams p, int32_t *flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    __shared__ int sweep_changed, any_changed;
    const int tid = threadIdx.x;
    const int zt = blockIdx.x / p.nyt, yt = blockIdx.x - zt * p.nyt;
    const int nx = p.nx, ny = p.ny, nz = p.nz, gxw = p.gxw, pitch = p.pitch;
    const int z0 = zt * p.bz, y0 = yt * p.by;
    const int gz = p.bz + p.oz + p.hz, gy = p.by + p.oy + p.hy;          // staged planes / rows
    const int zf = z0 - p.oz, yf = y0 - p.oy;                             // array plane / row of staged plane / row 0
    const int ngx = 2 * gxw;
    const int plane_words = gy * pitch;
    unsigned *S = lds, *M = lds + gz * plane_words;
    const size_t plane_elems = (size_t)ny * (size_t)nx;
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx;
    const unsigned border32 = p.border ? 0xffffffffu : 0u;
    if (tid == 0) { sweep_changed = 0; any_changed = 0; }

    // ---- stage the block (+ halo): state and mask as bits; out-of-array positions hold the border bit / mask 0
    for (int r = tid; r < gz * gy; r += kFillNT) {
        S[r * pitch] = border32;
        S[r * pitch + gxw + 1] = border32;
        M[r * pitch] = 0u;
        M[r * pitch + gxw + 1] = 0u;
    }
    const int ngran = gz * gy * ngx;
    for (int g = tid; g < ngran; g += kFillNT) {
        const int zi = g / (gy * ngx), rem = g - zi * (gy * ngx);
        const int yi = rem / ngx, col = rem - yi * ngx;
        const int z = zf + zi, y = yf + yi;
        const bool inside = (unsigned)z < (unsigned)nz && (unsigned)y < (unsigned)ny && 16 * col < nx;
        unsigned s16 = border32, m16 = 0u;
        if (inside) {
            const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(in + (size_t)z * plane_elems), 0, (int)plane_bytes + (RG ? 16 : 0), 0x00020000);
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(msk + (size_t)z * plane_elems), 0, (int)plane_bytes + (RG ? 16 : 0), 0x00020000);
            const unsigned off = (unsigned)(y * nx + 16 * col);
            const unsigned valid = nx - 16 * col >= 16 ? 0xffffu : ((1u << (nx - 16 * col)) - 1u);
            s16 = (pack16(__builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0)) & valid) | (border32 & ~valid);
            m16 = pack16(__builtin_amdgcn_raw_buffer_load_b128(rm, off, 0, 0)) & valid;
        }
        const int bo = ((zi * gy + yi) * pitch + 1) * 4 + 2 * col;
        *reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(S) + bo) = (unsigned short)s16;
        *reinterpret_cast<unsigned short *>(reinterpret_cast<unsigned char *>(M) + bo) = (unsigned short)m16;
    }
    __syncthreads();

    // ---- this thread's row: block plane zi, block row yi
    const int zi = tid / p.by, yi = tid - zi * p.by;
    const bool own = zi < p.bz && z0 + zi < nz && y0 + yi < ny;
    const int rbase = ((zi + p.oz) * gy + yi + p.oy) * pitch + 1;
    for (int sweep = 0; sweep < kFillSweeps; sweep++) {
        bool mine = false;
        if (own) {
            // (1) the taps of the window's rows
            for (int wc = 0; wc < gxw; wc++) {
                const int c = rbase + wc;
                const unsigned old = S[c], mk = M[c];
                if ((mk & ~old) == 0u) continue;                 // nothing left to set in this word
                unsigned acc = 0u;
                for (int r = 0; r < p.nrows; r++) {
                    const int tz = p.rows[r][0], ty = p.rows[r][2];
                    unsigned mm = (unsigned)p.rows[r][1];
                    const unsigned *a = S + c + ((tz - p.oz) * gy + (ty - p.oy)) * pitch;
                    const unsigned C = a[0];
                    unsigned L = 0, R = 0;
                    if (mm & ~(1u << p.ox)) { L = a[-1]; R = a[1]; }
                    while (mm) {
                        const int tx = __builtin_ctz(mm);
                        mm &= mm - 1;
                        const int dx = tx - p.ox;
                        acc |= dx == 0 ? C : dx > 0 ? __builtin_amdgcn_alignbit(R, C, (unsigned)dx) : __builtin_amdgcn_alignbit(C, L, (unsigned)(32 + dx));
                    }
                }
                const unsigned nw = old | (acc & mk);
                if (nw != old) { S[c] = nw; mine = true; }
            }
            // (2) fill along x inside the runs of the mask: upwards (towards larger x) ...
            if (p.fill_up) {
                unsigned cin = 0u;
                for (int wc = 0; wc < gxw; wc++) {
                    const int c = rbase + wc;
                    const unsigned s = S[c], mk = M[c], t = s & mk;
                    const unsigned long long U = (unsigned long long)mk + t + cin;
                    const unsigned u = (unsigned)U;
                    cin = (unsigned)(U >> 32);
                    const unsigned nw = s | (mk & (~u | t));
                    if (nw != s) { S[c] = nw; mine = true; }
                }
            }
            // ... and downwards: the same on bit-reversed words, from the last word to the first
            if (p.fill_down) {
                unsigned cin = 0u;
                for (int wc = gxw - 1; wc >= 0; wc--) {
                    const int c = rbase + wc;
                    const unsigned s = S[c], mk = __builtin_bitreverse32(M[c]), t = __builtin_bitreverse32(s) & mk;
                    const unsigned long long U = (unsigned long long)mk + t + cin;
                    const unsigned u = (unsigned)U;
                    cin = (unsigned)(U >> 32);
                    const unsigned nw = s | __builtin_bitreverse32(mk & (~u | t));
                    if (nw != s) { S[c] = nw; mine = true; }
                }
            }
        }
        if (mine) sweep_changed = 1;
        __syncthreads();
        const int ch = sweep_changed;
        __syncthreads();
        if (!ch) break;
        if (tid == 0) { sweep_changed = 0; any_changed = 1; }
        __syncthreads();
    }
    __syncthreads();

    // ---- write the block back (every voxel of the volume belongs to exactly one block)
    const int ogran = p.bz * p.by * ngx;
    for (int g = tid; g < ogran; g += kFillNT) {
        const int zo = g / (p.by * ngx), rem = g - zo * (p.by * ngx);
        const int yo = rem / ngx, col = rem - yo * ngx;
        const int z = z0 + zo, y = y0 + yo;
        if (z >= nz || y >= ny || 16 * col >= nx) continue;
        const unsigned w = *reinterpret_cast<const unsigned short *>(
            reinterpret_cast<const unsigned char *>(S) + (((zo + p.oz) * gy + yo + p.oy) * pitch + 1) * 4 + 2 * col);
        const u32x4 v = unpack16(w);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(out + (size_t)z * plane_elems), 0, (int)plane_bytes, 0x00020000);
        const unsigned o = (unsigned)(y * nx + 16 * col);
        const int nv = min(nx - 16 * col, 16);
        if (!RG || nv == 16) {
            __builtin_amdgcn_raw_buffer_store_b128(v, rout, o, 0, 0);
        } else {
            const unsigned o4 = (nv & 8) ? 8u : 0u, o2 = o4 + ((nv & 4) ? 4u : 0u), o1 = o2 + ((nv & 2) ? 2u : 0u);
            const unsigned d4 = (nv & 8) ? v.z : v.x;
            const unsigned d2s = o2 >= 8 ? (o2 >= 12 ? v.w : v.z) : (o2 >= 4 ? v.y : v.x);
            const unsigned d1w = o1 >= 8 ? (o1 >= 12 ? v.w : v.z) : (o1 >= 4 ? v.y : v.x);
            if (nv & 8) __builtin_amdgcn_raw_buffer_store_b64((u32x2){v.x, v.y}, rout, o, 0, 0);
            if (nv & 4) __builtin_amdgcn_raw_buffer_store_b32(d4, rout, o + o4, 0, 0);
            if (nv & 2) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)d2s, rout, o + o2, 0, 0);
            if (nv & 1) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(d1w >> (8 * (o1 & 3u))), rout, o + o1, 0, 0);
        }
    }
    if (tid == 0 && any_changed && flag) atomicOr(flag, 1);
}

static Knob g_fill_on{1};

// One launch of the masked dilation's block-wise fill towards its fixed point (see bitfill3_kernel).  `structure` /
// `origins` in the erosion form mi_binary_erosion takes with invert = 1 (the Python layer has mirrored the structure).
// MI_ERR_UNSUPPORTED (nothing launched) outside the envelope.
int bitfill3(const mi_array *in, const mi_array *out, const uint8_t *structure, const int64_t *sshape, const int *origins,
             const mi_array *mask, int border_value, int32_t *flag, hipStream_t s)
{
#define NOPE(msg) do { set_error("bitfill3: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (!g_fill_on || !g_bm_on) NOPE("switched off");
    if (!mask) NOPE("needs a mask (an unmasked dilation until stable fills the volume)");
    if (in->ndim != 3 || dtype_size(in->dtype) != 1 || dtype_size(out->dtype) != 1) NOPE("3-D 1-byte volumes only");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nx < 64 || nx > 2048) NOPE("rows of 64 .. 2048 voxels");
    if (ny * nx >= ((int64_t)1 << 31) || nz > (1 << 24) || ny > (1 << 24)) NOPE("plane too large");
    if (g_bm_on != 2 && nz * ny * nx < (1 << 18)) NOPE("small volume");
    if (((uintptr_t)in->data & 15) || ((uintptr_t)out->data & 15) || ((uintptr_t)mask->data & 15)) NOPE("needs 16-byte aligned data");
    FillParams p;
    memset(&p, 0, sizeof(p));
    int w[3], off[3];
    for (int d = 0; d < 3; d++) {
        if (sshape[d] < 1 || sshape[d] > 9) NOPE("structure extent > 9");
        w[d] = (int)sshape[d];
        off[d] = (int)(sshape[d] / 2 + origins[d]);
        if (off[d] < 0 || off[d] >= sshape[d]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    // monotone only when a voxel's own value is among its taps: the tap at offset 0 is structure[off]
    if (!structure[((int64_t)off[0] * w[1] + off[1]) * w[2] + off[2]]) NOPE("the structure does not hold its own centre: the iteration is not monotone");
    const bool ragged = (nx & 15) != 0;
    if (ragged) {
        if (!g_bm_ragged) NOPE("rows that are not a multiple of 16 bytes: switched off");
        const mi_array *arrs[2] = {in, mask};
        for (const mi_array *a : arrs) {
            void *base = nullptr;
            size_t size = 0;
            if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)a->data) != hipSuccess) {
                (void)hipGetLastError();
                NOPE("the extent of the allocation is unknown");
            }
            if ((uintptr_t)base + size < (uintptr_t)a->data + (size_t)(nz * ny * nx) + 16) NOPE("no 16 readable bytes after the array");
        }
    }
    p.nx = (int)nx; p.ny = (int)ny; p.nz = (int)nz;
    p.oz = off[0]; p.oy = off[1]; p.ox = off[2];
    p.hz = w[0] - 1 - off[0]; p.hy = w[1] - 1 - off[1];
    p.border = border_value != 0;
    int n = 0;
    for (int tz = 0; tz < w[0]; tz++)
        for (int ty = 0; ty < w[1]; ty++) {
            unsigned m = 0;
            for (int tx = 0; tx < w[2]; tx++)
                if (structure[((int64_t)tz * w[1] + ty) * w[2] + tx]) m |= 1u << tx;
            if (!m) continue;
            if (n == kBmMaxRows) NOPE("structure has too many rows");
            p.rows[n][0] = tz; p.rows[n][1] = (int)m; p.rows[n][2] = ty; n++;
            if (tz == off[0] && ty == off[1]) {
                p.fill_up = off[2] >= 1 && (m >> (off[2] - 1) & 1u);       // tap at dx = -1: x takes x - 1's value
                p.fill_down = off[2] + 1 < w[2] && (m >> (off[2] + 1) & 1u);
            }
        }
    p.nrows = n;
    p.gxw = (int)((nx + 31) / 32);
    p.pitch = p.gxw + 2;
    if (!(p.pitch & 1)) p.pitch++;
    // block: 16 x 16 rows (one per thread), fewer planes when whole rows of bits are long (state + mask <= 60 KiB)
    p.by = (int)std::min<int64_t>(16, ny);
    p.bz = (int)std::min<int64_t>(kFillNT / p.by, nz);
    auto lds_bytes = [&](int bz) { return (size_t)2 * (bz + p.oz + p.hz) * (p.by + p.oy + p.hy) * p.pitch * 4; };
    while (p.bz > 1 && lds_bytes(p.bz) > 60 * 1024) p.bz--;
    if (lds_bytes(p.bz) > 60 * 1024) NOPE("rows too wide for a block");
    p.nzt = (int)((nz + p.bz - 1) / p.bz); p.nyt = (int)((ny + p.by - 1) / p.by);
    const size_t lds = lds_bytes(p.bz);
    const unsigned char *ip = (const unsigned char *)in->data, *mp = (const unsigned char *)mask->data;
    unsigned char *op = (unsigned char *)out->data;
    static PerDeviceOnce attr;
    if (!attr) {
        MI_HIP(hipFuncSetAttribute((const void *)bitfill3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        MI_HIP(hipFuncSetAttribute((const void *)bitfill3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        attr = true;
    }
    const unsigned grid = (unsigned)(p.nzt * p.nyt);
    if (ragged) hipLaunchKernelGGL((bitfill3_kernel<true>), dim3(grid), dim3(kFillNT), lds, s, ip, op, mp, p, flag);
    else hipLaunchKernelGGL((bitfill3_kernel<false>), dim3(grid), dim3(kFillNT), lds, s, ip, op, mp, p, flag);
    MI_HIP(hipGetLastError());
    note_kernel("mi::bitfill3_kernel<%s> grid=%u block=%dx%d rows x planes (masked dilation filled to its fixed point block by block: row fills by carry, 1 bit per voxel)",
                ragged ? "ragged" : "aligned", grid, p.by, p.bz);
    return MI_OK;
#undef NOPE
}

// k fused iterations on a 3-D volume of 1-byte voxels; MI_ERR_UNSUPPORTED (nothing launched) outside the envelope.
int bitmorph3(const mi_array *in, const mi_array *out, const uint8_t *structure, const int64_t *sshape, const int *origins,
              const mi_array *mask, int border_value, int invert, int k, int32_t *flags, hipStream_t s, int open_close)
{
    // open_close: 0 = k iterations of one operation (`invert` says which); 1 = opening, 2 = closing with k iterations of
    // each half in ONE launch (2 k stages; `invert` is ignored)
#define NOPE(msg) do { set_error("bitmorph3: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (!g_bm_on) NOPE("switched off (mi_debug_set_bitmorph)");
    const int kiter = k;
    if (open_close) {
        invert = open_close == 2;                           // closing starts with the dilation
        k = 2 * k;
        for (int d = 0; d < in->ndim && d < 3; d++)
            if (!(sshape[d] & 1) || origins[d] != 0) NOPE("opening / closing in one launch: odd structure extents, origin 0");
    }
    if (dtype_size(in->dtype) != 1 || dtype_size(out->dtype) != 1) NOPE("1-byte volumes only");
    if (in->ndim != 3 && !(in->ndim == 2 && g_bm_2d)) NOPE("3-D volumes (and 2-D images) only");
    if (k < 1 || k > kBmMaxK) NOPE("1 .. 8 fused iterations");
    // a 2-D image is a one-plane volume with a one-plane structure: the y tiles (x tiles beyond 1024 columns) are the
    // parallelism, every workgroup a one-plane "stream" -- no plane pipeline, many short workgroups
    const int pad = 3 - in->ndim;
    const int64_t nz = pad ? 1 : in->shape[0], ny = in->shape[1 - pad], nx = in->shape[2 - pad];
    int w[3] = {1, 1, 1}, off[3] = {0, 0, 0};
    for (int d = 0; d < in->ndim; d++) {
        if (sshape[d] < 1 || sshape[d] > 31) NOPE("structure extent > 31");
        w[pad + d] = (int)sshape[d];
        off[pad + d] = (int)(sshape[d] / 2 + origins[d]);
        if (off[pad + d] < 0 || off[pad + d] >= sshape[d]) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    // images: one launch of the byte kernel is as fast or faster (13-15 us, host bound; 8192^2: 27 against 44 us); what the
    // bit kernel saves there are LAUNCHES -- fused iterations up to ~12 Mpixels (2048^2 x 4 iterations: 40 -> 21 us) and
    // opening / closing at every size (50 -> 16-21 us; profiles/r6_binary_images.txt)
    if (pad && g_bm_2d != 2 && !(open_close || (k >= 2 && ny * nx <= (int64_t)12 << 20)))
        NOPE("images: single iterations (and long fused runs on large images) stay on the byte kernel");
    if (nx < 64) NOPE("rows of at least 64 bytes");
    const bool ragged = (nx & 15) != 0;
    if (ragged) {
        // the ragged build reads up to 16 bytes past the last row of a plane (masked out): past the last plane that is past
        // the array -- only taken when the allocation the array lies in (ours or anybody's) has those bytes
        if (!g_bm_ragged) NOPE("rows that are not a multiple of 16 bytes: switched off");
        const mi_array *arrs[2] = {in, mask};
        for (const mi_array *a : arrs) {
            if (!a) continue;
            void *base = nullptr;
            size_t size = 0;
            if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)a->data) != hipSuccess) {
                (void)hipGetLastError();
                NOPE("rows that are not a multiple of 16 bytes: the extent of the allocation is unknown");
            }
            const uintptr_t end = (uintptr_t)a->data + (size_t)(nz * ny * nx);
            if ((uintptr_t)base + size < end + 16) NOPE("rows that are not a multiple of 16 bytes: no 16 readable bytes after the array");
        }
    }
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
    p.kflip = open_close ? kiter : 0;
    p.invert = invert != 0;
    p.border = invert ? !border_value : (border_value != 0);
    // rows (dz, dy) of the structure that hold a tap, grouped by their dx mask (the kernel ANDs the rows of a group word
    // by word and shifts once per group); a group is padded to an even count by repeating its last row (AND is idempotent)
    {
        struct Row { int tz, ty; unsigned m; };
        int n = 0;
        // pass 0: the structure as the first operation takes it; pass 1 (opening / closing): the other orientation.  The
        // caller's structure is the EROSION's; a dilation is the erosion of the complement by the mirrored structure.
        for (int pass = 0; pass < (open_close ? 2 : 1); pass++) {
            const bool mirrored = open_close ? (pass == 0) == (open_close == 2) : false;
            std::vector<Row> rows;
            for (int tz = 0; tz < w[0]; tz++)
                for (int ty = 0; ty < w[1]; ty++) {
                    unsigned m = 0;
                    for (int tx = 0; tx < w[2]; tx++) {
                        const int64_t idx = mirrored ? ((int64_t)(w[0] - 1 - tz) * w[1] + (w[1] - 1 - ty)) * w[2] + (w[2] - 1 - tx)
                                                     : ((int64_t)tz * w[1] + ty) * w[2] + tx;
                        if (structure[idx]) m |= 1u << tx;
                    }
                    if (m) rows.push_back({tz, ty, m});
                }
            std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.m < b.m; });
            const int n0 = n;
            for (size_t i = 0; i < rows.size();) {
                size_t j = i;
                while (j < rows.size() && rows[j].m == rows[i].m) j++;
                const size_t cnt = j - i, padded = cnt + (cnt & 1);
                if (n + (int)padded > kBmMaxRows) NOPE("structure has too many rows");
                for (size_t q = 0; q < padded; q++) {
                    const Row &r = rows[i + std::min(q, cnt - 1)];
                    p.rows[n][0] = r.tz | (q + 1 == padded ? 256 : 0);
                    p.rows[n][1] = (int)r.m;
                    p.rows[n][2] = r.ty - off[1];           // x pitch x 4 once the pitch is known
                    n++;
                }
                i = j;
            }
            if (pass == 0) p.nrows = n;                     // 0: an empty structure erodes nothing (output = true)
            else p.nrows2 = n - n0;
        }
    }
    p.ns = p.wz + 1;
    p.nms = mask ? k * (1 + p.hz) - p.hz + 1 : 0;

    // a structure the kernel has straight-line code for (origin 0)?
    int kind = 0;
    if (w[0] == 3 && w[1] == 3 && w[2] == 3 && off[0] == 1 && off[1] == 1 && off[2] == 1) {
        int cnt[4] = {0, 0, 0, 0};                          // set taps by city-block distance from the centre
        bool shells = true;
        for (int t = 0; t < 27; t++) {
            const int d = abs(t / 9 - 1) + abs(t / 3 % 3 - 1) + abs(t % 3 - 1);
            if (structure[t]) cnt[d]++;
        }
        const int full[4] = {1, 6, 12, 8};
        for (int d = 0; d < 4; d++) shells = shells && (cnt[d] == 0 || cnt[d] == full[d]);
        if (shells && cnt[0] == 1 && cnt[1] == 6) {
            if (cnt[2] == 0 && cnt[3] == 0) kind = 1;
            else if (cnt[2] == 12 && cnt[3] == 0) kind = 2;
            else if (cnt[2] == 12 && cnt[3] == 8) kind = 3;
        }
    }
    if (mask && kind > 1) kind = 0;                           // masked runs: the default structure and the table only
    if (g_bm_kind0) kind = 0;

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
    for (int r = 0; r < p.nrows + p.nrows2; r++) p.rows[r][2] *= p.pitch * 4;
    const int halo_y = k * (p.oy + p.hy);
    const int ngx = 2 * p.gxw;
    const int nt = kBmNT;
    const int lds_fixed = (p.oy + p.hy) * p.pitch + nt;      // slack rows + dump words
    const int max_gy = std::min(8 * nt / ngx, (int)((kBmMaxLds / 4 - lds_fixed) / ((size_t)((k + 1) * p.ns + p.nms) * p.pitch)));
    if (max_gy - halo_y < 2) NOPE("structure / iteration count too large for one tile");
    // at most three (else four) granules per thread -- 7 (5) resident waves per SIMD; 1024^3: 8-row tiles with NL 3 run
    // 384-400 us, 29-row tiles with NL 8 408 us -- while that leaves tiles at least as tall as their halo
    // (the run-time table pays a fixed scalar cost per step: large tiles amortise it -- ball(2) on 1024^3: 497 us on
    // 27-row tiles, 862 us on 8-row tiles)
    const int min_ty = std::max(halo_y, 4);
    const int max_gy4 = kind == 0 ? max_gy : 3 * nt / ngx - halo_y >= 4 * halo_y ? 3 * nt / ngx : 4 * nt / ngx;
    const int cus = device_cus();
    const int64_t slots = 2 * (int64_t)cus;                 // two resident workgroups per CU saturate the memory system
    int best_ty = 0, best_nzc = 1;
    const int ty_hi = (int)std::min<int64_t>((max_gy4 - halo_y >= min_ty ? std::min(max_gy, max_gy4) : max_gy) - halo_y, ny);
    {
        // the search is a few hundred candidates; one remembered plan per host thread makes a repeated call free (a
        // 256^3 call is ~15 us of GPU time: the search must not cost more than the launch)
        struct Key { int64_t nx, ny, nz; int k, w0, w1, w2, o0, o1, o2, mask, nt, cus, kind, ty, nzc; };
        static thread_local Key last_key = {};
        static thread_local bool have = false;
        const Key key = {nx, ny, nz, k, w[0], w[1], w[2], off[0], off[1], off[2], (mask ? 1 : 0) + 2 * open_close, nt, cus, kind, 0, 0};
        if (have && !memcmp(&key, &last_key, offsetof(Key, ty))) {
            best_ty = last_key.ty;
            best_nzc = last_key.nzc;
        } else {
            double best = 1e300;
            for (int ty = std::min(ty_hi, 2); ty <= ty_hi; ty++) {
                const int gy = ty + halo_y;
                const int64_t tiles = ((ny + ty - 1) / ty) * p.nxt;
                // what a step costs is what its threads issue: the staged granules rounded up to whole rounds of the
                // workgroup (and to an instantiated NL), the same for the output granules, a fixed part per stage
                int nlr = (gy * ngx + nt - 1) / nt;
                nlr = nlr == 7 ? 8 : nlr;
                const int nor = (ty * 2 * p.txw + nt - 1) / nt;
                // more than four granules per thread cost registers, i.e. resident waves (NL 3: 7 waves per SIMD, 4: 5,
                // 6: 4, 8: 3); the halo rows a small tile re-reads come out of the L2 (measured: 1024^3 runs 6 % faster
                // on 8-row tiles with NL 3 than on 29-row tiles with NL 8)
                const double occ = nlr <= 4 ? 1.0 : nlr <= 6 ? 1.12 : 1.25;
                const double step = ((double)nlr * nt * (mask ? 1.6 : 1.0) + 0.7 * nor * nt + 192.0 + 96.0 * k) * occ;
                for (int64_t rounds = 1; rounds <= 8; rounds++) {
                    // the largest chunk count that still fits `rounds` waves of workgroups, and one chunk fewer planes
                    const int64_t fit = std::max<int64_t>(1, std::min<int64_t>(nz, slots * rounds / tiles));
                    for (int64_t nzc : {fit, std::min<int64_t>(nz, fit + 1)}) {
                        const int chunk = (int)((nz + nzc - 1) / nzc);
                        const int real = (int)((nz + chunk - 1) / chunk);
                        const int64_t wgs = tiles * real;
                        const double cost = (double)((wgs + slots - 1) / slots) * (chunk + k * p.wz + 1) * step;
                        if (cost < best) { best = cost; best_ty = ty; best_nzc = real; }
                    }
                    if (fit >= nz) break;
                }
            }
            last_key = key;
            last_key.ty = best_ty;
            last_key.nzc = best_nzc;
            have = true;
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
    const size_t lds = ((size_t)((k + 1) * p.ns + p.nms) * p.gy * p.pitch + lds_fixed) * 4;
    if (lds > (size_t)kBmMaxLds) NOPE("does not fit LDS");
    const int nl = (p.gy * ngx + nt - 1) / nt;

    const unsigned char *ip = (const unsigned char *)in->data;
    unsigned char *op = (unsigned char *)out->data;
    const unsigned char *mp = mask ? (const unsigned char *)mask->data : nullptr;
#define GO4(NLV, KV, RGV) return mp ? launch_bitmorph<true, NLV, 256, (KV) <= 1 ? (KV) : 0, RGV>(ip, op, mp, p, lds, flags, s) \
                                    : launch_bitmorph<false, NLV, 256, KV, RGV>(ip, op, mp, p, lds, flags, s)
#define GO3(NLV, KV) do { if (ragged) { GO4(NLV, KV, true); } else { GO4(NLV, KV, false); } } while (0)
#define GO(NLV) do { if (kind == 1) { GO3(NLV, 1); } else if (kind == 2) { GO3(NLV, 2); } else if (kind == 3) { GO3(NLV, 3); } else { GO3(NLV, 0); } } while (0)
    if (nl <= 1) { GO(1); }
    if (nl <= 2) { GO(2); }
    if (nl <= 3) { GO(3); }
    if (nl <= 4) { GO(4); }
    if (nl <= 5) { GO(5); }
    if (nl <= 6) { GO(6); }
    GO(8);
#undef GO
#undef GO3
#undef GO4
#undef NOPE
}

}  // namespace mi

extern "C" int mi_debug_set_bitfill(int on)             // 0: runs until stable iterate the global operator (batches of fused iterations)
{
    mi::g_fill_on = on;
    return MI_OK;
}

extern "C" int mi_debug_set_bitmorph_ragged(int on)     // 0: rows that are not a multiple of 16 bytes keep the extended-rows / generic routes
{
    mi::g_bm_ragged = on;
    return MI_OK;
}

extern "C" int mi_debug_set_bitmorph_2d(int on)         // 0: 2-D images keep the byte kernel; 2: every image call the bit kernel can take
{
    mi::g_bm_2d = on;
    return MI_OK;
}

extern "C" int mi_debug_set_bitmorph_table(int on)      // 1: the run-time structure table even for the built-in structures
{
    mi::g_bm_kind0 = on;
    return MI_OK;
}

extern "C" int mi_debug_set_bitmorph(int on, int ty, int nzc)
{
    mi::g_bm_on = on;
    mi::g_bm_ty = ty;
    mi::g_bm_nzc = nzc;
    return MI_OK;
}
