// stencil3s.hip -- dense 3 x 3 x 3 / 5 x 5 x 5 (/ 7 x 7 x 7 with float accumulation) correlate of float32 volumes, z taken as a SCATTER over register
// accumulators ("each plane is read from LDS once").
//
// Reference path replaced: correlate / convolve with a small dense kernel, cupyimg/scipy/ndimage/filters.py:65-210 ->
// :441-495 (generated nested tap loop _filters_core.py:298-324: one global load per tap per voxel); accumulator type
// by `dtype_mode` (filters.py:74,470-487): "float" = float32, default = float64.
//
// Why another kernel (round 6).  stencil3_kernel (stencil3d.hip) keeps the wz input planes of the window in an LDS ring
// and re-reads every one of them for every output plane: W LDS passes over each plane, three 16-byte reads per row and
// pass, weights fetched from the kernel arguments inside the tap loop (as doubles, converted per use when the
// accumulator is float32).  3 x 3 x 3 on 512^3 ran at 0.31 of the HBM roofline, 5 x 5 x 5 at 0.08.  Here:
//
//   * a workgroup (8 waves) owns a 256 x TY column and streams along z; only the CURRENT plane (and the one being
//     staged) is in LDS -- 2 slots instead of W + 1;
//   * when plane q arrives, every lane adds its contribution to the W output planes it belongs to: W accumulator sets of
//     RW rows x 4 voxels stay in registers; the set that has seen its last plane is stored and becomes the set of the
//     plane that enters next (the plane loop is unrolled W times so that every set has a static name);
//   * an input row is read ONCE per plane (one 16-byte read per lane; the W/2 neighbours either side come from the
//     adjacent lanes by DPP, the tile's own halo by one small read in the edge lanes) and feeds W rows x W planes x W taps;
//   * weights are kernel arguments of the accumulator's type, addressed with compile-time indices: scalar registers
//     (the 125 of the 5^3 window do not fit -- as pairs for v_pk_fma_f32 they would need 250: in float mode they live in the
//     lanes of two vector registers and are read with one v_readlane per weight and plane; LDS-resident weights were tried
//     and lost: profiles/r6_stencil.txt);
//   * float32 accumulation uses v_pk_fma_f32 on x-pairs; float64 accumulation is v_mul_f64 + v_add_f64 in the order
//     (z, y, x) of the window -- the plane-by-plane scatter adds the taps of one output in exactly that order, so the
//     result is bit-identical to stencil3_kernel / corr3_kernel / SciPy's NI_Correlate.
//
// Envelope: float32 in and out, 3-D, window 3^3 or 5^3 (7^3: float accumulation only) with no zero weight (a zero weight is SKIPPED by the reference, which
// matters for non-finite samples: those windows stay on stencil3_kernel's mask path), origin 0 along x.
#include "nd_common.hpp"
#include "sep_common.hpp"

namespace mi {

constexpr int kSsNW = 8;            // waves per workgroup
constexpr int kSsPitch = 264;       // floats per LDS row: 4 halo + 256 + 4 halo

template <typename Acc>
struct ScatterParams {
    int nx, ny, nz;
    int oz, oy;                 // w/2 + origin along z, y
    int mode;
    float cval;
    int zc, nzc, nxt, nyt;
    int pad_;
    Acc w[343];                 // [tz][ty][tx] (W^3 of them)
};

// FMA64: float64 accumulation with v_fma_f64 -- chosen by the host only when every weight is a float32 value: the
// product of a float32 sample and such a weight is EXACT in float64 (24 + 24 significant bits), so the fused operation
// rounds once exactly where mul + add round once, and the result is still bit-identical to SciPy's; v_mul_f64 + v_add_f64
// cost 16 cycles per wave, v_fma_f64 8 (FP64 runs at half the FP32 rate on CDNA4), and the default mode is bound by them.
template <int W, typename Acc, int RW, bool FMA64 = false, bool RG = false>
__global__ void __launch_bounds__(kSsNW * 64)
stencil3s_kernel(const float *__restrict__ in, float *__restrict__ out, const ScatterParams<Acc> p)
{
    constexpr int RX = W / 2;
    constexpr int TY = kSsNW * RW;
    constexpr int ROWS = TY + W - 1;                      // staged rows of a plane
    constexpr int RPW = (ROWS + kSsNW - 1) / kSsNW;       // staged rows per wave
    constexpr int SLOT = ROWS * kSsPitch;
    constexpr bool F32 = std::is_same<Acc, float>::value;
    constexpr bool LANEW = W >= 5 && F32;                 // weights in the lanes of vector registers (see the tap loop)
    // lane layout of the weights: 5^3 -- lane k & 63 of register k >> 6; 7^3 -- window row g = tz * 7 + ty in register g / 9,
    // lanes 7 (g % 9) .. + 6 (a row's seven weights in ONE register: the asm statement of a row has 30 operands as it is)
    constexpr int NWR = W == 5 ? 2 : W == 7 ? 6 : 1;
    __shared__ __attribute__((aligned(16))) float ring[2 * SLOT];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b = blockIdx.x;
    const int total = p.nxt * p.nyt * p.nzc;
    if ((total & 7) == 0) b = (b & 7) * (total >> 3) + (b >> 3);     // one contiguous tile range per XCD
    const int per_chunk = p.nxt * p.nyt;
    const int zci = b / per_chunk;
    const int rem = b - zci * per_chunk;
    const int yt = rem / p.nxt, xt = rem - yt * p.nxt;

    const int nx = p.nx, ny = p.ny, nz = p.nz, mode = p.mode;
    const int x0 = xt * 256, y0 = yt * TY;
    const int zs = zci * p.zc, ze = min(zs + p.zc, nz);
    const int nout = ze - zs;
    const int ty_act = min(TY, ny - y0);
    // RG -- rows of any length (r6, late; a build of its own: the aligned one keeps its registers): 16-byte buffer loads need
    // 4-byte alignment only, so a row is staged from wherever it starts; the last lane of a row's last tile holds `tail` floats of
    // its row followed by the head of the next one (zeros beyond the plane), which the right-hand halo floats -- written at the
    // row's true end, after the row -- overwrite in LDS
    const int width = RG ? min(256, nx - x0) : 4 * min(64, (nx - x0) >> 2);
    const int nlanes = (width + 3) >> 2;
    const int tail = RG ? width - 4 * (nlanes - 1) : 4;   // floats of its row the last lane holds (4: a whole float4)
    const unsigned plane_bytes = (unsigned)ny * (unsigned)nx * 4u;
    const size_t plane_elems = (size_t)ny * (size_t)nx;

    // ---- staging recipe (as stencil3_kernel): wave w stages rows w, w + 8, ...; lanes 0..7 the eight halo floats
    unsigned voff_main[RPW], voff_halo[RPW];
    bool row_const[RPW];
    int lds_row[RPW];
    const bool halo_lane = lane < 8;
    const int xh = lane < 4 ? x0 - 4 + lane : x0 + width + (lane - 4);
    const int xsrc = halo_lane ? bmap_near<int>(xh, nx, mode) : -1;
    const int halo_pos = lane < 4 ? lane : 4 + width + (lane - 4);
#pragma unroll
    for (int k = 0; k < RPW; k++) {
        const int j = wave + kSsNW * k;
        const int ysrc = j < ROWS ? bmap_near<int>(y0 - p.oy + j, ny, mode) : -2;
        row_const[k] = ysrc == -1;
        lds_row[k] = j < ROWS ? j * kSsPitch : -1;
        voff_main[k] = (ysrc >= 0 && lane < nlanes) ? (unsigned)(ysrc * nx + x0 + 4 * lane) * 4u : kOOB;
        voff_halo[k] = (ysrc >= 0 && xsrc >= 0) ? (unsigned)(ysrc * nx + xsrc) * 4u : kOOB;
    }
    const bool halo_const = halo_lane && xsrc < 0;        // only in constant mode

    u32x4 pm[RPW];
    float ph[RPW];
    bool pconst = false;
    auto fetch = [&](int q) {                             // input plane q of the chunk (0 = plane zs - oz)
        int zsrc = zs - p.oz + q;
        if ((unsigned)zsrc >= (unsigned)nz) zsrc = bmap<int>(zsrc, nz, mode);
        pconst = zsrc < 0;
        zsrc = __builtin_amdgcn_readfirstlane(max(zsrc, 0));
        const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(in + (size_t)zsrc * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            pm[k] = __builtin_amdgcn_raw_buffer_load_b128(rin, pconst ? kOOB : voff_main[k], 0, 0);
            ph[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rin, pconst ? kOOB : voff_halo[k], 0, 0));
        }
    };
    auto stage = [&](int q) {                             // registers -> slot q & 1
        float *slot = ring + (q & 1) * SLOT;
#pragma unroll
        for (int k = 0; k < RPW; k++) {
            if (lds_row[k] < 0) continue;
            const bool c = pconst || row_const[k];
            float4 f = make_float4(__uint_as_float(pm[k].x), __uint_as_float(pm[k].y), __uint_as_float(pm[k].z), __uint_as_float(pm[k].w));
            if (c) f = make_float4(p.cval, p.cval, p.cval, p.cval);
            if (lane < nlanes) *reinterpret_cast<float4 *>(slot + lds_row[k] + 4 + 4 * lane) = f;
            if (halo_lane) slot[lds_row[k] + halo_pos] = (c || halo_const) ? p.cval : ph[k];
        }
    };

    const int r0 = wave * RW;
    unsigned ovoff[RW];
#pragma unroll
    for (int rr = 0; rr < RW; rr++)
        ovoff[rr] = (r0 + rr < ty_act && lane < nlanes) ? (unsigned)((y0 + r0 + rr) * nx + x0 + 4 * lane) * 4u : kOOB;
    const bool first_lane = lane == 0, last_lane = lane == nlanes - 1;

    // accumulator sets: A[a] belongs to one output plane at a time
    Acc A[W][RW][4];
#pragma unroll
    for (int a = 0; a < W; a++)
#pragma unroll
        for (int rr = 0; rr < RW; rr++)
#pragma unroll
            for (int c = 0; c < 4; c++) A[a][rr][c] = (Acc)0;

    f32x2 B[W][RW][2];                                     // LANEW: the same sets as register PAIRS (what the asm tap loop updates)
#pragma unroll
    for (int a = 0; a < W; a++)
#pragma unroll
        for (int rr = 0; rr < RW; rr++) B[a][rr][0] = B[a][rr][1] = (f32x2){0.f, 0.f};
    float wlane[NWR];
#pragma unroll
    for (int r = 0; r < NWR; r++) wlane[r] = 0.f;
    if constexpr (LANEW && W == 5) {
        wlane[0] = p.w[lane];
        wlane[1] = lane + 64 < W * W * W ? p.w[lane + 64] : 0.f;
    }
    if constexpr (LANEW && W == 7) {
#pragma unroll
        for (int r = 0; r < NWR; r++) {
            const int g = 9 * r + lane / 7;
            wlane[r] = (lane < 63 && g < W * W) ? p.w[g * 7 + lane % 7] : 0.f;
        }
    }
    const int nsteps = nout + W - 1;
    fetch(0);
    stage(0);
    __syncthreads();

    // one plane: PH = q mod W (static).  Set PH starts here (tap plane tz = 0 of output q), set (PH + 1) % W takes its
    // last tap plane (tz = W - 1 of output q - W + 1) and is stored.
    auto step = [&](auto ph_tag, int q) {
        constexpr int PH = decltype(ph_tag)::value;
        if (q + 1 < nsteps) fetch(q + 1);                 // in flight during the tap loop
#pragma unroll
        for (int rr = 0; rr < RW; rr++)
#pragma unroll
            for (int c = 0; c < 4; c++) A[PH][rr][c] = (Acc)0;
#pragma unroll
        for (int rr = 0; rr < RW; rr++) B[PH][rr][0] = B[PH][rr][1] = (f32x2){0.f, 0.f};
        const float *slot = ring + (q & 1) * SLOT + 4 + 4 * lane;
        // the 4 + 2 RX samples a lane needs of staged row r0 + i: its own four, RX either side from the neighbouring lanes
        // (DPP), the tile halo in the edge lanes
        auto read_row = [&](int i, float (&d)[4 + 2 * RX]) {
            const float *rowp = slot + (r0 + i) * kSsPitch;
            const float4 C = *reinterpret_cast<const float4 *>(rowp);
            // the RX samples either side: the neighbouring lanes' C, the tile halo in the edge lanes
            d[RX] = C.x; d[RX + 1] = C.y; d[RX + 2] = C.z; d[RX + 3] = C.w;
            if constexpr (RX == 1) {
                float hl = 0.f, hr = 0.f;
                if (first_lane) hl = rowp[-1];
                if (last_lane) hr = rowp[4];
                const float l = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(C.w), 0x138, 0xf, 0xf, false));
                const float r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(C.x), 0x130, 0xf, 0xf, false));
                d[0] = first_lane ? hl : l;
                d[5] = last_lane ? hr : r;
            } else if constexpr (RX == 3) {
                float hl[3] = {0.f, 0.f, 0.f}, hr[3] = {0.f, 0.f, 0.f};
                if (first_lane) { hl[0] = rowp[-3]; hl[1] = rowp[-2]; hl[2] = rowp[-1]; }
                if (last_lane) { hr[0] = rowp[4]; hr[1] = rowp[5]; hr[2] = rowp[6]; }
                const float cv[4] = {C.x, C.y, C.z, C.w};
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const float l = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cv[1 + j]), 0x138, 0xf, 0xf, false));
                    const float r = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(cv[j]), 0x130, 0xf, 0xf, false));
                    d[j] = first_lane ? hl[j] : l;
                    d[7 + j] = last_lane ? hr[j] : r;
                }
            } else {
                float2 hl = make_float2(0.f, 0.f), hr = make_float2(0.f, 0.f);
                if (first_lane) hl = *reinterpret_cast<const float2 *>(rowp - 2);
                if (last_lane) hr = *reinterpret_cast<const float2 *>(rowp + 4);
                const float l0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(C.z), 0x138, 0xf, 0xf, false));
                const float l1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(C.w), 0x138, 0xf, 0xf, false));
                const float r0v = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(C.x), 0x130, 0xf, 0xf, false));
                const float r1v = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(C.y), 0x130, 0xf, 0xf, false));
                d[0] = first_lane ? hl.x : l0;
                d[1] = first_lane ? hl.y : l1;
                d[6] = last_lane ? hr.x : r0v;
                d[7] = last_lane ? hr.y : r1v;
            }
        };
        if constexpr (LANEW) {
            // 5^3, float accumulation: the 125 weights live in the LANES of two registers (lane k of wlane[k >> 6] = w[k]) and
            // come out by v_readlane where they are used -- ONE per weight and plane: the window is walked one row ty at a
            // time with the RW input rows that share it live, so the four packed FMAs of a weight stand together.  As
            // kernel arguments they were 250 scalar registers (pairs for v_pk_fma_f32): spilled to lanes by the compiler and
            // read back 3 700 times per 2 500 FMAs.  (The registers pass through an empty asm per plane: the reads must not
            // be hoisted out of the plane loop, where they would be 125 live scalars again.)
            unsigned wb[NWR];
#pragma unroll
            for (int r = 0; r < NWR; r++) {
                wb[r] = __float_as_uint(wlane[r]);
                asm volatile("" : "+v"(wb[r]));
            }
            static_assert(!LANEW || RW == 2, "the lane-weight tap loop is written for two rows per wave");
            // pairs of a staged row: P[j] = (d[j], d[j + 1]) -- what v_pk_fma_f32 takes for the output pairs (0, 1) [tap tx: j = tx]
            // and (2, 3) [j = tx + 2]
            constexpr int NP = 3 + 2 * RX;
            f32x2 P[RW + W - 1][NP];
            auto read_pairs = [&](auto II) {
                constexpr int i = decltype(II)::value;
                float d[4 + 2 * RX];
                read_row(i, d);
#pragma unroll
                for (int j = 0; j < NP; j++) P[i][j] = (f32x2){d[j], d[j + 1]};
            };
            static_for<RW - 1>([&](auto II) { read_pairs(II); });
            static_for<W>([&](auto TY) {
                constexpr int ty = decltype(TY)::value;
                read_pairs(std::integral_constant<int, ty + RW - 1>{});
                static_for<W>([&](auto TZ) {
                    constexpr int tz = decltype(TZ)::value;
                    constexpr int a = (PH - tz + 2 * W) % W;      // the set of output plane q - tz
                    // The W weights of window row (tz, ty) and their 4 W packed FMAs as ONE statement: a weight is ONE
                    // v_readlane into the low half of a scalar pair (op_sel_hi 0 on that operand: both halves of the packed
                    // product take the low dword, the high scalar is never read), two pairs in turn, so that the 2 wait states a
                    // VALU-written scalar needs before a VALU reads it are filled with the previous weight's FMAs.  Written in
                    // C++ (readlane builtin + splat) the compiler hoisted all 125 reads of the 5^3 window, spilled the 250 scalars
                    // back into lanes and fetched each with two v_readlane + s_nop: 1 300 issue slots per plane for 500 FMAs.
                    // (asm operands inside a lambda do not capture: name the registers first)
                    f32x2 &a0 = B[a][0][0], &a1 = B[a][0][1], &b0 = B[a][1][0], &b1 = B[a][1][1];
                    const f32x2 pa0 = P[ty][0], pa1 = P[ty][1], pa2 = P[ty][2], pa3 = P[ty][3], pa4 = P[ty][4], pa5 = P[ty][5], pa6 = P[ty][6];
                    const f32x2 pb0 = P[ty + 1][0], pb1 = P[ty + 1][1], pb2 = P[ty + 1][2], pb3 = P[ty + 1][3], pb4 = P[ty + 1][4],
                                pb5 = P[ty + 1][5], pb6 = P[ty + 1][6];
#define MI_SS_FMA4(S, LO, HI)                                                   \
    "v_pk_fma_f32 %[a0], %[pa" #LO "], " S ", %[a0] op_sel_hi:[1,0,1]\n\t"      \
    "v_pk_fma_f32 %[a1], %[pa" #HI "], " S ", %[a1] op_sel_hi:[1,0,1]\n\t"      \
    "v_pk_fma_f32 %[b0], %[pb" #LO "], " S ", %[b0] op_sel_hi:[1,0,1]\n\t"      \
    "v_pk_fma_f32 %[b1], %[pb" #HI "], " S ", %[b1] op_sel_hi:[1,0,1]\n\t"
                    if constexpr (W == 5) {
                        constexpr int k0 = (tz * W + ty) * W;
                        const unsigned w0 = wb[k0 >> 6], w1 = wb[(k0 + 1) >> 6], w2 = wb[(k0 + 2) >> 6], w3 = wb[(k0 + 3) >> 6],
                                       w4 = wb[(k0 + 4) >> 6];
                        asm volatile(
                            "v_readlane_b32 s90, %[w0], %[k0]\n\t"
                            "v_readlane_b32 s92, %[w1], %[k1]\n\t"
                            "s_nop 0\n\t"
                            MI_SS_FMA4("s[90:91]", 0, 2)
                            "v_readlane_b32 s90, %[w2], %[k2]\n\t"
                            MI_SS_FMA4("s[92:93]", 1, 3)
                            "v_readlane_b32 s92, %[w3], %[k3]\n\t"
                            MI_SS_FMA4("s[90:91]", 2, 4)
                            "v_readlane_b32 s90, %[w4], %[k4]\n\t"
                            MI_SS_FMA4("s[92:93]", 3, 5)
                            MI_SS_FMA4("s[90:91]", 4, 6)
                            : [a0] "+v"(a0), [a1] "+v"(a1), [b0] "+v"(b0), [b1] "+v"(b1)
                            : [pa0] "v"(pa0), [pa1] "v"(pa1), [pa2] "v"(pa2), [pa3] "v"(pa3), [pa4] "v"(pa4), [pa5] "v"(pa5), [pa6] "v"(pa6),
                              [pb0] "v"(pb0), [pb1] "v"(pb1), [pb2] "v"(pb2), [pb3] "v"(pb3), [pb4] "v"(pb4), [pb5] "v"(pb5), [pb6] "v"(pb6),
                              [w0] "v"(w0), [w1] "v"(w1), [w2] "v"(w2), [w3] "v"(w3), [w4] "v"(w4),
                              [k0] "n"(k0 & 63), [k1] "n"((k0 + 1) & 63), [k2] "n"((k0 + 2) & 63), [k3] "n"((k0 + 3) & 63),
                              [k4] "n"((k0 + 4) & 63)
                            : "s90", "s91", "s92", "s93");
                    } else {
                        constexpr int g = tz * W + ty, k0 = 7 * (g % 9);
                        const f32x2 pa7 = P[ty][NP - 2], pa8 = P[ty][NP - 1], pb7 = P[ty + 1][NP - 2], pb8 = P[ty + 1][NP - 1];
                        const unsigned w0 = wb[g / 9 < NWR ? g / 9 : 0];
                        asm volatile(
                            "v_readlane_b32 s90, %[w0], %[k0]\n\t"
                            "v_readlane_b32 s92, %[w0], %[k1]\n\t"
                            "s_nop 0\n\t"
                            MI_SS_FMA4("s[90:91]", 0, 2)
                            "v_readlane_b32 s90, %[w0], %[k2]\n\t"
                            MI_SS_FMA4("s[92:93]", 1, 3)
                            "v_readlane_b32 s92, %[w0], %[k3]\n\t"
                            MI_SS_FMA4("s[90:91]", 2, 4)
                            "v_readlane_b32 s90, %[w0], %[k4]\n\t"
                            MI_SS_FMA4("s[92:93]", 3, 5)
                            "v_readlane_b32 s92, %[w0], %[k5]\n\t"
                            MI_SS_FMA4("s[90:91]", 4, 6)
                            "v_readlane_b32 s90, %[w0], %[k6]\n\t"
                            MI_SS_FMA4("s[92:93]", 5, 7)
                            MI_SS_FMA4("s[90:91]", 6, 8)
                            : [a0] "+v"(a0), [a1] "+v"(a1), [b0] "+v"(b0), [b1] "+v"(b1)
                            : [pa0] "v"(pa0), [pa1] "v"(pa1), [pa2] "v"(pa2), [pa3] "v"(pa3), [pa4] "v"(pa4), [pa5] "v"(pa5), [pa6] "v"(pa6),
                              [pa7] "v"(pa7), [pa8] "v"(pa8),
                              [pb0] "v"(pb0), [pb1] "v"(pb1), [pb2] "v"(pb2), [pb3] "v"(pb3), [pb4] "v"(pb4), [pb5] "v"(pb5), [pb6] "v"(pb6),
                              [pb7] "v"(pb7), [pb8] "v"(pb8),
                              [w0] "v"(w0),
                              [k0] "n"(k0), [k1] "n"(k0 + 1), [k2] "n"(k0 + 2), [k3] "n"(k0 + 3), [k4] "n"(k0 + 4), [k5] "n"(k0 + 5),
                              [k6] "n"(k0 + 6)
                            : "s90", "s91", "s92", "s93");
                    }
#undef MI_SS_FMA4
                });
            });
        } else {
#pragma unroll
            for (int i = 0; i < RW + W - 1; i++) {
                float d[4 + 2 * RX];
                read_row(i, d);
#pragma unroll
                for (int ty = 0; ty < W; ty++) {
                    const int rr = i - ty;
                    if (rr < 0 || rr >= RW) continue;         // compile time
#pragma unroll
                    for (int tz = 0; tz < W; tz++) {
                        const int a = (PH - tz + 2 * W) % W;   // compile time: the set of output plane q - tz
                        if constexpr (F32) {
#pragma unroll
                            for (int tx = 0; tx < W; tx++) {
                                const f32x2 w2 = splat2(p.w[(tz * W + ty) * W + tx]);
                                const f32x2 lo = fma2((f32x2){d[tx], d[tx + 1]}, w2, (f32x2){A[a][rr][0], A[a][rr][1]});
                                const f32x2 hi = fma2((f32x2){d[tx + 2], d[tx + 3]}, w2, (f32x2){A[a][rr][2], A[a][rr][3]});
                                A[a][rr][0] = lo.x; A[a][rr][1] = lo.y; A[a][rr][2] = hi.x; A[a][rr][3] = hi.y;
                            }
                        } else {
#pragma unroll
                            for (int tx = 0; tx < W; tx++) {
                                const double wv = p.w[(tz * W + ty) * W + tx];
#pragma unroll
                                for (int c = 0; c < 4; c++) {
                                    if constexpr (FMA64) A[a][rr][c] = __builtin_fma((double)d[c + tx], wv, A[a][rr][c]);
                                    else A[a][rr][c] += (double)d[c + tx] * wv;
                                }
                            }
                        }
                    }
                }
            }
        }
        // the output plane that has now seen all W input planes
        {
            constexpr int a = (PH + 1) % W;
            const int s = q - (W - 1);
            const bool live = s >= 0 && s < nout;
            const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(
                (void *)(out + (size_t)(zs + (live ? s : 0)) * plane_elems), 0, (int)plane_bytes, 0x00020000);
#pragma unroll
            for (int rr = 0; rr < RW; rr++) {
                u32x4 u;
                if constexpr (LANEW) {
                    u.x = __float_as_uint(B[a][rr][0].x); u.y = __float_as_uint(B[a][rr][0].y);
                    u.z = __float_as_uint(B[a][rr][1].x); u.w = __float_as_uint(B[a][rr][1].y);
                } else {
                    u.x = __float_as_uint((float)A[a][rr][0]); u.y = __float_as_uint((float)A[a][rr][1]);
                    u.z = __float_as_uint((float)A[a][rr][2]); u.w = __float_as_uint((float)A[a][rr][3]);
                }
                if (!RG || tail == 4) {                       // (uniform)
                    __builtin_amdgcn_raw_buffer_store_b128(u, rout, live ? ovoff[rr] : kOOB, 0, 0);
                } else {
                    // the last lane stores `tail` floats in pieces; every lane issues every store, with an out-of-range offset
                    // where it has nothing to write (no divergent control flow around the stores)
                    const unsigned o = live ? ovoff[rr] : kOOB;
                    __builtin_amdgcn_raw_buffer_store_b128(u, rout, last_lane ? kOOB : o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2){u.x, u.y}, rout, (last_lane && (tail & 2)) ? o : kOOB, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32((tail & 2) ? u.z : u.x, rout, (last_lane && (tail & 1) && o != kOOB) ? o + ((tail & 2) ? 8u : 0u) : kOOB, 0, 0);
                }
            }
        }
        if (q + 1 < nsteps) stage(q + 1);                 // the other slot: nobody reads it in this step
        __syncthreads();
    };

    for (int q0 = 0; q0 < nsteps; q0 += W) {
        static_for<W>([&](auto ph) {
            const int q = q0 + decltype(ph)::value;
            if (q < nsteps) step(ph, q);
        });
    }
}

template <int W, typename Acc, int RW, bool FMA64, bool RG = false>
static int launch_scatter(const float *in, float *out, ScatterParams<Acc> &p, hipStream_t s)
{
    if constexpr (!RG) {
        if (p.nx & 3) return launch_scatter<W, Acc, RW, FMA64, true>(in, out, p, s);
    }
    constexpr int TY = kSsNW * RW;
    const size_t lds = (size_t)2 * (TY + W - 1) * kSsPitch * sizeof(float);
    p.nxt = (p.nx + 255) / 256;
    p.nyt = (p.ny + TY - 1) / TY;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds));
    const int64_t slots = (int64_t)device_cus() * per_cu;
    const int64_t tiles = (int64_t)p.nxt * p.nyt;
    double best = 1e300;
    int best_nzc = 1;
    for (int nzc = 1; nzc <= std::min(p.nz, 64); nzc++) {
        const int chunk = (p.nz + nzc - 1) / nzc;
        const int real = (p.nz + chunk - 1) / chunk;
        const double rounds = (double)((tiles * real + slots - 1) / slots);
        const double cost = rounds * (chunk + W - 1 + 2.0);
        if (cost < best) { best = cost; best_nzc = real; }
    }
    p.zc = (p.nz + best_nzc - 1) / best_nzc;
    p.nzc = (p.nz + p.zc - 1) / p.zc;
    const int64_t total = tiles * p.nzc;
    if (total > 0x7fffffff) { set_error("stencil3s: too many tiles"); return MI_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL((stencil3s_kernel<W, Acc, RW, FMA64, RG>), dim3((unsigned)total), dim3(kSsNW * 64), 0, s, in, out, p);
    MI_HIP(hipGetLastError());
    note_kernel("mi::stencil3s_kernel<%d,%s,%d%s> grid=%lld (dense %dx%dx%d correlate, z scattered over register accumulators, %s)",
                W, std::is_same<Acc, float>::value ? "float" : "double", RW, RG ? ",ragged" : "", (long long)total, W, W, W,
                std::is_same<Acc, float>::value ? "v_pk_fma_f32" : FMA64 ? "f64 fma in window order: float32-valued weights, exact products"
                                                                         : "f64 mul + add in window order");
    return MI_OK;
}

static Knob g_scatter_on{1};

template <int W, typename Acc, int RW, bool FMA64 = false>
static int run_scatter(const mi_array *in, const mi_array *out, const double *weights, const int *off, int mode, double cval,
                       hipStream_t s)
{
    ScatterParams<Acc> p;
    memset(&p, 0, sizeof(p));
    p.nx = (int)in->shape[2]; p.ny = (int)in->shape[1]; p.nz = (int)in->shape[0];
    p.oz = off[0]; p.oy = off[1];
    p.mode = mode;
    p.cval = (float)cval;
    for (int k = 0; k < W * W * W; k++) p.w[k] = (Acc)weights[k];
    return launch_scatter<W, Acc, RW, FMA64>((const float *)in->data, (float *)out->data, p, s);
}

// MI_ERR_UNSUPPORTED (nothing launched) outside the envelope: the caller goes on to stencil3_tiled.
int stencil3_scatter(const mi_array *in, const mi_array *out, const double *weights, const int64_t *wshape, const int *origins,
                     int mode, double cval, bool acc_f32, hipStream_t s)
{
#define NOPE(msg) do { set_error("stencil3s: %s", msg); return MI_ERR_UNSUPPORTED; } while (0)
    if (!g_scatter_on) NOPE("switched off (mi_debug_set_stencil_scatter)");
    if (in->dtype != MI_F32 || out->dtype != MI_F32 || in->ndim != 3) NOPE("3-D float32 volumes only");
    const int W = (int)wshape[0];
    if ((W != 3 && W != 5 && W != 7) || wshape[1] != W || wshape[2] != W) NOPE("3 x 3 x 3, 5 x 5 x 5 or 7 x 7 x 7 windows only");
    if (W == 7 && !acc_f32) NOPE("7 x 7 x 7: float accumulation only (the float64 window is bound by the FP64 pipe on either kernel)");
    int off[3];
    for (int d = 0; d < 3; d++) {
        off[d] = W / 2 + origins[d];
        if (off[d] < 0 || off[d] >= W) { set_error("invalid origin"); return MI_ERR_INVALID_ARG; }
    }
    if (off[2] != W / 2) NOPE("origin 0 along x only");
    for (int k = 0; k < W * W * W; k++)
        if (weights[k] == 0.0 || (acc_f32 && (float)weights[k] == 0.0f)) NOPE("a zero weight is skipped by the reference: mask path");
    const int64_t nz = in->shape[0], ny = in->shape[1], nx = in->shape[2];
    if (nx < 8) NOPE("rows of at least 8 samples");
    if (ny * nx * 4 >= ((int64_t)1 << 31) || nz > (1 << 24) || ny > (1 << 24)) NOPE("plane too large");
    if (W > nz || W > ny) NOPE("window longer than the array");
    if (nz * ny * nx < (1 << 16)) NOPE("small volume");
    if (((uintptr_t)in->data & 3) || ((uintptr_t)out->data & 3)) NOPE("needs 4-byte aligned data");
    if (mode == MI_MODE_CONSTANT && (double)(float)cval != cval && !std::isnan(cval)) NOPE("cval is not a float32 value");
#undef NOPE
    bool f32w = true;                                      // every weight a float32 value: exact products (see FMA64)
    for (int k = 0; k < W * W * W; k++) f32w = f32w && (double)(float)weights[k] == weights[k];
    if (W == 3) {
        if (acc_f32) return run_scatter<3, float, 2>(in, out, weights, off, mode, cval, s);
        return f32w ? run_scatter<3, double, 2, true>(in, out, weights, off, mode, cval, s)
                    : run_scatter<3, double, 2>(in, out, weights, off, mode, cval, s);
    }
    if (W == 7) return run_scatter<7, float, 2>(in, out, weights, off, mode, cval, s);
    if (acc_f32) return run_scatter<5, float, 2>(in, out, weights, off, mode, cval, s);
    if (f32w && nz * ny * nx >= ((int64_t)1 << 25)) return run_scatter<5, double, 2, true>(in, out, weights, off, mode, cval, s);   // 256^3: the ring kernel is 10 % faster
    // 250 f64 mul + add per voxel: bound by the FP64 pipe either way, and the LDS-ring kernel is as fast (1.66 vs 1.68 ms
    // on 512^3, faster on 256^3)
    set_error("stencil3s: 5 x 5 x 5 with float64 weights and float64 accumulation stays on the LDS-ring kernel");
    return MI_ERR_UNSUPPORTED;
}

}  // namespace mi

extern "C" int mi_debug_set_stencil_scatter(int on) { mi::g_scatter_on = on; return MI_OK; }
