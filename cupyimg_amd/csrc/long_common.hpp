// long_common.hpp -- what the LDS-DMA staged tile kernels share (sep3d_long.hip: fused long separable filters;
// minmax3d_f32.hip: fused float32 min / max): tile geometry, the four-DMA statement of a wave, lane-shift helpers.
#pragma once
#include "sep_common.hpp"
#include "stream3d.hpp"

namespace mi {

constexpr int kLongTY = 16;           // output rows per tile = waves per workgroup
constexpr int kLongRowsMax = 32;      // raw rows per plane (TY + 17 - 1)
constexpr int kLongRec = 1024 + 64;   // LDS bytes per raw row: 256 floats + 16 halo floats
constexpr int kLongNB = 4;            // planes in LDS: one being x-filtered, one being y-read, two in flight
constexpr int kLongRawBytes = kLongNB * kLongRowsMax * kLongRec;
constexpr int kLongMaxChunk = 1024;   // planes per z chunk (ztab in LDS)
constexpr int kLongHyBytes = 2 * kLongTY * 64;     // y-filtered halo blocks: [2 planes][16 rows][4 blocks of 16 bytes]


typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// r4: which loads may be NON-TEMPORAL.  scripts/diag/copy_bw.hip: a 512 MiB -> 512 MiB copy with `nt` on the stores only
// moves 6.0 TB/s on this chip, with `nt` on the loads as well 6.45 TB/s.  Marking EVERY staged row nt made the stencil
// kernels slower (D' 197 -> 282 us, E-slab 1.61 -> 1.92 ms, C 407 -> 424 us: rows that a neighbouring workgroup re-reads a
// moment later were no longer found in the L2), so only rows that no other workgroup reads take the hint: of a tile's
// 16 + W - 1 staged rows those with index W - 1 .. 15 (MI_STREAM_NT = 1; 0 = none, the round-3 policy) -- and only
// for volumes that cannot live in the 256 MiB Infinity Cache anyway: 512^3 float32 gains 9-13 % at 3 .. 9 taps
// (uniform 5: 186 -> 169 us), while 256^3 (128 MiB in + out, MALL-resident when filtered repeatedly) LOST 20 % and
// 64 x 1024^2 / 200 x 500 x 760 (0.5-0.6 GiB) 3-9 % (profiles/r4_stream_nt.txt): stream_nt_for().
#ifndef MI_STREAM_NT
#define MI_STREAM_NT 1
#endif
constexpr long long kStreamNtMinBytes = 768ll << 20;       // input + output bytes from which the hint is given: 3 x the MALL
extern Knob g_stream_nt;                                   // test hook: -1 = by size and row width (default), 0 = never, 1 = always
// `x_tiles`: tiles a row is cut into.  A row is also read -- 8 floats at either end -- by the x-neighbouring tiles' halo
// loads; with one or two tiles per row those run next to it on the same XCD and still find the line, with four or eight
// (1024 / 2048-wide rows) they do not, and the hint costs 4-11 % (profiles/r4_stream_nt.txt, second table).
inline int stream_nt_for(long long in_plus_out_bytes, int x_tiles)
{
    const int k = g_stream_nt;
    return k < 0 ? (in_plus_out_bytes >= kStreamNtMinBytes && x_tiles <= 2 ? 1 : 0) : (k != 0);
}

#define MI_DMA_TWO_ROWS(NT_A)                                                                                         \
    asm volatile(                                                                                                    \
        "s_mov_b32 %0, m0\n\t"                                                                                       \
        "s_mov_b32 m0, %6\n\t"                                                                                       \
        "s_nop 0\n\t"                                                                                                \
        "buffer_load_dwordx4 %1, %5, 0 offen" NT_A " lds\n\t"                                                        \
        "s_add_u32 m0, m0, 0x400\n\t"                                                                                \
        "s_mov_b64 exec, 0xffff\n\t"                                                                                 \
        "buffer_load_dword %2, %5, 0 offen lds\n\t"                                                                  \
        "s_mov_b64 exec, -1\n\t"                                                                                     \
        "s_add_u32 m0, m0, %7\n\t"                                                                                   \
        "s_nop 0\n\t"                                                                                                \
        "buffer_load_dwordx4 %3, %5, 0 offen lds\n\t"                                                                \
        "s_add_u32 m0, m0, 0x400\n\t"                                                                                \
        "s_mov_b64 exec, 0xffff\n\t"                                                                                 \
        "buffer_load_dword %4, %5, 0 offen lds\n\t"                                                                  \
        "s_mov_b64 exec, -1\n\t"                                                                                     \
        "s_mov_b32 m0, %0"                                                                                           \
        : "=&s"(keep)                                                                                                \
        : "v"(va), "v"(vha), "v"(vb), "v"(vhb), "s"(rsrc), "s"(rec), "n"(16 * kLongRec - 1024)                        \
        : "memory", "scc")

// nt_a (wave-uniform): row A is read by this workgroup only -> non-temporal
__device__ __forceinline__ void dma_two_rows(u32x4_t rsrc, unsigned va, unsigned vha, unsigned vb, unsigned vhb, unsigned rec,
                                             bool nt_a = false)
{
    unsigned keep;
#if MI_STREAM_NT
    if (nt_a) { MI_DMA_TWO_ROWS(" nt"); return; }
#endif
    MI_DMA_TWO_ROWS("");
}

__device__ __forceinline__ float4 dpp4_shr(const float4 keep, const float4 v)
{
    return make_float4(dpp_from_left(keep.x, v.x), dpp_from_left(keep.y, v.y), dpp_from_left(keep.z, v.z), dpp_from_left(keep.w, v.w));
}
__device__ __forceinline__ float4 dpp4_shl(const float4 keep, const float4 v)
{
    return make_float4(dpp_from_right(keep.x, v.x), dpp_from_right(keep.y, v.y), dpp_from_right(keep.z, v.z), dpp_from_right(keep.w, v.w));
}


}  // namespace mi
