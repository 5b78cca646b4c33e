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

// The four LDS-DMAs a wave issues per plane, as ONE statement (M0 = wave-uniform LDS destination, saved and
// restored around it): row A (16 bytes per lane, destination rec + 16 * lane), its halo (4 bytes per lane, lanes
// 0..15 only, at rec + 1024), then the same for row B, whose record lies 16 records further.
__device__ __forceinline__ void dma_two_rows(u32x4_t rsrc, unsigned va, unsigned vha, unsigned vb, unsigned vhb, unsigned rec)
{
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %5, 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_mov_b64 exec, 0xffff\n\t"
        "buffer_load_dword %2, %5, 0 offen lds\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_add_u32 m0, m0, %7\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %3, %5, 0 offen lds\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_mov_b64 exec, 0xffff\n\t"
        "buffer_load_dword %4, %5, 0 offen lds\n\t"
        "s_mov_b64 exec, -1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(va), "v"(vha), "v"(vb), "v"(vhb), "s"(rsrc), "s"(rec), "n"(16 * kLongRec - 1024)
        : "memory", "scc");
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
