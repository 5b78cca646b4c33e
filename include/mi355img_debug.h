/* mi355img_debug.h -- test and tuning entry points of libmi355img.so.
 *
 * NOT part of the drop-in boundary (include/mi355img.h): nothing in the product path calls these; they exist so
 * that tests can force a particular kernel (to compare two device paths bit for bit), tuning sweeps can walk tile
 * shapes, and bench.py can name the kernel it timed.  The setters write process-wide knobs held as relaxed atomics
 * (mi::Knob, csrc/common.hpp): flipping one while other threads dispatch is not a data race, but it does change
 * which kernel those threads' NEXT calls take -- keep them out of production code.  All return MI_OK.
 */
#ifndef MI355IMG_DEBUG_H
#define MI355IMG_DEBUG_H

#include "mi355img.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fused separable float32 kernels (csrc/separable3d.hip, sep3d_long.hip, stream3d.hip) */
int mi_debug_set_sep3d_cfg(int cfg);          /* lean kernel tile shape variant (0 = cost model) */
int mi_debug_set_sep3d_zchunks(int n);        /* number of z chunks (0 = cost model) */
int mi_debug_set_sep3d_dbg(int flags);        /* ablations: 1 no x/z math, 2 no stores, 4 no loads, 8 no y math */
int mi_debug_set_sep3d_kernel(int k);         /* 1 = always the general (ws) kernel */
int mi_debug_set_sep3d_zrev(int on);          /* 0 = every z chunk streams upwards */
int mi_debug_set_sep3d_image2d(int on);       /* 0 = images take the tiled volume kernel */
int mi_debug_set_sep3d_long(int k);           /* 0 auto (9..17 taps, and 3..7 taps on large volumes with full tiles), 1 never the long kernel, 2 long kernel for all of 3..17 taps */
int mi_debug_set_sep3d_box(int k);            /* 0 auto, 1 never the running-sum box kernel */
int mi_debug_set_sep3d_ragged(int k);         /* r5: 1 (default) the 3 / 5 / 7-tap kernel takes rows that are not a multiple of 4 floats, 0 refuse them (r6: the switch also covers the 9 .. 17-tap LDS-DMA kernel's ragged build) */
int mi_debug_set_median27(int k);             /* r5: 1 (default) 3 x 3 x 3 medians of volumes on the shared-sort streaming kernel, 0 the per-voxel network */
int mi_debug_set_long_const0(int k);          /* r5: 1 (default) constant mode with cval == 0 on the r3 long kernel, 0 the r2 kernel */
int mi_debug_set_long_zchunks(int n);
int mi_debug_set_long_same(int on);
int mi_debug_set_long_cfg(int k);             /* long kernel, builds with -DMI_LONG_TUNE only: tuning variant of the 17-tap kernel (0 = product) */
int mi_debug_set_long_rows(int k);            /* long kernel: 0 auto (r3 kernel), 1 the r2 instruction stream (kept for 9 / 13 / 17 taps as the comparator), 2 (MI_LONG_TUNE builds, 17 taps) r2 stream with two rows per wave, 4 the r4 kernel with the y pass on the matrix cores (9 / 13 / 17 taps) */
int mi_debug_set_long_dbg(int flags);         /* long kernel ablations: 1 y reads one row, 2 no x pass, 4 no z scatter, 8 no DMA, 16 no stores, 32 no halo table */
int mi_debug_set_stream_fused_max(int taps);
int mi_debug_set_xcd_swizzle(int k);
int mi_debug_set_stream_wpb(int n);
int mi_debug_set_stream_slice(int n);
int mi_debug_set_stream_min_chunk(int n);
int mi_debug_set_minmax_f32_fused(int on);
int mi_debug_stream_pass(const float *in, float *out, int nz, int ny, int nx, int axis, const float *wav, int wa,
                         int oa, int ma, const float *wxv, int wx, int mx, float cval, mi_stream stream);
/* plain float4 copy kernel (grid-stride, `blocks` workgroups of 256): the practical HBM ceiling for a byte count */
int mi_debug_copy_f32(const float *in, float *out, int64_t n, int blocks, mi_stream stream);
int mi_debug_copy_f32_nt(const float *in, float *out, int64_t n, int blocks, mi_stream stream);   /* r4: 4 loads in flight, non-temporal loads and stores */
/* name of the last kernel the calling thread's separable-filter call dispatched ("" if none), NUL terminated */
int mi_debug_last_kernel(char *buf, size_t n);

/* counter-based synthetic float32 data (csrc/synth.hip; CPU twin: oracle/synth.py): value i = f(first_index + i, seed) */
int mi_debug_fill_synthetic_f32(float *out, int64_t n, uint64_t first_index, uint64_t seed, mi_stream stream);

/* pool test hook (csrc/runtime.hip): allocate `nbytes` for work on `stream`, report the block, free it again */
int mi_debug_pool_probe(size_t nbytes, mi_stream stream, void **block);

/* ---- other kernel families */
int mi_debug_generation(void);                /* advances with every mi_debug_set_* call: part of the keys of host-side refusal caches */
int mi_debug_set_stencil_scatter(int on);     /* csrc/stencil3s.hip: 0 = dense 3^3 / 5^3 windows stay on stencil3_kernel */
int mi_debug_set_binary_tiled(int on);        /* csrc/binary.hip: 0 = generic binary erosion kernel */
int mi_debug_set_bitfill(int on);                     /* 0: runs until stable iterate the global operator (no block-wise fill) */
int mi_debug_set_u8_ragged(int on);           /* 0: uint8 cubic min / max on rows that are not a multiple of 16 bytes goes back to the extended-rows route (mm3u8_ragged_kernel off) */
int mi_debug_set_s16_ragged(int on);          /* the same switch for the uint16 / int16 kernel (mm3s16_ragged_kernel) */
int mi_debug_set_bitmorph_ragged(int on);             /* 0: rows that are not a multiple of 16 bytes keep the extended-rows / generic routes */
int mi_debug_set_bitmorph_2d(int on);                 /* 0: 2-D images keep the byte kernel (binary3d.hip) */
int mi_debug_set_bitmorph_table(int on);              /* 1: the run-time structure table even for the built-in 3 x 3 x 3 structures */
int mi_debug_set_bitmorph(int on, int ty, int nzc);   /* csrc/bitmorph3d.hip: 0 = byte kernels; ty / nzc > 0 override the tile planner */
int mi_debug_set_stencil(int on);             /* csrc/correlate_nd.hip: 0 = generic n-D correlate */
int mi_debug_set_minmax_tiled(int on);
int mi_debug_set_rank_sorted(int on);
int mi_debug_set_rank_median(int on);
int mi_debug_set_u8_fused(int k);
int mi_debug_set_interp_generic(int on);
int mi_debug_set_affine_gz(int k);            /* LDS-staged affine kernel: workgroups along z (0 = one per tile; fewer = each walks several tiles of its column) */
int mi_debug_set_affine_rowblend(int k);      /* row-blend affine kernel (x axis untouched): 0 off, 1 on (default) */
int mi_debug_set_map_zstream(int k);          /* z-streaming map_coordinates kernel: 0 off, 1 on (default), 2 on with every step on the L1 gathers, 3 on with the exact box reduction (first r4 kernel) */
int mi_debug_set_map_zchunks(int k);          /* its z chunks (0 = planner) */
int mi_debug_set_map_zvariant(int k);         /* its instance: 10 x voxels per thread + planes of coordinates in flight (81, 82, 41; 0 = default) */
int mi_debug_set_affine_zstream(int k);       /* z-streaming affine kernel (axis 0 decoupled): 0 off, 1 auto, 32 / 64 tile height */
int mi_debug_set_affine_zchunks(int k);       /* its z chunks (0 = planner) */
int mi_debug_set_affine_box_kib(int k);       /* LDS-staged affine kernel: box budget per workgroup in KiB (0 = default 64; 36 = the r3 / r4 rule) */
int mi_debug_set_affine_dbg(int k);           /* LDS-staged affine kernel ablations: 1 no box DMA, 2 no interpolation, 4 no stores (timing only) */
int mi_debug_set_interp_c1(int k);            /* csrc/interp_fast.hip: 0 = round-2 order-1 constant-mode kernels, 1 = r3 (default), 2 = r3, narrow stores */
int mi_debug_set_spline_gain_first(int on);
int mi_debug_set_spline_threads(int n);
int mi_debug_set_spline_chunk(int n);
int mi_debug_set_spline_rows(int k);
int mi_debug_set_cubic_separable(int on);
int mi_debug_set_cubic_diag(int on);
int mi_debug_set_spline_rows_lds(int on);     /* spline prefilter along the contiguous axis: 0 = the tiled kernel (lines through memory twice per pole), 1 = whole lines in LDS when they fit (default), 4 .. 64 = that with so many lines per wave */
int mi_debug_set_spline_fast(int k);          /* r5 prefilter kernels (csrc/spline_fast.hip: one memory sweep per axis): 0 = never, 1 = volumes with >= 16384 lines (default), 2 = any line count */
int mi_debug_set_cubic_rowblend(int on);      /* order-3 affine on float32 coefficients, x axis to itself: 0 = the gather kernel, 1 = the row-blend kernel (default) */
int mi_debug_set_cubic_zstream(int on);       /* order-3 affine on float32 coefficients, axis 0 decoupled: 0 = the gather kernel, 1 = the z-streaming kernel (default), bits on top of 1: 2 = every wave on its gather path, 4 = no limit on the angle, 8 = grid-wrap / grid-constant too */
int mi_debug_set_cubic_box(int on);           /* order-3 affine, all three axes coupled: 1 = taps out of an LDS-staged box when it fits (r5, default), 0 = the gather kernel; bits 2 / 4 / 8: timing ablations */
int mi_debug_set_resample_fast(int on);       /* diagonal order-3 transforms on volumes (zoom, shift): 1 = r5 passes (x from LDS-staged spans, z with the plane window in registers), 0 = r3 passes (bit-identical) */
int mi_debug_set_cubic_zfactor(int on);       /* order-3 affine, stream axis decoupled: 1 = cubic3_zfactor_kernel (r5: in-plane values once per input plane; float32 rounding away from the gather kernel), 0 = cubic3_zstream_kernel (r4b: bit-identical to the gather kernel) */
int mi_debug_set_stream_nt(int k);             /* non-temporal staging of rows no other workgroup reads (sep3d_long3 <= 9 taps, mm3f32_long, mm3u8_split <= 5): -1 by volume size (default), 0 never, 1 always */
int mi_debug_set_pipe_normal_priority(int on); /* slab pipeline: comm stream at normal instead of high priority (read by mi_slab_pipe_create) */

#ifdef __cplusplus
}
#endif
#endif
