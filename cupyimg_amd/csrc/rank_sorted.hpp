// rank_sorted.hpp -- the register-resident sorting-network rank kernel (see minmax.hip: mi_rank_filter) and its
// launcher.  The networks are fully unrolled (P = 64: 672 compare-exchanges), which makes each instantiation slow
// to compile (the 24 of them took eleven minutes in one translation unit): the explicit instantiations are spread
// over rank_sorted_*.hip so that they build in parallel; minmax.hip only sees the declaration.
#pragma once
#include "nd_common.hpp"

namespace mi {

template <int N, typename F>
__device__ __forceinline__ void rank_static_for(F &&f)
{
    if constexpr (N > 0) {
        rank_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Footprints of up to 64 samples, values held as float (float32, 8- and 16-bit integers: exact) or as
// double (float64, 32-bit integers): the samples stay in registers (every index below is static after unrolling) and go through a bitonic
// sorting network padded with +inf -- P (log2 P)(log2 P + 1) / 4 compare-exchanges of one v_min + one
// v_max each, no scratch memory.  The reference picks per-size selection networks
// (_filters_optimal_medians.py); one network per padded size covers every rank.
// N > 0 / RANK >= 0: tap count and rank known at compile time (the medians of 5 x 5 and 3 x 3 x 3 windows): the +inf
// padding folds away and every compare-exchange that cannot reach output RANK is dead code.
template <typename T, typename V, int P, int N = 0, int RANK = -1>
__global__ void __launch_bounds__(256)
rank3_sorted_kernel(const T *__restrict__ in, T *__restrict__ out, Geom3 g, Taps3 tt, int mode, V cval, int rank)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsTaps lt = stage_taps(tt, smem);
    const Vox3 v = locate3(g);
    if (!v.valid) return;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)((unsigned)g.nz * g.ny * g.nx * sizeof(T)), 0x00020000);
    V vals[P];
    const int n = N > 0 ? N : tt.ntaps;
    if (v.interior) {
        const unsigned base = (unsigned)v.lin * (unsigned)sizeof(T);
        rank_static_for<P>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            vals[t] = t < n ? (V)buf_load<T>(rin, base + (unsigned)(lt.lin[t] * (int)sizeof(T))) : (V)INFINITY;
        });
    } else {
        rank_static_for<P>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            if (t < n) {
                const int pos = tap_pos3(g, v, lt, t, mode);
                vals[t] = pos < 0 ? cval : (V)buf_load<T>(rin, (unsigned)pos * (unsigned)sizeof(T));
            } else {
                vals[t] = (V)INFINITY;
            }
        });
    }
    // stage s of the network: block size k = 2 << (stage row), distance j; every index is a compile-time constant
    constexpr int LOGP = P == 16 ? 4 : (P == 32 ? 5 : (P == 64 ? 6 : 7));
    rank_static_for<LOGP>([&](auto KK) {
        constexpr int k = 2 << decltype(KK)::value;
        rank_static_for<decltype(KK)::value + 1>([&](auto JJ) {
            constexpr int j = (k >> 1) >> decltype(JJ)::value;
            rank_static_for<P>([&](auto II) {
                constexpr int i = decltype(II)::value;
                constexpr int l = i ^ j;
                if constexpr (l > i) {
                    const V a = vals[i], b = vals[l];
                    const V lo = a < b ? a : b, hi = a < b ? b : a;
                    if constexpr ((i & k) == 0) { vals[i] = lo; vals[l] = hi; }
                    else { vals[i] = hi; vals[l] = lo; }
                }
            });
        });
    });
    V res = vals[0];
    if constexpr (RANK >= 0) {
        res = vals[RANK];
    } else {
        rank_static_for<P - 1>([&](auto TT) {
            constexpr int t = decltype(TT)::value + 1;
            res = rank == t ? vals[t] : res;
        });
    }
    out[v.lin] = (T)res;
}


template <typename T, typename V, int P>
int run_rank_sorted(const T *in, T *out, const Geom3 &g, const Taps3 &tt, int mode, V cval, int rank, hipStream_t s)
{
    hipLaunchKernelGGL((rank3_sorted_kernel<T, V, P>), grid3(g), dim3(64, 4, 1), taps3_lds_bytes(tt), s, in, out, g, tt, mode, cval, rank);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

// the median of N samples (N = 25: 5 x 5, N = 27: 3 x 3 x 3), network pruned at compile time
template <typename T, typename V, int N>
int run_median_sorted(const T *in, T *out, const Geom3 &g, const Taps3 &tt, int mode, V cval, hipStream_t s)
{
    hipLaunchKernelGGL((rank3_sorted_kernel<T, V, 32, N, N / 2>), grid3(g), dim3(64, 4, 1), taps3_lds_bytes(tt), s, in, out, g, tt, mode,
                       cval, N / 2);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

#define MI_MEDIAN_SORTED_INST(T, V, N) \
    template int run_median_sorted<T, V, N>(const T *, T *, const Geom3 &, const Taps3 &, int, V, hipStream_t)

#define MI_RANK_SORTED_INST(T, V, P) \
    template int run_rank_sorted<T, V, P>(const T *, T *, const Geom3 &, const Taps3 &, int, V, int, hipStream_t)

}  // namespace mi
