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

// Footprints of up to 128 samples: the samples stay in registers (every index below is static after unrolling) and go through a
// bitonic sorting network padded with the largest key -- P (log2 P)(log2 P + 1) / 4 compare-exchanges, no scratch memory.  The
// reference picks per-size selection networks (_filters_optimal_medians.py); one network per padded size covers every rank.
// r5: the network sorts KEYS (RankKey below): 32-bit integers for everything but float64 -- the integer types as they are, float32
// through the order-preserving map of its bit pattern -- so a compare-exchange is v_min_i32 + v_max_i32.  On float values it was
// v_cmp_lt_f32, two wait states for VCC, two v_cndmask (a < b ? a : b is not v_min_f32 when NaNs may be about).  The keys are
// totally ordered: -0 sorts below +0, NaNs with the sign bit clear sort above +inf (where numpy.sort puts them), those with it set
// below -inf.  (The 3 x 3 x 3 median of a volume has a kernel of its own that shares its sorting between windows: median3d.hip.)
template <typename T>
struct RankKey {                       // the integer types up to 32 bits
    using K = std::conditional_t<std::is_same<T, uint32_t>::value, unsigned, int>;
    static __device__ __forceinline__ K pad() { return std::is_same<T, uint32_t>::value ? (K)0xffffffffu : (K)0x7fffffff; }
    static __device__ __forceinline__ K raw(T v) { return (K)v; }
    static __device__ __forceinline__ K finish(K r) { return r; }
    static __device__ __forceinline__ K key(T v) { return (K)v; }
    static __device__ __forceinline__ T value(K k) { return (T)k; }
};
template <>
struct RankKey<float> {
    using K = int;
    static __device__ __forceinline__ K pad() { return 0x7fffffff; }
    // raw(): the sample as it is loaded (no arithmetic: nothing in the loop that issues the loads waits for memory);
    // finish(): raw -> key, applied to all samples at once afterwards.  finish(pad()) == pad().
    static __device__ __forceinline__ K raw(float v) { return __float_as_int(v); }
    static __device__ __forceinline__ K finish(K b) { return b ^ ((b >> 31) & 0x7fffffff); }
    static __device__ __forceinline__ K key(float v) { return finish(raw(v)); }
    static __device__ __forceinline__ float value(K k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }
};
template <>
struct RankKey<double> {               // no 64-bit integer min / max: compare and select on the values
    using K = double;
    static __device__ __forceinline__ K pad() { return (double)INFINITY; }
    static __device__ __forceinline__ K raw(double v) { return v; }
    static __device__ __forceinline__ K finish(K r) { return r; }
    static __device__ __forceinline__ K key(double v) { return v; }
    static __device__ __forceinline__ double value(K k) { return k; }
};

// V: the type the caller hands the fill value over in (float for float32 and the 8- / 16-bit integers, double otherwise: exact)
// N > 0 / RANK >= 0: tap count and rank known at compile time (the medians of 5 x 5 and 3 x 3 x 3 windows): the +inf
// padding folds away and every compare-exchange that cannot reach output RANK is dead code.
template <typename T, typename V, int P, int N = 0, int RANK = -1>
__global__ void __launch_bounds__(256)
rank3_sorted_kernel(const T *__restrict__ in, T *__restrict__ out, Geom3 g, Taps3 tt, int mode, V cval, int rank)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const LdsTaps lt = stage_taps(tt, smem);
    // (r5, measured and dropped: a workgroup walking 16 planes so that the tap table is staged once -- 10 % on uint8, but the loop
    // keeps the geometry and the tap descriptors live and spilled 73 scalar registers; one plane per workgroup)
    const Vox3 v = locate3(g);
    if (!v.valid) return;
    const __amdgpu_buffer_rsrc_t rin =
        __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)((unsigned)g.nz * g.ny * g.nx * sizeof(T)), 0x00020000);
    using RK = RankKey<T>;
    using K = typename RK::K;
    K vals[P];
    const K craw = RK::raw((T)cval);
    const int n = N > 0 ? N : tt.ntaps;
    if (v.interior) {
        const unsigned base = (unsigned)v.lin * (unsigned)sizeof(T);
        rank_static_for<P>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            vals[t] = t < n ? RK::raw(buf_load<T>(rin, base + (unsigned)(lt.lin[t] * (int)sizeof(T)))) : RK::pad();
        });
    } else {
        rank_static_for<P>([&](auto TT) {
            constexpr int t = decltype(TT)::value;
            if (t < n) {
                const int pos = tap_pos3(g, v, lt, t, mode);
                vals[t] = pos < 0 ? craw : RK::raw(buf_load<T>(rin, (unsigned)pos * (unsigned)sizeof(T)));
            } else {
                vals[t] = RK::pad();
            }
        });
    }
    // raw samples -> keys, all at once: with the conversion inside the loops above -- one basic block per tap when the tap count is
    // a run-time number -- every load was followed by a wait for memory (r5: the float32 kernels ran 1.5 x the uint8 ones)
    rank_static_for<P>([&](auto TT) { vals[decltype(TT)::value] = RK::finish(vals[decltype(TT)::value]); });
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
                    const K a = vals[i], b = vals[l];
                    const K lo = a < b ? a : b, hi = a < b ? b : a;
                    if constexpr ((i & k) == 0) { vals[i] = lo; vals[l] = hi; }
                    else { vals[i] = hi; vals[l] = lo; }
                }
            });
        });
    });
    K res = vals[0];
    if constexpr (RANK >= 0) {
        res = vals[RANK];
    } else {
        rank_static_for<P - 1>([&](auto TT) {
            constexpr int t = decltype(TT)::value + 1;
            res = rank == t ? vals[t] : res;
        });
    }
    out[v.lin] = RK::value(res);
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
