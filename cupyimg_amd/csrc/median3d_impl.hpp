// median3d_impl.hpp (median3d*.hip) -- the 3 x 3 x 3 median of a volume (median_filter(size=3), the commonest rank filter on MRI volumes) as a
// z-streaming kernel that SHARES its sorting work between neighbouring voxels (r5).
//
// Reference path replaced: median_filter -> rank_filter -> one thread per voxel sorting its own 27 samples
// (cupyimg/scipy/ndimage/filters.py:1612-1650 median_filter, 1712-1850 _rank_filter / _get_rank_kernel: shell sort or the selection networks of _filters_optimal_medians.py per voxel).  The register sorting network of rank_sorted.hpp
// does the same per voxel: 27 gathers and ~270 min / max per voxel plus their addressing -- 700 vector instructions per wave
// and voxel, instruction bound (profiles/r5_rank_filters.txt).
//
// Here the 27 samples of a window are sorted ALONG THE THREE AXES IN TURN, and each partial result serves every window it is
// part of:
//   z: a thread owns one (y, x) column and walks z with the last three planes in registers; sorted = (min3, med3, max3): 3
//      instructions (v_min3 / v_med3 / v_max3 on 32-bit keys: RankKey of rank_sorted.hpp), used by 9 windows;
//   x: the z-sorted triples of the lanes left and right (2 x 3 wave shifts) -> for every z rank the sorted triple along x: 9
//      instructions, used by 3 windows (rows y - 1, y, y + 1): exchanged through LDS, one barrier per plane;
//   y: of the 27 positions (i, j, k) of the cube sorted along all three axes, (i+1)(j+1)(k+1) - 1 samples are known to lie
//      below and (3-i)(3-j)(3-k) - 1 above: only the 19 positions with both counts <= 13 can hold the median, each ONE
//      instruction (the k-th of a triple along y), and the median of the window is the median of those 19;
//   the median of 19 values in that partial order: a comparator network found by search (scripts/gen_median27_network.py:
//      Batcher's network pruned by the 980 monotone 0/1 labelings of the 3 x 3 x 3 poset), median27_net.hpp.
// A workgroup is 16 waves = 16 rows of 64 columns and produces 14 x 62 outputs per plane; boundary modes are index maps of the
// thread's own (y, x) and of the plane index, settled before the loop (`constant`: the fill value's key).
#pragma once
#include "rank_sorted.hpp"
#include "sep_common.hpp"

namespace mi {

struct Med27Params {
    int nx, ny, nz;
    int mx, my, mz;
    int nxt, nyt, nzc, zc;
    double cval;
};

constexpr int kM27Rows = 16, kM27OutRows = kM27Rows - 2, kM27OutCols = 62;

// key operations of the kernel: 32-bit integer keys (RankKey of rank_sorted.hpp: every dtype but float64; U: compared unsigned)
template <bool U>
struct KeyOps32 {
    using K = int;
    static __device__ __forceinline__ int min3(int a, int b, int c)
    {
        int r;
        if constexpr (U) asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        else asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    }
    static __device__ __forceinline__ int med3(int a, int b, int c)
    {
        int r;
        if constexpr (U) asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        else asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    }
    static __device__ __forceinline__ int max3(int a, int b, int c)
    {
        int r;
        if constexpr (U) asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        else asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    }
    using C = std::conditional_t<U, unsigned, int>;
    static __device__ __forceinline__ int mn(int a, int b) { return (C)a < (C)b ? a : b; }
    static __device__ __forceinline__ int mx(int a, int b) { return (C)a > (C)b ? a : b; }
    static __device__ __forceinline__ int left(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false); }
    static __device__ __forceinline__ int right(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
};
// float64: the values themselves with v_min_f64 / v_max_f64 (IEEE minNum / maxNum: a NaN in the window is passed over -- there
// are no 64-bit integer min / max instructions for an order-preserving key as float32 has); the three order statistics of a
// triple share their first two instructions (common subexpressions): 6 for all three
struct KeyOps64 {
    using K = double;
    static __device__ __forceinline__ double mn(double a, double b) { return __builtin_fmin(a, b); }
    static __device__ __forceinline__ double mx(double a, double b) { return __builtin_fmax(a, b); }
    static __device__ __forceinline__ double min3(double a, double b, double c) { return mn(mn(a, b), c); }
    static __device__ __forceinline__ double max3(double a, double b, double c) { return mx(mx(a, b), c); }
    static __device__ __forceinline__ double med3(double a, double b, double c) { return mx(mn(a, b), mn(mx(a, b), c)); }
    static __device__ __forceinline__ double shift(double v, bool to_right)
    {
        const long long b = __builtin_bit_cast(long long, v);
        int lo = (int)b, hi = (int)(b >> 32);
        if (to_right) {
            lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xf, 0xf, false);
            hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xf, 0xf, false);
        } else {
            lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);
            hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
        }
        return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    }
    static __device__ __forceinline__ double left(double v) { return shift(v, false); }
    static __device__ __forceinline__ double right(double v) { return shift(v, true); }
};
template <typename T>
using KeyOpsFor = std::conditional_t<std::is_same<T, double>::value, KeyOps64, KeyOps32<std::is_same<T, uint32_t>::value>>;

}  // namespace mi
#include "median27_net.hpp"
namespace mi {

template <typename T, int R>
__global__ void __launch_bounds__(kM27Rows * 64)
median27_stream_kernel(const T *__restrict__ in, T *__restrict__ out, const Med27Params p)
{
    using KO = KeyOpsFor<T>;
    using K = typename KO::K;
    using Net = Rank27Net<KO, R>;
    using RK = RankKey<T>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    K *lds = reinterpret_cast<K *>(smem);                      // [2][rows][9][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    int b = blockIdx.x;
    const int per_chunk = p.nxt * p.nyt;
    const int zci = b / per_chunk;
    const int rem = b - zci * per_chunk;
    const int yt = rem / p.nxt, xt = rem - yt * p.nxt;
    const int nx = p.nx, ny = p.ny, nz = p.nz;
    const int zs = zci * p.zc, ze = min(zs + p.zc, nz);

    const int x = xt * kM27OutCols - 1 + lane, y = yt * kM27OutRows - 1 + wave;
    // columns further out than one beyond the array are nobody's neighbour: clamped, so that the index maps stay cheap
    const int xsrc = bmap<int>(min(x, nx), nx, p.mx), ysrc = bmap<int>(min(y, ny), ny, p.my);
    const bool have = xsrc >= 0 && ysrc >= 0;
    const unsigned off = have ? (unsigned)(ysrc * nx + xsrc) * (unsigned)sizeof(T) : kOOB;
    const bool is_out = lane >= 1 && lane <= kM27OutCols && wave >= 1 && wave <= kM27OutRows && x < nx && y < ny;
    const K ckey = (K)RK::key((T)p.cval);
    const size_t plane_elems = (size_t)ny * (size_t)nx;
    const unsigned plane_bytes = (unsigned)(plane_elems * sizeof(T));

    auto fetch = [&](int s) -> K {
        // the sample of this thread's column in the plane of step s, as a key
        int zsrc = zs - 1 + s;
        if ((unsigned)zsrc >= (unsigned)nz) zsrc = bmap<int>(zsrc, nz, p.mz);         // the first and the last plane only: the index map's divisions are scalar code every wave would run every step
        const int zz = __builtin_amdgcn_readfirstlane(max(zsrc, 0));
        const __amdgpu_buffer_rsrc_t rin =
            __builtin_amdgcn_make_buffer_rsrc((void *)(in + (size_t)zz * plane_elems), 0, (int)plane_bytes, 0x00020000);
        const T v = buf_load<T>(rin, zsrc >= 0 ? off : kOOB);
        return (have && zsrc >= 0) ? (K)RK::key(v) : ckey;
    };

    const int nsteps = ze - zs + 2;
    K k0 = 0, k1 = fetch(0), k2 = fetch(1);
    K nxt = nsteps > 2 ? fetch(2) : (K)0;
    K *wr = lds + (wave * 9) * 64 + lane;
    for (int s = 2; s < nsteps; s++) {
        k0 = k1; k1 = k2; k2 = nxt;
        if (s + 1 < nsteps) nxt = fetch(s + 1);
        // z
        const K L = KO::min3(k0, k1, k2), M = KO::med3(k0, k1, k2), H = KO::max3(k0, k1, k2);
        // x: Q[i][j] = the j-th along x of the i-th along z
        const K Ll = KO::left(L), Lr = KO::right(L), Ml = KO::left(M), Mr = KO::right(M), Hl = KO::left(H), Hr = KO::right(H);
        K *w = wr + (s & 1) * (kM27Rows * 9 * 64);
        const K q[9] = {KO::min3(Ll, L, Lr), KO::med3(Ll, L, Lr), KO::max3(Ll, L, Lr),
                        KO::min3(Ml, M, Mr), KO::med3(Ml, M, Mr), KO::max3(Ml, M, Mr),
                        KO::min3(Hl, H, Hr), KO::med3(Hl, H, Hr), KO::max3(Hl, H, Hr)};
#pragma unroll
        for (int t = 0; t < 9; t++) w[t * 64] = q[t];
        __syncthreads();
        if (wave >= 1 && wave <= kM27OutRows) {
            K c[Net::NC];
            // the candidates (i, j, k) of rank R: i = rank along z, j along x, k along y; the rows above and below: -+ 9 * 64 ints
            Net::candidates(w - 9 * 64, q, w + 9 * 64, c);
            const K med = Net::select(c);
            if (is_out) out[(size_t)(zs + s - 2) * plane_elems + (size_t)(y * nx + x)] = RK::value((typename RK::K)med);
        }
    }
}

template <typename T, int R>
static int launch_median27(const T *in, T *out, int nz, int ny, int nx, int mx, int my, int mz, double cval, hipStream_t s)
{
    Med27Params p;
    p.nx = nx; p.ny = ny; p.nz = nz; p.mx = mx; p.my = my; p.mz = mz; p.cval = cval;
    p.nxt = (nx + kM27OutCols - 1) / kM27OutCols;
    p.nyt = (ny + kM27OutRows - 1) / kM27OutRows;
    // z chunks: at least two workgroups per CU in the launch, chunks of at least 8 planes (each re-reads two)
    const int64_t tiles = (int64_t)p.nxt * p.nyt;
    int nzc = (int)std::min<int64_t>(std::max<int64_t>(1, (2 * (int64_t)device_cus() + tiles - 1) / tiles), std::max(1, nz / 8));
    p.zc = (nz + nzc - 1) / nzc;
    p.nzc = (nz + p.zc - 1) / p.zc;
    const size_t lds = (size_t)2 * kM27Rows * 9 * 64 * sizeof(typename KeyOpsFor<T>::K);
    static PerDeviceOnce attr_done;
    if (!attr_done) {
        MI_HIP(hipFuncSetAttribute((const void *)median27_stream_kernel<T, R>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const int64_t total = tiles * p.nzc;
    note_kernel("mi::median27_stream_kernel<%d> grid=%d (rank %d of the 3 x 3 x 3 window: sorted along z, x, y in turn, shared between windows)", R,
                (int)total, R);
    hipLaunchKernelGGL((median27_stream_kernel<T, R>), dim3((unsigned)total), dim3(kM27Rows * 64), lds, s, in, out, p);
    MI_HIP(hipGetLastError());
    return MI_OK;
}

int median27_enabled();          // median3d.hip: mi_debug_set_median27

// rank `rank` of the full 3 x 3 x 3 window of a C-contiguous volume (cval: already the value the dtype holds), or
// MI_ERR_UNSUPPORTED (minmax.hip: mi_rank_filter tries it first).  ALL: every rank 1 .. 25 is built for T, else the median only.
template <typename T, bool ALL>
int run_rank27_impl(const T *in, T *out, int64_t nz, int64_t ny, int64_t nx, int mode, double cval, int rank, hipStream_t s)
{
    if (!median27_enabled()) return MI_ERR_UNSUPPORTED;
    if (nx < 8 || ny < 2 || nz < 2 || ny * nx * 8 >= ((int64_t)1 << 31) || nz * ny * nx < (1 << 12) || nz > (1 << 24)) return MI_ERR_UNSUPPORTED;
    if ((const void *)in == (const void *)out) return MI_ERR_UNSUPPORTED;
    const int z = (int)nz, y = (int)ny, x = (int)nx;
    if (rank == 13) return launch_median27<T, 13>(in, out, z, y, x, mode, mode, mode, cval, s);
    if constexpr (ALL) {
        switch (rank) {
#define MI_R27(R) case R: return launch_median27<T, R>(in, out, z, y, x, mode, mode, mode, cval, s);
        MI_R27(1) MI_R27(2) MI_R27(3) MI_R27(4) MI_R27(5) MI_R27(6) MI_R27(7) MI_R27(8) MI_R27(9) MI_R27(10) MI_R27(11) MI_R27(12)
        MI_R27(14) MI_R27(15) MI_R27(16) MI_R27(17) MI_R27(18) MI_R27(19) MI_R27(20) MI_R27(21) MI_R27(22) MI_R27(23) MI_R27(24) MI_R27(25)
#undef MI_R27
        }
    }
    return MI_ERR_UNSUPPORTED;
}

template <typename T>
int run_rank27(const T *in, T *out, int64_t nz, int64_t ny, int64_t nx, int mode, double cval, int rank, hipStream_t s);
#define MI_RANK27_INST(T, ALL)                                                                                                  \
    template <> int run_rank27<T>(const T *in, T *out, int64_t nz, int64_t ny, int64_t nx, int mode, double cval, int rank, hipStream_t s) \
    {                                                                                                                           \
        return run_rank27_impl<T, ALL>(in, out, nz, ny, nx, mode, cval, rank, s);                                               \
    }

}  // namespace mi
