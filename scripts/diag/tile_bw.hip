// tile_bw.hip -- what the MEMORY SIDE of a z-streaming tile kernel can reach on this MI355X, without any arithmetic:
// a workgroup owns a (TY rows x TX floats) tile of an N^3 float volume and walks down a chunk of planes, reading NS
// streams (same tile of NS arrays: NS = 1 the affine / filter kernels, NS = 4 map_coordinates: three coordinate arrays
// + the input) and writing one, one plane prefetched ahead.  Round 4: the z-streaming interpolation kernels (64-float
// rows) all sit at 5.4-5.5 TB/s of traffic, the fused filter kernel (256-float rows) at 6.1-6.3 -- is that the row
// segment, the number of streams, the chunking, or the power-of-two distance between the arrays?
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/diag/bin/tile_bw scripts/diag/tile_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct P { int n, tx4, ty, zc, nzc, ntx, nty; long stride4; int xcd; };

// U float4 per thread and plane and stream (TY * TX / 4 = 256 U)
template <int U, int NS, bool NT>
__global__ void __launch_bounds__(256) k_tile(const f4 *__restrict__ in, f4 *__restrict__ out, const P p)
{
    const int total = p.ntx * p.nty * p.nzc;
    int t = blockIdx.x;
    if (p.xcd && (total & 7) == 0) t = (t & 7) * (total >> 3) + (t >> 3);
    const int txi = t % p.ntx, tyi = (t / p.ntx) % p.nty, zci = t / (p.ntx * p.nty);
    const int zs = zci * p.zc, ze = min(zs + p.zc, p.n);
    const int n4 = p.n / 4;
    long off[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int idx = threadIdx.x + u * 256;
        const int row = idx / p.tx4, c = idx - row * p.tx4;
        off[u] = (long)(tyi * p.ty + row) * n4 + txi * p.tx4 + c;
    }
    const long plane4 = (long)p.n * n4;
    f4 v[2][NS][U];
    auto load = [&](int z, int b) {
#pragma unroll
        for (int s = 0; s < NS; s++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const f4 *a = in + s * p.stride4 + z * plane4 + off[u];
                v[b][s][u] = NT ? __builtin_nontemporal_load(a) : *a;
            }
    };
    auto store = [&](int z, int b) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            f4 r = v[b][0][u];
#pragma unroll
            for (int s = 1; s < NS; s++) r += v[b][s][u];
            f4 *a = out + z * plane4 + off[u];
            if (NT) __builtin_nontemporal_store(r, a); else *a = r;
        }
    };
    load(zs, 0);
    for (int z = zs; z < ze; z += 2) {
        if (z + 1 < ze) load(z + 1, 1);
        store(z, 0);
        if (z + 1 < ze) {
            if (z + 2 < ze) load(z + 2, 0);
            store(z + 1, 1);
        }
    }
}

template <int U, int NS>
static void run(const char *what, const f4 *in, f4 *out, P p, bool nt)
{
    const int total = p.ntx * p.nty * p.nzc;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&]() {
        if (nt) hipLaunchKernelGGL((k_tile<U, NS, true>), dim3(total), dim3(256), 0, 0, in, out, p);
        else hipLaunchKernelGGL((k_tile<U, NS, false>), dim3(total), dim3(256), 0, 0, in, out, p);
    };
    for (int i = 0; i < 30; i++) launch();
    CHECK(hipDeviceSynchronize());
    const int reps = 30;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = (double)(NS + 1) * p.n * (double)p.n * p.n * 4.0;
    printf("%-34s NS %d tile %3d x %3d  chunks %3d (grid %5d) xcd %d nt %d pad %6ld B: %7.1f us  %7.1f GB/s\n", what, NS, p.ty, p.tx4 * 4, p.nzc, total,
           p.xcd, (int)nt, (long)(p.stride4 * 16 - (long)p.n * p.n * p.n * 4), us, bytes / us * 1e-3);
    fflush(stdout);
}

int main(int argc, char **)
{
    const int n = 512;
    const long vol = (long)n * n * n * 4;
    const long pad_max = 1 << 20;
    f4 *in, *out;
    CHECK(hipMalloc(&in, 4 * (vol + pad_max)));
    CHECK(hipMalloc(&out, vol));
    CHECK(hipMemset(in, 0, 4 * (vol + pad_max)));
    CHECK(hipMemset(out, 0, vol));
    auto geo = [&](int tx, int ty, int nzc, long pad, int xcd) {
        P p; p.n = n; p.tx4 = tx / 4; p.ty = ty; p.nzc = nzc; p.zc = (n + nzc - 1) / nzc; p.ntx = n / tx; p.nty = n / ty;
        p.stride4 = (vol + pad) / 16; p.xcd = xcd; return p;
    };
    const bool sweep = argc > 1;
    if (sweep) {
        // which address bits separate the streams' channels?  (4R+1W, 64-float rows, 16 chunks, nontemporal)
        for (long pad : {0L, 128L, 256L, 512L, 1024L, 2048L, 4096L, 8192L, 16384L, 32768L, 65536L, 256L + 4096L, 256L + 16384L, 768L, 1280L, 81920L + 256L})
            run<2, 4>("4R+1W  64-float rows, padded", in, out, geo(64, 32, 16, pad, 1), true);
        for (long pad : {0L, 256L, 16384L}) {
            run<2, 4>("4R+1W  64-float rows, no xcd map", in, out, geo(64, 32, 16, pad, 0), true);
            run<2, 4>("4R+1W 256-float rows, padded", in, out, geo(256, 8, 16, pad, 1), true);
        }
        return 0;
    }
    for (int nt = 0; nt < 2; nt++) {
        // one read stream + one write stream: row segment and tile height at 512 f4 per plane (U = 2) and 1024 (U = 4)
        for (int nzc : {4, 8, 16}) {
            run<2, 1>("1R+1W  64-float rows", in, out, geo(64, 32, nzc, 0, 1), nt);
            run<2, 1>("1R+1W 128-float rows", in, out, geo(128, 16, nzc, 0, 1), nt);
            run<2, 1>("1R+1W 256-float rows", in, out, geo(256, 8, nzc, 0, 1), nt);
            run<2, 1>("1R+1W 512-float rows", in, out, geo(512, 4, nzc, 0, 1), nt);
            run<4, 1>("1R+1W 256-float rows", in, out, geo(256, 16, nzc / 2 ? nzc / 2 : 1, 0, 1), nt);
        }
        run<2, 1>("1R+1W  64-float rows, no xcd map", in, out, geo(64, 32, 8, 0, 0), nt);
        // four read streams + one write stream (map_coordinates), arrays exactly 512 MiB apart / padded
        for (int nzc : {8, 16}) {
            run<2, 4>("4R+1W  64-float rows", in, out, geo(64, 32, nzc, 0, 1), nt);
            run<2, 4>("4R+1W 128-float rows", in, out, geo(128, 16, nzc, 0, 1), nt);
            run<2, 4>("4R+1W 256-float rows", in, out, geo(256, 8, nzc, 0, 1), nt);
            run<2, 4>("4R+1W 512-float rows", in, out, geo(512, 4, nzc, 0, 1), nt);
        }
        for (long pad : {4096L, 65536L, 1L << 20, 81920L + 256}) run<2, 4>("4R+1W  64-float rows, padded", in, out, geo(64, 32, 16, pad, 1), nt);
    }
    return 0;
}
