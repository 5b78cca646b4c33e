// copy_bw.hip -- what this MI355X sustains for a 512 MiB -> 512 MiB float copy under different access patterns (round 4:
// is the in-tree float4 copy kernel, 5.8 TB/s, really the ceiling the headline kernel should be priced against? the
// hardware guide quotes ~6.3 TB/s as achievable).  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/diag/bin/copy_bw scripts/diag/copy_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// A: grid-stride float4 copy (the in-tree comparator)
__global__ void __launch_bounds__(256) k_stride(const f4 *__restrict__ in, f4 *__restrict__ out, long n4)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
// B: U loads in flight per thread before the stores, grid-stride over blocks of U * blockDim
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_unroll(const f4 *__restrict__ in, f4 *__restrict__ out, long n4)
{
    const long span = (long)blockDim.x * U;
    for (long b = (long)blockIdx.x * span; b < n4; b += (long)gridDim.x * span) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long i = b + (long)u * blockDim.x + threadIdx.x;
            if (i < n4) v[u] = NT ? __builtin_nontemporal_load(in + i) : in[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long i = b + (long)u * blockDim.x + threadIdx.x;
            if (i < n4) { if (NT) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
        }
    }
}
// C: every workgroup owns ONE contiguous chunk (n4 / gridDim) and streams through it, U loads in flight
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_chunk(const f4 *__restrict__ in, f4 *__restrict__ out, long n4)
{
    const long per = (n4 + gridDim.x - 1) / gridDim.x;
    const long b0 = (long)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
    const long span = (long)blockDim.x * U;
    for (long b = b0; b < b1; b += span) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long i = b + (long)u * blockDim.x + threadIdx.x;
            if (i < b1) v[u] = NT ? __builtin_nontemporal_load(in + i) : in[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long i = b + (long)u * blockDim.x + threadIdx.x;
            if (i < b1) { if (NT) __builtin_nontemporal_store(v[u], out + i); else out[i] = v[u]; }
        }
    }
}
// D: read only (sum kept alive) / write only
__global__ void __launch_bounds__(256) k_read(const f4 *__restrict__ in, f4 *__restrict__ out, long n4)
{
    f4 s = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) s += in[i];
    if (s.x == 12345.678f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write(const f4 *__restrict__ in, f4 *__restrict__ out, long n4)
{
    const f4 v = {1, 2, 3, 4};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) out[i] = v;
}

// E: buffer stores with the cache-policy bits spelt out (gfx940+: aux bit 0 = sc0, bit 1 = nt, bit 4 = sc1); loads likewise
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int AUXS, int AUXL, bool READ>
__global__ void __launch_bounds__(256) k_buf(const f4 *__restrict__ in, f4 *__restrict__ out, long n4)
{
    // 2 GiB windows do not matter here: 512 MiB arrays
    const __amdgpu_buffer_rsrc_t ri = __builtin_amdgcn_make_buffer_rsrc((void *)in, 0, (int)(n4 * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, (int)(n4 * 16), 0x00020000);
    const long span = (long)blockDim.x * 4;
    for (long b = (long)blockIdx.x * span; b < n4; b += (long)gridDim.x * span) {
        u4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned off = (unsigned)(b + (long)u * blockDim.x + threadIdx.x) * 16u;
            if (READ) v[u] = __builtin_amdgcn_raw_buffer_load_b128(ri, off, 0, AUXL);
            else v[u] = (u4){1u, 2u, 3u, off};
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const unsigned off = (unsigned)(b + (long)u * blockDim.x + threadIdx.x) * 16u;
            __builtin_amdgcn_raw_buffer_store_b128(v[u], ro, off, 0, AUXS);
        }
    }
}

template <typename K>
static double run(const char *name, K kern, int blocks, const f4 *in, f4 *out, long n4, double bytes)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 250; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, n4);     // ~45 ms: settled clocks
    CHECK(hipDeviceSynchronize());
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 100; i++) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, n4);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms / 100 < best) best = ms / 100;
    }
    printf("%-44s blocks %6d  %7.1f us  %7.1f GB/s  (%.3f of 8 TB/s)\n", name, blocks, best * 1e3, bytes / (best * 1e-3) / 1e9, bytes / (best * 1e-3) / 8e12);
    fflush(stdout);
    return best;
}

int main()
{
    const long n = 512L * 512 * 512, n4 = n / 4;
    f4 *in, *out;
    CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&out, n * 4));
    CHECK(hipMemset(in, 1, n * 4));
    const double rw = 2.0 * n * 4, one = 1.0 * n * 4;
    for (int blocks : {1024, 65536}) run("A grid-stride float4", k_stride, blocks, in, out, n4, rw);
    for (int blocks : {4096, 8192, 16384}) {
        run("B unroll 4", k_unroll<4, false>, blocks, in, out, n4, rw);
        run("B unroll 8", k_unroll<8, false>, blocks, in, out, n4, rw);
        run("B unroll 4 nontemporal", k_unroll<4, true>, blocks, in, out, n4, rw);
        run("B unroll 8 nontemporal", k_unroll<8, true>, blocks, in, out, n4, rw);
    }
    for (int blocks : {8192}) {
        run("C contiguous chunk per workgroup, unroll 4", k_chunk<4, false>, blocks, in, out, n4, rw);
        run("C contiguous chunk, unroll 4, nontemporal", k_chunk<4, true>, blocks, in, out, n4, rw);
        run("C contiguous chunk, unroll 8, nontemporal", k_chunk<8, true>, blocks, in, out, n4, rw);
    }
    for (int blocks : {1024, 4096}) {
        run("D read only", k_read, blocks, in, out, n4, one);
        run("D write only", k_write, blocks, in, out, n4, one);
    }
    for (int blocks : {2048, 4096, 8192}) {
        run("E write only, buffer store aux 0", k_buf<0, 0, false>, blocks, in, out, n4, one);
        run("E write only, buffer store aux 2 (nt)", k_buf<2, 0, false>, blocks, in, out, n4, one);
        run("E write only, buffer store aux 1 (sc0)", k_buf<1, 0, false>, blocks, in, out, n4, one);
        run("E write only, buffer store aux 16 (sc1)", k_buf<16, 0, false>, blocks, in, out, n4, one);
        run("E write only, buffer store aux 18 (sc1 nt)", k_buf<18, 0, false>, blocks, in, out, n4, one);
        run("E write only, buffer store aux 19 (sc0 sc1 nt)", k_buf<19, 0, false>, blocks, in, out, n4, one);
        run("E copy, load aux 0 store aux 2", k_buf<2, 0, true>, blocks, in, out, n4, rw);
        run("E copy, load nt store nt", k_buf<2, 2, true>, blocks, in, out, n4, rw);
        run("E copy, load nt store sc1 nt", k_buf<18, 2, true>, blocks, in, out, n4, rw);
        run("E copy, load sc1 nt store sc0 sc1 nt", k_buf<19, 18, true>, blocks, in, out, n4, rw);
        run("E copy, load aux 0 store aux 0", k_buf<0, 0, true>, blocks, in, out, n4, rw);
    }
    CHECK(hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0));
    return 0;
}
