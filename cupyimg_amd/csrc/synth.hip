// synth.hip -- counter-based synthetic volumes for the benchmarks (bench.py --config E; SURVEY.md section 8d: "generated
// on device ... from a counter-based generator so 32 GiB never has to exist on the host; same generator in the CPU
// restatement").  Value of global linear index i under `seed`:
//     h_k = splitmix64(seed + 4 i + k),  u_k = (h_k >> 42) * 2^-22,  x = ((u_0 + u_1 + u_2 + u_3) - 2) * sqrt(3)
// i.e. an Irwin-Hall(4) sample scaled to unit variance.  Every operation is exact in float32 except the final
// multiplication (one correctly rounded IEEE product), so oracle/synth.py reproduces it bit for bit with NumPy.
// Bench / test utility: declared in include/mi355img_debug.h, not part of the drop-in boundary.
#include "common.hpp"

namespace mi {

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) synth_f32_kernel(float *__restrict__ out, long long n, unsigned long long first,
                                                        unsigned long long seed)
{
    const float kScale = 1.7320508075688772f;     // sqrtf(3)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const unsigned long long c = seed + 4ull * (first + (unsigned long long)i);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) s += (float)(unsigned)(splitmix64(c + k) >> 42) * 0x1p-22f;
        out[i] = (s - 2.0f) * kScale;
    }
}

}  // namespace mi

extern "C" int mi_debug_fill_synthetic_f32(float *out, int64_t n, uint64_t first_index, uint64_t seed, mi_stream stream)
{
    MI_REQUIRE(out || n == 0, MI_ERR_INVALID_ARG, "out is NULL");
    if (n <= 0) return MI_OK;
    const int64_t want = (n + 255) / 256;
    const int blocks = (int)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(mi::synth_f32_kernel, dim3(blocks), dim3(256), 0, mi::resolve_stream(stream), out, (long long)n,
                       (unsigned long long)first_index, (unsigned long long)seed);
    MI_HIP(hipGetLastError());
    return MI_OK;
}
