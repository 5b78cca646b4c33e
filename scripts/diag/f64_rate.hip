// f64_rate.hip -- issue rate of the float64 VALU instructions the order-1 interpolation kernels use for their
// coordinate split (gfx950).  4 waves per SIMD on every CU, ITER x 32 independent instructions per wave; prints ns per
// wave instruction per SIMD (hipEvents), to be read against v_fma_f32 in the same table.
// Build: hipcc --offload-arch=gfx950 -O3 -o f64_rate scripts/diag/f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x

#define KERNEL(NAME, BODY, CLOBBERS...)                                                        \
    __global__ void __launch_bounds__(1024) NAME(float *out, int iters, double a, double b)    \
    {                                                                                          \
        double x = a + threadIdx.x, y = b;                                                     \
        float r;                                                                               \
        asm volatile("v_mov_b32 v2, %1\n v_mov_b32 v3, %2\n v_mov_b32 v4, %3\n v_mov_b32 v5, %4\n" \
                     "s_mov_b32 s20, %5\n"                                                     \
                     "1:\n" BODY                                                               \
                     "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"       \
                     "v_mov_b32 %0, v32\n"                                                     \
                     : "=v"(r)                                                                 \
                     : "v"((unsigned)__double_as_longlong(x)), "v"((unsigned)(__double_as_longlong(x) >> 32)), \
                       "v"((unsigned)__double_as_longlong(y)), "v"((unsigned)(__double_as_longlong(y) >> 32)), "s"(iters) \
                     : "v2", "v3", "v4", "v5", "s20", "scc", "vcc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", \
                       "s22", "s23", "s24", "s25");                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                        \
    }

KERNEL(k_fma32, REP8("v_fma_f32 v32, v2, v4, v32\n v_fma_f32 v34, v2, v4, v34\n v_fma_f32 v36, v2, v4, v36\n v_fma_f32 v38, v2, v4, v38\n"))
KERNEL(k_add64, REP8("v_add_f64 v[32:33], v[2:3], v[4:5]\n v_add_f64 v[34:35], v[2:3], v[4:5]\n v_add_f64 v[36:37], v[2:3], v[4:5]\n v_add_f64 v[38:39], v[2:3], v[4:5]\n"))
KERNEL(k_fma64, REP8("v_fma_f64 v[32:33], v[2:3], v[4:5], v[32:33]\n v_fma_f64 v[34:35], v[2:3], v[4:5], v[34:35]\n v_fma_f64 v[36:37], v[2:3], v[4:5], v[36:37]\n v_fma_f64 v[38:39], v[2:3], v[4:5], v[38:39]\n"))
KERNEL(k_mul64, REP8("v_mul_f64 v[32:33], v[2:3], v[4:5]\n v_mul_f64 v[34:35], v[2:3], v[4:5]\n v_mul_f64 v[36:37], v[2:3], v[4:5]\n v_mul_f64 v[38:39], v[2:3], v[4:5]\n"))
KERNEL(k_fract64, REP8("v_fract_f64 v[32:33], v[2:3]\n v_fract_f64 v[34:35], v[2:3]\n v_fract_f64 v[36:37], v[2:3]\n v_fract_f64 v[38:39], v[2:3]\n"))
KERNEL(k_cvti32, REP8("v_cvt_i32_f64 v32, v[2:3]\n v_cvt_i32_f64 v34, v[2:3]\n v_cvt_i32_f64 v36, v[2:3]\n v_cvt_i32_f64 v38, v[2:3]\n"))
KERNEL(k_cvtf32, REP8("v_cvt_f32_f64 v32, v[2:3]\n v_cvt_f32_f64 v34, v[2:3]\n v_cvt_f32_f64 v36, v[2:3]\n v_cvt_f32_f64 v38, v[2:3]\n"))
KERNEL(k_cvtf64i, REP8("v_cvt_f64_i32 v[32:33], v2\n v_cvt_f64_i32 v[34:35], v2\n v_cvt_f64_i32 v[36:37], v2\n v_cvt_f64_i32 v[38:39], v2\n"))
KERNEL(k_cmp64, REP8("v_cmp_gt_f64 vcc, v[2:3], v[4:5]\n v_cmp_gt_f64 s[22:23], v[2:3], v[4:5]\n v_cmp_neq_f64 vcc, v[2:3], v[4:5]\n v_cmp_neq_f64 s[24:25], v[2:3], v[4:5]\n"))
KERNEL(k_floor64, REP8("v_floor_f64 v[32:33], v[2:3]\n v_floor_f64 v[34:35], v[2:3]\n v_floor_f64 v[36:37], v[2:3]\n v_floor_f64 v[38:39], v[2:3]\n"))
KERNEL(k_cmp32, REP8("v_cmp_gt_f32 vcc, v2, v4\n v_cmp_gt_f32 s[22:23], v2, v4\n v_cmp_neq_f32 vcc, v2, v4\n v_cmp_neq_f32 s[24:25], v2, v4\n"))
KERNEL(k_cndmask, REP8("v_cndmask_b32 v32, v2, v4, vcc\n v_cndmask_b32 v34, v2, v4, vcc\n v_cndmask_b32 v36, v2, v4, vcc\n v_cndmask_b32 v38, v2, v4, vcc\n"))
KERNEL(k_mad64, REP8("v_mad_u64_u32 v[32:33], s[22:23], v2, v4, v[2:3]\n v_mad_u64_u32 v[34:35], s[22:23], v2, v4, v[2:3]\n v_mad_u64_u32 v[36:37], s[22:23], v2, v4, v[2:3]\n v_mad_u64_u32 v[38:39], s[22:23], v2, v4, v[2:3]\n"))

KERNEL(k_cnd_sgpr, REP8("v_cndmask_b32_e64 v32, v2, v4, s[22:23]\n v_cndmask_b32_e64 v34, v2, v4, s[22:23]\n v_cndmask_b32_e64 v36, v2, v4, s[22:23]\n v_cndmask_b32_e64 v38, v2, v4, s[22:23]\n"))
KERNEL(k_cnd_mix, REP8("v_cndmask_b32 v32, v2, v4, vcc\n v_fma_f32 v34, v2, v4, v34\n v_fma_f32 v36, v2, v4, v36\n v_fma_f32 v38, v2, v4, v38\n"))
KERNEL(k_cnd_init, "s_mov_b64 vcc, 0x0f0f0f0f\n s_mov_b64 s[22:23], 0x33333333\n" REP8("v_cndmask_b32 v32, v2, v4, vcc\n v_cndmask_b32 v34, v2, v4, vcc\n v_cndmask_b32_e64 v36, v2, v4, s[22:23]\n v_cndmask_b32_e64 v38, v2, v4, s[22:23]\n"))
KERNEL(k_cmp_cnd, REP8("v_cmp_gt_f32 vcc, v2, v4\n v_cndmask_b32 v32, v2, v4, vcc\n v_cmp_gt_f32 vcc, v4, v2\n v_cndmask_b32 v34, v2, v4, vcc\n"))
KERNEL(k_mov, REP8("v_mov_b32 v32, v2\n v_mov_b32 v34, v4\n v_mov_b32 v36, v2\n v_mov_b32 v38, v4\n"))
KERNEL(k_bfi, REP8("v_bfi_b32 v32, v5, v2, v4\n v_bfi_b32 v34, v5, v2, v4\n v_bfi_b32 v36, v5, v2, v4\n v_bfi_b32 v38, v5, v2, v4\n"))
KERNEL(k_mov_exec, "s_mov_b64 s[22:23], exec\n s_mov_b32 s24, 1\n s_mov_b32 s25, 0x80000000\n" REP8("s_mov_b64 exec, s[24:25]\n v_mov_b32 v32, v2\n v_mov_b32 v34, v4\n v_mov_b32 v36, v2\n v_mov_b32 v38, v4\n s_mov_b64 exec, s[22:23]\n"))
KERNEL(k_pkfma, REP8("v_pk_fma_f32 v[32:33], v[2:3], v[4:5], v[32:33]\n v_pk_fma_f32 v[34:35], v[2:3], v[4:5], v[34:35]\n v_pk_fma_f32 v[36:37], v[2:3], v[4:5], v[36:37]\n v_pk_fma_f32 v[38:39], v[2:3], v[4:5], v[38:39]\n"))

typedef void (*kern_t)(float *, int, double, double);
static void run(const char *name, kern_t k)
{
    const int blocks = 256, threads = 1024, iters = 10000;
    float *out;
    (void)hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 100.37, 1.5);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%-18s %8.3f ms  %.3f ns per wave instruction per SIMD\n", name, ms, ms * 1e6 / ((double)iters * 32.0 * 4.0));
    (void)hipFree(out);
}

int main()
{
    run("v_fma_f32", k_fma32);
    run("v_add_f64", k_add64);
    run("v_fma_f64", k_fma64);
    run("v_mul_f64", k_mul64);
    run("v_fract_f64", k_fract64);
    run("v_floor_f64", k_floor64);
    run("v_cvt_i32_f64", k_cvti32);
    run("v_cvt_f32_f64", k_cvtf32);
    run("v_cvt_f64_i32", k_cvtf64i);
    run("v_cmp_*_f64", k_cmp64);
    run("v_cmp_*_f32", k_cmp32);
    run("v_cndmask_b32", k_cndmask);
    run("v_mad_u64_u32", k_mad64);
    run("cndmask sgpr mask", k_cnd_sgpr);
    run("cndmask+3 fma", k_cnd_mix);
    run("cndmask mask set", k_cnd_init);
    run("cmp+cndmask pairs", k_cmp_cnd);
    run("v_mov_b32", k_mov);
    run("v_bfi_b32", k_bfi);
    run("4 v_mov in exec win", k_mov_exec);
    run("v_pk_fma_f32", k_pkfma);
    return 0;
}
