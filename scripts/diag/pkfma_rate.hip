// pkfma_rate.hip -- issue rate of v_pk_fma_f32 in the operand patterns the long separable kernel uses (gfx950).
// Each variant runs ITER x 32 instructions per wave with 4 waves per SIMD on every CU and reports cycles per wave
// instruction per SIMD (s_memtime = shader clock) and the shader clock against s_memrealtime (100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -o pkfma_rate scripts/diag/pkfma_rate.hip ; run: ./pkfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

// variant 0: z-pass form.  32 independent accumulators (16 pairs in banks 0/1, 16 in banks 2/3 as allocated by the
// asm below), src0 = SGPR pair, src1 = one shared VGPR pair.
template <int V>
__global__ void __launch_bounds__(1024) rate_kernel(float *out, unsigned long long *clk, int iters, float w0, float w1)
{
    f32x2 x = {(float)threadIdx.x, 1.0f}, y = {2.0f, (float)threadIdx.x};
    f32x2 w = {w0, w1};
    unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    if constexpr (V == 0) {
        // explicit registers: accumulators v[32:95] as 32 pairs; x in v[2:3] (banks 2,3), y in v[4:5] (banks 0,1)
        asm volatile(
            "v_mov_b32 v2, %1\n v_mov_b32 v3, %2\n v_mov_b32 v4, %3\n v_mov_b32 v5, %4\n"
            "s_mov_b32 s20, %5\n"
            "1:\n"
            // even accumulator pairs (v32,v33 -> banks 0,1) with x (banks 2,3): no conflict by the 4-bank model
            "v_pk_fma_f32 v[32:33], %6, v[2:3], v[32:33]\n v_pk_fma_f32 v[36:37], %6, v[2:3], v[36:37]\n"
            "v_pk_fma_f32 v[40:41], %6, v[2:3], v[40:41]\n v_pk_fma_f32 v[44:45], %6, v[2:3], v[44:45]\n"
            "v_pk_fma_f32 v[48:49], %6, v[2:3], v[48:49]\n v_pk_fma_f32 v[52:53], %6, v[2:3], v[52:53]\n"
            "v_pk_fma_f32 v[56:57], %6, v[2:3], v[56:57]\n v_pk_fma_f32 v[60:61], %6, v[2:3], v[60:61]\n"
            "v_pk_fma_f32 v[64:65], %6, v[2:3], v[64:65]\n v_pk_fma_f32 v[68:69], %6, v[2:3], v[68:69]\n"
            "v_pk_fma_f32 v[72:73], %6, v[2:3], v[72:73]\n v_pk_fma_f32 v[76:77], %6, v[2:3], v[76:77]\n"
            "v_pk_fma_f32 v[80:81], %6, v[2:3], v[80:81]\n v_pk_fma_f32 v[84:85], %6, v[2:3], v[84:85]\n"
            "v_pk_fma_f32 v[88:89], %6, v[2:3], v[88:89]\n v_pk_fma_f32 v[92:93], %6, v[2:3], v[92:93]\n"
            "v_pk_fma_f32 v[32:33], %6, v[2:3], v[32:33]\n v_pk_fma_f32 v[36:37], %6, v[2:3], v[36:37]\n"
            "v_pk_fma_f32 v[40:41], %6, v[2:3], v[40:41]\n v_pk_fma_f32 v[44:45], %6, v[2:3], v[44:45]\n"
            "v_pk_fma_f32 v[48:49], %6, v[2:3], v[48:49]\n v_pk_fma_f32 v[52:53], %6, v[2:3], v[52:53]\n"
            "v_pk_fma_f32 v[56:57], %6, v[2:3], v[56:57]\n v_pk_fma_f32 v[60:61], %6, v[2:3], v[60:61]\n"
            "v_pk_fma_f32 v[64:65], %6, v[2:3], v[64:65]\n v_pk_fma_f32 v[68:69], %6, v[2:3], v[68:69]\n"
            "v_pk_fma_f32 v[72:73], %6, v[2:3], v[72:73]\n v_pk_fma_f32 v[76:77], %6, v[2:3], v[76:77]\n"
            "v_pk_fma_f32 v[80:81], %6, v[2:3], v[80:81]\n v_pk_fma_f32 v[84:85], %6, v[2:3], v[84:85]\n"
            "v_pk_fma_f32 v[88:89], %6, v[2:3], v[88:89]\n v_pk_fma_f32 v[92:93], %6, v[2:3], v[92:93]\n"
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %0, v32, v93\n"
            : "=v"(x.x)
            : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "s"(iters), "s"(w)
            : "v2", "v3", "v4", "v5", "s20", "scc", "v32", "v33", "v36", "v37", "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53",
              "v56", "v57", "v60", "v61", "v64", "v65", "v68", "v69", "v72", "v73", "v76", "v77", "v80", "v81", "v84", "v85", "v88",
              "v89", "v92", "v93");
    } else if constexpr (V == 1) {
        // same, but the shared source in the SAME banks as the accumulators (v[4:5] banks 0,1 with v[32:33] banks 0,1)
        asm volatile(
            "v_mov_b32 v2, %1\n v_mov_b32 v3, %2\n v_mov_b32 v4, %3\n v_mov_b32 v5, %4\n"
            "s_mov_b32 s20, %5\n"
            "1:\n"
            "v_pk_fma_f32 v[32:33], %6, v[4:5], v[32:33]\n v_pk_fma_f32 v[36:37], %6, v[4:5], v[36:37]\n"
            "v_pk_fma_f32 v[40:41], %6, v[4:5], v[40:41]\n v_pk_fma_f32 v[44:45], %6, v[4:5], v[44:45]\n"
            "v_pk_fma_f32 v[48:49], %6, v[4:5], v[48:49]\n v_pk_fma_f32 v[52:53], %6, v[4:5], v[52:53]\n"
            "v_pk_fma_f32 v[56:57], %6, v[4:5], v[56:57]\n v_pk_fma_f32 v[60:61], %6, v[4:5], v[60:61]\n"
            "v_pk_fma_f32 v[64:65], %6, v[4:5], v[64:65]\n v_pk_fma_f32 v[68:69], %6, v[4:5], v[68:69]\n"
            "v_pk_fma_f32 v[72:73], %6, v[4:5], v[72:73]\n v_pk_fma_f32 v[76:77], %6, v[4:5], v[76:77]\n"
            "v_pk_fma_f32 v[80:81], %6, v[4:5], v[80:81]\n v_pk_fma_f32 v[84:85], %6, v[4:5], v[84:85]\n"
            "v_pk_fma_f32 v[88:89], %6, v[4:5], v[88:89]\n v_pk_fma_f32 v[92:93], %6, v[4:5], v[92:93]\n"
            "v_pk_fma_f32 v[32:33], %6, v[4:5], v[32:33]\n v_pk_fma_f32 v[36:37], %6, v[4:5], v[36:37]\n"
            "v_pk_fma_f32 v[40:41], %6, v[4:5], v[40:41]\n v_pk_fma_f32 v[44:45], %6, v[4:5], v[44:45]\n"
            "v_pk_fma_f32 v[48:49], %6, v[4:5], v[48:49]\n v_pk_fma_f32 v[52:53], %6, v[4:5], v[52:53]\n"
            "v_pk_fma_f32 v[56:57], %6, v[4:5], v[56:57]\n v_pk_fma_f32 v[60:61], %6, v[4:5], v[60:61]\n"
            "v_pk_fma_f32 v[64:65], %6, v[4:5], v[64:65]\n v_pk_fma_f32 v[68:69], %6, v[4:5], v[68:69]\n"
            "v_pk_fma_f32 v[72:73], %6, v[4:5], v[72:73]\n v_pk_fma_f32 v[76:77], %6, v[4:5], v[76:77]\n"
            "v_pk_fma_f32 v[80:81], %6, v[4:5], v[80:81]\n v_pk_fma_f32 v[84:85], %6, v[4:5], v[84:85]\n"
            "v_pk_fma_f32 v[88:89], %6, v[4:5], v[88:89]\n v_pk_fma_f32 v[92:93], %6, v[4:5], v[92:93]\n"
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %0, v32, v93\n"
            : "=v"(x.x)
            : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "s"(iters), "s"(w)
            : "v2", "v3", "v4", "v5", "s20", "scc", "v32", "v33", "v36", "v37", "v40", "v41", "v44", "v45", "v48", "v49", "v52", "v53",
              "v56", "v57", "v60", "v61", "v64", "v65", "v68", "v69", "v72", "v73", "v76", "v77", "v80", "v81", "v84", "v85", "v88",
              "v89", "v92", "v93");
    } else if constexpr (V == 2) {
        // plain v_fma_f32, 32 independent accumulators, SGPR weight
        asm volatile(
            "v_mov_b32 v2, %1\n s_mov_b32 s20, %5\n"
            "1:\n"
            REP8("v_fma_f32 v32, %6, v2, v32\n v_fma_f32 v37, %6, v2, v37\n v_fma_f32 v42, %6, v2, v42\n v_fma_f32 v47, %6, v2, v47\n")
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %0, v32, v47\n"
            : "=v"(x.x)
            : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "s"(iters), "s"(w0)
            : "v2", "s20", "scc", "v32", "v37", "v42", "v47");
    } else if constexpr (V == 3) {
        // x-pass form: two dependent chains, data operand with op_sel broadcast, SGPR pair swapped
        asm volatile(
            "v_mov_b32 v2, %1\n v_mov_b32 v3, %2\n v_mov_b32 v4, %3\n v_mov_b32 v5, %4\n"
            "s_mov_b32 s20, %5\n"
            "1:\n"
            REP8("v_pk_fma_f32 v[32:33], v[2:3], %6, v[32:33] op_sel:[0,1,0] op_sel_hi:[0,0,1]\n"
                 "v_pk_fma_f32 v[34:35], v[2:3], %6, v[34:35] op_sel:[0,1,0] op_sel_hi:[0,0,1]\n"
                 "v_pk_fma_f32 v[32:33], v[4:5], %6, v[32:33] op_sel:[1,0,0]\n"
                 "v_pk_fma_f32 v[34:35], v[4:5], %6, v[34:35] op_sel:[1,0,0]\n")
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %0, v32, v35\n"
            : "=v"(x.x)
            : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "s"(iters), "s"(w)
            : "v2", "v3", "v4", "v5", "s20", "scc", "v32", "v33", "v34", "v35");
    } else if constexpr (V == 4) {
        // z-pass form with all-VGPR operands (weight pair in v[6:7])
        asm volatile(
            "v_mov_b32 v2, %1\n v_mov_b32 v3, %2\n v_mov_b32 v6, %3\n v_mov_b32 v7, %4\n"
            "s_mov_b32 s20, %5\n"
            "1:\n"
            REP8("v_pk_fma_f32 v[32:33], v[6:7], v[2:3], v[32:33]\n v_pk_fma_f32 v[36:37], v[6:7], v[2:3], v[36:37]\n"
                 "v_pk_fma_f32 v[40:41], v[6:7], v[2:3], v[40:41]\n v_pk_fma_f32 v[44:45], v[6:7], v[2:3], v[44:45]\n")
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %0, v32, v45\n"
            : "=v"(x.x)
            : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "s"(iters), "s"(w)
            : "v2", "v3", "v6", "v7", "s20", "scc", "v32", "v33", "v36", "v37", "v40", "v41", "v44", "v45");
    } else if constexpr (V == 5) {
        // DPP wave shifts (the x pass's lane exchange)
        asm volatile(
            "v_mov_b32 v2, %1\n s_mov_b32 s20, %5\n"
            "1:\n"
            REP8("v_mov_b32_dpp v32, v2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v33, v2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                 "v_mov_b32_dpp v34, v2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v35, v2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n")
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %0, v32, v35\n"
            : "=v"(x.x)
            : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "s"(iters), "s"(w)
            : "v2", "s20", "scc", "v32", "v33", "v34", "v35");
    } else {
        // row-local DPP (row_shr) for comparison
        asm volatile(
            "v_mov_b32 v2, %1\n s_mov_b32 s20, %5\n"
            "1:\n"
            REP8("v_mov_b32_dpp v32, v2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v33, v2 row_shl:1 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b32_dpp v34, v2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v35, v2 row_shl:1 row_mask:0xf bank_mask:0xf\n")
            "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"
            "v_add_f32 %0, v32, v35\n"
            : "=v"(x.x)
            : "v"(x.x), "v"(x.y), "v"(y.x), "v"(y.y), "s"(iters), "s"(w)
            : "v2", "s20", "scc", "v32", "v33", "v34", "v35");
    }
    unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x.x;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = t1 - t0;
        clk[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int V>
static void run(const char *name, int waves_per_simd)
{
    const int blocks = 256, threads = 256 * waves_per_simd, iters = 20000;
    float *out;
    unsigned long long *clk;
    hipMalloc(&out, sizeof(float) * blocks * 1024);
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(rate_kernel<V>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters, 1.0001f, 0.9999f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    double cyc = 0, wall = 0;
    for (int b = 0; b < blocks; b++) { cyc += (double)h[2 * b]; wall += (double)h[2 * b + 1]; }
    cyc /= blocks; wall /= blocks;
    const double instr_per_simd = (double)iters * 32.0 * waves_per_simd;
    // s_memtime counts at a fixed 100 MHz on gfx9xx parts where readcyclecounter lowers to it; report both readings
    printf("%-46s waves/SIMD %d: %8.3f ms, memtime ticks %.3e, realtime ticks %.3e (100 MHz -> %.3f ms), "
           "ns per wave instruction per SIMD %.3f\n", name, waves_per_simd, ms, cyc, wall, wall / 1e5, ms * 1e6 / instr_per_simd);
    hipFree(out);
    hipFree(clk);
}

int main()
{
    for (int w : {1, 4}) {
        run<0>("pk_fma z form, acc/src in different banks", w);
        run<1>("pk_fma z form, acc/src in the same banks", w);
        run<4>("pk_fma z form, all VGPR operands", w);
        run<3>("pk_fma x form (2 chains, op_sel)", w);
        run<2>("v_fma_f32 (plain), 4 accumulators x8", w);
        run<5>("v_mov_b32_dpp wave_shr/shl", w);
        run<6>("v_mov_b32_dpp row_shr/shl", w);
    }
    return 0;
}
