// Issue cost of the VALU operations the rasterizer's blend loop is made of, on gfx950: cycles per wave64 instruction per SIMD
// with the SIMD saturated (8 waves per SIMD, independent accumulators).  hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b0 = 1.0001f, b1 = 0.9999f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) asm volatile(REP8("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        if (OP == 1) asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        if (OP == 2) asm volatile(REP8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        if (OP == 3) asm volatile(REP8("v_cmp_le_f32 vcc, %4, %0\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_le_f32 vcc, %4, %2\n v_cndmask_b32 %3, %3, %5, vcc\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
        if (OP == 4) asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                                  : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));
        if (OP == 5) asm volatile(REP8("v_min_f32 %0, %0, %4\n v_min_f32 %1, %1, %4\n v_rcp_f32 %2, %2\n v_min_f32 %3, %3, %4\n")
                                  : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        if (OP == 6) asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %5, %5\n v_pk_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %5, %5\n")
                                  : "+v"(p0), "+v"(a1), "+v"(p2), "+v"(a3) : "v"(q), "v"(b0));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p1.y + p2.x + p3.y;
}
template <int OP>
static void run(const char* name, float* d, int waves_per_simd) {
    const int iters = 4000, blocks = 256 * waves_per_simd;      // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double instr_per_simd = (double)iters * 32.0 * waves_per_simd;         // if the blocks spread evenly over 256 CUs
    printf("%-44s %d waves/SIMD: %.3f ms, %.2f cycles per wave-instruction per SIMD at %.0f MHz\n", name, waves_per_simd, ms,
           ms * 1e-3 * clk_khz * 1e3 / instr_per_simd, clk_khz * 1e-3);
}
int main() {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 4, 8}) {
        run<0>("v_fma_f32", d, w);
        run<1>("v_pk_fma_f32", d, w);
        run<4>("v_pk_add_f32 / v_pk_mul_f32", d, w);
        run<2>("v_exp_f32", d, w);
        run<5>("3 v_min_f32 + 1 v_rcp_f32", d, w);
        run<3>("v_cmp_le_f32 + v_cndmask_b32 pairs", d, w);
        run<6>("v_pk_fma_f32 / v_fma_f32 alternating", d, w);
    }
    return 0;
}
