// Issue cost of the candidate instructions for a LONE wavefront on a SIMD (the state k_rollout_quad runs in): cycles per
// instruction of independent streams and of one dependent chain, for v_fma_f32 / v_fmac_f32_dpp / v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32 / v_mov_b32_dpp / v_sin_f32 / v_rcp_f32.  (experiment: is hand-packed float2 algebra worth building?)
//   hipcc --offload-arch=gfx950 -O2 -o issue_rate issue_rate.hip && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define DPP " quad_perm:[1,2,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
// 8 independent accumulators v[0..7] (pairs for the packed forms), operands a, b
#define KERNEL(name, body)                                                                                     \
    __global__ void name(float *out, long long *ticks, int iters) {                                            \
        float a = out[threadIdx.x], b = out[64 + threadIdx.x];                                                 \
        float2 A = {a, b}, B = {b, a};                                                                         \
        float x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;            \
        float2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = p0, p5 = p1, p6 = p2, p7 = p3;  \
        long long t0 = clock64();                                                                              \
        for (int i = 0; i < iters; i++) {                                                                      \
            asm volatile(body : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), \
                         "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)        \
                         : "v"(a), "v"(b), "v"(A), "v"(B));                                                    \
        }                                                                                                      \
        long long t1 = clock64();                                                                              \
        out[128 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y; \
        if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;                                                     \
    }
// operands: %0-%7 scalars, %8-%15 pairs, %16 a, %17 b, %18 A, %19 B
KERNEL(k_fma_ind, REP8("v_fma_f32 %0, %16, %17, %0\n v_fma_f32 %1, %16, %17, %1\n v_fma_f32 %2, %16, %17, %2\n v_fma_f32 %3, %16, %17, %3\n"
                       "v_fma_f32 %4, %16, %17, %4\n v_fma_f32 %5, %16, %17, %5\n v_fma_f32 %6, %16, %17, %6\n v_fma_f32 %7, %16, %17, %7\n"))
KERNEL(k_fma_dep, REP64("v_fma_f32 %0, %16, %0, %17\n"))
KERNEL(k_fmac_ind, REP8("v_fmac_f32 %0, %16, %17\n v_fmac_f32 %1, %16, %17\n v_fmac_f32 %2, %16, %17\n v_fmac_f32 %3, %16, %17\n"
                        "v_fmac_f32 %4, %16, %17\n v_fmac_f32 %5, %16, %17\n v_fmac_f32 %6, %16, %17\n v_fmac_f32 %7, %16, %17\n"))
KERNEL(k_fmac_dpp_ind, REP8("v_fmac_f32_dpp %0, %16, %17" DPP "\n v_fmac_f32_dpp %1, %16, %17" DPP "\n v_fmac_f32_dpp %2, %16, %17" DPP "\n v_fmac_f32_dpp %3, %16, %17" DPP "\n"
                            "v_fmac_f32_dpp %4, %16, %17" DPP "\n v_fmac_f32_dpp %5, %16, %17" DPP "\n v_fmac_f32_dpp %6, %16, %17" DPP "\n v_fmac_f32_dpp %7, %16, %17" DPP "\n"))
KERNEL(k_fmac_dpp_dep, REP64("v_fmac_f32_dpp %0, %16, %17" DPP "\n"))
KERNEL(k_mov_dpp_ind, REP8("v_mov_b32_dpp %0, %16" DPP "\n v_mov_b32_dpp %1, %16" DPP "\n v_mov_b32_dpp %2, %16" DPP "\n v_mov_b32_dpp %3, %16" DPP "\n"
                           "v_mov_b32_dpp %4, %16" DPP "\n v_mov_b32_dpp %5, %16" DPP "\n v_mov_b32_dpp %6, %16" DPP "\n v_mov_b32_dpp %7, %16" DPP "\n"))
KERNEL(k_pk_fma_ind, REP8("v_pk_fma_f32 %8, %18, %19, %8\n v_pk_fma_f32 %9, %18, %19, %9\n v_pk_fma_f32 %10, %18, %19, %10\n v_pk_fma_f32 %11, %18, %19, %11\n"
                          "v_pk_fma_f32 %12, %18, %19, %12\n v_pk_fma_f32 %13, %18, %19, %13\n v_pk_fma_f32 %14, %18, %19, %14\n v_pk_fma_f32 %15, %18, %19, %15\n"))
KERNEL(k_pk_fma_dep, REP64("v_pk_fma_f32 %8, %18, %8, %19\n"))
KERNEL(k_pk_mul_ind, REP8("v_pk_mul_f32 %8, %18, %19\n v_pk_mul_f32 %9, %18, %19\n v_pk_mul_f32 %10, %18, %19\n v_pk_mul_f32 %11, %18, %19\n"
                          "v_pk_mul_f32 %12, %18, %19\n v_pk_mul_f32 %13, %18, %19\n v_pk_mul_f32 %14, %18, %19\n v_pk_mul_f32 %15, %18, %19\n"))
KERNEL(k_pk_add_ind, REP8("v_pk_add_f32 %8, %18, %8\n v_pk_add_f32 %9, %18, %9\n v_pk_add_f32 %10, %18, %10\n v_pk_add_f32 %11, %18, %11\n"
                          "v_pk_add_f32 %12, %18, %12\n v_pk_add_f32 %13, %18, %13\n v_pk_add_f32 %14, %18, %14\n v_pk_add_f32 %15, %18, %15\n"))
KERNEL(k_pk_add_dep, REP64("v_pk_add_f32 %8, %18, %8\n"))
// mixed stream as in the solve: a packed multiply-add feeding two scalar DPP multiply-adds (cross-format dependency)
KERNEL(k_pk_then_dpp, REP8("v_pk_fma_f32 %8, %18, %19, %8\n v_pk_fma_f32 %9, %18, %19, %9\n v_fmac_f32_dpp %0, %16, %17" DPP "\n v_fmac_f32_dpp %1, %16, %17" DPP "\n"
                           "v_pk_fma_f32 %10, %18, %19, %10\n v_pk_fma_f32 %11, %18, %19, %11\n v_fmac_f32_dpp %2, %16, %17" DPP "\n v_fmac_f32_dpp %3, %16, %17" DPP "\n"))
KERNEL(k_sin_ind, REP8("v_sin_f32 %0, %16\n v_sin_f32 %1, %16\n v_sin_f32 %2, %16\n v_sin_f32 %3, %16\n v_sin_f32 %4, %16\n v_sin_f32 %5, %16\n v_sin_f32 %6, %16\n v_sin_f32 %7, %16\n"))
KERNEL(k_rcp_ind, REP8("v_rcp_f32 %0, %16\n v_rcp_f32 %1, %16\n v_rcp_f32 %2, %16\n v_rcp_f32 %3, %16\n v_rcp_f32 %4, %16\n v_rcp_f32 %5, %16\n v_rcp_f32 %6, %16\n v_rcp_f32 %7, %16\n"))
KERNEL(k_rcp_dep, REP64("v_rcp_f32 %0, %0\n"))
KERNEL(k_nop, REP64("s_nop 0\n"))

template <class F>
void run(const char *name, F kern, int waves_per_simd, float *out, long long *ticks) {
    const int iters = 2000, blocks = 1;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves_per_simd * 4 > 64 ? 64 : 64), 0, 0, out, ticks, iters);   // warm
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, out, ticks, iters);
    hipDeviceSynchronize();
    long long t;
    hipMemcpy(&t, ticks, sizeof t, hipMemcpyDeviceToHost);
    printf("%-16s %7.2f ticks of s_memtime per instruction (64 per iteration, %d iterations)\n", name, (double)t / (64.0 * iters), iters);
}
int main() {
    float *out; long long *ticks;
    hipMalloc(&out, 4096); hipMemset(out, 0, 4096); hipMalloc(&ticks, 4096);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s, shader clock %d MHz (s_memtime counts at a fixed 100 MHz on gfx9: ticks x clock/100 = cycles)\n", p.gcnArchName, p.clockRate / 1000);
#define RUN(k) run(#k, k, 1, out, ticks)
    RUN(k_nop); RUN(k_fma_ind); RUN(k_fma_dep); RUN(k_fmac_ind); RUN(k_fmac_dpp_ind); RUN(k_fmac_dpp_dep); RUN(k_mov_dpp_ind);
    RUN(k_pk_fma_ind); RUN(k_pk_fma_dep); RUN(k_pk_mul_ind); RUN(k_pk_add_ind); RUN(k_pk_add_dep); RUN(k_pk_then_dpp);
    RUN(k_sin_ind); RUN(k_rcp_ind); RUN(k_rcp_dep);
    return 0;
}
