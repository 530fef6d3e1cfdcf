// Issue-rate microbenchmark for gfx950: wave-instructions per microsecond per SIMD for single
// instruction types (inline asm, 4 independent chains of 4) at 1, 2 and 4 wavefronts per SIMD,
// measured in real time (the shader clock moves with the load, so cycles are not a stable unit).
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_rate issue_rate.hip && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 20000;

#define REP4(S) S S S S
#define OP2(INS) \
    REP4(asm volatile(INS " %0, %0, %4\n" INS " %1, %1, %4\n" INS " %2, %2, %4\n" INS " %3, %3, %4" \
                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));)

template <int KIND>
__global__ __launch_bounds__(256) void k(double *out, double seed)
{
    double b = 1.0000001 + seed, c = 1e-9;
    double x0 = seed + threadIdx.x * 1e-9, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, e = (int)seed;
    for (int it = 0; it < ITERS; ++it) {
        if constexpr (KIND == 0) {
            REP4(asm volatile("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));)
        } else if constexpr (KIND == 1) {
            OP2("v_mul_f64")
        } else if constexpr (KIND == 2) {
            OP2("v_add_f64")
        } else if constexpr (KIND == 3) {
            REP4(asm volatile("v_ldexp_f64 %0, %0, %4\nv_ldexp_f64 %1, %1, %4\nv_ldexp_f64 %2, %2, %4\nv_ldexp_f64 %3, %3, %4"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(e));)
        } else if constexpr (KIND == 4) {
            REP4(asm volatile("v_rndne_f64 %0, %0\nv_rndne_f64 %1, %1\nv_rndne_f64 %2, %2\nv_rndne_f64 %3, %3"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        } else if constexpr (KIND == 5) {
            REP4(asm volatile("v_cvt_i32_f64 %0, %4\nv_cvt_i32_f64 %1, %5\nv_cvt_i32_f64 %2, %6\nv_cvt_i32_f64 %3, %7"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));)
        } else if constexpr (KIND == 6) {
            REP4(asm volatile("v_rcp_f64 %0, %0\nv_rcp_f64 %1, %1\nv_rcp_f64 %2, %2\nv_rcp_f64 %3, %3"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        } else if constexpr (KIND == 7) {
            REP4(asm volatile("v_max_f64 %0, %0, %4\nv_max_f64 %1, %1, %4\nv_max_f64 %2, %2, %4\nv_max_f64 %3, %3, %4"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));)
        } else if constexpr (KIND == 8) {
            REP4(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));)
        } else if constexpr (KIND == 9) {
            REP4(asm volatile("v_max_i32 %0, %0, %4\nv_max_i32 %1, %1, %4\nv_max_i32 %2, %2, %4\nv_max_i32 %3, %3, %4"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(e));)
        } else if constexpr (KIND == 10) {
            REP4(asm volatile("v_mov_b64 %0, %4\nv_mov_b64 %1, %4\nv_mov_b64 %2, %4\nv_mov_b64 %3, %4"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b));)
        } else if constexpr (KIND == 11) { // the gather + matvec pattern of one step: 16 dpp, 16 fma
            REP4(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_fma_f64 %4, %4, %8, %9\nv_fma_f64 %5, %5, %8, %9\nv_fma_f64 %6, %6, %8, %9\nv_fma_f64 %7, %7, %8, %9"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3)
                              : "v"(b), "v"(c));)
        } else if constexpr (KIND == 12) { // one dependent chain of fma
            REP4(asm volatile("v_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2\nv_fma_f64 %0, %0, %1, %2"
                              : "+v"(x0) : "v"(b), "v"(c));)
        } else if constexpr (KIND == 14) { // fp64 compare into VCC (the select tree of the Viterbi kernels)
            REP4(asm volatile("v_cmp_gt_f64 vcc, %0, %4\nv_cmp_gt_f64 vcc, %1, %4\nv_cmp_gt_f64 vcc, %2, %4\nv_cmp_gt_f64 vcc, %3, %4"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b) : "vcc");)
        } else if constexpr (KIND == 15) { // 64-bit unsigned compare (the same order for non-negative doubles)
            REP4(asm volatile("v_cmp_gt_u64 vcc, %0, %4\nv_cmp_gt_u64 vcc, %1, %4\nv_cmp_gt_u64 vcc, %2, %4\nv_cmp_gt_u64 vcc, %3, %4"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b) : "vcc");)
        } else if constexpr (KIND == 16) { // select
            REP4(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\nv_cndmask_b32 %1, %1, %4, vcc\nv_cndmask_b32 %2, %2, %4, vcc\nv_cndmask_b32 %3, %3, %4, vcc"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(e) : "vcc");)
        } else if constexpr (KIND == 17) { // 32-bit compare
            REP4(asm volatile("v_cmp_gt_u32 vcc, %0, %4\nv_cmp_gt_u32 vcc, %1, %4\nv_cmp_gt_u32 vcc, %2, %4\nv_cmp_gt_u32 vcc, %3, %4"
                              : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(e) : "vcc");)
        } else if constexpr (KIND == 18) { // the node of the select tree: compare, select index, max
            REP4(asm volatile("v_cmp_gt_f64 vcc, %0, %4\nv_cndmask_b32 %2, %2, %5, vcc\nv_max_f64 %0, %0, %4\n"
                              "v_cmp_gt_f64 vcc, %1, %4\nv_cndmask_b32 %3, %3, %5, vcc\nv_max_f64 %1, %1, %4"
                              : "+v"(x0), "+v"(x1), "+v"(i0), "+v"(i1) : "v"(b), "v"(e) : "vcc");)
        } else if constexpr (KIND == 13) { // s_nop 1 between (hazard filler cost)
            REP4(asm volatile("v_fma_f64 %0, %0, %4, %5\ns_nop 1\nv_fma_f64 %1, %1, %4, %5\ns_nop 1\nv_fma_f64 %2, %2, %4, %5\ns_nop 1\nv_fma_f64 %3, %3, %4, %5\ns_nop 1"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(b), "v"(c));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + i0 + i1 + i2 + i3;
}

template <int KIND>
int run(const char *name, double *out, int ncu, int per_iter = 16)
{
    printf("%-34s", name);
    const int ws[3] = {1, 2, 4};
    for (int w : ws) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int r = 0; r < 3; ++r) // warm and let the clock settle under this load
            hipLaunchKernelGGL(k<KIND>, dim3(ncu * w), dim3(256), 0, 0, out, 0.0);
        CHECK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; ++r)
            hipLaunchKernelGGL(k<KIND>, dim3(ncu * w), dim3(256), 0, 0, out, 0.0);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("  w=%d: %6.0f inst/us/SIMD", w, (double)per_iter * ITERS * w * 5 / (ms * 1e3));
    }
    printf("\n");
    return 0;
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int ncu = p.multiProcessorCount;
    printf("%s  CUs=%d  (600 inst/us/SIMD = one instruction per 4 cycles at 2.4 GHz)\n", p.name, ncu);
    double *out;
    CHECK(hipMalloc(&out, (size_t)ncu * 4 * 256 * 8));
    run<0>("v_fma_f64", out, ncu);
    run<12>("v_fma_f64 dependent chain", out, ncu);
    run<1>("v_mul_f64", out, ncu);
    run<2>("v_add_f64", out, ncu);
    run<7>("v_max_f64", out, ncu);
    run<3>("v_ldexp_f64", out, ncu);
    run<4>("v_rndne_f64", out, ncu);
    run<5>("v_cvt_i32_f64", out, ncu);
    run<6>("v_rcp_f64", out, ncu);
    run<10>("v_mov_b64", out, ncu);
    run<8>("v_mov_b32_dpp", out, ncu);
    run<9>("v_max_i32", out, ncu);
    run<11>("16 dpp + 16 fma_f64", out, ncu, 32);
    run<13>("v_fma_f64 + s_nop 1 (VALU only)", out, ncu);
    run<14>("v_cmp_gt_f64", out, ncu);
    run<15>("v_cmp_gt_u64", out, ncu);
    run<17>("v_cmp_gt_u32", out, ncu);
    run<16>("v_cndmask_b32", out, ncu);
    run<18>("cmp_f64 + cndmask + max_f64 (x2)", out, ncu, 24);
    return 0;
}
