// dpp64.hip -- semantics and issue rate of the cross-lane forms the wide family uses:
// v_fmac_f64_dpp row_newbcast, v_permlane32_swap, v_permlane16_swap.
//   hipcc --offload-arch=gfx950 -O3 -o dpp64 dpp64.hip && ./dpp64
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k_sem(double *out)
{
    const int l = threadIdx.x;
    double a = (double)l, one = 1.0, acc = 0.0;
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(one));
    out[l] = acc; // expect 16*(l/16) + 3
    int v0 = l, v1 = 100 + l;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v0), "+v"(v1));
    out[64 + l] = v0;
    out[128 + l] = v1;
    int w0 = l, w1 = 100 + l;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(w0), "+v"(w1));
    out[192 + l] = w0;
    out[256 + l] = w1;
}

template <int MODE>
__global__ void k_rate(double *out, int iters)
{
    double a = threadIdx.x * 1e-3, w = 1.0000001;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (int it = 0; it < iters; ++it) {
#define STEP(c, i)                                                                              \
    if (MODE == 0)                                                                              \
        asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(c) : "v"(a), "v"(w));                    \
    else                                                                                        \
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #i " row_mask:0xf bank_mask:0xf"  \
                     : "+v"(c) : "v"(a), "v"(w));
        STEP(c0, 0) STEP(c1, 1) STEP(c2, 2) STEP(c3, 3) STEP(c4, 4) STEP(c5, 5) STEP(c6, 6) STEP(c7, 7)
        STEP(c0, 8) STEP(c1, 9) STEP(c2, 10) STEP(c3, 11) STEP(c4, 12) STEP(c5, 13) STEP(c6, 14) STEP(c7, 15)
    }
    out[blockIdx.x * 64 + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

int main()
{
    double *d, h[320];
    hipMalloc(&d, 1 << 24);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[5] = {"newbcast:3", "p32swap v0", "p32swap v1", "p16swap v0", "p16swap v1"};
    for (int r = 0; r < 5; ++r) {
        printf("%-11s", names[r]);
        for (int l = 0; l < 64; l += 4)
            printf(" %g", h[r * 64 + l]);
        printf("\n");
    }
    // issue rate per SIMD with 1, 2 and 4 wavefronts per SIMD (1024 SIMDs): is one wavefront's
    // stream of (DPP) fp64 FMAs enough to keep the VALU busy?
    for (int mode = 0; mode < 2; ++mode) {
        for (int wps = 1; wps <= 4; wps *= 2) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            const int iters = 20000, blocks = 1024 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0)
                    hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(64), 0, 0, d, iters);
                else
                    hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(64), 0, 0, d, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s, %d wave(s) per SIMD: %.1f instr/us/SIMD\n",
                   mode ? "v_fmac_f64_dpp row_newbcast" : "v_fmac_f64", wps,
                   16.0 * iters * wps / (ms * 1e3));
        }
    }
    return 0;
}
