// Do v_mfma_f64_16x16x4 and v_fma_f64 share an execution pipe on gfx950?
// Every workgroup = 4 wavefronts (one per SIMD), 2 workgroups per CU resident: 2 wavefronts per SIMD.
//   mode 0: all wavefronts run a v_fma_f64 stream        (16 independent chains)
//   mode 1: all wavefronts run a v_mfma_f64_16x16x4 stream (4 independent accumulators)
//   mode 2: first half of the workgroups FMA, second half MFMA: one wavefront of each per SIMD
//           (workgroups b and b + 256 share a CU; the parity of b would select the XCD instead)
//   mode 3 / 4: only the FMA / only the MFMA half of mode 2 (one wavefront per SIMD)
// If the pipes were separate, mode 2 would take max(t0, t1) / ... i.e. about the longer of the two
// at half the wavefronts each; if shared, about (t0 + t1) / 2.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double *out, int iters, int mode)
{
    const bool second = blockIdx.x >= gridDim.x / 2;
    const bool do_mfma = mode == 1 || ((mode == 2 || mode == 4) && second);
    double r = 0.0;
    if ((mode == 3 && second) || (mode == 4 && !second))
        return;
    if (!do_mfma) {
        double a[16];
        for (int i = 0; i < 16; ++i)
            a[i] = threadIdx.x * 1e-3 + i;
        const double m = 0.999999, c = 1e-7;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        for (int i = 0; i < 16; ++i)
            r += a[i];
    } else {
        d4 D[4];
        for (int i = 0; i < 4; ++i)
            D[i] = (d4){0.0, 0.0, 0.0, 0.0};
        const double x = threadIdx.x * 1e-3, y = 1e-3;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(D[i]) : "v"(x), "v"(y));
        for (int i = 0; i < 4; ++i)
            r += D[i][0] + D[i][1] + D[i][2] + D[i][3];
    }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main()
{
    const int blocks = 512, iters = 20000; // 256 CUs x 2 workgroups
    double *d;
    (void)hipMalloc(&d, blocks * 256 * sizeof(double));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, mode);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best)
                best = ms;
        }
        // per wavefront: mode 0: 16 iters FMA instr; mode 1: 4 iters MFMA instr
        printf("mode %d: %.3f ms  (FMA instr/wavefront %d, MFMA instr/wavefront %d)\n", mode, best,
               (mode == 1 || mode == 4) ? 0 : 16 * iters, (mode == 0 || mode == 3) ? 0 : 4 * iters);
    }
    return 0;
}
