// Does v_mfma_f64_16x16x4 honour the execution mask?  (hipcc --offload-arch=gfx950 -O2)
// Lanes 16..31 (k = 1 of both operands) are switched off around the instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(double *out, int mode)
{
    const int lane = threadIdx.x;
    const double a = 1.0, b = (double)lane;
    d4 D = {-1.0, -1.0, -1.0, -1.0};
    d4 Z = {0.0, 0.0, 0.0, 0.0};
    if (mode == 0 || (lane >> 4) != 1)
        D = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, Z, 0, 0, 0);
    for (int r = 0; r < 4; ++r)
        out[lane * 4 + r] = D[r];
}
int main()
{
    double *d, h[256];
    hipMalloc(&d, sizeof(h));
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (expected with all lanes read: D[r] = 4 j + 96, j = lane & 15)\n", mode);
        for (int lane = 0; lane < 64; lane += 5)
            printf("  lane %2d: %g %g %g %g\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
    }
    return 0;
}
