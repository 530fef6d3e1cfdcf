// tools/micro/dp_chain.hip -- latency of a chain of dependent fp64 additions on one wavefront (what bounds
// a step of the order-faithful any-N kernels, gen_kernels.hpp): 4096 dependent v_add_f64, timed with the
// shader cycle counter and the 100 MHz wall clock; the same with 1, 2 and 4 wavefronts per SIMD-sharing workgroup, and with an LDS read
// feeding every addition.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dp_chain tools/micro/dp_chain.hip && /tmp/dp_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_chain(double *out, long long *cyc, int n, double inc)
{
    double s = threadIdx.x;
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
#pragma unroll 16
    for (int i = 0; i < n; ++i)
        asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(inc));
    const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0)
    {
        cyc[0] = c1 - c0;
        cyc[1] = w1 - w0;
    }
}
__global__ void k_chain_lds(double *out, long long *cyc, int n)
{
    __shared__ double x[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x)
        x[i] = 1.0 / (1 + i);
    __syncthreads();
    double s = 0.0;
    const long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    if (threadIdx.x == 0) {
        for (int i = 0; i + 8 <= n; i += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = x[(i + u) & 4095];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += v[u];
        }
    }
    const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0)
    {
        cyc[0] = c1 - c0;
        cyc[1] = w1 - w0;
    }
}
int main()
{
    double *out;
    long long *cyc, h[8];
    hipMalloc(&out, 1 << 20);
    hipMalloc(&cyc, 64);
    const int n = 4096;
    for (int tpb : {64, 256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_chain, dim3(1), dim3(tpb), 0, 0, out, cyc, n, 1e-9);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
            if (rep)
                printf("chain of %d dependent v_add_f64, %4d threads in one workgroup: %.2f shader cycles, %.2f ns (100 MHz wall clock) per add; launch %.3f ms\n",
                       n, tpb, (double)h[0] / n, (double)h[1] / n * 10.0, ms);
        }
    }
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_chain_lds, dim3(1), dim3(256), 0, 0, out, cyc, n);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        if (rep)
            printf("thread 0 sums %d LDS entries in order, eight loads at a time: %.2f shader cycles, %.2f ns per entry\n", n,
                   (double)h[0] / n, (double)h[1] / n * 10.0);
    }
    return 0;
}
