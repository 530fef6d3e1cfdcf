// mfma_lat.hip -- issue / latency of the fp64 matrix instructions on gfx950, one wavefront per SIMD:
//   v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 with 1, 2, 4 independent accumulator chains,
//   cycles per instruction from the shader clock (s_memtime) of wavefront 0.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_lat mfma_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256) void k16(double *out, unsigned long long *clk, int iters)
{
    d4 D[CH];
    for (int i = 0; i < CH; ++i)
        D[i] = (d4){0.0, 0.0, 0.0, 0.0};
    const double x = threadIdx.x * 1e-3, y = 1e-3;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < CH; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(D[i]) : "v"(x), "v"(y));
    double r = 0.0;
    for (int i = 0; i < CH; ++i)
        r += D[i][0] + D[i][1] + D[i][2] + D[i][3];
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

template <int CH>
__global__ __launch_bounds__(256) void k4(double *out, unsigned long long *clk, int iters)
{
    double D[CH];
    for (int i = 0; i < CH; ++i)
        D[i] = 0.0;
    const double x = threadIdx.x * 1e-3, y = 1e-3;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < CH; ++i)
            asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(D[i]) : "v"(x), "v"(y));
    double r = 0.0;
    for (int i = 0; i < CH; ++i)
        r += D[i];
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

// dependent chain: MFMA -> VALU multiply -> (as A operand) MFMA : the round trip of one recursion step
__global__ __launch_bounds__(256) void kchain(double *out, unsigned long long *clk, int iters)
{
    d4 D = (d4){0.0, 0.0, 0.0, 0.0};
    double x = threadIdx.x * 1e-3;
    const double y = 1e-3;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=v"(D) : "v"(x), "v"(y));
        asm volatile("s_nop 15\n\ts_nop 3\n\tv_mul_f64 %0, %1, %2" : "=v"(x) : "v"(D[0]), "v"(y));
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

// LDS round trip with a workgroup barrier: write, barrier, read (one wavefront per SIMD)
__global__ __launch_bounds__(256) void klds(double *out, unsigned long long *clk, int iters)
{
    __shared__ double sx[256 * 2];
    double x = threadIdx.x * 1e-3;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        sx[(it & 1) * 256 + threadIdx.x] = x;
        __syncthreads();
        x += sx[(it & 1) * 256 + ((threadIdx.x + 64) & 255)];
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

template <typename K>
static void run(const char *name, K kern, int blocks, int iters, int per_iter, double *d, unsigned long long *dclk)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e9f;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, dclk, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) {
            best = ms;
            (void)hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost);
        }
    }
    const double n = (double)iters * per_iter;
    printf("%-34s blocks %4d: %8.3f ms  %7.1f ns/instr  %7.1f cycles/instr  clock %.2f GHz\n", name, blocks, best,
           1e6 * best / n, (double)h[0] / n, (double)h[0] / ((double)h[1] * 10.0));
}

int main()
{
    double *d;
    unsigned long long *dclk;
    (void)hipMalloc(&d, 2048 * 256 * sizeof(double));
    (void)hipMalloc(&dclk, 16);
    const int it = 20000;
    for (int blocks : {1, 256, 512}) {
        run("mfma 16x16x4 f64, 1 chain", k16<1>, blocks, it, 1, d, dclk);
        run("mfma 16x16x4 f64, 2 chains", k16<2>, blocks, it, 2, d, dclk);
        run("mfma 16x16x4 f64, 4 chains", k16<4>, blocks, it, 4, d, dclk);
        run("mfma 4x4x4 4b f64, 1 chain", k4<1>, blocks, it, 1, d, dclk);
        run("mfma 4x4x4 4b f64, 2 chains", k4<2>, blocks, it, 2, d, dclk);
        run("mfma 4x4x4 4b f64, 4 chains", k4<4>, blocks, it, 4, d, dclk);
        run("mfma 4x4x4 4b f64, 8 chains", k4<8>, blocks, it, 8, d, dclk);
        run("mfma16 -> v_mul -> mfma16 round trip", kchain, blocks, it, 1, d, dclk);
        run("lds write/barrier/read round trip", klds, blocks, it, 1, d, dclk);
    }
    return 0;
}
