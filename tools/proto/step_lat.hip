// step_lat.hip -- where do the cycles of one tile step go?  Stages added one at a time, one workgroup of
// four wavefronts per CU, cycles per iteration from the shader clock of wavefront 0.
//   S0: 16 dependent v_mfma_f64_16x16x4 with distinct operand registers
//   S1: + one workgroup barrier per iteration
//   S2: + the A operands read from LDS (8 ds_read2_b64, all issued first)
//   S3: + the result (times a constant) written to the other LDS buffer: the real dependency cycle
//   S4: + 4 global stores per lane
//   S5: S3 with the reads issued pairwise next to their use
// build: hipcc --offload-arch=gfx950 -O3 -o step_lat step_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int prow(int rho)
{
    const int q = rho & 3, r = rho >> 2;
    return 8 * (q & 1) + 4 * (q >> 1) + r;
}

template <int STAGE>
__global__ __launch_bounds__(256) void kstep(double *out, unsigned long long *clk, int iters)
{
    constexpr int PX = 66;
    __shared__ double sX[2 * 16 * PX];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, s = lane & 15, q = lane >> 4;
    double B[16], Areg[16];
    for (int i = 0; i < 16; ++i) {
        B[i] = 1.0 / 64 + 1e-6 * (tid + i);
        Areg[i] = 1.0 + 1e-3 * i;
    }
    for (int e = tid; e < 2 * 16 * PX; e += 256)
        sX[e] = 1.0;
    int xw[4];
    for (int r = 0; r < 4; ++r)
        xw[r] = prow(q + 4 * r) * PX + 16 * w + s;
    const int xr = prow(s) * PX + q;
    __syncthreads();
    double keep = 0.0;
    const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        const double *X = sX + (it & 1) * 16 * PX;
        double *Xn = sX + ((it & 1) ^ 1) * 16 * PX;
        d4 acc;
        if constexpr (STAGE == 6 || STAGE == 7) {
            // contiguous operand layout: lane (m, q) reads 16 consecutive doubles of row m as 4 x 16 bytes
            // S6: all four reads first; S7: one read ahead of its four matrix instructions
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 *Xv = reinterpret_cast<const d2 *>(X + prow(s) * PX + 16 * q);
            d2 a2[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2)
                a2[k2] = Xv[k2];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[kk >> 1][kk & 1], B[kk], kk == 0 ? d4{0, 0, 0, 0} : acc, 0, 0, 0);
            if constexpr (STAGE == 6) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            }
        } else if constexpr (STAGE == 8) {
            // single 8-byte reads, all first
            double av[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                av[kk] = X[xr + 66 * (kk & 1) + 4 * kk]; // (offsets that cannot pair into ds_read2)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], B[kk], kk == 0 ? d4{0, 0, 0, 0} : acc, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        } else if constexpr (STAGE == 5) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[xr + 4 * kk], B[kk], kk == 0 ? d4{0, 0, 0, 0} : acc, 0, 0, 0);
        } else {
            double av[16];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                av[kk] = STAGE >= 2 ? X[xr + 4 * kk] : Areg[kk];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], B[kk], kk == 0 ? d4{0, 0, 0, 0} : acc, 0, 0, 0);
            if constexpr (STAGE >= 2) {
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
            }
        }
        if constexpr (STAGE >= 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Xn[xw[r]] = acc[r] * (1.0 / 64) + 0.5;
        } else {
            keep += acc[0] + acc[1] + acc[2] + acc[3];
            Areg[it & 15] = keep * 1e-30 + 1.0;
        }
        if constexpr (STAGE == 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[((size_t)(blockIdx.x * 16 + q + 4 * r) * 2000 + (it % 2000)) * 64 + 16 * w + s] = acc[r];
        }
        if constexpr (STAGE >= 1)
            __syncthreads();
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    out[blockIdx.x * 256 + tid] = keep + sX[tid];
    if (blockIdx.x == 0 && tid == 0) {
        clk[0] = c1 - c0;
        clk[1] = w1 - w0;
    }
}

template <typename K>
static void run(const char *name, K kern, int blocks, int iters, double *d, unsigned long long *dclk)
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
    printf("%-52s blocks %4d: %8.3f ms  %7.1f ns/step  %7.1f cycles/step  clock %.2f GHz\n", name, blocks, best,
           1e6 * best / iters, (double)h[0] / iters, (double)h[0] / ((double)h[1] * 10.0));
}

int main()
{
    double *d;
    unsigned long long *dclk;
    (void)hipMalloc(&d, (size_t)512 * 16 * 2000 * 64 * sizeof(double));
    (void)hipMalloc(&dclk, 16);
    const int it = 4000;
    for (int blocks : {256, 512}) {
        run("S0 16 dependent mfma", kstep<0>, blocks, it, d, dclk);
        run("S1 + barrier", kstep<1>, blocks, it, d, dclk);
        run("S2 + A operands from LDS (reads first)", kstep<2>, blocks, it, d, dclk);
        run("S3 + result written to LDS (dependency cycle)", kstep<3>, blocks, it, d, dclk);
        run("S4 + global stores", kstep<4>, blocks, it, d, dclk);
        run("S5 = S3, reads next to their use", kstep<5>, blocks, it, d, dclk);
        run("S6 = S3, 4 x 16-byte reads first", kstep<6>, blocks, it, d, dclk);
        run("S7 = S3, 16-byte reads one ahead", kstep<7>, blocks, it, d, dclk);
        run("S8 = S3, 16 x 8-byte reads first", kstep<8>, blocks, it, d, dclk);
    }
    return 0;
}
