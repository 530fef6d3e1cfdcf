// tile_proto.hip -- prototype / microbenchmark of the row-batched MFMA forward recursion
// (16 trajectory segments per workgroup, alpha-tile[16 x N] . A[N x N] on v_mfma_f64_16x16x4).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../bhmm_amd/csrc -o tile_proto tile_proto.hip
// run:   ./tile_proto [steps] [tiles]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "host_common.hpp"
#include "wide_kernels.hpp"

using namespace bhmm;
typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                         \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

__device__ __forceinline__ int row16_max(int v)
{
    v = max(v, dpp_i32<0xB1>(v));
    v = max(v, dpp_i32<0x4E>(v));
    v = max(v, dpp_i32<0x141>(v)); // row_half_mirror
    v = max(v, dpp_i32<0x140>(v)); // row_mirror
    return v;
}

__device__ __forceinline__ void keep_alive(double x) { asm volatile("" ::"v"(x)); }

// physical LDS row of tile row rho = q + 4 r
__device__ __forceinline__ int prow(int rho)
{
    const int q = rho & 3, r = rho >> 2;
    return 8 * (q & 1) + 4 * (q >> 1) + r;
}

// VAR: 0 reads next to their use, emission after the matrix instructions (four asm blocks)
//      1 all operand reads first (sched_group_barrier), emission as in 0
//      2 reads first; emission of the NEXT step after the LDS write (before the barrier), four chains interleaved (C++)
//      3 reads first; emission of the next step interleaved with the matrix instructions (sched_group_barrier)
//      4 as 2, but NO emission at all (p = 1): what the recursion alone costs
template <int N, int VAR, int WPS>
__global__ __launch_bounds__(256, WPS) void k_tile_fwd(const double *__restrict__ A, const double *__restrict__ mu,
                                                       const double *__restrict__ ga, const double *__restrict__ gb,
                                                       double gmg, const double *__restrict__ obs, int T,
                                                       int nsteps, double *__restrict__ alpha, int store, unsigned long long *clk)
{
    unsigned long long tA = 0, tB = 0, tC = 0;
    const unsigned long long cs = __builtin_readcyclecounter(), ws = wall_clock64();
    constexpr int KK = N / 4;
    constexpr int PX = N + 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sX = smem;
    int *sE = reinterpret_cast<int *>(sX + 2 * 16 * PX);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int s = lane & 15, q = lane >> 4;
    double Breg[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
        Breg[kk] = A[(4 * kk + q) * N + 16 * w + s];
    for (int e = tid; e < 16 * N; e += 256)
        sX[(e / N) * PX + (e % N)] = 1.0 / N;
    const int j = 16 * w + s;
    const double mu_j = mu[j], ga_j = ga[j], gb_j = gb[j];
    const double *orow[4];
    double *arow[4];
    int xw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t row = (int64_t)blockIdx.x * 16 + q + 4 * r;
        orow[r] = obs + row * T;
        arow[r] = alpha + row * (int64_t)T * N;
        xw[r] = prow(q + 4 * r) * PX;
    }
    const int xr = prow(s) * PX + q;
    constexpr int PF = 4;
    double oring[PF][4];
#pragma unroll
    for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            oring[u][r] = orow[r][u < nsteps ? u : nsteps - 1];
    double pcur[4] = {1.0, 1.0, 1.0, 1.0};
    auto emit_c = [&](const double (&o)[4], double (&p)[4]) __attribute__((always_inline)) {
        double d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            d[r] = o[r] - mu_j;
        double wv[4], qv[4];
        int tl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double u = fmin(fma(d[i] * d[i], ga_j, gb_j), 1.0);
            const double t = gmg - u;
            wv[i] = (t - gmg) + u;
            tl[i] = __double2loint(t);
            qv[i] = 0x1.e3991e644e6abp+92;
        }
        constexpr double C[10] = {-0x1.b6740fc28f781p+84, 0x1.62c157ee59177p+76, -0x1.ffcb55e82f22cp+67,
                                  0x1.4309126056718p+59,  -0x1.5d87fe9cc5d6fp+50, 0x1.3b2ab6fbde0f7p+41,
                                  -0x1.c6b08d703d48ap+31, 0x1.ebfbdff82c3b9p+21,  -0x1.62e42fefa3a17p+11, 1.0};
#pragma unroll
        for (int k = 0; k < 10; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                qv[i] = __builtin_fma(qv[i], wv[i], C[k]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            p[i] = ldexp(qv[i], tl[i]);
    };
    if constexpr (VAR == 2 || VAR == 3) {
        double o0[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            o0[r] = oring[0][r];
        emit_c(o0, pcur);
    }
    __syncthreads();
    int eP[4] = {0, 0, 0, 0};
    for (int tb = 0; tb < nsteps; tb += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int t = tb + u;
            if (t >= nsteps)
                break;
            double o[4], on[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = oring[u][r];
                on[r] = oring[(u + 1) & 3][r];
                const int tn = t + PF < nsteps ? t + PF : nsteps - 1;
                oring[u][r] = orow[r][tn];
            }
            const double *X = sX + (u & 1) * 16 * PX;
            double *Xn = sX + ((u & 1) ^ 1) * 16 * PX;
            const unsigned long long c0 = clk ? __builtin_readcyclecounter() : 0;
            d4 acc;
            if constexpr (VAR == 0) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[xr + 4 * kk], Breg[kk], kk == 0 ? d4{0.0, 0.0, 0.0, 0.0} : acc, 0, 0, 0);
            } else {
                double av[KK];
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
                    av[kk] = X[xr + 4 * kk];
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], Breg[kk], kk == 0 ? d4{0.0, 0.0, 0.0, 0.0} : acc, 0, 0, 0);
            }
            double pnext[4] = {1.0, 1.0, 1.0, 1.0};
            if constexpr (VAR == 3) {
                emit_c(on, pnext);
                __builtin_amdgcn_sched_group_barrier(0x100, KK / 2, 0);
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                }
            } else if constexpr (VAR != 0) {
                __builtin_amdgcn_sched_group_barrier(0x100, KK / 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, KK, 0);
            }
            unsigned long long c1 = 0;
            if (clk) {
                keep_alive(acc[0]);
                c1 = __builtin_readcyclecounter();
            }
            int E[4] = {0, 0, 0, 0};
            if ((u & 3) == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rho = q + 4 * r;
                    E[r] = max(max(sE[rho], sE[16 + rho]), max(sE[32 + rho], sE[48 + rho]));
                    eP[r] += E[r];
                }
            }
            int pm[4] = {-(1 << 28), -(1 << 28), -(1 << 28), -(1 << 28)};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double p;
                if constexpr (VAR <= 1)
                    p = gauss_pdf_issue(o[r] - mu_j, ga_j, gb_j, gmg);
                else
                    p = pcur[r];
                double v = acc[r] * p;
                if ((u & 3) == 3)
                    v = ldexp(v, -E[r]);
                Xn[xw[r] + j] = v;
                if (store)
                    arow[r][(int64_t)t * N + j] = v;
                if ((u & 3) == 2)
                    pm[r] = max(pm[r], v > 0.0 ? exponent_of(v) : -(1 << 28));
            }
            if ((u & 3) == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = row16_max(pm[r]);
                    if (s == 0)
                        sE[16 * w + q + 4 * r] = m;
                }
            }
            if constexpr (VAR == 2) {
                __builtin_amdgcn_sched_barrier(0);
                emit_c(on, pcur);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (VAR == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pcur[r] = pnext[r];
            }
            unsigned long long c2 = 0;
            if (clk)
                c2 = __builtin_readcyclecounter();
            __syncthreads();
            if (clk) {
                const unsigned long long c3 = __builtin_readcyclecounter();
                tA += c1 - c0;
                tB += c2 - c1;
                tC += c3 - c2;
            }
        }
    }
    if (clk && blockIdx.x == 0 && tid == 0) {
        clk[0] = tA;
        clk[1] = tB;
        clk[2] = tC;
        clk[3] = __builtin_readcyclecounter() - cs;
        clk[4] = wall_clock64() - ws;
    }
    if (!store) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            arow[r][j] = sX[xw[r] + j] + eP[r];
    }
}

// Two tiles per workgroup, advanced alternately by the same four wavefronts (explicit "ping-pong"): while the
// matrix instructions of one tile run, the operands of the other tile's next product -- written one half-step
// earlier -- are already on their way from LDS, so the exchange of a tile hides behind the other tile's chain.
// A's block in registers is shared by both tiles.  No emission (p = 1): what the recursion alone costs, to be
// compared with v4.
template <int N>
__global__ __launch_bounds__(256, 1) void k_tile_fwd_pp(const double *__restrict__ A, int T, int nsteps,
                                                        double *__restrict__ alpha)
{
    constexpr int KK = N / 4;
    constexpr int PX = N + 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sX = smem;                                            // [tile][buffer][16 * PX]
    int *sE = reinterpret_cast<int *>(sX + 4 * 16 * PX);          // [tile][64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int s = lane & 15, q = lane >> 4;
    double Breg[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
        Breg[kk] = A[(4 * kk + q) * N + 16 * w + s];
    for (int e = tid; e < 4 * 16 * PX; e += 256)
        sX[e] = 1.0 / N;
    for (int e = tid; e < 128; e += 256)
        sE[e] = 0;
    const int j = 16 * w + s;
    int xw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        xw[r] = prow(q + 4 * r) * PX;
    const int xr = prow(s) * PX + q;
    __syncthreads();
    double av[2][KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
        av[0][kk] = sX[xr + 4 * kk]; // tile 0, buffer 0
    int eP[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int tb = 0; tb < nsteps; tb += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int tile = 0; tile < 2; ++tile) {
                // operands of the OTHER tile's next product (tile 1: this step; tile 0: the next step)
                {
                    const int ob = tile == 0 ? (u & 1) : ((u + 1) & 1);
                    const double *Xo = sX + ((tile ^ 1) * 2 + ob) * 16 * PX;
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk)
                        av[tile ^ 1][kk] = Xo[xr + 4 * kk];
                }
                d4 acc;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tile][kk], Breg[kk], kk == 0 ? d4{0.0, 0.0, 0.0, 0.0} : acc, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, KK, 0); // the LDS reads first
                __builtin_amdgcn_sched_group_barrier(0x008, KK, 0); // then the matrix instructions
                double *Xn = sX + (tile * 2 + ((u & 1) ^ 1)) * 16 * PX;
                int *sEt = sE + 64 * tile;
                int E[4] = {0, 0, 0, 0};
                if (u == 3) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rho = q + 4 * r;
                        E[r] = max(max(sEt[rho], sEt[16 + rho]), max(sEt[32 + rho], sEt[48 + rho]));
                        eP[tile][r] += E[r];
                    }
                }
                int pm[4] = {-(1 << 28), -(1 << 28), -(1 << 28), -(1 << 28)};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = acc[r];
                    if (u == 3)
                        v = ldexp(v, -E[r]);
                    Xn[xw[r] + j] = v;
                    if (u == 2)
                        pm[r] = max(pm[r], v > 0.0 ? exponent_of(v) : -(1 << 28));
                }
                if (u == 2) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = row16_max(pm[r]);
                        if (s == 0)
                            sEt[16 * w + q + 4 * r] = m;
                    }
                }
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int tile = 0; tile < 2; ++tile)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = ((int64_t)blockIdx.x * 2 + tile) * 16 + q + 4 * r;
            alpha[row * (int64_t)T * N + j] = sX[(tile * 2 + (nsteps & 1)) * 16 * PX + xw[r] + j] + eP[tile][r];
        }
}

// naive reference: one workgroup of N threads per row, normalised every step
template <int N>
__global__ void k_ref_fwd(const double *A, const double *mu, const double *sig, const double *obs, int T,
                          int nsteps, double *alpha)
{
    __shared__ double x[N], red[N];
    const int j = threadIdx.x;
    const int64_t row = blockIdx.x;
    x[j] = 1.0 / N;
    __syncthreads();
    for (int t = 0; t < nsteps; ++t) {
        double acc = 0.0;
        for (int i = 0; i < N; ++i)
            acc += x[i] * A[i * N + j];
        const double z = (obs[row * T + t] - mu[j]) / sig[j];
        const double v = acc * exp(-0.5 * z * z) / (sqrt(2.0 * M_PI) * sig[j]);
        red[j] = v;
        __syncthreads();
        double sum = 0.0;
        for (int i = 0; i < N; ++i)
            sum += red[i];
        __syncthreads();
        x[j] = v / sum;
        alpha[(row * T + t) * N + j] = x[j];
        __syncthreads();
    }
}

template <int N, int VAR, int WPS>
static void run(const char *name, int tiles, int T, int nsteps, const double *dA, const double *dmu,
                const double *dga, const double *dgb, double gmg, const double *dobs, double *dalpha,
                const std::vector<double> &ref, int refrows)
{
    constexpr int PX = N + 2;
    const size_t sm = ((size_t)2 * 16 * PX) * sizeof(double) + 64 * sizeof(int);
    auto kern = k_tile_fwd<N, VAR, WPS>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // correctness on the first refrows rows (store = 1, few tiles)
    double maxdev = 0.0;
    if (refrows > 0) {
        const int ct = (refrows + 15) / 16;
        hipLaunchKernelGGL(kern, dim3(ct), dim3(256), sm, 0, dA, dmu, dga, dgb, gmg, dobs, T, nsteps, dalpha, 1, (unsigned long long *)nullptr);
        CK(hipDeviceSynchronize());
        std::vector<double> h((size_t)refrows * T * N);
        for (int r = 0; r < refrows; ++r)
            CK(hipMemcpy(h.data() + (size_t)r * T * N, dalpha + (size_t)r * T * N, (size_t)nsteps * N * sizeof(double),
                         hipMemcpyDeviceToHost));
        for (int r = 0; r < refrows; ++r)
            for (int t = 0; t < nsteps; ++t) {
                double sum = 0.0;
                for (int j = 0; j < N; ++j)
                    sum += h[((size_t)r * T + t) * N + j];
                for (int j = 0; j < N; ++j) {
                    const double a = h[((size_t)r * T + t) * N + j] / sum, b = ref[((size_t)r * T + t) * N + j];
                    const double d = fabs(a - b) / fmax(b, 1e-280);
                    if (b > 1e-200 && d > maxdev)
                        maxdev = d;
                    if (!(sum > 0.0))
                        maxdev = 1e300;
                }
            }
    }
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), sm, 0, dA, dmu, dga, dgb, gmg, dobs, T, nsteps, dalpha, 0, (unsigned long long *)nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms);
    }
    CK(hipGetLastError());
    // with store
    float bests = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), sm, 0, dA, dmu, dga, dgb, gmg, dobs, T, nsteps, dalpha, 1, (unsigned long long *)nullptr);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        bests = fminf(bests, ms);
    }
    {
        unsigned long long *dclk, h[5];
        CK(hipMalloc(&dclk, 64));
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), sm, 0, dA, dmu, dga, dgb, gmg, dobs, T, nsteps, dalpha, 0, dclk);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, dclk, sizeof(h), hipMemcpyDeviceToHost));
        printf("    phases (cycles/step, wave 0 of tile 0): mfma-issue %.0f  emit+write %.0f  barrier %.0f  total %.0f  clock %.2f GHz\n",
               (double)h[0] / nsteps, (double)h[1] / nsteps, (double)h[2] / nsteps, (double)h[3] / nsteps,
               (double)h[3] / ((double)h[4] * 10.0));
        CK(hipFree(dclk));
    }
    const double rowsteps = (double)tiles * 16 * nsteps;
    printf("%-28s N=%d tiles=%d steps=%d: %.3f ms (no store) %.3f ms (store)  %.1f ns/tile-step  "
           "%.2f TFLOP/s  maxdev %.2e\n",
           name, N, tiles, nsteps, best, bests, 1e6 * best / nsteps, 2.0 * N * N * rowsteps / (best * 1e-3) / 1e12,
           maxdev);
}

int main(int argc, char **argv)
{
    const int nsteps = argc > 1 ? atoi(argv[1]) : 2000;
    const int T = nsteps;
    constexpr int N = 64;
    const int maxtiles = 1024;
    const int64_t rows = (int64_t)maxtiles * 16;
    std::vector<double> A(N * N), mu(N), sig(N), ga(N), gb(N);
    srand(1);
    for (int i = 0; i < N; ++i) {
        double sum = 0.0;
        for (int j = 0; j < N; ++j) {
            A[i * N + j] = (rand() / (double)RAND_MAX) * 0.01 + (i == j ? 0.9 : 0.0) + (abs(i - j) == 1 ? 0.04 : 0.0);
            sum += A[i * N + j];
        }
        for (int j = 0; j < N; ++j)
            A[i * N + j] /= sum;
        mu[i] = -5.0 + 10.0 * i / (N - 1);
        sig[i] = 0.5 + 1.5 * i / (N - 1);
    }
    double gmg;
    gauss_pdf_constants(N, N, sig.data(), ga.data(), gb.data(), &gmg);
    std::vector<double> obs((size_t)rows * T);
    for (auto &o : obs)
        o = -6.0 + 12.0 * (rand() / (double)RAND_MAX);
    double *dA, *dmu, *dsig, *dga, *dgb, *dobs, *dalpha, *dref;
    CK(hipMalloc(&dA, N * N * 8));
    CK(hipMalloc(&dmu, N * 8));
    CK(hipMalloc(&dsig, N * 8));
    CK(hipMalloc(&dga, N * 8));
    CK(hipMalloc(&dgb, N * 8));
    CK(hipMalloc(&dobs, obs.size() * 8));
    CK(hipMalloc(&dalpha, (size_t)rows * T * N * 8));
    const int refrows = 32;
    CK(hipMalloc(&dref, (size_t)refrows * T * N * 8));
    CK(hipMemcpy(dA, A.data(), N * N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dmu, mu.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsig, sig.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dga, ga.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dgb, gb.data(), N * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dobs, obs.data(), obs.size() * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_ref_fwd<N>, dim3(refrows), dim3(N), 0, 0, dA, dmu, dsig, dobs, T, nsteps, dref);
    CK(hipDeviceSynchronize());
    std::vector<double> ref((size_t)refrows * T * N);
    CK(hipMemcpy(ref.data(), dref, ref.size() * 8, hipMemcpyDeviceToHost));

    for (int tiles : {256, 512}) {
        run<N, 0, 2>("v0 baseline", tiles, T, nsteps, dA, dmu, dga, dgb, gmg, dobs, dalpha, ref, refrows);
        run<N, 1, 2>("v1 reads first", tiles, T, nsteps, dA, dmu, dga, dgb, gmg, dobs, dalpha, ref, refrows);
        run<N, 2, 2>("v2 emission before barrier", tiles, T, nsteps, dA, dmu, dga, dgb, gmg, dobs, dalpha, ref, refrows);
        run<N, 3, 2>("v3 emission between mfma", tiles, T, nsteps, dA, dmu, dga, dgb, gmg, dobs, dalpha, ref, refrows);
        run<N, 4, 2>("v4 no emission", tiles, T, nsteps, dA, dmu, dga, dgb, gmg, dobs, dalpha, 0 ? ref : ref, 0);
        {   // two tiles per workgroup, alternated explicitly: tiles / 2 workgroups
            constexpr int PX = N + 2;
            const size_t sm = ((size_t)4 * 16 * PX) * sizeof(double) + 128 * sizeof(int);
            auto kern = k_tile_fwd_pp<N>;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(kern, dim3(tiles / 2), dim3(256), sm, 0, dA, T, nsteps, dalpha);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = fminf(best, ms);
            }
            CK(hipGetLastError());
            // the same rows through v4 (one tile per workgroup) for the comparison of the results
            std::vector<double> a((size_t)64 * N), b((size_t)64 * N);
            for (int r = 0; r < 64; ++r)
                CK(hipMemcpy(a.data() + (size_t)r * N, dalpha + (size_t)r * T * N, N * sizeof(double), hipMemcpyDeviceToHost));
            auto k4 = k_tile_fwd<N, 4, 2>;
            const size_t sm4 = ((size_t)2 * 16 * PX) * sizeof(double) + 64 * sizeof(int);
            hipLaunchKernelGGL(k4, dim3(4), dim3(256), sm4, 0, dA, dmu, dga, dgb, gmg, dobs, T, nsteps, dalpha, 0, (unsigned long long *)nullptr);
            CK(hipDeviceSynchronize());
            for (int r = 0; r < 64; ++r)
                CK(hipMemcpy(b.data() + (size_t)r * N, dalpha + (size_t)r * T * N, N * sizeof(double), hipMemcpyDeviceToHost));
            double md = 0.0;
            for (size_t e = 0; e < a.size(); ++e)
                md = fmax(md, fabs(a[e] - b[e]) / fmax(fabs(b[e]), 1e-300));
            const double rowsteps = (double)tiles * 16 * nsteps;
            printf("%-28s N=%d tiles=%d steps=%d: %.3f ms (no store)  %.1f ns per step of the %d workgroups  %.2f TFLOP/s  "
                   "final vectors vs v4: %.2e\n", "v5 two tiles per workgroup", N, tiles, nsteps, best, 1e6 * best / nsteps, tiles / 2,
                   2.0 * N * N * rowsteps / (best * 1e-3) / 1e12, md);
        }
    }
    return 0;
}
