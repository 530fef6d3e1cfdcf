// synth_api.hip -- synthetic observation trajectories generated on the device (SURVEY.md 8d:
// "counter-based generator implemented in the build, identical on host and device; big inputs
// are generated on the GPU box").  Measurement / test support of the C ABI: the BASELINE
// workloads (1024 x 1e6 discrete steps = 4 GB of observations) are drawn where they are used
// instead of travelling over PCIe.  Recipe restated from bhmm/util/testsystems.py:26-65,159-160
// and bhmm/hmm/generic_hmm.py:435-507 (hidden path by inverse CDF of pi / rows of A, then one
// emission per step); the uniforms come from the engine's counter-based stream, so a host
// restatement (tests/synth_host.py) reproduces every trajectory bit for bit (discrete).
#include <math.h>

#include <algorithm>
#include <string>
#include <vector>

#include "ctx.hpp"

namespace bhmm {
int invalid_arg(const std::string &msg);

namespace {

__device__ __forceinline__ double synth_uniform(uint64_t seed, uint64_t x)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (x + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// first index j in [0, n) with u < cdf[j] (cdf[n-1] is +inf-safe: the last index is returned)
__device__ __forceinline__ int cdf_pick(const double *cdf, int n, double u)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (u < cdf[mid])
            hi = mid;
        else
            lo = mid + 1;
    }
    return lo;
}

// One thread per trajectory (serial in t: the hidden path is a Markov chain), 64 trajectories
// per workgroup, 64 steps buffered per thread in LDS and written out as whole 256-byte rows so
// that the stores are coalesced although every thread walks its own trajectory.
//   stream position of step t of trajectory k: 2 * ((k0 + k) * T + t) (state), + 1 (emission);
//   k0 = index of this call's first trajectory in a larger (sharded) set
template <bool GAUSS>
__global__ __launch_bounds__(64) void k_synth(const double *cdfA, const double *cdfpi,
                                              const double *par0, const double *par1, int n, int M,
                                              int K, int64_t T, uint64_t seed, void *obs_out,
                                              uint8_t *states_out, int64_t k0)
{
    extern __shared__ double sm[];
    double *sA = sm;              // [n*n] row CDFs of A
    double *sE = sm + n * n;      // discrete: [n*M] row CDFs of B; gaussian: mu[n], sigma[n]
    const int ne = GAUSS ? 2 * n : n * M;
    for (int e = threadIdx.x; e < n * n; e += 64)
        sA[e] = cdfA[e];
    for (int e = threadIdx.x; e < ne; e += 64)
        sE[e] = GAUSS ? (e < n ? par0[e] : par1[e - n]) : par0[e];
    using OT = typename std::conditional<GAUSS, double, int32_t>::type;
    OT *tile = reinterpret_cast<OT *>(sE + ne); // [64 threads][64 steps]
    uint8_t *stile = reinterpret_cast<uint8_t *>(tile + 64 * 64);
    __syncthreads();
    const int lane = threadIdx.x;
    const int64_t k = (int64_t)blockIdx.x * 64 + lane;
    const bool live = k < K;
    int s = 0;
    OT *out = static_cast<OT *>(obs_out);
    for (int64_t tb = 0; tb < T; tb += 64) {
        const int cnt = (int)std::min<int64_t>(64, T - tb);
        if (live) {
            for (int q = 0; q < cnt; ++q) {
                const int64_t t = tb + q;
                const uint64_t pos = 2 * (uint64_t)((k0 + k) * T + t);
                const double us = synth_uniform(seed, pos);
                s = t == 0 ? cdf_pick(cdfpi, n, us) : cdf_pick(sA + s * n, n, us);
                const double ue = synth_uniform(seed, pos + 1);
                OT o;
                if constexpr (GAUSS) {
                    // Box-Muller on (ue, a second uniform derived from the same position)
                    const double u2 = synth_uniform(seed ^ 0xD1B54A32D192ED03ull, pos + 1);
                    const double r = sqrt(-2.0 * log(1.0 - ue)); // 1-ue in (0,1]
                    o = sE[s] + sE[n + s] * r * cos(6.283185307179586 * u2);
                } else {
                    o = (OT)cdf_pick(sE + (int64_t)s * M, M, ue);
                }
                tile[lane * 64 + ((q + lane) & 63)] = o; // skewed: conflict-free both ways
                stile[lane * 64 + ((q + lane) & 63)] = (uint8_t)s;
            }
        }
        __syncthreads();
        // row r of the tile = 64 consecutive steps of trajectory (block*64 + r)
        for (int r = 0; r < 64; ++r) {
            const int64_t kk = (int64_t)blockIdx.x * 64 + r;
            if (kk < K && lane < cnt) {
                out[kk * T + tb + lane] = tile[r * 64 + ((lane + r) & 63)];
                if (states_out)
                    states_out[kk * T + tb + lane] = stile[r * 64 + ((lane + r) & 63)];
            }
        }
        __syncthreads();
    }
}

} // namespace
} // namespace bhmm

using namespace bhmm;

extern "C" int bhmm_synth_observations_at(void *obs_dev, uint8_t *states_dev, int device,
                                          void *stream, int kind, const double *A, const double *pi,
                                          const double *par0, const double *par1, int N, int M,
                                          int K, int64_t T, uint64_t seed, int64_t first_traj);

extern "C" int bhmm_synth_observations(void *obs_dev, uint8_t *states_dev, int device, void *stream,
                                       int kind, const double *A, const double *pi,
                                       const double *par0, const double *par1, int N, int M, int K,
                                       int64_t T, uint64_t seed)
{
    return bhmm_synth_observations_at(obs_dev, states_dev, device, stream, kind, A, pi, par0, par1, N,
                                      M, K, T, seed, 0);
}

extern "C" int bhmm_synth_observations_at(void *obs_dev, uint8_t *states_dev, int device,
                                          void *stream, int kind, const double *A, const double *pi,
                                          const double *par0, const double *par1, int N, int M,
                                          int K, int64_t T, uint64_t seed, int64_t first_traj)
{
    if (!obs_dev || !A || !pi || !par0 || N < 1 || N > 255 || K < 1 || T < 1 || first_traj < 0)
        return invalid_arg("bad argument");
    if (kind != BHMM_EMIT_GAUSSIAN && kind != BHMM_EMIT_DISCRETE)
        return invalid_arg("kind must be gaussian or discrete");
    if (kind == BHMM_EMIT_GAUSSIAN && !par1)
        return invalid_arg("gaussian emissions need means and sigmas");
    if (kind == BHMM_EMIT_DISCRETE && M < 1)
        return invalid_arg("discrete emissions need M >= 1");
    BHMM_HIP(hipSetDevice(device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    // row CDFs on the host, in the order a host restatement takes them (plain running sums; the
    // last entry is never compared, cdf_pick returns the last index)
    std::vector<double> cA((size_t)N * N), cpi(N), e0, e1;
    for (int i = 0; i < N; ++i) {
        double acc = 0.0;
        for (int j = 0; j < N; ++j)
            cA[(size_t)i * N + j] = (acc += A[(size_t)i * N + j]);
    }
    {
        double acc = 0.0;
        for (int j = 0; j < N; ++j)
            cpi[j] = (acc += pi[j]);
    }
    if (kind == BHMM_EMIT_DISCRETE) {
        e0.resize((size_t)N * M);
        for (int i = 0; i < N; ++i) {
            double acc = 0.0;
            for (int o = 0; o < M; ++o)
                e0[(size_t)i * M + o] = (acc += par0[(size_t)i * M + o]);
        }
    } else {
        e0.assign(par0, par0 + N);
        e1.assign(par1, par1 + N);
    }
    const size_t nd = cA.size() + cpi.size() + e0.size() + e1.size();
    double *d = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), nd * sizeof(double));
    if (e != hipSuccess)
        return hip_fail(e, "hipMalloc");
    double *dA = d, *dpi = dA + cA.size(), *d0 = dpi + cpi.size(), *d1 = d0 + e0.size();
    hipError_t rc = hipMemcpyAsync(dA, cA.data(), cA.size() * sizeof(double), hipMemcpyHostToDevice, st);
    if (rc == hipSuccess)
        rc = hipMemcpyAsync(dpi, cpi.data(), cpi.size() * sizeof(double), hipMemcpyHostToDevice, st);
    if (rc == hipSuccess)
        rc = hipMemcpyAsync(d0, e0.data(), e0.size() * sizeof(double), hipMemcpyHostToDevice, st);
    if (rc == hipSuccess && !e1.empty())
        rc = hipMemcpyAsync(d1, e1.data(), e1.size() * sizeof(double), hipMemcpyHostToDevice, st);
    if (rc == hipSuccess) {
        const bool gauss = kind == BHMM_EMIT_GAUSSIAN;
        const size_t sm = ((size_t)N * N + (gauss ? 2 * (size_t)N : (size_t)N * M)) * sizeof(double) +
                          64 * 64 * (gauss ? sizeof(double) : sizeof(int32_t)) + 64 * 64;
        const dim3 grid((unsigned)((K + 63) / 64)), blk(64);
        if (gauss) {
            if (sm > 64 * 1024)
                rc = hipFuncSetAttribute((const void *)k_synth<true>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            if (rc == hipSuccess)
                hipLaunchKernelGGL(k_synth<true>, grid, blk, sm, st, (const double *)dA,
                                   (const double *)dpi, (const double *)d0, (const double *)d1, N, M, K,
                                   T, seed, obs_dev, states_dev, first_traj);
        } else {
            if (sm > 64 * 1024)
                rc = hipFuncSetAttribute((const void *)k_synth<false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
            if (rc == hipSuccess)
                hipLaunchKernelGGL(k_synth<false>, grid, blk, sm, st, (const double *)dA,
                                   (const double *)dpi, (const double *)d0, (const double *)nullptr, N,
                                   M, K, T, seed, obs_dev, states_dev, first_traj);
        }
        if (rc == hipSuccess)
            rc = hipGetLastError();
    }
    if (rc == hipSuccess)
        rc = hipStreamSynchronize(st);
    (void)hipFree(d);
    if (rc != hipSuccess)
        return hip_fail(rc, "bhmm_synth_observations");
    return BHMM_OK;
}
