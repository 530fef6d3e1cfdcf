// wide_kernels.hpp -- E-step kernels for 9..64 hidden states (BASELINE configs[3]: N = 64).
//
// The state vector no longer fits one lane, so the mapping flips: NP = 16, 32 or 64 lanes
// cooperate on one trajectory, lane j owns state j (64/NP trajectories per wavefront), and the
// recursions run serially in t per trajectory.  Chunking in time (as for N <= 8) would cost N
// forward recursions per step in the prescan -- 64x at N = 64 -- which is more than the chip
// gains from the extra parallelism at the trajectory counts of configs[3]; see DESIGN.md.
//
//   k_wide_fwd : alpha (row-major, the reference layout) + per-trajectory log-likelihood
//                (_hidden.c:16-66).  Lane j keeps column j of A in registers; alpha_t is
//                exchanged through LDS (wave-uniform broadcast reads).
//   k_wide_bwd : beta in registers only; gamma, xi row i (C'[i][:] in registers, rank-1
//                update per step), emission statistics.  A lives in LDS with padded rows
//                (lane i reads row i conflict-free); p o beta is exchanged through LDS.
//   k_wide_finalize : fixed-order sums over trajectories -> packed statistics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "estep_kernels.hpp"

namespace bhmm {

// model parameters of the wide family live in one device buffer:
//   A[n*n] | pi[n] | mu[n] | 1/sigma[n] | 1/(sqrt(2pi) sigma)[n] | sigma[n]
struct WideModel {
    const double *A, *pi, *mu, *isig, *cnorm, *sigma;
    const double *B; // [n][M] row-major (discrete)
    int n, M;
};

template <int NP>
__device__ __forceinline__ double wgroup_sum(double v)
{
    // lanes of one row (16) first, on DPP; then across rows with wave shuffles
    v += xchg_f64<1>(v);
    v += xchg_f64<2>(v);
    v += xchg_f64<4>(v);
    {
        const int lo = dpp_i32<0x140>(__double2loint(v)); // row_mirror: i <-> 15 - i
        const int hi = dpp_i32<0x140>(__double2hiint(v));
        v += __hiloint2double(hi, lo);
    }
    if constexpr (NP >= 32)
        v += __shfl_xor(v, 16, 64);
    if constexpr (NP >= 64)
        v += __shfl_xor(v, 32, 64);
    return v;
}

template <int NP>
__device__ __forceinline__ int wgroup_max(int v)
{
    v = max(v, xchg_i32<1>(v));
    v = max(v, xchg_i32<2>(v));
    v = max(v, xchg_i32<4>(v));
    v = max(v, dpp_i32<0x140>(v));
    if constexpr (NP >= 32)
        v = max(v, __shfl_xor(v, 16, 64));
    if constexpr (NP >= 64)
        v = max(v, __shfl_xor(v, 32, 64));
    return v;
}

template <int NP>
__device__ __forceinline__ unsigned long long wgroup_mask(int lane)
{
    if constexpr (NP == 64)
        return ~0ull;
    else
        return ((1ull << NP) - 1) << (lane / NP * NP);
}

// emission probability of MY state at global step gt (+ outlier rule over the group)
template <int NP, int KIND>
__device__ __forceinline__ double wide_emit(const WideModel &m, int j, bool real, int64_t gt,
                                            const void *obs_rm, double mu_j, double is_j,
                                            double cn_j, unsigned long long gmask, double &o,
                                            int &sym)
{
    double p = 0.0;
    if constexpr (KIND == EMIT_GAUSS) {
        o = static_cast<const double *>(obs_rm)[gt];
        const double z = (o - mu_j) * is_j;
        p = real ? cn_j * exp(-0.5 * z * z) : 0.0;
        if ((__ballot(p != 0.0) & gmask) == 0ull)
            p = real ? 1.0 : 0.0; // outputmodel.py:126-130
    } else if constexpr (KIND == EMIT_DISC) {
        sym = static_cast<const int32_t *>(obs_rm)[gt];
        p = real ? m.B[(int64_t)j * m.M + sym] : 0.0;
    } else {
        p = real ? static_cast<const double *>(obs_rm)[gt * m.n + j] : 0.0;
    }
    return p;
}

// Segments: a trajectory may be cut into time segments that are processed by different lane
// groups (one segment per trajectory = the plain serial recursion).  With several segments the
// boundary vectors come from warm-ups over the W steps before / after the segment and are
// verified afterwards by k_wide_check (same scheme as k_estep<..., SPEC>, see estep_sweep.hpp); alpha of
// the step before a segment is read from the previous segment's output by the backward pass.
struct Segs {
    const int32_t *traj; // trajectory of the segment
    const int64_t *t0;   // first step inside the trajectory
    const int32_t *len;
    int nseg;
    int W;
};

template <int NP, int KIND>
__global__ __launch_bounds__(64) void k_wide_fwd(const WideModel m, const int64_t *off, const Segs sg,
                                                 const void *obs_rm, double *alpha_rm,
                                                 double *logL_seg, double *a_entry, double *a_exit)
{
    constexpr int GP = 64 / NP;
    __shared__ __attribute__((aligned(16))) double xch[GP][NP];
    const int lane = threadIdx.x;
    const int gi = lane / NP, j = lane % NP;
    const int s = blockIdx.x * GP + gi;
    if (s >= sg.nseg)
        return;
    const int n = m.n;
    const bool real = j < n;
    const int k = sg.traj[s];
    const int64_t o0 = off[k];
    const int64_t t0 = sg.t0[s], t1 = t0 + sg.len[s];
    if (t1 <= t0) {
        if (j == 0)
            logL_seg[s] = 0.0;
        return;
    }
    const unsigned long long gmask = wgroup_mask<NP>(lane);
    double Acol[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        Acol[i] = (real && i < n) ? m.A[(int64_t)i * n + j] : 0.0;
    const double mu_j = (KIND == EMIT_GAUSS && real) ? m.mu[j] : 0.0;
    const double is_j = (KIND == EMIT_GAUSS && real) ? m.isig[j] : 0.0;
    const double cn_j = (KIND == EMIT_GAUSS && real) ? m.cnorm[j] : 0.0;
    const double pi_j = real ? m.pi[j] : 0.0;

    const int64_t tw = (t0 - sg.W > 0) ? t0 - sg.W : 0; // warm-up start (0: exact start)
    double a = real ? 1.0 / (double)n : 0.0, P = 1.0;
    int eP = 0;
    for (int64_t t = tw; t < t1; ++t) {
        double o;
        int sym;
        const double p = wide_emit<NP, KIND>(m, j, real, o0 + t, obs_rm, mu_j, is_j, cn_j, gmask,
                                             o, sym);
        double nj;
        if (t == 0) {
            nj = pi_j * p;
        } else {
            xch[gi][j] = a;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < NP; i += 2) {
                const double2 x = *reinterpret_cast<const double2 *>(&xch[gi][i]);
                acc[(i / 2) & 3] = fma(x.x, Acol[i], acc[(i / 2) & 3]);
                acc[(i / 2) & 3] = fma(x.y, Acol[i + 1], acc[(i / 2) & 3]);
            }
            nj = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * p;
        }
        const double c = wgroup_sum<NP>(nj);
        a = nj * fast_rcp(c);
        if (t >= t0) {
            int e;
            P = frexp(P * c, &e);
            eP += e;
            if (real)
                alpha_rm[(o0 + t) * n + j] = a;
        } else if (t == t0 - 1 && real) {
            a_entry[(int64_t)s * n + j] = a; // the entry vector this segment derived
        }
    }
    if (real)
        a_exit[(int64_t)s * n + j] = a;
    if (j == 0)
        logL_seg[s] = log(P) + (double)eP * 0.693147180559945309417232121458;
}

// statistics per segment: [n*n C' rows | n sum gamma | (gauss) n sum gamma d | n sum gamma d^2]
// discrete symbol table: [nseg][n][M] (dstat)
template <int NP, int KIND>
__global__ __launch_bounds__(64) void k_wide_bwd(const WideModel m, const int64_t *off, const Segs sg,
                                                 const void *obs_rm, const double *alpha_rm,
                                                 double *gamma_rm, double *gamma0, double *part,
                                                 double *dstat, double *b_exit, double *b_entry)
{
    constexpr int GP = 64 / NP;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sA = smem;                          // [NP][NP + 1] padded rows of A
    double *xb = smem + NP * (NP + 1);          // [GP][NP]  (NP*(NP+1) is even: 16-byte aligned)
    const int lane = threadIdx.x;
    const int gi = lane / NP, i = lane % NP;
    const int n = m.n;
    for (int e = lane; e < NP * NP; e += 64) {
        const int r = e / NP, c = e % NP;
        sA[r * (NP + 1) + c] = (r < n && c < n) ? m.A[(int64_t)r * n + c] : 0.0;
    }
    __syncthreads();
    const int s = blockIdx.x * GP + gi;
    if (s >= sg.nseg)
        return;
    const bool real = i < n;
    const int k = sg.traj[s];
    const int64_t o0 = off[k];
    const int64_t T = off[k + 1] - o0;
    const int64_t t0 = sg.t0[s], t1 = t0 + sg.len[s];
    const int S = n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0);
    double *mypart = part + (int64_t)s * S;
    double *mytab = (KIND == EMIT_DISC) ? dstat + (int64_t)s * n * m.M : nullptr;
    if (KIND == EMIT_DISC && real)
        for (int q = 0; q < m.M; ++q)
            mytab[(int64_t)i * m.M + q] = 0.0;
    double Crow[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c)
        Crow[c] = 0.0;
    double sgm = 0.0, sd = 0.0, sdd = 0.0;
    if (t1 > t0) {
        const unsigned long long gmask = wgroup_mask<NP>(lane);
        const double mu_i = (KIND == EMIT_GAUSS && real) ? m.mu[i] : 0.0;
        const double is_i = (KIND == EMIT_GAUSS && real) ? m.isig[i] : 0.0;
        const double cn_i = (KIND == EMIT_GAUSS && real) ? m.cnorm[i] : 0.0;
        const double *arow = sA + i * (NP + 1);
        double *xg = xb + gi * NP;
        // one backward step: b <- A (p o b), rescaled by a power of two; returns A (p o b)[i]
        auto back = [&](double p, double b) {
            xg[i] = p * b;
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < NP; c += 2) {
                const double2 x = *reinterpret_cast<const double2 *>(&xg[c]);
                acc[(c / 2) & 3] = fma(arow[c], x.x, acc[(c / 2) & 3]);
                acc[(c / 2) & 3] = fma(arow[c + 1], x.y, acc[(c / 2) & 3]);
            }
            return (acc[0] + acc[1]) + (acc[2] + acc[3]);
        };
        auto rescale = [&](double br) {
            const int E = wgroup_max<NP>(br > 0.0 ? exponent_of(br) : -(1 << 28));
            return ldexp(br, -E);
        };

        double b = real ? 1.0 / (double)n : 0.0; // _hidden.c:79-88
        if (t1 < T) { // warm-up from te down to t1: beta at t1 - 1
            const int64_t te = (t1 - 1 + sg.W < T - 1) ? t1 - 1 + sg.W : T - 1;
            for (int64_t t = te; t >= t1; --t) {
                double o;
                int sym;
                const double p = wide_emit<NP, KIND>(m, i, real, o0 + t, obs_rm, mu_i, is_i, cn_i,
                                                     gmask, o, sym);
                b = rescale(back(p, b));
            }
            if (real)
                b_exit[(int64_t)s * n + i] = b;
        }
        double a = real ? alpha_rm[(o0 + t1 - 1) * n + i] : 0.0;
        double gam;
        {
            const double g = a * b;
            gam = g * fast_rcp(wgroup_sum<NP>(g));
        }
        for (int64_t t = t1 - 1; t >= t0; --t) {
            double o = 0.0;
            int sym = 0;
            const double p = wide_emit<NP, KIND>(m, i, real, o0 + t, obs_rm, mu_i, is_i, cn_i,
                                                 gmask, o, sym);
            // consume gamma_t
            sgm += gam;
            if constexpr (KIND == EMIT_GAUSS) {
                const double d = o - mu_i;
                const double gd = gam * d;
                sd += gd;
                sdd = fma(gd, d, sdd);
            }
            if constexpr (KIND == EMIT_DISC)
                if (real)
                    mytab[(int64_t)i * m.M + sym] += gam; // row i is private to this lane
            if (gamma_rm && real)
                gamma_rm[(o0 + t) * n + i] = gam;
            if (t == 0) {
                if (real)
                    gamma0[(int64_t)k * n + i] = gam;
                break;
            }
            const double ap = real ? alpha_rm[(o0 + t - 1) * n + i] : 0.0;
            const double br = back(p, b);
            const double q = ap * br;
            const double rS = fast_rcp(wgroup_sum<NP>(q));
            gam = q * rS;
            const double w = ap * rS;
#pragma unroll
            for (int c = 0; c < NP; c += 2) {
                const double2 x = *reinterpret_cast<const double2 *>(&xg[c]);
                Crow[c] = fma(w, x.x, Crow[c]);
                Crow[c + 1] = fma(w, x.y, Crow[c + 1]);
            }
            b = rescale(br);
            if (t == t0 && real) // beta one step before this segment, as derived here
                b_entry[(int64_t)s * n + i] = b;
        }
    }
    if (real) {
#pragma unroll
        for (int c = 0; c < NP; ++c)
            if (c < n)
                mypart[(int64_t)i * n + c] = Crow[c];
        mypart[n * n + i] = sgm;
        if constexpr (KIND == EMIT_GAUSS) {
            mypart[n * n + n + i] = sd;
            mypart[n * n + 2 * n + i] = sdd;
        }
    }
}

// boundary consistency of a segmented run (see k_spec_check): one thread per segment
static __global__ void k_wide_check(const Segs sg, int n, const double *a_entry,
                                    const double *a_exit, const double *b_exit,
                                    const double *b_entry, double tol, unsigned int *result)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= sg.nseg || sg.len[s] == 0 || sg.t0[s] == 0)
        return;
    double dev = 0.0;
    auto cmp = [&](const double *x, const double *y) {
        double sx = 0.0, sy = 0.0;
        for (int j = 0; j < n; ++j) {
            sx += x[j];
            sy += y[j];
        }
        if (!(sx > 0.0) || !(sy > 0.0)) {
            dev = 1.0;
            return;
        }
        for (int j = 0; j < n; ++j) {
            const double xs = x[j] / sx, ys = y[j] / sy;
            const double d = fabs(xs - ys);
            const double r = (ys > 1e-280) ? d / ys : (d > 1e-280 ? 1.0 : 0.0);
            dev = fmax(dev, r);
        }
    };
    cmp(a_entry + (int64_t)s * n, a_exit + (int64_t)(s - 1) * n);
    cmp(b_exit + (int64_t)(s - 1) * n, b_entry + (int64_t)s * n);
    if (!(dev <= tol))
        atomicAdd(&result[0], 1u);
    atomicMax(&result[1], __float_as_uint((float)fmin(dev, 1.0)));
}

// packed statistics (bhmm_amd.h layout) from the per-trajectory partials; one wavefront per
// output entry, trajectory-strided partial sums + fixed shuffle tree.
template <int KIND>
__global__ __launch_bounds__(64) void k_wide_finalize(const WideModel m, int K, int nseg,
                                                      const double *part, const double *dstat,
                                                      const double *logL_k, const double *gamma0,
                                                      double *stats)
{
    const int n = m.n;
    const int S = n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0);
    const int MN = (KIND == EMIT_DISC) ? n * m.M : 0;
    const int oG0 = 1, oC = 1 + n, oSG = oC + n * n, oE = oSG + n;
    const int lane = threadIdx.x;
    int e = blockIdx.x;
    double s = 0.0;
    if (e < S) {
        for (int k = lane; k < nseg; k += 64)
            s += part[(int64_t)k * S + e];
        s = wave_sum(s);
        if (lane == 0) {
            if (e < n * n)
                stats[oC + e] = s * m.A[e];
            else if (e < n * n + n)
                stats[oSG + (e - n * n)] = s;
            else
                stats[oE + (e - n * n - n)] = s;
        }
        return;
    }
    e -= S;
    if (e < MN) {
        for (int k = lane; k < nseg; k += 64)
            s += dstat[(int64_t)k * MN + e];
        s = wave_sum(s);
        if (lane == 0)
            stats[oE + e] = s;
        return;
    }
    e -= MN;
    if (e < n) {
        for (int k = lane; k < K; k += 64)
            s += gamma0[(int64_t)k * n + e];
        s = wave_sum(s);
        if (lane == 0)
            stats[oG0 + e] = s;
        return;
    }
    for (int k = lane; k < K; k += 64)
        s += logL_k[k];
    s = wave_sum(s);
    if (lane == 0)
        stats[0] = s;
}

// plain scaled backward pass with the reference normalisation, beta row-major (_hidden.c:69-110)
template <int NP>
__global__ __launch_bounds__(64) void k_wide_beta(const WideModel m, const int64_t *off, int K,
                                                  const double *pobs_rm, double *beta_rm)
{
    constexpr int GP = 64 / NP;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sA = smem;
    double *xb = smem + NP * (NP + 1);
    const int lane = threadIdx.x;
    const int gi = lane / NP, i = lane % NP;
    const int n = m.n;
    for (int e = lane; e < NP * NP; e += 64) {
        const int r = e / NP, c = e % NP;
        sA[r * (NP + 1) + c] = (r < n && c < n) ? m.A[(int64_t)r * n + c] : 0.0;
    }
    __syncthreads();
    const int k = blockIdx.x * GP + gi;
    if (k >= K)
        return;
    const bool real = i < n;
    const int64_t o0 = off[k];
    const int64_t T = off[k + 1] - o0;
    if (T <= 0)
        return;
    const double *arow = sA + i * (NP + 1);
    double *xg = xb + gi * NP;
    double b = real ? 1.0 / (double)n : 0.0;
    if (real)
        beta_rm[(o0 + T - 1) * n + i] = b;
    for (int64_t t = T - 1; t >= 1; --t) {
        const double p = real ? pobs_rm[(o0 + t) * n + i] : 0.0;
        xg[i] = p * b;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NP; c += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(&xg[c]);
            acc[(c / 2) & 3] = fma(arow[c], x.x, acc[(c / 2) & 3]);
            acc[(c / 2) & 3] = fma(arow[c + 1], x.y, acc[(c / 2) & 3]);
        }
        const double br = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        b = br * fast_rcp(wgroup_sum<NP>(br));
        if (real)
            beta_rm[(o0 + t - 1) * n + i] = b;
    }
}

// xi-counts from given alpha, beta, pobs (row-major), _hidden.c:148-183, for 9..64 states.
// One group of NP lanes per time slab; lane i accumulates row i of C' = sum_t alpha_t[i] w_t[j]
// with w = pobs_{t+1} o beta_{t+1} / S_t; the factor A[i][j] is applied when the slabs are summed.
template <int NP>
__global__ __launch_bounds__(64) void k_wide_xi(const double *A, const double *pobs,
                                                const double *alpha, const double *beta, int n,
                                                int64_t T, int nslab, double *part)
{
    constexpr int GP = 64 / NP;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sA = smem;
    double *xb = smem + NP * (NP + 1);
    const int lane = threadIdx.x;
    const int gi = lane / NP, i = lane % NP;
    for (int e = lane; e < NP * NP; e += 64) {
        const int r = e / NP, c = e % NP;
        sA[r * (NP + 1) + c] = (r < n && c < n) ? A[(int64_t)r * n + c] : 0.0;
    }
    __syncthreads();
    const int slab = blockIdx.x * GP + gi;
    if (slab >= nslab)
        return;
    const bool real = i < n;
    const int64_t per = (T - 1 + nslab - 1) / nslab;
    const int64_t t0 = slab * per, t1 = (t0 + per < T - 1) ? t0 + per : T - 1;
    const double *arow = sA + i * (NP + 1);
    double *xg = xb + gi * NP;
    double Crow[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c)
        Crow[c] = 0.0;
    for (int64_t t = t0; t < t1; ++t) {
        xg[i] = real ? pobs[(t + 1) * n + i] * beta[(t + 1) * n + i] : 0.0;
        const double a = real ? alpha[t * n + i] : 0.0;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NP; c += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(&xg[c]);
            acc[(c / 2) & 3] = fma(arow[c], x.x, acc[(c / 2) & 3]);
            acc[(c / 2) & 3] = fma(arow[c + 1], x.y, acc[(c / 2) & 3]);
        }
        const double q = a * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
        const double w = a * fast_rcp(wgroup_sum<NP>(q));
#pragma unroll
        for (int c = 0; c < NP; c += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(&xg[c]);
            Crow[c] = fma(w, x.x, Crow[c]);
            Crow[c + 1] = fma(w, x.y, Crow[c + 1]);
        }
    }
    if (real)
#pragma unroll
        for (int c = 0; c < NP; ++c)
            if (c < n)
                part[((int64_t)slab * n + i) * n + c] = Crow[c];
}

static __global__ __launch_bounds__(64) void k_wide_xi_sum(const double *A, const double *part,
                                                           int n, int nslab, double *C)
{
    const int e = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nslab; k += 64)
        s += part[(int64_t)k * n * n + e];
    s = wave_sum(s);
    if (threadIdx.x == 0)
        C[e] = s * A[e];
}

} // namespace bhmm
