// host_api.cpp -- C-ABI entry points of the host-side model updates (include/bhmm_amd.h, last
// section): ONE call per EM iteration (bhmm_mstep) and ONE call per Gibbs sweep
// (bhmm_gibbs_parameters) for everything the reference does in Python / numpy / msmtools between
// two passes over the trajectories.  No device code.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <vector>

#include "../../include/bhmm_amd.h"
#include "host_model.hpp"

using namespace bhmm::host;

extern "C" int bhmm_mle_reversible(double *P, int64_t *iterations, const double *C, int n,
                                   int64_t maxiter, double maxerr)
{
    if (!P || !C || n < 1)
        return bhmm::invalid_arg("NULL argument or empty matrix");
    double tot = 0.0;
    for (size_t e = 0; e < (size_t)n * n; ++e)
        tot += C[e];
    if (!(tot > 0.0))
        return bhmm::invalid_arg("count matrix without counts");
    const int64_t it = mle_reversible(C, n, maxiter, maxerr, P);
    if (iterations)
        *iterations = it;
    return BHMM_OK;
}

extern "C" int bhmm_host_connected_sets(int32_t *label, const double *C, int n, double mincount,
                                        int strong)
{
    if (!label || !C || n < 1)
        return bhmm::invalid_arg("NULL argument or empty matrix");
    const Sets sets = connected_sets(C, n, mincount, strong != 0);
    for (size_t s = 0; s < sets.size(); ++s)
        for (int i : sets[s])
            label[i] = (int32_t)s;
    return (int)BHMM_OK;
}

extern "C" int bhmm_host_stationary_vector(double *pi, const double *P, int n)
{
    if (!pi || !P || n < 1)
        return bhmm::invalid_arg("NULL argument or empty matrix");
    stationary_vector(P, n, pi);
    return BHMM_OK;
}

extern "C" int bhmm_host_estimate_tmatrix(double *P, const double *C, int n, int reversible,
                                    const double *fixed_pi, int64_t maxiter, double maxerr,
                                    double mincount, int64_t *iterations)
{
    if (!P || !C || n < 1)
        return bhmm::invalid_arg("NULL argument or empty matrix");
    return estimate_P(C, n, reversible != 0, fixed_pi, maxiter, maxerr, mincount, P, iterations);
}

extern "C" int bhmm_host_partial_rev(double *P, const double *C, int n, const int32_t *in_set,
                                     int64_t maxiter, double maxerr)
{
    if (!P || !C || !in_set || n < 1)
        return bhmm::invalid_arg("NULL argument or empty matrix");
    std::vector<char> mask(n);
    int ns = 0;
    for (int i = 0; i < n; ++i)
        ns += (mask[i] = in_set[i] != 0);
    if (ns == 0 || ns == n)
        return bhmm::invalid_arg("the reversible set must be a proper, non-empty subset");
    partial_rev(C, n, mask, maxiter, maxerr, P);
    return BHMM_OK;
}

extern "C" int bhmm_host_is_reversible(const double *P, int n)
{
    if (!P || n < 1)
        return -1;
    return is_reversible(P, n) ? 1 : 0;
}

extern "C" int bhmm_host_sample_reversible(double *X, const double *C, int n, int64_t nsweeps, uint64_t base, int lanes)
{
    if (!X || !C || n < 1 || nsweeps < 0 || (lanes != 0 && lanes != 1 && lanes != 4))
        return -1;
    sample_reversible_sweeps_base(C, n, nsweeps, base, X, lanes);
    return (lanes == 1) ? 1 : reversible_sampler_lanes();
}

extern "C" int bhmm_host_rng_draws(double *out, int64_t count, int what, double param,
                                   uint64_t seed, uint64_t stream)
{
    if (!out || count < 0)
        return bhmm::invalid_arg("NULL output");
    Rng rng(seed, stream);
    for (int64_t i = 0; i < count; ++i)
        out[i] = what == 0 ? rng.u01() : what == 1 ? rng.normal() : what == 2 ? rng.gamma(param)
                                                                               : rng.u01_open();
    return BHMM_OK;
}

// ---- emission M-steps ------------------------------------------------------------------------
// gaussian.py:214-272 from  sum gamma, sum gamma (o - mu_old), sum gamma (o - mu_old)^2: the new mean
// is mu_old + <d>, the variance around the NEW mean <d^2> - <d>^2.
static int gaussian_mstep(int n, const double *w, const double *sgd, const double *sgdd,
                          const double *mu_old, double *mu, double *sigma)
{
    for (int i = 0; i < n; ++i) {
        const double m1 = sgd[i] / w[i], m2 = sgdd[i] / w[i];
        mu[i] = mu_old[i] + m1;
        sigma[i] = sqrt(std::max(m2 - m1 * m1, 0.0));
    }
    for (int i = 0; i < n; ++i)
        if (sigma[i] < std::numeric_limits<double>::epsilon()) { // (NaN passes, as in numpy)
            bhmm::set_error("at least one sigma is too small to continue.");
            return BHMM_ERR_SIGMA;
        }
    return BHMM_OK;
}

extern "C" int bhmm_mstep(int kind, int n, int M, const double *stats, const double *T_old,
                          const double *par0_old, const double *par1_old, int reversible,
                          int stationary, const double *fixed_pi, int64_t maxiter, double maxerr,
                          double mincount, double *T_new, double *pi_new, double *par0_new,
                          double *par1_new, int32_t *info, double *warm_state)
{
    if (!stats || !T_new || !pi_new || n < 1)
        return bhmm::invalid_arg("bhmm_mstep: NULL argument or no states");
    if (reversible < 0 && !T_old)
        return bhmm::invalid_arg("bhmm_mstep: reversible = -1 (decide from the model) needs T_old");
    const double *g0 = stats + 1, *C = stats + 1 + n, *sc = stats + 1 + n + (size_t)n * n;
    const double *emis = sc + n;
    // maximum_likelihood.py:306-308: reversible iff the CURRENT transition matrix is
    const bool rev = reversible < 0 ? is_reversible(T_old, n) : reversible != 0;
    int64_t its = 0;
    int rc = estimate_P(C, n, rev, stationary ? fixed_pi : nullptr, maxiter, maxerr, mincount, T_new,
                        &its, warm_state);
    if (rc)
        return rc;
    if (stationary) {
        if (fixed_pi)
            memcpy(pi_new, fixed_pi, n * sizeof(double));
        else
            stationary_distribution(T_new, C, n, mincount, pi_new);
    } else if (fixed_pi) {
        memcpy(pi_new, fixed_pi, n * sizeof(double));
    } else {
        double tot = 0.0;
        for (int i = 0; i < n; ++i)
            tot += g0[i];
        for (int i = 0; i < n; ++i)
            pi_new[i] = g0[i] / tot;
    }
    if (kind == BHMM_EMIT_GAUSSIAN) {
        if (!par0_old || !par0_new || !par1_new)
            return bhmm::invalid_arg("bhmm_mstep: gaussian emissions need means in, means / sigmas out");
        if ((rc = gaussian_mstep(n, sc, emis, emis + n, par0_old, par0_new, par1_new)))
            return rc;
    } else if (kind == BHMM_EMIT_DISCRETE) {
        if (!par0_new || M < 1)
            return bhmm::invalid_arg("bhmm_mstep: discrete emissions need B out");
        for (int i = 0; i < n; ++i) { // discrete.py:202-215: row-normalised weighted symbol counts
            double rs = 0.0;
            for (int k = 0; k < M; ++k)
                rs += emis[(size_t)i * M + k];
            for (int k = 0; k < M; ++k)
                par0_new[(size_t)i * M + k] = emis[(size_t)i * M + k] / rs;
        }
    }
    if (info) {
        info[0] = rev ? 1 : 0;
        info[1] = (int32_t)std::min<int64_t>(its, 2147483647);
    }
    (void)par1_old;
    return BHMM_OK;
}

// ---- Gibbs parameter step --------------------------------------------------------------------
extern "C" int bhmm_gibbs_parameters(int kind, int n, int M, const double *path_stats,
                                     const double *prior_C, const double *prior_n0,
                                     const double *prior_B, int reversible, int stationary,
                                     int64_t nsteps, uint64_t seed, uint64_t sweep, double *T,
                                     double *p0, double *par0, double *par1, int32_t *info)
{
    if (!path_stats || !T || !p0 || n < 1)
        return bhmm::invalid_arg("bhmm_gibbs_parameters: NULL argument or no states");
    const size_t nn = (size_t)n * n;
    const double *Cint = path_stats, *n0 = path_stats + nn, *emis = path_stats + nn + n;
    Rng rng(seed, sweep);
    // (1) emission parameters, bayesian_sampling.py:333-339
    if (kind == BHMM_EMIT_GAUSSIAN) {
        if (!par0 || !par1)
            return bhmm::invalid_arg("bhmm_gibbs_parameters: gaussian emissions need means and sigmas");
        // gaussian.py:303-318: mu ~ N(mean of the state's observations, sigma^2 / n), then
        // sigma from the Jeffreys-prior posterior given the NEW mean (chi-square with n - 1)
        const double *cnt = emis, *sd = emis + n, *sdd = emis + 2 * n;
        for (int i = 0; i < n; ++i) {
            const double ni = nearbyint(cnt[i]);
            if (ni > 0.0) {
                const double mu_old = par0[i];
                const double mean_obs = mu_old + sd[i] / ni;
                par0[i] = rng.normal() * par1[i] / sqrt(ni) + mean_obs;
                if (ni > 1.0) {
                    const double chi2 = rng.chisquare(ni - 1.0);
                    const double shift = par0[i] - mu_old;
                    const double s2 = (sdd[i] - 2.0 * shift * sd[i]) / ni + shift * shift;
                    par1[i] = sqrt(std::max(s2, 0.0)) / sqrt(chi2 / ni);
                }
            }
        }
    } else if (kind == BHMM_EMIT_DISCRETE) {
        if (!par0 || M < 1)
            return bhmm::invalid_arg("bhmm_gibbs_parameters: discrete emissions need B");
        std::vector<double> cnt(M);
        for (int i = 0; i < n; ++i) { // discrete.py:243-251: Dirichlet over the positive counts
            for (int k = 0; k < M; ++k)
                cnt[k] = emis[(size_t)i * M + k] + (prior_B ? prior_B[(size_t)i * M + k] : 0.0);
            rng.dirichlet(cnt.data(), M, par0 + (size_t)i * M);
        }
    }
    // (2) transition matrix, bayesian_sampling.py:341-360
    std::vector<double> C(nn);
    for (size_t e = 0; e < nn; ++e)
        C[e] = Cint[e] + (prior_C ? prior_C[e] : 0.0);
    if (reversible) {
        if (connected_sets(C.data(), n, 0.0, true).size() != 1) {
            bhmm::set_error("Encountered disconnected count matrix with sampling option reversible. "
                            "Use prior to ensure connectivity or use reversible=False.");
            return BHMM_ERR_DISCONNECTED;
        }
        std::vector<double> P0(nn), pi(n), X(nn);
        mle_reversible(C.data(), n, 10000, 1e-8, P0.data());
        for (int i = 0; i < n; ++i) // consistent sparsity pattern (:352-357)
            for (int j = 0; j < n; ++j)
                if (P0[(size_t)i * n + j] + P0[(size_t)j * n + i] == 0.0)
                    C[(size_t)i * n + j] = 0.0;
        stationary_vector(P0.data(), n, pi.data());
        double tot = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                X[(size_t)i * n + j] = 0.5 * (pi[i] * P0[(size_t)i * n + j] + pi[j] * P0[(size_t)j * n + i]);
                tot += X[(size_t)i * n + j];
            }
        for (size_t e = 0; e < nn; ++e)
            X[e] /= tot;
        sample_reversible_sweeps(C.data(), n, nsteps, rng, X.data());
        for (int i = 0; i < n; ++i) {
            double rs = 0.0;
            for (int j = 0; j < n; ++j)
                rs += X[(size_t)i * n + j];
            for (int j = 0; j < n; ++j)
                T[(size_t)i * n + j] = X[(size_t)i * n + j] / rs;
        }
    } else {
        // independent Dirichlet rows over the positive entries; a row without counts stays put
        for (int i = 0; i < n; ++i) {
            bool any = false;
            for (int j = 0; j < n; ++j) {
                T[(size_t)i * n + j] = 0.0;
                any = any || C[(size_t)i * n + j] > 0.0;
            }
            if (any)
                rng.dirichlet(&C[(size_t)i * n], n, &T[(size_t)i * n]);
            else
                T[(size_t)i * n + i] = 1.0;
        }
    }
    // (3) initial distribution, :362-370
    if (stationary) {
        stationary_distribution(T, C.data(), n, 0.0, p0);
    } else {
        std::vector<double> w(n);
        for (int i = 0; i < n; ++i) {
            w[i] = n0[i] + (prior_n0 ? prior_n0[i] : 0.0);
            p0[i] = 0.0;
        }
        rng.dirichlet(w.data(), n, p0);
    }
    if (info)
        info[0] = (int32_t)std::min<uint64_t>(rng.ctr, 2147483647u);
    return BHMM_OK;
}
