/*
 * bhmm_oracle.c -- CPU restatement of the bhmm forward-backward / Baum-Welch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP kernels in
 * bhmm_amd/csrc and the "port" CPU baseline of bench.py.  Nothing in the product path
 * (bhmm_amd/) may import, link or call it; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * Every function restates the arithmetic (operation order included, see SURVEY.md
 * Appendix A) of one reference routine and cites it.  Paths are relative to the
 * reference tree.  Parity status: PINNED -- tests/test_oracle.py checks every function
 * against tests/golden/ fixtures generated from the reference's own C and Python
 * kernels (tests/golden/gen_golden.py) and against oracle/_ref (the reference C
 * sources compiled in place) when that library is present.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).  Contraction is
 * disabled so the mul/add sequences round exactly like the reference's x86-64 build.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_NO_MEM 2 /* bhmm/hidden/impl_c/_hidden.h:5 */
#define ORC_ERR_CHOICE 3 /* stands in for exit(1) at _hidden.c:299-304 */

/* ---- emission models ------------------------------------------------------------- */

/* bhmm/output_models/impl_c/_gaussian.c:5-21 */
static double orc_gauss_pdf(double o, double mu, double sigma)
{
    const double norm = 1.0 / (sqrt(2.0 * M_PI) * sigma);
    const double z = (o - mu) / sigma;
    return norm * exp(-0.5 * z * z);
}

/* bhmm/output_models/impl_c/_gaussian.c:45-70 followed by the outlier rule of
 * bhmm/output_models/outputmodel.py:119-131 (rows summing to exactly 0 become all ones
 * when ignore_outliers is set; default True for Gaussian, gaussian.py:36).
 * Returns the number of outlier rows. */
long orc_pobs_gaussian(const double *obs, long T, const double *mu, const double *sigma,
                       int N, int ignore_outliers, double *pobs)
{
    long noutl = 0;
    for (long t = 0; t < T; ++t) {
        double *row = pobs + t * (long)N;
        for (int i = 0; i < N; ++i)
            row[i] = orc_gauss_pdf(obs[t], mu[i], sigma[i]);
        if (ignore_outliers) {
            double s = 0.0;
            for (int i = 0; i < N; ++i)
                s += row[i];
            if (s == 0.0) {
                for (int i = 0; i < N; ++i)
                    row[i] = 1.0;
                ++noutl;
            }
        }
    }
    return noutl;
}

/* bhmm/output_models/discrete.py:130-157: pobs[t,:] = B[:, obs[t]] */
void orc_pobs_discrete(const int32_t *obs, long T, const double *B, int N, int M, double *pobs)
{
    for (long t = 0; t < T; ++t)
        for (int i = 0; i < N; ++i)
            pobs[t * (long)N + i] = B[(long)i * M + obs[t]];
}

/* bhmm/output_models/impl_c/_discrete.c:1-32: pout[i, obs[t]] += w[t, i] (caller zeroes
 * and row-normalises, discrete.py:202-215) */
void orc_update_pout(const int32_t *obs, const double *w, long T, int N, int M, double *pout)
{
    for (long t = 0; t < T; ++t) {
        const int32_t o = obs[t];
        for (int i = 0; i < N; ++i)
            pout[(long)i * M + o] += w[t * (long)N + i];
    }
}

/* ---- hidden kernels ---------------------------------------------------------------- */

/* bhmm/hidden/impl_c/_hidden.c:16-66 */
double orc_forward(double *alpha, const double *A, const double *pobs, const double *pi,
                   int N, long T)
{
    double c = 0.0;
    for (int i = 0; i < N; ++i) {
        alpha[i] = pi[i] * pobs[i];
        c += alpha[i];
    }
    double logL = log(c);
    if (c != 0)
        for (int i = 0; i < N; ++i)
            alpha[i] /= c;

    for (long t = 1; t < T; ++t) {
        const double *prev = alpha + (t - 1) * N;
        double *cur = alpha + t * N;
        const double *p = pobs + t * N;
        c = 0.0;
        for (int j = 0; j < N; ++j) {
            double s = 0.0;
            for (int i = 0; i < N; ++i)
                s += prev[i] * A[i * N + j];
            cur[j] = s * p[j];
            c += cur[j];
        }
        if (c != 0)
            for (int j = 0; j < N; ++j)
                cur[j] /= c;
        logL += log(c);
    }
    return logL;
}

/* bhmm/hidden/impl_c/_hidden.c:69-110 */
void orc_backward(double *beta, const double *A, const double *pobs, int N, long T)
{
    double c = 0.0;
    double *last = beta + (T - 1) * N;
    for (int i = 0; i < N; ++i) {
        last[i] = 1.0;
        c += last[i];
    }
    for (int i = 0; i < N; ++i)
        last[i] /= c;

    for (long t = T - 2; t >= 0; --t) {
        const double *nb = beta + (t + 1) * N;
        const double *np = pobs + (t + 1) * N;
        double *cur = beta + t * N;
        c = 0.0;
        for (int i = 0; i < N; ++i) {
            double s = 0.0;
            for (int j = 0; j < N; ++j)
                s += A[i * N + j] * np[j] * nb[j];
            cur[i] = s;
            c += s;
        }
        if (c != 0)
            for (int j = 0; j < N; ++j)
                cur[j] /= c;
    }
}

/* bhmm/hidden/api.py:176-186 (numpy multiply / dot-with-ones / divide); row sum taken
 * in ascending state order like the unbound C twin at _hidden.c:113-131. */
void orc_gamma(double *gamma, const double *alpha, const double *beta, int N, long T)
{
    for (long t = 0; t < T; ++t) {
        double s = 0.0;
        for (int i = 0; i < N; ++i) {
            gamma[t * N + i] = alpha[t * N + i] * beta[t * N + i];
            s += gamma[t * N + i];
        }
        for (int i = 0; i < N; ++i)
            gamma[t * N + i] /= s;
    }
}

/* bhmm/hidden/api.py:191-211 */
void orc_state_counts(double *counts, const double *gamma, int N, long T)
{
    for (int i = 0; i < N; ++i)
        counts[i] = 0.0;
    for (long t = 0; t < T; ++t)
        for (int i = 0; i < N; ++i)
            counts[i] += gamma[t * N + i];
}

/* bhmm/hidden/impl_c/_hidden.c:148-183 */
int orc_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                          const double *beta, int N, long T)
{
    for (int k = 0; k < N * N; ++k)
        C[k] = 0.0;
    double *x = (double *)malloc((size_t)N * N * sizeof(double));
    if (!x)
        return ORC_ERR_NO_MEM;
    for (long t = 0; t + 1 < T; ++t) {
        double S = 0.0;
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                x[i * N + j] = alpha[t * N + i] * A[i * N + j] * pobs[(t + 1) * N + j] *
                               beta[(t + 1) * N + j];
                S += x[i * N + j];
            }
        for (int k = 0; k < N * N; ++k)
            C[k] += x[k] / S;
    }
    free(x);
    return ORC_OK;
}

/* first maximum wins, strict '>' scan: bhmm/hidden/impl_c/_hidden.c:186-200 */
static int orc_argmax(const double *v, int N)
{
    int a = 0;
    double m = v[0];
    for (int i = 1; i < N; ++i)
        if (v[i] > m) {
            m = v[i];
            a = i;
        }
    return a;
}

/* bhmm/hidden/impl_c/_hidden.c:203-281 */
int orc_viterbi(int32_t *path, const double *A, const double *pobs, const double *pi, int N,
                long T)
{
    double *v = (double *)malloc(sizeof(double) * N);
    double *vn = (double *)malloc(sizeof(double) * N);
    double *h = (double *)malloc(sizeof(double) * N);
    int32_t *ptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)T * N);
    int rc = ORC_OK;
    if (!v || !vn || !h || !ptr) {
        rc = ORC_ERR_NO_MEM;
        goto done;
    }
    double S = 0.0;
    for (int i = 0; i < N; ++i) {
        v[i] = pobs[i] * pi[i];
        S += v[i];
    }
    for (int i = 0; i < N; ++i)
        v[i] /= S;
    for (long t = 1; t < T; ++t) {
        S = 0.0;
        for (int j = 0; j < N; ++j) {
            for (int i = 0; i < N; ++i)
                h[i] = v[i] * A[i * N + j];
            const int best = orc_argmax(h, N);
            ptr[t * N + j] = best;
            vn[j] = pobs[t * N + j] * v[best] * A[best * N + j];
            S += vn[j];
        }
        for (int i = 0; i < N; ++i)
            vn[i] /= S;
        double *tmp = v;
        v = vn;
        vn = tmp;
    }
    path[T - 1] = orc_argmax(v, N);
    for (long t = T - 2; t >= 0; --t)
        path[t] = ptr[(t + 1) * N + path[t + 1]];
done:
    free(v);
    free(vn);
    free(h);
    free(ptr);
    return rc;
}

/* Backward path sampling, bhmm/hidden/impl_c/_hidden.c:330-378 with _normalize
 * (:307-319) and the inverse-CDF draw of _random_choice (:283-305).  The uniform for
 * step t is u[t] (u[T-1] is consumed first, matching the order of rand() calls when the
 * caller fills u in reverse: u[T-1] = first rand(), ...).  If u == NULL the libc
 * generator is used exactly like the reference: r = rand()/(RAND_MAX+1.0). */
int orc_sample_path(int32_t *path, const double *alpha, const double *A, int N, long T,
                    const double *u)
{
    double *ps = (double *)malloc(sizeof(double) * N);
    if (!ps)
        return ORC_ERR_NO_MEM;
    int rc = ORC_OK;
    for (long t = T - 1; t >= 0; --t) {
        for (int i = 0; i < N; ++i)
            ps[i] = (t == T - 1) ? alpha[t * N + i] : alpha[t * N + i] * A[i * N + path[t + 1]];
        double s = 0.0;
        for (int i = 0; i < N; ++i)
            s += ps[i];
        for (int i = 0; i < N; ++i)
            ps[i] /= s;
        const double r = u ? u[t] : (double)rand() / ((double)RAND_MAX + 1.0);
        double acc = 0.0;
        int pick = -1;
        for (int i = 0; i < N; ++i) {
            acc += ps[i];
            if (acc >= r) {
                pick = i;
                break;
            }
        }
        if (pick < 0) {
            rc = ORC_ERR_CHOICE;
            break;
        }
        path[t] = pick;
    }
    free(ps);
    return rc;
}

/* bhmm/hidden/impl_c/_hidden.c:321-327 */
void orc_set_seed(int seed)
{
    if (seed >= 0)
        srand((unsigned)seed);
}

/* Fill u[0..T) with the libc stream in the order _sample_path consumes it (t = T-1 first). */
void orc_fill_uniforms_libc(double *u, long T)
{
    for (long t = T - 1; t >= 0; --t)
        u[t] = (double)rand() / ((double)RAND_MAX + 1.0);
}

/* ---- hidden-path statistics used by the Gibbs sweep --------------------------------- */

/* Transition count matrix of hidden paths at lag 1 (bhmm/hmm/generic_hmm.py:297-319,
 * msmtools count_matrix 'sliding' at lag 1) and first-step counts (:321-334). */
void orc_path_counts(const int32_t *path, long T, int N, int64_t *C, int64_t *n0)
{
    if (T <= 0)
        return;
    n0[path[0]] += 1;
    for (long t = 0; t + 1 < T; ++t)
        C[(long)path[t] * N + path[t + 1]] += 1;
}

/* ---- E-step driver (one trajectory) -------------------------------------------------- */

/* Restates MaximumLikelihoodEstimator._forward_backward
 * (bhmm/estimators/maximum_likelihood.py:221-269) for one trajectory with a Gaussian
 * (kind 0) or discrete (kind 1) output model, materialising pobs/alpha/beta/gamma exactly
 * like the reference (no fusion).  work must hold 3*T*N doubles.  gamma (T*N) and
 * C (N*N) are outputs.  Returns the log-likelihood through *logL. */
int orc_estep_one(int kind, const void *obs, long T, int N, int M, const double *A,
                  const double *pi, const double *par0, const double *par1, double *work,
                  double *gamma, double *C, double *logL)
{
    double *pobs = work;
    double *alpha = work + (size_t)T * N;
    double *beta = work + 2 * (size_t)T * N;
    if (kind == 0)
        orc_pobs_gaussian((const double *)obs, T, par0, par1, N, 1, pobs);
    else
        orc_pobs_discrete((const int32_t *)obs, T, par0, N, M, pobs);
    *logL = orc_forward(alpha, A, pobs, pi, N, T);
    orc_backward(beta, A, pobs, N, T);
    orc_gamma(gamma, alpha, beta, N, T);
    return orc_transition_counts(C, A, pobs, alpha, beta, N, T);
}

/* Gaussian emission M-step, bhmm/output_models/gaussian.py:214-272 (two passes: means
 * first, then variances around the NEW means).  obs/gamma are K concatenated trajectories
 * with offsets off[K+1]. */
void orc_estimate_gaussian(const double *obs, const double *gamma, const int64_t *off, int K,
                           int N, double *mu, double *sigma)
{
    double *wsum = (double *)calloc((size_t)N, sizeof(double));
    for (int i = 0; i < N; ++i)
        mu[i] = 0.0;
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < N; ++i) {
            double dot = 0.0, ws = 0.0;
            for (int64_t t = off[k]; t < off[k + 1]; ++t) {
                dot += gamma[t * N + i] * obs[t];
                ws += gamma[t * N + i];
            }
            mu[i] += dot;
            wsum[i] += ws;
        }
    for (int i = 0; i < N; ++i)
        mu[i] /= wsum[i];
    for (int i = 0; i < N; ++i)
        sigma[i] = 0.0;
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < N; ++i) {
            double dot = 0.0;
            for (int64_t t = off[k]; t < off[k + 1]; ++t) {
                const double d = obs[t] - mu[i];
                dot += gamma[t * N + i] * (d * d);
            }
            sigma[i] += dot;
        }
    for (int i = 0; i < N; ++i)
        sigma[i] = sqrt(sigma[i] / wsum[i]);
    free(wsum);
}
