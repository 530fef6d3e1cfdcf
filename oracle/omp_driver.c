/* omp_driver.c -- TEST / BENCH INFRASTRUCTURE (never linked into the product).
 *
 * The "all host cores" leg of bench.py's cpu_baseline: the per-trajectory call sequence of
 * bhmm/estimators/maximum_likelihood.py:249-265 (p_obs, forward, backward, gamma, xi counts) over
 * a batch of equally long trajectories, one OpenMP task per trajectory, every thread with its own
 * buffers.  The reference is single-threaded by construction (maximum_likelihood.py:26-27); this is
 * what it would take to use the whole host for it -- trajectories are independent given the model.
 *
 * Built twice by oracle/Makefile:
 *   _ref/libbhmm_ref_omp.so  with the reference's own C sources (-DORC_USE_REF: _forward, _backward,
 *                            _compute_transition_counts, _p_obs from bhmm/hidden/impl_c/_hidden.c,
 *                            bhmm/output_models/impl_c/_gaussian.c), when the reference tree exists;
 *   liboracle_omp.so         with this repo's restatement (bhmm_oracle.c).
 * gamma (numpy in the reference, hidden/api.py:176-186) and the discrete p_obs gather
 * (discrete.py:150-153) are plain loops here; the outlier rule is outputmodel.py:126-130.
 */
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_USE_REF
double _forward(double *alpha, const double *A, const double *pobs, const double *pi, int N, int T);
void _backward(double *beta, const double *A, const double *pobs, int N, int T);
int _compute_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                               const double *beta, int N, int T);
void _p_obs(double *o, double *mus, double *sigmas, int N, int T, double *p);
#define FWD(al, A, p, pi, N, T) _forward(al, A, p, pi, N, (int)(T))
#define BWD(be, A, p, N, T) _backward(be, A, p, N, (int)(T))
#define XI(C, A, p, al, be, N, T) _compute_transition_counts(C, A, p, al, be, N, (int)(T))
#else
double orc_forward(double *alpha, const double *A, const double *pobs, const double *pi, int N, long T);
void orc_backward(double *beta, const double *A, const double *pobs, int N, long T);
int orc_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                          const double *beta, int N, long T);
void orc_pobs_gaussian(const double *o, long T, const double *mu, const double *sigma, int N, int outlier,
                       double *p);
#define FWD(al, A, p, pi, N, T) orc_forward(al, A, p, pi, N, T)
#define BWD(be, A, p, N, T) orc_backward(be, A, p, N, T)
#define XI(C, A, p, al, be, N, T) orc_transition_counts(C, A, p, al, be, N, T)
#endif

/* obs: K x T (doubles for kind 0, int32 for kind 1).  logL[K] out.  Returns the number of threads used
 * (negative: allocation failure). */
int orc_estep_batch_omp(int kind, const void *obs, int K, long T, int N, int M, const double *A,
                        const double *pi, const double *par0, const double *par1, int nthreads,
                        double *logL)
{
    int fail = 0, used = 1;
    if (nthreads > 0)
        omp_set_num_threads(nthreads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
        const size_t rows = (size_t)T * N;
        double *buf = (double *)malloc((4 * rows + (size_t)N * N) * sizeof(double));
        double *mu = (double *)malloc(2 * (size_t)N * sizeof(double));
        if (!buf || !mu) {
#pragma omp atomic write
            fail = 1;
        } else {
            double *pobs = buf, *alpha = buf + rows, *beta = buf + 2 * rows, *gamma = buf + 3 * rows,
                   *C = buf + 4 * rows;
            if (kind == 0) {
                memcpy(mu, par0, (size_t)N * sizeof(double));
                memcpy(mu + N, par1, (size_t)N * sizeof(double));
            }
#pragma omp for schedule(dynamic, 1)
            for (int k = 0; k < K; ++k) {
                if (kind == 0) {
                    const double *o = (const double *)obs + (size_t)k * T;
#ifdef ORC_USE_REF
                    _p_obs((double *)o, mu, mu + N, N, (int)T, pobs);
                    for (long t = 0; t < T; ++t) { /* outputmodel.py:126-130 */
                        double s = 0.0;
                        for (int i = 0; i < N; ++i)
                            s += pobs[t * N + i];
                        if (s == 0.0)
                            for (int i = 0; i < N; ++i)
                                pobs[t * N + i] = 1.0;
                    }
#else
                    orc_pobs_gaussian(o, T, mu, mu + N, N, 1, pobs);
#endif
                } else {
                    const int32_t *o = (const int32_t *)obs + (size_t)k * T;
                    for (long t = 0; t < T; ++t)
                        for (int i = 0; i < N; ++i)
                            pobs[t * N + i] = par0[(size_t)i * M + o[t]];
                }
                logL[k] = FWD(alpha, A, pobs, pi, N, T);
                BWD(beta, A, pobs, N, T);
                for (long t = 0; t < T; ++t) { /* hidden/api.py:176-186 */
                    double s = 0.0;
                    for (int i = 0; i < N; ++i) {
                        gamma[t * N + i] = alpha[t * N + i] * beta[t * N + i];
                        s += gamma[t * N + i];
                    }
                    for (int i = 0; i < N; ++i)
                        gamma[t * N + i] /= s;
                }
                XI(C, A, pobs, alpha, beta, N, T);
            }
        }
        free(buf);
        free(mu);
    }
    return fail ? -1 : used;
}
