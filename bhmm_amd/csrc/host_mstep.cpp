// host_mstep.cpp -- host-side M-step helper of the C ABI: the reversible maximum-likelihood
// transition matrix (the fixed-point iteration bhmm obtains from msmtools,
// bhmm/estimators/_tmatrix_disconnected.py:94-105 -> msmest.transition_matrix(reversible=True)).
// O(N^2) per iteration and thousands of iterations at maxerr = 1e-12: in numpy that is tens of
// milliseconds per EM iteration next to a 1 ms E-step.  Same arithmetic and iteration as
// bhmm_amd/estimators/_tmatrix.py:mle_reversible (the restatement the tests check it against).
// No device code here.
#include <math.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/bhmm_amd.h"

namespace bhmm {
int invalid_arg(const std::string &msg);
}

extern "C" int bhmm_mle_reversible(double *P, int64_t *iterations, const double *C, int n,
                                   int64_t maxiter, double maxerr)
{
    if (!P || !C || n < 1)
        return bhmm::invalid_arg("NULL argument or empty matrix");
    const size_t nn = (size_t)n * n;
    std::vector<double> C2(nn), X(nn), csum(n), xsum(n), q(n), xnew(n);
    double tot = 0.0;
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            C2[(size_t)i * n + j] = C[(size_t)i * n + j] + C[(size_t)j * n + i];
            s += C[(size_t)i * n + j];
        }
        csum[i] = s;
    }
    for (size_t e = 0; e < nn; ++e)
        tot += C2[e];
    if (!(tot > 0.0))
        return bhmm::invalid_arg("count matrix without counts");
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            X[(size_t)i * n + j] = C2[(size_t)i * n + j] / tot;
            s += X[(size_t)i * n + j];
        }
        xsum[i] = s;
    }
    int64_t it = 0;
    double err = 1.0;
    while (err > maxerr && it < maxiter) {
        for (int i = 0; i < n; ++i)
            q[i] = csum[i] / xsum[i];
        double total = 0.0;
        for (int i = 0; i < n; ++i) {
            double *xr = &X[(size_t)i * n];
            const double *cr = &C2[(size_t)i * n];
            for (int j = 0; j < n; ++j) {
                const double c = cr[j];
                const double v = c == 0.0 ? 0.0 : c / (q[i] + q[j]);
                xr[j] = v;
                total += v;
            }
        }
        err = 0.0;
        for (int i = 0; i < n; ++i) {
            double *xr = &X[(size_t)i * n];
            double s = 0.0;
            for (int j = 0; j < n; ++j) {
                xr[j] /= total;
                s += xr[j];
            }
            xnew[i] = s;
            const double d = fabs(s - xsum[i]);
            if (d > err || d != d)
                err = d;
        }
        xsum.swap(xnew);
        ++it;
    }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j)
            s += X[(size_t)i * n + j];
        for (int j = 0; j < n; ++j)
            P[(size_t)i * n + j] = X[(size_t)i * n + j] / s;
    }
    if (iterations)
        *iterations = it;
    return BHMM_OK;
}
