// host_common.hpp -- host helpers shared by the translation units of libbhmm_amd.so
#pragma once
#include <math.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <limits>

#include "ctx.hpp"
#include "estep_kernels.hpp"

namespace bhmm {

inline int pad_states(int n) { return n <= 2 ? 2 : (n <= 4 ? 4 : 8); }

// Constants of gauss_pdf() (estep_sweep.hpp): p = 2^(s - 4096 u), u = a (o - mu)^2 + b, for n real
// states (npad >= n entries are written: padded states get a tiny a and b = 1, i.e. u >= 1, p = 0); computed in
// extended precision and rounded once.  An invalid sigma makes MG NaN (every density NaN).
inline void gauss_pdf_constants(int n, int npad, const double *sigma, double *a, double *b, double *MG)
{
    const long double sqrt2pi = sqrtl(2.0L * 3.14159265358979323846264338327950288L);
    bool valid = true;
    long double cmax = 0.0L;
    for (int i = 0; i < n; ++i) {
        const long double sg = sigma[i];
        valid = valid && sg > 0.0L && std::isfinite(sigma[i]) && std::isfinite((double)(1.0L / (sqrt2pi * sg)));
        if (valid)
            cmax = std::max(cmax, 1.0L / (sqrt2pi * sg));
    }
    const long double s = valid ? ceill(log2l(cmax)) : 0.0L;
    for (int i = 0; i < npad; ++i) {
        if (i < n && valid) {
            const long double sg = sigma[i];
            a[i] = (double)(1.44269504088896340735992468100189214L / (2.0L * sg * sg) / 4096.0L);
            b[i] = (double)((s - log2l(1.0L / (sqrt2pi * sg))) / 4096.0L);
        } else {
            a[i] = 1e-300; // (not 0: an infinite observation must give u = inf, not inf * 0 = NaN)
            b[i] = 1.0;
        }
    }
    *MG = valid ? (double)(1649267441664.0L + s / 4096.0L) : std::numeric_limits<double>::quiet_NaN();
}

// ---- model marshalling -----------------------------------------------------------------
template <int N>
inline void fill_model(Model<N> &m, int n, int kind, int M, const double *A, const double *pi,
                       const double *par0, const double *par1)
{
    memset(&m, 0, sizeof(m));
    m.nreal = n;
    m.M = M;
    m.dcopies = 1;
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j)
            m.A[i * N + j] = (i < n && j < n) ? A[i * n + j] : (i == j ? 1.0 : 0.0);
    for (int i = 0; i < n; ++i)
        m.pi[i] = pi ? pi[i] : 0.0;
    if (kind == EMIT_GAUSS)
        for (int i = 0; i < n; ++i) {
            m.e0[i] = par0[i];
            m.e1[i] = 1.0 / par1[i];
            m.e2[i] = 1.0 / (sqrt(2.0 * M_PI) * par1[i]); // _gaussian.c:18
            m.e3[i] = par1[i];
        }
    if (kind == EMIT_GAUSS)
        gauss_pdf_constants(n, N, par1, m.e4, m.e5, &m.emg);
}


inline int pad_states_pub(int n) { return pad_states(n); }
template <int N>
inline void fill_model_pub(Model<N> &m, int n, int kind, int M, const double *A, const double *pi,
                           const double *par0, const double *par1)
{
    fill_model<N>(m, n, kind, M, A, pi, par0, par1);
}

} // namespace bhmm
