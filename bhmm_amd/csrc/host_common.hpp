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
    if (kind == EMIT_GAUSS) {
        // gauss_pdf() (estep_sweep.hpp): p = 2^(s - 4096 u), u = e4 (o - mu)^2 + e5 -- the constants in
        // extended precision, rounded once
        bool valid = true;
        long double cmax = 0.0L;
        for (int i = 0; i < n; ++i) {
            const long double sg = par1[i];
            valid = valid && sg > 0.0L && std::isfinite(par1[i]) && std::isfinite(m.e2[i]) &&
                    m.e2[i] > 0.0;
            if (valid)
                cmax = std::max(cmax, 1.0L / (sqrtl(2.0L * 3.14159265358979323846264338327950288L) * sg));
        }
        const long double s = valid ? ceill(log2l(cmax)) : 0.0L;
        for (int i = 0; i < N; ++i) {
            if (i < n && valid) {
                const long double sg = par1[i];
                const long double cn = 1.0L / (sqrtl(2.0L * 3.14159265358979323846264338327950288L) * sg);
                m.e4[i] = (double)(1.44269504088896340735992468100189214L / (2.0L * sg * sg) / 4096.0L);
                m.e5[i] = (double)((s - log2l(cn)) / 4096.0L);
            } else {
                m.e4[i] = 0.0;
                m.e5[i] = 1.0;
            }
        }
        m.emg = valid ? (double)(1649267441664.0L + s / 4096.0L) : std::numeric_limits<double>::quiet_NaN();
    }
}


inline int pad_states_pub(int n) { return pad_states(n); }
template <int N>
inline void fill_model_pub(Model<N> &m, int n, int kind, int M, const double *A, const double *pi,
                           const double *par0, const double *par1)
{
    fill_model<N>(m, n, kind, M, A, pi, par0, par1);
}

} // namespace bhmm
