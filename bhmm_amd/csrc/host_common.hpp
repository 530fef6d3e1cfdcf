// host_common.hpp -- host helpers shared by the translation units of libbhmm_amd.so
#pragma once
#include <math.h>
#include <string.h>

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
}


inline int pad_states_pub(int n) { return pad_states(n); }
template <int N>
inline void fill_model_pub(Model<N> &m, int n, int kind, int M, const double *A, const double *pi,
                           const double *par0, const double *par1)
{
    fill_model<N>(m, n, kind, M, A, pi, par0, par1);
}

} // namespace bhmm
