// host_model.cpp -- see host_model.hpp.  Host-only C++ (no device code, no HIP headers).
//
// Parity status: the partially reversible iteration restates
// _tmatrix_disconnected.py:126-190 and is pinned by fixtures written from that function
// (tests/golden/host_tmatrix.npz).  The reversible maximum-likelihood fixed point, the estimator
// with a fixed stationary vector and the reversible posterior sampler live in the unvendored
// msmtools package in the reference ("parity unpinned"): they are restated from the publications
// cited at each function and checked through their defining properties.
#include "host_model.hpp"
#include "host_rev_sampler.hpp"

#include <math.h>

#include <algorithm>
#include <limits>
#include <numeric>

#include "../../include/bhmm_amd.h"

namespace bhmm {
namespace host {

// ---- connected sets ------------------------------------------------------------------------
Sets connected_sets(const double *C, int n, double mincount, bool strong)
{
    // reachability by Warshall's closure on bit rows: O(n^2 * n/64)
    const int W = (n + 63) / 64;
    std::vector<uint64_t> reach((size_t)n * W, 0);
    auto set = [&](int i, int j) { reach[(size_t)i * W + (j >> 6)] |= (uint64_t)1 << (j & 63); };
    auto get = [&](int i, int j) { return (reach[(size_t)i * W + (j >> 6)] >> (j & 63)) & 1u; };
    for (int i = 0; i < n; ++i) {
        set(i, i);
        for (int j = 0; j < n; ++j)
            if (C[(size_t)i * n + j] > mincount) {
                set(i, j);
                if (!strong)
                    set(j, i);
            }
    }
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i)
            if (get(i, k)) {
                uint64_t *ri = &reach[(size_t)i * W];
                const uint64_t *rk = &reach[(size_t)k * W];
                for (int w = 0; w < W; ++w)
                    ri[w] |= rk[w];
            }
    std::vector<int> label(n, -1);
    Sets sets;
    for (int i = 0; i < n; ++i) {
        if (label[i] >= 0)
            continue;
        std::vector<int> s;
        for (int j = i; j < n; ++j)
            if (label[j] < 0 && get(i, j) && get(j, i)) {
                label[j] = i;
                s.push_back(j);
            }
        sets.push_back(s);
    }
    std::stable_sort(sets.begin(), sets.end(), [](const std::vector<int> &a, const std::vector<int> &b) {
        if (a.size() != b.size())
            return a.size() > b.size();
        return a[0] < b[0];
    });
    return sets;
}

// ---- stationary vector ---------------------------------------------------------------------
void stationary_vector(const double *P, int n, double *pi)
{
    if (n == 1) {
        pi[0] = 1.0;
        return;
    }
    std::vector<double> A(P, P + (size_t)n * n);
    bool reducible = false;
    for (int k = n - 1; k >= 1 && !reducible; --k) {
        double s = 0.0;
        for (int j = 0; j < k; ++j)
            s += A[(size_t)k * n + j];
        if (!(s > 0.0)) {
            reducible = true;
            break;
        }
        for (int i = 0; i < k; ++i)
            A[(size_t)i * n + k] /= s;
        for (int i = 0; i < k; ++i) {
            const double f = A[(size_t)i * n + k];
            if (f == 0.0)
                continue;
            for (int j = 0; j < k; ++j)
                A[(size_t)i * n + j] += f * A[(size_t)k * n + j];
        }
    }
    if (!reducible) {
        pi[0] = 1.0;
        double tot = 1.0;
        for (int k = 1; k < n; ++k) {
            double s = 0.0;
            for (int i = 0; i < k; ++i)
                s += pi[i] * A[(size_t)i * n + k];
            pi[k] = s;
            tot += s;
        }
        for (int k = 0; k < n; ++k)
            pi[k] /= tot;
        return;
    }
    // reducible block: lazy power iteration (x <- (x + x P) / 2) from the uniform vector
    std::vector<double> x(n, 1.0 / n), y(n);
    for (int64_t it = 0; it < 200000; ++it) {
        for (int j = 0; j < n; ++j)
            y[j] = 0.5 * x[j];
        for (int i = 0; i < n; ++i) {
            const double xi = 0.5 * x[i];
            if (xi == 0.0)
                continue;
            const double *row = &P[(size_t)i * n];
            double rs = 0.0;
            for (int j = 0; j < n; ++j)
                rs += row[j];
            if (!(rs > 0.0)) {
                y[i] += xi;
                continue;
            }
            for (int j = 0; j < n; ++j)
                y[j] += xi * row[j] / rs;
        }
        double tot = 0.0, err = 0.0;
        for (int j = 0; j < n; ++j)
            tot += y[j];
        for (int j = 0; j < n; ++j) {
            y[j] /= tot;
            err = std::max(err, fabs(y[j] - x[j]));
        }
        x.swap(y);
        if (err < 1e-16)
            break;
    }
    for (int j = 0; j < n; ++j)
        pi[j] = x[j];
}

// ---- reversible maximum-likelihood estimators ------------------------------------------------
// Fixed point  x_ij <- (c_ij + c_ji) / (c_i / x_i + c_j / x_j),  P_ij = x_ij / x_i
// (Bowman et al. 2009; Prinz et al. 2011, Eq. 29-31; Trendelkamp-Schroer et al. 2015, Alg. 1).
int64_t mle_reversible(const double *C, int n, int64_t maxiter, double maxerr, double *P,
                       double *xsum_state)
{
    const size_t nn = (size_t)n * n;
    std::vector<double> C2(nn), X(nn), csum(n), xsum(n), q(n);
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
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            X[(size_t)i * n + j] = C2[(size_t)i * n + j] / tot;
            s += X[(size_t)i * n + j];
        }
        xsum[i] = s;
    }
    // The iteration's state is the vector of row sums.  A caller that solves a sequence of slowly
    // changing problems (EM: the counts move little from one iteration to the next) hands the
    // previous solution back in: same fixed point (it is unique for a strongly connected C), same
    // stopping rule, a fraction of the iterations.  xsum_state = [valid (0/1) | n row sums].
    if (xsum_state && xsum_state[0] == 1.0) {
        bool ok = true;
        double t = 0.0;
        for (int i = 0; i < n; ++i) {
            ok = ok && xsum_state[1 + i] > 0.0 && std::isfinite(xsum_state[1 + i]);
            t += xsum_state[1 + i];
        }
        if (ok)
            for (int i = 0; i < n; ++i)
                xsum[i] = xsum_state[1 + i] / t;
    }
    // g: one step of the fixed-point map on the row sums (X is left holding the matrix that belongs
    // to g(x), normalised to total mass one)
    auto step = [&](const std::vector<double> &x, std::vector<double> &gx) {
        for (int i = 0; i < n; ++i)
            q[i] = csum[i] / x[i];
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
        for (int i = 0; i < n; ++i) {
            double *xr = &X[(size_t)i * n];
            double s = 0.0;
            for (int j = 0; j < n; ++j) {
                xr[j] /= total;
                s += xr[j];
            }
            gx[i] = s;
        }
    };
    // Plain iteration x <- g(x).  (Anderson acceleration of this map was tried in round 3: a fifth
    // of the iterations on dense count matrices, but on sparse ones it converged -- residual 1e-14 --
    // to OTHER fixed points of the map with a far lower likelihood in 13 of 400 random cases, which
    // the plain iteration started in the interior never did.  Not worth 40 us per EM iteration.)
    std::vector<double> gx(n);
    int64_t it = 0;
    double err = 1.0;
    while (err > maxerr && it < maxiter) {
        step(xsum, gx);
        ++it;
        err = 0.0;
        for (int i = 0; i < n; ++i) {
            const double d = fabs(gx[i] - xsum[i]);
            if (d > err || d != d)
                err = d;
        }
        xsum.swap(gx);
    }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j)
            s += X[(size_t)i * n + j];
        for (int j = 0; j < n; ++j)
            P[(size_t)i * n + j] = X[(size_t)i * n + j] / s;
    }
    if (xsum_state) {
        xsum_state[0] = it > 0 ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i)
            xsum_state[1 + i] = xsum[i];
    }
    return it;
}

// Reversible MLE with a given stationary vector (Trendelkamp-Schroer & Noe, J. Chem. Phys. 138,
// 164113 (2013)): Lagrange-multiplier fixed point.
void mle_reversible_fixed_pi(const double *C, const double *pi, int n, int64_t maxiter,
                             double maxerr, double *P)
{
    const size_t nn = (size_t)n * n;
    std::vector<double> C2(nn), lam(n), lnew(n);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            C2[(size_t)i * n + j] = C[(size_t)i * n + j] + C[(size_t)j * n + i];
            s += C2[(size_t)i * n + j];
        }
        lam[i] = s == 0.0 ? 1.0 : 0.5 * s;
    }
    int64_t it = 0;
    double err = 1.0;
    while (err > maxerr && it < maxiter) {
        err = 0.0;
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int j = 0; j < n; ++j) {
                const double c = C2[(size_t)i * n + j];
                if (c > 0.0)
                    s += c * pi[j] * lam[i] / (lam[i] * pi[j] + lam[j] * pi[i]);
            }
            lnew[i] = s == 0.0 ? lam[i] : s;
            const double d = fabs(lnew[i] - lam[i]) / std::max(lam[i], 1e-300);
            if (d > err || d != d)
                err = d;
        }
        lam.swap(lnew);
        ++it;
    }
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            const double c = C2[(size_t)i * n + j];
            double v = 0.0;
            if (c > 0.0 && j != i)
                v = c * pi[j] / (lam[i] * pi[j] + lam[j] * pi[i]);
            P[(size_t)i * n + j] = v;
            s += v;
        }
        P[(size_t)i * n + i] = 1.0 - s;
    }
}

// _tmatrix_disconnected.py:126-190: the rows in S are reversible among themselves and keep their
// outgoing counts; writes rows S of the full matrix P.  Same operation order as the reference
// (sums over the S block first, then over the outgoing block).
int64_t partial_rev(const double *C, int n, const std::vector<char> &in_S, int64_t maxiter,
                    double maxerr, double *P)
{
    std::vector<int> S, O;
    for (int i = 0; i < n; ++i)
        (in_S[i] ? S : O).push_back(i);
    const int ns = (int)S.size(), no = (int)O.size();
    std::vector<double> ATA((size_t)ns * ns), B((size_t)ns * no), X((size_t)ns * ns),
        Y((size_t)ns * no), countsums(ns), rowsums(ns), d(ns), rnew(ns);
    for (int a = 0; a < ns; ++a) {
        double s = 0.0;
        for (int j = 0; j < n; ++j)
            s += C[(size_t)S[a] * n + j];
        countsums[a] = s;
        for (int b = 0; b < ns; ++b)
            ATA[(size_t)a * ns + b] = C[(size_t)S[a] * n + S[b]] + C[(size_t)S[b] * n + S[a]];
        for (int b = 0; b < no; ++b)
            B[(size_t)a * no + b] = C[(size_t)S[a] * n + O[b]];
    }
    auto normalise = [&]() {
        double tx = 0.0, ty = 0.0;
        for (double v : X)
            tx += v;
        for (double v : Y)
            ty += v;
        const double tot = tx + ty;
        for (double &v : X)
            v /= tot;
        for (double &v : Y)
            v /= tot;
    };
    auto sums = [&](std::vector<double> &out) {
        for (int a = 0; a < ns; ++a) {
            double sx = 0.0, sy = 0.0;
            for (int b = 0; b < ns; ++b)
                sx += X[(size_t)a * ns + b];
            for (int b = 0; b < no; ++b)
                sy += Y[(size_t)a * no + b];
            out[a] = sx + sy;
        }
    };
    for (size_t e = 0; e < X.size(); ++e)
        X[e] = 0.5 * ATA[e];
    Y = B;
    normalise();
    sums(rowsums);
    int64_t it = 0;
    double err = 1.0;
    while (err > maxerr && it < maxiter) {
        for (int a = 0; a < ns; ++a)
            d[a] = countsums[a] / rowsums[a];
        for (int a = 0; a < ns; ++a) {
            for (int b = 0; b < ns; ++b)
                X[(size_t)a * ns + b] = ATA[(size_t)a * ns + b] / (d[a] + d[b]);
            for (int b = 0; b < no; ++b)
                Y[(size_t)a * no + b] = B[(size_t)a * no + b] / d[a];
        }
        normalise();
        sums(rnew);
        err = 0.0;
        for (int a = 0; a < ns; ++a) {
            const double e = fabs(rnew[a] - rowsums[a]);
            if (e > err || e != e)
                err = e;
        }
        rowsums.swap(rnew);
        ++it;
    }
    for (int a = 0; a < ns; ++a) {
        double *row = &P[(size_t)S[a] * n];
        for (int b = 0; b < ns; ++b)
            row[S[b]] = X[(size_t)a * ns + b];
        for (int b = 0; b < no; ++b)
            row[O[b]] = Y[(size_t)a * no + b];
        double s = 0.0;
        for (int j = 0; j < n; ++j)
            s += row[j];
        for (int j = 0; j < n; ++j)
            row[j] /= s;
    }
    return it;
}

static void submatrix(const double *A, int n, const std::vector<int> &s, std::vector<double> &out)
{
    const int m = (int)s.size();
    out.resize((size_t)m * m);
    for (int a = 0; a < m; ++a)
        for (int b = 0; b < m; ++b)
            out[(size_t)a * m + b] = A[(size_t)s[a] * n + s[b]];
}

int estimate_P(const double *C, int n, bool reversible, const double *fixed_pi, int64_t maxiter,
               double maxerr, double mincount, double *P, int64_t *iterations, double *warm)
{
    bool warm_used = false;
    int64_t its = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            P[(size_t)i * n + j] = i == j ? 1.0 : 0.0;
    std::vector<double> Cs, Ps;
    if (reversible && !fixed_pi) {
        for (const std::vector<int> &s : connected_sets(C, n, mincount, true)) {
            std::vector<char> mask(n, 0);
            for (int i : s)
                mask[i] = 1;
            double out = 0.0;
            for (int i : s)
                for (int j = 0; j < n; ++j)
                    if (!mask[j])
                        out += C[(size_t)i * n + j];
            if (out > std::numeric_limits<double>::epsilon()) {
                its += partial_rev(C, n, mask, maxiter, maxerr, P);
            } else if (s.size() > 1) {
                const int m = (int)s.size();
                submatrix(C, n, s, Cs);
                Ps.resize((size_t)m * m);
                // (the warm start only applies while ALL states form one closed set)
                const bool whole = m == n && warm != nullptr;
                its += mle_reversible(Cs.data(), m, maxiter, maxerr, Ps.data(), whole ? warm : nullptr);
                warm_used = warm_used || whole;
                for (int a = 0; a < m; ++a)
                    for (int b = 0; b < m; ++b)
                        P[(size_t)s[a] * n + s[b]] = Ps[(size_t)a * m + b];
            }
        }
    } else {
        for (const std::vector<int> &s : connected_sets(C, n, mincount, false)) {
            const int m = (int)s.size();
            submatrix(C, n, s, Cs);
            Ps.assign((size_t)m * m, 0.0);
            if (!reversible) {
                // row-normalisation; an empty row gets C_ii = 1 (_tmatrix_disconnected.py:110-115)
                for (int a = 0; a < m; ++a) {
                    double rs = 0.0;
                    for (int b = 0; b < m; ++b)
                        rs += Cs[(size_t)a * m + b];
                    if (rs == 0.0) {
                        Cs[(size_t)a * m + a] = 1.0;
                        rs = 1.0;
                    }
                    for (int b = 0; b < m; ++b)
                        Ps[(size_t)a * m + b] = Cs[(size_t)a * m + b] / rs;
                }
            } else {
                std::vector<double> pis(m);
                double tot = 0.0;
                for (int a = 0; a < m; ++a)
                    tot += fixed_pi[s[a]];
                for (int a = 0; a < m; ++a)
                    pis[a] = fixed_pi[s[a]] / tot;
                mle_reversible_fixed_pi(Cs.data(), pis.data(), m, maxiter, maxerr, Ps.data());
            }
            for (int a = 0; a < m; ++a)
                for (int b = 0; b < m; ++b)
                    P[(size_t)s[a] * n + s[b]] = Ps[(size_t)a * m + b];
        }
    }
    if (warm && !warm_used)
        warm[0] = 0.0; // another structure this time: the stored solution does not carry over
    if (iterations)
        *iterations = its;
    return BHMM_OK;
}

void stationary_distribution(const double *P, const double *C, int n, double mincount, double *pi)
{
    double ctot = 0.0;
    for (size_t e = 0; e < (size_t)n * n; ++e)
        ctot += C[e];
    std::fill(pi, pi + n, 0.0);
    std::vector<double> Ps, ps;
    for (const std::vector<int> &s : connected_sets(C, n, mincount, false)) {
        const int m = (int)s.size();
        double w = 0.0;
        for (int i : s)
            for (int j = 0; j < n; ++j)
                w += C[(size_t)i * n + j];
        w /= ctot;
        submatrix(P, n, s, Ps);
        ps.resize(m);
        stationary_vector(Ps.data(), m, ps.data());
        for (int a = 0; a < m; ++a)
            pi[s[a]] = w * ps[a];
    }
    double tot = 0.0;
    for (int i = 0; i < n; ++i)
        tot += pi[i];
    for (int i = 0; i < n; ++i)
        pi[i] /= tot;
}

bool is_reversible(const double *P, int n)
{
    std::vector<double> Ps, pi;
    for (const std::vector<int> &s : connected_sets(P, n, 0.0, false)) {
        const int m = (int)s.size();
        submatrix(P, n, s, Ps);
        // is_transition_matrix: non-negative, rows sum to one (numpy.allclose: 1e-8 + 1e-5 |1|)
        for (int a = 0; a < m; ++a) {
            double rs = 0.0;
            for (int b = 0; b < m; ++b) {
                if (!(Ps[(size_t)a * m + b] >= -1e-10))
                    return false;
                rs += Ps[(size_t)a * m + b];
            }
            if (!(fabs(rs - 1.0) <= 1e-8 + 1e-5))
                return false;
        }
        pi.resize(m);
        stationary_vector(Ps.data(), m, pi.data());
        for (int a = 0; a < m; ++a)
            for (int b = 0; b < m; ++b) {
                const double x = pi[a] * Ps[(size_t)a * m + b], y = pi[b] * Ps[(size_t)b * m + a];
                if (!(fabs(x - y) <= 1e-8 + 1e-5 * fabs(y))) // numpy.allclose(X, X.T)
                    return false;
            }
    }
    return true;
}

// ---- random numbers --------------------------------------------------------------------------
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static const uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;

Rng::Rng(uint64_t seed, uint64_t stream) : key(mix64(mix64(seed) + GOLDEN * (stream + 1))) {}

uint64_t Rng::bits()
{
    ++ctr;
    return mix64(key + GOLDEN * ctr);
}
double Rng::u01() { return (double)(bits() >> 11) * (1.0 / 9007199254740992.0); }
double Rng::u01_open() { return ((double)(bits() >> 12) + 0.5) * (1.0 / 4503599627370496.0); }

// Standard normal by the ziggurat method (Marsaglia & Tsang 2000, in Doornik's 2005 formulation with
// 128 blocks): one 64-bit draw decides block and abscissa, 98.8 % of the draws return after one
// multiplication and one comparison.
namespace {
struct Ziggurat {
    static const int C = 128;
    double X[C + 1], R[C];
    Ziggurat()
    {
        const double r = 3.442619855899, v = 9.91256303526217e-3;
        double f = exp(-0.5 * r * r);
        X[0] = v / f;
        X[1] = r;
        X[C] = 0.0;
        for (int i = 2; i < C; ++i) {
            X[i] = sqrt(-2.0 * log(v / X[i - 1] + f));
            f = exp(-0.5 * X[i] * X[i]);
        }
        for (int i = 0; i < C; ++i)
            R[i] = X[i + 1] / X[i];
    }
};
const Ziggurat zig;
} // namespace

double Rng::normal()
{
    for (;;) {
        const uint64_t w = bits();
        const int i = (int)(w & 127);
        // 53 high bits -> u in (-1, 1)
        const double u = (double)(int64_t)(w >> 11) * (1.0 / 4503599627370496.0) - 1.0;
        if (fabs(u) < zig.R[i])
            return u * zig.X[i];
        if (i == 0) { // tail beyond r
            const double r = 3.442619855899;
            double x, y;
            do {
                x = log(u01_open()) / r;
                y = log(u01_open());
            } while (-2.0 * y < x * x);
            return u < 0.0 ? x - r : r - x;
        }
        const double x = u * zig.X[i];
        const double f0 = exp(-0.5 * (zig.X[i] * zig.X[i] - x * x));
        const double f1 = exp(-0.5 * (zig.X[i + 1] * zig.X[i + 1] - x * x));
        if (f1 + u01() * (f0 - f1) < 1.0)
            return x;
    }
}

double Rng::gamma(double k)
{
    if (!(k > 0.0))
        return 0.0;
    if (k < 1.0) {
        const double g = gamma(k + 1.0);
        return g * pow(u01_open(), 1.0 / k);
    }
    const double d = k - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        const double x = normal();
        double v = 1.0 + c * x;
        if (v <= 0.0)
            continue;
        v = v * v * v;
        const double u = u01_open();
        const double x2 = x * x;
        if (u < 1.0 - 0.0331 * x2 * x2)
            return d * v;
        if (log(u) < 0.5 * x2 + d * (1.0 - v + log(v)))
            return d * v;
    }
}

double Rng::beta(double a, double b)
{
    const double x = gamma(a), y = gamma(b);
    return x / (x + y);
}

void Rng::dirichlet(const double *alpha, int n, double *out)
{
    double tot = 0.0, atot = 0.0;
    std::vector<double> g(n, 0.0);
    for (int i = 0; i < n; ++i)
        if (alpha[i] > 0.0) {
            g[i] = gamma(alpha[i]);
            tot += g[i];
            atot += alpha[i];
        }
    if (tot > 0.0 && std::isfinite(tot)) {
        for (int i = 0; i < n; ++i)
            if (alpha[i] > 0.0)
                out[i] = g[i] / tot;
        return;
    }
    // every draw underflowed (all concentration parameters tiny): the limit distribution is a
    // vertex of the simplex, chosen with probability alpha_i / sum alpha
    double r = u01() * atot;
    int pick = -1;
    for (int i = 0; i < n; ++i)
        if (alpha[i] > 0.0) {
            pick = i;
            r -= alpha[i];
            if (r < 0.0)
                break;
        }
    for (int i = 0; i < n; ++i)
        if (alpha[i] > 0.0)
            out[i] = i == pick ? 1.0 : 0.0;
}

// ---- reversible posterior sampler ------------------------------------------------------------
// host_rev_sampler.hpp: the element updates of a round side by side in SIMD lanes, every update on a random
// stream of its own -- the same chain whatever the vector width.  `lanes` = 1 / 4 picks the instantiation
// (tests), 0 the widest the CPU runs.  The caller's generator gives the base of the streams (one draw).
void sample_reversible_sweeps_avx2(const double *C, int n, int64_t nsweeps, uint64_t base, double *X);

int reversible_sampler_lanes()
{
    static const int forced = getenv("BHMM_AMD_REV_LANES") ? atoi(getenv("BHMM_AMD_REV_LANES")) : 0;
    if (forced == 1)
        return 1;
#if defined(__x86_64__)
    static const bool avx = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return avx ? 4 : 1;
#else
    return 1;
#endif
}

void sample_reversible_sweeps_base(const double *C, int n, int64_t nsweeps, uint64_t base, double *X, int lanes)
{
    if (lanes == 0)
        lanes = reversible_sampler_lanes();
#if defined(__x86_64__)
    if (lanes == 4 && reversible_sampler_lanes() == 4) {
        sample_reversible_sweeps_avx2(C, n, nsweeps, base, X);
        return;
    }
#endif
    revs::sample_reversible_sweeps_v<revs::V1>(C, n, nsweeps, base, X);
}

void sample_reversible_sweeps(const double *C, int n, int64_t nsweeps, Rng &rng, double *X)
{
    sample_reversible_sweeps_base(C, n, nsweeps, rng.bits(), X, 0);
}

} // namespace host
} // namespace bhmm
