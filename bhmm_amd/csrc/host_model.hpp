// host_model.hpp -- host-side model arithmetic of the EM iteration / Gibbs sweep that surrounds the
// device passes: the O(n^2 .. n^3) work of bhmm/estimators/_tmatrix_disconnected.py and of the
// parameter draws of bhmm/estimators/bayesian_sampling.py:333-373.  Pure C++ (no HIP): the same
// translation units build under -fsanitize=address,undefined for the CPU sanitizer run
// (oracle/Makefile: `make asan`).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace bhmm {
int invalid_arg(const std::string &msg);
void set_error(const std::string &msg);

namespace host {

typedef std::vector<std::vector<int>> Sets;

// _tmatrix_disconnected.py:28-43: (strongly | weakly) connected sets of the graph C > mincount,
// sorted by decreasing size, ties by smallest member
Sets connected_sets(const double *C, int n, double mincount, bool strong);

// stationary vector of a stochastic matrix (Grassmann-Taksar-Heyman elimination: no subtractions;
// a reducible block falls back to a lazy power iteration from the uniform vector)
void stationary_vector(const double *P, int n, double *pi);

// the fixed point of bhmm_mle_reversible (host_mstep.cpp); returns the iteration count
// xsum_state (optional, 1 + n doubles, in/out): warm start / final state of the row sums
int64_t mle_reversible(const double *C, int n, int64_t maxiter, double maxerr, double *P,
                       double *xsum_state = nullptr);
// reversible MLE with a given stationary vector
void mle_reversible_fixed_pi(const double *C, const double *pi, int n, int64_t maxiter,
                             double maxerr, double *P);
// _tmatrix_disconnected.py:126-190: rows `in_S` of P (n x n)
int64_t partial_rev(const double *C, int n, const std::vector<char> &in_S, int64_t maxiter,
                    double maxerr, double *P);
// _tmatrix_disconnected.py:68-123.  fixed_pi may be NULL.  Returns BHMM_OK / error code.
// warm (optional, 1 + n doubles, in/out): state of the reversible fixed point carried from call to call
int estimate_P(const double *C, int n, bool reversible, const double *fixed_pi, int64_t maxiter,
               double maxerr, double mincount, double *P, int64_t *iterations,
               double *warm = nullptr);
// _tmatrix_disconnected.py:229-251 with a count matrix
void stationary_distribution(const double *P, const double *C, int n, double mincount, double *pi);
// _tmatrix_disconnected.py:213-226
bool is_reversible(const double *P, int n);

// Counter-based generator: draw i of stream `key` is the SplitMix64 finaliser of
// key + golden * (i + 1) -- the device's uniform01 (path_kernels.hpp) -- so a chain is a function of
// (seed, sweep) alone and a host restatement reproduces it.
struct Rng {
    uint64_t key;
    uint64_t ctr = 0;
    explicit Rng(uint64_t seed, uint64_t stream);
    uint64_t bits();
    double u01();          // [0, 1)
    double u01_open();     // (0, 1)
    double normal();       // ziggurat (Marsaglia & Tsang 2000)
    double gamma(double k);// Marsaglia & Tsang 2000; k < 1 by the u^(1/k) boost
    double beta(double a, double b);
    double chisquare(double df) { return 2.0 * gamma(0.5 * df); }
    void dirichlet(const double *alpha, int n, double *out); // entries with alpha <= 0 stay untouched
};

// Reversible transition-matrix posterior draw: Gibbs sampler on the symmetric flux matrix X
// (Trendelkamp-Schroer, Wu, Paul, Noe, J. Chem. Phys. 143, 174101 (2015), Sec. IV), prior x_ij^-1.
// X (n x n, symmetric, in/out) is advanced by `nsweeps` full sweeps over all element pairs.
void sample_reversible_sweeps(const double *C, int n, int64_t nsweeps, Rng &rng, double *X);
// ... with the base of the per-update random streams given; lanes: 0 = the widest instantiation the CPU runs,
// 1 = one lane, 4 = AVX2 (falls back to one lane where the CPU has none): the same result either way
void sample_reversible_sweeps_base(const double *C, int n, int64_t nsweeps, uint64_t base, double *X, int lanes);
int reversible_sampler_lanes();

} // namespace host
} // namespace bhmm
