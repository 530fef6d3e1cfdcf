// asan_driver.cpp -- CPU sanitizer run ("race detection / sanitizers"; TEST
// INFRASTRUCTURE).  `make -C oracle asan` builds this file together with the product's host-only
// translation units (bhmm_amd/csrc/host_model.cpp, host_api.cpp, plan.hpp) and the oracle's C
// restatement with -fsanitize=address,undefined and runs it: ragged lengths, T = 1, 1e5 short
// trajectories, re-plan paths, structured / degenerate count matrices, all emission kinds.  Any
// sanitizer report aborts with a non-zero exit code (-fno-sanitize-recover).  GPU code is not
// covered (no GPU AddressSanitizer on this pool).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../bhmm_amd/csrc/host_model.hpp"
#include "../bhmm_amd/csrc/plan.hpp"
#include "../include/bhmm_amd.h"

namespace bhmm {
static std::string g_err;
int invalid_arg(const std::string &msg)
{
    g_err = msg;
    return BHMM_ERR_INVALID;
}
void set_error(const std::string &msg) { g_err = msg; }
} // namespace bhmm

extern "C" {
long orc_pobs_gaussian(const double *obs, long T, const double *mu, const double *sigma, int N,
                       int ignore_outliers, double *pobs);
void orc_pobs_discrete(const int32_t *obs, long T, const double *B, int N, int M, double *pobs);
double orc_forward(double *alpha, const double *A, const double *pobs, const double *pi, int N, long T);
void orc_backward(double *beta, const double *A, const double *pobs, int N, long T);
void orc_gamma(double *gamma, const double *alpha, const double *beta, int N, long T);
int orc_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                          const double *beta, int N, long T);
int orc_viterbi(int32_t *path, const double *A, const double *pobs, const double *pi, int N, long T);
int orc_sample_path(int32_t *path, const double *alpha, const double *A, int N, long T, const double *u);
void orc_path_counts(const int32_t *path, long T, int N, int64_t *C, int64_t *n0);
}

static uint64_t rng_state = 88172645463325252ull;
static double urand()
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0);
}

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            fprintf(stderr, "asan_driver: check failed at line %d: %s\n", __LINE__, #cond); \
            exit(3);                                                                 \
        }                                                                            \
    } while (0)

static void check_chunk_plan(const std::vector<int64_t> &off, int K, int N, int chunk, bool allow_mult)
{
    bhmm::plan::ChunkPlan p;
    const int64_t total = off[K] - off[0];
    CHECK(bhmm::plan::plan_chunks(off, K, N, total, chunk, allow_mult, 64, p));
    CHECK(p.Gp % 64 == 0 && p.Gp >= p.G && (int)p.ctraj.size() == p.Gp);
    int64_t covered = 0;
    for (int g = 0; g < p.G; ++g) {
        CHECK(p.clen[g] >= 1 && p.clen[g] <= p.Lmax);
        const int k = p.ctraj[g];
        CHECK(k >= 0 && k < K && p.cgoff[g] == off[k] + p.ct0[g]);
        CHECK(p.ct0[g] + p.clen[g] <= off[k + 1] - off[k]);
        covered += p.clen[g];
    }
    CHECK(covered == total);
    for (int k = 0; k < K; ++k) {
        CHECK(p.traj_c0[k] <= p.traj_c0[k + 1]);
        int64_t t = 0;
        for (int g = p.traj_c0[k]; g < p.traj_c0[k + 1]; ++g) { // chunks tile the trajectory in order
            CHECK(p.ct0[g] == t);
            t += p.clen[g];
        }
        CHECK(t == off[k + 1] - off[k]);
    }
    if (p.nG > 0) {
        CHECK((int)p.g0.size() == p.nG && (int)p.gt.size() == K + 1 && p.gt[K] == p.nG);
        for (int q = 0; q < p.nG; ++q)
            CHECK(p.g0[q] < p.g1[q] && p.g1[q] <= p.G);
    }
}

static void check_seg_plan(const std::vector<int64_t> &off, int K, int64_t seglen, int mult)
{
    bhmm::plan::SegPlan s;
    bhmm::plan::plan_segments(off, K, seglen, mult, s);
    int64_t covered = 0;
    for (size_t q = 0; q < s.traj.size(); ++q) {
        CHECK(s.len[q] >= 1);
        CHECK(s.t0[q] % 4 == 0); // starts at multiples of four
        covered += s.len[q];
    }
    CHECK(covered == off[K] - off[0]);
    CHECK((int)s.traj0.size() == K + 1 && s.traj0[K] == (int32_t)s.traj.size());
    // tiles of the row-batched kernels, per direction: every segment exactly once, 16 slots per tile, empty
    // slots only at the end of a class, and (inside a class) longest rows first
    for (int dir = 0; dir < 2; ++dir) {
        std::vector<int32_t> ts;
        bhmm::plan::plan_tiles(s, off, dir == 1, ts);
        CHECK(ts.size() % 16 == 0);
        std::vector<int> seen(s.traj.size(), 0);
        for (size_t q = 0; q < ts.size(); ++q)
            if (ts[q] >= 0) {
                CHECK((size_t)ts[q] < s.traj.size());
                seen[ts[q]]++;
            }
        for (size_t q = 0; q < s.traj.size(); ++q)
            CHECK(seen[q] == (s.len[q] > 0 ? 1 : 0));
        for (size_t q = 0; q + 1 < ts.size(); ++q) {
            if (ts[q] < 0 || ts[q + 1] < 0)
                continue;
            const int a = ts[q], b = ts[q + 1];
            const int64_t Ta = off[s.traj[a] + 1] - off[s.traj[a]], Tb = off[s.traj[b] + 1] - off[s.traj[b]];
            const bool ea = dir ? s.t0[a] + s.len[a] >= Ta : s.t0[a] == 0;
            const bool eb = dir ? s.t0[b] + s.len[b] >= Tb : s.t0[b] == 0;
            CHECK(ea || !eb);                      // rows without a warm-up come first
            if (ea == eb)
                CHECK(s.len[a] >= s.len[b]);       // longest first inside a class
        }
    }
    if (seglen > 0 && mult == 1) {
        std::vector<int64_t> mid;
        bhmm::plan::plan_forward_mids(off, K, seglen, mid);
        CHECK(mid.size() == s.traj.size());
        for (size_t q = 0; q < mid.size(); ++q)
            CHECK(mid[q] == -1 || (mid[q] > s.t0[q] && mid[q] < s.t0[q] + s.len[q]));
    }
}

static void planners()
{
    // ragged lengths incl. T = 1, T = 0 (skipped), one long trajectory
    {
        std::vector<int64_t> lens = {1, 7, 0, 100000, 33, 1, 2, 4097, 12345};
        std::vector<int64_t> off(lens.size() + 1, 0);
        for (size_t k = 0; k < lens.size(); ++k)
            off[k + 1] = off[k] + lens[k];
        for (int N : {2, 4, 8})
            for (int chunk : {0, 1, 32, 64, 1000, 1 << 20})
                check_chunk_plan(off, (int)lens.size(), N, chunk, true);
        for (int64_t seglen : {(int64_t)0, (int64_t)4, (int64_t)100, (int64_t)4096, (int64_t)1 << 40})
            for (int mult : {1, 2})
                check_seg_plan(off, (int)lens.size(), seglen, mult);
    }
    // 1e5 short trajectories
    {
        const int K = 100000;
        std::vector<int64_t> off(K + 1, 0);
        for (int k = 0; k < K; ++k)
            off[k + 1] = off[k] + 1 + (int64_t)(urand() * 40);
        check_chunk_plan(off, K, 8, 0, true);
        check_chunk_plan(off, K, 2, 16, true);
        check_seg_plan(off, K, 16, 1);
        check_seg_plan(off, K, 0, 1);
    }
    // very long chunks: the tripled / doubled plan and the re-plan back to the default count
    {
        const int K = 512;
        std::vector<int64_t> off(K + 1, 0);
        for (int k = 0; k < K; ++k)
            off[k + 1] = off[k] + 1190000;
        bhmm::plan::ChunkPlan fine, coarse;
        CHECK(bhmm::plan::plan_chunks(off, K, 8, off[K], 0, true, 64, fine));
        CHECK(bhmm::plan::plan_chunks(off, K, 8, off[K], 0, false, 64, coarse));
        CHECK(fine.chunk_mult > 1 && coarse.chunk_mult == 1 && coarse.L > fine.L && fine.G > coarse.G);
        check_chunk_plan(off, K, 8, 0, true);
        check_chunk_plan(off, K, 8, 0, false);
        check_seg_plan(off, K, 1190000 / 8, 1);
        check_seg_plan(off, K, 1190000 / 8, 2);
    }
    // a single trajectory of 2^31 + 5 steps (64-bit offsets)
    {
        std::vector<int64_t> off = {0, ((int64_t)1 << 31) + 5};
        check_chunk_plan(off, 1, 8, 0, true);
        check_seg_plan(off, 1, (int64_t)1 << 21, 2);
    }
}

static void host_model()
{
    using namespace bhmm::host;
    for (int trial = 0; trial < 600; ++trial) {
        const int n = 1 + (int)(urand() * 11);
        std::vector<double> C((size_t)n * n), P((size_t)n * n), pi(n);
        const double zero_p = trial % 4 == 0 ? 0.0 : 0.2 * (trial % 4);
        for (double &c : C)
            c = urand() < zero_p ? 0.0 : urand() * (trial % 3 == 0 ? 1e6 : 10.0);
        if (trial % 7 == 0) // an empty row and column
            for (int j = 0; j < n; ++j)
                C[j] = C[(size_t)j * n] = 0.0;
        int64_t its = 0;
        for (int rev = 0; rev < 2; ++rev) {
            CHECK(estimate_P(C.data(), n, rev != 0, nullptr, 20000, 1e-10, 1e-16, P.data(), &its) == 0);
            for (int i = 0; i < n; ++i) {
                double rs = 0.0;
                for (int j = 0; j < n; ++j)
                    rs += P[(size_t)i * n + j];
                CHECK(fabs(rs - 1.0) < 1e-9);
            }
        }
        double tot = 0.0;
        for (double c : C)
            tot += c;
        if (tot > 0.0) {
            stationary_distribution(P.data(), C.data(), n, 0.0, pi.data());
            double s = 0.0;
            for (double v : pi)
                s += v;
            CHECK(fabs(s - 1.0) < 1e-9);
        }
        (void)is_reversible(P.data(), n);
        std::vector<double> fixed(n);
        double ft = 0.0;
        for (double &v : fixed)
            ft += (v = urand() + 0.05);
        for (double &v : fixed)
            v /= ft;
        CHECK(estimate_P(C.data(), n, true, fixed.data(), 2000, 1e-10, 0.0, P.data(), &its) == 0);
        // one-call M-step, every kind
        const int M = 1 + (int)(urand() * 5);
        for (int kind = 0; kind < 3; ++kind) {
            const int esz = kind == 0 ? 2 * n : (kind == 1 ? n * M : 0);
            std::vector<double> stats(1 + n + (size_t)n * n + n + esz), Told((size_t)n * n, 1.0 / n),
                mu(n, 0.0), sg(n, 1.0), Tn((size_t)n * n), pin(n), p0((size_t)n * std::max(M, 1)), p1(n),
                warm(1 + n, 0.0);
            for (double &v : stats)
                v = urand() * 100 + 1.0;
            if (kind == 0)
                for (int i = 0; i < n; ++i) { // sum gamma d^2 >= (sum gamma d)^2 / sum gamma
                    const double w = stats[1 + n + (size_t)n * n + i];
                    const double m1 = urand() - 0.5;
                    stats[1 + 2 * n + (size_t)n * n + i] = m1 * w;
                    stats[1 + 3 * n + (size_t)n * n + i] = (m1 * m1 + 0.3) * w;
                }
            int32_t info[2];
            const int rc = bhmm_mstep(kind, n, M, stats.data(), Told.data(), mu.data(), sg.data(),
                                      trial % 3 - 1, trial % 2, trial % 5 == 0 ? fixed.data() : nullptr,
                                      5000, 1e-10, 1e-16, Tn.data(), pin.data(), p0.data(), p1.data(),
                                      info, warm.data());
            CHECK(rc == BHMM_OK);
            CHECK(bhmm_mstep(kind, n, M, stats.data(), Told.data(), mu.data(), sg.data(), 1, 0, nullptr,
                             5000, 1e-10, 1e-16, Tn.data(), pin.data(), p0.data(), p1.data(), info,
                             warm.data()) == BHMM_OK); // second call: warm start
        }
        // Gibbs parameter step, every kind, reversible or not
        for (int kind = 0; kind < 3; ++kind) {
            const int esz = kind == 0 ? 3 * n : (kind == 1 ? n * M : 0);
            std::vector<double> ps((size_t)n * n + n + esz), prior((size_t)n * n, 0.5), pn0(n, 0.25),
                T((size_t)n * n), p0(n), e0((size_t)n * std::max(M, 1), 1.0 / std::max(M, 1)), e1(n, 1.0);
            for (size_t e = 0; e < (size_t)n * n; ++e)
                ps[e] = C[e] == 0.0 ? 0.0 : floor(C[e]);
            for (int i = 0; i < n; ++i)
                ps[(size_t)n * n + i] = floor(urand() * 3);
            for (int e = 0; e < esz; ++e)
                ps[(size_t)n * n + n + e] = floor(urand() * 50);
            if (kind == 0)
                for (int i = 0; i < n; ++i)
                    ps[(size_t)n * n + n + 2 * n + i] += 60.0; // sum d^2 large enough
            int32_t info[1];
            for (int rev = 0; rev < 2; ++rev) {
                const int rc = bhmm_gibbs_parameters(kind, n, M, ps.data(), prior.data(), pn0.data(),
                                                     nullptr, rev, trial % 2, 5, 17, (uint64_t)trial,
                                                     T.data(), p0.data(), e0.data(), e1.data(), info);
                CHECK(rc == BHMM_OK);
                for (int i = 0; i < n; ++i) {
                    double rs = 0.0;
                    for (int j = 0; j < n; ++j)
                        rs += T[(size_t)i * n + j];
                    CHECK(fabs(rs - 1.0) < 1e-9);
                }
            }
        }
    }
    // error paths
    double C2[4] = {5, 0, 0, 5}, T2[4], p2[2], ps2[6] = {5, 0, 0, 5, 1, 1};
    CHECK(bhmm_gibbs_parameters(2, 2, 0, ps2, nullptr, nullptr, nullptr, 1, 0, 5, 1, 0, T2, p2, nullptr,
                                nullptr, nullptr) == BHMM_ERR_DISCONNECTED);
    CHECK(bhmm_mle_reversible(T2, nullptr, C2, 0, 10, 1e-8) == BHMM_ERR_INVALID);
    double out[64];
    for (int what = 0; what < 4; ++what)
        CHECK(bhmm_host_rng_draws(out, 64, what, 0.3, 5, 9) == BHMM_OK);
}

static void oracle_kernels()
{
    for (int trial = 0; trial < 60; ++trial) {
        const int N = 1 + (int)(urand() * 9);
        const long T = trial % 5 == 0 ? 1 : 1 + (long)(urand() * 300);
        const int M = 1 + (int)(urand() * 6);
        std::vector<double> A((size_t)N * N), pi(N), mu(N), sg(N), B((size_t)N * M), obs(T), u(T);
        std::vector<int32_t> sym(T), path(T);
        for (int i = 0; i < N; ++i) {
            double rs = 0.0;
            for (int j = 0; j < N; ++j)
                rs += (A[(size_t)i * N + j] = urand() + 0.01);
            for (int j = 0; j < N; ++j)
                A[(size_t)i * N + j] /= rs;
            pi[i] = 1.0 / N;
            mu[i] = i - 0.5 * N;
            sg[i] = 0.5 + urand();
            rs = 0.0;
            for (int k = 0; k < M; ++k)
                rs += (B[(size_t)i * M + k] = urand() + 0.01);
            for (int k = 0; k < M; ++k)
                B[(size_t)i * M + k] /= rs;
        }
        for (long t = 0; t < T; ++t) {
            obs[t] = (urand() - 0.5) * N;
            sym[t] = (int32_t)(urand() * M) % M;
            u[t] = urand();
        }
        std::vector<double> pobs((size_t)T * N), alpha((size_t)T * N), beta((size_t)T * N),
            gamma((size_t)T * N), Cc((size_t)N * N);
        if (trial % 2)
            orc_pobs_gaussian(obs.data(), T, mu.data(), sg.data(), N, 1, pobs.data());
        else
            orc_pobs_discrete(sym.data(), T, B.data(), N, M, pobs.data());
        const double ll = orc_forward(alpha.data(), A.data(), pobs.data(), pi.data(), N, T);
        CHECK(ll == ll);
        orc_backward(beta.data(), A.data(), pobs.data(), N, T);
        orc_gamma(gamma.data(), alpha.data(), beta.data(), N, T);
        CHECK(orc_transition_counts(Cc.data(), A.data(), pobs.data(), alpha.data(), beta.data(), N, T) == 0);
        CHECK(orc_viterbi(path.data(), A.data(), pobs.data(), pi.data(), N, T) == 0);
        CHECK(orc_sample_path(path.data(), alpha.data(), A.data(), N, T, u.data()) == 0);
        std::vector<int64_t> cnt((size_t)N * N), n0(N);
        orc_path_counts(path.data(), T, N, cnt.data(), n0.data());
    }
}

int main()
{
    planners();
    host_model();
    oracle_kernels();
    printf("asan_driver: ok (planners, host model, oracle kernels)\n");
    return 0;
}
