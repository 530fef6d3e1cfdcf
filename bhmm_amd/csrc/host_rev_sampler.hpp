// host_rev_sampler.hpp -- the reversible transition-matrix posterior sampler of the Gibbs sweep
// (bayesian_sampling.py:341-360 -> msmtools sample_tmatrix, restated from Trendelkamp-Schroer, Wu, Paul,
// Noe, J. Chem. Phys. 143, 174101 (2015), Sec. IV; parity unpinned) in a form
// whose element updates run side by side in SIMD lanes.
//
// One sweep = every diagonal element from its Beta conditional, then every off-diagonal pair by an
// independence Metropolis step with a Gamma proposal, in round-robin order: n - 1 rounds of n / 2 pairs with
// pairwise disjoint indices -- the updates of a round touch disjoint rows, so they are independent and one
// round is ONE vector of updates (4 lanes with AVX2: all four pairs of a round at 8 states).  Every update
// has a random stream of its own, keyed by (base, sweep, slot): which numbers an update consumes -- the
// rejection loops of the Gamma proposals run per lane -- does not depend on what its neighbours in the
// vector do, so the chain is the same whatever the vector width; the one-lane instantiation (host_model.cpp)
// and the AVX2 instantiation (host_rev_avx2.cpp) are held to the same bits by tests/test_host_native.py.
// All arithmetic is IEEE (+ - * / sqrt fma) on both sides; log and exp are the polynomials below, not libm
// (except in the rare scalar side paths -- ziggurat wedge / tail, degenerate rows --, which are per lane).
//
// V supplies: W lanes of double (D), of uint64 (U), masks (M) and the operations used below.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace bhmm {
namespace host {
namespace revs {

static const uint64_t GOLDEN64 = 0x9E3779B97F4A7C15ull;
static inline uint64_t mix64s(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ziggurat tables of the standard normal (Marsaglia & Tsang 2000, 128 blocks): shared by every instantiation
struct ZigTables {
    double X[129], R[128];
    ZigTables()
    {
        const double r = 3.442619855899, v = 9.91256303526217e-3;
        double f = exp(-0.5 * r * r);
        X[0] = v / f;
        X[1] = r;
        X[128] = 0.0;
        for (int i = 2; i < 128; ++i) {
            X[i] = sqrt(-2.0 * log(v / X[i - 1] + f));
            f = exp(-0.5 * X[i] * X[i]);
        }
        for (int i = 0; i < 128; ++i)
            R[i] = X[i + 1] / X[i];
    }
};
inline const ZigTables &zig_tables()
{
    static const ZigTables t;
    return t;
}

// ---- one scalar lane stream (the side paths; also what the one-lane instantiation is made of) --------------
struct LaneRng {
    uint64_t key, ctr;
    uint64_t bits()
    {
        ++ctr;
        return mix64s(key + GOLDEN64 * ctr);
    }
    static double to01(uint64_t w) // [0, 1), 52 bits
    {
        const uint64_t b = (w >> 12) | 0x3FF0000000000000ull;
        double d;
        memcpy(&d, &b, 8);
        return d - 1.0;
    }
    double u01() { return to01(bits()); }
    double u01_open() { return to01(bits()) + 0x1p-53; } // (0, 1)
};

// the part of a normal draw that did not return on the fast path: block i, abscissa u (in [-1, 1))
inline double normal_slow(LaneRng &g, int i, double u)
{
    const ZigTables &z = zig_tables();
    for (;;) {
        if (i == 0) { // tail beyond r
            const double r = 3.442619855899;
            double x, y;
            do {
                x = log(g.u01_open()) / r;
                y = log(g.u01_open());
            } while (-2.0 * y < x * x);
            return u < 0.0 ? x - r : r - x;
        }
        const double x = u * z.X[i];
        const double f0 = exp(-0.5 * (z.X[i] * z.X[i] - x * x));
        const double f1 = exp(-0.5 * (z.X[i + 1] * z.X[i + 1] - x * x));
        if (f1 + g.u01() * (f0 - f1) < 1.0)
            return x;
        // rejected: a fresh draw, fast path included
        const uint64_t w = g.bits();
        i = (int)(w & 127);
        const uint64_t b = (w >> 12) | 0x4000000000000000ull;
        double d;
        memcpy(&d, &b, 8);
        u = d - 3.0;
        if (fabs(u) < z.R[i])
            return u * z.X[i];
    }
}

// ---- vector mathematics on V ---------------------------------------------------------------------------
// log(x) for positive normal x (fdlibm's algorithm: x = 2^k m, m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1),
// log m = 2 s + s z R(z)); lanes with other arguments give unspecified values and must be masked by the caller
template <class V>
inline typename V::D vlog(typename V::D x)
{
    typedef typename V::D D;
    typedef typename V::U U;
    const U bits = V::as_u(x);
    // m in [sqrt(1/2), sqrt(2)): add 0x95f64 << 32 to the high word trick, in 64-bit form
    const U adj = V::addu(bits, V::setu(0x00095F6400000000ull));
    const U ku = V::srlu(adj, 52);
    const D k = V::sub(V::u2d_small(ku), V::set1(1023.0)); // (exponent + 1023) - 1023
    // mantissa with the exponent of m: clear exponent, set to 0x3ff or 0x3fe according to adj's bit 52
    const U mant = V::andu(bits, V::setu(0x000FFFFFFFFFFFFFull));
    const U carry = V::andu(V::srlu(V::addu(mant, V::setu(0x00095F6400000000ull)), 52), V::setu(1)); // 1: m < 1 branch
    const U ebits = V::subu(V::setu(0x3FF0000000000000ull), V::sllu(carry, 52));
    const D m = V::as_d(V::oru(mant, ebits));
    const D f = V::sub(m, V::set1(1.0));
    const D s = V::div(f, V::add(V::set1(2.0), f));
    const D z = V::mul(s, s), w = V::mul(z, z);
    const D t1 = V::mul(w, V::fma(w, V::fma(w, V::set1(1.531383769920937332e-01), V::set1(2.222219843214978396e-01)),
                                  V::set1(3.999999999940941908e-01)));
    const D t2 = V::mul(z, V::fma(w, V::fma(w, V::fma(w, V::set1(1.479819860511658591e-01), V::set1(1.818357216161805012e-01)),
                                             V::set1(2.857142874366239149e-01)),
                                  V::set1(6.666666666666735130e-01)));
    const D R = V::add(t2, t1);
    const D hfsq = V::mul(V::set1(0.5), V::mul(f, f));
    // k ln2_hi - ((hfsq - (s (hfsq + R) + k ln2_lo)) - f)
    const D inner = V::fma(s, V::add(hfsq, R), V::mul(k, V::set1(1.90821492927058770002e-10)));
    return V::sub(V::mul(k, V::set1(6.93147180369123816490e-01)), V::sub(V::sub(hfsq, inner), f));
}

// log(1 + x), x > -1: log(u) + (x - (u - 1)) / u with u = 1 + x rounded
template <class V>
inline typename V::D vlog1p(typename V::D x)
{
    typedef typename V::D D;
    const D u = V::add(V::set1(1.0), x);
    const D c = V::div(V::sub(x, V::sub(u, V::set1(1.0))), u);
    return V::add(vlog<V>(u), c);
}

// log(1 + x) for |x| < 2^-6 by its series to x^10 (truncation < 1e-21): no division, no range reduction
template <class V>
inline typename V::D vlog1p_small(typename V::D x)
{
    typedef typename V::D D;
    D q = V::set1(-1.0 / 10.0);
    q = V::fma(q, x, V::set1(1.0 / 9.0));
    q = V::fma(q, x, V::set1(-1.0 / 8.0));
    q = V::fma(q, x, V::set1(1.0 / 7.0));
    q = V::fma(q, x, V::set1(-1.0 / 6.0));
    q = V::fma(q, x, V::set1(1.0 / 5.0));
    q = V::fma(q, x, V::set1(-1.0 / 4.0));
    q = V::fma(q, x, V::set1(1.0 / 3.0));
    q = V::fma(q, x, V::set1(-1.0 / 2.0));
    q = V::fma(q, x, V::set1(1.0));
    return V::mul(q, x);
}

// exp(x) for x <= 0 (clamped at -700): x = k ln2 + r, the degree-11 polynomial of estep_sweep.hpp (exp_nonpos)
template <class V>
inline typename V::D vexp_nonpos(typename V::D x)
{
    typedef typename V::D D;
    typedef typename V::U U;
    x = V::max(x, V::set1(-700.0));
    const D k = V::rint(V::mul(x, V::set1(0x1.71547652b82fep+0)));
    D r = V::fma(k, V::set1(-0x1.62e42fefa39efp-1), x);
    r = V::fma(k, V::set1(-0x1.abc9e3b39803fp-56), r);
    D q = V::set1(0x1.ad7e38e167506p-26);
    q = V::fma(q, r, V::set1(0x1.28ae7908135d8p-22));
    q = V::fma(q, r, V::set1(0x1.71df27c33abefp-19));
    q = V::fma(q, r, V::set1(0x1.a01998fd42e01p-16));
    q = V::fma(q, r, V::set1(0x1.a01a012882c92p-13));
    q = V::fma(q, r, V::set1(0x1.6c16c184889e3p-10));
    q = V::fma(q, r, V::set1(0x1.111111112836cp-7));
    q = V::fma(q, r, V::set1(0x1.55555555506eap-5));
    q = V::fma(q, r, V::set1(0x1.55555555554f7p-3));
    q = V::fma(q, r, V::set1(0x1.000000000000ap-1));
    q = V::fma(q, r, V::set1(1.0));
    q = V::fma(q, r, V::set1(1.0));
    // 2^k, k in [-1010, 0]: exponent field k + 1023 (k + 1023 as a double, its integer through the 2^52 trick)
    const U e = V::andu(V::as_u(V::add(V::add(k, V::set1(1023.0)), V::set1(4503599627370496.0))), V::setu(0x7FFull));
    return V::mul(q, V::as_d(V::sllu(e, 52)));
}

// ---- W random streams -------------------------------------------------------------------------------
template <class V>
struct Streams {
    typename V::U key, ctr;
    // one more 64-bit draw for the lanes of `m` (the others keep their position)
    typename V::U bits(typename V::M m)
    {
        ctr = V::addu(ctr, V::andu(V::mask_u(m), V::setu(1)));
        return V::mix64(V::addu(key, V::mulu_const(ctr, GOLDEN64)));
    }
    static typename V::D to01(typename V::U w)
    {
        return V::sub(V::as_d(V::oru(V::srlu(w, 12), V::setu(0x3FF0000000000000ull))), V::set1(1.0));
    }
    LaneRng lane(int l) const { return LaneRng{V::lane_u(key, l), V::lane_u(ctr, l)}; }
    void put(int l, const LaneRng &g) { ctr = V::set_lane_u(ctr, l, g.ctr); }
};

template <class V>
inline typename V::D vnormal(Streams<V> &st, typename V::M need)
{
    typedef typename V::D D;
    typedef typename V::U U;
    typedef typename V::M M;
    const ZigTables &z = zig_tables();
    const U w = st.bits(need);
    const U idx = V::andu(w, V::setu(127));
    const D u = V::sub(V::as_d(V::oru(V::srlu(w, 12), V::setu(0x4000000000000000ull))), V::set1(3.0));
    const D Ri = V::gather(z.R, idx), Xi = V::gather(z.X, idx);
    D x = V::mul(u, Xi);
    const M fast = V::lt(V::abs(u), Ri);
    const M slow = V::andm(need, V::notm(fast));
    if (V::any(slow))
        for (int l = 0; l < V::W; ++l)
            if (V::lane_m(slow, l)) {
                LaneRng g = st.lane(l);
                x = V::set_lane_d(x, l, normal_slow(g, (int)V::lane_u(idx, l), V::lane_d(u, l)));
                st.put(l, g);
            }
    return x;
}

// Gamma(k, 1) for the lanes of `need` (k > 0 there); Marsaglia & Tsang 2000, k < 1 by the u^(1/k) boost
template <class V>
inline typename V::D vgamma(Streams<V> &st, typename V::D k, typename V::M need)
{
    typedef typename V::D D;
    typedef typename V::M M;
    const M small = V::andm(need, V::lt(k, V::set1(1.0)));
    const D kk = V::blend(small, V::add(k, V::set1(1.0)), k);
    const D d = V::sub(kk, V::set1(1.0 / 3.0));
    const D c = V::div(V::set1(1.0), V::sqrt(V::mul(V::set1(9.0), d)));
    D res = V::set1(0.0);
    M pending = need;
    while (V::any(pending)) {
        const D x = vnormal<V>(st, pending);
        const D v1 = V::fma(c, x, V::set1(1.0));
        const M vpos = V::gt(v1, V::set1(0.0));
        const D v = V::mul(V::mul(v1, v1), v1);
        const D u = V::add(Streams<V>::to01(st.bits(pending)), V::set1(0x1p-53));
        const D x2 = V::mul(x, x);
        M acc = V::lt(u, V::fma(V::mul(V::set1(-0.0331), x2), x2, V::set1(1.0)));
        const M undecided = V::andm(V::andm(pending, vpos), V::notm(acc));
        if (V::any(undecided)) {
            const D vs = V::blend(vpos, v, V::set1(1.0)); // (masked lanes: a harmless argument)
            const D rhs = V::fma(d, V::add(V::sub(V::set1(1.0), vs), vlog<V>(vs)), V::mul(V::set1(0.5), x2));
            acc = V::orm(acc, V::lt(vlog<V>(u), rhs));
        }
        acc = V::andm(V::andm(acc, vpos), pending);
        res = V::blend(acc, V::mul(d, v), res);
        pending = V::andm(pending, V::notm(acc));
    }
    if (V::any(small)) { // g * u^(1/k), u in (0, 1): exp(log(u) / k)
        const D u = V::add(Streams<V>::to01(st.bits(small)), V::set1(0x1p-53));
        const D ks = V::blend(small, k, V::set1(1.0));
        res = V::blend(small, V::mul(res, vexp_nonpos<V>(V::div(vlog<V>(u), ks))), res);
    }
    return res;
}

template <class V>
inline typename V::M vpositive(typename V::D x) // x > 1e-300 and finite
{
    return V::andm(V::gt(x, V::set1(1e-300)), V::lt(x, V::set1(1.7976931348623157e308)));
}

// scalar side path of an off-diagonal update: no Gamma proposal exists (degenerate rows): log-uniform random walk
inline double offdiag_random_walk(LaneRng &g, double v0, double v1, double v2, double c0, double c1, double c2)
{
    if (v0 == 0.0)
        return v0;
    const double step = g.u01() - 0.5;
    const double vn = v0 * exp(step);
    if (vn > 1e-300 && std::isfinite(vn)) {
        const double dl = c0 * step - c1 * log1p((vn - v0) / (v0 + v1)) - c2 * log1p((vn - v0) / (v0 + v2));
        if (dl >= 0.0 || g.u01() < exp(dl))
            v0 = vn;
    }
    return v0;
}

// One vector of off-diagonal updates: target density of v = x_ij (i != j) given everything else,
//   f(v) ~ v^(c0 - 1) (v + v1)^(-c1) (v + v2)^(-c2),   c0 = c_ij + c_ji, c1 = c_i, c2 = c_j,
// an independence Metropolis step with a Gamma(k, theta) proposal matched to the maximum of v f(v) and to its
// curvature there (Sec. IV C of the paper).  Returns the new values for the lanes of `act`.
template <class V>
inline typename V::D voffdiag(Streams<V> &st, typename V::D v0, typename V::D v1, typename V::D v2, typename V::D c0,
                              typename V::D c1, typename V::D c2, typename V::M act)
{
    typedef typename V::D D;
    typedef typename V::M M;
    const D one = V::set1(1.0);
    // What does not depend on the proposal is taken off the chain of dependent operations (the latency of one
    // round, seven times per sweep, is what this sampler costs): the uniform of the acceptance test and its
    // logarithm -- the test is log u < dl --, the reciprocals of the old value's terms, 1 / (2 a).
    const M v0z = V::eq(v0, V::set1(0.0));
    const D v0s = V::blend(v0z, one, v0);
    const D lu = vlog<V>(V::add(Streams<V>::to01(st.bits(act)), V::set1(0x1p-53))); // log of a uniform in (0, 1)
    const D rv0 = V::div(one, v0s), rv1 = V::div(one, V::add(v0s, v1)), rv2 = V::div(one, V::add(v0s, v2));
    const D a = V::sub(V::add(c1, c2), c0);
    const D inv2a = V::div(V::set1(0.5), a);
    const D b = V::fma(V::sub(c1, c0), v2, V::mul(V::sub(c2, c0), v1));
    const D c = V::mul(V::mul(V::sub(V::set1(0.0), c0), v1), v2);
    const D disc = V::sub(V::mul(b, b), V::mul(V::mul(V::set1(4.0), a), c));
    const D vbar = V::mul(V::sub(V::sqrt(V::max(disc, V::set1(0.0))), b), inv2a);
    M good = V::andm(V::andm(act, vpositive<V>(vbar)), V::ge(disc, V::set1(0.0)));
    const D vb = V::blend(good, vbar, one);
    const D r0 = V::div(one, vb), r1 = V::div(one, V::add(vb, v1)), r2 = V::div(one, V::add(vb, v2));
    const D h = V::sub(V::fma(V::mul(c1, r1), r1, V::mul(V::mul(c2, r2), r2)), V::mul(V::mul(c0, r0), r0));
    const D k = V::mul(V::mul(V::sub(V::set1(0.0), h), vb), vb), ith = V::mul(V::sub(V::set1(0.0), h), vb);
    // (k, 1 / theta and theta positive and finite: theta = 1 / ith > 1e-300  <=>  ith < 1e300)
    good = V::andm(good, V::andm(V::andm(vpositive<V>(k), vpositive<V>(ith)), V::lt(ith, V::set1(1e300))));
    D out = v0;
    if (V::any(good)) {
        const D ks = V::blend(good, k, one), is = V::blend(good, ith, one);
        const D theta = V::div(one, is); // (beside the Gamma draw, not behind it)
        const D vn = V::mul(vgamma<V>(st, ks, good), theta);
        const M ok = V::andm(good, vpositive<V>(vn));
        // log [f(vn) / q(vn)] - log [f(v0) / q(v0)],  q(v) ~ v^(k-1) exp(-v / theta)
        const D vns = V::blend(ok, vn, one);
        const D dv = V::sub(vns, v0s);
        // relative changes x = dv / (v0 + .): with the counts of a long trajectory the proposal sits within a per
        // mille of the old value and log(1 + x) is its series; a lane with a larger step takes the full logarithm of
        // the ratio -- chosen per lane by its own x, so the result does not depend on the neighbours in the vector
        const D x0 = V::mul(dv, rv0), x1 = V::mul(dv, rv1), x2 = V::mul(dv, rv2);
        const D lim = V::set1(0x1p-6);
        const M small = V::andm(V::andm(V::lt(V::abs(x0), lim), V::lt(V::abs(x1), lim)), V::lt(V::abs(x2), lim));
        D g0 = vlog1p_small<V>(x0), g1 = vlog1p_small<V>(x1), g2 = vlog1p_small<V>(x2);
        const M large = V::andm(V::andm(ok, V::notm(v0z)), V::notm(small));
        if (V::any(large)) {
            g0 = V::blend(large, vlog<V>(V::mul(vns, rv0)), g0);
            g1 = V::blend(large, vlog<V>(V::mul(V::add(vns, v1), rv1)), g1);
            g2 = V::blend(large, vlog<V>(V::mul(V::add(vns, v2), rv2)), g2);
        }
        D dl = V::mul(V::sub(c0, ks), g0);
        dl = V::sub(dl, V::mul(c1, g1));
        dl = V::sub(dl, V::mul(c2, g2));
        dl = V::fma(dv, is, dl);
        // accept with probability min(1, exp(dl)): log u < dl (an element that was zero takes the proposal)
        const M accept = V::orm(V::lt(lu, dl), v0z);
        out = V::blend(V::andm(ok, accept), vn, v0);
    }
    const M rw = V::andm(act, V::notm(good));
    if (V::any(rw))
        for (int l = 0; l < V::W; ++l)
            if (V::lane_m(rw, l)) {
                LaneRng g = st.lane(l);
                out = V::set_lane_d(out, l, offdiag_random_walk(g, V::lane_d(v0, l), V::lane_d(v1, l), V::lane_d(v2, l),
                                                                V::lane_d(c0, l), V::lane_d(c1, l), V::lane_d(c2, l)));
                st.put(l, g);
            }
    return out;
}

struct Pair {
    int i, j;
    double c0;
};

// X (n x n, symmetric, in/out) is advanced by `nsweeps` full sweeps.  base: one 64-bit draw of the caller's
// generator -- update `slot` of sweep s draws from the stream keyed mix64(base + golden (s * slots + slot + 1)).
template <class V>
void sample_reversible_sweeps_v(const double *C, int n, int64_t nsweeps, uint64_t base, double *X)
{
    typedef typename V::D D;
    typedef typename V::M M;
    const int W = V::W;
    std::vector<double> csum(n), rs(n);
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j)
            s += C[(size_t)i * n + j];
        csum[i] = s;
    }
    // round-robin ("circle method") scan: rounds of pairs with pairwise disjoint indices
    std::vector<std::vector<Pair>> rounds;
    const int m = n + (n & 1);
    int npairs = 0;
    for (int r = 0; r + 1 < m; ++r) {
        std::vector<Pair> rd;
        for (int k = 0; k < m / 2; ++k) {
            int a = k == 0 ? m - 1 : (r + k) % (m - 1);
            int b = k == 0 ? r : (r - k + (m - 1)) % (m - 1);
            if (a >= n || b >= n)
                continue;
            if (a < b)
                std::swap(a, b);
            const double c0 = C[(size_t)a * n + b] + C[(size_t)b * n + a];
            if (c0 > 0.0)
                rd.push_back(Pair{a, b, c0});
        }
        npairs += (int)rd.size();
        rounds.push_back(rd);
    }
    const uint64_t slots = (uint64_t)n + (uint64_t)npairs;
    auto rowsums = [&]() {
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int j = 0; j < n; ++j)
                s += X[(size_t)i * n + j];
            rs[i] = s;
        }
    };
    rowsums();
    double lv0[8], lv1[8], lv2[8], lc0[8], lc1[8], lc2[8], lo[8];
    uint64_t lkey[8];
    bool lact[8];
    for (int64_t sweep = 0; sweep < nsweeps; ++sweep) {
        const uint64_t sbase = base + GOLDEN64 * ((uint64_t)sweep * slots);
        // diagonal elements: x_ii / x_i ~ Beta(c_ii, c_i - c_ii) given the rest of the row
        for (int i0 = 0; i0 < n; i0 += W) {
            for (int l = 0; l < W; ++l) {
                const int i = i0 + l;
                const double cii = i < n ? C[(size_t)i * n + i] : 0.0;
                const double rest = i < n ? csum[i] - cii : 0.0;
                lact[l] = i < n && cii > 1e-300 && std::isfinite(cii) && rest > 1e-300 && std::isfinite(rest);
                lc0[l] = lact[l] ? cii : 1.0;
                lc1[l] = lact[l] ? rest : 1.0;
                lkey[l] = mix64s(sbase + GOLDEN64 * ((uint64_t)(i < n ? i : 0) + 1));
            }
            const M act = V::load_m(lact);
            if (!V::any(act))
                continue;
            Streams<V> st{V::load_u(lkey), V::setu(0)};
            const D gx = vgamma<V>(st, V::load_d(lc0), act);
            const D gy = vgamma<V>(st, V::load_d(lc1), act);
            V::store_d(lo, V::div(gx, V::add(gx, gy)));
            for (int l = 0; l < W; ++l)
                if (lact[l]) {
                    const int i = i0 + l;
                    const double t = lo[l];
                    const double rest = rs[i] - X[(size_t)i * n + i];
                    const double x = t / (1.0 - t) * rest;
                    if (x > 1e-300 && std::isfinite(x)) {
                        X[(size_t)i * n + i] = x;
                        rs[i] = rest + x;
                    }
                }
        }
        // off-diagonal pairs, round by round
        uint64_t slot = (uint64_t)n;
        for (const std::vector<Pair> &rd : rounds)
            for (size_t p0 = 0; p0 < rd.size(); p0 += W) {
                for (int l = 0; l < W; ++l) {
                    const bool in = p0 + l < rd.size();
                    lact[l] = in;
                    if (in) {
                        const Pair &p = rd[p0 + l];
                        const double x0 = X[(size_t)p.i * n + p.j];
                        lv0[l] = x0;
                        lv1[l] = rs[p.i] - x0;
                        lv2[l] = rs[p.j] - x0;
                        lc0[l] = p.c0;
                        lc1[l] = csum[p.i];
                        lc2[l] = csum[p.j];
                    } else {
                        lv0[l] = lv1[l] = lv2[l] = 1.0;
                        lc0[l] = 1.0;
                        lc1[l] = lc2[l] = 2.0;
                    }
                    lkey[l] = mix64s(sbase + GOLDEN64 * (slot + (uint64_t)l + 1));
                }
                Streams<V> st{V::load_u(lkey), V::setu(0)};
                const D xn = voffdiag<V>(st, V::load_d(lv0), V::load_d(lv1), V::load_d(lv2), V::load_d(lc0), V::load_d(lc1),
                                         V::load_d(lc2), V::load_m(lact));
                V::store_d(lo, xn);
                for (int l = 0; l < W; ++l)
                    if (lact[l]) {
                        const Pair &p = rd[p0 + l];
                        X[(size_t)p.i * n + p.j] = X[(size_t)p.j * n + p.i] = lo[l];
                        rs[p.i] = lv1[l] + lo[l];
                        rs[p.j] = lv2[l] + lo[l];
                    }
                slot += (uint64_t)std::min<size_t>(W, rd.size() - p0);
            }
        // The conditionals are covariant under a common factor of X (Beta and Gamma-proposal steps alike), so the
        // normalisation sum X = 1 is only bookkeeping against drift: every 16th sweep and at the end (n^2 divisions and
        // the row sums again were 4 % of a sweep at 8 states); in between the running row sums stand.
        if ((sweep & 15) == 15 || sweep + 1 == nsweeps) {
            double tot = 0.0;
            for (size_t e = 0; e < (size_t)n * n; ++e)
                tot += X[e];
            for (size_t e = 0; e < (size_t)n * n; ++e)
                X[e] /= tot;
            rowsums();
        }
    }
}

// ---- the one-lane instantiation's V ----------------------------------------------------------------
struct V1 {
    static const int W = 1;
    typedef double D;
    typedef uint64_t U;
    typedef bool M;
    static D set1(double x) { return x; }
    static U setu(uint64_t x) { return x; }
    static D add(D a, D b) { return a + b; }
    static D sub(D a, D b) { return a - b; }
    static D mul(D a, D b) { return a * b; }
    static D div(D a, D b) { return a / b; }
    static D sqrt(D a) { return ::sqrt(a); }
    static D fma(D a, D b, D c) { return ::fma(a, b, c); }
    static D max(D a, D b) { return a > b ? a : b; } // (second operand on unordered, like vmaxpd)
    static D min(D a, D b) { return a < b ? a : b; }
    static D abs(D a) { return ::fabs(a); }
    static D rint(D a) { return ::nearbyint(a); }
    static M lt(D a, D b) { return a < b; }
    static M gt(D a, D b) { return a > b; }
    static M ge(D a, D b) { return a >= b; }
    static M eq(D a, D b) { return a == b; }
    static M andm(M a, M b) { return a && b; }
    static M orm(M a, M b) { return a || b; }
    static M notm(M a) { return !a; }
    static bool any(M a) { return a; }
    static D blend(M m, D a, D b) { return m ? a : b; }
    static U as_u(D a) { U u; memcpy(&u, &a, 8); return u; }
    static D as_d(U a) { D d; memcpy(&d, &a, 8); return d; }
    static U addu(U a, U b) { return a + b; }
    static U subu(U a, U b) { return a - b; }
    static U andu(U a, U b) { return a & b; }
    static U oru(U a, U b) { return a | b; }
    static U srlu(U a, int s) { return a >> s; }
    static U sllu(U a, int s) { return a << s; }
    static U mask_u(M m) { return m ? ~0ull : 0ull; }
    static U mulu_const(U a, uint64_t c) { return a * c; }
    static U mix64(U z) { return mix64s(z); }
    static D u2d_small(U a) { return (double)(int64_t)a; } // a < 2^52
    static D gather(const double *t, U idx) { return t[idx]; }
    static bool lane_m(M m, int) { return m; }
    static double lane_d(D a, int) { return a; }
    static uint64_t lane_u(U a, int) { return a; }
    static D set_lane_d(D, int, double x) { return x; }
    static U set_lane_u(U, int, uint64_t x) { return x; }
    static D load_d(const double *p) { return p[0]; }
    static U load_u(const uint64_t *p) { return p[0]; }
    static M load_m(const bool *p) { return p[0]; }
    static void store_d(double *p, D a) { p[0] = a; }
};

} // namespace revs
} // namespace host
} // namespace bhmm
