// estep_sweep.hpp -- k_estep: the fused forward/backward sweep of the E-step (2, 4, 8 states).
//
// Lane mapping (estep_kernels.hpp): H = N/2 lanes per chunk, lane q owns the
// state pair (2q, 2q+1).  What this kernel changes is the instruction stream of the four
// loops, which bound the E-step (fp64 VALU):
//   * alpha is carried up to a power of two (the group's largest entry in [0.5, 1)); gamma and
//     xi are normalised by S_t = sum_i alpha_{t-1}[i] (A (p_t o beta_t))[i] in the backward
//     sweep (hidden/api.py:176-186), where the scale of alpha cancels, and the log-likelihood
//     (_hidden.c:57-66) is the exponent sum plus one log per chunk.  This removes the sum /
//     reciprocal / multiply chain from every forward step.
//   * the outlier rule of the gaussian model (outputmodel.py:119-131) is evaluated lazily: an
//     all-zero emission row makes the new vector exactly zero, which the exponent extraction
//     sees for free; the rule itself runs in a branch that is taken only then.
//   * exp() of the gaussian density is a branch-free kernel for non-positive arguments.
//   * every load is issued two steps (main sweeps) or four steps (warm-ups) ahead of its use;
//     the warm-ups of the speculative boundaries read the trajectory-major copy of the
//     observations, which needs no walk over the chunk table.
#pragma once

#include <type_traits>

#include "estep_kernels.hpp"

#ifndef ESTEP_SCALE_EVERY
#define ESTEP_SCALE_EVERY 4 // branch-free sweeps rescale every so many steps
#endif

namespace bhmm {

// exp(x) for x <= 0 to about 1 ulp: x = k ln2 + r, |r| <= ln2/2, exp(r) = 1 + r + r^2 q(r) with
// q the degree-9 minimax polynomial of tools/gen_exp_poly.py (approximation error 7.5e-18).
// Arguments below -750 (including -inf) give 0 like exp(); a NaN argument also gives 0 -- the
// caller restores NaN propagation in its outlier branch (fix_outlier).
__device__ __forceinline__ double exp_nonpos(double x)
{
    x = fmax(x, -750.0);
    const double k = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = fma(k, -0x1.62e42fefa39efp-1, x);
    r = fma(k, -0x1.abc9e3b39803fp-56, r);
    double q = 0x1.ad7e38e167506p-26;
    q = fma(q, r, 0x1.28ae7908135d8p-22);
    q = fma(q, r, 0x1.71df27c33abefp-19);
    q = fma(q, r, 0x1.a01998fd42e01p-16);
    q = fma(q, r, 0x1.a01a012882c92p-13);
    q = fma(q, r, 0x1.6c16c184889e3p-10);
    q = fma(q, r, 0x1.111111112836cp-7);
    q = fma(q, r, 0x1.55555555506eap-5);
    q = fma(q, r, 0x1.55555555554f7p-3);
    q = fma(q, r, 0x1.000000000000ap-1);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    return ldexp(q, (int)k);
}

[[maybe_unused]] static __global__ void k_exp_nonpos(const double *x, double *y, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        y[i] = exp_nonpos(x[i]);
}

// Gaussian density of one state, cn exp(-(o - mu)^2 / (2 sigma^2)), in 7 + GAUSS_PDF_DEG VALU
// instructions (cn * exp_nonpos(d * d * nh) takes 23; the two densities per lane and step were 57 %
// of the forward sweep's instruction stream).  With v = log2 of the density and s the integer
// emg_s >= log2 of the largest cn (host_common.hpp: fill_model):
//   u = clamp(a d^2 + b)        = (s - v) / 4096 in [0, 1]; the clamp is an output modifier
//   t = MG - u                  MG = 1.5 2^40 + s / 4096: t rounds to a multiple of 2^-12, the low
//                               dword of its mantissa is s + k', k' = rint(v - s), and
//                               t - MG = k' / 4096 exactly
//   w = u + (t - MG)            = -(v - s - k') / 4096 = -f / 4096 exactly, |f| <= 1/2
//   p = 2^(s + k') P(w)         P(w) = 2^f: minimax polynomial of tools/gen_exp2_poly.py in the
//                               exactly rescaled variable (max. relative error 2.8e-16)
// u above 1 (density below 2^(s - 4096), +-inf observations) gives 0 like exp(); a padded state
// has a = 0, b = 1.  The clamp modifier turns NaN into 0 (DX10_CLAMP), which would make a NaN
// observation a perfect hit: observations containing NaN put the context on the CAREFUL kernels at
// upload, where NANSAFE uses an IEEE minimum instead (NaN -> u = 1 -> p = 0 -> fix_outlier turns the
// row back into NaN).  An invalid sigma makes MG NaN and with it every density.
#ifndef GAUSS_PDF_DEG
#define GAUSS_PDF_DEG 10
#endif
template <bool NANSAFE>
__device__ __forceinline__ double gauss_pdf(double d, double a, double b, double MG)
{
    const double dd = d * d;
    double u;
    if constexpr (NANSAFE)
        u = fmin(fma(dd, a, b), 1.0);
    else
        u = fmin(fmax(fma(dd, a, b), 0.0), 1.0); // folded into the fma's clamp modifier
    const double t = MG - u;
    const double w = (t - MG) + u;
#if GAUSS_PDF_DEG == 10
    double q = 0x1.e3991e644e6abp+92;
    q = fma(q, w, -0x1.b6740fc28f781p+84);
    q = fma(q, w, 0x1.62c157ee59177p+76);
    q = fma(q, w, -0x1.ffcb55e82f22cp+67);
    q = fma(q, w, 0x1.4309126056718p+59);
    q = fma(q, w, -0x1.5d87fe9cc5d6fp+50);
    q = fma(q, w, 0x1.3b2ab6fbde0f7p+41);
    q = fma(q, w, -0x1.c6b08d703d48ap+31);
    q = fma(q, w, 0x1.ebfbdff82c3b9p+21);
    q = fma(q, w, -0x1.62e42fefa3a17p+11);
#else
    double q = -0x1.e7aa0f6005d3cp+100;
    q = fma(q, w, 0x1.e620fb765be15p+92);
    q = fma(q, w, -0x1.b526788b3b73dp+84);
    q = fma(q, w, 0x1.62bfc3c1c8a7cp+76);
    q = fma(q, w, -0x1.ffcbfba7b89b6p+67);
    q = fma(q, w, 0x1.43091310bf6b0p+59);
    q = fma(q, w, -0x1.5d87fe78cf26dp+50);
    q = fma(q, w, 0x1.3b2ab6fb9f413p+41);
    q = fma(q, w, -0x1.c6b08d7049fd1p+31);
    q = fma(q, w, 0x1.ebfbdff82c5adp+21);
    q = fma(q, w, -0x1.62e42fefa39efp+11);
#endif
    q = fma(q, w, 1.0);
    return ldexp(q, __double2loint(t));
}

[[maybe_unused]] static __global__ void k_gauss_pdf(const double *o, double *y, int64_t n, double mu,
                                                    double a, double b, double MG, int nansafe)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        y[i] = nansafe ? gauss_pdf<true>(o[i] - mu, a, b, MG) : gauss_pdf<false>(o[i] - mu, a, b, MG);
}

// compile-time loop index helpers: unrolled<K>(f) calls f(integral_constant<int, 0..K-1>);
// sc_at<J> says whether step J of an unrolled group rescales (the last one of every
// ESTEP_SCALE_EVERY)
template <int K, typename F, int J = 0>
__device__ __forceinline__ void unrolled(F &&f)
{
    if constexpr (J < K) {
        f(std::integral_constant<int, J>());
        unrolled<K, F, J + 1>(static_cast<F &&>(f));
    }
}
template <int J>
using sc_at = std::integral_constant<bool, (J % ESTEP_SCALE_EVERY) == ESTEP_SCALE_EVERY - 1>;

// my two states' slice of the emission model
struct EmisPair {
    double mu[2]; // gaussian: mean
    double a[2];  // gaussian: gauss_pdf() constants of the state
    double b[2];
    double MG;
};

// observation at position pos of the trajectory-major copy (same concatenation as the input)
template <int N, int KIND>
__device__ __forceinline__ ObsIn load_obs_rm(const void *obs_rm, int64_t pos, int q, int nreal)
{
    ObsIn in;
    in.o = 0.0;
    in.sym = 0;
    in.pp = make_double2(0.0, 0.0);
    if constexpr (KIND == EMIT_GAUSS) {
        in.o = static_cast<const double *>(obs_rm)[pos];
    } else if constexpr (KIND == EMIT_DISC) {
        in.sym = static_cast<const int32_t *>(obs_rm)[pos];
    } else {
        const double *row = static_cast<const double *>(obs_rm) + pos * nreal;
        in.pp.x = (2 * q < nreal) ? row[2 * q] : 0.0;
        in.pp.y = (2 * q + 1 < nreal) ? row[2 * q + 1] : 0.0;
    }
    return in;
}

// Per-lane cursor into the CI observations: one pointer that moves by whole records, so that the
// loads of neighbouring steps differ only in their immediate offset.
template <int N, int KIND>
struct ObsCursor {
    using T = typename std::conditional<KIND == EMIT_GAUSS, double,
              typename std::conditional<KIND == EMIT_DISC, int32_t, double2>::type>::type;
    static constexpr int STRIDE = (KIND == EMIT_EXPL) ? (N / 2) * 64 : 64; // elements per record
    const T *p;
    __device__ __forceinline__ ObsCursor(const void *obs_ci, int64_t rec, int cl, int q)
    {
        p = static_cast<const T *>(obs_ci) + rec * STRIDE + (KIND == EMIT_EXPL ? cl * (N / 2) + q : cl);
    }
    __device__ __forceinline__ ObsIn at(int drec) const
    {
        ObsIn in;
        in.o = 0.0;
        in.sym = 0;
        in.pp = make_double2(0.0, 0.0);
        if constexpr (KIND == EMIT_GAUSS)
            in.o = p[drec * STRIDE];
        else if constexpr (KIND == EMIT_DISC)
            in.sym = p[drec * STRIDE];
        else
            in.pp = p[drec * STRIDE];
        return in;
    }
    __device__ __forceinline__ void move(int drec) { p += drec * STRIDE; }
};

// emission probabilities of my two states WITHOUT the outlier rule; d = o - mu (gaussian)
// where the transposed emission matrix B^T [M][N] of the discrete model lives: staged in LDS, or
// -- alphabets too large for that -- in global memory (two pointers, so that the LDS accesses
// stay LDS instructions)
struct BtSrc {
    const double *lds;
    const double *glb;
    bool big;
};
// NANSAFE (the per-step-checked kernels) additionally rescues rows in the denormal range: if no
// state of the chunk has a probability of 2^-959 or more -- an observation ~38 sigma from every state,
// caller-supplied rows of 1e-320 -- sums over the row are denormal, their reciprocals infinite, and
// the fused density no longer rounds like the reference's exp-then-scale.  Such a row (rare: one
// divergent branch) is re-evaluated in the reference's own operation order (_gaussian.c:5-21) and
// returned times 2^900; the return value is that exponent (else 0).  gamma and xi do not see the
// factor, the forward sweep takes it off its exponent count.  An all-zero row is left to the
// outlier rule.
template <int N, int KIND, bool NANSAFE>
__device__ __forceinline__ int emit_raw(const Model<N> &m, unsigned long long gmask, int nreal,
                                        const ObsIn &in, const BtSrc &Bt, int q,
                                        const EmisPair &em, double (&p)[2], double (&d)[2])
{
    if constexpr (KIND == EMIT_GAUSS) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            d[b] = in.o - em.mu[b];
            p[b] = gauss_pdf<NANSAFE>(d[b], em.a[b], em.b[b], em.MG);
        }
    } else if constexpr (KIND == EMIT_DISC) {
        const int64_t e = (int64_t)in.sym * N + 2 * q;
        const double2 x = Bt.big ? *reinterpret_cast<const double2 *>(Bt.glb + e)
                                 : *reinterpret_cast<const double2 *>(Bt.lds + e);
        p[0] = x.x;
        p[1] = x.y;
        d[0] = d[1] = 0.0;
    } else {
        p[0] = in.pp.x;
        p[1] = in.pp.y;
        d[0] = d[1] = 0.0;
    }
    if constexpr (NANSAFE) {
        const bool big = p[0] >= 0x1p-959 || p[1] >= 0x1p-959;
        // single ENTRIES in the denormal range beside representable ones (a sparse transition matrix can
        // make such an entry the only way on): its products with alpha / beta would be rounded to one or
        // two bits, and differently in the two sweeps, which scale their vectors differently -- counts
        // off by 6e-4 where the reference keeps 1e-15 (tests/golden/cases/gauss8_denormal_entries_*).
        // The row is taken times 2^900 (exact) like an all-tiny one, unless an entry is large enough for
        // that to overflow further on.
        const bool tiny = (p[0] > 0.0 && p[0] < 0x1p-959) || (p[1] > 0.0 && p[1] < 0x1p-959);
        if (__builtin_expect((__ballot(tiny) & gmask) != 0ull, 0)) {
            if ((__ballot(big) & gmask) != 0ull &&
                (__ballot(p[0] > 0x1p+100 || p[1] > 0x1p+100) & gmask) == 0ull) {
                p[0] = ldexp(p[0], 900);
                p[1] = ldexp(p[1], 900);
                return 900;
            }
        }
        if (__builtin_expect((__ballot(big) & gmask) == 0ull, 0)) {
            if constexpr (KIND == EMIT_GAUSS) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int st = 2 * q + b;
                    const double z = (in.o - m.e0[st]) / m.e3[st];
                    p[b] = st < nreal ? m.e2[st] * exp(-0.5 * z * z) : 0.0;
                }
            }
            const bool nz = p[0] != 0.0 || p[1] != 0.0;
            if ((__ballot(nz) & gmask) != 0ull) {
                p[0] = ldexp(p[0], 900);
                p[1] = ldexp(p[1], 900);
                return 900;
            }
        }
    }
    return 0;
}

// The outlier rule, outputmodel.py:126-130: a row of pobs that is zero for every state becomes a
// row of ones.  Called only after a zero/denormal product was seen; returns true if p changed.
// A NaN observation (emit_raw turned it into p = 0) is turned back into NaN here.
template <int N, int KIND>
__device__ __forceinline__ bool fix_outlier(const ObsIn &in, int q, int nreal,
                                            unsigned long long gmask, double (&p)[2])
{
    if constexpr (KIND != EMIT_GAUSS) {
        return false;
    } else {
        const bool nz = (p[0] != 0.0) || (p[1] != 0.0);
        if ((__ballot(nz) & gmask) != 0ull)
            return false;
        double o = in.o, one = 1.0;
        asm volatile("" : "+v"(o), "+v"(one)); // keeps this arithmetic inside the rare branch
        one = (o != o) ? o : one;
        p[0] = (2 * q < nreal) ? one : 0.0;
        p[1] = (2 * q + 1 < nreal) ? one : 0.0;
        return true;
    }
}

// high dword of a non-negative double orders like the value; its top bits are the exponent
__device__ __forceinline__ bool tiny_hi(int hm) { return hm < 0x00100000; }
// ... below 2^-400: what the branch-free sweeps (rescaling every ESTEP_SCALE_EVERY steps) report.
// gamma and xi multiply an alpha row by a beta row, so each may use up only half of the exponent
// range; entries more than 2^-600 below the largest one of their vector may then be denormal,
// which is far below anything that reaches the statistics.
__device__ __forceinline__ bool small_hi(int hm) { return hm < ((1023 - 400) << 20); }

// a <- 2^ne (s o p), the group's largest entry brought into [0.5, 1); returns the exponent
// removed (-ne).
//   CAREFUL: every step is rescaled; for the gaussian model a zero / denormal result takes the
//   slow branch with the outlier rule.
//   otherwise the step is branch-free, rescales only where SCALE is set (every
//   ESTEP_SCALE_EVERY-th step of the unrolled loops -- powers of two, so the results do not
//   depend on the schedule) and records the smallest maximum seen at those points (hmin); the
//   kernel reports chunks where that fell below 2^-400 and the host repeats the E-step with the
//   CAREFUL instantiation.
template <int N, int KIND, bool CAREFUL, bool SCALE>
__device__ __forceinline__ int scaled_emit(const ObsIn &in, int q, int nreal,
                                           unsigned long long gmask, const double (&s)[2],
                                           double (&p)[2], double (&a)[2], int &hmin)
{
    constexpr int H = N / 2;
    double n0, n1;
    int hm;
    int extra = 0; // CAREFUL: 900 if the products were re-formed from the row times 2^900
    if constexpr (CAREFUL && KIND == EMIT_GAUSS) {
        for (;;) { // runs once; a second time only after the outlier rule replaced p
            n0 = s[0] * p[0];
            n1 = s[1] * p[1];
            hm = grp_max_i32<H>(max(__double2hiint(n0), __double2hiint(n1)));
            if (__builtin_expect(__ballot(tiny_hi(hm)) == 0ull, 1))
                break;
            // A step whose likelihood is in the denormal range although the emission row is not (the
            // mass sits on a state the observation excludes, the others contribute 1e-66 x 1e-257;
            // soak seed 16001 case 2087): the products of normal factors underflow to a few bits --
            // the reference's _hidden.c:57-66 loses them the same way, at another scale.  Re-formed
            // from the row times 2^900 (exact) they keep all bits; the 900 comes off the exponent count.
            {
                const double q0 = s[0] * ldexp(p[0], 900), q1 = s[1] * ldexp(p[1], 900);
                const int hq = grp_max_i32<H>(max(__double2hiint(q0), __double2hiint(q1)));
                if (__ballot(tiny_hi(hm) && !tiny_hi(hq) && hq < (2046 << 20)) != 0ull) {
                    if (tiny_hi(hm) && !tiny_hi(hq) && hq < (2046 << 20)) {
                        n0 = q0;
                        n1 = q1;
                        hm = hq;
                        extra = 900;
                    }
                    if (__ballot(tiny_hi(hm)) == 0ull)
                        break;
                }
            }
            if (__ballot(fix_outlier<N, KIND>(in, q, nreal, gmask, p)) == 0ull)
                break;
        }
    } else {
        n0 = s[0] * p[0];
        n1 = s[1] * p[1];
        if constexpr (!CAREFUL && !SCALE) {
            a[0] = n0;
            a[1] = n1;
            return 0;
        }
        hm = grp_max_i32<H>(max(__double2hiint(n0), __double2hiint(n1)));
        hmin = min(hmin, hm);
    }
    // a zero or denormal maximum (exponent field 0) is scaled by 2^1022 -- the bookkeeping stays
    // exact, the next steps finish the normalisation
    const int ne = 1022 - (hm >> 20);
    a[0] = ldexp(n0, ne);
    a[1] = ldexp(n1, ne);
    return -ne - extra;
}

// All-gather of a state-pair over the H lanes of a chunk.  ESTEP_LDS_GATHER: through a
// per-wavefront LDS exchange area (one 16-byte write, H 16-byte broadcast reads: LDS
// instructions, which leave the VALU to the arithmetic); otherwise on DPP quad permutes (2 VALU
// moves per double).  LDS operations of one wavefront execute in order, so the reads see the
// writes of the same step and no barrier is needed -- only the compiler has to keep the order.
#ifndef ESTEP_LDS_GATHER
#define ESTEP_LDS_GATHER 1
#endif
#ifndef ESTEP_DISC_P1_LDS
#define ESTEP_DISC_P1_LDS 0 // measured on configs[2]: P1 7.98 ms through LDS, 7.24 ms on DPP (the B lookup already sits in the LDS latency chain)
#endif
template <int N, bool USE_LDS>
struct Gather {
    static constexpr int H = N / 2;
    static constexpr bool LDS = USE_LDS && H > 1;
    double2 *w;       // my slot
    const double2 *r; // slot of lane 0 of my group; lane k is k * (64 / H) further
    __device__ __forceinline__ Gather(double *area)
    {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        double2 *base = reinterpret_cast<double2 *>(area) + wv * 64;
        w = base + (lane % H) * (64 / H) + lane / H;
        r = base + lane / H;
    }
    __device__ __forceinline__ void operator()(const double (&pair)[2], double (&full)[N]) const
    {
        if constexpr (LDS) {
            *w = make_double2(pair[0], pair[1]);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < H; ++k) {
                const double2 v = r[k * (64 / H)];
                full[2 * k] = v.x;
                full[2 * k + 1] = v.y;
            }
            asm volatile("" ::: "memory");
        } else {
            grp_gather<N>(pair, full);
        }
    }
};

// Scheduling fence: the exchange above has a latency of a few hundred cycles; independent
// arithmetic (the exp of the emission density) is placed between issuing it and using its result,
// and the compiler is kept from moving it back (ESTEP_SHADOW=0: leave the order to the compiler).
#ifndef ESTEP_SHADOW
#define ESTEP_SHADOW 1
#endif
__device__ __forceinline__ void sched_fence()
{
#if ESTEP_SHADOW
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// s = (a^T A)[my two states] from the gathered vector and my two columns of A
template <int N>
__device__ __forceinline__ void fwd_dot(const double (&af)[N], const double (&Ac)[N][2],
                                        double (&s)[2])
{
    s[0] = af[0] * Ac[0][0];
    s[1] = af[0] * Ac[0][1];
#pragma unroll
    for (int i = 1; i < N; ++i) {
        s[0] = fma(af[i], Ac[i][0], s[0]);
        s[1] = fma(af[i], Ac[i][1], s[1]);
    }
}
template <int N, class G>
__device__ __forceinline__ void fwd_matvec(const G &gather, const double (&a)[2],
                                           const double (&Ac)[N][2], double (&s)[2])
{
    double af[N];
    gather(a, af);
    fwd_dot<N>(af, Ac, s);
}

// r = (A bb)[my two states], bf = all-gather of bb
template <int N, class G>
__device__ __forceinline__ void bwd_matvec(const G &gather, const double (&bb)[2],
                                           const double (&Ar)[2][N], double (&bf)[N],
                                           double (&r)[2])
{
    gather(bb, bf);
    r[0] = Ar[0][0] * bf[0];
    r[1] = Ar[1][0] * bf[0];
#pragma unroll
    for (int j = 1; j < N; ++j) {
        r[0] = fma(Ar[0][j], bf[j], r[0]);
        r[1] = fma(Ar[1][j], bf[j], r[1]);
    }
}

// One backward step without statistics: bnew <- 2^ne A (p o b) (rescaled as in scaled_emit); p
// may be replaced by the outlier row.  bf (the gathered p o b) and r (the unscaled product) are
// returned for the xi accumulation of the caller.
template <int N, int KIND, bool CAREFUL, bool SCALE, class G>
__device__ __forceinline__ void beta_step(const G &gather, const ObsIn &in, int q,
                                          int nreal, unsigned long long gmask,
                                          const double (&Ar)[2][N], double (&p)[2],
                                          const double (&b)[2], double (&bf)[N], double (&r)[2],
                                          double (&bnew)[2], int &hmin, int &eacc)
{
    constexpr int H = N / 2;
    int hm;
    if constexpr (CAREFUL) {
        // p o beta in the denormal range although neither factor is (a state that explains this
        // observation well but the future badly): the sums formed from it would be denormal and
        // their reciprocals infinite.  gamma, xi and the rescaled beta only see the vector up to
        // a factor, so p is brought up by 2^900 first (exact; rare: one uniform branch).
        const int hb = grp_max_i32<H>(max(__double2hiint(p[0] * b[0]), __double2hiint(p[1] * b[1])));
        if (__builtin_expect(__ballot(hb < (64 << 20)) != 0ull, 0)) {
            if (hb < (64 << 20)) {
                p[0] = ldexp(p[0], 900);
                p[1] = ldexp(p[1], 900);
            }
        }
    }
    if constexpr (CAREFUL && KIND == EMIT_GAUSS) {
        for (;;) { // runs once; a second time only after the outlier rule replaced p
            const double bb[2] = {p[0] * b[0], p[1] * b[1]};
            bwd_matvec<N>(gather, bb, Ar, bf, r);
            hm = grp_max_i32<H>(max(__double2hiint(r[0]), __double2hiint(r[1])));
            if (__builtin_expect(__ballot(tiny_hi(hm)) == 0ull, 1))
                break;
            if (__ballot(fix_outlier<N, KIND>(in, q, nreal, gmask, p)) == 0ull)
                break;
        }
    } else {
        const double bb[2] = {p[0] * b[0], p[1] * b[1]};
        bwd_matvec<N>(gather, bb, Ar, bf, r);
        if constexpr (!CAREFUL && !SCALE) {
            bnew[0] = r[0];
            bnew[1] = r[1];
            return;
        }
        hm = grp_max_i32<H>(max(__double2hiint(r[0]), __double2hiint(r[1])));
        hmin = min(hmin, hm);
    }
    const int ne = 1022 - (hm >> 20); // see scaled_emit
    bnew[0] = ldexp(r[0], ne);
    bnew[1] = ldexp(r[1], ne);
    eacc -= ne; // exponent removed so far
}

// =========================================================================================
// k_estep<N, KIND, SPEC, GAMMA, CAREFUL>: forward sweep (alpha -> CI workspace, chunk log-likelihood), then
// backward sweep with gamma / xi / emission statistics in registers.
//   SPEC: chunk-boundary vectors by warm-up over W steps, verified afterwards by k_spec_check
//   (below); otherwise they are read from k_stitch.
//   A hidden Markov filter forgets its initial condition, so after enough steps the warm-up
//   result no longer depends on the uniform start vector.  k_spec_check (k_tail) compares, at
//   every chunk boundary, the vector one chunk assumed with the vector its neighbour actually
//   computed (componentwise relative tolerance).  If all boundaries agree the whole chain is
//   exact to that tolerance: chunk 0 starts from the true initial condition and the normalised
//   recursion is non-expansive in Hilbert's projective metric, so deviations add at most
//   linearly.  Otherwise the host re-runs the E-step with the prescan / stitch kernels, which
//   are exact unconditionally.
//   GAMMA: the instantiation that can store the gamma rows (gamma_ci may still be null).
//   PHASE: PH_ALL -- everything in one launch.  PH_FWDROWS -- forward sweep only, for the Gibbs
//   hidden-path step, which samples from alpha and is indifferent to its scale (_hidden.c:330-378):
//   every alpha row is stored ROUNDED TO fp32 (rows32, passed in the gamma_ci slot; any
//   power-of-two scale) and only the rows of the steps s % FWD_CKPT == 0 in fp64.  k_smp_maps
//   decides a draw from the fp32 row when the decision is clear at that precision and otherwise
//   rebuilds the exact fp64 row from the nearest stored one (path_kernels.hpp) -- half the alpha
//   traffic of storing fp64 rows, which bound both kernels.  PH_P1 / PH_P2 -- the E-step in two launches:
//   P1 has twice the workgroups, the first half run the forward sweep of their record group, the
//   second half only the backward warm-up (beta at the chunk's last step); neither needs the xi
//   accumulators, so four wavefronts fit a SIMD where PH_ALL has two, and the two halves hide
//   each other's latencies.  P2 is the backward sweep, started from the vectors P1 left in
//   a_exit / alpha_entry / beta_exit.
// Workgroup = one CI record group (64 chunks) = 32*N threads.
// =========================================================================================
#ifndef ESTEP_PF_F
#define ESTEP_PF_F 4 // steps per prefetch register set, forward sweep (observations)
#endif
#ifndef ESTEP_PF_B
#define ESTEP_PF_B 2 // ... backward sweep (observations + alpha)
#endif
#ifndef ESTEP_CKPT
#define ESTEP_CKPT 1 // keep every second alpha row in HBM, rebuild the others
#endif
#ifndef ESTEP_WAVES
#define ESTEP_WAVES 2
#endif
// E-step: every ESTEP_CK_OF(kind)-th alpha row reaches HBM, the backward sweep rebuilds the others.
// Two for the gaussian kind.  Four for the discrete kind: without the exponentials its forward
// launch (P1) is bound by HBM WRITES (3.2-3.8 TB/s is what the memory system sustains) with half of
// the VALU idle, while the backward launch (P2) is VALU-bound -- a quarter of the rows halves P1's
// bytes for one more rebuilt row per four steps in P2 (configs[2]: 24.6 -> see DESIGN.md section 9).
#define ESTEP_CK_OF(KINDV) ((KINDV) == EMIT_DISC ? 4 : 2)
#ifndef FWD_CKPT
#define FWD_CKPT 16 // forward-only pass (Gibbs step): fp64 alpha rows of the steps s % FWD_CKPT == 0
#endif
enum { PH_ALL = 0, PH_FWDROWS = 1, PH_P1 = 2, PH_P2 = 3 };

// Boundary vectors carried from one E-step to the next (round 3).  In an EM sequence the model moves
// little per iteration, so a chunk's warm-up need not start from the uniform vector W steps out: it
// starts from the PREVIOUS E-step's alpha (beta) da (db) steps before (after) the chunk -- off by
// the effect of the model change instead of by O(1) -- and runs only that many steps.  The boundary
// check afterwards is the same, so the result is as exact as before; a failed check repeats the
// E-step with full warm-ups.  alpha comes from the stored workspace rows (k_carry_alpha after the
// sweep), beta is captured by the backward sweep itself when `cap` steps remain.
struct Carry {
    const double *a_in = nullptr;  // [Gp][N] alpha da[g] + 1 steps before chunk g's first step
    const double *b_in = nullptr;  // [Gp][N] beta db[g] steps after chunk g's last step
    const int32_t *da = nullptr;   // [Gp] warm-up steps from a_in (0: none, full warm-up)
    const int32_t *db = nullptr;   // [Gp]
    double *b_out = nullptr;       // PH_P2: capture for the next E-step, [Gp][N] (entry g - 1)
    int32_t *db_out = nullptr;     // ... and its distance
    int cap = 0;                   // ... taken when `cap` steps of the chunk remain (multiple of 8).
                                   // The sweep is split there whether or not b_out is set: the split
                                   // changes how the lanes of a wavefront line up in time, hence the
                                   // order of the discrete kind's count atomics -- a call that does
                                   // not capture must round like one that does
};

template <int N, int KIND, bool SPEC, bool GAMMA, bool CAREFUL, int PHASE, bool BIGM = false>
__device__ __forceinline__ void estep_body(
    const Model<N> &m, const Chunks &ch, const void *obs_ci, const void *obs_rm,
    const int64_t *toff,   // [K+1] trajectory offsets (time steps)
    const double *Bt_g, double *alpha_entry, double *beta_exit,
    double *a_exit,        // SPEC: [G][N] alpha at each chunk's last step (any scale)
    double *b_entry,       // SPEC: [G][N] beta one step before each chunk, as the chunk derived it
    int W,                 // SPEC: warm-up length
    double *ws,            // CI workspace: alpha up to a power of two per step
    double *gamma_ci,      // CI gamma, or nullptr
    double *logL_chunk,    // [G] log of the product of the chunk's scaling factors
    double *gamma0,        // [K][N] gamma at t = 0 of every trajectory
    double *partials,      // [gridDim.x][S] register statistics per workgroup
    double *disc_partials, // [gridDim.x][M*N] discrete emission statistics per workgroup
    unsigned int *flags,   // !CAREFUL: flags[2] counts chunks that met a zero / denormal vector
    int32_t *ea_ci,        // PH_P1 -> PH_P2: [Gp] exponent of each chunk's last alpha row, then CI
                           // [record][64] cumulative exponents of the stored alpha rows
    const Carry &carry)    // PH_P1 / PH_P2: boundary vectors carried between E-steps (or all null)
{
    using SL = StatLayout<N, KIND>;
    constexpr int H = N / 2;
    constexpr int NW = (64 * H + 63) / 64; // wavefronts per workgroup
    constexpr bool FWDONLY = PHASE == PH_FWDROWS;
    constexpr bool HAS_BWD = PHASE == PH_ALL || PHASE == PH_P2;
    static_assert(SPEC || PHASE == PH_ALL, "the split phases are speculative-only");
    // PH_P1: workgroup b and b + gridDim/2 share record group b (forward / backward warm-up)
    const int bidx = PHASE == PH_P1 ? (int)(blockIdx.x % (gridDim.x / 2)) : (int)blockIdx.x;
    const bool role_f = PHASE != PH_P1 || blockIdx.x < gridDim.x / 2;
    const bool role_b = PHASE != PH_P1 || blockIdx.x >= gridDim.x / 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *red = smem;                                     // [NW][S]
    // discrete emission counts (_discrete.c:22-30) by LDS atomics: one table per wavefront where
    // LDS allows (m.dcopies == NW), so that the order of the additions -- hence the last bits of
    // the result -- does not depend on how the wavefronts of a workgroup interleave.
    // Alphabets whose tables do not fit the LDS (m.bt_global): B^T is read from global memory
    // (M N 8 bytes, L2-resident) and the counts are added with global fp64 atomics to one of
    // DISC_GLOBAL_TABLES replicated tables in disc_partials (zeroed by the host; their sums are
    // then no longer bit-reproducible from run to run).
    constexpr bool big = KIND == EMIT_DISC && BIGM; // (compile-time: the LDS path pays nothing)
    const int Mlds = (KIND == EMIT_DISC && !big) ? m.M : 0;
    const BtSrc Bt = {smem + NW * SL::S, Bt_g, big};        // [M][N]
    double *dstat0 = smem + NW * SL::S + Mlds * N;          // [dcopies][M][N]
    const int dcopies = (KIND == EMIT_DISC && !big) ? m.dcopies : 0;
    double *dstat = dstat0 + (dcopies > 1 ? (threadIdx.x >> 6) * (m.M * N) : 0);
    double *dstat_g = disc_partials + (int64_t)(blockIdx.x % DISC_GLOBAL_TABLES) * (m.M * N);
    // the discrete kind keeps its LDS unit busy with the emission table and the count atomics:
    // its all-gather runs on DPP instead (measured: 13 % faster there, equal for the gaussian)
    // (P1 and the forward-only pass keep no counts: there the discrete kind exchanges through LDS too)
    const Gather<N, ESTEP_LDS_GATHER && (KIND != EMIT_DISC || ESTEP_DISC_P1_LDS * (PHASE == PH_P1 || PHASE == PH_FWDROWS))>
        gather(dstat0 + dcopies * Mlds * N);
    int hmin = 0x7fffffff;
    [[maybe_unused]] int wmax = 0; // branch-free sweeps: largest alpha / S seen (high dword), see the self-check
#ifdef ESTEP_CLOCKPROBE
    const unsigned long long pc0 = __builtin_readcyclecounter(), pr0 = wall_clock64();
    unsigned long long pr1 = pr0, pr2 = pr0, pr3 = pr0;
#endif
    if constexpr (KIND == EMIT_DISC) {
        if (!big) {
            stage_Bt<N>(smem + NW * SL::S, Bt_g, m.M);
            for (int i = threadIdx.x; i < dcopies * m.M * N; i += blockDim.x)
                dstat0[i] = 0.0;
            __syncthreads();
        }
    }
    const int cl = threadIdx.x / H; // chunk within the record group == CI lane
    const int q = threadIdx.x % H;  // my state pair
    const int64_t g = (int64_t)bidx * 64 + cl;
    const int len = ch.len[g];
    const int64_t t0 = ch.t0[g];
    const int64_t goff = ch.goff[g];
    const bool first = (t0 == 0);
    const int nreal = m.nreal;
    const unsigned long long gmask = ((1ull << H) - 1) << ((threadIdx.x & 63) / H * H);
    const int64_t rec0 = ci_rec(g, 0, ch.Lmax);
    // exponent rows behind the [Gp] per-chunk entries (Gp = chunks of the launch)
    int32_t *const ea_rows =
        ea_ci + (int64_t)(PHASE == PH_P1 ? gridDim.x / 2 : gridDim.x) * 64;

    EmisPair em;
    double pi2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        em.mu[b] = m.e0[2 * q + b];
        em.a[b] = m.e4[2 * q + b];
        em.b[b] = m.e5[2 * q + b];
        pi2[b] = m.pi[2 * q + b];
    }
    em.MG = m.emg;
    const double u0 = (2 * q < nreal) ? 1.0 / (double)nreal : 0.0;
    const double u1 = (2 * q + 1 < nreal) ? 1.0 / (double)nreal : 0.0;

    double Cacc[2][N], sg[2], sd[2], sdd[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        sg[b] = sd[b] = sdd[b] = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j)
            Cacc[b][j] = 0.0;
    }

    if (len > 0) {
        double a[2];
        double2 aent = make_double2(0.0, 0.0); // the vector this chunk was entered with
        // my two columns of A (forward products; with ESTEP_CKPT also the backward sweep)
        double Ac[N][2];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            Ac[i][0] = m.A[i * N + 2 * q];
            Ac[i][1] = m.A[i * N + 2 * q + 1];
        }
        // ---------------- forward sweep (_hidden.c:16-66) ------------------------------
        if constexpr (PHASE == PH_P2) {
            // the forward sweep ran in PH_P1: its last vector and the entry vector
            const double2 x = *reinterpret_cast<const double2 *>(a_exit + g * N + 2 * q);
            a[0] = x.x;
            a[1] = x.y;
            aent = *reinterpret_cast<const double2 *>(alpha_entry + g * N + 2 * q);
        } else if (role_f) {
            int eP = 0;       // sum of the exponents removed
            double Sin = 1.0; // sum of the entry vector
            int s = 0;
            if (first) {
                // alpha_0 = pi o p_0, _hidden.c:28-39
                const ObsIn in = load_obs<N, KIND>(obs_ci, rec0, cl, q);
                double p[2], d[2];
                const int pe = emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, in, Bt, q, em, p, d);
                eP = scaled_emit<N, KIND, CAREFUL, true>(in, q, nreal, gmask, pi2, p, a, hmin) - pe;
                *ci_pair(ws, rec0, N, q, cl) = make_double2(a[0], a[1]);
                if constexpr (PHASE == PH_P1)
                    ea_rows[rec0 * 64 + cl] = eP;
                s = 1;
            } else {
                if constexpr (SPEC) {
                    // warm-up over the nw steps before the chunk, from the uniform vector or
                    // -- where that reaches the start of the trajectory -- exactly from pi
                    int nw = (int)(t0 < (int64_t)W ? t0 : (int64_t)W);
                    // carried start (only where the full warm-up would not reach the trajectory's
                    // start, which is exact and stays)
                    int dca = 0;
                    if constexpr (PHASE == PH_P1)
                        if (carry.a_in && t0 > (int64_t)W)
                            dca = carry.da[g];
                    if (dca > 0)
                        nw = dca;
                    int64_t pos = goff - nw;
                    auto wstep = [&](const ObsIn &in, auto sc) {
                        double p[2], d[2], sv[2], af[N];
                        gather(a, af);
                        sched_fence();
                        emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, in, Bt, q, em, p, d);
                        sched_fence();
                        fwd_dot<N>(af, Ac, sv);
                        (void)scaled_emit<N, KIND, CAREFUL, decltype(sc)::value>(in, q, nreal, gmask,
                                                                                 sv, p, a, hmin);
                    };
                    if (dca > 0) {
                        const double2 x = *reinterpret_cast<const double2 *>(carry.a_in + g * N + 2 * q);
                        a[0] = x.x;
                        a[1] = x.y;
                    } else if ((int64_t)nw == t0) {
                        const ObsIn in = load_obs_rm<N, KIND>(obs_rm, pos, q, nreal);
                        double p[2], d[2];
                        emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, in, Bt, q, em, p, d);
                        (void)scaled_emit<N, KIND, CAREFUL, true>(in, q, nreal, gmask, pi2, p, a, hmin);
                        ++pos;
                        --nw;
                    } else {
                        a[0] = u0;
                        a[1] = u1;
                    }
                    for (int i = nw & 3; i > 0; --i) {
                        const ObsIn in = load_obs_rm<N, KIND>(obs_rm, pos, q, nreal);
                        wstep(in, std::true_type());
                        ++pos;
                    }
                    nw &= ~3;
                    if (nw > 0) {
                        ObsIn c[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            c[j] = load_obs_rm<N, KIND>(obs_rm, pos + j, q, nreal);
                        for (; nw > 0; nw -= 4) {
                            pos += 4;
                            ObsIn nx[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                nx[j] = c[j];
                            if (nw > 4) {
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    nx[j] = load_obs_rm<N, KIND>(obs_rm, pos + j, q, nreal);
                            }
                            wstep(c[0], sc_at<0>());
                            wstep(c[1], sc_at<1>());
                            wstep(c[2], sc_at<2>());
                            wstep(c[3], sc_at<3>());
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                c[j] = nx[j];
                        }
                    }
                    aent = make_double2(a[0], a[1]);
                    *reinterpret_cast<double2 *>(alpha_entry + g * N + 2 * q) = aent;
                } else {
                    // entry vector from k_stitch (power-of-two scaled)
                    aent = *reinterpret_cast<const double2 *>(alpha_entry + g * N + 2 * q);
                    a[0] = aent.x;
                    a[1] = aent.y;
                }
                Sin = grp_sum<H>(a[0] + a[1]);
            }
#ifdef ESTEP_CLOCKPROBE
            pr1 = wall_clock64();
#endif
            constexpr int RS = (N / 2) * 64; // double2 elements per CI record of N doubles
            // Which alpha rows reach HBM.  The forward main sweep is bound by HBM write bandwidth
            // when every row is stored, so with ESTEP_CKPT only the rows of
            // even steps are, plus the last rows of the chunk that the single steps of the
            // backward sweep read; the backward sweep rebuilds alpha_{s-1} from alpha_{s-2} and
            // the emission of step s-1, which it needs anyway.
            auto fstep = [&](const ObsIn &in, double2 &out, auto sc) {
                double p[2], d[2], sv[2], af[N];
                gather(a, af);
                sched_fence();
                const int pe = emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, in, Bt, q, em, p, d);
                sched_fence();
                fwd_dot<N>(af, Ac, sv);
                eP += scaled_emit<N, KIND, CAREFUL, decltype(sc)::value>(in, q, nreal, gmask, sv, p,
                                                                        a, hmin) - pe;
                out = make_double2(a[0], a[1]);
            };
            // PH_P1: with every stored row goes the exponent removed so far (eP), one int per chunk;
            // PH_P2 forms the gamma / xi normalisers from it without a reciprocal
            int32_t *pe = nullptr;
            // which fp64 rows reach HBM: the E-step keeps every second one (CKS = 2, the backward
            // sweep rebuilds the others); the forward-only pass keeps every FWD_CKPT-th (and all
            // rows in fp32, see PH_FWDROWS above)
            constexpr int PF = ESTEP_PF_F;
            constexpr bool CKPT = ESTEP_CKPT && !FWDONLY;
            constexpr int CKS = FWDONLY ? FWD_CKPT : (CKPT ? ESTEP_CK_OF(KIND) : 1);
            constexpr int ALIGN = FWDONLY ? 2 * PF : (CKS > 2 ? CKS : 2);
            static_assert(PF % 2 == 0 && (FWDONLY ? CKS % (2 * PF) == 0 : (2 * PF) % CKS == 0),
                          "stored rows sit at fixed positions inside the unrolled groups");
            [[maybe_unused]] float2 *pw32 = nullptr;
            constexpr int RS32 = (N / 2) * 64; // float2 elements per fp32 CI record
            if constexpr (FWDONLY) {
                float2 *rows32 = reinterpret_cast<float2 *>(gamma_ci);
                if (first) // row 0 (written in fp64 above)
                    rows32[rec0 * RS32 + cl * (N / 2) + q] = make_float2((float)a[0], (float)a[1]);
                pw32 = rows32 + (rec0 + s) * RS32 + cl * (N / 2) + q;
            }
            auto single = [&](ObsCursor<N, KIND> &po, double2 *&pw, bool store) {
                double2 o;
                fstep(po.at(0), o, std::true_type());
                if (!FWDONLY || store)
                    *pw = o;
                if constexpr (FWDONLY) {
                    *pw32 = make_float2((float)o.x, (float)o.y);
                    pw32 += RS32;
                }
                if constexpr (PHASE == PH_P1)
                    *pe = eP;
                po.move(1);
                pw += RS;
                pe += 64;
            };
            ObsCursor<N, KIND> po(obs_ci, rec0 + s, cl, q);
            double2 *pw = ci_pair(ws, rec0 + s, N, q, cl);
            pe = ea_rows + (rec0 + s) * 64 + cl;
            // first chunk of a trajectory (s = 1): bring the group base to an aligned step
            while ((s & (ALIGN - 1)) && s < len) {
                single(po, pw, (s % CKS) == 0);
                ++s;
            }
            // groups of 2 PF steps (two register sets, each loaded PF..2PF-1 steps before its
            // use), then the remaining 0 .. 2PF-1 steps one by one
            const int tail = (len - s) % (2 * PF);
            int rem = len - s - tail;
            if (rem > 0) {
                ObsIn x[PF], y[PF];
                double2 ox[PF], oy[PF];
#pragma unroll
                for (int j = 0; j < PF; ++j)
                    x[j] = po.at(j);
                for (; rem > 0; rem -= 2 * PF) {
#pragma unroll
                    for (int j = 0; j < PF; ++j)
                        y[j] = po.at(PF + j);
                    unrolled<PF>([&](auto j) {
                        fstep(x[j], ox[j], sc_at<j>());
                        if constexpr (FWDONLY) {
                            pw32[j * RS32] = make_float2((float)ox[j].x, (float)ox[j].y);
                            if (j == 0 && (s % CKS) == 0)
                                pw[0] = ox[j];
                        } else if constexpr (j % CKS == 0) {
                            pw[j * RS] = ox[j];
                            if constexpr (PHASE == PH_P1)
                                pe[j * 64] = eP;
                        }
                    });
                    if (rem > 2 * PF) {
#pragma unroll
                        for (int j = 0; j < PF; ++j)
                            x[j] = po.at(2 * PF + j);
                    }
                    unrolled<PF>([&](auto j) {
                        fstep(y[j], oy[j], sc_at<PF + j>());
                        if constexpr (FWDONLY) {
                            pw32[(PF + j) * RS32] = make_float2((float)oy[j].x, (float)oy[j].y);
                        } else if constexpr ((PF + j) % CKS == 0) {
                            pw[(PF + j) * RS] = oy[j];
                            if constexpr (PHASE == PH_P1)
                                pe[(PF + j) * 64] = eP;
                        }
                    });
                    if constexpr (CKPT) {
                        // the backward sweep reads the last (len-1) % (2 CKS) + 1 rows directly: if
                        // the single steps below do not cover them, all rows of the last group are kept
                        if (rem == 2 * PF && tail < 2 * CKS) {
#pragma unroll
                            for (int j = 0; j < PF; ++j) {
                                if (j % CKS != 0)
                                    pw[j * RS] = ox[j];
                                if ((PF + j) % CKS != 0)
                                    pw[(PF + j) * RS] = oy[j];
                            }
                        }
                    }
                    po.move(2 * PF);
                    pw += 2 * PF * RS;
                    pe += 2 * PF * 64;
                    if constexpr (FWDONLY) {
                        pw32 += 2 * PF * RS32;
                        s += 2 * PF;
                    }
                }
            }
            for (int i = tail; i > 0; --i) // step len - i
                single(po, pw, ((len - i) % CKS) == 0);
#ifdef ESTEP_CLOCKPROBE
            pr2 = wall_clock64();
#endif
            const double Sfin = grp_sum<H>(a[0] + a[1]);
            if (q == 0)
                logL_chunk[g] = log(Sfin / Sin) + (double)eP * 0.693147180559945309417232121458;
            if constexpr (SPEC)
                *reinterpret_cast<double2 *>(a_exit + g * N + 2 * q) = make_double2(a[0], a[1]);
            if constexpr (PHASE == PH_P1)
                ea_ci[g] = eP;
        }

        if constexpr (!FWDONLY) {
        if (role_b) {
        // ---------------- backward sweep (_hidden.c:69-110, hidden/api.py:176-186) ----------
        double Ar[2][N];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            Ar[0][i] = m.A[(2 * q) * N + i];
            Ar[1][i] = m.A[(2 * q + 1) * N + i];
        }
        const int k = ch.traj[g];
        double b2[2], gam[2];
        if constexpr (SPEC && PHASE != PH_P2) {
            // beta at my last step: constant at the end of the trajectory (_hidden.c:79-88),
            // otherwise warmed up backwards over the nw steps after the chunk -- which is the
            // same start vector where the warm-up reaches the end of the trajectory
            const int64_t after = (toff[k + 1] - toff[k]) - (t0 + len);
            int nw = (int)(after < (int64_t)W ? after : (int64_t)W);
            b2[0] = u0;
            b2[1] = u1;
            if constexpr (PHASE == PH_P1) {
                // carried start (only where the warm-up would not reach the trajectory's end, whose
                // constant vector is exact and stays)
                const int dcb = (carry.b_in && after > (int64_t)W) ? carry.db[g] : 0;
                if (dcb > 0) {
                    nw = dcb;
                    const double2 x = *reinterpret_cast<const double2 *>(carry.b_in + g * N + 2 * q);
                    b2[0] = x.x;
                    b2[1] = x.y;
                }
            }
            int64_t pos = goff + len + nw - 1;
            auto wstep = [&](const ObsIn &in, auto sc) {
                double p[2], d[2], bf[N], r[2], bn[2];
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, in, Bt, q, em, p, d);
                int unused = 0;
                beta_step<N, KIND, CAREFUL, decltype(sc)::value>(gather, in, q, nreal, gmask, Ar, p,
                                                                 b2, bf, r, bn, hmin, unused);
                b2[0] = bn[0];
                b2[1] = bn[1];
            };
            for (int i = nw & 3; i > 0; --i) {
                const ObsIn in = load_obs_rm<N, KIND>(obs_rm, pos, q, nreal);
                wstep(in, std::true_type());
                --pos;
            }
            nw &= ~3;
            if (nw > 0) {
                ObsIn c[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    c[j] = load_obs_rm<N, KIND>(obs_rm, pos - j, q, nreal);
                for (; nw > 0; nw -= 4) {
                    pos -= 4;
                    ObsIn nx[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        nx[j] = c[j];
                    if (nw > 4) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            nx[j] = load_obs_rm<N, KIND>(obs_rm, pos - j, q, nreal);
                    }
                    wstep(c[0], sc_at<0>());
                    wstep(c[1], sc_at<1>());
                    wstep(c[2], sc_at<2>());
                    wstep(c[3], sc_at<3>());
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        c[j] = nx[j];
                }
            }
            *reinterpret_cast<double2 *>(beta_exit + g * N + 2 * q) = make_double2(b2[0], b2[1]);
        } else {
            const double2 x = *reinterpret_cast<const double2 *>(beta_exit + g * N + 2 * q);
            b2[0] = x.x;
            b2[1] = x.y;
        }
        if constexpr (HAS_BWD) {
        // Normaliser of gamma and xi.  S = sum_i alpha[i] (A (p o beta))[i] is the same bilinear form
        // at every step of the chunk, only the powers of two removed from alpha (ea, recorded by
        // PH_P1) and from beta (Eb) differ: PH_P2 takes one reciprocal per chunk (rS0, at the
        // last step, where alpha carries the exponent ebase) and scales it, 1/S = rS0 2^(ea + Eb -
        // ebase), instead of a sum over the group and a reciprocal per step.  The single steps and
        // the other instantiations sum and divide as before.
        constexpr bool EXPO = PHASE == PH_P2 && !CAREFUL;
        double rS0 = 0.0;
        int Eb = 0;
        const int ebase = EXPO ? ea_ci[g] : 0;
        {
#ifdef ESTEP_CLOCKPROBE
            pr3 = wall_clock64();
#endif
            gam[0] = a[0] * b2[0];
            gam[1] = a[1] * b2[1];
            rS0 = fast_rcp(grp_sum<H>(gam[0] + gam[1]));
            gam[0] *= rS0;
            gam[1] *= rS0;
        }
        // consume gamma_s: state counts, emission statistics, optional gamma row
        constexpr int RS = (N / 2) * 64; // double2 elements per CI record of N doubles
        auto consume = [&](const ObsIn &in, const double (&d)[2], double2 *gdst) {
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                sg[b] += gam[b];
                if constexpr (KIND == EMIT_GAUSS) {
                    const double gd = gam[b] * d[b];
                    sd[b] += gd;
                    sdd[b] = fma(gd, d[b], sdd[b]);
                }
                if constexpr (KIND == EMIT_DISC) { // _discrete.c:22-30
                    if (big)
                        atomicAdd(&dstat_g[(int64_t)in.sym * N + 2 * q + b], gam[b]);
                    else
                        atomicAdd(&dstat[in.sym * N + 2 * q + b], gam[b]);
                }
            }
            if constexpr (GAMMA)
                if (gamma_ci)
                    *gdst = make_double2(gam[0], gam[1]);
        };
        // step s with apv = alpha_{s-1} and the emission row p of step s: consume gamma_s, then
        // the pair (s-1, s) gives the xi contribution, gamma_{s-1} and beta_{s-1}
        auto bcore = [&](const ObsIn &in, double (&p)[2], const double (&d)[2], const double2 &apv,
                         double2 *gdst, auto sc) {
            double bf[N], r[2], bn[2];
            consume(in, d, gdst);
            beta_step<N, KIND, CAREFUL, decltype(sc)::value>(gather, in, q, nreal, gmask, Ar, p, b2,
                                                             bf, r, bn, hmin, Eb);
            double q0 = apv.x * r[0], q1 = apv.y * r[1];
            double S = grp_sum<H>(q0 + q1);
            if constexpr (CAREFUL) {
                // alpha concentrated on states whose (A (p o beta)) is in the denormal range although
                // p o beta as a whole is not (sparse A, narrow states): S is denormal, 1 / S infinite.
                // The vectors are brought up by 2^900 (exact), which gamma and xi do not see.
                if (__builtin_expect(__ballot(!(S >= 0x1p-959) && S > 0.0) != 0ull, 0)) {
                    // The products are formed again from p 2^900 rather than scaled afterwards: r was
                    // rounded in the denormal range (19 bits left in tests/golden/cases/
                    // disc4_M1150_9501_20.npz: one count off by 2e-6), and xi / gamma only come out
                    // normalised if S is the sum of the very products that are accumulated.  Rows with
                    // an entry above 2^100 (narrow Gaussians; p already raised by beta_step) would
                    // overflow and keep the scaling of the rounded values.
                    const bool me = !(S >= 0x1p-959) && S > 0.0;
                    const bool roomy = (__ballot(p[0] > 0x1p+100 || p[1] > 0x1p+100) & gmask) == 0ull;
                    const int kp = me && roomy ? 900 : 0, kr = me && !roomy ? 900 : 0;
                    const double bb[2] = {ldexp(p[0], kp) * b2[0], ldexp(p[1], kp) * b2[1]};
                    bwd_matvec<N>(gather, bb, Ar, bf, r);
                    r[0] = ldexp(r[0], kr);
                    r[1] = ldexp(r[1], kr);
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        bf[j] = ldexp(bf[j], kr);
                    q0 = apv.x * r[0];
                    q1 = apv.y * r[1];
                    S = grp_sum<H>(q0 + q1);
                }
            }
            const double rS = fast_rcp(S);
            gam[0] = q0 * rS;
            gam[1] = q1 * rS;
            const double w0 = apv.x * rS, w1 = apv.y * rS;
            if constexpr (!CAREFUL)
                wmax = max(wmax, max(__double2hiint(w0), __double2hiint(w1)));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                Cacc[0][j] = fma(w0, bf[j], Cacc[0][j]);
                Cacc[1][j] = fma(w1, bf[j], Cacc[1][j]);
            }
            b2[0] = bn[0];
            b2[1] = bn[1];
        };
        auto bstep = [&](const ObsIn &in, const double2 &apv, double2 *gdst, auto sc) {
            double p[2], d[2];
            emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, in, Bt, q, em, p, d);
            bcore(in, p, d, apv, gdst, sc);
        };
        // steps s and s-1 from the stored row alpha_{s-2}: alpha_{s-1} = (alpha_{s-2} A) o p_{s-1}
        // up to a scale, which gamma and xi do not see
        // second half of a backward step, from the gathered p o beta: beta_{s-1}, gamma_{s-1}, xi
        auto bfinish = [&](const double (&bf)[N], const double2 &apv, int ea, auto sc) {
            double r[2];
            r[0] = Ar[0][0] * bf[0];
            r[1] = Ar[1][0] * bf[0];
#pragma unroll
            for (int j = 1; j < N; ++j) {
                r[0] = fma(Ar[0][j], bf[j], r[0]);
                r[1] = fma(Ar[1][j], bf[j], r[1]);
            }
            const double q0 = apv.x * r[0], q1 = apv.y * r[1];
            double rS;
            if constexpr (EXPO)
                rS = ldexp(rS0, ea + Eb - ebase);
            else
                rS = fast_rcp(grp_sum<H>(q0 + q1));
            gam[0] = q0 * rS;
            gam[1] = q1 * rS;
            const double w0 = apv.x * rS, w1 = apv.y * rS;
            if constexpr (!CAREFUL)
                wmax = max(wmax, max(__double2hiint(w0), __double2hiint(w1)));
#pragma unroll
            for (int j = 0; j < N; ++j) {
                Cacc[0][j] = fma(w0, bf[j], Cacc[0][j]);
                Cacc[1][j] = fma(w1, bf[j], Cacc[1][j]);
            }
            if constexpr (decltype(sc)::value) {
                const int hm = grp_max_i32<H>(max(__double2hiint(r[0]), __double2hiint(r[1])));
                hmin = min(hmin, hm);
                const int ne = 1022 - (hm >> 20);
                b2[0] = ldexp(r[0], ne);
                b2[1] = ldexp(r[1], ne);
                Eb -= ne;
            } else {
                b2[0] = r[0];
                b2[1] = r[1];
            }
        };
        // steps s and s-1 from the stored row alpha_{s-2}: alpha_{s-1} = (alpha_{s-2} A) o p_{s-1}
        // up to a scale, which gamma and xi do not see.  Branch-free variant: the three LDS
        // exchanges of the pair are issued as early as their inputs allow and the two exp
        // evaluations sit in their shadows.
        auto bpair = [&](const ObsIn &hi, const ObsIn &lo, const double2 &alo, int ea,
                         double2 *gdst, auto sc_hi, auto sc_lo) {
            double p_hi[2], d_hi[2], p_lo[2], d_lo[2], sv[2], ah[2];
            const double al[2] = {alo.x, alo.y};
            // (per-step-checked kernels, every emission kind: the rebuilt alpha row is RESCALED.  Left
            // at the magnitude of its emission row -- explicit rows of 1e-222 -- its product with
            // A (p o beta) underflowed to zero, S = 0, gamma and the counts NaN: found by the explicit
            // E-step check of tests/sweeps/stress_small.py, tests/golden/cases/explicit2_nan_counts_*)
            if constexpr (CAREFUL) {
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, hi, Bt, q, em, p_hi, d_hi);
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, lo, Bt, q, em, p_lo, d_lo);
                fwd_matvec<N>(gather, al, Ac, sv);
                int unused = 0x7fffffff;
                (void)scaled_emit<N, KIND, CAREFUL, false>(lo, q, nreal, gmask, sv, p_lo, ah,
                                                           unused);
                bcore(hi, p_hi, d_hi, make_double2(ah[0], ah[1]), gdst, sc_hi);
                bcore(lo, p_lo, d_lo, alo, gdst - RS, sc_lo);
            } else {
                double afl[N], bf[N];
                gather(al, afl); // (1) alpha_{s-2}, for the rebuild
                sched_fence();
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, hi, Bt, q, em, p_hi, d_hi);
                sched_fence();
                fwd_dot<N>(afl, Ac, sv);
                consume(hi, d_hi, gdst);
                {
                    const double bb[2] = {p_hi[0] * b2[0], p_hi[1] * b2[1]};
                    gather(bb, bf); // (2) p o beta of step s
                }
                sched_fence();
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, lo, Bt, q, em, p_lo, d_lo);
                sched_fence();
                ah[0] = sv[0] * p_lo[0];
                ah[1] = sv[1] * p_lo[1];
                bfinish(bf, make_double2(ah[0], ah[1]), ea, sc_hi);
                consume(lo, d_lo, gdst - RS);
                {
                    const double bb[2] = {p_lo[0] * b2[0], p_lo[1] * b2[1]};
                    gather(bb, bf); // (3) p o beta of step s-1
                }
                bfinish(bf, alo, ea, sc_lo);
            }
        };
        // Four steps s .. s-3 from the stored row alpha_{s-4} (ESTEP_CK_OF == 4): the three rows in
        // between are rebuilt first, forwards (each needs the emission row of its own step, which
        // the backward steps below use again), then the four backward steps run as in bpair.
        [[maybe_unused]] auto bquad = [&](const ObsIn &x3, const ObsIn &x2, const ObsIn &x1,
                                          const ObsIn &x0, const double2 &alo, int ea, double2 *gdst,
                                          auto sc3, auto sc2, auto sc1, auto sc0) {
            double p3[2], d3[2], p2[2], d2[2], p1[2], d1[2], p0[2], d0[2], sv[2];
            double a1[2], a2[2], a3[2], af[N], bf[N];
            const double al[2] = {alo.x, alo.y};
            if constexpr (CAREFUL) {
                // per-step-checked kernels: every rebuilt row is RESCALED (three steps of emission
                // probabilities of 1e-150 took the unscaled rows to zero, S = 0, gamma and the counts
                // NaN -- tests/golden/cases/disc5_tiny_B_8001_1411.npz) and the backward steps are the
                // checked ones of bcore
                int unused = 0x7fffffff;
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, x0, Bt, q, em, p0, d0);
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, x1, Bt, q, em, p1, d1);
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, x2, Bt, q, em, p2, d2);
                emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, x3, Bt, q, em, p3, d3);
                fwd_matvec<N>(gather, al, Ac, sv);
                (void)scaled_emit<N, KIND, CAREFUL, false>(x0, q, nreal, gmask, sv, p0, a1, unused);
                fwd_matvec<N>(gather, a1, Ac, sv);
                (void)scaled_emit<N, KIND, CAREFUL, false>(x1, q, nreal, gmask, sv, p1, a2, unused);
                fwd_matvec<N>(gather, a2, Ac, sv);
                (void)scaled_emit<N, KIND, CAREFUL, false>(x2, q, nreal, gmask, sv, p2, a3, unused);
                bcore(x3, p3, d3, make_double2(a3[0], a3[1]), gdst, sc3);
                bcore(x2, p2, d2, make_double2(a2[0], a2[1]), gdst - RS, sc2);
                bcore(x1, p1, d1, make_double2(a1[0], a1[1]), gdst - 2 * RS, sc1);
                bcore(x0, p0, d0, alo, gdst - 3 * RS, sc0);
                (void)ea;
                (void)af;
                (void)bf;
                return;
            }
            gather(al, af); // alpha_{s-4}
            sched_fence();
            emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, x0, Bt, q, em, p0, d0);
            emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, x1, Bt, q, em, p1, d1);
            sched_fence();
            fwd_dot<N>(af, Ac, sv);
            a1[0] = sv[0] * p0[0]; // alpha_{s-3}
            a1[1] = sv[1] * p0[1];
            gather(a1, af);
            sched_fence();
            emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, x2, Bt, q, em, p2, d2);
            emit_raw<N, KIND, CAREFUL>(m, gmask, nreal, x3, Bt, q, em, p3, d3);
            sched_fence();
            fwd_dot<N>(af, Ac, sv);
            a2[0] = sv[0] * p1[0]; // alpha_{s-2}
            a2[1] = sv[1] * p1[1];
            gather(a2, af);
            fwd_dot<N>(af, Ac, sv);
            a3[0] = sv[0] * p2[0]; // alpha_{s-1}
            a3[1] = sv[1] * p2[1];
            auto back = [&](const ObsIn &in, const double (&p)[2], const double (&d)[2],
                            const double2 &apv, double2 *gd, auto sc) {
                consume(in, d, gd);
                const double bb[2] = {p[0] * b2[0], p[1] * b2[1]};
                gather(bb, bf);
                bfinish(bf, apv, ea, sc);
            };
            back(x3, p3, d3, make_double2(a3[0], a3[1]), gdst, sc3);
            back(x2, p2, d2, make_double2(a2[0], a2[1]), gdst - RS, sc2);
            back(x1, p1, d1, make_double2(a1[0], a1[1]), gdst - 2 * RS, sc1);
            back(x0, p0, d0, alo, gdst - 3 * RS, sc0);
        };
        // the observation of step 0 is needed last: fetch it now
        const ObsIn in0 = ObsCursor<N, KIND>(obs_ci, rec0, cl, q).at(0);
        double2 *const pg0 = GAMMA && gamma_ci ? ci_pair(gamma_ci, rec0, N, q, cl) : nullptr;
        {
            // steps len-1 .. 1; po / pa / pg point at the records of the step about to be done
            ObsCursor<N, KIND> po(obs_ci, rec0 + len - 1, cl, q);
            const double2 *pa = ci_pair(ws, rec0 + len - 1, N, q, cl);
            double2 *pg = pg0 + (int64_t)(len - 1) * RS;
            int rem = len - 1;
            if constexpr (ESTEP_CKPT != 0 && ESTEP_CK_OF(KIND) == 4 &&
                          !(CAREFUL && KIND == EMIT_GAUSS)) {
                // (len-1) % 8 single steps on stored rows, then groups of two quads: two register
                // sets {obs_s .. obs_{s-3}, alpha_{s-4}}, each loaded 4..7 steps before its use
                for (int i = rem % 8; i > 0; --i) {
                    bstep(po.at(0), pa[-RS], pg, std::true_type());
                    po.move(-1);
                    pa -= RS;
                    pg -= RS;
                }
                rem -= rem % 8;
                if (rem > 0) {
                    const int32_t *pea = ea_rows + (rec0 + rem) * 64 + cl;
                    ObsIn x[4], y[4];
                    double2 xa = pa[-4 * RS], ya;
                    int xe = EXPO ? pea[-4 * 64] : 0, ye = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        x[j] = po.at(-j);
                    // (PH_P2 with a capture request: the same loop runs in two stretches, down to
                    // `cap` remaining steps and then to the end; in between b2 -- beta at my local
                    // step cap, cap + 1 steps after the last step of the chunk before me -- is
                    // left for that chunk's next warm-up.  No instruction is added to the loop.)
                    int stop = (PHASE == PH_P2 && carry.cap > 0 && rem > carry.cap) ? carry.cap : 0;
                    // (a real loop, not two copies of the body: both stretches must execute the SAME
                    // instructions, or a run with a capture would round differently from one without)
#pragma clang loop unroll(disable)
                    for (int stretch = 0; stretch < 2; ++stretch) {
                    for (; rem > stop; rem -= 8) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            y[j] = po.at(-4 - j);
                        ya = pa[-8 * RS];
                        if constexpr (EXPO)
                            ye = pea[-8 * 64];
                        bquad(x[0], x[1], x[2], x[3], xa, xe, pg, sc_at<0>(), sc_at<1>(), sc_at<2>(),
                              sc_at<3>());
                        if (rem > 8) {
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                x[j] = po.at(-8 - j);
                            xa = pa[-12 * RS];
                            if constexpr (EXPO)
                                xe = pea[-12 * 64];
                        }
                        bquad(y[0], y[1], y[2], y[3], ya, ye, pg - 4 * RS, sc_at<0>(), sc_at<1>(),
                              sc_at<2>(), sc_at<3>());
                        po.move(-8);
                        pa -= 8 * RS;
                        pg -= 8 * RS;
                        pea -= 8 * 64;
                    }
                    if (stop == 0)
                        break;
                    if (carry.b_out && !first) {
                        *reinterpret_cast<double2 *>(carry.b_out + (g - 1) * N + 2 * q) =
                            make_double2(b2[0], b2[1]);
                        if (q == 0)
                            carry.db_out[g - 1] = stop + 1;
                    }
                    stop = 0;
                    }
                }
            } else if constexpr (ESTEP_CKPT != 0) {
                // (len-1) % 4 single steps on stored rows, then groups of two pairs: two register
                // sets {obs_s, obs_{s-1}, alpha_{s-2}}, each loaded 2..3 steps before its use
                for (int i = rem % 4; i > 0; --i) {
                    bstep(po.at(0), pa[-RS], pg, std::true_type());
                    po.move(-1);
                    pa -= RS;
                    pg -= RS;
                }
                rem -= rem % 4;
                if (rem > 0) {
                    // exponents of the stored rows: same record index as the rows themselves
                    const int32_t *pea = ea_rows + (rec0 + rem) * 64 + cl;
                    ObsIn xh = po.at(0), xl = po.at(-1), yh, yl;
                    double2 xa = pa[-2 * RS], ya;
                    int xe = EXPO ? pea[-2 * 64] : 0, ye = 0;
                    int stop = (PHASE == PH_P2 && carry.cap > 0 && rem > carry.cap) ? carry.cap : 0;
#pragma clang loop unroll(disable)
                    for (int stretch = 0; stretch < 2; ++stretch) { // (see the quad loop above)
                    for (; rem > stop; rem -= 4) {
                        yh = po.at(-2);
                        yl = po.at(-3);
                        ya = pa[-4 * RS];
                        if constexpr (EXPO)
                            ye = pea[-4 * 64];
                        bpair(xh, xl, xa, xe, pg, sc_at<0>(), sc_at<1>());
                        if (rem > 4) {
                            xh = po.at(-4);
                            xl = po.at(-5);
                            xa = pa[-6 * RS];
                            if constexpr (EXPO)
                                xe = pea[-6 * 64];
                        }
                        bpair(yh, yl, ya, ye, pg - 2 * RS, sc_at<2>(), sc_at<3>());
                        po.move(-4);
                        pa -= 4 * RS;
                        pg -= 4 * RS;
                        pea -= 4 * 64;
                    }
                    if (stop == 0)
                        break;
                    if (carry.b_out && !first) {
                        *reinterpret_cast<double2 *>(carry.b_out + (g - 1) * N + 2 * q) =
                            make_double2(b2[0], b2[1]);
                        if (q == 0)
                            carry.db_out[g - 1] = stop + 1;
                    }
                    stop = 0;
                    }
                }
            } else {
                constexpr int PF = ESTEP_PF_B;
                for (int i = rem % (2 * PF); i > 0; --i) { // as in the forward sweep
                    bstep(po.at(0), pa[-RS], pg, std::true_type());
                    po.move(-1);
                    pa -= RS;
                    pg -= RS;
                }
                rem -= rem % (2 * PF);
                if (rem > 0) {
                    ObsIn x[PF], y[PF];
                    double2 u[PF], v[PF];
#pragma unroll
                    for (int j = 0; j < PF; ++j) {
                        x[j] = po.at(-j);
                        u[j] = pa[-(j + 1) * RS];
                    }
                    for (; rem > 0; rem -= 2 * PF) {
#pragma unroll
                        for (int j = 0; j < PF; ++j) {
                            y[j] = po.at(-(PF + j));
                            v[j] = pa[-(PF + j + 1) * RS];
                        }
                        unrolled<PF>([&](auto j) { bstep(x[j], u[j], pg - j * RS, sc_at<j>()); });
                        if (rem > 2 * PF) {
#pragma unroll
                            for (int j = 0; j < PF; ++j) {
                                x[j] = po.at(-(2 * PF + j));
                                u[j] = pa[-(2 * PF + j + 1) * RS];
                            }
                        }
                        unrolled<PF>([&](auto j) {
                            bstep(y[j], v[j], pg - (PF + j) * RS, sc_at<PF + j>());
                        });
                        po.move(-2 * PF);
                        pa -= 2 * PF * RS;
                        pg -= 2 * PF * RS;
                    }
                }
            }
        }
        // step 0
        if (first) {
            double d[2] = {0.0, 0.0};
            if constexpr (KIND == EMIT_GAUSS) {
                d[0] = in0.o - em.mu[0];
                d[1] = in0.o - em.mu[1];
            }
            consume(in0, d, pg0);
            *reinterpret_cast<double2 *>(gamma0 + (int64_t)k * N + 2 * q) =
                make_double2(gam[0], gam[1]);
        } else {
            bstep(in0, aent, pg0, std::true_type());
            if constexpr (SPEC) // beta one step before this chunk: what the previous chunk assumed
                *reinterpret_cast<double2 *>(b_entry + g * N + 2 * q) = make_double2(b2[0], b2[1]);
        }
        if constexpr (!CAREFUL) {
            // Self-check of the branch-free sweep: every step contributes unit gamma mass.  A chunk
            // that does not come out at its length (an intermediate product in the denormal range
            // that the scale tracking did not see, anything non-finite) is reported like a tiny
            // vector: the host repeats the E-step with the per-step-checked kernels.
            const double mass = grp_sum<H>(sg[0] + sg[1]);
            if (!(fabs(mass - (double)len) <= 1e-8 * (double)len))
                hmin = 0;
            // ... and a state that carries weight although its (A (p o beta)) entry is below 2^-910
            // -- gamma_i = w_i r_i with w_i = alpha_i / S at 2^850 or more -- has that entry, or the
            // products it is summed from, rounded in the denormal range: gamma still sums to one, the
            // counts do not (one of them off by 2e-6 in tests/golden/cases/disc4_M1150_9501_20.npz).
            if (wmax >= ((1023 + 850) << 20))
                hmin = 0;
        }
        } // HAS_BWD
        } // role_b
        } // !FWDONLY
    }

    if constexpr (!CAREFUL) {
        if (__ballot(small_hi(hmin)) != 0ull && (threadIdx.x & 63) == 0)
            atomicAdd(&flags[2], 1u);
    }
#ifdef ESTEP_CLOCKPROBE
    if (threadIdx.x == 0) {
        unsigned long long *o = reinterpret_cast<unsigned long long *>(flags + 4) + 8 * (size_t)blockIdx.x;
        o[0] = pc0;
        o[1] = __builtin_readcyclecounter();
        o[2] = pr0;
        o[3] = wall_clock64();
        o[4] = pr1;
        o[5] = pr2;
        o[6] = pr3;
    }
#endif

    // ---------------- workgroup reduction of the register statistics ----------------------
    // entry e of the statistics vector is owned by lane q = (state of e) / 2; sum over the
    // chunks of the wavefront (lanes with equal q), then over wavefronts through LDS
    if constexpr (HAS_BWD) {
        const int lane = threadIdx.x & 63;
        const int wv = threadIdx.x >> 6;
        double *mine = red + wv * SL::S;
        auto chunk_sum = [&](double v) {
#pragma unroll
            for (int h = 32; h >= H; h >>= 1)
                v += __shfl_xor(v, h, 64);
            return v;
        };
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double v = chunk_sum(Cacc[b][j]);
                if (lane < H)
                    mine[(2 * q + b) * N + j] = v;
            }
            const double v = chunk_sum(sg[b]);
            if (lane < H)
                mine[SL::NC + 2 * q + b] = v;
            if constexpr (KIND == EMIT_GAUSS) {
                const double v1 = chunk_sum(sd[b]);
                const double v2 = chunk_sum(sdd[b]);
                if (lane < H) {
                    mine[SL::NC + N + 2 * q + b] = v1;
                    mine[SL::NC + 2 * N + 2 * q + b] = v2;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < SL::S; i += blockDim.x) {
            double v = red[i];
#pragma unroll
            for (int w = 1; w < NW; ++w)
                v += red[w * SL::S + i];
            partials[(int64_t)blockIdx.x * SL::S + i] = v;
        }
        if constexpr (KIND == EMIT_DISC)
            for (int i = threadIdx.x; i < Mlds * N; i += blockDim.x)
            {
                double v = dstat0[i];
                for (int w = 1; w < dcopies; ++w)
                    v += dstat0[w * (m.M * N) + i];
                disc_partials[(int64_t)blockIdx.x * (m.M * N) + i] = v;
            }
    }
}

#define ESTEP_ARGS                                                                               \
    const Model<N> m, const Chunks ch, const void *obs_ci, const void *obs_rm, const int64_t *toff, \
        const double *Bt_g, double *alpha_entry, double *beta_exit, double *a_exit,              \
        double *b_entry, int W, double *ws, double *gamma_ci, double *logL_chunk, double *gamma0, \
        double *partials, double *disc_partials, unsigned int *flags, int32_t *ea_ci,          \
        const Carry carry
#define ESTEP_PASS                                                                               \
    m, ch, obs_ci, obs_rm, toff, Bt_g, alpha_entry, beta_exit, a_exit, b_entry, W, ws, gamma_ci,   \
        logL_chunk, gamma0, partials, disc_partials, flags, ea_ci, carry

// the sweeps that carry the xi accumulators: two wavefronts per SIMD, up to 256 VGPRs
template <int N, int KIND, bool SPEC, bool GAMMA, bool CAREFUL, int PHASE = PH_ALL>
__global__ __launch_bounds__(32 * N)
    __attribute__((amdgpu_waves_per_eu(ESTEP_WAVES, ESTEP_WAVES))) void k_estep(ESTEP_ARGS)
{
    static_assert(PHASE == PH_ALL || PHASE == PH_P2, "use k_estep_light");
    if constexpr (KIND == EMIT_DISC) {
        if (m.bt_global) { // alphabet beyond the LDS: tables in global memory (uniform branch)
            estep_body<N, KIND, SPEC, GAMMA, CAREFUL, PHASE, true>(ESTEP_PASS);
            return;
        }
    }
    estep_body<N, KIND, SPEC, GAMMA, CAREFUL, PHASE>(ESTEP_PASS);
}

// forward sweeps / warm-ups only (PH_P1, PH_FWDROWS): four wavefronts per SIMD
template <int N, int KIND, bool SPEC, bool GAMMA, bool CAREFUL, int PHASE>
__global__ __launch_bounds__(32 * N) __attribute__((amdgpu_waves_per_eu(4))) void k_estep_light(
    ESTEP_ARGS)
{
    static_assert(PHASE == PH_P1 || PHASE == PH_FWDROWS, "use k_estep");
    if constexpr (KIND == EMIT_DISC) {
        if (m.bt_global) {
            estep_body<N, KIND, SPEC, GAMMA, CAREFUL, PHASE, true>(ESTEP_PASS);
            return;
        }
    }
    estep_body<N, KIND, SPEC, GAMMA, CAREFUL, PHASE>(ESTEP_PASS);
}
#undef ESTEP_ARGS
#undef ESTEP_PASS

// debugging aid (BHMM_AMD_POISON=1): every byte of every compute unit's LDS set to 0xFF before a
// kernel that exchanges vectors through LDS starts, so that a kernel that reads LDS it has not written meets NaNs instead of whatever the
// previous kernel on that compute unit left there
[[maybe_unused]] static __global__ void k_lds_poison(int words)
{
    extern __shared__ unsigned int lds_all[];
    for (int e = threadIdx.x; e < words; e += blockDim.x)
        lds_all[e] = 0xFFFFFFFFu;
    __syncthreads();
    if (lds_all[(threadIdx.x * 7) % words] != 0xFFFFFFFFu) // (keeps the stores)
        __builtin_trap();
}
[[maybe_unused]] static void lds_poison(hipStream_t stream)
{
    static const bool poison = getenv("BHMM_AMD_POISON") != nullptr;
    if (!poison)
        return;
    const int bytes = 160 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lds_poison),
                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(k_lds_poison, dim3(2048), dim3(1024), bytes, stream, bytes / 4);
    (void)hipGetLastError();
}

} // namespace bhmm
