// wide_kernels.hpp -- E-step kernels for 9..64 hidden states (BASELINE configs[3]: N = 64).
//
// The state vector no longer fits one lane, so the mapping flips: NP = 16, 32 or 64 lanes
// cooperate on one trajectory, lane j owns state j (64/NP trajectories per wavefront), and the
// recursions run serially in t per trajectory.  Chunking in time (as for N <= 8) would cost N
// forward recursions per step in the prescan -- 64x at N = 64 -- which is more than the chip
// gains from the extra parallelism at the trajectory counts of configs[3]; see DESIGN.md.
//
//   k_wide_fwd : alpha (row-major, the reference layout) + per-trajectory log-likelihood
//                (_hidden.c:16-66).  Lane j keeps column j of A in registers; alpha_t is
//                exchanged through LDS (wave-uniform broadcast reads).
//   k_wide_bwd : beta in registers only; gamma, xi row i (C'[i][:] in registers, rank-1
//                update per step), emission statistics.  A lives in LDS with padded rows
//                (lane i reads row i conflict-free); p o beta is exchanged through LDS.
//   k_wide_finalize : fixed-order sums over trajectories -> packed statistics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "estep_sweep.hpp"

namespace bhmm {

// model parameters of the wide family live in one device buffer:
//   A[n*n] | pi[n] | mu[n] | 1/sigma[n] | 1/(sqrt(2pi) sigma)[n] | sigma[n] | ga[n] | gb[n]
struct WideModel {
    const double *A, *pi, *mu, *isig, *cnorm, *sigma;
    const double *ga, *gb; // gauss_pdf_issue(): per-state constants (host_common.hpp: gauss_pdf_constants)
    double gmg;            // ... and the common magic constant
    const double *B; // [n][M] row-major (discrete)
    int n, M;
};

// gauss_pdf<NANSAFE = true> (estep_sweep.hpp: the gaussian density with constant, exponent and range
// reduction fused, 18 instructions where cn * exp(-z*z/2) by range reduction took 24), Horner steps pinned
// like above.  A NaN observation gives 0 here, as it always did in this family (-> outlier row).
__device__ __forceinline__ double gauss_pdf_issue(double d, double a, double b, double MG)
{
    const double u = fmin(fma(d * d, a, b), 1.0);
    const double t = MG - u;
    const double w = (t - MG) + u;
    // the polynomial and the exponent as ONE block (separate statements: an s_nop between every two
    // of them, see dot16); same ten fused multiply-adds, coefficients in scalar registers
    double q;
    asm("v_fma_f64 %0, %2, %1, %3\n\t"
        "v_fma_f64 %0, %0, %1, %4\n\t"
        "v_fma_f64 %0, %0, %1, %5\n\t"
        "v_fma_f64 %0, %0, %1, %6\n\t"
        "v_fma_f64 %0, %0, %1, %7\n\t"
        "v_fma_f64 %0, %0, %1, %8\n\t"
        "v_fma_f64 %0, %0, %1, %9\n\t"
        "v_fma_f64 %0, %0, %1, %10\n\t"
        "v_fma_f64 %0, %0, %1, %11\n\t"
        "v_fma_f64 %0, %0, %1, 1.0\n\t"
        "v_ldexp_f64 %0, %0, %12"
        : "=&v"(q)
        : "v"(w), "v"(0x1.e3991e644e6abp+92), "s"(-0x1.b6740fc28f781p+84), "s"(0x1.62c157ee59177p+76),
          "s"(-0x1.ffcb55e82f22cp+67), "s"(0x1.4309126056718p+59), "s"(-0x1.5d87fe9cc5d6fp+50),
          "s"(0x1.3b2ab6fbde0f7p+41), "s"(-0x1.c6b08d703d48ap+31), "s"(0x1.ebfbdff82c3b9p+21),
          "s"(-0x1.62e42fefa3a17p+11), "v"(__double2loint(t)));
    return q;
}

// ---- 64 states, one trajectory segment per wavefront: cross-lane forms of gfx950 ----------------
// A matrix-vector product needs every lane to see every element of the state vector.  Going
// through LDS costs a write, a wait and 32 broadcast reads per step, and with one wavefront per
// SIMD (the recursion is serial, the registers are full) nothing hides those latencies.  Instead:
//   * v_permlane32_swap / v_permlane16_swap make four copies of the vector in which every row of
//     16 lanes holds row r of the original (6 swaps);
//   * v_fmac_f64 with the DPP modifier row_newbcast:i multiplies by lane i of the own row, i.e.
//     by element 16 r + i, in the same instruction: 64 FMAs per product and no other traffic.
// The DPP instructions are inline assembly (the compiler does not fold a 64-bit row_newbcast
// move into the FMA); a DPP read of a VGPR needs two wait states after the VALU write of that
// VGPR, which the first instruction of every row group provides itself.
constexpr __host__ __device__ int wide_pitch(int np) { return np + 1; } // odd: lane i reads row i conflict-free

struct Rows4 {
    double r[4];
};

__device__ __forceinline__ void swap32_f64(double &x, double &y)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi.x, (int)lo.x);
    y = __hiloint2double((int)hi.y, (int)lo.y);
}

__device__ __forceinline__ void swap16_f64(double &x, double &y)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi.x, (int)lo.x);
    y = __hiloint2double((int)hi.y, (int)lo.y);
}

// NP = 32 (two segments per wavefront, rows 0-1 and 2-3): r[k], k = 0 / 1, holds in every row of
// a segment that segment's row k.  NP = 16: a row is a segment, the vector itself.
template <int NP>
__device__ __forceinline__ Rows4 rows_of_group(double v);

// r[k]: every row of 16 lanes holds row k of v
__device__ __forceinline__ Rows4 rows_of(double v)
{
    Rows4 o;
    double lo = v, hi = v;
    swap32_f64(lo, hi); // lo: rows (0,1,0,1)   hi: rows (2,3,2,3)
    o.r[0] = lo;
    o.r[1] = lo;
    swap16_f64(o.r[0], o.r[1]); // (0,0,0,0) and (1,1,1,1)
    o.r[2] = hi;
    o.r[3] = hi;
    swap16_f64(o.r[2], o.r[3]);
    return o;
}

// v[k] (k = 0..3) are four 64-lane vectors; on return row k of v[I] is row I of the old v[k]:
// the 16 x 4 / 4 x 16 operand layout of v_mfma_f64_16x16x4 (lane = 16 k + i) for block I.
__device__ __forceinline__ void rows_transpose4(double (&v)[4])
{
    swap32_f64(v[0], v[2]);
    swap32_f64(v[1], v[3]);
    swap16_f64(v[0], v[1]);
    swap16_f64(v[2], v[3]);
}

typedef double wide_d4 __attribute__((ext_vector_type(4)));

template <int NP>
__device__ __forceinline__ Rows4 rows_of_group(double v)
{
    if constexpr (NP == 64) {
        return rows_of(v);
    } else if constexpr (NP == 32) {
        Rows4 o;
        o.r[0] = v;
        o.r[1] = v;
        swap16_f64(o.r[0], o.r[1]); // rows (0,0,2,2) and (1,1,3,3)
        o.r[2] = o.r[3] = 0.0;
        return o;
    } else {
        Rows4 o;
        o.r[0] = v;
        o.r[1] = o.r[2] = o.r[3] = 0.0;
        return o;
    }
}

template <int I, bool FIRST>
__device__ __forceinline__ void fmac_bcast(double &acc, const double &src, const double &w)
{
    if constexpr (FIRST)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(src), "v"(w), "n"(I));
    else
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                     : "+v"(acc)
                     : "v"(src), "v"(w), "n"(I));
}

// acc[i & 3] += v[16 ROW + i] * w(i), i = 0..15, with v given as its row copy `src`.
// ONE assembly block: as sixteen statements the compiler separates every fourth from its
// predecessor with an s_nop (its hazard model counts an inline statement as zero wait states, so the
// accumulator written four instructions earlier looks freshly written) -- a seventh of the forward
// kernel's instruction stream was such padding.
template <typename WF>
__device__ __forceinline__ void dot16(double (&acc)[4], const double &src, WF &&w)
{
#define BHMM_W(I) "v"(w(std::integral_constant<int, I>{}))
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %17 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %4, %18 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %4, %19 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %4, %20 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                 : "v"(src), BHMM_W(0), BHMM_W(1), BHMM_W(2), BHMM_W(3), BHMM_W(4), BHMM_W(5), BHMM_W(6),
                   BHMM_W(7), BHMM_W(8), BHMM_W(9), BHMM_W(10), BHMM_W(11), BHMM_W(12), BHMM_W(13),
                   BHMM_W(14), BHMM_W(15));
#undef BHMM_W
}

// sum / max over all 64 lanes, result in every lane (fixed order)
__device__ __forceinline__ double wave64_sum(double v)
{
    v += xchg_f64<1>(v);
    v += xchg_f64<2>(v);
    v += xchg_f64<4>(v);
    {
        const int lo = dpp_i32<0x140>(__double2loint(v)); // row_mirror
        const int hi = dpp_i32<0x140>(__double2hiint(v));
        v += __hiloint2double(hi, lo);
    }
    double a = v, b = v;
    swap32_f64(a, b); // rows (0,1,0,1) / (2,3,2,3)
    a += b;
    b = a;
    swap16_f64(a, b); // all rows: 0+2 / 1+3
    return a + b;
}

__device__ __forceinline__ int wave64_max(int v)
{
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    v = max(v, xchg_i32<1>(v));
    v = max(v, xchg_i32<2>(v));
    v = max(v, xchg_i32<4>(v));
    v = max(v, dpp_i32<0x140>(v));
    u2 s = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    v = max((int)s.x, (int)s.y);
    s = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    return max((int)s.x, (int)s.y);
}

template <int NP>
__device__ __forceinline__ double wgroup_sum(double v)
{
    // lanes of one row (16) first, on DPP; then across rows with wave shuffles
    if constexpr (NP == 64)
        return wave64_sum(v);
    v += xchg_f64<1>(v);
    v += xchg_f64<2>(v);
    v += xchg_f64<4>(v);
    {
        const int lo = dpp_i32<0x140>(__double2loint(v)); // row_mirror: i <-> 15 - i
        const int hi = dpp_i32<0x140>(__double2hiint(v));
        v += __hiloint2double(hi, lo);
    }
    if constexpr (NP >= 32)
        v += __shfl_xor(v, 16, 64);
    if constexpr (NP >= 64)
        v += __shfl_xor(v, 32, 64);
    return v;
}

template <int NP>
__device__ __forceinline__ int wgroup_max(int v)
{
    if constexpr (NP == 64)
        return wave64_max(v);
    v = max(v, xchg_i32<1>(v));
    v = max(v, xchg_i32<2>(v));
    v = max(v, xchg_i32<4>(v));
    v = max(v, dpp_i32<0x140>(v));
    if constexpr (NP >= 32)
        v = max(v, __shfl_xor(v, 16, 64));
    if constexpr (NP >= 64)
        v = max(v, __shfl_xor(v, 32, 64));
    return v;
}


template <int NP>
__device__ __forceinline__ unsigned long long wgroup_mask(int lane)
{
    if constexpr (NP == 64)
        return ~0ull;
    else
        return ((1ull << NP) - 1) << (lane / NP * NP);
}

// What one step reads from the observation stream.  Loading it is separated from using it so
// that the recursions can fetch WIDE_PF steps ahead: they are serial in t with one or two
// wavefronts per SIMD, and a load issued in the step that needs it costs a full memory latency
// (about 1 us under load) per step.
#ifndef WIDE_PF
#define WIDE_PF 8
#endif
struct WideIn {
    double o; // Gaussian: the observation; explicit: p_obs of my state
    int sym;  // discrete: the symbol
};

template <int KIND>
__device__ __forceinline__ WideIn wide_load(const WideModel &m, int j, bool real, int64_t gt,
                                            const void *obs_rm)
{
    WideIn in;
    in.o = 0.0;
    in.sym = 0;
    if constexpr (KIND == EMIT_GAUSS)
        in.o = static_cast<const double *>(obs_rm)[gt];
    else if constexpr (KIND == EMIT_DISC)
        in.sym = static_cast<const int32_t *>(obs_rm)[gt];
    else
        in.o = real ? static_cast<const double *>(obs_rm)[gt * m.n + j] : 0.0;
    return in;
}

// emission probability of MY state (+ outlier rule over the group)
// RESCUE (the kernels that normalise by a sum and its reciprocal every step): a row whose entries
// all lie below 2^-959 -- densities in the denormal range, an observation ~38 sigma from every
// state -- is returned times 2^900, *pexp = 900 (else 0): the reciprocal of a denormal sum is
// infinite.  The backward recursion does not see the factor; the forward pass takes it off its
// exponent count.
template <int NP, int KIND, bool RESCUE = false>
__device__ __forceinline__ double wide_emit(const WideModel &m, int j, bool real, const WideIn &in,
                                            double mu_j, double ga_j, double gb_j,
                                            unsigned long long gmask, int *pexp = nullptr)
{
    double p = 0.0;
    if constexpr (KIND == EMIT_GAUSS) {
        p = gauss_pdf_issue(in.o - mu_j, ga_j, gb_j, m.gmg); // lanes without a state: (0, 1) -> 0
        if constexpr (RESCUE) {
            // a row in the denormal range: the reference's own operation order (_gaussian.c:5-21)
            if (__builtin_expect((__ballot(p >= 0x1p-959) & gmask) == 0ull, 0)) {
                const double z = real ? (in.o - mu_j) / m.sigma[j] : 0.0;
                p = real ? m.cnorm[j] * exp(-0.5 * z * z) : 0.0;
            }
        }
        if ((__ballot(p != 0.0) & gmask) == 0ull)
            p = real ? 1.0 : 0.0; // outputmodel.py:126-130
    } else if constexpr (KIND == EMIT_DISC) {
        p = real ? m.B[(int64_t)j * m.M + in.sym] : 0.0;
    } else {
        p = in.o;
    }
    if constexpr (RESCUE) {
        int e = 0;
        if ((__ballot(p >= 0x1p-959) & gmask) == 0ull && (__ballot(p != 0.0) & gmask) != 0ull) {
            p = ldexp(p, 900);
            e = 900;
        }
        if (pexp)
            *pexp = e;
    }
    return p;
}

// p o beta in the denormal range although neither factor is (a state that explains this observation
// well but the future badly): sums formed from it would be denormal, their reciprocals infinite.
// The kernels that normalise every step only see the vector up to a factor: p times 2^900 (exact).
template <int NP>
__device__ __forceinline__ double rescue_product(double p, double b, unsigned long long gmask)
{
    const double x = p * b;
    if (__builtin_expect((__ballot(x >= 0x1p-959) & gmask) == 0ull, 0))
        if ((__ballot(x != 0.0) & gmask) != 0ull)
            return ldexp(p, 900);
    return p;
}

// Segments: a trajectory may be cut into time segments that are processed by different lane
// groups (one segment per trajectory = the plain serial recursion).  With several segments the
// boundary vectors come from warm-ups over the W steps before / after the segment and are
// verified afterwards by k_wide_check (same scheme as k_estep<..., SPEC>, see estep_sweep.hpp); alpha of
// the step before a segment is read from the previous segment's output by the backward pass.
struct Segs {
    const int32_t *traj; // trajectory of the segment
    const int64_t *t0;   // first step inside the trajectory
    const int32_t *len;
    int nseg;
    int W;
    // backward pass only: where the forward pass, if it runs on a finer plan, starts a segment of
    // its own inside this one (-1: nowhere) -- alpha there continues another chain of rescalings
    const int64_t *fmid = nullptr;
};

// LAZY (E-step only): alpha is carried un-normalised, up to a power of two that is refreshed every
// fourth step -- the per-step sum over the states and its reciprocal leave the serial chain.  The
// backward pass only uses alpha through scale-free ratios, and the log-likelihood of a segment
// telescopes: log sum(alpha at the last step) - log sum(alpha at the entry) + ln 2 * (exponents
// removed in between).  A vector whose largest element falls below 2^-900 between two refreshes
// raises flags[2]; the host then repeats the E-step with the per-step normalisation.
#define WIDE_TROUBLE_EXP (-900)
// FULL: n == NP (no padded lanes; the row pitch is a constant)
template <int NP, int KIND, bool LAZY = false, bool FULL = false>
__global__ __launch_bounds__(64) void k_wide_fwd(const WideModel m, const int64_t *off, const Segs sg,
                                                 const void *obs_rm, double *alpha_rm,
                                                 double *logL_seg, double *a_entry, double *a_exit,
                                                 unsigned int *flags = nullptr)
{
    constexpr int GP = 64 / NP;
    const int lane = threadIdx.x;
    const int gi = lane / NP, j = lane % NP;
    // one segment per wavefront (64 states): everything about the segment is wave-uniform
    const int s = (NP == 64) ? (int)blockIdx.x : (int)blockIdx.x * GP + gi;
    if (s >= sg.nseg)
        return;
    const int n = FULL ? NP : m.n;
    const bool real = FULL || j < n;
    const int k = sg.traj[s];
    const int64_t o0 = off[k];
    const int64_t t0 = sg.t0[s], t1 = t0 + sg.len[s];
    if (t1 <= t0) {
        if (j == 0)
            logL_seg[s] = 0.0;
        return;
    }
    const unsigned long long gmask = wgroup_mask<NP>(lane);
    double Acol[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        Acol[i] = (real && i < n) ? m.A[(int64_t)i * n + j] : 0.0;
    const double mu_j = (KIND == EMIT_GAUSS && real) ? m.mu[j] : 0.0;
    const double ga_j = (KIND == EMIT_GAUSS && real) ? m.ga[j] : 0.0; // gauss_pdf_issue() constants
    const double gb_j = (KIND == EMIT_GAUSS && real) ? m.gb[j] : 1.0;
    const double pi_j = real ? m.pi[j] : 0.0;

    const int64_t tw = (t0 - sg.W > 0) ? t0 - sg.W : 0; // warm-up start (0: exact start)
    double a = real ? 1.0 / (double)n : 0.0, P = 1.0;
    int eP = 0;
    double S_start = 1.0; // LAZY: sum of the vector the segment starts from
    bool trouble = false;
    WideIn ring[WIDE_PF];
#pragma unroll
    for (int u = 0; u < WIDE_PF; ++u)
        ring[u] = wide_load<KIND>(m, j, real, o0 + (tw + u < t1 ? tw + u : t1 - 1), obs_rm);
    // steps are counted relative to the warm-up start in 32 bits (scalar compares)
    const int nsteps = (int)(t1 - tw), r0 = (int)(t0 - tw);
    const bool from_start = tw == 0;
    for (int rb = 0; rb < nsteps; rb += WIDE_PF) {
#pragma unroll
        for (int u = 0; u < WIDE_PF; ++u) {
            const int r = rb + u;
            if (r >= nsteps)
                break;
            const int64_t t = tw + r;
            const WideIn in = ring[u];
            {
                const int rn = r + WIDE_PF < nsteps ? r + WIDE_PF : nsteps - 1;
                ring[u] = wide_load<KIND>(m, j, real, o0 + tw + rn, obs_rm);
            }
            int pexp = 0;
            const double p = wide_emit<NP, KIND, !LAZY>(m, j, real, in, mu_j, ga_j, gb_j, gmask, &pexp);
            double nj;
            if (from_start && r == 0) {
                nj = pi_j * p;
            } else {
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                {
                    const Rows4 ar = rows_of_group<NP>(a);
                    unrolled<NP / 16>([&](auto rc) {
                        constexpr int r = decltype(rc)::value;
                        dot16(acc, ar.r[r], [&](auto ic) -> const double & {
                            return Acol[16 * r + decltype(ic)::value];
                        });
                    });
                }
                nj = ((acc[0] + acc[1]) + (acc[2] + acc[3])) * p;
            }
            if constexpr (LAZY) {
                a = nj;
                if ((u & 3) == 3) {
                    const int E = wgroup_max<NP>(a > 0.0 ? exponent_of(a) : -(1 << 28));
                    trouble |= E < WIDE_TROUBLE_EXP;
                    a = ldexp(a, -E);
                    if (r >= r0)
                        eP += E;
                }
                if (r >= r0) {
                    if (real)
                        alpha_rm[(o0 + t) * n + j] = a;
                } else if (r == r0 - 1) {
                    if (real)
                        a_entry[(int64_t)s * n + j] = a;
                    S_start = wgroup_sum<NP>(a);
                }
            } else {
                double c = wgroup_sum<NP>(nj);
                // the new row in the denormal range although the emission row is not (narrow states:
                // the mass sits where this observation is all but impossible): its sum's reciprocal
                // would be infinite -- times 2^900, exactly, and 900 off the exponent count (the
                // backward pass: rescue_product; found by tests/sweeps/stress_small.py 2001 / 2860)
                if (__builtin_expect(!(c >= 0x1p-959) && c > 0.0, 0)) {
                    nj = ldexp(nj, 900);
                    c = wgroup_sum<NP>(nj);
                    pexp += 900;
                }
                a = nj * fast_rcp(c);
                if (r >= r0) {
                    int e;
                    P = frexp(P * c, &e);
                    eP += e - pexp;
                    if (real)
                        alpha_rm[(o0 + t) * n + j] = a;
                } else if (r == r0 - 1 && real) {
                    a_entry[(int64_t)s * n + j] = a; // the entry vector this segment derived
                }
            }
        }
    }
    if (real)
        a_exit[(int64_t)s * n + j] = a;
    if constexpr (LAZY) {
        const double S_end = wgroup_sum<NP>(a);
        trouble |= !(S_end > 0.0) || !(S_start > 0.0);
        if (j == 0) {
            logL_seg[s] = (log(S_end) - log(S_start)) + (double)eP * 0.693147180559945309417232121458;
            if (trouble)
                atomicOr(&flags[2], 1u);
        }
    } else {
        if (j == 0)
            logL_seg[s] = log(P) + (double)eP * 0.693147180559945309417232121458;
    }
}

// statistics per segment: [n*n C' rows | n sum gamma | (gauss) n sum gamma d | n sum gamma d^2]
// discrete symbol table: [nseg][n][M] (dstat)
// LAZY: beta is refreshed to a power of two every fourth step instead of every step (see k_wide_fwd)
// XIG (64 states, control experiment of round 3, BHMM_AMD_WIDE_XI_GEMM=1): no xi accumulators in this
// kernel; it stores the rows W_{t-1} = p_t o beta_t / S_t instead and the counts are the
// time-parallel GEMM C' = alpha^T W of k_wide_xi_gemm64.
template <int NP, int KIND, bool LAZY = false, bool FULL = false, bool XIG = false>
__global__ __launch_bounds__(64) void k_wide_bwd(const WideModel m, const int64_t *off, const Segs sg,
                                                 const void *obs_rm, const double *alpha_rm,
                                                 double *gamma_rm, double *gamma0, double *part,
                                                 double *dstat, double *b_exit, double *b_entry,
                                                 unsigned int *flags = nullptr, double *Wg = nullptr)
{
    constexpr int GP = 64 / NP;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // fewer than 64 lanes per segment: rows of A in LDS, padded (lane i reads row i); the 64-lane
    // kernel keeps row i in VGPRs and uses no LDS
    constexpr int PITCH = wide_pitch(NP);
    double *sA = smem;                          // [NP][PITCH]
    const int lane = threadIdx.x;
    const int gi = lane / NP, i = lane % NP;
    const int n = FULL ? NP : m.n;
    if constexpr (NP < 64) {
        for (int e = lane; e < NP * NP; e += 64) {
            const int r = e / NP, c = e % NP;
            sA[r * PITCH + c] = (r < n && c < n) ? m.A[(int64_t)r * n + c] : 0.0;
        }
        __syncthreads();
    }
    const int s = (NP == 64) ? (int)blockIdx.x : (int)blockIdx.x * GP + gi;
    if (s >= sg.nseg)
        return;
    const bool real = FULL || i < n;
    const int k = sg.traj[s];
    const int64_t o0 = off[k];
    const int64_t T = off[k + 1] - o0;
    const int64_t t0 = sg.t0[s], t1 = t0 + sg.len[s];
    const int S = n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0);
    double *mypart = part + (int64_t)s * S;
    double *mytab = (KIND == EMIT_DISC) ? dstat + (int64_t)s * n * m.M : nullptr;
    if (KIND == EMIT_DISC && real)
        for (int q = 0; q < m.M; ++q)
            mytab[(int64_t)i * m.M + q] = 0.0;
    // xi accumulators.  Fewer than 64 states: lane i keeps row i of C' and adds w_i x_c per step.
    // 64 states: the rank-1 updates of four consecutive steps are one rank-4 update on the
    // matrix cores -- C'(16 I.., 16 J..) += W_I X_J^T with v_mfma_f64_16x16x4, W_I / X_J being
    // the rows 16 I.. of the four buffered w / x vectors (rows_transpose4) -- so the 64 x 64
    // accumulators live in the accumulation registers, A's row i stays in the vector registers
    // and the step issues 4 matrix instructions instead of 64 FMAs.
    constexpr int NCROW = (NP == 64) ? 1 : NP;
    double Crow[NCROW];
#pragma unroll
    for (int c = 0; c < NCROW; ++c)
        Crow[c] = 0.0;
    wide_d4 Cacc[4][4];
    double wq[4] = {0.0, 0.0, 0.0, 0.0}, xq[4] = {0.0, 0.0, 0.0, 0.0};
    double Arow[NP == 64 ? 64 : 1];
    if constexpr (NP == 64) {
#pragma unroll
        for (int I = 0; I < 4; ++I)
#pragma unroll
            for (int J = 0; J < 4; ++J)
                Cacc[I][J] = wide_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 64; ++c)
            Arow[c] = (real && c < n) ? m.A[(int64_t)i * n + c] : 0.0;
    }
    auto xi_flush = [&]() {
        if constexpr (NP == 64 && !XIG) {
            double ow[4] = {wq[0], wq[1], wq[2], wq[3]}, ox[4] = {xq[0], xq[1], xq[2], xq[3]};
            rows_transpose4(ow);
            rows_transpose4(ox);
#pragma unroll
            for (int I = 0; I < 4; ++I)
#pragma unroll
                for (int J = 0; J < 4; ++J)
                    Cacc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ow[I], ox[J], Cacc[I][J], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                wq[q] = 0.0;
        }
    };
    double sgm = 0.0, sd = 0.0, sdd = 0.0;
    if (t1 > t0) {
        const unsigned long long gmask = wgroup_mask<NP>(lane);
        const double mu_i = (KIND == EMIT_GAUSS && real) ? m.mu[i] : 0.0;
        const double ga_i = (KIND == EMIT_GAUSS && real) ? m.ga[i] : 0.0; // gauss_pdf_issue() constants
        const double gb_i = (KIND == EMIT_GAUSS && real) ? m.gb[i] : 1.0;
        const double *arow = sA + i * PITCH;
        // one backward step: b <- A (p o b), rescaled by a power of two; returns A (p o b)[i]
        double xcur = 0.0; // p o b of the current step (back() sets it)
        Rows4 xrows;       // fewer than 64 states: its row copies
        auto back = [&](double p, double b) {
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
            if constexpr (NP == 64) {
                xcur = p * b;
                const Rows4 xr = rows_of(xcur);
                unrolled<4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    dot16(acc, xr.r[r], [&](auto ic) -> const double & {
                        return Arow[16 * r + decltype(ic)::value];
                    });
                });
            } else {
                // A's row from LDS (NP doubles per lane do not fit beside the xi row), p o b on
                // DPP row broadcasts
                xcur = p * b;
                xrows = rows_of_group<NP>(xcur);
                unrolled<NP / 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    double av[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        av[q] = arow[16 * r + q];
                    dot16(acc, xrows.r[r], [&](auto ic) -> const double & {
                        return av[decltype(ic)::value];
                    });
                });
            }
            return (acc[0] + acc[1]) + (acc[2] + acc[3]);
        };
        bool trouble = false;
        auto rescale = [&](double br) {
            const int E = wgroup_max<NP>(br > 0.0 ? exponent_of(br) : -(1 << 28));
            if constexpr (LAZY)
                trouble |= E < WIDE_TROUBLE_EXP;
            return ldexp(br, -E);
        };

        double b = real ? 1.0 / (double)n : 0.0; // _hidden.c:79-88
        if (t1 < T) { // warm-up from te down to t1: beta at t1 - 1
            const int64_t te = (t1 - 1 + sg.W < T - 1) ? t1 - 1 + sg.W : T - 1;
            WideIn ring[WIDE_PF];
#pragma unroll
            for (int u = 0; u < WIDE_PF; ++u)
                ring[u] = wide_load<KIND>(m, i, real, o0 + (te - u > t1 ? te - u : t1), obs_rm);
            const int nwarm = (int)(te - t1) + 1;
            for (int rb = 0; rb < nwarm; rb += WIDE_PF) {
#pragma unroll
                for (int u = 0; u < WIDE_PF; ++u) {
                    const int r = rb + u;
                    if (r >= nwarm)
                        break;
                    const WideIn in = ring[u];
                    {
                        const int rn = r + WIDE_PF < nwarm ? r + WIDE_PF : nwarm - 1;
                        ring[u] = wide_load<KIND>(m, i, real, o0 + te - rn, obs_rm);
                    }
                    double p = wide_emit<NP, KIND, !LAZY>(m, i, real, in, mu_i, ga_i, gb_i, gmask);
                    if constexpr (!LAZY)
                        p = rescue_product<NP>(p, b, gmask);
                    b = back(p, b);
                    if (!LAZY || (u & 3) == 3)
                        b = rescale(b);
                }
            }
            if (real)
                b_exit[(int64_t)s * n + i] = b;
        }
        double a = real ? alpha_rm[(o0 + t1 - 1) * n + i] : 0.0;
        double gam;
        {
            double g = a * b, Sg = wgroup_sum<NP>(g);
            if (!(Sg >= 0x1p-959) && Sg > 0.0) { // (as in the step below)
                g = a * ldexp(b, 900);
                Sg = wgroup_sum<NP>(g);
            }
            gam = g * fast_rcp(Sg);
        }
        // rings: observation of step t and alpha of step t - 1, WIDE_PF steps ahead
        WideIn ro[WIDE_PF];
        double ra[WIDE_PF];
        const int nmain = (int)(t1 - t0); // step r of the loop is t = t1 - 1 - r
        constexpr bool SPARSE_SUM = LAZY && NP == 64;
        // alpha_t was rescaled by the forward pass where (t - tw) % 4 == 3, tw its warm-up start
        const int phase = (int)((t1 - 1 - ((t0 - sg.W > 0) ? t0 - sg.W : 0)) & 3);
        const int64_t tmid = sg.fmid ? sg.fmid[s] : -1;
        double rS_keep = 0.0;
        auto fetch = [&](int u, int r) {
            const int64_t tt = t1 - 1 - (r < nmain ? r : nmain - 1);
            ro[u] = wide_load<KIND>(m, i, real, o0 + tt, obs_rm);
            ra[u] = real ? alpha_rm[(o0 + (tt > 0 ? tt - 1 : 0)) * n + i] : 0.0;
        };
#pragma unroll
        for (int u = 0; u < WIDE_PF; ++u)
            fetch(u, u);
        const bool to_start = t0 == 0;
        for (int rb = 0; rb < nmain; rb += WIDE_PF) {
#pragma unroll
            for (int u = 0; u < WIDE_PF; ++u) {
                const int r = rb + u;
                // no early exit: the matrix-core accumulators must have one definition per
                // iteration (a second loop exit makes the compiler copy all 128 of them)
                if (r < nmain) {
                const int64_t t = t1 - 1 - r;
                const bool last = r == nmain - 1;
                const WideIn in = ro[u];
                const double ap = ra[u];
                fetch(u, r + WIDE_PF);
                double p = wide_emit<NP, KIND, !LAZY>(m, i, real, in, mu_i, ga_i, gb_i, gmask);
                if constexpr (!LAZY)
                    p = rescue_product<NP>(p, b, gmask);
                // consume gamma_t
                sgm += gam;
                if constexpr (KIND == EMIT_GAUSS) {
                    const double d = in.o - mu_i;
                    const double gd = gam * d;
                    sd += gd;
                    sdd = fma(gd, d, sdd);
                }
                if constexpr (KIND == EMIT_DISC)
                    if (real)
                        mytab[(int64_t)i * m.M + in.sym] += gam; // row i is private to this lane
                if (gamma_rm && real)
                    gamma_rm[(o0 + t) * n + i] = gam;
                if (to_start && last) { // t == 0
                    if (real)
                        gamma0[(int64_t)k * n + i] = gam;
                } else {
                    double br = back(p, b);
                    double q = ap * br;
                    // sum_i alpha_{t-1}[i] (A (p_t o beta_t))[i] = sum_j alpha_t[j] beta_t[j]: as long as
                    // neither vector has been rescaled, the normaliser of gamma is the one of the
                    // step before.  The lazily scaled forward pass rescales every fourth step (known
                    // phase) and beta is rescaled on the same steps here, so three of four steps reuse
                    // the reciprocal (64 states; the chain restarts after four steps: no drift).
                    bool need = true;
                    if constexpr (SPARSE_SUM)
                        need = r == 0 || last || ((phase - r) & 3) == 3 || t == tmid;
                    if (need) {
                        double Ssum = wgroup_sum<NP>(q);
                        // alpha concentrated on states whose (A (p o beta)) is in the denormal range
                        // although p o beta as a whole is not (sparse A, narrow states): the sum is
                        // denormal, its reciprocal infinite.  Lazy kernels report it (the host
                        // repeats the E-step with this instantiation's non-lazy twin); here the
                        // product is redone with p times 2^900 (exact), which gamma, xi and the
                        // rescaled beta do not see.
                        if constexpr (LAZY) {
                            trouble |= !(Ssum >= 0x1p-959);
                        } else if (__builtin_expect(!(Ssum >= 0x1p-959) && Ssum > 0.0, 0)) {
                            br = back(ldexp(p, 900), b);
                            q = ap * br;
                            Ssum = wgroup_sum<NP>(q);
                        }
                        rS_keep = fast_rcp(Ssum);
                    }
                    const double rS = rS_keep;
                    gam = q * rS;
                    const double w = ap * rS;
                    if constexpr (NP == 64 && XIG) {
                        if (real)
                            Wg[(o0 + t - 1) * n + i] = xcur * rS;
                        (void)w;
                    } else if constexpr (NP == 64) {
                        wq[u & 3] = w;
                        xq[u & 3] = xcur;
                    } else {
                        unrolled<NP / 16>([&](auto rc) {
                            constexpr int r = decltype(rc)::value;
                            unrolled<16>([&](auto ic) {
                                constexpr int c = decltype(ic)::value;
                                fmac_bcast<c, c == 0>(Crow[16 * r + c], xrows.r[r], w);
                            });
                        });
                    }
                    bool refresh = !LAZY || (u & 3) == 3;
                    if constexpr (SPARSE_SUM)
                        refresh = ((phase - (r + 1)) & 3) == 3;
                    b = refresh ? rescale(br) : br;
                    if (last && real) // beta one step before this segment, as derived here
                        b_entry[(int64_t)s * n + i] = b;
                }
                }
                if constexpr (NP == 64)
                    if ((u & 3) == 3)
                        xi_flush(); // slots of steps that did not run hold w = 0
            }
        }
        if constexpr (LAZY) {
            // self-check of the lazily scaled pass: unit gamma mass per step (else: non-lazy rerun)
            const double mass = wgroup_sum<NP>(sgm);
            trouble |= !(fabs(mass - (double)nmain) <= 1e-8 * (double)nmain);
            if (trouble && i == 0)
                atomicOr(&flags[2], 1u);
        }
    }
    if constexpr (NP == 64) {
        // C / D layout of v_mfma_f64_16x16x4: column = lane & 15, row = (lane >> 4) + 4 r
#pragma unroll
        for (int I = 0; I < 4; ++I)
#pragma unroll
            for (int J = 0; J < 4; ++J)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * I + (lane >> 4) + 4 * r, col = 16 * J + (lane & 15);
                    if (FULL || (row < n && col < n))
                        mypart[(int64_t)row * n + col] = XIG ? 0.0 : Cacc[I][J][r];
                }
    }
    if (real) {
        if constexpr (NP < 64) {
#pragma unroll
            for (int c = 0; c < NP; ++c)
                if (c < n)
                    mypart[(int64_t)i * n + c] = Crow[c];
        }
        mypart[n * n + i] = sgm;
        if constexpr (KIND == EMIT_GAUSS) {
            mypart[n * n + n + i] = sd;
            mypart[n * n + 2 * n + i] = sdd;
        }
    }
}

// xi counts of 64 states as a time-parallel GEMM (control experiment, see k_wide_bwd<.., XIG>):
// every wavefront owns a stretch of time steps and the full 64 x 64 tile -- 16 v_mfma_f64_16x16x4
// per four steps on eight 512-byte loads, accumulators in 128 registers -- and leaves its partial
// sum in xipart[split]; k_wide_xi_reduce adds them in split order into segment 0's block of `part`.
[[maybe_unused]] static __global__ __launch_bounds__(64) void k_wide_xi_gemm64(const double *alpha, const double *W,
                                                        int64_t total, int nsplit, double *xipart)
{
    const int lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
    const int64_t per = ((total + nsplit - 1) / nsplit + 3) / 4 * 4;
    const int64_t tb = (int64_t)blockIdx.x * per, te = tb + per < total ? tb + per : total;
    wide_d4 acc[4][4];
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int J = 0; J < 4; ++J)
            acc[I][J] = wide_d4{0.0, 0.0, 0.0, 0.0};
    double a[4], b[4], an[4], bn[4];
    auto load = [&](int64_t t, double (&x)[4], double (&y)[4]) {
        const int64_t tt = t + lk;
        const bool ok = tt < te;
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            x[I] = ok ? alpha[tt * 64 + 16 * I + li] : 0.0;
            y[I] = ok ? W[tt * 64 + 16 * I + li] : 0.0;
        }
    };
    load(tb, a, b);
    for (int64_t t = tb; t < te; t += 4) {
        load(t + 4, an, bn); // (beyond te: zeros)
#pragma unroll
        for (int I = 0; I < 4; ++I)
#pragma unroll
            for (int J = 0; J < 4; ++J)
                acc[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[I], b[J], acc[I][J], 0, 0, 0);
#pragma unroll
        for (int I = 0; I < 4; ++I) {
            a[I] = an[I];
            b[I] = bn[I];
        }
    }
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int J = 0; J < 4; ++J)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                xipart[((int64_t)blockIdx.x * 64 + 16 * I + lk + 4 * r) * 64 + 16 * J + li] = acc[I][J][r];
}

[[maybe_unused]] static __global__ void k_wide_xi_reduce(const double *xipart, int nsplit, double *part)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 64 * 64)
        return;
    double v = 0.0;
    for (int s = 0; s < nsplit; ++s)
        v += xipart[(int64_t)s * 4096 + e];
    part[e] = v; // segment 0's C' block (the backward kernel wrote zeros there)
}

// rows W_{T-1} of every trajectory: no transition leaves the last step
[[maybe_unused]] static __global__ void k_wide_zero_last_rows(const int64_t *off, int K, int n, double *Wg)
{
    const int k = blockIdx.x;
    if (k < K && off[k + 1] > off[k])
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            Wg[(off[k + 1] - 1) * n + i] = 0.0;
}

// =========================================================================================
// k_wide_probe: the forgetting curve of THIS model on THIS data for 9..64 states (the role of
// k_forget_probe for N <= 8).  One lane group per (sample position, direction): the forward
// (dir 0) or backward (dir 1) recursion runs over the same stretch of observations from two
// start vectors -- uniform / all mass on one state -- and curve[dir][w] receives the largest
// componentwise relative deviation between the two normalised vectors after w + 1 steps (float
// bits, maximum over the samples).  The host reads the warm-up length of the time segments off
// the curve before the first E-step instead of finding it by a failed boundary check.
// =========================================================================================
template <int NP, int KIND>
__global__ __launch_bounds__(64) void k_wide_probe(const WideModel m, const void *obs_rm,
                                                   const int64_t *starts, int S, int Wmax,
                                                   unsigned int *curve)
{
    constexpr int GP = 64 / NP;
    const int lane = threadIdx.x;
    const int gi = lane / NP, j = lane % NP;
    const int id = blockIdx.x * GP + gi;
    const bool act = id < 2 * S;
    const int dir = act ? id / S : 0, idx = act ? id % S : 0;
    const int n = m.n;
    const bool real = j < n;
    const unsigned long long gmask = wgroup_mask<NP>(lane);
    const int64_t pos0 = starts[idx];
    // dir 0: column j of A (new_j = sum_c v_c A[c][j]); dir 1: row j (new_j = sum_c A[j][c] v_c)
    double Areg[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c)
        Areg[c] = (real && c < n) ? (dir == 0 ? m.A[(int64_t)c * n + j] : m.A[(int64_t)j * n + c]) : 0.0;
    const double mu_j = (KIND == EMIT_GAUSS && real) ? m.mu[j] : 0.0;
    const double ga_j = (KIND == EMIT_GAUSS && real) ? m.ga[j] : 0.0; // gauss_pdf_issue() constants
    const double gb_j = (KIND == EMIT_GAUSS && real) ? m.gb[j] : 1.0;
    double x = real ? 1.0 / (double)n : 0.0, y = (j == idx % n) ? 1.0 : 0.0;
    auto matvec = [&](double v) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        const Rows4 vr = rows_of_group<NP>(v);
        unrolled<NP / 16>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            dot16(acc, vr.r[r], [&](auto ic) -> const double & {
                return Areg[16 * r + decltype(ic)::value];
            });
        });
        return (acc[0] + acc[1]) + (acc[2] + acc[3]);
    };
    for (int w = 0; w < Wmax; ++w) {
        const int64_t t = dir == 0 ? pos0 + w : pos0 + Wmax - 1 - w;
        const WideIn in = wide_load<KIND>(m, j, real, t, obs_rm);
        const double p = wide_emit<NP, KIND, true>(m, j, real, in, mu_j, ga_j, gb_j, gmask);
        auto step = [&](double v) {
            const double r = dir == 0 ? matvec(v) * p : matvec(p * v);
            return r * fast_rcp(wgroup_sum<NP>(r));
        };
        x = step(x);
        y = step(y);
        const double d = fabs(x - y), lo = fmin(x, y);
        double dev = lo > 1e-280 ? d / lo : (d > 1e-280 ? 1.0 : 0.0);
        if (!(dev == dev))
            dev = 1.0;
        dev = fmin(dev, 1.0);
        const int devmax = wgroup_max<NP>(__float_as_int((float)dev)); // non-negative floats order as ints
        if (act && j == 0)
            atomicMax(&curve[dir * Wmax + w], (unsigned int)devmax);
        if (__all(__int_as_float(devmax) < 1e-14f)) // every chain of the wavefront has merged
            break;
    }
}

// boundary consistency of a segmented run (see k_spec_check): one thread per segment
// (sixteen lanes per boundary, 16 boundaries per workgroup of 256: launched with (nseg + 15) / 16 workgroups)
[[maybe_unused]] static __global__ void k_wide_check(const Segs sg, int n, const double *a_entry,
                                    const double *a_exit, const double *b_exit,
                                    const double *b_entry, double tol, unsigned int *result)
{
    const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
    const bool live = s < sg.nseg && sg.len[s] != 0 && sg.t0[s] != 0;
    auto sum16 = [](double v) {
        v += __shfl_xor(v, 8, 16);
        v += __shfl_xor(v, 4, 16);
        v += __shfl_xor(v, 2, 16);
        return v + __shfl_xor(v, 1, 16);
    };
    double dev = 0.0;
    auto cmp = [&](const double *x, const double *y) {
        double sx = 0.0, sy = 0.0;
        if (live)
            for (int j = l; j < n; j += 16) {
                sx += x[j];
                sy += y[j];
            }
        sx = sum16(sx);
        sy = sum16(sy);
        if (!live)
            return;
        if (!(sx > 0.0) || !(sy > 0.0)) {
            dev = 1.0;
            return;
        }
        for (int j = l; j < n; j += 16) {
            const double xs = x[j] / sx, ys = y[j] / sy;
            const double d = fabs(xs - ys);
            const double r = (ys > 1e-280) ? d / ys : (d > 1e-280 ? 1.0 : 0.0);
            dev = fmax(dev, r == r ? r : 1.0);
        }
    };
    if (a_entry) // (the two passes may run on different segment plans: one launch per plan)
        cmp(a_entry + (int64_t)s * n, a_exit + (int64_t)(s - 1) * n);
    if (b_exit)
        cmp(b_exit + (int64_t)(s - 1) * n, b_entry + (int64_t)s * n);
    dev = fmax(dev, __shfl_xor(dev, 8, 16));
    dev = fmax(dev, __shfl_xor(dev, 4, 16));
    dev = fmax(dev, __shfl_xor(dev, 2, 16));
    dev = fmax(dev, __shfl_xor(dev, 1, 16));
    if (!live || l != 0)
        return;
    if (!(dev <= tol))
        atomicAdd(&result[0], 1u);
    atomicMax(&result[1], __float_as_uint((float)fmin(dev, 1.0)));
}

// packed statistics (bhmm_amd.h layout) from the per-trajectory partials; one wavefront per
// output entry, trajectory-strided partial sums + fixed shuffle tree.
template <int KIND>
__global__ __launch_bounds__(64) void k_wide_finalize(const WideModel m, int K, int nseg, int ndtab,
                                                      const double *part, const double *dstat,
                                                      const double *logL_k, const double *gamma0,
                                                      double *stats)
{
    const int n = m.n;
    const int S = n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0);
    const int MN = (KIND == EMIT_DISC) ? n * m.M : 0;
    const int oG0 = 1, oC = 1 + n, oSG = oC + n * n, oE = oSG + n;
    const int lane = threadIdx.x;
    int e = blockIdx.x;
    double s = 0.0;
    if (e < S) {
        for (int k = lane; k < nseg; k += 64)
            s += part[(int64_t)k * S + e];
        s = wave_sum(s);
        if (lane == 0) {
            if (e < n * n)
                stats[oC + e] = s * m.A[e];
            else if (e < n * n + n)
                stats[oSG + (e - n * n)] = s;
            else
                stats[oE + (e - n * n - n)] = s;
        }
        return;
    }
    e -= S;
    if (e < MN) {
        for (int k = lane; k < ndtab; k += 64) // (ndtab symbol tables: one per segment, or four per tile)
            s += dstat[(int64_t)k * MN + e];
        s = wave_sum(s);
        if (lane == 0)
            stats[oE + e] = s;
        return;
    }
    e -= MN;
    if (e < n) {
        for (int k = lane; k < K; k += 64)
            s += gamma0[(int64_t)k * n + e];
        s = wave_sum(s);
        if (lane == 0)
            stats[oG0 + e] = s;
        return;
    }
    for (int k = lane; k < K; k += 64)
        s += logL_k[k];
    s = wave_sum(s);
    if (lane == 0)
        stats[0] = s;
}

// plain scaled backward pass with the reference normalisation, beta row-major (_hidden.c:69-110)
template <int NP>
__global__ __launch_bounds__(64) void k_wide_beta(const WideModel m, const int64_t *off, int K,
                                                  const double *pobs_rm, double *beta_rm)
{
    constexpr int GP = 64 / NP;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sA = smem;
    double *xb = smem + NP * (NP + 1);
    const int lane = threadIdx.x;
    const int gi = lane / NP, i = lane % NP;
    const int n = m.n;
    for (int e = lane; e < NP * NP; e += 64) {
        const int r = e / NP, c = e % NP;
        sA[r * (NP + 1) + c] = (r < n && c < n) ? m.A[(int64_t)r * n + c] : 0.0;
    }
    __syncthreads();
    const int k = blockIdx.x * GP + gi;
    if (k >= K)
        return;
    const bool real = i < n;
    const int64_t o0 = off[k];
    const int64_t T = off[k + 1] - o0;
    if (T <= 0)
        return;
    const double *arow = sA + i * (NP + 1);
    double *xg = xb + gi * NP;
    double b = real ? 1.0 / (double)n : 0.0;
    if (real)
        beta_rm[(o0 + T - 1) * n + i] = b;
    for (int64_t t = T - 1; t >= 1; --t) {
        const double p = rescue_product<NP>(real ? pobs_rm[(o0 + t) * n + i] : 0.0, b,
                                            wgroup_mask<NP>(lane));
        xg[i] = p * b;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NP; c += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(&xg[c]);
            acc[(c / 2) & 3] = fma(arow[c], x.x, acc[(c / 2) & 3]);
            acc[(c / 2) & 3] = fma(arow[c + 1], x.y, acc[(c / 2) & 3]);
        }
        const double br = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        b = br * fast_rcp(wgroup_sum<NP>(br));
        if (real)
            beta_rm[(o0 + t - 1) * n + i] = b;
    }
}

// xi-counts from given alpha, beta, pobs (row-major), _hidden.c:148-183, for 9..64 states.
// One group of NP lanes per time slab; lane i accumulates row i of C' = sum_t alpha_t[i] w_t[j]
// with w = pobs_{t+1} o beta_{t+1} / S_t; the factor A[i][j] is applied when the slabs are summed.
template <int NP>
__global__ __launch_bounds__(64) void k_wide_xi(const double *A, const double *pobs,
                                                const double *alpha, const double *beta, int n,
                                                int64_t T, int nslab, double *part)
{
    constexpr int GP = 64 / NP;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sA = smem;
    double *xb = smem + NP * (NP + 1);
    const int lane = threadIdx.x;
    const int gi = lane / NP, i = lane % NP;
    for (int e = lane; e < NP * NP; e += 64) {
        const int r = e / NP, c = e % NP;
        sA[r * (NP + 1) + c] = (r < n && c < n) ? A[(int64_t)r * n + c] : 0.0;
    }
    __syncthreads();
    const int slab = blockIdx.x * GP + gi;
    if (slab >= nslab)
        return;
    const bool real = i < n;
    const int64_t per = (T - 1 + nslab - 1) / nslab;
    const int64_t t0 = slab * per, t1 = (t0 + per < T - 1) ? t0 + per : T - 1;
    const double *arow = sA + i * (NP + 1);
    double *xg = xb + gi * NP;
    double Crow[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c)
        Crow[c] = 0.0;
    for (int64_t t = t0; t < t1; ++t) {
        xg[i] = real ? rescue_product<NP>(pobs[(t + 1) * n + i], beta[(t + 1) * n + i],
                                           wgroup_mask<NP>(lane)) * beta[(t + 1) * n + i] : 0.0;
        const double a = real ? alpha[t * n + i] : 0.0;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < NP; c += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(&xg[c]);
            acc[(c / 2) & 3] = fma(arow[c], x.x, acc[(c / 2) & 3]);
            acc[(c / 2) & 3] = fma(arow[c + 1], x.y, acc[(c / 2) & 3]);
        }
        const double q = a * ((acc[0] + acc[1]) + (acc[2] + acc[3]));
        const double w = a * fast_rcp(wgroup_sum<NP>(q));
#pragma unroll
        for (int c = 0; c < NP; c += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(&xg[c]);
            Crow[c] = fma(w, x.x, Crow[c]);
            Crow[c + 1] = fma(w, x.y, Crow[c + 1]);
        }
    }
    if (real)
#pragma unroll
        for (int c = 0; c < NP; ++c)
            if (c < n)
                part[((int64_t)slab * n + i) * n + c] = Crow[c];
}

[[maybe_unused]] static __global__ __launch_bounds__(64) void k_wide_xi_sum(const double *A, const double *part,
                                                           int n, int nslab, double *C)
{
    const int e = blockIdx.x;
    double s = 0.0;
    for (int k = threadIdx.x; k < nslab; k += 64)
        s += part[(int64_t)k * n * n + e];
    s = wave_sum(s);
    if (threadIdx.x == 0)
        C[e] = s * A[e];
}

} // namespace bhmm
