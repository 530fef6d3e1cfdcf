// gen_kernels.hpp -- hidden kernels for ANY number of states (the family behind N > 64): the C ABI
// must not refuse what bhmm/hidden/impl_c/_hidden.c:16-378 accepts.  Slow by design -- one workgroup
// per trajectory, serial in t, the transition matrix in LDS up to ~140 states and read from global
// memory (L2) in every step above that -- but order-faithful: compiled with -ffp-contract=off, every product and every sum below is taken in
// the order of the reference's scalar C, so forward / backward rows, Viterbi
// paths and sampled paths (given the uniforms) are those of the reference to the last bit wherever
// the reference's libm exp / log is not involved.  The xi counts, the one O(N^2) statistic, are a true
// dense GEMM here -- C' = alpha^T W over all time steps, W_t = p_{t+1} o beta_{t+1} / S_t -- and run
// on the matrix cores (v_mfma_f64_16x16x4), the case BASELINE.json's north_star reserves them for.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "path_kernels.hpp"

#include "draw_verify.hpp"

namespace bhmm {

constexpr int GEN_TPB = 256;   // threads per workgroup; thread q owns states q, q + 256, ...
constexpr int GEN_PS_SLABS = 16; // time slabs per trajectory of k_gen_path_stats
constexpr int GEN_MAXPT = 16;  // states per thread: up to 4096 states
constexpr int GEN_MAXN = GEN_TPB * GEN_MAXPT;

// sum of x[0 .. n) in ascending index order (the reference's loops), the same value in every thread
__device__ __forceinline__ double gen_ordered_sum(const double *x, int n, double *slot)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        // same additions in the same order; eight entries are loaded at a time and the next eight are
        // requested before the current eight are added (the chain of additions is what bounds this:
        // the LDS round trips hide behind it)
        double s = 0.0;
        const int nb = n / 8;
        double v[8], w[8];
        if (nb > 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = x[u];
        }
        int b = 0;
        for (; b + 2 <= nb; b += 2) { // two register sets in turn: no copies that would wait for a load
#pragma unroll
            for (int u = 0; u < 8; ++u)
                w[u] = x[8 * (b + 1) + u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += v[u];
            if (b + 2 < nb) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = x[8 * (b + 2) + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += w[u];
        }
        if (b < nb) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += v[u];
        }
        int i = 8 * nb;
        for (; i < n; ++i)
            s += x[i];
        *slot = s;
    }
    __syncthreads();
    return *slot;
}

// The transition matrix of a model with n^2 doubles + the vectors below 160 KB of LDS (n <= ~140) is
// copied there once per workgroup (ALDS): every step reads all of it, and the ~600 cycles of an L2
// round trip per eight products -- in front of a chain of dependent additions that the reference's
// order does not allow to split -- were most of a step (3 us per step at n = 65).
constexpr size_t GEN_LDS_LIMIT = 160 * 1024;
// The products of a batch are all formed before its chain of additions starts: left to itself the
// scheduler emits product, addition, product, addition, ... and every addition then waits for its own
// product as well as for its predecessor (twice the latency per term).
#define GEN_CHAIN_FENCE() __builtin_amdgcn_sched_barrier(0)

// sum_i x[i] * Ac[i * stride], in ascending i (the reference's loop; products and additions separate:
// -ffp-contract=off), eight products' operands loaded at a time, the next eight requested first
template <bool PRE>
__device__ __forceinline__ double gen_dot(const double *Ac, int64_t stride, const double *x, int n)
{
    double s = 0.0;
    int i = 0;
    if constexpr (!PRE) {
        // the matrix in global memory (more than ~140 states; several wavefronts per SIMD hide the
        // round trips between them, and the look-ahead below measured 10 % slower there)
        for (; i + 8 <= n; i += 8) {
            double av[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = Ac[(int64_t)(i + u) * stride];
                xv[u] = x[i + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                xv[u] = xv[u] * av[u];
            GEN_CHAIN_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += xv[u];
        }
    } else {
        const int nb = n / 8;
        double av[8], xv[8], aw[8], xw[8];
        if (nb > 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = Ac[(int64_t)u * stride];
                xv[u] = x[u];
            }
        }
        int b = 0;
        for (; b + 2 <= nb; b += 2) { // two register sets in turn (no copies that would wait for a load)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                aw[u] = Ac[(int64_t)(8 * (b + 1) + u) * stride];
                xw[u] = x[8 * (b + 1) + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                xv[u] = xv[u] * av[u];
            GEN_CHAIN_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += xv[u];
            if (b + 2 < nb) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av[u] = Ac[(int64_t)(8 * (b + 2) + u) * stride];
                    xv[u] = x[8 * (b + 2) + u];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                xw[u] = xw[u] * aw[u];
            GEN_CHAIN_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += xw[u];
        }
        if (b < nb) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                xv[u] = xv[u] * av[u];
            GEN_CHAIN_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += xv[u];
        }
        i = 8 * nb;
    }
    for (; i < n; ++i)
        s += x[i] * Ac[(int64_t)i * stride];
    return s;
}
// sum_j Ar[j * stride] * p[j] * b[j], in ascending j, (A p) b as _hidden.c:92-98 multiplies
template <bool PRE>
__device__ __forceinline__ double gen_dot3(const double *Ar, int64_t stride, const double *p,
                                           const double *b, int n)
{
    double s = 0.0;
    int j = 0;
    if constexpr (!PRE) {
        for (; j + 8 <= n; j += 8) {
            double av[8], pv[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = Ar[(int64_t)(j + u) * stride];
                pv[u] = p[j + u];
                bv[u] = b[j + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                av[u] = av[u] * pv[u] * bv[u];
            GEN_CHAIN_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += av[u];
        }
    } else {
        const int nb = n / 8;
        double av[8], pv[8], bv[8], aw[8], pw[8], bw[8];
        if (nb > 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                av[u] = Ar[(int64_t)u * stride];
                pv[u] = p[u];
                bv[u] = b[u];
            }
        }
        int q = 0;
        for (; q + 2 <= nb; q += 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                aw[u] = Ar[(int64_t)(8 * (q + 1) + u) * stride];
                pw[u] = p[8 * (q + 1) + u];
                bw[u] = b[8 * (q + 1) + u];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                av[u] = av[u] * pv[u] * bv[u];
            GEN_CHAIN_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += av[u];
            if (q + 2 < nb) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    av[u] = Ar[(int64_t)(8 * (q + 2) + u) * stride];
                    pv[u] = p[8 * (q + 2) + u];
                    bv[u] = b[8 * (q + 2) + u];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                aw[u] = aw[u] * pw[u] * bw[u];
            GEN_CHAIN_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += aw[u];
        }
        if (q < nb) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                av[u] = av[u] * pv[u] * bv[u];
            GEN_CHAIN_FENCE();
#pragma unroll
            for (int u = 0; u < 8; ++u)
                s += av[u];
        }
        j = 8 * nb;
    }
    for (; j < n; ++j)
        s += Ar[(int64_t)j * stride] * p[j] * b[j];
    return s;
}

// sum over the workgroup (any order: only the tolerance-compared statistics use it): butterfly inside
// each wavefront, the four partial sums through `red` -- one barrier; the caller alternates between two
// areas, and every step has further barriers, so an area is never rewritten while it is still read
__device__ __forceinline__ double gen_block_sum(double v, double *red)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// _hidden.c:16-66.  alpha rows (total, n) row-major, logL per trajectory.
template <bool ALDS>
__global__ __launch_bounds__(GEN_TPB) void k_gen_forward(const WideModel m, const int64_t *off, int K,
                                                          const double *pobs, double *alpha,
                                                          double *logL)
{
    extern __shared__ double gsm[];
    const int n = m.n, tid = threadIdx.x;
    double *xa = gsm, *ya = gsm + n, *slot = gsm + 2 * n;
    const double *Amat;
    if constexpr (ALDS) {
        double *Al = gsm + 2 * n + 2;
        for (int e = tid; e < n * n; e += GEN_TPB)
            Al[e] = m.A[e];
        Amat = Al;
    } else {
        Amat = m.A;
    }
    const int k = blockIdx.x;
    const int64_t t0 = off[k], T = off[k + 1] - t0;
    double ll = 0.0;
    __syncthreads();
    // The emission row of the next step is requested a step ahead (first state of each thread; a
    // barrier pins the load where it is written), and the logarithms are taken by the LAST thread --
    // idle below 256 states -- so that thread 0's ordered sum is the only serial stretch of a step.
    double pnext = (T > 0 && tid < n) ? pobs[t0 * n + tid] : 0.0, cprev = 1.0;
    for (int64_t t = 0; t < T; ++t) {
        const double *p = pobs + (t0 + t) * n;
        const double pmine = pnext;
        if (t + 1 < T && tid < n)
            pnext = p[n + tid];
        for (int j = tid; j < n; j += GEN_TPB) {
            const double pj = j == tid ? pmine : p[j];
            double a;
            if (t == 0)
                a = m.pi[j] * pj;
            else
                a = gen_dot<ALDS>(Amat + j, n, xa, n) * pj;
            ya[j] = a;
        }
        // (the logarithm of the previous step's sum: in the shadow of the other wavefronts' products)
        if (tid == GEN_TPB - 1 && t > 0)
            ll += log(cprev);
        const double c = gen_ordered_sum(ya, n, slot);
        cprev = c;
        double *out = alpha + (t0 + t) * n;
        for (int j = tid; j < n; j += GEN_TPB) {
            double a = ya[j];
            if (c != 0)
                a /= c;
            out[j] = a;
            xa[j] = a;
        }
        __syncthreads();
    }
    if (tid == GEN_TPB - 1) {
        if (T > 0)
            ll += log(cprev);
        logL[k] = ll;
    }
}

// _hidden.c:69-110 and, with STATS, everything else the E-step takes from the backward pass:
//   gamma_t = alpha_t o beta_t / sum (hidden/api.py:176-186)  -> sum_t gamma, gamma_0, emission
//   statistics (gaussian: moments about the old means; discrete: weighted symbol counts), optional
//   gamma rows;  W_t = p_{t+1} o beta_{t+1} / S_t with S_t = sum_i alpha_t[i] (A (p o beta))[i], the
//   xi normaliser of _hidden.c:168-179, for the GEMM C' = alpha^T W.
// At = A transposed (coalesced reads of a row of A by the thread that owns it).
template <int KIND, bool STATS, bool ALDS>
__global__ __launch_bounds__(GEN_TPB) void k_gen_backward(
    const WideModel m, const double *At, const int64_t *off, int K, const double *pobs,
    const void *obs_rm, const double *alpha, double *beta_out, double *W, double *gamma_out,
    double *part /* [K][3][n] */, double *g0 /* [K][n] */, double *symtab /* [n][M] */)
{
    extern __shared__ double gsm[];
    const int n = m.n, tid = threadIdx.x;
    double *nb = gsm, *np = gsm + n, *cur = gsm + 2 * n, *slot = gsm + 3 * n;
    const double *Atm; // A transposed: global, or this workgroup's copy in LDS (behind slot + red)
    if constexpr (ALDS) {
        double *Al = gsm + 3 * n + 1 + GEN_TPB;
        for (int e = tid; e < n * n; e += GEN_TPB)
            Al[e] = At[e];
        Atm = Al;
    } else {
        Atm = At;
    }
    __syncthreads();
    const int k = blockIdx.x;
    const int64_t t0 = off[k], T = off[k + 1] - t0;
    double sc[GEN_MAXPT], sd[GEN_MAXPT], sdd[GEN_MAXPT];
#pragma unroll
    for (int q = 0; q < GEN_MAXPT; ++q)
        sc[q] = sd[q] = sdd[q] = 0.0;
    for (int64_t t = T - 1; t >= 0; --t) {
        double Snorm = 1.0;
        // this step's rows of pobs and alpha, requested before the recursion (first state of each
        // thread; the barriers pin the loads here) instead of where they are used
        const double p_mine = tid < n ? pobs[(t0 + t) * n + tid] : 0.0;
        [[maybe_unused]] double a_mine = 0.0;
        if constexpr (STATS)
            a_mine = tid < n ? alpha[(t0 + t) * n + tid] : 0.0;
        if (t == T - 1) {
            for (int i = tid; i < n; i += GEN_TPB)
                cur[i] = 1.0;
        } else {
            for (int i = tid; i < n; i += GEN_TPB)
                cur[i] = gen_dot3<ALDS>(Atm + i, n, np, nb, n);
        }
        const double c = gen_ordered_sum(cur, n, slot);
        if constexpr (STATS) {
            if (t < T - 1) {
                // S_t = sum_i alpha_t[i] cur[i] (cur still unnormalised), then W_t from the row t+1
                const double *a = alpha + (t0 + t) * n;
                double loc = 0.0;
                // The next row's products p o beta in or near the denormal range (an observation tens
                // of sigma from every state): W = p o beta / S would be a quotient of two numbers with
                // a few digits each -- and S, formed from (A p) beta in the reference's order for the
                // beta row, does not even lose the SAME digits as the numerators (counts off by up to
                // 0.15 in tests/sweeps/stress_many_states.py).  There the products are formed with p
                // times 2^kx (exact) and S from exactly those products; kx cancels in W.
                int kx = 0;
                if (c > 0.0) {
                    int ec;
                    (void)frexp(c, &ec);
                    kx = -ec > 64 ? (-ec < 900 ? -ec : 900) : 0;
                }
                if (kx > 0) {
                    for (int i = tid; i < n; i += GEN_TPB) {
                        double s2 = 0.0;
                        const double *Ar = Atm + i;
                        for (int j = 0; j < n; ++j)
                            s2 += Ar[(int64_t)j * n] * (ldexp(np[j], kx) * nb[j]);
                        loc += a[i] * s2;
                    }
                } else {
                    for (int i = tid; i < n; i += GEN_TPB)
                        loc += (i == tid ? a_mine : a[i]) * cur[i];
                }
                Snorm = gen_block_sum(loc, slot + 1);
                double *wr = W + (t0 + t) * n;
                for (int j = tid; j < n; j += GEN_TPB)
                    wr[j] = ldexp(np[j], kx) * nb[j] / Snorm;
            } else {
                double *wr = W + (t0 + t) * n;
                for (int j = tid; j < n; j += GEN_TPB)
                    wr[j] = 0.0; // the last step of a trajectory has no transition
            }
        }
        // normalise (reference: divide by the row sum unless it is zero), publish as "next" row
        const double *p = pobs + (t0 + t) * n;
        double gl = 0.0;
        for (int i = tid; i < n; i += GEN_TPB) {
            double b = cur[i];
            if (c != 0)
                b /= c;
            nb[i] = b;
            np[i] = i == tid ? p_mine : p[i];
            if (beta_out)
                beta_out[(t0 + t) * n + i] = b;
            if constexpr (STATS)
                gl += (i == tid ? a_mine : alpha[(t0 + t) * n + i]) * b;
        }
        if constexpr (STATS) {
            const double gs = gen_block_sum(gl, slot + 5);
            [[maybe_unused]] double o = 0.0;
            [[maybe_unused]] int sym = 0;
            if constexpr (KIND == EMIT_GAUSS)
                o = static_cast<const double *>(obs_rm)[t0 + t];
            if constexpr (KIND == EMIT_DISC)
                sym = static_cast<const int32_t *>(obs_rm)[t0 + t];
            int q = 0;
            for (int i = tid; i < n; i += GEN_TPB, ++q) {
                const double g = (i == tid ? a_mine : alpha[(t0 + t) * n + i]) * nb[i] / gs;
                sc[q] += g;
                if constexpr (KIND == EMIT_GAUSS) {
                    const double d = o - m.mu[i];
                    sd[q] += g * d;
                    sdd[q] += g * (d * d);
                }
                if constexpr (KIND == EMIT_DISC)
                    atomicAdd(&symtab[(int64_t)i * m.M + sym], g);
                if (gamma_out)
                    gamma_out[(t0 + t) * n + i] = g;
                if (t == 0)
                    g0[(int64_t)k * n + i] = g;
            }
        }
        __syncthreads();
    }
    if constexpr (STATS) {
        int q = 0;
        for (int i = tid; i < n; i += GEN_TPB, ++q) {
            part[((int64_t)k * 3 + 0) * n + i] = sc[q];
            part[((int64_t)k * 3 + 1) * n + i] = sd[q];
            part[((int64_t)k * 3 + 2) * n + i] = sdd[q];
        }
        if (T <= 0)
            for (int i = tid; i < n; i += GEN_TPB)
                g0[(int64_t)k * n + i] = 0.0;
    }
}

// xi as a GEMM on the matrix cores:  part[split] (n x n) = sum over this split's steps of
// alpha_t^T W_t.  One wavefront per 16 x 16 tile of the output, four per workgroup (a 32 x 32 tile),
// v_mfma_f64_16x16x4: lane = 16 k + i holds A-operand element (i, k) and B-operand element (k, i);
// result register r of lane l is element (row (l >> 4) + 4 r, column l & 15).
typedef double gen_v4d __attribute__((ext_vector_type(4)));
// The same product for up to 128 states with every row of alpha and W read once per workgroup: a workgroup
// takes one time slab and the whole n x n result, wavefront I the row tile I (its alpha operand is loaded
// once per four steps and meets the NT operands of W, which the NT wavefronts share through the L1).
// k_gen_xi_gemm re-read every column block of alpha and W (n / 32) times: 2.1 ms where the bytes are 0.5.
template <int NT>
__global__ __launch_bounds__(64 * NT) void k_gen_xi_gemm_rows(const double *alpha, const double *W, int64_t total,
                                                              int n, int nsplit, double *part)
{
    const int I = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t per = ((total + nsplit - 1) / nsplit + 3) / 4 * 4;
    const int64_t tb = (int64_t)blockIdx.x * per, te = tb + per < total ? tb + per : total;
    const bool ia = 16 * I + li < n;
    gen_v4d acc[NT];
#pragma unroll
    for (int J = 0; J < NT; ++J)
        acc[J] = gen_v4d{0.0, 0.0, 0.0, 0.0};
    auto fetch = [&](int64_t t, double &a, double (&b)[NT]) __attribute__((always_inline)) {
        const int64_t tt = t + lk;
        const bool live = tt < te;
        a = (ia && live) ? alpha[tt * n + 16 * I + li] : 0.0;
#pragma unroll
        for (int J = 0; J < NT; ++J)
            b[J] = (live && 16 * J + li < n) ? W[tt * n + 16 * J + li] : 0.0;
    };
    // operands of the next three groups of four steps on their way while one group multiplies (round 5: with one
    // group ahead the kernel waited for memory two thirds of the time -- 620 us at 65 states, 128 x 10 000)
    constexpr int XPF = 4;
    double a[XPF], b[XPF][NT];
#pragma unroll
    for (int u = 0; u < XPF - 1; ++u)
        fetch(tb + 4 * u, a[u], b[u]);
    for (int64_t t = tb; t < te; t += 4 * XPF) {
#pragma unroll
        for (int u = 0; u < XPF; ++u) {
            fetch(t + 4 * (u + XPF - 1), a[(u + XPF - 1) % XPF], b[(u + XPF - 1) % XPF]); // (beyond te: zeros, no loads)
#pragma unroll
            for (int J = 0; J < NT; ++J)
                acc[J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u][J], acc[J], 0, 0, 0);
        }
    }
#pragma unroll
    for (int J = 0; J < NT; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + lk + 4 * r, col = 16 * J + li;
            if (row < n && col < n)
                part[((int64_t)blockIdx.x * n + row) * n + col] = acc[J][r];
        }
}

[[maybe_unused]] static __global__ __launch_bounds__(256) void k_gen_xi_gemm(const double *alpha, const double *W,
                                                     int64_t total, int n, int nsplit, double *part)
{
    const int tiles = (n + 31) / 32;
    const int ti = blockIdx.x / tiles, tj = blockIdx.x % tiles;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = ti * 32 + (wave >> 1) * 16, j0 = tj * 32 + (wave & 1) * 16;
    const int li = lane & 15, lk = lane >> 4;
    const int64_t per = ((total + nsplit - 1) / nsplit + 3) / 4 * 4;
    const int64_t tb = (int64_t)blockIdx.y * per, te = tb + per < total ? tb + per : total;
    const bool ia = i0 + li < n, jb = j0 + li < n;
    gen_v4d acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t t = tb; t < te; t += 4) {
        const int64_t tt = t + lk;
        const double a = (ia && tt < te) ? alpha[tt * n + i0 + li] : 0.0;
        const double b = (jb && tt < te) ? W[tt * n + j0 + li] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + lk + 4 * r, col = j0 + li;
        if (row < n && col < n)
            part[((int64_t)blockIdx.y * n + row) * n + col] = acc[r];
    }
}

// Packed statistics vector of the E-step (include/bhmm_amd.h: bhmm_ctx_stats_size), fixed summation
// order: [logL | gamma_0 sums (n) | C = A o sum_splits C' (n*n) | sum gamma (n) | emission block]
template <int KIND>
__global__ void k_gen_finalize(const WideModel m, int K, int nsplit, const double *xipart,
                               const double *part, const double *g0, const double *logLk,
                               const double *symtab, double *stats)
{
    const int n = m.n;
    const int64_t nn = (int64_t)n * n;
    const int64_t nout = 1 + n + nn + n + (KIND == EMIT_GAUSS ? 2 * n : (KIND == EMIT_DISC ? (int64_t)n * m.M : 0));
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nout;
         e += (int64_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        if (e == 0) {
            for (int k = 0; k < K; ++k)
                v += logLk[k];
        } else if (e < 1 + n) {
            for (int k = 0; k < K; ++k)
                v += g0[(int64_t)k * n + (e - 1)];
        } else if (e < 1 + n + nn) {
            const int64_t ij = e - 1 - n;
            for (int s = 0; s < nsplit; ++s)
                v += xipart[(int64_t)s * nn + ij];
            v *= m.A[ij];
        } else if (e < 1 + 2 * n + nn) {
            const int64_t i = e - 1 - n - nn;
            for (int k = 0; k < K; ++k)
                v += part[((int64_t)k * 3 + 0) * n + i];
        } else if (KIND == EMIT_GAUSS) {
            const int64_t r = e - 1 - 2 * n - nn;
            const int which = (int)(r / n) + 1;
            const int64_t i = r % n;
            for (int k = 0; k < K; ++k)
                v += part[((int64_t)k * 3 + which) * n + i];
        } else {
            v = symtab[e - 1 - 2 * n - nn];
        }
        stats[e] = v;
    }
}

// W rows from GIVEN alpha / beta / pobs (single-trajectory bhmm_transition_counts, _hidden.c:148-183):
// one workgroup per step t < T - 1.
[[maybe_unused]] static __global__ __launch_bounds__(GEN_TPB) void k_gen_w_rows(const double *At, int n, int64_t T,
                                                        const double *pobs, const double *alpha,
                                                        const double *beta, double *W)
{
    extern __shared__ double gsm[];
    double *pb = gsm, *red = gsm + n;
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;
    if (t >= T - 1) {
        for (int j = tid; j < n; j += GEN_TPB)
            W[t * n + j] = 0.0;
        return;
    }
    for (int j = tid; j < n; j += GEN_TPB)
        pb[j] = pobs[(t + 1) * n + j] * beta[(t + 1) * n + j];
    __syncthreads();
    double loc = 0.0;
    for (int i = tid; i < n; i += GEN_TPB) {
        double s = 0.0;
        for (int j = 0; j < n; ++j)
            s += At[(int64_t)j * n + i] * pb[j];
        loc += alpha[t * n + i] * s;
    }
    red[tid] = loc;
    __syncthreads();
    for (int w = GEN_TPB / 2; w > 0; w >>= 1) {
        if (tid < w)
            red[tid] += red[tid + w];
        __syncthreads();
    }
    const double S = red[0];
    for (int j = tid; j < n; j += GEN_TPB)
        W[t * n + j] = pb[j] / S;
}

// _hidden.c:203-281, forward part: back-pointers (one uint16 per (t, j)) and the final state
template <bool ALDS>
__global__ __launch_bounds__(GEN_TPB) void k_gen_viterbi_fwd(const WideModel m, const int64_t *off,
                                                              int K, const double *pobs,
                                                              uint16_t *ptr, int32_t *last_state)
{
    extern __shared__ double gsm[];
    const int n = m.n, tid = threadIdx.x;
    double *v = gsm, *vn = gsm + n, *slot = gsm + 2 * n;
    const int k = blockIdx.x;
    const int64_t t0 = off[k], T = off[k + 1] - t0;
    if (T <= 0)
        return;
    const double *Amat;
    if constexpr (ALDS) {
        double *Al = gsm + 2 * n + 2;
        for (int e = tid; e < n * n; e += GEN_TPB)
            Al[e] = m.A[e];
        Amat = Al;
    } else {
        Amat = m.A;
    }
    __syncthreads();
    for (int64_t t = 0; t < T; ++t) {
        const double *p = pobs + (t0 + t) * n;
        for (int j = tid; j < n; j += GEN_TPB) {
            if (t == 0) {
                vn[j] = p[j] * m.pi[j];
            } else {
                int best = 0;
                const double *Ac = Amat + j;
                double hm = v[0] * Ac[0];
                int i = 1;
                for (; i + 8 <= n; i += 8) {
                    double av[8], xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        av[u] = Ac[(int64_t)(i + u) * n];
                        xv[u] = v[i + u];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double h = xv[u] * av[u];
                        if (h > hm) {
                            hm = h;
                            best = i + u;
                        }
                    }
                }
                for (; i < n; ++i) {
                    const double h = v[i] * Ac[(int64_t)i * n];
                    if (h > hm) {
                        hm = h;
                        best = i;
                    }
                }
                ptr[(t0 + t) * n + j] = (uint16_t)best;
                vn[j] = p[j] * v[best] * Amat[(int64_t)best * n + j];
            }
        }
        const double S = gen_ordered_sum(vn, n, slot);
        for (int j = tid; j < n; j += GEN_TPB)
            v[j] = vn[j] / S;
        __syncthreads();
    }
    if (tid == 0) {
        int a = 0;
        double mx = v[0];
        for (int i = 1; i < n; ++i)
            if (v[i] > mx) {
                mx = v[i];
                a = i;
            }
        last_state[k] = a;
    }
}

// back-trace (_hidden.c:269-272): one thread per trajectory chases the pointers
template <typename PT>
__global__ void k_gen_viterbi_trace(const int64_t *off, int K, int n, const uint16_t *ptr,
                                    const int32_t *last_state, PT *path)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K)
        return;
    const int64_t t0 = off[k], T = off[k + 1] - t0;
    if (T <= 0)
        return;
    int s = last_state[k];
    path[t0 + T - 1] = (PT)s;
    for (int64_t t = T - 2; t >= 0; --t) {
        s = ptr[(t0 + t + 1) * n + s];
        path[t0 + t] = (PT)s;
    }
}

// _hidden.c:330-378 (+ _normalize :307-319, _random_choice :283-305): backward sampling from alpha
template <bool ALDS>
__global__ __launch_bounds__(GEN_TPB) void k_gen_sample(const WideModel m, const int64_t *off, int K,
                                                        const double *alpha, const double *u,
                                                        uint64_t seed, const int64_t *soff,
                                                        int32_t *path, int *status)
{
    extern __shared__ double gsm[];
    const int n = m.n, tid = threadIdx.x;
    double *ps = gsm, *qs = gsm + n, *slot = gsm + 2 * n;
    int *pick = reinterpret_cast<int *>(gsm + 2 * n + 1);
    [[maybe_unused]] double *Alt = gsm + 2 * n + 4; // ALDS: A transposed (column nxt of A contiguous)
    if constexpr (ALDS) {
        for (int e = tid; e < n * n; e += GEN_TPB)
            Alt[(e % n) * n + e / n] = m.A[e];
    }
    __syncthreads();
    const int k = blockIdx.x;
    const int64_t t0 = off[k], T = off[k + 1] - t0;
    const int64_t s0 = soff ? soff[k] : t0;
    int nxt = 0;
    for (int64_t t = T - 1; t >= 0; --t) {
        const double *a = alpha + (t0 + t) * n;
        for (int i = tid; i < n; i += GEN_TPB) {
            double av;
            if constexpr (ALDS)
                av = Alt[nxt * n + i];
            else
                av = m.A[(int64_t)i * n + nxt];
            ps[i] = (t == T - 1) ? a[i] : a[i] * av;
        }
        const double S = gen_ordered_sum(ps, n, slot);
        // the quotients of _normalize (:307-319) by the thread that owns the entry; the cumulative
        // search of _random_choice (:283-305) is the serial part
        for (int i = tid; i < n; i += GEN_TPB)
            qs[i] = ps[i] / S;
        __syncthreads();
        if (tid == 0) {
            const double r = u ? u[t0 + t] : uniform01(seed, (uint64_t)(s0 + t));
            double acc = 0.0;
            int pk = -1;
            int i = 0;
            for (; pk < 0 && i + 8 <= n; i += 8) {
                double v[8];
#pragma unroll
                for (int w = 0; w < 8; ++w)
                    v[w] = qs[i + w];
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    acc += v[w];
                    if (pk < 0 && acc >= r)
                        pk = i + w;
                }
            }
            for (; pk < 0 && i < n; ++i) {
                acc += qs[i];
                if (acc >= r)
                    pk = i;
            }
            *pick = pk;
        }
        __syncthreads();
        nxt = *pick;
        if (nxt < 0) {
            if (tid == 0)
                atomicExch(status, BHMM_ERR_CHOICE);
            return;
        }
        if (tid == 0)
            path[t0 + t] = nxt;
        __syncthreads();
    }
}

// Viterbi over time segments for 65..128 states (round 4): the scheme of k_wide_viterbi_seg
// (path_kernels.hpp: warm-up, bitwise boundary check, fix-up rounds from the predecessor's exact vector,
// checkpoints every 64th step) with TWO target states per lane -- j = lane and lane + 64 -- and the
// columns of A in LDS (sAT[j][i], pitch 130: sixteen lanes read their 16-byte pieces conflict-free).
// One wavefront per segment, eight per workgroup.  `pobs`: the (total, n) emission matrix (gen_pobs).
//   v_entry / v_exit [nseg][128];  ckpt [(total >> 6) + 1][128];  flag [nseg]
constexpr int GVS_PITCH = 130;
template <bool FIX>
__global__ __launch_bounds__(512) void k_gen_viterbi_seg(const WideModel m, const int64_t *off, const Segs sg,
                                                         const double *pobs, uint8_t *ptr, int32_t *last_state,
                                                         double *v_entry, double *v_exit, double *ckpt,
                                                         const uint8_t *flag, double *vall = nullptr,
                                                         double mend_tol = 0.0, unsigned int *notmet = nullptr)
{
    // FIX with mend_tol > 0: the mending round of k_wide_viterbi_seg (path_kernels.hpp) -- flagged segments run again
    // from the predecessor's vector only until they are within mend_tol of a kept vector of the first pass; every
    // vector of the repeated stretch goes to vall; a segment that reaches its end counts in notmet.
    extern __shared__ __attribute__((aligned(16))) double gvs_sm[];
    double *sAT = gvs_sm;                         // [128][GVS_PITCH]
    double *xv = gvs_sm + 128 * GVS_PITCH;        // [8][128]: the final-state search
    const int n = m.n;
    for (int e = threadIdx.x; e < 128 * 128; e += 512) {
        const int j = e >> 7, i = e & 127;
        sAT[j * GVS_PITCH + i] = (i < n && j < n) ? m.A[(int64_t)i * n + j] : 0.0;
    }
    __syncthreads(); // (the only one: from here on the wavefronts are on their own)
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sgi = blockIdx.x * 8 + w;
    if (sgi >= sg.nseg || sg.len[sgi] <= 0)
        return;
    if constexpr (FIX) {
        if (!flag[sgi])
            return;
    }
    const int j[2] = {lane, lane + 64};
    const bool real[2] = {j[0] < n, j[1] < n};
    const int k = sg.traj[sgi];
    const int64_t o0 = off[k], T = off[k + 1] - o0;
    const int64_t t0 = sg.t0[sgi], t1 = t0 + sg.len[sgi];
    const int64_t tw = FIX ? t0 : ((t0 - sg.W > 0) ? t0 - sg.W : 0);
    auto emis = [&](int64_t gt, int e) __attribute__((always_inline)) {
        return real[e] ? pobs[gt * n + j[e]] : 0.0;
    };
    double v[2], p_next[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        // (warm-up start; replaced at t = 0.  FIX: the predecessor's vector, t0 > 0 for a flagged segment)
        v[e] = FIX ? v_entry[(int64_t)sgi * 128 + j[e]] : (real[e] ? 1.0 / (double)n : 0.0);
        p_next[e] = emis(o0 + tw, e);
    }
    bool met = false;
    for (int64_t t = tw; t < t1; ++t) {
        double p[2], vn[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            p[e] = p_next[e];
            if (t + 1 < t1)
                p_next[e] = emis(o0 + t + 1, e); // independent of the recursion
        }
        if (t == 0) {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                vn[e] = real[e] ? p[e] * m.pi[j[e]] : 0.0; // _hidden.c:232
        } else {
            const Rows4 R0 = rows_of(v[0]), R1 = rows_of(v[1]);
            const bool nanfree = __ballot(v[0] != v[0] || v[1] != v[1]) == 0ull;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const double *col = sAT + j[e] * GVS_PITCH;
                double bh = 0.0;
                int bi = 0;
                auto argmax_rows = [&](auto usemax) __attribute__((always_inline)) {
                    constexpr bool USEMAX = decltype(usemax)::value;
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) {
                        if (16 * kk >= n) // (row blocks of padded states only: exact zeros, never a winner)
                            continue;
                        double hh[16], wv[16];
                        int ii[16];
#pragma unroll
                        for (int q = 0; q < 16; q += 2) {
                            const double2 y = *reinterpret_cast<const double2 *>(col + 16 * kk + q);
                            wv[q] = y.x;
                            wv[q + 1] = y.y;
                        }
                        auto wk = [&](auto ic) __attribute__((always_inline)) { return wv[decltype(ic)::value]; };
                        const double src = kk < 4 ? R0.r[kk & 3] : R1.r[kk & 3];
                        asm volatile("s_nop 1"); // (a DPP read needs two wait states after the write of its register)
                        prod8_bcast<0>(hh, src, wk);
                        prod8_bcast<8>(hh, src, wk);
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            ii[i] = 16 * kk + i;
#define BHMM_ARGMAX_LEVEL(W)                                                                 \
    _Pragma("unroll") for (int i = 0; i + W < 16; i += 2 * W)                                \
    {                                                                                        \
        const bool take = hh[i + W] > hh[i];                                                 \
        ii[i] = take ? ii[i + W] : ii[i];                                                    \
        if constexpr (USEMAX)                                                                \
            asm("v_max_f64 %0, %1, %2" : "=v"(hh[i]) : "v"(hh[i]), "v"(hh[i + W]));           \
        else                                                                                 \
            hh[i] = take ? hh[i + W] : hh[i];                                                \
    }
                        BHMM_ARGMAX_LEVEL(1)
                        BHMM_ARGMAX_LEVEL(2)
                        BHMM_ARGMAX_LEVEL(4)
                        BHMM_ARGMAX_LEVEL(8)
#undef BHMM_ARGMAX_LEVEL
                        const bool take = (kk == 0) || (hh[0] > bh); // first maximum: _hidden.c:186-200
                        bh = take ? hh[0] : bh;
                        bi = take ? ii[0] : bi;
                    }
                };
                if (nanfree)
                    argmax_rows(std::true_type{});
                else
                    argmax_rows(std::false_type{});
                if (real[e] && t >= t0)
                    ptr[(o0 + t) * n + j[e]] = (uint8_t)bi;
                const double b0 = __shfl(v[0], bi & 63, 64), b1 = __shfl(v[1], bi & 63, 64);
                const double bv = bi < 64 ? b0 : b1, bA = col[bi];
                vn[e] = p[e] * bv * bA; // _hidden.c:253: (p v[i^]) A[i^][j]
            }
        }
        // the normalising sum in ascending order (_hidden.c:256-259): S = fma(vn[i], 1, S), rounded once
        double S = 0.0;
        {
            const Rows4 Rn0 = rows_of(vn[0]), Rn1 = rows_of(vn[1]);
            const double one = 1.0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                sum16_bcast(S, Rn0.r[kk], one);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                if (64 + 16 * kk < n) // (the rest adds exact zeros)
                    sum16_bcast(S, Rn1.r[kk], one);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
            v[e] = vn[e] / S;
        if constexpr (!FIX) {
            if (t == t0 - 1) {
                v_entry[(int64_t)sgi * 128 + j[0]] = v[0];
                v_entry[(int64_t)sgi * 128 + j[1]] = v[1];
            }
        }
        if (vall && t >= t0) { // (every vector of the first pass / of a mended stretch, [total][n]: k_vit_margin)
            if (real[0])
                vall[(o0 + t) * n + j[0]] = v[0];
            if (real[1])
                vall[(o0 + t) * n + j[1]] = v[1];
        }
        if (((o0 + t) & 63) == 63 && t >= t0) {
            double *cp = ckpt + ((o0 + t) >> 6) * 128;
            if constexpr (FIX) {
                const double c0 = cp[j[0]], c1 = cp[j[1]];
                bool same = __double_as_longlong(c0) == __double_as_longlong(v[0]) &&
                            __double_as_longlong(c1) == __double_as_longlong(v[1]);
                if (mend_tol > 0.0) // (uniform)
                    same = same || (fabs(c0 - v[0]) <= mend_tol * c0 && (c0 == 0.0) == (v[0] == 0.0) &&
                                    fabs(c1 - v[1]) <= mend_tol * c1 && (c1 == 0.0) == (v[1] == 0.0));
                if (__ballot(!same) == 0ull) {
                    met = true;
                    break;
                }
            }
            cp[j[0]] = v[0];
            cp[j[1]] = v[1];
        }
    }
    if (met)
        return;
    if (FIX && notmet && lane == 0)
        atomicAdd(notmet, 1u);
    v_exit[(int64_t)sgi * 128 + j[0]] = v[0];
    v_exit[(int64_t)sgi * 128 + j[1]] = v[1];
    if (t1 == T) { // the trajectory's final state (_hidden.c:262-267: first maximum)
        xv[w * 128 + j[0]] = v[0];
        xv[w * 128 + j[1]] = v[1];
        if (lane == 0) {
            double bm = xv[w * 128];
            int bi = 0;
            for (int i = 1; i < n; ++i)
                if (xv[w * 128 + i] > bm) {
                    bm = xv[w * 128 + i];
                    bi = i;
                }
            last_state[k] = bi;
        }
    }
}

// Viterbi over time segments for 129 .. 256 states (round 5): FOUR segments ("rows") per workgroup of 256 threads,
// thread j = target state j of every row.  A no longer fits LDS (512 KB at 256 states) and the serial kernel above is
// bound by streaming it from L2 once per step and trajectory; here one pass over A serves four rows: candidate i's
// entry A[i][j] is loaded once (coalesced over j) and multiplied with v_r[i] of the four rows (LDS, [i][4]: two
// 16-byte broadcast reads).  The arithmetic is the reference's, operation for operation (_hidden.c:203-281:
// products v[i] A[i][j] in ascending i with a strict comparison -- the first maximum --, (p v[i^]) A[i^][j], the
// normalising sum in ascending order by one thread per row, IEEE division), so a segment that starts from the serial
// run's vector reproduces its vectors and back-pointers bit for bit.  First pass only (warm-up from the uniform
// vector, every vector kept for k_vit_margin): the host accepts it when every boundary is bit-identical or by
// the margins of the decisions on the path, and runs the serial kernel otherwise -- no fix-up rounds here.
//   v_entry / v_exit [nseg][256];  vall [total][n];  ptr one byte per (t, j)
typedef double gen_d2 __attribute__((ext_vector_type(2)));
constexpr int GVR_ROWS = 4;
// MEND (round 6): the mending round -- only the rows of flagged segments run, from the predecessor's vector (v_entry, which
// the check replaced by it) at their first step, and a row stops as soon as its vector is within mend_tol of the one
// the first pass kept at the same step (vall, looked at on every 64th global step); every vector and back-pointer of
// the repeated stretch replaces the first pass's.  Rows that reach their end count in notmet.
template <int R, int S, bool MEND = false>
__global__ __launch_bounds__(256 * S) void k_gen_viterbi_rows(const WideModel m, const int64_t *off, const Segs sg,
                                                              const double *pobs, uint8_t *ptr, int32_t *last_state,
                                                              double *v_entry, double *v_exit, double *vall,
                                                              const uint8_t *flag = nullptr, double mend_tol = 0.0,
                                                              unsigned int *notmet = nullptr)
{
    // S threads per target state (sp = threadIdx.x / 256, uniform per wavefront): candidate range sp of S, ascending.
    // A range's winner is its FIRST maximum; ranges are merged in ascending order with a strict comparison, which is
    // the first maximum over all candidates (_hidden.c:186-200).  The point of S: one wavefront per SIMD waits ten
    // cycles per dependent fp64 instruction; S wavefronts per SIMD fill the pipe without more segments (= warm-ups).
    static_assert(R == 4, "the candidate loop reads the four rows of v as two 16-byte pieces");
    extern __shared__ __attribute__((aligned(16))) double gvr_sm[];
    const int n = m.n, j = threadIdx.x & 255, sp = threadIdx.x >> 8, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double *vT = gvr_sm;             // [n][R]: v of the previous step
    double *vn = gvr_sm + R * n;     // [R][n]: this step's unnormalised vector
    double *sS = gvr_sm + 2 * R * n; // [R]
    double *mh = gvr_sm + 2 * R * n + R;                              // [S - 1][R][256]: winners of the ranges 1 .. S - 1
    int *mb = reinterpret_cast<int *>(mh + (S > 1 ? (S - 1) * R * 256 : 0)); // ... and their indices
    const bool real = j < n;
    __shared__ int sDiff[R];
    int sgi[R], k[R], nst[R];
    int64_t o0[R], T[R], t0[R], t1[R], tw[R];
    bool rowmet[R];
    int nmax = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        sgi[r] = blockIdx.x * R + r;
        const bool has = sgi[r] < sg.nseg && sg.len[sgi[r]] > 0 && (!MEND || flag[sgi[r]] != 0);
        k[r] = has ? sg.traj[sgi[r]] : 0;
        o0[r] = has ? off[k[r]] : 0;
        T[r] = has ? off[k[r] + 1] - o0[r] : 0;
        t0[r] = has ? sg.t0[sgi[r]] : 0;
        t1[r] = has ? t0[r] + sg.len[sgi[r]] : 0;
        tw[r] = has ? (MEND ? t0[r] : ((t0[r] - sg.W > 0) ? t0[r] - sg.W : 0)) : 0;
        nst[r] = has ? (int)(t1[r] - tw[r]) : 0;
        nmax = max(nmax, nst[r]);
        rowmet[r] = false;
    }
    if (MEND && nmax == 0)
        return; // (no flagged row here: uniform over the workgroup)
    if (real && sp == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) // (warm-up start, replaced at t = 0; MEND: the predecessor's vector)
            vT[j * R + r] = (MEND && nst[r] > 0) ? v_entry[(int64_t)sgi[r] * 256 + j] : 1.0 / (double)n;
    }
    __syncthreads();
    const double pi_j = real ? m.pi[j] : 0.0;
    const double *Ac = m.A + (real ? j : 0);
    // my candidates: [lo, hi)
    const int chunk = ((n + S - 1) / S + 7) / 8 * 8;
    const int lo = min(sp * chunk, n), hi = min(lo + chunk, n);
    for (int u = 0; u < nmax; ++u) {
        bool act[R];
        int64_t t[R];
        double p[R], hm[R];
        int best[R];
        bool anyrec = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            act[r] = u < nst[r];
            t[r] = tw[r] + u;
            p[r] = (act[r] && real && sp == 0) ? pobs[(o0[r] + t[r]) * n + j] : 0.0;
            best[r] = 0;
            hm[r] = -1.0; // (an empty range never wins: the products are >= 0)
            anyrec |= act[r] && t[r] > 0;
        }
        if (anyrec && real && lo < hi) { // (uniform per wavefront but for the idle threads)
            {
                const double a0 = Ac[(int64_t)lo * n];
                const gen_d2 x0 = *reinterpret_cast<const gen_d2 *>(vT + lo * R);
                const gen_d2 x1 = *reinterpret_cast<const gen_d2 *>(vT + lo * R + 2);
                hm[0] = x0[0] * a0;
                hm[1] = x0[1] * a0;
                hm[2] = x1[0] * a0;
                hm[3] = x1[1] * a0;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    best[r] = lo;
            }
            // batches of eight candidates, the column entries of the next three batches on their way
            const int i0 = lo + 1;
            // (S wavefronts per SIMD have 512 / S registers each and hide latencies among themselves: shallower rings)
            constexpr int NB = S > 1 ? 4 : 8, RD = S > 1 ? 2 : 4;
            const int nb = (hi - i0) / NB;
            double av[RD][NB];
            // (every load is issued unconditionally, beyond the column with a clamped row index: with the loads under
            // a branch the compiler's wait counts assumed the newest batch and the ring was one batch deep in effect)
            auto issue = [&](int b, double (&dst)[NB]) __attribute__((always_inline)) {
#pragma unroll
                for (int e = 0; e < NB; ++e)
                    dst[e] = Ac[(int64_t)min(i0 + NB * b + e, n - 1) * n];
            };
#pragma unroll
            for (int d = 0; d < RD - 1; ++d)
                issue(d, av[d]);
            // ... and the four rows of v for the next batch of candidates (a read issued where it is used costs its
            // whole LDS latency: 325 cycles per candidate instead of the 90 of its arithmetic)
            gen_d2 xv[2][NB][2];
            auto vread = [&](int b, gen_d2 (&dst)[NB][2]) __attribute__((always_inline)) {
#pragma unroll
                for (int e = 0; e < NB; ++e) {
                    const int ii = min(i0 + NB * b + e, n - 1);
                    dst[e][0] = *reinterpret_cast<const gen_d2 *>(vT + ii * R);
                    dst[e][1] = *reinterpret_cast<const gen_d2 *>(vT + ii * R + 2);
                }
            };
            vread(0, xv[0]);
#ifdef GVR_X_NOLOOP
            for (int b0 = 0; b0 < 0; b0 += RD) {
#else
            for (int b0 = 0; b0 < nb; b0 += RD) {
#endif
#pragma unroll
                for (int d = 0; d < RD; ++d) {
                    const int b = b0 + d;
                    issue(b + RD - 1, av[(d + RD - 1) % RD]);
                    vread(b + 1, xv[(d + 1) & 1]);
                    if (b < nb) { // (uniform)
#pragma unroll
                        for (int e = 0; e < NB; ++e) {
                            const int ii = i0 + NB * b + e;
                            const gen_d2 x0 = xv[d & 1][e][0], x1 = xv[d & 1][e][1];
                            const double a = av[d][e];
                            const double h0 = x0[0] * a, h1 = x0[1] * a, h2 = x1[0] * a, h3 = x1[1] * a; // _hidden.c:249
                            // (the running maximum by v_max_f64 instead of two selects: the same value unless the
                            // maximum so far is NaN -- and v is NaN in all components or in none, the normalising
                            // sum sees to that; then every product is NaN and no comparison holds either way)
                            best[0] = h0 > hm[0] ? ii : best[0];
                            best[1] = h1 > hm[1] ? ii : best[1];
                            best[2] = h2 > hm[2] ? ii : best[2];
                            best[3] = h3 > hm[3] ? ii : best[3];
                            hm[0] = __builtin_fmax(hm[0], h0);
                            hm[1] = __builtin_fmax(hm[1], h1);
                            hm[2] = __builtin_fmax(hm[2], h2);
                            hm[3] = __builtin_fmax(hm[3], h3);
                        }
                    }
                }
            }
            for (int i = i0 + NB * nb; i < hi; ++i) {
                const double a = Ac[(int64_t)i * n];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double h = vT[i * R + r] * a;
                    if (h > hm[r]) {
                        hm[r] = h;
                        best[r] = i;
                    }
                }
            }
        }
        if constexpr (S > 1) { // the ranges' winners to range 0, merged in ascending order
            if (sp > 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    mh[((sp - 1) * R + r) * 256 + j] = hm[r];
                    mb[((sp - 1) * R + r) * 256 + j] = best[r];
                }
            }
            __syncthreads();
            if (sp == 0) {
#pragma unroll
                for (int q = 1; q < S; ++q)
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double h = mh[((q - 1) * R + r) * 256 + j];
                        const int bq = mb[((q - 1) * R + r) * 256 + j];
                        if (h > hm[r]) {
                            hm[r] = h;
                            best[r] = bq;
                        }
                    }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double x = 0.0;
            if (act[r] && real && sp == 0) {
                if (t[r] == 0) {
                    x = p[r] * pi_j; // _hidden.c:232
                } else {
                    if (t[r] >= t0[r])
                        ptr[(o0[r] + t[r]) * n + j] = (uint8_t)best[r];
                    x = p[r] * vT[best[r] * R + r] * Ac[(int64_t)best[r] * n]; // _hidden.c:253: (p v[i^]) A[i^][j]
                }
            }
            if (real && sp == 0)
                vn[r * n + j] = x;
        }
        __syncthreads();
        if (lane == 0 && wid < R) { // the normalising sum in ascending order (_hidden.c:256-259), one thread per row
            const double *x = vn + wid * n;
            double s = 0.0;
            int i = 0;
#ifdef GVR_X_NOSUM
            i = n - 1;
#endif
            for (; i + 8 <= n; i += 8) {
                double w8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    w8[e] = x[i + e];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    s += w8[e];
            }
            for (; i < n; ++i)
                s += x[i];
            sS[wid] = s;
        }
        // MEND: rows at a kept step of the first pass compare with it (uniform: every thread knows every row's step)
        bool cp[R], anycp = false;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            cp[r] = MEND && act[r] && ((o0[r] + t[r]) & 63) == 63 && t[r] >= t0[r];
            anycp |= cp[r];
        }
        if (MEND && anycp && threadIdx.x < R)
            sDiff[threadIdx.x] = 0;
        __syncthreads();
        if (sp == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (act[r]) { // (the boundary vectors are compared over all 256 entries: zeros beyond n)
                    const double v = real ? vn[r * n + j] / sS[r] : 0.0;
                    if (real)
                        vT[j * R + r] = v;
                    if (t[r] == t0[r] - 1)
                        v_entry[(int64_t)sgi[r] * 256 + j] = v;
                    if (cp[r] && real) {
                        const double old = vall[(o0[r] + t[r]) * n + j];
                        const bool same = __double_as_longlong(old) == __double_as_longlong(v) ||
                                          (fabs(old - v) <= mend_tol * old && (old == 0.0) == (v == 0.0));
                        if (!same)
                            sDiff[r] = 1;
                    }
                    if (real && t[r] >= t0[r])
                        vall[(o0[r] + t[r]) * n + j] = v;
                    if (t[r] == t1[r] - 1)
                        v_exit[(int64_t)sgi[r] * 256 + j] = v;
                }
            }
        }
        __syncthreads();
        if (MEND && anycp) {
            nmax = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (cp[r] && sDiff[r] == 0) { // within the tolerance of the first pass's vector: its vectors stand from here
                    nst[r] = u + 1;
                    rowmet[r] = true;
                }
                nmax = max(nmax, nst[r]);
            }
        }
    }
    if (MEND) {
        if (notmet && threadIdx.x == 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (nst[r] > 0 && !rowmet[r])
                    atomicAdd(notmet, 1u);
        }
    }
    // the trajectory's final state (_hidden.c:262-267: first maximum) by the row that holds its last step
    if (lane == 0 && wid < R && nst[wid] > 0 && t1[wid] == T[wid] && !rowmet[wid]) {
        double bm = vT[wid];
        int bi = 0;
        for (int i = 1; i < n; ++i)
            if (vT[i * R + wid] > bm) {
                bm = vT[i * R + wid];
                bi = i;
            }
        last_state[k[wid]] = bi;
    }
}

// Backward draw over time segments for 65..512 states (round 4): the scheme of k_wide_sample_seg
// (path_kernels.hpp) with SPL states per lane -- state e * 64 + lane in slot e --, one wavefront per
// segment, column s_{t+1} of A read from the transposed copy `At` (coalesced).  The draws of different
// runs are coupled through the per-step uniforms; pass 0 starts W steps above a segment from state 0,
// k_wide_smp_check flags the segments that did not continue their successor's state, and the fix-up
// rounds (FIX) draw those again until the path they meet is the one already there.  The decision uses
// prefix sums from a DPP scan wherever a margin of 1e-12 S makes the reference's ordered chains
// (_hidden.c:283-319) redundant, and those chains themselves otherwise.
template <int SPL, bool FIX>
__global__ __launch_bounds__(256) void k_gen_sample_seg(const WideModel m, const double *At, const int64_t *off,
                                                        const Segs sg, const double *alpha, const double *u,
                                                        uint64_t seed, const int64_t *soff, int32_t *path,
                                                        int *status, int32_t *s_entry, int32_t *s_exit,
                                                        const uint8_t *flag, const DrawWatch watch)
{
    __shared__ double xs[4][64 * SPL];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int sgi = blockIdx.x * 4 + w;
    if (sgi >= sg.nseg || sg.len[sgi] <= 0)
        return;
    if constexpr (FIX) {
        if (!flag[sgi])
            return;
    }
    const int n = m.n;
    bool real[SPL];
#pragma unroll
    for (int e = 0; e < SPL; ++e)
        real[e] = e * 64 + lane < n;
    const int k = sg.traj[sgi];
    const int64_t o0 = off[k], T = off[k + 1] - o0;
    const int64_t s0 = soff ? soff[k] : o0;
    const int64_t t0 = sg.t0[sgi], t1 = t0 + sg.len[sgi];
    const int64_t ts = FIX ? t1 - 1 : ((t1 + sg.W < T ? t1 + sg.W : T) - 1);
    int nxt = FIX ? s_entry[sgi] : 0;
    double a_next[SPL];
#pragma unroll
    for (int e = 0; e < SPL; ++e)
        a_next[e] = real[e] ? alpha[(o0 + ts) * n + e * 64 + lane] : 0.0;
    bool met = false;
    for (int64_t t = ts; t >= t0; --t) {
        double ps[SPL], P[SPL];
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
            const double a = a_next[e];
            ps[e] = a;
            if (t != T - 1)
                ps[e] = real[e] ? a * At[(int64_t)nxt * n + e * 64 + lane] : 0.0; // _hidden.c:365
            if (t > t0)
                a_next[e] = real[e] ? alpha[(o0 + t - 1) * n + e * 64 + lane] : 0.0; // independent of the draw
        }
        const double r = u ? u[o0 + t] : uniform01(seed, (uint64_t)(s0 + t));
        double base = 0.0;
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
            P[e] = group_prefix_sum<64>(ps[e]) + base;
            base = __shfl(P[e], 63, 64);
        }
        const double Sf = base, thr = r * Sf;
        const bool ok = Sf > 1e-290 && Sf < 1e290; // (also false for NaN)
        bool near = !ok, wnear = false;
#pragma unroll
        for (int e = 0; e < SPL; ++e) {
            near |= real[e] && !(fabs(P[e] - thr) > 1e-12 * Sf);
            wnear |= real[e] && !(fabs(P[e] - thr) > watch.tol * Sf);
        }
        // within reach of the deviation the alpha rows were verified to (watch.tol = 64 x that deviation; 0 for
        // rows of the serial recursion): recorded below, decided again afterwards (draw_verify.hpp)
        const bool watched = watch.tol > 0.0 && __ballot(wnear) != 0ull;
        int pick = -1;
        if (__ballot(near) == 0ull) {
#pragma unroll
            for (int e = 0; e < SPL; ++e) {
                const unsigned long long ge = __ballot(real[e] && P[e] >= thr);
                if (pick < 0 && ge)
                    pick = e * 64 + (int)__builtin_ctzll(ge);
            }
        } else {
            // the reference's arithmetic: _normalize (ascending sum, quotients), then the first state
            // whose cumulative sum reaches r -- every lane runs the same chains on the LDS copy
#pragma unroll
            for (int e = 0; e < SPL; ++e)
                xs[w][e * 64 + lane] = ps[e];
            double S = 0.0;
            for (int i = 0; i < n; ++i)
                S += xs[w][i];
#pragma unroll
            for (int e = 0; e < SPL; ++e)
                xs[w][e * 64 + lane] = ps[e] / S;
            double acc = 0.0;
            for (int i = 0; i < n; ++i) {
                acc += xs[w][i];
                if (pick < 0 && acc >= r)
                    pick = i;
            }
        }
        if (pick < 0) {
            if (lane == 0 && t < t1)
                status[0] = BHMM_ERR_CHOICE;
            pick = n - 1;
        } else if (__builtin_expect(watched && lane == 0 && t < t1, 0)) {
            draw_record(watch, k, t, nxt, r, pick, -1.0);
        }
        nxt = pick;
        if constexpr (!FIX) {
            if (t == t1 && lane == 0)
                s_entry[sgi] = pick;
        }
        if (t < t1) {
            if constexpr (FIX) {
                if (path[o0 + t] == pick) {
                    met = true;
                    break;
                }
            }
            if (lane == 0)
                path[o0 + t] = pick;
        }
    }
    if (!met && lane == 0)
        s_exit[sgi] = nxt;
}

// hidden-path statistics (generic_hmm.py:297-334,398-431): transition / start counts by integer
// atomics (exact, order-free); per-state emission statistics without atomics -- thread q walks the
// trajectory once per state it owns (O(n T) per trajectory, small against the O(n^2 T) forward pass)
// and leaves per-trajectory partials that are added in trajectory order.
template <int KIND>
__global__ __launch_bounds__(GEN_TPB) void k_gen_path_stats(const WideModel m, const int64_t *off,
                                                            int K, const void *obs_rm,
                                                            const int32_t *path,
                                                            unsigned long long *cnt /* n*n + n */,
                                                            double *epart /* gauss [K][GEN_PS_SLABS][3][n] */,
                                                            unsigned long long *symcnt /* [n][M] */)
{
    // grid (K, GEN_PS_SLABS): a block takes one slab of the trajectory's steps; the emission partials of
    // the slabs are added in (trajectory, slab) order by k_gen_pack_path_stats (run-to-run identical)
    const int n = m.n, tid = threadIdx.x;
    const int k = blockIdx.x, slab = blockIdx.y;
    const int64_t t0 = off[k], T = off[k + 1] - t0;
    const int64_t ta = T * slab / GEN_PS_SLABS, tb = T * (slab + 1) / GEN_PS_SLABS;
    if (T > 0 && tid == 0 && slab == 0)
        atomicAdd(&cnt[(int64_t)n * n + path[t0]], 1ull);
    for (int64_t t = ta + tid; t < tb && t + 1 < T; t += GEN_TPB)
        atomicAdd(&cnt[(int64_t)path[t0 + t] * n + path[t0 + t + 1]], 1ull);
    if constexpr (KIND == EMIT_DISC) {
        const int32_t *sym = static_cast<const int32_t *>(obs_rm);
        for (int64_t t = ta + tid; t < tb; t += GEN_TPB)
            atomicAdd(&symcnt[(int64_t)path[t0 + t] * m.M + sym[t0 + t]], 1ull);
    }
    if constexpr (KIND == EMIT_GAUSS) {
        const double *o = static_cast<const double *>(obs_rm);
        for (int i = tid; i < n; i += GEN_TPB) {
            double c = 0.0, s = 0.0, ss = 0.0;
            const double mu = m.mu[i];
            for (int64_t t = ta; t < tb; ++t)
                if (path[t0 + t] == i) {
                    const double d = o[t0 + t] - mu;
                    c += 1.0;
                    s += d;
                    ss += d * d;
                }
            const int64_t row = (int64_t)k * GEN_PS_SLABS + slab;
            epart[(row * 3 + 0) * n + i] = c;
            epart[(row * 3 + 1) * n + i] = s;
            epart[(row * 3 + 2) * n + i] = ss;
        }
    }
}

template <int KIND>
__global__ void k_gen_pack_path_stats(const WideModel m, int K, const unsigned long long *cnt,
                                      const double *epart, const unsigned long long *symcnt,
                                      double *out)
{
    const int n = m.n;
    const int64_t nn = (int64_t)n * n + n;
    const int64_t esz = KIND == EMIT_GAUSS ? 3 * (int64_t)n : (KIND == EMIT_DISC ? (int64_t)n * m.M : 0);
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nn + esz;
         e += (int64_t)gridDim.x * blockDim.x) {
        double v;
        if (e < nn) {
            v = (double)cnt[e];
        } else if (KIND == EMIT_GAUSS) {
            const int64_t r = e - nn;
            v = 0.0;
            for (int64_t k = 0; k < (int64_t)K * GEN_PS_SLABS; ++k)
                v += epart[(k * 3 + r / n) * n + r % n];
        } else {
            v = (double)symcnt[e - nn];
        }
        out[e] = v;
    }
}

[[maybe_unused]] static __global__ void k_gen_transpose(const double *A, int n, double *At)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (int64_t)n * n)
        At[(e % n) * n + e / n] = A[e];
}

} // namespace bhmm
