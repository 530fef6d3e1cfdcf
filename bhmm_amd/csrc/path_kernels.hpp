// path_kernels.hpp -- order-faithful (bit-exact) kernels: Viterbi decoding and backward
// path sampling.  Compiled with -ffp-contract=off: every product and sum below rounds
// exactly like the reference's scalar C, so paths are
// bit-identical to bhmm/hidden/impl_c/_hidden.c:203-281 and :330-378 given the same pobs /
// alpha / uniforms.
//
// Viterbi is strictly serial in t (a chunked max-product would round differently), so its
// parallelism is: NP lanes per trajectory (lane j owns state j), 64/NP trajectories per
// wavefront, trajectories across the grid; vectors are exchanged through LDS and normalising
// sums are taken in ascending state order in every lane (k_wide_viterbi_*, all state counts).
// Path sampling is exact AND parallel in time: for fixed uniforms the draws are maps of the
// next state and maps compose exactly (k_smp_*); k_sample_path is the serial form behind the
// single-trajectory entry point.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "estep_sweep.hpp"
#include "wide_kernels.hpp"

#include "draw_verify.hpp"

namespace bhmm {

// splitmix64-based uniform in [0,1) for global step index x (device-generated stream)
__device__ __forceinline__ double uniform01(uint64_t seed, uint64_t x)
{
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (x + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

// Backward path sampling (_hidden.c:330-378): N lanes per trajectory, alpha row-major
// (T, nreal).  u == nullptr -> uniforms from uniform01(seed, global step).
// status[0] is set to BHMM_ERR_CHOICE if a draw finds no state (_hidden.c:299-304).
template <int N>
__global__ __launch_bounds__(64) void k_sample_path(const Model<N> m, const int64_t *off, int K,
                                                    const double *alpha_rm, const double *u,
                                                    uint64_t seed, int32_t *path, int *status)
{
    constexpr int GP = 64 / N;
    const int k = blockIdx.x * GP + threadIdx.x / N;
    const int i = threadIdx.x % N;
    if (k >= K)
        return;
    const int64_t o0 = off[k];
    const int64_t T = off[k + 1] - o0;
    if (T <= 0)
        return;
    const int n = m.nreal;
    double Arow[N];
#pragma unroll
    for (int jj = 0; jj < N; ++jj)
        Arow[jj] = m.A[i * N + jj];
    int nxt = 0;
    for (int64_t t = T - 1; t >= 0; --t) {
        const double a = (i < n) ? alpha_rm[(o0 + t) * n + i] : 0.0;
        double ps = a;
        if (t != T - 1) {
            double aij = Arow[0];
#pragma unroll
            for (int jj = 1; jj < N; ++jj)
                aij = (nxt == jj) ? Arow[jj] : aij;
            ps = a * aij; // _hidden.c:365
        }
        double S = 0.0;
#pragma unroll
        for (int q = 0; q < N; ++q)
            S += __shfl(ps, q, N); // _normalize, ascending (:310-314)
        const double pn = ps / S;
        const double r = u ? u[o0 + t] : uniform01(seed, (uint64_t)(o0 + t));
        double acc = 0.0;
        int pick = -1;
#pragma unroll
        for (int q = 0; q < N; ++q) {
            acc += __shfl(pn, q, N);
            if (pick < 0 && q < n && acc >= r) // first state with cumsum >= r (:290-296)
                pick = q;
        }
        if (pick < 0) {
            if (i == 0)
                status[0] = BHMM_ERR_CHOICE;
            pick = n - 1;
        }
        nxt = pick;
        if (i == 0)
            path[o0 + t] = pick;
    }
}

// =========================================================================================
// Chunk-parallel backward path sampling (batched Gibbs step), exact given the uniforms.
//
// For fixed uniforms the reference's draw at step t is a map of the next state,
//   g_t(j) = first i with  cumsum_i( alpha_t[.] A[., j] / sum ) >= u_t      (_hidden.c:359-372)
// (a constant map at t = T-1, :347-355).  Maps compose exactly, so the chunks of the E-step
// decomposition are processed independently:
//   k_smp_maps   : per chunk, F_c = g_{t0} o ... o g_{t1-1} as 8 nibbles (state at the first
//                  step of the NEXT chunk -> state at t0).  All 8 images are tracked until they
//                  coalesce (within tens of steps), then a single image -- which from there on IS
//                  the sampled path, whatever follows the chunk.  The kernel leaves both behind:
//                  the full map g_t of every step above the coalescence point, the state itself
//                  (one nibble) of every step below it.
//   k_smp_stitch : per trajectory, backwards over chunks: the state each chunk starts from.
//   k_smp_apply  : per chunk, walk those tables from the known start state: the path (optional
//                  output) and its statistics; alpha and the uniforms are not read again.
// One draw is decided on the unnormalised partial sums (c_i >= u*S) when the margin exceeds
// 1e-13*S -- 50x the worst-case rounding difference to the reference's normalise-then-cumsum
// arithmetic -- and falls back to exactly that arithmetic (IEEE division, ascending sums)
// otherwise, so the sampled path is identical to the reference's for the same uniforms.
// =========================================================================================
// SPEC (the map kernels, which also draw for next states the path may never take): a draw whose
// weights are all zero -- an impossible next state under a sparse transition matrix -- returns -1
// instead of raising; the caller marks that map entry, and only a walk that really uses it fails.
// gap_out (optional): min_i |c_i - r S| / S, the distance of the uniform from the nearest cumulative sum
template <int N, bool SPEC = false>
__device__ __forceinline__ int pick_state(const double (&a)[N], const double *col, double r, int n,
                                          int *status, double *gap_out = nullptr)
{
    double ps[N], c[N], S = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        ps[i] = col ? a[i] * col[i] : a[i]; // _hidden.c:349 / :365
        S += ps[i];                         // ascending: S is bit-identical to _normalize's sum
        c[i] = S;
    }
    const double t = r * S, tol = 1e-13 * S;
    int pick = -1;
    bool amb = !(S > 0.0);
    double gmin = 1e300;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double d = c[i] - t;
        const bool in = i < n;
        amb |= in && (fabs(d) <= tol);
        gmin = in ? fmin(gmin, fabs(d)) : gmin;
        if (pick < 0 && in && d >= 0.0)
            pick = i;
    }
    if (gap_out)
        *gap_out = S > 0.0 ? gmin / S : 0.0;
    if (amb) { // the reference's own arithmetic (_normalize + _random_choice)
        double acc = 0.0;
        pick = -1;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            acc += ps[i] / S;
            if (pick < 0 && i < n && acc >= r)
                pick = i;
        }
    }
    if (pick < 0) {
        if constexpr (SPEC) {
            if (!(S > 0.0))
                return -1;
        }
        *status = BHMM_ERR_CHOICE;
        pick = n - 1;
    }
    return pick;
}

template <int N>
__device__ __forceinline__ void stage_At(double *sAt, const Model<N> &m)
{
    for (int e = threadIdx.x; e < N * N; e += blockDim.x)
        sAt[(e % N) * N + e / N] = m.A[e]; // sAt[j][i] = A[i][j]
    __syncthreads();
}

// Part p of a chunk of `len` steps covers [s_lo, s_hi) (k_smp_maps and k_smp_apply agree on this)
__device__ __forceinline__ void smp_part_bounds(int len, int part, int P, int &s_lo, int &s_hi)
{
    s_lo = (int)((int64_t)len * part / P);
    s_hi = (int)((int64_t)len * (part + 1) / P);
}

// What a lane fetches for one step of a rebuild: observation, symbol or pobs row.
template <int N, int KIND>
struct SmpRaw {
    double o;    // gaussian: the observation
    int sym;     // discrete: the symbol
    double pv[KIND == EMIT_EXPL ? N : 1]; // explicit: the pobs row
};
template <int N, int KIND>
__device__ __forceinline__ void smp_fetch(const void *obs_ci, int64_t rec, int lane,
                                          SmpRaw<N, KIND> &raw)
{
    raw.o = 0.0;
    raw.sym = 0;
    if constexpr (KIND == EMIT_GAUSS)
        raw.o = static_cast<const double *>(obs_ci)[rec * 64 + lane];
    else if constexpr (KIND == EMIT_DISC)
        raw.sym = static_cast<const int32_t *>(obs_ci)[rec * 64 + lane];
    else
        ci_load<N>(static_cast<const double *>(obs_ci), rec, lane, raw.pv);
}

// next = 2^e ((prev^T A) o p): one forward step (_hidden.c:41-55) of ONE lane holding all N
// states, rescaled by a power of two like scaled_emit<CAREFUL> (the largest entry into [0.5, 1)),
// with the outlier rule of the gaussian model (outputmodel.py:126-130) applied when the product
// vanishes.  The emission probabilities are those of emit_raw (estep_sweep.hpp).
// sAt[j*N + i] = A[i][j]; sEm = [mu | -1/(2 sigma^2) | 1/(sqrt(2 pi) sigma)] (both LDS).
template <int N, int KIND>
__device__ __forceinline__ void smp_rebuild(const double (&prev)[N], const double *sAt,
                                            const double *sEm, const double *Bt_g,
                                            const SmpRaw<N, KIND> &raw, int n, double (&next)[N])
{
    // the model tables are re-read from LDS at every step: hoisted out of the step loop they would
    // occupy 2 N (N + 3) registers of every lane
    asm volatile("" ::: "memory");
    double sv[N];
    int hm = 0;
    bool nz = false;
    [[maybe_unused]] double pb[KIND == EMIT_DISC ? N : 1];
    if constexpr (KIND == EMIT_DISC) { // the symbol's row of B^T ([M][N], L2-resident)
        const double2 *brow = reinterpret_cast<const double2 *>(Bt_g + (int64_t)raw.sym * N);
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const double2 x = brow[i];
            pb[2 * i] = x.x;
            pb[2 * i + 1] = x.y;
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double acc = prev[0] * sAt[j * N];
#pragma unroll
        for (int i = 1; i < N; ++i)
            acc = fma(prev[i], sAt[j * N + i], acc);
        sv[j] = acc;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double pj;
        if constexpr (KIND == EMIT_GAUSS) {
            const double d = raw.o - sEm[j];
            pj = sEm[2 * N + j] * exp_nonpos(d * d * sEm[N + j]);
        } else if constexpr (KIND == EMIT_DISC) {
            pj = pb[j];
        } else {
            pj = raw.pv[j];
        }
        nz |= pj != 0.0;
        next[j] = sv[j] * pj;
        hm = max(hm, __double2hiint(next[j]));
    }
    if constexpr (KIND == EMIT_GAUSS) {
        if (__builtin_expect(hm < 0x00100000 && !nz, 0)) { // all-zero emission row
            const double one = (raw.o != raw.o) ? raw.o : 1.0; // a NaN observation stays NaN
            hm = 0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                next[j] = j < n ? sv[j] * one : 0.0;
                hm = max(hm, __double2hiint(next[j]));
            }
        }
    }
    const int ne = 1022 - (hm >> 20);
#pragma unroll
    for (int j = 0; j < N; ++j)
        next[j] = ldexp(next[j], ne);
}

// A draw decided from an alpha row that is only known to a relative accuracy (the fp32 copy):
// the state if the decision is clear by more than `tol` (relative to the normaliser), -1 if not.
// A normaliser below 1e-30 is never clear: the forward pass rescales its rows only every few
// steps, so after a run of very unlikely observations an fp32 row may sit in the denormal range
// (or be zero), where its relative accuracy is gone.
template <int N>
__device__ __forceinline__ int pick_clear(const double (&a)[N], const double *col, double r, int n,
                                          double tol)
{
    double c[N], S = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        S += col ? a[i] * col[i] : a[i];
        c[i] = S;
    }
    const double t = r * S, w = tol * S;
    int pick = 0;
    bool amb = !(S > 1e-30);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double d = c[i] - t;
        const bool in = i < n;
        amb |= in && (fabs(d) <= w);
        pick += (in && d < 0.0) ? 1 : 0; // c is non-decreasing: first i with c_i >= t
    }
    return (amb || pick >= n) ? -1 : pick;
}

// The maps are built on a partition finer than the E-step chunks (every chunk is cut into P
// parts, P workgroups walk the same 256 chunks): these kernels keep one lane per (chunk, part),
// so more parts = more wavefronts in flight; composition does not care where the cuts are.
// Once the 8 images have coalesced, every further step of the part is the same whatever follows
// the part, i.e. it IS the sampled path there.  Those states are kept, one nibble per step (8 steps
// per word, word w of part p of chunk g at nib[(p * W8 + w) * Gp + g], filled from the top of the
// part downwards), and dmark[part] = the lowest step that still depends on the part's successor:
// k_smp_apply takes the steps [dmark, end of part) from the full step maps and the rest from the
// nibbles.
//
// alpha.  The kernel is bound by reading the alpha rows (and the forward pass by writing them),
// so the forward pass leaves every row rounded to fp32 (rows32, 4 N bytes per step) and only the
// rows of the steps s % FWD_CKPT == 0 in fp64.  A draw compares partial sums of alpha_t[i] A[i][j]
// with u S: with an fp32 row these are known to 1.5e-7 S, so whenever every partial sum is further
// than SMP_TOL32 S from the threshold the fp32 row decides exactly as the fp64 row would.
// Otherwise (a few hundred of 2.6e7 steps on configs[4]) the lane rebuilds the exact fp64 row
// from the stored one below it with the forward recursion itself and decides with pick_state,
// i.e. as before.  rows32 == nullptr: every fp64 row is in alpha_ci (the exact fallback pass of a
// failed speculation wrote them) and is used directly.
constexpr double SMP_TOL32 = 1e-6;
// R32: the forward pass left fp32 rows (rows32 != nullptr) -- a template parameter, not a run-time test: with both
// kinds of rows in one loop the ring below is a bundle of phi registers that every load is copied into at once
template <int N, int KIND, bool R32>
__global__ __launch_bounds__(256, 4) void k_smp_maps(const Model<N> m, const Chunks ch,
                                                     const int64_t *off, const int64_t *soff,
                                                     const double *alpha_ci, const float *rows32,
                                                     const void *obs_ci, const double *Bt_g,
                                                     const double *u, uint64_t seed, int P,
                                                     uint32_t *Fmap, int *status, int32_t *dmark,
                                                     uint32_t *nib, int W8, int64_t Gp, uint32_t *gw,
                                                     int Lp, const DrawWatch watch)
{
    __shared__ __attribute__((aligned(16))) double sAt[N * N];
    __shared__ __attribute__((aligned(16))) double sEm[3 * N];
    if (threadIdx.x < N) {
        const int i = threadIdx.x;
        sEm[i] = m.e0[i];
        sEm[N + i] = -0.5 * m.e1[i] * m.e1[i];
        sEm[2 * N + i] = m.e2[i];
    }
    stage_At<N>(sAt, m);
    const int part = blockIdx.x % P;
    const int64_t g = (int64_t)(blockIdx.x / P) * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int len = ch.len[g];
    if (len == 0)
        return;
    const int k = ch.traj[g];
    const int64_t t0 = ch.t0[g], base = ch.goff[g];
    const int64_t Tk = off[k + 1] - off[k];
    const int64_t sbase = soff[k] + t0; // position of this chunk in the random stream
    const int n = m.nreal;
    int s_lo, s_hi;
    smp_part_bounds(len, part, P, s_lo, s_hi);
    uint32_t cur = 0x76543210u; // nibble j = image of next-part state j (identity)
    int dm = s_hi;
    uint32_t word = 0;
    uint32_t *mynib = nib + ((int64_t)part * W8) * Gp + g;
    // steps above the coalescence point: the whole map of the step (8 nibbles, next state ->
    // state), word j of part p of chunk g at gw[(p * Lp + j) * Gp + g] -- k_smp_apply then needs
    // neither alpha nor the uniforms again
    uint32_t *mygw = gw + ((int64_t)part * Lp) * Gp + g;
    // exact fp64 row of step s: the stored row at or below it, taken forward over the steps between
    auto exact_row = [&](int s, double (&a)[N]) {
        const int cb = R32 ? (s & ~(FWD_CKPT - 1)) : s;
        ci_load<N>(alpha_ci, ci_rec(g, cb, ch.Lmax), lane, a);
#pragma unroll 1
        for (int q = cb + 1; q <= s; ++q) {
            SmpRaw<N, KIND> raw;
            smp_fetch<N, KIND>(obs_ci, ci_rec(g, q, ch.Lmax), lane, raw);
            double nx[N];
            smp_rebuild<N, KIND>(a, sAt, sEm, Bt_g, raw, n, nx);
#pragma unroll
            for (int i = 0; i < N; ++i)
                a[i] = nx[i];
        }
    };
    // The walk is a dependent chain (the column of A a step needs is known only after the previous
    // draw) and its alpha rows come straight from HBM: they are requested SMP_MPF steps ahead.
    uint32_t alive_set = (1u << n) - 1u; // the part may be entered with any real state
#ifndef SMP_MPF_VALUE
#define SMP_MPF_VALUE 1 // steps of load-ahead of the fp32 alpha rows (measured 1 / 2 / 4: DESIGN.md section 5)
#endif
    constexpr int SMP_MPF = SMP_MPF_VALUE;
    constexpr int NF = N / 2; // float2 per fp32 row
    const int nst = s_hi - s_lo;
    float2 ring[SMP_MPF][NF];
    auto fetch32 = [&](int q, int jj) {
        const int sc = s_hi - 1 - (jj < nst ? jj : nst - 1);
        const float2 *p32 =
            reinterpret_cast<const float2 *>(rows32 + ci_rec(g, sc, ch.Lmax) * (int64_t)(N * 64)) +
            lane * NF;
        if constexpr (N >= 4) {
#pragma unroll
            for (int e = 0; e < N / 4; ++e) {
                const float4 x = reinterpret_cast<const float4 *>(p32)[e];
                ring[q][2 * e] = make_float2(x.x, x.y);
                ring[q][2 * e + 1] = make_float2(x.z, x.w);
            }
        } else {
            ring[q][0] = p32[0];
        }
    };
    if (R32 && nst > 0) {
#pragma unroll
        for (int q = 0; q < SMP_MPF; ++q)
            fetch32(q, q);
    }
    for (int jb = 0; jb < nst; jb += SMP_MPF) {
#pragma unroll
    for (int qq = 0; qq < SMP_MPF; ++qq) {
        const int j = jb + qq; // position from the top of the part
        if (j >= nst)
            break;
        const int s = s_hi - 1 - j;
        double a[N];
        if constexpr (R32) {
#pragma unroll
            for (int e = 0; e < NF; ++e) {
                a[2 * e] = ring[qq][e].x;
                a[2 * e + 1] = ring[qq][e].y;
            }
            // (the slot's last use BEFORE it is requested again: scheduled the other way round, the request lands in
            // other registers and is waited for at once to be copied into the slot -- no load-ahead at all)
#pragma unroll
            for (int e = 0; e < N; ++e)
                asm volatile("" : "+v"(a[e]));
            __builtin_amdgcn_sched_barrier(0);
            fetch32(qq, j + SMP_MPF);
        } else {
            ci_load<N>(alpha_ci, ci_rec(g, s, ch.Lmax), lane, a);
        }
        const double r = u ? u[base + s] : uniform01(seed, (uint64_t)(sbase + s));
        const bool last = (t0 + s == Tk - 1);
        // Only the images that are still alive need a draw: the next state of this step is one of
        // the nibbles of cur, whatever state the part is entered with.  Their number falls quickly
        // (8 -> 2..3 within a few steps -> 1), so the loop below is short on most steps; a lane
        // whose images have coalesced (or at the last step of a trajectory, a constant map) draws
        // once.
        // (alive_set = the set of nibbles of cur over the real entry states, carried along)
        uint32_t alive = alive_set;
        if (last)
            alive = 1u; // the draw does not depend on a next state (_hidden.c:347-355)
        const bool one = (alive & (alive - 1)) == 0;
        uint32_t G = 0;
        bool clear = R32;
        if (clear) { // decide from the fp32 row where that is safe
            uint32_t todo = alive;
            while (todo) {
                const int x = __ffs(todo) - 1;
                todo &= todo - 1;
                const int y = pick_clear<N>(a, last ? nullptr : sAt + x * N, r, n, SMP_TOL32);
                clear = clear && y >= 0;
                G |= (uint32_t)(y & 7) << (4 * x);
            }
        }
        if (__builtin_expect(!clear, 0)) { // the exact fp64 row, the reference's decision rule
            if (R32)
                exact_row(s, a);
            G = 0;
            uint32_t todo = alive;
            while (todo) {
                const int x = __ffs(todo) - 1;
                todo &= todo - 1;
                double gap;
                const int y = pick_state<N, true>(a, last ? nullptr : sAt + x * N, r, n, status, &gap);
                // a draw within reach of the deviation these alpha rows were verified to: recorded, decided
                // again on the serial recursion afterwards (draw_verify.hpp; only rows of a speculative pass)
                if (__builtin_expect(watch.tol > 0.0 && R32 && y >= 0 && gap <= watch.tol, 0))
                    draw_record(watch, ch.traj[g], t0 + s, x, r, y, gap); // (k re-read: not kept live for this)
                // (bit 3 of a nibble: no state can precede next state x here)
                G |= (y < 0 ? (8u | (uint32_t)(n - 1)) : (uint32_t)y) << (4 * x);
            }
        }
        // the images of the next step down: G over the alive set
        {
            uint32_t nset = 0, todo = alive;
            while (todo) {
                const int x = __ffs(todo) - 1;
                todo &= todo - 1;
                nset |= 1u << ((G >> (4 * x)) & 7u);
            }
            alive_set = nset;
        }
        if (one) {
            const uint32_t e1 = (G >> (4 * (__ffs(alive) - 1))) & 15u, x1 = e1 & 7u;
            if (e1 & 8u) // the path itself has no possible predecessor (_hidden.c:299-304)
                *status = BHMM_ERR_CHOICE;
            cur = x1 * 0x11111111u;
            if (last) {
                dm = s; // no successor state: k_smp_apply takes this step (a constant map)
                mygw[(int64_t)j * Gp] = cur;
            }
            word |= x1 << (4 * (j & 7));
        } else {
            dm = s;
            mygw[(int64_t)j * Gp] = G; // (entries of states that are not alive are never looked up)
            uint32_t nw = 0;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj)
                nw |= ((G >> (4 * ((cur >> (4 * jj)) & 7u))) & 7u) << (4 * jj);
            cur = nw;
        }
        if ((j & 7) == 7 || s == s_lo) {
            mynib[(int64_t)(j >> 3) * Gp] = word;
            word = 0;
        }
    }
    }
    Fmap[g * P + part] = cur;
    dmark[g * P + part] = dm;
}

// state at the first step of the NEXT part, for every (chunk, part) (0 behind a trajectory's
// last part, whose map is constant): walking a trajectory's maps from its end, next_state[i] = x,
// x <- F_i(x).  One wavefront per trajectory; the chain is not walked serially: every lane
// composes SMP_STITCH_SEG consecutive maps into one (8 nibbles: y -> F(y)), a wave-wide scan of
// those composites gives every lane the state its segment is entered with, and the lane re-walks
// its own segment.  (1024 maps per trajectory on configs[4]: 65 -> a few microseconds.)
constexpr int SMP_STITCH_SEG = 16;
__device__ __forceinline__ uint32_t smp_compose(uint32_t first, uint32_t then)
{
    uint32_t c = 0; // c(y) = then(first(y))
#pragma unroll
    for (int y = 0; y < 8; ++y)
        c |= ((then >> (4 * ((first >> (4 * y)) & 7u))) & 7u) << (4 * y);
    return c;
}
[[maybe_unused]] static __global__ __launch_bounds__(64) void k_smp_stitch(const int32_t *traj_c0, int K, int P,
                                                          const uint32_t *Fmap, int32_t *next_state,
                                                          const int32_t *start = nullptr)
{
    const int k = blockIdx.x;
    if (k >= K)
        return;
    const int lane = threadIdx.x;
    uint32_t x0 = start ? (uint32_t)start[k] : 0u; // Viterbi: the final state of the trajectory
    const int64_t lo = (int64_t)traj_c0[k] * P;
    int64_t hi = (int64_t)traj_c0[k + 1] * P - 1;
    constexpr int TILE = 64 * SMP_STITCH_SEG;
    while (hi >= lo) {
        const int64_t cnt = hi - lo + 1 < TILE ? hi - lo + 1 : TILE;
        // my segment: maps hi - (lane * SEG + j), j = 0 .. SEG-1 (walked in this order)
        uint32_t f[SMP_STITCH_SEG];
        uint32_t comp = 0x76543210u; // identity
#pragma unroll
        for (int j = 0; j < SMP_STITCH_SEG; ++j) {
            const int64_t i = (int64_t)lane * SMP_STITCH_SEG + j;
            f[j] = i < cnt ? Fmap[hi - i] : 0x76543210u;
        }
#pragma unroll
        for (int j = 0; j < SMP_STITCH_SEG; ++j)
            comp = smp_compose(comp, f[j]);
        // inclusive scan over lanes: M_l = comp_l o ... o comp_0 (comp_0 applied first)
        uint32_t M = comp;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t other = (uint32_t)__shfl_up((int)M, d, 64);
            if (lane >= d)
                M = smp_compose(other, M);
        }
        const uint32_t Mprev = (uint32_t)__shfl_up((int)M, 1, 64);
        uint32_t x = lane == 0 ? x0 : ((Mprev >> (4 * x0)) & 7u);
#pragma unroll
        for (int j = 0; j < SMP_STITCH_SEG; ++j) {
            const int64_t i = (int64_t)lane * SMP_STITCH_SEG + j;
            if (i < cnt) {
                next_state[hi - i] = (int32_t)x;
                x = (f[j] >> (4 * x)) & 7u;
            }
        }
        const uint32_t Mall = (uint32_t)__shfl((int)M, 63, 64);
        x0 = (Mall >> (4 * x0)) & 7u; // state entering the next tile
        hi -= cnt;
    }
}

// The same chain walked by ONE lane per trajectory: for batches of many short trajectories (a
// few maps each), where a wavefront per trajectory would be mostly idle.
constexpr int SMP_STITCH_TPB = 8;
[[maybe_unused]] static __global__ __launch_bounds__(64) void k_smp_stitch_serial(const int32_t *traj_c0, int K, int P,
                                                          const uint32_t *Fmap, int32_t *next_state,
                                                          const int32_t *start = nullptr)
{
    // every lane walks its own trajectory, so each wavefront-wide load / store touches one cache
    // line per active lane: few lanes per workgroup, many workgroups
    const int k = blockIdx.x * SMP_STITCH_TPB + threadIdx.x;
    if (threadIdx.x >= SMP_STITCH_TPB || k >= K)
        return;
    uint32_t x = start ? (uint32_t)start[k] : 0u; // Viterbi: the final state of the trajectory
    // the maps are loaded 16 at a time (their addresses do not depend on the walk), the walk itself
    // is register arithmetic
    int64_t hi = (int64_t)traj_c0[k + 1] * P - 1;
    const int64_t lo = (int64_t)traj_c0[k] * P;
    while (hi >= lo) {
        const int cnt = (int)((hi - lo + 1) < 16 ? (hi - lo + 1) : 16);
        uint32_t f[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) // clamped, unconditional: all 16 loads go out together
            f[j] = Fmap[hi - j > lo ? hi - j : lo];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < cnt) {
                next_state[hi - j] = (int32_t)x;
                x = (f[j] >> (4 * x)) & 7u;
            }
        }
        hi -= cnt;
    }
}

// walk every part from its known start state with what k_smp_maps left behind -- the full step maps
// above the part's coalescence point, one nibble per step below it -- and emit the sampled path
// (optional output) and the hidden-path statistics of the Gibbs sweep (generic_hmm.py:297-334,
// 398-431): integer transition / start counts (exact) and per-state emission sums, fused so that
// neither the path nor the observations are read again.  alpha and the uniforms are not needed here.
template <int N, int KIND>
__global__ __launch_bounds__(256) void k_smp_apply(const Model<N> m, const Chunks ch,
                                                   const int64_t *off, const void *obs_ci, int P,
                                                   const int32_t *next_state, int32_t *path,
                                                   unsigned long long *counts, double *epartials,
                                                   const int32_t *dmark, const uint32_t *nib, int W8,
                                                   int64_t Gp, const uint32_t *gw, int Lp, int *status = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) double lds[]; // discrete: [M][N] counts
    __shared__ unsigned int cnt[N * N + N];
    __shared__ double red[4][3 * N];
    // alphabets too large for an LDS table (m.bt_global): the symbol counts go to one of
    // DISC_GLOBAL_TABLES global tables in epartials (zeroed by the host) -- integer-valued, so
    // the sums stay exact and order-independent
    const bool big = KIND == EMIT_DISC && m.bt_global;
    double *gtab = epartials + (int64_t)(blockIdx.x % DISC_GLOBAL_TABLES) * (m.M * N);
    for (int e = threadIdx.x; e < N * N + N; e += blockDim.x)
        cnt[e] = 0u;
    if constexpr (KIND == EMIT_DISC)
        if (!big)
            for (int e = threadIdx.x; e < m.M * N; e += blockDim.x)
                lds[e] = 0.0;
    __syncthreads();
    const int part = blockIdx.x % P;
    const int64_t g = (int64_t)(blockIdx.x / P) * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int len = ch.len[g];
    double s0[N], s1[N], s2[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
        s0[i] = s1[i] = s2[i] = 0.0;
    // common shift of the gaussian moments: the mean of the state means (keeps |o - shift| of the
    // order of the spread of the means: the re-centring below loses nothing that matters)
    double cshift = 0.0;
    if constexpr (KIND == EMIT_GAUSS) {
        for (int i = 0; i < m.nreal; ++i)
            cshift += m.e0[i];
        cshift /= (double)m.nreal;
    }
    if (len > 0) {
        const int k = ch.traj[g];
        const int64_t t0 = ch.t0[g], base = ch.goff[g];
        const int64_t Tk = off[k + 1] - off[k];
        int s_lo, s_hi;
        smp_part_bounds(len, part, P, s_lo, s_hi);
        int nxt = next_state[g * P + part];
        const int dm = dmark[g * P + part]; // steps below dm: the state itself is recorded
        const uint32_t *mynib = nib + ((int64_t)part * W8) * Gp + g;
        const uint32_t *mygw = gw + ((int64_t)part * Lp) * Gp + g;
        // The walk is a dependent chain of table look-ups, one observation per step: everything it
        // reads is fetched SMP_PF steps ahead (a step would otherwise wait a full memory latency,
        // and there are only two wavefronts per SIMD to hide it).
        constexpr int SMP_PF = 4;
        const int nst = s_hi - s_lo;
        double oring[SMP_PF];
        int yring[SMP_PF];
        uint32_t gring[SMP_PF];
        auto fetch = [&](int q, int j) {
            const int jc = j < nst ? j : nst - 1;
            const int sc = s_hi - 1 - jc;
            const int64_t r = ci_rec(g, sc, ch.Lmax) * 64 + lane;
            if constexpr (KIND == EMIT_GAUSS)
                oring[q] = static_cast<const double *>(obs_ci)[r];
            if constexpr (KIND == EMIT_DISC)
                yring[q] = static_cast<const int32_t *>(obs_ci)[r];
            // above dm the step's full map, below it the word that holds the step's nibble
            gring[q] = sc >= dm ? mygw[(int64_t)jc * Gp] : mynib[(int64_t)(jc >> 3) * Gp];
        };
#pragma unroll
        for (int q = 0; q < SMP_PF; ++q) {
            oring[q] = 0.0;
            yring[q] = 0;
            gring[q] = 0u;
            if (nst > 0)
                fetch(q, q);
        }
        for (int jb = 0; jb < nst; jb += SMP_PF) {
#pragma unroll
        for (int q = 0; q < SMP_PF; ++q) {
            const int j = jb + q;
            if (j >= nst)
                break;
            const int s = s_hi - 1 - j;
            const double o_cur = oring[q];
            const int y_cur = yring[q];
            const uint32_t g_cur = gring[q];
            fetch(q, j + SMP_PF);
            (void)o_cur;
            (void)y_cur;
            const bool last = (t0 + s == Tk - 1);
            const int st = (int)((g_cur >> (4 * (s >= dm ? nxt : (j & 7)))) & 7u);
            if (s >= dm && ((g_cur >> (4 * nxt)) & 8u) && status) // entry marked by k_smp_maps:
                *status = BHMM_ERR_CHOICE;                        // no state can precede nxt here
            if (path)
                path[base + s] = st;
            if (!last)
                atomicAdd(&cnt[st * N + nxt], 1u);
            if (t0 + s == 0)
                atomicAdd(&cnt[N * N + st], 1u);
            if constexpr (KIND == EMIT_GAUSS) {
                // moments about the common shift cshift (one subtraction and one square per step,
                // an exact 0/1 weight per state); re-centred on the state means after the walk
                const double dc = o_cur - cshift, dd = dc * dc;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double w = (st == i) ? 1.0 : 0.0;
                    s0[i] += w;
                    s1[i] = fma(w, dc, s1[i]);
                    s2[i] = fma(w, dd, s2[i]);
                }
            }
            if constexpr (KIND == EMIT_DISC) {
                const int sym = y_cur;
                if (big)
                    atomicAdd(&gtab[(int64_t)sym * N + st], 1.0);
                else
                    atomicAdd(&lds[sym * N + st], 1.0); // integer-valued: exact, order-independent
            }
            nxt = st;
        }
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < N * N + N; e += blockDim.x)
        if (cnt[e])
            atomicAdd(&counts[e], (unsigned long long)cnt[e]);
    if constexpr (KIND == EMIT_GAUSS) {
        const int wv = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            // sum (o - mu_i)^k from the moments about cshift: e = mu_i - cshift
            const double e = m.e0[i] - cshift;
            s2[i] = s2[i] - 2.0 * e * s1[i] + e * e * s0[i];
            s1[i] = s1[i] - e * s0[i];
            const double a = wave_sum(s0[i]), b = wave_sum(s1[i]), c = wave_sum(s2[i]);
            if (lane == 0) {
                red[wv][i] = a;
                red[wv][N + i] = b;
                red[wv][2 * N + i] = c;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 3 * N; e += blockDim.x)
            epartials[(int64_t)blockIdx.x * 3 * N + e] =
                ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    }
    if constexpr (KIND == EMIT_DISC)
        if (!big)
            for (int e = threadIdx.x; e < m.M * N; e += blockDim.x)
                epartials[(int64_t)blockIdx.x * m.M * N + e] = lds[e];
}

// ---- small reference-shaped kernels on row-major arrays ------------------------------------
// gamma_t = alpha_t o beta_t / sum (hidden/api.py:176-186); one thread per time step.
[[maybe_unused]] static __global__ void k_gamma_rows(const double *alpha, const double *beta, double *gamma, int n,
                             int64_t T)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T)
        return;
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const double g = alpha[t * n + i] * beta[t * n + i];
        gamma[t * n + i] = g;
        s += g;
    }
    for (int i = 0; i < n; ++i)
        gamma[t * n + i] /= s;
}

// Gaussian pdf rows (_gaussian.c:45-70) + outlier rule; one thread per time step.
[[maybe_unused]] static __global__ void k_pobs_gaussian(const double *obs, const double *mu, const double *sigma,
                                double *pobs, int n, int64_t T, int ignore_outliers)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T)
        return;
    const double o = obs[t];
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        const double C = 1.0 / (sqrt(2.0 * M_PI) * sigma[i]);
        const double d = (o - mu[i]) / sigma[i];
        const double p = C * exp(-0.5 * d * d);
        pobs[t * n + i] = p;
        s += p;
    }
    if (ignore_outliers && s == 0.0)
        for (int i = 0; i < n; ++i)
            pobs[t * n + i] = 1.0;
}

// xi-counts from given alpha, beta (_hidden.c:148-183).  One thread per pair (t, t+1):
// x[i][j] = alpha_t[i] A[i][j] pobs_{t+1}[j] beta_{t+1}[j];  C += x / sum(x).
// Per-thread accumulation over a strided set of t, then LDS + per-block partials; the
// final fixed-order sum is done by k_sum_partials.  n <= 8.
template <int N>
__global__ __launch_bounds__(256) void k_xi_rows(const double *A, const double *pobs,
                                                 const double *alpha, const double *beta, int n,
                                                 int64_t T, double *partials)
{
    __shared__ double red[4][N * N];
    double acc[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j)
            acc[i][j] = 0.0;
    double Al[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j)
            Al[i][j] = (i < n && j < n) ? A[i * n + j] : 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t + 1 < T; t += stride) {
        double a[N], b[N], x[N][N], S = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            a[i] = (i < n) ? alpha[t * n + i] : 0.0;
            b[i] = (i < n) ? pobs[(t + 1) * n + i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double bj = (j < n) ? beta[(t + 1) * n + j] : 0.0;
                x[i][j] = a[i] * Al[i][j] * b[j] * bj; // left-to-right, _hidden.c:174
                S += x[i][j];
            }
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int j = 0; j < N; ++j)
                acc[i][j] += x[i][j] / S;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double v = wave_sum(acc[i][j]);
            if (lane == 0)
                red[wv][i * N + j] = v;
        }
    __syncthreads();
    for (int e = threadIdx.x; e < N * N; e += blockDim.x)
        partials[(int64_t)blockIdx.x * (N * N) + e] =
            ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
}

template <int N>
__global__ void k_sum_partials(const double *partials, int nblocks, int n, double *C)
{
    const int e = threadIdx.x;
    if (e >= N * N)
        return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b)
        s += partials[(int64_t)b * (N * N) + e];
    const int i = e / N, j = e % N;
    if (i < n && j < n)
        C[i * n + j] = s;
}

// weighted symbol counts pout[i][obs[t]] += w[t][i] (_discrete.c:1-32): one block per
// state-row slab; fp64 atomics into an LDS histogram, then one ordered flush.
[[maybe_unused]] static __global__ void k_update_pout(const int32_t *obs, const double *w, int64_t T, int n, int M,
                              double *pout_partials)
{
    extern __shared__ double hist[]; // [n][M]
    for (int e = threadIdx.x; e < n * M; e += blockDim.x)
        hist[e] = 0.0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += stride) {
        const int o = obs[t];
        for (int i = 0; i < n; ++i)
            atomicAdd(&hist[i * M + o], w[t * n + i]);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n * M; e += blockDim.x)
        pout_partials[(int64_t)blockIdx.x * n * M + e] = hist[e];
}

// dst[e] += sum_b partials[b][e]: one wavefront per entry (a single thread walking all blocks
// is bound by the latency of its dependent loads), fixed summation tree
[[maybe_unused]] static __global__ __launch_bounds__(64) void k_add_partials(const double *partials, int nblocks, int count,
                                                     double *dst)
{
    const int e = blockIdx.x;
    if (e >= count)
        return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64)
        s += partials[(int64_t)b * count + e];
    s = wave_sum(s);
    if (threadIdx.x == 0)
        dst[e] += s;
}

// Hidden-path statistics as ONE packed fp64 vector in the caller's layout,
// [counts n*n | n0 n | emission block], so that several ranks can sum them with a single
// all-reduce.  Counts are integers below 2^53: their fp64 sums are exact in any order.
//   cnt  : [Ns*Ns + Ns] integer counts with row stride Ns (the padded state count)
//   ered : gaussian [3][Ns]; discrete [M][Ns] when `transposed`, else [n][M]
[[maybe_unused]] static __global__ void k_pack_path_stats(const unsigned long long *cnt, const double *ered, int n, int Ns,
                                  int M, int kind, int transposed, double *out)
{
    const int nn = n * n;
    const int esz = kind == EMIT_GAUSS ? 3 * n : (kind == EMIT_DISC ? n * M : 0);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nn + n + esz;
         e += gridDim.x * blockDim.x) {
        double v;
        if (e < nn)
            v = (double)cnt[(e / n) * Ns + e % n];
        else if (e < nn + n)
            v = (double)cnt[Ns * Ns + (e - nn)];
        else {
            const int q = e - nn - n;
            if (kind == EMIT_GAUSS)
                v = ered[(q / n) * Ns + q % n];
            else
                v = transposed ? ered[(size_t)(q % M) * Ns + q / M] : ered[q];
        }
        out[e] = v;
    }
}

// =========================================================================================
// 9..64 states: same order-faithful recursions, NP lanes per trajectory, vectors exchanged
// through LDS so that every lane can take the index-ordered sums the reference takes.
// Back-pointers: one byte per (t, j), trajectory-major [T][n].
// =========================================================================================
// U trajectories are advanced in lock-step by one lane group: the recursion of a single
// trajectory is one long dependency chain (LDS exchange -> argmax -> LDS exchange -> ordered
// sum -> IEEE division), so a wavefront alone on its SIMD is latency bound; U independent chains
// interleave in the same instruction stream and hide each other's latencies.
template <int NP, int KIND, int U>
__global__ __launch_bounds__(64) void k_wide_viterbi_fwd(const WideModel m, const int64_t *off,
                                                         int K, const void *obs_rm,
                                                         uint8_t *ptr, int32_t *last_state)
{
    constexpr int GP = 64 / NP;
    constexpr int TL = NP < 16 ? NP : 16; // argmax tile
    __shared__ __attribute__((aligned(16))) double xv[GP][U][NP];
    __shared__ __attribute__((aligned(16))) double xn[GP][U][NP];
    const int lane = threadIdx.x;
    const int gi = lane / NP, j = lane % NP;
    const int kbase = (blockIdx.x * GP + gi) * U;
    if (kbase >= K)
        return;
    const int n = m.n;
    const bool real = j < n;
    const unsigned long long gmask = wgroup_mask<NP>(lane);
    double Acol[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        Acol[i] = (real && i < n) ? m.A[(int64_t)i * n + j] : 0.0;
    const double mu_j = (KIND == EMIT_GAUSS && real) ? m.mu[j] : 0.0;
    const double sg_j = (KIND == EMIT_GAUSS && real) ? m.sigma[j] : 1.0;
    const double cn_j = (KIND == EMIT_GAUSS && real) ? m.cnorm[j] : 0.0;
    const double pi_j = real ? m.pi[j] : 0.0;
    int64_t o0[U], T[U], Tmax = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int k = kbase + u;
        o0[u] = (k < K) ? off[k] : 0;
        T[u] = (k < K) ? off[k + 1] - off[k] : 0;
        Tmax = T[u] > Tmax ? T[u] : Tmax;
    }
    // emission probability of my state at global step gt (+ outlier rule over the group)
    auto emis = [&](int64_t gt) {
        double p;
        if constexpr (KIND == EMIT_GAUSS) {
            const double o = static_cast<const double *>(obs_rm)[gt];
            const double d = (o - mu_j) / sg_j;
            p = real ? cn_j * exp_nonpos(-0.5 * d * d) : 0.0; // _gaussian.c:18-20
            if ((__ballot(p != 0.0) & gmask) == 0ull)
                p = real ? 1.0 : 0.0;
        } else if constexpr (KIND == EMIT_DISC) {
            const int sym = static_cast<const int32_t *>(obs_rm)[gt];
            p = real ? m.B[(int64_t)j * m.M + sym] : 0.0;
        } else {
            p = real ? static_cast<const double *>(obs_rm)[gt * n + j] : 0.0;
        }
        return p;
    };
    double v[U], p_next[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        v[u] = 0.0;
        p_next[u] = (T[u] > 0) ? emis(o0[u]) : 0.0;
    }
    for (int64_t t = 0; t < Tmax; ++t) {
        double p[U], vn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            p[u] = p_next[u];
            if (t + 1 < T[u])
                p_next[u] = emis(o0[u] + t + 1); // independent of the recursion
        }
        if (t > 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                xv[gi][u][j] = v[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (t == 0) {
                vn[u] = p[u] * pi_j; // _hidden.c:232
            } else {
                // first-maximum argmax (_hidden.c:186-200) as a select tree per tile of <= 16
                // states: the later candidate wins only if strictly greater, which is exactly
                // the linear scan's result; tiles are folded in index order with the same rule
                double bh = 0.0, bv = 0.0, bA = 0.0;
                int bi = 0;
#pragma unroll
                for (int tl = 0; tl < NP; tl += TL) {
                    double vv[TL], hh[TL], aa[TL];
                    int ii[TL];
#pragma unroll
                    for (int i = 0; i < TL; i += 2) {
                        const double2 x = *reinterpret_cast<const double2 *>(&xv[gi][u][tl + i]);
                        vv[i] = x.x;
                        vv[i + 1] = x.y;
                    }
#pragma unroll
                    for (int i = 0; i < TL; ++i) {
                        aa[i] = Acol[tl + i];
                        hh[i] = vv[i] * aa[i]; // _hidden.c:249
                        ii[i] = tl + i;
                    }
#define BHMM_ARGMAX_LEVEL(W)                                           \
    if constexpr (TL > W) {                                            \
        _Pragma("unroll") for (int i = 0; i + W < TL; i += 2 * W)      \
        {                                                              \
            const bool take = hh[i + W] > hh[i];                       \
            hh[i] = take ? hh[i + W] : hh[i];                          \
            vv[i] = take ? vv[i + W] : vv[i];                          \
            aa[i] = take ? aa[i + W] : aa[i];                          \
            ii[i] = take ? ii[i + W] : ii[i];                          \
        }                                                              \
    }
                    BHMM_ARGMAX_LEVEL(1)
                    BHMM_ARGMAX_LEVEL(2)
                    BHMM_ARGMAX_LEVEL(4)
                    BHMM_ARGMAX_LEVEL(8)
#undef BHMM_ARGMAX_LEVEL
                    const bool take = (tl == 0) || (hh[0] > bh);
                    bh = take ? hh[0] : bh;
                    bv = take ? vv[0] : bv;
                    bA = take ? aa[0] : bA;
                    bi = take ? ii[0] : bi;
                }
                if (real && t < T[u])
                    ptr[(o0[u] + t) * n + j] = (uint8_t)bi;
                vn[u] = p[u] * bv * bA; // _hidden.c:253
            }
            xn[gi][u][j] = vn[u];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double S = 0.0;
#pragma unroll
            for (int tl = 0; tl < NP; tl += TL) {
                double xs[TL];
#pragma unroll
                for (int i = 0; i < TL; i += 2) {
                    const double2 x = *reinterpret_cast<const double2 *>(&xn[gi][u][tl + i]);
                    xs[i] = x.x;
                    xs[i + 1] = x.y;
                }
#pragma unroll
                for (int i = 0; i < TL; ++i)
                    S += xs[i]; // ascending order; padded states add exact zeros
            }
            if (t < T[u]) // a finished trajectory keeps its final v
                v[u] = vn[u] / S;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
        xv[gi][u][j] = v[u];
    if (j == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (kbase + u < K && T[u] > 0) {
                double bm = xv[gi][u][0];
                int bi = 0;
                for (int i = 1; i < n; ++i)
                    if (xv[gi][u][i] > bm) {
                        bm = xv[gi][u][i];
                        bi = i;
                    }
                last_state[kbase + u] = bi;
            }
        }
    }
}

// =========================================================================================
// 9..64 states, parallel over time segments (round 4): the same order-faithful step as
// k_wide_viterbi_fwd, one lane group per SEGMENT instead of per trajectory, and bit-identical to the
// serial run by construction:
//   pass 0   every segment that does not start its trajectory begins W steps early from the uniform
//            vector (the max-product recursion forgets its start: the survivors coalesce);
//   check    k_wide_vit_check compares the vector a segment arrived with at its first step (v_entry)
//            BITWISE with the one its predecessor computed there (v_exit).  In the typical run nine
//            boundaries in ten are identical to the bit (the two runs, a few ulp apart after the
//            survivors met, fall onto the same sequence of doubles); the others are flagged, and
//            their v_entry is replaced by the predecessor's vector;
//   fix-up   (FIX = true) a flagged segment is run again from that exact vector, without warm-up.  The
//            first pass left its vector every 64th step (`ckpt`); as soon as the repeated run
//            reproduces one of them bitwise, everything behind it -- back-pointers, v_exit -- is what
//            the first pass wrote, and the segment stops.  One that reaches its end writes a new
//            v_exit, which the next check compares with its successor's entry.
// When a check finds no difference, every segment started from the serial run's vector (induction
// from the first segment of each trajectory, which starts from pi exactly), so every back-pointer is
// the serial run's.  The host bounds the number of rounds and runs the serial kernel beyond it.
// Argmax: the select tree carries (product, index) only; v[i^] (a lane read) and A[i^][j] (LDS, shared
// by the WPB wavefronts of a workgroup) are looked up afterwards: three selects per node instead of seven.
//   v_entry / v_exit [nseg][NP];  ckpt [(total >> 6) + 1][NP];  flag [nseg]
// =========================================================================================
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// h[i] = fma(v[16 ROW + B + i], w(B + i), 0), i = 0..7, with v given as its row copy `src`: ONE assembly
// block per eight products (as single statements the compiler pads every one with an s_nop: its hazard
// model cannot see which operand of an inline statement is the DPP one).
template <int B, int HB = B, int HN = 16, typename WF>
__device__ __forceinline__ void prod8_bcast(double (&h)[HN], const double &src, WF &&w)
{
#define BHMM_W(I) "v"(w(std::integral_constant<int, B + I>{}))
    asm volatile("v_mov_b64 %0, 0\n\tv_mov_b64 %1, 0\n\tv_mov_b64 %2, 0\n\tv_mov_b64 %3, 0\n\t"
                 "v_mov_b64 %4, 0\n\tv_mov_b64 %5, 0\n\tv_mov_b64 %6, 0\n\tv_mov_b64 %7, 0\n\t"
                 "v_fmac_f64_dpp %0, %8, %9 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %1, %8, %10 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %2, %8, %11 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %3, %8, %12 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %4, %8, %13 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %5, %8, %14 row_newbcast:%22 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %6, %8, %15 row_newbcast:%23 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %7, %8, %16 row_newbcast:%24 row_mask:0xf bank_mask:0xf"
                 : "=&v"(h[HB + 0]), "=&v"(h[HB + 1]), "=&v"(h[HB + 2]), "=&v"(h[HB + 3]), "=&v"(h[HB + 4]),
                   "=&v"(h[HB + 5]), "=&v"(h[HB + 6]), "=&v"(h[HB + 7])
                 : "v"(src), BHMM_W(0), BHMM_W(1), BHMM_W(2), BHMM_W(3), BHMM_W(4), BHMM_W(5), BHMM_W(6),
                   BHMM_W(7), "n"(B + 0), "n"(B + 1), "n"(B + 2), "n"(B + 3), "n"(B + 4), "n"(B + 5),
                   "n"(B + 6), "n"(B + 7));
#undef BHMM_W
}

// S = (...((S + v[16 ROW]) + v[16 ROW + 1]) + ...) + v[16 ROW + 15], every sum rounded once
// (fma(x, 1, S)), with v given as its row copy `src`
__device__ __forceinline__ void sum16_bcast(double &S, const double &src, const double &one)
{
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "+v"(S)
                 : "v"(src), "v"(one));
}

// the four rows of a 64-state vector in one block (one leading wait instead of one per row and the compiler's
// protective s_nop between the blocks)
__device__ __forceinline__ void sum64_bcast(double &S, const double &r0, const double &r1, const double &r2,
                                            const double &r3, const double &one)
{
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %1, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %2, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %3, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp %0, %4, %5 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                 : "+v"(S)
                 : "v"(r0), "v"(r1), "v"(r2), "v"(r3), "v"(one));
}

__device__ __forceinline__ double gauss_exp_block(double x, double cn); // (below: cn * exp_nonpos(x), one block)

// First maximum of eight NaN-free non-negative candidates (`later > earlier`, strict: _hidden.c:186-200), as ONE
// block: value in h[0], index BASE + position in iw.  Compares into scalar pairs, v_max_f64 for the values,
// v_cndmask for the indices, ordered so that every select finds its mask two wait states old -- 22 instructions,
// one of them an s_nop; the compiler's tree takes 21 plus seven s_nop.  h[2], h[4], h[6] are destroyed.
template <int BASE>
__device__ __forceinline__ void argmax_oct_const(double (&h)[8], int &iw)
{
    int i1, i2, i3;
    unsigned long long s0, s1, s2, s3;
    asm("v_cmp_gt_f64 %[s0], %[h1], %[h0]\n\t"
        "v_cmp_gt_f64 %[s1], %[h3], %[h2]\n\t"
        "v_cmp_gt_f64 %[s2], %[h5], %[h4]\n\t"
        "v_cmp_gt_f64 %[s3], %[h7], %[h6]\n\t"
        "v_max_f64 %[h0], %[h0], %[h1]\n\t"
        "v_max_f64 %[h2], %[h2], %[h3]\n\t"
        "v_max_f64 %[h4], %[h4], %[h5]\n\t"
        "v_max_f64 %[h6], %[h6], %[h7]\n\t"
        "v_cndmask_b32_e64 %[i0], %[c0], %[c1], %[s0]\n\t"
        "v_cndmask_b32_e64 %[i1], %[c2], %[c3], %[s1]\n\t"
        "v_cndmask_b32_e64 %[i2], %[c4], %[c5], %[s2]\n\t"
        "v_cndmask_b32_e64 %[i3], %[c6], %[c7], %[s3]\n\t"
        "v_cmp_gt_f64 %[s0], %[h2], %[h0]\n\t"
        "v_cmp_gt_f64 %[s1], %[h6], %[h4]\n\t"
        "v_max_f64 %[h0], %[h0], %[h2]\n\t"
        "v_max_f64 %[h4], %[h4], %[h6]\n\t"
        "v_cndmask_b32_e64 %[i0], %[i0], %[i1], %[s0]\n\t"
        "v_cndmask_b32_e64 %[i2], %[i2], %[i3], %[s1]\n\t"
        "v_cmp_gt_f64 %[s0], %[h4], %[h0]\n\t"
        "v_max_f64 %[h0], %[h0], %[h4]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 %[i0], %[i0], %[i2], %[s0]"
        : [h0] "+v"(h[0]), [h2] "+v"(h[2]), [h4] "+v"(h[4]), [h6] "+v"(h[6]), [i0] "=&v"(iw), [i1] "=&v"(i1),
          [i2] "=&v"(i2), [i3] "=&v"(i3), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)
        : [h1] "v"(h[1]), [h3] "v"(h[3]), [h5] "v"(h[5]), [h7] "v"(h[7]), [c0] "n"(BASE), [c1] "n"(BASE + 1),
          [c2] "n"(BASE + 2), [c3] "n"(BASE + 3), [c4] "n"(BASE + 4), [c5] "n"(BASE + 5), [c6] "n"(BASE + 6),
          [c7] "n"(BASE + 7));
}
// The products feeding such a tree IN the block (one block = eight candidates end to end: the compiler pads the seam
// between two blocks with s_nop, and loads consumed by one block need one s_waitcnt instead of one per load).
#define BHMM_OCT_TREE                                                                              \
    "v_cmp_gt_f64 %[s0], %[h1], %[h0]\n\t"                                                         \
    "v_cmp_gt_f64 %[s1], %[h3], %[h2]\n\t"                                                         \
    "v_max_f64 %[h0], %[h0], %[h1]\n\t"                                                            \
    "v_max_f64 %[h2], %[h2], %[h3]\n\t"                                                            \
    "v_cndmask_b32_e64 %[i0], %[ib], %[ib]+1, %[s0]\n\t"                                           \
    "v_cndmask_b32_e64 %[i1], %[ib]+2, %[ib]+3, %[s1]\n\t"                                         \
    "v_cmp_gt_f64 %[s0], %[h5], %[h4]\n\t"                                                         \
    "v_cmp_gt_f64 %[s1], %[h7], %[h6]\n\t"                                                         \
    "v_max_f64 %[h4], %[h4], %[h5]\n\t"                                                            \
    "v_max_f64 %[h6], %[h6], %[h7]\n\t"                                                            \
    "v_cndmask_b32_e64 %[i2], %[ib]+4, %[ib]+5, %[s0]\n\t"                                         \
    "v_cndmask_b32_e64 %[i3], %[ib]+6, %[ib]+7, %[s1]\n\t"                                         \
    "v_cmp_gt_f64 %[s0], %[h2], %[h0]\n\t"                                                         \
    "v_cmp_gt_f64 %[s1], %[h6], %[h4]\n\t"                                                         \
    "v_max_f64 %[h0], %[h0], %[h2]\n\t"                                                            \
    "v_max_f64 %[h4], %[h4], %[h6]\n\t"                                                            \
    "v_cndmask_b32_e64 %[i0], %[i0], %[i1], %[s0]\n\t"                                             \
    "v_cndmask_b32_e64 %[i2], %[i2], %[i3], %[s1]\n\t"                                             \
    "v_cmp_gt_f64 %[s0], %[h4], %[h0]\n\t"                                                         \
    "v_max_f64 %[h0], %[h0], %[h4]\n\t"                                                            \
    "s_nop 0\n\t"                                                                                  \
    "v_cndmask_b32_e64 %[i0], %[i0], %[i2], %[s0]"
// candidates v[i] * a[i] with v from its LDS copy (y[0..7], already loaded): value of the first maximum in hw
template <int IB>
__device__ __forceinline__ void argmax_oct_mul(const double (&y)[8], const double *a, double &hw, int &iw)
{
    double h1, h2, h3, h4, h5, h6, h7;
    int i1, i2, i3;
    unsigned long long s0, s1;
    asm("v_mul_f64 %[h0], %[y0], %[a0]\n\t"
        "v_mul_f64 %[h1], %[y1], %[a1]\n\t"
        "v_mul_f64 %[h2], %[y2], %[a2]\n\t"
        "v_mul_f64 %[h3], %[y3], %[a3]\n\t"
        "v_mul_f64 %[h4], %[y4], %[a4]\n\t"
        "v_mul_f64 %[h5], %[y5], %[a5]\n\t"
        "v_mul_f64 %[h6], %[y6], %[a6]\n\t"
        "v_mul_f64 %[h7], %[y7], %[a7]\n\t" BHMM_OCT_TREE
        : [h0] "=&v"(hw), [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [h4] "=&v"(h4), [h5] "=&v"(h5),
          [h6] "=&v"(h6), [h7] "=&v"(h7), [i0] "=&v"(iw), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3),
          [s0] "=&s"(s0), [s1] "=&s"(s1)
        : [y0] "v"(y[0]), [y1] "v"(y[1]), [y2] "v"(y[2]), [y3] "v"(y[3]), [y4] "v"(y[4]), [y5] "v"(y[5]),
          [y6] "v"(y[6]), [y7] "v"(y[7]), [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]),
          [a4] "v"(a[4]), [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]), [ib] "n"(IB + 7 <= 64 ? IB : 0));
    if constexpr (IB + 7 > 64) // (not an inline constant any more: the position, then the base)
        iw += IB;
}
// candidates fma(v[16 ROW + LB + i] broadcast within the row, a[i], 0) with v given as its row copy `src`
// (LEAD: `src` was written by the instruction before -- a DPP read needs two wait states)
template <int IB, int LB, bool LEAD>
__device__ __forceinline__ void argmax_oct_bcast(const double &src, const double *a, double &hw, int &iw)
{
    double h1, h2, h3, h4, h5, h6, h7;
    int i1, i2, i3;
    unsigned long long s0, s1;
#define BHMM_OCT_BCAST                                                                                 \
    "v_mov_b64 %[h0], 0\n\tv_mov_b64 %[h1], 0\n\tv_mov_b64 %[h2], 0\n\tv_mov_b64 %[h3], 0\n\t"         \
    "v_mov_b64 %[h4], 0\n\tv_mov_b64 %[h5], 0\n\tv_mov_b64 %[h6], 0\n\tv_mov_b64 %[h7], 0\n\t"         \
    "v_fmac_f64_dpp %[h0], %[src], %[a0] row_newbcast:%[lb] row_mask:0xf bank_mask:0xf\n\t"            \
    "v_fmac_f64_dpp %[h1], %[src], %[a1] row_newbcast:%[lb]+1 row_mask:0xf bank_mask:0xf\n\t"          \
    "v_fmac_f64_dpp %[h2], %[src], %[a2] row_newbcast:%[lb]+2 row_mask:0xf bank_mask:0xf\n\t"          \
    "v_fmac_f64_dpp %[h3], %[src], %[a3] row_newbcast:%[lb]+3 row_mask:0xf bank_mask:0xf\n\t"          \
    "v_fmac_f64_dpp %[h4], %[src], %[a4] row_newbcast:%[lb]+4 row_mask:0xf bank_mask:0xf\n\t"          \
    "v_fmac_f64_dpp %[h5], %[src], %[a5] row_newbcast:%[lb]+5 row_mask:0xf bank_mask:0xf\n\t"          \
    "v_fmac_f64_dpp %[h6], %[src], %[a6] row_newbcast:%[lb]+6 row_mask:0xf bank_mask:0xf\n\t"          \
    "v_fmac_f64_dpp %[h7], %[src], %[a7] row_newbcast:%[lb]+7 row_mask:0xf bank_mask:0xf\n\t" BHMM_OCT_TREE
#define BHMM_OCT_BCAST_OPS                                                                             \
    : [h0] "=&v"(hw), [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [h4] "=&v"(h4), [h5] "=&v"(h5),  \
      [h6] "=&v"(h6), [h7] "=&v"(h7), [i0] "=&v"(iw), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3),  \
      [s0] "=&s"(s0), [s1] "=&s"(s1)                                                                   \
    : [src] "v"(src), [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [a4] "v"(a[4]),  \
      [a5] "v"(a[5]), [a6] "v"(a[6]), [a7] "v"(a[7]), [ib] "n"(IB + 7 <= 64 ? IB : 0), [lb] "n"(LB)
    if constexpr (LEAD)
        asm volatile("s_nop 1\n\t" BHMM_OCT_BCAST BHMM_OCT_BCAST_OPS);
    else
        asm volatile(BHMM_OCT_BCAST BHMM_OCT_BCAST_OPS);
    if constexpr (IB + 7 > 64) // (not an inline constant any more: the position, then the base)
        iw += IB;
#undef BHMM_OCT_BCAST
#undef BHMM_OCT_BCAST_OPS
}
#undef BHMM_OCT_TREE
// the same tree over eight (value, index) pairs -- the winners of eight argmax_oct_const blocks, in index order
__device__ __forceinline__ void argmax_oct_regs(double (&h)[8], const int (&ii)[8], int &iw)
{
    int i1, i2, i3;
    unsigned long long s0, s1, s2, s3;
    asm("v_cmp_gt_f64 %[s0], %[h1], %[h0]\n\t"
        "v_cmp_gt_f64 %[s1], %[h3], %[h2]\n\t"
        "v_cmp_gt_f64 %[s2], %[h5], %[h4]\n\t"
        "v_cmp_gt_f64 %[s3], %[h7], %[h6]\n\t"
        "v_max_f64 %[h0], %[h0], %[h1]\n\t"
        "v_max_f64 %[h2], %[h2], %[h3]\n\t"
        "v_max_f64 %[h4], %[h4], %[h5]\n\t"
        "v_max_f64 %[h6], %[h6], %[h7]\n\t"
        "v_cndmask_b32_e64 %[i0], %[c0], %[c1], %[s0]\n\t"
        "v_cndmask_b32_e64 %[i1], %[c2], %[c3], %[s1]\n\t"
        "v_cndmask_b32_e64 %[i2], %[c4], %[c5], %[s2]\n\t"
        "v_cndmask_b32_e64 %[i3], %[c6], %[c7], %[s3]\n\t"
        "v_cmp_gt_f64 %[s0], %[h2], %[h0]\n\t"
        "v_cmp_gt_f64 %[s1], %[h6], %[h4]\n\t"
        "v_max_f64 %[h0], %[h0], %[h2]\n\t"
        "v_max_f64 %[h4], %[h4], %[h6]\n\t"
        "v_cndmask_b32_e64 %[i0], %[i0], %[i1], %[s0]\n\t"
        "v_cndmask_b32_e64 %[i2], %[i2], %[i3], %[s1]\n\t"
        "v_cmp_gt_f64 %[s0], %[h4], %[h0]\n\t"
        "v_max_f64 %[h0], %[h0], %[h4]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32_e64 %[i0], %[i0], %[i2], %[s0]"
        : [h0] "+v"(h[0]), [h2] "+v"(h[2]), [h4] "+v"(h[4]), [h6] "+v"(h[6]), [i0] "=&v"(iw), [i1] "=&v"(i1),
          [i2] "=&v"(i2), [i3] "=&v"(i3), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)
        : [h1] "v"(h[1]), [h3] "v"(h[3]), [h5] "v"(h[5]), [h7] "v"(h[7]), [c0] "v"(ii[0]), [c1] "v"(ii[1]),
          [c2] "v"(ii[2]), [c3] "v"(ii[3]), [c4] "v"(ii[4]), [c5] "v"(ii[5]), [c6] "v"(ii[6]), [c7] "v"(ii[7]));
}

constexpr int WVS_WPB = 4;
#ifndef WVS_LDS_ROWS
#define WVS_LDS_ROWS 4 // 64 states: rows of 16 candidates whose v comes from LDS (0 .. 4; measured: DESIGN.md section 4)
#endif
template <int NP, int KIND, bool FIX>
__global__ __launch_bounds__(64 * WVS_WPB) void k_wide_viterbi_seg(
    const WideModel m, const int64_t *off, const Segs sg, const void *obs_rm, uint8_t *ptr,
    int32_t *last_state, double *v_entry, double *v_exit, double *ckpt, const uint8_t *flag,
    double *vall = nullptr, double mend_tol = 0.0, unsigned int *notmet = nullptr)
{
    // FIX with mend_tol > 0 ("mending" round, round 6): the segments whose entry vector was further than the
    // tolerance from the predecessor's are run again from the predecessor's vector only until they are within
    // that tolerance of a vector the first pass kept (every 64th step) -- from there on the first pass's
    // vectors stand, a splice like a boundary that is equal to the tolerance (k_vit_margin's budget counts
    // it).  Every vector of the repeated stretch goes to vall; a segment that reaches its end counts in notmet.
    constexpr int GP = 64 / NP;
    __shared__ __attribute__((aligned(16))) double xv[WVS_WPB][GP][NP];
    __shared__ double sA[NP * NP];
    // (64 states: one segment per wavefront -- told to the compiler, so that the step counter, the segment's bounds
    // and every test on them live in scalar registers instead of 64-bit vector compares and exec-mask branches)
    const int w = NP == 64 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int gi = lane / NP, j = lane % NP;
    const int n = m.n;
    for (int e = threadIdx.x; e < NP * NP; e += 64 * WVS_WPB)
        sA[e] = (e / NP < n && e % NP < n) ? m.A[(int64_t)(e / NP) * n + e % NP] : 0.0;
    __syncthreads(); // (the only one: from here on the wavefronts are on their own)
    const int sgi = (blockIdx.x * WVS_WPB + w) * GP + gi;
    if (sgi >= sg.nseg || sg.len[sgi] <= 0)
        return;
    if constexpr (FIX) {
        if (!flag[sgi])
            return;
    }
    const bool real = j < n;
    const unsigned long long gmask = wgroup_mask<NP>(lane);
    double Acol[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        Acol[i] = sA[i * NP + j];
    const double mu_j = (KIND == EMIT_GAUSS && real) ? m.mu[j] : 0.0;
    const double sg_j = (KIND == EMIT_GAUSS && real) ? m.sigma[j] : 1.0;
    const double rs_j = 1.0 / sg_j; // correctly rounded: IEEE division
    const double cn_j = (KIND == EMIT_GAUSS && real) ? m.cnorm[j] : 0.0; // (0 on padding lanes)
    const double pi_j = real ? m.pi[j] : 0.0;
    auto uni = [](int64_t x) __attribute__((always_inline)) { // (NP == 64: the same in every lane; say so)
        if constexpr (NP == 64)
            return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)x >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x));
        else
            return x;
    };
    const int k = (int)uni(sg.traj[sgi]);
    const int64_t o0 = uni(off[k]), T = uni(off[k + 1]) - o0;
    const int64_t t0 = uni(sg.t0[sgi]), t1 = t0 + uni(sg.len[sgi]);
    const int64_t tw = FIX ? t0 : ((t0 - sg.W > 0) ? t0 - sg.W : 0);
    double olane = 0.0;
    auto emis = [&](int64_t gt) __attribute__((always_inline)) {
        double p;
        if constexpr (KIND == EMIT_GAUSS) {
            // _gaussian.c:18-20 with the arithmetic of k_pobs_lanes (the same bits): the quotient by sigma from
            // the correctly rounded reciprocal, constant and exponential as one block -- 27 instructions where
            // the division and the statement-wise exponential took 60 (round 4 measured the in-step density
            // slower than the emission-matrix pass with those; with these it is 0.7 ms against 1.9 at configs[3])
            double o;
            if constexpr (NP == 64) {
                // (round 6) the observations of 64 steps in one register (lane l: step 64 b + l of this run), one
                // coalesced load per 64 steps, waited for where it is issued: a load per step is waited for in that
                // step with vmcnt(0) -- the counter the stores of the vectors and back-pointers count in, too
                const int rr = (int)(gt - (o0 + tw));
                if ((rr & 63) == 0) {
                    olane = tw + rr + lane < t1 ? static_cast<const double *>(obs_rm)[gt + lane] : 0.0;
                    asm volatile("" : "+v"(olane));
                }
                o = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(olane), rr & 63),
                                     __builtin_amdgcn_readlane(__double2loint(olane), rr & 63));
            } else {
                o = static_cast<const double *>(obs_rm)[gt];
            }
            const double x = o - mu_j;
            const double q0 = x * rs_j;
            const double d = fma(fma(-q0, sg_j, x), rs_j, q0); // == x / sigma
            p = gauss_exp_block(-0.5 * d * d, cn_j);
            if ((__ballot(p != 0.0) & gmask) == 0ull)
                p = real ? 1.0 : 0.0;
        } else if constexpr (KIND == EMIT_DISC) {
            const int sym = static_cast<const int32_t *>(obs_rm)[gt];
            p = real ? m.B[(int64_t)j * m.M + sym] : 0.0;
        } else {
            p = real ? static_cast<const double *>(obs_rm)[gt * n + j] : 0.0;
        }
        return p;
    };
    // (warm-up start; replaced at t = 0.  FIX: the predecessor's vector, t0 > 0 for a flagged segment)
    double v = FIX ? v_entry[(int64_t)sgi * NP + j] : (real ? 1.0 / (double)n : 0.0);
    if constexpr (NP == 64 && WVS_LDS_ROWS > 0)
        xv[w][gi][j] = v; // (LDS copy of v for the upper candidate rows of the next step)
    double p_next = emis(o0 + tw);
    bool met = false; // FIX: the run reproduced a vector of the first pass
    // (the step as a 32-bit count from the first one: 64-bit compares are vector instructions even on scalars)
    const int nst = (int)(t1 - tw), r0 = (int)(t0 - tw), rz = tw == 0 ? 0 : -1;
    const int ph = (int)((o0 + tw) & 63);
    for (int r = 0; r < nst; ++r) {
        const int64_t t = tw + r;
        const double p = p_next;
        // (the LDS copies of v two blocks of eight candidates ahead of their products: one wait per block)
        double2 ypre[2][4];
        auto prefetch = [&](auto qc) __attribute__((always_inline)) {
            constexpr int q8 = decltype(qc)::value;
            if constexpr (NP == 64 && q8 < NP / 8 && q8 / 2 >= NP / 16 - WVS_LDS_ROWS) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ypre[q8 & 1][q] = *reinterpret_cast<const double2 *>(&xv[w][gi][8 * q8 + 2 * q]);
            }
        };
        if constexpr (NP == 64 && WVS_LDS_ROWS == 4) { // (the first block's before the density: nothing else covers them)
            prefetch(std::integral_constant<int, 0>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        if (r + 1 < nst)
            p_next = emis(o0 + t + 1); // independent of the recursion
        double vn;
        if (r == rz) { // t == 0
            vn = p * pi_j; // _hidden.c:232
        } else {
            // Every lane needs every element of v: row copies (permlane swaps) and the DPP form
            // fma(v[16 k + i] broadcast within the row, A[16 k + i][j], 0) -- the product
            // v[i] * A[i][j] of _hidden.c:249, rounded once like the multiplication -- instead of
            // 32 broadcast reads from LDS, which was what bound this kernel with all CUs busy.
            // First-maximum argmax (_hidden.c:186-200) as a select tree per row of 16 states: the
            // later candidate wins only if strictly greater, which is exactly the linear scan's
            // result; rows are folded in index order with the same rule.
            const Rows4 R = rows_of_group<NP>(v);
            double bh = 0.0;
            int bi = 0;
            // USEMAX: without NaNs the winner's value is the maximum -- one v_max_f64 instead of two selects
            auto argmax_rows = [&](auto usemax) __attribute__((always_inline)) {
                constexpr bool USEMAX = decltype(usemax)::value;
#pragma unroll
                for (int k = 0; k < NP / 16; ++k) {
                    double hh[16];
                    int ii[16];
                    if (NP == 64 && k >= NP / 16 - WVS_LDS_ROWS) {
                        // (round 6) the upper rows of candidates read v from its LDS copy -- every lane the same
                        // address, 16 bytes a read: the LDS pipe idles in this kernel -- and multiply with ONE
                        // instruction; the row-broadcast form needs a cleared accumulator per product.  Same
                        // product, rounded once either way.
#pragma unroll
                        for (int q = 0; q < 16; q += 2) {
                            const double2 y = *reinterpret_cast<const double2 *>(&xv[w][gi][16 * k + q]);
                            hh[q] = y.x * Acol[16 * k + q];
                            hh[q + 1] = y.y * Acol[16 * k + q + 1];
                        }
                    } else {
                        auto wk = [&](auto ic) __attribute__((always_inline)) { return Acol[16 * k + decltype(ic)::value]; };
                        asm volatile("s_nop 1"); // (a DPP read needs two wait states after the write of its register)
                        prod8_bcast<0>(hh, R.r[k], wk);
                        prod8_bcast<8>(hh, R.r[k], wk);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        ii[i] = 16 * k + i;
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
                    const bool take = (k == 0) || (hh[0] > bh);
                    bh = take ? hh[0] : bh;
                    bi = take ? ii[0] : bi;
                }
            };
            // (round 6) NaN-free form, eight candidates per assembly block (argmax_oct): the compiler's tree pays a
            // wait-state s_nop between every compare and the select that reads its mask -- 63 issue slots per step
            auto argmax_rows_blocks = [&]() __attribute__((always_inline)) {
                double wh[8];
                int wi[8];
                if constexpr (WVS_LDS_ROWS < 4)
                    prefetch(std::integral_constant<int, 0>{});
                static_for<NP / 8>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int q8 = decltype(qc)::value, k = q8 / 2, o = q8 % 2;
                    prefetch(std::integral_constant<int, q8 + 1>{});
                    if constexpr (NP == 64 && q8 + 1 < NP / 8 && (q8 + 1) / 2 >= NP / 16 - WVS_LDS_ROWS)
                        __builtin_amdgcn_sched_barrier(0); // (or the loads sink to their use, below this block)
                    if constexpr (NP == 64 && k >= NP / 16 - WVS_LDS_ROWS) {
                        double y[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            y[2 * q] = ypre[q8 & 1][q].x, y[2 * q + 1] = ypre[q8 & 1][q].y;
                        argmax_oct_mul<8 * q8>(y, &Acol[8 * q8], wh[q8], wi[q8]);
                    } else {
                        argmax_oct_bcast<8 * q8, 8 * o, o == 0>(R.r[k], &Acol[8 * q8], wh[q8], wi[q8]);
                    }
                });
                if constexpr (NP == 16) {
                    const bool take = wh[1] > wh[0];
                    bh = take ? wh[1] : wh[0];
                    bi = take ? wi[1] : wi[0];
                } else if constexpr (NP == 32) {
                    bh = wh[0], bi = wi[0];
#pragma unroll
                    for (int q = 1; q < 4; ++q) {
                        const bool take = wh[q] > bh;
                        bh = take ? wh[q] : bh;
                        bi = take ? wi[q] : bi;
                    }
                } else {
                    argmax_oct_regs(wh, wi, bi);
                    bh = wh[0];
                }
            };
            if (__ballot(v != v) == 0ull) // (wave-uniform; v is NaN-free, A is, so are the products)
                argmax_rows_blocks();
            else
                argmax_rows(std::false_type{});
            if (real && r >= r0)
                ptr[(o0 + t) * n + j] = (uint8_t)bi;
            // (64 states: v[i^] from the LDS copy of v -- one read instead of two lane permutes and their index)
            double bv;
            if constexpr (NP == 64 && WVS_LDS_ROWS > 0)
                bv = xv[w][gi][bi];
            else
                bv = __shfl(v, bi, NP);
            const double bA = sA[bi * NP + j];
            vn = p * bv * bA; // _hidden.c:253: (p v[i^]) A[i^][j]
        }
        // the normalising sum in ascending order (_hidden.c:256-259), S = fma(vn[i], 1, S) = S + vn[i]
        // rounded once, on the row copies of vn (padded states add exact zeros)
        double S = 0.0;
        {
            const Rows4 Rn = rows_of_group<NP>(vn);
            const double one = 1.0;
            if constexpr (NP == 64) {
                sum64_bcast(S, Rn.r[0], Rn.r[1], Rn.r[2], Rn.r[3], one);
            } else {
#pragma unroll
                for (int k = 0; k < NP / 16; ++k)
                    sum16_bcast(S, Rn.r[k], one);
            }
        }
        v = vn / S;
        if constexpr (NP == 64 && WVS_LDS_ROWS > 0)
            xv[w][gi][j] = v; // (the next step's upper candidate rows read it from here)
        if constexpr (!FIX) {
            if (r == r0 - 1)
                v_entry[(int64_t)sgi * NP + j] = v;
        }
        if (vall && real && r >= r0) // (every vector of the first pass / of a mended stretch, [total][n]: k_vit_margin)
            vall[(o0 + t) * n + j] = v;
        if (((ph + r) & 63) == 63 && r >= r0) {
            double *cp = ckpt + ((o0 + t) >> 6) * NP + j;
            if constexpr (FIX) {
                const double cv = *cp;
                bool same = __double_as_longlong(cv) == __double_as_longlong(v);
                if (mend_tol > 0.0) // (uniform)
                    same = same || (fabs(cv - v) <= mend_tol * cv && (cv == 0.0) == (v == 0.0));
                if ((__ballot(!same) & gmask) == 0ull) {
                    met = true;
                    break;
                }
            }
            *cp = v;
        }
    }
    if (met)
        return;
    if (FIX && notmet && j == 0)
        atomicAdd(notmet, 1u);
    v_exit[(int64_t)sgi * NP + j] = v;
    if (t1 == T) { // the trajectory's final state (_hidden.c:262-267: first maximum)
        xv[w][gi][j] = v;
        if (j == 0) {
            double bm = xv[w][gi][0];
            int bi = 0;
            for (int i = 1; i < n; ++i)
                if (xv[w][gi][i] > bm) {
                    bm = xv[w][gi][i];
                    bi = i;
                }
            last_state[k] = bi;
        }
    }
}

// result[3] = segments whose entry vector is not bit-identical to the predecessor's exit vector; those
// are flagged, and their entry vector becomes the predecessor's (what the fix-up pass starts from).
// result[0] = those of them that are not even equal to `tol` relative in every component, with the same
// zero pattern (k_vit_margin's condition; tol = 0: not counted)
// flag[nseg + s] (tol > 0 only): segment s is one of the latter (the mending round's list)
template <int NP>
__global__ void k_wide_vit_check(const Segs sg, double *v_entry, const double *v_exit, uint8_t *flag,
                                 unsigned int *result, double tol = 0.0)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool differs = false, far = false;
    if (s < sg.nseg && sg.len[s] > 0 && sg.t0[s] != 0) {
        double *x = v_entry + (int64_t)s * NP;
        const double *y = v_exit + (int64_t)(s - 1) * NP;
        for (int j = 0; j < NP; ++j) {
            differs |= __double_as_longlong(x[j]) != __double_as_longlong(y[j]);
            far |= !(fabs(x[j] - y[j]) <= tol * y[j]) || ((x[j] == 0.0) != (y[j] == 0.0));
        }
        if (differs)
            for (int j = 0; j < NP; ++j)
                x[j] = y[j];
    }
    if (s < sg.nseg) {
        flag[s] = differs ? 1 : 0;
        if (tol > 0.0)
            flag[sg.nseg + s] = (differs && far) ? 1 : 0;
    }
    const unsigned long long d = __ballot(differs), f = __ballot(differs && far);
    if ((threadIdx.x & 63) == 0 && d)
        atomicAdd(&result[3], (unsigned int)__popcll(d));
    if ((threadIdx.x & 63) == 0 && f && tol > 0.0)
        atomicAdd(&result[0], (unsigned int)__popcll(f));
}

// =========================================================================================
// k_vit_margin (round 5): the segment-parallel first pass without fix-up rounds.  When every boundary of the
// first pass is equal to `tol` (most are equal to the bit; the rest are the two runs' rounding noise after the
// survivors met, which the max-product recursion never sheds exactly), the vectors of the first pass are the
// serial run's up to delta = (number of boundaries) tol + the roundings of both runs: the step
// v -> normalise(p o max_i v_i A_ij) is non-expansive in Hilbert's projective metric, four roundings per
// component and step are not common to all components.  The back-pointers may still differ from the serial
// run's where a decision was closer than that -- but the PATH only uses one decision per step: if the final
// state and every decision ON the path of the first pass was taken with a relative margin above `margin`
// (>= 16 delta, host), the serial run takes the same decisions there, and by induction from the last step its
// path is this one.  Step t of the path: j = path[t], i^ = path[t - 1], all
// candidates v_{t-1}[i] A[i][j] (_hidden.c:249) against the winner's.  Violations (a close or tied decision,
// a winner that is zero, denormal or not finite) are counted in result[2]; the host then runs the fix-up
// rounds, which need none of this.   vall [total][n]: every vector of the first pass.
// =========================================================================================
constexpr int VM_STEPS = 1024; // steps of a segment per workgroup of k_vit_margin (A^T is staged once per workgroup)
[[maybe_unused]] static __global__ void k_vm_transpose(const double *A, int n, double *At)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * n)
        At[(e % n) * n + e / n] = A[e];
}
// ALDS = false (more than 128 states: A^T does not fit LDS): `A` is A TRANSPOSED in global memory, read per step.
template <typename PT, int NC, bool ALDS = true>
__global__ __launch_bounds__(256) void k_vit_margin(const double *A, int n, const int64_t *off, const Segs sg,
                                                    const double *vall, const PT *path, double margin,
                                                    unsigned int *result)
{
    // workgroup (s, c): steps [VM_STEPS c, VM_STEPS c + VM_STEPS) of segment s, 64 per wavefront and turn; lane l holds
    // the path at "its" step and the one before (one coalesced read each), the rows of v sixteen steps at a time
    extern __shared__ double vm_sAT[]; // [j][i] = A[i][j], rows PA apart
    // (an odd pitch: the transposing writes of the staging -- consecutive lanes, consecutive j -- fall on different
    // banks; with pitch n = 64 all 64 lanes of a wavefront hit one bank pair, 4 us of the LDS pipe per workgroup)
    const int PA = ALDS ? (n | 1) : n;
    const int s = blockIdx.x, wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (s >= sg.nseg || sg.len[s] <= 0 || (int64_t)blockIdx.y * VM_STEPS >= sg.len[s])
        return;
    if constexpr (ALDS) {
        for (int e = threadIdx.x; e < n * n; e += 256)
            vm_sAT[(e % n) * PA + e / n] = A[e];
        __syncthreads();
    }
    const int k = sg.traj[s];
    const int64_t o0 = off[k], T = off[k + 1] - o0;
    const int64_t t1 = sg.t0[s] + sg.len[s];
    bool bad = false;
    for (int turn = 0; turn < VM_STEPS / 256; ++turn) {
    const int64_t tb = sg.t0[s] + (int64_t)blockIdx.y * VM_STEPS + 256 * turn + 64 * wid;
    if (tb >= t1)
        break;
    const int cnt = (int)(t1 - tb < 64 ? t1 - tb : 64);
    const int pj = lane < cnt ? (int)path[o0 + tb + lane] : 0;
    const int pi = (lane < cnt && tb + lane >= 1) ? (int)path[o0 + tb + lane - 1] : 0;
    constexpr int U = 16;
    for (int q0 = 0; q0 < cnt; q0 += U) {
        double vr[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = tb + q0 + u;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                vr[u][c] = (q0 + u < cnt && t >= 1 && lane + 64 * c < n) ? vall[(o0 + t - 1) * n + lane + 64 * c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t t = tb + q0 + u;
            if (q0 + u < cnt && t >= 1) { // (uniform)
                const int j = __shfl(pj, q0 + u, 64), ih = __shfl(pi, q0 + u, 64);
                const double *col = (ALDS ? vm_sAT : A) + (int64_t)j * PA;
                double h[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    h[c] = lane + 64 * c < n ? vr[u][c] * col[lane + 64 * c] : 0.0; // _hidden.c:249
                double hb = __shfl(h[0], ih & 63, 64);
#pragma unroll
                for (int c = 1; c < NC; ++c) {
                    const double hbc = __shfl(h[c], ih & 63, 64);
                    hb = (ih >> 6) == c ? hbc : hb;
                }
                const double lim = hb - margin * hb;
                bad |= !(hb >= 0x1p-960) || !(hb < 0x1p1000);
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    bad |= lane + 64 * c < n && lane + 64 * c != ih && !(h[c] < lim);
            }
        }
    }
    if (t1 == T && tb + cnt == t1) { // the final state (_hidden.c:262-267), by the wavefront that holds the last step
        const int j = __shfl(pj, cnt - 1, 64);
        const double *vp = vall + (o0 + T - 1) * n;
        const double vb = vp[j], lim = vb - margin * vb;
        bad |= !(vb >= 0x1p-960) || !(vb < 0x1p1000);
        for (int i = lane; i < n; i += 64)
            bad |= i != j && !(vp[i] < lim);
    }
    } // (turns)
    const unsigned long long b = __ballot(bad);
    if (lane == 0 && b)
        atomicAdd(&result[2], 1u);
}

// ---- instruction-count diet of the chunked Viterbi step (round 3) ------------------------------
// The chunked Viterbi kernel is bound by the length of its instruction stream (four wavefronts per
// SIMD at ~0.9 of the issue ceiling), so what follows removes instructions without touching a
// single rounding:
//  * exp_nonpos() as ONE assembly block: the same range reduction and the same twelve fused
//    multiply-adds (hence the same bits), coefficients in scalar registers, the constant of the
//    density multiplied in the block.  As separate statements the compiler protects every inline
//    v_fma_f64 against its successor with an s_nop (its hazard model does not look inside assembly),
//    eleven wasted issue slots per density; dependent v_fma_f64 need no wait state.
//  * the first-maximum argmax over eight products as one block: compares into scalar pairs,
//    v_max_f64 for the values, v_cndmask for the indices, ordered so that every select finds its
//    mask two wait states old (the gfx950 rule for a VALU read of a VALU-written SGPR) -- 21
//    instructions where the compiler's select tree took 26 plus five s_nop.  take = b > a, value
//    max(a, b): identical to `take ? b : a` for the non-negative finite products of this recursion.
//  * the division by sigma of _gaussian.c:18 as q0 = x r, d = q0 + r (x - q0 sigma) with
//    r = RN(1 / sigma): three instructions instead of the thirteen of an IEEE division by a
//    run-time value, and the correctly rounded quotient all the same (Markstein's theorem for a
//    correctly rounded reciprocal; checked here against x / sigma on 1.5e9 random and
//    special-mantissa pairs without a mismatch).  An infinite observation gives NaN instead of inf,
//    which the clamp of the exponent turns into the same exact zero.
__device__ __forceinline__ double gauss_exp_block(double x, double cn)
{
    x = fmax(x, -750.0);
    const double k = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = fma(k, -0x1.62e42fefa39efp-1, x);
    r = fma(k, -0x1.abc9e3b39803fp-56, r);
    const int ki = (int)k;
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
        "v_fma_f64 %0, %0, %1, 1.0\n\t"
        "v_ldexp_f64 %0, %0, %12\n\t"
        "v_mul_f64 %0, %13, %0"
        : "=&v"(q)
        : "v"(r), "v"(0x1.ad7e38e167506p-26), "s"(0x1.28ae7908135d8p-22), "s"(0x1.71df27c33abefp-19),
          "s"(0x1.a01998fd42e01p-16), "s"(0x1.a01a012882c92p-13), "s"(0x1.6c16c184889e3p-10),
          "s"(0x1.111111112836cp-7), "s"(0x1.55555555506eap-5), "s"(0x1.55555555554f7p-3),
          "s"(0x1.000000000000ap-1), "v"(ki), "v"(cn));
    return q;
}

// index of the FIRST maximum of h[0..7] (_hidden.c:186-200: `if (h > best)`), and the maximum when
// WANT_MAX.  h is destroyed.
template <bool WANT_MAX>
__device__ __forceinline__ int argmax8_block(double (&h)[8], double &mx)
{
    int i0, i1, i2, i3;
    unsigned long long s0, s1, s2, s3;
    if constexpr (WANT_MAX) {
        asm("v_cmp_gt_f64 %[s0], %[h1], %[h0]\n\t"
            "v_cmp_gt_f64 %[s1], %[h3], %[h2]\n\t"
            "v_cmp_gt_f64 %[s2], %[h5], %[h4]\n\t"
            "v_cmp_gt_f64 %[s3], %[h7], %[h6]\n\t"
            "v_max_f64 %[h0], %[h0], %[h1]\n\t"
            "v_max_f64 %[h2], %[h2], %[h3]\n\t"
            "v_max_f64 %[h4], %[h4], %[h5]\n\t"
            "v_max_f64 %[h6], %[h6], %[h7]\n\t"
            "v_cndmask_b32 %[i0], 0, 1, %[s0]\n\t"
            "v_cndmask_b32 %[i1], 2, 3, %[s1]\n\t"
            "v_cndmask_b32 %[i2], 4, 5, %[s2]\n\t"
            "v_cndmask_b32 %[i3], 6, 7, %[s3]\n\t"
            "v_cmp_gt_f64 %[s0], %[h2], %[h0]\n\t"
            "v_cmp_gt_f64 %[s1], %[h6], %[h4]\n\t"
            "v_max_f64 %[h0], %[h0], %[h2]\n\t"
            "v_max_f64 %[h4], %[h4], %[h6]\n\t"
            "v_cndmask_b32 %[i0], %[i0], %[i1], %[s0]\n\t"
            "v_cndmask_b32 %[i2], %[i2], %[i3], %[s1]\n\t"
            "v_cmp_gt_f64 %[s0], %[h4], %[h0]\n\t"
            "v_max_f64 %[h0], %[h0], %[h4]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %[i0], %[i0], %[i2], %[s0]"
            : [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3), [s0] "=&s"(s0),
              [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [h0] "+v"(h[0]), [h2] "+v"(h[2]),
              [h4] "+v"(h[4]), [h6] "+v"(h[6])
            : [h1] "v"(h[1]), [h3] "v"(h[3]), [h5] "v"(h[5]), [h7] "v"(h[7]));
        mx = h[0];
    } else {
        asm("v_cmp_gt_f64 %[s0], %[h1], %[h0]\n\t"
            "v_cmp_gt_f64 %[s1], %[h3], %[h2]\n\t"
            "v_cmp_gt_f64 %[s2], %[h5], %[h4]\n\t"
            "v_cmp_gt_f64 %[s3], %[h7], %[h6]\n\t"
            "v_max_f64 %[h0], %[h0], %[h1]\n\t"
            "v_max_f64 %[h2], %[h2], %[h3]\n\t"
            "v_max_f64 %[h4], %[h4], %[h5]\n\t"
            "v_max_f64 %[h6], %[h6], %[h7]\n\t"
            "v_cndmask_b32 %[i0], 0, 1, %[s0]\n\t"
            "v_cndmask_b32 %[i1], 2, 3, %[s1]\n\t"
            "v_cndmask_b32 %[i2], 4, 5, %[s2]\n\t"
            "v_cndmask_b32 %[i3], 6, 7, %[s3]\n\t"
            "v_cmp_gt_f64 %[s0], %[h2], %[h0]\n\t"
            "v_cmp_gt_f64 %[s1], %[h6], %[h4]\n\t"
            "v_max_f64 %[h0], %[h0], %[h2]\n\t"
            "v_max_f64 %[h4], %[h4], %[h6]\n\t"
            "v_cndmask_b32 %[i0], %[i0], %[i1], %[s0]\n\t"
            "v_cndmask_b32 %[i2], %[i2], %[i3], %[s1]\n\t"
            "v_cmp_gt_f64 %[s0], %[h4], %[h0]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %[i0], %[i0], %[i2], %[s0]"
            : [i0] "=&v"(i0), [i1] "=&v"(i1), [i2] "=&v"(i2), [i3] "=&v"(i3), [s0] "=&s"(s0),
              [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [h0] "+v"(h[0]), [h2] "+v"(h[2]),
              [h4] "+v"(h[4]), [h6] "+v"(h[6])
            : [h1] "v"(h[1]), [h3] "v"(h[3]), [h5] "v"(h[5]), [h7] "v"(h[7]));
        mx = 0.0;
    }
    return i0;
}

// the same for four products (models of up to four states: 16 chunks per wavefront instead of 8)
template <bool WANT_MAX>
__device__ __forceinline__ int argmax4_block(double (&h)[4], double &mx)
{
    int i0, i1;
    unsigned long long s0, s1;
    if constexpr (WANT_MAX) {
        asm("v_cmp_gt_f64 %[s0], %[h1], %[h0]\n\t"
            "v_cmp_gt_f64 %[s1], %[h3], %[h2]\n\t"
            "v_max_f64 %[h0], %[h0], %[h1]\n\t"
            "v_max_f64 %[h2], %[h2], %[h3]\n\t"
            "v_cndmask_b32 %[i0], 0, 1, %[s0]\n\t"
            "v_cndmask_b32 %[i1], 2, 3, %[s1]\n\t"
            "v_cmp_gt_f64 %[s0], %[h2], %[h0]\n\t"
            "v_max_f64 %[h0], %[h0], %[h2]\n\t"
            "s_nop 0\n\t"
            "v_cndmask_b32 %[i0], %[i0], %[i1], %[s0]"
            : [i0] "=&v"(i0), [i1] "=&v"(i1), [s0] "=&s"(s0), [s1] "=&s"(s1), [h0] "+v"(h[0]),
              [h2] "+v"(h[2])
            : [h1] "v"(h[1]), [h3] "v"(h[3]));
        mx = h[0];
    } else {
        asm("v_cmp_gt_f64 %[s0], %[h1], %[h0]\n\t"
            "v_cmp_gt_f64 %[s1], %[h3], %[h2]\n\t"
            "v_max_f64 %[h0], %[h0], %[h1]\n\t"
            "v_max_f64 %[h2], %[h2], %[h3]\n\t"
            "v_cndmask_b32 %[i0], 0, 1, %[s0]\n\t"
            "v_cndmask_b32 %[i1], 2, 3, %[s1]\n\t"
            "v_cmp_gt_f64 %[s0], %[h2], %[h0]\n\t"
            "s_nop 1\n\t"
            "v_cndmask_b32 %[i0], %[i0], %[i1], %[s0]"
            : [i0] "=&v"(i0), [i1] "=&v"(i1), [s0] "=&s"(s0), [s1] "=&s"(s1), [h0] "+v"(h[0]),
              [h2] "+v"(h[2])
            : [h1] "v"(h[1]), [h3] "v"(h[3]));
        mx = 0.0;
    }
    return i0;
}

template <int NP, bool WANT_MAX>
__device__ __forceinline__ int argmax_block(double (&h)[NP], double &mx)
{
    static_assert(NP == 4 || NP == 8, "argmax blocks for 4 and 8 products");
    if constexpr (NP == 8)
        return argmax8_block<WANT_MAX>(h, mx);
    else
        return argmax4_block<WANT_MAX>(h, mx);
}

// =========================================================================================
// k_viterbi_chunks: the same order-faithful recursion, parallel over the time chunks of the
// E-step plan, and still bit-identical to the serial run -- verified, not assumed:
//   * every chunk starts W steps early from a uniform vector (or exactly from pi where that
//     reaches the start of the trajectory); the max-product recursion forgets its start like the
//     filter does, and it is non-expansive in Hilbert's projective metric, so if at every chunk
//     boundary the vector a chunk assumed agrees with the one its predecessor computed to
//     tol (k_spec_check, componentwise relative), every vector of the run is within
//     (#chunks of the trajectory) x 2 tol of the serial run's;
//   * a back-pointer equals the serial run's if it is decided by more than that: the kernel
//     counts decisions whose runner-up is within `margin` (relative) of the winner -- exact
//     ties, the all-zero case and NaNs included.
// If no boundary is out of tolerance and no decision is that close, the back-pointers are the
// serial ones.  In practice the survivors coalesce and the boundary vectors come out
// BIT-IDENTICAL (k_viterbi_check); then the run is the serial run outright and close decisions
// are irrelevant.  Otherwise the host runs k_wide_viterbi_fwd.  Emission rows: k_pobs_all.
// =========================================================================================
// KIND: EMIT_EXPL -- `src` is the (total, n) emission matrix (k_pobs_lanes / k_pobs_all, or the
// caller's pobs); EMIT_DISC -- `src` is the int32 symbol stream and the probability is read from
// B directly (discrete.py:150-153: pobs = B[:, obs].T is a gather, there is nothing to precompute
// -- for BASELINE configs[2] the materialised matrix would be 65 GB written and read again).
// MARGIN = false: the close-decision count is left out (a sixth of the step's instructions).  The
// host runs this form first -- when every boundary vector comes out bit-identical to the
// predecessor's, which is the normal case, the run IS the serial run and the count is not looked at
// -- and repeats with MARGIN = true only otherwise.
// The steps run in blocks of VPF.  A block in which every chunk of the wavefront is in its warm-up,
// or every chunk is past it and at least two blocks from its end, takes a form without per-step
// conditions (no back-pointer store / an unconditional one, no clamp of the prefetch address, the
// outlier rule looked at only when a normalising sum is exactly zero); the first block of a
// trajectory, the block in which a warm-up ends and the last two blocks take the general form.
template <int NP, int KIND = EMIT_EXPL, bool MARGIN = true>
__global__ __launch_bounds__(64) void k_viterbi_chunks(const WideModel m, const Chunks ch, int G,
                                                       const int64_t *toff, const void *src,
                                                       int W, double margin, uint8_t *ptr,
                                                       int32_t *last_state, double *v_entry,
                                                       double *v_exit, unsigned int *flags,
                                                       int b_in_lds)
{
    // EMIT_GAUSS: `src` is the observation stream, the density of the lane's state is evaluated in
    // the step (same arithmetic as k_pobs_lanes, outlier rule by a ballot over the chunk's lanes)
    [[maybe_unused]] const double *pobs = static_cast<const double *>(src);
    [[maybe_unused]] const int32_t *syms = static_cast<const int32_t *>(src);
    constexpr int GP = 64 / NP;
    static_assert(NP == 4 || NP == 8, "argmax_block");
    __shared__ __attribute__((aligned(16))) double xv[GP][NP];
    __shared__ __attribute__((aligned(16))) double xn[GP][NP];
    // A for the winner's factor: the argmax carries only (product, index); v[i^] and
    // A[i^][j] are looked up afterwards (two LDS reads instead of four more registers per select)
    __shared__ double sA[NP * NP];
    extern __shared__ double sBdyn[]; // discrete: B [n][M] when the launch provides the room
    const int lane = threadIdx.x;
    for (int e = lane; e < NP * NP; e += 64)
        sA[e] = (e / NP < m.n && e % NP < m.n) ? m.A[(int64_t)(e / NP) * m.n + e % NP] : 0.0;
    [[maybe_unused]] const double *sBd = nullptr;
    if constexpr (KIND == EMIT_DISC) {
        if (b_in_lds) {
            for (int e = lane; e < m.n * m.M; e += 64)
                sBdyn[e] = m.B[e];
            sBd = sBdyn;
        }
    }
    __syncthreads();
    const int gi = lane / NP, j = lane % NP;
    const int64_t g = (int64_t)blockIdx.x * GP + gi;
    const int len = g < G ? ch.len[g] : 0;
    bool low = false;
    if (len > 0) {
        const int n = m.n;
        const bool real = j < n;
        const bool allreal = n == NP; // (uniform) every lane stores a back-pointer
        double Acol[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i)
            Acol[i] = (real && i < n) ? m.A[(int64_t)i * n + j] : 0.0;
        const double pi_j = real ? m.pi[j] : 0.0;
        const int64_t t0 = ch.t0[g], goff = ch.goff[g];
        const int k = ch.traj[g];
        const int nw = (int)(t0 < (int64_t)W ? t0 : (int64_t)W);
        const bool exact = (int64_t)nw == t0; // the warm-up reaches the start of the trajectory
        const int64_t gs = goff - nw;
        const int steps = nw + len;
        double v = real ? 1.0 / (double)n : 0.0;
        // what a step reads from memory is requested VPF steps ahead (it does not depend on the
        // recursion): the emission probability itself, or -- discrete -- the symbol, whose
        // probability then comes from B in LDS (or, for alphabets too large for it, from L2)
        constexpr int VPF = 4;
        double pring[VPF];
        int sring[VPF];
        auto request = [&](int q, int sidx) {
            const int64_t t = gs + (sidx < steps ? sidx : steps - 1);
            if constexpr (KIND == EMIT_DISC)
                sring[q] = syms[t];
            else if constexpr (KIND == EMIT_GAUSS)
                pring[q] = pobs[t]; // the observation
            else
                pring[q] = real ? pobs[t * n + j] : 0.0;
        };
        // the same without the clamp, for blocks at least VPF steps from the end of the chunk
        auto request_inside = [&](int q, int sidx) {
            const int64_t t = gs + sidx;
            if constexpr (KIND == EMIT_DISC)
                sring[q] = syms[t];
            else if constexpr (KIND == EMIT_GAUSS)
                pring[q] = pobs[t];
            else
                pring[q] = real ? pobs[t * n + j] : 0.0;
        };
        [[maybe_unused]] const double mu_j = (KIND == EMIT_GAUSS && real) ? m.mu[j] : 0.0;
        [[maybe_unused]] const double sg_j = (KIND == EMIT_GAUSS && real) ? m.sigma[j] : 1.0;
        [[maybe_unused]] const double rs_j = 1.0 / sg_j; // correctly rounded: IEEE division
        [[maybe_unused]] const double cn_j = (KIND == EMIT_GAUSS && real) ? m.cnorm[j] : 0.0;
        [[maybe_unused]] const unsigned long long grp =
            (NP == 64 ? ~0ull : ((1ull << NP) - 1)) << (lane / NP * NP);
        // emission probability of the lane's state from ring slot q (outlier rule not applied)
        auto emission = [&](int q) -> double {
            if constexpr (KIND == EMIT_DISC) {
                const int sym = sring[q];
                return !real ? 0.0 : (sBd ? sBd[j * m.M + sym] : m.B[(int64_t)j * m.M + sym]);
            } else if constexpr (KIND == EMIT_GAUSS) {
                const double x = pring[q] - mu_j;
                const double q0 = x * rs_j;
                const double d = fma(fma(-q0, sg_j, x), rs_j, q0); // == x / sigma, _gaussian.c:18
                return gauss_exp_block(-0.5 * d * d, cn_j); // cn_j = 0 on padding lanes
            } else {
                return pring[q];
            }
        };
        auto ordered_sum = [&]() -> double {
            double xs[NP];
#pragma unroll
            for (int i = 0; i < NP; i += 2) {
                const double2 x = *reinterpret_cast<const double2 *>(&xn[gi][i]);
                xs[i] = x.x;
                xs[i + 1] = x.y;
            }
            double S = 0.0 + xs[0]; // (the reference's sum starts from zero: -0 becomes +0)
#pragma unroll
            for (int i = 1; i < NP; ++i)
                S += xs[i]; // ascending order; padded states add exact zeros
            return S;
        };
        // one step without conditions on the step number.  MAIN: back-pointer stored at pq.
        auto fast_step = [&](auto main_tag, auto allreal_tag, double p, uint8_t *pq) {
            constexpr bool MAIN = decltype(main_tag)::value;
            constexpr bool ALLREAL = decltype(allreal_tag)::value;
            xv[gi][j] = v;
            double hh[NP];
            [[maybe_unused]] double h0[NP];
#pragma unroll
            for (int i = 0; i < NP; i += 2) {
                const double2 x = *reinterpret_cast<const double2 *>(&xv[gi][i]);
                hh[i] = x.x * Acol[i]; // _hidden.c:249
                hh[i + 1] = x.y * Acol[i + 1];
            }
            if constexpr (MAIN && MARGIN) {
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    h0[i] = hh[i];
            }
            double mx;
            const int ib = argmax_block<NP, MAIN && MARGIN>(hh, mx);
            if constexpr (MAIN) {
                if (ALLREAL || real)
                    *pq = (uint8_t)ib;
                if constexpr (MARGIN) {
                    const double thr = mx - margin * mx;
                    int cnt = 0;
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        cnt += (h0[i] >= thr) ? 1 : 0;
                    low |= real && cnt != 1;
                }
            }
            const double bv = xv[gi][ib], bA = sA[ib * NP + j];
            double vn = p * bv * bA; // _hidden.c:253: (p v[i^]) A[i^][j]
            xn[gi][j] = vn;
            double S = ordered_sum();
            if constexpr (KIND == EMIT_GAUSS) {
                // outputmodel.py:126-130 (an all-zero emission row counts as all ones): such a row
                // makes every product of its chunk zero, so it is looked for only then
                if (__any(S == 0.0)) {
                    if ((__ballot(p != 0.0) & grp) == 0ull)
                        p = real ? 1.0 : 0.0;
                    vn = p * bv * bA;
                    xn[gi][j] = vn;
                    S = ordered_sum();
                }
            }
            v = vn / S;
        };
#pragma unroll
        for (int q = 0; q < VPF; ++q) {
            pring[q] = 0.0;
            sring[q] = 0;
            request(q, q);
        }
        for (int sb = 0; sb < steps; sb += VPF) {
            // (per wavefront: the chunks of a wavefront differ in nw at the start of a trajectory
            // and by one step in length)
            const bool inside = sb + 2 * VPF <= steps && (sb > 0 || !exact);
            if (__all(inside && sb + VPF <= nw)) {
#pragma unroll
                for (int qq = 0; qq < VPF; ++qq) {
                    const double p = emission(qq);
                    request_inside(qq, sb + qq + VPF);
                    fast_step(std::false_type{}, std::false_type{}, p, nullptr);
                }
                if (sb + VPF == nw)
                    v_entry[g * NP + j] = v; // the vector this chunk starts from
                continue;
            }
            if (__all(inside && sb >= nw)) {
                uint8_t *pq = ptr + (gs + sb) * n + j;
                if (allreal) {
#pragma unroll
                    for (int qq = 0; qq < VPF; ++qq) {
                        const double p = emission(qq);
                        request_inside(qq, sb + qq + VPF);
                        fast_step(std::true_type{}, std::true_type{}, p, pq + qq * NP);
                    }
                } else {
#pragma unroll
                    for (int qq = 0; qq < VPF; ++qq) {
                        const double p = emission(qq);
                        request_inside(qq, sb + qq + VPF);
                        fast_step(std::true_type{}, std::false_type{}, p, pq + qq * n);
                    }
                }
                continue;
            }
#pragma unroll
        for (int qq = 0; qq < VPF; ++qq) {
            const int s = sb + qq;
            if (s >= steps)
                break;
            double p = emission(qq);
            if constexpr (KIND == EMIT_GAUSS) {
                if ((__ballot(p != 0.0) & grp) == 0ull)
                    p = real ? 1.0 : 0.0; // outputmodel.py:126-130
            }
            request(qq, s + VPF);
            double vn;
            if (exact && s == 0) {
                vn = p * pi_j; // _hidden.c:232
            } else {
                xv[gi][j] = v;
                // first-maximum argmax (_hidden.c:186-200) as a select tree, see
                // k_wide_viterbi_fwd
                double vv[NP], hh[NP], h0[NP];
                int ii[NP];
#pragma unroll
                for (int i = 0; i < NP; i += 2) {
                    const double2 x = *reinterpret_cast<const double2 *>(&xv[gi][i]);
                    vv[i] = x.x;
                    vv[i + 1] = x.y;
                }
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    hh[i] = vv[i] * Acol[i]; // _hidden.c:249
                    h0[i] = hh[i];
                    ii[i] = i;
                }
#define BHMM_ARGMAX_LEVEL(Wd)                                          \
    if constexpr (NP > Wd) {                                           \
        _Pragma("unroll") for (int i = 0; i + Wd < NP; i += 2 * Wd)    \
        {                                                              \
            const bool take = hh[i + Wd] > hh[i];                      \
            hh[i] = take ? hh[i + Wd] : hh[i];                         \
            ii[i] = take ? ii[i + Wd] : ii[i];                         \
        }                                                              \
    }
                BHMM_ARGMAX_LEVEL(1)
                BHMM_ARGMAX_LEVEL(2)
                BHMM_ARGMAX_LEVEL(4)
                BHMM_ARGMAX_LEVEL(8)
#undef BHMM_ARGMAX_LEVEL
                if (s >= nw) {
                    if (real)
                        ptr[(gs + s) * n + j] = (uint8_t)ii[0];
                    if constexpr (MARGIN) {
                        // candidates within `margin` of the winner: must be the winner alone
                        const double thr = hh[0] - margin * hh[0];
                        int cnt = 0;
#pragma unroll
                        for (int i = 0; i < NP; ++i)
                            cnt += (h0[i] >= thr) ? 1 : 0;
                        low |= real && cnt != 1;
                    }
                }
                vn = p * xv[gi][ii[0]] * sA[ii[0] * NP + j]; // _hidden.c:253: (p v[i^]) A[i^][j]
            }
            xn[gi][j] = vn;
            const double S = ordered_sum();
            v = vn / S;
            if (s == nw - 1)
                v_entry[g * NP + j] = v; // the vector this chunk starts from
        }
        }
        v_exit[g * NP + j] = v;
        if (t0 + len == toff[k + 1] - toff[k]) { // last chunk of the trajectory: final state
            xv[gi][j] = v;
            if (j == 0) {
                double bm = xv[gi][0];
                int bi = 0;
                for (int i = 1; i < n; ++i)
                    if (xv[gi][i] > bm) {
                        bm = xv[gi][i];
                        bi = i;
                    }
                int cnt = 0;
                for (int i = 0; i < n; ++i)
                    cnt += (xv[gi][i] >= bm - margin * bm) ? 1 : 0;
                low |= cnt != 1;
                last_state[k] = bi;
            }
        }
    }
    const unsigned long long lows = __ballot(low);
    if (lane == 0 && lows)
        atomicAdd(&flags[2], (unsigned int)__popcll(lows));
}

// boundary check of k_viterbi_chunks: result[0] boundaries out of tolerance, [1] largest
// deviation (float bits), [3] boundaries whose two vectors are not bit-identical.  When all are
// bit-identical the chunked run IS the serial run (by induction from the exact first chunk), and
// close decisions do not matter.
template <int NP>
__global__ void k_viterbi_check(const Chunks ch, int G, const double *v_entry, const double *v_exit,
                                double tol, unsigned int *result)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double dev = spec_dev_one<NP>(ch, G, g, v_entry, v_exit, nullptr, nullptr);
    spec_commit(dev, tol, result);
    bool differs = false;
    if (g < G && ch.len[g] > 0 && ch.t0[g] != 0) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
            differs |= __double_as_longlong(v_entry[g * NP + j]) !=
                       __double_as_longlong(v_exit[(g - 1) * NP + j]);
    }
    const unsigned long long d = __ballot(differs);
    if ((threadIdx.x & 63) == 0 && d)
        atomicAdd(&result[3], (unsigned int)__popcll(d));
}

// Back-trace of the chunked run, parallel over chunks like the path sampler: k_vit_walk<false>
// gives every chunk its map "state at my last step -> state at the last step of the previous
// chunk" (8 candidates walked by 8 lanes, packed as nibbles), k_smp_stitch chains the maps per
// trajectory from its final state, k_vit_walk<true> re-walks every chunk from its known last state
// and writes the path (_hidden.c:269-272).  Back-pointer rows are staged through LDS 64 steps at
// a time; a dependent chain of global byte loads would be latency bound.
// The survivors of a Viterbi recursion coalesce: a few dozen steps below the end of a chunk all
// eight candidates have walked into the same state, and from there down the path does not depend
// on what follows the chunk.  The map pass notices that at a tile boundary (step coal[g]), writes
// the path below it itself, and the second pass walks only the steps above -- one walk over the
// back-pointers instead of two.
template <int NP, bool APPLY, typename PT = int32_t>
__global__ __launch_bounds__(64) void k_vit_walk(const Chunks ch, int G, int n, const uint8_t *ptr,
                                                 const int32_t *end_state, uint32_t *maps,
                                                 PT *path, int32_t *coal)
{
    constexpr int GP = 64 / NP;
    __shared__ __attribute__((aligned(16))) uint8_t tile[GP][64 * NP];
    __shared__ int32_t outp[GP][64];
    const int lane = threadIdx.x;
    const int gi = lane / NP, e = lane % NP;
    const int64_t g = (int64_t)blockIdx.x * GP + gi;
    const int len = g < G ? ch.len[g] : 0;
    unsigned int packed = 0;
    if (len > 0) {
        const int64_t goff = ch.goff[g];
        const bool first = ch.t0[g] == 0;
        // row s of the chunk leads from the state at step s to the state at step s-1; row 0 of a
        // chunk that does not start its trajectory leads into the previous chunk (maps only)
        const int s_lo = (APPLY || first) ? 1 : 0;
        int cur = APPLY ? end_state[g] : e;
        // APPLY: the path below step `stop` is there already.  Maps: `agreed` once the candidates
        // have met; -1 = they never did (then the second pass walks the whole chunk)
        const int stop = APPLY ? coal[g] : -1;
        bool agreed = false;
        int met = -1;
        if (APPLY && e == 0)
            path[goff + len - 1] = (PT)cur;
        for (int hi = len - 1; hi >= s_lo && hi > stop; hi -= 64) {
            const int lo = hi - 63 > s_lo ? hi - 63 : s_lo;
            const int cnt = hi - lo + 1;
            const int64_t base = (goff + lo) * n;
            if ((n & 3) == 0) {
                // rows of 4 or 8 bytes: the tile as 32-bit words (a quarter of the loads and stores)
                const uint32_t *src = reinterpret_cast<const uint32_t *>(ptr + base);
                uint32_t *dst = reinterpret_cast<uint32_t *>(&tile[gi][0]);
                for (int b = e; b < cnt * n / 4; b += NP)
                    dst[b] = src[b];
            } else {
                for (int b = e; b < cnt * n; b += NP)
                    tile[gi][b] = ptr[base + b];
            }
            const bool writing = APPLY || agreed;
            for (int q = cnt - 1; q >= 0; --q) {
                cur = tile[gi][q * n + cur];
                if (writing && e == 0)
                    outp[gi][q] = cur; // path at step lo + q - 1
            }
            if (writing) {
                for (int q = e; q < cnt; q += NP)
                    if (lo - 1 + q >= 0) // (row 0 of a later chunk points into its predecessor)
                        path[goff + lo - 1 + q] = (PT)outp[gi][q];
            } else {
                const int c0 = __shfl(cur, 0, NP);
                const unsigned long long same = __ballot(cur == c0);
                if (((same >> (gi * NP)) & ((1ull << NP) - 1)) == ((1ull << NP) - 1)) {
                    agreed = true;
                    met = lo - 1; // every candidate is in the same state at this step
                }
            }
        }
        if (!APPLY && e == 0)
            coal[g] = met;
        packed = (unsigned int)cur << (4 * e);
    }
    if constexpr (!APPLY) {
#pragma unroll
        for (int h = 1; h < NP; h <<= 1)
            packed |= __shfl_xor(packed, h, 64);
        if (e == 0 && len > 0)
            maps[g] = packed;
    }
}

// Emission probabilities of all steps, row-major (total, n), fully parallel (one thread per
// step): takes exp / division / table gathers out of the serial Viterbi recursion, whose time
// is set by the length of its per-step instruction stream.  Same arithmetic as the fused form
// (_gaussian.c:18-20 with true division, outputmodel.py:126-130).
template <int KIND>
__global__ void k_pobs_all(const WideModel m, const void *obs_rm, int64_t total, double *pobs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total)
        return;
    const int n = m.n;
    if constexpr (KIND == EMIT_GAUSS) {
        const double o = static_cast<const double *>(obs_rm)[t];
        double s = 0.0;
        for (int j = 0; j < n; ++j) {
            const double d = (o - m.mu[j]) / m.sigma[j];
            const double p = m.cnorm[j] * exp_nonpos(-0.5 * d * d);
            pobs[t * n + j] = p;
            s = (p != 0.0) ? 1.0 : s;
        }
        if (s == 0.0)
            for (int j = 0; j < n; ++j)
                pobs[t * n + j] = 1.0;
    } else {
        const int sym = static_cast<const int32_t *>(obs_rm)[t];
        for (int j = 0; j < n; ++j)
            pobs[t * n + j] = m.B[(int64_t)j * m.M + sym];
    }
}

// The same rows with one thread per ELEMENT (step, state): consecutive lanes write consecutive
// doubles, so every store instruction of a wavefront is one contiguous 512-byte segment (the
// thread-per-step form above writes 64 lines of which it fills an eighth each, and runs at a
// quarter of the write rate).  NL = n must be a power of two: the outlier rule (all n
// probabilities zero, outputmodel.py:126-130) is a ballot over the aligned group of n lanes.
constexpr int POBS_LANES_R = 8; // rows of 256 / NL steps per workgroup
template <int KIND, int NL>
__global__ __launch_bounds__(256) void k_pobs_lanes(const WideModel m, const void *obs_rm,
                                                    int64_t total, double *pobs)
{
    // thread -> state j = threadIdx.x % NL for POBS_LANES_R groups of 256 / NL consecutive steps (consecutive
    // threads write consecutive elements).  Gaussian: the division by sigma of _gaussian.c:18 as the
    // correctly rounded quotient from the correctly rounded reciprocal and the exponential as one block
    // (gauss_exp_block above: the same bits as (o - mu) / sigma and cnorm * exp_nonpos(..), a third of the
    // instructions -- at 64 states this pass was 2.8 ms of a 23 ms Viterbi call on configs[3])
    const int j = threadIdx.x % NL;
    [[maybe_unused]] double mu_j = 0.0, sg_j = 1.0, rs_j = 1.0, cn_j = 0.0;
    if constexpr (KIND == EMIT_GAUSS) {
        mu_j = m.mu[j];
        sg_j = m.sigma[j];
        rs_j = 1.0 / sg_j; // correctly rounded: IEEE division
        cn_j = m.cnorm[j];
    }
    const int lane = threadIdx.x & 63;
    [[maybe_unused]] const unsigned long long grp = (NL == 64 ? ~0ull : ((1ull << NL) - 1)) << (lane / NL * NL);
    const int64_t e0 = (int64_t)blockIdx.x * (256 * POBS_LANES_R) + threadIdx.x;
#pragma unroll
    for (int r = 0; r < POBS_LANES_R; ++r) {
        const int64_t e = e0 + (int64_t)r * 256;
        const bool in = e < total * NL; // (uniform over a group of NL lanes: total * NL is a multiple of NL)
        const int64_t t = in ? e / NL : 0;
        if constexpr (KIND == EMIT_GAUSS) {
            const double x = static_cast<const double *>(obs_rm)[t] - mu_j;
            const double q0 = x * rs_j;
            const double d = fma(fma(-q0, sg_j, x), rs_j, q0); // == x / sigma
            double p = gauss_exp_block(-0.5 * d * d, cn_j);
            const unsigned long long nzm = __ballot(p != 0.0);
            if ((nzm & grp) == 0ull)
                p = 1.0; // outlier rule (outputmodel.py:126-130)
            if (in)
                pobs[e] = p;
        } else {
            const int sym = static_cast<const int32_t *>(obs_rm)[t];
            if (in)
                pobs[e] = m.B[(int64_t)j * m.M + sym];
        }
    }
}

// back-trace: one wavefront per trajectory stages 64 steps of back-pointers in LDS
// (coalesced), lane 0 chases them (_hidden.c:269-272)
// `bytes` back-pointer bytes (whole rows of n) to LDS: rows of whole 16-byte pieces (e.g. 64 states) go as 16-byte
// loads -- four per lane for 64 steps of 64 states instead of 64 byte loads
__device__ __forceinline__ void walk_tile_load(uint8_t *tile, const uint8_t *src, int bytes, int n, int lane)
{
    if ((n & 15) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(tile);
        for (int e = lane; e < (bytes >> 4); e += 64)
            d4[e] = s4[e];
    } else {
        for (int e = lane; e < bytes; e += 64)
            tile[e] = src[e];
    }
}

template <typename PT>
__global__ __launch_bounds__(64) void k_wide_viterbi_trace(const int64_t *off, int K, int n,
                                                           const uint8_t *ptr,
                                                           const int32_t *last_state,
                                                           PT *path)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * 64];
    __shared__ int32_t outp[64];
    const int k = blockIdx.x;
    const int lane = threadIdx.x;
    const int64_t o0 = off[k];
    const int64_t T = off[k + 1] - o0;
    if (T <= 0)
        return;
    int cur = last_state[k];
    if (lane == 0)
        path[o0 + T - 1] = (PT)cur;
    // steps t = hi .. lo (descending) use ptr[t] to produce path[t-1]
    for (int64_t hi = T - 1; hi >= 1; hi -= 64) {
        const int64_t lo = (hi - 63 > 1) ? hi - 63 : 1;
        const int cnt = (int)(hi - lo + 1);
        const int64_t base = (o0 + lo) * n;
        walk_tile_load(tile, ptr + base, cnt * n, n, lane);
        __syncthreads();
        if (lane == 0) {
            for (int q = cnt - 1; q >= 0; --q) {
                cur = tile[q * n + cur];
                outp[q] = cur; // path[lo + q - 1]
            }
        }
        __syncthreads();
        if (lane < cnt)
            path[o0 + lo + lane - 1] = (PT)outp[lane];
        cur = outp[0];
        __syncthreads();
    }
}

// back-trace of a segment-parallel run, parallel over the segments (the scheme of k_vit_walk, one lane
// per candidate state): APPLY = false -- lane c walks the back-pointers of segment s down from "state c
// at the segment's last step" and leaves the state that implies for the last step of segment s - 1
// (maps[s][c]); k_wide_vit_stitch chains the maps of a trajectory from its final state; APPLY = true --
// the segment is walked once more from its now known last state and the path is written
// (_hidden.c:269-272).  Back-pointer rows are staged through LDS 64 steps at a time.
// NC: candidate states per lane (1: up to 64 states, 2: up to 128; maps[s][64 NC])
template <bool APPLY, typename PT, int NC = 1>
__global__ __launch_bounds__(64) void k_wide_vit_walk(const int64_t *off, const Segs sg, int n,
                                                      const uint8_t *ptr, uint8_t *maps,
                                                      const uint8_t *end_state, PT *path)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * 64 * NC];
    __shared__ int32_t outp[64];
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= sg.nseg || sg.len[s] <= 0)
        return;
    const int k = sg.traj[s];
    const int64_t o0 = off[k];
    const int64_t t0 = sg.t0[s], t1 = t0 + sg.len[s];
    int cur[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c)
        cur[c] = APPLY ? (int)end_state[s] : (lane + 64 * c < n ? lane + 64 * c : 0);
    if (APPLY && lane == 0)
        path[o0 + t1 - 1] = (PT)cur[0];
    // steps t = hi .. lo (descending) use ptr[t] to produce the state at t - 1; APPLY stops above the
    // segment's first step (the state below it is the previous segment's last), the map pass goes on
    // to it -- except at the start of a trajectory, where there is nothing below
    const int64_t low = APPLY ? t0 + 1 : (t0 > 1 ? t0 : 1);
    for (int64_t hi = t1 - 1; hi >= low; hi -= 64) {
        const int64_t lo = (hi - 63 > low) ? hi - 63 : low;
        const int cnt = (int)(hi - lo + 1);
        const int64_t base = (o0 + lo) * n;
        walk_tile_load(tile, ptr + base, cnt * n, n, lane);
        __syncthreads();
        if constexpr (APPLY) {
            if (lane == 0) {
                for (int q = cnt - 1; q >= 0; --q) {
                    cur[0] = tile[q * n + cur[0]];
                    outp[q] = cur[0]; // path[lo + q - 1]
                }
            }
            __syncthreads();
            if (lane < cnt)
                path[o0 + lo + lane - 1] = (PT)outp[lane];
            cur[0] = outp[0];
        } else {
            for (int q = cnt - 1; q >= 0; --q)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    cur[c] = tile[q * n + cur[c]];
        }
        __syncthreads();
    }
    if constexpr (!APPLY) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            maps[(int64_t)s * 64 * NC + lane + 64 * c] = (uint8_t)cur[c];
    }
}

// end_state[s] for every segment: the trajectory's final state for its last segment, then map by map
[[maybe_unused]] static __global__ void k_wide_vit_stitch(const int32_t *traj0, int K, const uint8_t *maps,
                                                           int stride, const int32_t *last_state,
                                                           uint8_t *end_state)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K)
        return;
    const int s0 = traj0[k], s1 = traj0[k + 1];
    if (s1 <= s0)
        return;
    int cur = last_state[k];
    end_state[s1 - 1] = (uint8_t)cur;
    for (int s = s1 - 1; s > s0; --s) {
        cur = maps[(int64_t)s * stride + cur];
        end_state[s - 1] = (uint8_t)cur;
    }
}

template <int NP>
__global__ __launch_bounds__(64) void k_wide_sample_path(const WideModel m, const int64_t *off,
                                                         int K, const double *alpha_rm,
                                                         const double *u, uint64_t seed,
                                                         int32_t *path, int *status,
                                                         const int64_t *soff = nullptr)
{
    constexpr int GP = 64 / NP;
    __shared__ __attribute__((aligned(16))) double xs[GP][NP];
    const int lane = threadIdx.x;
    const int gi = lane / NP, i = lane % NP;
    const int k = blockIdx.x * GP + gi;
    if (k >= K)
        return;
    const int n = m.n;
    const bool real = i < n;
    const int64_t o0 = off[k];
    const int64_t T = off[k + 1] - o0;
    if (T <= 0)
        return;
    const int64_t s0 = soff ? soff[k] : o0; // position of this trajectory in the random stream
    int nxt = 0;
    for (int64_t t = T - 1; t >= 0; --t) {
        const double a = real ? alpha_rm[(o0 + t) * n + i] : 0.0;
        double ps = a;
        if (t != T - 1)
            ps = real ? a * m.A[(int64_t)i * n + nxt] : 0.0; // _hidden.c:365
        xs[gi][i] = ps;
        double S = 0.0;
#pragma unroll
        for (int q = 0; q < NP; ++q)
            S += xs[gi][q]; // _normalize, ascending
        const double pn = ps / S;
        xs[gi][i] = pn;
        const double r = u ? u[o0 + t] : uniform01(seed, (uint64_t)(s0 + t));
        double acc = 0.0;
        int pick = -1;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            acc += xs[gi][q];
            if (pick < 0 && q < n && acc >= r)
                pick = q;
        }
        if (pick < 0) {
            if (i == 0)
                status[0] = BHMM_ERR_CHOICE;
            pick = n - 1;
        }
        nxt = pick;
        if (i == 0)
            path[o0 + t] = pick;
    }
}

// Inclusive prefix sum over the NP lanes of a group (NP = 16: one row of 16 lanes, 32: two, 64: the
// wavefront) with DPP moves: row_shr 1, 2, 4, 8 inside the rows, row_bcast15 / row_bcast31 across them.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take_f64(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, ROW_MASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, ROW_MASK, 0xF, true);
    return __hiloint2double(hi, lo); // (lanes without a source, or masked out: 0)
}
template <int NP>
__device__ __forceinline__ double group_prefix_sum(double x)
{
    x += dpp_take_f64<0x111, 0xF>(x); // row_shr:1
    x += dpp_take_f64<0x112, 0xF>(x); // row_shr:2
    x += dpp_take_f64<0x114, 0xF>(x); // row_shr:4
    x += dpp_take_f64<0x118, 0xF>(x); // row_shr:8
    if constexpr (NP >= 32)
        x += dpp_take_f64<0x142, 0xA>(x); // row_bcast15 into rows 1 and 3
    if constexpr (NP >= 64)
        x += dpp_take_f64<0x143, 0xC>(x); // row_bcast31 into rows 2 and 3
    return x;
}

// =========================================================================================
// 9..64 states, backward sampling parallel over time segments (round 4).  The draw at step t is a
// function of (alpha_t, the state drawn at t + 1, the uniform of step t) -- the uniforms belong to the
// steps, not to the run -- so two runs that differ in where they started are COUPLED: once they draw
// the same state at some step they agree at every earlier one.
//   pass 0   a segment that does not end its trajectory starts W steps above its last step from state
//            0 (or from the exact rule at T - 1 where the warm-up reaches it) and notes the state it
//            drew for the step above its own (s_entry);
//   check    k_wide_smp_check: s_entry against the state the successor drew there (s_exit of s + 1);
//            where they differ the segment is flagged and s_entry becomes the successor's state;
//   fix-up   (FIX) a flagged segment is drawn again from that state and stops as soon as it draws the
//            state the path already holds at that step -- from there down nothing changes.  One
//            that reaches its first step writes a new s_exit for the next check.
// When a check finds nothing, every segment continued the state its successor drew: the path is the
// serial run's (_hidden.c:331-380), by induction from the last segment of each trajectory.
// A in LDS, transposed ([next state][i]: the lanes of a group read consecutive words).
// The draw itself (_hidden.c:283-305: normalise, then the first state whose cumulative sum reaches r)
// is decided WITHOUT its two ordered chains of n additions wherever that cannot change it: with the
// unnormalised prefix sums P_q of alpha_t[i] A[i][s_{t+1}] (a DPP scan, any order: within 64 eps S of
// the exact ones) the state is the first q with P_q >= r S, and the reference's normalised sums
// (within ~130 eps of P_q / S) order the same way unless some P_q lies within 1e-12 S of r S.  Only
// then -- or when S is not a normal number -- the step takes the reference's own arithmetic.
// =========================================================================================
template <int NP, bool FIX>
__global__ __launch_bounds__(64 * WVS_WPB) void k_wide_sample_seg(
    const WideModel m, const int64_t *off, const Segs sg, const double *alpha_rm, const double *u,
    uint64_t seed, int32_t *path, int *status, const int64_t *soff, int32_t *s_entry, int32_t *s_exit,
    const uint8_t *flag, const DrawWatch watch)
{
    constexpr int GP = 64 / NP;
    constexpr int TL = NP < 16 ? NP : 16;
    __shared__ __attribute__((aligned(16))) double xs[WVS_WPB][GP][NP];
    __shared__ __attribute__((aligned(16))) double xp[WVS_WPB][GP][NP];
    __shared__ double sAT[NP * NP];
    // (64 states: one segment per wavefront -- bounds, step count and the step's uniform in scalar registers)
    const int w = NP == 64 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int gi = lane / NP, i = lane % NP;
    const int n = m.n;
    for (int e = threadIdx.x; e < NP * NP; e += 64 * WVS_WPB) // sAT[nxt][i] = A[i][nxt]
        sAT[e] = (e / NP < n && e % NP < n) ? m.A[(int64_t)(e % NP) * n + e / NP] : 0.0;
    __syncthreads(); // (the only one)
    const int sgi = (blockIdx.x * WVS_WPB + w) * GP + gi;
    if (sgi >= sg.nseg || sg.len[sgi] <= 0)
        return;
    if constexpr (FIX) {
        if (!flag[sgi])
            return;
    }
    const bool real = i < n;
    const unsigned long long gmask = wgroup_mask<NP>(lane);
    auto uni = [](int64_t x) __attribute__((always_inline)) { // (NP == 64: the same in every lane; say so)
        if constexpr (NP == 64)
            return (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uint64_t)x >> 32)) << 32) |
                             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x));
        else
            return x;
    };
    const int k = (int)uni(sg.traj[sgi]);
    const int64_t o0 = uni(off[k]), T = uni(off[k + 1]) - o0;
    const int64_t s0 = soff ? uni(soff[k]) : o0; // position of this trajectory in the random stream
    const int64_t t0 = uni(sg.t0[sgi]), t1 = t0 + uni(sg.len[sgi]);
    const int64_t ts = FIX ? t1 - 1 : ((t1 + sg.W < T ? t1 + sg.W : T) - 1);
    int nxt = FIX ? s_entry[sgi] : 0;
    if constexpr (NP == 64)
        nxt = __builtin_amdgcn_readfirstlane(nxt);
    const int ic = real ? i : n - 1; // (padded lanes read a real entry and ignore it)
    bool met = false;
    // NP == 64, first pass: the drawn states of 64 steps in one register (lane l: step c0 + l), one coalesced
    // store per 64 steps -- a store per step keeps every wait of the loop waiting for it (one counter for loads and
    // stores), which is what made the ring useless when it was tried before
    constexpr bool PACKED_STORE = NP == 64 && !FIX;
    int pvec = 0;
    // (the step as a 32-bit count c = ts - t: 64-bit compares are vector instructions even on scalars)
    const int nst = (int)(ts - t0) + 1;                     // steps of this run
    const int cT = ts == T - 1 ? 0 : -1;                    // c of the trajectory's last step, if the run has it
    const int c1 = (int)(ts - t1);                          // c of step t1 (the one above the segment; -1: none)
    double rlane = 0.0; // NP == 64: lane l holds the uniform of step c0 + l, 64 steps at a time
    auto weight = [&](const int c, const double a) __attribute__((always_inline)) {
        return c != cT ? a * sAT[nxt * NP + i] : a; // _hidden.c:365 (padded states: 0 * 0)
    };
    auto step = [&](const int c, const double ps) __attribute__((always_inline)) {
        const int64_t t = ts - c;
        double r;
        if constexpr (NP == 64) {
            // (every lane computing the same counter-based uniform every step was 30 of the step's instructions)
            if ((c & 63) == 0) {
                const int64_t tl = t - lane;
                rlane = tl >= t0 ? (u ? u[o0 + tl] : uniform01(seed, (uint64_t)(s0 + tl))) : 0.0;
                // (waited for HERE, once in 64 steps: left pending, the load makes the read of rlane below a wait
                // for everything in flight -- the ring of alpha rows included -- in every step)
                asm volatile("" : "+v"(rlane));
            }
            r = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(rlane), c & 63),
                                 __builtin_amdgcn_readlane(__double2loint(rlane), c & 63));
        } else {
            r = u ? u[o0 + t] : uniform01(seed, (uint64_t)(s0 + t));
        }
        int pick = NP;
        bool watched = false;
        {
            const double P = group_prefix_sum<NP>(ps);
            const double Sf = __shfl(P, NP - 1, NP); // the group's total
            const double thr = r * Sf;
            const bool ok = Sf > 1e-290 && Sf < 1e290; // (also false for NaN)
            const bool near = !ok || !(fabs(P - thr) > 1e-12 * Sf);
            const unsigned long long nearm = __ballot(near && real) & gmask;
            // within reach of the deviation the alpha rows were verified to (watch.tol = 64 x that deviation;
            // 0 for rows of the serial recursion): recorded below, decided again afterwards (draw_verify.hpp)
            watched = watch.tol > 0.0 && (__ballot(real && !(fabs(P - thr) > watch.tol * Sf)) & gmask) != 0ull;
            if (nearm == 0ull) {
                const unsigned long long ge = __ballot(P >= thr) & gmask;
                pick = ge ? (int)__builtin_ctzll(ge) - gi * NP : NP;
            } else {
                // the reference's arithmetic, chain for chain
                xs[w][gi][i] = ps;
                double S = 0.0;
#pragma unroll 1 // (rare path: rolled -- fully unrolled, its 64 compare results cost the hot loop its scalar registers)
                for (int tl = 0; tl < NP; tl += TL) {
                    double x[TL];
#pragma unroll
                    for (int q = 0; q < TL; q += 2) {
                        const double2 y = *reinterpret_cast<const double2 *>(&xs[w][gi][tl + q]);
                        x[q] = y.x;
                        x[q + 1] = y.y;
                    }
#pragma unroll
                    for (int q = 0; q < TL; ++q)
                        S += x[q]; // _normalize, ascending
                }
                xp[w][gi][i] = ps / S;
                double acc = 0.0;
#pragma unroll 1 // (rare path: rolled -- fully unrolled, its 64 compare results cost the hot loop its scalar registers)
                for (int tl = 0; tl < NP; tl += TL) {
                    double x[TL];
#pragma unroll
                    for (int q = 0; q < TL; q += 2) {
                        const double2 y = *reinterpret_cast<const double2 *>(&xp[w][gi][tl + q]);
                        x[q] = y.x;
                        x[q + 1] = y.y;
                    }
#pragma unroll
                    for (int q = 0; q < TL; ++q) {
                        acc += x[q]; // _hidden.c:299-303: the first state whose cumulative sum reaches r
                        const int cand = (acc >= r) ? tl + q : NP;
                        pick = cand < pick ? cand : pick;
                    }
                }
            }
        }
        if (pick >= n) { // (a padded state's sum is the last real one's: it would have been drawn there)
            if (i == 0 && c > c1)
                status[0] = BHMM_ERR_CHOICE;
            pick = n - 1;
        } else if (__builtin_expect(watched && i == 0 && c > c1, 0)) {
            draw_record(watch, k, t, nxt, r, pick, -1.0);
        }
        nxt = pick;
        if constexpr (!FIX) {
            if (c == c1 && i == 0)
                s_entry[sgi] = pick;
        }
        if (c > c1) { // t < t1
            if constexpr (FIX) {
                if (path[o0 + t] == pick) {
                    met = true;
                    return;
                }
            }
            if constexpr (PACKED_STORE) {
                // c - c1 - 1 = 0, 1, ... counts the segment's own steps from its last one down
                const int cs = c - c1 - 1;
                pvec = lane == (cs & 63) ? pick : pvec;
                if ((cs & 63) == 63 || c == nst - 1) { // (64 collected, or the segment's first step)
                    const int64_t tl = t + (cs & 63) - lane; // lane l holds the step l below the block's top
                    if (lane <= (cs & 63))
                        path[o0 + tl] = pvec;
                }
            } else {
                if (i == 0)
                    path[o0 + t] = pick;
            }
        }
    };
    if constexpr (NP == 64 && !FIX) {
        // alpha rows four steps ahead of their draw (a row arrives from HBM after 1 - 2 us, a step takes a fifth of
        // that): a ring of four registers, the loop unrolled by four so that every slot has a name.  Every load is
        // issued unconditionally (beyond the run at a clamped step): with a load under a branch the compiler's wait
        // counts must assume it was not issued, and every wait becomes a wait for the newest load.
        constexpr int RING = 4;
        double ar[RING];
#pragma unroll
        for (int q = 0; q < RING; ++q)
            ar[q] = alpha_rm[(o0 + (ts - q > t0 ? ts - q : t0)) * n + ic];
        int cb = 0;
        for (; cb + RING <= nst; cb += RING) {
#pragma unroll
            for (int q = 0; q < RING; ++q) {
                const int64_t t = ts - (cb + q);
                // (the row's last use before the slot is loaded again: the same register, no copy at the loop's end
                // that would have to wait for the load)
                const double ps = weight(cb + q, real ? ar[q] : 0.0);
                ar[q] = alpha_rm[(o0 + (t - RING > t0 ? t - RING : t0)) * n + ic];
                step(cb + q, ps);
            }
        }
        for (int c = cb; c < nst; ++c) // (fewer than four steps left: their rows again, from the cache)
            step(c, weight(c, real ? alpha_rm[(o0 + ts - c) * n + ic] : 0.0));
    } else {
        double a_next = alpha_rm[(o0 + ts) * n + ic];
        for (int c = 0; c < nst; ++c) {
            const double a = real ? a_next : 0.0;
            if (c + 1 < nst)
                a_next = alpha_rm[(o0 + ts - c - 1) * n + ic]; // independent of the draw
            step(c, weight(c, a));
            if (FIX && met)
                break;
        }
    }
    if (!met && i == 0)
        s_exit[sgi] = nxt;
}

// result[3] = segments that did not continue the state their successor drew; flagged, s_entry replaced
[[maybe_unused]] static __global__ void k_wide_smp_check(const Segs sg, int32_t *s_entry, const int32_t *s_exit, uint8_t *flag,
                                 unsigned int *result)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    bool differs = false;
    if (s + 1 < sg.nseg && sg.len[s] > 0 && sg.traj[s + 1] == sg.traj[s]) {
        differs = s_entry[s] != s_exit[s + 1];
        if (differs)
            s_entry[s] = s_exit[s + 1];
    }
    if (s < sg.nseg)
        flag[s] = differs ? 1 : 0;
    const unsigned long long d = __ballot(differs);
    if ((threadIdx.x & 63) == 0 && d)
        atomicAdd(&result[3], (unsigned int)__popcll(d));
}

// hidden-path statistics for 9..64 states: one workgroup per trajectory; integer counts in
// LDS (exact), per-state emission sums through LDS fp64 atomics.
template <int KIND>
__global__ __launch_bounds__(256) void k_wide_path_stats(const WideModel m, const int64_t *off,
                                                         const void *obs_rm, const int32_t *path,
                                                         unsigned long long *counts, // [n*n+n]
                                                         double *epart,             // [K][esz]
                                                         int table_global) // the emission table of
                                                         // a big alphabet: epart itself (zeroed)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = m.n;
    const int esz = (KIND == EMIT_GAUSS) ? 3 * n : (KIND == EMIT_DISC ? n * m.M : 0);
    const int elds = table_global ? 0 : esz;
    unsigned int *cnt = reinterpret_cast<unsigned int *>(lds + elds);
    for (int e = threadIdx.x; e < elds; e += blockDim.x)
        lds[e] = 0.0;
    for (int e = threadIdx.x; e < n * n + n; e += blockDim.x)
        cnt[e] = 0u;
    __syncthreads();
    const int k = blockIdx.x;
    const int64_t o0 = off[k];
    const int64_t T = off[k + 1] - o0;
    for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
        const int st = path[o0 + t];
        if (t == 0)
            atomicAdd(&cnt[n * n + st], 1u);
        if (t + 1 < T)
            atomicAdd(&cnt[st * n + path[o0 + t + 1]], 1u);
        if constexpr (KIND == EMIT_GAUSS) {
            const double d = static_cast<const double *>(obs_rm)[o0 + t] - m.mu[st];
            atomicAdd(&lds[st], 1.0);
            atomicAdd(&lds[n + st], d);
            atomicAdd(&lds[2 * n + st], d * d);
        }
        if constexpr (KIND == EMIT_DISC) {
            const int64_t e = (int64_t)st * m.M + static_cast<const int32_t *>(obs_rm)[o0 + t];
            if (table_global)
                atomicAdd(&epart[(int64_t)blockIdx.x * esz + e], 1.0); // integer-valued: exact
            else
                atomicAdd(&lds[e], 1.0);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < n * n + n; e += blockDim.x)
        if (cnt[e])
            atomicAdd(&counts[e], (unsigned long long)cnt[e]);
    for (int e = threadIdx.x; e < elds; e += blockDim.x)
        epart[(int64_t)k * esz + e] = lds[e];
}

} // namespace bhmm
