// estep_kernels.hpp -- CDNA4 (gfx950) kernels of the batched E-step.
//
// Replaces the per-trajectory Python/C sequence of
// bhmm/estimators/maximum_likelihood.py:221-282 (p_obs -> forward -> backward -> gamma ->
// transition counts, then the host-side sums) by a parallel-in-time decomposition:
//
//   every trajectory is cut into time chunks; N/2 lanes own one chunk (a state pair each;
//   k_prescan: N lanes), so that a 64-wide wavefront advances 64 / (N/2) chunks per
//   instruction.  All per-step data (observations, alpha, gamma) live in HBM in a
//   "chunk-interleaved" (CI) layout  [group of 64 chunks][step][chunk][state]: the rows that
//   the chunks of one group touch at the same step are one contiguous record, and the 16-byte
//   elements of a wavefront's lanes (chunk-major, then state pair) are consecutive in it.
//
//   k_prescan  : per chunk, the N x N transfer matrix  prod_t A*diag(p_t)  (rows kept
//                exponent-normalised) -- makes the time recursion associative.
//   k_stitch   : per trajectory, sequential over chunks (N lanes per trajectory): exact
//                alpha at every chunk entry and beta at every chunk exit.
//   k_estep    : (estep_sweep.hpp) per chunk, forward sweep (alpha -> HBM, log-likelihood),
//                then the backward sweep that consumes alpha and accumulates gamma / xi /
//                emission sufficient statistics in registers.  beta and pobs never touch HBM.
//   k_rows     : forward-only / backward-only rows with the reference's normalisation (the
//                `hidden` API).
//   k_finalize : fixed-order reduction of the per-workgroup partials into the packed
//                statistics vector (the quantity that is all-reduced across GPUs).
//
// Arithmetic follows SURVEY.md Appendix A up to re-association (fp64, results within 1e-6
// relative of the reference; the bit-exact kernels -- Viterbi, path sampling -- live in
// path_kernels.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bhmm {

enum { EMIT_GAUSS = 0, EMIT_DISC = 1, EMIT_EXPL = 2 };
enum { MODE_ESTEP = 0, MODE_FWD = 1, MODE_BWD = 2 };

constexpr int BLOCK = 256;

// Model parameters, passed by value as kernel arguments (uniform -> scalar registers).
// N is the padded state count (2, 4 or 8); states >= nreal are inert (pi = 0, p = 0).
template <int N>
struct Model {
    double A[N * N];
    double pi[N];
    double e0[N]; // gaussian: mean
    double e1[N]; // gaussian: 1/sigma
    double e2[N]; // gaussian: 1/(sqrt(2 pi) sigma)      (_gaussian.c:18)
    double e3[N]; // gaussian: sigma (the bit-exact path kernels divide by it)
    double e4[N]; // gaussian, gauss_pdf(): log2(e) / (2 sigma^2) / 4096
    double e5[N]; // gaussian, gauss_pdf(): (emg_s - log2 e2) / 4096 >= 0; 1 for a padded state
    double emg;   // gaussian, gauss_pdf(): 1.5 * 2^40 + emg_s / 4096, emg_s = ceil(log2(max e2)); NaN if a
                  // sigma is not a positive finite number
    int nreal;
    int M; // number of symbols (discrete)
    int dcopies; // k_estep, discrete: LDS count tables per workgroup (one per wavefront, or 1)
    int bt_global; // discrete, alphabet too large for the LDS: B^T is read from global memory
                   // (L2-resident) and the emission counts go to DISC_GLOBAL_TABLES global tables
};
// big alphabets: number of replicated global count tables (workgroup b uses table b % this)
constexpr int DISC_GLOBAL_TABLES = 8;

// Chunk table (device pointers), one entry per lane of the launch; padded entries have
// len == 0.
struct Chunks {
    const int32_t *traj; // trajectory index of the chunk
    const int64_t *t0;   // first time step of the chunk inside its trajectory
    const int32_t *len;  // number of steps
    const int64_t *goff; // t0 + offset of the trajectory in the concatenated arrays
    int Lmax;            // record stride per wavefront (max chunk length)
};

// ---- CI addressing ------------------------------------------------------------------
__device__ __forceinline__ int64_t ci_rec(int64_t g, int s, int Lmax)
{
    return (g >> 6) * (int64_t)Lmax + s;
}

template <int N>
__device__ __forceinline__ void ci_load(const double *base, int64_t rec, int lane, double (&v)[N])
{
    const double2 *p =
        reinterpret_cast<const double2 *>(base + rec * (int64_t)(N * 64)) + lane * (N / 2);
#pragma unroll
    for (int q = 0; q < N / 2; ++q) {
        const double2 x = p[q];
        v[2 * q] = x.x;
        v[2 * q + 1] = x.y;
    }
}

template <int N>
__device__ __forceinline__ void ci_store(double *base, int64_t rec, int lane, const double (&v)[N])
{
    double2 *p = reinterpret_cast<double2 *>(base + rec * (int64_t)(N * 64)) + lane * (N / 2);
#pragma unroll
    for (int q = 0; q < N / 2; ++q)
        p[q] = make_double2(v[2 * q], v[2 * q + 1]);
}

// ---- emission probabilities, fused (never written to HBM) ------------------------------
// gaussian: _gaussian.c:5-21 + the outlier rule of outputmodel.py:119-131
// discrete: discrete.py:150-153 (column gather from B, staged transposed in LDS)
// explicit: pobs rows supplied by the caller (hidden/api.py signatures)
template <int N, int KIND>
__device__ __forceinline__ void emit(const Model<N> &m, const void *obs_ci, const double *Bt,
                                     int64_t rec, int lane, double (&p)[N], double &o, int &sym)
{
    if constexpr (KIND == EMIT_GAUSS) {
        o = static_cast<const double *>(obs_ci)[rec * 64 + lane];
        double mx = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double z = i < m.nreal ? (o - m.e0[i]) / m.e3[i] : 0.0; // (_gaussian.c:5-21, see k_prescan)
            p[i] = i < m.nreal ? m.e2[i] * exp(-0.5 * z * z) : 0.0;
            mx = fmax(mx, p[i]);
        }
        if (mx == 0.0) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                p[i] = (i < m.nreal) ? 1.0 : 0.0;
        }
    } else if constexpr (KIND == EMIT_DISC) {
        sym = static_cast<const int32_t *>(obs_ci)[rec * 64 + lane];
        const double2 *row = reinterpret_cast<const double2 *>(Bt + (int64_t)sym * N);
#pragma unroll
        for (int q = 0; q < N / 2; ++q) {
            const double2 x = row[q];
            p[2 * q] = x.x;
            p[2 * q + 1] = x.y;
        }
    } else {
        ci_load<N>(static_cast<const double *>(obs_ci), rec, lane, p);
    }
    // a row in the denormal range (cf. emit_raw, estep_sweep.hpp): times 2^900, exactly.  These
    // kernels only produce boundary vectors, which carry no scale.
    if constexpr (KIND != EMIT_DISC) {
        double mx = p[0];
#pragma unroll
        for (int i = 1; i < N; ++i)
            mx = fmax(mx, p[i]);
        if (mx < 0x1p-959) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                p[i] = ldexp(p[i], 900);
        }
    }
}

// exponent of an all-zero row of a transfer matrix (k_prescan, k_compose, k_stitch)
constexpr int ROW_EXP_NONE = -(1 << 28);

// exact power-of-two renormalisation of one row; returns the exponent removed
template <int N>
__device__ __forceinline__ int renorm_row(const double (&nr)[N], double (&dst)[N])
{
    double mx = nr[0];
#pragma unroll
    for (int j = 1; j < N; ++j)
        mx = fmax(mx, nr[j]);
    int e;
    (void)frexp(mx, &e);
#pragma unroll
    for (int j = 0; j < N; ++j)
        dst[j] = ldexp(nr[j], -e);
    return e;
}

template <int N>
__device__ __forceinline__ void stage_Bt(double *dst, const double *Bt_g, int M)
{
    for (int i = threadIdx.x; i < M * N; i += blockDim.x)
        dst[i] = Bt_g[i];
}

// ---- cross-lane helpers on DPP (no LDS round trip): exchanges inside aligned groups of
// 2, 4 or 8 lanes.  xor1/xor2 are quad permutes; xor4 is row_half_mirror followed by a quad
// reversal (lane i -> 7-i -> i^4).
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int H>
__device__ __forceinline__ int xchg_i32(int v)
{
    if constexpr (H == 1)
        return dpp_i32<0xB1>(v); // quad_perm [1,0,3,2]
    else if constexpr (H == 2)
        return dpp_i32<0x4E>(v); // quad_perm [2,3,0,1]
    else
        return dpp_i32<0x1B>(dpp_i32<0x141>(v)); // row_half_mirror, quad_perm [3,2,1,0]
}
template <int H>
__device__ __forceinline__ double xchg_f64(double x)
{
    const int lo = xchg_i32<H>(__double2loint(x));
    const int hi = xchg_i32<H>(__double2hiint(x));
    return __hiloint2double(hi, lo);
}
template <int N>
__device__ __forceinline__ int group_max(int v)
{
    v = max(v, xchg_i32<1>(v));
    if constexpr (N >= 4)
        v = max(v, xchg_i32<2>(v));
    if constexpr (N >= 8)
        v = max(v, xchg_i32<4>(v));
    return v;
}
// 1/x to ~1 ulp: hardware seed + two Newton steps (the full IEEE division sequence is only
// needed by the bit-exact path kernels)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ int exponent_of(double x)
{
    int e;
    (void)frexp(x, &e);
    return e;
}

// =========================================================================================
// k_prescan: chunk transfer matrices.
//   Mc = prod_{t in chunk} A diag(p_t)   (t = t0 .. t0+len-1; for the first chunk of a
//   trajectory the product starts at t = 1 and every row is seeded with pi o p_0, so that
//   e_0^T Mc is the unnormalised alpha at the chunk end).
//   Row r is the forward recursion started from unit vector e_r (_hidden.c:42-63 without
//   the division).  Mapping: N lanes per chunk, lane r owns row r (N doubles) and a private
//   register copy of A; the step's emission vector is computed one state per lane and
//   exchanged through a 64-byte LDS slot per chunk.  A workgroup of 64*N threads covers the
//   same 64 chunks as one wavefront of k_estep (one CI record group), so observation loads
//   stay inside one 512 B segment.  Rows are renormalised by a power of two every step
//   (exact); the exponent is carried separately:  true row r = 2^ex[r] * stored row r.
//   Output per chunk: N*N doubles row-major + N exponents (as doubles).
// =========================================================================================
template <int N, int KIND>
__global__ __launch_bounds__(64 * N) void k_prescan(const Model<N> m, const Chunks ch,
                                                    const void *obs_ci, const double *Bt_g,
                                                    double *Mbuf)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *sA = smem;             // [N][N]
    double *sP = smem + N * N;     // [64 chunks][N] emission exchange
    const double *sBt = sP + 64 * N; // [M][N] (discrete)
    for (int i = threadIdx.x; i < N * N; i += blockDim.x)
        sA[i] = m.A[i];
    if constexpr (KIND == EMIT_DISC) {
        if (m.bt_global)
            sBt = Bt_g;
        else
            stage_Bt<N>(sP + 64 * N, Bt_g, m.M);
    }
    __syncthreads();
    const int cl = threadIdx.x / N; // chunk within the record group == CI lane
    const int r = threadIdx.x % N;
    const int64_t g = (int64_t)blockIdx.x * 64 + cl;
    const int len = ch.len[g];
    if (len == 0)
        return;
    const bool first = (ch.t0[g] == 0);
    double A[N][N];
#pragma unroll
    for (int k = 0; k < N; ++k)
#pragma unroll
        for (int j = 0; j < N; j += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(sA + k * N + j);
            A[k][j] = x.x;
            A[k][j + 1] = x.y;
        }
    const double mu_r = m.e0[r], sg_r = m.e3[r], cn_r = m.e2[r];
    const bool real = r < m.nreal;
    double *slot = sP + cl * N;
    const unsigned long long gmask = ((N == 64) ? ~0ull : ((1ull << N) - 1))
                                     << ((threadIdx.x & 63) / N * N);

    double row[N];
#pragma unroll
    for (int j = 0; j < N; ++j)
        row[j] = (r == j) ? 1.0 : 0.0;
    int ex = 0;
    // raw input of the next step, loaded one iteration ahead (the recursion is serial; the
    // load latency would otherwise sit on the critical path of every step)
    auto load_in = [&](int64_t rec, double &o, int &sym, double &pv) {
        if constexpr (KIND == EMIT_GAUSS)
            o = static_cast<const double *>(obs_ci)[rec * 64 + cl];
        else if constexpr (KIND == EMIT_DISC)
            sym = static_cast<const int32_t *>(obs_ci)[rec * 64 + cl];
        else
            pv = static_cast<const double *>(obs_ci)[rec * (int64_t)(N * 64) + cl * N + r];
    };
    double o_n = 0.0, pv_n = 0.0;
    int sym_n = 0;
    load_in(ci_rec(g, 0, ch.Lmax), o_n, sym_n, pv_n);
    for (int s = 0; s < len; ++s) {
        const int64_t rec = ci_rec(g, s, ch.Lmax);
        const double o = o_n, pv = pv_n;
        const int sym = sym_n;
        if (s + 1 < len)
            load_in(rec + 1, o_n, sym_n, pv_n);
        double pr;
        if constexpr (KIND == EMIT_GAUSS) {
            // the reference's own operation order (_gaussian.c:5-21): at the bottom of the
            // denormal range this decides between "all zero" (outlier rule) and a last non-zero
            // entry, and the sweeps (emit_raw) decide it the same way
            const double z = real ? (o - mu_r) / sg_r : 0.0;
            pr = real ? cn_r * exp(-0.5 * z * z) : 0.0;
            if ((__ballot(pr != 0.0) & gmask) == 0ull) // outlier row, outputmodel.py:126-130
                pr = real ? 1.0 : 0.0;
            else if ((__ballot(pr >= 0x1p-959) & gmask) == 0ull) // denormal range: times 2^900, exactly
                pr = ldexp(pr, 900);
        } else if constexpr (KIND == EMIT_DISC) {
            pr = sBt[sym * N + r];
        } else {
            pr = pv;
            if ((__ballot(pr >= 0x1p-959) & gmask) == 0ull && (__ballot(pr != 0.0) & gmask) != 0ull)
                pr = ldexp(pr, 900);
        }
        slot[r] = pr;
        double p[N];
#pragma unroll
        for (int j = 0; j < N; j += 2) {
            const double2 x = *reinterpret_cast<const double2 *>(slot + j);
            p[j] = x.x;
            p[j + 1] = x.y;
        }
        double nr[N];
        if (first && s == 0) {
#pragma unroll
            for (int j = 0; j < N; ++j)
                nr[j] = m.pi[j] * p[j];
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
                double acc = row[0] * A[0][j];
#pragma unroll
                for (int k = 1; k < N; ++k)
                    acc = fma(row[k], A[k][j], acc);
                nr[j] = acc * p[j];
            }
        }
        ex += renorm_row<N>(nr, row);
    }
    double *out = Mbuf + g * (int64_t)(N * N + N);
    bool zero_row = true;
#pragma unroll
    for (int j = 0; j < N; j += 2) {
        *reinterpret_cast<double2 *>(out + r * N + j) = make_double2(row[j], row[j + 1]);
        zero_row = zero_row && row[j] == 0.0 && row[j + 1] == 0.0;
    }
    // a zero row (no path from state r explains the chunk: sparse A, exact zeros among the emission
    // probabilities) carries the exponent "minus infinity": it must not take part when the rows'
    // exponents are aligned downstream (the others would be shifted out of range)
    out[N * N + r] = zero_row ? (double)ROW_EXP_NONE : (double)ex;
}

// =========================================================================================
// k_compose: product of the transfer matrices of one group of consecutive chunks,
//   P_g = M_{c0} M_{c0+1} ... M_{c1-1},  same storage format as M (rows power-of-two
// normalised, exponents separate).  N lanes per group, lane r carries row r of the running
// product.  It makes k_stitch two-level: groups are stitched first (serial depth n/R), then
// the chunks inside every group in parallel (depth R) -- instead of depth n.
// =========================================================================================
template <int N>
__global__ __launch_bounds__(64) void k_compose(const int32_t *grp_c0, const int32_t *grp_c1,
                                                int nG, const double *Mbuf, double *Pbuf)
{
    constexpr int GP = 64 / N;
    constexpr int MS = N * N + N;
    constexpr int NEG = -(1 << 28);
    const int gidx = (int)blockIdx.x * GP + (int)threadIdx.x / N;
    const int r = threadIdx.x % N;
    if (gidx >= nG)
        return;
    const int c0 = grp_c0[gidx], c1 = grp_c1[gidx];
    double row[N];
    int ex = 0;
    {
        const double *src = Mbuf + (int64_t)c0 * MS;
#pragma unroll
        for (int j = 0; j < N; ++j)
            row[j] = src[r * N + j];
        ex = (int)src[N * N + r];
    }
    for (int c = c0 + 1; c < c1; ++c) {
        const double *src = Mbuf + (int64_t)c * MS;
        // weights w_k = row[k] * 2^(e_c[k] - E): align the exponents of the rows of M_c
        int ek[N], E = NEG;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            ek[k] = (int)src[N * N + k];
            if (row[k] > 0.0 && ek[k] > ROW_EXP_NONE / 2)
                E = max(E, ek[k] + exponent_of(row[k]));
        }
        double w[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
            w[k] = ek[k] > ROW_EXP_NONE / 2 ? ldexp(row[k], ek[k] - E) : 0.0;
        double nr[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            double acc = w[0] * src[j];
#pragma unroll
            for (int k = 1; k < N; ++k)
                acc = fma(w[k], src[k * N + j], acc);
            nr[j] = acc;
        }
        ex += (E == NEG ? 0 : E) + renorm_row<N>(nr, row);
    }
    double *out = Pbuf + (int64_t)gidx * MS;
    bool zero_row = ex <= ROW_EXP_NONE / 2;
    {
        bool all0 = true;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            out[r * N + j] = row[j];
            all0 = all0 && row[j] == 0.0;
        }
        zero_row = zero_row || all0;
    }
    out[N * N + r] = zero_row ? (double)ROW_EXP_NONE : (double)ex;
}

// =========================================================================================
// k_stitch: exact chunk-boundary vectors.  N lanes cooperate on one (trajectory,
// direction); lane r holds row r of the current transfer matrix.
//   forward : alpha_entry[c] ~ alpha at the step before chunk c
//             (_hidden.c:42-63 collapsed over a chunk:  a <- a^T Mc)
//   backward: beta_exit[c]   ~ beta at the last step of chunk c
//             (_hidden.c:91-109 collapsed:  b <- M_{c+1} b,  b_{T-1} = 1/N)
// Both vectors are kept up to a power-of-two scale (largest entry in [0.25, 8)); k_estep / k_rows
// normalises them where the reference's normalisation matters.  Cross-lane traffic is DPP
// only; the next PD matrices are prefetched into registers because the chain is latency bound.
// Blocks [0, nb) run the forward direction, [nb, 2 nb) the backward one.
// =========================================================================================
template <int N>
__global__ __launch_bounds__(64) void k_stitch(const int32_t *seg_c0, const int32_t *seg_c1, int K,
                                               int nb, int nreal, const double *Mbuf,
                                               const double *a_init, const double *b_init,
                                               double *alpha_entry, double *beta_exit)
{
    constexpr int GP = 64 / N; // trajectories per block
    constexpr int MS = N * N + N;
    constexpr int PD = 4;      // prefetch depth (chunks)
    constexpr int NEG = -(1 << 28);
    const bool bwd = (int)blockIdx.x >= nb;
    const int k = ((int)blockIdx.x % nb) * GP + (int)threadIdx.x / N;
    const int r = threadIdx.x % N;
    if (k >= K)
        return;
    // a segment is a run of consecutive chunks of one trajectory: the whole trajectory
    // (a_init == nullptr), or one group of the two-level scheme (vectors at the group
    // boundaries come from the group-level pass)
    const int c0 = seg_c0[k], c1 = seg_c1[k];
    const int nc = c1 - c0;

    double rows[PD][N], er[PD];
    // forward: row r in natural column order; backward: in xor order (entry q = M[r][r^q]),
    // matching the order the DPP all-gather delivers b in
    auto fetch = [&](int u, int c) {
        const double *src = Mbuf + (int64_t)c * MS + r * N;
        if (!bwd) {
#pragma unroll
            for (int j = 0; j < N; j += 2) {
                const double2 x = *reinterpret_cast<const double2 *>(src + j);
                rows[u][j] = x.x;
                rows[u][j + 1] = x.y;
            }
        } else {
#pragma unroll
            for (int q = 0; q < N; ++q)
                rows[u][q] = src[r ^ q];
        }
        er[u] = Mbuf[(int64_t)c * MS + N * N + r];
    };

    if (!bwd) {
        double a = a_init ? a_init[(int64_t)k * N + r]
                          : ((r == 0) ? 1.0 : 0.0); // e_0 selects row 0 of the seeded first chunk
#pragma unroll
        for (int u = 0; u < PD; ++u)
            if (u < nc)
                fetch(u, c0 + u);
        for (int cb = 0; cb < nc; cb += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int ci = cb + u;
                if (ci < nc) {
                    alpha_entry[(int64_t)(c0 + ci) * N + r] = a;
                    const int e = (int)er[u];
                    const bool use = a > 0.0 && e > ROW_EXP_NONE / 2; // (a zero row of M: see k_prescan)
                    const int E = group_max<N>(use ? e + exponent_of(a) : NEG);
                    const double w = use ? ldexp(a, e - E) : 0.0;
                    double buf[N];
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        buf[j] = w * rows[u][j];
                    if (ci + PD < nc)
                        fetch(u, c0 + ci + PD);
                    // reduce-scatter over the N lanes: lane j ends with sum_r w_r M[r][j]
                    if constexpr (N >= 8) {
                        const bool up = (r & 4) != 0;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const double keep = up ? buf[i + 4] : buf[i];
                            const double send = up ? buf[i] : buf[i + 4];
                            buf[i] = keep + xchg_f64<4>(send);
                        }
                    }
                    if constexpr (N >= 4) {
                        const bool up = (r & 2) != 0;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const double keep = up ? buf[i + 2] : buf[i];
                            const double send = up ? buf[i] : buf[i + 2];
                            buf[i] = keep + xchg_f64<2>(send);
                        }
                    }
                    {
                        const bool up = (r & 1) != 0;
                        const double keep = up ? buf[1] : buf[0];
                        const double send = up ? buf[0] : buf[1];
                        buf[0] = keep + xchg_f64<1>(send);
                    }
                    a = buf[0]; // largest entry of the group is in [0.25, N)
                }
            }
        }
    } else {
        double b = b_init ? b_init[(int64_t)k * N + r]
                          : ((r < nreal) ? 1.0 / (double)nreal : 0.0); // _hidden.c:79-88
        // chunk c (> c0) is consumed when producing the exit vector of chunk c-1
#pragma unroll
        for (int u = 0; u < PD; ++u)
            if (u < nc - 1)
                fetch(u, c1 - 1 - u);
        for (int cb = 0; cb < nc; cb += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int ci = cb + u; // counts chunks from the end
                if (ci < nc) {
                    const int c = c1 - 1 - ci;
                    beta_exit[(int64_t)c * N + r] = b;
                    if (c > c0) {
                        // all-gather b over the group in xor order: bx[q] = b of lane r^q
                        double bx[N];
                        bx[0] = b;
                        bx[1] = xchg_f64<1>(b);
                        if constexpr (N >= 4) {
                            bx[2] = xchg_f64<2>(bx[0]);
                            bx[3] = xchg_f64<2>(bx[1]);
                        }
                        if constexpr (N >= 8) {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                bx[4 + q] = xchg_f64<4>(bx[q]);
                        }
                        // s = sum_j M[r][j] b_j, summed in xor order j = r ^ q
                        double s = 0.0;
#pragma unroll
                        for (int q = 0; q < N; ++q)
                            s = fma(rows[u][q], bx[q], s);
                        const int e = (int)er[u];
                        if (ci + PD < nc - 1)
                            fetch(u, c - PD);
                        const bool use = s > 0.0 && e > ROW_EXP_NONE / 2;
                        const int E = group_max<N>(use ? e + exponent_of(s) : NEG);
                        b = use ? ldexp(s, e - E) : 0.0; // largest entry of the group in [0.5, 1)
                    }
                }
            }
        }
    }
}

// =========================================================================================
// Lane mapping of the streaming kernels (k_estep in estep_sweep.hpp, k_rows below):
// H = N/2 lanes cooperate on one chunk; lane q owns the state pair (2q, 2q+1), i.e. exactly one
// 16-byte element of every CI record, the two columns / rows of A that touch its states, and
// (k_estep) the two rows of the xi accumulator.  A workgroup of 64*H threads covers one CI
// record group (64 chunks).  Compared with one lane per chunk this cuts the per-lane register
// state by H (the 2*N*N-register xi accumulator was what limited occupancy to one wavefront per
// SIMD), keeps A in registers, and leaves the instruction count per chunk-step about equal.
// =========================================================================================
template <int N, int KIND>
struct StatLayout {
    static constexpr int NC = N * N;                             // xi accumulators
    static constexpr int NG = N;                                 // sum_t gamma
    static constexpr int NE = (KIND == EMIT_GAUSS) ? 2 * N : 0;  // sum gamma d, sum gamma d^2
    static constexpr int S = NC + NG + NE;
};

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1)
        v += __shfl_xor(v, h, 64);
    return v;
}

// value of lane K (0..H-1) of my H-lane group, H in {1,2,4}: one quad permute per dword
template <int H, int K>
__device__ __forceinline__ double grp_bcast(double x)
{
    if constexpr (H == 1) {
        return x;
    } else {
        constexpr int CTRL = (H == 4) ? (K | (K << 2) | (K << 4) | (K << 6))
                                      : (K | (K << 2) | ((2 + K) << 4) | ((2 + K) << 6));
        const int lo = dpp_i32<CTRL>(__double2loint(x));
        const int hi = dpp_i32<CTRL>(__double2hiint(x));
        return __hiloint2double(hi, lo);
    }
}
template <int H>
__device__ __forceinline__ double grp_sum(double x)
{
    if constexpr (H >= 2)
        x += xchg_f64<1>(x);
    if constexpr (H >= 4)
        x += xchg_f64<2>(x);
    return x;
}
template <int H>
__device__ __forceinline__ int grp_max_i32(int v)
{
    if constexpr (H >= 2)
        v = max(v, xchg_i32<1>(v));
    if constexpr (H >= 4)
        v = max(v, xchg_i32<2>(v));
    return v;
}
// all-gather: full[2k + b] = pair[b] of lane k
template <int N>
__device__ __forceinline__ void grp_gather(const double (&pair)[2], double (&full)[N])
{
    constexpr int H = N / 2;
    full[0] = grp_bcast<H, 0>(pair[0]);
    full[1] = grp_bcast<H, 0>(pair[1]);
    if constexpr (H >= 2) {
        full[2] = grp_bcast<H, 1>(pair[0]);
        full[3] = grp_bcast<H, 1>(pair[1]);
    }
    if constexpr (H >= 4) {
        full[4] = grp_bcast<H, 2>(pair[0]);
        full[5] = grp_bcast<H, 2>(pair[1]);
        full[6] = grp_bcast<H, 3>(pair[0]);
        full[7] = grp_bcast<H, 3>(pair[1]);
    }
}

// Raw per-step input of one lane: the observation (gaussian / discrete) or my pobs pair.
struct ObsIn {
    double o;
    int sym;
    double2 pp;
};
template <int N, int KIND>
__device__ __forceinline__ ObsIn load_obs(const void *obs_ci, int64_t rec, int cl, int q)
{
    ObsIn in;
    in.o = 0.0;
    in.sym = 0;
    in.pp = make_double2(0.0, 0.0);
    if constexpr (KIND == EMIT_GAUSS)
        in.o = static_cast<const double *>(obs_ci)[rec * 64 + cl];
    else if constexpr (KIND == EMIT_DISC)
        in.sym = static_cast<const int32_t *>(obs_ci)[rec * 64 + cl];
    else
        in.pp = *(reinterpret_cast<const double2 *>(static_cast<const double *>(obs_ci) +
                                                    rec * (int64_t)(N * 64)) +
                  cl * (N / 2) + q);
    return in;
}

// emission probabilities of MY two states (+ the outlier rule).  `is` holds sigma: the gaussian
// density is evaluated in the reference's own operation order (_gaussian.c:5-21; cf. k_prescan).
// A row without any entry of 2^-959 or more (densities / caller-supplied rows in the denormal range)
// is returned times 2^900 and the function returns 900 (else 0): the callers normalise by sums and
// their reciprocals, which are infinite for denormal sums; the forward rows take the exponent off
// their likelihood count.
template <int N, int KIND>
__device__ __forceinline__ int emit_pair(const Model<N> &m, const ObsIn &in, const double *Bt,
                                         int q, const double (&mu)[2], const double (&is)[2],
                                         const double (&cn)[2], unsigned long long gmask,
                                         double (&p)[2])
{
    if constexpr (KIND == EMIT_GAUSS) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const bool real = 2 * q + b < m.nreal;
            const double z = real ? (in.o - mu[b]) / is[b] : 0.0;
            p[b] = real ? cn[b] * exp(-0.5 * z * z) : 0.0;
        }
        if ((__ballot(p[0] != 0.0 || p[1] != 0.0) & gmask) == 0ull) {
            p[0] = (2 * q < m.nreal) ? 1.0 : 0.0; // outlier row, outputmodel.py:126-130
            p[1] = (2 * q + 1 < m.nreal) ? 1.0 : 0.0;
        }
    } else if constexpr (KIND == EMIT_DISC) {
        const double2 x = *reinterpret_cast<const double2 *>(Bt + (int64_t)in.sym * N + 2 * q);
        p[0] = x.x;
        p[1] = x.y;
    } else {
        p[0] = in.pp.x;
        p[1] = in.pp.y;
    }
    if ((__ballot(p[0] >= 0x1p-959 || p[1] >= 0x1p-959) & gmask) == 0ull &&
        (__ballot(p[0] != 0.0 || p[1] != 0.0) & gmask) != 0ull) {
        p[0] = ldexp(p[0], 900);
        p[1] = ldexp(p[1], 900);
        return 900;
    }
    return 0;
}

__device__ __forceinline__ double2 *ci_pair(double *base, int64_t rec, int N_, int q, int cl)
{
    return reinterpret_cast<double2 *>(base + rec * (int64_t)(N_ * 64)) + cl * (N_ / 2) + q;
}

// =========================================================================================
// k_rows<N, KIND, MODE>: forward (MODE_FWD) or backward (MODE_BWD) rows with the reference's own
// per-step normalisation (_hidden.c:16-66, :69-110), into the CI workspace -- the forward-only /
// backward-only passes of the `hidden` API, which must return the reference's normalised rows.
// Chunk-boundary vectors come from k_stitch (exact).  The E-step itself runs k_estep
// (estep_sweep.hpp), which carries alpha and beta up to powers of two instead.
// =========================================================================================
template <int N, int KIND, int MODE>
__global__ __launch_bounds__(32 * N) void k_rows(const Model<N> m, const Chunks ch,
                                                 const void *obs_ci, const double *Bt_g,
                                                 const double *alpha_entry,
                                                 const double *beta_exit,
                                                 double *ws,         // CI rows: alpha or beta
                                                 double *logL_chunk) // [G] (MODE_FWD)
{
    static_assert(MODE == MODE_FWD || MODE == MODE_BWD, "row passes only");
    constexpr int H = N / 2;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const double *Bt = smem; // [M][N]
    if constexpr (KIND == EMIT_DISC) {
        if (m.bt_global) {
            Bt = Bt_g;
        } else {
            stage_Bt<N>(smem, Bt_g, m.M);
            __syncthreads();
        }
    }
    const int cl = threadIdx.x / H; // chunk within the record group == CI lane
    const int q = threadIdx.x % H;  // my state pair
    const int64_t g = (int64_t)blockIdx.x * 64 + cl;
    const int len = ch.len[g];
    if (len <= 0)
        return;
    const bool first = (ch.t0[g] == 0);
    const unsigned long long gmask = ((1ull << H) - 1) << ((threadIdx.x & 63) / H * H);
    double mu[2], is[2], cn[2], pi2[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        mu[b] = m.e0[2 * q + b];
        is[b] = m.e3[2 * q + b]; // sigma (emit_pair)
        cn[b] = m.e2[2 * q + b];
        pi2[b] = m.pi[2 * q + b];
    }
    if constexpr (MODE == MODE_FWD) {
        double Ac[N][2], a[2];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            Ac[i][0] = m.A[i * N + 2 * q];
            Ac[i][1] = m.A[i * N + 2 * q + 1];
        }
        double P = 1.0; // running product of the scaling factors c_t, mantissa part
        int eP = 0;     // ... and its binary exponent: logL = log(P) + eP ln 2
        int s = 0;
        ObsIn nxt = load_obs<N, KIND>(obs_ci, ci_rec(g, 0, ch.Lmax), cl, q);
        if (first) {
            double p[2];
            const int pe0 = emit_pair<N, KIND>(m, nxt, Bt, q, mu, is, cn, gmask, p);
            if (len > 1)
                nxt = load_obs<N, KIND>(obs_ci, ci_rec(g, 1, ch.Lmax), cl, q);
            a[0] = pi2[0] * p[0];
            a[1] = pi2[1] * p[1];
            int pex = pe0;
            // the whole row in the denormal range (an explicit emission row may hold denormals: the
            // reciprocal of its sum would be infinite) although the factors are not zero: times 2^900,
            // exactly, and 900 off the exponent count -- the reference divides by the denormal sum
            if (grp_max_i32<H>(max(__double2hiint(a[0]), __double2hiint(a[1]))) < (64 << 20)) {
                a[0] = pi2[0] * ldexp(p[0], 900);
                a[1] = pi2[1] * ldexp(p[1], 900);
                pex += 900;
            }
            const double c = grp_sum<H>(a[0] + a[1]);
            const double rc = fast_rcp(c);
            a[0] *= rc;
            a[1] *= rc;
            P = frexp(c, &eP);
            eP -= pex;
            *ci_pair(ws, ci_rec(g, 0, ch.Lmax), N, q, cl) = make_double2(a[0], a[1]);
            s = 1;
        } else {
            // entry vector from k_stitch (power-of-two scaled): normalise, _hidden.c:57-59
            const double2 x = *reinterpret_cast<const double2 *>(alpha_entry + g * N + 2 * q);
            const double rS = fast_rcp(grp_sum<H>(x.x + x.y));
            a[0] = x.x * rS;
            a[1] = x.y * rS;
        }
        for (; s < len; ++s) {
            double p[2];
            const int64_t rec = ci_rec(g, s, ch.Lmax);
            const ObsIn cur = nxt;
            if (s + 1 < len) // issue the next step's load before this step's arithmetic
                nxt = load_obs<N, KIND>(obs_ci, rec + 1, cl, q);
            const int pe = emit_pair<N, KIND>(m, cur, Bt, q, mu, is, cn, gmask, p);
            double af[N];
            grp_gather<N>(a, af);
            double n0 = af[0] * Ac[0][0], n1 = af[0] * Ac[0][1];
#pragma unroll
            for (int i = 1; i < N; ++i) {
                n0 = fma(af[i], Ac[i][0], n0);
                n1 = fma(af[i], Ac[i][1], n1);
            }
            int pex = pe;
            if (grp_max_i32<H>(max(__double2hiint(n0 * p[0]), __double2hiint(n1 * p[1]))) < (64 << 20)) {
                p[0] = ldexp(p[0], 900); // (see the first step)
                p[1] = ldexp(p[1], 900);
                pex += 900;
            }
            n0 *= p[0];
            n1 *= p[1];
            const double c = grp_sum<H>(n0 + n1);
            const double rc = fast_rcp(c);
            a[0] = n0 * rc;
            a[1] = n1 * rc;
            int e;
            P = frexp(P * c, &e);
            eP += e - pex;
            *ci_pair(ws, rec, N, q, cl) = make_double2(a[0], a[1]);
        }
        if (q == 0)
            logL_chunk[g] = log(P) + (double)eP * 0.693147180559945309417232121458;
    } else {
        double Ar[2][N], b2[2];
#pragma unroll
        for (int i = 0; i < N; ++i) {
            Ar[0][i] = m.A[(2 * q) * N + i];
            Ar[1][i] = m.A[(2 * q + 1) * N + i];
        }
        {
            const double2 x = *reinterpret_cast<const double2 *>(beta_exit + g * N + 2 * q);
            const double rS = 1.0 / grp_sum<H>(x.x + x.y);
            b2[0] = x.x * rS;
            b2[1] = x.y * rS;
        }
        *ci_pair(ws, ci_rec(g, len - 1, ch.Lmax), N, q, cl) = make_double2(b2[0], b2[1]);
        for (int s = len - 1; s >= 1; --s) {
            double p[2];
            const ObsIn cur = load_obs<N, KIND>(obs_ci, ci_rec(g, s, ch.Lmax), cl, q);
            (void)emit_pair<N, KIND>(m, cur, Bt, q, mu, is, cn, gmask, p);
            {   // p o beta in the denormal range although neither factor is: p times 2^900
                const int hb = grp_max_i32<H>(max(__double2hiint(p[0] * b2[0]), __double2hiint(p[1] * b2[1])));
                if (hb < (64 << 20)) {
                    p[0] = ldexp(p[0], 900);
                    p[1] = ldexp(p[1], 900);
                }
            }
            const double bb2[2] = {p[0] * b2[0], p[1] * b2[1]};
            double bf[N];
            grp_gather<N>(bb2, bf);
            double r0 = Ar[0][0] * bf[0], r1 = Ar[1][0] * bf[0];
#pragma unroll
            for (int j = 1; j < N; ++j) {
                r0 = fma(Ar[0][j], bf[j], r0);
                r1 = fma(Ar[1][j], bf[j], r1);
            }
            const double rc = 1.0 / grp_sum<H>(r0 + r1);
            b2[0] = r0 * rc;
            b2[1] = r1 * rc;
            *ci_pair(ws, ci_rec(g, s - 1, ch.Lmax), N, q, cl) = make_double2(b2[0], b2[1]);
        }
    }
}

// =========================================================================================
// k_spec_check: boundary consistency of a speculative E-step.  For every chunk g that is not the
// first of its trajectory: the alpha vector g started from (warm-up) against the alpha its
// predecessor ended with, and the beta vector the predecessor started its backward sweep from
// (warm-up) against the beta that g derived for that step.  Vectors are compared after
// normalisation, componentwise relative.  result[0] counts violations, result[1] holds the bit
// pattern of the largest relative deviation seen (float), for adapting the warm-up length.
// =========================================================================================
// deviation at the boundary in front of chunk g (0 where there is none)
template <int N>
__device__ __forceinline__ double spec_dev_one(const Chunks &ch, int G, int64_t g,
                                               const double *alpha_entry, const double *a_exit,
                                               const double *beta_exit, const double *b_entry)
{
    if (g >= G || ch.len[g] == 0 || ch.t0[g] == 0)
        return 0.0;
    double dev = 0.0;
    auto cmp = [&](const double *x, const double *y) { // y: reference side
        double sx = 0.0, sy = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            sx += x[j];
            sy += y[j];
        }
        if (!(sx > 0.0) || !(sy > 0.0)) {
            dev = 1.0;
            return;
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double xs = x[j] / sx, ys = y[j] / sy;
            const double d = fabs(xs - ys);
            const double r = (ys > 1e-280) ? d / ys : (d > 1e-280 ? 1.0 : 0.0);
            dev = fmax(dev, r);
        }
    };
    cmp(alpha_entry + g * N, a_exit + (g - 1) * N);
    if (beta_exit) // forward-only passes verify alpha alone
        cmp(beta_exit + (g - 1) * N, b_entry + g * N);
    return dev == dev ? dev : 1.0;
}

// one pair of atomics per wavefront: count of boundaries out of tolerance, largest deviation
__device__ __forceinline__ void spec_commit(double dev, double tol, unsigned int *result)
{
    const unsigned long long bad = __ballot(!(dev <= tol));
    float m = (float)fmin(dev, 1.0);
#pragma unroll
    for (int h = 32; h >= 1; h >>= 1)
        m = fmaxf(m, __shfl_xor(m, h, 64));
    if ((threadIdx.x & 63) == 0) {
        if (bad)
            atomicAdd(&result[0], (unsigned int)__popcll(bad));
        // (nothing to report from a wavefront of first chunks: a million short trajectories are
        // 15 625 such wavefronts, and their atomics on this one word were 0.15 ms)
        // ... nor one whose largest deviation is below what the word holds already (a plain look
        // first: hundreds of atomics on one address serialise)
        if (m > 0.f && __float_as_uint(m) > __hip_atomic_load(&result[1], __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(&result[1], __float_as_uint(m));
    }
}

template <int N>
__device__ __forceinline__ void spec_check_one(const Chunks &ch, int G, int64_t g,
                                               const double *alpha_entry, const double *a_exit,
                                               const double *beta_exit, const double *b_entry,
                                               double tol, unsigned int *result)
{
    spec_commit(spec_dev_one<N>(ch, G, g, alpha_entry, a_exit, beta_exit, b_entry), tol, result);
}

template <int N>
__global__ void k_spec_check(const Chunks ch, int G, const double *alpha_entry, const double *a_exit,
                             const double *beta_exit, const double *b_entry, double tol,
                             unsigned int *result)
{
    spec_check_one<N>(ch, G, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, alpha_entry, a_exit,
                      beta_exit, b_entry, tol, result);
}

// alpha for the next E-step's shortened warm-ups (estep_sweep.hpp: Carry): for chunk g (not the first
// of its trajectory) the stored workspace row of chunk g - 1 closest below `Wc` steps before its end
// (rows at multiples of four are stored by every instantiation), and the number of warm-up steps
// from there to the row just before chunk g.
template <int N>
__global__ void k_carry_alpha(const Chunks ch, int G, const double *ws, int Wc, double *a_out,
                              int32_t *da_out)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G)
        return;
    int d = 0;
    if (ch.len[g] > 0 && ch.t0[g] != 0) {
        const int lp = ch.len[g - 1];
        const int idx = (lp - Wc) & ~3;
        if (lp - Wc >= 0 && idx >= 0) {
            d = lp - 1 - idx;
            const double *src = ws + ci_rec(g - 1, idx, ch.Lmax) * (int64_t)(N * 64) + ((g - 1) & 63) * N;
#pragma unroll
            for (int j = 0; j < N; ++j)
                a_out[g * N + j] = src[j];
        }
    }
    da_out[g] = d;
}

// =========================================================================================
// k_logl: per-trajectory log-likelihood = sum of its chunks' logs (one wavefront per
// trajectory, fixed summation tree -> run-to-run identical).
// k_finalize: one wavefront per output entry; fixed-order sums of the per-workgroup partials
// -> packed statistics (bhmm_amd.h layout).
// =========================================================================================
__device__ __forceinline__ void logl_one(int k, const int32_t *traj_c0, const double *logL_chunk,
                                         double *logL_k, double *mirror = nullptr)
{
    const int lane = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    const int a1 = traj_c0[k + 1];
    int c = traj_c0[k] + lane;
    for (; c + 192 < a1; c += 256) {
        const double v0 = logL_chunk[c], v1 = logL_chunk[c + 64];
        const double v2 = logL_chunk[c + 128], v3 = logL_chunk[c + 192];
        s0 += v0;
        s1 += v1;
        s2 += v2;
        s3 += v3;
    }
    for (; c < a1; c += 64)
        s0 += logL_chunk[c];
    double s = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0) {
        logL_k[k] = s;
        if (mirror)
            mirror[k] = s;
    }
}

[[maybe_unused]] static __global__ __launch_bounds__(64) void k_logl(const int32_t *traj_c0, int K,
                                                    const double *logL_chunk, double *logL_k)
{
    logl_one(blockIdx.x, traj_c0, logL_chunk, logL_k);
}

// Entry e of the finalisation (one wavefront): e < S register statistics, then the discrete
// count table, then gamma_0, then the total log-likelihood -- the sum of logL_k[0..K) or, if
// logL_k is null, left to the caller (the fused tail kernel, where the per-trajectory sums are
// produced by other workgroups of the same launch).
template <int N, int KIND>
__device__ __forceinline__ void finalize_one(int e, const Model<N> &m, int K, int nblocks,
                                             const double *partials, const double *disc_partials,
                                             const double *logL_k, const double *logL_chunk, int G,
                                             const double *gamma0, double *stats,
                                             double *mirror = nullptr)
{
    using SL = StatLayout<N, KIND>;
    auto put = [&](int idx, double v) { // mirror: contiguous copy for the single D2H transfer
        stats[idx] = v;
        if (mirror)
            mirror[idx] = v;
    };
    const int n = m.nreal;
    const int lane = threadIdx.x & 63; // one wavefront per entry (k_tail: four entries per workgroup)
    const int MN = (KIND == EMIT_DISC) ? m.M * N : 0;
    // packed offsets
    const int oG0 = 1, oC = 1 + n, oSG = oC + n * n, oE = oSG + n;
    double s = 0.0;
    // column sums over many rows: four independent partial sums per lane keep four loads in flight
    // (a million short trajectories are 15 625 rows: one dependent add per load was 0.27 ms)
    auto column_sum = [&](const double *col, int64_t stride, int rows) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int b = lane;
        for (; b + 192 < rows; b += 256) {
            const double v0 = col[(int64_t)b * stride], v1 = col[(int64_t)(b + 64) * stride];
            const double v2 = col[(int64_t)(b + 128) * stride], v3 = col[(int64_t)(b + 192) * stride];
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
        }
        for (; b < rows; b += 64)
            s0 += col[(int64_t)b * stride];
        return wave_sum((s0 + s1) + (s2 + s3));
    };
    if (e < SL::S) {
        s = column_sum(partials + e, SL::S, nblocks);
        if (lane != 0)
            return;
        if (e < SL::NC) {
            const int r = e / N, c = e % N;
            if (r < n && c < n)
                put(oC + r * n + c, s * m.A[r * N + c]); // xi = A o (alpha (x) b / S)
        } else if (e < SL::NC + N) {
            const int r = e - SL::NC;
            if (r < n)
                put(oSG + r, s);
        } else {
            const int w = (e - SL::NC - N) / N, r = (e - SL::NC - N) % N;
            if (r < n)
                put(oE + w * n + r, s);
        }
        return;
    }
    e -= SL::S;
    if (e < MN) {
        const int ndisc = m.bt_global ? DISC_GLOBAL_TABLES : nblocks;
        s = column_sum(disc_partials + e, MN, ndisc);
        const int sym = e / N, r = e % N;
        if (lane == 0 && r < n)
            put(oE + r * m.M + sym, s);
        return;
    }
    e -= MN;
    if (e < N) {
        if (!logL_k)
            return; // fused tail: summed over trajectory blocks there (k_tail), not by one wavefront
        s = column_sum(gamma0 + e, N, K);
        if (lane == 0 && e < n)
            put(oG0 + e, s);
        return;
    }
    if (!logL_k)
        return; // fused tail: the total is formed by the last trajectory workgroup (k_tail)
    s = column_sum(logL_k, 1, K);
    if (lane == 0)
        put(0, s);
}

template <int N, int KIND>
__global__ __launch_bounds__(64) void k_finalize(const Model<N> m, int K, int nblocks,
                                                 const double *partials,
                                                 const double *disc_partials,
                                                 const double *logL_k, const double *gamma0,
                                                 double *stats)
{
    finalize_one<N, KIND>(blockIdx.x, m, K, nblocks, partials, disc_partials, logL_k, nullptr, 0,
                          gamma0, stats);
}

// k_tail: everything after the sweep of a speculative E-step in one launch of 64-thread
// workgroups -- [0, nfin) finalisation entries, [nfin, nfin + nTB) trajectory blocks, then one
// thread per chunk boundary of the check.  The workgroups are independent of each other.
// A trajectory block owns TAIL_TPB consecutive trajectories: their log-likelihoods (the sum of
// each one's chunk logs, same summation tree whatever the number of trajectories), and the block's
// partial sums of logL and of gamma_0.  The block that finishes last adds the partials up in block
// order (word 3 of the verdict set counts finished blocks; it is cleared with the set) -- so the
// tail costs K / 64 short workgroups and two sums of K / 64 terms, not K workgroups and sums of K
// terms by one wavefront (1e6 short trajectories: 40 ms -> < 1 ms).
// flags_next: the verdict words of the NEXT E-step (the two sets alternate), cleared here so
// that no memset sits between E-steps.
constexpr int TAIL_TPB = 64; // trajectories per block when they are short (else one: tail_tpb())
// A block walks its trajectories one after the other, each with a wave-wide sum over its chunks:
// 64 per block is right for trajectories of a few chunks, one per block for long ones (configs[1]:
// 128 chunks per trajectory -- 64 of them in sequence per block cost 80 us instead of 25).
inline int tail_tpb(int64_t G, int K) { return G >= (int64_t)16 * K ? 1 : TAIL_TPB; }
// sum of the trajectory blocks' partials -> stats[0] (total log-likelihood) and stats[1..n]
// (sum_k gamma_k[0]); one wavefront: every lane takes whole records (1 + N independent loads in
// flight per record), then one wave sum per entry -- fixed order, no chain of load latencies
// entry e of the same sums by one wavefront of its own (k_tail_total: many trajectory blocks)
template <int N>
__device__ __forceinline__ void tail_total_entry(int e, int n, int nTB, const double *tb_part,
                                                 double *stats, double *mirror)
{
    const int lane = threadIdx.x & 63;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int t = lane;
    for (; t + 192 < nTB; t += 256) {
        const double v0 = tb_part[(int64_t)t * (1 + N) + e], v1 = tb_part[(int64_t)(t + 64) * (1 + N) + e];
        const double v2 = tb_part[(int64_t)(t + 128) * (1 + N) + e], v3 = tb_part[(int64_t)(t + 192) * (1 + N) + e];
        s0 += v0;
        s1 += v1;
        s2 += v2;
        s3 += v3;
    }
    for (; t < nTB; t += 64)
        s0 += tb_part[(int64_t)t * (1 + N) + e];
    const double sacc = wave_sum((s0 + s1) + (s2 + s3));
    if (lane == 0 && (e == 0 || e - 1 < n)) {
        stats[e] = sacc; // packed offsets: [0] total log-likelihood, [1 + i] sum_k gamma_k[0][i]
        mirror[e] = sacc;
    }
}

template <int N>
__device__ __forceinline__ void tail_total(int n, int nTB, const double *tb_part, double *stats,
                                           double *mirror)
{
    const int lane = threadIdx.x & 63;
    double acc[1 + N];
#pragma unroll
    for (int e = 0; e <= N; ++e)
        acc[e] = 0.0;
    for (int t = lane; t < nTB; t += 64) {
        double v[1 + N];
#pragma unroll
        for (int e = 0; e <= N; ++e)
            v[e] = __hip_atomic_load(tb_part + (int64_t)t * (1 + N) + e, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int e = 0; e <= N; ++e)
            acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e <= N; ++e) {
        const double sacc = wave_sum(acc[e]);
        if (lane == 0) {
            if (e == 0) {
                stats[0] = sacc;
                mirror[0] = sacc;
            } else if (e - 1 < n) {
                stats[1 + (e - 1)] = sacc; // packed offset of sum_k gamma_k[0]
                mirror[1 + (e - 1)] = sacc;
            }
        }
    }
}
// the same as its own (one-wavefront) launch, for batches with many trajectory blocks: a kernel
// boundary orders it behind k_tail without a fence and a ticket per block
template <int N>
__global__ __launch_bounds__(64) void k_tail_total(int n, int nTB, const double *tb_part,
                                                   double *stats, double *mirror)
{
    tail_total_entry<N>(blockIdx.x, n, nTB, tb_part, stats, mirror); // grid: 1 + N workgroups
}

// k_fold_rows: rows [b R, (b + 1) R) of a [rows][cols] table summed into row b of `out` (fixed order),
// one thread per column -- consecutive threads read consecutive doubles.  The finalisation sums a
// column per wavefront, which is a strided walk over the whole table: fine for the few hundred rows
// of a usual batch, 0.2 ms for the 15 625 rows a million short trajectories leave behind.  Folding
// first leaves it ~120 rows.
constexpr int FOLD_ROWS = 128;
[[maybe_unused]] static __global__ __launch_bounds__(256) void k_fold_rows(const double *tab, int rows, int cols,
                                                                           double *out)
{
    const int r0 = blockIdx.x * FOLD_ROWS;
    const int r1 = r0 + FOLD_ROWS < rows ? r0 + FOLD_ROWS : rows;
    for (int e = threadIdx.x; e < cols; e += 256) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            const double v0 = tab[(int64_t)r * cols + e], v1 = tab[(int64_t)(r + 1) * cols + e];
            const double v2 = tab[(int64_t)(r + 2) * cols + e], v3 = tab[(int64_t)(r + 3) * cols + e];
            s0 += v0;
            s1 += v1;
            s2 += v2;
            s3 += v3;
        }
        for (; r < r1; ++r)
            s0 += tab[(int64_t)r * cols + e];
        out[(int64_t)blockIdx.x * cols + e] = (s0 + s1) + (s2 + s3);
    }
}

constexpr int TAIL_WAVES = 4; // wavefronts per workgroup of k_tail, each with a role of its own
template <int N, int KIND>
__global__ __launch_bounds__(64 * TAIL_WAVES) void k_tail(const Model<N> m, const Chunks ch, int K, int G,
                                             int nblocks, int nfin, const int32_t *traj_c0,
                                             const double *partials, const double *disc_partials,
                                             const double *logL_chunk, const double *gamma0,
                                             const double *alpha_entry, const double *a_exit,
                                             const double *beta_exit, const double *b_entry,
                                             double tol, double *stats, double *logL_k,
                                             double *mirror, // [S stats | K logL_k], contiguous
                                             int S, unsigned int *flags, unsigned int *flags_next,
                                             double *tb_part, // [nTB][1 + N] block partials
                                             bool fused_total, int tpb)
{
    // (workgroups of four wavefronts: a million short trajectories are 31 000 roles, and as
    // one-wavefront workgroups their dispatch alone took 0.24 ms)
    const int b = blockIdx.x * TAIL_WAVES + (threadIdx.x >> 6);
    const int nTB = (K + tpb - 1) / tpb;
    const int lane = threadIdx.x & 63;
    if (b < nfin) {
        finalize_one<N, KIND>(b, m, K, nblocks, partials, disc_partials, nullptr, logL_chunk, G,
                              gamma0, stats, mirror);
        if (b == 0 && lane < 4)
            flags_next[lane] = 0u;
    } else if (b < nfin + nTB) {
        const int tb = b - nfin;
        const int k0 = tb * tpb;
        const int kn = K - k0 < tpb ? K - k0 : tpb;
        // chunk ranges of my trajectories: lane j holds the first chunk of trajectory k0 + j
        const int c_lo = traj_c0[k0 + (lane < kn ? lane : kn)];
        const int c_hi = __shfl_down(c_lo, 1, 64);
        const int c_end = traj_c0[k0 + kn];
        const int my_hi = lane == kn - 1 ? c_end : c_hi;
        const bool single = __all(lane >= kn || my_hi - c_lo <= 1);
        double mine = 0.0; // log-likelihood of trajectory k0 + lane
        if (single) {
            if (lane < kn && my_hi > c_lo)
                mine = logL_chunk[c_lo];
        } else {
            for (int j = 0; j < kn; ++j) {
                const int a0 = __shfl(c_lo, j, 64), a1 = j == kn - 1 ? c_end : __shfl(c_lo, j + 1, 64);
                // (one trajectory of a million steps is 31 250 chunks: four loads in flight per lane,
                // a dependent add per load took 0.15 ms)
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int c = a0 + lane;
                for (; c + 192 < a1; c += 256) {
                    const double v0 = logL_chunk[c], v1 = logL_chunk[c + 64];
                    const double v2 = logL_chunk[c + 128], v3 = logL_chunk[c + 192];
                    s0 += v0;
                    s1 += v1;
                    s2 += v2;
                    s3 += v3;
                }
                for (; c < a1; c += 64)
                    s0 += logL_chunk[c];
                double sacc = wave_sum((s0 + s1) + (s2 + s3));
                if (lane == j)
                    mine = sacc;
            }
        }
        if (lane < kn) {
            logL_k[k0 + lane] = mine;
            mirror[S + k0 + lane] = mine;
        }
        const double bsum = wave_sum(lane < kn ? mine : 0.0);
        double *mypart = tb_part + (int64_t)tb * (1 + N);
        if (lane == 0)
            mypart[0] = bsum;
#pragma unroll
        for (int e = 0; e < N; ++e) {
            const double g = wave_sum(lane < kn ? gamma0[(int64_t)(k0 + lane) * N + e] : 0.0);
            if (lane == 0)
                mypart[1 + e] = g;
        }
        if (!fused_total)
            return; // many trajectory blocks: k_tail_total adds the partials up after this launch
        __threadfence();
        unsigned int ticket = 0;
        if (lane == 0)
            ticket = atomicAdd(&flags[3], 1u);
        ticket = __shfl(ticket, 0, 64);
        if (ticket == (unsigned int)(nTB - 1)) {
            __threadfence();
            tail_total<N>(m.nreal, nTB, tb_part, stats, mirror);
        }
    } else {
        // (roles beyond the last boundary: spec_dev_one ignores g >= G)
        spec_check_one<N>(ch, G, (int64_t)(b - nfin - nTB) * 64 + lane, alpha_entry, a_exit,
                          beta_exit, b_entry, tol, flags);
    }
}

// =========================================================================================
// k_forget_probe: how fast does the filter forget under THIS model on THIS data?  For a sample
// of positions the forward (dir 0) and backward (dir 1) recursions are run over the same stretch
// of observations from two different start vectors (uniform / all mass on one state), and
// curve[dir][w] receives the largest componentwise relative deviation between the two normalised
// vectors after w + 1 steps (float bits; maximum over the samples).  The host reads the warm-up
// length of the speculative boundaries off this curve (calibrate_warmup in bhmm_amd.hip) instead
// of guessing it; the boundary check still verifies every E-step.  One thread per (sample, dir),
// plain per-thread arithmetic -- this runs once per data set, not per E-step.
// =========================================================================================
template <int N, int KIND>
__global__ __launch_bounds__(64) void k_forget_probe(const Model<N> m, const void *obs_rm,
                                                     const double *Bt_g, const int64_t *starts,
                                                     int S, int Wmax, unsigned int *curve)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    const bool act = id < 2 * S;
    const int dir = act ? id / S : 0, idx = act ? id % S : 0;
    const int n = m.nreal;
    const int64_t pos0 = starts[idx];
    double x[N], y[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        x[j] = j < n ? 1.0 / (double)n : 0.0;
        y[j] = (j == idx % n) ? 1.0 : 0.0;
    }
    for (int w = 0; w < Wmax; ++w) {
        const int64_t t = dir == 0 ? pos0 + w : pos0 + Wmax - 1 - w;
        double p[N];
        if constexpr (KIND == EMIT_GAUSS) {
            const double o = static_cast<const double *>(obs_rm)[t];
            double mx = 0.0;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double z = (o - m.e0[j]) * m.e1[j];
                p[j] = j < n ? m.e2[j] * exp(-0.5 * z * z) : 0.0;
                mx = fmax(mx, p[j]);
            }
            if (mx == 0.0) {
#pragma unroll
                for (int j = 0; j < N; ++j)
                    p[j] = j < n ? 1.0 : 0.0;
            }
        } else if constexpr (KIND == EMIT_DISC) {
            const int sym = static_cast<const int32_t *>(obs_rm)[t];
#pragma unroll
            for (int j = 0; j < N; ++j)
                p[j] = Bt_g[(int64_t)sym * N + j];
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j)
                p[j] = j < n ? static_cast<const double *>(obs_rm)[t * n + j] : 0.0;
        }
        auto step = [&](double (&v)[N]) {
            double r[N], sum = 0.0;
            if (dir == 0) { // (v A) o p
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    double a = 0.0;
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        a = fma(v[i], m.A[i * N + j], a);
                    r[j] = a * p[j];
                }
            } else { // A (p o v)
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    double a = 0.0;
#pragma unroll
                    for (int j = 0; j < N; ++j)
                        a = fma(m.A[i * N + j], p[j] * v[j], a);
                    r[i] = i < n ? a : 0.0;
                }
            }
#pragma unroll
            for (int j = 0; j < N; ++j)
                sum += r[j];
            const double rs = 1.0 / sum;
#pragma unroll
            for (int j = 0; j < N; ++j)
                v[j] = r[j] * rs;
        };
        step(x);
        step(y);
        double dev = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double d = fabs(x[j] - y[j]), lo = fmin(x[j], y[j]);
            dev = fmax(dev, lo > 1e-280 ? d / lo : (d > 1e-280 ? 1.0 : 0.0));
        }
        if (!(dev == dev))
            dev = 1.0;
        dev = fmin(dev, 1.0);
        if (act)
            atomicMax(&curve[dir * Wmax + w], __float_as_uint((float)dev));
        if (__all(dev < 1e-14)) // all chains of the wavefront have merged (target: 1e-13)
            break;
    }
}

// =========================================================================================
// layout conversion kernels (one lane per chunk; CI side is coalesced)
// =========================================================================================
// has_nan (gaussian observations): set to 1 if some observation is NaN -- the context then stays on
// the CAREFUL kernels, whose densities keep NaN apart from a perfect hit (gauss_pdf, estep_sweep.hpp)
template <typename T>
__global__ void k_pack_scalar(const Chunks ch, const T *src, T *dst_ci, int32_t *has_nan)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int len = ch.len[g];
    const int64_t off = ch.goff[g];
    bool nan = false;
    for (int s = 0; s < len; ++s) {
        const T v = src[off + s];
        nan |= v != v;
        dst_ci[ci_rec(g, s, ch.Lmax) * 64 + lane] = v;
    }
    if (nan)
        *has_nan = 1;
}

// rows of nreal doubles (row-major) -> CI records of N doubles (zero padded)
template <int N>
__global__ void k_pack_rows(const Chunks ch, const double *src, int nreal, double *dst_ci)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int len = ch.len[g];
    const int64_t off = ch.goff[g];
    for (int s = 0; s < len; ++s) {
        double v[N];
#pragma unroll
        for (int i = 0; i < N; ++i)
            v[i] = (i < nreal) ? src[(off + s) * nreal + i] : 0.0;
        ci_store<N>(dst_ci, ci_rec(g, s, ch.Lmax), lane, v);
    }
}

template <int N>
__global__ void k_unpack_rows(const Chunks ch, const double *src_ci, int nreal, double *dst,
                              int only_traj, int64_t dst_shift)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const int len = ch.len[g];
    if (only_traj >= 0 && ch.traj[g] != only_traj)
        return;
    const int64_t off = ch.goff[g] - dst_shift;
    for (int s = 0; s < len; ++s) {
        double v[N];
        ci_load<N>(src_ci, ci_rec(g, s, ch.Lmax), lane, v);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < nreal)
                dst[(off + s) * nreal + i] = v[i];
    }
}

} // namespace bhmm
