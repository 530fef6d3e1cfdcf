// tile_kernels.hpp -- row-batched E-step recursions on the fp64 matrix cores, 64 states and more.
//
// Reference loops covered: bhmm/hidden/impl_c/_hidden.c:42-63 (forward), :91-109 (backward),
// :148-183 (xi counts), hidden/api.py:176-186 (gamma), with the emission rows of
// output_models/impl_c/_gaussian.c:5-21 / discrete.py:130-157 fused in.
//
// The 9..64-state kernels of wide_kernels.hpp run one trajectory segment per wavefront: a
// matrix-VECTOR product per step on DPP FMAs, one or two wavefronts per SIMD, about 690 cycles per
// step against the 256 pipe cycles its 4096 FMAs need.  Here SIXTEEN segments ("rows") form a tile
// and one workgroup of four wavefronts advances the whole tile by one step with
//
//     alpha-tile [16 x N] . A [N x N]      on v_mfma_f64_16x16x4_f64
//
// Output column tile c (16 states) belongs to wavefront c mod 4; its block of A (all N rows, 16
// columns) stays in that wavefront's registers as the B operand; the tile of the previous step is
// all-gathered through LDS into the A-operand layout (lane (m, k) <- row m, state 4 kk + k).  The
// emission row and the power-of-two rescaling are elementwise on the C/D layout
// (lane (s, q), register r  <->  row q + 4 r, state 16 c + s).  The backward kernel does the same
// with A transposed and adds the xi counts as a second group of matrix instructions per step:
// C'[16 I.., 16 J..] += (alpha_{t-1} / S)[rows, 16 I..]^T (p_t o beta_t)[rows, 16 J..], K = the tile's
// rows -- the C/D-layout registers of alpha ARE the A operand, the LDS copy of p o beta the B operand.
//
// Scaling.  alpha and beta are carried up to powers of two (refreshed every fourth step from the
// row maxima the wavefronts exchange one step earlier); the normaliser of gamma and xi is the same
// bilinear form at every step of a segment, sum_j alpha_t[j] beta_t[j] = S0 2^(exponents removed
// since), so it is summed across the four wavefronts ONCE per segment (and once more for the
// transition into the segment, whose alpha row belongs to the neighbour's chain).  The forward pass
// leaves the exponent it removed at step t in exps[t].  Rows that leave the range this covers
// (a vector below 2^-900, a denormal normaliser, a gamma mass that is off) raise flags[2] and the
// host repeats the E-step with the per-step-normalising kernels (wide_kernels.hpp /
// gen_kernels.hpp) -- exactly the contract of the lazily scaled 9..64-state kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wide_kernels.hpp"

namespace bhmm {

#ifndef TILE_PF
#define TILE_PF 4 // steps of load-ahead (a tile step takes about half a microsecond)
#endif
static_assert(TILE_PF % 4 == 0, "the rescaling phase is the position inside the unrolled group");

template <int NT>
struct TileGeo {
    static constexpr int NP = 16 * NT;                   // padded state count
    static constexpr int TPW = (NT + 3) / 4;             // column tiles per wavefront
    static constexpr int KK = NP / 4;                    // K steps of one product
    static constexpr int PX = (NP + 31) / 32 * 32 + 2;   // pitch of a tile row in LDS: == 2 (mod 32)
};

// physical LDS row of tile row rho = q + 4 r.  Rows q and q + 1 of one register index lie eight
// apart, so that both read patterns are conflict-free with the pitch above: the matvec operand
// (lanes (m, k): row m, column 4 kk + k) and the xi operand (lanes (s, q): row q + 4 r, column 16 J + s).
__device__ __forceinline__ int tile_prow(int rho)
{
    const int q = rho & 3, r = rho >> 2;
    return 8 * (q & 1) + 4 * (q >> 1) + r;
}

__device__ __forceinline__ int row16_max_i32(int v)
{
    v = max(v, dpp_i32<0xB1>(v));
    v = max(v, dpp_i32<0x4E>(v));
    v = max(v, dpp_i32<0x141>(v)); // row_half_mirror
    v = max(v, dpp_i32<0x140>(v)); // row_mirror
    return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}
// sum over the 16 lanes of a row, result in every lane (fixed order)
__device__ __forceinline__ double row16_sum(double v)
{
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return v;
}

// which segment sits in which tile row: tile_seg[16 * tile + rho] (-1: empty)
struct TilePlan {
    const int32_t *tile_seg;
    int ntiles;
};

// what one step reads for my four rows
template <int KIND, int TPW>
struct TileIn {
    double o[KIND == EMIT_GAUSS ? 4 : 1];
    int sym[KIND == EMIT_DISC ? 4 : 1];
    double p[KIND == EMIT_EXPL ? TPW : 1][4];
};

// emission probability of state (tile ct, lane s) for row r of this lane
template <int KIND, int TPW>
__device__ __forceinline__ double tile_emit(const WideModel &m, const TileIn<KIND, TPW> &in, int c, int r,
                                            int j, bool real, double mu_j, double ga_j, double gb_j)
{
    if constexpr (KIND == EMIT_GAUSS)
        return gauss_pdf_issue(in.o[r] - mu_j, ga_j, gb_j, m.gmg);
    else if constexpr (KIND == EMIT_DISC)
        return real ? m.B[(int64_t)j * m.M + in.sym[r]] : 0.0;
    else
        return in.p[c][r];
}

// Four densities of ONE state at four observations (the four tile rows of a lane): gauss_pdf_issue with
// the four Horner chains side by side, so that no fused multiply-add waits for its predecessor.
__device__ __forceinline__ void gauss_pdf4_issue(const double (&d)[4], double a, double b, double MG,
                                                 double (&p)[4])
{
    double w[4], q[4];
    int tl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double u = fmin(fma(d[i] * d[i], a, b), 1.0);
        const double t = MG - u;
        w[i] = (t - MG) + u;
        tl[i] = __double2loint(t);
        q[i] = 0x1.e3991e644e6abp+92;
    }
    constexpr double C[10] = {-0x1.b6740fc28f781p+84, 0x1.62c157ee59177p+76, -0x1.ffcb55e82f22cp+67,
                              0x1.4309126056718p+59,  -0x1.5d87fe9cc5d6fp+50, 0x1.3b2ab6fbde0f7p+41,
                              -0x1.c6b08d703d48ap+31, 0x1.ebfbdff82c3b9p+21,  -0x1.62e42fefa3a17p+11, 1.0};
#pragma unroll
    for (int k = 0; k < 10; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            q[i] = __builtin_fma(q[i], w[i], C[k]);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        p[i] = ldexp(q[i], tl[i]);
}

// emission probabilities of my states (tiles c) for my four rows
template <int NT, int KIND, int TPW>
__device__ __forceinline__ void tile_emit4(const WideModel &m, const TileIn<KIND, TPW> &in, int w, int s,
                                           const bool (&real)[TPW], const double (&mu_j)[TPW],
                                           const double (&ga_j)[TPW], const double (&gb_j)[TPW],
                                           double (&p)[TPW][4])
{
#ifdef TILE_X_NOEMIT
#pragma unroll
    for (int c = 0; c < TPW; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            p[c][r] = real[c] ? 0.05 : 0.0;
    return;
#endif
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        if constexpr (KIND == EMIT_GAUSS) {
            double d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                d[r] = in.o[r] - mu_j[c];
            gauss_pdf4_issue(d, ga_j[c], gb_j[c], m.gmg, p[c]); // (lanes without a state: a = 0, b = 1 -> 0)
        } else if constexpr (KIND == EMIT_DISC) {
            const int j = 16 * (w + 4 * c) + s;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p[c][r] = real[c] ? m.B[(int64_t)j * m.M + in.sym[r]] : 0.0;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p[c][r] = in.p[c][r];
        }
    }
}

// A tile advances in groups of four steps (the rescaling phase).  Most groups are uniform over the
// tile's 16 rows -- every row still in its warm-up, or every row in its main part -- and run without
// any per-row predicate; the groups around segment entries and exits take the general path.
enum { TM_WARM = 0, TM_MAIN = 1, TM_GEN = 2 };

template <int V>
using tile_ic = std::integral_constant<int, V>;
typedef double tile_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int tile_all_min(int v)
{
    v = min(v, __shfl_xor(v, 16, 64));
    return min(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ int tile_all_max(int v)
{
    v = max(v, __shfl_xor(v, 16, 64));
    return max(v, __shfl_xor(v, 32, 64));
}

// Workgroup = EIGHT wavefronts on one tile (SPLIT).  Measured on gfx950 (tools/proto/step_lat.hip,
// tile_proto.hip, the in-kernel probe): the matrix instructions of a step are a dependent chain
// (1024 cycles at 64 states) that OCCUPIES THE SIMD'S VECTOR UNIT -- no other vector instruction of
// either wavefront of the SIMD issues meanwhile --, 16-byte LDS reads feed it at 1390 cycles per step
// where 8-byte reads need 1830, and four global stores per lane cost it another 220.  So the roles
// are split:
//   wavefronts 0-3 ("matrix"): operand reads, matrix instructions, times the emission row, LDS write;
//   wavefronts 4-7 ("stream"): the emission rows one step ahead (into LDS), the observation
//       stream (read once per tile, passed on through LDS), alpha rows between the LDS tile and HBM
//       in 16-byte pieces, exponent bookkeeping -- wavefront 4 + w shares the SIMD of matrix
//       wavefront w and works while that one waits for the exchange of the step.
// One barrier per step for all eight.  The K index of the products is (q, kk) <-> state q * KK + kk,
// so that the operand of lane (m, q) is 8 * KK consecutive bytes of row m of the tile.
// !SPLIT (more than 64 states: two column tiles per wavefront, whose blocks of A need half of a
// 512-register budget): four wavefronts that take both roles in turn.
constexpr int TILE_THREADS = 512;
template <bool SPLIT>
constexpr int tile_threads() { return SPLIT ? TILE_THREADS : TILE_THREADS / 2; }

// emission rows handed from the stream wavefronts to the matrix wavefronts: [slot][w][c][half][lane][2]
template <int TPW>
__device__ __forceinline__ int tile_p_index(int slot, int w, int c, int h, int lane)
{
    return ((((slot * 4 + w) * TPW + c) * 2 + h) * 64 + lane) * 2;
}

// =========================================================================================
// k_tile_fwd: alpha rows (row-major, up to a power of two per row) for the main part of every
// segment, the exponents removed (exps[global step], eP_seg[segment]), the vectors at the segment
// entry (after the warm-up) and exit for the boundary check and the log-likelihood.
// FULL: n == 16 NT (no padded states).
// =========================================================================================
template <int NT, int KIND, bool FULL, bool SPLIT>
__global__ __launch_bounds__(tile_threads<SPLIT>()) void k_tile_fwd(const WideModel m, const int64_t *off, const Segs sg,
                                                                    const TilePlan tp, const void *obs_rm,
                                                                    double *alpha_rm, int32_t *exps, int32_t *eP_seg,
                                                                    double *a_entry, double *a_exit, unsigned int *flags,
                                                                    unsigned long long *probe = nullptr)
{
#ifndef BHMM_TILE_PROBE_BUILD
    // (production build: the in-kernel cycle probe compiles away -- its run-time tests cost every wavefront some
    // fifteen instruction slots per step even when no probe buffer is given; -DBHMM_TILE_PROBE_BUILD brings it back)
    probe = nullptr;
#endif
    using G = TileGeo<NT>;
    constexpr int TPW = G::TPW, KK = G::KK, PX = G::PX, NP = G::NP;
    __shared__ __attribute__((aligned(16))) double sX[2 * 16 * PX];
    __shared__ __attribute__((aligned(16))) double sP[2 * 4 * TPW * 2 * 64 * 2];
    __shared__ int sE[64];
    __shared__ __attribute__((aligned(16))) double sObs[16 * 16]; // observations of 16 steps: [step & 15][4 q + r]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool matrix = !SPLIT || wid < 4, stream = !SPLIT || wid >= 4;
    const int w = wid & 3;
    const int s = lane & 15, q = lane >> 4;
    const int n = FULL ? NP : m.n;

    // ---- my four rows (both roles: lane (s, q), register r <-> row q + 4 r) --------------------
    int nst[4], r0[4], nlast[4];
    int64_t ob[4]; // global step index of the row's first (warm-up) step
    bool fs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int sgi = tp.tile_seg[(int64_t)blockIdx.x * 16 + q + 4 * r];
        nst[r] = 0;
        r0[r] = 0;
        ob[r] = 0;
        fs[r] = false;
        if (sgi >= 0 && sg.len[sgi] > 0) {
            const int64_t o0 = off[sg.traj[sgi]], t0 = sg.t0[sgi], t1 = t0 + sg.len[sgi];
            const int64_t tw = (t0 - sg.W > 0) ? t0 - sg.W : 0;
            nst[r] = (int)(t1 - tw);
            r0[r] = (int)(t0 - tw);
            ob[r] = o0 + tw;
            fs[r] = tw == 0;
        }
        nlast[r] = nst[r] > 0 ? nst[r] - 1 : 0;
    }
    // (every wavefront holds all 16 rows: these are uniform over the workgroup)
    // (uniform over the workgroup by construction -- said to the compiler, so that the loop and the tests on these
    // bounds are scalar instructions instead of vector compares and exec-mask branches)
    const int nmax = __builtin_amdgcn_readfirstlane(tile_all_max(max(max(nst[0], nst[1]), max(nst[2], nst[3]))));
    const int g4 = (nmax + 3) & ~3;
    // steps [g2, g3): every row of the tile is inside its main part
    const int g2 = __builtin_amdgcn_readfirstlane(tile_all_max(max(max(r0[0], r0[1]), max(r0[2], r0[3]))));
    const int g3 = __builtin_amdgcn_readfirstlane(tile_all_min(min(min(nst[0], nst[1]), min(nst[2], nst[3])))) - 1;

    for (int e = tid; e < 16 * PX; e += tile_threads<SPLIT>())
        sX[e] = (e % PX) < n ? 1.0 / (double)n : 0.0; // warm-ups start from the uniform vector
    bool real[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c)
        real[c] = (w + 4 * c < NT) && (FULL || 16 * (w + 4 * c) + s < n);
    // groups of four steps; the matrix part only distinguishes the first step (rows that start their
    // trajectory take pi o p_0 instead of the product)
    auto run = [&](auto &&step) __attribute__((always_inline)) {
        int rs = 0;
        if (g4 >= 4) {
            step(0, tile_ic<0>{}, tile_ic<TM_GEN>{});
            step(1, tile_ic<1>{}, tile_ic<TM_MAIN>{});
            step(2, tile_ic<2>{}, tile_ic<TM_MAIN>{});
            step(3, tile_ic<3>{}, tile_ic<TM_MAIN>{});
            rs = 4;
        }
        for (; rs + 4 <= g4; rs += 4) {
            step(rs, tile_ic<0>{}, tile_ic<TM_MAIN>{});
            step(rs + 1, tile_ic<1>{}, tile_ic<TM_MAIN>{});
            step(rs + 2, tile_ic<2>{}, tile_ic<TM_MAIN>{});
            step(rs + 3, tile_ic<3>{}, tile_ic<TM_MAIN>{});
        }
    };

    // ================= the matrix part ==========================================================
    double Breg[TPW * KK], pi_j[TPW]; // my blocks of A (B operand)
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int j = 16 * (w + 4 * c) + s;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int i = q * KK + kk;
            Breg[c * KK + kk] = (matrix && real[c] && (FULL || i < n)) ? m.A[(int64_t)i * n + j] : 0.0;
        }
        pi_j[c] = real[c] ? m.pi[j] : 0.0;
    }
    int xw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        xw[r] = tile_prow(q + 4 * r) * PX;
    const int xr = tile_prow(s) * PX + q * KK; // my operand: KK consecutive doubles of row s
    unsigned long long pc = 0; // (probe: end of the previous step's work)
    auto m_step = [&](int rs, auto uc, auto mc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, MODE = decltype(mc)::value;
        const bool pr = probe && blockIdx.x == gridDim.x - 1 && wid == 0;
        const unsigned long long c0 = pr ? __builtin_readcyclecounter() : 0;
        const double *X = sX + (u & 1) * 16 * PX; // (groups of four steps: the buffer is the step's parity)
        double *Xn = sX + ((u & 1) ^ 1) * 16 * PX;
        tile_d2 pl[TPW][2];
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            pl[c][0] = *reinterpret_cast<const tile_d2 *>(&sP[tile_p_index<TPW>(u & 1, w, c, 0, lane)]);
            pl[c][1] = *reinterpret_cast<const tile_d2 *>(&sP[tile_p_index<TPW>(u & 1, w, c, 1, lane)]);
        }
        wide_d4 acc[TPW];
        // (operands in pieces of eight: all of them first is no faster and costs the registers)
        constexpr int CH = KK % 8 == 0 ? 8 : (KK % 4 == 0 ? 4 : 2);
#pragma unroll
        for (int k0 = 0; k0 < KK; k0 += CH) {
            tile_d2 av[CH / 2];
#pragma unroll
            for (int k2 = 0; k2 < CH / 2; ++k2)
                av[k2] = *reinterpret_cast<const tile_d2 *>(X + xr + k0 + 2 * k2);
#pragma unroll
            for (int kk = k0; kk < k0 + CH; ++kk)
#pragma unroll
                for (int c = 0; c < TPW; ++c)
                    // (a column tile beyond NT: its block of A is zero, the product is computed all the same -- the
                    // wavefronts with two real tiles set the pace of the step, and a chain without branches keeps the
                    // accumulators where they are: with the branch, 770 register moves per step at NT = 6)
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[(kk - k0) >> 1][(kk - k0) & 1], Breg[c * KK + kk],
                                                                  kk == 0 ? wide_d4{0.0, 0.0, 0.0, 0.0} : acc[c], 0, 0, 0);
        }
        unsigned long long c1 = 0;
        if (pr) {
            asm volatile("" ::"v"(acc[0][0]));
            c1 = __builtin_readcyclecounter();
        }
        // the exponent this step removes: row maxima of the step before, over the four wavefronts
        int E[4] = {0, 0, 0, 0};
        if constexpr (u == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rho = q + 4 * r;
                E[r] = max(max(sE[rho], sE[16 + rho]), max(sE[32 + rho], sE[48 + rho]));
            }
        }
        int pm[4] = {-(1 << 28), -(1 << 28), -(1 << 28), -(1 << 28)};
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            if (NT % 4 == 0 || w + 4 * c < NT) {
                const int j = 16 * (w + 4 * c) + s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double p = pl[c][r >> 1][r & 1];
                    double v = acc[c][r] * p;
                    if constexpr (MODE == TM_GEN)
                        if (fs[r] && rs == 0)
                            v = pi_j[c] * p;
                    if constexpr (u == 3)
                        v = ldexp(v, -E[r]);
                    Xn[xw[r] + j] = v;
                    if constexpr (u == 2)
                        pm[r] = max(pm[r], v > 0.0 ? exponent_of(v) : -(1 << 28));
                }
            }
        }
        if constexpr (u == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mx = row16_max_i32(pm[r]);
                if (s == 0)
                    sE[16 * w + q + 4 * r] = mx;
            }
        }
        if (pr && lane == 0) {
            const unsigned long long c2 = __builtin_readcyclecounter();
            probe[0] += c1 - c0;
            probe[1] += c2 - c1;
            probe[2] += pc ? c0 - pc : 0;
            probe[3] += 1;
            pc = c2;
        }
    };

    // ================= the stream part ==========================================================
    double mu_j[TPW], ga_j[TPW], gb_j[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int j = 16 * (w + 4 * c) + s;
        mu_j[c] = (KIND == EMIT_GAUSS && stream && real[c]) ? m.mu[j] : 0.0;
        ga_j[c] = (KIND == EMIT_GAUSS && stream && real[c]) ? m.ga[j] : 0.0;
        gb_j[c] = (KIND == EMIT_GAUSS && stream && real[c]) ? m.gb[j] : 1.0;
    }
    // the row whose alpha I carry to HBM: 16 lanes per row, NP / 16 states each -- FULL: consecutive ones;
    // otherwise pairs, pair e / 2 at 2 (lane & 15) + 16 e (sixteen lanes on 256 consecutive bytes, one 16-byte
    // piece per pair when n is even, see k_tile_bwd)
    constexpr int SPL = FULL ? NP / 16 : 2 * ((NP + 31) / 32); // (pairs: up to the row's pitch, never beyond it)
    const int srow = (w * 64 + lane) >> 4, sch = ((w * 64 + lane) & 15) * (FULL ? SPL : 2);
    auto spos = [&](int e) __attribute__((always_inline)) { return FULL ? e : 16 * e; }; // (e even)
    const bool n_even = FULL || (n & 1) == 0;
    int s_seg = -1, s_nst = 0, s_r0 = 0;
    int64_t s_ob = 0;
    // the observation stream: read ONCE per tile -- wavefront 4 loads, per group of four steps, one
    // value per (row, step), lane = 4 row + step, and passes them on through LDS; everybody reads
    // the four rows of a step with two 16-byte reads
    const int lrow = lane >> 2, ldt = lane & 3;
    int64_t l_ob = 0;
    int l_last = 0;
    if (stream) {
        const int sgi = tp.tile_seg[(int64_t)blockIdx.x * 16 + srow];
        s_seg = sgi;
        if (sgi >= 0 && sg.len[sgi] > 0) {
            const int64_t o0 = off[sg.traj[sgi]], t0 = sg.t0[sgi], t1 = t0 + sg.len[sgi];
            const int64_t tw = (t0 - sg.W > 0) ? t0 - sg.W : 0;
            s_nst = (int)(t1 - tw);
            s_r0 = (int)(t0 - tw);
            s_ob = o0 + tw;
        }
        if (w == 0) {
            const int lgi = tp.tile_seg[(int64_t)blockIdx.x * 16 + lrow];
            if (lgi >= 0 && sg.len[lgi] > 0) {
                const int64_t o0 = off[sg.traj[lgi]], t0 = sg.t0[lgi], t1 = t0 + sg.len[lgi];
                const int64_t tw = (t0 - sg.W > 0) ? t0 - sg.W : 0;
                l_ob = o0 + tw;
                l_last = (int)(t1 - tw) - 1;
            }
        }
    }
    const int sxr = tile_prow(srow) * PX + sch;
    const int lpos = 4 * (lrow & 3) + (lrow >> 2); // row q + 4 r sits at position 4 q + r
    auto obs_load = [&](int step) __attribute__((always_inline)) -> double {
        const int64_t g = l_ob + min(step, l_last);
        if constexpr (KIND == EMIT_DISC)
            return __hiloint2double(0, static_cast<const int32_t *>(obs_rm)[g]);
        else
            return static_cast<const double *>(obs_rm)[g];
    };
    double pend = 0.0; // the group two ahead, on its way
    // what the emission row of step rs is computed from
    auto fetch_in = [&](TileIn<KIND, TPW> &in, int rs) __attribute__((always_inline)) {
        if constexpr (KIND == EMIT_GAUSS) {
            const tile_d2 lo = *reinterpret_cast<const tile_d2 *>(&sObs[(rs & 15) * 16 + 4 * q]);
            const tile_d2 hi = *reinterpret_cast<const tile_d2 *>(&sObs[(rs & 15) * 16 + 4 * q + 2]);
            in.o[0] = lo[0];
            in.o[1] = lo[1];
            in.o[2] = hi[0];
            in.o[3] = hi[1];
        } else if constexpr (KIND == EMIT_DISC) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                in.sym[r] = __double2loint(sObs[(rs & 15) * 16 + 4 * q + r]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = min(rs, nlast[r]);
#pragma unroll
                for (int c = 0; c < TPW; ++c)
                    in.p[c][r] = real[c] ? static_cast<const double *>(obs_rm)[(ob[r] + rr) * n + 16 * (w + 4 * c) + s] : 0.0;
            }
        }
    };
    // emission row of a step into slot (step & 1)
    auto emit_to_lds = [&](const TileIn<KIND, TPW> &in, int slot) __attribute__((always_inline)) {
        double p[TPW][4];
        tile_emit4<NT, KIND, TPW>(m, in, w, s, real, mu_j, ga_j, gb_j, p);
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            *reinterpret_cast<tile_d2 *>(&sP[tile_p_index<TPW>(slot, w, c, 0, lane)]) = tile_d2{p[c][0], p[c][1]};
            *reinterpret_cast<tile_d2 *>(&sP[tile_p_index<TPW>(slot, w, c, 1, lane)]) = tile_d2{p[c][2], p[c][3]};
        }
    };
    // alpha of step rs (in LDS buffer (rs + 1) & 1 after that step's barrier) to HBM
    const int64_t s_abase = s_ob * n + sch;
    auto store_row = [&](int rs) __attribute__((always_inline)) {
#ifdef TILE_X_NOSTORE
        return;
#endif
        const double *X = sX + ((rs + 1) & 1) * 16 * PX + sxr;
        auto put = [&](double *dst) __attribute__((always_inline)) {
            if (n_even) {
#pragma unroll
                for (int e = 0; e < SPL; e += 2)
                    if (FULL || sch + spos(e) < n)
                        *reinterpret_cast<tile_d2 *>(dst + spos(e)) = *reinterpret_cast<const tile_d2 *>(X + spos(e));
            } else {
#pragma unroll
                for (int e = 0; e < SPL; e += 2) {
                    const tile_d2 x2 = *reinterpret_cast<const tile_d2 *>(X + spos(e));
                    if (sch + spos(e) < n)
                        dst[spos(e)] = x2[0];
                    if (sch + spos(e) + 1 < n)
                        dst[spos(e) + 1] = x2[1];
                }
            }
        };
        if (rs >= g2 && rs < g3) { // (uniform: every row of the tile in its main part)
            put(alpha_rm + s_abase + (int64_t)rs * n);
            return;
        }
        if (rs < 0 || rs >= s_nst)
            return;
        if constexpr (FULL) { // (8-byte pieces outside the uniform groups: measured 11 % faster for the whole kernel
                              // at 64 states than the 16-byte form below -- profiles/r05)
            double *dst = nullptr;
            if (rs >= s_r0)
                dst = alpha_rm + s_abase + (int64_t)rs * n;
            else if (rs == s_r0 - 1)
                dst = a_entry + (int64_t)s_seg * n + sch;
            if (dst) {
#pragma unroll
                for (int e = 0; e < SPL; ++e)
                    dst[e] = X[e];
            }
            if (rs == s_nst - 1) {
                double *dx = a_exit + (int64_t)s_seg * n + sch;
#pragma unroll
                for (int e = 0; e < SPL; ++e)
                    dx[e] = X[e];
            }
        } else {
            if (rs >= s_r0)
                put(alpha_rm + s_abase + (int64_t)rs * n);
            else if (rs == s_r0 - 1)
                put(a_entry + (int64_t)s_seg * n + sch);
            if (rs == s_nst - 1)
                put(a_exit + (int64_t)s_seg * n + sch);
        }
    };
    int eP[4] = {0, 0, 0, 0};
    unsigned int trouble = 0u; // (bit 0: a vector below 2^-900)
    auto s_step = [&](int rs, auto uc, auto) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        const bool pr = probe && blockIdx.x == gridDim.x - 1 && wid == (SPLIT ? 4 : 0);
        const unsigned long long c0 = pr ? __builtin_readcyclecounter() : 0;
        // alpha of the previous step: LDS -> HBM
        store_row(rs - 1);
        if constexpr (KIND != EMIT_EXPL && u == 0) {
            if (w == 0) { // the observations of the group two ahead go to LDS, the next ones are fetched
                sObs[((rs + 8 + ldt) & 15) * 16 + lpos] = pend;
                pend = obs_load(rs + 12 + ldt);
            }
        }
        // bookkeeping of the exponents (wavefront 5, one lane per row)
        if constexpr (u == 3) {
            if (w == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rho = q + 4 * r;
                    const int E = max(max(sE[rho], sE[16 + rho]), max(sE[32 + rho], sE[48 + rho]));
                    const bool act = rs < nst[r];
                    trouble |= (act && E < WIDE_TROUBLE_EXP) ? 1u : 0u;
                    if (act && rs >= r0[r]) {
                        eP[r] += E;
                        if (s == 0)
                            exps[ob[r] + rs] = E;
                    }
                }
            }
        }
        const unsigned long long c1 = pr ? __builtin_readcyclecounter() : 0;
        // the emission row of the next step
        TileIn<KIND, TPW> ein;
        fetch_in(ein, rs + 1);
        emit_to_lds(ein, (u + 1) & 1);
        if (pr && lane == 0) {
            const unsigned long long c2 = __builtin_readcyclecounter();
            probe[4] += c1 - c0;
            probe[5] += c2 - c1;
            probe[7] += 1;
        }
    };
    auto s_prologue1 = [&]() __attribute__((always_inline)) {
        if constexpr (KIND != EMIT_EXPL) {
            if (w == 0) {
                sObs[ldt * 16 + lpos] = obs_load(ldt);
                sObs[(4 + ldt) * 16 + lpos] = obs_load(4 + ldt);
                pend = obs_load(8 + ldt);
            }
        }
    };
    auto s_prologue2 = [&]() __attribute__((always_inline)) {
        TileIn<KIND, TPW> in0;
        fetch_in(in0, 0);
        emit_to_lds(in0, 0);
    };
    auto s_epilogue = [&]() __attribute__((always_inline)) {
        store_row(g4 - 1);
        if (w == 1 && s == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int sgi = tp.tile_seg[(int64_t)blockIdx.x * 16 + q + 4 * r];
                if (sgi >= 0)
                    eP_seg[sgi] = eP[r];
            }
        }
        if (trouble)
            atomicOr(&flags[2], trouble);
    };

    // ================= the loop ===================================================================
    if constexpr (SPLIT) {
        if (matrix) {
            // (the serial chain runs here: its instructions go first whenever both wavefronts of a
            // SIMD have one ready)
            __builtin_amdgcn_s_setprio(3);
            __syncthreads(); // (the stream wavefronts: first observations in LDS ...
            __syncthreads(); //  ... first emission row in LDS)
            run([&](int rs, auto uc, auto mc) __attribute__((always_inline)) {
                m_step(rs, uc, mc);
                __syncthreads();
            });
        } else {
            s_prologue1();
            __syncthreads();
            s_prologue2();
            __syncthreads();
            run([&](int rs, auto uc, auto mc) __attribute__((always_inline)) {
                s_step(rs, uc, mc);
                __syncthreads();
            });
            s_epilogue();
        }
    } else {
        s_prologue1();
        __syncthreads();
        s_prologue2();
        __syncthreads();
        run([&](int rs, auto uc, auto mc) __attribute__((always_inline)) {
            m_step(rs, uc, mc);
            s_step(rs, uc, mc);
            __syncthreads();
        });
        s_epilogue();
    }
}

// log-likelihood of every segment from what k_tile_fwd left: log sum(exit vector) - log sum(entry
// vector) + ln 2 * removed exponents (the entry vector of a segment that starts its trajectory is
// exact: pi o p_0 carries the whole likelihood, nothing is subtracted)
// (sixteen lanes per segment: launched with (nseg + 15) / 16 workgroups of 256)
[[maybe_unused]] static __global__ void k_tile_logl(const Segs sg, int n, const double *a_entry,
                                                    const double *a_exit, const int32_t *eP_seg,
                                                    double *logL_seg, unsigned int *flags)
{
    const int s = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
    const bool live = s < sg.nseg && sg.len[s] > 0;
    const bool warm = live && sg.t0[s] > 0;
    double se = 0.0, sx = 0.0;
    if (live)
        for (int j = l; j < n; j += 16)
            sx += a_exit[(int64_t)s * n + j];
    if (warm)
        for (int j = l; j < n; j += 16)
            se += a_entry[(int64_t)s * n + j];
    for (int h = 8; h >= 1; h >>= 1) {
        sx += __shfl_xor(sx, h, 16);
        se += __shfl_xor(se, h, 16);
    }
    if (s >= sg.nseg || l != 0)
        return;
    if (!live) {
        logL_seg[s] = 0.0;
        return;
    }
    if (!warm)
        se = 1.0;
    if (!(sx > 0.0) || !(se > 0.0))
        atomicOr(&flags[2], 2u);
    logL_seg[s] = (log(sx) - log(se)) + (double)eP_seg[s] * 0.693147180559945309417232121458;
}

// =========================================================================================
// k_tile_bwd: beta in registers only; gamma, xi counts, emission statistics per tile.
//   part  [tile][n*n C' | n sum gamma | (gauss) n sum gamma d | n sum gamma d^2]
//   dstat [tile][4][n][M] (discrete: one table per lane row q, so that no two lanes share an entry)
// XIG: no xi accumulators; the rows W_{t-1} = p_t o beta_t / S are stored instead (counts by the
// time-parallel GEMM of gen_kernels.hpp) -- for state counts whose accumulators do not fit.
//
// One iteration of the matrix wavefronts (step us = time t of a row): the matrix instructions of
// beta_{t-1} = A (p_t o beta_t) | rescale, x' = p_{t-1} o beta_{t-1} into the other LDS buffer -- the
// serial chain -- | then, while that exchange is in flight: the xi matrix instructions of the
// transition t-1 -> t (operands: alpha_{t-1} in registers, p_t o beta_t still in this step's LDS
// buffer), gamma_{t-1} and the emission statistics | barrier.  The stream wavefronts supply the
// emission rows and the observations two steps ahead.
// =========================================================================================
template <int NT, int KIND, bool FULL, bool XIG, bool SPLIT>
__global__ __launch_bounds__(tile_threads<SPLIT>()) void k_tile_bwd(const WideModel m, const int64_t *off, const Segs sg,
                                                           const TilePlan tp, const void *obs_rm,
                                                           const double *alpha_rm, const int32_t *exps,
                                                           double *gamma_rm, double *gamma0, double *part,
                                                           double *dstat, double *b_exit, double *b_entry,
                                                           unsigned int *flags, double *Wg, unsigned long long *probe = nullptr)
{
#ifndef BHMM_TILE_PROBE_BUILD
    // (production build: the in-kernel cycle probe compiles away -- its run-time tests cost every wavefront some
    // fifteen instruction slots per step even when no probe buffer is given; -DBHMM_TILE_PROBE_BUILD brings it back)
    probe = nullptr;
#endif
    using G = TileGeo<NT>;
    constexpr int TPW = G::TPW, KK = G::KK, PX = G::PX, NP = G::NP;
    static_assert(!(XIG && SPLIT), "the W rows are stored by the both-role wavefronts (w_store)");
    __shared__ __attribute__((aligned(16))) double sX[2 * 16 * PX];
    __shared__ __attribute__((aligned(16))) double sP[2 * 4 * TPW * 2 * 64 * 2];
    __shared__ __attribute__((aligned(16))) double sObs[16 * 16]; // observations of 16 steps: [step & 15][4 q + r]
    __shared__ __attribute__((aligned(16))) double sAl[2 * 16 * PX]; // alpha_{t-1} tile of a step, laid out like sX
    __shared__ __attribute__((aligned(16))) int sEx[2 * 16];  // exponent the forward pass removed at t, [slot][4 q + r]
    __shared__ int sE[64];
    __shared__ double sS[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool matrix = !SPLIT || wid < 4;
    const int w = wid & 3;
    const int s = lane & 15, q = lane >> 4;
    const int n = FULL ? NP : m.n;

    // ---- the tile's rows: step us of the tile is time ttop - us of the row.  One lane per row works
    // the numbers out and leaves them in LDS; the matrix wavefronts only read them in the general
    // steps (their registers are needed elsewhere), the stream wavefronts keep their four rows'.
    __shared__ int sMeta[16 * 8]; // per row: seg, nwarm, nst, trj, ttop, (pad), gtop lo, gtop hi
    if (tid < 16) {
        const int sgi = tp.tile_seg[(int64_t)blockIdx.x * 16 + tid];
        int v_nwarm = 0, v_nst = 0, v_trj = 0, v_ttop = 0;
        int64_t v_gtop = 0;
        if (sgi >= 0 && sg.len[sgi] > 0) {
            const int k = sg.traj[sgi];
            const int64_t o0 = off[k], T = off[k + 1] - o0, t0 = sg.t0[sgi], t1 = t0 + sg.len[sgi];
            const int64_t te = (t1 - 1 + sg.W < T - 1) ? t1 - 1 + sg.W : T - 1;
            v_nwarm = t1 < T ? (int)(te - t1) + 1 : 0;
            v_nst = v_nwarm + (int)(t1 - t0);
            v_ttop = (int)(t1 - 1 + v_nwarm);
            v_gtop = o0 + v_ttop;
            v_trj = k;
        }
        int *mrow = sMeta + 8 * tid;
        mrow[0] = sgi;
        mrow[1] = v_nwarm;
        mrow[2] = v_nst;
        mrow[3] = v_trj;
        mrow[4] = v_ttop;
        mrow[5] = 0;
        mrow[6] = (int)(v_gtop & 0xffffffffll);
        mrow[7] = (int)(v_gtop >> 32);
    }
    __syncthreads();
    // (volatile: these reads must stay where they are written -- hoisted out of the loops they would
    // occupy the registers this exists to free)
    // (... and the LDS address space spelled out: through a generic volatile pointer every one of them is a flat
    // load with s_waitcnt vmcnt(0) behind it, i.e. a drain of all global stores in flight -- 3 150 cycles per
    // step of the W-row stores of 128 states, profiles/r05)
    typedef const volatile __attribute__((address_space(3))) int tile_lds_cvint;
    auto meta = [&](int r, int k) __attribute__((always_inline)) {
        return ((tile_lds_cvint *)sMeta)[8 * (q + 4 * r) + k];
    };
    auto meta_gtop = [&](int r) __attribute__((always_inline)) {
        return (int64_t)(((uint64_t)(uint32_t)meta(r, 7) << 32) | (uint32_t)meta(r, 6));
    };
    int nmax = 0, nstmin = 1 << 30, emin = 1 << 30, emax = 0;
    for (int rho = 0; rho < 16; ++rho) {
        const int vw = sMeta[8 * rho + 1], vn = sMeta[8 * rho + 2];
        nmax = max(nmax, vn);
        nstmin = min(nstmin, vn);
        emin = min(emin, vw);
        emax = max(emax, vw);
    }
    // Iteration us does the back half of step us and the front half of step us + 1.
    // [0, g1): both are warm-up steps of every row, and none reads alpha yet (also not four steps
    // ahead); [g2, g3): both are main-part steps of every row (entered, before the last step, t > 0
    // also for what is fetched ahead); the rest: general
    // (read from LDS, i.e. into vector registers: uniform all the same, and said so -- scalar loop control)
    nmax = __builtin_amdgcn_readfirstlane(nmax), nstmin = __builtin_amdgcn_readfirstlane(nstmin);
    emin = __builtin_amdgcn_readfirstlane(emin), emax = __builtin_amdgcn_readfirstlane(emax);
    const int g4 = (nmax + 3) & ~3;
    const int g1 = min(max(emin - 2 - TILE_PF, 0) & ~3, g4);
    const int g2 = min((emax + 1 + 3) & ~3, g4);
    const int g3 = min(max(g2, max(nstmin - 2 - TILE_PF, 0) & ~3), g4);

    bool real[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c)
        real[c] = (w + 4 * c < NT) && (FULL || 16 * (w + 4 * c) + s < n);
    // which steps need the sums over all states (both roles take the barriers of those exchanges)
    auto any_enter_at = [&](int us) __attribute__((always_inline)) {
        bool a = false;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            a |= us == meta(r, 1) && meta(r, 2) > 0;
        return (bool)__any(a); // (every wavefront holds all 16 rows: uniform over the workgroup)
    };
    auto any_last_at = [&](int us) __attribute__((always_inline)) {
        bool a = false;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            a |= us == meta(r, 2) - 1 && meta(r, 4) - us > 0;
        return (bool)__any(a);
    };
    // groups [lo, hi) in mode mc; the last group of a phase takes the general path (its last
    // iteration does the front half of the next phase's first step) and fetches for it
    auto run = [&](auto &&step, int lo, int hi, auto mc) __attribute__((always_inline)) {
        int us = lo;
        for (; us + 8 <= hi; us += 4) {
            step(us, tile_ic<0>{}, mc, mc);
            step(us + 1, tile_ic<1>{}, mc, mc);
            step(us + 2, tile_ic<2>{}, mc, mc);
            step(us + 3, tile_ic<3>{}, mc, mc);
        }
        for (; us + 4 <= hi; us += 4) {
            step(us, tile_ic<0>{}, tile_ic<TM_GEN>{}, tile_ic<TM_GEN>{});
            step(us + 1, tile_ic<1>{}, tile_ic<TM_GEN>{}, tile_ic<TM_GEN>{});
            step(us + 2, tile_ic<2>{}, tile_ic<TM_GEN>{}, tile_ic<TM_GEN>{});
            step(us + 3, tile_ic<3>{}, tile_ic<TM_GEN>{}, tile_ic<TM_GEN>{});
        }
    };

    // ================= the stream part ============================================================
    // (every vector instruction here competes with the matrix instructions for the SIMD: the
    // observation stream is read once per tile and passed on through LDS, alpha comes in
    // 16-byte pieces, one lane per (row, four states), and is laid out for the matrix wavefronts
    // by the LDS -- see k_tile_fwd)
    double mu_j[TPW], ga_j[TPW], gb_j[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int i = 16 * (w + 4 * c) + s;
        mu_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.mu[i] : 0.0; // (both parts use it)
        ga_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.ga[i] : 0.0;
        gb_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.gb[i] : 1.0;
    }
    auto rmeta = [&](int row, int k) __attribute__((always_inline)) { return sMeta[8 * row + k]; };
    auto rmeta_gtop = [&](int row) __attribute__((always_inline)) {
        return (int64_t)(((uint64_t)(uint32_t)rmeta(row, 7) << 32) | (uint32_t)rmeta(row, 6));
    };
    // ---- alpha_{t-1}: my row and my NP / 16 states of it.  FULL: SPL consecutive states.  Otherwise pairs of
    // states, pair e / 2 at 2 (lane & 15) + 16 e: sixteen lanes on 256 consecutive bytes of a row, one 16-byte
    // access per pair when n is even (a pair is then inside the row or outside it, and every row starts on a
    // 16-byte boundary), two 8-byte ones when it is odd
    constexpr int SPL = FULL ? NP / 16 : 2 * ((NP + 31) / 32); // (pairs: up to the row's pitch, never beyond it)
    const int srow = (w * 64 + lane) >> 4, sch = ((w * 64 + lane) & 15) * (FULL ? SPL : 2);
    auto spos = [&](int e) __attribute__((always_inline)) { return FULL ? e : 16 * e; }; // (e even)
    const bool n_even = FULL || (n & 1) == 0;
    const int s_nwarm = rmeta(srow, 1), s_nst = rmeta(srow, 2), s_ttop = rmeta(srow, 4);
    const int64_t s_abase = rmeta_gtop(srow) * n + sch;
    const int sxr = tile_prow(srow) * PX + sch;
    struct ARow {
        double v[SPL];
    };
    auto loadA = [&](ARow &a, int us) __attribute__((always_inline)) {
        // alpha_{t-1} from the last warm-up step on (it becomes alpha_t of the first main step)
#ifdef TILE_X_NOLOADA
        const bool wanta = false;
#else
        const bool wanta = us + 1 >= s_nwarm && us < s_nst && s_ttop - us > 0;
#endif
        const double *src = alpha_rm + s_abase - ((int64_t)us + 1) * n;
        if (n_even) {
#pragma unroll
            for (int e = 0; e < SPL; e += 2) {
                const bool in = wanta && (FULL || sch + spos(e) < n);
                const tile_d2 t2 = in ? *reinterpret_cast<const tile_d2 *>(src + spos(e)) : tile_d2{0.0, 0.0};
                a.v[e] = t2[0];
                a.v[e + 1] = t2[1];
            }
        } else {
#pragma unroll
            for (int e = 0; e < SPL; e += 2) {
                a.v[e] = (wanta && sch + spos(e) < n) ? src[spos(e)] : 0.0;
                a.v[e + 1] = (wanta && sch + spos(e) + 1 < n) ? src[spos(e) + 1] : 0.0;
            }
        }
    };
    auto a_to_lds = [&](const ARow &a, int slot) __attribute__((always_inline)) {
        double *dst = sAl + slot * 16 * PX + sxr;
#pragma unroll
        for (int e = 0; e < SPL; e += 2)
            *reinterpret_cast<tile_d2 *>(dst + spos(e)) = tile_d2{a.v[e], a.v[e + 1]};
    };
    // ---- the exponent the forward pass removed at time t (wavefront 5, one lane per row)
    const int xrow = lane & 15;
    const int x_nwarm = rmeta(xrow, 1), x_nst = rmeta(xrow, 2), x_ttop = rmeta(xrow, 4);
    const int64_t x_gtop = rmeta_gtop(xrow);
    const int xpos = 4 * (xrow & 3) + (xrow >> 2);
    auto loadX = [&](int us) __attribute__((always_inline)) -> int {
        return (w == 1 && lane < 16 && us >= x_nwarm && us < x_nst && ((x_ttop - us) & 3) == 3) ? exps[x_gtop - us] : 0;
    };
    // ---- the observation stream (wavefront 4: lane = 4 row + step of a group of four)
    const int lrow = lane >> 2, ldt = lane & 3;
    const int64_t l_gtop = rmeta_gtop(lrow);
    const int l_last = rmeta(lrow, 2) > 0 ? rmeta(lrow, 2) - 1 : 0;
    const int lpos = 4 * (lrow & 3) + (lrow >> 2); // row q + 4 r sits at position 4 q + r
    auto obs_load = [&](int step) __attribute__((always_inline)) -> double {
        const int64_t g = l_gtop - min(step, l_last);
        if constexpr (KIND == EMIT_DISC)
            return __hiloint2double(0, static_cast<const int32_t *>(obs_rm)[g]);
        else
            return static_cast<const double *>(obs_rm)[g];
    };
    double pend = 0.0; // the group two ahead, on its way
    // what the emission row of step us is computed from
    auto fetch_in = [&](TileIn<KIND, TPW> &in, int us) __attribute__((always_inline)) {
        if constexpr (KIND == EMIT_GAUSS) {
            const tile_d2 lo = *reinterpret_cast<const tile_d2 *>(&sObs[(us & 15) * 16 + 4 * q]);
            const tile_d2 hi = *reinterpret_cast<const tile_d2 *>(&sObs[(us & 15) * 16 + 4 * q + 2]);
            in.o[0] = lo[0];
            in.o[1] = lo[1];
            in.o[2] = hi[0];
            in.o[3] = hi[1];
        } else if constexpr (KIND == EMIT_DISC) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                in.sym[r] = __double2loint(sObs[(us & 15) * 16 + 4 * q + r]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nl = meta(r, 2) > 0 ? meta(r, 2) - 1 : 0;
                const int64_t g = meta_gtop(r) - min(us, nl);
#pragma unroll
                for (int c = 0; c < TPW; ++c)
                    in.p[c][r] = real[c] ? static_cast<const double *>(obs_rm)[g * n + 16 * (w + 4 * c) + s] : 0.0;
            }
        }
    };
    auto emit_to_lds = [&](const TileIn<KIND, TPW> &in, int slot) __attribute__((always_inline)) {
        double p[TPW][4];
        tile_emit4<NT, KIND, TPW>(m, in, w, s, real, mu_j, ga_j, gb_j, p);
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            *reinterpret_cast<tile_d2 *>(&sP[tile_p_index<TPW>(slot, w, c, 0, lane)]) = tile_d2{p[c][0], p[c][1]};
            *reinterpret_cast<tile_d2 *>(&sP[tile_p_index<TPW>(slot, w, c, 1, lane)]) = tile_d2{p[c][2], p[c][3]};
        }
    };
    ARow ringA[TILE_PF];
    int ringX[TILE_PF];
    auto s_prologue1 = [&]() __attribute__((always_inline)) {
        if constexpr (KIND != EMIT_EXPL) {
            if (w == 0) {
                sObs[ldt * 16 + lpos] = obs_load(ldt);
                sObs[(4 + ldt) * 16 + lpos] = obs_load(4 + ldt);
                pend = obs_load(8 + ldt);
            }
        }
#pragma unroll
        for (int u = 0; u < TILE_PF; ++u) {
            loadA(ringA[u], u);
            ringX[u] = loadX(u);
        }
        a_to_lds(ringA[0], 0);
        loadA(ringA[0], TILE_PF);
        if (w == 1 && lane < 16)
            sEx[xpos] = ringX[0];
        ringX[0] = loadX(TILE_PF);
    };
    auto s_prologue2 = [&]() __attribute__((always_inline)) {
        TileIn<KIND, TPW> in0;
        fetch_in(in0, 0);
        emit_to_lds(in0, 0);
        fetch_in(in0, 1);
        emit_to_lds(in0, 1);
    };
    auto s_step = [&](int us, auto uc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        // alpha / exponent for the back half of step us + 1; the loads of four steps further
        a_to_lds(ringA[(u + 1) & 3], (u + 1) & 1);
        loadA(ringA[(u + 1) & 3], us + 1 + TILE_PF);
        if (w == 1 && lane < 16)
            sEx[((u + 1) & 1) * 16 + xpos] = ringX[(u + 1) & 3];
        ringX[(u + 1) & 3] = loadX(us + 1 + TILE_PF);
        if constexpr (KIND != EMIT_EXPL && u == 0) {
            if (w == 0) { // the observations of the group two ahead go to LDS, the next ones are fetched
                sObs[((us + 8 + ldt) & 15) * 16 + lpos] = pend;
                pend = obs_load(us + 12 + ldt);
            }
        }
        // the emission row of step us + 2
        TileIn<KIND, TPW> ein;
        fetch_in(ein, us + 2);
        emit_to_lds(ein, u & 1);
    };

    // ================= the matrix part ============================================================
    double Breg[TPW * KK]; // my blocks of A^T (B operand of the beta product)
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int i = 16 * (w + 4 * c) + s; // my state: row i of A
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int j = q * KK + kk;
            Breg[c * KK + kk] = (matrix && real[c] && (FULL || j < n)) ? m.A[(int64_t)i * n + j] : 0.0;
        }
    }
    int xw[4], xq[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        xw[r] = tile_prow(q + 4 * r) * PX;
        xq[r] = xw[r] + s; // xi operand: row q + 4 r, column 16 J + s
    }
    const int xr = tile_prow(s) * PX + q * KK;

    // xi accumulators: C'[16 (w + 4 c) .., 16 J ..]
    wide_d4 Cacc[XIG ? 1 : TPW][XIG ? 1 : NT];
    if constexpr (!XIG) {
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int J = 0; J < NT; ++J)
                Cacc[c][J] = wide_d4{0.0, 0.0, 0.0, 0.0};
    }
    double sgm[TPW], sd[TPW], sdd[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c)
        sgm[c] = sd[c] = sdd[c] = 0.0;
    double *mytab = nullptr;
    if constexpr (KIND == EMIT_DISC) {
        mytab = dstat + ((int64_t)blockIdx.x * 4 + q) * n * m.M;
        if (matrix) {
#pragma unroll
            for (int c = 0; c < TPW; ++c)
                if (real[c])
                    for (int z = 0; z < m.M; ++z)
                        mytab[(int64_t)(16 * (w + 4 * c) + s) * m.M + z] = 0.0;
        }
    }

    // alpha of time t - 1 and the exponent removed at t: from the LDS slot the stream wavefronts filled
    struct AIn {
        double ap[TPW][4];
        int ex[4];
    };
    auto fetch_a = [&](int slot, AIn &in) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                // (a column tile beyond NT does not exist in the LDS tile: what lies there -- another row,
                // another array, whatever the previous kernel left -- may be a NaN pattern, and 0 x NaN would
                // reach the row sums: the transient of 65 states)
                in.ap[c][r] = (NT % 4 == 0 || w + 4 * c < NT) ? sAl[slot * 16 * PX + xw[r] + 16 * (w + 4 * c) + s] : 0.0;
        typedef int tile_i4 __attribute__((ext_vector_type(4)));
        const tile_i4 e = *reinterpret_cast<const tile_i4 *>(&sEx[slot * 16 + 4 * q]);
        in.ex[0] = e[0];
        in.ex[1] = e[1];
        in.ex[2] = e[2];
        in.ex[3] = e[3];
    };

    // state of my rows: beta; fcur = 1 / (S0 2^(exponents removed since the row entered its main part)),
    // the factor that normalises alpha_t o beta_t to gamma_t (0 outside the main part)
    double beta[TPW][4];
    double fcur[4] = {0.0, 0.0, 0.0, 0.0}, mass = 0.0;
#pragma unroll
    for (int c = 0; c < TPW; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            beta[c][r] = real[c] ? 1.0 / (double)n : 0.0; // _hidden.c:79-88 (any scale)
    unsigned int trouble = 0u; // (bits: 4 S0, 8 entry normaliser, 16 a vector below 2^-900, 32 gamma mass)
    // sum over all states of v (my states, my four rows), exchanged through sS: two barriers (which
    // the stream wavefronts take as well)
    auto rows_sum = [&](const double (&v)[4], double (&out)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double ps = row16_sum(v[r]);
            if (s == 0)
                sS[16 * w + q + 4 * r] = ps;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rho = q + 4 * r;
            out[r] = (sS[rho] + sS[16 + rho]) + (sS[32 + rho] + sS[48 + rho]);
        }
        __syncthreads();
    };

    // front half of step us (time t), the chain part: [a row enters its main part: normaliser S0] ;
    // x = p_t o beta_t into LDS buffer us & 1; the gamma factors of the step
    auto front_chain = [&](int us, auto uc, auto mc, double (&fg)[4], bool (&mainr)[4],
                           const tile_d2 (&pl)[TPW][2], const double (&acur)[TPW][4]) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, MODE = decltype(mc)::value;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            mainr[r] = MODE == TM_MAIN;
        if constexpr (MODE == TM_GEN) {
            bool enter[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mainr[r] = us >= meta(r, 1) && us < meta(r, 2);
                enter[r] = us == meta(r, 1) && meta(r, 2) > 0;
            }
            if (any_enter_at(us)) {
                // the warm-up's beta for the boundary check, and the normaliser of the whole
                // segment, S0 = sum_j alpha_t*[j] beta_t*[j]
                double g[4] = {0.0, 0.0, 0.0, 0.0}, S0[4];
#pragma unroll
                for (int c = 0; c < TPW; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        g[r] += acur[c][r] * beta[c][r];
                        if (enter[r] && meta(r, 1) > 0 && real[c])
                            b_exit[(int64_t)meta(r, 0) * n + 16 * (w + 4 * c) + s] = beta[c][r];
                    }
                rows_sum(g, S0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (enter[r]) {
                        trouble |= (!(S0[r] >= 0x1p-959) || !(S0[r] < 0x1p1000)) ? 4u : 0u;
                        fcur[r] = fast_rcp(S0[r]);
                    }
            }
        }
        double *Xn = sX + (u & 1) * 16 * PX;
#pragma unroll
        for (int c = 0; c < TPW; ++c)
            if (NT % 4 == 0 || w + 4 * c < NT)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Xn[xw[r] + 16 * (w + 4 * c) + s] = pl[c][r >> 1][r & 1] * beta[c][r];
        if constexpr (MODE != TM_WARM) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                fg[r] = mainr[r] ? fcur[r] : 0.0;
        }
    };
    // ... and off the chain: gamma_t and the emission statistics of the rows in their main part
    // (ol: the observations of the step, position r of my lane row)
    auto front_stats = [&](int us, auto mc, const double (&fg)[4], const bool (&mainr)[4],
                           const tile_d2 (&ol)[2], const double (&acur)[TPW][4]) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mc)::value;
#ifdef TILE_X_NOSTATS
        return;
#endif
        if constexpr (MODE != TM_WARM) {
            const int64_t usn = (int64_t)us * n;
#pragma unroll
            for (int c = 0; c < TPW; ++c) {
                const int j = 16 * (w + 4 * c) + s;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double gam = acur[c][r] * beta[c][r] * fg[r];
                    mass += gam;
                    sgm[c] += gam;
                    if constexpr (KIND == EMIT_GAUSS) {
                        const double d = ol[r >> 1][r & 1] - mu_j[c];
                        const double gd = gam * d;
                        sd[c] += gd;
                        sdd[c] = fma(gd, d, sdd[c]);
                    }
                    if constexpr (KIND == EMIT_DISC) {
                        if (real[c] && mainr[r]) {
                            const int sym = __double2loint(ol[r >> 1][r & 1]);
                            mytab[(int64_t)j * m.M + sym] += gam;
                        }
                    }
                    if (real[c] && mainr[r]) {
                        if (gamma_rm)
                            gamma_rm[meta_gtop(r) * n + j - usn] = gam;
                        if constexpr (MODE == TM_GEN)
                            if (meta(r, 4) - us == 0)
                                gamma0[(int64_t)meta(r, 3) * n + j] = gam;
                    }
                }
            }
        }
    };
    // emission row / observations of a step from the LDS slot the stream wavefronts filled
    auto fetch_p = [&](int us, tile_d2 (&pl)[TPW][2], tile_d2 (&ol)[2]) __attribute__((always_inline)) {
        const int slot = us & 1;
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            pl[c][0] = *reinterpret_cast<const tile_d2 *>(&sP[tile_p_index<TPW>(slot, w, c, 0, lane)]);
            pl[c][1] = *reinterpret_cast<const tile_d2 *>(&sP[tile_p_index<TPW>(slot, w, c, 1, lane)]);
        }
        if constexpr (KIND != EMIT_EXPL) { // (discrete: the symbol is the low word)
            ol[0] = *reinterpret_cast<const tile_d2 *>(&sObs[(us & 15) * 16 + 4 * q]);
            ol[1] = *reinterpret_cast<const tile_d2 *>(&sObs[(us & 15) * 16 + 4 * q + 2]);
        } else {
            ol[0] = ol[1] = tile_d2{0.0, 0.0};
        }
    };

    // prologue: the front half of step 0
    auto m_prologue = [&]() __attribute__((always_inline)) {
        tile_d2 pl[TPW][2], ol[2];
        double fg[4] = {0.0, 0.0, 0.0, 0.0};
        bool mainr[4];
        // rows without a warm-up (end of the trajectory) start in the main part: alpha_{T-1}
        double a0[TPW][4];
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                a0[c][r] = (meta(r, 2) > 0 && meta(r, 1) == 0 && real[c])
                               ? alpha_rm[meta_gtop(r) * n + 16 * (w + 4 * c) + s] : 0.0;
        fetch_p(0, pl, ol);
        front_chain(0, tile_ic<0>{}, tile_ic<TM_GEN>{}, fg, mainr, pl, a0);
        front_stats(0, tile_ic<TM_GEN>{}, fg, mainr, ol, a0);
    };

    // XIG: W_{t-1} = (p_t o beta_t) fx of my row, my NP / 16 states of it -- the piece of the LDS tile and the
    // address the alpha loads of the stream part use (one address per lane and step, 16-byte pieces), instead of
    // eight scattered 8-byte stores per lane from the matrix layout
    double fxs = 0.0;
    auto w_store = [&](int us, auto uc, auto mc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, MODE = decltype(mc)::value;
#ifdef TILE_X_NOW
        return;
#endif
        if constexpr (XIG && MODE != TM_WARM) {
            const bool want = MODE == TM_MAIN || (us >= s_nwarm && us < s_nst && s_ttop - us > 0);
            if (want) {
                const double *Xs = sX + (u & 1) * 16 * PX + sxr;
                double *dst = Wg + s_abase - ((int64_t)us + 1) * n;
                if (n_even) {
#pragma unroll
                    for (int e = 0; e < SPL; e += 2) {
                        const tile_d2 x2 = *reinterpret_cast<const tile_d2 *>(Xs + spos(e));
                        if (FULL || sch + spos(e) < n)
                            *reinterpret_cast<tile_d2 *>(dst + spos(e)) = tile_d2{x2[0] * fxs, x2[1] * fxs};
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < SPL; e += 2) {
                        const tile_d2 x2 = *reinterpret_cast<const tile_d2 *>(Xs + spos(e));
                        if (sch + spos(e) < n)
                            dst[spos(e)] = x2[0] * fxs;
                        if (sch + spos(e) + 1 < n)
                            dst[spos(e) + 1] = x2[1] * fxs;
                    }
                }
            }
        }
    };
    unsigned long long pc = 0; // (probe: end of the previous step's work)
    auto m_step = [&](int us, auto uc, auto mc) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value, MODE = decltype(mc)::value;
        bool mainr[4], lastr[4];
        int tt[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mainr[r] = MODE == TM_MAIN;
            lastr[r] = false;
            tt[r] = 1;
        }
        if constexpr (MODE == TM_GEN) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mainr[r] = us >= meta(r, 1) && us < meta(r, 2);
                tt[r] = meta(r, 4) - us;
                lastr[r] = us == meta(r, 2) - 1 && tt[r] > 0; // the transition into the segment
            }
        }
#ifdef TILE_X_PROBE_REGS
        const bool pr = false;
#else
        const bool pr = probe && blockIdx.x == gridDim.x - 1 && wid == 0;
#endif
        const unsigned long long c0 = pr ? __builtin_readcyclecounter() : 0;
        AIn in;
        fetch_a(u & 1, in);
        // ---- back half of step us: beta_{t-1} (raw) = A (p_t o beta_t) ----------------------------
        const double *X = sX + (u & 1) * 16 * PX;
        tile_d2 pl[TPW][2], ol[2];
        fetch_p(us + 1, pl, ol); // emission row / observations of step us + 1
        wide_d4 acc[TPW];
        constexpr int CH = KK % 8 == 0 ? 8 : (KK % 4 == 0 ? 4 : 2); // (operands in pieces of eight, see k_tile_fwd)
#pragma unroll
        for (int k0 = 0; k0 < KK; k0 += CH) {
            tile_d2 av[CH / 2];
#pragma unroll
            for (int k2 = 0; k2 < CH / 2; ++k2)
                av[k2] = *reinterpret_cast<const tile_d2 *>(X + xr + k0 + 2 * k2);
#pragma unroll
            for (int kk = k0; kk < k0 + CH; ++kk)
#pragma unroll
                for (int c = 0; c < TPW; ++c)
                    // (a column tile beyond NT: its block of A is zero, the product is computed all the same -- the
                    // wavefronts with two real tiles set the pace of the step, and a chain without branches keeps the
                    // accumulators where they are: with the branch, 770 register moves per step at NT = 6)
                    acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[(kk - k0) >> 1][(kk - k0) & 1], Breg[c * KK + kk],
                                                                  kk == 0 ? wide_d4{0.0, 0.0, 0.0, 0.0} : acc[c], 0, 0, 0);
        }
        unsigned long long c1 = 0;
        if (pr) {
            asm volatile("" ::"v"(acc[0][0]));
            c1 = __builtin_readcyclecounter();
        }
        // factors of xi for the transition t-1 -> t: the exponent the forward pass removed at t comes off
        double fx[4] = {0.0, 0.0, 0.0, 0.0}, fnx[4] = {0.0, 0.0, 0.0, 0.0};
        if constexpr (MODE != TM_WARM) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                fnx[r] = ldexp(fcur[r], -in.ex[r]);
                fx[r] = (mainr[r] && tt[r] > 0) ? fnx[r] : 0.0;
            }
            if constexpr (MODE == TM_GEN) {
                if (any_last_at(us)) {
                    // alpha_{t0-1} belongs to the neighbouring segment's chain: its own normaliser
                    double g[4] = {0.0, 0.0, 0.0, 0.0}, SL[4];
#pragma unroll
                    for (int c = 0; c < TPW; ++c)
                        if (NT % 4 == 0 || w + 4 * c < NT)
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                g[r] += in.ap[c][r] * acc[c][r];
                    rows_sum(g, SL);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (lastr[r]) {
                            trouble |= (!(SL[r] >= 0x1p-959) || !(SL[r] < 0x1p1000)) ? 8u : 0u;
                            fx[r] = fast_rcp(SL[r]);
                        }
                }
            }
        }
        // rescale (every fourth step, by the row maxima exchanged one step earlier)
        int E[4] = {0, 0, 0, 0};
        if constexpr (u == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rho = q + 4 * r;
                E[r] = max(max(sE[rho], sE[16 + rho]), max(sE[32 + rho], sE[48 + rho]));
                trouble |= ((MODE != TM_GEN || us < meta(r, 2)) && E[r] < WIDE_TROUBLE_EXP) ? 16u : 0u;
            }
        }
        int pm[4] = {-(1 << 28), -(1 << 28), -(1 << 28), -(1 << 28)};
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double b = (NT % 4 == 0 || w + 4 * c < NT) ? acc[c][r] : 0.0;
                if constexpr (u == 3)
                    b = ldexp(b, -E[r]);
                if constexpr (u == 2)
                    pm[r] = max(pm[r], b > 0.0 ? exponent_of(b) : -(1 << 28));
                if constexpr (MODE == TM_GEN)
                    if (lastr[r] && real[c])
                        b_entry[(int64_t)meta(r, 0) * n + 16 * (w + 4 * c) + s] = b; // beta one step before the segment
                beta[c][r] = b;
            }
        if constexpr (MODE != TM_WARM) { // ... and the exponent just removed from beta goes on
#pragma unroll
            for (int r = 0; r < 4; ++r)
                fcur[r] = u == 3 ? ldexp(fnx[r], E[r]) : fnx[r];
        }
        if constexpr (u == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mx = row16_max_i32(pm[r]);
                if (s == 0)
                    sE[16 * w + q + 4 * r] = mx;
            }
        }
        // ---- front half of step us + 1, the chain part: x' into the other buffer ------------------
        double fg[4] = {0.0, 0.0, 0.0, 0.0};
        bool mainn[4];
        front_chain(us + 1, tile_ic<(u + 1) & 3>{}, mc, fg, mainn, pl, in.ap);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long c2 = pr ? __builtin_readcyclecounter() : 0;
        // ---- off the chain: xi of the transition t-1 -> t -----------------------------------------
#ifndef TILE_X_NOXI
        if constexpr (MODE != TM_WARM) {
            if constexpr (!XIG) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double xb[NT];
#pragma unroll
                    for (int J = 0; J < NT; ++J)
                        xb[J] = X[xq[r] + 16 * J];
#pragma unroll
                    for (int c = 0; c < TPW; ++c)
                        if (NT % 4 == 0 || w + 4 * c < NT) {
                            const double aw = in.ap[c][r] * fx[r];
#pragma unroll
                            for (int J = 0; J < NT; ++J)
                                Cacc[c][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, xb[J], Cacc[c][J], 0, 0, 0);
                        }
                }
            } else {
                // (the rows W_{t-1} leave in the stream layout -- w_store below; row 4 w + q of that layout
                // is register w of this lane)
                fxs = w == 0 ? fx[0] : (w == 1 ? fx[1] : (w == 2 ? fx[2] : fx[3]));
            }
        }
#endif
        front_stats(us + 1, mc, fg, mainn, ol, in.ap);
        if (pr && lane == 0) {
            const unsigned long long c3 = __builtin_readcyclecounter();
            const int o = MODE == TM_WARM ? 0 : 8;
            probe[o + 0] += c1 - c0;
            probe[o + 1] += c2 - c1;
            probe[o + 2] += c3 - c2;
            probe[o + 3] += pc ? c0 - pc : 0;
            probe[o + 4] += 1;
            pc = c3;
        }
    };

    // ================= the loop ===================================================================
    auto run_all = [&](auto &&step) __attribute__((always_inline)) {
        run(step, 0, g1, tile_ic<TM_WARM>{});
        run(step, g1, g2, tile_ic<TM_GEN>{});
        run(step, g2, g3, tile_ic<TM_MAIN>{});
        run(step, g3, g4, tile_ic<TM_GEN>{});
    };
    if constexpr (SPLIT) {
        if (!matrix) {
            s_prologue1();
            __syncthreads(); // (the first observations are in LDS)
            s_prologue2();
            __syncthreads(); // (the emission rows of steps 0 and 1, alpha for step 0 are in LDS)
            if (any_enter_at(0)) { // (the barriers of the matrix wavefronts' exchanges, here and below)
                __syncthreads();
                __syncthreads();
            }
            __syncthreads(); // (end of the prologue)
            run_all([&](int us, auto uc, auto mc, auto) __attribute__((always_inline)) {
                constexpr int MODE = decltype(mc)::value;
                if constexpr (MODE == TM_GEN) {
                    if (any_last_at(us)) {
                        __syncthreads();
                        __syncthreads();
                    }
                    if (any_enter_at(us + 1)) {
                        __syncthreads();
                        __syncthreads();
                    }
                }
                s_step(us, uc);
                __syncthreads();
            });
            __syncthreads(); // (the gamma mass check)
            __syncthreads();
            return;
        }
        __builtin_amdgcn_s_setprio(3); // (the serial chain runs here, see k_tile_fwd)
        __syncthreads();
        __syncthreads();
        m_prologue();
        __syncthreads();
        run_all([&](int us, auto uc, auto mc, auto) __attribute__((always_inline)) {
            m_step(us, uc, mc);
            __syncthreads();
        });
    } else {
        s_prologue1();
        __syncthreads();
        s_prologue2();
        __syncthreads();
        m_prologue();
        __syncthreads();
#ifdef TILE_X_PROBE_REGS
        // (variant builds only: phase times of the both-role wavefront 0 of the last workgroup, summed in
        // registers -- the run-time probe above adds to global memory in every step, and the wait that needs
        // drains the step's stores)
        unsigned long long pa[2][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
        const bool prr = probe && blockIdx.x == gridDim.x - 1 && wid == 0;
        run_all([&](int us, auto uc, auto mc, auto) __attribute__((always_inline)) {
            constexpr int o = decltype(mc)::value == TM_WARM ? 0 : 1;
            const unsigned long long t0 = prr ? __builtin_readcyclecounter() : 0;
            m_step(us, uc, mc);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t1 = prr ? __builtin_readcyclecounter() : 0;
            w_store(us, uc, mc);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t2 = prr ? __builtin_readcyclecounter() : 0;
            s_step(us, uc);
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t3 = prr ? __builtin_readcyclecounter() : 0;
            __syncthreads();
            const unsigned long long t4 = prr ? __builtin_readcyclecounter() : 0;
            pa[o][0] += t1 - t0;
            pa[o][1] += t2 - t1;
            pa[o][2] += t3 - t2;
            pa[o][3] += t4 - t3;
            pa[o][4] += 1;
        });
        if (prr && lane == 0)
            for (int o = 0; o < 2; ++o)
                for (int i = 0; i < 5; ++i)
                    probe[32 + 8 * o + i] = pa[o][i];
#else
        run_all([&](int us, auto uc, auto mc, auto) __attribute__((always_inline)) {
            m_step(us, uc, mc);
            w_store(us, uc, mc);
            s_step(us, uc);
            __syncthreads();
        });
#endif
    }

    // ---- self-check: unit gamma mass per step of every row -------------------------------------
    {
        // (the mass of all rows of my lane row q; the wanted number likewise)
        const double mq[4] = {mass, 0.0, 0.0, 0.0};
        double tot[4], want = 0.0;
        rows_sum(mq, tot);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            want += (double)(meta(r, 2) - meta(r, 1));
        trouble |= !(fabs(tot[0] - want) <= 1e-8 * want) ? 32u : 0u;
        if (trouble)
            atomicOr(&flags[2], trouble);
    }
    // ---- the tile's partial statistics --------------------------------------------------------
    const int S = n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0);
    double *mypart = part + (int64_t)blockIdx.x * S;
    if constexpr (!XIG) {
#pragma unroll
        for (int c = 0; c < TPW; ++c)
            if (NT % 4 == 0 || w + 4 * c < NT)
#pragma unroll
                for (int J = 0; J < NT; ++J)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * (w + 4 * c) + q + 4 * r, col = 16 * J + s;
                        if (FULL || (row < n && col < n))
                            mypart[(int64_t)row * n + col] = Cacc[c][J][r];
                    }
    }
    // emission statistics of state 16 (w + 4 c) + s: the four lane rows q hold disjoint tile rows
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        double a0 = sgm[c], a1 = sd[c], a2 = sdd[c];
        a0 += __shfl_xor(a0, 16, 64);
        a0 += __shfl_xor(a0, 32, 64);
        if constexpr (KIND == EMIT_GAUSS) {
            a1 += __shfl_xor(a1, 16, 64);
            a1 += __shfl_xor(a1, 32, 64);
            a2 += __shfl_xor(a2, 16, 64);
            a2 += __shfl_xor(a2, 32, 64);
        }
        if (real[c] && q == 0) {
            const int j = 16 * (w + 4 * c) + s;
            mypart[n * n + j] = a0;
            if constexpr (KIND == EMIT_GAUSS) {
                mypart[n * n + n + j] = a1;
                mypart[n * n + 2 * n + j] = a2;
            }
        }
    }
}

// packed statistics (bhmm_amd.h layout) of a pass whose xi counts come from the time-parallel GEMM
// (XIG): C = A o sum of the GEMM's split partials; everything else from the tiles' partial blocks.
// One wavefront per output entry, fixed summation order.
template <int KIND>
__global__ __launch_bounds__(64) void k_tile_finalize_xig(const WideModel m, int K, int ntiles, int nsplit,
                                                          const double *xipart, const double *part,
                                                          const double *dstat, const double *logL_k,
                                                          const double *gamma0, double *stats)
{
    const int n = m.n;
    const int64_t nn = (int64_t)n * n;
    const int S = n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0); // (the tiles' block layout, C' part unused)
    const int nE = KIND == EMIT_GAUSS ? 2 * n : 0;
    const int64_t MN = KIND == EMIT_DISC ? (int64_t)n * m.M : 0;
    const int64_t oG0 = 1, oC = 1 + n, oSG = oC + nn, oE = oSG + n;
    const int lane = threadIdx.x;
    int64_t e = blockIdx.x;
    double s = 0.0;
    if (e < nn) {
        for (int k = lane; k < nsplit; k += 64)
            s += xipart[(int64_t)k * nn + e];
        s = wave_sum(s);
        if (lane == 0)
            stats[oC + e] = s * m.A[e];
        return;
    }
    e -= nn;
    if (e < n + nE) {
        for (int k = lane; k < ntiles; k += 64)
            s += part[(int64_t)k * S + nn + e];
        s = wave_sum(s);
        if (lane == 0)
            stats[(e < n ? oSG : oE - n) + e] = s;
        return;
    }
    e -= n + nE;
    if (e < MN) {
        for (int k = lane; k < 4 * ntiles; k += 64)
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

} // namespace bhmm
