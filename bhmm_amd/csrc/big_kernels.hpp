// big_kernels.hpp -- row-batched E-step recursions on the fp64 matrix cores for MORE THAN 128 states
// (129 .. 512), where the transition matrix no longer fits the register file of a compute unit.
//
// Reference loops covered: bhmm/hidden/impl_c/_hidden.c:42-63 (forward), :91-109 (backward), the
// normalisers of :148-183 (xi counts; the counts themselves by the time-parallel GEMM of
// gen_kernels.hpp over the rows W this pass stores) and hidden/api.py:176-186 (gamma), with the
// emission rows of output_models/impl_c/_gaussian.c:5-21 / discrete.py:130-157 fused in.
//
// Same tiling as tile_kernels.hpp -- sixteen trajectory segments ("rows") per workgroup, one step of
// all of them is alpha-tile[16 x N] . A[N x N] on v_mfma_f64_16x16x4_f64, four wavefronts, column tile
// ct (16 states) on wavefront ct mod 4, the tile of the previous step all-gathered through LDS -- but
//   * A's blocks are STREAMED: a packed copy in MFMA operand order (k_big_pack: [column tile][kk][lane],
//     512 contiguous bytes per matrix instruction) stays resident in L2 (0.5 MB at 256 states, 2 MB at
//     512) and every wavefront pulls its column tiles' blocks through a ring of register buffers, three
//     blocks of 16 matrix instructions ahead of their use.  N^2 * 8 bytes per step and workgroup: at 256
//     states 0.5 MB per 6.8 us of matrix instructions = 77 GB/s per compute unit.
//   * a wavefront's column tiles are INDEPENDENT accumulator chains (N / 64 of them), so the matrix
//     instructions never wait for each other as they do at 64 states;
//   * the vectors are normalised at EVERY step, like the reference's (two small exchanges through LDS
//     per step -- nothing next to >= 256 matrix instructions per wavefront): no lazy power-of-two
//     scaling, no exponent bookkeeping; log c_t is accumulated per row as _hidden.c:57-66 does.
// A step whose normaliser is zero or not finite (an all-zero emission row -> the outlier rule of
// outputmodel.py:126-130, a zero-probability observation) raises flags[2]: the host repeats the
// E-step with the order-faithful kernels of gen_kernels.hpp, which implement those rules.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tile_kernels.hpp"

namespace bhmm {

constexpr int BIG_KC = 16;   // matrix instructions per streamed block of A

template <int TPW>
struct BigGeo {
    static constexpr int NP = 64 * TPW; // padded state count: TPW column tiles on each of four wavefronts
    static constexpr int KK = NP / 4;   // K steps of one product
    static constexpr int PX = NP + 2;   // pitch of a tile row in LDS: == 2 (mod 32), see tile_prow
    static constexpr int NBLK = (KK / BIG_KC) * TPW;
    static constexpr size_t smem = (size_t)(2 * 16 * PX + 3 * 64) * sizeof(double);
};

// A in operand order, zero-padded: element (ct, kk, lane = (s, q)) of
//   Bf: A[q KK + kk][16 ct + s]   (forward:  alpha-tile . A)
//   Bb: A[16 ct + s][q KK + kk]   (backward: x-tile . A^T)
// at index ((ct KK/2 + kk/2) 64 + lane) 2 + (kk & 1): sixteen bytes per lane and pair of K steps.
[[maybe_unused]] static __global__ void k_big_pack(const double *A, int n, int NP, double *Bf, double *Bb)
{
    const int KK = NP / 4, NT = NP / 16;
    const int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (e >= (int64_t)NT * KK * 64)
        return;
    const int lane = (int)(e & 63), kk = (int)((e >> 6) % KK), ct = (int)((e >> 6) / KK);
    const int s = lane & 15, q = lane >> 4;
    const int i = q * KK + kk, j = 16 * ct + s;
    const int64_t o = (((int64_t)ct * (KK / 2) + kk / 2) * 64 + lane) * 2 + (kk & 1);
    const bool in = i < n && j < n;
    Bf[o] = in ? A[(int64_t)i * n + j] : 0.0;
    Bb[o] = in ? A[(int64_t)j * n + i] : 0.0;
}

// blocks of A in flight: the ring of register buffers lives ACROSS the steps (the matrix is the same at
// every step), so that the first blocks of step t + 1 are requested during the last blocks of step t --
// i.e. BEFORE step t's alpha / W / gamma stores.  Memory operations of a wavefront retire in order
// (vmcnt): with the ring refilled at the top of a step, its first matrix instruction waited for the
// previous step's stores to reach HBM (~7 of 11 us per step at 256 states, profiles/r05).
template <int TPW>
constexpr int big_ring() // (divides the block count, so that the slots wrap with the blocks)
{
    return BigGeo<TPW>::NBLK % 4 == 0 ? 4 : (BigGeo<TPW>::NBLK % 3 == 0 ? 3 : (BigGeo<TPW>::NBLK % 5 == 0 ? 5 : 7));
}

template <int TPW>
struct BigRing {
    tile_d2 v[big_ring<TPW>()][BIG_KC / 2];
};

template <int TPW, int B>
__device__ __forceinline__ void big_issue(BigRing<TPW> &ring, const double *__restrict__ Bp, int w, int lane)
{
    using G = BigGeo<TPW>;
    constexpr int k0 = (B / TPW) * BIG_KC, c = B % TPW;
    // (the global address space spelled out: behind the opaque copy of big_stream_ptr the compiler no longer knows
    // it, and flat loads count against the LDS counter as well -- every wait for the tile's operands would drain
    // the ring)
    typedef const tile_d2 __attribute__((address_space(1))) big_gd2;
    big_gd2 *src = (big_gd2 *)(reinterpret_cast<const tile_d2 *>(Bp) + ((int64_t)(w + 4 * c) * (G::KK / 2) + k0 / 2) * 64 + lane);
#ifdef BIG_X_NOSTREAM // (experiment builds, tools/proto/big_variants.sh: what does a piece of the step cost?)
    (void)src;
#pragma unroll
    for (int k2 = 0; k2 < BIG_KC / 2; ++k2)
        ring.v[B % big_ring<TPW>()][k2] = tile_d2{1.0 / 256, 1.0 / 256};
#else
#pragma unroll
    for (int k2 = 0; k2 < BIG_KC / 2; ++k2)
        ring.v[B % big_ring<TPW>()][k2] = src[(int64_t)k2 * 64];
#endif
}

// before the first step: blocks 0 .. RING - 2
template <int TPW>
__device__ __forceinline__ void big_prime(BigRing<TPW> &ring, const double *__restrict__ Bp, int w, int lane)
{
    big_issue<TPW, 0>(ring, Bp, w, lane);
    if constexpr (big_ring<TPW>() > 2)
        big_issue<TPW, 1>(ring, Bp, w, lane);
    if constexpr (big_ring<TPW>() > 3)
        big_issue<TPW, 2>(ring, Bp, w, lane);
    if constexpr (big_ring<TPW>() > 4)
        big_issue<TPW, 3>(ring, Bp, w, lane);
    if constexpr (big_ring<TPW>() > 5)
        big_issue<TPW, 4>(ring, Bp, w, lane);
    if constexpr (big_ring<TPW>() > 6)
        big_issue<TPW, 5>(ring, Bp, w, lane);
}

// tile[16 x NP] (LDS buffer X) times the streamed matrix Bp: acc[c] = column tile w + 4 c of the product
// (C/D layout: lane (s, q), register r <-> row q + 4 r, state 16 (w + 4 c) + s).  On entry the ring holds
// (or has in flight) blocks 0 .. RING - 2; on exit those of the next step.
template <int TPW>
__device__ __forceinline__ void big_product(const double *X, int xr, const double *__restrict__ Bp, int w, int lane,
                                            BigRing<TPW> &ring, wide_d4 (&acc)[TPW])
{
    using G = BigGeo<TPW>;
    constexpr int KK = G::KK, NBLK = G::NBLK, KC = BIG_KC, RING = big_ring<TPW>();
    static_assert(NBLK % RING == 0 && NBLK >= RING, "ring slots must wrap with the blocks");
    tile_d2 av[2][KC / 2];
    // block b: K chunk b / TPW, column tile w + 4 (b % TPW)
    auto for_blocks = [&](auto &&self, auto bc) __attribute__((always_inline)) -> void {
        constexpr int b = decltype(bc)::value;
        if constexpr (b < NBLK) {
            constexpr int k0 = (b / TPW) * KC, c = b % TPW;
            // (the scheduler would hoist every block's loads to the top of the step -- independent loads --
            // and spill a thousand registers: nothing moves across a block boundary)
            __builtin_amdgcn_sched_barrier(0);
            big_issue<TPW, (b + RING - 1) % NBLK>(ring, Bp, w, lane);
            // the tile's operands of a K chunk serve all TPW column tiles; those of the NEXT chunk are read
            // while this chunk's matrix instructions run (the read latency was exposed once per block)
            if constexpr (c == 0 && k0 + KC < KK) {
#pragma unroll
                for (int k2 = 0; k2 < KC / 2; ++k2)
                    av[((b / TPW) + 1) & 1][k2] = *reinterpret_cast<const tile_d2 *>(X + xr + k0 + KC + 2 * k2);
            }
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[(b / TPW) & 1][kk >> 1][kk & 1], ring.v[b % RING][kk >> 1][kk & 1],
                                                              (b / TPW == 0 && kk == 0) ? wide_d4{0.0, 0.0, 0.0, 0.0} : acc[c],
                                                              0, 0, 0);
            self(self, tile_ic<b + 1>{});
        }
    };
#pragma unroll
    for (int c = 0; c < TPW; ++c)
        acc[c] = wide_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k2 = 0; k2 < KC / 2; ++k2)
        av[0][k2] = *reinterpret_cast<const tile_d2 *>(X + xr + 2 * k2);
    for_blocks(for_blocks, tile_ic<0>{});
    __builtin_amdgcn_sched_barrier(0);
}

// The packed matrix is the same at every step, so the compiler hoists its loads out of the time loop --
// i.e. tries to keep all of A in registers, which works up to 192 states (144 doubles per lane, and is
// then the best there is) and spills thousands of registers beyond.  An opaque copy of the pointer per
// step keeps the loads where they are written.
template <int TPW>
__device__ __forceinline__ const double *big_stream_ptr(const double *Bp)
{
    if constexpr (TPW > 3)
        asm volatile("" : "+s"(Bp));
    return Bp;
}

// what the emission row of a step is computed from (my four rows)
template <int KIND>
struct BigObs {
    double o[4];
    int sym[4];
};

// =========================================================================================
// k_big_fwd: normalised alpha rows (row-major) for the main part of every segment, the segment's
// log-likelihood sum_t log c_t, the vectors at the segment entry (after the warm-up) and exit.
// =========================================================================================
template <int TPW, int KIND>
__global__ __launch_bounds__(256) void k_big_fwd(const WideModel m, const double *__restrict__ Bf, const int64_t *off,
                                                 const Segs sg, const TilePlan tp, const void *obs_rm,
                                                 double *alpha_rm, double *logL_seg, double *a_entry, double *a_exit,
                                                 unsigned int *flags)
{
    using G = BigGeo<TPW>;
    constexpr int PX = G::PX, KK = G::KK;
    extern __shared__ __attribute__((aligned(16))) double big_smem[];
    double *sX = big_smem;               // [2][16][PX]
    double *sS = big_smem + 2 * 16 * PX; // [3][64] partial row sums
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, q = lane >> 4;
    const int n = m.n;

    int seg[4], nst[4], r0[4];
    int64_t ob[4];
    bool fs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int sgi = tp.tile_seg[(int64_t)blockIdx.x * 16 + q + 4 * r];
        seg[r] = sgi;
        nst[r] = r0[r] = 0;
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
    }
    const int nmax = tile_all_max(max(max(nst[0], nst[1]), max(nst[2], nst[3])));
    for (int e = tid; e < 2 * 16 * PX; e += 256)
        sX[e] = (e % PX) < n ? 1.0 / (double)n : 0.0; // warm-ups start from the uniform vector
    bool real[TPW];
    double mu_j[TPW], ga_j[TPW], gb_j[TPW], pi_j[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int j = 16 * (w + 4 * c) + s;
        real[c] = j < n;
        mu_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.mu[j] : 0.0;
        ga_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.ga[j] : 0.0;
        gb_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.gb[j] : 1.0;
        pi_j[c] = real[c] ? m.pi[j] : 0.0;
    }
    int xw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        xw[r] = tile_prow(q + 4 * r) * PX;
    const int xr = tile_prow(s) * PX + q * KK;
    auto obs_at = [&](BigObs<KIND> &in, int rs) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t g = ob[r] + min(rs, nst[r] > 0 ? nst[r] - 1 : 0);
            if constexpr (KIND == EMIT_GAUSS)
                in.o[r] = static_cast<const double *>(obs_rm)[g];
            else if constexpr (KIND == EMIT_DISC)
                in.sym[r] = static_cast<const int32_t *>(obs_rm)[g];
        }
    };
    auto emit = [&](const BigObs<KIND> &in, int rs, double (&p)[TPW][4]) __attribute__((always_inline)) {
#ifdef BIG_X_NOEMIT
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                p[c][r] = real[c] ? 0.05 : 0.0;
        return;
#endif
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            const int j = 16 * (w + 4 * c) + s;
            if constexpr (KIND == EMIT_GAUSS) {
                double d[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    d[r] = in.o[r] - mu_j[c];
                gauss_pdf4_issue(d, ga_j[c], gb_j[c], m.gmg, p[c]); // (lanes without a state: a = 0, b = 1 -> 0)
            } else if constexpr (KIND == EMIT_DISC) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    p[c][r] = real[c] ? m.B[(int64_t)j * m.M + in.sym[r]] : 0.0;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t g = ob[r] + min(rs, nst[r] > 0 ? nst[r] - 1 : 0);
                    p[c][r] = real[c] ? static_cast<const double *>(obs_rm)[g * n + j] : 0.0;
                }
            }
        }
    };
    // alpha row of my rows at the current step (running pointers: no 64-bit multiply per store; the global address
    // space spelled out -- as plain pointers in an array they became flat stores, which count against the LDS
    // counter too: every wait for the next product's operands waited for the alpha rows to reach memory)
    typedef double __attribute__((address_space(1))) big_gdouble;
    big_gdouble *arow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        arow[r] = (big_gdouble *)(alpha_rm + ob[r] * n);
    double lm[4] = {1.0, 1.0, 1.0, 1.0}; // log-likelihood of my rows: product of the c_t, mantissa ...
    int le[4] = {0, 0, 0, 0};            // ... and exponent
    unsigned int trouble = 0u;
    BigObs<KIND> cur, nxt;
    obs_at(cur, 0);
    BigRing<TPW> ring;
    big_prime<TPW>(ring, Bf, w, lane);
    __syncthreads();
    for (int rs = 0; rs < nmax; ++rs) {
        const double *X = sX + (rs & 1) * 16 * PX;
        double *Xn = sX + ((rs & 1) ^ 1) * 16 * PX;
        obs_at(nxt, rs + 1); // (the next step's observations are on their way during the product)
        wide_d4 acc[TPW];
        big_product<TPW>(X, xr, big_stream_ptr<TPW>(Bf), w, lane, ring, acc);
        double p[TPW][4];
        emit(cur, rs, p);
        double v[TPW][4], ps[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[c][r] = ((fs[r] && rs == 0) ? pi_j[c] : acc[c][r]) * p[c][r]; // _hidden.c:29-33 / :44-52
                ps[r] += v[c][r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double t = row16_sum(ps[r]);
            if (s == 0)
                sS[16 * w + q + 4 * r] = t;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rho = q + 4 * r;
            const double cs = (sS[rho] + sS[16 + rho]) + (sS[32 + rho] + sS[48 + rho]); // c_t, _hidden.c:53-56
            const bool act = rs < nst[r], mainp = act && rs >= r0[r];
            trouble |= (act && !(cs > 0x1p-1000 && cs < 0x1p1000)) ? 1u : 0u;
            const double inv = act ? fast_rcp(cs) : 0.0; // (1 ulp; the IEEE division sequence is 30 instructions)
#ifndef BIG_X_NOLOG
            // _hidden.c:57-66 sums log c_t; here the c_t of a row are multiplied up (mantissa in [0.5, 1) and
            // exponent apart: no range to leave) and the logarithm is taken once at the end -- the whole
            // wavefront would execute log() at every step for four lanes' sake
            if (mainp) {
                int e;
                lm[r] = frexp(lm[r] * cs, &e);
                le[r] += e;
            }
#endif
#pragma unroll
            for (int c = 0; c < TPW; ++c) {
                const int j = 16 * (w + 4 * c) + s;
                const double a = v[c][r] * inv;
                Xn[xw[r] + j] = a;
                if (real[c] && act) {
                    if (mainp) {
#ifndef BIG_X_NOSTORE
                        arow[r][16 * (w + 4 * c) + s] = a; // (= alpha_rm[(ob[r] + rs) * n + j])
#endif
                    }
                    else if (rs == r0[r] - 1)
                        a_entry[(int64_t)seg[r] * n + j] = a;
                    if (rs == nst[r] - 1)
                        a_exit[(int64_t)seg[r] * n + j] = a;
                }
            }
        }
        cur = nxt;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            arow[r] += n;
        __syncthreads();
    }
    if (s == 0 && w == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (seg[r] >= 0)
                logL_seg[seg[r]] = log(lm[r]) + (double)le[r] * 0.693147180559945309417232121458;
    }
    if (trouble)
        atomicOr(&flags[2], trouble);
}

// =========================================================================================
// k_big_bwd: beta in registers only; gamma, the rows W_{t-1} = p_t o beta_t / S_{t-1} for the xi GEMM,
// emission statistics per tile.
//   part  [tile][3 n]  sum gamma | (gauss) sum gamma d | sum gamma d^2
//   dstat [tile][4][n][M]  (discrete: one table per lane row q, so that no two lanes share an entry)
// One iteration (time t of a row): x = p_t o beta_t into LDS; beta_raw = x-tile . A^T; with
// alpha_{t-1}: S = sum_i alpha_{t-1}[i] beta_raw[i] (the reference's normaliser of the transition
// t-1 -> t, _hidden.c:168-179, and of gamma_{t-1}, hidden/api.py:176-186), W_{t-1} = x / S,
// beta_{t-1} = beta_raw / sum(beta_raw) (_hidden.c:100-108); gamma_t = alpha_t o beta_t / sum.
// =========================================================================================
template <int TPW, int KIND>
__global__ __launch_bounds__(256) void k_big_bwd(const WideModel m, const double *__restrict__ Bb, const int64_t *off,
                                                 const Segs sg, const TilePlan tp, const void *obs_rm,
                                                 const double *alpha_rm, double *gamma_rm, double *gamma0,
                                                 double *part, double *dstat, double *b_exit, double *b_entry,
                                                 unsigned int *flags, double *Wg)
{
    using G = BigGeo<TPW>;
    constexpr int PX = G::PX, KK = G::KK;
    extern __shared__ __attribute__((aligned(16))) double big_smem[];
    double *sX = big_smem;               // [2][16][PX] (one buffer used per step, alternating)
    double *sS = big_smem + 2 * 16 * PX; // [3][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = lane & 15, q = lane >> 4;
    const int n = m.n;

    // my four rows: step us of the tile is time ttop - us of the row
    int seg[4], nwarm[4], nst[4], trj[4], ttop[4];
    int64_t gtop[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int sgi = tp.tile_seg[(int64_t)blockIdx.x * 16 + q + 4 * r];
        seg[r] = sgi;
        nwarm[r] = nst[r] = trj[r] = ttop[r] = 0;
        gtop[r] = 0;
        if (sgi >= 0 && sg.len[sgi] > 0) {
            const int k = sg.traj[sgi];
            const int64_t o0 = off[k], T = off[k + 1] - o0, t0 = sg.t0[sgi], t1 = t0 + sg.len[sgi];
            const int64_t te = (t1 - 1 + sg.W < T - 1) ? t1 - 1 + sg.W : T - 1;
            nwarm[r] = t1 < T ? (int)(te - t1) + 1 : 0;
            nst[r] = nwarm[r] + (int)(t1 - t0);
            ttop[r] = (int)(t1 - 1 + nwarm[r]);
            gtop[r] = o0 + ttop[r];
            trj[r] = k;
        }
    }
    const int nmax = tile_all_max(max(max(nst[0], nst[1]), max(nst[2], nst[3])));
    bool real[TPW];
    double mu_j[TPW], ga_j[TPW], gb_j[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        const int j = 16 * (w + 4 * c) + s;
        real[c] = j < n;
        mu_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.mu[j] : 0.0;
        ga_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.ga[j] : 0.0;
        gb_j[c] = (KIND == EMIT_GAUSS && real[c]) ? m.gb[j] : 1.0;
    }
    int xw[4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
        xw[r] = tile_prow(q + 4 * r) * PX;
    const int xr = tile_prow(s) * PX + q * KK;
    double *mytab = nullptr;
    if constexpr (KIND == EMIT_DISC) {
        mytab = dstat + ((int64_t)blockIdx.x * 4 + q) * n * m.M;
#pragma unroll
        for (int c = 0; c < TPW; ++c)
            if (real[c])
                for (int z = 0; z < m.M; ++z)
                    mytab[(int64_t)(16 * (w + 4 * c) + s) * m.M + z] = 0.0;
    }
    auto obs_at = [&](BigObs<KIND> &in, int us) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t g = gtop[r] - min(us, nst[r] > 0 ? nst[r] - 1 : 0);
            if constexpr (KIND == EMIT_GAUSS)
                in.o[r] = static_cast<const double *>(obs_rm)[g];
            else if constexpr (KIND == EMIT_DISC)
                in.sym[r] = static_cast<const int32_t *>(obs_rm)[g];
        }
    };
    auto emit = [&](const BigObs<KIND> &in, int us, double (&p)[TPW][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < TPW; ++c) {
            const int j = 16 * (w + 4 * c) + s;
            if constexpr (KIND == EMIT_GAUSS) {
                double d[4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    d[r] = in.o[r] - mu_j[c];
                gauss_pdf4_issue(d, ga_j[c], gb_j[c], m.gmg, p[c]);
            } else if constexpr (KIND == EMIT_DISC) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    p[c][r] = real[c] ? m.B[(int64_t)j * m.M + in.sym[r]] : 0.0;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t g = gtop[r] - min(us, nst[r] > 0 ? nst[r] - 1 : 0);
                    p[c][r] = real[c] ? static_cast<const double *>(obs_rm)[g * n + j] : 0.0;
                }
            }
        }
    };
    // alpha at time ttop - us of my rows (zero where the row does not read it: warm-up, finished)
    auto alpha_at = [&](int us, double (&a)[TPW][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool want = us >= nwarm[r] - 1 && us < nst[r] + 1 && ttop[r] - us >= 0;
            const double *src = alpha_rm + (gtop[r] - us) * n;
#pragma unroll
            for (int c = 0; c < TPW; ++c)
                a[c][r] = (want && real[c]) ? src[16 * (w + 4 * c) + s] : 0.0;
        }
    };
    double beta[TPW][4], acur[TPW][4], sgm[TPW], sd[TPW], sdd[TPW];
#pragma unroll
    for (int c = 0; c < TPW; ++c) {
        sgm[c] = sd[c] = sdd[c] = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            beta[c][r] = real[c] ? 1.0 / (double)n : 0.0; // _hidden.c:79-88
    }
    unsigned int trouble = 0u;
    BigObs<KIND> cur, nxt;
    obs_at(cur, 0);
    alpha_at(0, acur);
    BigRing<TPW> ring;
    big_prime<TPW>(ring, Bb, w, lane);
    for (int us = 0; us < nmax; ++us) {
        double *X = sX + (us & 1) * 16 * PX;
        obs_at(nxt, us + 1);
        double p[TPW][4], pg[4] = {0.0, 0.0, 0.0, 0.0};
        emit(cur, us, p);
        bool mainp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mainp[r] = us >= nwarm[r] && us < nst[r];
            if (us == nwarm[r] && nst[r] > 0 && nwarm[r] > 0) { // the warm-up's beta, for the boundary check
#pragma unroll
                for (int c = 0; c < TPW; ++c)
                    if (real[c])
                        b_exit[(int64_t)seg[r] * n + 16 * (w + 4 * c) + s] = beta[c][r];
            }
        }
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                X[xw[r] + 16 * (w + 4 * c) + s] = p[c][r] * beta[c][r];  // x = p_t o beta_t
                pg[r] = fma(acur[c][r], beta[c][r], pg[r]);              // gamma_t's normaliser
            }
        __syncthreads();
        wide_d4 acc[TPW];
        big_product<TPW>(X, xr, big_stream_ptr<TPW>(Bb), w, lane, ring, acc);
        // alpha_{t-1}.  (Requested BEFORE the product these loads queue ahead of the ring's: memory operations retire
        // in order, the first block of A then waits for HBM -- measured 7.3 -> 7.8 ms at 256 states, twice, round 5)
        double aprev[TPW][4];
        alpha_at(us + 1, aprev);
        double pb[4] = {0.0, 0.0, 0.0, 0.0}, pS[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < TPW; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pb[r] += acc[c][r];
                pS[r] = fma(aprev[c][r], acc[c][r], pS[r]);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double t0 = row16_sum(pg[r]), t1 = row16_sum(pb[r]), t2 = row16_sum(pS[r]);
            if (s == 0) {
                sS[16 * w + q + 4 * r] = t0;
                sS[64 + 16 * w + q + 4 * r] = t1;
                sS[128 + 16 * w + q + 4 * r] = t2;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rho = q + 4 * r;
            const double Sg = (sS[rho] + sS[16 + rho]) + (sS[32 + rho] + sS[48 + rho]);
            const double Sb = (sS[64 + rho] + sS[80 + rho]) + (sS[96 + rho] + sS[112 + rho]);
            const double SS = (sS[128 + rho] + sS[144 + rho]) + (sS[160 + rho] + sS[176 + rho]);
            const bool act = us < nst[r];
            const int tt = ttop[r] - us;         // the time of this step
            const bool trans = mainp[r] && tt > 0; // the transition tt - 1 -> tt is this row's
            trouble |= (act && !(Sb > 0x1p-1000 && Sb < 0x1p1000)) ? 2u : 0u;
            trouble |= (mainp[r] && !(Sg > 0x1p-1000 && Sg < 0x1p1000)) ? 4u : 0u;
            trouble |= (trans && !(SS > 0x1p-1000 && SS < 0x1p1000)) ? 8u : 0u;
            const double ig = mainp[r] ? fast_rcp(Sg) : 0.0, ib = act ? fast_rcp(Sb) : 0.0, iS = trans ? fast_rcp(SS) : 0.0;
            const int64_t grow = (gtop[r] - us) * n;
#pragma unroll
            for (int c = 0; c < TPW; ++c) {
                const int j = 16 * (w + 4 * c) + s;
                const double gam = acur[c][r] * beta[c][r] * ig; // (beta is still beta_t here)
                sgm[c] += gam;
                if constexpr (KIND == EMIT_GAUSS) {
                    const double d = cur.o[r] - mu_j[c];
                    const double gd = gam * d;
                    sd[c] += gd;
                    sdd[c] = fma(gd, d, sdd[c]);
                }
                if (real[c] && mainp[r]) {
                    if constexpr (KIND == EMIT_DISC)
                        mytab[(int64_t)j * m.M + cur.sym[r]] += gam;
                    if (gamma_rm)
                        gamma_rm[grow + j] = gam;
                    if (tt == 0)
                        gamma0[(int64_t)trj[r] * n + j] = gam;
                    if (trans)
                        Wg[grow - n + j] = X[xw[r] + j] * iS; // row t - 1 (x is still in this step's LDS buffer)
                }
                const double b = acc[c][r] * ib; // beta_{t-1}
                if (real[c] && us == nst[r] - 1 && tt > 0)
                    b_entry[(int64_t)seg[r] * n + j] = b; // beta one step before the segment
                beta[c][r] = b;
                acur[c][r] = aprev[c][r];
            }
        }
        cur = nxt;
    }
    if (trouble)
        atomicOr(&flags[2], trouble);
    double *mypart = part + (int64_t)blockIdx.x * 3 * n;
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
            mypart[j] = a0;
            mypart[n + j] = a1;
            mypart[2 * n + j] = a2;
        }
    }
}

// =========================================================================================
// k_big_xi_gemm: C' = alpha^T W over all time steps (the xi counts of _hidden.c:148-183 up to the factor
// A[i][j], W_t = p_{t+1} o beta_{t+1} / S_t as k_big_bwd left them; W of a trajectory's last step is zero).
// Workgroup (bi, bj, slab): the 32 TI x 32 TI block (bi, bj) of C' over time slab `slab` (TI = 4: 128 x 128;
// TI = 3: 96 x 96, for 65 .. 96 states, where the larger block is mostly padding); wavefront (wi, wj) its
// quarter as TI x TI tiles of v_mfma_f64_16x16x4 -- K = four consecutive time steps, the operands
// straight from HBM in operand order (lane (s, q): row t + q, state 16 I + s: 128 contiguous bytes per
// sixteen lanes), three K steps ahead.  Every row of alpha and W is read ceil(n / 128) times in all.
//   xipart [slab][n][n]   (summed in slab order by k_big_finalize)
// =========================================================================================
template <int TI>
[[maybe_unused]] static __global__ __launch_bounds__(256) void k_big_xi_gemm(const double *__restrict__ alpha,
                                                                           const double *__restrict__ W, int64_t total,
                                                                           int n, int nb, int nsplit, double *xipart)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int s = lane & 15, q = lane >> 4;
    const int blk = blockIdx.x % (nb * nb), slab = blockIdx.x / (nb * nb);
    const int i0 = 32 * TI * (blk / nb) + 16 * TI * (wv >> 1), j0 = 32 * TI * (blk % nb) + 16 * TI * (wv & 1);
    // time slab, a multiple of four steps
    const int64_t per = ((total + nsplit - 1) / nsplit + 3) & ~(int64_t)3;
    const int64_t tb = (int64_t)slab * per, te = tb + per < total ? tb + per : total;
    wide_d4 acc[TI][TI];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j)
            acc[i][j] = wide_d4{0.0, 0.0, 0.0, 0.0};
    bool ci[TI], cj[TI];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        ci[i] = i0 + 16 * i + s < n;
        cj[i] = j0 + 16 * i + s < n;
    }
    constexpr int PF = 3;
    double ra[PF + 1][TI], rb[PF + 1][TI];
    auto load = [&](int slot, int64_t t) __attribute__((always_inline)) {
        const bool in = t + q < te;
        const double *pa = alpha + (t + q) * n + i0 + s, *pb = W + (t + q) * n + j0 + s;
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            ra[slot][i] = (in && ci[i]) ? pa[16 * i] : 0.0;
            rb[slot][i] = (in && cj[i]) ? pb[16 * i] : 0.0;
        }
    };
#pragma unroll
    for (int u = 0; u < PF; ++u)
        load(u, tb + 4 * u);
    auto body = [&](auto uc, int64_t t) __attribute__((always_inline)) {
        constexpr int u = decltype(uc)::value;
        load((u + PF) & 3, t + 4 * PF);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[u][i], rb[u][j], acc[i][j], 0, 0, 0);
    };
    for (int64_t t = tb; t < te; t += 16) { // (four K steps per round: the ring slots are compile-time)
        body(tile_ic<0>{}, t);
        body(tile_ic<1>{}, t + 4);
        body(tile_ic<2>{}, t + 8);
        body(tile_ic<3>{}, t + 12);
    }
    double *out = xipart + (int64_t)slab * n * n;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + 16 * i + q + 4 * r, col = j0 + 16 * j + s; // (C/D layout)
                if (row < n && col < n)
                    out[(int64_t)row * n + col] = acc[i][j][r];
            }
}

// packed statistics (bhmm_amd.h layout): C = A o sum of the xi GEMM's split partials, the rest from the
// tiles' partial blocks.  One wavefront per output entry, fixed summation order.
template <int KIND>
__global__ __launch_bounds__(64) void k_big_finalize(const WideModel m, int K, int ntiles, int nsplit,
                                                     const double *xipart, const double *part, const double *dstat,
                                                     const double *logL_k, const double *gamma0, double *stats)
{
    const int n = m.n;
    const int64_t nn = (int64_t)n * n;
    const int nE = KIND == EMIT_GAUSS ? 2 * n : 0;
    const int64_t MN = KIND == EMIT_DISC ? (int64_t)n * m.M : 0;
    const int64_t oG0 = 1, oC = 1 + n, oSG = oC + nn, oE = oSG + n;
    const int lane = threadIdx.x;
    for (int64_t e = blockIdx.x; e < nn + n + nE + MN + n + 1; e += gridDim.x) {
        double s = 0.0;
        int64_t f = e;
        if (f < nn) {
            for (int k = lane; k < nsplit; k += 64)
                s += xipart[(int64_t)k * nn + f];
            s = wave_sum(s);
            if (lane == 0)
                stats[oC + f] = s * m.A[f];
            continue;
        }
        f -= nn;
        if (f < n + nE) {
            for (int k = lane; k < ntiles; k += 64)
                s += part[(int64_t)k * 3 * n + f];
            s = wave_sum(s);
            if (lane == 0)
                stats[(f < n ? oSG : oE - n) + f] = s;
            continue;
        }
        f -= n + nE;
        if (f < MN) {
            for (int k = lane; k < 4 * ntiles; k += 64)
                s += dstat[(int64_t)k * MN + f];
            s = wave_sum(s);
            if (lane == 0)
                stats[oE + f] = s;
            continue;
        }
        f -= MN;
        if (f < n) {
            for (int k = lane; k < K; k += 64)
                s += gamma0[(int64_t)k * n + f];
            s = wave_sum(s);
            if (lane == 0)
                stats[oG0 + f] = s;
            continue;
        }
        for (int k = lane; k < K; k += 64)
            s += logL_k[k];
        s = wave_sum(s);
        if (lane == 0)
            stats[0] = s;
    }
}

} // namespace bhmm
