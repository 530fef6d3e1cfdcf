// wide_api.hip -- host side of the 9..64-state kernel family (wide_kernels.hpp).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "host_common.hpp"
#include "plan.hpp"
#include "wide_kernels.hpp"
#include "tile_kernels.hpp"

namespace bhmm {
int invalid_arg(const std::string &msg);

static int wide_np(int n) { return n <= 16 ? 16 : (n <= 32 ? 32 : 64); }

// upload the model into ctx->d_wmodel and describe it
static int wide_model(bhmm_ctx *c, int kind, const double *A, const double *pi, const double *par0,
                      const double *par1, WideModel &m)
{
    const int n = c->n;
    std::vector<double> h((size_t)n * n + 7 * n, 0.0);
    memcpy(h.data(), A, sizeof(double) * n * n);
    double *hp = h.data() + (size_t)n * n;
    for (int i = 0; i < n; ++i) {
        hp[i] = pi ? pi[i] : 0.0;
        if (kind == EMIT_GAUSS) {
            hp[n + i] = par0[i];
            hp[2 * n + i] = 1.0 / par1[i];
            hp[3 * n + i] = 1.0 / (sqrt(2.0 * M_PI) * par1[i]);
            hp[4 * n + i] = par1[i];
        }
    }
    m.gmg = 0.0;
    if (kind == EMIT_GAUSS)
        gauss_pdf_constants(n, n, par1, hp + 5 * n, hp + 6 * n, &m.gmg);
    int rc = c->d_wmodel.ensure(h.size());
    if (rc)
        return rc;
    BHMM_HIP(hipMemcpyAsync(c->d_wmodel.p, h.data(), h.size() * sizeof(double),
                            hipMemcpyHostToDevice, c->stream));
    m.A = c->d_wmodel.p;
    m.pi = m.A + (size_t)n * n;
    m.mu = m.pi + n;
    m.isig = m.mu + n;
    m.cnorm = m.isig + n;
    m.sigma = m.cnorm + n;
    m.ga = m.sigma + n;
    m.gb = m.ga + n;
    m.n = n;
    m.M = c->M;
    m.B = nullptr;
    if (kind == EMIT_DISC) {
        if ((rc = c->d_Brm.ensure((size_t)n * c->M)))
            return rc;
        BHMM_HIP(hipMemcpyAsync(c->d_Brm.p, par0, (size_t)n * c->M * sizeof(double),
                                hipMemcpyHostToDevice, c->stream));
        m.B = c->d_Brm.p;
    }
    BHMM_HIP(hipStreamSynchronize(c->stream)); // h is a temporary
    return BHMM_OK;
}

static Segs segs_of(bhmm_ctx *c, int which)
{
    Segs sg;
    sg.traj = c->d_wseg_traj[which].p;
    sg.t0 = c->d_wseg_t0[which].p;
    sg.len = c->d_wseg_len[which].p;
    sg.nseg = c->w_nseg[which];
    // (segment starts and the warm-up are multiples of four: the lazily scaled passes then rescale on
    // the steps t % 4 == 3 whatever plan each of them runs on)
    sg.W = (c->spec_W + 3) & ~3;
    sg.fmid = (which == 1 && c->w_nseg[2] > c->w_nseg[1]) ? c->d_wseg_fmid.p : nullptr;
    return sg;
}

// the plan the forward pass runs on when the backward pass runs on `which`
static int wide_fwd_plan(const bhmm_ctx *c, int which)
{
    return (which == 1 && c->w_nseg[2] > c->w_nseg[1]) ? 2 : which;
}

// 64 states: the lazily scaled E-step runs on the row-batched matrix-core kernels (tile_kernels.hpp)
// 33..64 states (64 lanes anyway: four column tiles, the last ones partly padded) and, through tile_gen.hip,
// 65..128
static bool wide_tile(const bhmm_ctx *c)
{
    return c->tile_latched && ((c->n > 32 && c->n <= 64 && !c->gen) || (c->gen && c->n <= 512));
}

template <int KIND>
static int tile_launch_fwd(bhmm_ctx *c, const WideModel &m, int which)
{
    const Segs sg = segs_of(c, which);
    const TilePlan tp{c->d_tile_seg[which].p, c->w_ntiles[which]};
    // BHMM_AMD_TILE_PROBE=1: cycles of the phases of a step (workgroup 0), printed after the pass -- in a library
    // built with -DBHMM_TILE_PROBE_BUILD; the production build compiles the probe out of the kernels (the phase
    // cycles then print as zeros, the kernel time of BHMM_AMD_TILE_PROBE=2 is unaffected)
    static const bool probe_on = getenv("BHMM_AMD_TILE_PROBE") != nullptr;
    unsigned long long *probe = nullptr;
    if (probe_on) {
        int rc = c->d_probe.ensure(4096);
        if (rc)
            return rc;
        probe = reinterpret_cast<unsigned long long *>(c->d_probe.p);
        BHMM_HIP(hipMemsetAsync(probe, 0, 64, c->stream));
        BHMM_HIP(hipEventRecord(c->ev[5], c->stream));
    }
    static const bool probe_in = probe_on && atoi(getenv("BHMM_AMD_TILE_PROBE")) == 1; // (2: kernel times only)
    lds_poison(c->stream);
    if (c->n == 64)
        hipLaunchKernelGGL((k_tile_fwd<4, KIND, true, true>), dim3(tp.ntiles), dim3(TILE_THREADS), 0, c->stream, m,
                           (const int64_t *)c->d_offsets.p, sg, tp, (const void *)c->d_obs_rm.p, c->d_alpha_rm.p,
                           c->d_wexp.p, c->d_wePseg.p, c->d_waentry.p, c->d_waexit.p, c->d_specres.p,
                           probe_in ? probe : (unsigned long long *)nullptr);
    else
        hipLaunchKernelGGL((k_tile_fwd<4, KIND, false, true>), dim3(tp.ntiles), dim3(TILE_THREADS), 0, c->stream, m,
                           (const int64_t *)c->d_offsets.p, sg, tp, (const void *)c->d_obs_rm.p, c->d_alpha_rm.p,
                           c->d_wexp.p, c->d_wePseg.p, c->d_waentry.p, c->d_waexit.p, c->d_specres.p,
                           probe_in ? probe : (unsigned long long *)nullptr);
    BHMM_HIP(hipGetLastError());
    if (probe_on) {
        unsigned long long h[8];
        BHMM_HIP(hipEventRecord(c->ev[1], c->stream));
        BHMM_HIP(hipMemcpyAsync(h, probe, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        BHMM_HIP(hipStreamSynchronize(c->stream));
        float kms = 0.f;
        (void)hipEventElapsedTime(&kms, c->ev[5], c->ev[1]);
        fprintf(stderr, "tile fwd kernel %.3f ms; ", kms);
        if (h[3])
            fprintf(stderr, "tile fwd probe: matrix [operands+matrix %.0f | emission row, write %.0f | barrier %.0f] "
                        "stream [store, loads %.0f | emission %.0f | barrier %.0f] cycles/step (%llu steps)\n",
                (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], (double)h[4] / h[7],
                (double)h[5] / h[7], (double)h[6] / h[7], h[3]);
    }
    hipLaunchKernelGGL(k_tile_logl, dim3((sg.nseg + 15) / 16), dim3(256), 0, c->stream, sg, c->n,
                       (const double *)c->d_waentry.p, (const double *)c->d_waexit.p,
                       (const int32_t *)c->d_wePseg.p, c->d_wlogLseg.p, c->d_specres.p);
    hipLaunchKernelGGL(k_logl, dim3(c->K), dim3(64), 0, c->stream,
                       (const int32_t *)c->d_wseg_traj0[which].p, c->K,
                       (const double *)c->d_wlogLseg.p, c->d_logLk.p);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

template <int KIND>
static int tile_launch_bwd(bhmm_ctx *c, const WideModel &m, int which, bool store_gamma, double *stats_dev)
{
    const Segs sg = segs_of(c, which);
    const TilePlan tp{c->d_tile_segb[which].p, c->w_ntilesb[which]};
    double *gam = store_gamma ? c->d_gamma_ci.p : (double *)nullptr;
    static const bool probe_on = getenv("BHMM_AMD_TILE_PROBE") != nullptr;
    unsigned long long *probe = nullptr;
    if (probe_on) {
        int rc = c->d_probe.ensure(4096);
        if (rc)
            return rc;
        probe = reinterpret_cast<unsigned long long *>(c->d_probe.p) + 16;
        BHMM_HIP(hipMemsetAsync(probe, 0, 128, c->stream));
        BHMM_HIP(hipEventRecord(c->ev[5], c->stream));
    }
    lds_poison(c->stream);
    if (c->n == 64)
        hipLaunchKernelGGL((k_tile_bwd<4, KIND, true, false, true>), dim3(tp.ntiles), dim3(TILE_THREADS), 0, c->stream, m,
                           (const int64_t *)c->d_offsets.p, sg, tp, (const void *)c->d_obs_rm.p,
                           (const double *)c->d_alpha_rm.p, (const int32_t *)c->d_wexp.p, gam, c->d_gamma0.p,
                           c->d_partials.p, c->d_dpartials.p, c->d_wbexit.p, c->d_wbentry.p, c->d_specres.p,
                           (double *)nullptr,
                           atoi(getenv("BHMM_AMD_TILE_PROBE") ? getenv("BHMM_AMD_TILE_PROBE") : "0") == 1 ? probe : (unsigned long long *)nullptr);
    else
        hipLaunchKernelGGL((k_tile_bwd<4, KIND, false, false, true>), dim3(tp.ntiles), dim3(TILE_THREADS), 0, c->stream, m,
                           (const int64_t *)c->d_offsets.p, sg, tp, (const void *)c->d_obs_rm.p,
                           (const double *)c->d_alpha_rm.p, (const int32_t *)c->d_wexp.p, gam, c->d_gamma0.p,
                           c->d_partials.p, c->d_dpartials.p, c->d_wbexit.p, c->d_wbentry.p, c->d_specres.p,
                           (double *)nullptr,
                           atoi(getenv("BHMM_AMD_TILE_PROBE") ? getenv("BHMM_AMD_TILE_PROBE") : "0") == 1 ? probe : (unsigned long long *)nullptr);
    BHMM_HIP(hipGetLastError());
    if (probe_on) {
        unsigned long long h[16];
        BHMM_HIP(hipEventRecord(c->ev[3], c->stream));
        BHMM_HIP(hipMemcpyAsync(h, probe, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        BHMM_HIP(hipStreamSynchronize(c->stream));
        float kms = 0.f;
        (void)hipEventElapsedTime(&kms, c->ev[5], c->ev[3]);
        fprintf(stderr, "tile bwd kernel %.3f ms\n", kms);
        for (int o = 0; o < 16; o += 8)
            if (h[o + 4])
                fprintf(stderr, "tile bwd probe (%s steps): operands+matrix %.0f | rescale, x' write %.0f | xi, statistics %.0f | "
                                "barrier %.0f cycles/step (%llu steps)\n", o ? "main" : "warm-up", (double)h[o] / h[o + 4],
                        (double)h[o + 1] / h[o + 4], (double)h[o + 2] / h[o + 4], (double)h[o + 3] / h[o + 4], h[o + 4]);
    }
    const int n = c->n;
    const int nfin = n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0) + (KIND == EMIT_DISC ? n * c->M : 0) +
                     n + 1;
    hipLaunchKernelGGL((k_wide_finalize<KIND>), dim3(nfin), dim3(64), 0, c->stream, m, c->K, tp.ntiles,
                       4 * tp.ntiles, (const double *)c->d_partials.p, (const double *)c->d_dpartials.p,
                       (const double *)c->d_logLk.p, (const double *)c->d_gamma0.p, stats_dev);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

template <int NP, int KIND>
static int wide_launch_fwd(bhmm_ctx *c, const WideModel &m, int which, bool lazy = false)
{
    constexpr int GP = 64 / NP;
    if (lazy && wide_tile(c))
        return tile_launch_fwd<KIND>(c, m, which);
    which = wide_fwd_plan(c, which);
    const Segs sg = segs_of(c, which);
    if (lazy && NP == 64 && c->n == 64) // the shape of BASELINE configs[3]
        hipLaunchKernelGGL((k_wide_fwd<NP, KIND, true, NP == 64>), dim3((sg.nseg + GP - 1) / GP),
                           dim3(64), 0, c->stream, m, (const int64_t *)c->d_offsets.p, sg,
                           (const void *)c->d_obs_rm.p, c->d_alpha_rm.p, c->d_wlogLseg.p,
                           c->d_waentry.p, c->d_waexit.p, c->d_specres.p);
    else if (lazy)
        hipLaunchKernelGGL((k_wide_fwd<NP, KIND, true>), dim3((sg.nseg + GP - 1) / GP), dim3(64), 0,
                           c->stream, m, (const int64_t *)c->d_offsets.p, sg,
                           (const void *)c->d_obs_rm.p, c->d_alpha_rm.p, c->d_wlogLseg.p,
                           c->d_waentry.p, c->d_waexit.p, c->d_specres.p);
    else
        hipLaunchKernelGGL((k_wide_fwd<NP, KIND, false>), dim3((sg.nseg + GP - 1) / GP), dim3(64), 0,
                           c->stream, m, (const int64_t *)c->d_offsets.p, sg,
                           (const void *)c->d_obs_rm.p, c->d_alpha_rm.p, c->d_wlogLseg.p,
                           c->d_waentry.p, c->d_waexit.p, (unsigned int *)nullptr);
    BHMM_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_logl, dim3(c->K), dim3(64), 0, c->stream,
                       (const int32_t *)c->d_wseg_traj0[which].p, c->K,
                       (const double *)c->d_wlogLseg.p, c->d_logLk.p);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

template <int NP, int KIND>
static int wide_launch_bwd(bhmm_ctx *c, const WideModel &m, int which, bool store_gamma,
                           double *stats_dev, bool lazy = false)
{
    constexpr int GP = 64 / NP;
    if (lazy && wide_tile(c))
        return tile_launch_bwd<KIND>(c, m, which, store_gamma, stats_dev);
    const Segs sg = segs_of(c, which);
    // A's rows in LDS for fewer than 64 lanes per segment; the 64-lane kernel keeps them in VGPRs
    const size_t sm = NP == 64 ? 0 : (size_t)(NP * wide_pitch(NP)) * sizeof(double);
    double *gam = store_gamma ? c->d_gamma_ci.p : (double *)nullptr;
    // control experiment: xi counts as a separate time-parallel GEMM
    static const bool xi_gemm = getenv("BHMM_AMD_WIDE_XI_GEMM") != nullptr;
    if (lazy && NP == 64 && c->n == 64 && xi_gemm) {
        const int nsplit = 4096;
        int rc;
        if ((rc = c->d_gW.ensure((size_t)c->total * 64)) || (rc = c->d_gxipart.ensure((size_t)nsplit * 4096)))
            return rc;
        hipLaunchKernelGGL(k_wide_zero_last_rows, dim3(c->K), dim3(64), 0, c->stream,
                           (const int64_t *)c->d_offsets.p, c->K, 64, c->d_gW.p);
        hipLaunchKernelGGL((k_wide_bwd<NP, KIND, true, NP == 64, NP == 64>), dim3((sg.nseg + GP - 1) / GP),
                           dim3(64), sm, c->stream, m, (const int64_t *)c->d_offsets.p, sg,
                           (const void *)c->d_obs_rm.p, (const double *)c->d_alpha_rm.p, gam,
                           c->d_gamma0.p, c->d_partials.p, c->d_dpartials.p, c->d_wbexit.p,
                           c->d_wbentry.p, c->d_specres.p, c->d_gW.p);
        hipLaunchKernelGGL(k_wide_xi_gemm64, dim3(nsplit), dim3(64), 0, c->stream,
                           (const double *)c->d_alpha_rm.p, (const double *)c->d_gW.p, c->total, nsplit,
                           c->d_gxipart.p);
        hipLaunchKernelGGL(k_wide_xi_reduce, dim3(16), dim3(256), 0, c->stream,
                           (const double *)c->d_gxipart.p, nsplit, c->d_partials.p);
    } else if (lazy && NP == 64 && c->n == 64)
        hipLaunchKernelGGL((k_wide_bwd<NP, KIND, true, NP == 64>), dim3((sg.nseg + GP - 1) / GP),
                           dim3(64), sm, c->stream, m, (const int64_t *)c->d_offsets.p, sg,
                           (const void *)c->d_obs_rm.p, (const double *)c->d_alpha_rm.p, gam,
                           c->d_gamma0.p, c->d_partials.p, c->d_dpartials.p, c->d_wbexit.p,
                           c->d_wbentry.p, c->d_specres.p);
    else if (lazy)
        hipLaunchKernelGGL((k_wide_bwd<NP, KIND, true>), dim3((sg.nseg + GP - 1) / GP), dim3(64), sm,
                           c->stream, m, (const int64_t *)c->d_offsets.p, sg,
                           (const void *)c->d_obs_rm.p, (const double *)c->d_alpha_rm.p, gam,
                           c->d_gamma0.p, c->d_partials.p, c->d_dpartials.p, c->d_wbexit.p,
                           c->d_wbentry.p, c->d_specres.p);
    else
        hipLaunchKernelGGL((k_wide_bwd<NP, KIND, false>), dim3((sg.nseg + GP - 1) / GP), dim3(64), sm,
                           c->stream, m, (const int64_t *)c->d_offsets.p, sg,
                           (const void *)c->d_obs_rm.p, (const double *)c->d_alpha_rm.p, gam,
                           c->d_gamma0.p, c->d_partials.p, c->d_dpartials.p, c->d_wbexit.p,
                           c->d_wbentry.p, (unsigned int *)nullptr);
    BHMM_HIP(hipGetLastError());
    const int n = c->n;
    const int nfin = n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0) + (KIND == EMIT_DISC ? n * c->M : 0) +
                     n + 1;
    hipLaunchKernelGGL((k_wide_finalize<KIND>), dim3(nfin), dim3(64), 0, c->stream, m, c->K, sg.nseg,
                       sg.nseg, (const double *)c->d_partials.p, (const double *)c->d_dpartials.p,
                       (const double *)c->d_logLk.p, (const double *)c->d_gamma0.p, stats_dev);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

#define WIDE_DISPATCH(c, KIND, fn, ...)                               \
    ((c)->N == 16 ? fn<16, KIND>(__VA_ARGS__)                         \
     : (c)->N == 32 ? fn<32, KIND>(__VA_ARGS__)                       \
                    : fn<64, KIND>(__VA_ARGS__))

// Segment length that fills the chip.  64 states: one segment per wavefront, and the recursion
// is issue-bound with a single wavefront per SIMD already -- a SIMD that gets a second wavefront
// takes almost twice as long, so the plan aims at exactly one per SIMD (or a multiple).  Fewer
// states run 64 / N segments per wavefront and are latency-bound: four wavefronts per SIMD.
static int64_t wide_fill_len(const bhmm_ctx *c)
{
    const int64_t groups_per_wave = 64 / c->N;
    if (wide_tile(c)) { // 16 segments per workgroup, tile_per_cu workgroups per compute unit
        const int64_t want = 16 * (int64_t)(c->num_simd / 4) * c->tile_per_cu;
        return ((c->total + want - 1) / want + 3) & ~(int64_t)3;
    }
    const int64_t want = (c->N == 64) ? (int64_t)c->num_simd : 4 * (int64_t)c->num_simd * groups_per_wave;
    return (c->total + want - 1) / want;
}

// segment plan `which` with segments of at most seglen steps (seglen <= 0: one per trajectory)
static int wide_plan(bhmm_ctx *c, int which, int64_t seglen, int mult = 1)
{
    plan::SegPlan sp; // (plan.hpp: pure host code, also built under the CPU sanitizers)
    plan::plan_segments(c->offsets, c->K, seglen, mult, sp);
    std::vector<int32_t> &st = sp.traj, &sl = sp.len, &s0 = sp.traj0;
    std::vector<int64_t> &stt = sp.t0;
    s0[c->K] = (int32_t)st.size();
    const int ns = (int)st.size();
    c->w_nseg[which] = ns;
    int rc;
    if ((rc = c->d_wseg_traj[which].ensure(std::max(ns, 1))) ||
        (rc = c->d_wseg_len[which].ensure(std::max(ns, 1))) ||
        (rc = c->d_wseg_t0[which].ensure(std::max(ns, 1))) ||
        (rc = c->d_wseg_traj0[which].ensure(c->K + 1)))
        return rc;
    BHMM_HIP(hipMemcpy(c->d_wseg_traj[which].p, st.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(c->d_wseg_len[which].p, sl.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(c->d_wseg_t0[which].p, stt.data(), ns * sizeof(int64_t), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(c->d_wseg_traj0[which].p, s0.data(), (c->K + 1) * sizeof(int32_t),
                       hipMemcpyHostToDevice));
    c->w_ntiles[which] = 0;
    if (wide_tile(c)) {
        // tiles of 16 segments of about the same number of steps (a tile runs as long as its longest row)
        std::vector<int32_t> ts;
        plan::plan_tiles(sp, c->offsets, false, ts);
        c->w_ntiles[which] = (int)(ts.size() / 16);
        if ((rc = c->d_tile_seg[which].ensure(std::max<size_t>(ts.size(), 16))))
            return rc;
        BHMM_HIP(hipMemcpy(c->d_tile_seg[which].p, ts.data(), ts.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        plan::plan_tiles(sp, c->offsets, true, ts);
        c->w_ntilesb[which] = (int)(ts.size() / 16);
        if ((rc = c->d_tile_segb[which].ensure(std::max<size_t>(ts.size(), 16))))
            return rc;
        BHMM_HIP(hipMemcpy(c->d_tile_segb[which].p, ts.data(), ts.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return BHMM_OK;
}

// plan 1 with segments of seglen steps and, for 64 states, the forward pass's own plan 2 with half
// of that (its kernel fits two wavefronts per SIMD; measured on configs[3]: 4.6 instead of 5.1 ms)
// as long as its segments stay four warm-ups long
static int wide_plan_segments(bhmm_ctx *c, int64_t seglen)
{
    int rc = wide_plan(c, 1, seglen);
    if (rc)
        return rc;
    c->w_nseg[2] = 0;
    // (every segment of plan 1 cut in two: the segment count stays a multiple of the SIMD count)
    if (c->N == 64 && !wide_tile(c) && c->wseg_split && seglen / 2 >= 4 * (int64_t)c->spec_W && seglen >= 128) {
        if ((rc = wide_plan(c, 2, seglen, 2)))
            return rc;
        // for every segment of plan 1: the start of a plan-2 segment strictly inside it (-1: none)
        std::vector<int64_t> mid;
        plan::plan_forward_mids(c->offsets, c->K, seglen, mid);
        if ((int)mid.size() != c->w_nseg[1])
            return BHMM_ERR_INVALID; // (cannot happen: both loops cut the trajectories the same way)
        if ((rc = c->d_wseg_fmid.ensure(std::max<size_t>(mid.size(), 1))))
            return rc;
        BHMM_HIP(hipMemcpy(c->d_wseg_fmid.p, mid.data(), mid.size() * sizeof(int64_t), hipMemcpyHostToDevice));
    }
    return rc;
}

int wide_plan_pub(bhmm_ctx *c, int which, int64_t seglen) { return wide_plan(c, which, seglen); }
Segs wide_segs_pub(bhmm_ctx *c, int which) { return segs_of(c, which); }

int wide_model_pub(bhmm_ctx *c, int kind, const double *A, const double *pi, const double *par0,
                   const double *par1, WideModel &m)
{
    return wide_model(c, kind, A, pi, par0, par1, m);
}

int wide_alloc(bhmm_ctx *c)
{
    const int n = c->n;
    c->N = wide_np(n);
    const int S = n * n + 3 * n;
    int rc;
    if ((rc = wide_plan(c, 0, 0)))
        return rc;
    // time-segmented plan: ~4 lane groups per SIMD, but segments long against the warm-up
    c->w_nseg[1] = 0;
    {
        int64_t seglen = c->wseg_len;
        if (seglen <= 0)
            seglen = std::max<int64_t>(wide_fill_len(c), (wide_tile(c) ? 2 : 8) * (int64_t)c->spec_W);
        int64_t maxT = 0;
        for (int k = 0; k < c->K; ++k)
            maxT = std::max(maxT, c->offsets[k + 1] - c->offsets[k]);
        c->wseg_cur_len = seglen;
        if (c->wseg_enabled && maxT > seglen && (rc = wide_plan_segments(c, seglen)))
            return rc;
    }
    // (buffers sized for the finest plan there can be: plan 2 has at most twice the segments of plan 1)
    const int nsmax = std::max(std::max(c->w_nseg[0], 2 * c->w_nseg[1] + c->K), 1);
    if ((rc = c->d_alpha_rm.ensure((size_t)c->total * n)) ||
        (rc = c->d_logLk.ensure(std::max(c->K, 1))) || (rc = c->d_wlogLseg.ensure(nsmax)) ||
        (rc = c->d_gamma0.ensure((size_t)std::max(c->K, 1) * n)) ||
        (rc = c->d_partials.ensure((size_t)nsmax * S)) ||
        (rc = c->d_waentry.ensure((size_t)nsmax * n)) || (rc = c->d_waexit.ensure((size_t)nsmax * n)) ||
        (rc = c->d_wbexit.ensure((size_t)nsmax * n)) || (rc = c->d_wbentry.ensure((size_t)nsmax * n)) ||
        (rc = c->d_specres.ensure(4)) ||
        (rc = c->d_stats.ensure(1 + n + n * n + n + std::max(2 * n, n * c->M))))
        return rc;
    if (c->kind == EMIT_DISC && (rc = c->d_dpartials.ensure((size_t)std::max(nsmax, 4 * (nsmax / 16 + 2)) * n * c->M)))
        return rc;
    if (wide_tile(c) && ((rc = c->d_wexp.ensure((size_t)std::max<int64_t>(c->total, 1))) ||
                         (rc = c->d_wePseg.ensure(nsmax))))
        return rc;
    if (!c->h_specres)
        BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_specres), 4 * sizeof(unsigned int),
                               hipHostMallocDefault));
    BHMM_HIP(hipMemsetAsync(c->d_gamma0.p, 0, (size_t)std::max(c->K, 1) * n * sizeof(double),
                            c->stream));
    return BHMM_OK;
}

// Measure the forgetting curve (k_wide_probe) and set the warm-up length of the time segments
// from it: W = first length after which two chains started differently agree to 1e-13 at every
// sampled position, + 15 %.  W_out = 0: no statement (trajectories too short, curve not below
// the target within Wmax).
template <int NP, int KIND>
static int wide_probe_run(bhmm_ctx *c, const WideModel &m, int *W_out)
{
    *W_out = 0;
    int64_t maxT = 0;
    for (int k = 0; k < c->K; ++k)
        maxT = std::max(maxT, c->offsets[k + 1] - c->offsets[k]);
    const int Wmax = (int)std::min<int64_t>(8192, maxT / 2) / 8 * 8;
    if (Wmax < 64)
        return BHMM_OK;
    std::vector<int> longk;
    for (int k = 0; k < c->K; ++k)
        if (c->offsets[k + 1] - c->offsets[k] >= Wmax)
            longk.push_back(k);
    const int S = 256;
    std::vector<int64_t> starts(S);
    for (int i = 0; i < S; ++i) { // positions relative to the first observation of the context
        const int k = longk[i % longk.size()];
        const int64_t room = c->offsets[k + 1] - c->offsets[k] - Wmax + 1;
        const int64_t rep = i / (int64_t)longk.size(), reps = (S + longk.size() - 1) / longk.size();
        starts[i] = c->offsets[k] + (room - 1) * rep / std::max<int64_t>(reps - 1, 1);
    }
    const size_t bytes = S * sizeof(int64_t) + 2 * (size_t)Wmax * sizeof(unsigned int);
    int rc;
    if ((rc = c->d_probe.ensure(bytes)))
        return rc;
    int64_t *d_starts = reinterpret_cast<int64_t *>(c->d_probe.p);
    unsigned int *d_curve = reinterpret_cast<unsigned int *>(d_starts + S);
    BHMM_HIP(hipMemcpyAsync(d_starts, starts.data(), S * sizeof(int64_t), hipMemcpyHostToDevice,
                            c->stream));
    BHMM_HIP(hipMemsetAsync(d_curve, 0, 2 * (size_t)Wmax * sizeof(unsigned int), c->stream));
    constexpr int GP = 64 / NP;
    hipLaunchKernelGGL((k_wide_probe<NP, KIND>), dim3((2 * S + GP - 1) / GP), dim3(64), 0, c->stream, m,
                       (const void *)c->d_obs_rm.p, (const int64_t *)d_starts, S, Wmax, d_curve);
    BHMM_HIP(hipGetLastError());
    std::vector<float> curve(2 * (size_t)Wmax);
    BHMM_HIP(hipMemcpyAsync(curve.data(), d_curve, curve.size() * sizeof(float), hipMemcpyDeviceToHost,
                            c->stream));
    BHMM_HIP(hipStreamSynchronize(c->stream)); // starts / curve are temporaries
    int last = -1;
    for (int w = 0; w < Wmax; ++w)
        if (std::max(curve[w], curve[Wmax + w]) >= 1e-13f)
            last = w;
    if (last + 2 >= Wmax)
        return BHMM_OK; // not forgotten within Wmax: no statement
    // the boundary check looks at every segment boundary, the probe at S positions: the worst
    // boundary lags the worst sample (measured: 1.3x in warm-up steps); an E-step that fails the
    // check costs eight good ones, a longer warm-up a few per cent
    int W = (int)std::ceil(1.5 * (last + 2));
    *W_out = std::max(16, (W + 7) / 8 * 8);
    return BHMM_OK;
}

// First pass over these observations with this family: measure how fast the model forgets
// (wide_probe_run) and plan the segments for that warm-up length (at least four warm-ups long, or
// what fills the chip).
static int wide_calibrate(bhmm_ctx *c, const WideModel &m)
{
    int rc = BHMM_OK;
    c->spec_calibrated = true;
    int W = 0;
    switch (c->kind) {
    case EMIT_GAUSS:
        rc = WIDE_DISPATCH(c, EMIT_GAUSS, wide_probe_run, c, m, &W);
        break;
    case EMIT_DISC:
        rc = WIDE_DISPATCH(c, EMIT_DISC, wide_probe_run, c, m, &W);
        break;
    default:
        rc = WIDE_DISPATCH(c, EMIT_EXPL, wide_probe_run, c, m, &W);
    }
    if (rc)
        return rc;
    const int W_planned = c->spec_W;
    if (W > 0)
        c->spec_W = W;
    if (W > W_planned) { // longer warm-ups than the plan assumed: longer segments
        int64_t maxT = 0;
        for (int k = 0; k < c->K; ++k)
            maxT = std::max(maxT, c->offsets[k + 1] - c->offsets[k]);
        int64_t seglen = c->wseg_len;
        if (seglen <= 0)
            seglen = std::max<int64_t>(wide_fill_len(c), (wide_tile(c) ? 2 : 4) * (int64_t)W);
        seglen = std::max(seglen, c->wseg_cur_len); // never more segments than allocated for
        if (seglen >= maxT) {
            c->wseg_given_up = true;
        } else if (seglen > c->wseg_cur_len) {
            c->wseg_cur_len = seglen;
            if ((rc = wide_plan_segments(c, seglen)))
                return rc;
            if (c->w_nseg[1] <= c->w_nseg[0])
                c->wseg_given_up = true;
        }
    }
    return BHMM_OK;
}

int wide_forward(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                 const double *par1)
{
    WideModel m;
    int rc = wide_model(c, c->kind, A, pi, par0, par1, m);
    if (rc)
        return rc;
    switch (c->kind) { // plan 0: one segment per trajectory, the exact serial recursion
    case EMIT_GAUSS:
        return WIDE_DISPATCH(c, EMIT_GAUSS, wide_launch_fwd, c, m, 0);
    case EMIT_DISC:
        return WIDE_DISPATCH(c, EMIT_DISC, wide_launch_fwd, c, m, 0);
    default:
        return WIDE_DISPATCH(c, EMIT_EXPL, wide_launch_fwd, c, m, 0);
    }
}

// alpha rows for the backward draw (path_api.hip): the time-segmented forward pass of the E-step
// (lazily scaled / matrix-core kernels where the context uses them -- the draw normalises
// alpha_t[i] A[i][s_{t+1}] itself, any positive factor per row cancels), boundaries verified to 1e-11
// like the E-step's; anything else -- the exact serial recursion of wide_forward.
int wide_forward_draw(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                      const double *par1)
{
    WideModel m;
    int rc = wide_model(c, c->kind, A, pi, par0, par1, m);
    if (rc)
        return rc;
    if (c->spec_enabled && c->wseg_enabled) {
        if (!c->spec_calibrated && c->w_nseg[1] > c->w_nseg[0] && (rc = wide_calibrate(c, m)))
            return rc;
        if (!c->wseg_given_up && c->w_nseg[1] > c->w_nseg[0]) {
            const bool lazy = !c->careful && !c->wide_careful;
            BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 3 * sizeof(unsigned int), c->stream));
            switch (c->kind) {
            case EMIT_GAUSS:
                rc = WIDE_DISPATCH(c, EMIT_GAUSS, wide_launch_fwd, c, m, 1, lazy);
                break;
            case EMIT_DISC:
                rc = WIDE_DISPATCH(c, EMIT_DISC, wide_launch_fwd, c, m, 1, lazy);
                break;
            default:
                rc = WIDE_DISPATCH(c, EMIT_EXPL, wide_launch_fwd, c, m, 1, lazy);
            }
            if (rc)
                return rc;
            const Segs sgf = segs_of(c, wide_fwd_plan(c, 1));
            hipLaunchKernelGGL(k_wide_check, dim3((sgf.nseg + 15) / 16), dim3(256), 0, c->stream, sgf,
                               c->n, (const double *)c->d_waentry.p, (const double *)c->d_waexit.p,
                               (const double *)nullptr, (const double *)nullptr, 1e-11, c->d_specres.p);
            BHMM_HIP(hipGetLastError());
            BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 3 * sizeof(unsigned int),
                                    hipMemcpyDeviceToHost, c->stream));
            BHMM_HIP(hipStreamSynchronize(c->stream));
            c->draw_fwd_segmented = c->h_specres[0] == 0 && (!lazy || c->h_specres[2] == 0);
            if (c->draw_fwd_segmented) {
                float dev; // (largest boundary deviation the check saw: what the draws' watch is sized by)
                memcpy(&dev, &c->h_specres[1], sizeof(float));
                c->draw_alpha_dev = dev;
                return BHMM_OK;
            }
        }
    }
    c->draw_fwd_segmented = false;
    c->draw_alpha_dev = 0.0;
    return wide_forward(c, A, pi, par0, par1);
}

int wide_estep(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
               const double *par1, double *stats_dev, int flags)
{
    WideModel m;
    int rc = wide_model(c, c->kind, A, pi, par0, par1, m);
    if (rc)
        return rc;
    const bool sg = (flags & BHMM_FLAG_STORE_GAMMA) != 0;
    if (sg && (rc = c->d_gamma_ci.ensure((size_t)c->total * c->n)))
        return rc;
    auto run = [&](int which, bool lazy) -> int {
        int r;
        c->tile_used = lazy && wide_tile(c);
        // event intervals of this family: [0] forward pass, [2] backward pass + statistics
        BHMM_HIP(hipEventRecord(c->ev[0], c->stream));
#define BHMM_WIDE_PASSES(KINDV)                                                                  \
    do {                                                                                         \
        if ((r = WIDE_DISPATCH(c, KINDV, wide_launch_fwd, c, m, which, lazy)))                   \
            return r;                                                                            \
        BHMM_HIP(hipEventRecord(c->ev[1], c->stream));                                           \
        BHMM_HIP(hipEventRecord(c->ev[2], c->stream));                                           \
        r = WIDE_DISPATCH(c, KINDV, wide_launch_bwd, c, m, which, sg, stats_dev, lazy);          \
    } while (0)
        switch (c->kind) {
        case EMIT_GAUSS:
            BHMM_WIDE_PASSES(EMIT_GAUSS);
            break;
        case EMIT_DISC:
            BHMM_WIDE_PASSES(EMIT_DISC);
            break;
        default:
            BHMM_WIDE_PASSES(EMIT_EXPL);
        }
#undef BHMM_WIDE_PASSES
        if (r)
            return r;
        BHMM_HIP(hipEventRecord(c->ev[3], c->stream));
        BHMM_HIP(hipEventRecord(c->ev[4], c->stream));
        return BHMM_OK;
    };
    if (c->wseg_enabled && !c->spec_calibrated && c->w_nseg[1] > c->w_nseg[0] && (rc = wide_calibrate(c, m)))
        return rc;
    if (c->wseg_enabled && !c->wseg_given_up && c->w_nseg[1] > c->w_nseg[0]) {
        // time-segmented run with warm-up boundaries, verified afterwards
        // lazily scaled kernels unless an earlier E-step on these data left their range
        const bool lazy = !c->careful && !c->wide_careful;
        BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 3 * sizeof(unsigned int), c->stream));
        if ((rc = run(1, lazy)))
            return rc;
        const Segs sgs = segs_of(c, 1), sgf = segs_of(c, wide_fwd_plan(c, 1));
        hipLaunchKernelGGL(k_wide_check, dim3((sgf.nseg + 15) / 16), dim3(256), 0, c->stream, sgf,
                           c->n, (const double *)c->d_waentry.p, (const double *)c->d_waexit.p,
                           (const double *)nullptr, (const double *)nullptr, 1e-11, c->d_specres.p);
        hipLaunchKernelGGL(k_wide_check, dim3((sgs.nseg + 15) / 16), dim3(256), 0, c->stream, sgs,
                           c->n, (const double *)nullptr, (const double *)nullptr,
                           (const double *)c->d_wbexit.p, (const double *)c->d_wbentry.p, 1e-11,
                           c->d_specres.p);
        BHMM_HIP(hipGetLastError());
        BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 3 * sizeof(unsigned int),
                                hipMemcpyDeviceToHost, c->stream));
        // the statistics (and, for moderately many trajectories, the log-likelihoods) travel with the verdict
        // words: ONE host round trip per verified E-step -- bhmm_estep_fetch finds them in the pinned buffer
        const int S = bhmm_ctx_stats_size(c);
        const bool pre = stats_dev && c->h_pinned && c->h_pinned_n >= (size_t)S + (size_t)c->K;
        if (pre) {
            BHMM_HIP(hipMemcpyAsync(c->h_pinned, stats_dev, (size_t)S * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            if (c->K <= 4096)
                BHMM_HIP(hipMemcpyAsync(c->h_pinned + S, c->d_logLk.p, (size_t)c->K * sizeof(double),
                                        hipMemcpyDeviceToHost, c->stream));
        }
        BHMM_HIP(hipStreamSynchronize(c->stream));
        if (lazy)
            c->wide_trouble = c->h_specres[2];
        if (lazy && c->h_specres[2] != 0) {
            // a vector left the range the lazy scaling covers: per-step normalisation from now on
            c->wide_careful = true;
            c->careful_retry = true;
            return wide_estep(c, A, pi, par0, par1, stats_dev, flags);
        }
        float dev;
        memcpy(&dev, &c->h_specres[1], sizeof(float));
        c->spec_last_dev = dev;
        if (c->h_specres[0] == 0) {
            c->spec_ok++;
            c->ev_pending = true;
            // The probe's warm-up carries a margin for the boundaries it did not sample; what the check
            // found at EVERY boundary says how much of it this model needs.  First E-step on these
            // observations only (so that repeated calls stay bit-identical): more than two decades
            // inside the tolerance -> 10 % shorter and once more, at most four times, aiming 30 times
            // inside the 1e-11 of the check.
            if (wide_tile(c) && lazy && !c->spec_W_fixed && c->tile_settle < 4 && dev > 0.f && dev < 1e-13f) {
                const double f = std::max(log(3e-13) / log((double)dev), 0.9);
                const int Wn = ((int)ceil(c->spec_W * f) + 7) / 8 * 8;
                if (Wn < c->spec_W) {
                    ++c->tile_settle;
                    c->tile_W_good = c->spec_W;
                    c->spec_W = Wn;
                    return wide_estep(c, A, pi, par0, par1, stats_dev, flags);
                }
            }
            c->tile_settle = 4;
            // An EM sequence moves the model, and with it the length the filter needs to forget its start
            // (configs[3]: the worst boundary went 5e-14 -> 1.3e-11 over 14 iterations at a fixed warm-up,
            // profiles/r05).  A check that fails costs this call a second pass, so the warm-up FOLLOWS the
            // measured deviation whenever the model has changed since the previous call: it aims at 3e-13
            // (1.5 decades inside the tolerance), one decade being spec_W / 12.5 steps.  Calls that repeat a
            // model never change it (bit-identical results, as before).
            if (wide_tile(c) && lazy && !c->spec_W_fixed && c->carry_delta > 0.0 && dev > 0.f) {
                const double dec = log10((double)dev / 3e-13);
                int Wn = c->spec_W;
                if (dev > 2e-12f)
                    Wn = ((int)ceil(c->spec_W * (1.0 + dec / 12.5)) + 7) / 8 * 8;
                else if (dev < 2e-14f)
                    Wn = std::max(16, ((int)ceil(c->spec_W * (1.0 + 0.5 * dec / 12.5)) + 7) / 8 * 8);
                if ((int64_t)Wn <= c->wseg_cur_len) // (a warm-up may be as long as a segment; beyond: re-plan below)
                    c->spec_W = Wn;
            }
            if (pre) {
                c->prefetched = true;
                c->logLk_prefetched = c->K <= 4096;
            }
            return BHMM_OK;
        }
        if (c->tile_W_good > c->spec_W && c->tile_settle > 0 && c->tile_settle <= 4) {
            // a refinement too far: back to the warm-up that verified, for good
            c->spec_W = c->tile_W_good;
            c->tile_W_good = 0;
            c->tile_settle = 5;
            return wide_estep(c, A, pi, par0, par1, stats_dev, flags);
        }
        c->spec_fail++;
        if (wide_tile(c) && lazy && !c->spec_W_fixed && c->wide_retry < 2) {
            // the tile kernels once more with the warm-up the measured deviation asks for (it decays
            // geometrically with the warm-up length), as long as it is no longer than a segment: 10 ms
            // instead of the 300 ms of the serial plan below
            const double d = std::min(std::max((double)dev, 1e-300), 0.5);
            const double f = std::min(std::max(log(3e-13) / log(d), 1.08), 2.0);
            const int Wn = ((int)ceil(c->spec_W * f) + 7) / 8 * 8;
            if ((int64_t)Wn <= c->wseg_cur_len) {
                ++c->wide_retry;
                c->spec_W = Wn;
                return wide_estep(c, A, pi, par0, par1, stats_dev, flags);
            }
        }
        // The deviation decays geometrically with the warm-up length (the filter forgets its start
        // vector): extrapolate to where it reaches a tenth of the tolerance, lengthen the
        // segments to at least four warm-ups and try again at the next call; give up (serial
        // plan) after three re-plans or when the segments would no longer split a trajectory.
        int64_t maxT = 0;
        for (int k = 0; k < c->K; ++k)
            maxT = std::max(maxT, c->offsets[k + 1] - c->offsets[k]);
        const double d = std::min(std::max((double)dev, 1e-300), 0.5);
        double f = log(1e-12) / log(d);
        f = std::min(std::max(f, 1.25), 8.0);
        const int64_t Wn = ((int64_t)ceil(c->spec_W * f) + 7) / 8 * 8;
        int64_t seglen = c->wseg_len > 0 ? (int64_t)c->wseg_len : (wide_tile(c) ? 2 : 4) * Wn;
        if (c->wseg_len <= 0)
            seglen = std::max(seglen, wide_fill_len(c));
        seglen = std::max(seglen, c->wseg_cur_len); // never more segments than allocated for
        if (c->wide_replans >= 3 || seglen >= maxT || Wn >= maxT / 2) {
            c->wseg_given_up = true;
        } else {
            ++c->wide_replans;
            c->spec_W = (int)Wn;
            if (seglen > c->wseg_cur_len) {
                c->wseg_cur_len = seglen;
                if ((rc = wide_plan_segments(c, seglen)))
                    return rc;
            }
            if (c->w_nseg[1] <= c->w_nseg[0])
                c->wseg_given_up = true;
        }
    }
    if ((rc = run(0, false)))
        return rc;
    c->ev_pending = true;
    return BHMM_OK;
}

// beta (row-major) for explicit pobs into ctx->d_alpha_rm (reused as the output buffer)
int wide_backward(bhmm_ctx *c, const double *A)
{
    WideModel m;
    int rc = wide_model(c, EMIT_EXPL, A, nullptr, nullptr, nullptr, m);
    if (rc)
        return rc;
    const int NP = c->N, GP = 64 / NP;
    const size_t sm = (size_t)(NP * wide_pitch(NP) + GP * NP) * sizeof(double);
    const dim3 grid((c->K + GP - 1) / GP), blk(64);
    const double *pobs = reinterpret_cast<const double *>(c->d_obs_rm.p);
    if (NP == 16)
        hipLaunchKernelGGL((k_wide_beta<16>), grid, blk, sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, c->K, pobs, c->d_alpha_rm.p);
    else if (NP == 32)
        hipLaunchKernelGGL((k_wide_beta<32>), grid, blk, sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, c->K, pobs, c->d_alpha_rm.p);
    else
        hipLaunchKernelGGL((k_wide_beta<64>), grid, blk, sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, c->K, pobs, c->d_alpha_rm.p);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

// xi-counts from host alpha / beta / pobs (hidden API), 9..64 states
int wide_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                           const double *beta, int n, int64_t T)
{
    const int NP = wide_np(n), GP = 64 / NP;
    const int nslab = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (T - 1 + 63) / 64));
    double *dA, *dp, *da, *db, *dpart, *dC;
    const size_t cnt = (size_t)T * n;
    std::vector<void *> ptrs;
    auto alloc = [&](double **p, size_t count) {
        hipError_t e = hipMalloc(reinterpret_cast<void **>(p), std::max<size_t>(count, 1) * sizeof(double));
        if (e == hipSuccess)
            ptrs.push_back(*p);
        return e;
    };
    auto cleanup = [&]() {
        for (void *p : ptrs)
            (void)hipFree(p);
    };
    if (alloc(&dA, (size_t)n * n) || alloc(&dp, cnt) || alloc(&da, cnt) || alloc(&db, cnt) ||
        alloc(&dpart, (size_t)nslab * n * n) || alloc(&dC, (size_t)n * n)) {
        cleanup();
        (void)hipGetLastError();
        set_error("hipMalloc failed");
        return BHMM_ERR_NO_MEM;
    }
    hipError_t e = hipMemcpy(dA, A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dp, pobs, cnt * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(da, alpha, cnt * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, beta, cnt * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(dpart, 0, (size_t)nslab * n * n * sizeof(double));
    if (e == hipSuccess) {
        const size_t sm = (size_t)(NP * wide_pitch(NP) + GP * NP) * sizeof(double);
        const dim3 grid((nslab + GP - 1) / GP), blk(64);
        if (NP == 16)
            hipLaunchKernelGGL((k_wide_xi<16>), grid, blk, sm, 0, (const double *)dA, (const double *)dp,
                               (const double *)da, (const double *)db, n, T, nslab, dpart);
        else if (NP == 32)
            hipLaunchKernelGGL((k_wide_xi<32>), grid, blk, sm, 0, (const double *)dA, (const double *)dp,
                               (const double *)da, (const double *)db, n, T, nslab, dpart);
        else
            hipLaunchKernelGGL((k_wide_xi<64>), grid, blk, sm, 0, (const double *)dA, (const double *)dp,
                               (const double *)da, (const double *)db, n, T, nslab, dpart);
        hipLaunchKernelGGL(k_wide_xi_sum, dim3(n * n), dim3(64), 0, 0, (const double *)dA,
                           (const double *)dpart, n, nslab, dC);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(C, dC, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess)
        return hip_fail(e, "wide_transition_counts");
    return BHMM_OK;
}

} // namespace bhmm
