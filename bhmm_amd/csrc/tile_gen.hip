// tile_gen.hip -- more than 64 (up to 128) hidden states on the row-batched matrix-core kernels
// (tile_kernels.hpp): the E-step of the any-N family (gen_api.hip) without its order-faithful,
// one-workgroup-per-trajectory recursions -- those stay for Viterbi / path sampling (bit-exact) and as
// the fallback of this path.  Reference: bhmm/hidden/impl_c/_hidden.c:42-63,91-109,148-183.
//
// Time segments with verified warm-up boundaries as for 64 states (wide_api.hip); the warm-up length
// is calibrated with forward passes alone (each leaves, at every segment boundary, the distance between
// the warmed-up vector and the one the neighbouring segment computed -- the curve k_wide_probe measures
// for up to 64 states); the xi counts are the time-parallel GEMM of gen_kernels.hpp over the rows
// W_{t-1} = p_t o beta_t / S the backward pass stores.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "host_common.hpp"
#include "plan.hpp"
#include "gen_kernels.hpp"
#include "tile_kernels.hpp"
#include "big_kernels.hpp"
#include "tile_gen_launch.hpp"

namespace bhmm {
int wide_plan_pub(bhmm_ctx *c, int which, int64_t seglen);
Segs wide_segs_pub(bhmm_ctx *c, int which);
int big_launch_fwd(bhmm_ctx *c, const WideModel &m);
int big_launch_bwd(bhmm_ctx *c, const WideModel &m, double *gam, double *stats_dev);

// 65 .. 128 states: tile_kernels.hpp (A's blocks in registers); 129 .. 512: big_kernels.hpp (streamed from L2)
bool tile_gen_capable(const bhmm_ctx *c) { return c->tile_latched && c->gen && c->n <= 512; }

TILE_GEN_LAUNCH_DECL(extern, 5)
TILE_GEN_LAUNCH_DECL(extern, 6)
TILE_GEN_LAUNCH_DECL(extern, 7)
TILE_GEN_LAUNCH_DECL(extern, 8)

namespace {

// above this many states the kernels of big_kernels.hpp run the E-step (BHMM_AMD_BIG_FROM: experiments)
static int big_from()
{
    static const int v = getenv("BHMM_AMD_BIG_FROM") ? atoi(getenv("BHMM_AMD_BIG_FROM")) : 128;
    return v;
}

int64_t max_len(const bhmm_ctx *c)
{
    int64_t maxT = 0;
    for (int k = 0; k < c->K; ++k)
        maxT = std::max(maxT, c->offsets[k + 1] - c->offsets[k]);
    return maxT;
}

// segments that fill the chip (16 per workgroup, one workgroup per compute unit), but at least two
// warm-ups long
int64_t fill_len(const bhmm_ctx *c)
{
    const int64_t want = 16 * (int64_t)(c->num_simd / 4) * c->tile_per_cu;
    return std::max<int64_t>(((c->total + want - 1) / want + 3) & ~(int64_t)3, 16);
}

int plan_for(bhmm_ctx *c, int W)
{
    // (more than 128 states: a step is tens of microseconds of matrix instructions and there are seldom enough
    // segments for every compute unit -- filling the chip is worth more than short warm-ups relative to the
    // segments: half a warm-up is long enough there)
    const int64_t wmin = c->n > big_from() ? std::max<int64_t>(W / 2, 32) : 2 * (int64_t)W;
    int64_t seglen = c->wseg_len > 0 ? (int64_t)c->wseg_len : std::max<int64_t>(fill_len(c), wmin);
    seglen = std::max(seglen, c->wseg_cur_len); // never more segments than allocated for
    if (seglen >= max_len(c))
        seglen = 0; // one segment per trajectory: no boundaries to verify
    if (c->w_nseg[1] > 0 && seglen == c->wseg_cur_len)
        return BHMM_OK;
    c->wseg_cur_len = seglen;
    return wide_plan_pub(c, 1, seglen);
}

// (the dispatch macros below paste the function name)
#define launch_fwd tile_gen_launch_fwd
#define launch_bwd tile_gen_launch_bwd

#define TILE_GEN_NT(fn, KINDV, ...)                                                                      \
    (c->n <= 80 ? fn<5, KINDV>(__VA_ARGS__) : c->n <= 96 ? fn<6, KINDV>(__VA_ARGS__)                      \
     : c->n <= 112 ? fn<7, KINDV>(__VA_ARGS__) : fn<8, KINDV>(__VA_ARGS__))
#define TILE_GEN_DISPATCH_128(fn, ...)                                                                   \
    (c->kind == EMIT_GAUSS  ? TILE_GEN_NT(fn, EMIT_GAUSS, __VA_ARGS__)                                   \
     : c->kind == EMIT_DISC ? TILE_GEN_NT(fn, EMIT_DISC, __VA_ARGS__)                                    \
                            : TILE_GEN_NT(fn, EMIT_EXPL, __VA_ARGS__))
// (more than 128 states: big_api.hip)
#define TILE_GEN_DISPATCH(fn, ...) (c->n > big_from() ? big_##fn(__VA_ARGS__) : TILE_GEN_DISPATCH_128(fn, __VA_ARGS__))

// boundary check of one direction (0 forward, 1 backward): flags -> host
int run_check(bhmm_ctx *c, int dir)
{
    const Segs sg = wide_segs_pub(c, 1);
    if (dir == 0)
        hipLaunchKernelGGL(k_wide_check, dim3((sg.nseg + 15) / 16), dim3(256), 0, c->stream, sg, c->n,
                           (const double *)c->d_waentry.p, (const double *)c->d_waexit.p,
                           (const double *)nullptr, (const double *)nullptr, 1e-11, c->d_specres.p);
    else
        hipLaunchKernelGGL(k_wide_check, dim3((sg.nseg + 15) / 16), dim3(256), 0, c->stream, sg, c->n,
                           (const double *)nullptr, (const double *)nullptr, (const double *)c->d_wbexit.p,
                           (const double *)c->d_wbentry.p, 1e-11, c->d_specres.p);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

int read_flags(bhmm_ctx *c)
{
    BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 3 * sizeof(unsigned int), hipMemcpyDeviceToHost,
                            c->stream));
    BHMM_HIP(hipStreamSynchronize(c->stream));
    return BHMM_OK;
}

// The warm-up length from forward passes: deviation at the segment boundaries after W steps, twice,
// and the geometric decay between the two (the filter forgets its start vector) extrapolated to 1e-13.
int calibrate(bhmm_ctx *c, const WideModel &m, bool *usable)
{
    *usable = false;
    const int64_t maxT = max_len(c);
    int W = std::max(16, (c->spec_W_fixed ? c->spec_W : 32) / 8 * 8);
    double prevW = 0.0, prevdev = 1.0;
    for (int it = 0; it < 8; ++it) {
        c->spec_W = W;
        int rc = plan_for(c, W);
        if (rc)
            return rc;
        if (c->w_nseg[1] <= c->w_nseg[0]) { // no time segmentation (short trajectories): nothing to verify
            *usable = true;
            return BHMM_OK;
        }
        BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 3 * sizeof(unsigned int), c->stream));
        if ((rc = TILE_GEN_DISPATCH(launch_fwd, c, m)) || (rc = run_check(c, 0)) || (rc = read_flags(c)))
            return rc;
        if (c->h_specres[2]) { // left the range of the lazily scaled kernels: the order-faithful family
            c->wide_trouble = c->h_specres[2];
            c->tile_reason = 1;
            return BHMM_OK;
        }
        float devf;
        memcpy(&devf, &c->h_specres[1], sizeof(float));
        const double dev = std::max((double)devf, 1e-300);
        c->spec_last_dev = devf;
        if (c->h_specres[0] == 0 && dev <= 3e-13) {
            *usable = true;
            return BHMM_OK;
        }
        if (c->spec_W_fixed) {
            c->tile_reason = 6;
            return BHMM_OK; // the caller's warm-up does not verify: not ours to change
        }
        double Wn;
        if (prevW > 0.0 && dev < 0.5 * prevdev) {
            const double rate = log(prevdev / dev) / ((double)W - prevW); // per step
            Wn = W + 1.25 * log(dev / 1e-13) / rate;
        } else {
            Wn = 2.0 * W;
        }
        prevW = W;
        prevdev = dev;
        W = (int)std::min<double>(std::max(Wn, W + 8.0), 4.0 * W + 64.0);
        W = (W + 7) / 8 * 8;
        if (W >= maxT / 2) {
            c->tile_reason = 3;
            return BHMM_OK; // chains that do not forget within the trajectories: serial family
        }
    }
    c->tile_reason = 2;
    return BHMM_OK;
}

} // namespace

int tile_gen_alloc(bhmm_ctx *c)
{
    const int n = c->n;
    c->N = 64; // (not used by this family; wide_fill_len never sees these contexts)
    int rc;
    if ((rc = wide_plan_pub(c, 0, 0)))
        return rc;
    c->w_nseg[1] = 0;
    c->wseg_cur_len = 0;
    c->spec_calibrated = c->spec_W_fixed && false;
    c->wseg_given_up = false;
    c->wide_careful = false;
    c->tile_reason = 0;
    // buffers for the finest plan there can be
    const int64_t minlen = c->wseg_len > 0 ? (int64_t)c->wseg_len : fill_len(c);
    int64_t nsmax = c->K;
    for (int k = 0; k < c->K; ++k)
        nsmax += (c->offsets[k + 1] - c->offsets[k]) / std::max<int64_t>(minlen & ~(int64_t)3, 4) + 1;
    const size_t S = (size_t)n * n + 3 * n;
    const size_t ntmax = (size_t)nsmax / 16 + 3;
    if ((rc = c->d_wlogLseg.ensure(nsmax)) || (rc = c->d_wePseg.ensure(nsmax)) ||
        (rc = c->d_waentry.ensure((size_t)nsmax * n)) || (rc = c->d_waexit.ensure((size_t)nsmax * n)) ||
        (rc = c->d_wbexit.ensure((size_t)nsmax * n)) || (rc = c->d_wbentry.ensure((size_t)nsmax * n)) ||
        (rc = c->d_wexp.ensure((size_t)std::max<int64_t>(c->total, 1))) || (rc = c->d_specres.ensure(4)) ||
        (rc = c->d_partials.ensure(ntmax * S)))
        return rc;
    if (c->kind == EMIT_DISC && (rc = c->d_dpartials.ensure(4 * ntmax * (size_t)n * c->M)))
        return rc;
    if (!c->h_specres)
        BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_specres), 4 * sizeof(unsigned int),
                               hipHostMallocDefault));
    BHMM_HIP(hipMemsetAsync(c->d_gamma0.p, 0, (size_t)std::max(c->K, 1) * n * sizeof(double), c->stream));
    return BHMM_OK;
}

// *done: statistics are in stats_dev, verified; else the caller runs the order-faithful kernels
int tile_gen_estep(bhmm_ctx *c, const WideModel &m, double *stats_dev, int flags, bool *done)
{
    *done = false;
    c->tile_used = false;
    if (!tile_gen_capable(c) || c->wseg_given_up || c->wide_careful || !c->wseg_enabled)
        return BHMM_OK;
    int rc;
    if (!c->spec_calibrated) {
        c->spec_calibrated = true;
        bool usable = false;
        if ((rc = calibrate(c, m, &usable)))
            return rc;
        if (!usable) {
            c->wseg_given_up = true;
            return BHMM_OK;
        }
    }
    const bool sg = (flags & BHMM_FLAG_STORE_GAMMA) != 0;
    if (sg && (rc = c->d_gamma_ci.ensure((size_t)c->total * c->n)))
        return rc;
    double *gam = sg ? c->d_gamma_ci.p : nullptr;
    for (int attempt = 0; attempt < 3; ++attempt) {
        BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 3 * sizeof(unsigned int), c->stream));
        BHMM_HIP(hipEventRecord(c->ev[0], c->stream));
        if ((rc = TILE_GEN_DISPATCH(launch_fwd, c, m)))
            return rc;
        BHMM_HIP(hipEventRecord(c->ev[1], c->stream));
        BHMM_HIP(hipEventRecord(c->ev[2], c->stream));
        if ((rc = TILE_GEN_DISPATCH(launch_bwd, c, m, gam, stats_dev)))
            return rc;
        BHMM_HIP(hipEventRecord(c->ev[3], c->stream));
        BHMM_HIP(hipEventRecord(c->ev[4], c->stream));
        const bool segmented = c->w_nseg[1] > c->w_nseg[0];
        if (segmented && ((rc = run_check(c, 0)) || (rc = run_check(c, 1))))
            return rc;
        if ((rc = read_flags(c)))
            return rc;
        c->wide_trouble = c->h_specres[2];
        if (c->h_specres[2]) {
            // A self-check fired: the kernels are deterministic, so these data leave the lazily scaled
            // kernels' range every time (a run that was merely repeated would hide an uninitialised read --
            // round 4's LDS over-read was found because such a flag did NOT repeat).  Nothing of a flagged
            // run is used; the context moves to the order-faithful family for these observations.
            c->wide_careful = true; // out of the lazily scaled kernels' range on these data: stay away
            c->tile_reason = 4;
            return BHMM_OK;
        }
        float devf;
        memcpy(&devf, &c->h_specres[1], sizeof(float));
        c->spec_last_dev = devf;
        if (!segmented || c->h_specres[0] == 0) {
            c->spec_ok++;
            c->ev_pending = true;
            c->tile_used = true;
            *done = true;
            // (first E-step on these observations: more than two decades inside the tolerance -> 10 %
            // shorter and once more, at most four times; wide_api.hip)
            if (segmented && !c->spec_W_fixed && c->tile_settle < 4 && devf > 0.f && devf < 1e-13f) {
                const double f = std::max(log(3e-13) / log((double)devf), 0.9);
                const int Wn = std::max(16, ((int)ceil(c->spec_W * f) + 7) / 8 * 8);
                if (Wn < c->spec_W) {
                    ++c->tile_settle;
                    c->tile_W_good = c->spec_W;
                    c->spec_W = Wn;
                    *done = false;
                    attempt = -1; // (not an attempt after a failure)
                    continue;
                }
            }
            c->tile_settle = 4;
            return BHMM_OK;
        }
        if (c->tile_W_good > c->spec_W && c->tile_settle > 0 && c->tile_settle <= 4) {
            c->spec_W = c->tile_W_good; // a refinement too far: back to the warm-up that verified, for good
            c->tile_W_good = 0;
            c->tile_settle = 5;
            attempt = -1;
            continue;
        }
        c->spec_fail++;
        // the model has moved to slower forgetting: extrapolate (geometric decay) and try again
        if (c->spec_W_fixed)
            break;
        const double d = std::min(std::max((double)devf, 1e-300), 0.5);
        const double f = std::min(std::max(log(1e-13) / log(d), 1.25), 4.0);
        const int Wn = ((int)ceil(c->spec_W * f) + 7) / 8 * 8;
        if (Wn >= max_len(c) / 2)
            break;
        c->spec_W = Wn;
        if ((rc = plan_for(c, Wn)))
            return rc;
    }
    c->wseg_given_up = true;
    c->tile_reason = 5;
    return BHMM_OK;
}

// alpha rows (row-major, any positive factor per row) for the backward draw of 65..128 states from the
// tile forward pass, boundaries verified to 1e-11; *done = false: the caller runs the serial forward pass
int tile_gen_forward_draw(bhmm_ctx *c, const WideModel &m, bool *done)
{
    *done = false;
    if (!tile_gen_capable(c) || c->wseg_given_up || c->wide_careful || !c->wseg_enabled || !c->spec_enabled)
        return BHMM_OK;
    int rc;
    if (!c->spec_calibrated) {
        c->spec_calibrated = true;
        bool usable = false;
        if ((rc = calibrate(c, m, &usable)))
            return rc;
        if (!usable) {
            c->wseg_given_up = true;
            return BHMM_OK;
        }
    }
    BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 3 * sizeof(unsigned int), c->stream));
    if ((rc = TILE_GEN_DISPATCH(launch_fwd, c, m)))
        return rc;
    const bool segmented = c->w_nseg[1] > c->w_nseg[0];
    if (segmented && (rc = run_check(c, 0)))
        return rc;
    if ((rc = read_flags(c)))
        return rc;
    *done = c->h_specres[2] == 0 && (!segmented || c->h_specres[0] == 0);
    float dev = 0.f; // (largest boundary deviation the check saw: what the draws' watch is sized by)
    if (segmented)
        memcpy(&dev, &c->h_specres[1], sizeof(float));
    c->draw_alpha_dev = *done ? dev : 0.0;
    return BHMM_OK;
}

} // namespace bhmm
