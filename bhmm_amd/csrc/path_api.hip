// path_api.hip -- C ABI entry points backed by the order-faithful kernels (Viterbi, path
// sampling) and the small reference-shaped row-major kernels.  This translation unit is
// compiled with -ffp-contract=off (see Makefile): no fused multiply-adds, so products and
// sums round like the reference's scalar C.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "host_common.hpp"
#include "plan.hpp"
#include "path_kernels.hpp"
#include "draw_verify.hpp"

namespace bhmm {
int invalid_arg(const std::string &msg);
int wide_model_pub(bhmm_ctx *c, int kind, const double *A, const double *pi, const double *par0,
                   const double *par1, WideModel &m);
int wide_forward_draw(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                      const double *par1);
int wide_forward(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                 const double *par1);
int wide_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                           const double *beta, int n, int64_t T);
int forward_ci(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
               const double *par1);
int forward_ci_verdict(bhmm_ctx *c, bool *ok);
// gen_api.hip (more than 64 states)
int gen_viterbi_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                    const double *par1, void *paths_out, int out_fmt);
int gen_sample_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                   const double *par1, const double *u, uint64_t seed, int32_t *paths,
                   int64_t *counts, int64_t *n0, double *emis, double *stats_dev);
int gen_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                          const double *beta, int N, int64_t T);
int gen_sample_path(int32_t *path, const double *alpha, const double *A, const double *u, int N,
                    int64_t T);
int unpack_ws_rows(bhmm_ctx *c, double *dst_dev);
Chunks chunks_pub(const bhmm_ctx *c);

// ---- draws inside the reach of the alpha rows' verified deviation (draw_verify.hpp) ---------------
// d_dv: [count | disagree | unconverged | checked] | DrawEvent[DRAW_EVENT_CAP] | model
static inline size_t dv_model_offset() { return 16 + (size_t)DRAW_EVENT_CAP * sizeof(DrawEvent); }

// watch for one sampling call: tol <= 0 (or the option off) leaves it switched off.  The four counters are
// cleared on the stream.
// count_slot: a counter the caller clears with its own fills (saves a dispatch per Gibbs sweep)
int draw_watch_prepare(bhmm_ctx *c, double tol, DrawWatch &w, unsigned int *count_slot)
{
    w.ev = nullptr;
    w.count = nullptr;
    w.tol = 0.0;
    c->draw_events = c->draw_checked = c->draw_redone = 0;
    if (!c->draw_watch || !(tol > 0.0))
        return BHMM_OK;
    const size_t msz = ((size_t)c->n * c->n + 3 * (size_t)c->n +
                        (c->kind == EMIT_DISC ? (size_t)c->n * c->M : 0)) * sizeof(double);
    int rc = c->d_dv.ensure(dv_model_offset() + msz);
    if (rc)
        return rc;
    if (!count_slot)
        BHMM_HIP(hipMemsetAsync(c->d_dv.p, 0, 16, c->stream));
    w.count = count_slot ? count_slot : reinterpret_cast<unsigned int *>(c->d_dv.p);
    w.ev = reinterpret_cast<DrawEvent *>(c->d_dv.p + 16);
    w.tol = c->draw_watch_tol > 0.0 ? c->draw_watch_tol : tol;
    return BHMM_OK;
}

// `count` draws were recorded: those with gap <= thr are decided again on the windowed serial recursion.
// *ok = every such decision stands (else the caller repeats the call on exact alpha rows).
int draw_verify_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0, const double *par1,
                    unsigned int count, double thr, int64_t Wlong, bool *ok)
{
    *ok = false;
    c->draw_events = count;
    if (count > DRAW_EVENT_CAP)
        return BHMM_OK; // (more than the list holds: the exact rows decide)
    const int n = c->n;
    double *md = reinterpret_cast<double *>(c->d_dv.p + dv_model_offset());
    BHMM_HIP(hipMemcpyAsync(md, A, (size_t)n * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BHMM_HIP(hipMemcpyAsync(md + (size_t)n * n, pi, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    double *mp0 = md + (size_t)n * n + n;
    if (c->kind == EMIT_GAUSS) {
        BHMM_HIP(hipMemcpyAsync(mp0, par0, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        BHMM_HIP(hipMemcpyAsync(mp0 + n, par1, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    } else if (c->kind == EMIT_DISC) {
        BHMM_HIP(hipMemcpyAsync(mp0, par0, (size_t)n * c->M * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    unsigned int *res = reinterpret_cast<unsigned int *>(c->d_dv.p);
    BHMM_HIP(hipMemsetAsync(res + 1, 0, 12, c->stream)); // (the three result words; rare path)
    const size_t sm = (4 * (size_t)n + 8) * sizeof(double);
    hipLaunchKernelGGL(k_draw_verify, dim3(count), dim3(256), sm, c->stream,
                       reinterpret_cast<const DrawEvent *>(c->d_dv.p + 16), (int)count, (const double *)md, n, c->M,
                       c->kind, (const void *)c->d_obs_rm.p, (const int64_t *)c->d_offsets.p, Wlong, thr, res + 1);
    BHMM_HIP(hipGetLastError());
    unsigned int h[4] = {0, 0, 0, 0};
    BHMM_HIP(hipMemcpyAsync(h, res, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    BHMM_HIP(hipStreamSynchronize(c->stream));
    c->draw_checked = h[3];
    *ok = h[1] == 0 && h[2] == 0 && !(c->draw_test_redo && h[3] > 0);
    return BHMM_OK;
}

namespace {

struct Tmp { // scoped raw device allocations for the context-free entry points
    std::vector<void *> ptrs;
    ~Tmp()
    {
        for (void *p : ptrs)
            (void)hipFree(p);
    }
    template <typename T>
    int alloc(T **out, size_t count)
    {
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            set_error("hipMalloc failed");
            return BHMM_ERR_NO_MEM;
        }
        ptrs.push_back(p);
        *out = static_cast<T *>(p);
        return BHMM_OK;
    }
};

template <int N>
int sample_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
               const double *par1, const double *u, uint64_t seed, int32_t *paths, int64_t *counts,
               int64_t *n0, double *emis, double *stats_dev, bool defer_check = true, bool exact = false)
{
    // alpha -> CI workspace.  The boundary check of a speculative forward pass is only enqueued:
    // its verdict comes back with the results below, and a failed check repeats the call with the
    // waiting form (which then lengthens the warm-up / falls back to the exact pass).
    static const bool no_defer = getenv("BHMM_AMD_NO_DEFER") != nullptr; // (kernel experiments)
    c->fwd_defer = defer_check && !no_defer && !exact;
    // exact: the transfer-matrix pass (every boundary vector computed, none assumed) -- the repeat of a call
    // in which a watched draw did not stand (draw_verify.hpp)
    const bool spec_saved = c->spec_enabled;
    if (exact)
        c->spec_enabled = false;
    int rc = forward_ci(c, A, pi, par0, par1);
    c->spec_enabled = spec_saved;
    c->fwd_defer = false;
    if (rc)
        return rc;
    // A deferred verdict must be consumed on EVERY way out (allocation failure, launch error ...):
    // left pending, its bookkeeping is skipped and a later call reads stale verdict words.
    struct PendingVerdict {
        bhmm_ctx *c;
        ~PendingVerdict()
        {
            if (!c->fwd_pending)
                return;
            c->fwd_pending = false;
            (void)hipStreamSynchronize(c->stream);
            bool ok = false;
            (void)forward_ci_verdict(c, &ok); // (the call is failing anyway: outcome unused)
        }
    } pending_guard{c};
    const int K = c->K, n = c->n;
    const size_t nstat = (size_t)N * N + N;
    // parts per chunk for the map kernels (BHMM_AMD_SMP_PARTS overrides: kernel experiments)
    static const int parts_env = getenv("BHMM_AMD_SMP_PARTS") ? atoi(getenv("BHMM_AMD_SMP_PARTS")) : 0;
    // (k_smp_maps keeps within 128 VGPRs, i.e. four wavefronts per SIMD: 8 parts per chunk fill
    // them on the default plan of 32768 chunks; measured on configs[4]: P = 4 0.56, 8 0.48, 16 0.52 ms
    // for maps + stitch + apply)
    // Few, short chunks (the shard of one of eight ranks: 32 x 1e5 steps -> 16352 chunks of 196 steps) gave ONE
    // part per chunk, i.e. a quarter of a wavefront per SIMD walking 196 dependent steps: k_smp_maps took 325 us
    // of a 550 us path step.  Parts as short as 16 steps until the lanes fill the chip (round 6).
    int Pfill = 1;
    while (Pfill < 16 && c->Lmax / (2 * Pfill) >= 16 && (int64_t)c->G * Pfill < 262144)
        Pfill *= 2;
    const int P = parts_env > 0 ? parts_env : std::max(Pfill, c->Lmax >= 512 ? 8 : (c->Lmax >= 256 ? 4 : 1));
    const int nblk = (c->Gp / BLOCK) * P;
    const size_t esz = c->kind == EMIT_GAUSS ? 3 * (size_t)N : (c->kind == EMIT_DISC ? (size_t)c->M * N : 0);
    // scratch2: [path] | status | part maps | next-part states | lowest non-final step per part |
    // nibbles of the final steps (8 per word)
    const size_t npath = paths ? (size_t)c->total : 0;
    const int Lp = c->Lmax / P + 2;          // steps per part (upper bound)
    const int W8 = (Lp + 7) / 8 + 1;
    // ... | full step maps above the coalescence point (one word per step)
    if ((rc = c->d_scratch2.ensure((npath + 4 + (3 + (size_t)W8 + (size_t)Lp) * (size_t)c->Gp * P) *
                                   sizeof(int32_t))))
        return rc;
    int32_t *path = paths ? reinterpret_cast<int32_t *>(c->d_scratch2.p) : nullptr;
    int *status = nullptr; // (lives behind the counts, see below)
    uint32_t *fmap = reinterpret_cast<uint32_t *>(reinterpret_cast<int32_t *>(c->d_scratch2.p) + npath + 4);
    int32_t *nstate = reinterpret_cast<int32_t *>(fmap + (size_t)c->Gp * P);
    int32_t *dmark = nstate + (size_t)c->Gp * P;
    uint32_t *nib = reinterpret_cast<uint32_t *>(dmark + (size_t)c->Gp * P);
    uint32_t *gw = nib + (size_t)W8 * c->Gp * P;
    const int64_t Gp64 = c->Gp;
    // scratch: counts | reduced emission | status | emission partials | u   (the first three are
    // zeroed by ONE fill: every fill is a dispatch of its own, ~5 us on the stream)
    // emission partials: one table per workgroup, or (big discrete alphabets) a few global ones
    const bool bigM = c->kind == EMIT_DISC && c->bt_global;
    const size_t ntab = bigM ? (size_t)DISC_GLOBAL_TABLES : (size_t)nblk;
    const size_t dbl = nstat + esz + 1 + ntab * esz + (u ? (size_t)c->total : 0) + 8;
    if ((rc = c->d_scratch.ensure(dbl * sizeof(double))))
        return rc;
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(c->d_scratch.p);
    double *ered = reinterpret_cast<double *>(c->d_scratch.p) + nstat;
    status = reinterpret_cast<int *>(ered + esz);
    double *epart = ered + esz + 1;
    BHMM_HIP(hipMemsetAsync(cnt, 0, (nstat + esz + 1) * sizeof(double), c->stream));
    if (bigM)
        BHMM_HIP(hipMemsetAsync(epart, 0, ntab * esz * sizeof(double), c->stream));
    double *udev = nullptr;
    if (u) {
        udev = epart + ntab * esz;
        BHMM_HIP(hipMemcpyAsync(udev, u, (size_t)c->total * sizeof(double), hipMemcpyHostToDevice,
                                c->stream));
    }
    Model<N> m;
    fill_model_pub<N>(m, n, c->kind, c->M, A, pi, par0, par1);
    m.bt_global = bigM ? 1 : 0;
    // rows of a speculative pass: draws within 64 x the tolerance of its boundary check are recorded (the
    // measured deviation, usually far smaller, is applied when they are looked at)
    DrawWatch watch;
    if ((rc = draw_watch_prepare(c, c->rows32_valid && !exact ? 64.0 * c->spec_tol : 0.0, watch,
                                 reinterpret_cast<unsigned int *>(status) + 1))) // (second word of the cleared status slot)
        return rc;
    {
        // exact chunk-parallel sampling: maps per part, stitch, apply + statistics
        const Chunks chs = chunks_pub(c);
        const int64_t *offd = c->d_offsets.p;
        const void *obs_ci = c->d_obs_ci.p;
        const int64_t *soffd = c->d_soff.p ? c->d_soff.p : c->d_offsets.p;
        const double *wsd = c->d_ws.p, *Btd = c->d_Bt.p;
        // fp32 copies of the alpha rows, if the forward pass that just ran wrote them
        const float *r32 = c->rows32_valid ? c->d_ws32.p : nullptr;
#define BHMM_SMP_MAPS(KINDV, R32V)                                                                         \
    hipLaunchKernelGGL((k_smp_maps<N, KINDV, R32V>), dim3(nblk), dim3(BLOCK), 0, c->stream, m, chs, offd, soffd, \
                       wsd, r32, obs_ci, Btd, (const double *)udev, seed, P, fmap, status, dmark, nib, W8, Gp64, \
                       gw, Lp, watch)
        if (c->kind == EMIT_GAUSS) {
            if (r32)
                BHMM_SMP_MAPS(EMIT_GAUSS, true);
            else
                BHMM_SMP_MAPS(EMIT_GAUSS, false);
        } else if (c->kind == EMIT_DISC) {
            if (r32)
                BHMM_SMP_MAPS(EMIT_DISC, true);
            else
                BHMM_SMP_MAPS(EMIT_DISC, false);
        } else {
            if (r32)
                BHMM_SMP_MAPS(EMIT_EXPL, true);
            else
                BHMM_SMP_MAPS(EMIT_EXPL, false);
        }
#undef BHMM_SMP_MAPS
        BHMM_HIP(hipGetLastError());
        if ((int64_t)c->G * P >= (int64_t)32 * K) // long chains: one wavefront per trajectory
            hipLaunchKernelGGL(k_smp_stitch, dim3(K), dim3(64), 0, c->stream,
                               (const int32_t *)c->d_traj_c0.p, K, P, (const uint32_t *)fmap, nstate);
        else
            hipLaunchKernelGGL(k_smp_stitch_serial, dim3((K + SMP_STITCH_TPB - 1) / SMP_STITCH_TPB),
                               dim3(64), 0, c->stream, (const int32_t *)c->d_traj_c0.p, K, P,
                               (const uint32_t *)fmap, nstate);
        BHMM_HIP(hipGetLastError());
        const int32_t *ns = nstate, *dmk = dmark;
        const uint32_t *nb = nib, *gwc = gw;
        if (c->kind == EMIT_GAUSS)
            hipLaunchKernelGGL((k_smp_apply<N, EMIT_GAUSS>), dim3(nblk), dim3(BLOCK), 0, c->stream, m,
                               chs, offd, obs_ci, P, ns, path, cnt, epart, dmk, nb, W8, Gp64, gwc, Lp, status);
        else if (c->kind == EMIT_DISC) {
            const size_t smd = bigM ? 0 : (size_t)c->M * N * sizeof(double);
            if (smd > 64 * 1024)
                BHMM_HIP(hipFuncSetAttribute((const void *)(k_smp_apply<N, EMIT_DISC>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)smd));
            hipLaunchKernelGGL((k_smp_apply<N, EMIT_DISC>), dim3(nblk), dim3(BLOCK), smd, c->stream, m,
                               chs, offd, obs_ci, P, ns, path, cnt, epart, dmk, nb, W8, Gp64, gwc, Lp, status);
        }
        else
            hipLaunchKernelGGL((k_smp_apply<N, EMIT_EXPL>), dim3(nblk), dim3(BLOCK), 0, c->stream, m,
                               chs, offd, obs_ci, P, ns, path, cnt, epart, dmk, nb, W8, Gp64, gwc, Lp, status);
        BHMM_HIP(hipGetLastError());
        if (esz) {
            hipLaunchKernelGGL(k_add_partials, dim3((unsigned)esz), dim3(64), 0,
                               c->stream, (const double *)epart, (int)ntab, (int)esz, ered);
            BHMM_HIP(hipGetLastError());
        }
    }
    // counts | reduced emission statistics | [status, watched draws] are contiguous on the device: ONE copy, into
    // pinned memory where it fits (a pageable destination makes every small copy a round trip of its own)
    const size_t nres = nstat + esz + 1;
    std::vector<double> hres_v;
    double *hres = nullptr;
    if (nres <= 8192) {
        if (!c->h_small)
            BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_small), 8192 * sizeof(double), hipHostMallocDefault));
        hres = c->h_small;
    } else {
        hres_v.resize(nres);
        hres = hres_v.data();
    }
    int hstatus = 0;
    if (stats_dev) {
        // statistics stay on the device, packed for the caller's all-reduce
        hipLaunchKernelGGL(k_pack_path_stats, dim3(1), dim3(256), 0, c->stream,
                           (const unsigned long long *)cnt, (const double *)ered, n, N, c->M, c->kind,
                           1, stats_dev);
        BHMM_HIP(hipGetLastError());
        BHMM_HIP(hipMemcpyAsync(hres + nres - 1, status, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    } else {
        BHMM_HIP(hipMemcpyAsync(hres, cnt, nres * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    const unsigned long long *hc = reinterpret_cast<const unsigned long long *>(hres);
    const double *he = hres + nstat;
    const int *hst = reinterpret_cast<const int *>(hres + nres - 1); // [status | watched draws]
    if (paths)
        BHMM_HIP(hipMemcpyAsync(paths, path, (size_t)c->total * sizeof(int32_t),
                                hipMemcpyDeviceToHost, c->stream));
    BHMM_HIP(hipStreamSynchronize(c->stream));
    if (c->fwd_pending) {
        c->fwd_pending = false;
        bool ok = false;
        if ((rc = forward_ci_verdict(c, &ok)))
            return rc;
        if (!ok) // boundaries did not verify: everything above ran on wrong alpha rows
            return sample_run<N>(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, stats_dev,
                                 false);
    }
    hstatus = hst[0];
    const unsigned int nwatched = watch.count ? (unsigned int)hst[1] : 0u;
    c->draw_alpha_dev = watch.count ? (double)c->spec_last_dev : 0.0;
    if (nwatched) {
        // draws inside 64 x the deviation the boundary check measured: decided again on the serial recursion
        // over a long window; if one does not stand, the whole call again on the transfer-matrix rows
        bool ok = false;
        const double thr = c->draw_watch_tol > 0.0 ? c->draw_watch_tol
                                                   : 64.0 * std::max((double)c->spec_last_dev, 1e-16);
        if ((rc = draw_verify_run(c, A, pi, par0, par1, nwatched, thr, 8 * (int64_t)std::max(c->spec_W, 64), &ok)))
            return rc;
        if (!ok) {
            const unsigned int ev = c->draw_events, ck = c->draw_checked;
            rc = sample_run<N>(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, stats_dev, false, true);
            c->draw_events = ev;
            c->draw_checked = ck;
            c->draw_redone = 1;
            return rc;
        }
    }
    if (hstatus) {
        set_error("random choice found no state: alpha/A not normalisable (_hidden.c:299-304)");
        return hstatus;
    }
    if (stats_dev)
        return BHMM_OK;
    if (counts)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                counts[i * n + j] = (int64_t)hc[i * N + j];
    if (n0)
        for (int i = 0; i < n; ++i)
            n0[i] = (int64_t)hc[N * N + i];
    if (emis) {
        if (c->kind == EMIT_GAUSS)
            for (int w = 0; w < 3; ++w)
                for (int i = 0; i < n; ++i)
                    emis[w * n + i] = he[w * N + i];
        else if (c->kind == EMIT_DISC)
            for (int i = 0; i < n; ++i)
                for (int o = 0; o < c->M; ++o)
                    emis[(size_t)i * c->M + o] = he[(size_t)o * N + i];
    }
    return BHMM_OK;
}

// time segments of at most seglen steps for the segment-parallel Viterbi pass (which = 0) / backward
// sampler (which = 1); rebuilt only when the length changes
int wide_path_plan(bhmm_ctx *c, int which, int64_t seglen, Segs &sg)
{
    bhmm_ctx::PathPlan &pp = c->pplan[which];
    if (pp.nseg == 0 || pp.seglen != seglen) {
        plan::SegPlan sp;
        plan::plan_segments(c->offsets, c->K, seglen, 1, sp);
        const int ns = (int)sp.traj.size();
        int rc;
        if ((rc = pp.traj.ensure(std::max(ns, 1))) || (rc = pp.len.ensure(std::max(ns, 1))) ||
            (rc = pp.t0.ensure(std::max(ns, 1))) || (rc = pp.traj0.ensure(c->K + 1)))
            return rc;
        BHMM_HIP(hipMemcpyAsync(pp.traj0.p, sp.traj0.data(), (c->K + 1) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        BHMM_HIP(hipMemcpyAsync(pp.traj.p, sp.traj.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        BHMM_HIP(hipMemcpyAsync(pp.len.p, sp.len.data(), ns * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        BHMM_HIP(hipMemcpyAsync(pp.t0.p, sp.t0.data(), ns * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
        BHMM_HIP(hipStreamSynchronize(c->stream)); // (the vectors go out of scope)
        pp.nseg = ns;
        pp.seglen = seglen;
        pp.maxlen = 0;
        for (int32_t l : sp.len)
            pp.maxlen = std::max<int64_t>(pp.maxlen, l);
    }
    sg.traj = pp.traj.p;
    sg.len = pp.len.p;
    sg.t0 = pp.t0.p;
    sg.nseg = pp.nseg;
    return BHMM_OK;
}

// ---- 9..64 states ---------------------------------------------------------------------
// out_fmt: 0 = int32 paths to a host buffer (the reference's type, hidden.pyx:161-162),
// 1 = one byte per step to a host buffer, 2 = one byte per step into a device buffer of the
// context's device (written by the back-trace kernels directly, no copy at all)
int wide_viterbi_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                     const double *par1, void *paths_out, int out_fmt)
{
    WideModel m;
    int rc = wide_model_pub(c, c->kind, A, pi, par0, par1, m);
    if (rc)
        return rc;
    const int K = c->K, n = c->n, NP = c->wide ? c->N : 8, GP = 64 / NP;
    c->viterbi_chunked = false;
    c->vit_mended = 0;
    const size_t gpad = c->wide ? 0 : (size_t)c->Gp;
    if ((rc = c->d_scratch.ensure((size_t)c->total * n)) ||
        (rc = c->d_scratch2.ensure(((size_t)c->total + K + 3 * gpad) * sizeof(int32_t))))
        return rc;
    uint8_t *ptr = reinterpret_cast<uint8_t *>(c->d_scratch.p);
    int32_t *last = reinterpret_cast<int32_t *>(c->d_scratch2.p);
    int32_t *path = last + K;
    uint8_t *path8 = out_fmt == 2 ? static_cast<uint8_t *>(paths_out) : reinterpret_cast<uint8_t *>(path);
    uint32_t *vmaps = reinterpret_cast<uint32_t *>(path + c->total); // chunked run: chunk maps,
    int32_t *vend = reinterpret_cast<int32_t *>(vmaps + gpad);       // state at each chunk's end
    int32_t *vcoal = vend + gpad; // step below which the map pass wrote the path itself (k_vit_walk)
    // U trajectories per lane group.  Measured on configs[1] (256 trajectories): the kernel is
    // bound by the instruction stream of each wavefront, not by latency, and SIMDs are plentiful
    // (32 of 1024 busy), so U = 1 is fastest (U = 4 was 3.5x slower); the parameter stays for
    // batches with more trajectories than SIMD slots.
    const int U = 1;
    const dim3 grid((K + GP * U - 1) / (GP * U)), blk(64);
    const void *obs = c->d_obs_rm.p;
    const int64_t *off = c->d_offsets.p;
    // emission probabilities in a separate, fully parallel pass when the (total, n) matrix
    // fits (it is what the reference materialises anyway, maximum_likelihood.py:345-347)
    int vkind = c->kind;
    // n <= 8 with a chunk plan: the chunk-parallel kernel evaluates the emission itself (gathers
    // from B / evaluates the gaussian density): no (total, n) emission matrix is written and re-read
    const bool disc_direct = !c->wide && c->kind != EMIT_EXPL && c->spec_enabled && c->G > K;
    // 9..64 states over time segments, discrete: the kernel gathers from B itself (no (total, n) matrix).
    // Measured for the Gaussian kind too: the division by sigma and the exponential in every step,
    // warm-ups included, cost more than the matrix pass saves (64 states: 9.0 -> 9.7 ms, 16: 0.67 -> 0.81).
    // (round 6: the Gaussian density in the step too, with the reciprocal-based quotient and the one-block
    // exponential of k_pobs_lanes -- at 64 states: 6.5 GB less written and read again per call)
    static const bool gauss_matrix = getenv("BHMM_AMD_VIT_GAUSS_MATRIX") != nullptr; // (experiments: the round-5 way)
    const bool wide_direct = c->wide && (c->kind == EMIT_DISC || (c->kind == EMIT_GAUSS && !gauss_matrix)) &&
                             c->spec_enabled && !c->vit_seg_given_up;
    if (c->kind != EMIT_EXPL && !disc_direct && !wide_direct) {
        size_t freeb = 0, totb = 0;
        const size_t need = (size_t)c->total * n * sizeof(double);
        if (hipMemGetInfo(&freeb, &totb) == hipSuccess && need + ((size_t)1 << 30) < freeb + c->d_alpha_rm.n * sizeof(double) &&
            c->d_alpha_rm.ensure((size_t)c->total * n) == BHMM_OK) {
            const dim3 pg((unsigned)((c->total + 255) / 256)), pb(256);
            const dim3 pl((unsigned)(((size_t)c->total * n + 256 * POBS_LANES_R - 1) / (256 * POBS_LANES_R)));
#define BHMM_POBS_LANES(NLV)                                                                        \
    do {                                                                                            \
        if (c->kind == EMIT_GAUSS)                                                                  \
            hipLaunchKernelGGL((k_pobs_lanes<EMIT_GAUSS, NLV>), pl, pb, 0, c->stream, m, obs,       \
                               c->total, c->d_alpha_rm.p);                                          \
        else                                                                                        \
            hipLaunchKernelGGL((k_pobs_lanes<EMIT_DISC, NLV>), pl, pb, 0, c->stream, m, obs,        \
                               c->total, c->d_alpha_rm.p);                                          \
    } while (0)
            // one thread per element where n is a power of two (coalesced stores), else per step
            if (n == 2)
                BHMM_POBS_LANES(2);
            else if (n == 4)
                BHMM_POBS_LANES(4);
            else if (n == 8)
                BHMM_POBS_LANES(8);
            else if (n == 16)
                BHMM_POBS_LANES(16);
            else if (n == 32)
                BHMM_POBS_LANES(32);
            else if (n == 64)
                BHMM_POBS_LANES(64);
            else if (c->kind == EMIT_GAUSS)
                hipLaunchKernelGGL((k_pobs_all<EMIT_GAUSS>), pg, pb, 0, c->stream, m, obs, c->total,
                                   c->d_alpha_rm.p);
            else
                hipLaunchKernelGGL((k_pobs_all<EMIT_DISC>), pg, pb, 0, c->stream, m, obs, c->total,
                                   c->d_alpha_rm.p);
#undef BHMM_POBS_LANES
            BHMM_HIP(hipGetLastError());
            obs = c->d_alpha_rm.p;
            vkind = EMIT_EXPL;
        } else {
            (void)hipGetLastError();
        }
    }
#define BHMM_WV(NPV, KINDV)                                                                     \
    hipLaunchKernelGGL((k_wide_viterbi_fwd<NPV, KINDV, 1>), grid,                                  \
                       blk, 0, c->stream, m, off, K, obs, ptr, last)
#define BHMM_WV_KIND(NPV)                                 \
    do {                                                  \
        if (vkind == EMIT_GAUSS)                          \
            BHMM_WV(NPV, EMIT_GAUSS);                     \
        else if (vkind == EMIT_DISC)                      \
            BHMM_WV(NPV, EMIT_DISC);                      \
        else                                              \
            BHMM_WV(NPV, EMIT_EXPL);                      \
    } while (0)
    // back-trace of an accepted chunk-parallel run (k_vit_walk, maps -> stitch -> paths)
    // (models of up to four states: four lanes per chunk, 16 chunks per wavefront)
    const bool vit4 = !c->wide && n <= 4;
    auto launch_walks_np = [&](auto npc) {
        constexpr int VNP = decltype(npc)::value;
        const Chunks chs = chunks_pub(c);
        const dim3 wg((c->G + 64 / VNP - 1) / (64 / VNP));
        if (out_fmt == 0)
            hipLaunchKernelGGL((k_vit_walk<VNP, false, int32_t>), wg, dim3(64), 0, c->stream, chs, c->G, n,
                               (const uint8_t *)ptr, (const int32_t *)nullptr, vmaps, path, vcoal);
        else
            hipLaunchKernelGGL((k_vit_walk<VNP, false, uint8_t>), wg, dim3(64), 0, c->stream, chs, c->G, n,
                               (const uint8_t *)ptr, (const int32_t *)nullptr, vmaps, path8, vcoal);
        if ((int64_t)c->G >= (int64_t)32 * K)
            hipLaunchKernelGGL(k_smp_stitch, dim3(K), dim3(64), 0, c->stream,
                               (const int32_t *)c->d_traj_c0.p, K, 1, (const uint32_t *)vmaps, vend,
                               (const int32_t *)last);
        else
            hipLaunchKernelGGL(k_smp_stitch_serial, dim3((K + SMP_STITCH_TPB - 1) / SMP_STITCH_TPB),
                               dim3(64), 0, c->stream, (const int32_t *)c->d_traj_c0.p, K, 1,
                               (const uint32_t *)vmaps, vend, (const int32_t *)last);
        if (out_fmt == 0)
            hipLaunchKernelGGL((k_vit_walk<VNP, true, int32_t>), wg, dim3(64), 0, c->stream, chs, c->G, n,
                               (const uint8_t *)ptr, (const int32_t *)vend, (uint32_t *)nullptr, path, vcoal);
        else
            hipLaunchKernelGGL((k_vit_walk<VNP, true, uint8_t>), wg, dim3(64), 0, c->stream, chs, c->G, n,
                               (const uint8_t *)ptr, (const int32_t *)vend, (uint32_t *)nullptr, path8, vcoal);
    };
    auto launch_walks = [&]() {
        if (vit4)
            launch_walks_np(std::integral_constant<int, 4>{});
        else
            launch_walks_np(std::integral_constant<int, 8>{});
    };
    bool walks_in_flight = false;
    // n <= 8 with a chunk plan: the chunk-parallel run first; its back-pointers are accepted only
    // if every chunk boundary verifies and no decision was close (k_viterbi_chunks)
    bool done = false;
    if (!c->wide && (vkind == EMIT_EXPL || disc_direct) && c->spec_enabled && c->G > K) {
        int maxchunks = 1;
        for (int k = 0; k < K; ++k)
            maxchunks = std::max(maxchunks, c->traj_c0[k + 1] - c->traj_c0[k]);
        const double tol = 1e-11, margin = std::max(1e-7, 16.0 * tol * maxchunks);
        if ((rc = c->d_aentry.ensure((size_t)c->Gp * 8)) || (rc = c->d_aexit.ensure((size_t)c->Gp * 8)) ||
            (rc = c->d_specres.ensure(4)))
            return rc;
        if (!c->h_specres)
            BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_specres), 4 * sizeof(unsigned int),
                                   hipHostMallocDefault));
        const Chunks chs = chunks_pub(c);
        // discrete: B in LDS when it is small enough to leave four wavefronts per SIMD their room
        const size_t smB = (disc_direct && c->kind == EMIT_DISC &&
                            (size_t)n * c->M * sizeof(double) <= 16 * 1024)
                               ? (size_t)n * c->M * sizeof(double) : 0;
        // A warm-up too short for some boundary (the survivors of that stretch had not met yet) shows
        // as a boundary out of tolerance: the run is repeated with twice, then four times the warm-up
        // (one more pass of ~1 ms each) before the serial kernel (tens of ms) has to decide.
        // The warm-up of THIS recursion (round 4): the max-product chains coalesce faster than the
        // E-step's filter forgets to 1e-13 (configs[1]: bit-identical boundaries from W ~ 128 on, the
        // E-step's probe says 280).  The first call on new observations starts from the E-step's
        // length; every later call that verified tries three quarters of the last good length, until
        // one does not verify -- that one is repeated with the last good length in the same call, and
        // the search ends (at most one lost pass per set of observations).
        int W_try = c->vit_W > 0 ? c->vit_W : c->spec_W;
        bool exploring = false;
        if (c->vit_W > 0 && c->vit_explore && !c->spec_W_fixed) {
            const int Wn = std::max(32, (c->vit_W * 3 / 4 + 7) / 8 * 8);
            if (Wn < c->vit_W && Wn > c->vit_bad) {
                W_try = Wn;
                exploring = true;
            } else {
                c->vit_explore = false;
            }
        }
        for (int attempt = 0; attempt < 3; ++attempt) {
        if (attempt > 0) {
            if (exploring) { // the shorter warm-up did not verify: back to the one that did
                c->vit_bad = W_try;
                c->vit_explore = false;
                exploring = false;
                W_try = c->vit_W;
            } else {
                W_try *= 2;
            }
        }
        // first without the close-decision count; bit-identical boundaries make it irrelevant
        for (int pass = 0; pass < 2; ++pass) {
            BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
#define BHMM_VC_NP(VNP, KINDV, MARGINV)                                                             \
    hipLaunchKernelGGL((k_viterbi_chunks<VNP, KINDV, MARGINV>),                                     \
                       dim3((c->G + 64 / VNP - 1) / (64 / VNP)), dim3(64), smB, c->stream, m, chs,   \
                       c->G, off, obs, W_try, margin, ptr, last, c->d_aentry.p, c->d_aexit.p,       \
                       c->d_specres.p, smB ? 1 : 0)
#define BHMM_VC(KINDV, MARGINV)                 \
    do {                                        \
        if (vit4)                               \
            BHMM_VC_NP(4, KINDV, MARGINV);      \
        else                                    \
            BHMM_VC_NP(8, KINDV, MARGINV);      \
    } while (0)
            if (disc_direct && c->kind == EMIT_DISC) {
                if (pass == 0)
                    BHMM_VC(EMIT_DISC, false);
                else
                    BHMM_VC(EMIT_DISC, true);
            } else if (disc_direct) {
                if (pass == 0)
                    BHMM_VC(EMIT_GAUSS, false);
                else
                    BHMM_VC(EMIT_GAUSS, true);
            } else {
                if (pass == 0)
                    BHMM_VC(EMIT_EXPL, false);
                else
                    BHMM_VC(EMIT_EXPL, true);
            }
#undef BHMM_VC
#undef BHMM_VC_NP
            BHMM_HIP(hipGetLastError());
            if (vit4)
                hipLaunchKernelGGL((k_viterbi_check<4>), dim3((c->G + 255) / 256), dim3(256), 0, c->stream,
                                   chs, c->G, (const double *)c->d_aentry.p,
                                   (const double *)c->d_aexit.p, tol, c->d_specres.p);
            else
                hipLaunchKernelGGL((k_viterbi_check<8>), dim3((c->G + 255) / 256), dim3(256), 0, c->stream,
                                   chs, c->G, (const double *)c->d_aentry.p,
                                   (const double *)c->d_aexit.p, tol, c->d_specres.p);
            BHMM_HIP(hipGetLastError());
            BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 4 * sizeof(unsigned int),
                                    hipMemcpyDeviceToHost, c->stream));
            // the normal case is "all boundaries bit-identical": the back-trace is enqueued behind
            // the first pass before its verdict is known (one host round trip less); the walks only
            // read back-pointers, which are valid state indices whatever the verdict, and are
            // repeated after any further pass
            const bool speculative = attempt == 0 && pass == 0;
            if (speculative)
                launch_walks();
            BHMM_HIP(hipStreamSynchronize(c->stream));
            walks_in_flight = speculative && c->h_specres[3] == 0;
            if (c->h_specres[3] == 0 || c->h_specres[0] != 0)
                break; // the serial run outright / out of tolerance: the count cannot help
        }
        const bool accepted = c->h_specres[3] == 0 || (c->h_specres[0] == 0 && c->h_specres[2] == 0);
        if (exploring && !accepted)
            continue; // (a shorter warm-up must be as good as the longer one was, not merely tolerable)
        if (c->h_specres[0] == 0) {
            c->vit_W = W_try; // (what worked is where the next call on these observations starts)
            break; // every boundary within tolerance: a longer warm-up changes nothing
        }
        }
        // all boundaries bit-identical: it is the serial run; else within tolerance and no
        // close decision
        done = c->h_specres[3] == 0 || (c->h_specres[0] == 0 && c->h_specres[2] == 0);
        c->viterbi_chunked = done;
        float dev;
        memcpy(&dev, &c->h_specres[1], sizeof(float));
        c->spec_last_dev = dev;
        c->viterbi_close = c->h_specres[2];
    }
    // 9..64 states: one lane group per time segment (k_wide_viterbi_seg); accepted only if EVERY
    // segment arrives at its first step with the bit pattern its predecessor left there -- then the
    // back-pointers are the serial run's, by induction from the exact first segment of each trajectory
    if (c->wide && c->spec_enabled && !c->vit_seg_given_up) {
        // warm-up: its own, not the E-step's.  Every length is exact (bitwise check + fix-up rounds), so the length
        // only trades warm-up steps (W / segment length of the first pass) against rounds (1.1 ms each at configs[3],
        // whatever the number of flagged segments) -- and the E-step's length says little about that: its boundary
        // check asks for 1e-11 on every boundary, which after a dozen EM iterations on configs[3] takes 904 steps,
        // while the max-product vectors are within rounding noise of their predecessors' after 128 (one round
        // either way: 17.6 ms against 23.6).  Start at 128 steps (less if the filter forgets faster); a call that
        // needed three or more rounds doubles the length for the next call on these observations, up to the E-step's.
        const int W_estep = std::max(64, c->spec_W > 0 ? (c->spec_W + 7) / 8 * 8 : 128);
        int W_try = c->vit_W > 0 ? c->vit_W : std::min(128, W_estep);
        const bool exploring = false;
        if ((rc = c->d_specres.ensure(4)))
            return rc;
        if (!c->h_specres)
            BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_specres), 4 * sizeof(unsigned int),
                                   hipHostMallocDefault));
        // the path-margin acceptance (k_vit_margin): every vector of the first pass is kept ([total][n], in the
        // buffer of the 65..128-state E-step's W rows), boundaries equal to vm_tol count as usable
        const double vm_tol = 1e-12;
        double *vall = nullptr;
        int64_t maxT = 0;
        // (up to 64 states one fix-up round is short -- 1.1 ms at configs[3] -- and cheaper than keeping the
        // vectors and checking the margins: only observations that needed two or more rounds before take this way)
        if (c->vit_margin && c->vit_margin_want && c->d_gW.ensure((size_t)c->total * n) == BHMM_OK) {
            vall = c->d_gW.p;
            for (int k = 0; k < K; ++k)
                maxT = std::max(maxT, c->offsets[k + 1] - c->offsets[k]);
        } else {
            (void)hipGetLastError();
        }
        c->vit_margin_used = 0;
        c->vit_margin_close = 0;
        // back-trace over the segments: maps, stitch, apply
        auto seg_walks = [&]() -> int {
            // on a plan of its own, eight times finer than the pass's: a walk is a chain of dependent look-ups, its
            // time the length of a segment (configs[3]: 2 x 0.49 ms on the 2048 segments of the pass)
            Segs sgw;
            int rcw;
            static const int walk_div = getenv("BHMM_AMD_WALK_DIV") ? atoi(getenv("BHMM_AMD_WALK_DIV")) : 8;
            if ((rcw = wide_path_plan(c, 2, std::max<int64_t>(256, c->pplan[0].seglen / walk_div), sgw)))
                return rcw;
            sgw.W = 0;
            if ((rcw = c->d_vmaps.ensure((size_t)sgw.nseg * 64)) || (rcw = c->d_vend.ensure((size_t)sgw.nseg)))
                return rcw;
            hipLaunchKernelGGL((k_wide_vit_walk<false, uint8_t>), dim3(sgw.nseg), dim3(64), 0, c->stream, off, sgw, n,
                               (const uint8_t *)ptr, c->d_vmaps.p, (const uint8_t *)nullptr, (uint8_t *)nullptr);
            hipLaunchKernelGGL(k_wide_vit_stitch, dim3((K + 63) / 64), dim3(64), 0, c->stream,
                               (const int32_t *)c->pplan[2].traj0.p, K, (const uint8_t *)c->d_vmaps.p, 64,
                               (const int32_t *)last, c->d_vend.p);
            if (out_fmt == 0)
                hipLaunchKernelGGL((k_wide_vit_walk<true, int32_t>), dim3(sgw.nseg), dim3(64), 0, c->stream, off, sgw, n,
                                   (const uint8_t *)ptr, (uint8_t *)nullptr, (const uint8_t *)c->d_vend.p, path);
            else
                hipLaunchKernelGGL((k_wide_vit_walk<true, uint8_t>), dim3(sgw.nseg), dim3(64), 0, c->stream, off, sgw, n,
                                   (const uint8_t *)ptr, (uint8_t *)nullptr, (const uint8_t *)c->d_vend.p, path8);
            BHMM_HIP(hipGetLastError());
            return BHMM_OK;
        };
        for (int attempt = 0; attempt < 2 && !done; ++attempt) {
            if (attempt > 0) {
                W_try *= 2;
            }
            // two lane groups' worth of segments per SIMD, none shorter than two warm-ups
            const int64_t want = (int64_t)c->vit_seg_per_simd * c->num_simd * GP;
            const int64_t seglen = std::max<int64_t>((c->total + want - 1) / want, c->vit_seg_warmups * (int64_t)W_try);
            Segs sg;
            if ((rc = wide_path_plan(c, 0, seglen, sg)))
                return rc;
            if (sg.nseg <= K)
                break; // nothing to gain: one segment per trajectory is the serial kernel
            sg.W = W_try;
            if ((rc = c->d_aentry.ensure((size_t)sg.nseg * NP)) || (rc = c->d_aexit.ensure((size_t)sg.nseg * NP)))
                return rc;
            const dim3 sgrid((sg.nseg + GP * WVS_WPB - 1) / (GP * WVS_WPB)), sblk(64 * WVS_WPB);
            if ((rc = c->d_vckpt.ensure(((size_t)(c->total >> 6) + 1) * NP)) ||
                (rc = c->d_vflag.ensure(2 * (size_t)sg.nseg))) // ([nseg] flagged | [nseg] flagged and further than vm_tol)
                return rc;
#define BHMM_WVS(NPV, KINDV, FIXV)                                                                    \
    hipLaunchKernelGGL((k_wide_viterbi_seg<NPV, KINDV, FIXV>), sgrid, sblk, 0, c->stream, m, off, sg,  \
                       obs, ptr, last, c->d_aentry.p, c->d_aexit.p, c->d_vckpt.p,                     \
                       (const uint8_t *)c->d_vflag.p, FIXV ? (double *)nullptr : vall)
#define BHMM_WVS_KIND(NPV, FIXV)                                                                      \
    do {                                                                                              \
        if (vkind == EMIT_GAUSS)                                                                      \
            BHMM_WVS(NPV, EMIT_GAUSS, FIXV);                                                          \
        else if (vkind == EMIT_DISC)                                                                  \
            BHMM_WVS(NPV, EMIT_DISC, FIXV);                                                           \
        else                                                                                          \
            BHMM_WVS(NPV, EMIT_EXPL, FIXV);                                                           \
        hipLaunchKernelGGL((k_wide_vit_check<NPV>), dim3((sg.nseg + 255) / 256), dim3(256), 0,      \
                           c->stream, sg, c->d_aentry.p, (const double *)c->d_aexit.p, c->d_vflag.p,  \
                           c->d_specres.p, (!FIXV && vall) ? vm_tol : 0.0);                           \
    } while (0)
#define BHMM_WVS_NP(FIXV)              \
    do {                               \
        if (NP == 16)                  \
            BHMM_WVS_KIND(16, FIXV);   \
        else if (NP == 32)             \
            BHMM_WVS_KIND(32, FIXV);   \
        else                           \
            BHMM_WVS_KIND(64, FIXV);   \
    } while (0)
#define BHMM_WVS_MEND(NPV, KINDV)                                                                      \
    hipLaunchKernelGGL((k_wide_viterbi_seg<NPV, KINDV, true>), sgrid, sblk, 0, c->stream, m, off, sg,   \
                       obs, ptr, last, c->d_aentry.p, c->d_aexit.p, c->d_vckpt.p,                      \
                       (const uint8_t *)c->d_vflag.p + sg.nseg, vall, vm_tol, c->d_specres.p + 1)
#define BHMM_WVS_MEND_KIND(NPV)                   \
    do {                                          \
        if (vkind == EMIT_GAUSS)                  \
            BHMM_WVS_MEND(NPV, EMIT_GAUSS);       \
        else if (vkind == EMIT_DISC)              \
            BHMM_WVS_MEND(NPV, EMIT_DISC);        \
        else                                      \
            BHMM_WVS_MEND(NPV, EMIT_EXPL);        \
    } while (0)
            // pass 0 with warm-ups, then fix-up rounds while any boundary is not bit-identical
            const int max_rounds = 12;
            int round = 0;
            bool margin_accepted = false;
            bool allow_mend = c->vit_mend, mended = false;
            c->vit_mended = 0;
            for (; round <= max_rounds; ++round) {
                BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                lds_poison(c->stream);
                if (round == 0)
                    BHMM_WVS_NP(false);
                else
                    BHMM_WVS_NP(true);
                BHMM_HIP(hipGetLastError());
                BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 4 * sizeof(unsigned int),
                                        hipMemcpyDeviceToHost, c->stream));
                BHMM_HIP(hipStreamSynchronize(c->stream));
                if (round == 0) {
                    c->vit_seg_mismatch = (int)c->h_specres[3];
                    c->vit_far = (int)c->h_specres[0];
                }
                if (c->h_specres[3] == 0)
                    break;
                int spliced = 0;
                if (round == 0 && vall && c->h_specres[0] != 0 && (int64_t)c->h_specres[0] * 2 <= sg.nseg && allow_mend) {
                    mended = true;
                    // Some boundaries are further than vm_tol from their predecessors' vectors (the warm-up was
                    // too short THERE; the max-product vectors of a metastable model need several times the
                    // filter's forgetting length at a few boundaries -- configs[3]: 276 of 2048 after 256 steps,
                    // none after 904): those segments alone are run again from the predecessor's vector until
                    // they are within vm_tol of a kept vector of the first pass (k_wide_viterbi_seg, mend_tol)
                    // instead of lengthening every warm-up.  If one reaches its end the rounds decide.
                    BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                    lds_poison(c->stream);
                    if (NP == 16)
                        BHMM_WVS_MEND_KIND(16);
                    else if (NP == 32)
                        BHMM_WVS_MEND_KIND(32);
                    else
                        BHMM_WVS_MEND_KIND(64);
                    BHMM_HIP(hipGetLastError());
                    unsigned int notmet = 0;
                    BHMM_HIP(hipMemcpyAsync(&notmet, c->d_specres.p + 1, sizeof(unsigned int), hipMemcpyDeviceToHost,
                                            c->stream));
                    BHMM_HIP(hipStreamSynchronize(c->stream));
                    c->vit_mended = (int)c->h_specres[0];
                    if (notmet == 0) {
                        spliced = (int)c->h_specres[0];
                        c->h_specres[0] = 0;
                    }
                }
                if (round == 0 && vall && c->h_specres[0] == 0) {
                    // every boundary (and splice) within vm_tol: the path of this pass, and the margins of the
                    // decisions on it
                    const int maxseg = (int)((maxT + seglen - 1) / seglen) + 1 + spliced;
                    const double margin = std::max(1e-10, 16.0 * (2e-15 * (double)maxT + vm_tol * maxseg));
                    if ((rc = seg_walks()))
                        return rc;
                    BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                    const size_t smm = (size_t)n * (n | 1) * sizeof(double); // (odd pitch, k_vit_margin)
                    const dim3 mgrid(sg.nseg, (unsigned)((c->pplan[0].maxlen + VM_STEPS - 1) / VM_STEPS)); // (the longest REAL segment)
                    static const bool vm_global = getenv("BHMM_AMD_VM_GLOBAL") != nullptr; // (experiment: A^T from L2)
                    if (vm_global) {
                        if ((rc = c->d_gAt.ensure((size_t)n * n)))
                            return rc;
                        hipLaunchKernelGGL(k_vm_transpose, dim3((n * n + 255) / 256), dim3(256), 0, c->stream, m.A, n, c->d_gAt.p);
                        if (out_fmt == 0)
                            hipLaunchKernelGGL((k_vit_margin<int32_t, 1, false>), mgrid, dim3(256), 0, c->stream, (const double *)c->d_gAt.p, n, off,
                                               sg, (const double *)vall, (const int32_t *)path, margin, c->d_specres.p);
                        else
                            hipLaunchKernelGGL((k_vit_margin<uint8_t, 1, false>), mgrid, dim3(256), 0, c->stream, (const double *)c->d_gAt.p, n, off,
                                               sg, (const double *)vall, (const uint8_t *)path8, margin, c->d_specres.p);
                    } else if (out_fmt == 0)
                        hipLaunchKernelGGL((k_vit_margin<int32_t, 1>), mgrid, dim3(256), smm, c->stream, m.A, n, off,
                                           sg, (const double *)vall, (const int32_t *)path, margin, c->d_specres.p);
                    else
                        hipLaunchKernelGGL((k_vit_margin<uint8_t, 1>), mgrid, dim3(256), smm, c->stream, m.A, n, off,
                                           sg, (const double *)vall, (const uint8_t *)path8, margin, c->d_specres.p);
                    BHMM_HIP(hipGetLastError());
                    BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 4 * sizeof(unsigned int),
                                            hipMemcpyDeviceToHost, c->stream));
                    BHMM_HIP(hipStreamSynchronize(c->stream));
                    c->vit_margin_close = (int)c->h_specres[2];
                    if (c->h_specres[2] == 0) {
                        c->vit_margin_used = 1;
                        margin_accepted = true;
                        break;
                    }
                    // (a close decision on the path: the rounds decide)
                }
                if (round == 0 && mended) {
                    // The rounds compare BITWISE with the kept vectors of the first pass; a mended segment now
                    // holds exact vectors up to its splice and first-pass vectors behind it, so a repeated run
                    // would stop at the first kept vector and take the rest for exact.  A pass that was mended and
                    // then not accepted is therefore run again from scratch, without mending (one first pass lost).
                    allow_mend = false;
                    mended = false;
                    round = -1;
                }
            }
#undef BHMM_WVS_NP
#undef BHMM_WVS_KIND
#undef BHMM_WVS
#undef BHMM_WVS_MEND_KIND
#undef BHMM_WVS_MEND
            c->vit_seg_rounds = round;
            // (a round runs as long as its longest flagged segment needs to fall onto a vector of the first pass
            // again: 1.1 ms at configs[3] on white-noise observations, 6.7 ms -- half a first pass -- on
            // observations drawn from the model, where 1800 of 2048 boundaries carry rounding noise.  The margins
            // cost about 1.3 ms there: wanted from the next call on when rounds were many, or the flagged
            // segments more than a quarter)
            if (round >= 2 || (round >= 1 && !margin_accepted && (int64_t)c->vit_seg_mismatch * 4 > sg.nseg))
                c->vit_margin_want = true;
            const bool accepted = c->h_specres[3] == 0 || margin_accepted;
            // (How many boundaries the first pass left to the fix-up does not say whether the warm-up was
            // too short -- most of them are rounding noise, and a round costs the same for one segment as
            // for a thousand: doubling the warm-up on that count was measured slower everywhere.)
            if (exploring && !accepted)
                continue;
            if (accepted) {
                done = true;
                // (boundaries further than 1e-12 apart keep the margin rule from being asked: a longer warm-up for
                // the next call, like after three or more rounds -- never beyond the E-step's)
                const bool longer = (vall && c->vit_far > 0 && !margin_accepted) || (round >= 3 && !margin_accepted);
                c->vit_W = (longer && W_try < W_estep) ? std::min(2 * W_try, W_estep) : W_try;
            }
        }
        if (!done && c->pplan[0].nseg > K)
            c->vit_seg_given_up = true; // these observations go to the serial kernel from now on
        c->viterbi_chunked = done;
        if (done) {
            if (!c->vit_margin_used && (rc = seg_walks())) // (a margin-accepted pass has its path already)
                return rc;
            walks_in_flight = true;
        }
    }
    if (done)
        ;
    else if (NP == 8)
        BHMM_WV_KIND(8);
    else if (NP == 16)
        BHMM_WV_KIND(16);
    else if (NP == 32)
        BHMM_WV_KIND(32);
    else
        BHMM_WV_KIND(64);
    BHMM_HIP(hipGetLastError());
    if (done) {
        if (!walks_in_flight)
            launch_walks();
    } else if (out_fmt == 0) {
        hipLaunchKernelGGL(k_wide_viterbi_trace<int32_t>, dim3(K), dim3(64), 0, c->stream, off, K, n,
                           (const uint8_t *)ptr, (const int32_t *)last, path);
    } else {
        hipLaunchKernelGGL(k_wide_viterbi_trace<uint8_t>, dim3(K), dim3(64), 0, c->stream, off, K, n,
                           (const uint8_t *)ptr, (const int32_t *)last, path8);
    }
    BHMM_HIP(hipGetLastError());
    if (out_fmt == 2) { // the paths are where the caller wants them; the call still completes them
        BHMM_HIP(hipStreamSynchronize(c->stream));
        return BHMM_OK;
    }
    // large results into a pageable buffer: pin it for the transfer (a pageable destination goes
    // through the runtime's staging buffers at a fraction of the link rate); a buffer the caller
    // pinned already (hipHostMalloc / hipHostRegister, torch pin_memory) is used as it is
    const size_t pbytes = (size_t)c->total * (out_fmt == 0 ? sizeof(int32_t) : sizeof(uint8_t));
    hipPointerAttribute_t attr;
    bool caller_pinned = hipPointerGetAttributes(&attr, paths_out) == hipSuccess &&
                         attr.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    const bool pinned = !caller_pinned && pbytes >= ((size_t)8 << 20) &&
                        hipHostRegister(paths_out, pbytes, hipHostRegisterDefault) == hipSuccess;
    if (!pinned)
        (void)hipGetLastError();
    hipError_t ce = hipMemcpyAsync(paths_out, out_fmt == 0 ? static_cast<const void *>(path)
                                                           : static_cast<const void *>(path8),
                                   pbytes, hipMemcpyDeviceToHost, c->stream);
    if (ce == hipSuccess)
        ce = hipStreamSynchronize(c->stream);
    if (pinned)
        (void)hipHostUnregister(paths_out);
    BHMM_HIP(ce);
    return BHMM_OK;
}

int wide_sample_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                    const double *par1, const double *u, uint64_t seed, int32_t *paths,
                    int64_t *counts, int64_t *n0, double *emis, double *stats_dev)
{
    // alpha (row-major, any scale per row) in d_alpha_rm; the repeat of a call in which a watched draw did not
    // stand (draw_verify.hpp) takes the serial recursion
    int rc = c->draw_force_exact ? wide_forward(c, A, pi, par0, par1) : wide_forward_draw(c, A, pi, par0, par1);
    if (rc)
        return rc;
    if (c->draw_force_exact) {
        c->draw_fwd_segmented = false;
        c->draw_alpha_dev = 0.0;
    }
    WideModel m;
    if ((rc = wide_model_pub(c, c->kind, A, pi, par0, par1, m)))
        return rc;
    const int K = c->K, n = c->n, NP = c->N, GP = 64 / NP;
    // rows of a segmented pass: draws within 64 x the deviation its boundary check measured are recorded
    DrawWatch watch;
    if ((rc = draw_watch_prepare(c, c->draw_fwd_segmented ? 64.0 * std::max(c->draw_alpha_dev, 1e-16) : 0.0, watch, nullptr)))
        return rc;
    const size_t nstat = (size_t)n * n + n;
    const size_t esz = c->kind == EMIT_GAUSS ? 3 * (size_t)n
                                             : (c->kind == EMIT_DISC ? (size_t)n * c->M : 0);
    if ((rc = c->d_scratch2.ensure(((size_t)c->total + 4) * sizeof(int32_t))))
        return rc;
    int32_t *path = reinterpret_cast<int32_t *>(c->d_scratch2.p);
    int *status = reinterpret_cast<int *>(path + c->total);
    const size_t dbl = nstat + (size_t)K * esz + esz + (u ? (size_t)c->total : 0) + 8;
    if ((rc = c->d_scratch.ensure(dbl * sizeof(double))))
        return rc;
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(c->d_scratch.p);
    double *epart = reinterpret_cast<double *>(c->d_scratch.p) + nstat;
    double *ered = epart + (size_t)K * esz;
    double *udev = nullptr;
    if (u) {
        udev = ered + esz;
        BHMM_HIP(hipMemcpyAsync(udev, u, (size_t)c->total * sizeof(double), hipMemcpyHostToDevice,
                                c->stream));
    }
    BHMM_HIP(hipMemsetAsync(cnt, 0, nstat * sizeof(unsigned long long), c->stream));
    BHMM_HIP(hipMemsetAsync(status, 0, sizeof(int), c->stream));
    if (esz)
        BHMM_HIP(hipMemsetAsync(ered, 0, esz * sizeof(double), c->stream));
    const dim3 grid((K + GP - 1) / GP), blk(64);
    const int64_t *off = c->d_offsets.p;
    auto launch_serial = [&]() {
        if (NP == 16)
            hipLaunchKernelGGL((k_wide_sample_path<16>), grid, blk, 0, c->stream, m, off, K,
                               (const double *)c->d_alpha_rm.p, (const double *)udev, seed, path, status,
                               (const int64_t *)c->d_soff.p);
        else if (NP == 32)
            hipLaunchKernelGGL((k_wide_sample_path<32>), grid, blk, 0, c->stream, m, off, K,
                               (const double *)c->d_alpha_rm.p, (const double *)udev, seed, path, status,
                               (const int64_t *)c->d_soff.p);
        else
            hipLaunchKernelGGL((k_wide_sample_path<64>), grid, blk, 0, c->stream, m, off, K,
                               (const double *)c->d_alpha_rm.p, (const double *)udev, seed, path, status,
                               (const int64_t *)c->d_soff.p);
    };
    // parallel over time segments where that fills more of the device than the trajectories do
    // (k_wide_sample_seg): the draws are coupled through the per-step uniforms, segments that did not
    // continue their successor's state are drawn again until none is left
    c->smp_segmented = false;
    bool seg_done = false;
    if (c->spec_enabled) {
        const int64_t want = (int64_t)c->smp_seg_per_simd * c->num_simd * GP;
        const int64_t seglen = std::max<int64_t>((c->total + want - 1) / want, 64);
        Segs sg;
        if ((rc = wide_path_plan(c, 1, seglen, sg)))
            return rc;
        if (sg.nseg > K) {
            if (c->smp_W <= 0)
                c->smp_W = 64;
            sg.W = c->smp_W;
            if ((rc = c->d_sentry.ensure((size_t)sg.nseg)) || (rc = c->d_sexit.ensure((size_t)sg.nseg)) ||
                (rc = c->d_vflag.ensure((size_t)sg.nseg)) || (rc = c->d_specres.ensure(4)))
                return rc;
            if (!c->h_specres)
                BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_specres), 4 * sizeof(unsigned int),
                                       hipHostMallocDefault));
            const dim3 sgrid((sg.nseg + GP * WVS_WPB - 1) / (GP * WVS_WPB)), sblk(64 * WVS_WPB);
#define BHMM_WSS(NPV, FIXV)                                                                              \
    hipLaunchKernelGGL((k_wide_sample_seg<NPV, FIXV>), sgrid, sblk, 0, c->stream, m, off, sg,           \
                       (const double *)c->d_alpha_rm.p, (const double *)udev, seed, path, status,       \
                       (const int64_t *)c->d_soff.p, c->d_sentry.p, c->d_sexit.p,                       \
                       (const uint8_t *)c->d_vflag.p, watch)
#define BHMM_WSS_NP(FIXV)          \
    do {                           \
        if (NP == 16)              \
            BHMM_WSS(16, FIXV);    \
        else if (NP == 32)         \
            BHMM_WSS(32, FIXV);    \
        else                       \
            BHMM_WSS(64, FIXV);    \
    } while (0)
            const int max_rounds = 16;
            int round = 0;
            for (; round <= max_rounds; ++round) {
                BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                lds_poison(c->stream);
                if (round == 0)
                    BHMM_WSS_NP(false);
                else
                    BHMM_WSS_NP(true);
                hipLaunchKernelGGL(k_wide_smp_check, dim3((sg.nseg + 255) / 256), dim3(256), 0, c->stream, sg,
                                   c->d_sentry.p, (const int32_t *)c->d_sexit.p, c->d_vflag.p, c->d_specres.p);
                BHMM_HIP(hipGetLastError());
                BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 4 * sizeof(unsigned int),
                                        hipMemcpyDeviceToHost, c->stream));
                BHMM_HIP(hipMemcpyAsync(&c->h_specres[0], status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                BHMM_HIP(hipStreamSynchronize(c->stream));
                if (round == 0)
                    c->smp_seg_mismatch = (int)c->h_specres[3];
                if (c->h_specres[3] == 0)
                    break;
            }
#undef BHMM_WSS_NP
#undef BHMM_WSS
            c->smp_seg_rounds = round;
            // (a draw that found no state may belong to a segment that was drawn again afterwards:
            // the serial kernel decides such a call)
            seg_done = c->h_specres[3] == 0 && c->h_specres[0] == 0;
            if (!seg_done)
                BHMM_HIP(hipMemsetAsync(status, 0, sizeof(int), c->stream));
            // more than a tenth of the segments left to the fix-up: a longer warm-up next time
            if ((int64_t)c->smp_seg_mismatch * 10 > sg.nseg && c->smp_W < 4096)
                c->smp_W *= 2;
            c->smp_segmented = seg_done;
        }
    }
    if (!seg_done) {
        if (c->draw_fwd_segmented) { // (the serial draw has no watch: it reads rows of the serial recursion)
            if ((rc = wide_forward(c, A, pi, par0, par1)))
                return rc;
            c->draw_fwd_segmented = false;
            c->draw_alpha_dev = 0.0;
        }
        launch_serial();
    } else if (watch.count) {
        // watched draws are decided again on the serial recursion over a long window; if one does not stand, the
        // whole call again on the rows of the serial recursion
        unsigned int nwatched = 0;
        BHMM_HIP(hipMemcpyAsync(&nwatched, watch.count, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        BHMM_HIP(hipStreamSynchronize(c->stream));
        if (nwatched) {
            bool ok = false;
            if ((rc = draw_verify_run(c, A, pi, par0, par1, nwatched, watch.tol, 8 * (int64_t)std::max(c->spec_W, 64), &ok)))
                return rc;
            if (!ok) {
                const unsigned int ev = c->draw_events, ck = c->draw_checked;
                c->draw_force_exact = true;
                rc = wide_sample_run(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, stats_dev);
                c->draw_force_exact = false;
                c->draw_events = ev;
                c->draw_checked = ck;
                c->draw_redone = 1;
                return rc;
            }
        }
    }
    BHMM_HIP(hipGetLastError());
    if (counts || n0 || emis || stats_dev) {
        // the per-trajectory emission table in LDS, or -- alphabets too large for it -- in epart itself
        const bool gtab = c->kind == EMIT_DISC &&
                          esz * sizeof(double) + nstat * sizeof(unsigned int) > (size_t)150 * 1024;
        const size_t sm = (gtab ? 0 : esz * sizeof(double)) + nstat * sizeof(unsigned int);
        if (gtab)
            BHMM_HIP(hipMemsetAsync(epart, 0, (size_t)K * esz * sizeof(double), c->stream));
        const void *obs = c->d_obs_rm.p;
        if (c->kind == EMIT_GAUSS)
            hipLaunchKernelGGL((k_wide_path_stats<EMIT_GAUSS>), dim3(K), dim3(256), sm, c->stream, m,
                               off, obs, (const int32_t *)path, cnt, epart, 0);
        else if (c->kind == EMIT_DISC) {
            if (sm > 64 * 1024)
                BHMM_HIP(hipFuncSetAttribute((const void *)(k_wide_path_stats<EMIT_DISC>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
            hipLaunchKernelGGL((k_wide_path_stats<EMIT_DISC>), dim3(K), dim3(256), sm, c->stream, m,
                               off, obs, (const int32_t *)path, cnt, epart, gtab ? 1 : 0);
        } else
            hipLaunchKernelGGL((k_wide_path_stats<EMIT_EXPL>), dim3(K), dim3(256), sm, c->stream, m,
                               off, obs, (const int32_t *)path, cnt, epart, 0);
        BHMM_HIP(hipGetLastError());
        if (esz) {
            hipLaunchKernelGGL(k_add_partials, dim3((unsigned)esz), dim3(64), 0,
                               c->stream, (const double *)epart, K, (int)esz, ered);
            BHMM_HIP(hipGetLastError());
        }
    }
    std::vector<unsigned long long> hc(nstat);
    std::vector<double> he(esz);
    int hstatus = 0;
    if (stats_dev) {
        hipLaunchKernelGGL(k_pack_path_stats, dim3(16), dim3(256), 0, c->stream,
                           (const unsigned long long *)cnt, (const double *)ered, n, n, c->M, c->kind,
                           0, stats_dev);
        BHMM_HIP(hipGetLastError());
    } else {
        BHMM_HIP(hipMemcpyAsync(hc.data(), cnt, nstat * sizeof(unsigned long long),
                                hipMemcpyDeviceToHost, c->stream));
        if (esz)
            BHMM_HIP(hipMemcpyAsync(he.data(), ered, esz * sizeof(double), hipMemcpyDeviceToHost,
                                    c->stream));
    }
    BHMM_HIP(hipMemcpyAsync(&hstatus, status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (paths)
        BHMM_HIP(hipMemcpyAsync(paths, path, (size_t)c->total * sizeof(int32_t),
                                hipMemcpyDeviceToHost, c->stream));
    BHMM_HIP(hipStreamSynchronize(c->stream));
    if (hstatus) {
        set_error("random choice found no state: alpha/A not normalisable (_hidden.c:299-304)");
        return hstatus;
    }
    if (stats_dev)
        return BHMM_OK;
    if (counts)
        for (size_t e = 0; e < (size_t)n * n; ++e)
            counts[e] = (int64_t)hc[e];
    if (n0)
        for (int i = 0; i < n; ++i)
            n0[i] = (int64_t)hc[(size_t)n * n + i];
    if (emis && esz)
        memcpy(emis, he.data(), esz * sizeof(double)); // gaussian [3][n]; discrete [n][M]
    return BHMM_OK;
}

struct TmpCtx {
    bhmm_ctx *c = nullptr;
    ~TmpCtx() { bhmm_ctx_destroy(c); }
};

int cur_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        d = 0;
    }
    return d;
}

} // namespace

int wide_path_plan_pub(bhmm_ctx *c, int which, int64_t seglen, Segs &sg) { return wide_path_plan(c, which, seglen, sg); }

} // namespace bhmm

using namespace bhmm;

extern "C" {

int bhmm_viterbi_batch(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                       const double *par1, int32_t *paths)
{
    if (c)
        lds_poison(c->stream); // (debugging aid, BHMM_AMD_POISON=1 only)
    if (!c || c->kind < 0)
        return invalid_arg("no observations loaded");
    if (!A || !pi || !paths)
        return invalid_arg("NULL argument");
    if (c->kind == BHMM_EMIT_GAUSSIAN && (!par0 || !par1))
        return invalid_arg("gaussian emissions need means and sigmas");
    if (c->kind == BHMM_EMIT_DISCRETE && !par0)
        return invalid_arg("discrete emissions need B");
    BHMM_HIP(hipSetDevice(c->device));
    if (c->gen)
        return gen_viterbi_run(c, A, pi, par0, par1, paths, 0);
    // all state counts up to 64 use the LDS-exchange kernels (k_wide_viterbi_*)
    return wide_viterbi_run(c, A, pi, par0, par1, paths, 0);
}

int bhmm_viterbi_batch_u8(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                          const double *par1, uint8_t *paths, int paths_on_device)
{
    if (c)
        lds_poison(c->stream); // (debugging aid, BHMM_AMD_POISON=1 only)
    if (!c || c->kind < 0)
        return invalid_arg("no observations loaded");
    if (!A || !pi || !paths)
        return invalid_arg("NULL argument");
    if (c->kind == BHMM_EMIT_GAUSSIAN && (!par0 || !par1))
        return invalid_arg("gaussian emissions need means and sigmas");
    if (c->kind == BHMM_EMIT_DISCRETE && !par0)
        return invalid_arg("discrete emissions need B");
    BHMM_HIP(hipSetDevice(c->device));
    if (c->gen)
        return gen_viterbi_run(c, A, pi, par0, par1, paths, paths_on_device ? 2 : 1);
    return wide_viterbi_run(c, A, pi, par0, par1, paths, paths_on_device ? 2 : 1);
}

int bhmm_sample_paths(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                      const double *par1, const double *u, uint64_t seed, int32_t *paths,
                      int64_t *counts, int64_t *n0, double *emis)
{
    if (c)
        lds_poison(c->stream); // (debugging aid, BHMM_AMD_POISON=1 only)
    if (!c || c->kind < 0)
        return invalid_arg("no observations loaded");
    if (!A || !pi)
        return invalid_arg("NULL argument");
    BHMM_HIP(hipSetDevice(c->device));
    if (c->gen)
        return gen_sample_run(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, nullptr);
    if (c->wide)
        return wide_sample_run(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, nullptr);
    switch (c->N) {
    case 2:
        return sample_run<2>(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, nullptr);
    case 4:
        return sample_run<4>(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, nullptr);
    default:
        return sample_run<8>(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, nullptr);
    }
}

int bhmm_ctx_set_stream_offsets(bhmm_ctx *c, const int64_t *soff)
{
    if (!c || c->kind < 0)
        return invalid_arg("no observations loaded");
    BHMM_HIP(hipSetDevice(c->device));
    if (!soff) {
        c->d_soff.release();
        return BHMM_OK;
    }
    int rc = c->d_soff.ensure((size_t)std::max(c->K, 1));
    if (rc)
        return rc;
    BHMM_HIP(hipMemcpy(c->d_soff.p, soff, (size_t)c->K * sizeof(int64_t), hipMemcpyHostToDevice));
    return BHMM_OK;
}

int bhmm_ctx_path_stats_size(const bhmm_ctx *c)
{
    if (!c || c->kind < 0)
        return 0;
    const int n = c->n;
    return n * n + n + (c->kind == EMIT_GAUSS ? 3 * n : (c->kind == EMIT_DISC ? n * c->M : 0));
}

int bhmm_sample_paths_dev(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                          const double *par1, const double *u, uint64_t seed, int32_t *paths,
                          double *stats_dev)
{
    if (c)
        lds_poison(c->stream); // (debugging aid, BHMM_AMD_POISON=1 only)
    if (!c || c->kind < 0)
        return invalid_arg("no observations loaded");
    if (!A || !pi || !stats_dev)
        return invalid_arg("NULL argument");
    BHMM_HIP(hipSetDevice(c->device));
    if (c->gen)
        return gen_sample_run(c, A, pi, par0, par1, u, seed, paths, nullptr, nullptr, nullptr,
                              stats_dev);
    if (c->wide)
        return wide_sample_run(c, A, pi, par0, par1, u, seed, paths, nullptr, nullptr, nullptr,
                               stats_dev);
    switch (c->N) {
    case 2:
        return sample_run<2>(c, A, pi, par0, par1, u, seed, paths, nullptr, nullptr, nullptr, stats_dev);
    case 4:
        return sample_run<4>(c, A, pi, par0, par1, u, seed, paths, nullptr, nullptr, nullptr, stats_dev);
    default:
        return sample_run<8>(c, A, pi, par0, par1, u, seed, paths, nullptr, nullptr, nullptr, stats_dev);
    }
}

int bhmm_viterbi(int32_t *path, const double *A, const double *pobs, const double *pi, int N,
                 int64_t T)
{
    if (!path || !A || !pobs || !pi || N < 1 || T < 1)
        return invalid_arg("NULL argument or empty problem");
    TmpCtx t;
    int rc = bhmm_ctx_create(&t.c, cur_device(), nullptr);
    if (rc)
        return rc;
    const int64_t off[2] = {0, T};
    rc = bhmm_ctx_set_observations(t.c, BHMM_EMIT_EXPLICIT, pobs, off, 1, N, 0, 0, 0);
    if (rc)
        return rc;
    return bhmm_viterbi_batch(t.c, A, pi, nullptr, nullptr, path);
}

int bhmm_sample_path(int32_t *path, const double *alpha, const double *A, const double *u, int N,
                     int64_t T)
{
    if (!path || !alpha || !A || !u || N < 1 || T < 1)
        return invalid_arg("NULL argument or empty problem");
    if (N > 64)
        return gen_sample_path(path, alpha, A, u, N, T);
    Tmp tmp;
    double *d_alpha, *d_u;
    int64_t *d_off;
    int32_t *d_path;
    int *d_status;
    int rc;
    if ((rc = tmp.alloc(&d_alpha, (size_t)T * N)) || (rc = tmp.alloc(&d_u, (size_t)T)) ||
        (rc = tmp.alloc(&d_off, 2)) || (rc = tmp.alloc(&d_path, (size_t)T)) ||
        (rc = tmp.alloc(&d_status, 1)))
        return rc;
    const int64_t off[2] = {0, T};
    BHMM_HIP(hipMemcpy(d_alpha, alpha, (size_t)T * N * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(d_u, u, (size_t)T * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(d_off, off, sizeof(off), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemset(d_status, 0, sizeof(int)));
    if (N > 8) {
        double *d_A;
        if ((rc = tmp.alloc(&d_A, (size_t)N * N)))
            return rc;
        BHMM_HIP(hipMemcpy(d_A, A, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice));
        WideModel m;
        memset(&m, 0, sizeof(m));
        m.A = d_A;
        m.n = N;
        if (N <= 16)
            hipLaunchKernelGGL((k_wide_sample_path<16>), dim3(1), dim3(64), 0, 0, m,
                               (const int64_t *)d_off, 1, (const double *)d_alpha,
                               (const double *)d_u, (uint64_t)0, d_path, d_status);
        else if (N <= 32)
            hipLaunchKernelGGL((k_wide_sample_path<32>), dim3(1), dim3(64), 0, 0, m,
                               (const int64_t *)d_off, 1, (const double *)d_alpha,
                               (const double *)d_u, (uint64_t)0, d_path, d_status);
        else
            hipLaunchKernelGGL((k_wide_sample_path<64>), dim3(1), dim3(64), 0, 0, m,
                               (const int64_t *)d_off, 1, (const double *)d_alpha,
                               (const double *)d_u, (uint64_t)0, d_path, d_status);
        BHMM_HIP(hipGetLastError());
        int stw = 0;
        BHMM_HIP(hipMemcpy(path, d_path, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost));
        BHMM_HIP(hipMemcpy(&stw, d_status, sizeof(int), hipMemcpyDeviceToHost));
        if (stw) {
            set_error("random choice found no state: p not normalised (_hidden.c:299-304)");
            return stw;
        }
        return BHMM_OK;
    }
    const int NP = pad_states_pub(N);
    std::vector<double> pi(N, 0.0);
#define BHMM_SAMPLE_CASE(NN)                                                                    \
    {                                                                                           \
        Model<NN> m;                                                                            \
        fill_model_pub<NN>(m, N, EMIT_EXPL, 0, A, pi.data(), nullptr, nullptr);                 \
        hipLaunchKernelGGL((k_sample_path<NN>), dim3(1), dim3(64), 0, 0, m,                     \
                           (const int64_t *)d_off, 1, (const double *)d_alpha,                  \
                           (const double *)d_u, (uint64_t)0, d_path, d_status);                 \
    }
    if (NP == 2)
        BHMM_SAMPLE_CASE(2)
    else if (NP == 4)
        BHMM_SAMPLE_CASE(4)
    else
        BHMM_SAMPLE_CASE(8)
    BHMM_HIP(hipGetLastError());
    int st = 0;
    BHMM_HIP(hipMemcpy(path, d_path, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost));
    BHMM_HIP(hipMemcpy(&st, d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (st) {
        set_error("random choice found no state: p not normalised (_hidden.c:299-304)");
        return st;
    }
    return BHMM_OK;
}

int bhmm_libc_uniforms(double *u, int64_t T, int seed)
{
    if (!u || T < 0)
        return invalid_arg("NULL argument");
    if (seed >= 0)
        srand((unsigned)seed); // _hidden.c:321-327
    for (int64_t t = T - 1; t >= 0; --t)
        u[t] = (double)rand() / ((double)RAND_MAX + 1.0); // _hidden.c:285-287
    return BHMM_OK;
}

int bhmm_state_probabilities(double *gamma, const double *alpha, const double *beta, int N,
                             int64_t T)
{
    if (!gamma || !alpha || !beta || N < 1 || T < 1)
        return invalid_arg("NULL argument or empty problem");
    Tmp tmp;
    double *da, *db, *dg;
    int rc;
    const size_t cnt = (size_t)T * N;
    if ((rc = tmp.alloc(&da, cnt)) || (rc = tmp.alloc(&db, cnt)) || (rc = tmp.alloc(&dg, cnt)))
        return rc;
    BHMM_HIP(hipMemcpy(da, alpha, cnt * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(db, beta, cnt * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gamma_rows, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, 0,
                       (const double *)da, (const double *)db, dg, N, T);
    BHMM_HIP(hipGetLastError());
    BHMM_HIP(hipMemcpy(gamma, dg, cnt * sizeof(double), hipMemcpyDeviceToHost));
    return BHMM_OK;
}

int bhmm_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                           const double *beta, int N, int64_t T)
{
    if (!C || !A || !pobs || !alpha || !beta || N < 1 || T < 1)
        return invalid_arg("NULL argument or empty problem");
    if (N > 64)
        return gen_transition_counts(C, A, pobs, alpha, beta, N, T);
    if (N > 8)
        return wide_transition_counts(C, A, pobs, alpha, beta, N, T);
    Tmp tmp;
    double *dA, *dp, *da, *db, *dpart, *dC;
    int rc;
    const size_t cnt = (size_t)T * N;
    const int nblk = (int)std::min<int64_t>(1024, (T + 255) / 256);
    const int NP = pad_states_pub(N);
    if ((rc = tmp.alloc(&dA, (size_t)N * N)) || (rc = tmp.alloc(&dp, cnt)) ||
        (rc = tmp.alloc(&da, cnt)) || (rc = tmp.alloc(&db, cnt)) ||
        (rc = tmp.alloc(&dpart, (size_t)nblk * NP * NP)) || (rc = tmp.alloc(&dC, (size_t)N * N)))
        return rc;
    BHMM_HIP(hipMemcpy(dA, A, (size_t)N * N * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(dp, pobs, cnt * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(da, alpha, cnt * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(db, beta, cnt * sizeof(double), hipMemcpyHostToDevice));
#define BHMM_XI_CASE(NN)                                                                        \
    {                                                                                           \
        hipLaunchKernelGGL((k_xi_rows<NN>), dim3(nblk), dim3(256), 0, 0, (const double *)dA,    \
                           (const double *)dp, (const double *)da, (const double *)db, N, T,    \
                           dpart);                                                              \
        hipLaunchKernelGGL((k_sum_partials<NN>), dim3(1), dim3(64), 0, 0,                       \
                           (const double *)dpart, nblk, N, dC);                                 \
    }
    if (NP == 2)
        BHMM_XI_CASE(2)
    else if (NP == 4)
        BHMM_XI_CASE(4)
    else
        BHMM_XI_CASE(8)
    BHMM_HIP(hipGetLastError());
    BHMM_HIP(hipMemcpy(C, dC, (size_t)N * N * sizeof(double), hipMemcpyDeviceToHost));
    return BHMM_OK;
}

int bhmm_pobs_gaussian(double *pobs, const double *obs, const double *mu, const double *sigma,
                       int N, int64_t T, int ignore_outliers)
{
    if (!pobs || !obs || !mu || !sigma || N < 1 || T < 1)
        return invalid_arg("NULL argument or empty problem");
    Tmp tmp;
    double *dobs, *dmu, *dsig, *dp;
    int rc;
    if ((rc = tmp.alloc(&dobs, (size_t)T)) || (rc = tmp.alloc(&dmu, (size_t)N)) ||
        (rc = tmp.alloc(&dsig, (size_t)N)) || (rc = tmp.alloc(&dp, (size_t)T * N)))
        return rc;
    BHMM_HIP(hipMemcpy(dobs, obs, (size_t)T * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(dmu, mu, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(dsig, sigma, (size_t)N * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_pobs_gaussian, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, 0,
                       (const double *)dobs, (const double *)dmu, (const double *)dsig, dp, N, T,
                       ignore_outliers);
    BHMM_HIP(hipGetLastError());
    BHMM_HIP(hipMemcpy(pobs, dp, (size_t)T * N * sizeof(double), hipMemcpyDeviceToHost));
    return BHMM_OK;
}

int bhmm_update_pout(double *pout, const int32_t *obs, const double *weights, int64_t T, int N,
                     int M)
{
    if (!pout || !obs || !weights || N < 1 || M < 1 || T < 1)
        return invalid_arg("NULL argument or empty problem");
    if ((size_t)N * M * sizeof(double) > 64 * 1024)
        return invalid_arg("N*M too large for the LDS histogram");
    Tmp tmp;
    int32_t *dobs;
    double *dw, *dpart, *dout;
    int rc;
    const int nblk = (int)std::min<int64_t>(512, (T + 255) / 256);
    if ((rc = tmp.alloc(&dobs, (size_t)T)) || (rc = tmp.alloc(&dw, (size_t)T * N)) ||
        (rc = tmp.alloc(&dpart, (size_t)nblk * N * M)) || (rc = tmp.alloc(&dout, (size_t)N * M)))
        return rc;
    BHMM_HIP(hipMemcpy(dobs, obs, (size_t)T * sizeof(int32_t), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(dw, weights, (size_t)T * N * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(dout, pout, (size_t)N * M * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_update_pout, dim3(nblk), dim3(256), (size_t)N * M * sizeof(double), 0,
                       (const int32_t *)dobs, (const double *)dw, T, N, M, dpart);
    hipLaunchKernelGGL(k_add_partials, dim3(N * M), dim3(64), 0, 0,
                       (const double *)dpart, nblk, N * M, dout);
    BHMM_HIP(hipGetLastError());
    BHMM_HIP(hipMemcpy(pout, dout, (size_t)N * M * sizeof(double), hipMemcpyDeviceToHost));
    return BHMM_OK;
}

} // extern "C"
