// gen_api.hip -- host side of the any-N family (gen_kernels.hpp): contexts with more than 64 hidden
// states.  bhmm/hidden/impl_c/_hidden.c:16-378 has no limit on N; neither has the C ABI.  Everything
// is trajectory-major and materialised like the reference does (pobs, alpha, beta / W rows); one
// workgroup per trajectory, no time decomposition.  Compiled with -ffp-contract=off.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "host_common.hpp"
#include "gen_kernels.hpp"

namespace bhmm {
int invalid_arg(const std::string &msg);
int wide_model_pub(bhmm_ctx *c, int kind, const double *A, const double *pi, const double *par0,
                   const double *par1, WideModel &m);
// tile_gen.hip: up to 128 states on the row-batched matrix-core kernels
bool tile_gen_capable(const bhmm_ctx *c);
int tile_gen_alloc(bhmm_ctx *c);
int tile_gen_estep(bhmm_ctx *c, const WideModel &m, double *stats_dev, int flags, bool *done);
int tile_gen_forward_draw(bhmm_ctx *c, const WideModel &m, bool *done);
int wide_path_plan_pub(bhmm_ctx *c, int which, int64_t seglen, Segs &sg);
int draw_watch_prepare(bhmm_ctx *c, double tol, DrawWatch &w, unsigned int *count_slot);
int draw_verify_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0, const double *par1,
                    unsigned int count, double thr, int64_t Wlong, bool *ok);

namespace {

size_t gen_smem(int n, int vectors, int extra) { return ((size_t)vectors * n + extra) * sizeof(double); }
// the workgroup's own copy of the transition matrix behind the vectors, where it fits (gen_kernels.hpp, ALDS)
size_t gen_a_bytes(int n) { return (size_t)n * n * sizeof(double); }
bool gen_a_in_lds(int n, size_t vectors_bytes) { return vectors_bytes + gen_a_bytes(n) <= GEN_LDS_LIMIT; }

template <typename F>
int gen_set_smem(F *fn, size_t sm)
{
    if (sm > 64 * 1024)
        BHMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    return BHMM_OK;
}

// emission rows of all steps (the reference materialises them too, maximum_likelihood.py:249-252)
int gen_pobs(bhmm_ctx *c, const WideModel &m, const double **pobs)
{
    if (c->kind == EMIT_EXPL) {
        *pobs = reinterpret_cast<const double *>(c->d_obs_rm.p);
        return BHMM_OK;
    }
    int rc = c->d_gpobs.ensure((size_t)c->total * c->n);
    if (rc)
        return rc;
    const dim3 pg((unsigned)((c->total + 255) / 256)), pb(256);
    if (c->kind == EMIT_GAUSS)
        hipLaunchKernelGGL((k_pobs_all<EMIT_GAUSS>), pg, pb, 0, c->stream, m, (const void *)c->d_obs_rm.p,
                           c->total, c->d_gpobs.p);
    else
        hipLaunchKernelGGL((k_pobs_all<EMIT_DISC>), pg, pb, 0, c->stream, m, (const void *)c->d_obs_rm.p,
                           c->total, c->d_gpobs.p);
    BHMM_HIP(hipGetLastError());
    *pobs = c->d_gpobs.p;
    return BHMM_OK;
}

int gen_transposed(bhmm_ctx *c, const WideModel &m)
{
    const int n = c->n;
    int rc = c->d_gAt.ensure((size_t)n * n);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_gen_transpose, dim3((unsigned)(((size_t)n * n + 255) / 256)), dim3(256), 0,
                       c->stream, m.A, n, c->d_gAt.p);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

int gen_launch_forward(bhmm_ctx *c, const WideModel &m, const double *pobs)
{
    size_t sm = gen_smem(c->n, 2, 2);
    const bool alds = gen_a_in_lds(c->n, sm);
    int rc;
    if (alds) {
        sm += gen_a_bytes(c->n);
        if ((rc = gen_set_smem(k_gen_forward<true>, sm)))
            return rc;
        hipLaunchKernelGGL(k_gen_forward<true>, dim3(c->K), dim3(GEN_TPB), sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, c->K, pobs, c->d_alpha_rm.p, c->d_logLk.p);
    } else {
        if ((rc = gen_set_smem(k_gen_forward<false>, sm)))
            return rc;
        hipLaunchKernelGGL(k_gen_forward<false>, dim3(c->K), dim3(GEN_TPB), sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, c->K, pobs, c->d_alpha_rm.p, c->d_logLk.p);
    }
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

// xi GEMM + reduction geometry
int gen_nsplit(const bhmm_ctx *c)
{
    const int tiles = (c->n + 31) / 32;
    const int64_t want = 4096 / ((int64_t)tiles * tiles) + 1;
    return (int)std::max<int64_t>(1, std::min<int64_t>({want, (c->total + 63) / 64, (int64_t)256}));
}

} // namespace

int gen_alloc(bhmm_ctx *c)
{
    const int n = c->n;
    const size_t rows = (size_t)c->total * n;
    const int K = std::max(c->K, 1);
    const size_t S = 1 + (size_t)n + (size_t)n * n + n + std::max<size_t>(2 * (size_t)n, (size_t)n * c->M);
    int rc;
    if ((rc = c->d_alpha_rm.ensure(rows)) || (rc = c->d_logLk.ensure(K)) ||
        (rc = c->d_gamma0.ensure((size_t)K * n)) || (rc = c->d_stats.ensure(S)))
        return rc;
    if (tile_gen_capable(c) && (rc = tile_gen_alloc(c)))
        return rc;
    return BHMM_OK;
}

int gen_forward(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                const double *par1)
{
    WideModel m;
    int rc = wide_model_pub(c, c->kind, A, pi, par0, par1, m);
    if (rc)
        return rc;
    const double *pobs = nullptr;
    if ((rc = gen_pobs(c, m, &pobs)))
        return rc;
    return gen_launch_forward(c, m, pobs);
}

// beta rows of _hidden.c:69-110 (explicit pobs) into d_alpha_rm
int gen_backward(bhmm_ctx *c, const double *A)
{
    WideModel m;
    std::vector<double> pi(c->n, 1.0 / c->n);
    int rc = wide_model_pub(c, EMIT_EXPL, A, pi.data(), nullptr, nullptr, m);
    if (rc || (rc = gen_transposed(c, m)))
        return rc;
    const double *pobs = reinterpret_cast<const double *>(c->d_obs_rm.p);
    size_t sm = gen_smem(c->n, 3, 1 + GEN_TPB);
    const bool alds = gen_a_in_lds(c->n, sm);
    sm += alds ? gen_a_bytes(c->n) : 0;
#define BHMM_GEN_BETA(ALDSV)                                                                         \
    do {                                                                                             \
        if ((rc = gen_set_smem(k_gen_backward<EMIT_EXPL, false, ALDSV>, sm)))                        \
            return rc;                                                                               \
        hipLaunchKernelGGL((k_gen_backward<EMIT_EXPL, false, ALDSV>), dim3(c->K), dim3(GEN_TPB), sm, \
                           c->stream, m, (const double *)c->d_gAt.p, (const int64_t *)c->d_offsets.p, \
                           c->K, pobs, (const void *)nullptr, (const double *)nullptr,               \
                           c->d_alpha_rm.p, (double *)nullptr, (double *)nullptr, (double *)nullptr, \
                           (double *)nullptr, (double *)nullptr);                                    \
    } while (0)
    if (alds)
        BHMM_GEN_BETA(true);
    else
        BHMM_GEN_BETA(false);
#undef BHMM_GEN_BETA
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

int gen_estep(bhmm_ctx *c, const double *A, const double *pi, const double *par0, const double *par1,
              double *stats_dev, int flags)
{
    WideModel m;
    int rc = wide_model_pub(c, c->kind, A, pi, par0, par1, m);
    if (rc)
        return rc;
    {
        // up to 128 states: the row-batched matrix-core kernels (tolerance-compared statistics); the
        // order-faithful kernels below where those do not apply or left their range
        bool done = false;
        if ((rc = tile_gen_estep(c, m, stats_dev, flags, &done)) || done)
            return rc;
    }
    const int n = c->n, K = c->K;
    const bool sg = (flags & BHMM_FLAG_STORE_GAMMA) != 0;
    const int nsplit = gen_nsplit(c);
    if ((rc = gen_transposed(c, m)) || (rc = c->d_gW.ensure((size_t)c->total * n)) ||
        (rc = c->d_gpart.ensure((size_t)std::max(K, 1) * 3 * n)) ||
        (rc = c->d_gxipart.ensure((size_t)nsplit * n * n)) ||
        (c->kind == EMIT_DISC && (rc = c->d_gsym.ensure((size_t)n * c->M))) ||
        (sg && (rc = c->d_gamma_ci.ensure((size_t)c->total * n))))
        return rc;
    BHMM_HIP(hipEventRecord(c->ev[0], c->stream));
    const double *pobs = nullptr;
    if ((rc = gen_pobs(c, m, &pobs)))
        return rc;
    BHMM_HIP(hipEventRecord(c->ev[1], c->stream));
    if ((rc = gen_launch_forward(c, m, pobs)))
        return rc;
    BHMM_HIP(hipEventRecord(c->ev[2], c->stream));
    if (c->kind == EMIT_DISC)
        BHMM_HIP(hipMemsetAsync(c->d_gsym.p, 0, (size_t)n * c->M * sizeof(double), c->stream));
    size_t sm = gen_smem(n, 3, 1 + GEN_TPB);
    const bool alds = gen_a_in_lds(n, sm);
    sm += alds ? gen_a_bytes(n) : 0;
    double *gam = sg ? c->d_gamma_ci.p : nullptr;
#define BHMM_GEN_BWD2(KINDV, ALDSV)                                                                  \
    do {                                                                                             \
        if ((rc = gen_set_smem(k_gen_backward<KINDV, true, ALDSV>, sm)))                             \
            return rc;                                                                               \
        hipLaunchKernelGGL((k_gen_backward<KINDV, true, ALDSV>), dim3(K), dim3(GEN_TPB), sm,         \
                           c->stream, m, (const double *)c->d_gAt.p, (const int64_t *)c->d_offsets.p, \
                           K, pobs, (const void *)c->d_obs_rm.p, (const double *)c->d_alpha_rm.p,    \
                           (double *)nullptr, c->d_gW.p, gam, c->d_gpart.p, c->d_gamma0.p,           \
                           c->d_gsym.p);                                                             \
    } while (0)
#define BHMM_GEN_BWD(KINDV)                                                                          \
    do {                                                                                             \
        if (alds)                                                                                    \
            BHMM_GEN_BWD2(KINDV, true);                                                              \
        else                                                                                         \
            BHMM_GEN_BWD2(KINDV, false);                                                             \
    } while (0)
    if (c->kind == EMIT_GAUSS)
        BHMM_GEN_BWD(EMIT_GAUSS);
    else if (c->kind == EMIT_DISC)
        BHMM_GEN_BWD(EMIT_DISC);
    else
        BHMM_GEN_BWD(EMIT_EXPL);
#undef BHMM_GEN_BWD
#undef BHMM_GEN_BWD2
    BHMM_HIP(hipGetLastError());
    const int tiles = (n + 31) / 32;
    hipLaunchKernelGGL(k_gen_xi_gemm, dim3(tiles * tiles, nsplit), dim3(256), 0, c->stream,
                       (const double *)c->d_alpha_rm.p, (const double *)c->d_gW.p, c->total, n, nsplit,
                       c->d_gxipart.p);
    BHMM_HIP(hipGetLastError());
    BHMM_HIP(hipEventRecord(c->ev[3], c->stream));
    const dim3 fg(256), fb(256);
#define BHMM_GEN_FIN(KINDV)                                                                          \
    hipLaunchKernelGGL((k_gen_finalize<KINDV>), fg, fb, 0, c->stream, m, K, nsplit,                  \
                       (const double *)c->d_gxipart.p, (const double *)c->d_gpart.p,                 \
                       (const double *)c->d_gamma0.p, (const double *)c->d_logLk.p,                  \
                       (const double *)c->d_gsym.p, stats_dev)
    if (c->kind == EMIT_GAUSS)
        BHMM_GEN_FIN(EMIT_GAUSS);
    else if (c->kind == EMIT_DISC)
        BHMM_GEN_FIN(EMIT_DISC);
    else
        BHMM_GEN_FIN(EMIT_EXPL);
#undef BHMM_GEN_FIN
    BHMM_HIP(hipGetLastError());
    BHMM_HIP(hipEventRecord(c->ev[4], c->stream));
    c->ev_pending = true;
    return BHMM_OK;
}

// out_fmt as wide_viterbi_run: 0 int32 host, 1 uint8 host, 2 uint8 device (N <= 256 for the bytes)
int gen_viterbi_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                    const double *par1, void *paths_out, int out_fmt)
{
    const int n = c->n, K = c->K;
    if (out_fmt != 0 && n > 256)
        return invalid_arg("one byte per step holds at most 256 states: use bhmm_viterbi_batch (int32)");
    WideModel m;
    int rc = wide_model_pub(c, c->kind, A, pi, par0, par1, m);
    if (rc)
        return rc;
    c->viterbi_chunked = false;
    c->vit_mended = 0;
    const double *pobs = nullptr;
    if ((rc = gen_pobs(c, m, &pobs)) ||
        (rc = c->d_scratch.ensure((size_t)c->total * n * sizeof(uint16_t))) ||
        (rc = c->d_scratch2.ensure(((size_t)c->total + K) * sizeof(int32_t))))
        return rc;
    uint16_t *ptr = reinterpret_cast<uint16_t *>(c->d_scratch.p);
    int32_t *last = reinterpret_cast<int32_t *>(c->d_scratch2.p);
    int32_t *path = last + K;
    // up to 128 states: over time segments (k_gen_viterbi_seg), accepted only when every segment started
    // from the bit pattern its predecessor computed -- then the back-pointers are the serial run's
    if (n <= 128 && c->spec_enabled && !c->vit_seg_given_up) {
        bool done = false;
        uint8_t *ptr8 = reinterpret_cast<uint8_t *>(c->d_scratch.p);
        const size_t smv = (size_t)(128 * GVS_PITCH + 8 * 128) * sizeof(double);
        if ((rc = gen_set_smem(k_gen_viterbi_seg<false>, smv)) || (rc = gen_set_smem(k_gen_viterbi_seg<true>, smv)) ||
            (rc = c->d_specres.ensure(4)))
            return rc;
        if (!c->h_specres)
            BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_specres), 4 * sizeof(unsigned int),
                                   hipHostMallocDefault));
        // warm-up: the max-product survivors of these larger models meet later than the filter forgets (128
        // states, 128 x 10 000: 236 of 2048 boundaries further than 1e-12 apart after the E-step's 120 steps, 4
        // after 240, none after 480), and a fix-up round costs half a first pass here: four times the E-step's
        // length, and segments as short as one warm-up (round 5; with the path-margin acceptance 25.7 -> 11 ms)
        // -- but never longer than the segments that fill the chip (a slowly forgetting model keeps the E-step's
        // length and its segment count; the rounds then do what the margins cannot)
        const int seg_warmups = c->vit_margin ? 1 : c->vit_seg_warmups;
        const int W0 = std::max(64, c->spec_W > 0 ? (c->spec_W + 7) / 8 * 8 : 128);
        const int64_t fill0 = ((c->total + (int64_t)c->vit_seg_per_simd * c->num_simd - 1) /
                               ((int64_t)c->vit_seg_per_simd * c->num_simd) + 7) / 8 * 8;
        // (round 6: with the mending round the E-step's length is where the margin route starts too -- the few
        // boundaries that need the fourfold length are repaired alone; the length doubles for the next call, up to
        // that fourfold, when a pass could not be mended or needed three or more rounds)
        const int W_cap = (int)std::max<int64_t>(W0, std::min<int64_t>(4 * (int64_t)W0, fill0));
        int W_try = c->vit_W > 0 ? c->vit_W : ((c->vit_margin && !c->vit_mend) ? W_cap : W0);
        Segs sg;
        // the path-margin acceptance of the first pass (k_vit_margin, path_kernels.hpp; see wide_viterbi_run)
        const double vm_tol = 1e-12;
        double *vall = nullptr;
        int64_t maxT = 0;
        const size_t smm = (size_t)n * (n | 1) * sizeof(double); // (odd pitch, k_vit_margin)
        if (c->vit_margin && c->d_gW.ensure((size_t)c->total * n) == BHMM_OK &&
            gen_set_smem(k_vit_margin<int32_t, 2>, smm) == BHMM_OK && gen_set_smem(k_vit_margin<uint8_t, 2>, smm) == BHMM_OK) {
            vall = c->d_gW.p;
            for (int k = 0; k < K; ++k)
                maxT = std::max(maxT, c->offsets[k + 1] - c->offsets[k]);
        } else {
            (void)hipGetLastError();
        }
        c->vit_margin_used = 0;
        c->vit_margin_close = 0;
        const int64_t *off = c->d_offsets.p;
        uint8_t *p8 = out_fmt == 2 ? static_cast<uint8_t *>(paths_out) : reinterpret_cast<uint8_t *>(path);
        // back-trace over the segments: maps, stitch, apply
        auto seg_walks = [&]() -> int {
            int rcw;
            if ((rcw = c->d_vmaps.ensure((size_t)sg.nseg * 128)) || (rcw = c->d_vend.ensure((size_t)sg.nseg)))
                return rcw;
            hipLaunchKernelGGL((k_wide_vit_walk<false, uint8_t, 2>), dim3(sg.nseg), dim3(64), 0, c->stream, off, sg, n,
                               (const uint8_t *)ptr8, c->d_vmaps.p, (const uint8_t *)nullptr, (uint8_t *)nullptr);
            hipLaunchKernelGGL(k_wide_vit_stitch, dim3((K + 63) / 64), dim3(64), 0, c->stream,
                               (const int32_t *)c->pplan[0].traj0.p, K, (const uint8_t *)c->d_vmaps.p, 128,
                               (const int32_t *)last, c->d_vend.p);
            if (out_fmt == 0)
                hipLaunchKernelGGL((k_wide_vit_walk<true, int32_t, 2>), dim3(sg.nseg), dim3(64), 0, c->stream, off, sg, n,
                                   (const uint8_t *)ptr8, (uint8_t *)nullptr, (const uint8_t *)c->d_vend.p, path);
            else
                hipLaunchKernelGGL((k_wide_vit_walk<true, uint8_t, 2>), dim3(sg.nseg), dim3(64), 0, c->stream, off, sg, n,
                                   (const uint8_t *)ptr8, (uint8_t *)nullptr, (const uint8_t *)c->d_vend.p, p8);
            BHMM_HIP(hipGetLastError());
            return BHMM_OK;
        };
        for (int attempt = 0; attempt < 2 && !done; ++attempt) {
            if (attempt > 0)
                W_try *= 2;
            const int64_t want = (int64_t)c->vit_seg_per_simd * c->num_simd;
            const int64_t seglen = std::max<int64_t>((c->total + want - 1) / want, seg_warmups * (int64_t)W_try);
            if ((rc = wide_path_plan_pub(c, 0, seglen, sg)))
                return rc;
            if (sg.nseg <= K)
                break;
            sg.W = W_try;
            if ((rc = c->d_aentry.ensure((size_t)sg.nseg * 128)) || (rc = c->d_aexit.ensure((size_t)sg.nseg * 128)) ||
                (rc = c->d_vckpt.ensure(((size_t)(c->total >> 6) + 1) * 128)) || (rc = c->d_vflag.ensure(2 * (size_t)sg.nseg)))
                return rc;
            const dim3 sgrid((sg.nseg + 7) / 8), sblk(512);
            int round = 0;
            bool margin_accepted = false;
            bool allow_mend = c->vit_mend, mended = false;
            c->vit_mended = 0;
            for (; round <= 12; ++round) {
                BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                lds_poison(c->stream);
                if (round == 0)
                    hipLaunchKernelGGL(k_gen_viterbi_seg<false>, sgrid, sblk, smv, c->stream, m, (const int64_t *)c->d_offsets.p,
                                       sg, pobs, ptr8, last, c->d_aentry.p, c->d_aexit.p, c->d_vckpt.p,
                                       (const uint8_t *)c->d_vflag.p, vall);
                else
                    hipLaunchKernelGGL(k_gen_viterbi_seg<true>, sgrid, sblk, smv, c->stream, m, (const int64_t *)c->d_offsets.p,
                                       sg, pobs, ptr8, last, c->d_aentry.p, c->d_aexit.p, c->d_vckpt.p,
                                       (const uint8_t *)c->d_vflag.p);
                hipLaunchKernelGGL((k_wide_vit_check<128>), dim3((sg.nseg + 255) / 256), dim3(256), 0, c->stream, sg,
                                   c->d_aentry.p, (const double *)c->d_aexit.p, c->d_vflag.p, c->d_specres.p,
                                   (round == 0 && vall) ? vm_tol : 0.0);
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
                    // the mending round of wide_viterbi_run (path_api.hip): the segments further than vm_tol from their
                    // predecessors' vectors alone run again up to a kept vector of the first pass
                    mended = true;
                    BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                    lds_poison(c->stream);
                    hipLaunchKernelGGL(k_gen_viterbi_seg<true>, sgrid, sblk, smv, c->stream, m, (const int64_t *)c->d_offsets.p,
                                       sg, pobs, ptr8, last, c->d_aentry.p, c->d_aexit.p, c->d_vckpt.p,
                                       (const uint8_t *)c->d_vflag.p + sg.nseg, vall, vm_tol, c->d_specres.p + 1);
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
                    // every boundary (and splice) within vm_tol: the path of this pass, and the margins of the decisions on it
                    const int maxseg = (int)((maxT + seglen - 1) / seglen) + 1 + spliced;
                    const double margin = std::max(1e-10, 16.0 * (2e-15 * (double)maxT + vm_tol * maxseg));
                    if ((rc = seg_walks()))
                        return rc;
                    BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                    const dim3 mgrid(sg.nseg, (unsigned)((c->pplan[0].maxlen + VM_STEPS - 1) / VM_STEPS)); // (the longest REAL segment)
                    if (out_fmt == 0)
                        hipLaunchKernelGGL((k_vit_margin<int32_t, 2>), mgrid, dim3(256), smm, c->stream, m.A, n, off, sg,
                                           (const double *)vall, (const int32_t *)path, margin, c->d_specres.p);
                    else
                        hipLaunchKernelGGL((k_vit_margin<uint8_t, 2>), mgrid, dim3(256), smm, c->stream, m.A, n, off, sg,
                                           (const double *)vall, (const uint8_t *)p8, margin, c->d_specres.p);
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
                    // (the rounds compare bitwise with the kept vectors: a pass that was mended and then not accepted is
                    // run again from scratch, without mending -- see wide_viterbi_run)
                    allow_mend = false;
                    mended = false;
                    round = -1;
                }
            }
            c->vit_seg_rounds = round;
            if (c->h_specres[3] == 0 || margin_accepted) {
                done = true;
                // (what converged is where the next call on these observations starts; longer after a pass whose far
                // boundaries could not be mended or that needed three or more rounds)
                const bool longer = (vall && c->vit_far > 0 && !margin_accepted) || (round >= 3 && !margin_accepted);
                c->vit_W = (longer && W_try < W_cap) ? std::min(2 * W_try, W_cap) : W_try;
            }
        }
        if (!done && c->pplan[0].nseg > K)
            c->vit_seg_given_up = true; // these observations go to the serial kernel from now on
        c->viterbi_chunked = done;
        if (done) {
            if (!c->vit_margin_used && (rc = seg_walks())) // (a margin-accepted pass has its path already)
                return rc;
            if (out_fmt == 0)
                BHMM_HIP(hipMemcpyAsync(paths_out, path, (size_t)c->total * sizeof(int32_t),
                                        hipMemcpyDeviceToHost, c->stream));
            else if (out_fmt == 1)
                BHMM_HIP(hipMemcpyAsync(paths_out, p8, (size_t)c->total, hipMemcpyDeviceToHost, c->stream));
            BHMM_HIP(hipStreamSynchronize(c->stream));
            return BHMM_OK;
        }
    }
    // 129 .. 256 states (round 5): four segments per workgroup share every pass over A (k_gen_viterbi_rows); first
    // pass only -- accepted when every boundary is bit-identical or by the margins of the decisions on its path
    // (k_vit_margin), else the serial kernel below decides
    if (n > 128 && n <= 256 && c->spec_enabled && c->vit_margin && !c->vit_seg_given_up) {
        uint8_t *ptr8 = reinterpret_cast<uint8_t *>(c->d_scratch.p);
        uint8_t *p8 = out_fmt == 2 ? static_cast<uint8_t *>(paths_out) : reinterpret_cast<uint8_t *>(path);
        const int64_t *off = c->d_offsets.p;
        const int64_t want = (int64_t)GVR_ROWS * (c->num_simd / 4); // one workgroup of four segments per compute unit
        const int64_t fill0 = ((c->total + want - 1) / want + 7) / 8 * 8;
        const int W0 = std::max(64, c->spec_W > 0 ? (c->spec_W + 7) / 8 * 8 : 128);
        // (six E-step forgetting lengths, at most one and a half fill lengths: at 256 states one boundary of 749 was
        // still 1e-12 off after 504 steps, none after 750 -- and a pass that is not accepted is lost time)
        // (round 6: with the mending round the E-step's length; a pass that is not accepted doubles it for the next call)
        const int W_try = c->vit_W > 0 ? c->vit_W
                          : (c->vit_mend ? W0
                                         : (int)std::max<int64_t>(W0, std::min<int64_t>(6 * (int64_t)W0, (3 * fill0 / 2 + 7) / 8 * 8)));
        const int64_t seglen = std::max<int64_t>(fill0, W_try);
        Segs sg;
        if ((rc = wide_path_plan_pub(c, 0, seglen, sg)))
            return rc;
        int64_t maxT = 0;
        for (int k = 0; k < K; ++k)
            maxT = std::max(maxT, c->offsets[k + 1] - c->offsets[k]);
        c->vit_margin_used = 0;
        c->vit_margin_close = 0;
        c->vit_seg_rounds = 0;
        // threads per target state (candidate ranges): 2 (BHMM_AMD_GVR_S = 1 / 4: experiments; measured 15.8 / 13.3 /
        // 14.1 ms at 129 states with 1 / 2 / 4)
        static const int gvr_s = getenv("BHMM_AMD_GVR_S") ? atoi(getenv("BHMM_AMD_GVR_S")) : 2;
        const int GVR_S = gvr_s == 1 ? 1 : (gvr_s == 2 ? 2 : 4);
        const size_t smr = (size_t)(2 * GVR_ROWS * n + GVR_ROWS) * sizeof(double) +
                           (size_t)(GVR_S - 1) * GVR_ROWS * 256 * (sizeof(double) + sizeof(int));
        if (sg.nseg > K && (rc = gen_transposed(c, m)) == BHMM_OK && c->d_gW.ensure((size_t)c->total * n) == BHMM_OK &&
            c->d_aentry.ensure((size_t)sg.nseg * 256) == BHMM_OK && c->d_aexit.ensure((size_t)sg.nseg * 256) == BHMM_OK &&
            c->d_vflag.ensure(2 * (size_t)sg.nseg) == BHMM_OK && c->d_specres.ensure(4) == BHMM_OK &&
            c->d_vmaps.ensure((size_t)sg.nseg * 256) == BHMM_OK && c->d_vend.ensure((size_t)sg.nseg) == BHMM_OK) {
            if (!c->h_specres)
                BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_specres), 4 * sizeof(unsigned int),
                                       hipHostMallocDefault));
            sg.W = W_try;
            const double vm_tol = 1e-12;
            double *vall = c->d_gW.p;
            BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
            lds_poison(c->stream);
#define BHMM_GVR(SV)                                                                                                  \
    hipLaunchKernelGGL((k_gen_viterbi_rows<GVR_ROWS, SV>), dim3((sg.nseg + GVR_ROWS - 1) / GVR_ROWS), dim3(256 * SV), smr, \
                       c->stream, m, off, sg, pobs, ptr8, last, c->d_aentry.p, c->d_aexit.p, vall)
            if (GVR_S == 1)
                BHMM_GVR(1);
            else if (GVR_S == 2)
                BHMM_GVR(2);
            else
                BHMM_GVR(4);
#undef BHMM_GVR
            hipLaunchKernelGGL((k_wide_vit_check<256>), dim3((sg.nseg + 255) / 256), dim3(256), 0, c->stream, sg,
                               c->d_aentry.p, (const double *)c->d_aexit.p, c->d_vflag.p, c->d_specres.p, vm_tol);
            // the back-trace of this pass (needed either way)
            auto rows_walks = [&]() -> int {
                hipLaunchKernelGGL((k_wide_vit_walk<false, uint8_t, 4>), dim3(sg.nseg), dim3(64), 0, c->stream, off, sg, n,
                                   (const uint8_t *)ptr8, c->d_vmaps.p, (const uint8_t *)nullptr, (uint8_t *)nullptr);
                hipLaunchKernelGGL(k_wide_vit_stitch, dim3((K + 63) / 64), dim3(64), 0, c->stream,
                                   (const int32_t *)c->pplan[0].traj0.p, K, (const uint8_t *)c->d_vmaps.p, 256,
                                   (const int32_t *)last, c->d_vend.p);
                if (out_fmt == 0)
                    hipLaunchKernelGGL((k_wide_vit_walk<true, int32_t, 4>), dim3(sg.nseg), dim3(64), 0, c->stream, off, sg, n,
                                       (const uint8_t *)ptr8, (uint8_t *)nullptr, (const uint8_t *)c->d_vend.p, path);
                else
                    hipLaunchKernelGGL((k_wide_vit_walk<true, uint8_t, 4>), dim3(sg.nseg), dim3(64), 0, c->stream, off, sg, n,
                                       (const uint8_t *)ptr8, (uint8_t *)nullptr, (const uint8_t *)c->d_vend.p, p8);
                BHMM_HIP(hipGetLastError());
                return BHMM_OK;
            };
            if ((rc = rows_walks()))
                return rc;
            BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost,
                                    c->stream));
            BHMM_HIP(hipStreamSynchronize(c->stream));
            c->vit_seg_mismatch = (int)c->h_specres[3];
            c->vit_far = (int)c->h_specres[0];
            bool accepted = c->h_specres[3] == 0;
            int spliced = 0;
            if (!accepted && c->h_specres[0] != 0 && (int64_t)c->h_specres[0] * 2 <= sg.nseg && c->vit_mend) {
                // the mending round (see wide_viterbi_run): the rows of the segments further than vm_tol from their
                // predecessors' vectors run again up to a vector the first pass kept; then the back-trace again
                BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                lds_poison(c->stream);
#define BHMM_GVR_MEND(SV)                                                                                                    \
    hipLaunchKernelGGL((k_gen_viterbi_rows<GVR_ROWS, SV, true>), dim3((sg.nseg + GVR_ROWS - 1) / GVR_ROWS), dim3(256 * SV), smr, \
                       c->stream, m, off, sg, pobs, ptr8, last, c->d_aentry.p, c->d_aexit.p, vall,                           \
                       (const uint8_t *)c->d_vflag.p + sg.nseg, vm_tol, c->d_specres.p + 1)
                if (GVR_S == 1)
                    BHMM_GVR_MEND(1);
                else if (GVR_S == 2)
                    BHMM_GVR_MEND(2);
                else
                    BHMM_GVR_MEND(4);
#undef BHMM_GVR_MEND
                BHMM_HIP(hipGetLastError());
                unsigned int notmet = 0;
                BHMM_HIP(hipMemcpyAsync(&notmet, c->d_specres.p + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
                BHMM_HIP(hipStreamSynchronize(c->stream));
                c->vit_mended = (int)c->h_specres[0];
                if (notmet == 0) {
                    spliced = (int)c->h_specres[0];
                    c->h_specres[0] = 0;
                    if ((rc = rows_walks()))
                        return rc;
                }
            }
            if (!accepted && c->h_specres[0] == 0) {
                const int maxseg = (int)((maxT + seglen - 1) / seglen) + 1 + spliced;
                const double margin = std::max(1e-10, 16.0 * (2e-15 * (double)maxT + vm_tol * maxseg));
                BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                const dim3 mgrid(sg.nseg, (unsigned)((c->pplan[0].maxlen + VM_STEPS - 1) / VM_STEPS)); // (the longest REAL segment)
                if (out_fmt == 0)
                    hipLaunchKernelGGL((k_vit_margin<int32_t, 4, false>), mgrid, dim3(256), 0, c->stream,
                                       (const double *)c->d_gAt.p, n, off, sg, (const double *)vall, (const int32_t *)path,
                                       margin, c->d_specres.p);
                else
                    hipLaunchKernelGGL((k_vit_margin<uint8_t, 4, false>), mgrid, dim3(256), 0, c->stream,
                                       (const double *)c->d_gAt.p, n, off, sg, (const double *)vall, (const uint8_t *)p8,
                                       margin, c->d_specres.p);
                BHMM_HIP(hipGetLastError());
                BHMM_HIP(hipMemcpyAsync(c->h_specres, c->d_specres.p, 4 * sizeof(unsigned int), hipMemcpyDeviceToHost,
                                        c->stream));
                BHMM_HIP(hipStreamSynchronize(c->stream));
                c->vit_margin_close = (int)c->h_specres[2];
                accepted = c->h_specres[2] == 0;
                c->vit_margin_used = accepted ? 1 : 0;
            }
            if (accepted) {
                c->viterbi_chunked = true;
                c->vit_W = W_try;
                c->vit_rows_fail = 0;
                if (out_fmt == 0)
                    BHMM_HIP(hipMemcpyAsync(paths_out, path, (size_t)c->total * sizeof(int32_t), hipMemcpyDeviceToHost,
                                            c->stream));
                else if (out_fmt == 1)
                    BHMM_HIP(hipMemcpyAsync(paths_out, p8, (size_t)c->total, hipMemcpyDeviceToHost, c->stream));
                BHMM_HIP(hipStreamSynchronize(c->stream));
                return BHMM_OK;
            }
            // One pass that was not accepted (a close decision under THIS model -- an early EM iterate, say) sends
            // this call to the serial kernel; the next call tries again with twice the warm-up, and only a second
            // failure in a row (or a warm-up that would exceed half a trajectory) gives the observations up.
            if (++c->vit_rows_fail >= 2 || 4 * (int64_t)W_try > maxT)
                c->vit_seg_given_up = true;
            else
                c->vit_W = 2 * W_try;
        } else {
            (void)hipGetLastError();
        }
    }
    size_t sm = gen_smem(n, 2, 2);
    if (gen_a_in_lds(n, sm)) {
        sm += gen_a_bytes(n);
        if ((rc = gen_set_smem(k_gen_viterbi_fwd<true>, sm)))
            return rc;
        hipLaunchKernelGGL(k_gen_viterbi_fwd<true>, dim3(K), dim3(GEN_TPB), sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, K, pobs, ptr, last);
    } else {
        if ((rc = gen_set_smem(k_gen_viterbi_fwd<false>, sm)))
            return rc;
        hipLaunchKernelGGL(k_gen_viterbi_fwd<false>, dim3(K), dim3(GEN_TPB), sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, K, pobs, ptr, last);
    }
    BHMM_HIP(hipGetLastError());
    const dim3 tg((K + 63) / 64), tb(64);
    if (out_fmt == 0) {
        hipLaunchKernelGGL(k_gen_viterbi_trace<int32_t>, tg, tb, 0, c->stream,
                           (const int64_t *)c->d_offsets.p, K, n, (const uint16_t *)ptr,
                           (const int32_t *)last, path);
        BHMM_HIP(hipGetLastError());
        BHMM_HIP(hipMemcpyAsync(paths_out, path, (size_t)c->total * sizeof(int32_t),
                                hipMemcpyDeviceToHost, c->stream));
    } else {
        uint8_t *p8 = out_fmt == 2 ? static_cast<uint8_t *>(paths_out) : reinterpret_cast<uint8_t *>(path);
        hipLaunchKernelGGL(k_gen_viterbi_trace<uint8_t>, tg, tb, 0, c->stream,
                           (const int64_t *)c->d_offsets.p, K, n, (const uint16_t *)ptr,
                           (const int32_t *)last, p8);
        BHMM_HIP(hipGetLastError());
        if (out_fmt == 1)
            BHMM_HIP(hipMemcpyAsync(paths_out, p8, (size_t)c->total, hipMemcpyDeviceToHost, c->stream));
    }
    BHMM_HIP(hipStreamSynchronize(c->stream));
    return BHMM_OK;
}

// alpha_dev: rows to sample from (the context's own forward pass when NULL)
int gen_sample_run(bhmm_ctx *c, const double *A, const double *pi, const double *par0,
                   const double *par1, const double *u, uint64_t seed, int32_t *paths,
                   int64_t *counts, int64_t *n0, double *emis, double *stats_dev)
{
    WideModel m;
    int rc = wide_model_pub(c, c->kind, A, pi, par0, par1, m);
    if (rc)
        return rc;
    // alpha rows in d_alpha_rm: from the tile forward pass over time segments where it verifies
    // (65..128 states; the draw normalises alpha_t[i] A[i][s_{t+1}] itself, any factor per row cancels),
    // else from the serial recursion
    // (the repeat of a call in which a watched draw did not stand, draw_verify.hpp: the serial recursion)
    bool fwd_seg = false;
    if (!c->draw_force_exact && (rc = tile_gen_forward_draw(c, m, &fwd_seg)))
        return rc;
    c->draw_fwd_segmented = fwd_seg;
    if (!fwd_seg) {
        c->draw_alpha_dev = 0.0;
        if ((rc = gen_forward(c, A, pi, par0, par1)))
            return rc;
    }
    // rows of a segmented pass: draws within 64 x the deviation its boundary check measured are recorded
    DrawWatch watch;
    if ((rc = draw_watch_prepare(c, fwd_seg ? 64.0 * std::max(c->draw_alpha_dev, 1e-16) : 0.0, watch, nullptr)))
        return rc;
    const int n = c->n, K = c->K;
    const size_t nstat = (size_t)n * n + n;
    const size_t esz = c->kind == EMIT_GAUSS ? 3 * (size_t)n : (c->kind == EMIT_DISC ? (size_t)n * c->M : 0);
    const size_t nsym = c->kind == EMIT_DISC ? (size_t)n * c->M : 0;
    if ((rc = c->d_scratch2.ensure(((size_t)c->total + 4) * sizeof(int32_t))) ||
        (rc = c->d_scratch.ensure((nstat + nsym + (size_t)std::max(K, 1) * GEN_PS_SLABS * 3 * n + nstat + esz +
                                   (u ? (size_t)c->total : 0) + 8) * sizeof(double))))
        return rc;
    int32_t *path = reinterpret_cast<int32_t *>(c->d_scratch2.p);
    int *status = reinterpret_cast<int *>(path + c->total);
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(c->d_scratch.p);
    unsigned long long *symcnt = cnt + nstat;
    double *epart = reinterpret_cast<double *>(symcnt + nsym);
    double *packed = epart + (size_t)std::max(K, 1) * GEN_PS_SLABS * 3 * n;
    double *udev = nullptr;
    if (u) {
        udev = packed + nstat + esz;
        BHMM_HIP(hipMemcpyAsync(udev, u, (size_t)c->total * sizeof(double), hipMemcpyHostToDevice,
                                c->stream));
    }
    BHMM_HIP(hipMemsetAsync(cnt, 0, (nstat + nsym) * sizeof(unsigned long long), c->stream));
    BHMM_HIP(hipMemsetAsync(status, 0, sizeof(int), c->stream));
    // up to 512 states: the draw over time segments (k_gen_sample_seg), coupled through the per-step
    // uniforms; segments that did not continue their successor's state are drawn again until none is left
    c->smp_segmented = false;
    bool seg_done = false;
    if (c->spec_enabled && n <= 512) {
        const int64_t want = (int64_t)c->smp_seg_per_simd * c->num_simd;
        const int64_t seglen = std::max<int64_t>((c->total + want - 1) / want, 64);
        Segs sg;
        if ((rc = wide_path_plan_pub(c, 1, seglen, sg)))
            return rc;
        if (sg.nseg > K) {
            if (c->smp_W <= 0)
                c->smp_W = 64;
            sg.W = c->smp_W;
            if ((rc = gen_transposed(c, m)) || (rc = c->d_sentry.ensure((size_t)sg.nseg)) ||
                (rc = c->d_sexit.ensure((size_t)sg.nseg)) || (rc = c->d_vflag.ensure((size_t)sg.nseg)) ||
                (rc = c->d_specres.ensure(4)))
                return rc;
            if (!c->h_specres)
                BHMM_HIP(hipHostMalloc(reinterpret_cast<void **>(&c->h_specres), 4 * sizeof(unsigned int),
                                       hipHostMallocDefault));
            const dim3 sgrid((sg.nseg + 3) / 4), sblk(256);
#define BHMM_GSS(SPLV, FIXV)                                                                            \
    hipLaunchKernelGGL((k_gen_sample_seg<SPLV, FIXV>), sgrid, sblk, 0, c->stream, m,                    \
                       (const double *)c->d_gAt.p, (const int64_t *)c->d_offsets.p, sg,                 \
                       (const double *)c->d_alpha_rm.p, (const double *)udev, seed,                     \
                       (const int64_t *)c->d_soff.p, path, status, c->d_sentry.p, c->d_sexit.p,         \
                       (const uint8_t *)c->d_vflag.p, watch)
#define BHMM_GSS_SPL(FIXV)       \
    do {                         \
        if (n <= 128)            \
            BHMM_GSS(2, FIXV);   \
        else if (n <= 256)       \
            BHMM_GSS(4, FIXV);   \
        else                     \
            BHMM_GSS(8, FIXV);   \
    } while (0)
            const int max_rounds = 16;
            int round = 0;
            for (; round <= max_rounds; ++round) {
                BHMM_HIP(hipMemsetAsync(c->d_specres.p, 0, 4 * sizeof(unsigned int), c->stream));
                lds_poison(c->stream);
                if (round == 0)
                    BHMM_GSS_SPL(false);
                else
                    BHMM_GSS_SPL(true);
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
#undef BHMM_GSS_SPL
#undef BHMM_GSS
            c->smp_seg_rounds = round;
            // (a draw that found no state may belong to a segment that was drawn again afterwards:
            // the serial kernel decides such a call)
            seg_done = c->h_specres[3] == 0 && c->h_specres[0] == 0;
            if (!seg_done)
                BHMM_HIP(hipMemsetAsync(status, 0, sizeof(int), c->stream));
            if ((int64_t)c->smp_seg_mismatch * 10 > sg.nseg && c->smp_W < 4096)
                c->smp_W *= 2;
            c->smp_segmented = seg_done;
        }
    }
    size_t sm = gen_smem(n, 2, 4);
    if (!seg_done && fwd_seg) { // (the serial draw has no watch: it reads rows of the serial recursion)
        if ((rc = gen_forward(c, A, pi, par0, par1)))
            return rc;
        c->draw_fwd_segmented = false;
        c->draw_alpha_dev = 0.0;
    }
    if (seg_done && watch.count) {
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
                rc = gen_sample_run(c, A, pi, par0, par1, u, seed, paths, counts, n0, emis, stats_dev);
                c->draw_force_exact = false;
                c->draw_events = ev;
                c->draw_checked = ck;
                c->draw_redone = 1;
                return rc;
            }
        }
    }
    if (seg_done) {
        ;
    } else if (gen_a_in_lds(n, sm)) {
        sm += gen_a_bytes(n);
        if ((rc = gen_set_smem(k_gen_sample<true>, sm)))
            return rc;
        hipLaunchKernelGGL(k_gen_sample<true>, dim3(K), dim3(GEN_TPB), sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, K, (const double *)c->d_alpha_rm.p,
                           (const double *)udev, seed, (const int64_t *)c->d_soff.p, path, status);
    } else {
        if ((rc = gen_set_smem(k_gen_sample<false>, sm)))
            return rc;
        hipLaunchKernelGGL(k_gen_sample<false>, dim3(K), dim3(GEN_TPB), sm, c->stream, m,
                           (const int64_t *)c->d_offsets.p, K, (const double *)c->d_alpha_rm.p,
                           (const double *)udev, seed, (const int64_t *)c->d_soff.p, path, status);
    }
    BHMM_HIP(hipGetLastError());
    int hstatus = 0;
    BHMM_HIP(hipMemcpyAsync(&hstatus, status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    BHMM_HIP(hipStreamSynchronize(c->stream));
    if (hstatus) {
        set_error("random choice found no state: alpha/A not normalisable (_hidden.c:299-304)");
        return hstatus;
    }
    const void *obs = c->d_obs_rm.p;
    const dim3 pg(256), pb(256);
#define BHMM_GEN_PS(KINDV)                                                                          \
    do {                                                                                            \
        hipLaunchKernelGGL((k_gen_path_stats<KINDV>), dim3(K, GEN_PS_SLABS), dim3(GEN_TPB), 0, c->stream, m,      \
                           (const int64_t *)c->d_offsets.p, K, obs, (const int32_t *)path, cnt,     \
                           epart, symcnt);                                                          \
        hipLaunchKernelGGL((k_gen_pack_path_stats<KINDV>), pg, pb, 0, c->stream, m, K,              \
                           (const unsigned long long *)cnt, (const double *)epart,                  \
                           (const unsigned long long *)symcnt, stats_dev ? stats_dev : packed);     \
    } while (0)
    if (c->kind == EMIT_GAUSS)
        BHMM_GEN_PS(EMIT_GAUSS);
    else if (c->kind == EMIT_DISC)
        BHMM_GEN_PS(EMIT_DISC);
    else
        BHMM_GEN_PS(EMIT_EXPL);
#undef BHMM_GEN_PS
    BHMM_HIP(hipGetLastError());
    std::vector<double> hp;
    if (!stats_dev && (counts || n0 || emis)) {
        hp.resize(nstat + esz);
        BHMM_HIP(hipMemcpyAsync(hp.data(), packed, hp.size() * sizeof(double), hipMemcpyDeviceToHost,
                                c->stream));
    }
    if (paths)
        BHMM_HIP(hipMemcpyAsync(paths, path, (size_t)c->total * sizeof(int32_t), hipMemcpyDeviceToHost,
                                c->stream));
    BHMM_HIP(hipStreamSynchronize(c->stream));
    if (!hp.empty()) {
        if (counts)
            for (size_t e = 0; e < (size_t)n * n; ++e)
                counts[e] = (int64_t)llround(hp[e]);
        if (n0)
            for (int i = 0; i < n; ++i)
                n0[i] = (int64_t)llround(hp[(size_t)n * n + i]);
        if (emis && esz)
            memcpy(emis, hp.data() + nstat, esz * sizeof(double));
    }
    return BHMM_OK;
}

// single-trajectory entry points with host arrays and more than 64 states ---------------------------
int gen_transition_counts(double *C, const double *A, const double *pobs, const double *alpha,
                          const double *beta, int N, int64_t T)
{
    if (N > GEN_MAXN)
        return invalid_arg("more than 4096 hidden states are not supported");
    double *d = nullptr;
    const size_t rows = (size_t)T * N, nn = (size_t)N * N;
    const int tiles = (N + 31) / 32;
    const int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>({4096 / ((int64_t)tiles * tiles) + 1,
                                                                    (T + 63) / 64, (int64_t)256}));
    const size_t need = 4 * rows + 2 * nn + (size_t)nsplit * nn + nn;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), need * sizeof(double));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc failed");
        return BHMM_ERR_NO_MEM;
    }
    struct Free {
        double *p;
        ~Free() { (void)hipFree(p); }
    } guard{d};
    double *dp = d, *da = dp + rows, *db = da + rows, *dW = db + rows, *dA = dW + rows, *dAt = dA + nn,
           *dpart = dAt + nn, *dC = dpart + (size_t)nsplit * nn;
    BHMM_HIP(hipMemcpy(dp, pobs, rows * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(da, alpha, rows * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(db, beta, rows * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(dA, A, nn * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gen_transpose, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, 0,
                       (const double *)dA, N, dAt);
    const size_t sm = ((size_t)N + GEN_TPB) * sizeof(double);
    hipLaunchKernelGGL(k_gen_w_rows, dim3((unsigned)T), dim3(GEN_TPB), sm, 0, (const double *)dAt, N, T,
                       (const double *)dp, (const double *)da, (const double *)db, dW);
    hipLaunchKernelGGL(k_gen_xi_gemm, dim3(tiles * tiles, nsplit), dim3(256), 0, 0, (const double *)da,
                       (const double *)dW, T, N, nsplit, dpart);
    BHMM_HIP(hipGetLastError());
    std::vector<double> hpart((size_t)nsplit * nn);
    BHMM_HIP(hipMemcpy(hpart.data(), dpart, hpart.size() * sizeof(double), hipMemcpyDeviceToHost));
    (void)dC;
    for (size_t ij = 0; ij < nn; ++ij) {
        double v = 0.0;
        for (int s = 0; s < nsplit; ++s)
            v += hpart[(size_t)s * nn + ij];
        C[ij] = v * A[ij];
    }
    return BHMM_OK;
}

int gen_sample_path(int32_t *path, const double *alpha, const double *A, const double *u, int N,
                    int64_t T)
{
    if (N > GEN_MAXN)
        return invalid_arg("more than 4096 hidden states are not supported");
    double *d = nullptr;
    const size_t rows = (size_t)T * N, nn = (size_t)N * N;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d), (rows + nn + (size_t)T + 8) * sizeof(double) +
                                                                ((size_t)T + 4) * sizeof(int32_t));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipMalloc failed");
        return BHMM_ERR_NO_MEM;
    }
    struct Free {
        double *p;
        ~Free() { (void)hipFree(p); }
    } guard{d};
    double *da = d, *dA = da + rows, *du = dA + nn;
    int64_t *doff = reinterpret_cast<int64_t *>(du + T);
    int32_t *dpath = reinterpret_cast<int32_t *>(doff + 2);
    int *status = reinterpret_cast<int *>(dpath + T);
    const int64_t off[2] = {0, T};
    BHMM_HIP(hipMemcpy(da, alpha, rows * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(dA, A, nn * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(du, u, (size_t)T * sizeof(double), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemcpy(doff, off, sizeof(off), hipMemcpyHostToDevice));
    BHMM_HIP(hipMemset(status, 0, sizeof(int)));
    WideModel m;
    memset(&m, 0, sizeof(m));
    m.A = dA;
    m.n = N;
    const size_t sm = gen_smem(N, 2, 4);
    int rc = gen_set_smem(k_gen_sample<false>, sm);
    if (rc)
        return rc;
    hipLaunchKernelGGL(k_gen_sample<false>, dim3(1), dim3(GEN_TPB), sm, 0, m, (const int64_t *)doff, 1,
                       (const double *)da, (const double *)du, (uint64_t)0, (const int64_t *)nullptr,
                       dpath, status);
    BHMM_HIP(hipGetLastError());
    int hstatus = 0;
    BHMM_HIP(hipMemcpy(&hstatus, status, sizeof(int), hipMemcpyDeviceToHost));
    if (hstatus) {
        set_error("random choice found no state: alpha/A not normalisable (_hidden.c:299-304)");
        return hstatus;
    }
    BHMM_HIP(hipMemcpy(path, dpath, (size_t)T * sizeof(int32_t), hipMemcpyDeviceToHost));
    return BHMM_OK;
}

} // namespace bhmm
