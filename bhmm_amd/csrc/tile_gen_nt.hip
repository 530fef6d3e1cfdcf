// tile_gen_nt.hip -- the kernels of tile_kernels.hpp for ONE column-tile count (-DTILE_GEN_NT_VALUE=5 .. 8;
// 65 .. 128 states) and their launches: forward pass + log-likelihoods, backward pass + W rows + the xi GEMM +
// finalize.  See tile_gen.hip for the driver (plans, boundary checks, fallbacks).
#include "tile_gen_launch.hpp"
#include "gen_kernels.hpp"
#include "tile_kernels.hpp"
#include "big_kernels.hpp"

namespace bhmm {

template <int NT, int KIND>
int tile_gen_launch_fwd(bhmm_ctx *c, const WideModel &m)
{
    lds_poison(c->stream);
    const Segs sg = wide_segs_pub(c, 1);
    const TilePlan tp{c->d_tile_seg[1].p, c->w_ntiles[1]};
    // up to 96 states the forward kernel fits the eight-wavefront form (matrix + stream wavefronts, 225
    // registers); the backward kernel does not (it would spill 440 registers), nor does either at 128
    constexpr bool FWD_SPLIT = true;
    static const bool probe_on = getenv("BHMM_AMD_TILE_PROBE") != nullptr && atoi(getenv("BHMM_AMD_TILE_PROBE")) == 1;
    unsigned long long *probe = nullptr;
    if (probe_on) {
        int rc = c->d_probe.ensure(4096);
        if (rc)
            return rc;
        probe = reinterpret_cast<unsigned long long *>(c->d_probe.p);
        BHMM_HIP(hipMemsetAsync(probe, 0, 64, c->stream));
    }
    hipLaunchKernelGGL((k_tile_fwd<NT, KIND, false, FWD_SPLIT>), dim3(tp.ntiles), dim3(tile_threads<FWD_SPLIT>()), 0,
                       c->stream, m, (const int64_t *)c->d_offsets.p, sg, tp, (const void *)c->d_obs_rm.p,
                       c->d_alpha_rm.p, c->d_wexp.p, c->d_wePseg.p, c->d_waentry.p, c->d_waexit.p,
                       c->d_specres.p, probe);
    BHMM_HIP(hipGetLastError());
    if (probe_on) {
        unsigned long long h[8];
        BHMM_HIP(hipMemcpyAsync(h, probe, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        BHMM_HIP(hipStreamSynchronize(c->stream));
        if (h[3] && h[7])
            fprintf(stderr, "tile fwd<%d> probe: matrix [operands+matrix %.0f | emission row, write %.0f | barrier %.0f] "
                            "stream [store, loads %.0f | emission %.0f | barrier %.0f] cycles/step (%llu steps)\n", NT,
                    (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], (double)h[4] / h[7],
                    (double)h[5] / h[7], (double)h[6] / h[7], h[3]);
    }
    hipLaunchKernelGGL(k_tile_logl, dim3((sg.nseg + 15) / 16), dim3(256), 0, c->stream, sg, c->n,
                       (const double *)c->d_waentry.p, (const double *)c->d_waexit.p,
                       (const int32_t *)c->d_wePseg.p, c->d_wlogLseg.p, c->d_specres.p);
    hipLaunchKernelGGL(k_logl, dim3(c->K), dim3(64), 0, c->stream, (const int32_t *)c->d_wseg_traj0[1].p,
                       c->K, (const double *)c->d_wlogLseg.p, c->d_logLk.p);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

template <int NT, int KIND>
int tile_gen_launch_bwd(bhmm_ctx *c, const WideModel &m, double *gam, double *stats_dev)
{
    lds_poison(c->stream);
    const Segs sg = wide_segs_pub(c, 1);
    const TilePlan tp{c->d_tile_segb[1].p, c->w_ntilesb[1]};
    const int n = c->n;
    // time slabs of the xi GEMM: one workgroup of NT wavefronts each; four per compute unit hide the loads
    const int nsplit = (int)std::max<int64_t>(1, std::min<int64_t>((c->total + 63) / 64, (int64_t)c->num_simd));
    int rc;
    if ((rc = c->d_gW.ensure((size_t)c->total * n)) || (rc = c->d_gxipart.ensure((size_t)nsplit * n * n)))
        return rc;
    hipLaunchKernelGGL(k_wide_zero_last_rows, dim3(c->K), dim3(64), 0, c->stream, (const int64_t *)c->d_offsets.p,
                       c->K, n, c->d_gW.p);
    // BHMM_AMD_TILE_PROBE=1: cycles of the phases of a step (last workgroup, wavefront 0), printed after the pass
    static const bool probe_on = getenv("BHMM_AMD_TILE_PROBE") != nullptr && atoi(getenv("BHMM_AMD_TILE_PROBE")) == 1;
    unsigned long long *probe = nullptr;
    if (probe_on) {
        if ((rc = c->d_probe.ensure(4096)))
            return rc;
        probe = reinterpret_cast<unsigned long long *>(c->d_probe.p) + 16;
        BHMM_HIP(hipMemsetAsync(probe, 0, 48 * 8, c->stream));
    }
    hipLaunchKernelGGL((k_tile_bwd<NT, KIND, false, true, false>), dim3(tp.ntiles), dim3(tile_threads<false>()), 0,
                       c->stream, m, (const int64_t *)c->d_offsets.p, sg, tp, (const void *)c->d_obs_rm.p,
                       (const double *)c->d_alpha_rm.p, (const int32_t *)c->d_wexp.p, gam, c->d_gamma0.p,
                       c->d_partials.p, c->d_dpartials.p, c->d_wbexit.p, c->d_wbentry.p, c->d_specres.p,
                       c->d_gW.p, probe);
    BHMM_HIP(hipGetLastError());
    if (probe_on) {
        unsigned long long h[48];
        BHMM_HIP(hipMemcpyAsync(h, probe, sizeof(h), hipMemcpyDeviceToHost, c->stream));
        BHMM_HIP(hipStreamSynchronize(c->stream));
#ifdef TILE_X_PROBE_REGS
        for (int o = 0; o < 2; ++o)
            if (h[32 + 8 * o + 4])
                fprintf(stderr, "tile bwd<%d> phases (%s steps): matrix part %.0f | W rows %.0f | stream part %.0f | barrier %.0f "
                                "cycles/step (%llu steps)\n", NT, o ? "main / general" : "warm-up",
                        (double)h[32 + 8 * o] / h[32 + 8 * o + 4], (double)h[32 + 8 * o + 1] / h[32 + 8 * o + 4],
                        (double)h[32 + 8 * o + 2] / h[32 + 8 * o + 4], (double)h[32 + 8 * o + 3] / h[32 + 8 * o + 4],
                        h[32 + 8 * o + 4]);
#endif
        for (int o = 0; o < 16; o += 8)
            if (h[o + 4])
                fprintf(stderr, "tile bwd<%d> probe (%s steps): operands+matrix %.0f | rescale, x' write %.0f | W rows, statistics %.0f | "
                                "stream part + barrier %.0f cycles/step (%llu steps)\n", NT, o ? "main" : "warm-up",
                        (double)h[o] / h[o + 4], (double)h[o + 1] / h[o + 4], (double)h[o + 2] / h[o + 4],
                        (double)h[o + 3] / h[o + 4], h[o + 4]);
    }
    // xi counts: C' = alpha^T W over all time steps.  BHMM_AMD_XI_ROWS=1: round 4's kernel (one workgroup per
    // time slab computes the whole n x n block, NT wavefronts); default: k_big_xi_gemm (128 x 128 blocks of
    // 4 x 4 matrix tiles per wavefront, operands three K steps ahead)
    // (up to 80 states the 128 x 128 blocks are mostly padding: 4.80 against 4.69 ms at 65 states, 4.23 against 4.63
    // at 128 -- profiles/r05)
    static const bool xi_rows_env = getenv("BHMM_AMD_XI_ROWS") != nullptr;
    const bool xi_rows = xi_rows_env;
    const int nsl = xi_rows ? nsplit : std::min(nsplit, 2 * c->num_simd / 4); // (two workgroups per compute unit)
    if (!xi_rows) {
        if (n <= 96) // (one 96 x 96 block, 3 x 3 tiles per wavefront)
            hipLaunchKernelGGL(k_big_xi_gemm<3>, dim3(nsl), dim3(256), 0, c->stream, (const double *)c->d_alpha_rm.p,
                               (const double *)c->d_gW.p, c->total, n, 1, nsl, c->d_gxipart.p);
        else
            hipLaunchKernelGGL(k_big_xi_gemm<4>, dim3(nsl), dim3(256), 0, c->stream, (const double *)c->d_alpha_rm.p,
                               (const double *)c->d_gW.p, c->total, n, 1, nsl, c->d_gxipart.p);
    } else
    switch ((n + 15) / 16) {
    case 5:
        hipLaunchKernelGGL((k_gen_xi_gemm_rows<5>), dim3(nsplit), dim3(320), 0, c->stream, (const double *)c->d_alpha_rm.p,
                           (const double *)c->d_gW.p, c->total, n, nsplit, c->d_gxipart.p);
        break;
    case 6:
        hipLaunchKernelGGL((k_gen_xi_gemm_rows<6>), dim3(nsplit), dim3(384), 0, c->stream, (const double *)c->d_alpha_rm.p,
                           (const double *)c->d_gW.p, c->total, n, nsplit, c->d_gxipart.p);
        break;
    case 7:
        hipLaunchKernelGGL((k_gen_xi_gemm_rows<7>), dim3(nsplit), dim3(448), 0, c->stream, (const double *)c->d_alpha_rm.p,
                           (const double *)c->d_gW.p, c->total, n, nsplit, c->d_gxipart.p);
        break;
    default:
        hipLaunchKernelGGL((k_gen_xi_gemm_rows<8>), dim3(nsplit), dim3(512), 0, c->stream, (const double *)c->d_alpha_rm.p,
                           (const double *)c->d_gW.p, c->total, n, nsplit, c->d_gxipart.p);
    }
    const int64_t nfin = (int64_t)n * n + n + (KIND == EMIT_GAUSS ? 2 * n : 0) +
                         (KIND == EMIT_DISC ? (int64_t)n * c->M : 0) + n + 1;
    hipLaunchKernelGGL((k_tile_finalize_xig<KIND>), dim3((unsigned)nfin), dim3(64), 0, c->stream, m, c->K, tp.ntiles,
                       nsl, (const double *)c->d_gxipart.p, (const double *)c->d_partials.p,
                       (const double *)c->d_dpartials.p, (const double *)c->d_logLk.p,
                       (const double *)c->d_gamma0.p, stats_dev);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}


TILE_GEN_LAUNCH_DECL(, TILE_GEN_NT_VALUE)

} // namespace bhmm
