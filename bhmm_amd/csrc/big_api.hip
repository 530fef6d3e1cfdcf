// big_api.hip -- host side of the row-batched matrix-core E-step for more than 128 (up to 512) hidden
// states (big_kernels.hpp).  The driver -- time segments, warm-up calibration, boundary verification, the
// fallback to the order-faithful any-N kernels -- is tile_gen.hip's; this file launches the two passes.
// Reference: bhmm/hidden/impl_c/_hidden.c:42-63,91-109,148-183.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "host_common.hpp"
#include "gen_kernels.hpp"
#include "big_kernels.hpp"

namespace bhmm {
Segs wide_segs_pub(bhmm_ctx *c, int which);

namespace {
// ---- more than 128 states: big_kernels.hpp ---------------------------------------------------------
template <typename F>
int big_set_smem(F *fn, size_t sm)
{
    if (sm > 64 * 1024)
        BHMM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)sm));
    return BHMM_OK;
}

template <int TPW, int KIND>
int big_fwd_t(bhmm_ctx *c, const WideModel &m)
{
    using G = BigGeo<TPW>;
    lds_poison(c->stream);
    const Segs sg = wide_segs_pub(c, 1);
    const TilePlan tp{c->d_tile_seg[1].p, c->w_ntiles[1]};
    // A in matrix-operand order for both passes (the model of THIS call: the backward pass follows)
    const size_t npk = (size_t)G::NP * G::NP;
    int rc;
    if ((rc = c->d_bigBf.ensure(npk)) || (rc = c->d_bigBb.ensure(npk)) || (rc = big_set_smem(k_big_fwd<TPW, KIND>, G::smem)))
        return rc;
    hipLaunchKernelGGL(k_big_pack, dim3((unsigned)((npk + 255) / 256)), dim3(256), 0, c->stream, m.A, c->n, G::NP,
                       c->d_bigBf.p, c->d_bigBb.p);
    hipLaunchKernelGGL((k_big_fwd<TPW, KIND>), dim3(tp.ntiles), dim3(256), G::smem, c->stream, m,
                       (const double *)c->d_bigBf.p, (const int64_t *)c->d_offsets.p, sg, tp, (const void *)c->d_obs_rm.p,
                       c->d_alpha_rm.p, c->d_wlogLseg.p, c->d_waentry.p, c->d_waexit.p, c->d_specres.p);
    BHMM_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_logl, dim3(c->K), dim3(64), 0, c->stream, (const int32_t *)c->d_wseg_traj0[1].p, c->K,
                       (const double *)c->d_wlogLseg.p, c->d_logLk.p);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

template <int TPW, int KIND>
int big_bwd_t(bhmm_ctx *c, const WideModel &m, double *gam, double *stats_dev)
{
    using G = BigGeo<TPW>;
    lds_poison(c->stream);
    const Segs sg = wide_segs_pub(c, 1);
    const TilePlan tp{c->d_tile_segb[1].p, c->w_ntilesb[1]};
    const int n = c->n;
    // the xi counts: C' = alpha^T W over the rows W this pass stores (k_big_xi_gemm: 128 x 128 blocks of C',
    // time slabs that give every compute unit a workgroup or two)
    // block edge 32 TI (TI x TI tiles per wavefront, TI = 3 .. 5) and nb x nb blocks: the combination with the
    // fewest padded tiles (129 .. 160 states: one 160 x 160 block instead of four 128 x 128 ones)
    int ti = 4, nb = (n + 127) / 128;
    for (int t = 3; t <= 5; ++t) { // (6 x 6 tiles per wavefront: measured slower than 4 x 4 on more blocks)
        const int b = (n + 32 * t - 1) / (32 * t);
        if ((b * t) * (b * t) < (nb * ti) * (nb * ti) || ((b * t) * (b * t) == (nb * ti) * (nb * ti) && b < nb)) {
            ti = t;
            nb = b;
        }
    }
    static const int ti_env = getenv("BHMM_AMD_XI_TI") ? atoi(getenv("BHMM_AMD_XI_TI")) : 0; // (experiments)
    if (ti_env >= 3 && ti_env <= 6) {
        ti = ti_env;
        nb = (n + 32 * ti - 1) / (32 * ti);
    }
    const int nsplit = (int)std::max<int64_t>(
        1, std::min<int64_t>({(int64_t)(2 * c->num_simd / 4) / (nb * nb), (c->total + 255) / 256, (int64_t)256}));
    int rc;
    if ((rc = c->d_gW.ensure((size_t)c->total * n)) || (rc = c->d_gxipart.ensure((size_t)nsplit * n * n)) ||
        (rc = big_set_smem(k_big_bwd<TPW, KIND>, G::smem)))
        return rc;
    hipLaunchKernelGGL(k_wide_zero_last_rows, dim3(c->K), dim3(64), 0, c->stream, (const int64_t *)c->d_offsets.p,
                       c->K, n, c->d_gW.p);
    hipLaunchKernelGGL((k_big_bwd<TPW, KIND>), dim3(tp.ntiles), dim3(256), G::smem, c->stream, m,
                       (const double *)c->d_bigBb.p, (const int64_t *)c->d_offsets.p, sg, tp, (const void *)c->d_obs_rm.p,
                       (const double *)c->d_alpha_rm.p, gam, c->d_gamma0.p, c->d_partials.p, c->d_dpartials.p,
                       c->d_wbexit.p, c->d_wbentry.p, c->d_specres.p, c->d_gW.p);
    BHMM_HIP(hipGetLastError());
#define BIG_XI(TIV)                                                                                              \
    hipLaunchKernelGGL(k_big_xi_gemm<TIV>, dim3(nb * nb * nsplit), dim3(256), 0, c->stream,                        \
                       (const double *)c->d_alpha_rm.p, (const double *)c->d_gW.p, c->total, n, nb, nsplit,        \
                       c->d_gxipart.p)
    if (ti == 3)
        BIG_XI(3);
    else if (ti == 5)
        BIG_XI(5);
    else if (ti == 6)
        BIG_XI(6);
    else
        BIG_XI(4);
#undef BIG_XI
    hipLaunchKernelGGL((k_big_finalize<KIND>), dim3(4096), dim3(64), 0, c->stream, m, c->K, tp.ntiles, nsplit,
                       (const double *)c->d_gxipart.p, (const double *)c->d_partials.p,
                       (const double *)c->d_dpartials.p, (const double *)c->d_logLk.p, (const double *)c->d_gamma0.p,
                       stats_dev);
    BHMM_HIP(hipGetLastError());
    return BHMM_OK;
}

// (column tiles per wavefront: 3, 4, 5, 6, 8 for up to 192, 256, 320, 384, 512 states; seven -- 448 states --
// was measured: its backward kernel needs a ring of seven blocks and spills, 22.9 ms against 20.1 on the
// 512-state kernel at 400 states, 128 x 4000)
#define BIG_TPW(fn, KINDV, ...)                                              \
    (c->n <= 128   ? fn<2, KINDV>(__VA_ARGS__)                               \
     : c->n <= 192 ? fn<3, KINDV>(__VA_ARGS__)                               \
     : c->n <= 256 ? fn<4, KINDV>(__VA_ARGS__)                               \
     : c->n <= 320 ? fn<5, KINDV>(__VA_ARGS__)                               \
     : c->n <= 384 ? fn<6, KINDV>(__VA_ARGS__)                               \
                   : fn<8, KINDV>(__VA_ARGS__))
#define BIG_DISPATCH(fn, ...)                                                \
    (c->kind == EMIT_GAUSS  ? BIG_TPW(fn, EMIT_GAUSS, __VA_ARGS__)           \
     : c->kind == EMIT_DISC ? BIG_TPW(fn, EMIT_DISC, __VA_ARGS__)            \
                            : BIG_TPW(fn, EMIT_EXPL, __VA_ARGS__))

} // namespace

int big_launch_fwd(bhmm_ctx *c, const WideModel &m) { return BIG_DISPATCH(big_fwd_t, c, m); }

int big_launch_bwd(bhmm_ctx *c, const WideModel &m, double *gam, double *stats_dev)
{
    return BIG_DISPATCH(big_bwd_t, c, m, gam, stats_dev);
}

} // namespace bhmm
