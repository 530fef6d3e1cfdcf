// tile_gen_launch.hpp -- launches of the row-batched matrix-core kernels for NT = 5 .. 8 column tiles (65 .. 128
// states).  The templates are instantiated in tile_gen_nt.hip, once per NT (one translation unit each: the
// twelve kernel pairs take eight minutes in one), and only declared for tile_gen.hip.
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "host_common.hpp"
#include "plan.hpp"
#include "wide_kernels.hpp"

namespace bhmm {
Segs wide_segs_pub(bhmm_ctx *c, int which);

template <int NT, int KIND>
int tile_gen_launch_fwd(bhmm_ctx *c, const WideModel &m);
template <int NT, int KIND>
int tile_gen_launch_bwd(bhmm_ctx *c, const WideModel &m, double *gam, double *stats_dev);

#define TILE_GEN_LAUNCH_DECL(X, NTV)                                                                              \
    X template int tile_gen_launch_fwd<NTV, EMIT_GAUSS>(bhmm_ctx *, const WideModel &);                           \
    X template int tile_gen_launch_fwd<NTV, EMIT_DISC>(bhmm_ctx *, const WideModel &);                            \
    X template int tile_gen_launch_fwd<NTV, EMIT_EXPL>(bhmm_ctx *, const WideModel &);                            \
    X template int tile_gen_launch_bwd<NTV, EMIT_GAUSS>(bhmm_ctx *, const WideModel &, double *, double *);       \
    X template int tile_gen_launch_bwd<NTV, EMIT_DISC>(bhmm_ctx *, const WideModel &, double *, double *);        \
    X template int tile_gen_launch_bwd<NTV, EMIT_EXPL>(bhmm_ctx *, const WideModel &, double *, double *);
} // namespace bhmm
